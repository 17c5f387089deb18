/* sprof.c - a preloadable sampling profiler for the host side of the drop-in
 * (no perf / gdb on the GPU boxes): a signal to the main thread every 100 us of
 * wall time (a CLOCK_MONOTONIC POSIX timer: ITIMER_PROF only ticks at HZ), the
 * interrupted PCs are histogrammed in 64 byte buckets and printed as
 * [object+offset] at exit; tools/ubench/sprof_resolve.py names them with nm.   gcc -O2 -shared -fPIC -o libsprof.so sprof.c -ldl
 *         LD_PRELOAD=libsprof.so:liba2amd_units.so ref_bench ...             */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>
#include <ucontext.h>

#define MAXS 200000
static void *pcs[MAXS];
static int tids[MAXS];
static __thread int my_tid;
static volatile int npcs;

static void on_prof(int sig, siginfo_t *si, void *uc_)
{
	ucontext_t *uc = (ucontext_t *)uc_;
	int i = __sync_fetch_and_add(&npcs, 1);
	if(i < MAXS)
	{
		pcs[i] = (void *)uc->uc_mcontext.gregs[REG_RIP];
		tids[i] = my_tid;
	}
	else
		npcs = MAXS;
}

static int cmp(const void *a, const void *b) { return strcmp(*(char *const *)a, *(char *const *)b); }

static void dump(void)
{
	static char names[MAXS][96];
	static char *ptr[MAXS];
	int i, n = npcs, run;
	signal(SIGPROF, SIG_IGN);
	{
		/* keep the samples of the thread that spends most time outside libc
		 * (sleeping threads sit in its futex wrappers): the engine's */
		int best = 0, bestn = -1, j, k, seen[64], nseen = 0;
		for(i = 0; i < n; ++i)
		{
			for(k = 0; k < nseen && seen[k] != tids[i]; ++k)
				;
			if(k == nseen && nseen < 64)
				seen[nseen++] = tids[i];
		}
		for(k = 0; k < nseen; ++k)
		{
			int cnt = 0;
			for(j = 0; j < n; j += 7)
			{
				Dl_info di;
				if(tids[j] == seen[k] && dladdr(pcs[j], &di) && di.dli_fname && !strstr(di.dli_fname, "libc.so"))
					++cnt;
			}
			if(cnt > bestn)
			{
				bestn = cnt;
				best = seen[k];
			}
		}
		for(i = j = 0; i < n; ++i)
			if(tids[i] == best)
				pcs[j++] = pcs[i];
		fprintf(stderr, "sprof: thread %d has %d of %d samples\n", best, j, n);
		n = j;
	}
	for(i = 0; i < n; ++i)
	{
		Dl_info di;
		if(dladdr(pcs[i], &di) && di.dli_fname)
		{
			const char *b = strrchr(di.dli_fname, '/');
			snprintf(names[i], 96, "[%s+%lx]", b ? b + 1 : di.dli_fname,
					(unsigned long)((char *)pcs[i] - (char *)di.dli_fbase) & ~0x3fUL);
		}
		else
			snprintf(names[i], 96, "[?]");
		ptr[i] = names[i];
	}
	qsort(ptr, n, sizeof(char *), cmp);
	fprintf(stderr, "sprof: %d samples\n", n);
	for(i = 0; i < n; i += run)
	{
		for(run = 1; i + run < n && !strcmp(ptr[i], ptr[i + run]); ++run)
			;
		if(run * 1000 >= n)	/* >= 0.1 % */
			fprintf(stderr, "sprof %6.2f%%  %s\n", run * 100.0 / n, ptr[i]);
	}
}

static void arm(void)
{
	struct sigevent ev;
	struct itimerspec it = { { 0, 100000 }, { 0, 100000 } };
	timer_t t;
	int ms = getenv("SPROF_DELAY_MS") ? atoi(getenv("SPROF_DELAY_MS")) : 0;
	if(ms)	/* (let the HIP runtime come up first: its start-up does not survive EINTR) */
	{
		it.it_value.tv_sec = ms / 1000;
		it.it_value.tv_nsec = ms % 1000 * 1000000 + 1;
	}
	my_tid = (int)syscall(SYS_gettid);
	memset(&ev, 0, sizeof(ev));
	ev.sigev_notify = SIGEV_THREAD_ID;
	ev.sigev_signo = SIGPROF;
	ev._sigev_un._tid = my_tid;
	timer_create(CLOCK_MONOTONIC, &ev, &t);	/* (CPU-time clocks only tick at HZ) */
	timer_settime(t, 0, &it, NULL);
}

struct start { void *(*fn)(void *); void *arg; };
static void *trampoline(void *p)
{
	struct start st = *(struct start *)p;
	static int created;
	free(p);
	if(__sync_fetch_and_add(&created, 1) < 1)	/* the application's first thread; not the HIP runtime's helpers */
		arm();
	return st.fn(st.arg);
}

int pthread_create(pthread_t *th, const pthread_attr_t *attr, void *(*fn)(void *), void *arg)
{
	static int (*real)(pthread_t *, const pthread_attr_t *, void *(*)(void *), void *);
	struct start *st = malloc(sizeof(*st));
	if(!real)
		real = dlsym(RTLD_NEXT, "pthread_create");
	st->fn = fn;
	st->arg = arg;
	return real(th, attr, trampoline, st);
}

__attribute__((constructor)) static void init(void)
{
	struct sigaction sa;
	memset(&sa, 0, sizeof(sa));
	sa.sa_sigaction = on_prof;
	sa.sa_flags = SA_SIGINFO | SA_RESTART;
	sigaction(SIGPROF, &sa, NULL);
	arm();
	atexit(dump);
}
