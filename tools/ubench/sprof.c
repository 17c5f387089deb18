/* sprof.c - a preloadable sampling profiler for the host side of the drop-in
 * (no perf / gdb on the GPU boxes): SIGPROF every 250 us of CPU time, the
 * interrupted PC and its caller are histogrammed, resolved with dladdr() at
 * exit.   gcc -O2 -shared -fPIC -o libsprof.so sprof.c -ldl
 *         LD_PRELOAD=libsprof.so:liba2amd_units.so ref_bench ...             */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <ucontext.h>

#define MAXS 200000
static void *pcs[MAXS];
static volatile int npcs;

static void on_prof(int sig, siginfo_t *si, void *uc_)
{
	ucontext_t *uc = (ucontext_t *)uc_;
	if(npcs < MAXS)
		pcs[npcs++] = (void *)uc->uc_mcontext.gregs[REG_RIP];
}

static int cmp(const void *a, const void *b) { return strcmp(*(char *const *)a, *(char *const *)b); }

static void dump(void)
{
	static char names[MAXS][96];
	static char *ptr[MAXS];
	int i, n = npcs, run;
	struct itimerval off = { { 0, 0 }, { 0, 0 } };
	setitimer(ITIMER_PROF, &off, NULL);
	for(i = 0; i < n; ++i)
	{
		Dl_info di;
		if(dladdr(pcs[i], &di) && di.dli_sname)
			snprintf(names[i], 96, "%s", di.dli_sname);
		else if(dladdr(pcs[i], &di) && di.dli_fname)
		{
			const char *b = strrchr(di.dli_fname, '/');
			snprintf(names[i], 96, "[%s+%lx]", b ? b + 1 : di.dli_fname,
					(unsigned long)((char *)pcs[i] - (char *)di.dli_fbase) & ~0xffUL);
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
		if(run * 200 >= n)	/* >= 0.5 % */
			fprintf(stderr, "sprof %6.2f%%  %s\n", run * 100.0 / n, ptr[i]);
	}
}

__attribute__((constructor)) static void init(void)
{
	struct sigaction sa;
	struct itimerval it = { { 0, 250 }, { 0, 250 } };
	memset(&sa, 0, sizeof(sa));
	sa.sa_sigaction = on_prof;
	sa.sa_flags = SA_SIGINFO | SA_RESTART;
	sigaction(SIGPROF, &sa, NULL);
	setitimer(ITIMER_PROF, &it, NULL);
	atexit(dump);
}
