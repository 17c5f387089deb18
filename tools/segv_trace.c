/* debugging aid: LD_PRELOAD this to get a backtrace (module + offset) on SIGSEGV; map with addr2line -e <so> <offset> */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void on_segv(int sig)
{
	void *bt[48];
	int n = backtrace(bt, 48);
	backtrace_symbols_fd(bt, n, 2);
	_exit(139);
}
__attribute__((constructor)) static void init(void) { signal(SIGSEGV, on_segv); }
