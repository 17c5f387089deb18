#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void h(int s){ void *b[40]; int n=backtrace(b,40); backtrace_symbols_fd(b,n,2); _exit(99);} 
__attribute__((constructor)) static void init(void){ signal(SIGSEGV,h); }
