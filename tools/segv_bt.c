/* tools/segv_bt.c — LD_PRELOAD helper for the GPU box: a native backtrace of the thread that takes a SIGSEGV / SIGABRT (pytest's
 * faulthandler shows Python frames only, and the library's rank threads have none).
 *   gcc -shared -fPIC -O1 -o /tmp/segv_bt.so tools/segv_bt.c;  LD_PRELOAD=/tmp/segv_bt.so python3 -m pytest ... */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
static void on_fault(int sig) {
    void *bt[64];
    const char *msg = "\n== native backtrace of the faulting thread ==\n";
    (void)!write(2, msg, strlen(msg));
    backtrace_symbols_fd(bt, backtrace(bt, 64), 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
/* also callable late (ctypes), after the HIP runtime and the profiler have installed their own handlers */
void segv_bt_install(void) { signal(SIGSEGV, on_fault); signal(SIGABRT, on_fault); signal(SIGBUS, on_fault); }
__attribute__((constructor)) static void install(void) { segv_bt_install(); }
