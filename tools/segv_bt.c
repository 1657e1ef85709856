/* tools/segv_bt.c — helper for the GPU box: a native backtrace of the thread that takes a SIGSEGV / SIGBUS / SIGABRT (pytest's
 * faulthandler shows Python frames only, and the library's rank threads have none).  The handler runs on an alternate stack (a fault
 * that is a stack overflow still gets its backtrace) and can be installed again at any time — the HIP runtime, RCCL or a profiler may
 * replace it when they come up: tests/conftest.py re-installs it before every test when HJ_TEST_SEGV_BT names the built library.
 *   gcc -shared -fPIC -O1 -o /tmp/segv_bt.so tools/segv_bt.c;  HJ_TEST_SEGV_BT=/tmp/segv_bt.so python3 -m pytest ... */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static void on_fault(int sig, siginfo_t *si, void *uc) {
    (void)uc;
    void *bt[64];
    char line[128];
    const char *msg = "\n== native backtrace of the faulting thread ==\n";
    (void)!write(2, msg, strlen(msg));
    int n = 0;
    const char *p = "signal "; while (*p) line[n++] = *p++;
    line[n++] = (char)('0' + sig / 10); line[n++] = (char)('0' + sig % 10);
    p = ", fault address 0x"; while (*p) line[n++] = *p++;
    unsigned long a = (unsigned long)si->si_addr;
    for (int s = 60; s >= 0; s -= 4) line[n++] = "0123456789abcdef"[(a >> s) & 15];
    line[n++] = '\n';
    (void)!write(2, line, (size_t)n);
    backtrace_symbols_fd(bt, backtrace(bt, 64), 2);
    signal(sig, SIG_DFL);
    raise(sig);
}
void segv_bt_install(void) {
    static char *stack;
    if (!stack) stack = malloc(1 << 16);
    stack_t ss; ss.ss_sp = stack; ss.ss_size = 1 << 16; ss.ss_flags = 0;
    sigaltstack(&ss, 0); /* (of the calling thread: the others fault on their own stacks) */
    struct sigaction sa; memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_fault; sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
    sigaction(SIGSEGV, &sa, 0); sigaction(SIGBUS, &sa, 0); sigaction(SIGABRT, &sa, 0);
}
__attribute__((constructor)) static void install(void) { segv_bt_install(); }
