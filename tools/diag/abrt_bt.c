/* Diagnostic preload (tools/r05_soak5.sh): on SIGABRT / SIGSEGV / SIGBUS print the C backtrace of the thread that raised it to
 * stderr, then die by the default action.  gcc -shared -fPIC -O1 -o libabrt_bt.so abrt_bt.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void on_fatal(int sig)
{
    void* frames[64];
    const char* msg = sig == SIGABRT ? "\n*** abrt_bt: SIGABRT, backtrace of the raising thread:\n" : "\n*** abrt_bt: fatal signal, backtrace:\n";
    (void)!write(2, msg, strlen(msg));
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void)
{
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_fatal;
    sigaction(SIGABRT, &sa, 0);
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
}
