// LD_PRELOAD helper for crash hunts on the GPU box (no gdb there): prints a native backtrace on SIGSEGV / SIGBUS / SIGABRT, then
// re-raises.  build: gcc -shared -fPIC -O1 -g tools/segv_trace.c -o tools/_bin/libsegv_trace.so
// use:   LD_PRELOAD=tools/_bin/libsegv_trace.so python -m pytest -p no:faulthandler ...
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
#include <stdlib.h>
#include <sys/syscall.h>
static int g_fd = 2;   // FP_SEGV_TRACE_FILE: write there (pytest captures fd 2)
static void handler(int sig, siginfo_t *si, void *ctx) {
  (void)ctx;
  void *bt[64];
  char msg[128];
  int n = snprintf(msg, sizeof msg, "\n[segv_trace] signal %d at address %p, thread %ld\n", sig, si ? si->si_addr : 0, (long)syscall(SYS_gettid));
  (void)!write(g_fd, msg, n);
  n = backtrace(bt, 64);
  backtrace_symbols_fd(bt, n, g_fd);
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) void segv_trace_install(void) {
  const char *f = getenv("FP_SEGV_TRACE_FILE");
  if (f && g_fd == 2) { int fd = open(f, O_WRONLY | O_CREAT | O_APPEND, 0644); if (fd >= 0) g_fd = fd; }
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_RESETHAND;
  static char stack[1 << 16];
  stack_t ss = {stack, 0, sizeof stack};
  sigaltstack(&ss, 0);
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
}
