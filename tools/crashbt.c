/* LD_PRELOAD shim for crash hunting on the GPU box (test infrastructure, not product).
 *
 * Installs SIGSEGV / SIGABRT / SIGBUS handlers BEFORE Python starts. Python's faulthandler installs its own handlers later and,
 * after dumping the Python stacks, re-raises into the previously installed handler -- this one -- which prints the C backtrace of
 * the FAULTING thread (module + offset, resolvable with llvm-symbolizer / addr2line against the same binaries) together with the
 * thread id and the fault address, then re-raises with the default action so the exit status stays the signal's.
 *
 * Build: gcc -O1 -g -fPIC -shared -o tools/_crashbt.so tools/crashbt.c
 * Use:   LD_PRELOAD=tools/_crashbt.so LIBC_FATAL_STDERR_=1 PYTHONFAULTHANDLER=1 python -m pytest ...
 */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <ucontext.h>
#include <stdio.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static int g_fd = 2; /* a private duplicate of the ORIGINAL stderr, taken in the constructor: pytest redirects fd 2 into its capture file later */
static void put(const char *s) { (void)!write(g_fd, s, strlen(s)); }

static void put_hex(unsigned long v) {
    char b[2 + 16 + 1];
    int i = 0;
    b[i++] = '0';
    b[i++] = 'x';
    for (int s = 60; s >= 0; s -= 4) b[i++] = "0123456789abcdef"[(v >> s) & 15];
    b[i] = 0;
    put(b);
}

static void handler(int sig, siginfo_t *si, void *uc) {
    void *frames[96];
    put("\n[crashbt] signal ");
    put_hex((unsigned long)sig);
    put(" tid ");
    put_hex((unsigned long)syscall(SYS_gettid));
    put(" pid ");
    put_hex((unsigned long)getpid());
    put(" fault address ");
    put_hex((unsigned long)(si ? si->si_addr : 0));
    put(" rip ");
    put_hex(uc ? (unsigned long)((ucontext_t *)uc)->uc_mcontext.gregs[REG_RIP] : 0);
    put("\n");
    {   /* the dying thread's name (hsa / rccl / torch helper threads name themselves) */
        char path[64], name[64] = "";
        snprintf(path, sizeof path, "/proc/self/task/%ld/comm", (long)syscall(SYS_gettid));
        int cf = open(path, O_RDONLY);
        if (cf >= 0) {
            ssize_t r = read(cf, name, sizeof name - 1);
            if (r > 0) name[r] = 0;
            close(cf);
        }
        put("[crashbt] thread name: ");
        put(name);
    }
    int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, g_fd);
    put("[crashbt] end of backtrace\n");
    /* a copy of the memory map makes the offsets resolvable later */
    FILE *f = fopen("/proc/self/maps", "r");
    if (f) {
        char line[512];
        put("[crashbt] executable mappings:\n");
        while (fgets(line, sizeof line, f))
            if (strstr(line, " r-xp ") || strstr(line, " r-x ")) put(line);
        fclose(f);
    }
    struct sigaction dfl;
    memset(&dfl, 0, sizeof dfl);
    dfl.sa_handler = SIG_DFL;
    sigaction(sig, &dfl, 0);
    raise(sig);
}

/* ---- optional heap sentinels (CRASHBT_CHURN=<threads>) --------------------------------------------------------------------------
 * Each sentinel thread keeps a ring of small malloc() chunks of the sizes runtime bookkeeping objects have (32 .. 512 bytes), filled
 * with a pattern, and verifies the pattern before it frees them: a write through a dangling pointer that lands in a re-used chunk is
 * REPORTED (chunk address, size, offset, bytes found) instead of surfacing much later as a corrupted free list. The threads also make
 * re-use of freshly freed chunks by ANOTHER thread likely, which is what a process with RCCL / runtime helper threads does. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>

static volatile int g_churn_stop;
static volatile unsigned long g_churn_hits;

unsigned long crashbt_churn_hits(void) { return g_churn_hits; }

static void *churn(void *arg) {
    enum { RING = 4096 };
    static __thread unsigned char *ring[RING];
    static __thread unsigned short sz[RING];
    uint64_t rng = 0x9e3779b97f4a7c15ull ^ (uintptr_t)arg;
    unsigned i = 0;
    while (!g_churn_stop) {
        rng = rng * 6364136223846793005ull + 1442695040888963407ull;
        unsigned slot = i++ % RING;
        if (ring[slot]) {
            unsigned char *c = ring[slot];
            for (unsigned k = 0; k < sz[slot]; k++)
                if (c[k] != 0x5a) {
                    g_churn_hits++;
                    put("\n[crashbt] SENTINEL: foreign write into live chunk ");
                    put_hex((unsigned long)c);
                    put(" size ");
                    put_hex(sz[slot]);
                    put(" offset ");
                    put_hex(k);
                    put(" bytes:");
                    for (unsigned j = k; j < sz[slot] && j < k + 32; j++) {
                        char b[4] = {' ', "0123456789abcdef"[c[j] >> 4], "0123456789abcdef"[c[j] & 15], 0};
                        put(b);
                    }
                    put("\n");
                    break;
                }
            free(c);
        }
        unsigned n = 32 + (unsigned)((rng >> 33) % 31) * 16; /* 32 .. 512 */
        ring[slot] = malloc(n);
        sz[slot] = (unsigned short)n;
        if (ring[slot]) memset(ring[slot], 0x5a, n);
        if ((i & 63) == 0) usleep(50);
    }
    return 0;
}

__attribute__((constructor)) static void crashbt_install(void) {
    int d = fcntl(2, F_DUPFD_CLOEXEC, 200);
    if (d >= 0) g_fd = d;
    static char alt[1 << 16];
    stack_t ss = {.ss_sp = alt, .ss_size = sizeof alt, .ss_flags = 0};
    sigaltstack(&ss, 0);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGABRT, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
    void *warm[4];
    backtrace(warm, 4); /* loads libgcc now, not inside the handler */
    const char *c = getenv("CRASHBT_CHURN");
    int n = c ? atoi(c) : 0;
    for (int t = 0; t < n && t < 16; t++) {
        pthread_t th;
        pthread_create(&th, 0, churn, (void *)(uintptr_t)(t + 1));
        pthread_detach(th);
    }
}
