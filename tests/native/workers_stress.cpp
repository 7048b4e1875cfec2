// Stress of csrc/rpt_workers.hpp (no GPU): batches of every size from two submitting threads at once, every item exactly once,
// nothing touched after parallel_for has returned (the job and its argument live on the caller's stack).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cerrno>
#include <cstring>
#include <sys/wait.h>
#include <thread>
#include <unistd.h>
#include <vector>
#include "../../relativitypathtracer_amd/csrc/rpt_workers.hpp"

struct Batch {
    std::atomic<int> hits[64];
    int payload[64];
};
static void item(void *arg, int i) {
    Batch &b = *(Batch *)arg;
    b.hits[i].fetch_add(1);
    volatile double x = b.payload[i];
    for (int k = 0; k < 200; k++) x = x * 1.0000001 + 1e-9;      // about a microsecond
    b.payload[i] = (int)x;
}

static int submit(int rounds, unsigned seed) {
    int bad = 0;
    for (int r = 0; r < rounds; r++) {
        seed = seed * 1664525u + 1013904223u;
        const int n = 1 + (int)((seed >> 16) % 64);
        Batch b;
        for (int i = 0; i < 64; i++) { b.hits[i].store(0); b.payload[i] = i; }
        rpth::Workers::instance().parallel_for(n, item, &b);
        for (int i = 0; i < 64; i++) bad += b.hits[i].load() != (i < n ? 1 : 0);
        std::memset((void *)&b, 0xAB, sizeof b);                  // a late worker would trip over this
        if ((seed >> 8) % 97 == 0) std::this_thread::sleep_for(std::chrono::microseconds(400));   // let the workers fall asleep
    }
    return bad;
}

// fork() while a thread keeps the pool busy: the child has no workers and inherits whatever the counters (and a mutex a worker
// may have held) said at that instant.  Its batches must still complete — on the calling thread alone (pthread_atfork child
// handler in rpt_workers.hpp) — and it must report threads 0.
static int fork_mode(int forks) {
    // the pool exists before anything forks (the library creates it in rpt_create): a fork DURING its construction — a function-local
    // static being initialised by another thread — would leave the child waiting for an initialiser that does not exist there
    (void)rpth::Workers::instance().threads();
    std::atomic<bool> stop{false};
    std::thread busy([&] { while (!stop.load()) submit(50, 7u); });
    int bad = 0;
    for (int f = 0; f < forks; f++) {
        std::this_thread::sleep_for(std::chrono::microseconds(150 + 37 * (f % 11)));      // some forks catch the workers asleep, some at work
        const pid_t pid = fork();
        if (pid == 0) {
            alarm(20);                                            // a child that hangs dies of SIGALRM and is counted
            const int b = submit(200, 100u + (unsigned)f);
            _exit(b ? 1 : (rpth::Workers::instance().threads() != 0 ? 2 : 0));
        }
        int status = 0;
        pid_t w = -1;
        if (pid > 0) do { w = waitpid(pid, &status, 0); } while (w < 0 && errno == EINTR);
        if (pid < 0 || w != pid || !WIFEXITED(status) || WEXITSTATUS(status) != 0) {
            bad++;
            std::fprintf(stderr, "fork %d: pid %d waitpid %d errno %d exited %d code %d signalled %d signal %d\n", f, (int)pid, (int)w, errno,
                         pid > 0 && w == pid ? WIFEXITED(status) : -1, pid > 0 && w == pid && WIFEXITED(status) ? WEXITSTATUS(status) : -1,
                         pid > 0 && w == pid ? WIFSIGNALED(status) : -1, pid > 0 && w == pid && WIFSIGNALED(status) ? WTERMSIG(status) : -1);
        }
    }
    stop.store(true);
    busy.join();
    std::printf("threads %d forks %d bad %d\n", rpth::Workers::instance().threads(), forks, bad);
    return bad ? 1 : 0;
}

int main(int argc, char **argv) {
    if (argc > 2 && std::strcmp(argv[1], "--fork") == 0) return fork_mode(std::atoi(argv[2]));
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 20000;
    int bad_a = 0, bad_b = 0;
    std::thread ta([&] { bad_a = submit(rounds, 1u); });
    std::thread tb([&] { bad_b = submit(rounds, 2u); });
    ta.join();
    tb.join();
    std::printf("threads %d rounds %d bad %d\n", rpth::Workers::instance().threads(), 2 * rounds, bad_a + bad_b);
    return bad_a + bad_b ? 1 : 0;
}
