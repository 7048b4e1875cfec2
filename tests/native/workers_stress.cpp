// Stress of csrc/rpt_workers.hpp (no GPU): batches of every size from two submitting threads at once, every item exactly once,
// nothing touched after parallel_for has returned (the job and its argument live on the caller's stack).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
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

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 20000;
    int bad_a = 0, bad_b = 0;
    std::thread ta([&] { bad_a = submit(rounds, 1u); });
    std::thread tb([&] { bad_b = submit(rounds, 2u); });
    ta.join();
    tb.join();
    std::printf("threads %d rounds %d bad %d\n", rpth::Workers::instance().threads(), 2 * rounds, bad_a + bad_b);
    return bad_a + bad_b ? 1 : 0;
}
