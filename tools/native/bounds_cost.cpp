// Host cost of the screen bounds, per object: the PROPOSAL (csrc/rpt_screen_bounds.hpp: outline sampling) and the PROOF
// (csrc/rpt_bounds_certify.hpp).  Reads records written by tools/bounds_cost.py: {int32 interval, int32 has_root, float root[6],
// rpt_object} per object; prints microseconds per object (best of several passes over the whole file), the share of
// proposals that claim something, of those proven, and the segment tests per proven claim.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../relativitypathtracer_amd/csrc/rpt_bounds_certify.hpp"

struct Rec { int interval, has_root; float root[6]; rpt_object o; };

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<Rec> recs;
    Rec r;
    while (std::fread(&r.interval, 4, 1, f) == 1 && std::fread(&r.has_root, 4, 1, f) == 1 && std::fread(r.root, 4, 6, f) == 6 && std::fread(&r.o, sizeof r.o, 1, f) == 1) recs.push_back(r);
    std::fclose(f);
    if (recs.empty()) return 2;
    const int passes = argc > 2 ? std::atoi(argv[2]) : 7;
    double best_prop = 1e300, best_both = 1e300;
    size_t claims = 0, proven = 0, tests = 0;
    volatile float sink = 0.0f;
    for (int p = 0; p < passes; p++) {
        auto t0 = std::chrono::steady_clock::now();
        for (const Rec &x : recs) { const rptb::Rect q = rptb::proposed_object_rect(x.o, x.interval, x.has_root ? x.root : nullptr); sink = sink + q.u0; }
        auto t1 = std::chrono::steady_clock::now();
        claims = proven = tests = 0;
        for (const Rec &x : recs) {
            rptb::cert::Stats st{0, 0, 0, 0};
            const rptb::Rect q = rptb::certified_object_rect(x.o, x.interval, x.has_root ? x.root : nullptr, &st);
            sink = sink + q.u0;
            if (st.reason >= 0) { claims++; if (st.reason == 0) { proven++; tests += (size_t)st.tests; } }
        }
        auto t2 = std::chrono::steady_clock::now();
        best_prop = std::min(best_prop, std::chrono::duration<double>(t1 - t0).count());
        best_both = std::min(best_both, std::chrono::duration<double>(t2 - t1).count());
    }
    const double n = (double)recs.size();
    std::printf("%zu objects: proposal %.2f us/object, proposal + proof %.2f us/object (proof %.2f); %zu claim something, %zu proven (%.2f %%), %.1f segment tests per proven claim\n",
                recs.size(), best_prop / n * 1e6, best_both / n * 1e6, (best_both - best_prop) / n * 1e6, claims, proven, claims ? 100.0 * proven / claims : 0.0,
                proven ? (double)tests / proven : 0.0);
    return 0;
}
