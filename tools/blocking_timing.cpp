// Host-only timing of the upload plan (softbody-webgpu_amd/csrc/sb_blocking.h) on a w x h lattice: stage times to stderr.
//   g++ -O2 -std=c++17 -pthread -Isoftbody-webgpu_amd/csrc tools/blocking_timing.cpp -o /tmp/blocking_timing && SB_UPLOAD_TIMING=1 /tmp/blocking_timing 1000 1000 1100 5
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "sb_blocking.h"

int main(int argc, char **argv)
{
    const uint32_t w = argc > 1 ? atoi(argv[1]) : 1000, h = argc > 2 ? atoi(argv[2]) : 1000, target = argc > 3 ? atoi(argv[3]) : 1100,
                   K = argc > 4 ? atoi(argv[4]) : 5;
    const uint32_t P = w * h;
    std::vector<float> px(P), py(P);
    SbHostBeams beams;
    auto add = [&](uint32_t a, uint32_t b) { SbHostBeam s{}; s.a = a; s.b = b; beams.push_back(s); };
    for (uint32_t x = 0; x < w; x++)
        for (uint32_t y = 0; y < h; y++) {
            const uint32_t i = x * h + y;
            px[i] = 30.0f * x;
            py[i] = 30.0f * y;
            if (y + 1 < h) add(i, i + 1);
            if (x + 1 < w) add(i, i + h);
            if (y + 1 < h && x + 1 < w) add(i, i + h + 1);
        }
    for (int rep = 0; rep < 3; rep++) {
        SbBlocking t;
        const auto t0 = std::chrono::steady_clock::now();
        sb_build_blocking(t, px, py, beams, target, K);
        fprintf(stderr, "== total %.2f ms (tiles %u, entries %zu)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
                t.ntiles, t.ent_la.size());
        // redundancy of the plan: beam evaluations and particle integrations per launch over K x (beams, particles)
        double ev = 0, in = 0;
        for (uint32_t k = 0; k < t.ntiles; k++) {
            for (uint32_t m = 0; m < K; m++) ev += t.lvl_cnt[(size_t)k * K + m];
            for (uint32_t r = 0; r < K; r++) in += t.ring_cnt[(size_t)k * (K + 1) + r];
        }
        fprintf(stderr, "   K %u: beam evaluations x%.3f, particle integrations x%.3f, largest region %u, most entries %u\n", K, ev / ((double)K * beams.size()),
                in / ((double)K * P), t.max_region, t.max_entries);
    }
    for (int rep = 0; rep < 3; rep++) { // the single-substep tiling of the same scene (spatial-hash mode, block_substeps = 1)
        SbTiling tl;
        const auto t0 = std::chrono::steady_clock::now();
        sb_build_tiling(tl, px, py, beams, 1024);
        fprintf(stderr, "== tiling %.2f ms (tiles %u, copies %zu, cut %llu)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
                tl.ntiles, tl.copy_slot.size(), (unsigned long long)tl.cut_beams);
    }
    return 0;
}
