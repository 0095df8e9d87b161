// Host-only check of softbody-webgpu_amd/csrc/sb_tiling.h (the scene partition of SB_PATH_TILED):
// built and run by tests/test_tiling_cpu.py with plain g++ (no HIP needed).
//   usage: tiling_check <width> <height> <target> <seed> <mode>     mode 0 = lattice, 1 = random graph
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>

#include "sb_tiling.h"

#define REQUIRE(c)                                                  \
    do {                                                            \
        if (!(c)) {                                                 \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                               \
        }                                                           \
    } while (0)

static unsigned long long rng_state;
static double rnd() { rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(rng_state >> 11) / 9007199254740992.0; }

int main(int argc, char **argv)
{
    if (argc < 6) return 2;
    const uint32_t w = atoi(argv[1]), h = atoi(argv[2]), target = atoi(argv[3]);
    rng_state = strtoull(argv[4], nullptr, 10);
    const int mode = atoi(argv[5]);
    const uint32_t P = w * h;
    std::vector<float> px(P), py(P);
    SbHostBeams beams;
    for (uint32_t x = 0; x < w; x++)
        for (uint32_t y = 0; y < h; y++) {
            uint32_t i = x * h + y;
            px[i] = mode ? (float)(rnd() * 1000.0) : 30.0f * x + (float)rnd();
            py[i] = mode ? (float)(rnd() * 1000.0) : 30.0f * y + (float)rnd();
            if (mode == 0) {
                auto add = [&](uint32_t a, uint32_t b) { SbHostBeam s{}; s.a = a; s.b = b; beams.push_back(s); };
                if (y + 1 < h) add(i, i + 1);
                if (x + 1 < w) add(i, i + h);
                if (y + 1 < h && x + 1 < w) add(i, i + h + 1);
            }
        }
    if (mode == 1) {
        px[3] = NAN; // non-finite positions must not break the partition
        py[5] = INFINITY;
        for (uint32_t k = 0; k < 3 * P; k++) {
            SbHostBeam s{};
            s.a = (uint32_t)(rnd() * P) % P;
            s.b = (uint32_t)(rnd() * P) % P; // arbitrary long-range beams, self-beams included
            beams.push_back(s);
        }
    }
    const uint32_t B = (uint32_t)beams.size();
    SbTiling t;
    sb_build_tiling(t, px, py, beams, target);

    // tiles partition the particles, no tile above the target, populations balanced
    REQUIRE(t.tile_p0.size() == t.ntiles + 1 && t.tile_p0[0] == 0 && t.tile_p0[t.ntiles] == P);
    REQUIRE(t.ntiles == (P + std::max(64u, target) - 1) / std::max(64u, target) || target < 64);
    std::vector<uint32_t> internal_of_slot(P, 0xFFFFFFFFu), tile_of(P);
    uint32_t mn = P, mx = 0;
    for (uint32_t k = 0; k < t.ntiles; k++) {
        uint32_t n = t.tile_p0[k + 1] - t.tile_p0[k];
        REQUIRE(n > 0 && n <= std::max(64u, target));
        mn = std::min(mn, n);
        mx = std::max(mx, n);
        for (uint32_t i = t.tile_p0[k]; i < t.tile_p0[k + 1]; i++) {
            REQUIRE(t.order[i] < P && internal_of_slot[t.order[i]] == 0xFFFFFFFFu);
            internal_of_slot[t.order[i]] = i;
            tile_of[i] = k;
            if (i > t.tile_p0[k]) REQUIRE(t.order[i - 1] < t.order[i]); // slot order inside a tile
        }
    }
    REQUIRE(mx - mn <= 1 + mx / 8);
    REQUIRE(t.max_own == mx);

    // halo lists: sorted, unique, foreign
    uint32_t max_all = 0;
    for (uint32_t k = 0; k < t.ntiles; k++) {
        for (uint32_t q = t.tile_h0[k]; q < t.tile_h0[k + 1]; q++) {
            REQUIRE(t.halo_idx[q] < P && tile_of[t.halo_idx[q]] != k);
            if (q > t.tile_h0[k]) REQUIRE(t.halo_idx[q - 1] < t.halo_idx[q]);
        }
        max_all = std::max(max_all, (t.tile_p0[k + 1] - t.tile_p0[k]) + (t.tile_h0[k + 1] - t.tile_h0[k]));
    }
    REQUIRE(max_all == t.max_all);

    // every beam: one copy per distinct endpoint tile, local indices resolve to the right particles,
    // copy_of_slot names the copy in endpoint A's tile; slices padded to a multiple of 4
    std::vector<int> copies(B, 0);
    uint64_t cut = 0;
    for (uint32_t k = 0; k < t.ntiles; k++) {
        REQUIRE((t.tile_b0[k + 1] - t.tile_b0[k]) % 4 == 0);
        const uint32_t own = t.tile_p0[k + 1] - t.tile_p0[k];
        for (uint32_t c = t.tile_b0[k]; c < t.tile_b0[k + 1]; c++) {
            const uint32_t s = t.copy_slot[c];
            if (s == 0xFFFFFFFFu) {
                REQUIRE(t.copy_la[c] == 0xFFFFFFFFu && t.copy_lb[c] == 0xFFFFFFFFu);
                continue;
            }
            REQUIRE(s < B);
            copies[s]++;
            auto resolve = [&](uint32_t l) { return l < own ? t.tile_p0[k] + l : t.halo_idx[t.tile_h0[k] + (l - own)]; };
            REQUIRE(t.copy_la[c] < own + (t.tile_h0[k + 1] - t.tile_h0[k]));
            REQUIRE(t.copy_lb[c] < own + (t.tile_h0[k + 1] - t.tile_h0[k]));
            REQUIRE(resolve(t.copy_la[c]) == internal_of_slot[beams[s].a]);
            REQUIRE(resolve(t.copy_lb[c]) == internal_of_slot[beams[s].b]);
            REQUIRE(t.copy_la[c] < own || t.copy_lb[c] < own); // at least one endpoint is owned here
        }
    }
    for (uint32_t s = 0; s < B; s++) {
        const uint32_t ta = tile_of[internal_of_slot[beams[s].a]], tb = tile_of[internal_of_slot[beams[s].b]];
        REQUIRE(copies[s] == (ta == tb ? 1 : 2));
        cut += ta != tb;
        const uint32_t c = t.copy_of_slot[s];
        REQUIRE(t.copy_slot[c] == s && c >= t.tile_b0[ta] && c < t.tile_b0[ta + 1]);
    }
    REQUIRE(cut == t.cut_beams);
    printf("TILING_OK tiles=%u particles=%u beams=%u cut=%llu max_own=%u max_all=%u\n", t.ntiles, P, B,
           (unsigned long long)cut, t.max_own, t.max_all);
    return 0;
}
