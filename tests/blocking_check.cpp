// Host-only check of softbody-webgpu_amd/csrc/sb_blocking.h (the plan of the temporally blocked kernel):
// built and run by tests/test_blocking_cpu.py with plain g++ (no HIP needed).
//   usage: blocking_check <width> <height> <target> <seed> <mode> <K>     mode 0 = lattice, 1 = random graph
// Two parts.  (1) Structural invariants of the plan.  (2) A dependency-exact emulation: a toy "physics" of
// 64-bit hashes with exactly the data flow of compute.wgsl (a beam reads its two endpoints and its own state and
// adds to both endpoints' wrapping integer sums; a particle reads its own state and its sum) is stepped k times
// globally and, tile by tile, with the schedule k_substep_blocked runs (prefixes per substep, entries riding along
// in groups, sums consumed and cleared by the particle phase): own particles and owned beams must agree exactly
// for every k <= K.  Any hole in the rings, the prefixes or the entry lists changes a hash.
#include <cstdio>
#include <cstdlib>
#include <set>

#include "sb_blocking.h"

#define REQUIRE(c)                                                  \
    do {                                                            \
        if (!(c)) {                                                 \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                               \
        }                                                           \
    } while (0)

static unsigned long long rng_state;
static double rnd() { rng_state = rng_state * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(rng_state >> 11) / 9007199254740992.0; }
static uint64_t mix(uint64_t x)
{
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31;
    return x;
}
// toy beam: new state and the two contributions
static void beam_eval(uint64_t xa, uint64_t xb, uint64_t y, uint64_t &y_new, uint64_t &fa, uint64_t &fb)
{
    const uint64_t f = mix(xa * 3 + mix(xb) + mix(y ^ 0x1234));
    y_new = mix(f ^ y);
    fa = f;
    fb = mix(f) | 1;
}
static uint64_t particle_eval(uint64_t x, uint64_t sum) { return mix(x + mix(sum ^ 0x9E3779B97F4A7C15ULL)); }

int main(int argc, char **argv)
{
    if (argc < 7) return 2;
    const uint32_t w = atoi(argv[1]), h = atoi(argv[2]), target = atoi(argv[3]);
    rng_state = strtoull(argv[4], nullptr, 10);
    const int mode = atoi(argv[5]);
    const uint32_t K = atoi(argv[6]);
    const uint32_t P = w * h;
    std::vector<float> px(P), py(P);
    SbHostBeams beams;
    auto add = [&](uint32_t a, uint32_t b) { SbHostBeam s{}; s.a = a; s.b = b; beams.push_back(s); };
    for (uint32_t x = 0; x < w; x++)
        for (uint32_t y = 0; y < h; y++) {
            const uint32_t i = x * h + y;
            px[i] = mode ? (float)(rnd() * 1000.0) : 30.0f * x + (float)rnd();
            py[i] = mode ? (float)(rnd() * 1000.0) : 30.0f * y + (float)rnd();
            if (mode == 0) {
                if (y + 1 < h) add(i, i + 1);
                if (x + 1 < w) add(i, i + h);
                if (y + 1 < h && x + 1 < w) add(i, i + h + 1);
                if (y > 0 && x + 1 < w && (i % 3) == 0) add(i, i + h - 1);
            }
        }
    if (mode == 1) { // random graph: mostly short beams, some long-range, some self beams, some parallel beams
        for (uint32_t i = 0; i < P; i++) {
            const int deg = (int)(rnd() * 4);
            for (int d = 0; d < deg; d++) {
                uint32_t j = rnd() < 0.9 ? (uint32_t)std::min<double>(P - 1, std::max(0.0, i + (rnd() - 0.5) * 40)) : (uint32_t)(rnd() * P);
                if (rnd() < 0.02) j = i;
                add(i, j);
                if (rnd() < 0.02) add(j, i);
            }
        }
        if (P > 10) px[3] = NAN, py[7] = INFINITY;
    }
    const uint32_t B = (uint32_t)beams.size();
    SbBlocking t;
    sb_build_blocking(t, px, py, beams, target, K);
    const uint32_t T = t.ntiles;
    REQUIRE(t.K == K && t.tile_p0.size() == T + 1 && t.tile_p0[T] == P);
    std::vector<uint32_t> internal_of_slot(P), tile_of(P);
    {
        std::vector<char> seen(P, 0);
        for (uint32_t i = 0; i < P; i++) {
            REQUIRE(t.order[i] < P && !seen[t.order[i]]);
            seen[t.order[i]] = 1;
            internal_of_slot[t.order[i]] = i;
        }
        for (uint32_t k = 0; k < T; k++)
            for (uint32_t i = t.tile_p0[k]; i < t.tile_p0[k + 1]; i++) tile_of[i] = k;
    }
    // owned beams: a bijection slot <-> g, tile-major, slot order inside the tile
    REQUIRE(t.tile_b0[T] == B);
    for (uint32_t k = 0; k < T; k++)
        for (uint32_t g = t.tile_b0[k]; g < t.tile_b0[k + 1]; g++) {
            const uint32_t s = t.beam_slot[g];
            REQUIRE(s < B && t.g_of_slot[s] == g && tile_of[internal_of_slot[beams[s].a]] == k);
        }
    // per tile: region, rings, entries
    uint64_t total_entries = 0;
    for (uint32_t k = 0; k < T; k++) {
        const uint32_t p0 = t.tile_p0[k], n_own = t.tile_p0[k + 1] - p0, h0 = t.tile_h0[k], nh = t.tile_h0[k + 1] - h0;
        const uint32_t *rc = &t.ring_cnt[(size_t)k * (K + 1)], *lc = &t.lvl_cnt[(size_t)k * K];
        REQUIRE(rc[0] == n_own && rc[K] == n_own + nh);
        // independent BFS
        std::vector<int> ring(P, -1);
        std::vector<uint32_t> fr;
        for (uint32_t i = p0; i < p0 + n_own; i++) ring[i] = 0, fr.push_back(i);
        for (uint32_t r = 1; r <= K; r++) {
            std::vector<uint32_t> nx;
            std::set<uint32_t> fset(fr.begin(), fr.end());
            for (uint32_t s = 0; s < B; s++) {
                const uint32_t a = internal_of_slot[beams[s].a], b = internal_of_slot[beams[s].b];
                if (fset.count(a) && ring[b] < 0) ring[b] = (int)r, nx.push_back(b);
                if (fset.count(b) && ring[a] < 0) ring[a] = (int)r, nx.push_back(a);
            }
            fr.swap(nx);
            uint32_t cnt = 0;
            for (uint32_t i = 0; i < P; i++) cnt += ring[i] >= 0 && ring[i] <= (int)r;
            REQUIRE(rc[r] == cnt);
        }
        std::vector<uint32_t> region(n_own + nh);
        for (uint32_t q = 0; q < n_own; q++) region[q] = p0 + q;
        for (uint32_t q = 0; q < nh; q++) {
            region[n_own + q] = t.halo_idx[h0 + q];
            REQUIRE(ring[region[n_own + q]] >= 1);
            if (q) REQUIRE(ring[region[n_own + q - 1]] <= ring[region[n_own + q]]); // sorted by ring
        }
        for (uint32_t r = 0; r <= K; r++)
            for (uint32_t q = (r ? rc[r - 1] : 0); q < rc[r]; q++) REQUIRE(ring[region[q]] == (int)r);
        // entries: every beam with an endpoint of ring <= K-1 exactly once, sorted by the smaller ring, own beams first
        const uint32_t e0 = t.tile_e0[k], ne = t.tile_e0[k + 1] - e0, n_ownb = t.tile_b0[k + 1] - t.tile_b0[k];
        total_entries += ne;
        std::set<uint32_t> listed;
        uint32_t nstate = 0;
        for (uint32_t j = 0; j < ne; j++) {
            const uint32_t s = t.ent_slot[e0 + j];
            REQUIRE(s < B && listed.insert(s).second);
            const uint32_t a = internal_of_slot[beams[s].a], b = internal_of_slot[beams[s].b];
            REQUIRE(t.ent_la[e0 + j] < region.size() && region[t.ent_la[e0 + j]] == a);
            REQUIRE(t.ent_lb[e0 + j] < region.size() && region[t.ent_lb[e0 + j]] == b);
            const int m = std::min(ring[a], ring[b]);
            REQUIRE(m >= 0 && m <= (int)K - 1);
            for (uint32_t mm = 0; mm < K; mm++) REQUIRE((j < lc[mm]) == (m <= (int)mm));
            if (j < n_ownb) REQUIRE(t.g_of_slot[s] == t.tile_b0[k] + j);
            else REQUIRE(t.ent_state[t.tile_s0[k] + nstate++] == t.g_of_slot[s] && tile_of[a] != k);
        }
        REQUIRE(nstate == t.tile_s0[k + 1] - t.tile_s0[k]);
        for (uint32_t s = 0; s < B; s++) {
            const int ra = ring[internal_of_slot[beams[s].a]], rb = ring[internal_of_slot[beams[s].b]];
            const bool need = (ra >= 0 && ra <= (int)K - 1) || (rb >= 0 && rb <= (int)K - 1);
            REQUIRE(need == (listed.count(s) != 0));
        }
        // neighbour tiles = owners of the halo
        std::set<uint32_t> owners;
        for (uint32_t q = 0; q < nh; q++) owners.insert(tile_of[region[n_own + q]]);
        REQUIRE(owners.size() == t.tile_n0[k + 1] - t.tile_n0[k]);
        for (uint32_t i = t.tile_n0[k]; i < t.tile_n0[k + 1]; i++) REQUIRE(owners.count(t.tile_nb[i]));
    }
    REQUIRE(total_entries == t.tile_e0[T] && t.slot_e0[B] == total_entries);
    for (uint32_t s = 0; s < B; s++)
        for (uint32_t e = t.slot_e0[s]; e < t.slot_e0[s + 1]; e++) REQUIRE(t.ent_slot[t.slot_ent[e]] == s);

    // ---- dependency-exact emulation
    std::vector<uint64_t> x0(P), y0(B);
    for (auto &v : x0) v = mix((uint64_t)(rnd() * 1e18));
    for (auto &v : y0) v = mix((uint64_t)(rnd() * 1e18));
    const uint32_t TT = 8, G = 2; // threads per tile and group width of the emulated schedule (any values exercise the logic)
    for (uint32_t k_run = 1; k_run <= K; k_run++) {
        // global reference: k_run substeps (particles and beam states in internal / g order)
        std::vector<uint64_t> x = x0, y = y0;
        for (uint32_t s = 0; s < k_run; s++) {
            std::vector<uint64_t> sum(P, 0), yn(B);
            for (uint32_t g = 0; g < B; g++) {
                const uint32_t sl = t.beam_slot[g], a = internal_of_slot[beams[sl].a], b = internal_of_slot[beams[sl].b];
                uint64_t fa, fb;
                beam_eval(x[a], x[b], y[g], yn[g], fa, fb);
                sum[a] += fa;
                sum[b] += fb;
            }
            for (uint32_t i = 0; i < P; i++) x[i] = particle_eval(x[i], sum[i]);
            y.swap(yn);
        }
        // per tile, the kernel's schedule
        for (uint32_t k = 0; k < T; k++) {
            const uint32_t p0 = t.tile_p0[k], n_own = t.tile_p0[k + 1] - p0, h0 = t.tile_h0[k];
            const uint32_t *rc = &t.ring_cnt[(size_t)k * (K + 1)], *lc = &t.lvl_cnt[(size_t)k * K];
            const uint32_t e0 = t.tile_e0[k], b0 = t.tile_b0[k], n_ownb = t.tile_b0[k + 1] - b0, s0 = t.tile_s0[k];
            const uint32_t np_load = rc[k_run], ne_load = lc[k_run - 1];
            const uint32_t cap = t.max_region;
            std::vector<uint64_t> lx(cap + 2, 0), lsum(cap + 2, 0), ly(ne_load);
            for (uint32_t q = 0; q < np_load; q++) lx[q] = x0[q < n_own ? p0 + q : t.halo_idx[h0 + q - n_own]];
            lx[cap] = 11, lx[cap + 1] = 22;
            for (uint32_t j = 0; j < ne_load; j++) ly[j] = y0[j < n_ownb ? b0 + j : t.ent_state[s0 + j - n_ownb]];
            const uint32_t maxb = (std::max(ne_load, 1u) + TT - 1) / TT;
            for (uint32_t s = 1; s <= k_run; s++) {
                const uint32_t nbl = lc[k_run - s], npr = rc[k_run - s];
                for (uint32_t tid = 0; tid < TT; tid++)
                    for (uint32_t i0 = 0; i0 < maxb + G; i0 += G) {
                        if (!(tid + i0 * TT < nbl)) continue;
                        for (uint32_t u = 0; u < G; u++) {
                            const uint32_t j = tid + (i0 + u) * TT;
                            const bool real = j < ne_load;
                            const uint32_t la = real ? t.ent_la[e0 + j] : cap, lb = real ? t.ent_lb[e0 + j] : cap + 1;
                            uint64_t yd = 5, &yy = real ? ly[j] : yd, fa, fb, yn;
                            beam_eval(lx[la], lx[lb], yy, yn, fa, fb);
                            yy = yn;
                            lsum[la] += fa;
                            lsum[lb] += fb;
                        }
                    }
                for (uint32_t q = 0; q < npr; q++) {
                    lx[q] = particle_eval(lx[q], lsum[q]);
                    lsum[q] = 0;
                }
            }
            for (uint32_t q = 0; q < n_own; q++) REQUIRE(lx[q] == x[p0 + q]);
            for (uint32_t j = 0; j < n_ownb; j++) REQUIRE(ly[j] == y[b0 + j]);
        }
    }
    printf("BLOCKING_OK tiles=%u entries=%llu max_region=%u max_entries=%u\n", T, (unsigned long long)total_entries, t.max_region,
           t.max_entries);
    return 0;
}
