// sb_tiling.h -- host-side partition of a scene into particle tiles for SB_PATH_TILED.
//
// A tile is a spatially compact set of <= `target` particles (recursive coordinate bisection
// of the uploaded positions) that one workgroup keeps in LDS for a whole substep.  Every beam
// is stored in the slice of the tile that owns its endpoints; a beam whose endpoints live in
// two tiles ("cut beam") is stored in BOTH slices, each with its own copy of the dynamic state
// (target/last/strain/stress).  The two copies see identical inputs (read-buffer positions) and
// run identical arithmetic, so they stay bit-identical forever; each tile applies only the force
// on the endpoint it owns.  That is what removes every global atomic and every inter-workgroup
// dependency from the substep (DESIGN.md "cut beams").
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <thread>
#include <memory>
#include <vector>

struct SbHostBeam {
    uint32_t a, b;   // endpoints as particle mapping SLOTS
    uint32_t da, db; // endpoints as particle DATA indices (what the caller's record holds)
    float f[9];      // length, target, last, spring, damp, yield, limit, strain, stress
};

struct SbTiling {
    uint32_t ntiles = 0;
    std::vector<uint32_t> order;        // internal particle -> slot (tiles are contiguous ranges)
    std::vector<uint32_t> tile_p0;      // [ntiles+1]
    std::vector<uint32_t> tile_b0;      // [ntiles+1] beam-copy ranges (padded to x4 with dead copies)
    std::vector<uint32_t> tile_h0;      // [ntiles+1] halo ranges
    std::vector<uint32_t> halo_idx;     // internal particle index of each halo entry
    std::vector<uint32_t> copy_la, copy_lb; // tile-local endpoint indices; 0xFFFFFFFF = padding
    std::vector<uint32_t> copy_slot;    // beam slot of each copy; 0xFFFFFFFF = padding
    std::vector<uint32_t> copy_of_slot; // beam slot -> the copy in the tile that owns endpoint A
    uint32_t max_own = 0, max_all = 0;
    uint64_t cut_beams = 0;
};

namespace sbt {

// std::vector whose resize(n) / vector(n) leave the elements uninitialised (trivial types only): the plan's arrays are tens
// of megabytes each and every one is fully overwritten by a parallel loop right after it is sized -- zero-filling them first
// costs a serial pass and takes every page fault on one thread (r02: 36 of the plan's 92 ms per million particles)
template <typename T>
struct default_init_alloc : std::allocator<T> {
    template <typename U> struct rebind { using other = default_init_alloc<U>; };
    using std::allocator<T>::allocator;
    template <typename U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
    template <typename U, typename... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
template <typename T> using uvec = std::vector<T, default_init_alloc<T>>;
} // namespace sbt
using SbHostBeams = sbt::uvec<SbHostBeam>; // (3 M records of 52 bytes: see above)
namespace sbt {

// f(begin, end) over [0, n) in contiguous chunks on a few host threads (uploads of millions of records)
template <typename F>
inline void parallel_ranges(size_t n, size_t min_chunk, F f)
{
    unsigned hw = std::thread::hardware_concurrency();
    const size_t nt = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(hw ? hw : 4u, 16u), n / std::max<size_t>(min_chunk, 1)));
    if (nt <= 1) {
        f((size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    for (size_t w = 0; w < nt; w++) th.emplace_back([=, &f] { f(n * w / nt, n * (w + 1) / nt); });
    for (auto &t : th) t.join();
}

// in-place inclusive prefix sum of v[1..n] (v[0] stays: the CSR convention "v[i + 1] += v[i]"), two passes on a few threads
template <typename V>
inline void parallel_csr_scan(V &v)
{
    const size_t n = v.size();
    unsigned hw = std::thread::hardware_concurrency();
    const size_t nt = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(hw ? hw : 4u, 16u), n >> 16));
    if (nt <= 1) {
        for (size_t i = 1; i < n; i++) v[i] += v[i - 1];
        return;
    }
    std::vector<uint64_t> part(nt + 1, 0);
    {
        std::vector<std::thread> th;
        for (size_t w = 0; w < nt; w++)
            th.emplace_back([&, w] {
                uint64_t s = 0;
                for (size_t i = n * w / nt; i < n * (w + 1) / nt; i++) s += v[i];
                part[w + 1] = s;
            });
        for (auto &t : th) t.join();
    }
    for (size_t w = 0; w < nt; w++) part[w + 1] += part[w];
    std::vector<std::thread> th;
    for (size_t w = 0; w < nt; w++)
        th.emplace_back([&, w] {
            auto run = (typename V::value_type)part[w];
            for (size_t i = n * w / nt; i < n * (w + 1) / nt; i++) {
                run += v[i];
                v[i] = run;
            }
        });
    for (auto &t : th) t.join();
}

struct Splitter {
    const std::vector<float> &x, &y;
    std::vector<uint32_t> &order;
    std::vector<uint32_t> &tile_p0; // [tiles + 1], sized by the caller; tile_p0[0] = 0
    uint32_t target;

    static float key(float v) { return std::isfinite(v) ? v : 0.0f; }

    // [lo,hi) becomes tiles first .. first+k-1 of near-equal population.  The two halves of a split touch disjoint
    // ranges of `order` and of `tile_p0`, so the upper levels of the recursion run their left halves on threads of
    // their own (`fan` = levels left to fan out).
    void split(uint32_t lo, uint32_t hi, uint32_t k, uint32_t first, int fan)
    {
        if (k <= 1) {
            std::sort(order.begin() + lo, order.begin() + hi); // slot order inside a tile
            tile_p0[first + 1] = hi;
            return;
        }
        float minx = INFINITY, maxx = -INFINITY, miny = INFINITY, maxy = -INFINITY;
        for (uint32_t i = lo; i < hi; i++) {
            float vx = key(x[order[i]]), vy = key(y[order[i]]);
            minx = std::min(minx, vx); maxx = std::max(maxx, vx);
            miny = std::min(miny, vy); maxy = std::max(maxy, vy);
        }
        const bool along_x = (maxx - minx) >= (maxy - miny);
        const std::vector<float> &c = along_x ? x : y;
        uint32_t kl = k / 2;
        uint32_t nl = (uint32_t)(((uint64_t)(hi - lo) * kl) / k);
        auto cmp = [&](uint32_t p, uint32_t q) {
            float a = key(c[p]), b = key(c[q]);
            return a < b || (a == b && p < q);
        };
        std::nth_element(order.begin() + lo, order.begin() + lo + nl, order.begin() + hi, cmp);
        if (fan > 0 && hi - lo > 65536) {
            std::thread left([=] { split(lo, lo + nl, kl, first, fan - 1); });
            split(lo + nl, hi, k - kl, first + kl, fan - 1);
            left.join();
        } else {
            split(lo, lo + nl, kl, first, 0);
            split(lo + nl, hi, k - kl, first + kl, 0);
        }
    }
};

// recursive coordinate bisection of P particles into ceil(P / target) tiles; fills order and tile_p0
inline void bisect(const std::vector<float> &px, const std::vector<float> &py, std::vector<uint32_t> &order,
                   std::vector<uint32_t> &tile_p0, uint32_t target)
{
    const uint32_t P = (uint32_t)px.size();
    order.resize(P);
    for (uint32_t i = 0; i < P; i++) order[i] = i;
    const uint32_t k = P ? (P + target - 1) / target : 0;
    tile_p0.assign((size_t)k + 1, 0);
    if (P) {
        Splitter sp{px, py, order, tile_p0, target};
        sp.split(0, P, k, 0, 4);
    }
}

} // namespace sbt

// px,py: position per particle slot; beams: per beam slot, endpoints as particle slots.
// Everything after the bisection runs per particle or per tile on a few host threads, over a particle -> incident-beams
// adjacency (r02: the serial version was 48-75 ms of a 1 M-particle upload).
inline void sb_build_tiling(SbTiling &t, const std::vector<float> &px, const std::vector<float> &py,
                            const SbHostBeams &beams, uint32_t target)
{
    const uint32_t P = (uint32_t)px.size(), B = (uint32_t)beams.size();
    target = std::max(64u, std::min(target, 16384u));
    sbt::bisect(px, py, t.order, t.tile_p0, target);
    const uint32_t T = t.ntiles = (uint32_t)t.tile_p0.size() - 1;
    sbt::uvec<uint32_t> internal_of_slot(P), tile_of(P);
    sbt::parallel_ranges(P, 1 << 16, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++) internal_of_slot[t.order[i]] = (uint32_t)i;
    });
    sbt::parallel_ranges(T, 16, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; k++)
            for (uint32_t i = t.tile_p0[k]; i < t.tile_p0[k + 1]; i++) tile_of[i] = (uint32_t)k;
    });
    // endpoints as internal indices; adjacency (filled with atomic cursors: the order inside a list is arbitrary, each
    // particle puts its OUT-beams, those with A == it, in slot order at the front of its list below)
    sbt::uvec<uint32_t> ba(B), bb(B);
    std::vector<uint32_t> adj0(P + 1, 0);
    sbt::parallel_ranges(B, 1 << 16, [&](size_t s0, size_t s1) {
        for (size_t s = s0; s < s1; s++) {
            ba[s] = internal_of_slot[beams[s].a];
            bb[s] = internal_of_slot[beams[s].b];
            __atomic_fetch_add(&adj0[ba[s] + 1], 1u, __ATOMIC_RELAXED);
            if (bb[s] != ba[s]) __atomic_fetch_add(&adj0[bb[s] + 1], 1u, __ATOMIC_RELAXED);
        }
    });
    sbt::parallel_csr_scan(adj0);
    sbt::uvec<uint32_t> adj(adj0[P]), out_deg(P);
    {
        sbt::uvec<uint32_t> cur(P);
        sbt::parallel_ranges(P, 1 << 16, [&](size_t i0, size_t i1) { std::copy(adj0.begin() + i0, adj0.begin() + i1, cur.begin() + i0); });
        sbt::parallel_ranges(B, 1 << 16, [&](size_t s0, size_t s1) {
            for (size_t s = s0; s < s1; s++) {
                adj[__atomic_fetch_add(&cur[ba[s]], 1u, __ATOMIC_RELAXED)] = (uint32_t)s;
                if (bb[s] != ba[s]) adj[__atomic_fetch_add(&cur[bb[s]], 1u, __ATOMIC_RELAXED)] = (uint32_t)s;
            }
        });
    }
    sbt::parallel_ranges(P, 1 << 14, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++) {
            uint32_t *l = adj.data() + adj0[i];
            const uint32_t n = adj0[i + 1] - adj0[i];
            uint32_t m = 0;
            for (uint32_t e = 0; e < n; e++)
                if (ba[l[e]] == i) std::swap(l[m++], l[e]);
            std::sort(l, l + m);      // out-beams in slot order: position = RANK among the beams that leave this particle
            std::sort(l + m, l + n);  // (in-beams in slot order too: the plan does not depend on thread timing)
            out_deg[i] = m;
        }
    });

    // pass 1, per tile: halo = the other endpoints of its particles' beams that live elsewhere; number of copies = every
    // beam that leaves one of its particles + every beam that arrives from another tile
    std::vector<std::vector<uint32_t>> halo(T);
    std::vector<uint32_t> ncopy(T, 0), ncut(T, 0);
    sbt::parallel_ranges(T, 16, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; k++) {
            auto &h = halo[k];
            uint32_t copies = 0, cut = 0;
            for (uint32_t p = t.tile_p0[k]; p < t.tile_p0[k + 1]; p++)
                for (uint32_t e = adj0[p]; e < adj0[p + 1]; e++) {
                    const uint32_t s = adj[e], q = ba[s] == p ? bb[s] : ba[s];
                    const bool away = tile_of[q] != k;
                    if (away) h.push_back(q);
                    if (ba[s] == p) copies++;
                    else if (away) copies++, cut++; // (a beam between two particles of this tile is A's alone)
                }
            std::sort(h.begin(), h.end());
            h.erase(std::unique(h.begin(), h.end()), h.end());
            ncopy[k] = copies;
            ncut[k] = cut;
        }
    });
    t.tile_h0.assign(T + 1, 0);
    t.tile_b0.assign(T + 1, 0);
    t.max_own = t.max_all = 0;
    t.cut_beams = 0;
    for (uint32_t k = 0; k < T; k++) {
        t.tile_h0[k + 1] = t.tile_h0[k] + (uint32_t)halo[k].size();
        t.tile_b0[k + 1] = t.tile_b0[k] + (ncopy[k] + 3) / 4 * 4;
        const uint32_t own = t.tile_p0[k + 1] - t.tile_p0[k];
        t.max_own = std::max(t.max_own, own);
        t.max_all = std::max(t.max_all, own + (uint32_t)halo[k].size());
        t.cut_beams += ncut[k];
    }
    t.halo_idx.resize(t.tile_h0[T]);

    // pass 2, per tile: its copies, ordered by the beam's RANK among the beams that leave its endpoint A (0 = A's first
    // beam in slot order, 1 = its second ...) and then by A -- not by slot.  In slot order a particle's three or four beams
    // sit next to each other, so consecutive lanes of the kernel gather the same LDS position and add to the same LDS force
    // words: 43 % of the LDS cycles of k_substep_tiled were bank conflicts (r01 counters).  By rank, consecutive lanes work
    // on consecutive particles ("all the +y beams, then all the +x beams, then the diagonals").  The integer force sums do
    // not care about the order, so the result keeps its bits.
    const uint32_t total = t.tile_b0[T];
    t.copy_la.resize(total);
    t.copy_lb.resize(total);
    t.copy_slot.resize(total);
    t.copy_of_slot.resize(B);
    sbt::parallel_ranges(T, 16, [&](size_t k0, size_t k1) {
        struct Copy { uint64_t key; uint32_t la, lb, slot, own; };
        std::vector<Copy> tmp;
        for (size_t k = k0; k < k1; k++) {
            const auto &h = halo[k];
            std::copy(h.begin(), h.end(), t.halo_idx.begin() + t.tile_h0[k]);
            const uint32_t p0 = t.tile_p0[k], n_own = t.tile_p0[k + 1] - p0;
            auto local = [&](uint32_t internal) -> uint32_t {
                if (tile_of[internal] == k) return internal - p0;
                return n_own + (uint32_t)(std::lower_bound(h.begin(), h.end(), internal) - h.begin());
            };
            tmp.clear();
            for (uint32_t p = p0; p < p0 + n_own; p++) {
                const uint32_t *l = adj.data() + adj0[p];
                const uint32_t n = adj0[p + 1] - adj0[p], m = out_deg[p];
                for (uint32_t e = 0; e < m; e++) // beams that leave p: rank e
                    tmp.push_back(Copy{((uint64_t)e << 32) | (p - p0), p - p0, local(bb[l[e]]), l[e], 1u});
                for (uint32_t e = m; e < n; e++) { // beams that arrive at p from another tile: the rank they have at their A
                    const uint32_t s = l[e], a = ba[s];
                    if (tile_of[a] == k) continue;
                    const uint32_t *la_list = adj.data() + adj0[a];
                    const uint32_t rank = (uint32_t)(std::lower_bound(la_list, la_list + out_deg[a], s) - la_list);
                    const uint32_t la = local(a);
                    tmp.push_back(Copy{((uint64_t)rank << 32) | la, la, p - p0, s, 0u});
                }
            }
            std::sort(tmp.begin(), tmp.end(), [](const Copy &x, const Copy &y) { return x.key != y.key ? x.key < y.key : x.slot < y.slot; });
            uint32_t c = t.tile_b0[k];
            for (const Copy &q : tmp) {
                t.copy_la[c] = q.la;
                t.copy_lb[c] = q.lb;
                t.copy_slot[c] = q.slot;
                if (q.own) t.copy_of_slot[q.slot] = c;
                c++;
            }
            for (; c < t.tile_b0[k + 1]; c++) t.copy_la[c] = t.copy_lb[c] = t.copy_slot[c] = 0xFFFFFFFFu; // padding to a multiple of 4
        }
    });
}
