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
inline void sb_build_tiling(SbTiling &t, const std::vector<float> &px, const std::vector<float> &py,
                            const SbHostBeams &beams, uint32_t target)
{
    const uint32_t P = (uint32_t)px.size(), B = (uint32_t)beams.size();
    target = std::max(64u, std::min(target, 16384u));
    sbt::bisect(px, py, t.order, t.tile_p0, target);
    t.ntiles = (uint32_t)t.tile_p0.size() - 1;
    std::vector<uint32_t> internal_of_slot(P), tile_of(P);
    for (uint32_t i = 0; i < P; i++) internal_of_slot[t.order[i]] = i;
    for (uint32_t k = 0; k < t.ntiles; k++)
        for (uint32_t i = t.tile_p0[k]; i < t.tile_p0[k + 1]; i++) tile_of[i] = k;

    // pass 1: count copies and collect halo candidates per tile
    std::vector<uint32_t> ncopy(t.ntiles + 1, 0);
    std::vector<std::vector<uint32_t>> halo(t.ntiles);
    t.cut_beams = 0;
    for (uint32_t s = 0; s < B; s++) {
        uint32_t ia = internal_of_slot[beams[s].a], ib = internal_of_slot[beams[s].b];
        uint32_t ta = tile_of[ia], tb = tile_of[ib];
        ncopy[ta]++;
        if (tb != ta) {
            ncopy[tb]++;
            halo[ta].push_back(ib);
            halo[tb].push_back(ia);
            t.cut_beams++;
        }
    }
    t.tile_h0.assign(t.ntiles + 1, 0);
    t.tile_b0.assign(t.ntiles + 1, 0);
    t.max_own = t.max_all = 0;
    for (uint32_t k = 0; k < t.ntiles; k++) {
        auto &h = halo[k];
        std::sort(h.begin(), h.end());
        h.erase(std::unique(h.begin(), h.end()), h.end());
        t.tile_h0[k + 1] = t.tile_h0[k] + (uint32_t)h.size();
        t.tile_b0[k + 1] = t.tile_b0[k] + (ncopy[k] + 3) / 4 * 4;
        uint32_t own = t.tile_p0[k + 1] - t.tile_p0[k];
        t.max_own = std::max(t.max_own, own);
        t.max_all = std::max(t.max_all, own + (uint32_t)h.size());
    }
    t.halo_idx.resize(t.tile_h0[t.ntiles]);
    for (uint32_t k = 0; k < t.ntiles; k++) std::copy(halo[k].begin(), halo[k].end(), t.halo_idx.begin() + t.tile_h0[k]);

    // pass 2: the copies of every tile, ordered by the beam's RANK among the beams that leave its endpoint A (0 = A's first
    // beam in slot order, 1 = its second ...) and then by A -- not by slot.  In slot order a particle's three or four beams
    // sit next to each other, so consecutive lanes of the kernel gather the same LDS position and add to the same LDS force
    // words: 43 % of the LDS cycles of k_substep_tiled were bank conflicts (r01 counters).  By rank, consecutive lanes work
    // on consecutive particles ("all the +y beams, then all the +x beams, then the diagonals").  The integer force sums do
    // not care about the order, so the result keeps its bits.
    const uint32_t total = t.tile_b0[t.ntiles];
    struct Copy { uint64_t key; uint32_t la, lb, slot, own; };
    sbt::uvec<Copy> tmp(total);
    std::vector<uint32_t> cursor(t.tile_b0.begin(), t.tile_b0.end() - 1), out_cnt(P, 0);
    auto local = [&](uint32_t tile, uint32_t internal) -> uint32_t {
        if (tile_of[internal] == tile) return internal - t.tile_p0[tile];
        const auto &h = halo[tile];
        uint32_t pos = (uint32_t)(std::lower_bound(h.begin(), h.end(), internal) - h.begin());
        return (t.tile_p0[tile + 1] - t.tile_p0[tile]) + pos;
    };
    for (uint32_t s = 0; s < B; s++) {
        const uint32_t ia = internal_of_slot[beams[s].a], ib = internal_of_slot[beams[s].b];
        const uint32_t ta = tile_of[ia], tb = tile_of[ib];
        const uint64_t rank = out_cnt[ia]++;
        const uint32_t la = local(ta, ia);
        tmp[cursor[ta]++] = Copy{(rank << 32) | la, la, local(ta, ib), s, 1u};
        if (tb != ta) {
            const uint32_t lb_a = local(tb, ia);
            tmp[cursor[tb]++] = Copy{(rank << 32) | lb_a, lb_a, local(tb, ib), s, 0u};
        }
    }
    t.copy_la.resize(total);
    t.copy_lb.resize(total);
    t.copy_slot.resize(total);
    t.copy_of_slot.assign(B, 0);
    sbt::parallel_ranges(t.ntiles, 16, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; k++) {
            const uint32_t c0 = t.tile_b0[k], c1 = cursor[k], c2 = t.tile_b0[k + 1];
            std::sort(tmp.begin() + c0, tmp.begin() + c1,
                      [](const Copy &x, const Copy &y) { return x.key != y.key ? x.key < y.key : x.slot < y.slot; });
            for (uint32_t c = c0; c < c1; c++) {
                t.copy_la[c] = tmp[c].la;
                t.copy_lb[c] = tmp[c].lb;
                t.copy_slot[c] = tmp[c].slot;
                if (tmp[c].own) t.copy_of_slot[tmp[c].slot] = c;
            }
            for (uint32_t c = c1; c < c2; c++) t.copy_la[c] = t.copy_lb[c] = t.copy_slot[c] = 0xFFFFFFFFu; // padding to a multiple of 4
        }
    });
}
