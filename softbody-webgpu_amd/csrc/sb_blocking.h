// sb_blocking.h -- host-side plan for the TEMPORALLY BLOCKED substep kernel (k_substep_blocked).
//
// One launch advances every tile by K substeps out of LDS and registers.  That is possible because a
// substep moves information exactly one beam hop (compute.wgsl:103-130: a beam reads its two endpoints;
// :183-187: a particle reads the forces of its own beams), so the state of a tile's own particles after K
// substeps depends only on the particles within K hops of the tile and on the beams between them:
//   ring(p)   = beam-graph distance of particle p from the tile's own particles (0 = own), for ring <= K
//   region    = own particles, then the halo sorted by ring
//   entries   = every beam with an endpoint of ring <= K-1, sorted by the smaller ring of its endpoints
// In substep s of a launch of k substeps (s = 1..k) the tile evaluates the entries whose smaller ring is
// <= k-s and integrates the particles of ring <= k-s: exactly the part of the region whose inputs are
// still the true state.  Everything is redundant, deterministic arithmetic on the same inputs the owner
// uses, so the result is bit-identical to k single substeps.  Per launch a tile reads its region once
// and writes its own particles and its own beams once: HBM traffic and launch boundaries per substep drop
// by about K x (2/3 after the redundant ring work and the halo gathers are paid).
//
// Beam state is NOT duplicated here (the single-substep tiling keeps a private copy per tile): each beam
// has one owner (the tile of its endpoint A) and one slot g in the state arrays, which are double
// buffered like the particles (other tiles read state t while the owner writes state t+k).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "sb_tiling.h"

struct SbBlocking {
    uint32_t ntiles = 0, K = 0;
    std::vector<uint32_t> order;     // internal particle -> slot (tiles are contiguous ranges)
    std::vector<uint32_t> tile_p0;   // [T+1] own particles
    std::vector<uint32_t> tile_h0;   // [T+1] halo ranges
    sbt::uvec<uint32_t> halo_idx;  // internal particle index of each halo entry (sorted by ring, then index)
    sbt::uvec<uint32_t> ring_cnt;  // [T][K+1] region particles with ring <= r (ring_cnt[t][0] = own)
    std::vector<uint32_t> tile_b0;   // [T+1] owned beams = ranges of the state arrays
    sbt::uvec<uint32_t> beam_slot; // [NB] mapping slot of state index g
    sbt::uvec<uint32_t> g_of_slot; // [B]
    std::vector<uint32_t> tile_e0;   // [T+1] entry ranges
    sbt::uvec<uint32_t> ent_la, ent_lb, ent_slot; // region-local endpoint indices, beam slot
    sbt::uvec<uint32_t> lvl_cnt;   // [T][K] entries whose smaller ring is <= m (m = 0..K-1)
    std::vector<uint32_t> tile_s0;   // [T+1] ranges of ent_state (entries past the tile's own beams)
    sbt::uvec<uint32_t> ent_state; // state index g of each non-owned entry
    sbt::uvec<uint32_t> slot_e0;     // [B+1] CSR beam slot -> absolute entry indices (delete pass)
    sbt::uvec<uint32_t> slot_ent;
    std::vector<uint32_t> tile_n0;   // [T+1] CSR tile -> tiles that own some of its halo particles
    sbt::uvec<uint32_t> tile_nb;
    uint32_t max_region = 0, max_entries = 0, max_own = 0;
    // per launch depth k = 1..K (index k; [0] unused): what a launch of k substeps needs and loads -- the largest region
    // (particles of ring <= k) and entry prefix (smaller ring <= k-1) of any tile, and their totals over the tiles
    std::vector<uint32_t> region_at, entries_at;
    std::vector<uint64_t> sum_region_at, sum_entries_at;
    // ... and by class, what the kernel's slot classes must hold: the largest halo (ring 1..k) and the most halo entries
    // (entries past the tile's own beams) of any tile at depth k; the most beams any tile owns
    std::vector<uint32_t> halo_at, halo_entries_at;
    uint32_t max_ownb = 0;
};

namespace sbt {

template <typename F>
inline void parallel_tiles(uint32_t n, F f)
{
    unsigned hw = std::thread::hardware_concurrency();
    const uint32_t nt = std::max(1u, std::min<uint32_t>(hw ? hw : 4u, std::min(16u, (n + 31) / 32)));
    if (nt <= 1) {
        f(0u, 0u, n);
        return;
    }
    std::vector<std::thread> th;
    for (uint32_t w = 0; w < nt; w++)
        th.emplace_back([=, &f] { f(w, (uint32_t)((uint64_t)n * w / nt), (uint32_t)((uint64_t)n * (w + 1) / nt)); });
    for (auto &t : th) t.join();
}

} // namespace sbt

// px,py: position per particle slot; beams: per beam slot, endpoints as particle slots.
inline void sb_build_blocking(SbBlocking &t, const std::vector<float> &px, const std::vector<float> &py,
                              const SbHostBeams &beams, uint32_t target, uint32_t K)
{
    const uint32_t P = (uint32_t)px.size(), B = (uint32_t)beams.size();
    const bool timing = getenv("SB_UPLOAD_TIMING") != nullptr;
    auto tick = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[sb blocking] %-26s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tick).count());
        tick = now;
    };
    target = std::max(64u, std::min(target, 16384u));
    t.K = K;
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

    mark("bisection");
    // beam endpoints as internal indices; adjacency (particle -> incident beam slots)
    // (filled by several threads with atomic cursors: the order inside a particle's list is arbitrary, and nothing below
    // depends on it -- out-lists, frontiers and entry lists are sorted after they are collected)
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
    sbt::uvec<uint32_t> adj(adj0[P]);
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
    mark("adjacency");
    // Order of the beams inside a tile, for the owned state arrays and for every entry list: by the beam's RANK among
    // the beams that leave its endpoint A (0 = A's first beam in slot order, 1 = its second ...), then by A.  Consecutive
    // lanes of the kernel then work on consecutive particles -- in a lattice "all the +y beams, then all the +x beams,
    // then the diagonals" -- so the endpoint gathers and the integer force adds of a wave fall into distinct LDS banks
    // instead of three or four lanes hitting the same particle (r02: 48 % of the LDS cycles were bank conflicts with
    // the entries in slot order).
    // Per particle (threads over particle ranges): its out-beams are the entries of its adjacency list with A == it, put
    // in slot order in place at the front of the list; the rank of a beam is its position there.
    sbt::uvec<uint32_t> rank(B), out_deg(P);
    sbt::parallel_ranges(P, 1 << 14, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++) {
            uint32_t *l = adj.data() + adj0[i];
            const uint32_t n = adj0[i + 1] - adj0[i];
            uint32_t m = 0;
            for (uint32_t e = 0; e < n; e++)
                if (ba[l[e]] == i) std::swap(l[m++], l[e]);
            std::sort(l, l + m);
            for (uint32_t e = 0; e < m; e++) rank[l[e]] = e;
            out_deg[i] = m;
        }
    });
    // owned beams: tile of endpoint A.  Each tile collects, sorts and numbers its own (threads over tiles).
    t.tile_b0.assign(T + 1, 0);
    sbt::parallel_ranges(T, 16, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; k++) {
            uint32_t n = 0;
            for (uint32_t i = t.tile_p0[k]; i < t.tile_p0[k + 1]; i++) n += out_deg[i];
            t.tile_b0[k + 1] = n;
        }
    });
    for (uint32_t k = 0; k < T; k++) t.tile_b0[k + 1] += t.tile_b0[k];
    t.beam_slot.resize(B);
    t.g_of_slot.resize(B);
    {
        // (keys travel with the elements: a comparator that looks rank[] and ba[] up by slot misses the cache on every compare)
        struct Key { uint64_t rank_a; uint32_t slot; };
        sbt::uvec<Key> keyed(B);
        sbt::parallel_ranges(T, 16, [&](size_t k0, size_t k1) {
            for (size_t k = k0; k < k1; k++) {
                uint32_t g = t.tile_b0[k];
                for (uint32_t i = t.tile_p0[k]; i < t.tile_p0[k + 1]; i++)
                    for (uint32_t e = 0; e < out_deg[i]; e++) keyed[g++] = Key{((uint64_t)e << 32) | i, adj[adj0[i] + e]};
                std::sort(keyed.begin() + t.tile_b0[k], keyed.begin() + t.tile_b0[k + 1],
                          [](const Key &x, const Key &y) { return x.rank_a != y.rank_a ? x.rank_a < y.rank_a : x.slot < y.slot; });
                for (uint32_t q = t.tile_b0[k]; q < t.tile_b0[k + 1]; q++) {
                    t.beam_slot[q] = keyed[q].slot;
                    t.g_of_slot[keyed[q].slot] = q;
                }
            }
        });
    }

    mark("owned beams");
    // per tile: rings, region, entries (tiles are independent: a few host threads)
    struct TileOut {
        std::vector<uint32_t> halo, ring_cnt, la, lb, slot, lvl_cnt, state, nb;
    };
    std::vector<TileOut> out(T);
    sbt::parallel_tiles(T, [&](uint32_t, uint32_t k0, uint32_t k1) {
        // what a tile knows about the particles of its region: local index and ring.  A small open-addressing table per
        // worker, emptied through its own list of used cells after every tile (three scene-sized arrays per worker, filled
        // before the first tile, were 9 bytes x P x 16 workers: 2.3 GB of host memory at 16 M particles)
        struct RegionMap {
            std::vector<uint32_t> key, local, used;
            std::vector<uint8_t> ring;
            uint32_t mask = 0;
            explicit RegionMap(uint32_t cap = 1u << 13) { resize(cap); }
            void resize(uint32_t cap)
            {
                key.assign(cap, 0xFFFFFFFFu);
                local.assign(cap, 0u);
                ring.assign(cap, 0);
                mask = cap - 1u;
            }
            static uint32_t hash(uint32_t q) { return q * 2654435761u; }
            // cell of q, or of the empty cell where q would go
            uint32_t cell(uint32_t q) const
            {
                uint32_t c = (hash(q) >> 7) & mask;
                while (key[c] != q && key[c] != 0xFFFFFFFFu) c = (c + 1u) & mask;
                return c;
            }
            bool has(uint32_t q) const { return key[cell(q)] == q; }
            void put(uint32_t q, uint32_t l, uint8_t r)
            {
                if ((used.size() + 1) * 2 > key.size()) { // half full: double
                    std::vector<uint32_t> k2, l2;
                    std::vector<uint8_t> r2;
                    for (uint32_t c : used) {
                        k2.push_back(key[c]);
                        l2.push_back(local[c]);
                        r2.push_back(ring[c]);
                    }
                    resize((uint32_t)key.size() * 2u);
                    used.clear();
                    for (size_t j = 0; j < k2.size(); j++) put(k2[j], l2[j], r2[j]);
                }
                const uint32_t c = cell(q);
                key[c] = q;
                local[c] = l;
                ring[c] = r;
                used.push_back(c);
            }
            void clear()
            {
                for (uint32_t c : used) key[c] = 0xFFFFFFFFu;
                used.clear();
            }
        } reg;
        std::vector<uint32_t> frontier, next;
        struct Ent { uint64_t key; uint32_t a, slot, m; }; // key = (smaller ring, not-owned, rank): the sort order, carried along
        std::vector<Ent> ents;
        for (uint32_t k = k0; k < k1; k++) {
            TileOut &o = out[k];
            const uint32_t p0 = t.tile_p0[k], p1 = t.tile_p0[k + 1], n_own = p1 - p0;
            reg.clear();
            frontier.clear();
            for (uint32_t i = p0; i < p1; i++) frontier.push_back(i); // (own particles are told by their range, not by the table)
            auto in_region = [&](uint32_t q) { return (q >= p0 && q < p1) || reg.has(q); };
            auto ring_of = [&](uint32_t q) -> uint32_t { return (q >= p0 && q < p1) ? 0u : reg.ring[reg.cell(q)]; };
            auto local_of = [&](uint32_t q) -> uint32_t { return (q >= p0 && q < p1) ? q - p0 : reg.local[reg.cell(q)]; };
            o.ring_cnt.assign(K + 1, n_own);
            o.halo.clear();
            for (uint32_t r = 1; r <= K; r++) {
                next.clear();
                for (uint32_t p : frontier)
                    for (uint32_t e = adj0[p]; e < adj0[p + 1]; e++) {
                        const uint32_t s = adj[e], q = ba[s] == p ? bb[s] : ba[s];
                        if (!in_region(q)) {
                            reg.put(q, 0u, (uint8_t)r);
                            next.push_back(q);
                        }
                    }
                std::sort(next.begin(), next.end());
                for (uint32_t q : next) {
                    reg.local[reg.cell(q)] = n_own + (uint32_t)o.halo.size();
                    o.halo.push_back(q);
                }
                o.ring_cnt[r] = n_own + (uint32_t)o.halo.size();
                frontier.swap(next);
            }
            // entries: every beam with an endpoint of ring <= K-1, once
            ents.clear();
            auto visit = [&](uint32_t p) {
                if (ring_of(p) > K - 1) return;
                for (uint32_t e = adj0[p]; e < adj0[p + 1]; e++) {
                    const uint32_t s = adj[e];
                    const uint32_t a = ba[s], b = bb[s];
                    // the beam is emitted by endpoint A, unless A lies outside the rings that emit (then by B)
                    const bool a_emits = in_region(a) && ring_of(a) <= K - 1;
                    if (p == a ? true : !a_emits) {
                        if (p != a && p != b) continue;
                        const uint32_t m = std::min<uint32_t>(ring_of(a), ring_of(b));
                        ents.push_back(Ent{((uint64_t)m << 40) | ((uint64_t)(tile_of[a] == k ? 0u : 1u) << 32) | rank[s], a, s, m});
                    }
                }
            };
            for (uint32_t i = p0; i < p1; i++) visit(i);
            for (uint32_t q : o.halo) visit(q);
            std::sort(ents.begin(), ents.end(), [](const Ent &x, const Ent &y) {
                if (x.key != y.key) return x.key < y.key;
                if (x.a != y.a) return x.a < y.a;
                return x.slot < y.slot;
            });
            const size_t n = ents.size();
            o.la.resize(n);
            o.lb.resize(n);
            o.slot.resize(n);
            o.state.clear();
            o.lvl_cnt.assign(K, 0);
            for (size_t j = 0; j < n; j++) {
                const uint32_t s = ents[j].slot;
                o.la[j] = local_of(ba[s]);
                o.lb[j] = local_of(bb[s]);
                o.slot[j] = s;
                if ((ents[j].key >> 32) & 1u) o.state.push_back(t.g_of_slot[s]);
                for (uint32_t m = ents[j].m; m < K; m++) o.lvl_cnt[m]++;
            }
            // tiles that own my halo particles (their acceleration flags decide whether halo a's are read)
            o.nb.clear();
            for (uint32_t q : o.halo) o.nb.push_back(tile_of[q]);
            std::sort(o.nb.begin(), o.nb.end());
            o.nb.erase(std::unique(o.nb.begin(), o.nb.end()), o.nb.end());
        }
    });

    mark("rings + entries per tile");
    t.tile_h0.assign(T + 1, 0);
    t.tile_e0.assign(T + 1, 0);
    t.tile_s0.assign(T + 1, 0);
    t.tile_n0.assign(T + 1, 0);
    t.max_region = t.max_entries = t.max_own = 0;
    for (uint32_t k = 0; k < T; k++) {
        t.tile_h0[k + 1] = t.tile_h0[k] + (uint32_t)out[k].halo.size();
        t.tile_e0[k + 1] = t.tile_e0[k] + (uint32_t)out[k].la.size();
        t.tile_s0[k + 1] = t.tile_s0[k] + (uint32_t)out[k].state.size();
        t.tile_n0[k + 1] = t.tile_n0[k] + (uint32_t)out[k].nb.size();
        const uint32_t own = t.tile_p0[k + 1] - t.tile_p0[k];
        t.max_own = std::max(t.max_own, own);
        t.max_region = std::max(t.max_region, own + (uint32_t)out[k].halo.size());
        t.max_entries = std::max(t.max_entries, (uint32_t)out[k].la.size());
    }
    t.region_at.assign(K + 1, 0);
    t.entries_at.assign(K + 1, 0);
    t.halo_at.assign(K + 1, 0);
    t.halo_entries_at.assign(K + 1, 0);
    t.max_ownb = 0;
    t.sum_region_at.assign(K + 1, 0);
    t.sum_entries_at.assign(K + 1, 0);
    for (uint32_t k = 0; k < T; k++)
        for (uint32_t d = 1; d <= K; d++) {
            t.region_at[d] = std::max(t.region_at[d], out[k].ring_cnt[d]);
            t.entries_at[d] = std::max(t.entries_at[d], out[k].lvl_cnt[d - 1]);
            t.sum_region_at[d] += out[k].ring_cnt[d];
            t.sum_entries_at[d] += out[k].lvl_cnt[d - 1];
            const uint32_t own = t.tile_p0[k + 1] - t.tile_p0[k], ownb = t.tile_b0[k + 1] - t.tile_b0[k];
            t.halo_at[d] = std::max(t.halo_at[d], out[k].ring_cnt[d] - own);
            t.halo_entries_at[d] = std::max(t.halo_entries_at[d], out[k].lvl_cnt[d - 1] - std::min(out[k].lvl_cnt[d - 1], ownb));
            t.max_ownb = std::max(t.max_ownb, ownb);
        }
    t.halo_idx.resize(t.tile_h0[T]);
    t.ring_cnt.resize((size_t)T * (K + 1));
    t.lvl_cnt.resize((size_t)T * K);
    const uint32_t E = t.tile_e0[T];
    t.ent_la.resize(E);
    t.ent_lb.resize(E);
    t.ent_slot.resize(E);
    t.ent_state.resize(t.tile_s0[T]);
    t.tile_nb.resize(t.tile_n0[T]);
    t.slot_e0.resize(B + 1);
    sbt::parallel_ranges(B + 1, 1 << 16, [&](size_t s0, size_t s1) { std::fill(t.slot_e0.begin() + s0, t.slot_e0.begin() + s1, 0u); });
    sbt::parallel_ranges(T, 16, [&](size_t k0, size_t k1) {
        for (size_t k = k0; k < k1; k++) {
            const TileOut &o = out[k];
            std::copy(o.halo.begin(), o.halo.end(), t.halo_idx.begin() + t.tile_h0[k]);
            std::copy(o.ring_cnt.begin(), o.ring_cnt.end(), t.ring_cnt.begin() + (size_t)k * (K + 1));
            std::copy(o.lvl_cnt.begin(), o.lvl_cnt.end(), t.lvl_cnt.begin() + (size_t)k * K);
            std::copy(o.la.begin(), o.la.end(), t.ent_la.begin() + t.tile_e0[k]);
            std::copy(o.lb.begin(), o.lb.end(), t.ent_lb.begin() + t.tile_e0[k]);
            std::copy(o.slot.begin(), o.slot.end(), t.ent_slot.begin() + t.tile_e0[k]);
            std::copy(o.state.begin(), o.state.end(), t.ent_state.begin() + t.tile_s0[k]);
            std::copy(o.nb.begin(), o.nb.end(), t.tile_nb.begin() + t.tile_n0[k]);
        }
    });
    sbt::parallel_ranges(E, 1 << 16, [&](size_t e0, size_t e1) {
        for (size_t e = e0; e < e1; e++) __atomic_fetch_add(&t.slot_e0[t.ent_slot[e] + 1], 1u, __ATOMIC_RELAXED);
    });
    sbt::parallel_csr_scan(t.slot_e0);
    t.slot_ent.resize(E);
    {
        sbt::uvec<uint32_t> cur(B);
        sbt::parallel_ranges(B, 1 << 16, [&](size_t s0, size_t s1) { std::copy(t.slot_e0.begin() + s0, t.slot_e0.begin() + s1, cur.begin() + s0); });
        sbt::parallel_ranges(E, 1 << 16, [&](size_t e0, size_t e1) {
            for (size_t e = e0; e < e1; e++) t.slot_ent[__atomic_fetch_add(&cur[t.ent_slot[e]], 1u, __ATOMIC_RELAXED)] = (uint32_t)e;
        });
    }
    mark("merge + slot->entries");
}
