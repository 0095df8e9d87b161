// sb_partition.cpp -- generic x-slab partition of ANY scene into per-rank scenes with ghost zones (host only).
//
// particles shard by spatial slab; each rank also holds a ghost zone `depth` beam hops deep that it
// steps redundantly and that its neighbours refresh every `depth` substeps (include/softbody.h "multi-GPU halo
// exchange").  softbody-webgpu_amd/halo.py's slab_scene GENERATES a lattice per rank; this splits a scene that
// already exists -- a loaded snapshot (engineMapping.ts:407-430), the reference's default scene (main.ts:188-253),
// anything a BufferMapper built -- in the reference's own buffer layouts, for the Python and the Node host alike.
//
//   owner(p)   = the slab p's x coordinate falls in; slabs hold equal numbers of particles (cut in x order)
//   ghosts(r)  = everything within `depth` beam hops of rank r's own particles and (contact_reach > 0) of every
//                particle whose x lies within contact_reach of the x range of r's own particles
//   local scene= own + ghost particles and every beam between two of them, data indices and slots renumbered by a
//                MONOTONE map (the collision loop's slot order and its index tie-break survive, compute.wgsl:144,153)
//   owner(beam)= owner of its endpoint A; a rank's ghost beams are refreshed (target, last) by that owner
// Both sides of an exchange list the traded records in ascending GLOBAL data index, so no list has to travel.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <new>
#include <string>
#include <memory>
#include <vector>

#include "../../include/softbody.h"

void sb_set_create_error(const char *msg); // sb_api.hip: what sb_last_error(NULL) returns

namespace {

struct Peer {
    uint32_t rank;
    std::vector<uint32_t> ghost_p, send_p, ghost_b, send_b; // local data indices
};

struct Rank {
    std::vector<uint32_t> particles;   // global particle SLOTS of the local particles, ascending
    std::vector<uint32_t> beams;       // global beam SLOTS of the local beams, ascending
    std::vector<uint32_t> p_local_of_data, b_local_of_data; // local data index per local slot position
    std::vector<uint32_t> p_global_data, b_global_data;     // per LOCAL DATA index: global data index
    std::vector<uint8_t> p_owned, b_owned;                  // per LOCAL DATA index
    uint32_t n_owned_p = 0, n_owned_b = 0;
    std::vector<Peer> peers;
};

} // namespace

struct sb_partition {
    uint32_t layout = 0, world = 0, depth = 0, P = 0, B = 0, maxP = 0, maxB = 0;
    std::vector<uint8_t> metadata, particles, beams;  // copies of the caller's records (active ones are read)
    std::vector<uint32_t> p_data_of_slot, b_data_of_slot;
    std::vector<Rank> ranks;
};

static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline float rdf(const uint8_t *p) { float v; memcpy(&v, p, 4); return v; }

#define PFAIL(code, ...)                                 \
    do {                                                 \
        char _b[400];                                    \
        snprintf(_b, sizeof _b, __VA_ARGS__);            \
        sb_set_create_error(_b);                         \
        return (code);                                   \
    } while (0)

static sb_status partition_create_impl(uint32_t layout, uint32_t maxP, uint32_t maxB, const uint8_t *md, const uint8_t *mp,
                                       const uint8_t *pd, const uint8_t *bd, uint32_t world, uint32_t depth, float reach,
                                       sb_partition **out)
{
    if (layout != SB_LAYOUT_V1 && layout != SB_LAYOUT_V2) PFAIL(SB_ERR_INVALID, "sb_partition_create: unknown layout %u", layout);
    if (!world || world > 4096) PFAIL(SB_ERR_INVALID, "sb_partition_create: world %u", world);
    if (world > 1 && depth == 0) PFAIL(SB_ERR_INVALID, "sb_partition_create: depth 0 (a ghost zone at least one beam hop deep is needed)");
    const uint32_t P = rd32(md + 4), B = rd32(md + 24);
    if (P > maxP || B > maxB) PFAIL(SB_ERR_INVALID, "sb_partition_create: counts (%u/%u) exceed capacity (%u/%u)", P, B, maxP, maxB);
    const uint32_t bstride = layout == SB_LAYOUT_V1 ? SB_BEAM_STRIDE_V1 : SB_BEAM_STRIDE_V2, isz = layout == SB_LAYOUT_V1 ? 2 : 4;
    auto map_get = [&](size_t id) -> uint32_t {
        if (isz == 2) { uint16_t v; memcpy(&v, mp + 2 * id, 2); return v; }
        return rd32(mp + 4 * id);
    };
    std::unique_ptr<sb_partition> holder(new sb_partition()); // released to the caller on success only: an exception or an
    sb_partition *pt = holder.get();                          // early return below frees it
    pt->layout = layout; pt->world = world; pt->depth = depth; pt->P = P; pt->B = B; pt->maxP = maxP; pt->maxB = maxB;
    pt->metadata.assign(md, md + SB_METADATA_BYTES);
    pt->particles.assign(pd, pd + (size_t)maxP * SB_PARTICLE_STRIDE);
    pt->beams.assign(bd, bd + (size_t)maxB * bstride);
    pt->p_data_of_slot.resize(P);
    pt->b_data_of_slot.resize(B);
    std::vector<uint32_t> slot_of_data(maxP, 0xFFFFFFFFu);
    for (uint32_t s = 0; s < P; s++) {
        const uint32_t idx = map_get(s);
        if (idx >= maxP || slot_of_data[idx] != 0xFFFFFFFFu) {
            PFAIL(SB_ERR_INVALID, "sb_partition_create: particle slot %u maps to a bad or doubly mapped data index %u", s, idx);
        }
        slot_of_data[idx] = s;
        pt->p_data_of_slot[s] = idx;
    }
    std::vector<uint32_t> ba(B), bb(B); // endpoints as particle SLOTS
    for (uint32_t s = 0; s < B; s++) {
        const uint32_t idx = map_get((size_t)maxP + s);
        if (idx >= maxB) { PFAIL(SB_ERR_INVALID, "sb_partition_create: beam slot %u maps to data index %u >= max_beams", s, idx); }
        pt->b_data_of_slot[s] = idx;
        const uint8_t *rec = pt->beams.data() + (size_t)idx * bstride;
        uint32_t a, b;
        if (layout == SB_LAYOUT_V1) { const uint32_t pr = rd32(rec); a = pr & 0xffffu; b = pr >> 16; }
        else { a = rd32(rec); b = rd32(rec + 4); }
        if (a >= maxP || b >= maxP || slot_of_data[a] == 0xFFFFFFFFu || slot_of_data[b] == 0xFFFFFFFFu) {
            PFAIL(SB_ERR_INVALID, "sb_partition_create: beam slot %u references particle data index %u/%u that no slot maps to", s, a, b);
        }
        ba[s] = slot_of_data[a];
        bb[s] = slot_of_data[b];
    }
    // owner of every particle (by slot): equal-population slabs in x order
    std::vector<float> x(P);
    for (uint32_t s = 0; s < P; s++) {
        const float v = rdf(pt->particles.data() + (size_t)pt->p_data_of_slot[s] * SB_PARTICLE_STRIDE);
        x[s] = std::isfinite(v) ? v : 0.0f;
    }
    std::vector<uint32_t> by_x(P), owner(P);
    for (uint32_t s = 0; s < P; s++) by_x[s] = s;
    std::sort(by_x.begin(), by_x.end(), [&](uint32_t p, uint32_t q) { return x[p] < x[q] || (x[p] == x[q] && pt->p_data_of_slot[p] < pt->p_data_of_slot[q]); });
    for (uint32_t k = 0; k < P; k++) owner[by_x[k]] = (uint32_t)(((uint64_t)k * world) / std::max(P, 1u));
    // adjacency by slot
    std::vector<uint32_t> adj0(P + 1, 0), adj;
    for (uint32_t s = 0; s < B; s++) { adj0[ba[s] + 1]++; if (bb[s] != ba[s]) adj0[bb[s] + 1]++; }
    for (uint32_t i = 0; i < P; i++) adj0[i + 1] += adj0[i];
    adj.resize(adj0[P]);
    { std::vector<uint32_t> cur(adj0.begin(), adj0.end() - 1);
      for (uint32_t s = 0; s < B; s++) { adj[cur[ba[s]]++] = bb[s]; if (bb[s] != ba[s]) adj[cur[bb[s]]++] = ba[s]; } }

    pt->ranks.resize(world);
    std::vector<uint32_t> stamp(P, 0xFFFFFFFFu), local_of_slot(P, 0), frontier, next;
    std::vector<std::vector<uint32_t>> local_p_of_slot(world); // per rank: local DATA index per global slot (or ~0)
    for (uint32_t r = 0; r < world; r++) {
        Rank &R = pt->ranks[r];
        frontier.clear();
        float xmin = INFINITY, xmax = -INFINITY;
        for (uint32_t s = 0; s < P; s++)
            if (owner[s] == r) { stamp[s] = r; frontier.push_back(s); xmin = std::min(xmin, x[s]); xmax = std::max(xmax, x[s]); }
        // the contact band first, then `depth` beam hops around own particles AND band: a band particle whose own beam
        // neighbours were missing would be wrong after one substep while possibly touching an owned particle
        if (reach > 0.0f && world > 1)
            for (uint32_t s = 0; s < P; s++)
                if (stamp[s] != r && x[s] >= xmin - reach && x[s] <= xmax + reach) { stamp[s] = r; frontier.push_back(s); }
        for (uint32_t d = 0; d < depth && world > 1; d++) {
            next.clear();
            for (uint32_t p : frontier)
                for (uint32_t e = adj0[p]; e < adj0[p + 1]; e++)
                    if (stamp[adj[e]] != r) { stamp[adj[e]] = r; next.push_back(adj[e]); }
            frontier.swap(next);
        }
        for (uint32_t s = 0; s < P; s++) if (stamp[s] == r) R.particles.push_back(s); // ascending slot
        // local data index = rank of the global data index among the local particles (monotone)
        std::vector<uint32_t> by_data(R.particles);
        std::sort(by_data.begin(), by_data.end(), [&](uint32_t p, uint32_t q) { return pt->p_data_of_slot[p] < pt->p_data_of_slot[q]; });
        local_p_of_slot[r].assign(P, 0xFFFFFFFFu);
        R.p_global_data.resize(by_data.size());
        R.p_owned.resize(by_data.size());
        for (uint32_t i = 0; i < by_data.size(); i++) {
            local_p_of_slot[r][by_data[i]] = i;
            R.p_global_data[i] = pt->p_data_of_slot[by_data[i]];
            R.p_owned[i] = owner[by_data[i]] == r;
            R.n_owned_p += R.p_owned[i];
        }
        R.p_local_of_data.resize(R.particles.size());
        for (uint32_t k = 0; k < R.particles.size(); k++) R.p_local_of_data[k] = local_p_of_slot[r][R.particles[k]];
        for (uint32_t s = 0; s < B; s++) if (stamp[ba[s]] == r && stamp[bb[s]] == r) R.beams.push_back(s);
        std::vector<uint32_t> bby_data(R.beams);
        std::sort(bby_data.begin(), bby_data.end(), [&](uint32_t p, uint32_t q) { return pt->b_data_of_slot[p] < pt->b_data_of_slot[q]; });
        std::vector<uint32_t> local_b_of_slot_r(B, 0xFFFFFFFFu);
        R.b_global_data.resize(bby_data.size());
        R.b_owned.resize(bby_data.size());
        for (uint32_t i = 0; i < bby_data.size(); i++) {
            local_b_of_slot_r[bby_data[i]] = i;
            R.b_global_data[i] = pt->b_data_of_slot[bby_data[i]];
            R.b_owned[i] = owner[ba[bby_data[i]]] == r;
            R.n_owned_b += R.b_owned[i];
        }
        R.b_local_of_data.resize(R.beams.size());
        for (uint32_t k = 0; k < R.beams.size(); k++) R.b_local_of_data[k] = local_b_of_slot_r[R.beams[k]];
    }
    // peers and trade lists (ascending global data index on both sides)
    for (uint32_t r = 0; r < world; r++) {
        Rank &R = pt->ranks[r];
        std::vector<std::vector<uint32_t>> gp(world), gb(world);
        for (uint32_t i = 0; i < R.p_global_data.size(); i++)
            if (!R.p_owned[i]) gp[owner[slot_of_data[R.p_global_data[i]]]].push_back(i);
        for (uint32_t k = 0; k < R.beams.size(); k++) {
            const uint32_t s = R.beams[k], o = owner[ba[s]];
            if (o != r) gb[o].push_back(R.b_local_of_data[k]);
        }
        for (uint32_t s = 0; s < world; s++) {
            std::sort(gb[s].begin(), gb[s].end()); // local data order == global data order (monotone map)
            if (gp[s].empty() && gb[s].empty()) continue;
            Peer pr;
            pr.rank = s;
            pr.ghost_p.swap(gp[s]);
            pr.ghost_b.swap(gb[s]);
            R.peers.push_back(std::move(pr));
        }
    }
    // the send side mirrors the peer's ghost side; a rank that only sends to s still lists s as a peer
    for (uint32_t r = 0; r < world; r++)
        for (const Peer &pr : pt->ranks[r].peers) {
            Rank &S = pt->ranks[pr.rank]; // pr.rank owns what r holds as ghosts
            auto it = std::find_if(S.peers.begin(), S.peers.end(), [&](const Peer &q) { return q.rank == r; });
            if (it == S.peers.end()) {
                Peer q;
                q.rank = r;
                it = S.peers.insert(std::upper_bound(S.peers.begin(), S.peers.end(), q, [](const Peer &u, const Peer &v) { return u.rank < v.rank; }), q);
            }
        }
    for (uint32_t r = 0; r < world; r++)
        for (Peer &pr : pt->ranks[r].peers) {
            const Rank &S = pt->ranks[pr.rank];
            // what pr.rank holds as ghosts owned by r: r sends those
            for (const Peer &q : S.peers) {
                if (q.rank != r) continue;
                for (uint32_t i : q.ghost_p) {
                    const uint32_t slot = slot_of_data[S.p_global_data[i]];
                    if (local_p_of_slot[r][slot] == 0xFFFFFFFFu) { PFAIL(SB_ERR_INVALID, "sb_partition_create: internal error (sent particle not local)"); }
                    pr.send_p.push_back(local_p_of_slot[r][slot]);
                }
                // beams: match by global data index
                std::vector<uint32_t> want;
                for (uint32_t i : q.ghost_b) want.push_back(S.b_global_data[i]);
                const Rank &R = pt->ranks[r];
                for (uint32_t g : want) {
                    auto it = std::lower_bound(R.b_global_data.begin(), R.b_global_data.end(), g);
                    if (it == R.b_global_data.end() || *it != g) {
                                    PFAIL(SB_ERR_INVALID, "sb_partition_create: rank %u holds a ghost beam whose owner %u does not hold it (depth %u too small?)", pr.rank, r, depth);
                    }
                    pr.send_b.push_back((uint32_t)(it - R.b_global_data.begin()));
                }
            }
        }
    for (uint32_t r = 0; r < world; r++)
        if (pt->ranks[r].peers.size() > SB_MAX_PEERS) {
            PFAIL(SB_ERR_UNSUPPORTED, "sb_partition_create: rank %u would trade with %zu ranks (at most %d): fewer, wider slabs", r, pt->ranks[r].peers.size(), SB_MAX_PEERS);
        }
    *out = holder.release();
    return SB_OK;
}

#define PGUARD(call)                                                                      \
    try { return (call); }                                                                \
    catch (const std::bad_alloc &) { sb_set_create_error("out of host memory"); return SB_ERR_OOM; } \
    catch (const std::exception &ex) { sb_set_create_error(ex.what()); return SB_ERR_INVALID; }

extern "C" {

sb_status sb_partition_create(uint32_t layout, uint32_t max_particles, uint32_t max_beams, const void *metadata, const void *mapping,
                              const void *particles, const void *beams, uint32_t world, uint32_t depth, float contact_reach,
                              sb_partition **out)
{
    if (!metadata || !mapping || !particles || (!beams && max_beams) || !out) { sb_set_create_error("sb_partition_create: null argument"); return SB_ERR_INVALID; }
    *out = nullptr;
    PGUARD(partition_create_impl(layout, max_particles, max_beams, (const uint8_t *)metadata, (const uint8_t *)mapping,
                                 (const uint8_t *)particles, (const uint8_t *)beams, world, depth, contact_reach, out))
}

sb_status sb_partition_destroy(sb_partition *p)
{
    delete p;
    return SB_OK;
}

sb_status sb_partition_rank_counts(const sb_partition *p, uint32_t rank, uint32_t counts[8])
{
    if (!p || !counts || rank >= p->world) return SB_ERR_INVALID;
    const Rank &R = p->ranks[rank];
    counts[0] = (uint32_t)R.particles.size();
    counts[1] = (uint32_t)R.beams.size();
    counts[2] = R.n_owned_p;
    counts[3] = R.n_owned_b;
    counts[4] = (uint32_t)R.peers.size();
    counts[5] = p->depth;
    counts[6] = p->P;
    counts[7] = p->B;
    return SB_OK;
}

sb_status sb_partition_layout(const sb_partition *p, uint32_t *layout)
{
    if (!p || !layout) return SB_ERR_INVALID;
    *layout = p->layout;
    return SB_OK;
}

sb_status sb_partition_rank_scene(const sb_partition *p, uint32_t rank, uint32_t max_particles, uint32_t max_beams, void *metadata,
                                  void *mapping, void *particles, void *beams)
{
    if (!p || rank >= p->world || !metadata || !mapping || !particles || (!beams && max_beams)) return SB_ERR_INVALID;
    const Rank &R = p->ranks[rank];
    const uint32_t nP = (uint32_t)R.particles.size(), nB = (uint32_t)R.beams.size();
    if (max_particles < nP || max_beams < nB || (p->layout == SB_LAYOUT_V1 && (max_particles > 65536 || max_beams > 65536))) {
        sb_set_create_error("sb_partition_rank_scene: capacities do not hold the rank's scene");
        return SB_ERR_INVALID;
    }
    const uint32_t bstride = p->layout == SB_LAYOUT_V1 ? SB_BEAM_STRIDE_V1 : SB_BEAM_STRIDE_V2, isz = p->layout == SB_LAYOUT_V1 ? 2 : 4;
    uint8_t *md = (uint8_t *)metadata, *mp = (uint8_t *)mapping, *pd = (uint8_t *)particles, *bd = (uint8_t *)beams;
    memcpy(md, p->metadata.data(), SB_METADATA_BYTES);
    memcpy(md + 4, &nP, 4);
    memcpy(md + 24, &nB, 4);
    memcpy(md + 40, &max_particles, 4);
    memcpy(md + 44, &max_beams, 4);
    auto map_set = [&](size_t id, uint32_t v) {
        if (isz == 2) { const uint16_t h = (uint16_t)v; memcpy(mp + 2 * id, &h, 2); }
        else memcpy(mp + 4 * id, &v, 4);
    };
    // identity beyond the active slots, as BufferMapper.writeState leaves it (engineMapping.ts:505-516)
    for (uint32_t s = 0; s < max_particles; s++) map_set(s, s);
    for (uint32_t s = 0; s < max_beams; s++) map_set((size_t)max_particles + s, s);
    for (uint32_t k = 0; k < nP; k++) map_set(k, R.p_local_of_data[k]);
    for (uint32_t k = 0; k < nB; k++) map_set((size_t)max_particles + k, R.b_local_of_data[k]);
    memset(pd, 0, (size_t)max_particles * SB_PARTICLE_STRIDE);
    if (max_beams) memset(bd, 0, (size_t)max_beams * bstride);
    for (uint32_t i = 0; i < nP; i++)
        memcpy(pd + (size_t)i * SB_PARTICLE_STRIDE, p->particles.data() + (size_t)R.p_global_data[i] * SB_PARTICLE_STRIDE, SB_PARTICLE_STRIDE);
    // global particle data index -> local data index, for the endpoints
    std::vector<uint32_t> local_of_global(p->maxP, 0xFFFFFFFFu);
    for (uint32_t i = 0; i < nP; i++) local_of_global[R.p_global_data[i]] = i;
    for (uint32_t i = 0; i < nB; i++) {
        const uint8_t *src = p->beams.data() + (size_t)R.b_global_data[i] * bstride;
        uint8_t *dst = bd + (size_t)i * bstride;
        memcpy(dst, src, bstride);
        if (p->layout == SB_LAYOUT_V1) {
            const uint32_t pr = rd32(src), a = local_of_global[pr & 0xffffu], b = local_of_global[pr >> 16];
            const uint32_t w = (a & 0xffffu) | (b << 16);
            memcpy(dst, &w, 4);
        } else {
            const uint32_t a = local_of_global[rd32(src)], b = local_of_global[rd32(src + 4)];
            memcpy(dst, &a, 4);
            memcpy(dst + 4, &b, 4);
        }
    }
    return SB_OK;
}

sb_status sb_partition_rank_ids(const sb_partition *p, uint32_t rank, uint32_t *particle_global, uint8_t *particle_owned,
                                uint32_t *beam_global, uint8_t *beam_owned)
{
    if (!p || rank >= p->world) return SB_ERR_INVALID;
    const Rank &R = p->ranks[rank];
    if (particle_global) std::copy(R.p_global_data.begin(), R.p_global_data.end(), particle_global);
    if (particle_owned) std::copy(R.p_owned.begin(), R.p_owned.end(), particle_owned);
    if (beam_global) std::copy(R.b_global_data.begin(), R.b_global_data.end(), beam_global);
    if (beam_owned) std::copy(R.b_owned.begin(), R.b_owned.end(), beam_owned);
    return SB_OK;
}

sb_status sb_partition_peer_counts(const sb_partition *p, uint32_t rank, uint32_t j, uint32_t *peer_rank, uint32_t counts[4])
{
    if (!p || rank >= p->world || !peer_rank || !counts || j >= p->ranks[rank].peers.size()) return SB_ERR_INVALID;
    const Peer &q = p->ranks[rank].peers[j];
    *peer_rank = q.rank;
    counts[0] = (uint32_t)q.ghost_p.size();
    counts[1] = (uint32_t)q.send_p.size();
    counts[2] = (uint32_t)q.ghost_b.size();
    counts[3] = (uint32_t)q.send_b.size();
    return SB_OK;
}

sb_status sb_partition_peer_lists(const sb_partition *p, uint32_t rank, uint32_t j, uint32_t *ghost_p, uint32_t *send_p,
                                  uint32_t *ghost_b, uint32_t *send_b)
{
    if (!p || rank >= p->world || j >= p->ranks[rank].peers.size()) return SB_ERR_INVALID;
    const Peer &q = p->ranks[rank].peers[j];
    if (ghost_p) std::copy(q.ghost_p.begin(), q.ghost_p.end(), ghost_p);
    if (send_p) std::copy(q.send_p.begin(), q.send_p.end(), send_p);
    if (ghost_b) std::copy(q.ghost_b.begin(), q.ghost_b.end(), ghost_b);
    if (send_b) std::copy(q.send_b.begin(), q.send_b.end(), send_b);
    return SB_OK;
}

} // extern "C"
