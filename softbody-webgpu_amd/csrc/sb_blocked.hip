// sb_blocked.hip -- k_substep_blocked: K substeps per launch out of LDS and registers (gfx950, wave64).
//
// The plan (which particles and beams a tile needs, ring by ring) is sb_blocking.h's; the arithmetic is
// sb_physics.h's, the same functions the single-substep kernels call.  One 512-thread workgroup per tile:
//   load    region particles (own: coalesced; halo: gathered) -> positions in LDS, velocities and
//           accelerations in registers; the tile's entry words and beam states -> registers; materials -> LDS
//   k x     beam phase: every thread walks its entries up to this substep's prefix (compute.wgsl:103-130),
//                       integer force sums by ds_add, new target/last stay in registers
//           particle phase: consume and clear the sums (compute.wgsl:171-201), new position back to LDS
//   store   own particles -> WRITE buffers, own beams -> WRITE state arrays (strain/stress: last substep of a call)
// Between launches nothing but the true state lives in HBM, so uploads, read-backs, the delete pass and the ghost
// refresh of the multi-GPU path see what they always saw.  Collisions are not blocked (their reach is spatial,
// not along beams): SB_COLLIDE_GRID keeps the single-substep kernels of sb_kernels.hip.
#include "sb_engine.h"

#ifndef SB_BK_WAVES
#define SB_BK_WAVES 4 // waves per SIMD the register budget is sized for (2 tiles per CU at 512 threads per tile)
#endif
#ifndef SB_BK_G
#define SB_BK_G 2      // entries evaluated side by side (independent dependency chains in one instruction stream)
#endif

SB_DEV uint32_t sbb_tile_of_block(uint32_t b, uint32_t n)
{
    uint32_t q = n >> 3, r = n & 7, x = b & 7, k = b >> 3; // contiguous runs of tiles per XCD (sb_kernels.hip)
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

struct SbBlockedPlan {
    const uint32_t *tile_p0, *tile_h0, *halo_idx, *ring_cnt, *tile_b0, *tile_e0, *tile_s0, *ent_word, *ent_state, *lvl_cnt,
        *tile_n0, *tile_nb;
    const float *ent_length; // MAT == 1: rest length per entry
    const float *mat_tab;    // [nmat][SB_MAT_ROW]
    uint32_t ntiles, K, cap, nmat, dummy_word;
};

struct SbBlockedState {
    const float *target_r, *last_r;
    float *target_w, *last_w, *strain, *stress;
    uint32_t *broken;
    // "plastic" flags, per tile and per state buffer (like the acceleration flags): 0 = every beam the tile owns still has
    // target_length == length, bit for bit (compute.wgsl:113-116 has never fired for them), in BOTH state buffers -- such a
    // tile neither reads nor writes its targets, and tiles around it do not gather them: 33 of a launch's 147 MB on a
    // scene that has not yielded (BASELINE config 2).  Set for good when a beam of the tile yields (that launch stores the
    // tile's targets, every later one reads and stores them), by an upload that holds yielded beams, by a ghost refresh.
    const uint32_t *plastic_r;
    uint32_t *plastic_w;
};

#define SB_BK_CAP (SB_BK_MAXP * SB_BK_T + 2u + 128u) // LDS records: a full region, the two dummy endpoints, 64 more dummy pairs
#define SB_BK_PAD0 (SB_BK_MAXP * SB_BK_T + 2u)        // ... one pair per lane, for PADDING slots (a slot of a class the tile
// does not fill): were they all the one dummy beam, a wave of padding would send 64 lanes' force sums to the same two LDS
// words -- same-address atomics retire one lane at a time and hold the LDS pipeline meanwhile for every wave of the CU (first
// build with slot classes, r03: +4 us per substep from 142 padding entries per tile).  Entries removed by the delete pass
// stay the one dummy beam: they are few.
// A thread's particle and entry slots come in two CLASSES with registers of their own: the first SB_BK_OWNP particle slots
// hold particles the tile owns (q = tid + i T), the others its halo (q = n_own + tid + (i - OWNP) T); the first SB_BK_OWNB
// entry slots hold beams the tile owns, the others the halo entries.  Until r03 one slot could hold either, so the gathers of
// the second round of loads wrote registers the first round's loads were still in flight for -- and the compiler put a full
// s_waitcnt vmcnt(0) in front of EVERY gather (vmcnt retires in order): some fifteen serialised memory round trips per tile
// where two were meant (profiles/r03_ablation.txt: 42 of a launch's 78 us were load and store phases).
#define SB_LANDED(x) asm volatile("" : "+v"(x)) /* a use: the compiler's wait for x goes HERE (inside a rare branch, not at its join) */
#define SB_BK_DUMMY_A (SB_BK_MAXP * SB_BK_T)      // (what the host packs into dummy_word, sb_api.hip upload_blocked)
#define SB_BK_DUMMY_B (SB_BK_MAXP * SB_BK_T + 1u)
#define SB_BK_ROW 8u // LDS material row: length, 1/length, spring, damp, yield, yield*length, length*limit, limit

#ifndef SB_BK_WAVES_AUX
#define SB_BK_WAVES_AUX SB_BK_WAVES
#endif
// TRACK (SB_COLLIDE_GRID engines, sb_api.hip hybrid_step): the launch runs while every neighbour list of the spatial hash is
// empty, i.e. while the collision loop of compute.wgsl:142-170 is a no-op -- which holds as long as no particle has moved
// more than the hash's skin relative to the common drift since the lists were made.  So the kernel measures what its own
// particles move, substep by substep, against the drift (cx, cy) and leaves the largest sum any of its threads saw in
// dmax[tile] (every particle is integrated by one thread throughout, so a thread's sum of per-substep maxima bounds the path
// of each of its particles); k_hybrid_validate adds the launch's maximum to the hash's running bound and raises *bad when
// that exceeds the skin -- the launches queued behind then return at once and the host redoes the block substep by substep.
struct SbTrack {
    // r04: a tracked launch VALIDATES THE LAUNCH BEFORE IT in its own prologue (the separate k_hybrid_validate launch behind every
    // tracked launch cost 4.6 us per six substeps; only the last launch of a run still gets one).  What a launch leaves for its
    // successor: 64 slots {largest displacement sum by atomic max, the sample (dx, dy) of tiles 0 .. 63; slot 0's fourth word: a beam was
    // flagged}, triple buffered by launch number like the spatial hash's slots (sb_physics.h SbGridStep); the running state
    // (SbHybridCtl) in two blocks: every workgroup reads `q_in` and computes the verdict for itself -- one wave, 64 loads -- and
    // workgroup 0 writes the updated block `q_out` for the next launch.  Nothing is communicated inside a launch.
    const SbHybridCtl *q_in;
    SbHybridCtl *q_out;
    const float4 *slots_prev; // what the launch before left (nullptr: nothing to validate -- the first launch of a run)
    float4 *slots_out, *slots_zero;
    uint32_t k_prev;          // substeps of the launch before
    uint32_t *broken_prev;    // its break flags, waiting for the verdict (merged into `broken_ok` by the tiles that own the beams)
    uint32_t *broken_ok;
};
#define SB_HY_SLOTS 64u  // (slot 0's fourth word: the launch flagged a beam)
template <int MAT, bool AUX, bool PLAIN, bool TRACK>
__global__ __launch_bounds__(SB_BK_T) __attribute__((amdgpu_waves_per_eu(AUX ? SB_BK_WAVES_AUX : SB_BK_WAVES, AUX ? SB_BK_WAVES_AUX : SB_BK_WAVES))) void k_substep_blocked(
    SbParticleArrays r, SbParticleArrays w, SbBlockedPlan bp, SbBlockedState bs, uint32_t k_run, const SbConsts c, SbParams prm,
    const uint32_t *__restrict__ acc_flag_r, uint32_t *acc_flag_w, SbTrack tr)
{
    float moved_sum = 0.0f; // TRACK: sum over the substeps of this thread's largest drift-relative displacement
    float track_cx = 0.0f, track_cy = 0.0f;
    bool prev_broke = false; // TRACK: the launch before flagged beams, and it counts: its flags are merged below
    // TRACK: what the verdict on the launch before needs is REQUESTED here and consumed behind the acceleration / plastic flags
    // below, which are a trip to the L2 of their own anyway (consumed at once, the verdict put a second trip in front of every
    // workgroup's first request: +3.8 us per launch)
    uint32_t hy_wv = 0u;
    float4 hy_sl = make_float4(0.f, 0.f, 0.f, 0.f);
    if (TRACK) {
        // the verdict on the launch before this one, by every workgroup for itself (uniform: everybody reads the same words -- through
        // the VECTOR path, a word per lane: the launch before wrote them, and the scalar cache is not refreshed between launches)
        const uint32_t lane = threadIdx.x & 63u, nw = (uint32_t)(sizeof(SbHybridCtl) / 4u);
        // (no inline asm up here to force the vector path: an asm statement in front of the tile-table loads counts as a store that
        // may alias them, and they stop being scalar loads -- ten vector loads and their waits in every workgroup's first microsecond;
        // the per-lane indices do it by themselves, and the flag rides in slot 0's fourth word)
        hy_wv = ((const uint32_t *)tr.q_in)[lane < nw ? lane : 0u];
        if (tr.slots_prev) hy_sl = tr.slots_prev[lane];
    }
    // static LDS layout: every address below is a register plus an immediate offset
    __shared__ float2 s_pos[SB_BK_CAP];
    __shared__ int s_fx[SB_BK_CAP], s_fy[SB_BK_CAP]; // fixed-point force sums (x and y apart: consecutive particles, consecutive banks)
    extern __shared__ __attribute__((aligned(16))) float s_mat[]; // [nmat][SB_BK_ROW]

    const uint32_t tile = sbb_tile_of_block(blockIdx.x, bp.ntiles), tid = threadIdx.x, K = bp.K;
    const uint32_t p0 = bp.tile_p0[tile], n_own = bp.tile_p0[tile + 1] - p0, h0 = bp.tile_h0[tile];
    const uint32_t *rc = bp.ring_cnt + (size_t)tile * (K + 1), *lc = bp.lvl_cnt + (size_t)tile * K;
    const uint32_t b0 = bp.tile_b0[tile], n_ownb = bp.tile_b0[tile + 1] - b0, e0 = bp.tile_e0[tile], s0 = bp.tile_s0[tile];
    const uint32_t np_load = rc[k_run];       // ring <= k_run: everything this launch reads
    const uint32_t np_move = rc[k_run - 1];   // ring <= k_run - 1: everything it integrates at least once
    const uint32_t ne_load = lc[k_run - 1];
    const uint32_t lmask = (1u << SB_BK_LBITS) - 1u;

    // acceleration flags (DESIGN.md 4.1 "zero accelerations"): own tile, and the tiles that own the halo
    const bool acc_r = __hip_atomic_load(&acc_flag_r[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    const bool acc_w_dirty = __hip_atomic_load(&acc_flag_w[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    const bool own_plastic = __hip_atomic_load(&bs.plastic_r[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    bool nb_acc = false, nb_plastic_lane = false;
    {
        // (every WAVE looks at all the neighbours -- a tile has a dozen -- so that the plastic verdict is wave-uniform without
        // a barrier in front of the requests below; the acceleration verdict is only needed after one, further down)
        const uint32_t n0 = bp.tile_n0[tile], nn = bp.tile_n0[tile + 1] - n0;
        for (uint32_t i = tid & 63u; i < nn; i += 64u) {
            const uint32_t nb = bp.tile_nb[n0 + i];
            nb_acc |= __hip_atomic_load(&acc_flag_r[nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            nb_plastic_lane |= __hip_atomic_load(&bs.plastic_r[nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        }
    }
    const bool nb_plastic = __builtin_amdgcn_ballot_w64(nb_plastic_lane) != 0ull;
    if (TRACK) {
        // (every word of the block into a SCALAR register by its own readlane: copied through a struct in private memory the words
        // came back as vector values, the early return below became a divergent branch and every tile-table load behind it a
        // vector load)
        static_assert(sizeof(SbHybridCtl) == 44, "SbHybridCtl: eleven words, named below");
#define SB_Q_U(i) ((uint32_t)__builtin_amdgcn_readlane((int)hy_wv, (i)))
#define SB_Q_F(i) __uint_as_float(SB_Q_U(i))
        float qD = SB_Q_F(0), qCx = SB_Q_F(1), qCy = SB_Q_F(2), qcx = SB_Q_F(3), qcy = SB_Q_F(4);
        const float qskin = SB_Q_F(5);
        uint32_t qbad = SB_Q_U(6), qdone = SB_Q_U(7), qsub = SB_Q_U(8);
        const uint32_t qpad = SB_Q_U(9), qfail = SB_Q_U(10);
#undef SB_Q_F
#undef SB_Q_U
        if (qbad == 0u && tr.slots_prev) {
            float m = hy_sl.x, sx = hy_sl.y, sy = hy_sl.z;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { // (a butterfly: the same sums in every wave of every workgroup)
                m = fmaxf(m, __shfl_xor(m, off, 64));
                sx += __shfl_xor(sx, off, 64);
                sy += __shfl_xor(sy, off, 64);
            }
            m = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(m)));
            sx = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(sx)));
            sy = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(sy)));
            const float D = qD + m;
            const bool ok = D <= qskin && qdone != qfail; // (NaN-safe; fail_at: the forced roll-back of the tests)
            if (!ok) {
                qbad = 1u;
            } else {
                const float ns = (float)min(bp.ntiles, SB_HY_SLOTS);
                float mx = sb_div(sx, ns), my = sb_div(sy, ns); // (the drift this launch measures against: any estimate keeps the bound valid)
                if (!(sb_abs(mx) < 1.0e30f) || !(sb_abs(my) < 1.0e30f)) mx = my = 0.0f;
                qD = D;
                qCx += (float)tr.k_prev * qcx; // (the drift the launch before measured against)
                qCy += (float)tr.k_prev * qcy;
                qcx = mx;
                qcy = my;
                qdone += 1u;
                qsub += tr.k_prev;
                prev_broke = __builtin_amdgcn_readlane((int)__float_as_uint(hy_sl.w), 0) != 0;
            }
        }
        // (workgroup 0's two stores sit HERE, behind the tile-table loads: a global store in front of them makes them vector loads --
        // the scalar cache is not coherent with this kernel's own stores, and the compiler cannot tell the arrays apart)
        if (blockIdx.x == 0u && threadIdx.x < SB_HY_SLOTS) tr.slots_zero[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (blockIdx.x == 0u && threadIdx.x == 0u) {
            SbHybridCtl o;
            o.D = qD; o.Cx = qCx; o.Cy = qCy; o.cx = qcx; o.cy = qcy; o.skin = qskin;
            o.bad = qbad; o.done = qdone; o.substeps = qsub; o.pad_ = qpad; o.fail_at = qfail;
            *tr.q_out = o;
        }
        if (qbad != 0u) return; // (uniform; nothing has been written: the launches behind return the same way)
        track_cx = qcx;
        track_cy = qcy;
    }

    if (TRACK && prev_broke) { // (rare) the flags the launch before raised on THIS tile's beams count now: into the mask the way back reads
        for (uint32_t wd = (b0 >> 5) + tid; wd <= ((b0 + n_ownb) >> 5) && n_ownb; wd += SB_BK_T) {
            const uint32_t lo = max(b0, wd * 32u), hi = min(b0 + n_ownb, wd * 32u + 32u);
            if (hi <= lo) continue;
            const uint32_t mask = (hi - lo == 32u ? 0xFFFFFFFFu : ((1u << (hi - lo)) - 1u)) << (lo - wd * 32u);
            const uint32_t bits = __hip_atomic_load(&tr.broken_prev[wd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & mask;
            if (bits) {
                atomicOr(&tr.broken_ok[wd], bits);
                atomicAnd(&tr.broken_prev[wd], ~bits);
            }
        }
    }
    // ---- load, in two waves of requests: (1) everything whose address follows from the tile tables -- the INDICES of the halo
    // and of the halo entries' states first (the gathers wait for them, and a wait for the k-th request is a wait for the k - 1
    // before it), then own particles, own beam states and the entry words -- all issued back to back; (2) the gathers the
    // indices name, into registers no request of (1) writes (slot classes, above).
    // Every request below is UNCONDITIONAL -- a lane with nothing to fetch reads element 0 of the array and discards it: a
    // conditional load is a branch, and behind a dozen exec-mask branches the compiler's bookkeeping of what is in flight
    // degrades to "wait for everything" at the first use of the first index.
    const uint32_t nh_load = np_load - n_own, nh_move = np_move - n_own, nhe = ne_load - min(n_ownb, ne_load);
    uint32_t hidx[SB_BK_HALOP], sidx[SB_BK_HALOB];
#pragma unroll
    for (int i = 0; i < SB_BK_HALOP; i++) {
        const uint32_t h = tid + (uint32_t)i * SB_BK_T;
        hidx[i] = bp.halo_idx[h < nh_load ? h0 + h : 0u];
    }
#pragma unroll
    for (int i = 0; i < SB_BK_HALOB; i++) {
        const uint32_t j = tid + (uint32_t)i * SB_BK_T;
        sidx[i] = bp.ent_state[j < nhe ? s0 + j : 0u];
    }
    float2 pp[SB_BK_MAXP], pv[SB_BK_MAXP], pa[SB_BK_MAXP];
#pragma unroll
    for (int i = 0; i < SB_BK_OWNP; i++) {
        const uint32_t q = tid + (uint32_t)i * SB_BK_T;
        pp[i] = r.pos[q < n_own ? p0 + q : 0u];
        pv[i] = r.vel[q < n_own ? p0 + q : 0u];
        pa[i] = make_float2(0.f, 0.f);
    }
    float tg[SB_BK_MAXB], ls[SB_BK_MAXB], ln[SB_BK_MAXB];
#pragma unroll
    for (int i = 0; i < SB_BK_OWNB; i++) {
        const uint32_t j = tid + (uint32_t)i * SB_BK_T;
        ls[i] = bs.last_r[j < n_ownb ? b0 + j : 0u];
    }
    uint32_t word[SB_BK_MAXB];
#pragma unroll
    for (int i = 0; i < SB_BK_MAXB; i++) {
        const uint32_t j = i < (int)SB_BK_OWNB ? tid + (uint32_t)i * SB_BK_T : n_ownb + tid + (uint32_t)(i - (int)SB_BK_OWNB) * SB_BK_T;
        const bool have = i < (int)SB_BK_OWNB ? j < n_ownb : j < ne_load;
        word[i] = bp.ent_word[have ? e0 + j : 0u];
        ln[i] = MAT == 1 ? bp.ent_length[have ? e0 + j : 0u] : 1.0f;
    }
    // own targets: only a tile that has yielded fetches them (the others all read element 0: one cache line); last of this
    // round, because the address waits for the tile's flag
#pragma unroll
    for (int i = 0; i < SB_BK_OWNB; i++) {
        const uint32_t j = tid + (uint32_t)i * SB_BK_T;
        tg[i] = bs.target_r[(own_plastic && j < n_ownb) ? b0 + j : 0u];
    }
    // MAT == 1 (rest lengths that do not fit the dictionary: every scene built the way the reference's editor builds beams):
    // the rest length of every entry is staged in LDS behind the material rows and read per evaluation; its reciprocal is
    // recomputed there (the short exact form).  Until r03 both sat in registers, 24 of the 128, and the variant spilled 34-47.
    float *s_len = s_mat + SB_BK_ROW * bp.nmat;
    // (2) the gathers
#pragma unroll
    for (int i = 0; i < SB_BK_HALOP; i++) {
        const uint32_t h = tid + (uint32_t)i * SB_BK_T;
        const uint32_t at = h < nh_load ? hidx[i] : 0u;
        pp[SB_BK_OWNP + i] = r.pos[at];
        pv[SB_BK_OWNP + i] = r.vel[h < nh_move ? at : 0u];
        pa[SB_BK_OWNP + i] = make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < SB_BK_HALOB; i++) {
        const uint32_t j = tid + (uint32_t)i * SB_BK_T;
        const uint32_t at = j < nhe ? sidx[i] : 0u;
        tg[SB_BK_OWNB + i] = bs.target_r[nb_plastic ? at : 0u]; // (no tile around has yielded: nothing to gather)
        ls[SB_BK_OWNB + i] = bs.last_r[at];
    }
    // (what the lanes with nothing to fetch hold instead: a unit beam between this lane's own pair of dummy records)
    const uint32_t pad_word = (SB_BK_PAD0 + 2u * (tid & 63u)) | ((SB_BK_PAD0 + 2u * (tid & 63u) + 1u) << SB_BK_LBITS);
#pragma unroll
    for (int i = 0; i < SB_BK_MAXB; i++) {
        const uint32_t j = i < (int)SB_BK_OWNB ? tid + (uint32_t)i * SB_BK_T : n_ownb + tid + (uint32_t)(i - (int)SB_BK_OWNB) * SB_BK_T;
        const bool have = i < (int)SB_BK_OWNB ? j < n_ownb : j < ne_load;
        word[i] = have ? word[i] : pad_word;
        tg[i] = have ? tg[i] : 1.0f;
        ls[i] = have ? ls[i] : 1.0f;
        if (MAT == 1) ln[i] = have ? ln[i] : 1.0f;
    }
#pragma unroll
    for (int i = 0; i < SB_BK_MAXP; i++) {
        const uint32_t q = i < (int)SB_BK_OWNP ? tid + (uint32_t)i * SB_BK_T : n_own + tid + (uint32_t)(i - (int)SB_BK_OWNP) * SB_BK_T;
        const bool moves = i < (int)SB_BK_OWNP ? q < n_own : q < np_move;
        pv[i] = moves ? pv[i] : make_float2(0.f, 0.f);
    }
    // accelerations: almost never (DESIGN.md 4.1 "zero accelerations"), so behind everything else -- a branch in the middle of
    // the requests above makes the compiler lose count of what is in flight and wait for all of it
    if (acc_r) {
#pragma unroll
        for (int i = 0; i < SB_BK_OWNP; i++) {
            const uint32_t q = tid + (uint32_t)i * SB_BK_T;
            if (q < n_own) pa[i] = r.acc[p0 + q];
        }
    }
    // halo accelerations: only when a tile that owns part of the halo has any
    if (__syncthreads_or(nb_acc ? 1 : 0)) {
#pragma unroll
        for (int i = 0; i < SB_BK_HALOP; i++) {
            const uint32_t h = tid + (uint32_t)i * SB_BK_T;
            if (h < nh_move) pa[SB_BK_OWNP + i] = r.acc[hidx[i]];
        }
    }
    // LDS index of a particle slot: own q = tid + i T, halo q = n_own + tid + (i - OWNP) T
    const uint32_t q_halo = n_own + tid;
#define SB_SLOT_Q(i) ((i) < (int)SB_BK_OWNP ? tid + (uint32_t)(i) * SB_BK_T : q_halo + (uint32_t)((i) - (int)SB_BK_OWNP) * SB_BK_T)
#pragma unroll
    for (int i = 0; i < SB_BK_MAXP; i++) {
        const uint32_t q = SB_SLOT_Q(i);
        const bool have = i < (int)SB_BK_OWNP ? q < n_own : q < np_load;
        if (have) {
            s_pos[q] = pp[i];
            s_fx[q] = 0;
            s_fy[q] = 0;
        }
    }
    if (MAT == 1) {
#pragma unroll
        for (int i = 0; i < SB_BK_MAXB; i++) s_len[tid + (uint32_t)i * SB_BK_T] = ln[i];
    }
    if (tid < 65u) { // the endpoints of dead entries and of every lane's padding entries: unit beams nobody owns
        const uint32_t a = tid == 64u ? SB_BK_DUMMY_A : SB_BK_PAD0 + 2u * tid;
        s_pos[a] = make_float2(0.f, 0.f);
        s_pos[a + 1u] = make_float2(1.f, 0.f);
        s_fx[a] = s_fy[a] = s_fx[a + 1u] = s_fy[a + 1u] = 0;
    }
    for (uint32_t i = tid; i < bp.nmat; i += SB_BK_T) { // host row: length, spring, damp, yield, limit, 1/length
        const float *h = bp.mat_tab + 6u * i;
        float *d = s_mat + SB_BK_ROW * i;
        d[0] = h[0];
        d[1] = h[5];
        d[2] = h[1] * 65536.0f; // (spring, damp) times the force scale: SbBeamMat::sd
        d[3] = h[2] * 65536.0f;
        d[4] = h[3];
        d[5] = h[3] * h[0]; // yield_strain * length, the first product of compute.wgsl:115
        d[6] = h[0] * h[4]; // length * strain_break_limit, :117
        d[7] = h[4];
    }
    uint32_t brk = 0u; // bit i: entry i of this thread crossed its break limit in some substep
    bool any_acc = false;
    __syncthreads();
    // targets nobody fetched (tiles that have never yielded: plastic flags above) are the rest lengths
    if (!own_plastic || !nb_plastic) {
#pragma unroll
        for (int i = 0; i < SB_BK_MAXB; i++) {
            const bool own = i < (int)SB_BK_OWNB;
            const uint32_t j = own ? tid + (uint32_t)i * SB_BK_T : n_ownb + tid + (uint32_t)(i - (int)SB_BK_OWNB) * SB_BK_T;
            const bool have = own ? j < n_ownb : j < ne_load;
            if (have && !(own ? own_plastic : nb_plastic))
                tg[i] = MAT == 2 ? s_mat[SB_BK_ROW * (word[i] >> (2u * SB_BK_LBITS))] : s_len[tid + (uint32_t)i * SB_BK_T];
        }
    }

    // this substep's prefixes (entries / particles whose inputs are still the true state); the next substep's are
    // requested a whole substep ahead, and all of them are wave-uniform
    uint32_t nbl = __builtin_amdgcn_readfirstlane(lc[k_run - 1]), npr = __builtin_amdgcn_readfirstlane(rc[k_run - 1]);
    // SB_BK_ABLATE (diagnostic builds, never shipped; tools/blocked_ablation.sh): 16 = load and store phases only, 1 = no beam
    // phase, 2 = no particle arithmetic, 4 = no force atomics, 32 = no barriers (wrong results, timing only)
#ifndef SB_BK_ABLATE
#define SB_BK_ABLATE 0
#endif
    for (uint32_t s = 1; s <= k_run && !((SB_BK_ABLATE & 16) && prm.time_step >= 0.0f); s++) {
        uint32_t nbl_next = 0u, npr_next = 0u;
        if (s < k_run) {
            nbl_next = lc[k_run - s - 1];
            npr_next = rc[k_run - s - 1];
        }
        // ---- strain / stress (compute.wgsl:122-123): outputs of the LAST substep of a call, for owned beams only.  A short pass
        // of its own over the own slots in front of that substep's beam phase -- the beam's length once more, then the
        // operations of sb_beam_eval in the same order -- while entry words and beam states are the only live arrays.  (r02 / r03
        // had it inside the groups of the beam phase, sharing their endpoint and material reads: 17 - 21 spilled registers in
        // the variant that ends every call.)
        if (AUX && s == k_run) {
            // (branch-free: slots with nothing to report -- padding, beams a delete pass removed -- evaluate their unit beam
            // between dummy records like everybody else and store into this thread's dump element behind the last beam; six
            // exec-mask branches cost the variant its scalar registers, whose spills then spilled in turn)
            const uint32_t dump = bp.tile_b0[bp.ntiles] + tid;
#pragma unroll
            for (int i = 0; i < (int)SB_BK_OWNB; i++) {
                uint32_t j = tid + (uint32_t)i * SB_BK_T;
                asm volatile("" : "+v"(j)); // (or the store addresses of a thread are hoisted out of the substep loop: 48 VGPRs)
                uint32_t wd = word[i];
                asm volatile("" : "+v"(wd));
                const uint32_t at = (j < n_ownb && wd != bp.dummy_word) ? b0 + j : dump;
                const float4 *row = (const float4 *)(s_mat + SB_BK_ROW * (wd >> (2u * SB_BK_LBITS)));
                const float2 qa = s_pos[wd & lmask], qb = s_pos[(wd >> SB_BK_LBITS) & lmask];
                const float4 r0 = row[0]; // length, 1/length, spring', damp' (times the force scale: the scaling is exact)
                const float yield_strain = row[1].x;
                float inv_length = r0.y;
                if (MAT != 2) {
                    const float len_i = s_len[tid + (uint32_t)i * SB_BK_T];
                    inv_length = sb_wave_all(len_i >= 0x1p-45f && len_i <= 0x1p45f) ? sb_rcp_gated(len_i) : sb_div(1.0f, len_i);
                }
                const float len = sb_beam_length(qa, qb);
                const float force_mag = (tg[i] - len) * (r0.z * 0x1p-16f) + (ls[i] - len) * (r0.w * 0x1p-16f); // :110
                const float strain_v = (len - tg[i]) * inv_length;                                           // :112
                bs.stress[at] = force_mag * (1.0f / 20.0f);                                                  // :122
                bs.strain[at] = sb_div(sb_abs(strain_v), yield_strain);                                      // :123
                asm volatile("" ::: "memory"); // one slot at a time: six interleaved evaluations do not fit beside the substep loop's registers
            }
        }
        // ---- beam phase: groups of SB_BK_G entries in straight-line code.  An entry past the prefix (its inputs
        // are no longer the true state) may ride along in a group: its force lands on particles that are not
        // integrated any more and its state is never stored (owned entries are always inside the prefix).
#pragma unroll
        for (int i0 = 0; i0 < SB_BK_MAXB; i0 += SB_BK_G) {
            // (the entry number of the group's first slot: own slots count from 0, halo slots from the tile's own beams)
            const uint32_t j0 = i0 < (int)SB_BK_OWNB ? tid + (uint32_t)i0 * SB_BK_T : n_ownb + tid + (uint32_t)(i0 - (int)SB_BK_OWNB) * SB_BK_T;
            if ((i0 < (int)SB_BK_OWNB ? j0 < n_ownb : j0 < nbl) && !((SB_BK_ABLATE & 1) && prm.time_step >= 0.0f)) {
                float2 qa[SB_BK_G], qb[SB_BK_G];
                SbBeamMat mt[SB_BK_G];
                uint32_t la[SB_BK_G], lb[SB_BK_G];
                float t_in[SB_BK_G], l_in[SB_BK_G];
                float spring_s[SB_BK_G], damp_s[SB_BK_G]; // (times the force scale: SbBeamMat::sd)
                int32_t fa[SB_BK_G][2], fb[SB_BK_G][2];
                bool broken[SB_BK_G];
#pragma unroll
                for (int u = 0; u < SB_BK_G; u++) {
                    const int i = i0 + u;
                    asm volatile("" : "+v"(word[i])); // keep the unpacking inside the loop: hoisted it is 3 registers per entry
                    la[u] = word[i] & lmask;
                    lb[u] = (word[i] >> SB_BK_LBITS) & lmask;
                    // the material row as two 16-byte reads (ds_read_b128: 4 LDS cycles each; the 12-byte form costs 8)
                    const float4 *row = (const float4 *)(s_mat + SB_BK_ROW * (word[i] >> (2u * SB_BK_LBITS)));
                    qa[u] = s_pos[la[u]];
                    qb[u] = s_pos[lb[u]];
                    const float4 r0 = row[0], r1 = row[1]; // length, 1/length, spring', damp' | yield, yield*length, length*limit, limit
                    mt[u].sd = sb_v2{r0.z, r0.w};
                    spring_s[u] = r0.z;
                    damp_s[u] = r0.w;
                    mt[u].yield_strain = r1.x;
                    if (MAT == 2) {
                        mt[u].length = r0.x;
                        mt[u].inv_length = r0.y;
                        mt[u].yl = r1.y;
                        mt[u].ll = r1.z;
                    } else {
                        const float len_i = s_len[tid + (uint32_t)i * SB_BK_T];
                        mt[u].length = len_i;
                        mt[u].yl = r1.x * len_i;
                        mt[u].ll = len_i * r1.w;
                    }
                    t_in[u] = tg[i];
                    l_in[u] = ls[i];
                }
                if (MAT == 1) { // 1 / length (IEEE), as the short exact form whenever every lane's length is an ordinary number
                    unsigned long long odd = 0ull;
#pragma unroll
                    for (int u = 0; u < SB_BK_G; u++)
                        odd |= __builtin_amdgcn_ballot_w64(!(mt[u].length >= 0x1p-45f)) | __builtin_amdgcn_ballot_w64(!(mt[u].length <= 0x1p45f));
#pragma unroll
                    for (int u = 0; u < SB_BK_G; u++) mt[u].inv_length = odd == 0ull ? sb_rcp_gated(mt[u].length) : sb_div(1.0f, mt[u].length);
                }
                bool mirrored;
                sb_beam_group<SB_BK_G>(qa, qb, mt, t_in, l_in, fa, fb, mirrored, broken);
#pragma unroll
                for (int u = 0; u < SB_BK_G; u++) {
                    const int i = i0 + u;
                    tg[i] = t_in[u];
                    ls[i] = l_in[u];
                    if (__builtin_expect(broken[u], 0)) brk |= 1u << i;
                }
                if ((SB_BK_ABLATE & 4) && prm.time_step >= 0.0f) {
                } else if (__builtin_expect(mirrored, 1)) { // A's share is minus B's: ds_sub of the same integers (sb_beam_group)
#pragma unroll
                    for (int u = 0; u < SB_BK_G; u++) {
                        __hip_atomic_fetch_sub(&s_fx[la[u]], fb[u][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_sub(&s_fy[la[u]], fb[u][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(&s_fx[lb[u]], fb[u][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(&s_fy[lb[u]], fb[u][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < SB_BK_G; u++) {
                        atomicAdd(&s_fx[la[u]], fa[u][0]);
                        atomicAdd(&s_fy[la[u]], fa[u][1]);
                        atomicAdd(&s_fx[lb[u]], fb[u][0]);
                        atomicAdd(&s_fy[lb[u]], fb[u][1]);
                    }
                }
            }
        }
        if (!(SB_BK_ABLATE & 32)) __syncthreads();
        // ---- particle phase: consume and clear the complete sums (compute.wgsl:171-201, :184-185)
        float moved_now = 0.0f;
#pragma unroll
        for (int i = 0; i < SB_BK_MAXP; i++) {
            const uint32_t q = SB_SLOT_Q(i);
            if (i < (int)SB_BK_OWNP ? q < n_own : q < npr) { // (own particles are inside every prefix)
                SbParticle particle;
                particle.p = s_pos[q];
                particle.v = pv[i];
                particle.a = pa[i];
                const int fx = s_fx[q], fy = s_fy[q];
                s_fx[q] = 0;
                s_fy[q] = 0;
                const float2 p_old = particle.p;
                if (!((SB_BK_ABLATE & 2) && prm.time_step >= 0.0f)) sb_particle_finish<PLAIN>(prm, c, particle, fx, fy);
                if (TRACK && i < (int)SB_BK_OWNP) { // (as sb_substep_tiled measures it: sqrt(2) x the larger component)
                    const float ddx = particle.p.x - p_old.x, ddy = particle.p.y - p_old.y;
                    const float m = fmaxf(sb_abs(ddx - track_cx), sb_abs(ddy - track_cy)) * 1.4142137f;
                    moved_now = fmaxf(moved_now, m < 1.0e30f ? m : 1.0e30f); // (NaN reads as huge)
                    if (i == 0 && tid == 0 && s == k_run) { // this tile's sample for the next launch's drift estimate
                        if (tile < SB_HY_SLOTS) { // (the sample of tiles 0 .. 63: the slot's y and z, which nobody else writes)
                            __hip_atomic_store((uint32_t *)&tr.slots_out[tile].y, __float_as_uint(sb_abs(ddx) < 1.0e30f ? ddx : 0.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store((uint32_t *)&tr.slots_out[tile].z, __float_as_uint(sb_abs(ddy) < 1.0e30f ? ddy : 0.0f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
                s_pos[q] = particle.p;
                pv[i] = particle.v;
                pa[i] = particle.a;
            }
        }
        if (TRACK) moved_sum += moved_now;
        if (!(SB_BK_ABLATE & 32)) __syncthreads();
        nbl = __builtin_amdgcn_readfirstlane(nbl_next);
        npr = __builtin_amdgcn_readfirstlane(npr_next);
    }

    // ---- store: own particles, own beams (their slots: the first of each class).  The indices are re-derived from a thread
    // number the compiler cannot see through: computed from `tid` they are loop-invariant, get hoisted above the substep loop
    // as a dozen 64-bit addresses, are spilled there for want of registers and reloaded here -- scratch loads in front of the
    // stores, i.e. waits on the in-order memory counter between them.
    uint32_t tid_s = tid;
    asm volatile("" : "+v"(tid_s));
#pragma unroll
    for (int i = 0; i < SB_BK_OWNP; i++) {
        const uint32_t q = tid_s + (uint32_t)i * SB_BK_T;
        if (q < n_own) {
            sb_store_wt(&w.pos[p0 + q], s_pos[q]); // (write-through: sb_physics.h; config 2 at 1 M: 8.15 -> 8.32e10)
            sb_store_wt(&w.vel[p0 + q], pv[i]);
            const bool nz = (__float_as_uint(pa[i].x) | __float_as_uint(pa[i].y)) != 0u; // -0.0 counts
            any_acc |= nz;
            if (nz || acc_w_dirty) w.acc[p0 + q] = pa[i];
        }
    }
    // has a beam of a so far unyielded tile yielded in this launch?  (compute.wgsl:113-116 moved its target off its length)
    bool yielded = false;
    if (!own_plastic) {
#pragma unroll
        for (int i = 0; i < SB_BK_OWNB; i++) {
            const uint32_t j = tid_s + (uint32_t)i * SB_BK_T;
            const float len_i = MAT == 2 ? s_mat[SB_BK_ROW * (word[i] >> (2u * SB_BK_LBITS))] : s_len[tid + (uint32_t)i * SB_BK_T];
            yielded |= j < n_ownb && word[i] != bp.dummy_word && __float_as_uint(tg[i]) != __float_as_uint(len_i);
        }
    }
    const bool plastic_w = own_plastic || __syncthreads_or(yielded ? 1 : 0) != 0;
    if (tid == 0) bs.plastic_w[tile] = plastic_w ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < SB_BK_OWNB; i++) {
        const uint32_t j = tid_s + (uint32_t)i * SB_BK_T;
        if (j < n_ownb) {
            if (__builtin_expect(word[i] != bp.dummy_word, 1)) {
                if (plastic_w) sb_store_wt(&bs.target_w[b0 + j], tg[i]);
                sb_store_wt(&bs.last_w[b0 + j], ls[i]);
                if ((brk >> i) & 1u) {
                    atomicOr(&bs.broken[(b0 + j) >> 5], 1u << ((b0 + j) & 31u));
                    if (TRACK) __hip_atomic_store((uint32_t *)&tr.slots_out[0].w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                // a beam removed by a delete pass keeps its last state, which still has to travel to the other buffer: load,
                // wait and store all INSIDE this branch (the wait at the join of the conditional form ran on every beam and
                // drained the stores before it, one memory round trip per slot)
                float keep_t = bs.target_r[b0 + j], keep_l = bs.last_r[b0 + j];
                SB_LANDED(keep_t);
                SB_LANDED(keep_l);
                bs.target_w[b0 + j] = keep_t;
                bs.last_w[b0 + j] = keep_l;
            }
        }
    }
    const int wg_any = __syncthreads_or(any_acc ? 1 : 0);
    if (tid == 0) acc_flag_w[tile] = wg_any ? 1u : 0u;
    if (TRACK) {
        __shared__ float s_moved[SB_BK_T / 64u];
        float m = moved_sum < 1.0e30f ? moved_sum : 1.0e30f;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        if ((tid & 63u) == 0u) s_moved[tid >> 6] = m;
        __syncthreads();
        if (tid == 0) {
            float b = 0.0f;
            for (uint32_t v = 0; v < SB_BK_T / 64u; v++) b = fmaxf(b, s_moved[v]);
            // (sums are >= 0 and never NaN: their bit patterns order like the numbers)
            (void)__hip_atomic_fetch_max((uint32_t *)&tr.slots_out[tile % SB_HY_SLOTS].x, __float_as_uint(b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// behind the LAST tracked launch of a run (every other launch is validated by its successor's prologue, SbTrack): the same verdict
// from the same slots, the running state into the block the host reads, the launch's break flags kept or dropped
__global__ __launch_bounds__(64) void k_hybrid_validate(const float4 *__restrict__ slots, uint32_t ntiles, const SbHybridCtl *q_in, SbHybridCtl *q_out,
                                                        uint32_t k, uint32_t *broken_new, uint32_t *broken_ok, uint32_t nwords)
{
    const uint32_t lane = threadIdx.x;
    const float4 sl = slots[lane];
    const bool flagged = __builtin_amdgcn_readlane((int)__float_as_uint(sl.w), 0) != 0;
    SbHybridCtl q = *q_in;
    bool ok = false;
    if (q.bad == 0u) {
        float m = sl.x, sx = sl.y, sy = sl.z;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { // (the butterfly of the launches' own prologues: the same numbers)
            m = fmaxf(m, __shfl_xor(m, off, 64));
            sx += __shfl_xor(sx, off, 64);
            sy += __shfl_xor(sy, off, 64);
        }
        const float D = q.D + m;
        ok = D <= q.skin && q.done != q.fail_at;
        if (!ok) {
            q.bad = 1u;
        } else {
            const float ns = (float)min(ntiles, SB_HY_SLOTS);
            float mx = sb_div(sx, ns), my = sb_div(sy, ns);
            if (!(sb_abs(mx) < 1.0e30f) || !(sb_abs(my) < 1.0e30f)) mx = my = 0.0f;
            q.D = D;
            q.Cx += (float)k * q.cx;
            q.Cy += (float)k * q.cy;
            q.cx = mx;
            q.cy = my;
            q.done += 1u;
            q.substeps += k;
        }
    }
    if (lane == 0u) *q_out = q;
    if (flagged) // break flags the launch raised (almost never any): kept if the launch counts, dropped if not
        for (uint32_t i = lane; i < nwords; i += 64u) {
            const uint32_t bits = broken_new[i];
            if (bits) {
                if (ok) broken_ok[i] |= bits;
                broken_new[i] = 0u;
            }
        }
}

// beam state between the two device layouts of an SB_COLLIDE_GRID engine that can run blocked (sb_api.hip hybrid_step): the
// tiled layout (one copy per tile that holds an endpoint) is where the state lives between API calls; the blocked layout
// (one slot per beam) borrows it for a run of blocked launches
// One workgroup per tile of the blocked plan: its own beams are entries e0 .. of the tile, in the order of the state slots
// b0 .. (k_substep_blocked), so the rest length of each is at hand -- and with it the tile's plastic flag as of NOW: 0 = every
// own beam's target is its rest length, bit for bit (what the tile's launches then neither read nor write).
__global__ __launch_bounds__(256) void k_hybrid_to_blocked(const float *__restrict__ t_target, const float *__restrict__ t_last,
                                                           const uint32_t *__restrict__ copy_of_g, const uint32_t *__restrict__ tile_b0,
                                                           const uint32_t *__restrict__ tile_e0, const uint32_t *__restrict__ ent_word,
                                                           const float *__restrict__ ent_length, const float *__restrict__ mat_tab,
                                                           float *b_target, float *b_target_other, float *b_last, uint32_t *plastic_a,
                                                           uint32_t *plastic_b)
{
    const uint32_t tile = blockIdx.x, b0 = tile_b0[tile], nb = tile_b0[tile + 1] - b0, e0 = tile_e0[tile];
    bool differs = false;
    for (uint32_t j = threadIdx.x; j < nb; j += 256u) {
        const uint32_t g = b0 + j, cpy = copy_of_g[g];
        const float t = t_target[cpy];
        const float rest = ent_length ? ent_length[e0 + j] : mat_tab[6u * (ent_word[e0 + j] >> (2u * SB_BK_LBITS))];
        differs |= __float_as_uint(t) != __float_as_uint(rest);
        b_target[g] = t;
        b_target_other[g] = t; // (a tile that does not yield never stores its targets: both buffers must hold them)
        b_last[g] = t_last[cpy];
    }
    const int any = __syncthreads_or(differs ? 1 : 0);
    if (threadIdx.x == 0u) plastic_a[tile] = plastic_b[tile] = any ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_hybrid_to_tiled(const float *__restrict__ b_target, const float *__restrict__ b_last,
                                                         const float *__restrict__ b_strain, const float *__restrict__ b_stress,
                                                         uint32_t *b_broken, const uint32_t *__restrict__ g_of_copy, uint32_t nc,
                                                         const uint32_t *__restrict__ t_pair, float *t_target, float *t_last,
                                                         float *t_strain, float *t_stress, uint32_t *t_broken, int aux)
{
    const uint32_t cpy = blockIdx.x * 256u + threadIdx.x;
    if (cpy >= nc) return;
    const uint32_t g = g_of_copy[cpy];
    if (g == 0xFFFFFFFFu) return; // padding copy
    // a copy a delete pass has removed keeps the record it died with (that is what a read-back returns for its data index): the
    // blocked layout's slot of that beam holds whatever its dummy entry computed, or a strain from before its last single substeps
    if (t_pair[cpy] == 0xFFFFFFFFu) return;
    t_target[cpy] = b_target[g];
    t_last[cpy] = b_last[g];
    if (aux) {
        t_strain[cpy] = b_strain[g];
        t_stress[cpy] = b_stress[g];
    }
    if ((b_broken[g >> 5] >> (g & 31u)) & 1u) atomicOr(&t_broken[cpy >> 5], 1u << (cpy & 31u));
}
// entries of beams the tiled layout's delete passes have removed since the blocked plan last looked
__global__ __launch_bounds__(256) void k_hybrid_sync_dead(const uint32_t *__restrict__ t_pair, const uint32_t *__restrict__ copy_slot,
                                                          uint32_t nc, const uint32_t *__restrict__ slot_e0,
                                                          const uint32_t *__restrict__ slot_ent, uint32_t *ent_word, uint32_t dummy_word)
{
    const uint32_t cpy = blockIdx.x * 256u + threadIdx.x;
    if (cpy >= nc || copy_slot[cpy] == 0xFFFFFFFFu || t_pair[cpy] != 0xFFFFFFFFu) return;
    const uint32_t sl = copy_slot[cpy];
    for (uint32_t en = slot_e0[sl]; en < slot_e0[sl + 1]; en++) ent_word[slot_ent[en]] = dummy_word;
}

// delete pass of the blocked layout: every entry of a flagged beam (the owner's and the halo copies in other
// tiles) turns into the dummy beam; the mapping slot records the pass, as in k_delete (sb_kernels.hip)
__global__ __launch_bounds__(256) void k_delete_blocked(uint32_t *ent_word, const uint32_t *__restrict__ beam_slot,
                                                        const uint32_t *__restrict__ slot_e0,
                                                        const uint32_t *__restrict__ slot_ent, uint32_t nwords, uint32_t nbeam,
                                                        uint32_t *broken, uint32_t *dead_gen, uint32_t gen, uint32_t dummy_word)
{
    const uint32_t wd = blockIdx.x * 256u + threadIdx.x;
    if (wd >= nwords) return;
    uint32_t bits = broken[wd];
    if (!bits) return;
    broken[wd] = 0;
    while (bits) {
        const uint32_t k = __ffs(bits) - 1;
        bits &= bits - 1;
        const uint32_t g = wd * 32 + k;
        if (g >= nbeam) break;
        const uint32_t s = beam_slot[g];
        for (uint32_t e = slot_e0[s]; e < slot_e0[s + 1]; e++) ent_word[slot_ent[e]] = dummy_word;
        dead_gen[s] = gen;
    }
}

static inline uint32_t cdiv_b(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// How a call of n substeps is cut into launches of at most kmax substeps each.  A launch pays a fixed part (its load and
// store phases, the boundary) and a part that grows a little faster than its depth (deeper rings: more redundant work), so
// neither "as deep as possible" nor "always the sweet spot" is right: 20 substeps are cheaper as 7 + 7 + 6 than as
// 6 + 6 + 6 + 2, 960 are cheaper as 160 x 6 than as 137 x 7 + 1.  Candidates are BALANCED splits (L launches of depth
// ceil(n / L) or one less) for every L from the fewest possible on; they are priced with the launch times measured on
// BASELINE config 2 (1 M particles, r03 depth sweep of the de-serialised kernel: about 4 us fixed + 12 us per substep, a
// little less per substep the deeper the launch; the shape of the curve, not its scale, is what decides).
// Returns L; *first = ceil(n / L), *n_first = how many launches have that depth (the others have first - 1).
static const float kLaunchCostUs[SB_BK_KMAX + 1] = {0.0f, 18.0f, 29.5f, 41.0f, 53.0f, 64.5f, 76.5f, 88.7f, 101.0f}; // r03: 5, 6, 7 measured
uint32_t sbk_split_call(uint32_t n, uint32_t kmax, bool fewest, uint32_t *first, uint32_t *n_first)
{
    kmax = kmax < 1u ? 1u : (kmax > SB_BK_KMAX ? SB_BK_KMAX : kmax);
    uint32_t best_L = 0;
    float best = 0.0f;
    const uint32_t L_min = (n + kmax - 1) / kmax;
    for (uint32_t L = L_min; L <= n && L <= (fewest ? L_min : L_min * 2u + 1u); L++) { // (beyond twice the fewest launches nothing gets cheaper)
        const uint32_t hi = (n + L - 1) / L, n_hi = n - (hi - 1) * L;
        const float cost = (float)n_hi * kLaunchCostUs[hi] + (float)(L - n_hi) * kLaunchCostUs[hi - 1];
        if (!best_L || cost < best) {
            best_L = L;
            best = cost;
        }
    }
    *first = best_L ? (n + best_L - 1) / best_L : 0u;
    *n_first = best_L ? n - (*first - 1) * best_L : 0u;
    return best_L;
}

// the mode-1 variants keep one float per entry in dynamic LDS: with the static part that is more than the 64 KB a launch may
// take without saying so (gfx950 has 160 KB per CU: two such workgroups still fit)
static void allow_large_lds(int device)
{
    static bool done[64] = {};
    if (device < 0 || device >= 64 || done[device]) return;
    done[device] = true;
    const int bytes = 100 * 1024;
#define SB_ALLOW(A, PL, TR) (void)hipFuncSetAttribute((const void *)k_substep_blocked<1, A, PL, TR>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)
    SB_ALLOW(false, false, false); SB_ALLOW(false, true, false); SB_ALLOW(true, false, false); SB_ALLOW(true, true, false);
    SB_ALLOW(false, false, true); SB_ALLOW(false, true, true); SB_ALLOW(true, false, true); SB_ALLOW(true, true, true);
#undef SB_ALLOW
    (void)hipGetLastError(); // (not sticky)
}

// one launch of k substeps on the blocked layout `bk` (the engine's own, or the one an SB_COLLIDE_GRID engine keeps beside
// its tiled layout); flips the particle and beam-state buffers
static void launch_one(sb_engine *e, SbBlockedDev &bk, uint32_t k, bool aux, bool track)
{
    SbBlockedPlan bp{bk.d_tile_p0, bk.d_tile_h0, bk.d_halo_idx, bk.d_ring_cnt, bk.d_tile_b0, bk.d_tile_e0, bk.d_tile_s0, bk.d_ent_word,
                     bk.d_ent_state, bk.d_lvl_cnt, bk.d_tile_n0, bk.d_tile_nb, bk.d_ent_length, bk.d_mat, bk.ntiles, bk.plan_K, bk.cap,
                     bk.nmat, bk.dummy_word};
    SbBlockedState bs{bk.d_target[bk.cur], bk.d_last[bk.cur], bk.d_target[bk.cur ^ 1u], bk.d_last[bk.cur ^ 1u], bk.d_strain, bk.d_stress,
                      bk.d_broken, bk.d_plastic[bk.cur], bk.d_plastic[bk.cur ^ 1u]};
    SbParticleArrays r = e->part[e->cur], w = e->part[e->cur ^ 1];
    SbTrack tr{};
    if (track) { // (launch number bk.seq: its slots, the block it reads and the one it writes, its own mask of break flags)
        float4 *slots = (float4 *)bk.d_hslots;
        const uint32_t set = bk.seq % 3u, stride = SB_HY_SLOTS;
        tr.q_in = bk.d_q + bk.qpar;
        tr.q_out = bk.d_q + (bk.qpar ^ 1u);
        tr.slots_prev = bk.run_launches ? slots + ((bk.seq + 2u) % 3u) * stride : nullptr;
        tr.slots_out = slots + set * stride;
        tr.slots_zero = slots + ((bk.seq + 1u) % 3u) * stride;
        tr.k_prev = bk.k_prev;
        tr.broken_prev = bk.d_broken_new[(bk.seq + 1u) & 1u];
        tr.broken_ok = bk.d_broken;
        bs.broken = bk.d_broken_new[bk.seq & 1u]; // (merged into bk.d_broken by the launch behind it, or by k_hybrid_validate)
        bk.qpar ^= 1u;
        bk.seq = (bk.seq + 1u) % 6u; // (its parity picks a mask of break flags, its remainder by 3 a set of slots: a counter that never wraps out of step)
        bk.run_launches++;
        bk.k_prev = k;
    }
#define SB_LAUNCH_B(M, A, PL, TR)                                                                                            \
    k_substep_blocked<M, A, PL, TR><<<bk.ntiles, SB_BK_T, bk.lds_bytes, e->stream>>>(r, w, bp, bs, k, e->consts, e->prm,        \
                                                                                    e->d_acc_flag[e->cur], e->d_acc_flag[e->cur ^ 1], tr)
#define SB_LAUNCH_BT(M, A, PL) do { if (track) SB_LAUNCH_B(M, A, PL, true); else SB_LAUNCH_B(M, A, PL, false); } while (0)
#define SB_LAUNCH_BA(M, PL) do { if (aux) SB_LAUNCH_BT(M, true, PL); else SB_LAUNCH_BT(M, false, PL); } while (0)
    // the constants of THIS launch (they ride in its kernarg): the reference's defaults take the plain particle phase
    const bool plain = e->consts.drag_exp == 2.0f && e->consts.mouse_active == 0u;
    if (bk.ntiles) {
        if (bk.mat_mode == 2) {
            if (plain) SB_LAUNCH_BA(2, true); else SB_LAUNCH_BA(2, false);
        } else {
            if (plain) SB_LAUNCH_BA(1, true); else SB_LAUNCH_BA(1, false);
        }
    }
#undef SB_LAUNCH_BA
#undef SB_LAUNCH_BT
#undef SB_LAUNCH_B
    e->cur ^= 1;
    bk.cur ^= 1u;
    e->substeps_done += k;
}

// The HIP runtime resolves a kernel the first time it is launched (tens of microseconds each).  A call of n substeps ends with
// the strain / stress variant and runs the lean one before it, so a short first call -- the five warm-up substeps of the bench
// protocol are ONE launch -- leaves the lean variant to be resolved inside the next, timed, call.  Called at upload for the
// plan's material mode: every variant a call may launch.
void sbk_preload_blocked(const SbBlockedDev &bk, bool tracked)
{
    hipFuncAttributes a;
#define SB_TOUCH(M, A, PL, TR) (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_substep_blocked<M, A, PL, TR>))
#define SB_TOUCH_M(M)                                      \
    do {                                                   \
        if (tracked) {                                     \
            SB_TOUCH(M, false, false, true);               \
            SB_TOUCH(M, false, true, true);                \
            SB_TOUCH(M, true, false, true);                \
            SB_TOUCH(M, true, true, true);                 \
        } else {                                           \
            SB_TOUCH(M, false, false, false);              \
            SB_TOUCH(M, false, true, false);               \
            SB_TOUCH(M, true, false, false);               \
            SB_TOUCH(M, true, true, false);                \
        }                                                  \
    } while (0)
    if (bk.mat_mode == 2) SB_TOUCH_M(2);
    else SB_TOUCH_M(1);
#undef SB_TOUCH_M
#undef SB_TOUCH
    (void)hipGetLastError();
}

// n substeps as the launches sbk_split_call chooses; the last launch of a call also stores strain/stress when write_aux
void sbk_launch_blocked(sb_engine *e, uint32_t n, bool write_aux)
{
    if (e->bk.mat_mode == 1) allow_large_lds(e->device);
    uint32_t k_hi = 0, n_hi = 0;
    sbk_split_call(n, e->bk.K, e->bk.fixed_depth, &k_hi, &n_hi);
    while (n) {
        const uint32_t k = n_hi ? k_hi : k_hi - 1u; // the deeper launches first
        if (n_hi) n_hi--;
        launch_one(e, e->bk, k, write_aux && k == n, false);
        e->beams.target = e->bk.d_target[e->bk.cur]; // what read-back, halo pack/unpack and the next launch see
        e->beams.last = e->bk.d_last[e->bk.cur];
        n -= k;
    }
}

// ---- SB_COLLIDE_GRID engines with a blocked plan beside the tiled one (e->hy; sb_api.hip hybrid_step drives these)
void sbk_hybrid_to_blocked(sb_engine *e)
{
    SbBlockedDev &h = e->hy;
    const uint32_t B = h.nbeams;
    if (!B) return;
    if (h.synced_delete_gen != e->delete_gen) { // beams the tiled layout has removed since: their entries die here too
        k_hybrid_sync_dead<<<cdiv_b(e->nbeam, 256), 256, 0, e->stream>>>(e->beams.pair, e->beams.slot, e->nbeam, h.d_slot_e0, h.d_slot_ent,
                                                                        h.d_ent_word, h.dummy_word);
        h.synced_delete_gen = e->delete_gen;
    }
    k_hybrid_to_blocked<<<h.ntiles, 256, 0, e->stream>>>(e->beams.target, e->beams.last, h.d_copy_of_g, h.d_tile_b0, h.d_tile_e0, h.d_ent_word,
                                                         h.mat_mode == 1 ? h.d_ent_length : nullptr, h.d_mat, h.d_target[h.cur],
                                                         h.d_target[h.cur ^ 1u], h.d_last[h.cur], h.d_plastic[0], h.d_plastic[1]);
    (void)hipMemsetAsync(h.d_broken, 0, (size_t)cdiv_b(B, 32) * 4, e->stream);
    for (int b = 0; b < 2; b++) (void)hipMemsetAsync(h.d_broken_new[b], 0, (size_t)cdiv_b(B, 32) * 4, e->stream);
    (void)hipMemsetAsync(h.d_hslots, 0, 3u * SB_HY_SLOTS * sizeof(float4), e->stream); // (a run starts from clean slots whatever ended the one before)
}
void sbk_hybrid_to_tiled(sb_engine *e, bool aux)
{
    SbBlockedDev &h = e->hy;
    if (!e->nbeam) return;
    k_hybrid_to_tiled<<<cdiv_b(e->nbeam, 256), 256, 0, e->stream>>>(h.d_target[h.cur], h.d_last[h.cur], h.d_strain, h.d_stress, h.d_broken,
                                                                   h.d_g_of_copy, e->nbeam, e->beams.pair, e->beams.target, e->beams.last,
                                                                   e->beams.strain, e->beams.stress, e->d_broken, aux ? 1 : 0);
}
// `count` launches of the depths in ks[], each followed by its validation; nothing here waits
void sbk_hybrid_launch(sb_engine *e, const uint32_t *ks, uint32_t count, bool aux_last)
{
    SbBlockedDev &h = e->hy;
    if (h.mat_mode == 1) allow_large_lds(e->device);
    h.run_launches = 0; // (the first launch of a run has nobody to validate)
    for (uint32_t i = 0; i < count; i++) launch_one(e, h, ks[i], aux_last && i + 1 == count, true);
    if (count) { // ... and the last one nobody behind it: the verdict on it by a launch of its own, into the block the host reads
        const uint32_t last = (h.seq + 5u) % 6u; // (the launch before the next one)
        k_hybrid_validate<<<1, 64, 0, e->stream>>>((const float4 *)h.d_hslots + (last % 3u) * SB_HY_SLOTS, h.ntiles, h.d_q + h.qpar,
                                                  h.d_q + (h.qpar ^ 1u), ks[count - 1], h.d_broken_new[last & 1u], h.d_broken, cdiv_b(h.nbeams, 32));
        h.qpar ^= 1u;
        h.validate_launches++;
    }
}

void sbk_launch_delete_blocked(sb_engine *e)
{
    if (!e->nbeam) return;
    const uint32_t nwords = cdiv_b(e->nbeam, 32);
    k_delete_blocked<<<cdiv_b(nwords, 256), 256, 0, e->stream>>>(e->bk.d_ent_word, e->beams.slot, e->bk.d_slot_e0, e->bk.d_slot_ent,
                                                                 nwords, e->nbeam, e->d_broken, e->d_dead_gen, ++e->delete_gen,
                                                                 e->bk.dummy_word);
}
