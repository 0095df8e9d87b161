// sb_kernels.hip -- HIP kernels of the substep for gfx950 (CDNA4, wave64).
//
// Two device schedules of the same canonical substep (S0 of SURVEY.md 8(a) A3: all beams
// from the read state, then all particles):
//   SB_PATH_ATOMIC  k_beams_atomic (global i32 atomics, compute.wgsl:127-130 as written)
//                   + k_particles  (compute.wgsl:134-202)
//   SB_PATH_TILED   k_substep_tiled: one workgroup per particle tile; beam forces are summed
//                   with LDS integer atomics and consumed in the same launch.
// Both are HBM-bandwidth-bound (no MFMA: < 1 flop/byte).
#include <algorithm>
#include <cstring>
#include <type_traits>

#include "sb_engine.h"

#define SB_BLOCK 256
#ifndef SB_ABLATE
#define SB_ABLATE 0
#endif
#define SB_TILE_BLOCK 512 // threads per tile workgroup: 8 waves (measured best: 256 x 4 20.6 us, 512 x 2 19.8 us, 1024 x 1 20.3 us)
#ifndef SB_UNROLL
#define SB_UNROLL 2     // owned particles / beam copies each thread has in flight at a time
#endif
#define SB_MAT_ROW 6u // length, spring, damp, yield, limit, 1/length

// ---------------------------------------------------------------- SB_PATH_ATOMIC

__global__ __launch_bounds__(SB_BLOCK) void k_beams_atomic(SbBeamArrays b, uint32_t n,
                                                           const float2 *__restrict__ pos,
                                                           int2 *forces, uint32_t *broken)
{
    uint32_t i = blockIdx.x * SB_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t ia = b.ia[i];
    if (ia == 0xFFFFFFFFu) return; // removed by a delete pass
    uint32_t ib = b.ib[i];
    const float length = b.length[i];
    SbBeamResult r = sb_beam_eval<true>(pos[ia], pos[ib], length, sb_div(1.0f, length), b.target[i], b.last[i],
                                        b.spring[i], b.damp[i], b.yield[i], b.limit[i]);
    b.target[i] = r.target_length;
    b.last[i] = r.last_length;
    b.strain[i] = r.strain;
    b.stress[i] = r.stress;
    atomicAdd(&forces[ia].x, r.ax);
    atomicAdd(&forces[ia].y, r.ay);
    atomicAdd(&forces[ib].x, r.bx);
    atomicAdd(&forces[ib].y, r.by);
    if (r.broken) atomicOr(&broken[i >> 5], 1u << (i & 31));
}

// Collision loop over every other slot in ascending order (compute.wgsl:144-170), staged
// through LDS 256 positions at a time.  Internal particle order == slot order on this path.
template <int MODE>
__global__ __launch_bounds__(SB_BLOCK) void k_particles(SbParticleArrays r, SbParticleArrays w,
                                                        int2 *forces, uint32_t P,
                                                        const SbConsts c, SbParams prm,
                                                        const uint32_t *__restrict__ pidx, SbGrid grid_all,
                                                        SbGridJob job)
{
    __shared__ float2 s_pos[SB_BLOCK];
    __shared__ SbGridShared s_grid;
    SbGrid grid = grid_all;
    if (MODE == SB_COLLIDE_GRID) { // the decision (sb_physics.h SbGridCtl); this path runs the classic schedule only
        const SbGridEarly early = sb_grid_begin(job.step);
        sb_grid_stage(job.step, early, s_grid, 1u);
        __syncthreads();
        if (threadIdx.x == 0) sb_grid_decide(job.step, grid_all, s_grid, 1u, true);
        __syncthreads();
        if (s_grid.now.abort != 0u) return; // (uniform; nothing has been written)
        if (blockIdx.x == 0 && threadIdx.x < SB_GRID_SLOTS) job.step.slots_zero[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
        grid = sb_grid_view(grid_all, s_grid.now.cur, &s_grid.now.geo, s_grid.now.Cx, s_grid.now.Cy);
    }
    uint32_t i = blockIdx.x * SB_BLOCK + threadIdx.x;
    bool active = i < P;
    SbParticle particle, self;
    if (active) {
        particle.p = r.pos[i];
        particle.v = r.vel[i];
        particle.a = r.acc[i];
    } else {
        particle.p = particle.v = particle.a = make_float2(0.f, 0.f);
    }
    self = particle;                                            // :141
    const float elasticity_coeff = sb_div(c.elasticity + 1.0f, 2.0f); // :143
    if (MODE == SB_COLLIDE_ALLPAIRS) {
        const float two_r = prm.particle_radius * 2.0f;
        for (uint32_t base = 0; base < P; base += SB_BLOCK) {
            __syncthreads();
            if (base + threadIdx.x < P) s_pos[threadIdx.x] = r.pos[base + threadIdx.x];
            __syncthreads();
            uint32_t cnt = min((uint32_t)SB_BLOCK, P - base);
            if (active) {
                for (uint32_t k = 0; k < cnt; k++) {
                    uint32_t o = base + k;
                    float2 op = s_pos[k];
                    float dx = op.x - self.p.x, dy = op.y - self.p.y;
                    float dist = sb_length(dx, dy);
                    if (o != i && (dist == 0.0f || dist < two_r))
                        sb_collide_pair(prm, c.friction, elasticity_coeff, particle, self, pidx[i], pidx[o],
                                        op, r.vel[o]);
                }
            }
        }
    }
    if (MODE == SB_COLLIDE_GRID && active) {
        sb_collide_slow(grid, s_grid.now.fresh != 0u, prm, c.friction, elasticity_coeff, particle, self, i,
                        pidx, r.pos, r.vel);
    }
    float moved = 0.0f;
    if (active) {
        int2 f = forces[i];
        forces[i] = make_int2(0, 0); // atomicExchange(..., 0), :184-185
        sb_particle_finish(prm, c, particle, f.x, f.y);
        if (MODE == SB_COLLIDE_GRID) {
            const float dx = particle.p.x - self.p.x, dy = particle.p.y - self.p.y;
            moved = fmaxf(sb_abs(dx - s_grid.now.cx), sb_abs(dy - s_grid.now.cy)) * 1.4142137f;
            if (threadIdx.x == 0) sb_store_sample_displacement(job.step.slots_out, dx, dy);
        }
        w.pos[i] = particle.p;
        w.vel[i] = particle.v;
        w.acc[i] = particle.a;
    }
    if (MODE == SB_COLLIDE_GRID) sb_store_block_displacement(job.step.slots_out, moved);
}


// ---------------------------------------------------------------- SB_PATH_TILED

// Blocks are dealt round-robin over the 8 XCDs (block b -> XCD b % 8, MI355X_MICROARCH.md
// "Workgroup dispatch"); tiles that are neighbours in the k-d order share halo particles, so
// give each XCD a contiguous run of tiles and let its private L2 serve the shared lines.
// Bijective for any n (speed only, never correctness).
SB_DEV uint32_t sb_tile_of_block(uint32_t b, uint32_t n)
{
    uint32_t q = n >> 3, r = n & 7, x = b & 7, k = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// One workgroup = one particle tile for one whole substep:
//   phase 0  own + halo positions of the READ state -> LDS; LDS force accumulators = 0
//   phase 1  stream the tile's beam slice (coalesced SoA), gather endpoints from LDS,
//            evaluate compute.wgsl:103-130, ds_add the fixed-point force onto OWNED endpoints,
//            write back target/last/strain/stress
//   phase 2  consume the complete sums: compute.wgsl:171-201 per owned particle -> WRITE state
// HBM traffic per substep = beam slice (36 B read + 16 B written per copy) + particles
// (24 B read + 24 B written) + halo positions (8 B each, mostly L2 hits): the force
// accumulator (compute.wgsl:68-69) never leaves the CU.
#define SB_TILED_PARAMS                                                                                             \
    SbParticleArrays r, SbParticleArrays w, SbBeamArrays b, const uint32_t *__restrict__ tile_p0,                   \
        const uint32_t *__restrict__ tile_b0, const uint32_t *__restrict__ tile_h0,                                 \
        const uint32_t *__restrict__ halo_idx, uint32_t ntiles, uint32_t cap_all, uint32_t cap_own, uint32_t lbits, \
        const float *__restrict__ mat_tab, uint32_t nmat, const SbConsts c, SbParams prm, uint32_t *broken,         \
        const uint32_t *__restrict__ pidx, SbGrid grid_all, SbGridJob job, const uint32_t *__restrict__ acc_flag_r, \
        uint32_t *acc_flag_w
#define SB_TILED_ARGS                                                                                               \
    r, w, b, tile_p0, tile_b0, tile_h0, halo_idx, ntiles, cap_all, cap_own, lbits, mat_tab, nmat, c, prm, broken,   \
        pidx, grid_all, job, acc_flag_r, acc_flag_w

template <int MODE, int MAT, bool AUX>
__device__ __forceinline__ void sb_substep_tiled(SB_TILED_PARAMS)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sb_lds[];
    __shared__ SbGridShared s_grid;
    SbGridEarly early;
#ifdef SB_STAMPS
    const uint64_t sb_t_start = wall_clock64();
#endif
    float2 *s_pos = (float2 *)sb_lds;
    int *s_f = (int *)(s_pos + cap_all);
    float *s_mat = (float *)(s_f + 2 * cap_all); // accumulators exist for halo slots too: their sums are never read

    const uint32_t tile = sb_tile_of_block(blockIdx.x, ntiles);
    const uint32_t p0 = tile_p0[tile], n_own = tile_p0[tile + 1] - p0;
    const uint32_t h0 = tile_h0[tile], n_halo = tile_h0[tile + 1] - h0;
    const uint32_t b0 = tile_b0[tile], nb = tile_b0[tile + 1] - b0;
    const uint32_t tid = threadIdx.x;

    // Phase 0.  Everything this workgroup will ever read from the particle arrays is requested
    // up front, back to back: the halo indices, the owned positions, and (for phase 2) the owned
    // velocities and accelerations of the first SB_UNROLL x SB_TILE_BLOCK particles.  The dependent halo
    // position gather goes out as soon as its indices are back.
    // acc_flag[buffer][tile] == 0 guarantees that every acceleration of the tile in that buffer IS
    // zero (true for almost every tile: compute.wgsl:188 zeroes a, only border friction :192,:196
    // sets it), so the 8 B/particle read and the 8 B/particle write of zeros can both be skipped.
    // (flags are rewritten by every launch: read them at agent scope, not through the scalar cache)
    const bool acc_r = __hip_atomic_load(&acc_flag_r[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    const bool acc_w_dirty = __hip_atomic_load(&acc_flag_w[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    uint32_t hidx = 0;
    const bool has_halo = tid < n_halo;
    if (has_halo) hidx = halo_idx[h0 + tid];
    float2 pp[SB_UNROLL], pv[SB_UNROLL], pa[SB_UNROLL];
#pragma unroll
    for (int u = 0; u < SB_UNROLL; u++) {
        const uint32_t i = tid + (uint32_t)u * SB_TILE_BLOCK;
        if (i < n_own) {
            pp[u] = r.pos[p0 + i];
            pv[u] = r.vel[p0 + i];
            pa[u] = acc_r ? r.acc[p0 + i] : make_float2(0.f, 0.f);
        }
    }
    // SB_COLLIDE_GRID: last substep's displacement slots and the control block, requested together with the own particles (cold
    // lines, a full trip to memory like them; the wait counter retires in order, so requested any earlier they would stand in
    // front of the acceleration flags and halo indices, which come out of the L2): they turn into this substep's decision
    // behind the first barrier (sb_physics.h SbGridCtl)
    if (MODE == SB_COLLIDE_GRID) early = sb_grid_begin(job.step);
    float2 hp = make_float2(0.f, 0.f);
    if (has_halo) hp = r.pos[hidx];
#pragma unroll
    for (int u = 0; u < SB_UNROLL; u++) {
        const uint32_t i = tid + (uint32_t)u * SB_TILE_BLOCK;
        if (i < n_own) {
            s_pos[i] = pp[u];
            s_f[2 * i] = 0;
            s_f[2 * i + 1] = 0;
        }
    }
    for (uint32_t i = tid + SB_UNROLL * SB_TILE_BLOCK; i < n_own; i += SB_TILE_BLOCK) { // tiles above 1024 particles
        s_pos[i] = r.pos[p0 + i];
        s_f[2 * i] = 0;
        s_f[2 * i + 1] = 0;
    }
    if (has_halo) {
        s_pos[n_own + tid] = hp;
        s_f[2 * (n_own + tid)] = 0;
        s_f[2 * (n_own + tid) + 1] = 0;
    }
    for (uint32_t i = tid + SB_TILE_BLOCK; i < n_halo; i += SB_TILE_BLOCK) {
        s_pos[n_own + i] = r.pos[halo_idx[h0 + i]];
        s_f[2 * (n_own + i)] = 0;
        s_f[2 * (n_own + i) + 1] = 0;
    }
    if (MAT != 0)
        for (uint32_t i = tid; i < nmat * SB_MAT_ROW; i += SB_TILE_BLOCK) s_mat[i] = mat_tab[i];
    // SB_COLLIDE_GRID: the length of each particle's neighbour list is fetched here, long before phase 2 walks it
    // (on the substep right after a hash build the lists do not exist yet: phase 2 makes them)
    uint32_t ncand[SB_UNROLL];
    bool fresh = false;
    float drift_x = 0.0f, drift_y = 0.0f; // SbGridCtl: the common displacement this substep is measured against
    float drift_Cx = 0.0f, drift_Cy = 0.0f; // ... and the drift accumulated since the hash in use was built
    if (MODE == SB_COLLIDE_GRID) {
#pragma unroll
        for (int u = 0; u < SB_UNROLL; u++) { // (requested whatever the decision will be: on a list-making substep the main pass is skipped)
            const uint32_t i = tid + (uint32_t)u * SB_TILE_BLOCK;
            ncand[u] = i < n_own ? grid_all.nl_count[p0 + i] : 0u;
        }
        SB_STAMP(job.step, 0);
        sb_grid_stage(job.step, early, s_grid, 1u);
        SB_STAMP(job.step, 1);
    }
    __syncthreads();
    SbGrid grid = grid_all;
    bool pushing = false;
    if (MODE == SB_COLLIDE_GRID) {
        SB_STAMP(job.step, 2);
        // what every thread needs of the decision: lane 0 of wave 0 has left it in LDS (uniform: broadcast reads).  The whole
        // block is thread 0's business -- in workgroup 0, which publishes it for the next launch, and on the rare substeps that
        // make lists or push the next hash, where every workgroup reads it (behind a barrier)
        const SbGridFlags gf = s_grid.hot.f;
        const bool whole = blockIdx.x == 0 || __builtin_amdgcn_readfirstlane(gf.fresh | gf.pushing | gf.abort) != 0u;
        if (whole && tid == 0) sb_grid_decide(job.step, grid_all, s_grid, 1u, true);
        drift_x = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s_grid.hot.cx))); // uniform: SGPRs
        drift_y = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s_grid.hot.cy)));
        drift_Cx = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s_grid.hot.Cx)));
        drift_Cy = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s_grid.hot.Cy)));
        SB_STAMP(job.step, 3);
        if (__builtin_amdgcn_readfirstlane(gf.abort) != 0u) return; // (nothing has been written; workgroup 0 has told the launches behind)
        if (blockIdx.x == 0 && tid < SB_GRID_SLOTS) job.step.slots_zero[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        fresh = __builtin_amdgcn_readfirstlane(gf.fresh) != 0u;
        pushing = __builtin_amdgcn_readfirstlane(gf.pushing) != 0u;
        SB_STAMP(job.step, 4);
    }

    // Lagged schedule (SbGridCtl): the tail of the substep before this one ordered the next hash.  Every workgroup pushes its own
    // particles -- the READ state, which sits in LDS -- into the OTHER hash buffer; the lists in use serve this substep, the
    // next one makes its lists from what is pushed here.  Nobody reads these stores before the launch has retired.
    if (MODE == SB_COLLIDE_GRID && pushing) {
        __syncthreads(); // (thread 0 has chosen the geometry)
        const SbGridGeom pg = sb_grid_geom_load(&s_grid.now.pgeo);
        const uint32_t buf = __builtin_amdgcn_readfirstlane(s_grid.now.cur) ^ 1u;
        uint32_t n_out = 0u;
        // (two particles per thread side by side: both returning exchanges in flight before either record needs its answer)
#pragma unroll 1
        for (uint32_t i0 = tid; i0 < n_own; i0 += 2u * SB_TILE_BLOCK) {
            const uint32_t i1 = i0 + SB_TILE_BLOCK;
            const bool two = i1 < n_own;
            const float2 pa0 = s_pos[i0], pa1 = s_pos[two ? i1 : i0];
            const uint32_t sl0 = job.pslot[p0 + i0], sl1 = job.pslot[p0 + (two ? i1 : i0)];
            uint32_t c0, c1 = 0u;
            bool o0 = false, o1 = false;
            const unsigned long long old0 = sb_grid_push_begin(job.build, buf, pg, p0 + i0, pa0, &c0, &o0);
            unsigned long long old1 = 0ull;
            if (two) old1 = sb_grid_push_begin(job.build, buf, pg, p0 + i1, pa1, &c1, &o1);
            n_out += (o0 ? 1u : 0u) + (o1 ? 1u : 0u);
            sb_grid_push_end(job.build, buf, pg, p0 + i0, pa0, sl0, c0, old0);
            if (two) sb_grid_push_end(job.build, buf, pg, p0 + i1, pa1, sl1, c1, old1);
        }
        sb_grid_count_outside(&job.build.outside[pg.gen & 3u], n_out);
    }

    // Beam phase.  The slice is walked in batches of SB_UNROLL x SB_TILE_BLOCK copies: all global loads of a
    // batch are issued back to back (and the next batch's before this one is evaluated), so a wave
    // keeps ~2 x 3 x SB_UNROLL loads in flight instead of paying two dependent HBM latencies per beam.
    const uint32_t lmask = (1u << lbits) - 1u;
    uint32_t wd[SB_UNROLL], wn[SB_UNROLL];
    float tg[SB_UNROLL], ls[SB_UNROLL], ln[SB_UNROLL], tgn[SB_UNROLL], lsn[SB_UNROLL], lnn[SB_UNROLL];
    auto fetch = [&](uint32_t j0, uint32_t *wo, float *to, float *lo, float *leno) {
#pragma unroll
        for (int u = 0; u < SB_UNROLL; u++) {
            const uint32_t j = j0 + (uint32_t)u * SB_TILE_BLOCK;
            const bool ok = j < nb;
            const uint32_t cidx = b0 + (ok ? j : 0u);
            wo[u] = ok ? b.pair[cidx] : 0xFFFFFFFFu;
            to[u] = b.target[cidx];
            lo[u] = b.last[cidx];
            leno[u] = (MAT != 2) ? b.length[cidx] : 0.0f;
        }
    };
    if (nb) fetch(tid, wd, tg, ls, ln);
    for (uint32_t j0 = tid; j0 < nb + tid; j0 += SB_UNROLL * SB_TILE_BLOCK) {
        const uint32_t jn = j0 + SB_UNROLL * SB_TILE_BLOCK;
        if (jn < nb + tid) fetch(jn, wn, tgn, lsn, lnn);
#pragma unroll
        for (int u = 0; u < SB_UNROLL; u++) {
            const uint32_t word = wd[u];
            if (word != 0xFFFFFFFFu) { // padding, out of range, or removed by a delete pass
                const uint32_t c = b0 + j0 + (uint32_t)u * SB_TILE_BLOCK;
                const uint32_t la = word & lmask, lb = (word >> lbits) & lmask;
                float length, inv_length, spring, damp, yield, limit;
                if (MAT != 0) {
                    const uint32_t m = SB_MAT_ROW * (word >> (2u * lbits)); // 32-bit LDS index
                    length = MAT == 2 ? s_mat[m] : ln[u];
                    spring = s_mat[m + 1u];
                    damp = s_mat[m + 2u];
                    yield = s_mat[m + 3u];
                    limit = s_mat[m + 4u];
                    inv_length = MAT == 2 ? s_mat[m + 5u] : sb_div(1.0f, length);
                } else {
                    length = ln[u];
                    spring = b.spring[c];
                    damp = b.damp[c];
                    yield = b.yield[c];
                    limit = b.limit[c];
                    inv_length = sb_div(1.0f, length);
                }
                const float target = tg[u];
#if SB_ABLATE & 1 // diagnostic build: beam arithmetic replaced by a data-dependent dummy
                SbBeamResult res;
                {
                    float2 qa = s_pos[la], qb = s_pos[lb];
                    res.target_length = target;
                    res.last_length = ls[u] + (qa.x - qb.x) * spring + length * damp + yield * limit + inv_length;
                    res.strain = res.stress = 0.f;
                    res.ax = __float_as_int(qa.y);
                    res.ay = __float_as_int(qb.y);
                    res.bx = res.ax ^ 5;
                    res.by = res.ay ^ 9;
                    res.broken = false;
                }
#else
                SbBeamResult res = sb_beam_eval<AUX>(s_pos[la], s_pos[lb], length, inv_length, target, ls[u], spring, damp,
                                                     yield, limit);
#endif
                // target_length only moves on plastic yield (compute.wgsl:113-116): store it when it did
                if (__float_as_uint(res.target_length) != __float_as_uint(target)) b.target[c] = res.target_length;
                sb_store_wt(&b.last[c], res.last_length);
                if (AUX) { // strain/stress: outputs only (:122-123), stored by the last substep of a call
                    b.strain[c] = res.strain;
                    b.stress[c] = res.stress;
                }
#if !(SB_ABLATE & 32) // diagnostic build: no LDS force accumulation
                // both endpoints unconditionally: a halo endpoint lands in an accumulator nobody reads,
                // which is cheaper than two exec-mask branches per beam
                atomicAdd(&s_f[2 * la], res.ax);
                atomicAdd(&s_f[2 * la + 1], res.ay);
                atomicAdd(&s_f[2 * lb], res.bx);
                atomicAdd(&s_f[2 * lb + 1], res.by);
#endif
                if (res.broken) atomicOr(&broken[c >> 5], 1u << (c & 31));
            }
        }
#pragma unroll
        for (int u = 0; u < SB_UNROLL; u++) {
            wd[u] = wn[u];
            tg[u] = tgn[u];
            ls[u] = lsn[u];
            ln[u] = lnn[u];
        }
    }
    __syncthreads();

    if (MODE == SB_COLLIDE_GRID) {
        SB_STAMP(job.step, 5);
        // a substep that makes its lists: from the block thread 0 decided (behind the barrier that ends the beam phase);
        // otherwise the hash, its geometry and its age are those of the block this launch read
        // (everything uniform, and told so: the hash pointers are selected in scalar registers, not per lane)
        const SbGridCtl &blk = fresh ? s_grid.now : s_grid.prev;
        const uint32_t cur = __builtin_amdgcn_readfirstlane(blk.cur);
        if (fresh) {
            drift_Cx = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s_grid.now.Cx)));
            drift_Cy = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s_grid.now.Cy)));
        }
        grid = sb_grid_view(grid_all, cur, &blk.geo, drift_Cx, drift_Cy);
    }
    // Phase 2: consume the complete force sums (compute.wgsl:171-201) -> WRITE state.
    bool any_acc = false;
    float moved = 0.0f;
    // `slow` (compile-time at each call site): this particle takes sb_collide_slow instead of walking its list
    auto finish = [&](uint32_t i, float2 vel, float2 acc, uint32_t count, auto slow) {
        const uint32_t g = p0 + i;
        SbParticle particle;
        particle.p = s_pos[i];
        particle.v = vel;
        particle.a = acc;
        const float2 p_old = particle.p;
        if (MODE == SB_COLLIDE_GRID) {
            const SbParticle self = particle; // :141
#if !(SB_ABLATE & 64) // diagnostic build: no collision scan
            const float ec = sb_div(c.elasticity + 1.0f, 2.0f); // :143
            if (decltype(slow)::value) sb_collide_slow(grid, fresh, prm, c.friction, ec, particle, self, g, pidx, r.pos, r.vel);
            else sb_collide_list(grid, count, prm, c.friction, ec, particle, self, g, pidx, r.pos, r.vel);
#endif
        }
#if SB_ABLATE & 2 // diagnostic build: particle arithmetic replaced by a data-dependent dummy
        particle.v.x += (float)s_f[2 * i] * prm.time_step;
        particle.v.y += (float)s_f[2 * i + 1] * c.gravity_y;
#else
        sb_particle_finish(prm, c, particle, s_f[2 * i], s_f[2 * i + 1]);
#endif
        if (MODE == SB_COLLIDE_GRID) {
            const float dx = particle.p.x - p_old.x, dy = particle.p.y - p_old.y;
            moved = fmaxf(moved, fmaxf(sb_abs(dx - drift_x), sb_abs(dy - drift_y)) * 1.4142137f);
            if (i == 0u) sb_store_sample_displacement(job.step.slots_out, dx, dy); // i == 0 is thread 0's first particle
        }
        sb_store_wt(&w.pos[g], particle.p); // (write-through: sb_physics.h)
        sb_store_wt(&w.vel[g], particle.v);
        const bool nz = (__float_as_uint(particle.a.x) | __float_as_uint(particle.a.y)) != 0u; // -0.0 counts
        any_acc |= nz;
        if (nz || acc_w_dirty) w.acc[g] = particle.a;
    };
    // A substep that makes its lists: the whole tile's, by the workgroup together through LDS (sb_lists_cooperative) -- after which
    // it is a substep like any other, walked by the main pass below with the lengths just written.  A tile whose particles have
    // scattered beyond what the LDS area holds has every particle walk the hash for itself in the second pass instead.
    if (MODE == SB_COLLIDE_GRID && fresh && job.coop_bytes != 0u) {
        const SbGridGeom gm = sb_grid_geom_load(grid.geo);
        SB_STAMP(job.step, 8);
        if (sb_lists_cooperative(grid, gm, p0, n_own, job.pslot, job.islot, sb_lds + job.coop_off, job.coop_bytes
#ifdef SB_STAMPS
                                 , job.step, sb_t_start
#endif
                                 )) { // (uniform)
            fresh = false;
#pragma unroll
            for (int u = 0; u < SB_UNROLL; u++) {
                const uint32_t i = tid + (uint32_t)u * SB_TILE_BLOCK;
                ncand[u] = i < n_own ? grid.nl_count[p0 + i] : 0u; // (written by this very thread)
            }
            SB_STAMP(job.step, 9);
        }
    }
    // main pass: every particle whose list can simply be walked (all of them, on almost every substep)
    bool any_slow = false;
    if (!fresh) {
#pragma unroll
        for (int u = 0; u < SB_UNROLL; u++) {
            const uint32_t i = tid + (uint32_t)u * SB_TILE_BLOCK;
            if (i < n_own) {
                const uint32_t count = MODE == SB_COLLIDE_GRID ? ncand[u] : 0u;
                if (MODE == SB_COLLIDE_GRID && count == SB_NL_OVERFLOW) any_slow = true;
                else finish(i, pv[u], pa[u], count, std::false_type{});
            }
        }
        for (uint32_t i = tid + SB_UNROLL * SB_TILE_BLOCK; i < n_own; i += SB_TILE_BLOCK) {
            const uint32_t count = MODE == SB_COLLIDE_GRID ? grid.nl_count[p0 + i] : 0u;
            if (MODE == SB_COLLIDE_GRID && count == SB_NL_OVERFLOW) any_slow = true;
            else finish(i, r.vel[p0 + i], acc_r ? r.acc[p0 + i] : make_float2(0.f, 0.f), count, std::false_type{});
        }
    }
    // second pass, SB_COLLIDE_GRID only: the substep after a hash build (every particle's list is made first), or the
    // particles of piles that overflow their lists
    // second pass, SB_COLLIDE_GRID only: the substep after a hash build where the tile could not make its lists together (every
    // particle makes its own first), or the particles of piles that overflow their lists
    if (MODE == SB_COLLIDE_GRID && (fresh || __syncthreads_or(any_slow ? 1 : 0))) {
#pragma unroll 1
        for (uint32_t i = tid; i < n_own; i += SB_TILE_BLOCK) {
            if (!fresh && grid.nl_count[p0 + i] != SB_NL_OVERFLOW) continue;
            finish(i, r.vel[p0 + i], acc_r ? r.acc[p0 + i] : make_float2(0.f, 0.f), 0u, std::true_type{});
        }
    }
#if !(SB_ABLATE & 128) // diagnostic build: no displacement tracking
    if (MODE == SB_COLLIDE_GRID) sb_store_block_displacement(job.step.slots_out, moved);
#endif
    const int wg_any = __syncthreads_or(any_acc ? 1 : 0);
    if (tid == 0) acc_flag_w[tile] = wg_any ? 1u : 0u;
    if (MODE == SB_COLLIDE_GRID) SB_STAMP(job.step, 6);
#ifdef SB_STAMPS
    if (MODE == SB_COLLIDE_GRID && blockIdx.x == gridDim.x / 2u && threadIdx.x == 0u && job.step.stamps[9] > job.step.stamps[8] && job.step.stamps[8] > job.step.stamps[5]) {
        for (int k = 0; k < 12; k++) job.step.stamps[16 + k] = job.step.stamps[k]; // the last launch that made its lists together
        job.step.stamps[8] = job.step.stamps[9] = 0u;
    }
#endif
}

// The two entry points.  With the collision walk compiled in, the body wants 71 VGPRs: 7 waves per SIMD, i.e.
// three 8-wave workgroups per CU and 768 resident slots for ~1000 tiles (a second, mostly empty round).  Capping
// it at 64 VGPRs (a few spilled lanes) keeps four workgroups per CU; the collision-free variants (50-58 VGPRs)
// are left alone.
template <int MAT, bool AUX>
__global__ __launch_bounds__(SB_TILE_BLOCK) void k_substep_tiled(SB_TILED_PARAMS)
{
    sb_substep_tiled<SB_COLLIDE_OFF, MAT, AUX>(SB_TILED_ARGS);
}
template <int MAT, bool AUX>
__global__ __launch_bounds__(SB_TILE_BLOCK) __attribute__((amdgpu_waves_per_eu(SB_GRID_WAVES, 8))) void k_substep_tiled_grid(SB_TILED_PARAMS)
{
    sb_substep_tiled<SB_COLLIDE_GRID, MAT, AUX>(SB_TILED_ARGS);
}

// ---------------------------------------------------------------- spatial hash build (the helper launch)
// r04: the per-substep helper launch of rounds 1-3 (k_grid_maintain: reduce the displacement slots, decide, rebuild when the
// bound demands it) is gone from the steady state.  The decision is taken by the last workgroup of every particle kernel
// (sb_grid_tail, sb_physics.h); in the lagged schedule the particle kernels also push the hash themselves, one substep ahead.
// What is left is this helper, launched by the host
//   * forced: when the HOST knows the lists are worthless (first substep after an upload, ghost refresh, the hybrid's fresh
//     start, recovery from an abort): builds from the current state and publishes the state the next substep reads;
//   * unforced, in front of every substep of a CLASSIC stretch (violent scenes, the atomic path): builds when the tail of the
//     substep before ordered it (`need_build`), else returns at once.
// ONE pass, no device-wide barrier, no residency requirement (sb_grid_push_begin / _end).  (Round 1 counted, scanned and
// scattered into cell-sorted records: three device-wide barriers, 70-90 us per build; tools/grid_phases.py.)
#define SB_BUILD_BLOCKS 128u // (r03, pile / soup: 32 -> 38.9 / 57.4, 64 -> 37.7 / 53.9, 128 -> 37.3 / 52.8 us per substep)
#ifndef SB_MT
#define SB_MT 1024u // threads per workgroup of k_grid_build
#endif

__global__ __launch_bounds__(SB_MT) void k_grid_build(SbGridStep t, uint32_t force, const float2 *__restrict__ pos,
                                                      const uint32_t *__restrict__ pslot, SbGrid g, SbGridBuild w)
{
    __shared__ SbGridShared s_grid;
    const uint32_t tid = threadIdx.x;
    const SbGridEarly early = sb_grid_begin(t);
    sb_grid_stage(t, early, s_grid, 1u);
    __syncthreads();
    if (tid == 0) {
        if (force) { // the host knows the lists are worthless: a hash of the current state, whatever the bound says.  Every
            // workgroup computes the same block from the same words; block 0 publishes it, `settled`: the substep that
            // follows adopts it as it stands
            const SbGridCtl &E = s_grid.prev;
            SbGridCtl &N = s_grid.now; // (a copy of E so far: sb_grid_stage)
            N.fresh = 1u;
            N.need_build = N.pushing = N.abort = 0u;
            N.settled = N.transient = 1u;
            N.cur = E.cur ^ 1u;
            N.builds = E.builds + 1u;
            N.since = 1u;
            N.accum = N.Cx = N.Cy = 0.0f;
            N.geo = sb_grid_geom_for(g, E.geo.skin, (E.geo.wide != 0u || E.wide_next != 0u) ? 1u : 0u);
            N.geo.gen = N.builds;
            if (blockIdx.x == 0u) {
                const uint32_t *src = (const uint32_t *)&s_grid.now;
                uint32_t *dst = (uint32_t *)(t.ctl + (t.par ^ 1u));
                for (uint32_t k = 0; k < sizeof(SbGridCtl) / 4u; k++) SB_AGENT_STORE(&dst[k], src[k]);
                SB_AGENT_STORE(&t.outside[(N.geo.gen + 1u) & 3u], 0u);
            }
        } else {
            sb_grid_decide(t, g, s_grid, 1u, false); // what the substep behind this launch will decide
        }
    }
    __syncthreads();
    if (!force && (s_grid.now.need_build == 0u || s_grid.now.abort != 0u)) return;
    const SbGridGeom geo = sb_grid_geom_load(&s_grid.now.geo);
    const uint32_t buf = __builtin_amdgcn_readfirstlane(s_grid.now.cur), P = t.P;
    // ---- every particle: its cell, itself pushed on the front of that cell's list, its record
    // (four particles per thread and round, so that four returning atomics are in flight per lane)
    const uint32_t nthreads = gridDim.x * SB_MT, gtid = blockIdx.x * SB_MT + tid;
    uint32_t n_out = 0u;
    for (uint32_t i0 = gtid; i0 < P; i0 += 4u * nthreads) {
        uint32_t c[4], slot[4];
        unsigned long long old[4];
        float2 p[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = i0 + (uint32_t)u * nthreads;
            p[u] = i < P ? pos[i] : make_float2(0.f, 0.f);
            slot[u] = i < P ? pslot[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = i0 + (uint32_t)u * nthreads;
            if (i < P) {
                bool o = false;
                old[u] = sb_grid_push_begin(w, buf, geo, i, p[u], &c[u], &o);
                n_out += o ? 1u : 0u;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = i0 + (uint32_t)u * nthreads;
            if (i < P) sb_grid_push_end(w, buf, geo, i, p[u], slot[u], c[u], old[u]);
        }
    }
    sb_grid_count_outside(&w.outside[geo.gen & 3u], n_out);
}

// The decision ahead of time, for the host's look between launches (sb_api.hip hybrid_substeps): what the next substep would
// decide from the displacement slots of the last one, published `settled` -- the substep that follows adopts it as it stands.
__global__ __launch_bounds__(SB_GRID_SLOTS) void k_grid_settle(SbGridStep t, SbGrid g)
{
    __shared__ SbGridShared s_grid;
    const SbGridEarly early = sb_grid_begin(t);
    sb_grid_stage(t, early, s_grid, 0u);
    __syncthreads();
    if (threadIdx.x == 0) sb_grid_decide(t, g, s_grid, 0u, true);
}

// ---------------------------------------------------------------- delete pass (compute.wgsl:205-246)

// Canonical semantics: every copy of a beam flagged since the last pass stops
// acting; its mapping slot records the number of the pass that removed it.  The host replays
// the stable in-place compactions, pass by pass, when it reads the mapping back.
__global__ __launch_bounds__(SB_BLOCK) void k_delete(SbBeamArrays b, uint32_t nwords, uint32_t nbeam,
                                                     uint32_t *broken, uint32_t *dead_gen, uint32_t gen, int tiled)
{
    uint32_t w = blockIdx.x * SB_BLOCK + threadIdx.x;
    if (w >= nwords) return;
    uint32_t bits = broken[w];
    if (!bits) return;
    broken[w] = 0;
    while (bits) {
        uint32_t k = __ffs(bits) - 1;
        bits &= bits - 1;
        uint32_t i = w * 32 + k;
        if (i >= nbeam) break;
        if (tiled) b.pair[i] = 0xFFFFFFFFu;
        else b.ia[i] = 0xFFFFFFFFu;
        dead_gen[b.slot[i]] = gen; // both copies of a cut beam store the same value
    }
}

// ---------------------------------------------------------------- halo exchange helpers

// send lists -> packed float buffer (6 floats per particle, 2 per beam, at the configured offsets)
// What an owner sends for one of its beams: {target_length, last_length} -- or, once its delete pass has removed the beam,
// a last_length of SB_HALO_DEAD (a NaN payload no arithmetic produces): the ghost copies on the neighbour die on the OWNER's
// word only, never on their own redundant evaluation, whose inputs may already have been invalid.
#define SB_HALO_DEAD 0x7FC0DEADu
SB_DEV float2 sb_halo_beam_record(const SbBeamArrays &b, uint32_t cpy, const uint32_t *__restrict__ dead_gen)
{
    if (dead_gen[b.slot[cpy]] != 0u) return make_float2(0.0f, __uint_as_float(SB_HALO_DEAD));
    return make_float2(b.target[cpy], b.last[cpy]);
}

__global__ __launch_bounds__(SB_BLOCK) void k_halo_pack(SbParticleArrays c, SbBeamArrays b,
                                                        const uint32_t *__restrict__ plist,
                                                        const uint32_t *__restrict__ poff, uint32_t np,
                                                        const uint32_t *__restrict__ blist,
                                                        const uint32_t *__restrict__ boff, uint32_t nb, float *dst,
                                                        const uint32_t *__restrict__ dead_gen)
{
    uint32_t k = blockIdx.x * SB_BLOCK + threadIdx.x;
    if (k < np) {
        uint32_t i = plist[k];
        float2 p = c.pos[i], v = c.vel[i], a = c.acc[i];
        float2 *o = (float2 *)(dst + poff[k]);
        o[0] = p;
        o[1] = v;
        o[2] = a;
    } else if (k < np + nb) {
        uint32_t j = k - np, cpy = blist[j];
        *(float2 *)(dst + boff[j]) = sb_halo_beam_record(b, cpy, dead_gen);
    }
}

// What a refresh may do to the per-tile promises: a ghost particle that arrives with an acceleration breaks
// "every acceleration of this tile is zero", a ghost beam whose target arrives changed (its owner's copy yielded) breaks "no beam
// of this tile has ever yielded" -- for THAT tile (found by bisection of the tile tables; both events are rare), not for all of
// them as three memsets per refresh used to say.  (The spatial hash rebins after a refresh -- ghosts jumped: the host's business,
// sbk_launch_halo_unpack.)
struct SbHaloFlags {
    uint32_t *acc_flag;          // the current particle buffer's flags (nullptr: no tiles)
    const uint32_t *tile_p0;     // [ntiles + 1]
    uint32_t ntiles;
    uint32_t *plastic0, *plastic1; // blocked layout only (else nullptr), with
    const uint32_t *tile_b0;     // its beams per tile
};
SB_DEV uint32_t sb_range_of(const uint32_t *__restrict__ first, uint32_t n, uint32_t i) // largest t < n with first[t] <= i
{
    uint32_t lo = 0u, hi = n;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (first[mid] <= i) lo = mid;
        else hi = mid;
    }
    return lo;
}

// packed float buffer -> ghost lists (every device copy of a ghost beam is refreshed)
__global__ __launch_bounds__(SB_BLOCK) void k_halo_unpack(SbParticleArrays c, SbBeamArrays b,
                                                          const uint32_t *__restrict__ plist,
                                                          const uint32_t *__restrict__ poff, uint32_t np,
                                                          const uint2 *__restrict__ blist,
                                                          const uint32_t *__restrict__ boff, uint32_t nbc,
                                                          const float *__restrict__ src, uint32_t *broken,
                                                          const uint32_t *__restrict__ dead_gen, SbHaloFlags flags)
{
    uint32_t k = blockIdx.x * SB_BLOCK + threadIdx.x;
    if (k < np) {
        uint32_t i = plist[k];
        const float2 *in = (const float2 *)(src + poff[k]);
        const float2 a = in[2];
        c.pos[i] = in[0];
        c.vel[i] = in[1];
        c.acc[i] = a;
        if (flags.acc_flag && ((__float_as_uint(a.x) | __float_as_uint(a.y)) != 0u)) // (-0.0 counts, as in the substep kernels)
            SB_AGENT_STORE(&flags.acc_flag[sb_range_of(flags.tile_p0, flags.ntiles, i)], 1u);
    } else if (k < np + nbc) {
        uint2 e = blist[k - np];
        float2 tl = *(const float2 *)(src + boff[e.y]);
        if (__float_as_uint(tl.y) == SB_HALO_DEAD) { // removed by its owner: flag this copy for sb_halo_delete_ghosts
            if (dead_gen[b.slot[e.x]] == 0u) atomicOr(&broken[e.x >> 5], 1u << (e.x & 31u));
        } else {
            if (flags.plastic0 && __float_as_uint(b.target[e.x]) != __float_as_uint(tl.x)) {
                const uint32_t t = sb_range_of(flags.tile_b0, flags.ntiles, e.x);
                SB_AGENT_STORE(&flags.plastic0[t], 1u);
                SB_AGENT_STORE(&flags.plastic1[t], 1u);
            }
            b.target[e.x] = tl.x;
            b.last[e.x] = tl.y;
        }
    }
}

// Ghost copies evaluate their own break condition like everybody else, but on inputs that may be invalid near the outer edge
// of the ghost zone: their flags are dropped before a delete pass (sb_delete_pass on an engine with a halo).
__global__ __launch_bounds__(SB_BLOCK) void k_halo_clear_ghost_flags(const uint2 *__restrict__ blist, uint32_t nbc, uint32_t *broken)
{
    uint32_t k = blockIdx.x * SB_BLOCK + threadIdx.x;
    if (k < nbc) {
        const uint32_t c = blist[k].x;
        atomicAnd(&broken[c >> 5], ~(1u << (c & 31u)));
    }
}

// ---------------------------------------------------------------- direct peer exchange
#define SB_MAILBOX_FLAGS_BYTES 256

struct SbPeerRoute {
    uint32_t n;
    uint32_t begin[SB_MAX_PEERS], end[SB_MAX_PEERS]; // my packed send layout, floats
    float *dst[SB_MAX_PEERS];                        // where begin[j] lands in neighbour j's current receive buffer
};

__device__ __forceinline__ float *sb_peer_route(const SbPeerRoute &r, uint32_t off)
{
    float *o = nullptr;
#pragma unroll
    for (uint32_t j = 0; j < SB_MAX_PEERS; j++)
        if (j < r.n && off >= r.begin[j] && off < r.end[j]) o = r.dst[j] + (off - r.begin[j]);
    return o;
}

// k_halo_pack with the destination resolved per record: the stores go over xGMI into the neighbours'
// fine-grained mailboxes; they are complete (acknowledged) when the kernel retires.
__global__ __launch_bounds__(SB_BLOCK) void k_halo_pack_peer(SbParticleArrays c, SbBeamArrays b,
                                                             const uint32_t *__restrict__ plist,
                                                             const uint32_t *__restrict__ poff, uint32_t np,
                                                             const uint32_t *__restrict__ blist,
                                                             const uint32_t *__restrict__ boff, uint32_t nb,
                                                             SbPeerRoute route, const uint32_t *__restrict__ dead_gen)
{
    // one float2 per lane, consecutive lanes -> consecutive 8-byte words of the mailbox: the stores that
    // cross xGMI are fully coalesced (a lane per 24-byte record would leave every 64-byte packet a third full)
    uint32_t t = blockIdx.x * SB_BLOCK + threadIdx.x;
    if (t < 3 * np) {
        uint32_t k = t / 3, part = t - 3 * k, i = plist[k];
        const float2 *src = part == 0 ? c.pos : part == 1 ? c.vel : c.acc;
        float2 *o = (float2 *)sb_peer_route(route, poff[k]);
        if (o) o[part] = src[i];
    } else if (t < 3 * np + nb) {
        uint32_t j = t - 3 * np, cpy = blist[j];
        float2 *o = (float2 *)sb_peer_route(route, boff[j]);
        if (o) *o = sb_halo_beam_record(b, cpy, dead_gen);
    }
}

struct SbPeerSignal {
    uint32_t n, seq;
    uint32_t *remote[SB_MAX_PEERS]; // my flag in each neighbour's mailbox
    uint32_t *local;                // my mailbox's flags: slot j is written by neighbour j
    uint64_t limit_ticks;           // wall_clock64 ticks (100 MHz) before the wait gives up
    uint32_t *err;                  // pinned host word
};

// One lane per neighbour: publish "my data for exchange `seq` is in your mailbox", then wait until the
// neighbour has said the same.  Runs as its own launch between pack and unpack, so the data stores are
// released before the flag and the unpack kernel starts (and acquires) after the flag was seen.  The
// wait is bounded: a neighbour that never arrives sets *err instead of leaving a wave spinning.
__global__ __launch_bounds__(64) void k_peer_signal_wait(SbPeerSignal s)
{
    uint32_t j = threadIdx.x;
    if (j >= s.n) return;
    __hip_atomic_store(s.remote[j], s.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    uint64_t t0 = wall_clock64();
    while ((int32_t)(__hip_atomic_load(&s.local[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - s.seq) < 0) {
        if (wall_clock64() - t0 > s.limit_ticks) {
            __hip_atomic_fetch_or(s.err, 1u << j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}

// ---------------------------------------------------------------- launchers

static inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// the schedule of the next substep's hash work (SbGridCtl): lagged unless the engine is inside a classic stretch, runs the
// atomic path (its particle kernel does not push), or SB_GRID_MODE says otherwise (A/B runs, the fallback's tests)
uint32_t sbk_grid_mode(const sb_engine *e)
{
    static const int forced = [] {
        const char *v = getenv("SB_GRID_MODE");
        return !v ? -1 : (!strcmp(v, "classic") ? (int)SB_GRID_CLASSIC : (!strcmp(v, "lagged") ? (int)SB_GRID_LAGGED : -1));
    }();
    if (e->path != SB_PATH_TILED) return SB_GRID_CLASSIC;
    if (forced >= 0) return (uint32_t)forced;
    return e->grid_classic_left ? SB_GRID_CLASSIC : SB_GRID_LAGGED;
}

// what a launch that decides needs: the control blocks and the displacement slots by the number of the substep it is (or is in front of)
static SbGridStep sbk_grid_step(const sb_engine *e, uint32_t sched)
{
    const uint32_t nblk = e->path == SB_PATH_TILED ? e->ntiles : cdiv(e->P, SB_BLOCK);
    const uint64_t k = e->grid_executed; // substeps run so far (64 bits: k % 3 has to go on across 2^32 substeps, two days of a busy engine)
    float4 *slots = (float4 *)e->d_blk_max;
    SbGridStep st{e->d_grid_ctl, e->grid_par, sched, slots + (size_t)(k % 3u) * SB_GRID_SLOTS, slots + (size_t)((k + 1u) % 3u) * SB_GRID_SLOTS,
                  slots + (size_t)((k + 2u) % 3u) * SB_GRID_SLOTS, 1.0f / (float)std::min<uint32_t>(std::max(nblk, 1u), SB_GRID_SLOTS), e->d_grid_outside, e->P};
#ifdef SB_STAMPS
    st.stamps = e->dev_err + 4; // (pinned host memory, mapped: sb_get_info "grid_stamp_<k>")
#endif
    return st;
}
static SbGridBuild sbk_grid_arrays(const sb_engine *e)
{
    return SbGridBuild{{e->d_head[0], e->d_head[1]}, {e->d_cell_of[0], e->d_cell_of[1]}, {e->d_rec[0], e->d_rec[1]}, e->d_grid_outside};
}
static uint32_t sbk_build_blocks(const sb_engine *e)
{
    static const uint32_t max_blocks = [] { // tuning knob (no residency requirement: the build has no device-wide barrier)
        const char *v = getenv("SB_MAINTAIN_BLOCKS");
        const long n = v ? atol(v) : 0;
        return n >= 1 && n <= 4096 ? (uint32_t)n : SB_BUILD_BLOCKS;
    }();
    return std::min(std::max(cdiv(e->P, SB_MT * 4u), 1u), max_blocks);
}

// the host's look between launches: the decision ahead of time (e->d_grid_ctl[e->grid_par] then holds it, `settled`)
void sbk_launch_grid_settle(sb_engine *e)
{
    k_grid_settle<<<1, SB_GRID_SLOTS, 0, e->stream>>>(sbk_grid_step(e, sbk_grid_mode(e)), e->grid);
    e->grid_par ^= 1u;
}

void sbk_launch_substep(sb_engine *e, bool write_aux)
{
    SbParticleArrays r = e->part[e->cur], w = e->part[e->cur ^ 1];
    const uint32_t mode = e->opt.collision_mode;
    SbGridJob job{};
    if (mode == SB_COLLIDE_GRID && e->P) {
        const uint32_t sched = sbk_grid_mode(e);
        const SbGridBuild gb = sbk_grid_arrays(e);
        // the helper: forced when the host knows the lists are worthless; unforced in front of every substep of the classic
        // schedule (it takes the substep's decision itself and builds when that says so); not at all in the lagged steady state
        if (e->grid_force) {
            k_grid_build<<<sbk_build_blocks(e), SB_MT, 0, e->stream>>>(sbk_grid_step(e, sched), 1u, r.pos, e->d_pslot, e->grid, gb);
            e->grid_par ^= 1u; // (it published the state the substep adopts)
            e->grid_force = false;
            e->grid_helper_launches++;
        } else if (sched == SB_GRID_CLASSIC) {
            k_grid_build<<<sbk_build_blocks(e), SB_MT, 0, e->stream>>>(sbk_grid_step(e, sched), 0u, r.pos, e->d_pslot, e->grid, gb);
            e->grid_helper_launches++;
        }
        job.step = sbk_grid_step(e, sched);
        job.build = gb;
        job.pslot = e->d_pslot;
        job.islot = e->d_islot;
        job.coop_off = e->lds_coop_off;
        job.coop_bytes = e->d_islot ? e->lds_coop : 0u;
    }
    if (e->path == SB_PATH_ATOMIC) {
        if (e->nbeam)
            k_beams_atomic<<<cdiv(e->nbeam, SB_BLOCK), SB_BLOCK, 0, e->stream>>>(e->beams, e->nbeam, r.pos,
                                                                               e->d_forces, e->d_broken);
        if (e->P) {
            dim3 g(cdiv(e->P, SB_BLOCK));
#define SB_LAUNCH_P(M) k_particles<M><<<g, SB_BLOCK, 0, e->stream>>>(r, w, e->d_forces, e->P, e->consts, e->prm, e->d_pidx, e->grid, job)
            if (mode == SB_COLLIDE_ALLPAIRS) SB_LAUNCH_P(SB_COLLIDE_ALLPAIRS);
            else if (mode == SB_COLLIDE_GRID) SB_LAUNCH_P(SB_COLLIDE_GRID);
            else SB_LAUNCH_P(SB_COLLIDE_OFF);
#undef SB_LAUNCH_P
        }
    } else if (e->ntiles) {
    const size_t lds_launch = (mode == SB_COLLIDE_GRID && job.coop_bytes) ? (size_t)e->lds_coop_off + e->lds_coop : e->lds_bytes;
#define SB_LAUNCH_T(K, T, A) K<T, A><<<e->ntiles, SB_TILE_BLOCK, lds_launch, e->stream>>>(                            \
        r, w, e->beams, e->d_tile_p0, e->d_tile_b0, e->d_tile_h0, e->d_halo_idx, e->ntiles, e->tile_cap_all,       \
        e->tile_cap_own, e->lbits, e->d_mat, e->nmat, e->consts, e->prm, e->d_broken, e->d_pidx, e->grid,          \
        job, e->d_acc_flag[e->cur], e->d_acc_flag[e->cur ^ 1])
#define SB_LAUNCH_TA(K, T) do { if (write_aux) SB_LAUNCH_T(K, T, true); else SB_LAUNCH_T(K, T, false); } while (0)
#define SB_LAUNCH_TM(K) do { if (e->mat_mode == 2) SB_LAUNCH_TA(K, 2); else if (e->mat_mode == 1) SB_LAUNCH_TA(K, 1); else SB_LAUNCH_TA(K, 0); } while (0)
        if (mode == SB_COLLIDE_GRID) SB_LAUNCH_TM(k_substep_tiled_grid);
        else SB_LAUNCH_TM(k_substep_tiled);
#undef SB_LAUNCH_TM
#undef SB_LAUNCH_TA
#undef SB_LAUNCH_T
    }
    if (mode == SB_COLLIDE_GRID && e->P) {
        e->grid_par ^= 1u;   // (workgroup 0 of that launch publishes the other block)
        e->grid_executed++;  // (an abort takes the count back: sb_api.hip grid_substeps)
    }
    e->cur ^= 1;
    e->substeps_done++;
}

// How close is the closest pair any neighbour list holds?  Asked by the host between launches (sb_api.hip hybrid_substeps),
// never on the substep path.  Squared distance (float bits; +inf = every list is empty, 0 = a list overflowed or holds a NaN),
// per workgroup into `blk`, then one small launch takes the minimum -- NOT one atomic per wave on a shared word: same-address
// traffic from the whole device costs ~30 ns per access here (a flag stored by the list makers themselves cost the pile 60 us
// per substep).
__global__ __launch_bounds__(SB_BLOCK) void k_lists_min_d2(const uint32_t *__restrict__ nl_count, const uint32_t *__restrict__ nl,
                                                           uint32_t stride, const float2 *__restrict__ pos, uint32_t P, float *blk)
{
    __shared__ float s_min[SB_BLOCK / 64];
    float best = INFINITY;
    for (uint32_t i = blockIdx.x * SB_BLOCK + threadIdx.x; i < P; i += gridDim.x * SB_BLOCK) {
        const uint32_t n = nl_count[i];
        if (n == SB_NL_OVERFLOW) best = 0.0f;
        else if (n) {
            const float2 p = pos[i];
            for (uint32_t k = 0; k < n; k++) {
                const float2 q = pos[nl[(size_t)k * stride + i]];
                const float dx = q.x - p.x, dy = q.y - p.y, d2 = dx * dx + dy * dy;
                best = d2 == d2 ? fminf(best, d2) : 0.0f;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) best = fminf(best, __shfl_xor(best, o));
    if ((threadIdx.x & 63u) == 0u) s_min[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0u) {
        for (uint32_t w = 1; w < SB_BLOCK / 64; w++) best = fminf(best, s_min[w]);
        blk[blockIdx.x] = best;
    }
}
__global__ __launch_bounds__(256) void k_min_of(const float *__restrict__ blk, uint32_t n, float *out)
{
    __shared__ float s_min[4];
    float best = INFINITY;
    for (uint32_t i = threadIdx.x; i < n; i += 256u) best = fminf(best, blk[i]);
    for (int o = 32; o > 0; o >>= 1) best = fminf(best, __shfl_xor(best, o));
    if ((threadIdx.x & 63u) == 0u) s_min[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0u) *out = fminf(fminf(s_min[0], s_min[1]), fminf(s_min[2], s_min[3]));
}

void sbk_launch_lists_min_d2(sb_engine *e)
{
    const uint32_t blocks = std::min(std::max(cdiv(e->P, SB_BLOCK * 4u), 1u), 1024u);
    k_lists_min_d2<<<blocks, SB_BLOCK, 0, e->stream>>>(e->grid.nl_count, e->grid.nl, e->grid.nl_stride, e->part[e->cur].pos, e->P,
                                                      (float *)e->d_grid_nonempty + 1);
    k_min_of<<<1, 256, 0, e->stream>>>((const float *)e->d_grid_nonempty + 1, blocks, (float *)e->d_grid_nonempty);
}

void sbk_launch_delete(sb_engine *e)
{
    if (!e->nbeam) return;
    uint32_t nwords = cdiv(e->nbeam, 32);
    k_delete<<<cdiv(nwords, SB_BLOCK), SB_BLOCK, 0, e->stream>>>(e->beams, nwords, e->nbeam, e->d_broken,
                                                                 e->d_dead_gen, ++e->delete_gen, e->path == SB_PATH_TILED);
}

void sbk_launch_halo_clear_ghost_flags(sb_engine *e)
{
    if (!e->n_ghost_b_copies) return;
    k_halo_clear_ghost_flags<<<cdiv(e->n_ghost_b_copies, SB_BLOCK), SB_BLOCK, 0, e->stream>>>(e->d_ghost_b, e->n_ghost_b_copies, e->d_broken);
}

void sbk_launch_halo_pack(sb_engine *e, float *dst)
{
    uint32_t n = e->n_send_p + e->n_send_b;
    if (!n) return;
    k_halo_pack<<<cdiv(n, SB_BLOCK), SB_BLOCK, 0, e->stream>>>(e->part[e->cur], e->beams, e->d_send_p, e->d_send_p_off,
                                                              e->n_send_p, e->d_send_b, e->d_send_b_off, e->n_send_b, dst, e->d_dead_gen);
}

void sbk_launch_halo_unpack(sb_engine *e, const float *src)
{
    uint32_t n = e->n_ghost_p + e->n_ghost_b_copies;
    if (!n) return;
    SbHaloFlags flags{};
    if (e->ntiles) {
        flags.acc_flag = e->d_acc_flag[e->cur];
        flags.tile_p0 = e->bk.K ? e->bk.d_tile_p0 : e->d_tile_p0;
        flags.ntiles = e->ntiles;
        if (e->bk.K) {
            flags.plastic0 = e->bk.d_plastic[0];
            flags.plastic1 = e->bk.d_plastic[1];
            flags.tile_b0 = e->bk.d_tile_b0;
        }
    }
    if (e->d_grid_ctl) e->grid_force = true; // ghosts jumped: the next substep starts with a forced build
    k_halo_unpack<<<cdiv(n, SB_BLOCK), SB_BLOCK, 0, e->stream>>>(e->part[e->cur], e->beams, e->d_ghost_p, e->d_ghost_p_off,
                                                                e->n_ghost_p, e->d_ghost_b, e->d_ghost_b_off,
                                                                e->n_ghost_b_copies, src, e->d_broken, e->d_dead_gen, flags);
}

static inline size_t sb_mailbox_stride(uint32_t recv_floats)
{
    return ((size_t)recv_floats * 4 + 255) & ~(size_t)255;
}

void sbk_launch_peer_exchange(sb_engine *e)
{
    const uint32_t seq = ++e->peer_seq, par = seq & 1u;
    SbPeerRoute route{};
    SbPeerSignal sig{};
    route.n = sig.n = e->n_peers;
    sig.seq = seq;
    for (uint32_t j = 0; j < e->n_peers; j++) {
        char *box = (char *)e->peer_box[j];
        route.begin[j] = e->peer_begin[j];
        route.end[j] = e->peer_begin[j] + e->peer_len[j];
        route.dst[j] = (float *)(box + SB_MAILBOX_FLAGS_BYTES + (size_t)par * e->peer_stride[j]) + e->peer_dst[j];
        sig.remote[j] = (uint32_t *)box + e->peer_slot[j];
    }
    sig.local = (uint32_t *)e->mailbox;
    sig.limit_ticks = (uint64_t)e->peer_timeout_ms * 100000ull;
    sig.err = e->dev_err;
    uint32_t n = 3 * e->n_send_p + e->n_send_b;
    if (n)
        k_halo_pack_peer<<<cdiv(n, SB_BLOCK), SB_BLOCK, 0, e->stream>>>(e->part[e->cur], e->beams, e->d_send_p,
                                                                       e->d_send_p_off, e->n_send_p, e->d_send_b,
                                                                       e->d_send_b_off, e->n_send_b, route, e->d_dead_gen);
    k_peer_signal_wait<<<1, 64, 0, e->stream>>>(sig);
    const float *src = (const float *)((char *)e->mailbox + SB_MAILBOX_FLAGS_BYTES +
                                       (size_t)par * sb_mailbox_stride(e->recv_floats));
    sbk_launch_halo_unpack(e, src);
}

