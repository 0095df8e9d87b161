// sb_engine.h -- internal engine state behind the C ABI of include/softbody.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/softbody.h"
#include "sb_physics.h"
#include "sb_tiling.h"
#include "sb_blocking.h"

// Device SoA of the beam working set.  One entry per beam COPY: the atomic path keeps one
// copy per active slot (slot order); the tiled path keeps per-tile slices in which a beam
// cut by a tile boundary appears once in each of its two tiles (DESIGN.md "cut beams").
struct SbBeamArrays {
    uint32_t *ia, *ib;  // atomic path: internal particle indices of the endpoints
    uint32_t *pair;     // tiled path: local A | local B << lbits | material << 2*lbits; 0xFFFFFFFF = dead
    float *length, *target, *last, *spring, *damp, *yield, *limit, *strain, *stress;
    uint32_t *slot;     // beam mapping slot (as of upload) this copy belongs to
};

struct SbParticleArrays {
    float2 *pos, *vel, *acc;
};

// waves per SIMD the hash-on single-substep kernel is compiled for (k_substep_tiled_grid: 8 = a cap of 64 VGPRs and four 8-wave
// workgroups per CU, which is also the share of the CU's LDS a workgroup may use; 6 = 80 VGPRs, three workgroups)
#ifndef SB_GRID_WAVES
#define SB_GRID_WAVES 8
#endif
#define SB_BK_LBITS 12u  // bits per region-local endpoint index of the blocked kernel (regions of up to 4094 particles)
#ifndef SB_BK_T
#define SB_BK_T 512u     // threads per tile workgroup of k_substep_blocked
#endif
#ifndef SB_BK_MAXB
#define SB_BK_MAXB (6144 / SB_BK_T) // entries per thread    -> a tile evaluates at most 6144 beams per substep
#endif
#ifndef SB_BK_MAXP
#define SB_BK_MAXP (2560 / SB_BK_T) // particles per thread  -> a region holds at most 2560 particles
#endif
// slot classes (sb_blocked.hip): own / halo particle slots and own / halo entry slots per thread
#define SB_BK_OWNP 2u                        // a tile owns at most 1024 particles ...
#define SB_BK_HALOP (SB_BK_MAXP - SB_BK_OWNP) // ... and reads at most 1536 more
#define SB_BK_OWNB 6u                        // a tile owns at most 3072 beams ...
#define SB_BK_HALOB (SB_BK_MAXB - SB_BK_OWNB) // ... and evaluates at most 3072 more
#define SB_BK_KMAX 8u
// When the caller does not say: the plan is made SB_BK_KPLAN substeps deep (if the regions fit), and a call of n substeps is
// cut into launches of at most that depth by sbk_split_call (sb_blocked.hip), which prices the candidates with the measured
// launch times: a long call runs at SB_BK_KLONG substeps per launch (1 M particles: 5 -> 13.8, 6 -> 13.2, 7 -> 13.4 us per
// substep), a short one in the fewest launches (20 substeps: 7 + 7 + 6, not 6 + 6 + 6 + 2).
#define SB_BK_KPLAN 7u
#define SB_BK_KLONG 6u

// the hybrid's running state on the device (sb_api.hip hybrid_step)
struct SbHybridCtl {
    float D, Cx, Cy;     // the hash's displacement bound and accumulated drift, carried through the blocked launches
    float cx, cy, skin;  // the drift every launch measures against; the budget
    uint32_t bad, done;  // sticky: a launch went over the budget (its result is discarded); launches that stayed within it
    uint32_t substeps;   // ... and the substeps they advanced
    uint32_t pad_;       // (r03: any_broken -- the flag rides behind the launch's slots now, sb_blocked.hip SbTrack)
    uint32_t fail_at;    // tests: the launch with this number fails whatever it measured (0xFFFFFFFF: none)
};

// device side of the temporally blocked plan (sb_blocking.h, sb_blocked.hip)
struct SbBlockedDev {
    uint32_t K = 0;           // most substeps a launch may advance (the deepest prefix of the plan that fits the kernel's slots); 0 = the engine is not running blocked
    uint32_t plan_K = 0;      // depth the plan was made for (the stride of ring_cnt / lvl_cnt); a launch of k <= K loads the depth-k prefix only
    uint32_t cap = 0;         // region capacity = local index of the first dummy endpoint
    uint32_t dummy_word = 0;  // entry word of dead and padding entries
    uint32_t cur = 0;         // which beam-state buffer holds the current state
    uint32_t *d_tile_p0 = nullptr, *d_tile_h0 = nullptr, *d_halo_idx = nullptr, *d_ring_cnt = nullptr, *d_tile_b0 = nullptr,
             *d_tile_e0 = nullptr, *d_tile_s0 = nullptr, *d_ent_word = nullptr, *d_ent_state = nullptr, *d_lvl_cnt = nullptr,
             *d_tile_n0 = nullptr, *d_tile_nb = nullptr, *d_slot_e0 = nullptr, *d_slot_ent = nullptr;
    float *d_ent_length = nullptr;
    uint32_t *d_ent_word0 = nullptr; // d_ent_word as uploaded (delete passes turn entries into dummies; an upload that keeps the plan undoes that)
    std::vector<uint32_t> h_beam_slot, h_tile_b0; // blocked beam -> mapping slot, beams per tile (host copies, for the same)
    float *d_target[2] = {nullptr, nullptr}, *d_last[2] = {nullptr, nullptr};
    uint32_t *d_plastic[2] = {nullptr, nullptr}; // per state buffer, per tile: 0 = every owned beam's target is still its rest length (sb_blocked.hip)
    bool pristine = false;    // as of the upload: no tile had a yielded beam (the traffic model leaves the targets out then)
    // what a launch works on besides the plan (the engine's own arrays when the blocked layout is the engine's layout; arrays of
    // its own when it sits beside the tiled layout of an SB_COLLIDE_GRID engine: e->hy)
    float *d_mat = nullptr;   // [nmat][6]
    uint32_t nmat = 0, mat_mode = 0, ntiles = 0, nbeams = 0;
    size_t lds_bytes = 0;
    float *d_strain = nullptr, *d_stress = nullptr;
    uint32_t *d_broken = nullptr;
    // hybrid only
    uint32_t *d_broken_new[2] = {nullptr, nullptr}; // break flags of the tracked launches in flight, by launch number parity (sb_blocked.hip SbTrack)
    uint32_t *d_copy_of_g = nullptr, *d_g_of_copy = nullptr; // blocked beam -> a tiled copy of it; tiled copy -> blocked beam
    SbHybridCtl *d_q = nullptr;
    float *d_hslots = nullptr;        // what a tracked launch leaves for the launch that validates it: 3 sets of 64 slots + a flag entry (float4 each)
    uint32_t qpar = 0, seq = 0, run_launches = 0, k_prev = 0; // which of the two SbHybridCtl blocks is current; launch number; launches of the run so far; depth of the last
    uint32_t synced_delete_gen = 0;   // delete passes of the tiled layout this plan has seen
    float rate = 0.0f;                // growth of the displacement bound per substep, as the last tracked run measured it (0: not known)
    uint32_t fail_streak = 0;         // tracked runs refused in a row (each doubles the stretch of single substeps before the next look)
    uint32_t slow_chunk = 0, slow_left = 0; // substeps to run substep by substep before looking again whether the scene is quiet (doubles while it is not)
    uint64_t launches_ok = 0, launches_failed = 0, substeps_blocked = 0, validate_launches = 0; // (validate_launches: k_hybrid_validate launches -- one per run since r04)
    uint32_t k_long = 0;      // substeps per launch of a long call (what the traffic model prices)
    bool fixed_depth = false; // the caller named the depth (sb_options.block_substeps): every call runs in the fewest launches
    uint64_t entries = 0, halo_entries = 0, halo_particles = 0; // totals of the whole plan (depth K)
    uint64_t entries_at[SB_BK_KMAX + 1] = {}, region_at[SB_BK_KMAX + 1] = {}; // totals over the tiles of what a launch of depth k loads
};

struct sb_engine {
    sb_options opt{};
    SbParams prm{};
    uint32_t subticks = 64;
    uint32_t path = SB_PATH_ATOMIC;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> marks;   // sb_mark: created on first use
    std::string err;

    bool loaded = false;
    uint32_t P = 0;        // active particles at upload
    uint32_t B = 0;        // active beams at upload
    uint32_t nbeam = 0;    // beam copies on the device
    uint32_t cur = 0;      // which particle buffer holds the current state
    uint64_t substeps_done = 0;

    // host shadows (copy semantics: filled from the caller's buffers, never aliased)
    std::vector<uint8_t> h_metadata; // 112 B
    std::vector<uint8_t> h_mapping;  // full capacity, caller's layout
    std::vector<uint32_t> h_pidx;    // internal particle -> data index
    std::vector<uint32_t> h_pslot;   // internal particle -> mapping slot
    std::vector<uint32_t> h_copy_of_slot; // beam slot -> index of the copy read back for it
    SbHostBeams h_beams;      // static part of every active beam record, per slot
    // what an upload of the same topology reuses (sb_api.hip rewrite_scene_state): copy / blocked beam -> mapping slot, the
    // blocked plan's beams per tile, a pristine copy of the one device array delete passes write into
    std::vector<uint32_t> h_slot_of_copy;
    uint32_t *d_live0 = nullptr;     // entry words (blocked) / endpoint words (tiled) / first endpoints (atomic) as uploaded
    size_t live_words = 0;
    uint32_t uploads_kept = 0;       // uploads that kept the plan (sb_info "uploads_kept")
    // An upload that only REMOVED beams from the scene on the device keeps the plan too (sb_api.hip rewrite_scene_state): the
    // engine's own slots stay those of the upload the plan was made for, the removed ones die on the device like beams a delete
    // pass removed, and the caller's slots (0 .. beam count of the latest upload) map onto the engine's: h_user_slot[u], strictly
    // increasing; empty = identity.  Every path that speaks the caller's slots goes through sb_user_slot() / sb_user_beams().
    std::vector<uint32_t> h_user_slot;
    uint32_t uploads_edited = 0;     // ... of them, uploads that removed beams (sb_info "uploads_edited")

    // device state
    SbParticleArrays part[2]{};
    SbBeamArrays beams{};
    SbConsts consts{};             // metadata bytes 48..111, passed to every launch BY VALUE (kernarg):
                                   // scalar-cache lines are not reliably refreshed between launches on this stack
    uint32_t *d_pidx = nullptr;    // data index per internal particle (collision tie-break :153)
    uint32_t *d_pslot = nullptr;
    uint32_t *d_islot = nullptr;   // internal particle per slot (SB_COLLIDE_GRID on the tiled path: sb_lists_cooperative)
    uint32_t lds_coop_off = 0, lds_coop = 0; // ... and that function's area behind the substep kernel's own dynamic LDS (bytes; 0: none)
    int2 *d_forces = nullptr;      // atomic path accumulator (compute.wgsl:68-69)
    uint32_t *d_broken = nullptr;  // bit per beam copy: flagged this frame (compute.wgsl:86-88)
    uint32_t *d_dead_gen = nullptr;  // per beam slot: number of the delete pass that removed it (0 = live)
    uint32_t delete_gen = 0;
    // halo exchange lists (internal particle indices; beam copy indices)
    uint32_t *d_ghost_p = nullptr, *d_send_p = nullptr;
    uint32_t n_ghost_p = 0, n_send_p = 0;
    uint32_t *d_send_b = nullptr;      // one copy per sent beam
    uint2 *d_ghost_b = nullptr;        // (copy index, position in the ghost beam list), every copy
    uint32_t n_ghost_b = 0, n_send_b = 0, n_ghost_b_copies = 0;
    uint32_t *d_send_p_off = nullptr, *d_send_b_off = nullptr;   // float offsets into the packed send buffer
    uint32_t *d_ghost_p_off = nullptr, *d_ghost_b_off = nullptr; // float offsets into the packed recv buffer
    uint32_t send_floats = 0, recv_floats = 0;                   // extents of the two packed layouts
    // direct peer exchange (sb_peer_*): own mailbox, neighbours' mailboxes, per-neighbour routing
    void *mailbox = nullptr;
    std::vector<void *> mapped;        // opened IPC mappings (closed with the scene)
    uint32_t n_peers = 0, peer_seq = 0, peer_timeout_ms = 0;
    void *peer_box[8] = {};
    uint32_t peer_stride[8] = {}, peer_begin[8] = {}, peer_len[8] = {}, peer_dst[8] = {}, peer_slot[8] = {};

    // tiled path
    uint32_t ntiles = 0, tile_cap_own = 0, tile_cap_all = 0;
    uint32_t nhalo = 0;
    uint32_t *d_tile_p0 = nullptr;    // [ntiles+1] first internal particle of each tile
    uint32_t *d_tile_b0 = nullptr;    // [ntiles+1] first beam copy of each tile
    uint32_t *d_tile_h0 = nullptr;    // [ntiles+1] first halo entry of each tile
    uint32_t *d_halo_idx = nullptr;   // internal particle index of each halo entry
    size_t lds_bytes = 0;
    uint32_t *d_acc_flag[2] = {nullptr, nullptr}; // per particle buffer, per tile: 0 = every acc is zero
    SbBlockedDev bk;
    SbBlockedDev hy;                  // SB_COLLIDE_GRID: a blocked plan BESIDE the tiled layout, for the stretches in which nothing is within reach (hy.K != 0: available)
    struct SbHybridPending *hy_pending = nullptr; // ... being made on a side thread since the upload; it goes to the device the first
                                      // time the scene is found quiet (sb_api.hip hybrid_substeps), so a pile never pays for it
    std::vector<uint32_t> h_tile_p0;  // the tiled layout's tiles (host copy: the two plans must agree on them)
    // beam word packing and material dictionary (tiled path)
    uint32_t lbits = 16;      // bits per tile-local endpoint index
    uint32_t mat_mode = 0;    // 0: per-copy parameter arrays; 1: table of (spring,damp,yield,limit) + per-copy length;
                              // 2: table of (length,spring,damp,yield,limit)
    uint32_t nmat = 0;
    float *d_mat = nullptr;   // [nmat][6] = length, spring, damp, yield_strain, strain_break_limit, 1/length

    // spatial hash (SB_COLLIDE_GRID): two hash buffers, the decision state, what the particle kernels' tails need (sb_physics.h SbGridCtl)
    SbGrid grid{};
    uint32_t ncell = 0;               // nx*ny (+1 spare entry in the per-cell arrays)
    unsigned long long *d_head[2] = {}; // per cell: (build number << 32) | first record of its list (SbGrid::head)
    uint32_t *d_cell_of[2] = {};      // per particle: cell at that buffer's last build
    float4 *d_rec[2] = {};            // per particle: {x, y at the build, slot, next record of its cell's list}
    SbGridCtl *d_grid_ctl = nullptr;  // [2] decision state by substep parity (device resident: no host sync per substep)
    uint32_t *d_blk_max = nullptr;    // float4[3][SB_GRID_SLOTS]: what the particle kernels measure (sb_physics.h SbGridStep), by substep number % 3
    uint32_t *d_nl_count = nullptr, *d_nl = nullptr; // neighbour lists (SbGrid)
    uint32_t *d_grid_nonempty = nullptr; // sbk_launch_lists_min_d2: [0] = squared distance of the closest listed pair (+inf: every list empty), [1 ..] per-workgroup minima
    uint32_t *d_grid_outside = nullptr; // [4] particles each hash build found outside its frame, by build number & 3
    uint32_t grid_par = 0;            // the SbGridCtl block the next launch reads
    SbGridCtl grid_ctl0[2]{};         // the decision state a fresh upload starts from
    size_t grid_heads = 0, grid_slots = 0; // entries of each d_head, of d_blk_max
    bool grid_force = false;          // the next substep starts with a forced helper launch (upload, ghost refresh, hybrid, abort recovery)
    uint32_t grid_classic_left = 0, grid_classic_chunk = 0; // substeps left in the classic stretch after an abort; its length (doubles per abort, decays when calm)
    uint32_t grid_calm = 0;           // lagged substeps since the last abort
    uint64_t grid_executed = 0;       // host mirror of SbGridCtl::executed (which wraps at 2^32; this one picks the slot set, k % 3, and must not): single substeps run since the upload
    uint64_t grid_aborts = 0, grid_helper_launches = 0, grid_classic_substeps = 0; // statistics (sb_get_info)
    uint32_t *dev_err = nullptr;      // pinned host word: bounded device-side waits report here (sb_sync reads it)
    // pinned staging of uploads and read-backs (sb_api.hip: stage_*): two chunks, filled / drained by several host threads while
    // the other one is on the wire (hipMemcpy from pageable memory stages through ONE thread: ~5 GB/s)
    uint8_t *stage[2] = {nullptr, nullptr};
    hipEvent_t stage_done[2] = {nullptr, nullptr};
    bool stage_busy[2] = {false, false};
    int stage_cur = 0;
    std::thread reaper;               // frees the host arrays of the last upload (sb_api.hip: SbUploadTrash)

    size_t device_bytes = 0;
    std::vector<void *> allocs;                             // freed with the scene (not pooled)
    std::vector<std::pair<void *, size_t>> pool_used, pool_free; // device blocks of the scene / kept for the next upload (sb_api.hip dev_alloc)
};

// sb_kernels.hip
void sbk_launch_substep(sb_engine *e, bool write_aux);
uint32_t sbk_grid_mode(const sb_engine *e); // SB_GRID_LAGGED / SB_GRID_CLASSIC: the schedule the next substep's launch will run
void sbk_launch_grid_settle(sb_engine *e);   // the next substep's decision ahead of time, into e->d_grid_ctl[e->grid_par] (the host's look)
void sbk_launch_delete(sb_engine *e);
void sbk_launch_lists_min_d2(sb_engine *e); // e->d_grid_nonempty[0] = squared distance of the closest pair any neighbour list holds (float; hybrid look)
void sbk_launch_halo_clear_ghost_flags(sb_engine *e);
void sbk_launch_halo_pack(sb_engine *e, float *dst);
void sbk_launch_halo_unpack(sb_engine *e, const float *src);
void sbk_launch_peer_exchange(sb_engine *e);
// sb_blocked.hip
void sbk_launch_blocked(sb_engine *e, uint32_t n, bool write_aux);
void sbk_preload_blocked(const SbBlockedDev &bk, bool tracked); // resolve every kernel variant a call on this plan may launch (upload time)
uint32_t sbk_split_call(uint32_t n, uint32_t kmax, bool fewest, uint32_t *first, uint32_t *n_first); // launches: n_first of depth first, the rest first - 1
void sbk_launch_delete_blocked(sb_engine *e);
void sbk_hybrid_to_blocked(sb_engine *e);
void sbk_hybrid_to_tiled(sb_engine *e, bool aux);
void sbk_hybrid_launch(sb_engine *e, const uint32_t *ks, uint32_t count, bool aux_last);
