/*
 * softbody.h -- C ABI of the MI355X-native softbody physics step.
 *
 * Drop-in boundary for ONE path of spsquared/softbody-webgpu: the per-substep
 * physics step (`compute_update` / `compute_delete`, src/shaders/compute.wgsl:90-246)
 * as driven by src/engineWorker.ts.  The reference has no FFI: its boundary is the
 * set of WebGPU calls engineWorker.ts makes on seven storage buffers, in the byte
 * layouts src/engineMapping.ts defines.  Each entry point below replaces one of
 * those call groups (cited per function) and moves raw bytes in those layouts.
 *
 * Conventions
 *   - plain C types only; every function returns an sb_status (0 = ok);
 *     sb_last_error() gives the message for the last failure on that engine
 *     (engine == NULL: the last sb_create failure of the calling thread).
 *   - COPY semantics: no host pointer is retained after a call returns (the JS
 *     ArrayBuffers are owned and later mutated by BufferMapper,
 *     engineMapping.ts:364-367).
 *   - one call at a time per engine, like the reference's AsyncLock
 *     (src/lock.ts:4-19; engineWorker.ts:553,584,632).  When do calls return?
 *       SB_COLLIDE_OFF / SB_COLLIDE_ALLPAIRS, and SB_PATH_ATOMIC with any collision mode:
 *         sb_step / sb_frame / sb_delete_pass only ENQUEUE work on the engine's HIP stream;
 *         sb_sync, sb_load_buffers and sb_step_timed wait for it.
 *       SB_COLLIDE_GRID on the tiled path (the default of sb_default_options): sb_step and
 *         sb_frame MAY WAIT for the stream, like every call of the reference does
 *         (engineWorker.ts:632-633,686-688: `await queue.onSubmittedWorkDone()` on both
 *         sides of a frame).  The spatial hash keeps itself valid on the device; what
 *         the host owes it is one look when a call's substeps have been issued (did a
 *         substep move somebody farther than predicted?  -- then the launches behind it
 *         returned at once and are issued again behind a fresh hash), and the stretches
 *         that run several substeps per launch (nothing within reach of anything) are
 *         sized from a look at the device as well.  So a call returns when its LAST
 *         substep has been issued and everything before the last look has run: for a
 *         1 M-particle scene a 64-substep frame holds the calling thread for about
 *         2.4 ms of its 2.4 ms (a Node host should call from a worker thread, as the
 *         reference itself does: engine.ts:138).  sb_delete_pass, sb_write_user_input
 *         and the sb_halo_* / sb_peer_* calls still only enqueue.
 *       A wait POLLS the stream for as long as the work in flight should take (busily for
 *         the first 8 ms, then every ~50 us between short sleeps; 0.2 s at most) before it
 *         parks the thread: being woken costs 0.2 - 0.5 ms on some hosts, more than many
 *         of these waits.  SB_WAIT_SPIN_US=0 in the environment parks always.
 *   - there is no CPU fallback: without a usable HIP device sb_create fails.
 */
#ifndef SOFTBODY_H
#define SOFTBODY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SB_ABI_VERSION 1

typedef struct sb_engine sb_engine;

typedef enum sb_status {
    SB_OK = 0,
    SB_ERR_INVALID = 1,     /* bad argument / buffer too small / inconsistent scene */
    SB_ERR_HIP = 2,         /* a HIP runtime call failed (message carries hipGetErrorString) */
    SB_ERR_NO_DEVICE = 3,   /* no usable gfx950 device (engineWorker.ts:86,93,98 throw TypeError) */
    SB_ERR_OOM = 4,
    SB_ERR_STATE = 5,       /* call order (e.g. step before write_buffers) */
    SB_ERR_UNSUPPORTED = 6
} sb_status;

/* host buffer layouts (src/engineMapping.ts) */
#define SB_LAYOUT_V1 1 /* reference: u16 mapping, beam = packed u16 pair + 9 f32, stride 40 */
#define SB_LAYOUT_V2 2 /* wide: u32 mapping, beam = u32 a, u32 b + 9 f32, stride 44 */

/* particle-particle collision broad phase (compute.wgsl:142-170) */
#define SB_COLLIDE_OFF 0      /* skip the collision loop (BASELINE config 2) */
#define SB_COLLIDE_ALLPAIRS 1 /* the reference's O(P^2) scan, LDS-tiled */
#define SB_COLLIDE_GRID 2     /* spatial hash + neighbour lists; same pair set and summation order => same bits
                               * as SB_COLLIDE_ALLPAIRS.  The default (sb_default_options). */

/* device schedule of one substep */
#define SB_PATH_AUTO 0
#define SB_PATH_ATOMIC 1 /* beam kernel with global i32 atomics + particle kernel */
#define SB_PATH_TILED 2  /* fused LDS-tiled substep: forces never leave the CU */

#define SB_METADATA_BYTES 112
#define SB_PARTICLE_STRIDE 24
#define SB_BEAM_STRIDE_V1 40
#define SB_BEAM_STRIDE_V2 44
#define SB_USER_INPUT_BYTES 32
#define SB_USER_INPUT_OFFSET 80

typedef struct sb_options {
    uint32_t struct_size;    /* = sizeof(sb_options) */
    float bounds_size;       /* engineWorker.ts:39 (fixed 1000 there; an option here) */
    float particle_radius;   /* engineWorker.ts:40,89 */
    uint32_t subticks;       /* engineWorker.ts:41,90: rounded UP to even; time_step = 1/subticks (:331) */
    uint32_t max_particles;  /* capacity; BufferMapper.maxParticles (engineMapping.ts:362) */
    uint32_t max_beams;      /* capacity; BufferMapper.maxBeams (engineMapping.ts:363) */
    uint32_t layout;         /* SB_LAYOUT_* */
    uint32_t collision_mode; /* SB_COLLIDE_* */
    uint32_t path;           /* SB_PATH_* */
    uint32_t tile_particles; /* SB_PATH_TILED: target particles per tile (0 = the engine picks: fewest rounds of resident tiles).
                              * An UPPER BOUND where several substeps run per launch (collisions off, or nothing within reach): a tile of
                              * that kernel owns at most 1024 particles and 3072 beams, larger targets are lowered to fit */
    int32_t device_ordinal;  /* HIP device */
    float grid_skin;         /* SB_COLLIDE_GRID: cells are 2r + 2*skin wide and the hash is rebuilt only when
                              * some particle may have moved more than `skin` (relative to the scene's common
                              * drift) since the last build.  0 = default: adaptive, starting at 0.4 r; a hash that
                              * is worn out within 3 substeps is followed by one with the smallest doubled skin (up
                              * to 1.6 r) that promises two substeps, one that lasted 64 by one half as wide; > 0 =
                              * that skin, fixed; negative = rebuild every substep */
    uint32_t block_substeps; /* SB_PATH_TILED: substeps one launch advances out of LDS and registers (temporal
                              * blocking over beam-hop rings; same bits as single substeps).  With SB_COLLIDE_OFF
                              * always; with SB_COLLIDE_GRID for the stretches of a run in which every neighbour
                              * list of the spatial hash is empty (the collision loop is then a no-op; the engine
                              * tracks what the particles move and redoes, substep by substep, a launch that used
                              * up the hash's skin).
                              * 0 = default: a plan 7 substeps deep, each call cut into the cheapest balanced
                              * launches (long calls 6 per launch, 20 substeps as 7 + 7 + 6); N > 1 = every call in
                              * the fewest launches of at most N (at most 8); 1 = one launch per substep.  Lowered
                              * automatically to the deepest plan whose regions fit the kernel's registers and LDS */
    uint32_t reserved[3];
} sb_options;

/* Fill with the reference defaults: bounds 1000, radius 10, subticks 64, 65536/65536, v1,
 * spatial-hash collisions (the bits of the reference's all-pairs scan), auto path, device 0. */
void sb_default_options(sb_options *opts);

/* Replaces the WGPUSoftbodyEngineWorker constructor's device/buffer/pipeline creation
 * (engineWorker.ts:83-176, 312-343). */
sb_status sb_create(const sb_options *opts, sb_engine **out);

/* engineWorker.ts:711-717 destroy(). */
sb_status sb_destroy(sb_engine *e);

/* Replaces writeBuffers() (engineWorker.ts:580-597): uploads metadata, mapping, particle data
 * (-> buffer A) and beam data; zeroes the force accumulators, the delete mask and buffer B.
 * Buffers are the BufferMapper ArrayBuffers (engineMapping.ts:342-345) at FULL capacity:
 * metadata 112 B, mapping (max_particles+max_beams) entries, particles max_particles*24 B,
 * beams max_beams*stride B; the *_bytes arguments are checked against that.
 * An upload with the topology of the scene already on the device (same counts and mapping, every beam between the same
 * two particles with the same rest length and material; no ghost zones configured) keeps the engine's plan and only moves
 * state -- about a tenth of the time of an upload that plans (sb_get_info "uploads_kept" counts them); the results are
 * those of a fresh engine either way.  So does an upload that only REMOVED beams from that scene (at most an eighth of them; the
 * beams that are left in their old order under any valid mapping, which is what BufferMapper.writeState produces after
 * removeBeam calls or after a run whose delete passes removed beams, engineMapping.ts:452-459,500-518): the removed beams die on
 * the device like beams a delete pass removed, counts / records / mapping read back exactly as from a fresh engine
 * ("uploads_edited" counts those).  An upload that ADDS a beam plans again. */
sb_status sb_write_buffers(sb_engine *e, const void *metadata, size_t metadata_bytes,
                           const void *mapping, size_t mapping_bytes,
                           const void *particles, size_t particles_bytes,
                           const void *beams, size_t beams_bytes);

/* Replaces Metadata.writeUserInput (engineMapping.ts:323-325, engineWorker.ts:636-642):
 * the 32 bytes at metadata offset 80 (user_strength, mouse_active, mouse_pos, mouse_vel,
 * applied_force). */
sb_status sb_write_user_input(sb_engine *e, const void *bytes32);

/* Replaces the PHYSICS_CONSTANTS round trip (engineWorker.ts:497-507): the 8 floats at
 * metadata offset 48 (gravity.xy, border_elasticity, border_friction, elasticity, friction,
 * drag_coeff, drag_exp), without re-uploading the scene. */
sb_status sb_set_physics_constants(sb_engine *e, const float constants8[8]);
sb_status sb_get_physics_constants(sb_engine *e, float constants8[8]);

/* Replaces the compute pass of frame() (engineWorker.ts:646-665): `subticks` x compute_update
 * with alternating read/write particle buffers, then one compute_delete. */
sb_status sb_frame(sb_engine *e);

/* n x compute_update only (benchmark granularity: one "step" = one substep).  Any n; the
 * engine tracks which particle buffer is current. */
sb_status sb_step(sb_engine *e, uint32_t n_substeps);

/* one compute_delete (compute.wgsl:205-246, canonical semantics: stable compaction). */
sb_status sb_delete_pass(sb_engine *e);

/* device.queue.onSubmittedWorkDone() (engineWorker.ts:633,687). */
sb_status sb_sync(sb_engine *e);

/* sb_step bracketed by HIP events on the engine's stream; *ms = device time of the n substeps. */
sb_status sb_step_timed(sb_engine *e, uint32_t n_substeps, float *ms);

/* Time marks that do not stop the host: sb_mark records HIP event number `slot` (0 .. SB_MAX_MARKS-1) at the current end of
 * the engine's stream and returns at once; sb_mark_elapsed waits for mark b and gives the device time from mark a to mark
 * b.  For loops that interleave sb_step with other stream work (the ghost refresh of the multi-GPU path: the bench times
 * the substep launches and the exchanges of every rank this way without a host synchronisation per exchange). */
#define SB_MAX_MARKS 4096
sb_status sb_mark(sb_engine *e, uint32_t slot);
sb_status sb_mark_elapsed(sb_engine *e, uint32_t a, uint32_t b, float *ms);

/* Replaces loadBuffers() (engineWorker.ts:548-579): metadata, mapping, particles (current
 * buffer) and beams back into host ArrayBuffers of full capacity.  Only records reachable
 * through the mapping are written; other bytes of the caller's buffers are left as they are.
 * Any pointer may be NULL to skip that buffer. */
sb_status sb_load_buffers(sb_engine *e, void *metadata, size_t metadata_bytes,
                          void *mapping, size_t mapping_bytes,
                          void *particles, size_t particles_bytes,
                          void *beams, size_t beams_bytes);

/* counts as the device sees them (metadata.particle_i_c / beam_i_c after deletes). */
sb_status sb_get_counts(sb_engine *e, uint32_t *particles, uint32_t *beams);

/* introspection for benches/tests: key = "path", "tiles", "beam_copies", "halo_particles",
 * "device_bytes", "substeps_done", "kernels_per_substep", "substep_hbm_bytes" (the HBM bytes one substep
 * launch has to move with the data layout the engine holds: the launched kernel's own compulsory traffic),
 * "grid_cells", "grid_builds", "grid_wide", "grid_skin_x1000", "material_mode", "materials", "local_index_bits". */
sb_status sb_get_info(sb_engine *e, const char *key, uint64_t *value);

/* ---- multi-GPU halo exchange; one engine per rank/GPU, each holding its
 * slab of the scene PLUS a ghost zone k beam-hops deep copied from its neighbours.  Ghost
 * particles and ghost beams are stepped like any others (redundantly; the arithmetic is
 * deterministic, so the copies agree bit for bit while their inputs are valid); every k substeps
 * the owners overwrite them: p,v,a (6 floats) per ghost particle, target_length,last_length
 * (2 floats) per ghost beam.  Lists are DATA indices into this engine's particle / beam buffers,
 * in the order the peer packs them.  The exchange itself (RCCL send/recv on the packed device
 * buffers) is the caller's; see softbody-webgpu_amd/halo.py. */
sb_status sb_halo_configure(sb_engine *e, const uint32_t *ghost_particles, uint32_t n_ghost_particles,
                            const uint32_t *send_particles, uint32_t n_send_particles,
                            const uint32_t *ghost_beams, uint32_t n_ghost_beams,
                            const uint32_t *send_beams, uint32_t n_send_beams);
/* Optional: where each list entry lives inside the packed buffers, as FLOAT offsets (6 floats per
 * particle entry, 2 per beam entry).  Default (or NULL): all particles back to back, then all beams.
 * A caller with several neighbours uses this to make each neighbour's share one contiguous segment
 * (one send and one receive per neighbour).  Call after sb_halo_configure. */
sb_status sb_halo_set_layout(sb_engine *e, const uint32_t *send_particle_off, const uint32_t *send_beam_off,
                             const uint32_t *ghost_particle_off, const uint32_t *ghost_beam_off);
/* current state of the send lists -> DEVICE buffer of (6*n_send_particles + 2*n_send_beams) floats
 * (particles first), enqueued on the engine's stream. */
sb_status sb_halo_pack(sb_engine *e, void *device_dst);
/* DEVICE buffer of (6*n_ghost_particles + 2*n_ghost_beams) floats -> current state of the ghost
 * lists, enqueued on the engine's stream. */
sb_status sb_halo_unpack(sb_engine *e, const void *device_src);
/* Beams that BREAK in a multi-GPU run.  A flagged beam keeps acting until the delete pass at the end of its frame
 * (compute.wgsl:205-246, engineWorker.ts:663-664), so only that pass has to agree between ranks, and a beam must
 * disappear from every rank that holds a copy in the same pass.  The owner decides: on an engine with a halo
 * sb_delete_pass drops the flags the GHOST copies raised themselves (their inputs may have been invalid) and removes
 * the rest; from then on the owner's record of a removed beam travels as "dead" in every pack (a NaN payload in
 * last_length); unpack flags the local copies of such beams, and sb_halo_delete_ghosts removes them.  A frame on
 * every rank is therefore:  substeps with a refresh every `depth` of them  ->  sb_delete_pass  ->  one more refresh
 * (pack / exchange / unpack, or sb_peer_exchange)  ->  sb_halo_delete_ghosts.   halo.py Exchanger.frame() and
 * host/halo.js PeerExchanger.frame() do exactly that. */
sb_status sb_halo_delete_ghosts(sb_engine *e);
/* ---- direct neighbour exchange over peer mappings (xGMI stores into the neighbour's mailbox) ----
 * The alternative to moving the packed buffers with a collective library: every exchange is three
 * launches on the engine's own stream (pack straight into the neighbours' mailboxes, signal + wait,
 * unpack), no host synchronisation.  A mailbox is fine-grained device memory: 64 u32 sequence flags
 * (256 B), then two receive buffers (alternating by exchange parity) of the packed receive layout,
 * each rounded up to 256 B.  Order of calls: sb_halo_configure [+ sb_halo_set_layout] ->
 * sb_peer_mailbox on every rank -> trade the 64-byte handles -> sb_peer_map for each neighbour in
 * another process (a neighbour engine in the same process passes its local pointer directly) ->
 * sb_peer_connect -> sb_peer_exchange whenever the ghost zone must be refreshed.  Every rank must
 * call sb_peer_exchange the same number of times.  A neighbour that does not show up within
 * `timeout_ms` makes the wait give up (no hung wave): the next sb_sync reports SB_ERR_HIP. */
#define SB_MAX_PEERS 8
sb_status sb_peer_mailbox(sb_engine *e, void **local_mailbox, void *ipc_handle_64_bytes, uint64_t *mailbox_bytes);
sb_status sb_peer_map(sb_engine *e, const void *ipc_handle_64_bytes, void **mapped_mailbox);
/* per neighbour j: its mailbox, the float count of ITS packed receive layout, my send segment
 * [send_begin, send_begin+send_len) (floats, in my packed send layout), where that segment starts in
 * its receive layout (floats), and which flag slot of its mailbox is mine (= my position in its
 * neighbour order).  My own flag slot for neighbour j is j. */
sb_status sb_peer_connect(sb_engine *e, uint32_t n_peers, void *const *mailboxes, const uint32_t *peer_recv_floats,
                          const uint32_t *send_begin, const uint32_t *send_len, const uint32_t *dst_begin,
                          const uint32_t *their_slot, uint32_t timeout_ms);
sb_status sb_peer_exchange(sb_engine *e);

/* ---- generic x-slab partition of ANY scene into per-rank scenes with ghost zones (host only, no GPU needed) ----
 * The reference has no counterpart (one browser tab, one GPU); this is what lets a scene built by BufferMapper
 * (engineMapping.ts:432-527) or loaded from a snapshot (:407-430) run on several GPUs.  Input: the four host
 * buffers of sb_write_buffers.  owner(particle) = equal-population slab of its x coordinate; ghosts of a rank =
 * everything within `depth` beam hops of its own particles and (contact_reach > 0) of every particle whose x lies
 * within contact_reach of the x range of its own particles (so that contacts across slab faces are computed on
 * both sides: choose at least depth * max(2r + the distance a particle moves in one substep, the longest beam)).  Every rank's scene keeps
 * the relative order of slots and of data indices, so the collision loop's slot order and its index tie-break
 * (compute.wgsl:144,153) are those of the whole scene.  owner(beam) = owner of its endpoint A.  Exchange every
 * `depth` substeps (sb_halo_* / sb_peer_*): both sides list the traded records in ascending GLOBAL data index.
 * Errors are reported through sb_last_error(NULL).  Limits: ghost zones are redundant computation, valid while
 * information travels at most one hop per substep -- contacts between particles of slabs that are not neighbours
 * in x are missed, and break flags do not cross ranks. */
typedef struct sb_partition sb_partition;
sb_status sb_partition_create(uint32_t layout, uint32_t max_particles, uint32_t max_beams, const void *metadata, const void *mapping,
                              const void *particles, const void *beams, uint32_t world, uint32_t depth, float contact_reach,
                              sb_partition **out);
sb_status sb_partition_destroy(sb_partition *p);
/* counts = { local particles, local beams, owned particles, owned beams, peers, depth, global particles, global beams } */
sb_status sb_partition_rank_counts(const sb_partition *p, uint32_t rank, uint32_t counts[8]);
/* the layout the partition was created with (SB_LAYOUT_V1 / SB_LAYOUT_V2): it fixes the record sizes of what
 * sb_partition_rank_scene writes -- mapping indices of 2 or 4 bytes, beam records of SB_BEAM_STRIDE_V1 or _V2 bytes */
sb_status sb_partition_layout(const sb_partition *p, uint32_t *layout);
/* the rank's scene in the partition's layout, into caller buffers of the given capacities (>= the local counts) */
sb_status sb_partition_rank_scene(const sb_partition *p, uint32_t rank, uint32_t max_particles, uint32_t max_beams, void *metadata,
                                  void *mapping, void *particles, void *beams);
/* per LOCAL data index: the global data index and whether this rank owns it (any pointer may be NULL) */
sb_status sb_partition_rank_ids(const sb_partition *p, uint32_t rank, uint32_t *particle_global, uint8_t *particle_owned,
                                uint32_t *beam_global, uint8_t *beam_owned);
/* peer j (ascending rank): its rank and the lengths { ghost particles, sent particles, ghost beams, sent beams } ... */
sb_status sb_partition_peer_counts(const sb_partition *p, uint32_t rank, uint32_t j, uint32_t *peer_rank, uint32_t counts[4]);
/* ... and the lists themselves, LOCAL data indices in the order sb_halo_configure expects (any pointer may be NULL) */
sb_status sb_partition_peer_lists(const sb_partition *p, uint32_t rank, uint32_t j, uint32_t *ghost_particles, uint32_t *send_particles,
                                  uint32_t *ghost_beams, uint32_t *send_beams);

/* the engine's hipStream_t, so a caller can order its own work (RCCL send/recv) after it. */
sb_status sb_get_stream(sb_engine *e, void **hip_stream);

const char *sb_last_error(const sb_engine *e);
uint32_t sb_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
