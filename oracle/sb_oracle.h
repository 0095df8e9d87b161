/*
 * sb_oracle.h -- CPU ORACLE for the softbody physics step.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's WGSL physics
 * (/root/reference/src/shaders/compute.wgsl:90-246) operating directly on the
 * reference's seven storage buffers (src/engineWorker.ts:136-176) in the byte
 * layouts src/engineMapping.ts defines.  It exists so tests can check the HIP
 * product path and so bench.py can time a CPU baseline.
 *
 *   Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 *   include, link, load or execute anything under oracle/.  The product
 *   (softbody-webgpu_amd/) never does.
 *
 * PARITY STATUS: "parity unpinned" by the reference.  The reference ships no
 * tests, fixtures, golden vectors or recorded outputs
 * and its only implementation is WGSL run through a browser's WebGPU, which
 * does not exist in this environment.  The oracle is therefore pinned to the
 * WGSL text by analytic known-answer tests (tests/test_oracle_kat.py), each
 * citing the WGSL line it checks, not by outputs of the reference.
 *
 * Semantics implemented: "S0" of SURVEY.md section 8(a) A3 -- per substep all
 * beams first (complete integer force sums), then all particles; the limit
 * the racy reference approaches when beam threads win every race.
 */
#ifndef SB_ORACLE_H
#define SB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* buffer layouts */
#define SBO_LAYOUT_V1 1 /* reference: u16 mapping, packed u16 pair, beam stride 40 (engineMapping.ts:151,183-193) */
#define SBO_LAYOUT_V2 2 /* wide: u32 mapping, two u32 endpoints, beam stride 44 */

/* collision broad phase */
#define SBO_COLLIDE_OFF 0      /* skip compute.wgsl:144-170 entirely */
#define SBO_COLLIDE_ALLPAIRS 1 /* compute.wgsl:144-170 as written (O(P^2)) */
#define SBO_COLLIDE_GRID 2     /* uniform grid, contacts applied in ascending slot order => bit-identical to ALLPAIRS */

#define SBO_METADATA_BYTES 112 /* engineMapping.ts:239 */
#define SBO_PARTICLE_STRIDE 24 /* engineMapping.ts:103 */
#define SBO_BEAM_STRIDE_V1 40  /* engineMapping.ts:151 */
#define SBO_BEAM_STRIDE_V2 44

typedef struct sbo_params {
    float bounds_size;     /* compute.wgsl:1, engineWorker.ts:39,329 */
    float particle_radius; /* compute.wgsl:2, engineWorker.ts:40,330 */
    float time_step;       /* compute.wgsl:3, engineWorker.ts:331 (= 1/subticks) */
    int32_t layout;        /* SBO_LAYOUT_* */
    int32_t collision_mode;/* SBO_COLLIDE_* */
    int32_t threads;       /* OpenMP threads (<=1: scalar) */
} sbo_params;

/* One compute_update dispatch (compute.wgsl:90-203) under S0 semantics.
 * metadata is read only here; beams/particle_forces/delete_mappings are
 * updated in place; particles_write receives the new particle state. */
void sbo_update(const sbo_params *prm, const uint8_t *metadata, const uint8_t *particles_read,
                uint8_t *particles_write, uint8_t *beams, const uint8_t *mapping,
                int32_t *particle_forces, uint32_t *delete_mappings);

/* compute_delete (compute.wgsl:205-246), canonical INTENT semantics: in-place
 * stable compaction of flagged beam slots, beam_i_c decremented, bitmask
 * cleared.  delete_words = number of u32 words in delete_mappings. */
void sbo_delete(const sbo_params *prm, uint8_t *metadata, uint8_t *mapping,
                uint32_t *delete_mappings, size_t delete_words);

/* n_substeps dispatches alternating A->B, B->A starting with A as the read
 * buffer (engineWorker.ts:655-661).  Returns 0 if the final state is in A,
 * 1 if it is in B (odd n). */
int sbo_step(const sbo_params *prm, const uint8_t *metadata, uint8_t *particles_a,
             uint8_t *particles_b, uint8_t *beams, const uint8_t *mapping,
             int32_t *particle_forces, uint32_t *delete_mappings, uint32_t n_substeps);

/* One frame (engineWorker.ts:646-665): subticks x update then one delete. */
void sbo_frame(const sbo_params *prm, uint8_t *metadata, uint8_t *particles_a, uint8_t *particles_b,
               uint8_t *beams, uint8_t *mapping, int32_t *particle_forces,
               uint32_t *delete_mappings, size_t delete_words, uint32_t subticks);

/* canonical arithmetic helpers, exported for the KATs */
float sbo_pow(float x, float y);
int32_t sbo_f32_to_i32(float x);

#ifdef __cplusplus
}
#endif
#endif
