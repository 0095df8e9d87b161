/*
 * c_abi_frame.c -- the C ABI of include/softbody.h used from plain C (C99), no Python, no Node.
 *
 * Builds a 2-particle / 1-beam scene in the reference's v1 byte layout by hand
 * (src/engineMapping.ts:118-124, 178-194, 252-273), runs one frame on the GPU and prints the
 * particle state.  Build and run (tests/test_c_example.py does exactly this):
 *   gcc -std=c99 -Iinclude examples/c_abi_frame.c -o c_abi_frame \
 *       -Lsoftbody-webgpu_amd/csrc -lsoftbody_hip -Wl,-rpath,$PWD/softbody-webgpu_amd/csrc
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "softbody.h"

#define MAXP 8
#define MAXB 8

int main(void)
{
    sb_options o;
    sb_default_options(&o);
    o.max_particles = MAXP;
    o.max_beams = MAXB;
    o.collision_mode = SB_COLLIDE_OFF;
    sb_engine *e = NULL;
    if (sb_create(&o, &e) != SB_OK) {
        fprintf(stderr, "sb_create: %s\n", sb_last_error(NULL));
        return 2;
    }
    unsigned char metadata[SB_METADATA_BYTES] = {0};
    unsigned short mapping[MAXP + MAXB] = {0};
    float particles[MAXP * 6] = {0};
    unsigned char beams[MAXB * SB_BEAM_STRIDE_V1] = {0};
    unsigned int *mdu = (unsigned int *)metadata;
    float *mdf = (float *)metadata;
    mdu[0] = 3; mdu[1] = 2;            /* particle vertex count, particle instance count */
    mdu[5] = 2; mdu[6] = 1;            /* beam vertex count, beam instance count */
    mdu[10] = MAXP; mdu[11] = MAXB;
    mdf[12] = 0.0f; mdf[13] = -0.5f;   /* gravity */
    mdf[14] = 0.5f; mdf[15] = 0.2f; mdf[16] = 0.5f; mdf[17] = 0.1f; mdf[18] = 0.0f; mdf[19] = 2.0f;
    mdf[20] = 1.0f;                    /* user strength */
    particles[0] = 100.0f; particles[1] = 500.0f;          /* particle 0 */
    particles[6] = 210.0f; particles[7] = 500.0f;          /* particle 1: beam stretched by 10 */
    mapping[0] = 0; mapping[1] = 1; mapping[MAXP + 0] = 0;
    unsigned short ends[2] = {0, 1};
    float f[7] = {100.0f, 100.0f, 100.0f, 2.0f, 0.0f, 0.5f, 10.0f}; /* length target last spring damp yield limit */
    memcpy(beams, ends, 4);
    memcpy(beams + 4, f, sizeof f);
    if (sb_write_buffers(e, metadata, sizeof metadata, mapping, sizeof mapping, particles, sizeof particles, beams,
                         sizeof beams) != SB_OK ||
        sb_frame(e) != SB_OK ||
        sb_load_buffers(e, metadata, sizeof metadata, mapping, sizeof mapping, particles, sizeof particles, beams,
                        sizeof beams) != SB_OK) {
        fprintf(stderr, "engine call failed: %s\n", sb_last_error(e));
        return 3;
    }
    printf("p0 = (%.6f, %.6f) v0 = (%.6f, %.6f)\n", particles[0], particles[1], particles[2], particles[3]);
    printf("p1 = (%.6f, %.6f) v1 = (%.6f, %.6f)\n", particles[6], particles[7], particles[8], particles[9]);
    /* the spring pulls the two together symmetrically; gravity is the same on both */
    int ok = particles[2] > 0.0f && particles[8] < 0.0f && particles[2] == -particles[8] && particles[3] == particles[9];
    sb_destroy(e);
    puts(ok ? "C_ABI_OK" : "C_ABI_UNEXPECTED");
    return ok ? 0 : 1;
}
