"""Byte layouts of the engine's host-visible buffers (numpy views).

Mirrors /root/reference/src/engineMapping.ts byte for byte:
  * Particle  stride 24  = 6 x f32 [px, py, vx, vy, ax, ay]        (engineMapping.ts:103,118-124)
  * Beam v1   stride 40  = u16 a, u16 b, 9 x f32                   (engineMapping.ts:151,178-194)
  * Beam v2   stride 44  = u32 a, u32 b, 9 x f32                   (wide layout)
  * Metadata  112 bytes                                            (engineMapping.ts:239-262)
  * Mapping   u16 (v1) / u32 (v2) [maxParticles + maxBeams]        (engineMapping.ts:355,367-368)
  * Snapshot  v1 (engineMapping.ts:377-430) and a wide v2 with u32 section sizes

The v1 layout is the reference's; v2 only widens the indices so scenes beyond
65 536 particles fit (the reference hard-caps at u16, engineMapping.ts:362-363).
"""
import struct

import numpy as np

LAYOUT_V1 = 1
LAYOUT_V2 = 2

METADATA_BYTES = 112
PARTICLE_STRIDE = 24
BEAM_STRIDE = {LAYOUT_V1: 40, LAYOUT_V2: 44}

BEAM_FLOATS = ("length", "target_length", "last_length", "spring", "damp",
               "yield_strain", "strain_break_limit", "strain", "stress")

BEAM_DTYPE = {
    LAYOUT_V1: np.dtype([("a", "<u2"), ("b", "<u2")] + [(n, "<f4") for n in BEAM_FLOATS]),
    LAYOUT_V2: np.dtype([("a", "<u4"), ("b", "<u4")] + [(n, "<f4") for n in BEAM_FLOATS]),
}
assert BEAM_DTYPE[LAYOUT_V1].itemsize == 40 and BEAM_DTYPE[LAYOUT_V2].itemsize == 44

MAPPING_DTYPE = {LAYOUT_V1: np.dtype("<u2"), LAYOUT_V2: np.dtype("<u4")}

# metadata u32-word offsets (compute.wgsl:29-54)
MD_PARTICLE_V_C, MD_PARTICLE_I_C = 0, 1
MD_BEAM_V_C, MD_BEAM_I_C = 5, 6
MD_MAX_PARTICLES, MD_MAX_BEAMS = 10, 11
MD_CONSTANTS = 12      # 8 floats: gravity.xy, border_elasticity, border_friction, elasticity, friction, drag_coeff, drag_exp
MD_USER_INPUT = 20     # 8 words: user_strength, mouse_active(u32), mouse_pos.xy, mouse_vel.xy, applied_force.xy

DEFAULT_CONSTANTS = dict(gravity=(0.0, -0.5), border_elasticity=0.5, border_friction=0.2,
                         elasticity=0.5, friction=0.1, drag_coeff=0.001, drag_exp=2.0)  # engineMapping.ts:264-272


class Buffers:
    """The four host ArrayBuffers a BufferMapper owns (engineMapping.ts:342-345)."""

    def __init__(self, layout=LAYOUT_V1, max_particles=65536, max_beams=65536):
        if layout == LAYOUT_V1 and (max_particles > 65536 or max_beams > 65536):
            raise ValueError("v1 layout is limited to 65536 particles/beams (u16 indices)")
        self.layout = layout
        self.max_particles = int(max_particles)
        self.max_beams = int(max_beams)
        self.metadata = np.zeros(METADATA_BYTES // 4, dtype="<u4")
        self.particles = np.zeros((self.max_particles, 6), dtype="<f4")
        self.beams = np.zeros(self.max_beams, dtype=BEAM_DTYPE[layout])
        self.mapping = np.zeros(self.max_particles + self.max_beams, dtype=MAPPING_DTYPE[layout])
        # Metadata ctor, engineMapping.ts:252-273
        self.metadata[MD_PARTICLE_V_C] = 3
        self.metadata[MD_BEAM_V_C] = 2
        self.metadata[MD_MAX_PARTICLES] = self.max_particles
        self.metadata[MD_MAX_BEAMS] = self.max_beams
        self.user_strength = 1.0
        self.set_physics_constants(**DEFAULT_CONSTANTS)

    # ---- metadata accessors (engineMapping.ts:275-325)
    @property
    def _mdf(self):
        return self.metadata.view("<f4")

    @property
    def particle_count(self):
        return int(self.metadata[MD_PARTICLE_I_C])

    @particle_count.setter
    def particle_count(self, c):
        self.metadata[MD_PARTICLE_I_C] = c

    @property
    def beam_count(self):
        return int(self.metadata[MD_BEAM_I_C])

    @beam_count.setter
    def beam_count(self, c):
        self.metadata[MD_BEAM_I_C] = c

    def set_physics_constants(self, gravity, border_elasticity, border_friction, elasticity,
                              friction, drag_coeff, drag_exp):
        f = self._mdf
        f[MD_CONSTANTS + 0], f[MD_CONSTANTS + 1] = gravity
        f[MD_CONSTANTS + 2] = border_elasticity
        f[MD_CONSTANTS + 3] = border_friction
        f[MD_CONSTANTS + 4] = elasticity
        f[MD_CONSTANTS + 5] = friction
        f[MD_CONSTANTS + 6] = drag_coeff
        f[MD_CONSTANTS + 7] = drag_exp

    def get_physics_constants(self):
        f = self._mdf
        return dict(gravity=(float(f[12]), float(f[13])), border_elasticity=float(f[14]),
                    border_friction=float(f[15]), elasticity=float(f[16]), friction=float(f[17]),
                    drag_coeff=float(f[18]), drag_exp=float(f[19]))

    @property
    def user_strength(self):
        return float(self._mdf[MD_USER_INPUT])

    @user_strength.setter
    def user_strength(self, v):
        self._mdf[MD_USER_INPUT] = v

    def set_user_input(self, applied_force=(0.0, 0.0), mouse_pos=(0.0, 0.0), mouse_vel=(0.0, 0.0),
                       mouse_active=False):
        """engineMapping.ts:317-322"""
        f = self._mdf
        self.metadata[MD_USER_INPUT + 1] = 1 if mouse_active else 0
        f[MD_USER_INPUT + 2], f[MD_USER_INPUT + 3] = mouse_pos
        f[MD_USER_INPUT + 4], f[MD_USER_INPUT + 5] = mouse_vel
        f[MD_USER_INPUT + 6], f[MD_USER_INPUT + 7] = applied_force

    def user_input_bytes(self):
        """The 32 bytes Metadata.writeUserInput uploads (engineMapping.ts:323-325)."""
        return self.metadata[MD_USER_INPUT:MD_USER_INPUT + 8].tobytes()

    # ---- scene authoring (identity mapping, like BufferMapper.writeState, engineMapping.ts:500-517)
    def set_scene(self, particles, beams):
        """particles: (P,2|4|6) float array; beams: structured array of this layout's BEAM_DTYPE
        (endpoints are particle data indices)."""
        particles = np.asarray(particles, dtype="<f4")
        P = particles.shape[0]
        B = beams.shape[0]
        if P > self.max_particles or B > self.max_beams:
            raise ValueError("scene exceeds buffer capacity")
        self.particles[:] = 0
        self.particles[:P, :particles.shape[1]] = particles
        self.beams[:] = 0
        self.beams[:B] = beams
        self.mapping[:] = 0
        self.mapping[:P] = np.arange(P, dtype=self.mapping.dtype)
        self.mapping[self.max_particles:self.max_particles + B] = np.arange(B, dtype=self.mapping.dtype)
        self.particle_count = P
        self.beam_count = B

    def copy(self):
        o = Buffers.__new__(Buffers)
        o.layout, o.max_particles, o.max_beams = self.layout, self.max_particles, self.max_beams
        o.metadata = self.metadata.copy()
        o.particles = self.particles.copy()
        o.beams = self.beams.copy()
        o.mapping = self.mapping.copy()
        return o

    # ---- snapshots
    def create_snapshot(self):
        """engineMapping.ts:377-401 for v1 (including its u16 size fields, which wrap for
        P > 2730 / B > 1638 exactly as the reference's do); for v2 the same sections with a
        'SBW2' magic and u32 sizes."""
        P, B = self.particle_count, self.beam_count
        isz = self.mapping.dtype.itemsize
        pm = self.mapping[:P].tobytes()
        pd = self.particles[:P].tobytes()
        bm = self.mapping[self.max_particles:self.max_particles + B].tobytes()
        bd = self.beams[:B].tobytes()
        consts = self.metadata[MD_CONSTANTS:MD_CONSTANTS + 8].tobytes()
        if self.layout == LAYOUT_V1:
            head = struct.pack("<6H", (P * isz) & 0xFFFF, len(pd) & 0xFFFF, (B * isz) & 0xFFFF,
                               len(bd) & 0xFFFF, 32, 0)
        else:
            head = b"SBW2" + struct.pack("<5I", len(pm), len(pd), len(bm), len(bd), 32)
        return head + consts + pm + pd + bm + bd

    def load_snapshot(self, buf):
        """engineMapping.ts:407-430.  Returns False when the snapshot does not fit."""
        buf = bytes(buf)
        if self.layout == LAYOUT_V1:
            pms, pds, bms, bds, mds = struct.unpack_from("<5H", buf, 0)
            off = 12
        else:
            if buf[:4] != b"SBW2":
                return False
            pms, pds, bms, bds, mds = struct.unpack_from("<5I", buf, 4)
            off = 24
        # the reference compares byte sizes to element counts (engineMapping.ts:418); kept as is
        if pms > self.max_particles * (1 if self.layout == LAYOUT_V1 else 4) or \
           bms > self.max_beams * (1 if self.layout == LAYOUT_V1 else 4):
            return False
        isz = self.mapping.dtype.itemsize
        self.metadata[MD_CONSTANTS:MD_CONSTANTS + mds // 4] = np.frombuffer(buf, "<u4", mds // 4, off)
        off += mds
        P, B = pms // isz, bms // isz
        self.mapping[:P] = np.frombuffer(buf, self.mapping.dtype, P, off)
        off += pms
        self.particles.reshape(-1)[:pds // 4] = np.frombuffer(buf, "<f4", pds // 4, off)
        off += pds
        self.mapping[self.max_particles:self.max_particles + B] = np.frombuffer(buf, self.mapping.dtype, B, off)
        off += bms
        self.beams.view("u1").reshape(-1)[:bds] = np.frombuffer(buf, "u1", bds, off)
        self.particle_count = P
        self.beam_count = B
        return True
