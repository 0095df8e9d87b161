"""ctypes wrapper of the CPU oracle (TEST INFRASTRUCTURE ONLY, see sb_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
It steps the reference's seven buffers (engineWorker.ts:136-176) held as numpy arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

COLLIDE_OFF, COLLIDE_ALLPAIRS, COLLIDE_GRID = 0, 1, 2


class _Params(ctypes.Structure):
    _fields_ = [("bounds_size", ctypes.c_float), ("particle_radius", ctypes.c_float),
                ("time_step", ctypes.c_float), ("layout", ctypes.c_int32),
                ("collision_mode", ctypes.c_int32), ("threads", ctypes.c_int32)]


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libsboracle.so")
        if not os.path.exists(path):
            build()
        # libgomp's default active spinning collapses when the team size changes between calls
        # (measured here: 8 threads slower than 1); it reads this at load time
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        L = ctypes.CDLL(path)
        vp = ctypes.c_void_p
        L.sbo_update.argtypes = [ctypes.POINTER(_Params), vp, vp, vp, vp, vp, vp, vp]
        L.sbo_update.restype = None
        L.sbo_delete.argtypes = [ctypes.POINTER(_Params), vp, vp, vp, ctypes.c_size_t]
        L.sbo_delete.restype = None
        L.sbo_step.argtypes = [ctypes.POINTER(_Params), vp, vp, vp, vp, vp, vp, vp, ctypes.c_uint32]
        L.sbo_step.restype = ctypes.c_int
        L.sbo_pow.argtypes = [ctypes.c_float, ctypes.c_float]
        L.sbo_pow.restype = ctypes.c_float
        L.sbo_f32_to_i32.argtypes = [ctypes.c_float]
        L.sbo_f32_to_i32.restype = ctypes.c_int32
        _LIB = L
    return _LIB


_VARIANTS = {}


def variant_lib(name):
    """One of the OTHER conformant readings of the WGSL text (sb_oracle.c SBO_VARIANT: "v1" = normalize by two
    divisions, "v2" = strain by division, "v3" = both, "v4" = normalize by inverseSqrt, "fma" = contraction
    allowed), for tools/tolerance_study.py; never the oracle.  None if this CPU cannot run it (fma)."""
    if name in _VARIANTS:
        return _VARIANTS[name]
    if name == "fma" and " fma " not in open("/proc/cpuinfo").read().replace("\n", " "):
        _VARIANTS[name] = None
        return None
    subprocess.run(["make", "-s", "-C", _HERE, "libsboracle_%s.so" % name], check=True)
    L = ctypes.CDLL(os.path.join(_HERE, "libsboracle_%s.so" % name))
    vp = ctypes.c_void_p
    L.sbo_step.argtypes = [ctypes.POINTER(_Params), vp, vp, vp, vp, vp, vp, vp, ctypes.c_uint32]
    L.sbo_step.restype = ctypes.c_int
    L.sbo_delete.argtypes = [ctypes.POINTER(_Params), vp, vp, vp, ctypes.c_size_t]
    L.sbo_delete.restype = None
    _VARIANTS[name] = L
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class OracleEngine:
    """Plays the role of engineWorker.ts for the oracle: owns particlesA/B, beams, mapping,
    metadata, particleForces, deleteMappings; write_buffers/frame/step/load_buffers follow
    engineWorker.ts:580-597, 646-665, 548-579."""

    def __init__(self, bounds_size=1000.0, particle_radius=10.0, subticks=64, layout=1,
                 collision_mode=COLLIDE_ALLPAIRS, threads=1, variant=None):
        self._lib = lib() if variant is None else variant_lib(variant)
        if self._lib is None:
            raise RuntimeError("oracle variant %r cannot run on this CPU" % variant)
        self.subticks = int(-(-int(subticks) // 2) * 2)  # engineWorker.ts:90
        self.prm = _Params(np.float32(bounds_size), np.float32(particle_radius),
                           np.float32(np.float32(1.0) / np.float32(self.subticks)),
                           layout, collision_mode, threads)
        self.layout = layout
        self.final_in_b = 0

    def write_buffers(self, buf):
        """engineWorker.ts:580-597: copy metadata, mapping, particles->A, beams; zero forces,
        delete mask, B."""
        self.max_particles, self.max_beams = buf.max_particles, buf.max_beams
        self.metadata = buf.metadata.copy()
        self.mapping = buf.mapping.copy()
        self.particles_a = buf.particles.copy()
        self.particles_b = np.zeros_like(buf.particles)
        self.beams = buf.beams.copy()
        self.forces = np.zeros(2 * buf.max_particles, dtype=np.int32)
        self.delete_words = (buf.max_particles + buf.max_beams + 31) // 32
        self.delete = np.zeros(self.delete_words, dtype=np.uint32)
        self.final_in_b = 0

    def write_user_input(self, bytes32):
        self.metadata[20:28] = np.frombuffer(bytes32, "<u4", 8)

    def set_physics_constants(self, consts8):
        self.metadata.view("<f4")[12:20] = np.asarray(consts8, dtype="<f4")

    def step(self, n):
        a, b = (self.particles_a, self.particles_b) if not self.final_in_b else (self.particles_b, self.particles_a)
        r = self._lib.sbo_step(ctypes.byref(self.prm), _p(self.metadata), _p(a), _p(b), _p(self.beams),
                           _p(self.mapping), _p(self.forces), _p(self.delete), n)
        self.final_in_b ^= r

    def delete_pass(self):
        self._lib.sbo_delete(ctypes.byref(self.prm), _p(self.metadata), _p(self.mapping), _p(self.delete),
                         self.delete_words)

    def frame(self):
        self.step(self.subticks)
        self.delete_pass()

    def load_buffers(self, buf):
        """engineWorker.ts:548-579 into a layout.Buffers."""
        buf.metadata[:] = self.metadata
        buf.particles[:] = self.particles_b if self.final_in_b else self.particles_a
        buf.beams[:] = self.beams
        buf.mapping[:] = self.mapping
        return buf
