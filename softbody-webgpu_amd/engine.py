"""ctypes binding of the C ABI in include/softbody.h (libsoftbody_hip.so).

The HIP library is the product; this module only marshals numpy buffers into it.  There is
no CPU fallback: if the library is missing, lacks a declared symbol, or finds no GPU, the
calls raise.
"""
import ctypes
import os
import re

import numpy as np

from .layout import BEAM_STRIDE, LAYOUT_V1, METADATA_BYTES, PARTICLE_STRIDE, Buffers

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SOFTBODY_HIP_LIB") or os.path.join(_HERE, "csrc", "libsoftbody_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "softbody.h")

COLLIDE_OFF, COLLIDE_ALLPAIRS, COLLIDE_GRID = 0, 1, 2
PATH_AUTO, PATH_ATOMIC, PATH_TILED = 0, 1, 2

STATUS = {0: "SB_OK", 1: "SB_ERR_INVALID", 2: "SB_ERR_HIP", 3: "SB_ERR_NO_DEVICE", 4: "SB_ERR_OOM",
          5: "SB_ERR_STATE", 6: "SB_ERR_UNSUPPORTED"}


class EngineError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (STATUS.get(status, status), message))
        self.status = status


class SbOptions(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("bounds_size", ctypes.c_float),
                ("particle_radius", ctypes.c_float), ("subticks", ctypes.c_uint32),
                ("max_particles", ctypes.c_uint32), ("max_beams", ctypes.c_uint32),
                ("layout", ctypes.c_uint32), ("collision_mode", ctypes.c_uint32),
                ("path", ctypes.c_uint32), ("tile_particles", ctypes.c_uint32),
                ("device_ordinal", ctypes.c_int32), ("grid_skin", ctypes.c_float),
                ("block_substeps", ctypes.c_uint32), ("reserved", ctypes.c_uint32 * 3)]


_lib = None


def declared_symbols():
    """Every function include/softbody.h declares."""
    src = open(HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sb_[a-z_0-9]+)\s*\(", src)))


def load_library():
    """dlopen the HIP engine and check it exports every symbol the header declares."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % LIB_PATH)
    # One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so.7 /
    # libhsa-runtime64.so.1 and load them by path; if this library pulled in /opt/rocm's copies
    # first, a later `import torch` would bring up a SECOND runtime in the process, whose device
    # discovery fails ("No HIP GPUs are available").  Loading torch's copy first makes our
    # DT_NEEDED libamdhip64.so.7 resolve to the already-loaded one, so torch (RCCL, streams) and the
    # engine share a single runtime and can share hipStream_t handles (halo.TorchTransport).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    if missing:
        raise ImportError("libsoftbody_hip.so lacks symbols declared in softbody.h: %s" % missing)
    vp, sz, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32
    L.sb_default_options.argtypes = [ctypes.POINTER(SbOptions)]
    L.sb_default_options.restype = None
    L.sb_create.argtypes = [ctypes.POINTER(SbOptions), ctypes.POINTER(vp)]
    L.sb_destroy.argtypes = [vp]
    L.sb_write_buffers.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, sz]
    L.sb_load_buffers.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, sz]
    L.sb_write_user_input.argtypes = [vp, vp]
    L.sb_set_physics_constants.argtypes = [vp, vp]
    L.sb_get_physics_constants.argtypes = [vp, vp]
    L.sb_frame.argtypes = [vp]
    L.sb_step.argtypes = [vp, u32]
    L.sb_delete_pass.argtypes = [vp]
    L.sb_halo_delete_ghosts.argtypes = [vp]
    L.sb_sync.argtypes = [vp]
    L.sb_step_timed.argtypes = [vp, u32, ctypes.POINTER(ctypes.c_float)]
    L.sb_mark.argtypes = [vp, u32]
    L.sb_mark_elapsed.argtypes = [vp, u32, u32, ctypes.POINTER(ctypes.c_float)]
    L.sb_get_counts.argtypes = [vp, ctypes.POINTER(u32), ctypes.POINTER(u32)]
    L.sb_get_info.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64)]
    L.sb_halo_configure.argtypes = [vp, vp, u32, vp, u32, vp, u32, vp, u32]
    L.sb_halo_set_layout.argtypes = [vp, vp, vp, vp, vp]
    L.sb_halo_pack.argtypes = [vp, vp]
    L.sb_peer_mailbox.argtypes = [vp, ctypes.POINTER(vp), vp, ctypes.POINTER(ctypes.c_uint64)]
    L.sb_peer_map.argtypes = [vp, vp, ctypes.POINTER(vp)]
    L.sb_peer_connect.argtypes = [vp, u32, ctypes.POINTER(vp), vp, vp, vp, vp, vp, u32]
    L.sb_peer_exchange.argtypes = [vp]
    L.sb_halo_unpack.argtypes = [vp, vp]
    L.sb_get_stream.argtypes = [vp, ctypes.POINTER(vp)]
    f32 = ctypes.c_float
    L.sb_partition_create.argtypes = [u32, u32, u32, vp, vp, vp, vp, u32, u32, f32, ctypes.POINTER(vp)]
    L.sb_partition_destroy.argtypes = [vp]
    L.sb_partition_rank_counts.argtypes = [vp, u32, ctypes.POINTER(u32 * 8)]
    L.sb_partition_layout.argtypes = [vp, ctypes.POINTER(u32)]
    L.sb_partition_rank_scene.argtypes = [vp, u32, u32, u32, vp, vp, vp, vp]
    L.sb_partition_rank_ids.argtypes = [vp, u32, vp, vp, vp, vp]
    L.sb_partition_peer_counts.argtypes = [vp, u32, u32, ctypes.POINTER(u32), ctypes.POINTER(u32 * 4)]
    L.sb_partition_peer_lists.argtypes = [vp, u32, u32, vp, vp, vp, vp]
    L.sb_last_error.argtypes = [vp]
    L.sb_last_error.restype = ctypes.c_char_p
    L.sb_abi_version.restype = u32
    for name in declared_symbols():
        fn = getattr(L, name)
        if name not in ("sb_default_options", "sb_last_error", "sb_abi_version"):
            fn.restype = ctypes.c_int
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Engine:
    """Host-side handle of one engine; method names follow engineWorker.ts
    (writeBuffers / loadBuffers / frame) and the C ABI."""

    def __init__(self, bounds_size=1000.0, particle_radius=10.0, subticks=64, layout=LAYOUT_V1,
                 max_particles=65536, max_beams=65536, collision_mode=COLLIDE_GRID,
                 path=PATH_AUTO, tile_particles=0, device=0, grid_skin=0.0, block_substeps=0):
        L = load_library()
        o = SbOptions()
        L.sb_default_options(ctypes.byref(o))
        o.bounds_size, o.particle_radius, o.subticks = bounds_size, particle_radius, subticks
        o.max_particles, o.max_beams, o.layout = max_particles, max_beams, layout
        o.collision_mode, o.path, o.tile_particles, o.device_ordinal = collision_mode, path, tile_particles, device
        o.grid_skin = grid_skin
        o.block_substeps = block_substeps
        self._h = ctypes.c_void_p()
        st = L.sb_create(ctypes.byref(o), ctypes.byref(self._h))
        if st != 0:
            self._h = None
            raise EngineError(st, L.sb_last_error(None).decode())
        self.layout, self.max_particles, self.max_beams = layout, max_particles, max_beams
        self.collision_mode = collision_mode
        self.subticks = (subticks + 1) // 2 * 2

    def _check(self, st):
        if st != 0:
            raise EngineError(st, load_library().sb_last_error(self._h).decode())

    def destroy(self):
        if self._h:
            load_library().sb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def write_buffers(self, buf: Buffers):
        self._check(load_library().sb_write_buffers(
            self._h, _ptr(buf.metadata), buf.metadata.nbytes, _ptr(buf.mapping), buf.mapping.nbytes,
            _ptr(buf.particles), buf.particles.nbytes, _ptr(buf.beams), buf.beams.nbytes))

    def load_buffers(self, buf: Buffers):
        self._check(load_library().sb_load_buffers(
            self._h, _ptr(buf.metadata), buf.metadata.nbytes, _ptr(buf.mapping), buf.mapping.nbytes,
            _ptr(buf.particles), buf.particles.nbytes, _ptr(buf.beams), buf.beams.nbytes))
        return buf

    def write_user_input(self, bytes32):
        b = (ctypes.c_ubyte * 32).from_buffer_copy(bytes32)
        self._check(load_library().sb_write_user_input(self._h, ctypes.cast(b, ctypes.c_void_p)))

    def set_physics_constants(self, consts8):
        a = np.asarray(consts8, dtype="<f4")
        assert a.shape == (8,)
        self._check(load_library().sb_set_physics_constants(self._h, _ptr(a)))

    def get_physics_constants(self):
        a = np.zeros(8, dtype="<f4")
        self._check(load_library().sb_get_physics_constants(self._h, _ptr(a)))
        return a

    def frame(self):
        self._check(load_library().sb_frame(self._h))

    def step(self, n):
        self._check(load_library().sb_step(self._h, n))

    def delete_pass(self):
        self._check(load_library().sb_delete_pass(self._h))

    def halo_delete_ghosts(self):
        self._check(load_library().sb_halo_delete_ghosts(self._h))

    def sync(self):
        self._check(load_library().sb_sync(self._h))

    def step_timed(self, n):
        ms = ctypes.c_float()
        self._check(load_library().sb_step_timed(self._h, n, ctypes.byref(ms)))
        return ms.value

    def mark(self, slot):
        """Record time mark `slot` at the current end of the engine's stream (does not wait)."""
        self._check(load_library().sb_mark(self._h, slot))

    def mark_elapsed(self, a, b):
        """Device milliseconds from mark a to mark b (waits for b)."""
        ms = ctypes.c_float()
        self._check(load_library().sb_mark_elapsed(self._h, a, b, ctypes.byref(ms)))
        return ms.value

    def counts(self):
        p, b = ctypes.c_uint32(), ctypes.c_uint32()
        self._check(load_library().sb_get_counts(self._h, ctypes.byref(p), ctypes.byref(b)))
        return p.value, b.value

    def info(self, key):
        v = ctypes.c_uint64()
        self._check(load_library().sb_get_info(self._h, key.encode(), ctypes.byref(v)))
        return v.value

    def kernel_name(self):
        """Name (up to the template arguments) of the kernel that does the substeps of this engine, as rocprofv3
        lists it."""
        if self.info("path") != PATH_TILED:
            return "k_beams_atomic+k_particles"
        if self.info("substeps_per_launch") > 1:
            return "k_substep_blocked"
        return "k_substep_tiled_grid" if self.collision_mode == COLLIDE_GRID else "k_substep_tiled"

    def sync_quiet(self):
        """sync() that swallows a reported device-side timeout (used when abandoning a failed exchange set-up)."""
        try:
            self.sync()
        except EngineError:
            pass

    def halo_configure(self, ghost_particles, send_particles, ghost_beams=(), send_beams=()):
        a = [np.ascontiguousarray(x, dtype="<u4") for x in (ghost_particles, send_particles, ghost_beams, send_beams)]
        self._check(load_library().sb_halo_configure(self._h, _ptr(a[0]), a[0].size, _ptr(a[1]), a[1].size,
                                                     _ptr(a[2]), a[2].size, _ptr(a[3]), a[3].size))

    def halo_set_layout(self, send_particle_off, send_beam_off, ghost_particle_off, ghost_beam_off):
        a = [np.ascontiguousarray(x, dtype="<u4") for x in (send_particle_off, send_beam_off, ghost_particle_off,
                                                             ghost_beam_off)]
        self._check(load_library().sb_halo_set_layout(self._h, *[_ptr(x) for x in a]))

    def peer_mailbox(self):
        """(local device pointer, 64-byte IPC handle, bytes) of this engine's mailbox (allocated on first call)."""
        ptr, nbytes = ctypes.c_void_p(), ctypes.c_uint64()
        handle = ctypes.create_string_buffer(64)
        self._check(load_library().sb_peer_mailbox(self._h, ctypes.byref(ptr), handle, ctypes.byref(nbytes)))
        return ptr.value, handle.raw, nbytes.value

    def peer_map(self, handle):
        ptr = ctypes.c_void_p()
        self._check(load_library().sb_peer_map(self._h, ctypes.c_char_p(bytes(handle)), ctypes.byref(ptr)))
        return ptr.value

    def peer_connect(self, mailboxes, peer_recv_floats, send_begin, send_len, dst_begin, their_slot, timeout_ms=0):
        n = len(mailboxes)
        boxes = (ctypes.c_void_p * max(n, 1))(*mailboxes)
        a = [np.ascontiguousarray(x, dtype="<u4") for x in (peer_recv_floats, send_begin, send_len, dst_begin, their_slot)]
        self._check(load_library().sb_peer_connect(self._h, n, boxes, *[_ptr(x) for x in a], int(timeout_ms)))

    def peer_exchange(self):
        self._check(load_library().sb_peer_exchange(self._h))

    def halo_pack(self, device_ptr):
        self._check(load_library().sb_halo_pack(self._h, ctypes.c_void_p(device_ptr)))

    def halo_unpack(self, device_ptr):
        self._check(load_library().sb_halo_unpack(self._h, ctypes.c_void_p(device_ptr)))

    def stream(self):
        s = ctypes.c_void_p()
        self._check(load_library().sb_get_stream(self._h, ctypes.byref(s)))
        return s.value
