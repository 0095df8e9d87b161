"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every
symbol include/softbody.h declares; with no GPU present it fails LOUDLY (no CPU fallback)."""
import ctypes

import numpy as np
import pytest


def test_library_exports_every_declared_symbol(sb):
    import __graft_entry__ as ge
    ge.build()
    L = sb.engine.load_library()
    names = sb.engine.declared_symbols()
    assert len(names) >= 18 and "sb_write_buffers" in names and "sb_frame" in names
    for n in names:
        assert hasattr(L, n), n
    assert L.sb_abi_version() == 1


def test_default_options_match_reference_defaults(sb):
    L = sb.engine.load_library()
    o = sb.engine.SbOptions()
    L.sb_default_options(ctypes.byref(o))
    assert o.struct_size == ctypes.sizeof(sb.engine.SbOptions) == 64
    # engineWorker.ts:39-41, engineMapping.ts:362-363
    assert (o.bounds_size, o.particle_radius, o.subticks) == (1000.0, 10.0, 64)
    assert (o.max_particles, o.max_beams, o.layout) == (65536, 65536, 1)


def test_no_cpu_fallback(sb):
    """Without a GPU sb_create must fail with SB_ERR_NO_DEVICE, like the reference throwing
    TypeError when WebGPU is missing (engineWorker.ts:86,93,98)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(sb.engine.EngineError) as ei:
        sb.Engine()
    assert ei.value.status == 3 and "no CPU fallback" in str(ei.value)


def test_bad_options_rejected(sb):
    L = sb.engine.load_library()
    o = sb.engine.SbOptions()
    L.sb_default_options(ctypes.byref(o))
    h = ctypes.c_void_p()
    o.max_particles = 70000  # v1 is u16
    assert L.sb_create(ctypes.byref(o), ctypes.byref(h)) == 1
    assert b"65536" in L.sb_last_error(None)
    L.sb_default_options(ctypes.byref(o))
    o.struct_size = 8
    assert L.sb_create(ctypes.byref(o), ctypes.byref(h)) == 1


def test_layout_golden_bytes(sb):
    """Byte layouts derived by hand from engineMapping.ts (Particle.to :118-124, Beam.to :178-194,
    Metadata ctor :252-273)."""
    buf = sb.Buffers(1, 4, 4)
    beams = np.zeros(1, sb.layout.BEAM_DTYPE[1])
    beams[0] = (1, 2, 30.0, 29.0, 31.0, 50.0, 700.0, 0.2, 0.5, 0.0, 0.0)
    buf.set_scene(np.array([[1, 2, 3, 4, 5, 6], [0, 0, 0, 0, 0, 0], [7, 8, 0, 0, 0, 0]], "f4"), beams)
    assert buf.particles[0].tobytes() == np.array([1, 2, 3, 4, 5, 6], "<f4").tobytes()
    rec = buf.beams[:1].tobytes()
    assert len(rec) == 40
    assert rec[:4] == (1 | (2 << 16)).to_bytes(4, "little")          # packed u16 pair, compute.wgsl:99-100
    assert rec[4:32] == np.array([30, 29, 31, 50, 700, 0.2, 0.5], "<f4").tobytes()
    md = buf.metadata
    assert md.nbytes == 112
    assert md[0] == 3 and md[5] == 2 and md[10] == 4 and md[11] == 4    # :255-259
    assert md[1] == 3 and md[6] == 1
    assert md.view("<f4")[12:20].tolist() == pytest.approx([0, -0.5, 0.5, 0.2, 0.5, 0.1, 0.001, 2])
    assert md.view("<f4")[20] == 1.0                                  # userStrength, :263
    assert buf.mapping.dtype.itemsize == 2 and list(buf.mapping[:3]) == [0, 1, 2] and buf.mapping[4] == 0
    # v2 widens only the indices
    b2 = sb.Buffers(2, 4, 4)
    assert b2.beams.dtype.itemsize == 44 and b2.mapping.dtype.itemsize == 4


def test_default_scene_counts_and_snapshot_roundtrip(sb):
    """main.ts:188-246 -> 119 particles / 299 beams; snapshot v1 format engineMapping.ts:377-430."""
    buf = sb.scenes.default_buffers(1)
    assert (buf.particle_count, buf.beam_count) == (119, 299)
    snap = buf.create_snapshot()
    assert len(snap) == 12 + 32 + 119 * 2 + 119 * 24 + 299 * 2 + 299 * 40
    head = np.frombuffer(snap, "<u2", 6)
    assert list(head[:5]) == [238, 2856, 598, 11960, 32]
    other = sb.Buffers(1, 65536, 65536)
    assert other.load_snapshot(snap)
    assert other.particle_count == 119 and other.beam_count == 299
    assert np.array_equal(other.particles[:119], buf.particles[:119])
    assert other.beams[:299].tobytes() == buf.beams[:299].tobytes()
    # first rectangle: addRectangle(185, 10, 60, 2, 2, ...) main.ts:218
    assert buf.particles[:4, :2].tolist() == [[185, 10], [185, 70], [245, 10], [245, 70]]
    b0 = buf.beams[:6]
    assert [(int(x["a"]), int(x["b"])) for x in b0] == [(0, 1), (0, 2), (0, 3), (1, 3), (1, 2), (2, 3)]


def test_lattice_config2_shape(sb):
    """BASELINE config 2 generator: 1000x1000 -> 1M particles / 2 996 001 beams."""
    w, h = 100, 80
    buf = sb.scenes.lattice_buffers(w, h)
    assert buf.particle_count == w * h
    assert buf.beam_count == (h - 1) * w + (w - 1) * h + (w - 1) * (h - 1)
    assert 999 * 1000 * 2 + 999 * 999 == 2996001
