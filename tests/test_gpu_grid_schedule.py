"""The spatial hash keeps itself valid without a helper launch per substep (r04; csrc/sb_physics.h SbGridCtl): every substep
kernel takes the decision in its prologue, the LAGGED schedule pushes the next hash from inside the substep kernels one substep
ahead, a substep that moves somebody farther than predicted raises `abort` and the host recovers through a stretch of the
CLASSIC schedule (helper launch in front of every substep).  Whatever the schedule does, the particles are the reference's
collision loop (compute.wgsl:142-170), bit for bit: here against the oracle's all-pairs scan / grid mode, plus the
bookkeeping's own invariant (the bound really bounds)."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_parity import ALLPAIRS, GRID, assert_same

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CTL_WORDS = ["fresh", "need_build", "pushing", "abort", "cur", "executed", "builds", "since", "accum", "cx", "cy", "Cx", "Cy", "skin_min",
             "skin_max", "wide_next", "settled", "transient", "short_lived", "pad0", "pad1", "pad2", "skin"]   # (sb_physics.h SbGridCtl)
CTL_FLOATS = {"accum", "cx", "cy", "Cx", "Cy", "skin_min", "skin_max", "skin"}


def ctl(eng):
    """The SbGridCtl block the next launch reads (sb_get_info "grid_ctl_<word>")."""
    out = {}
    for i, name in enumerate(CTL_WORDS):
        v = eng.info("grid_ctl_%d" % i)
        out[name] = struct.unpack("<f", struct.pack("<I", v))[0] if name in CTL_FLOATS else v
    return out


def gas(sb, n_side=40, speed=8.0, seed=5, d=44.0):
    return sb.scenes.soup_buffers(n_side, n_side, d=d, origin=(40.0, 40.0), jitter=6.0, speed=speed, seed=seed)


def engines(sb, oracle, buf, bounds, ref_mode=ALLPAIRS, **kw):
    eng = sb.Engine(bounds_size=bounds, layout=buf.layout, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=GRID, **kw)
    ref = oracle.OracleEngine(bounds, 10.0, 64, buf.layout, ref_mode, threads=16)
    eng.write_buffers(buf)
    ref.write_buffers(buf)
    return eng, ref


def test_lagged_schedule_needs_no_helper_launches(sb, oracle):
    """(No beams in these gases: the engine never keeps a blocked plan beside them -- the switch SB_HYBRID is read once per process and
    is not touched here.)  A gas of 1600 free particles that meet all the time, 320 substeps in calls of 1 to 64: after the forced build of the
    first substep every hash is pushed by the substep kernels themselves (several of them), no launch is aborted, and the result
    is the all-pairs scan's."""
    buf = gas(sb)
    eng, ref = engines(sb, oracle, buf, 1900.0, tile_particles=128)
    assert eng.info("grid_schedule") == 0
    for n in (1, 2, 3, 64, 5, 64, 64, 1, 52, 64):
        eng.step(n)
        ref.step(n)
    got, exp = eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())
    info = {k: eng.info(k) for k in ("grid_builds", "grid_aborts", "grid_helper_launches", "grid_classic_substeps", "kernels_per_substep")}
    eng.destroy()
    assert np.isfinite(exp.particles).all()
    assert_same(got, exp, "gas, lagged schedule")
    assert not np.array_equal(got.particles[:, 2:4], buf.particles[:, 2:4])   # contacts did change velocities
    assert info["grid_builds"] >= 4 and info["grid_aborts"] == 0, info
    assert info["grid_helper_launches"] == 1 and info["grid_classic_substeps"] == 0 and info["kernels_per_substep"] == 1, info


def test_the_bound_bounds(sb, oracle):
    """The bookkeeping's own invariant, read off the device: after m substeps the control block says `accum` (D) and (Cx, Cy)
    (the drift C) for the READ state of substep m + 1 -- every particle's displacement since the hash in use was built, minus C,
    must be within D.  The hash of a forced build holds the uploaded positions (age = substeps since the upload)."""
    buf = sb.scenes.lattice_buffers(48, 40, d=30.0, origin=(200.0, 600.0), jitter=0.15, layout=2, velocity=(1.5, -2.0))
    P = buf.particle_count
    eng, ref = engines(sb, oracle, buf, 4000.0, ref_mode=GRID, tile_particles=256)
    p0 = buf.particles[:P, :2].astype("f8")
    checked = 0
    for m in range(1, 40):
        eng.step(1)
        c = ctl(eng)
        if c["builds"] != 1 or c["pushing"] or (c["fresh"] and m > 1):
            break   # (a second hash is on its way: its origin is no longer the upload)
        now = eng.load_buffers(buf.copy()).particles[:P, :2].astype("f8")
        # the block is the one substep m ran under: its bound covers the state substep m READ, i.e. after m - 1 substeps --
        # so compare with the positions one call earlier (kept from the last iteration)
        if m > 1:
            rel = np.abs(prev - p0 - np.array([c["Cx"], c["Cy"]])).max()
            assert rel <= c["accum"] * (1 + 1e-5) + 1e-6, (m, rel, c)
            checked += 1
        prev = now
    eng.destroy()
    assert checked >= 5, checked


def test_the_bound_covers_the_last_substep_when_the_hybrid_looks(sb, oracle):
    """the hybrid's look read the bound of the state BEFORE the last single substep and started its tracked
    run one substep short.  The look now takes the next substep's decision ahead of time (k_grid_settle) and a tracked run hands
    its bound back as a settled block: after every call that went through a look, `accum` and (Cx, Cy) must cover the CURRENT
    positions (the hash of the forced first build holds the uploaded ones) -- and the run is the oracle's, bit for bit."""
    buf = sb.scenes.lattice_buffers(48, 40, d=30.0, origin=(200.0, 600.0), jitter=0.15, layout=2, velocity=(1.5, -2.0))
    P = buf.particle_count
    eng, ref = engines(sb, oracle, buf, 4000.0, ref_mode=GRID, tile_particles=256)
    p0 = buf.particles[:P, :2].astype("f8")
    checked = 0
    for call in range(12):
        eng.step(5)
        ref.step(5)
        c = ctl(eng)
        if c["builds"] != 1:
            break   # (a second hash: its origin is no longer the upload)
        if c["settled"]:   # the block a tracked run (or a look) left: it describes the state the NEXT substep reads, i.e. now
            now = eng.load_buffers(buf.copy()).particles[:P, :2].astype("f8")
            rel = np.abs(now - p0 - np.array([c["Cx"], c["Cy"]])).max()
            assert rel <= c["accum"] * (1 + 1e-5) + 1e-6, (call, rel, c)
            checked += 1
    blocked = eng.info("hybrid_substeps")
    got, exp = eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())
    eng.destroy()
    assert_same(got, exp, "quiet lattice in calls of five")
    assert checked >= 3 and blocked >= 10, (checked, blocked)


def test_a_sudden_kick_aborts_and_recovers(sb, oracle):
    """The lagged schedule predicts the next substep's displacement from the last one's.  A user force that appears between two
    calls (engineWorker.ts:636-642: input is written per frame) and lasts two substeps throws every particle forward three to
    six units per substep where a fifth of one was predicted (the common drift the bound is measured against lags one substep
    behind): some launch finds its lists not known to be valid, raises `abort`, the launches behind it return at once, and
    the host redoes them behind a fresh hash in the classic schedule -- in which the gas, now fast, stays.  Same bits as the
    all-pairs scan; and the abort really happened."""
    buf = gas(sb, n_side=36, speed=6.0, seed=9)
    eng, ref = engines(sb, oracle, buf, 1800.0, tile_particles=128)
    eng.step(40)
    ref.step(40)
    for strength in (9000.0, -14000.0, 11000.0):   # applied_force x user_strength = acceleration: two substeps of it add strength / 32 to v
        ui = np.zeros(8, "f4")
        ui[0] = 1.0                         # user_strength
        ui[6], ui[7] = strength, 0.3 * strength   # applied_force
        eng.write_user_input(ui.tobytes())
        ref.write_user_input(ui.tobytes())
        eng.step(2)
        ref.step(2)
        ui[6] = ui[7] = 0.0
        eng.write_user_input(ui.tobytes())
        ref.write_user_input(ui.tobytes())
        eng.step(70)
        ref.step(70)
    got, exp = eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())
    info = {k: eng.info(k) for k in ("grid_builds", "grid_aborts", "grid_helper_launches", "grid_classic_substeps", "substeps_done")}
    eng.destroy()
    assert np.isfinite(exp.particles).all()
    assert_same(got, exp, "kicked gas")
    assert info["substeps_done"] == 40 + 3 * 72
    assert info["grid_aborts"] >= 1 and info["grid_classic_substeps"] >= 16, info


CLASSIC = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "oracle"))
import __graft_entry__ as ge
import oracle
sb = ge.load_package()
buf = sb.scenes.default_buffers(1, 256, 512)
eng = sb.Engine(layout=1, max_particles=buf.max_particles, max_beams=buf.max_beams)
ref = oracle.OracleEngine(1000.0, 10.0, 64, 1, 1, threads=4)
eng.write_buffers(buf); ref.write_buffers(buf)
assert eng.info("grid_schedule") == 1
for f in range(12):
    eng.frame(); ref.frame()
got, exp = eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())
assert np.array_equal(got.particles.view("u4"), exp.particles.view("u4")) and np.array_equal(got.beams.tobytes(), exp.beams.tobytes())
assert eng.info("grid_classic_substeps") == 12 * 64 and eng.info("grid_helper_launches") == 12 * 64 and eng.info("grid_aborts") == 0
assert eng.info("kernels_per_substep") == 2 and eng.info("grid_builds") > 2
print("CLASSIC_OK", eng.info("grid_builds"))
"""


NO_COOP = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "oracle"))
import __graft_entry__ as ge
import oracle
sb = ge.load_package()
pile, bounds = sb.scenes.blob_pile_buffers(8, 4, gap=19.6)
eng = sb.Engine(bounds_size=bounds, layout=2, max_particles=pile.max_particles, max_beams=pile.max_beams, collision_mode=2, tile_particles=128)
eng.write_buffers(pile)
eng.step(200)
got = eng.load_buffers(pile.copy())
ref = oracle.OracleEngine(bounds, 10.0, 64, 2, 2, threads=8)
ref.write_buffers(pile)
ref.step(200)
exp = ref.load_buffers(pile.copy())
assert np.array_equal(got.particles.view("u4"), exp.particles.view("u4")) and got.beams.tobytes() == exp.beams.tobytes()
assert eng.info("grid_builds") >= 3
print("COOP_RUN_OK")
"""


def test_lists_made_per_particle_when_the_cooperative_build_is_off(sb, oracle):
    """The neighbour lists of a tile are made by its workgroup together through LDS (sb_lists_cooperative); a tile that has
    scattered falls back on every particle walking the hash for itself.  SB_GRID_COOP=0 takes that path everywhere: same bits
    (a pile of touching blobs, 200 substeps, both against the oracle's grid mode)."""
    pile, bounds = sb.scenes.blob_pile_buffers(8, 4, gap=19.6)
    ref = oracle.OracleEngine(bounds, 10.0, 64, 2, GRID, threads=8)
    ref.write_buffers(pile)
    ref.step(200)
    exp = ref.load_buffers(pile.copy())
    eng = sb.Engine(bounds_size=bounds, layout=2, max_particles=pile.max_particles, max_beams=pile.max_beams, collision_mode=GRID, tile_particles=128)
    eng.write_buffers(pile)
    eng.step(200)
    got = eng.load_buffers(pile.copy())
    builds = eng.info("grid_builds")
    eng.destroy()
    assert_same(got, exp, "pile, lists made together")
    assert builds >= 3
    env = dict(os.environ, SB_GRID_COOP="0", SB_HYBRID="0", GRAFT_REPO_ROOT=ROOT)   # (read once per process: own process)
    p = subprocess.run([sys.executable, "-c", NO_COOP], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "COOP_RUN_OK" in p.stdout, p.stdout + p.stderr


def test_classic_schedule_forced(sb):
    """SB_GRID_MODE=classic (read once per process: own process): the fallback schedule on its own -- a helper launch in front of
    every substep, which takes the substep's decision itself and builds when that says so -- on the reference's default scene
    with the engine's default options, 12 frames against the oracle's all-pairs scan."""
    env = dict(os.environ, SB_GRID_MODE="classic", SB_HYBRID="0", GRAFT_REPO_ROOT=ROOT)
    p = subprocess.run([sys.executable, "-c", CLASSIC], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "CLASSIC_OK" in p.stdout, p.stdout + p.stderr


WRAP = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "oracle"))
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import __graft_entry__ as ge
import oracle
sb = ge.load_package()
# 1. a pile across the wrap (the device's 32-bit count of substeps passes 2^32 at substep 46): same bits as the oracle's grid mode
pile, bounds = sb.scenes.blob_pile_buffers(8, 4, gap=19.6)
eng = sb.Engine(bounds_size=bounds, layout=2, max_particles=pile.max_particles, max_beams=pile.max_beams, collision_mode=2, tile_particles=128)
ref = oracle.OracleEngine(bounds, 10.0, 64, 2, 2, threads=8)
eng.write_buffers(pile); ref.write_buffers(pile)
for n in (30, 30, 64, 100):
    eng.step(n); ref.step(n)
    got, exp = eng.load_buffers(pile.copy()), ref.load_buffers(pile.copy())
    assert np.array_equal(got.particles.view("u4"), exp.particles.view("u4")) and got.beams.tobytes() == exp.beams.tobytes(), n
assert eng.info("grid_builds") >= 3 and eng.info("grid_helper_launches") <= 1
eng.destroy()
# 2. a kick that makes a launch abort right behind the wrap: the host's roll-back counts across it
buf = sb.scenes.soup_buffers(36, 36, d=44.0, origin=(40.0, 40.0), jitter=6.0, speed=6.0, seed=9)
eng = sb.Engine(bounds_size=1800.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=2, tile_particles=128)
ref = oracle.OracleEngine(1800.0, 10.0, 64, 2, 1, threads=16)
eng.write_buffers(buf); ref.write_buffers(buf)
eng.step(40); ref.step(40)
ui = np.zeros(8, "f4"); ui[0] = 1.0; ui[6], ui[7] = 9000.0, 2700.0
eng.write_user_input(ui.tobytes()); ref.write_user_input(ui.tobytes())
eng.step(2); ref.step(2)
ui[6] = ui[7] = 0.0
eng.write_user_input(ui.tobytes()); ref.write_user_input(ui.tobytes())
eng.step(70); ref.step(70)
got, exp = eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())
assert np.isfinite(exp.particles).all()
assert np.array_equal(got.particles.view("u4"), exp.particles.view("u4"))
assert eng.info("grid_aborts") >= 1 and eng.info("substeps_done") == 112
print("WRAP_OK", eng.info("grid_aborts"))
"""


def test_the_substep_count_wraps_without_a_trace():
    """The device counts executed substeps in 32 bits; the host picks the set of displacement slots by its own count modulo 3 and
    rolls back after an abort by the difference of the two.  SB_GRID_EXECUTED0 (a test hook, read once per process) starts every
    upload 46 substeps short of 2^32 -- two days of a busy engine: a pile and a kicked gas cross the wrap bit for bit (until r04 the
    host's count was 32 bits wide too, and 2^32 is not a multiple of 3)."""
    env = dict(os.environ, SB_GRID_EXECUTED0=str(2 ** 32 - 46), SB_HYBRID="0", GRAFT_REPO_ROOT=ROOT)
    p = subprocess.run([sys.executable, "-c", WRAP], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "WRAP_OK" in p.stdout, p.stdout + p.stderr
