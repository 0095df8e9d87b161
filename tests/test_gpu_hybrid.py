"""SB_COLLIDE_GRID with a blocked plan beside the tiling (csrc/sb_api.hip hybrid_substeps): while every neighbour list of the
spatial hash is empty, the collision loop the reference always runs (compute.wgsl:142-170) is a no-op and the engine advances
K substeps per launch out of LDS and registers, tracking what the particles move; a launch that uses up the hash's skin is
not counted and its substeps are redone one by one.  Everything here is bit for bit against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_parity import GRID, OFF, assert_same

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(sb, oracle, buf, *, bounds, n=0, frames=0, calls=1, ref_mode=None, subticks=64):
    eng = sb.Engine(bounds_size=bounds, subticks=subticks, layout=buf.layout, max_particles=buf.max_particles, max_beams=buf.max_beams,
                    collision_mode=GRID)
    ref = oracle.OracleEngine(bounds, 10.0, subticks, buf.layout, GRID if ref_mode is None else ref_mode, threads=16)
    eng.write_buffers(buf)
    ref.write_buffers(buf)
    for _ in range(frames):
        eng.frame()
        ref.frame()
    for _ in range(calls if n else 0):
        eng.step(n)
        ref.step(n)
    got, exp = eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())
    info = {k: eng.info(k) for k in ("hybrid", "hybrid_substeps", "hybrid_failed", "grid_builds", "substeps_done", "hybrid_launches",
                                     "hybrid_validate_launches")}
    eng.destroy()
    return got, exp, info


def test_quiet_lattice_runs_blocked_and_matches_the_oracle(sb, oracle):
    """A falling 128 x 96 lattice at spacing 30 (nothing within 2r + 2 skin = 28): after the first stretch of single substeps
    that builds the hash, the run goes blocked -- and equals the oracle's grid mode AND its collision-free mode."""
    buf = sb.scenes.lattice_buffers(128, 96, d=30.0, origin=(300.0, 900.0), jitter=1.0, layout=2, velocity=(0.4, -1.0))
    got, exp, info = run(sb, oracle, buf, bounds=6000.0, n=150, calls=2)
    assert info["hybrid"] >= 2 and info["substeps_done"] == 300
    assert info["hybrid_substeps"] >= 200, info
    # r04: a tracked launch is validated by the prologue of the launch behind it; only the last launch of a run gets a launch for it
    assert info["hybrid_launches"] >= 30 and info["hybrid_validate_launches"] * 4 <= info["hybrid_launches"], info
    assert_same(got, exp, "quiet lattice, hybrid")
    _, off, _ = run(sb, oracle, buf, bounds=6000.0, n=150, calls=2, ref_mode=OFF)
    assert_same(got, off, "quiet lattice == collisions off")


def test_frames_with_yield_break_and_delete_passes(sb, oracle):
    """Plastic yield, break flags raised inside blocked launches, per-frame delete passes of the tiled layout: the flags cross
    from the blocked layout's mask into the tiled one, dead beams die in both plans."""
    buf = sb.scenes.lattice_buffers(40, 30, d=30.0, origin=(30.0, 30.0), spring=50.0, damp=100.0, yield_strain=0.05, strain_limit=0.12,
                                    layout=2, velocity=(-40.0, -35.0), slack=8, jitter=0.5)
    got, exp, info = run(sb, oracle, buf, bounds=4000.0, frames=4, n=9)
    assert exp.beam_count < buf.beam_count, "scene must break beams"
    assert info["hybrid_substeps"] > 0, info
    assert_same(got, exp, "yield + break + delete, hybrid")


def test_two_blobs_meet(sb, oracle):
    """Two lattice blobs on a collision course: quiet (blocked, in runs as long as the hash's skin lasts) while apart, blocked
    against the shrinking gap of the closest listed pair once they are within reach of each other, substep by substep when they
    touch -- bit-exact through the transitions.  (At four times the speed the skin lasts ten substeps and runs are not worth
    starting: test_fast_blobs_stay_on_single_substeps.)"""
    a = sb.scenes.lattice_buffers(24, 24, d=30.0, origin=(200.0, 400.0), jitter=0.5, layout=2, velocity=(4.0, 0.0))
    b = sb.scenes.lattice_buffers(24, 24, d=30.0, origin=(950.0, 415.0), jitter=0.5, layout=2, velocity=(-4.0, 0.0), seed=7)
    P, B = a.particle_count, a.beam_count
    pv = np.concatenate([a.particles[:P], b.particles[:P]])
    beams = np.concatenate([a.beams[:B], b.beams[:B]])
    beams["a"][B:] += P
    beams["b"][B:] += P
    buf = sb.Buffers(2, 2 * P, 2 * B)
    buf.set_scene(pv, beams)
    got, exp, info = run(sb, oracle, buf, bounds=3000.0, n=160, calls=6)
    assert info["hybrid_substeps"] > 100 and info["hybrid_substeps"] < 900, info
    assert_same(got, exp, "two blobs meet")
    _, off, _ = run(sb, oracle, buf, bounds=3000.0, n=160, calls=6, ref_mode=OFF)
    assert (off.particles[:2 * P] != exp.particles[:2 * P]).any(), "the blobs must really collide"


def test_fast_blobs_stay_on_single_substeps(sb, oracle):
    """The same two blobs at +-18: the displacement bound uses the skin up every ten substeps, the first tracked run measures
    that, and the engine goes back to single substeps (a run costs a dozen launches of fixed work) -- same bits either way."""
    a = sb.scenes.lattice_buffers(24, 24, d=30.0, origin=(200.0, 400.0), jitter=0.5, layout=2, velocity=(18.0, 0.0))
    b = sb.scenes.lattice_buffers(24, 24, d=30.0, origin=(1100.0, 415.0), jitter=0.5, layout=2, velocity=(-18.0, 0.0), seed=7)
    P, B = a.particle_count, a.beam_count
    pv = np.concatenate([a.particles[:P], b.particles[:P]])
    beams = np.concatenate([a.beams[:B], b.beams[:B]])
    beams["a"][B:] += P
    beams["b"][B:] += P
    buf = sb.Buffers(2, 2 * P, 2 * B)
    buf.set_scene(pv, beams)
    got, exp, info = run(sb, oracle, buf, bounds=3000.0, n=160, calls=6)
    assert info["hybrid_substeps"] < 200, info
    assert_same(got, exp, "fast blobs")


FORCED = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import __graft_entry__ as ge
sb, orc = ge.load_package(), ge.load_oracle()
buf = sb.scenes.lattice_buffers(96, 64, d=30.0, origin=(300.0, 700.0), jitter=1.0, layout=2, velocity=(0.3, -0.8))
eng = sb.Engine(bounds_size=5000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=2)
ref = orc.OracleEngine(5000.0, 10.0, 64, 2, orc.COLLIDE_GRID, threads=16)
eng.write_buffers(buf); ref.write_buffers(buf)
for n in (64, 37, 64, 120):
    eng.step(n); ref.step(n)
got, exp = eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())
print("failed", eng.info("hybrid_failed"), "blocked", eng.info("hybrid_substeps"))
assert eng.info("hybrid_failed") >= 3 and eng.info("hybrid_substeps") > 50
assert np.array_equal(got.particles.view("u4"), exp.particles.view("u4")) and got.beams.tobytes() == exp.beams.tobytes()
print("forced roll-backs ok")
"""


def test_forced_roll_backs_are_invisible():
    """SB_HYBRID_FAIL_EVERY=3: every third tracked launch is declared over budget whatever it measured; its substeps are redone
    one by one from the intact READ buffers.  (The variable is read once per process: a process of its own.)"""
    env = dict(os.environ, SB_HYBRID_FAIL_EVERY="3")
    p = subprocess.run([sys.executable, "-c", FORCED % ROOT], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0 and "forced roll-backs ok" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]


PARKED = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import __graft_entry__ as ge
sb, orc = ge.load_package(), ge.load_oracle()
buf = sb.scenes.lattice_buffers(96, 64, d=30.0, origin=(300.0, 700.0), jitter=1.0, layout=2, velocity=(0.3, -0.8))
eng = sb.Engine(bounds_size=5000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=2)
ref = orc.OracleEngine(5000.0, 10.0, 64, 2, orc.COLLIDE_GRID, threads=16)
eng.write_buffers(buf); ref.write_buffers(buf)
for n in (64, 37, 64, 120):
    eng.step(n); ref.step(n)
eng.frame(); ref.frame()
got, exp = eng.load_buffers(buf.copy()), ref.load_buffers(buf.copy())
assert eng.info("hybrid_substeps") > 100
assert np.array_equal(got.particles.view("u4"), exp.particles.view("u4")) and got.beams.tobytes() == exp.beams.tobytes()
print("parked waits ok", eng.info("hybrid_substeps"))
"""


def test_waits_that_park_instead_of_polling():
    """Every look and every run of the hybrid path ends in a wait of the host for the stream, which polls for as long as the work in
    flight should take before it parks the thread (sb_api.hip sb_stream_wait).  SB_WAIT_SPIN_US=0 (read once per process: a process
    of its own) parks always -- the other branch of every such wait: same bits."""
    env = dict(os.environ, SB_WAIT_SPIN_US="0")
    p = subprocess.run([sys.executable, "-c", PARKED % ROOT], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0 and "parked waits ok" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
