"""Worker of tests/test_gpu_hardening.py: runs with SB_MAINTAIN_BLOCKS above what the card can hold at once, so the
device-wide barrier of k_grid_maintain cannot complete; its bounded wait must give up, sb_sync must report it, and
the engine must work again after the next upload."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

sb = ge.load_package()
orc = ge.load_oracle()
assert int(os.environ["SB_MAINTAIN_BLOCKS"]) >= 1024
big = sb.scenes.soup_buffers(800, 800, d=40.0, origin=(1000.0, 30.0), jitter=10.0, speed=30.0)   # 640 000 particles: 625 workgroups wanted
eng = sb.Engine(bounds_size=40000.0, layout=2, max_particles=big.max_particles, max_beams=big.max_beams, collision_mode=2)
eng.write_buffers(big)
eng.step(1)            # the first substep always builds the hash
try:
    eng.sync()
    print("NO_ERROR_REPORTED")
    sys.exit(1)
except sb.engine.EngineError as exc:
    assert "barrier timed out" in str(exc), exc
    print("REPORTED:", exc)
# recovery: a scene small enough for the barrier (a few workgroups), same engine, same process
small = sb.scenes.soup_buffers(60, 50, d=40.0, origin=(1000.0, 30.0), jitter=10.0, speed=30.0)
small2 = sb.Buffers(2, big.max_particles, big.max_beams)
small2.set_scene(small.particles[:small.particle_count], small.beams[:0])
eng.write_buffers(small2)
ref = orc.OracleEngine(40000.0, 10.0, 64, 2, orc.COLLIDE_GRID, threads=4)
ref.write_buffers(small2)
eng.step(64)
ref.step(64)
eng.sync()
got, exp = eng.load_buffers(small2.copy()), ref.load_buffers(small2.copy())
P = small.particle_count
assert np.array_equal(got.particles[:P].view("u4"), exp.particles[:P].view("u4"))
eng.destroy()
print("RECOVERED_OK")
