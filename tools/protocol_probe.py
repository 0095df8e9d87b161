"""Where the wall time of the driver's 20-substep protocol goes: HIP-event time against host time of sb_step_timed, and of step + sync
without events (GPU box; profiles/r04_grid_schedules.txt, last section)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
sb = ge.load_package()
buf = sb.scenes.lattice_buffers(1000, 1000, d=30.0, origin=(1000.0, 1000.0), jitter=1.0, layout=2)
eng = sb.Engine(bounds_size=32000.0, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=0)
eng.write_buffers(buf)
eng.step(5); eng.sync()
for rep in range(6):
    t0 = time.perf_counter(); ms = eng.step_timed(20); t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
    print("step_timed(20): events %.1f us, wall %.1f us, + sync %.1f us" % (ms * 1e3, (t1 - t0) * 1e6, (t2 - t1) * 1e6))
    time.sleep(0.05)
for rep in range(4):
    t0 = time.perf_counter(); eng.step(20); t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
    print("step(20) returns after %.1f us, sync done at %.1f us" % ((t1 - t0) * 1e6, (t2 - t0) * 1e6))
    time.sleep(0.05)
for rep in range(3):
    t0 = time.perf_counter(); ms = eng.step_timed(21); t1 = time.perf_counter()
    print("step_timed(21): events %.1f us, wall %.1f us" % (ms * 1e3, (t1 - t0) * 1e6))
