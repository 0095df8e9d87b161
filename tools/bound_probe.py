"""How conservative is the hash's displacement bound?  After settling the config-3 pile, positions are read back every substep
and two bounds compared: the running SUM of per-substep maxima (what k_grid_maintain accumulates) against the maximum NET
displacement of any particle since the start (what a per-particle reference position would give).  Mean drift removed in both."""
import sys
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge
sb = ge.load_package()
which = sys.argv[1] if len(sys.argv) > 1 else "pile"
if which == "pile":
    buf, bounds = sb.scenes.config3_buffers()
else:
    buf, bounds = sb.scenes.soup_buffers(1000, 1000, d=40.0, origin=(1000.0, 30.0), jitter=10.0, speed=60.0), 42000.0
eng = sb.Engine(bounds_size=bounds, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=2)
eng.write_buffers(buf)
if which == "pile":
    for _ in range(sb.scenes.CONFIG3_SETTLE_FRAMES):
        eng.frame()
else:
    eng.step(64)
out = buf.copy()
P = buf.particle_count
p0 = eng.load_buffers(out).particles[:P, :2].astype("f8").copy()
prev, acc, C = p0.copy(), 0.0, np.zeros(2)
for s in range(1, 41):
    eng.step(1)
    p = eng.load_buffers(out).particles[:P, :2].astype("f8")
    d = p - prev
    c = d.mean(axis=0)
    acc += np.abs(d - c).max() * 1.4142137
    C += c
    net = np.abs(p - p0 - C).max() * 1.4142137
    prev = p.copy()
    if s % 4 == 0:
        print("%s substep %2d: sum of maxima %.3f   max net displacement %.3f   ratio %.2f" % (which, s, acc, net, acc / max(net, 1e-9)), flush=True)
