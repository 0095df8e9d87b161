"""How local is the motion that wears a hash out?  BASELINE config 3's pile: positions read back K substeps apart, the displacement
of every particle against the common drift, the maximum per 320 x 320-unit block of the scene (about a tile's worth of a packed
body) against the maximum over the scene -- the number that decides when EVERY tile's lists are remade.  (GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
sb = ge.load_package()
K = int(os.environ.get("K", "8"))
buf, bounds = sb.scenes.config3_buffers()
os.environ["SB_HYBRID"] = "0"
eng = sb.Engine(bounds_size=bounds, particle_radius=10.0, subticks=64, layout=2, max_particles=buf.max_particles,
                max_beams=buf.max_beams, collision_mode=2)
eng.write_buffers(buf)
for _ in range(sb.scenes.CONFIG3_SETTLE_FRAMES):
    eng.frame()
eng.step(64)
out = sb.layout.Buffers(layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams)
for w in range(6):
    eng.step(97)
    eng.load_buffers(out)
    n = out.particle_count
    a = out.particles[:n, :2].astype(np.float64).copy()
    eng.step(K)
    eng.load_buffers(out)
    b = out.particles[:n, :2].astype(np.float64)
    d = b - a
    d -= d.mean(axis=0)
    disp = np.hypot(d[:, 0], d[:, 1])
    cell = (np.floor(a[:, 0] / 320.0).astype(np.int64) << 20) + np.floor(a[:, 1] / 320.0).astype(np.int64)
    order = np.argsort(cell, kind="stable")
    cs, ds = cell[order], disp[order]
    starts = np.flatnonzero(np.r_[True, cs[1:] != cs[:-1]])
    bmax = np.maximum.reduceat(ds, starts)
    g = disp.max()
    q = np.quantile(disp, [0.5, 0.9, 0.99, 0.999, 0.9999])
    print("window %d: K %d  max displacement %.3f  particle quantiles 50/90/99/99.9/99.99 %%: %s" % (w, K, g, " ".join("%.3f" % v for v in q)))
    print("          %d blocks; share of blocks whose own max is above 1/2, 1/4, 1/8 of the scene's: %.3f %.3f %.3f; median block max %.3f"
          % (len(bmax), (bmax > g / 2).mean(), (bmax > g / 4).mean(), (bmax > g / 8).mean(), np.median(bmax)), flush=True)
print("hash builds so far", eng.info("grid_builds"), "skin", eng.info("grid_skin_x1000") / 1000.0)
eng.destroy()
