"""Which of bench.py's config-3 scenes really exercise the collision RESPONSE in the timed window?
Runs each with collisions off and with the spatial hash and reports how many particles the collision loop changed,
how fast things move, and whether everything stays finite."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import __graft_entry__ as ge
sb = ge.load_package()
scenes = {
    "--config3 (4000x250 lattice, spacing 22, on the floor)": (lambda: sb.scenes.lattice_buffers(4000, 250, d=22.0, origin=(1000.0, 10.0), jitter=1.0, layout=2), 4000 * 22.0 + 2000.0),
    "--soup (1000x1000 free particles, spacing 40, up to 60 units/s)": (lambda: sb.scenes.soup_buffers(1000, 1000, d=40.0, origin=(1000.0, 30.0), jitter=10.0, speed=60.0), 1000 * 40.0 + 2000.0),
}
for name, (make, bounds) in scenes.items():
    buf = make()
    out = {}
    for mode in (0, 2):
        eng = sb.Engine(bounds_size=float(bounds), layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=mode)
        eng.write_buffers(buf)
        for n in (64, 500, 500):
            eng.step(n)
            out.setdefault(mode, []).append(eng.load_buffers(buf.copy()).particles)
        if mode == 2:
            builds = eng.info("grid_builds")
        eng.destroy()
    print(name, flush=True)
    for k, n in enumerate((64, 564, 1064)):
        a, b = out[0][k], out[2][k]
        v = np.hypot(b[:, 2], b[:, 3]) / 64.0
        print("  after %4d substeps: changed by the collision loop %.4f of the particles, on the floor %d, |v|dt median %.3f max %.3f, finite %s"
              % (n, (a != b).any(axis=1).mean(), (b[:, 1] == 10.0).sum(), np.median(v), v.max(), np.isfinite(b).all()), flush=True)
    print("  hash builds in 1064 substeps: %d" % builds, flush=True)
