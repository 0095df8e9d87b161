"""Does bench.py's config-3 scene really exercise the collision RESPONSE in the timed window?
From the settled pile (and from later states of the timed window) one substep is run with the spatial hash and one
with collisions off; the share of particles whose new state differs is the share the collision loop of
compute.wgsl:142-170 changed in that substep.  Also: how fast things move, whether the pile is still a pile."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import __graft_entry__ as ge
sb = ge.load_package()
scenes = {
    "config 3 (blob pile of scenes.config3_buffers, settled 48 frames)": (lambda: sb.scenes.config3_buffers(), sb.scenes.CONFIG3_SETTLE_FRAMES),
    "--soup (1000x1000 free particles, spacing 40, up to 60 units/s)": (lambda: (sb.scenes.soup_buffers(1000, 1000, d=40.0, origin=(1000.0, 30.0), jitter=10.0, speed=60.0), 1000 * 40.0 + 2000.0), 0),
}
for name, (make, settle) in scenes.items():
    buf, bounds = make()
    P = buf.particle_count
    eng = sb.Engine(bounds_size=float(bounds), layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=2)
    eng.write_buffers(buf)
    for _ in range(settle):
        eng.frame()
    print(name, flush=True)
    done = 0
    builds0 = eng.info("grid_builds")
    for n in (64, 564, 1064):
        eng.step(n - done - 1)
        done = n
        before = eng.load_buffers(buf.copy())
        eng.step(1)
        on = eng.load_buffers(buf.copy()).particles[:P]
        off = sb.Engine(bounds_size=float(bounds), layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=0)
        off.write_buffers(before)
        off.step(1)
        b = off.load_buffers(buf.copy()).particles[:P]
        off.destroy()
        v = np.hypot(on[:, 2], on[:, 3]) / 64.0
        print("  substep %4d of the window: changed by the collision loop %.4f of the particles, on the floor %d, |v|dt median %.4f max %.3f, "
              "top of the pile %.0f, beams %d, finite %s"
              % (n, (on.view('u4') != b.view('u4')).any(axis=1).mean(), (on[:, 1] == 10.0).sum(), np.median(v), v.max(), on[:, 1].max(),
                 before.beam_count, np.isfinite(on).all()), flush=True)
    print("  hash builds in the 1064 substeps: %d" % (eng.info("grid_builds") - builds0), flush=True)
    eng.destroy()
