"""Looking for a config-3 scene (1 M particles, floor contact + self-collision ACTIVE) that stays a softbody
instead of bursting.  Steps in chunks, logs each chunk (time, speeds, hash builds), stops when a chunk gets slow."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import __graft_entry__ as ge
sb = ge.load_package()
log = open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "drop_probe.log"), "a")
def say(s):
    print(s, flush=True); log.write(s + "\n"); log.flush()
cases = [tuple(float(x) for x in c.split(",")) for c in sys.argv[1:]] or [(4000, 250, 22.0, 10.0, 0.0)]
for W, H, d, y0, vy in cases:
    W, H = int(W), int(H)
    buf = sb.scenes.lattice_buffers(W, H, d=d, origin=(1000.0, y0), jitter=0.5, layout=2, velocity=(0.0, vy))
    eng = sb.Engine(bounds_size=max(W * d + 2000.0, 32000.0), layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=2)
    eng.write_buffers(buf)
    say("case %dx%d d=%.1f y0=%.0f vy=%.0f" % (W, H, d, y0, vy))
    done = 0
    for chunk in range(40):
        t0 = time.perf_counter()
        ms = eng.step_timed(25)
        done += 25
        out = eng.load_buffers(buf.copy())
        p = out.particles
        v = np.hypot(p[:, 2], p[:, 3]) / 64.0
        say("  after %4d: %.1f us/substep, builds %d, |v|dt median %.3f p99.9 %.3f max %.2f, on floor %d, finite %s"
            % (done, ms * 1e3 / 25, eng.info("grid_builds"), np.median(v), np.percentile(v, 99.9), v.max(), (p[:, 1] == 10.0).sum(), np.isfinite(p).all()))
        if ms > 250.0:
            say("  chunk got slow: stopping this case")
            break
    eng.destroy()
