"""Config-3 scenes under the two hash schedules of SbGridCtl (sb_physics.h): time per substep, hashes built, aborts, helper
launches.  Usage: [SB_GRID_MODE=classic|lagged] python tools/grid_schedule_probe.py [pile] [soup] [floor] [quiet]  (GPU box; TILE=n: sb_options.tile_particles)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
sb = ge.load_package()
which = sys.argv[1:] or ["pile", "soup", "floor", "quiet"]
STEPS = int(os.environ.get("STEPS", "960"))
for name in which:
    if name == "pile":
        buf, bounds = sb.scenes.config3_buffers()
        settle = sb.scenes.CONFIG3_SETTLE_FRAMES
    elif name == "soup":
        buf = sb.scenes.soup_buffers(1000, 1000, d=40.0, origin=(1000.0, 30.0), jitter=10.0, speed=60.0)   # bench.py --soup
        bounds, settle = 42000.0, 0
    elif name == "floor":
        buf = sb.scenes.lattice_buffers(4000, 250, d=22.0, origin=(1000.0, 10.0), jitter=1.0, layout=2)
        bounds, settle = 4000 * 22.0 + 2000.0, 0
    else:
        buf = sb.scenes.lattice_buffers(1000, 1000, d=30.0, origin=(1000.0, 1000.0), jitter=1.0, layout=2)
        bounds, settle = 32000.0, 0
    os.environ["SB_HYBRID"] = "0"
    eng = sb.Engine(bounds_size=bounds, particle_radius=10.0, subticks=64, layout=2, max_particles=buf.max_particles,
                    max_beams=buf.max_beams, collision_mode=2, tile_particles=int(os.environ.get("TILE", "0")))
    eng.write_buffers(buf)
    for _ in range(settle):
        eng.frame()
    eng.step(64)
    eng.sync()
    keys = ("grid_builds", "grid_aborts", "grid_helper_launches", "grid_classic_substeps")
    before = {k: eng.info(k) for k in keys}
    ms = eng.step_timed(STEPS)
    eng.sync()
    after = {k: eng.info(k) - before[k] for k in keys}
    print("%-6s %s  %.2f us/substep  %s  skin %.1f  tiles %d" % (name, os.environ.get("SB_GRID_MODE", "auto"), ms * 1e3 / STEPS, after,
          eng.info("grid_skin_x1000") / 1000.0, eng.info("tiles")), flush=True)
    if os.environ.get("STAMPS"):   # a -DSB_STAMPS build: one mid-grid workgroup of the LAST launch, us since its start
        print("       stamps (us): begin-issued %.2f staged %.2f barrier1 %.2f decided %.2f barrier2 %.2f beams-done %.2f end %.2f"
              % tuple(eng.info("grid_stamp_%d" % k) / 100.0 for k in range(7)), flush=True)
        f = [eng.info("grid_stamp_%d" % (16 + k)) / 100.0 for k in range(12)]
        print("       last list-making launch (us): barrier1 %.2f decided %.2f beams-done %.2f coop-start %.2f rectangle %.2f records-in-LDS %.2f coop-done %.2f end %.2f"
              % (f[2], f[4], f[5], f[8], f[10], f[11], f[9], f[6]), flush=True)
    eng.destroy()
