"""Where a hash rebuild spends its time: phase stamps of block 0 of k_grid_maintain (10 ns ticks; block 0 is one of many, the launch ends with the slowest), on the config-3 pile and the soup."""
import sys
sys.path.insert(0, '.')
import __graft_entry__ as ge
sb = ge.load_package()
for name in ("pile", "soup"):
    if name == "pile":
        buf, bounds = sb.scenes.config3_buffers()
    else:
        buf, bounds = sb.scenes.soup_buffers(1000, 1000, d=40.0, origin=(1000.0, 30.0), jitter=10.0, speed=60.0), 42000.0
    eng = sb.Engine(bounds_size=bounds, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=2)
    eng.write_buffers(buf)
    for k in range(6):
        eng.step(40)
        st = [eng.info("grid_stamp_%d" % i) / 100.0 for i in range(2)]
        print(name, "cells", eng.info("grid_cells"), "builds", eng.info("grid_builds"),
              "stamps us: push + records %.1f | ticket %.1f  (total %.1f)" % (st[0], st[1] - st[0], st[1]), flush=True)
    eng.destroy()
