"""The spatial hash's decision state (SbGridCtl, sb_physics.h) substep by substep on a scene: which substeps push, make lists,
how the bound and the skin move.  Usage: python tools/grid_ctl_dump.py [quiet|pile|soup] [substeps] [stride]   (GPU box)"""
import os, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
sb = ge.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "quiet"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 96
stride = int(sys.argv[3]) if len(sys.argv) > 3 else 1
WORDS = ["fresh", "need_build", "pushing", "abort", "cur", "executed", "builds", "since", "accum", "cx", "cy", "Cx", "Cy", "skin_min", "skin_max",
         "wide_next", "settled", "transient", "short_lived", "pad0", "pad1", "pad2", "skin", "cell", "reach2", "nx", "ny", "x0", "y0", "wide", "gen", "pskin"]
FLOATS = {"accum", "cx", "cy", "Cx", "Cy", "skin_min", "skin_max", "skin", "cell", "reach2", "x0", "y0", "pskin"}
os.environ["SB_HYBRID"] = "0"
if name == "pile":
    buf, bounds = sb.scenes.config3_buffers()
elif name == "soup":
    buf, bounds = sb.scenes.soup_buffers(1000, 1000, d=40.0, origin=(1000.0, 30.0), jitter=10.0, speed=60.0), 42000.0
else:
    buf, bounds = sb.scenes.lattice_buffers(1000, 1000, d=30.0, origin=(1000.0, 1000.0), jitter=1.0, layout=2), 32000.0
eng = sb.Engine(bounds_size=bounds, particle_radius=10.0, subticks=64, layout=2, max_particles=buf.max_particles, max_beams=buf.max_beams, collision_mode=2)
eng.write_buffers(buf)
show = ("fresh", "pushing", "abort", "cur", "executed", "builds", "since", "accum", "cx", "cy", "skin", "pskin", "settled", "transient", "short_lived")
for k in range(0, n, stride):
    eng.step(stride)
    w = {}
    for i, nm in enumerate(WORDS):
        v = eng.info("grid_ctl_%d" % i)
        w[nm] = struct.unpack("<f", struct.pack("<I", v))[0] if nm in FLOATS else v
    print("after %4d:" % (k + stride), " ".join("%s=%s" % (a, ("%.4g" % w[a]) if a in FLOATS else w[a]) for a in show),
          "aborts=%d classic=%d" % (eng.info("grid_aborts"), eng.info("grid_classic_substeps")), flush=True)
eng.destroy()
