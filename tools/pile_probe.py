#!/usr/bin/env python3
"""Stability probe for BASELINE config 3 scenes, on the CPU oracle (test infrastructure, not the product):
a pile of lattice blobs resting on the floor and on each other.  Prints per frame the fastest particle, the
share of particles the collision loop changed in that frame's last substep and the top of the pile, so that a
scene generator can be chosen that is still a pile after thousands of substeps (the collision
response of compute.wgsl:164-168 pumps energy into crushed rows; deep beds burst)."""
import argparse
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--columns", type=int, default=12, help="blobs per layer")
ap.add_argument("--layers", type=int, default=4)
ap.add_argument("--bw", type=int, default=9, help="blob width in particles")
ap.add_argument("--bh", type=int, default=4)
ap.add_argument("--gap", type=float, default=21.0, help="clearance between neighbouring blobs at upload")
ap.add_argument("--frames", type=int, default=30)
ap.add_argument("--bounds", type=float, default=0.0)
a = ap.parse_args()
sb = ge.load_package()
orc = ge.load_oracle()
orc.build()
buf, bounds = sb.scenes.blob_pile_buffers(a.columns, a.layers, bw=a.bw, bh=a.bh, gap=a.gap, bounds=a.bounds or None)
P = buf.particle_count
print("pile: %d blobs, %d particles, %d beams, bounds %g" % (a.columns * a.layers, P, buf.beam_count, bounds), flush=True)
on = orc.OracleEngine(bounds, 10.0, 64, 2, orc.COLLIDE_GRID, threads=8)
on.write_buffers(buf)
for f in range(a.frames):
    on.step(63)
    before = on.load_buffers(buf.copy())
    off = orc.OracleEngine(bounds, 10.0, 64, 2, orc.COLLIDE_OFF, threads=8)
    off.write_buffers(before)
    off.step(1)
    on.step(1)
    on.delete_pass() if hasattr(on, "delete_pass") else None
    x, y = on.load_buffers(buf.copy()), off.load_buffers(buf.copy())
    changed = (x.particles[:P].view("u4") != y.particles[:P].view("u4")).any(axis=1).mean()
    v = np.hypot(x.particles[:P, 2], x.particles[:P, 3])
    print("frame %3d  max|v| %8.3f  mean|v| %7.4f  changed by collisions %.3f  top %7.1f  beams %d"
          % (f, v.max(), v.mean(), changed, x.particles[:P, 1].max(), x.beam_count), flush=True)
