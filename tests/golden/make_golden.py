#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory.  Run from the repo root:
    python tests/golden/make_golden.py

The reference has no fixtures of its own, cannot run here (WGSL through a
browser's WebGPU only) and ships no recorded outputs, so these vectors are produced by THIS repo:
  default_scene_v1.snapshot / _v2.snapshot
      the web app's initial scene (src/main.ts:188-246) serialised in the snapshot format of
      src/engineMapping.ts:377-401, by softbody-webgpu_amd/layout.py (a by-hand restatement of the
      byte layout, cross-checked against the independent JavaScript restatement in host/).
  default_scene_v1_after_2_frames.snapshot
      the same scene after 2 frames (128 substeps, all-pairs collisions) stepped by the CPU oracle
      (oracle/sb_oracle.c, S0 semantics).  Parity against the real WebGPU engine is "unpinned".
  lattice_8x6_after_100_substeps.npz
      a small jittered lattice (v2 layout, collisions off) before/after 100 oracle substeps.
  lattice_32x32_after_1000_substeps.json
      BASELINE config 1 (32 x 32 lattice of main.ts:220's material at (100, 100), spacing 25, collisions off,
      v2 layout): SHA-256 of the v2 snapshot after 1000 oracle substeps, its length, and the first and last particle
      records, for host/test/gpu.test.js (the whole Node path must reproduce the snapshot byte for byte).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sb = ge.load_package()
    orc = ge.load_oracle()
    orc.build()
    for lay in (1, 2):
        buf = sb.scenes.default_buffers(lay, 65536 if lay == 1 else 256, 65536 if lay == 1 else 512)
        open(os.path.join(HERE, "default_scene_v%d.snapshot" % lay), "wb").write(buf.create_snapshot())
    buf = sb.scenes.default_buffers(1)
    ref = orc.OracleEngine(1000.0, 10.0, 64, 1, orc.COLLIDE_ALLPAIRS)
    ref.write_buffers(buf)
    ref.frame()
    ref.frame()
    out = ref.load_buffers(buf.copy())
    open(os.path.join(HERE, "default_scene_v1_after_2_frames.snapshot"), "wb").write(out.create_snapshot())
    lat = sb.scenes.lattice_buffers(8, 6, d=25.0, origin=(100.0, 100.0), jitter=2.0, layout=2, strain_limit=0.5)
    ref = orc.OracleEngine(1000.0, 10.0, 64, 2, orc.COLLIDE_OFF)
    ref.write_buffers(lat)
    ref.step(100)
    out = ref.load_buffers(lat.copy())
    np.savez_compressed(os.path.join(HERE, "lattice_8x6_after_100_substeps.npz"),
                        particles_in=lat.particles, beams_in=lat.beams.view("u1"),
                        particles_out=out.particles, beams_out=out.beams.view("u1"))
    import hashlib
    import json
    lat = sb.scenes.lattice_buffers(32, 32, d=25.0, origin=(100.0, 100.0), spring=50.0, damp=700.0, yield_strain=0.2,
                                    strain_limit=0.5, layout=2)
    ref = orc.OracleEngine(1000.0, 10.0, 64, 2, orc.COLLIDE_OFF)
    ref.write_buffers(lat)
    ref.step(1000)
    out = ref.load_buffers(lat.copy())
    snap = out.create_snapshot()
    json.dump({"sha256": hashlib.sha256(snap).hexdigest(), "bytes": len(snap), "particles": out.particle_count,
               "beams": out.beam_count, "first_particle": [float(x) for x in out.particles[0]],
               "last_particle": [float(x) for x in out.particles[out.particle_count - 1]]},
              open(os.path.join(HERE, "lattice_32x32_after_1000_substeps.json"), "w"), indent=1)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
