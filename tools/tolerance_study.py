#!/usr/bin/env python3
"""The fp32 tolerance of this build against ANY conformant WebGPU run of the reference (north_star: "a stated fp32
tolerance after N steps").  The reference cannot run here, and the WGSL spec only bounds its
built-ins in ULP, so a live run may differ from the canonical arithmetic of DESIGN.md 2 in exactly these places:
    v1   normalize(v) = v / length(v)            two IEEE divisions        compute.wgsl:111,156,176
    v2   strain = (len - target) / length        one IEEE division         compute.wgsl:112
    v3   both
    v4   normalize(v) = v * inverseSqrt(dot(v,v))  the usual lowering       compute.wgsl:111,156,176
    fma  the compiler may contract a*b+c into one fma (-ffp-contract=fast -mfma)
Each is a build of the oracle (oracle/sb_oracle.c SBO_VARIANT; CPU, test infrastructure).  This script steps the
reference's default scene (main.ts:188-246, all-pairs collisions) and the BASELINE config-1 lattice (32 x 32, d = 25,
collisions off) with the canonical build and every variant and prints max |dp|, max |dv| after 64 and 1000 substeps.
The table is committed as profiles/r02_tolerance.txt and quoted in DESIGN.md 3; tests/test_tolerance_cpu.py holds
the numbers to the bound stated there."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

VARIANTS = ("v1", "v2", "v3", "v4", "fma")


def scenes(sb):
    lat = sb.scenes.lattice_buffers(32, 32, d=25.0, origin=(100.0, 100.0), spring=50.0, damp=700.0, yield_strain=0.2,
                                    strain_limit=0.5, jitter=2.0, layout=2)
    return {"default scene, all-pairs collisions (main.ts:188-246)": (sb.scenes.default_buffers(1, 256, 512), 1),
            "config-1 lattice 32x32, collisions off": (lat, 0)}


def run(orc, buf, mode, variant, checkpoints):
    e = orc.OracleEngine(1000.0, 10.0, 64, buf.layout, mode, threads=1, variant=variant)
    e.write_buffers(buf)
    out, done = {}, 0
    for n in checkpoints:
        e.step(n - done)
        done = n
        out[n] = e.load_buffers(buf.copy()).particles[:buf.particle_count].copy()
    return out


def study(checkpoints=(64, 1000)):
    sb = ge.load_package()
    orc = ge.load_oracle()
    orc.build()
    rows = []
    for name, (buf, mode) in scenes(sb).items():
        base = run(orc, buf, mode, None, checkpoints)
        for v in VARIANTS:
            if orc.variant_lib(v) is None:
                continue
            got = run(orc, buf, mode, v, checkpoints)
            for n in checkpoints:
                d = np.abs(got[n].astype("f8") - base[n].astype("f8"))
                rows.append(dict(scene=name, variant=v, substeps=n, dp=float(d[:, :2].max()), dv=float(d[:, 2:4].max()),
                                 differ=int((got[n].view("u4") != base[n].view("u4")).any(axis=1).sum()),
                                 particles=int(buf.particle_count)))
    return rows


if __name__ == "__main__":
    rows = study()
    print("%-58s %-4s %8s %12s %12s %s" % ("scene", "var", "substeps", "max |dp|", "max |dv|", "particles that differ"))
    for r in rows:
        print("%-58s %-4s %8d %12.3e %12.3e %d / %d" % (r["scene"], r["variant"], r["substeps"], r["dp"], r["dv"], r["differ"], r["particles"]))
