"""Register / scratch / occupancy table of every kernel in a HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python tools/kernel_resources.py softbody-webgpu_amd/csrc/sb_kernels.hip [filter]"""
import re, subprocess, sys, tempfile, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
flags = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize".split()
with tempfile.TemporaryDirectory() as d:
    p = subprocess.run(["/opt/rocm/bin/hipcc", *flags, *os.environ.get("EXTRA", "").split(), "-c", src, "-o", os.path.join(d, "k.o"),
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
rows, cur = [], None
for line in p.stderr.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    if flt and flt not in name: continue
    print("%-48s VGPR %3s  spill %3s  scratch %4s  SGPR %3s  occ %s  LDS %s" % (name[:48], r.get("VGPRs"), r.get("VGPRs Spill", r.get("VGPR Spill")),
          r.get("ScratchSize [bytes/lane]"), r.get("TotalSGPRs"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
