# Launch timeline of the default collision mode on config 2's scene (blocked launches + validations under the hash): rocprofv3 kernel
# trace of bench.py --collisions grid, per-kernel totals and the gaps between launches (GPU box).
ROOT=$PWD
OUT=$ROOT/gpurun_out/trace_hybrid
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 $ROOT/bench.py --collisions grid --steps 960 --warmup 64 --no-extra --no-cpu-baseline > $OUT/bench.log 2>&1
cd $ROOT
cut -c1-200 $OUT/bench.log | tail -2
python3 - $OUT/tr <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted((r for r in csv.DictReader(open(f))), key=lambda r: int(r["Start_Timestamp"]))
# the timed region: the last 960 substeps -> take the last 170 blocked launches and everything between
idx = [i for i, r in enumerate(rows) if "k_substep_blocked" in r["Kernel_Name"]]
first = idx[-160]
rows = rows[first:]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
by = {}
for r in rows:
    by.setdefault(r["Kernel_Name"].split("(")[0][:50], []).append(dur(r))
busy = sum(sum(v) for v in by.values())
print("span %.1f us, kernels busy %.1f us (%.1f %%), %d launches" % (span, busy, 100 * busy / span, len(rows)))
for k, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print("%-52s n %4d  mean %7.2f  total %9.1f  (%.1f %% of span)" % (k, len(d), sum(d) / len(d), sum(d), 100 * sum(d) / span))
gaps = [(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3 for i in range(len(rows) - 1)]
g = sorted(gaps)
print("gaps between launches: mean %.2f median %.2f p90 %.2f max %.2f total %.1f us" % (sum(gaps) / len(gaps), g[len(g) // 2], g[int(len(g) * .9)], g[-1], sum(gaps)))
big = [(x, rows[i]["Kernel_Name"][:30], rows[i + 1]["Kernel_Name"][:30]) for i, x in enumerate(gaps) if x > 20]
print("gaps above 20 us:", len(big), big[:6])
PY
rm -rf $OUT/tr
