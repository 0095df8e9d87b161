# Launch timeline of the soup scene (classic schedule: helper launch + substep kernel per substep): per-kernel durations, list-making
# launches by duration, building vs idle helper launches (GPU box).
ROOT=$PWD
OUT=$ROOT/gpurun_out/trace_soup
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 $ROOT/tools/grid_schedule_probe.py soup > $OUT/probe.log 2>&1
cd $ROOT
grep us/substep $OUT/probe.log
python3 - $OUT/tr <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted((r for r in csv.DictReader(open(f))), key=lambda r: int(r["Start_Timestamp"]))
sub = [i for i, r in enumerate(rows) if "k_substep_tiled_grid" in r["Kernel_Name"]][-960:]
first = sub[0]
rows = rows[first:]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
by = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:40]
    by.setdefault(k, []).append(dur(r))
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print("span %.1f us over %d substeps = %.2f per substep" % (span, 960, span / 960))
for k, d in by.items():
    s = sorted(d)
    print("%-42s n %4d  mean %7.2f  median %7.2f  p90 %7.2f  max %7.2f  total/substep %.2f" % (k, len(d), sum(d) / len(d), s[len(s) // 2], s[int(len(s) * .9)], s[-1], sum(d) / 960))
d = by[[k for k in by if "tiled_grid" in k][0]]
for lo, hi in ((0, 40), (40, 60), (60, 90), (90, 150), (150, 1e9)):
    x = [v for v in d if lo <= v < hi]
    if x: print("   substep launches of %3d - %3d us: %4d, mean %.2f" % (lo, min(hi, 999), len(x), sum(x) / len(x)))
# the helper: builds vs idle
h = by.get([k for k in by if "k_grid_build" in k][0] if any("k_grid_build" in k for k in by) else "", [])
x = [v for v in h if v > 10]; y = [v for v in h if v <= 10]
print("   helper launches: %d building (mean %.2f), %d idle (mean %.2f)" % (len(x), sum(x) / max(len(x), 1), len(y), sum(y) / max(len(y), 1)))
PY
rm -rf $OUT/tr
