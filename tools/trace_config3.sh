mkdir -p gpurun_out/r02d
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r02d/trace_cfg3 -- python3 $ROOT/bench.py --config3 --steps 400 --warmup 16 --no-cpu-baseline > $ROOT/gpurun_out/r02d/trace_cfg3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r02d/trace_soup -- python3 $ROOT/bench.py --soup --steps 400 --warmup 16 --no-cpu-baseline > $ROOT/gpurun_out/r02d/trace_soup.log 2>&1
cd $ROOT
for d in trace_cfg3 trace_soup; do echo == $d; tail -1 gpurun_out/r02d/$d.log | cut -c1-200; python3 - gpurun_out/r02d/$d <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("%-70s calls %6s avg us %9.2f total ms %8.2f  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
# rebuild substeps against the others: the maintain launches split at 10 us, and the particle launch that follows each
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    acc = {}
    for a, b in zip(rows, rows[1:]):
        if "k_grid_maintain" in a["Kernel_Name"] and "k_substep" in b["Kernel_Name"]:
            key = "rebuild" if dur(a) > 10.0 else "keep"
            acc.setdefault(key, []).append((dur(a), dur(b), (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3))
    for key, v in acc.items():
        n = len(v)
        print("  %-8s substeps %5d: maintain %.2f us, gap %.2f us, particle kernel %.2f us" % (key, n, sum(x[0] for x in v) / n, sum(x[2] for x in v) / n, sum(x[1] for x in v) / n))
PY
done
