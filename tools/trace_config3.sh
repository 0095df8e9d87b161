mkdir -p gpurun_out/r02c
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r02c/trace_cfg3 -- python3 $ROOT/bench.py --config3 --steps 400 --warmup 16 --no-cpu-baseline > $ROOT/gpurun_out/r02c/trace_cfg3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r02c/trace_soup -- python3 $ROOT/bench.py --soup --steps 400 --warmup 16 --no-cpu-baseline > $ROOT/gpurun_out/r02c/trace_soup.log 2>&1
cd $ROOT
for d in trace_cfg3 trace_soup; do echo == $d; tail -1 gpurun_out/r02c/$d.log | cut -c1-200; python3 - gpurun_out/r02c/$d <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("%-70s calls %6s avg us %9.2f total ms %8.2f  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
done
