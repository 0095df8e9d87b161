#!/bin/bash
# Per-launch durations of config 3's substep kernel from a rocprofv3 kernel trace (GPU box): since r04 the substep kernel is the
# only launch of a substep, and its launches come in three kinds -- ordinary, push (the workgroups push the next hash as they go)
# and list-making -- which the durations tell apart.  Usage: tools/trace_config3.sh [pile|soup|floor|quiet]   (env: SB_GRID_COOP=0,
# SB_GRID_MODE=classic for A/B runs); prints mean / median / percentiles and the split at 45 us.
SCENE=${1:-pile}
ROOT=$PWD
OUT=$ROOT/gpurun_out/trace_config3
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 $ROOT/tools/grid_schedule_probe.py $SCENE > $OUT/probe.log 2>&1
cd $ROOT
grep us/substep $OUT/probe.log
python3 - $OUT/tr <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted((r for r in csv.DictReader(open(f)) if "k_substep_tiled_grid" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows][-960:]   # the timed region of the probe
s, n = sorted(d), len(d)
print("launches %d: mean %.2f median %.2f p80 %.2f p90 %.2f p95 %.2f max %.2f us" % (n, sum(d) / n, s[n // 2], s[int(n * .8)], s[int(n * .9)], s[int(n * .95)], s[-1]))
slow, fast = [x for x in d if x > 45], [x for x in d if x <= 45]
print("   <= 45 us: %d launches, mean %.2f;  > 45 us (list-making): %d launches, mean %.2f" % (len(fast), sum(fast) / max(len(fast), 1), len(slow), sum(slow) / max(len(slow), 1)))
PY
rm -rf $OUT/tr
