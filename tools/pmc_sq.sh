#!/bin/bash
# SQ counter passes over the default bench (two rocprofv3 --pmc runs with --kernel-trace only), summarised per kernel.
# Usage: tools/pmc_sq.sh <outdir> [bench args]
OUT=$PWD/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 200 --warmup 16 --no-cpu-baseline --no-extra $*"
ROOT=$PWD
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_sq1 -- $BENCH > $OUT/pmc_sq1.log 2>&1 || echo "pmc sq1 failed"
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1 || echo "pmc sq2 failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || echo "trace failed"
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
d = collections.defaultdict(list)
for f in glob.glob(out + "/pmc_sq*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        d[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(d):
    if "substep" in k[0]:
        print("%-40s %-24s launches %4d  per launch %14.0f" % (k[0][:40], k[1], len(d[k]), sum(d[k]) / len(d[k])))
for f in glob.glob(out + "/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:60], r["Calls"], "avg us", float(r["AverageNs"]) / 1e3)
PY
