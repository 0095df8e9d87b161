#!/bin/bash
# Counter passes over a collision bench (rocprofv3 --pmc with --kernel-trace only), k_substep_tiled_grid split into the launches
# that make neighbour lists (the substep after a hash build) and the others by GRBM_GUI_ACTIVE of the same dispatch.
# Usage: tools/pmc_fresh.sh <outdir> [bench args, e.g. --soup]
OUT=$PWD/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 200 --warmup 16 --no-cpu-baseline --no-extra $*"
ROOT=$PWD
cd /tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
n=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  # (a fifth pass of five TA_* counters was refused by the hardware -- "exceeds the capabilities" -- and rocprofv3 then hung
  # until the box's silence guard killed the call: keep passes small and names taken from $OUT/avail.txt)
  n=$((n+1))
  rocprofv3 --pmc GRBM_GUI_ACTIVE $set --kernel-trace --output-format csv -d $OUT/pmc$n -- $BENCH > $OUT/pmc$n.log 2>&1 || echo "pmc pass $n failed: $set"
done
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pmc*/*/*counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "k_substep_tiled_grid" in r["Kernel_Name"]]
    by = collections.defaultdict(dict)
    for r in rows:
        by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    act = sorted(v.get("GRBM_GUI_ACTIVE", 0.0) for v in by.values())
    if not act:
        continue
    med = act[len(act) // 2]
    groups = {"keep": [v for v in by.values() if v.get("GRBM_GUI_ACTIVE", 0.0) <= 2.0 * med], "fresh": [v for v in by.values() if v.get("GRBM_GUI_ACTIVE", 0.0) > 2.0 * med]}
    names = sorted({k for v in by.values() for k in v})
    for name in names:
        print("%-40s" % name, "  ".join("%s (%4d launches) %14.0f" % (g, len(vs), sum(v.get(name, 0.0) for v in vs) / max(len(vs), 1)) for g, vs in groups.items()))
PY
