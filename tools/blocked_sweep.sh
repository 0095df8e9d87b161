#!/bin/bash
# Timing sweep of the blocked kernel on BASELINE config 2 (GPU box): launch depth (every call cut into launches of at most k).
#   gpurun -- 'tools/blocked_sweep.sh > gpurun_out/sweep.txt'      (DEPTHS="5 6 7" BENCH_ARGS="--width 2000 --height 2000" to vary)
cd "$(dirname "$0")/.."
one() { python bench.py --steps ${STEPS:-960} --warmup 64 --no-extra --no-cpu-baseline "$@" 2>>gpurun_out/sweep_stderr.log |
        python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.2f us per substep, %.1f us per launch of %d, tiles %d, %.3g p-steps/s' % (d['config']['kernel_us_per_substep'], r['avg_launch_us'], r['substeps_per_launch'], d['config']['tiles'], d['value']))"; }
for k in ${DEPTHS:-3 4 5 6 7}; do
  echo -n "depth=$k: "
  one --block-substeps $k $BENCH_ARGS
done
