#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, the default bench, rocprofv3 kernel trace + PMC passes.
# Usage: tools/gpu_profile.sh <round-tag> [tests|prof|bench|all]      (then: python tools/summarize_profiles.py <round-tag>)
# (the three parts fit one 20-minute gpurun call each; "all" is for a box without that limit)
# Every rocprofv3 run has the program itself after `--`; --pmc passes carry --kernel-trace only.
set -o pipefail
TAG=${1:-r04}
PART=${2:-all}
OUT=$PWD/gpurun_out/$TAG
ROOT=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
if [ $PART = tests ] || [ $PART = all ]; then
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
fi
if [ $PART = prof ] || [ $PART = all ]; then
python bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo bench failed; tail -5 $OUT/bench.err; exit 1; }
cut -c1-300 $OUT/bench.json
BENCH="python3 $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra"
CFG3="python3 $ROOT/bench.py --config3 --steps 200 --warmup 20 --no-cpu-baseline --no-extra"
cd /tmp
prof() { # outdir, then rocprofv3 arguments
  d=$1; shift
  rocprofv3 "$@" > $OUT/$d.log 2>&1 || { echo "$d failed"; tail -3 $OUT/$d.log; }
}
prof trace --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH
prof pmc_fetch --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH
prof pmc_write --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH
prof calib_fetch --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib_fetch -- $ROOT/tools/hbm_calib
prof calib_write --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/calib_write -- $ROOT/tools/hbm_calib
# where the dominant kernel's time goes: SQ counters, two passes of eight
prof pmc_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_sq1 -- $BENCH
prof pmc_sq2 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $BENCH
# the same workload with one substep per launch (the HBM-bound kernel of DESIGN.md 4.2): traffic
K1="python3 $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra --block-substeps 1"
prof pmc_fetch_k1 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_k1 -- $K1
prof pmc_write_k1 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_k1 -- $K1
# config 3 (the settled blob pile, spatial-hash collisions): kernel trace and HBM traffic of its two kernels
prof trace_cfg3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg3 -- $CFG3
prof pmc_fetch_cfg3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_cfg3 -- $CFG3
prof pmc_write_cfg3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_cfg3 -- $CFG3
cd $ROOT
fi
if [ $PART = bench ] || [ $PART = all ]; then
python3 bench.py --config3 --no-cpu-baseline > $OUT/bench_config3.json 2>/dev/null || echo "config3 bench failed"
python3 bench.py --block-substeps 1 --no-cpu-baseline --no-extra > $OUT/bench_single_substep.json 2>/dev/null || echo "k=1 bench failed"
python3 bench.py --collisions grid --no-cpu-baseline --no-extra > $OUT/bench_grid.json 2>/dev/null || echo "grid bench failed"
SB_HYBRID=0 python3 bench.py --collisions grid --no-cpu-baseline --no-extra > $OUT/bench_grid_no_hybrid.json 2>/dev/null || echo "grid bench (no hybrid) failed"
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_protocol.json 2>/dev/null || echo "driver-protocol bench failed"
python3 bench.py --rest-lengths current --no-cpu-baseline --no-extra > $OUT/bench_rest_lengths.json 2>/dev/null || echo "mode-1 bench failed"
python3 bench.py --lattice-on-floor --no-cpu-baseline --no-extra > $OUT/bench_config3_contacts.json 2>/dev/null || echo "lattice-on-floor bench failed"
python3 bench.py --soup --no-cpu-baseline > $OUT/bench_soup.json 2>/dev/null || echo "soup bench failed"
python3 tools/config3_contacts_check.py 2>/dev/null | grep -v "amdgpu.ids" > $OUT/config3_contacts_check.txt || echo "config3 check failed"
# one GPU's share of BASELINE configs 4 (500 x 4000 columns x rows) and 5 (1000 x 8000, mixed springs, dt = 1/128), and 4 M / 16 M particles
python3 bench.py --no-cpu-baseline --no-extra --width 500 --height 4000 --steps 500 > $OUT/cfg4_share.json 2>/dev/null || echo "cfg4 share failed"
python3 bench.py --no-cpu-baseline --no-extra --width 1000 --height 8000 --mixed-stiffness --subticks 128 --steps 200 --warmup 40 > $OUT/cfg5_share.json 2>/dev/null || echo "cfg5 share failed"
python3 bench.py --no-cpu-baseline --no-extra --width 2000 --height 2000 --steps 400 > $OUT/bench_4M.json 2>/dev/null || echo "4M failed"
python3 bench.py --no-cpu-baseline --no-extra --width 4000 --height 4000 --steps 200 --warmup 20 > $OUT/bench_16M.json 2>/dev/null || echo "16M failed"
# multi-GPU pieces that one card can show: cost of one ghost refresh (RCCL vs direct peer stores, loopback)
# and the whole N=2 bench code path with both ranks on this card, started bare (rehearsal, not a measurement)
python3 tools/exchange_cost.py 2>/dev/null | grep "^depth" > $OUT/exchange_cost.txt || echo "exchange cost probe failed"
python3 bench.py --gpus 2 --rehearse-one-gpu --steps 480 --warmup 48 2>/dev/null > $OUT/rehearse_2ranks.json || echo "rehearsal failed"
python3 bench.py --gpus 2 --rehearse-one-gpu --steps 20 --warmup 5 2>/dev/null > $OUT/rehearse_2ranks_driver_protocol.json || echo "rehearsal (driver protocol) failed"
# ... the same two ranks with the engine's default collision mode (hash on: blocked launches between the ghost refreshes since r04)
python3 bench.py --gpus 2 --rehearse-one-gpu --collisions grid --steps 480 --warmup 48 --no-cpu-baseline --no-extra 2>/dev/null > $OUT/rehearse_2ranks_grid.json || echo "rehearsal (grid) failed"
# the spatial hash's two schedules on the config-3 scenes (lagged by default, classic forced)
python3 tools/grid_schedule_probe.py 2>/dev/null | grep us/substep > $OUT/grid_schedule_probe.txt || echo "schedule probe failed"
SB_GRID_MODE=classic python3 tools/grid_schedule_probe.py 2>/dev/null | grep us/substep >> $OUT/grid_schedule_probe.txt || echo "schedule probe (classic) failed"
node softbody-webgpu_amd/host/bench.js > $OUT/node_bench.json 2> $OUT/node_bench.err || echo "node bench failed"
SB_UPLOAD_TIMING=1 python3 tools/upload_timing.py 2>&1 | grep -v "amdgpu.ids" > $OUT/upload_timing.txt || echo "upload timing failed"
fi
find $OUT -name "*.csv" | wc -l
du -sh $OUT
