#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, the default bench, rocprofv3 kernel trace + PMC passes.
# Usage: tools/gpu_profile.sh <round-tag>
set -o pipefail
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo bench failed; tail -5 $OUT/bench.err; exit 1; }
cat $OUT/bench.json | cut -c1-300
BENCH="python3 $PWD/bench.py --steps 200 --warmup 16 --no-cpu-baseline"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || { echo pmc fetch failed; tail -5 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 || { echo pmc write failed; tail -5 $OUT/pmc_write.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib_fetch -- $OLDPWD/tools/hbm_calib > $OUT/calib_fetch.log 2>&1 || { echo calib fetch failed; tail -5 $OUT/calib_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/calib_write -- $OLDPWD/tools/hbm_calib > $OUT/calib_write.log 2>&1 || { echo calib write failed; tail -5 $OUT/calib_write.log; exit 1; }
# where the dominant kernel's time goes: SQ counters, two passes of eight (own runs: --pmc with --kernel-trace only)
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_sq1 -- $BENCH > $OUT/pmc_sq1.log 2>&1 || echo "pmc sq1 failed"
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1 || echo "pmc sq2 failed"
# config 3 (grid collisions): bench line + kernel trace
python3 $OLDPWD/bench.py --collisions grid --no-cpu-baseline > $OUT/bench_grid.json 2> $OUT/bench_grid.err || { echo grid bench failed; tail -5 $OUT/bench_grid.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_grid -- python3 $OLDPWD/bench.py --collisions grid --steps 200 --warmup 16 --no-cpu-baseline > $OUT/trace_grid.log 2>&1 || { echo grid trace failed; exit 1; }
# config 3 with its contacts active (4000x250 lattice at spacing 22 resting on the floor)
python3 $OLDPWD/bench.py --lattice-on-floor --no-cpu-baseline --no-extra > $OUT/bench_config3_contacts.json 2>/dev/null || echo "config3 contacts bench failed"
python3 $OLDPWD/bench.py --soup --no-cpu-baseline > $OUT/bench_soup.json 2>/dev/null || echo "soup bench failed"
python3 $OLDPWD/tools/config3_contacts_check.py 2>/dev/null | grep -v "amdgpu.ids" > $OUT/config3_contacts_check.txt || echo "config3 check failed"
# one GPU's share of BASELINE configs 4 (500 x 4000 columns x rows) and 5 (1000 x 8000, mixed springs, dt = 1/128)
python3 $OLDPWD/bench.py --no-cpu-baseline --width 500 --height 4000 --steps 500 > $OUT/cfg4_share.json 2>/dev/null || echo "cfg4 share failed"
python3 $OLDPWD/bench.py --no-cpu-baseline --width 1000 --height 8000 --mixed-stiffness --subticks 128 --steps 200 --warmup 32 > $OUT/cfg5_share.json 2>/dev/null || echo "cfg5 share failed"
# multi-GPU pieces that one card can show: cost of one ghost refresh (RCCL vs direct peer stores, loopback)
# and the whole N=2 bench code path with both ranks on this card (rehearsal, not a measurement)
python3 $OLDPWD/tools/exchange_cost.py 2>/dev/null | grep "^depth" > $OUT/exchange_cost.txt || echo "exchange cost probe failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_exchange -- python3 $OLDPWD/tools/exchange_cost.py > $OUT/trace_exchange.log 2>&1 || echo "exchange trace failed"
(cd $OLDPWD && python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --rehearse-one-gpu --steps 480 --warmup 48 2>/dev/null | grep "^{" > $OUT/rehearse_2ranks.json) || echo "rehearsal failed"
(cd $OLDPWD && node softbody-webgpu_amd/host/bench.js > $OUT/node_bench.json 2> $OUT/node_bench.err) || echo "node bench failed"
find $OUT -name "*.csv" | wc -l
du -sh $OUT
