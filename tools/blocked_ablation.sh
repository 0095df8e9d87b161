#!/bin/bash
# Where does a launch of k_substep_blocked spend its time?  Builds diagnostic variants of the library (SB_BK_ABLATE bits, see
# sb_blocked.hip; never shipped) into scratch/ -- run this part HERE, hipcc cross-compiles -- and, with "run", times each of
# them on BASELINE config 2 on the GPU box (the .so files travel with the snapshot).
#   tools/blocked_ablation.sh build            (in the container)
#   gpurun -- 'tools/blocked_ablation.sh run > gpurun_out/ablation.txt'
set -e
cd "$(dirname "$0")/.."
CS=softbody-webgpu_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -Wno-unused-function"
VARIANTS="${VARIANTS:-0 16 1 2 4 32 3}"
if [ "$1" = build ]; then
    mkdir -p scratch
    for v in $VARIANTS; do
        ( cd $CS && /opt/rocm/bin/hipcc $FLAGS -DSB_BK_ABLATE=$v $EXTRA -shared -o ../../scratch/libsb_ablate_$v.so sb_api.hip sb_kernels.hip sb_blocked.hip sb_partition.cpp ) &
    done
    wait
    ls -la scratch/
else
    for v in $VARIANTS; do
        echo -n "SB_BK_ABLATE=$v  "
        SOFTBODY_HIP_LIB=$PWD/scratch/libsb_ablate_$v.so python bench.py --steps ${STEPS:-960} --warmup 64 --no-extra --no-cpu-baseline $BENCH_ARGS 2>/dev/null |
            python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f us per substep, %.1f us per launch' % (d['config']['kernel_us_per_substep'], d['roofline']['avg_launch_us']))"
    done
fi
