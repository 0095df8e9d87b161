run() { # label, lib, extra args
  L=$1; LIB=$2; shift 2
  SOFTBODY_HIP_LIB=$LIB python bench.py --no-cpu-baseline --no-extra --steps 840 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', '%.2f us/substep'%(d['ms_per_step']*1e3), 'tiles', d['config']['tiles'], 'K', d['roofline']['substeps_per_launch'], 'GB/s %.0f'%d['roofline']['achieved'], 'launch us %.1f' % d['roofline']['avg_launch_us'])"
}
D=softbody-webgpu_amd/csrc
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_halo.py -m gpu -x -q 2>&1 | tail -2
for k in 3 4 5 6 7; do run base_k$k $D/libsoftbody_hip.so --block-substeps $k; done
run base_4M_k6 $D/libsoftbody_hip.so --block-substeps 6 --width 2000 --height 2000 --steps 420
run cfg5_share_k6 $D/libsoftbody_hip.so --block-substeps 6 --width 1000 --height 8000 --mixed-stiffness --subticks 128 --steps 240
