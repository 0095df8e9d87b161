mkdir -p gpurun_out/ll; rm -f gpurun_out/ll/var.txt
for v in "-DSB_MT=1024u" "-DSB_MT=512u" "-DSB_MT=256u"; do
 for b in 64 128; do
  echo "== $v blocks $b" >> gpurun_out/ll/var.txt
  make -B -C softbody-webgpu_amd/csrc libsoftbody_hip.so EXTRA="$v" > /dev/null 2>&1 || exit 1
  if [ $b = 64 ]; then python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q -k "grid" 2>&1 | tail -1 >> gpurun_out/ll/var.txt; fi
  SB_MAINTAIN_BLOCKS=$b python bench.py --config3 --no-cpu-baseline --no-extra 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> gpurun_out/ll/var.txt
  SB_MAINTAIN_BLOCKS=$b python bench.py --soup --no-cpu-baseline --no-extra 2>&1 | grep -o '"ms_per_step": [0-9.]*' >> gpurun_out/ll/var.txt
 done
done
cat gpurun_out/ll/var.txt
