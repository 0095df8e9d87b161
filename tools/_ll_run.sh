for w in 1000 1050 1060; do for k in 5 6; do
  echo "width $w K $k: $(python bench.py --width $w --block-substeps $k --no-cpu-baseline --no-extra --steps 1680 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"substeps_per_launch": [0-9]*\|"region_particles": [0-9]*\|"tiles": [0-9]*' | tr '\n' ' ')"
done; done
