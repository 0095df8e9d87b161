run() { L=$1; shift
  python bench.py --no-cpu-baseline --no-extra --steps 800 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', '%.2f us/substep'%(d['ms_per_step']*1e3), 'tiles', d['config']['tiles'], 'K', d['roofline']['substeps_per_launch'])"
}
for t in 700 800 900 1024 1100 1200 1300; do for k in 4 5 6; do run tile${t}_k$k --tile $t --block-substeps $k; done; done
