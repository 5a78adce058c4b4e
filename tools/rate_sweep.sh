for shape in "36 100 2097152" "64 128 2097152" "100 300 1048576" "150 500 1048576" "250 600 524288" "300 1000 262144" "500 1500 131072" "800 2000 65536" "1200 3000 32768" "2000 4000 16384" "2048 8000 8192" "2100 4000 16384" "3000 5000 8192"; do
  set -- $shape
  for aff in 0 1; do for opt in 0 1; do
    python tools/geom_sweep.py --R $1 --F $2 --n $3 --iters 2 --geoms 0x0 --affine $aff --opt $opt 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('R=$1 F=$2 n=$3 affine=$aff opt=$opt', d['ms'], 'ms', d['gcups'], 'GCUPS', '%dx%d' % (d['group_lanes'], d['rows_per_lane']), 'lds', d['lds_per_wave'])"
  done; done
done
