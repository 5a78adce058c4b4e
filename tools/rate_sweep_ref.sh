for shape in "150 500 1048576" "150 2000 262144" "150 8000 65536" "150 20000 32768" "150 32000 16384" "500 8000 32768" "500 20000 16384"; do
  set -- $shape
  for aff in 0 1; do
    python tools/geom_sweep.py --R $1 --F $2 --n $3 --iters 2 --geoms 0x0 --affine $aff 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('R=$1 F=$2 n=$3 affine=$aff', d['ms'], 'ms', d['gcups'], 'GCUPS', '%dx%d' % (d['group_lanes'], d['rows_per_lane']), 'lds', d['lds_per_wave'], 'wpb', d['waves_per_block'])"
  done
done
