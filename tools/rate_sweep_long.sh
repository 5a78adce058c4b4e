for shape in "500 1500 131072" "800 2000 65536" "1000 2500 65536" "1200 3000 32768" "1600 3000 32768" "2000 4000 16384" "2048 8000 8192"; do
  set -- $shape
  for aff in 0 1; do for opt in 0 1; do
    VALIGN_HIP_DEBUG=force_long python tools/geom_sweep.py --R $1 --F $2 --n $3 --iters 2 --geoms 0x0 --affine $aff --opt $opt 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('force_long R=$1 F=$2 n=$3 affine=$aff opt=$opt', d['ms'], 'ms', d['gcups'], 'GCUPS')"
  done; done
done
