#!/bin/bash
# PMC passes over the long-read kernel (developer tool; run on the GPU box): VALU share, LDS, HBM.
# Usage: tools/pmc_long.sh <tag> [extra geom_sweep.py args, e.g. --affine 1]
set -u
TAG=${1:-long}; shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
i=0
for group in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
  "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  rocprofv3 --pmc $group --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -- python3 $R/tools/geom_sweep.py --R 2500 --F 5000 --n 65536 --iters 1 --geoms 0x0 "$@" > $R/gpurun_out/pmc_${TAG}_p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
