#!/bin/bash
# rocprofv3 PMC passes over tools/prof_kernel.py, one counter group per run (no trace domains
# mixed in).  Usage: tools/pmc_passes.sh <tag> [prof_kernel args...]   (run on the GPU box)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
i=0
for group in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
  "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  rocprofv3 --pmc $group --output-format csv -d $R/gpurun_out/pmc_${TAG}/p$i -- python3 $R/tools/prof_kernel.py "$@" > $R/gpurun_out/pmc_${TAG}_p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
