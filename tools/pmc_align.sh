#!/bin/bash
# PMC passes over the alignment path (developer tool; run on the GPU box).
# VALIGN_HIP_DEBUG=no_overlap: one fill and one traceback dispatch per million pairs (the default schedule cuts the batch
# 7/8 + 1/8 and per-dispatch means would mix the two sizes).
set -u
export VALIGN_HIP_DEBUG=no_overlap
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
i=0
for group in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
  "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $group --output-format csv -d $R/gpurun_out/pmc_align/p$i -- python3 $R/tools/align_bench.py --iters 1 > $R/gpurun_out/pmc_align_p$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
