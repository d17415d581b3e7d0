#!/bin/bash
# Counter passes over any tool of this directory (run on the GPU box, from the repo root):
#   bash tools/pmc_cmd.sh <outdir-under-gpurun_out> <tool.py> [args...]
# Each pass is its own rocprofv3 run with --pmc and the kernel trace only, as the pool requires.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
TOOL=$R/tools/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  [ -n "${PMC_ONLY:-}" ] && ! echo " $PMC_ONLY " | grep -q " $i " && continue
  timeout -k 10 240 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -- python3 $TOOL "$@" > $OUT/p$i.log 2>&1
  rc=$?
  echo "pass $i rc=$rc : $line" >> $OUT/passes.txt
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && break
done <<'PASSES'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE TA_BUSY_avr
FETCH_SIZE TCC_HIT_sum
WRITE_SIZE TCC_MISS_sum TCC_REQ_sum
PASSES
cat $OUT/passes.txt
