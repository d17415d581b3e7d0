#!/bin/bash
# HBM/fabric traffic of every kernel at the bench size (run on the GPU box from the repo root):
#   bash tools/pmc_traffic.sh <outdir-under-gpurun_out> [prof_run args, e.g. --size 17179869184]
# Separate --pmc passes (TCC has 4 slots; FETCH_SIZE takes 3, WRITE_SIZE 2).  The request-size
# counters give exact bytes where FETCH_SIZE is uncalibrated (MI355X_MICROARCH.md: FETCH_SIZE = RDREQ x 64 B).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/prof_run.py "$@" > $OUT/p$i.log 2>&1
  echo "pass $i rc=$? : $line" >> $OUT/passes.txt
done <<'PASSES'
FETCH_SIZE
WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum
PASSES
cat $OUT/passes.txt
