#!/bin/bash
# Chunk-size sweep of the bench workload (run on the GPU box from the repo root):
#   bash tools/chunk_sweep.sh <outdir-under-gpurun_out> [bench args]
# One bench line per chunk size (MH_BENCH_CHUNK = 256 / 512 / 1024), then the fabric-traffic counter
# passes (tools/pmc_traffic.sh) for the sizes named in SWEEP_TRAFFIC (default: 256).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
for c in 256 512 1024; do
  MH_BENCH_CHUNK=$c timeout -k 10 400 python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $OUT/bench_chunk$c.json 2> $OUT/bench_chunk$c.err
  echo "chunk $c rc=$?" >> $OUT/sweep.txt
done
for c in ${SWEEP_TRAFFIC:-256}; do
  MH_BENCH_CHUNK=$c bash $R/tools/pmc_traffic.sh $(basename $OUT)/traffic_chunk$c --size 17179869184
done
cat $OUT/sweep.txt
