#!/bin/bash
# Rehearsal of the N > 1 bench paths on ONE card (run on the GPU box from the repo root): two ranks share
# device 0, collectives go through gloo (staged via host memory).  Not a measurement of scaling — the
# 8-GPU node is the driver's — only a check that every mode runs and round-trips.
#   bash tools/rehearse_multi.sh <outdir-under-gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
export MH_BENCH_DEVICE=0
run() {  # name, bench args...
  local name=$1; shift
  timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      $R/bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err
  echo "$name rc=$?" >> $OUT/rehearse.txt
  python3 - "$OUT/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2].ljust(18), d["value"], d["scaling"], d["stages_ms"], "ok" if d["round_trip_bit_exact"] else "ROUND TRIP FAILED")
except Exception as e:
    print(sys.argv[2].ljust(18), "failed:", e)
PY
}
run default_strong_16g
run weak_zipf_2g --size 2147483648
run strong_zipf_4g --total-size 4294967296
run config4_1g --config 4 --size 1073741824
run order2_text_1g --order 2 --kind text --size 1073741824

run order2_text_1g_allreduce --order 2 --kind text --size 1073741824 --o2-exchange allreduce
# ONE rank through the same N > 1 code path on the real backend (RCCL): process group, all-reduce, all-gather,
# reduce-scatter, barrier, pre-shifted encode — the calls the driver's multi-GPU bench will make
run1() {
  local name=$1; shift
  MH_BENCH_DIST1=1 timeout -k 10 300 python3 $R/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err
  echo "$name rc=$?" >> $OUT/rehearse.txt
  python3 - "$OUT/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2].ljust(18), d["value"], d["config"]["backend"], d["stages_ms"], "ok" if d["round_trip_bit_exact"] else "ROUND TRIP FAILED")
except Exception as e:
    print(sys.argv[2].ljust(18), "failed:", e)
PY
}
run1 rccl_world1_16g
run1 rccl_world1_config4 --config 4
run1 rccl_world1_order2 --order 2 --kind text --size 4294967296

run1 rccl_world1_order2_allreduce --order 2 --kind text --size 4294967296 --o2-exchange allreduce
cat $OUT/rehearse.txt
