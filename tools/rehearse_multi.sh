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
if [ "${REHEARSE_SKIP_PAIRS:-0}" != 1 ]; then
run default_strong_16g
run weak_zipf_2g --size 2147483648
run strong_zipf_4g --total-size 4294967296
run config4_1g --config 4 --size 1073741824
run order2_text_1g --order 2 --kind text --size 1073741824
run config5_2x1g_strong --config 5 --total-size 2147483648      # [r5] BASELINE configs[4] as ONE stream split over the ranks

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
fi

# [r5] MORE ranks through bench.py's own launcher (`python bench.py --gpus N` starts its N ranks as a child process), all on
# device 0 with gloo.  VERDICT r04 asked for eight.  The GPU boxes of this pool end a run in which more than six processes have
# the card open ("process guard"), and the launcher's agent (python -m torch.distributed.run) counts as one: six ranks were
# killed with "7 processes had the GPU open (limit 6)" (gpurun_out/.last_call.json of that call, quoted in
# profiles/r05/rehearse/README.md).  So the most one card can rehearse is FIVE ranks — five shards of the 16 GiB stream (strong
# scaling, the default) and config 4 with 1 GiB per rank — plus four.  The eight-way split itself (shard bounds, the 8-entry
# all-gather, eight models from one all-reduce) runs in the CPU suite with world size 8 (tests/test_sharded_cpu.py).
runn() {  # name, ranks, bench args...
  local name=$1 n=$2; shift 2
  MH_BENCH_DEVICE=0 timeout -k 10 600 python3 $R/bench.py --gpus $n --backend gloo --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err
  echo "$name rc=$?" >> $OUT/rehearse.txt
  python3 - "$OUT/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2].ljust(18), d["value"], d["scaling"], "ranks", d["config"]["ranks"], d["stages_ms"], "ok" if d["round_trip_bit_exact"] else "ROUND TRIP FAILED")
except Exception as e:
    print(sys.argv[2].ljust(18), "failed:", e)
PY
}
if [ "${REHEARSE_MANY:-1}" = 1 ]; then
  runn strong5_gloo 5
  runn config4_5x1g_gloo 5 --config 4 --size 1073741824
  runn strong4_gloo 4
  runn config4_4x1g_gloo 4 --config 4 --size 1073741824
  cat $OUT/rehearse.txt
fi
