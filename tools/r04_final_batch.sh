#!/bin/bash
# Round-4 closing measurements (GPU box, repo root): the bench line at the metric's configuration and the other workloads.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_final; mkdir -p $O
b() { local name=$1; shift; timeout -k 10 400 python3 $R/bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; python3 - $O/$name.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("   ", d["value"], d["ms_per_step"], d["stages_ms"], d["round_trip_bit_exact"], d.get("decode_index_free"))
except Exception as e:
    print("    failed:", e)
PY
}
b bench16g_r04 --steps 20 --warmup 5
b bench4g_text --steps 5 --warmup 2 --size 4294967296 --kind text --no-cpu-baseline
b bench4g_uniform --steps 5 --warmup 2 --size 4294967296 --kind uniform --no-cpu-baseline
b bench4g_zipf --steps 5 --warmup 2 --size 4294967296 --no-cpu-baseline
b bench_config4_one_shard --steps 5 --warmup 2 --config 4 --no-cpu-baseline
b bench2g_shard --steps 20 --warmup 3 --size 2147483648 --no-cpu-baseline
b order2_text4g --steps 5 --warmup 2 --order 2 --kind text --size 4294967296 --no-cpu-baseline
b order2_text16g --steps 3 --warmup 1 --order 2 --kind text --size 17179869184 --no-cpu-baseline
for k in zipf text uniform; do echo "index-free $k"; timeout -k 10 200 python3 $R/tools/index_free_rate.py --size 4294967296 --kind $k 2>&1 | grep build_index | tail -1; done | tee $O/index_free_rate_4GiB.txt
