#!/bin/bash
# Round-5 closing measurements (GPU box, repo root): the bench line at the metric's configuration (now with this round's counters
# on it) and the other workloads.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_final; mkdir -p $O
b() { local name=$1; shift; timeout -k 10 400 python3 $R/bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; python3 - $O/$name.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("   ", d["value"], d["ms_per_step"], d["stages_ms"], d["round_trip_bit_exact"], "traffic", d["roofline"]["traffic"], {k: v for k, v in (d.get("decode_index_free") or {}).items() if k != "via_index"})
except Exception as e:
    print("    failed:", e)
PY
}
b bench16g_r05 --steps 20 --warmup 5
b bench_config2_256MiB --config 2 --steps 20 --warmup 3
b bench4g_text --steps 5 --warmup 2 --size 4294967296 --kind text --no-cpu-baseline
b bench4g_uniform --steps 5 --warmup 2 --size 4294967296 --kind uniform --no-cpu-baseline
b bench4g_zipf --steps 5 --warmup 2 --size 4294967296 --no-cpu-baseline
b bench_config4_one_shard --steps 5 --warmup 2 --config 4 --no-cpu-baseline
b bench2g_shard --steps 20 --warmup 3 --size 2147483648 --no-cpu-baseline
b order2_text4g --steps 5 --warmup 2 --order 2 --kind text --size 4294967296 --no-cpu-baseline
b order2_text16g --steps 3 --warmup 1 --order 2 --kind text --size 17179869184 --no-cpu-baseline
for k in zipf text; do echo "stream $k"; timeout -k 10 200 python3 $R/tools/stream_rate.py --size 4294967296 --kind $k 2>&1 | tail -1; done | tee $O/stream_rate_4GiB.txt
