#!/bin/bash
# Round 5: the bench records VERDICT r04 asked for (GPU box, repo root): config 2 at its own size with a kernel trace,
# config 4's shard, the other workloads.   bash tools/r05_records.sh [outdir-under-gpurun_out]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-r05_records}
mkdir -p $OUT
b() { name=$1; shift; timeout -k 10 400 python3 $R/bench.py "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; python3 -c "
import json,sys
d=json.load(open('$OUT/$name.json')); print('   ', d['value'], d['stages_ms'], 'fixed', d.get('fixed_cost_share'), 'ok' if d['round_trip_bit_exact'] else 'ROUND TRIP FAILED')"; }
b bench_config2_256MiB --config 2 --steps 20 --warmup 3
b bench_config4_one_shard --config 4 --no-cpu-baseline
b bench4g_uniform --kind uniform --size 4294967296 --no-cpu-baseline
b bench4g_text --kind text --size 4294967296 --no-cpu-baseline
b bench4g_zipf --size 4294967296 --no-cpu-baseline
b bench2g_shard --size 2147483648 --no-cpu-baseline --steps 10
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/config2_trace -- python3 $R/bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/config2_under_rocprof.json 2> $OUT/config2_trace.log
f=$(find $OUT/config2_trace -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $OUT/config2_kernel_stats.csv && grep mhk "$f" | cut -d, -f1-4 | cut -c1-150
