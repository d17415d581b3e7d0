#!/bin/bash
# Round-5 evidence batch (GPU box, repo root): the whole GPU suite, the bench line, kernel times of the bench command, counter
# passes at the bench size (order-1 kernels: profiles/secondary.json / traffic.json are made from these, with the source hashes
# of the kernels as they are NOW) and at 4 GiB for the two kernels of the index-free decode.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_prof; mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests_final.txt 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests_final.txt
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $O/bench16g_r05.json 2> $O/bench16g_r05.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench16g -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench16g_under_rocprof.json 2> $O/bench16g.err
echo "bench under rocprof rc=$?"
f=$(find $O/bench16g -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/bench16g_kernel_stats.csv
cd $R
bash tools/pmc_passes.sh r05_prof/pmc16g --size 17179869184 > $O/pmc_passes.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r05_prof/pmc16g $O/pmc16g_summary.json > $O/pmc16g_summary.txt 2>&1
bash tools/pmc_traffic.sh r05_prof/traffic16g --size 17179869184 > $O/pmc_traffic.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r05_prof/traffic16g $O/traffic16g_counters.json > $O/traffic16g_counters.txt 2>&1
PMC_ONLY="1 2 3 4 5" bash tools/pmc_cmd.sh r05_prof/stream4g stream_rate.py --size 4294967296 --reps 2 > $O/pmc_stream.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r05_prof/stream4g $O/stream4g_counters.json > $O/stream4g_counters.txt 2>&1
# keep the merged output small: the raw per-dispatch CSVs stay on the box
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*domain_stats.csv" -delete
du -sh $O; ls $O; cat $O/pmc16g/passes.txt $O/traffic16g/passes.txt 2>/dev/null | tail -12
