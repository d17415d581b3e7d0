#!/bin/bash
# Round-4 evidence batch (GPU box, repo root): kernel times of the bench command, counter passes at the bench size.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench16g -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench16g_under_rocprof.json 2> $O/bench16g.err
echo "bench under rocprof rc=$?"
cd $R
bash tools/pmc_passes.sh r04_prof/pmc16g --size 17179869184 > $O/pmc_passes.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r04_prof/pmc16g $O/pmc16g_summary.json > $O/pmc16g_summary.txt 2>&1
bash tools/pmc_traffic.sh r04_prof/traffic16g --size 17179869184 > $O/pmc_traffic.log 2>&1
python3 tools/pmc_summary.py gpurun_out/r04_prof/traffic16g $O/traffic16g_counters.json > $O/traffic16g_counters.txt 2>&1
cd /tmp
# (the histogram's head-byte probe needs a diagnostic build: make -C markov-huffman-coding_amd/csrc exp TAG=nohead EXPFLAGS=-DMH_HIST_PROBE_NOHEAD,
#  then: MH_LIB=.../libmhc_nohead.so rocprofv3 --pmc TCC_EA0_RDREQ_sum ... -- python3 tools/hist_only.py; record: profiles/r04/hist_head_byte_*.txt)
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/index4g/p1 -- python3 $R/tools/index_free_rate.py --size 4294967296 > $O/index4g_p1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TA_BUSY_avr --kernel-trace --output-format csv -d $O/index4g/p2 -- python3 $R/tools/index_free_rate.py --size 4294967296 > $O/index4g_p2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/index4g/p3 -- python3 $R/tools/index_free_rate.py --size 4294967296 > $O/index4g_p3.log 2>&1
python3 $R/tools/pmc_summary.py $O/index4g $O/index4g_counters.json > $O/index4g_counters.txt 2>&1
# keep the merged output small: the raw per-dispatch CSVs stay on the box
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; ls $O
