#!/bin/bash
# Round 5, VERDICT r04 item 2: where the two passes of the index-free decode spend their time (GPU box, repo root).
#   bash tools/r05_stream_probes.sh [outdir-under-gpurun_out] [size]
out=gpurun_out/${1:-r05_stream_probes}; size=${2:-4294967296}
mkdir -p $out
R=$(pwd)
run() { name=$1; shift; ( for kv in "$@"; do export "$kv"; done; echo "== $name $*"; timeout -k 10 200 python3 tools/stream_rate.py --size $size --kind ${KIND:-zipf} 2>&1 | tail -2 ) | tee -a $out/summary.txt; }
run base
run warm64 MH_INDEX_WARM_BITS=64
run warm96 MH_INDEX_WARM_BITS=96
run warm128 MH_INDEX_WARM_BITS=128
run warm256 MH_INDEX_WARM_BITS=256
run nostore MH_LIB=$R/markov-huffman-coding_amd/libmhc_diag.so MH_SEG_PROBE=1
KIND=text run text
KIND=text run text_nostore MH_LIB=$R/markov-huffman-coding_amd/libmhc_diag.so MH_SEG_PROBE=1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/trace -- python3 $R/tools/stream_rate.py --size $size > $R/$out/trace.log 2>&1
cd $R
f=$(find $out/trace -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -14 "$f" | cut -c1-200 | tee -a $out/summary.txt
