#!/bin/bash
# A/B of tile-decoder builds and their diagnostic switches on one box (GPU box, repo root):
#   bash tools/dec_sweep.sh <outdir-under-gpurun_out> <size-bytes> "<tag> [ENV=VAL ...]" ...
# tag "base" = libmhc.so, anything else = libmhc_<tag>.so (csrc/Makefile `exp`).  One bench line per spec.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
SIZE=$1; shift
mkdir -p $OUT
i=0
for spec in "$@"; do
  i=$((i+1))
  set -- $spec
  tag=$1; shift
  name=$(echo "$spec" | tr ' =/' '___')
  (
    [ "$tag" != base ] && export MH_LIB=$R/markov-huffman-coding_amd/libmhc_$tag.so
    for kv in "$@"; do export "$kv"; done
    timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 --size $SIZE ${BENCH_ARGS:-} > $OUT/$name.json 2> $OUT/$name.err
    echo "$spec rc=$?" >> $OUT/ab.txt
  )
  python3 - "$OUT/$name.json" "$spec" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2].ljust(44), d["stages_ms"], "ok" if d["round_trip_bit_exact"] else "WRONG-OUTPUT")
except Exception as e:
    print(sys.argv[2].ljust(44), "failed:", e)
PY
done | tee $OUT/summary.txt
