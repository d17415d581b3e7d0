#!/bin/bash
# A/B runs of experimental library builds (csrc/Makefile `exp`) on the GPU box, from the repo root:
#   bash tools/ab_bench.sh <outdir-under-gpurun_out> <tag> [<tag> ...] [-- bench args]
# tag "base" = the regular libmhc.so, "vN" = the regular library with MH_DEC_VARIANT=N, anything else =
# libmhc_<tag>.so.  One bench line per tag (no CPU baseline), then a summary.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
TAGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do TAGS+=("$1"); shift; done
[ $# -gt 0 ] && shift
for t in "${TAGS[@]}"; do
  unset MH_LIB MH_DEC_VARIANT
  case "$t" in
    base) ;;
    v[0-9]*) export MH_DEC_VARIANT=${t#v} ;;                 # decode A/B variant of the regular library
    *) export MH_LIB=$R/markov-huffman-coding_amd/libmhc_$t.so ;;
  esac
  timeout -k 10 300 python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $OUT/$t.json 2> $OUT/$t.err
  echo "$t rc=$?" >> $OUT/ab.txt
done
python3 - "$OUT" "${TAGS[@]}" <<'PY'
import json, sys
out = sys.argv[1]
for t in sys.argv[2:]:
    try:
        d = json.load(open("%s/%s.json" % (out, t)))
        print(t.ljust(12), d["value"], d["stages_ms"], "ok" if d["round_trip_bit_exact"] else "ROUND TRIP FAILED")
    except Exception as e:
        print(t.ljust(12), "failed:", e)
PY
