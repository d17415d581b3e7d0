#!/bin/bash
# Round 5, VERDICT r04 item 1: the tile decoder's second-level probes (csrc/mh_tile_probes.hpp) on one box, 4 GiB Zipf.
#   bash tools/r05_decode_probes.sh [outdir-under-gpurun_out] [size]
# Lines marked WRONG-OUTPUT are cost probes (MH_TILE_PROBE=2 keeps them from reporting what they decode).
out=${1:-r05_decode_probes}; size=${2:-4294967296}
export MH_BENCH_NO_INDEX_FREE=1
bash tools/dec_sweep.sh $out $size \
  "base" \
  "diag" \
  "diag MH_TILE_G=1 MH_TILE_PROBE=2" \
  "diag MH_TILE_G=1 MH_TILE_K=1 MH_TILE_PROBE=2" \
  "diag MH_TILE_G=2 MH_TILE_PROBE=2" \
  "diag MH_TILE_G=3 MH_TILE_PROBE=2" \
  "diag MH_TILE_G=4 MH_TILE_K=1 MH_TILE_PROBE=2" \
  "diag MH_TILE_G=5 MH_TILE_K=1 MH_TILE_PROBE=2" \
  "diag MH_TILE_G=4 MH_TILE_K=1 MH_TILE_WIN=1 MH_TILE_PROBE=2" \
  "diag MH_TILE_G=5 MH_TILE_K=1 MH_TILE_WIN=1 MH_TILE_PROBE=2" \
  "diag MH_TILE_G=5 MH_TILE_K=2 MH_TILE_PROBE=2" \
  "diag MH_TILE_G=6" \
  "diag MH_TILE_G=7" \
  "diag MH_TILE_G=6 MH_TILE_WIN=1" \
  "diag MH_TILE_G=7 MH_TILE_WIN=1" \
  "base"
