#!/bin/bash
# A/B of the tile decoder's geometry on one box: first-level width P x tiles per wave K, against the chunk decoder.
# usage (on the GPU box): tools/tile_sweep.sh <size-bytes> <outdir>
size=${1:-4294967296}; out=${2:-gpurun_out/tile_sweep}; mkdir -p $out
MH_BENCH_NO_FINE=1 timeout -k 10 300 python bench.py --size $size --steps 3 --no-cpu-baseline > $out/chunk.json 2>/dev/null
for P in ${PS:-6 7 8}; do for K in ${KS:-1 2 3}; do
  MH_TILE_P=$P MH_TILE_K=$K timeout -k 10 300 python bench.py --size $size --steps 3 --no-cpu-baseline > $out/p${P}_k${K}.json 2>/dev/null || echo "P=$P K=$K failed"
done; done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.load(open(f)); print(os.path.basename(f), d["stages_ms"], d["round_trip_bit_exact"])
    except Exception as e: print(f, "unreadable", e)
PY
