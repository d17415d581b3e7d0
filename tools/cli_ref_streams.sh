#!/bin/bash
# Streams written by the genuine reference binary (oracle/_ref), decompressed by bin/markovhuffman WITHOUT a sidecar
# index (GPU box, from the repo root): random bytes, "ABC" repeated, long runs — the index builder's hard cases.
#   bash tools/cli_ref_streams.sh [MiB per file, default 256]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
MIB=${1:-256}
W=/dev/shm/clichk; rm -rf $W; mkdir -p $W; cd $W
python3 - "$MIB" <<'PY'
import sys
import numpy as np
n = int(sys.argv[1]) << 20
np.random.default_rng(1).integers(0, 256, n, dtype=np.uint8).tofile("random.bin")
np.tile(np.frombuffer(b"ABC", dtype=np.uint8), n // 3 + 1)[:n].tofile("abc.bin")
(np.arange(n, dtype=np.int64) // 4096 % 256).astype(np.uint8).tofile("runs.bin")
PY
for f in random abc runs; do
  t0=$(date +%s.%N)
  $R/oracle/_ref/markovhuffman $f.bin -o $f.cm -d $f.e > /dev/null 2>&1
  t1=$(date +%s.%N)
  $R/bin/markovhuffman $f.cm -x -o $f.out -e $f.e > $f.log 2>&1
  t2=$(date +%s.%N)
  $R/oracle/_ref/markovhuffman $f.cm -x -o $f.ref -e $f.e > /dev/null 2>&1
  t3=$(date +%s.%N)
  if cmp -s $f.bin $f.out; then same=identical; else same=DIFFERENT; fi
  python3 -c "print('%-7s %d MiB: reference compress %.1f s, gpu cli decompress without index %.2f s (%s), reference decompress %.1f s' % ('$f', $MIB, $t1-$t0, $t2-$t1, '$same', $t3-$t2))"
done
rm -rf $W
