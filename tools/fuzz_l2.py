"""More random cases of tests/test_gpu_fuzz.py::l2_layout_case (the decoder's L2 layout) on the GPU box:
   python3 tools/fuzz_l2.py [first_seed [count]]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry
from oracle import mh_oracle as oracle
import test_gpu_fuzz as F
mhc = entry.load_package()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for seed in range(first, first + count):
    try:
        print(seed, F.l2_layout_case(mhc, oracle, seed), flush=True)
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed", seed, e, flush=True)
print("done, mismatches:", bad)
