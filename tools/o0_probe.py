import os, sys, ctypes
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import bench, __graft_entry__ as entry
n = 4 << 30
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
mhc = entry.load_package()
bench.CHUNK = 1024
data = bench.generate("zipf", n, 2, 0, dev)
codec = bench.Codec(mhc, n, dev)
counts0 = torch.bincount(data[: 1 << 28].to(torch.int64), minlength=256).cpu().numpy().astype("uint64")
model = mhc.Model.from_counts(counts0, 0)
print("layout", model.decode_layout(), "maxlen", model.max_code_len)
ev = lambda: torch.cuda.Event(enable_timing=True)
for it in range(3):
    e = [ev() for _ in range(3)]
    e[0].record(); codec.encode(model, data, 0x20); e[1].record()
    codec.decode(model); e[2].record(); torch.cuda.synchronize()
    nbits = int(codec.nbits[0].item())
    print("encode %.3f ms decode %.3f ms ratio %.4f" % (e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), nbits / 8 / n))
print("ok", bool(torch.equal(codec.decoded, data)))
