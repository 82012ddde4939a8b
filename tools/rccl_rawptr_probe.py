#!/usr/bin/env python3
"""Does torch.distributed's nccl (= RCCL) backend all-reduce a tensor that wraps a raw device pointer of the library?
One rank on one GPU (all this pool has): exercises the tensor path the SPLPAK_ND_DIST hook uses, not the transport."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
hip = ctypes.CDLL("libamdhip64.so")
ptr = ctypes.c_void_p()
n = 1 << 20
assert hip.hipMalloc(ctypes.byref(ptr), ctypes.c_size_t(8 * n)) == 0
class Raw:
    def __init__(self, p, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(p), False), "version": 2}
v = torch.as_tensor(Raw(ptr.value, n), device=torch.device("cuda", 0))
v.fill_(1.5)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    dist.all_reduce(v)
st.synchronize()
print("raw-pointer all-reduce over nccl:", float(v.sum()), "expected", 1.5 * n, "data_ptr matches:", v.data_ptr() == ptr.value)
dist.destroy_process_group()
