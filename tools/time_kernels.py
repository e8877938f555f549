#!/usr/bin/env python3
"""Development aid (GPU box): per-kernel times of the compress path without any output check (for timing-aid builds /
ZSMI_STOP_SEQ / ZSMI_STOP_LIT runs whose output is not a valid frame)."""
import sys, os
if not os.environ.get("ZSMI_LIB_FILE"):
    os.environ["ZSMI_DEBUG_LIB"] = "1"      # the library built with -DZSMI_DEBUG_HOOKS (zstandard_amd/_lib.py); ZSMI_LIB_FILE: a variant build
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data as D
from zstandard_amd import BatchCodec, _lib
n, cs, steps = 4096, 65536, 5
cls = os.environ.get("CLS", "zipf")          # a class of tests/_corpus.py (16 MiB of it, tiled) instead of the Zipf log
if cls == "zipf":
    data = D.zipf_log(n * cs)
elif cls == "random":
    data = np.random.default_rng(7).integers(0, 256, n * cs, dtype=np.uint8)
elif cls == "zeros":
    data = np.zeros(n * cs, dtype=np.uint8)
else:
    import _corpus as C
    one = np.frombuffer(C.CLASSES[cls](16 << 20), dtype=np.uint8)
    data = np.tile(one, (n * cs + len(one) - 1) // len(one))[:n * cs].copy()
dev = torch.device("cuda:0")
dsrc = torch.from_numpy(data).to(dev)
bc = BatchCodec(device=0); Z = _lib.lib()
off = np.arange(n, dtype=np.uint64) * cs; sz = np.full(n, cs, dtype=np.uint32)
bound = int(Z.zsmi_compressBound(cs)); doff = np.arange(n, dtype=np.uint64) * bound
ddst = torch.empty(n * bound, dtype=torch.uint8, device=dev); dsz = torch.empty(n, dtype=torch.int32, device=dev)
for _ in range(2): bc.compress_device(dsrc.data_ptr(), off, sz, ddst.data_ptr(), doff, dsz.data_ptr(), 3)
bc.sync(); bc.enable_timing(True)
for _ in range(steps): bc.compress_device(dsrc.data_ptr(), off, sz, ddst.data_ptr(), doff, dsz.data_ptr(), 3)
bc.sync()
print(cls, {k: round(v[0] / steps * 1e3, 4) for k, v in bc.kernel_times().items()})
