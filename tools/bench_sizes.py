#!/usr/bin/env python3
"""Development aid (GPU box): compress throughput and ratio for several chunk sizes / levels (device-resident, per-call wall time)."""
import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data as D, _oracle as O
from zstandard_amd import BatchCodec, _lib

def main():
    total = 256 << 20
    data = D.zipf_log(total)
    dev = torch.device("cuda:0")
    dsrc = torch.from_numpy(data).to(dev)
    bc = BatchCodec(device=0)
    Z = _lib.lib()
    for cs, lvl in ((65536, 3), (131072, 3), (131072, 1), (65536, 1), (100000, 3), (32768, 3), (1 << 20, 3)):
        n = total // cs
        off = (np.arange(n, dtype=np.uint64) * cs); sz = np.full(n, cs, dtype=np.uint32)
        bound = int(Z.zsmi_compressBound(cs)); doff = np.arange(n, dtype=np.uint64) * bound
        ddst = torch.empty(n * bound, dtype=torch.uint8, device=dev); dsz = torch.empty(n, dtype=torch.int32, device=dev)
        def run():
            bc.compress_device(dsrc.data_ptr(), off, sz, ddst.data_ptr(), doff, dsz.data_ptr(), lvl)
        for _ in range(2): run()
        bc.sync(); t0 = time.perf_counter()
        for _ in range(5): run()
        bc.sync(); dt = (time.perf_counter() - t0) / 5
        sizes = dsz.cpu().numpy().astype(np.uint32)
        assert (sizes < 0xFFFFFF88).all()
        # spot check: first and last frames decode (oracle D) to the input
        host = ddst.cpu().numpy()
        for i in (0, n - 1):
            f = host[int(doff[i]):int(doff[i]) + int(sizes[i])].tobytes()
            assert O.decompress(f, cs) == data[i * cs:(i + 1) * cs].tobytes()
        k = min(n, 64)
        ref = sum(len(O.zstd_compress(data[i * cs:(i + 1) * cs].tobytes(), lvl)) for i in range(k))
        print(f"chunk {cs:8d} L{lvl}: {n * cs / dt / 2**30:7.2f} GiB/s  ratio {n * cs / sizes.sum():.4f}  vs libzstd(first {k}) {ref / sizes[:k].sum():.4f}", flush=True)

if __name__ == "__main__":
    main()
