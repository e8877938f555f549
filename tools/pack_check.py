#!/usr/bin/env python3
"""GPU check of zsmi_packFramesDevice (run by tests/test_gpu_codec.py::test_pack_frames_device): a ragged batch is compressed on
the device, packed, and the packed run must be the frames back to back, each decoding under oracle D."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data as D, _oracle as O
from zstandard_amd import BatchCodec

def main():
    codec = BatchCodec(0)
    rng = np.random.default_rng(5)
    data = D.zipf_log(3 << 20, seed_lo=11)
    n = 97
    sizes = rng.integers(1, 70000, n).astype(np.uint32)
    offs = rng.integers(0, len(data) - 70000, n).astype(np.uint64)
    Z = codec.L
    bounds = np.array([Z.zsmi_compressBound(int(s)) for s in sizes], dtype=np.uint64)
    doffs = np.zeros(n, dtype=np.uint64); doffs[1:] = np.cumsum(bounds)[:-1]
    dev = torch.device("cuda:0")
    d_src = torch.from_numpy(data).to(dev)
    d_dst = torch.zeros(int(bounds.sum()), dtype=torch.uint8, device=dev)
    d_sizes = torch.zeros(n, dtype=torch.int32, device=dev)
    codec.compress_device(d_src.data_ptr(), offs, sizes, d_dst.data_ptr(), doffs, d_sizes.data_ptr(), 3)
    d_packed = torch.zeros(int(bounds.sum()), dtype=torch.uint8, device=dev)
    d_poffs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    codec.pack_device(d_dst.data_ptr(), doffs, d_sizes.data_ptr(), n, d_packed.data_ptr(), d_poffs.data_ptr())
    codec.sync()
    fs = d_sizes.cpu().numpy().astype(np.uint32); assert (fs < 0xFFFFFF88).all()
    po = d_poffs.cpu().numpy().astype(np.uint64)
    assert po[0] == 0 and (np.diff(po) == fs).all(), "packed offsets are not the running sum of the frame sizes"
    arena = d_dst.cpu().numpy(); packed = d_packed.cpu().numpy()
    want = np.concatenate([arena[int(doffs[i]):int(doffs[i]) + int(fs[i])] for i in range(n)])
    assert (packed[:len(want)] == want).all(), "packed bytes differ from the frames back to back"
    for i in (0, n // 2, n - 1):
        c = data[int(offs[i]):int(offs[i]) + int(sizes[i])].tobytes()
        assert O.decompress(packed[int(po[i]):int(po[i + 1])].tobytes(), len(c)) == c
    print("pack ok")
    return 0

if __name__ == "__main__":
    sys.exit(main())
