"""development aid: kernel times of one compress step without any output check"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from zstandard_amd import BatchCodec
import _data as D
n, cs = 4096, 65536
host = D.zipf_log(n * cs, threads=32)
bc = BatchCodec(0, torch.cuda.current_stream().cuda_stream)
d_src = torch.from_numpy(host).cuda()
stride = 66048
d_dst = torch.empty(n * stride, dtype=torch.uint8, device="cuda"); d_sizes = torch.zeros(n, dtype=torch.int32, device="cuda")
offs = np.arange(n, dtype=np.uint64) * cs; sizes = np.full(n, cs, dtype=np.uint32); doffs = np.arange(n, dtype=np.uint64) * stride
for _ in range(2): bc.compress_device(d_src.data_ptr(), offs, sizes, d_dst.data_ptr(), doffs, d_sizes.data_ptr(), 3)
torch.cuda.synchronize(); bc.enable_timing(True)
for _ in range(3): bc.compress_device(d_src.data_ptr(), offs, sizes, d_dst.data_ptr(), doffs, d_sizes.data_ptr(), 3)
torch.cuda.synchronize()
print({k: round(v[0] / 3 * 1e3, 3) for k, v in bc.kernel_times().items()}, "stop", os.environ.get("ZSMI_STOP_LIT"), os.environ.get("ZSMI_STOP_SEQ"))
