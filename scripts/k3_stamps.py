"""Scratch: in-kernel cycle stamps of the 64 x 128 (SK) tile body of gemm_tn (K3: Z = X^T Y)."""
import ctypes as C, os, sys, torch
os.environ.setdefault("DMDX_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dmd_era5_amd", "libdmdx_stamps.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
from dmd_era5_amd import _lib
K = default_kernels(); lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(1)
Xt = torch.randn((8760, 129780), generator=g, device="cuda")
Yt = torch.randn((60, 129780), generator=g, device="cuda")
buf = (C.c_ulonglong * 8)()
for _ in range(2):
    K.gemm_tn(Yt, Xt); torch.cuda.synchronize(); lib.dmdx_debug_read_stamps(buf, 1)
K.events = []
K.gemm_tn(Yt, Xt); torch.cuda.synchronize(); lib.dmdx_debug_read_stamps(buf, 1)
n = buf[4]
print("chunks", n, "avg cycles per chunk per wave: top %.0f  frags+mfma(3 k-steps) %.0f  vmcnt_wait %.0f  barrier %.0f  post(k-step 3, refill, fold) %.0f  total %.0f"
      % (buf[0]/n, buf[1]/n, buf[5]/n, buf[2]/n, buf[3]/n, (sum(buf[:4]) + buf[5])/n))
print("clock %.0f MHz; launch %.2f ms" % (100.0 * buf[6] / max(buf[7], 1), sum(e0.elapsed_time(e1) for _, _, e0, e1 in K.events)))
