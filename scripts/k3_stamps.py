"""Scratch: in-kernel cycle stamps of the 64 x 128 (SK) tile body of gemm_tn (K3: Z = X^T Y)."""
import ctypes as C, os, sys, torch
os.environ.setdefault("DMDX_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dmd_era5_amd", "libdmdx_stamps.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
from dmd_era5_amd import _lib
K = default_kernels(); lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(1)
n, l, mb = (int(a) for a in (sys.argv[1:4] + ["8760", "60", "129780"][len(sys.argv) - 1:]))   # e.g. 3653 70 130872: a cfg4 block
nb = int(sys.argv[4]) if len(sys.argv) > 4 else 0            # > 0: the batched launch over nb such blocks (gemm_tn_blocks)
Xt = torch.randn((n, mb), generator=g, device="cuda")
Yt = torch.randn((l, mb), generator=g, device="cuda")
if nb:
    Xs = [Xt] + [torch.randn((n, mb), generator=g, device="cuda") for _ in range(nb - 1)]
    Ys = [Yt] + [torch.randn((l, mb), generator=g, device="cuda") for _ in range(nb - 1)]
    call = lambda: K.gemm_tn_blocks(Xs, Ys)
else:
    call = lambda: K.gemm_tn(Yt, Xt)
buf = (C.c_ulonglong * 12)()
for _ in range(2):
    call(); torch.cuda.synchronize(); lib.dmdx_debug_read_stamps(buf, 1)
K.events = []
call(); torch.cuda.synchronize(); lib.dmdx_debug_read_stamps(buf, 1)
n = buf[4]
print(f"X^T Y with {Xt.shape[0]} x {Yt.shape[0]} columns, {max(nb, 1)} block(s) of {Xt.shape[1]} rows:", "chunks", n, "avg cycles per chunk per wave: top %.0f  frags+mfma(3 k-steps) %.0f  vmcnt_wait %.0f  barrier %.0f  post(k-step 3, refill, fold) %.0f  total %.0f"
      % (buf[0]/n, buf[1]/n, buf[5]/n, buf[2]/n, buf[3]/n, (sum(buf[:4]) + buf[5])/n))
print("clock %.0f MHz; launch %.2f ms" % (100.0 * buf[6] / max(buf[7], 1), sum(e0.elapsed_time(e1) for _, _, e0, e1 in K.events)))
