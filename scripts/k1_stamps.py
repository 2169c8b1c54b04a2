import ctypes as C, os, sys, torch
os.environ.setdefault("DMDX_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dmd_era5_amd", "libdmdx_stamps.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
from dmd_era5_amd import _lib
K = default_kernels(); lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(1)
Xt = torch.randn((8760, 129780), generator=g, device="cuda")
buf = (C.c_ulonglong * 12)()
K.syrk(Xt); torch.cuda.synchronize(); lib.dmdx_debug_read_stamps(buf, 1)
K.syrk(Xt); torch.cuda.synchronize(); lib.dmdx_debug_read_stamps(buf, 1)
n = buf[4]
print("chunks", n, "avg cycles per chunk per wave: dma_issue %.0f  frags+mfma %.0f  vmcnt_wait %.0f  barrier %.0f  post %.0f  total %.0f"
      % (buf[0]/n, buf[1]/n, buf[5]/n, buf[2]/n, buf[3]/n, (sum(buf[:4]) + buf[5])/n))
print("sustained core clock in the chunk loop: %.0f MHz (s_memtime / s_memrealtime * 100 MHz)" % (100.0 * buf[6] / max(buf[7], 1)))
