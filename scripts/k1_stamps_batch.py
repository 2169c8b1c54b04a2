"""Scratch (needs `make -C dmd_era5_amd/csrc stamps`): the cfg2 Gram as bench.py launches it (8 row blocks, one batched
launch): time inside the chunk loops (in-kernel stamps) against the launch's duration -- what is spent outside them
(unit prologue, partial-tile commit, last round, reduce kernel)."""
import ctypes as C, os, sys, torch
os.environ.setdefault("DMDX_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dmd_era5_amd", "libdmdx_stamps.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
from dmd_era5_amd import _lib
K = default_kernels(); lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(1)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 8
blocks = [torch.randn((8760, 129780), generator=g, device="cuda") for _ in range(NB)]
out = torch.empty((8760, 8760), dtype=torch.float64, device="cuda")
buf = (C.c_ulonglong * 12)()
for _ in range(2):
    K.syrk_blocks(blocks, out=out); torch.cuda.synchronize(); lib.dmdx_debug_read_stamps(buf, 1)
K.events = []
K.syrk_blocks(blocks, out=out); torch.cuda.synchronize(); lib.dmdx_debug_read_stamps(buf, 1)
ms = sum(a.elapsed_time(b) for _, _, a, b in K.events)
n = buf[4]
clock = 100.0 * buf[6] / max(buf[7], 1)
per = (sum(buf[:4]) + buf[5]) / n
loop_ms = n * per / 2048 / (clock * 1e3)          # 2048 waves resident
units = 2415 * 2 * NB
print(f"{NB} blocks: launch + reduce {ms:.1f} ms; {n} wave-chunks at {per:.0f} cycles (2 x 4096 = the matrix core) and {clock:.0f} MHz "
      f"= {loop_ms:.1f} ms inside the chunk loops ({100 * loop_ms / ms:.1f} %); outside them {ms - loop_ms:.1f} ms = "
      f"{1e3 * (ms - loop_ms) * 512 / units:.0f} us per unit ({units} units, 512 resident, {units / 512:.2f} rounds)")
print(f"per wave and unit: prologue {buf[8] / max(buf[10], 1) / clock:.1f} us, partial-tile commit (stores drained) {buf[9] / max(buf[10], 1) / clock:.1f} us, "
      f"chunk loop {buf[6] / max(buf[10], 1) / clock / 1e3:.3f} ms")
