"""Scratch: average lifetime of a K1 workgroup (one unit) from the clock probe of the product build, against the
launch duration: 512 resident workgroups x (launch / lifetime) rounds -- what the chip loses between units."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
K = default_kernels()
g = torch.Generator(device="cuda").manual_seed(1)
NB = 8
blocks = [torch.randn((8760, 129780), generator=g, device="cuda") for _ in range(NB)]
out = torch.empty((8760, 8760), dtype=torch.float64, device="cuda")
K.syrk_blocks(blocks, out=out); torch.cuda.synchronize()
for rep in range(3):
    ctr = torch.zeros(3, dtype=torch.int64, device="cuda")
    K.clock_probe(ctr)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); K.syrk_blocks(blocks, out=out); e1.record(); e1.synchronize()
    K.clock_probe(None)
    cyc, ticks, cnt = (int(v) for v in ctr.tolist())
    ms = e0.elapsed_time(e1)
    life_ms = ticks / cnt * 1e-5                      # 100 MHz ticks
    busy = cnt * life_ms / 512                         # if 512 workgroups were resident at every moment
    print(f"launch + reduce {ms:.2f} ms; {cnt} workgroups, average lifetime {life_ms:.3f} ms at {100.0 * cyc / ticks:.0f} MHz; "
          f"units x lifetime / 512 = {busy:.2f} ms ({100 * busy / ms:.1f} % of the launch); chunk loop at 8192 cycles per chunk: "
          f"{2028 * 8192 / (100.0 * cyc / ticks * 1e3):.3f} ms per unit", flush=True)
