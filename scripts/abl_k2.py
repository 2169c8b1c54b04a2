"""Scratch: timing-only ablations of the 16x16x4 K2 body (libdmdx_k2abl{1,2,4}.so built with
-DDMDX_K2_ABL: 1 no MFMAs, 2 no X loads, 4 no W staging / barrier; results are wrong) against the
product build, one subprocess each (the library is chosen at load time), same box.
Usage: python scripts/abl_k2.py   -> runs itself per library"""
import os, sys, subprocess, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, ROOT)
    from dmd_era5_amd.kernels import default_kernels
    K = default_kernels()
    mb, n, NB = 130872, 3653, 8
    g = torch.Generator(device="cuda").manual_seed(1)
    blocks = [torch.randn((n, mb), generator=g, device="cuda", dtype=torch.float32) for _ in range(NB)]
    for l in (70, 112, 128, 220):
        W = K.pitch(torch.randn((l, n), generator=g, device="cuda", dtype=torch.float32))
        out = torch.empty((l, mb), device="cuda", dtype=torch.float32)
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for B in blocks: K.skinny(B, W, out=out)
            e1.record(); e1.synchronize()
            if r: ts.append(e0.elapsed_time(e1))
        print(f"  l={l:4d}: {statistics.median(ts):7.2f} ms", flush=True)
else:
    for tag in ("", "k2abl1", "k2abl2", "k2abl4"):
        env = dict(os.environ)
        if tag: env["DMDX_LIB_PATH"] = os.path.join(ROOT, "dmd_era5_amd", f"libdmdx_{tag}.so")
        print(tag or "product", flush=True)
        subprocess.run([sys.executable, __file__, "child"], env=env)
