"""Scratch: K3 (Z = X^T Y over the 8 cfg2 row blocks, one launch) against the K-split knob."""
import os, sys, time, torch, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from dmd_era5_amd.kernels import default_kernels
    K = default_kernels()
    n, mb, l = 8760, 129780, int(sys.argv[2])
    g = torch.Generator(device="cuda").manual_seed(1)
    Xb = [torch.randn((n, mb), generator=g, device="cuda") for _ in range(8)]
    Yb = [torch.randn((l, mb), generator=g, device="cuda") for _ in range(8)]
    for _ in range(2): K.gemm_tn_blocks(Xb, Yb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): K.gemm_tn_blocks(Xb, Yb)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"l={l} ROUNDS={os.environ.get('DMDX_TN_ROUNDS','default')} MAX_CPS={os.environ.get('DMDX_TN_MAX_CPS','default')}: {ms:.2f} ms/pass -> {8*n*mb*4/ms/1e9:.2f} TB/s", flush=True)
else:
    for l in (20, 60):
        for rounds, cps in (("", ""), ("3", ""), ("4", ""), ("9", ""), ("12", ""), ("6", "4096"), ("3", "4096"), ("1", "4096"), ("2", "4096")):
            env = dict(os.environ)
            if rounds: env["DMDX_TN_ROUNDS"] = rounds
            if cps: env["DMDX_TN_MAX_CPS"] = cps
            subprocess.run([sys.executable, __file__, "child", str(l)], env=env)
