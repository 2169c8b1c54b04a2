"""Randomized-SVD companion of bench.py (same cfg2 matrix, sklearn-default randomized SVD as the
step): one JSON line with the whole-path GB/s and, for the two tall-skinny GEMM kernels that
stream X (K2: Y = X Q, K3: Z = X^T Y), the achieved HBM rate of the snapshot stream and the
algorithmic TFLOP/s per launch, from HIP events around the launches of one extra step.
    python scripts/bench_randomized.py [--steps 3] [--warmup 1] [--k 50]"""
import argparse, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dmd_era5_amd import svd as dsvd
from dmd_era5_amd.kernels import default_kernels

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--k", type=int, default=50)
a = ap.parse_args()
kern = default_kernels()
m, n, _, desc = bench.WORKLOADS["cfg2"]
blocks = bench.make_snapshot_blocks(m, n, 1234, torch.device("cuda"))
for B in blocks:
    kern.row_center_scale_(B, False)
step = lambda: dsvd.svd_randomized(blocks, a.k, random_state=0, kern=kern)
for _ in range(a.warmup):
    res = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    res = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
# kernel breakdown from one more step with HIP events around every launch (the ~300 event records
# cost host time, so that step is not the timed one)
kern.events = []
res = step()
torch.cuda.synchronize()
events, kern.events = kern.events, None
l = int(res.info["l"])
rows = {"skinny": [], "gemm_tn": []}
for name, shape, e0, e1 in events:
    if name in ("skinny", "skinny_gram") and shape[1] == n:        # (m_b, n, l): X streamed (with or without the fused Gram)
        rows["skinny"].append((shape[0], e0.elapsed_time(e1)))
    elif name in ("gemm_tn", "gemm_tn_blocks") and shape[0] > 4 * n:  # (K, na, nb[, blocks]): X streamed
        rows["gemm_tn"].append((shape[0], e0.elapsed_time(e1)))
out = {"metric": "randomized rank-r SVD GB/s on ERA5 snapshot matrix (X resident in HBM)",
       "value": m * n * 4.0 / dt / 1e9, "unit": "GB/s", "ms_per_step": dt * 1e3, "n_gpus": 1,
       "config": {"workload": desc.replace("method-of-snapshots", "randomized (sklearn defaults)").replace("rank-50", f"rank-{a.k}"), "k": a.k,
                  "l": l, "n_iter": int(res.info["n_iter"]), "passes_over_X": 2 * int(res.info["n_iter"]) + 2},
       "kernels": {}}
for name, label in (("skinny", "K2 skinny_kernel (Y = X Q)"), ("gemm_tn", "K3 (Z = X^T Y: 64x128-tile generic body, or K3s at l <= 32)")):
    if not rows[name]:
        continue
    mb = float(np.mean([r[0] for r in rows[name]]))
    ms = float(np.mean([r[1] for r in rows[name]]))
    out["kernels"][label] = {
        "launches_per_step": len(rows[name]), "ms_per_launch": ms,
        "roofline": {"bound": "hbm", "achieved": 4.0 * mb * n / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": 4.0 * mb * n / (ms * 1e-3) / 1e9 / 8000.0},
        "algorithmic_tflops": 2.0 * mb * n * l / (ms * 1e-3) / 1e12,
        "mfma_bound_note": "l is padded to %d columns: at the nominal 157.3 TFLOP/s the padded MFMA work alone takes "
                           "%.2f ms per launch" % (-(-l // 32) * 32, 2.0 * mb * n * (-(-l // 32) * 32) / 157.3e12 * 1e3)}
# the core clock the chip holds under each of the two kernels (per-workgroup s_memtime /
# s_memrealtime stamps, dmdx_set_clock_probe): both run the fp32 MFMA pipe while streaming X from
# HBM, and the clock under that load is ~2.0 GHz, not the 2.4 GHz of the nominal peak
lp = -(-l // 32) * 32
Qt = torch.randn((l, n), device="cuda", dtype=torch.float32)
Yb = [torch.randn((l, B.shape[1]), device="cuda", dtype=torch.float32) for B in blocks]


def probed(fn):
    ctr = torch.zeros(3, dtype=torch.int64, device="cuda")
    fn()
    torch.cuda.synchronize()
    kern.clock_probe(ctr)
    try:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
    finally:
        kern.clock_probe(None)
    cyc, ticks, wgs = (int(v) for v in ctr.tolist())
    return 100.0 * cyc / max(ticks, 1), e0.elapsed_time(e1)


for label, fn in (("K2 skinny_kernel (Y = X Q)", lambda: [kern.skinny(B, Qt) for B in blocks]),
                  ("K3 (Z = X^T Y: 64x128-tile generic body, or K3s at l <= 32)", lambda: kern.gemm_tn_blocks(blocks, Yb))):
    if label not in out["kernels"]:
        continue
    mhz, ms_pass = probed(fn)
    bound = 2.0 * m * n * lp / (157.3e12 * mhz / 2400.0) * 1e3
    out["kernels"][label].update({
        "core_clock_mhz": mhz, "ms_per_pass_over_X": ms_pass,
        "padded_mfma_bound_ms_at_held_clock": bound, "frac_of_mfma_bound_at_held_clock": bound / ms_pass,
        "hbm_bound_ms_at_6.3TBps": 4.0 * m * n / 6.3e12 * 1e3})
out["s_head"] = [float(x) for x in res.s[:3].cpu()]
print(json.dumps(out), flush=True)
