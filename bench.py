#!/usr/bin/env python3
"""Headline benchmark: rank-r SVD GB/s on the ERA5 snapshot matrix + fp32 MFMA fraction.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|small]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1] / SURVEY.md section 8d, "cfg2"): synthetic
X = A diag(sigma) B^T + eps on the device, m = 721*1440 = 1 038 240 space points x
n = 8760 hourly snapshots, fp32, sigma_i = 100*0.9^i (64 terms), eps ~ N(0, 0.01^2),
Philox seed 1234 (+rank), held as 8 row blocks of 129 780 space points, row-centred
with K5; rank-50 "standard" SVD (method of
snapshots): Gram (K1) -> top eigenpairs (fp64) -> U = X V S^-1 (K2) -> Rayleigh-Ritz
refinement.  A step = one full SVD with X resident in HBM.  With N > 1 every rank
holds its own 1 038 240-row shard (weak scaling, config-3 style) and the only
exchange is the all-reduce of the n x n Gram (+ one l x l) over RCCL.

Output: ONE JSON line on rank 0 (contract in the task statement) with
  value        = N * m * n * 4 bytes * K / wall      [GB/s]
  roofline     = fp32-MFMA roofline of the Gram kernel: algorithmic flops m*n*(n+1)
                 per launch / average launch time measured with HIP events on the
                 launch stream inside the timed region; peak 157.3 TFLOP/s
  cpu_baseline = the oracle's `svd_standard` (np.linalg.svd + slice == the reference's
                 call, era5_svd.py:251) timed on a bounded sample of the same X on
                 the host cores (N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 256 flop/clk x 2.4 GHz

WORKLOADS = {
    # name: (m, n, rank, description)
    "cfg2": (721 * 1440, 8760, 50,
             "cfg2: 1038240x8760 fp32 synthetic low-rank(64)+noise, row-centred, rank-50 "
             "method-of-snapshots SVD"),
    "small": (65536, 1024, 50, "small: 65536x1024 fp32 synthetic, rank-50 (smoke only)"),
}


def make_snapshot_blocks(m: int, n: int, seed: int, device, shard: int = 0) -> list[torch.Tensor]:
    """The snapshot matrix as row (space) blocks, each an (n, mb) fp32 device tensor
    (= the block's X^T), generated in time-slabs.  Row blocks keep the
    column stride short (TLB reach, see dmd_era5_amd/svd.py).

    ``shard``: row shard of ONE global matrix X = A diag(sigma) B^T + noise: the time factor B
    (n x 64) comes from ``seed`` on every rank, the space factor A and the noise of this shard
    from ``seed`` and ``shard`` -- the stacked matrix keeps rank 64 + noise whatever the number
    of ranks (independent B per rank would make it rank 64 N, a different problem per N)."""
    from dmd_era5_amd.svd import split_rows

    g = torch.Generator(device=device).manual_seed(seed)
    rank = 64
    B = torch.randn((n, rank), generator=g, device=device, dtype=torch.float32)
    if shard:
        g = torch.Generator(device=device).manual_seed(seed + 1000003 * shard)
    sig = 100.0 * 0.9 ** torch.arange(rank, device=device, dtype=torch.float32)
    Bs = B * sig
    blocks = []
    step = 1024
    for r0, r1 in split_rows(m):
        A = torch.randn((r1 - r0, rank), generator=g, device=device, dtype=torch.float32)
        Xb = torch.empty((n, r1 - r0), device=device, dtype=torch.float32)
        for j0 in range(0, n, step):
            j1 = min(n, j0 + step)
            blk = Xb[j0:j1]
            torch.matmul(Bs[j0:j1], A.T, out=blk)
            blk.add_(torch.randn(blk.shape, generator=g, device=device, dtype=torch.float32),
                     alpha=0.01)
        blocks.append(Xb)
    return blocks


def make_powerlaw_blocks(m: int, n: int, seed: int, device, shard: int = 0) -> list[torch.Tensor]:
    """The gap-free counterpart of :func:`make_snapshot_blocks` (ERA5 anomalies have power-law
    spectra, not a gap behind the wanted rank): X = X0 W with X0 (m x n) Gaussian noise and
    W = B diag(sigma) H^T / sqrt(m_total-ish), B and H random orthogonal (n x n), sigma_i = 100 / i
    over ALL n columns.  X0^T X0 / m = I + O(sqrt(n/m)), so the singular values of X are
    sigma_i sqrt(m) within the Marchenko-Pastur edge factors 1 +- sqrt(n/m) and every consecutive
    ratio is (i+1)/i: no gap anywhere.  Row blocks as (n, mb) fp32 tensors; the time factor W is
    the same for every shard (one global matrix, as in make_snapshot_blocks)."""
    from dmd_era5_amd.svd import split_rows

    g = torch.Generator(device=device).manual_seed(seed)
    Bq, _ = torch.linalg.qr(torch.randn((n, n), generator=g, device=device, dtype=torch.float32))
    Hq, _ = torch.linalg.qr(torch.randn((n, n), generator=g, device=device, dtype=torch.float32))
    sig = 100.0 / torch.arange(1, n + 1, device=device, dtype=torch.float32)
    Wt = (Hq * sig) @ Bq.T                      # (n, n) = W^T:  X_b^T = W^T X0_b^T
    del Bq, Hq
    if shard:
        g = torch.Generator(device=device).manual_seed(seed + 1000003 * shard)
    blocks = []
    for r0, r1 in split_rows(m):
        X0t = torch.randn((n, r1 - r0), generator=g, device=device, dtype=torch.float32)
        blocks.append(Wt @ X0t)
        del X0t
    return blocks


def calibrate(device) -> dict:
    """What this GPU sustains, next to the nominal peaks the roofline is priced against
    (SURVEY.md 8d): a register-only fp32 MFMA loop (dmdx_calib_mfma_f32, 2 waves per SIMD) and a
    4 GiB device-to-device copy.  Reported only; `roofline.peak` stays the nominal 157.3."""
    import ctypes as C

    from dmd_era5_amd import _lib

    lib = _lib.load()
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    sink = torch.zeros(4, dtype=torch.float32, device=device)
    flops = C.c_double(0.0)
    stream = torch.cuda.current_stream(device).cuda_stream
    best = 0.0
    for it in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.dmdx_calib_mfma_f32(200000, cus, sink.data_ptr(), C.byref(flops), stream), "calib")
        e1.record()
        e1.synchronize()
        if it:
            best = max(best, flops.value / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    a = torch.empty(1 << 30, dtype=torch.float32, device=device)
    b = torch.empty_like(a)
    copy = 0.0
    for it in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        e1.synchronize()
        if it:
            copy = max(copy, 2.0 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    del a, b
    return {"mfma_f32_tflops_measured": best, "mfma_f32_tflops_nominal": PEAK_FP32_MFMA_TFLOPS,
            "hbm_copy_TBps_measured": copy, "hbm_TBps_nominal": 8.0, "compute_units": cus,
            "how": "dmdx_calib_mfma_f32: 16 x 200000 v_mfma_f32_32x32x2_f32 per wave, 8 waves per CU; "
                   "torch copy of 4 GiB (read + write bytes)"}


def cpu_baseline(Xt: torch.Tensor, r: int) -> dict:
    """Oracle (np.linalg.svd + slice) on a bounded sample (leading rows of the first
    row block) of the same matrix."""
    from oracle import era5_oracle as orc

    n, m = Xt.shape
    cs = 4 if n >= 4096 else 1
    ms = min(m, 32445 if n >= 4096 else 8192)
    sample = Xt[::cs, :ms].contiguous().cpu().numpy()          # (n_s, m_s) C-order
    Xs = sample.T                                              # (m_s, n_s) F-order, as the reference's X
    try:
        from threadpoolctl import threadpool_info

        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    U, s, V = orc.svd_standard(Xs, r)
    dt = time.perf_counter() - t0
    return {
        "value": Xs.nbytes / dt / 1e9,
        "unit": "GB/s",
        "cores": int(cores),
        "kind": "port",
        "seconds": dt,
        "sample": f"oracle.svd_standard (np.linalg.svd + slice, the reference's era5_svd.py:251 "
                  f"call) on rows[0:{ms}] x every {cs}th column = {Xs.shape[0]}x{Xs.shape[1]} fp32 "
                  f"F-order of the same X; LAPACK cost per byte grows ~linearly with n, so the "
                  f"full-width rate is ~{cs}x lower",
        "s_head": [float(x) for x in s[:3]],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-calibrate", action="store_true",
                    help="skip the ~0.3 s on-box micro-benchmarks (register-only fp32 MFMA loop, HBM copy)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
        args.gpus = world
    # DMDX_BENCH_DEVICE / DMDX_DIST_BACKEND: rehearsal knobs (several ranks on ONE GPU over gloo,
    # to exercise the N > 1 code path on a single-GPU box); the driver's runs use neither.
    dev_index = int(os.environ.get("DMDX_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)

    from dmd_era5_amd import svd as dsvd
    from dmd_era5_amd.kernels import default_kernels

    kern = default_kernels()  # fails loudly if libdmdx.so is missing
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DMDX_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        comm = dsvd.TorchDistComm()
        # bring the RCCL communicator up outside the timed region whatever --warmup is
        dist.all_reduce(torch.zeros(1, device=device))
        torch.cuda.synchronize()
    else:
        dist = None
        comm = dsvd.Comm()

    m, n, r, desc = WORKLOADS[args.workload]
    blocks = make_snapshot_blocks(m, n, 1234, device, shard=rank)
    for Xb in blocks:
        kern.row_center_scale_(Xb, False)
    torch.cuda.synchronize()

    def step():
        return dsvd.svd_snapshots(blocks, r, comm=comm, kern=kern)

    # library handles / code objects (rocBLAS, rocSOLVER, libdmdx) are created on first use: prime
    # them on a toy problem so that --warmup 0 does not time their initialisation
    toy = torch.randn((2048, 16384), device=device, dtype=torch.float32)
    dsvd.svd_snapshots([toy[:, :8192].contiguous(), toy[:, 8192:].contiguous()], 8, kern=kern)
    del toy
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    res = None
    for _ in range(args.warmup):
        res = step()
    barrier()
    kern.events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    dt = time.perf_counter() - t0
    events, kern.events = kern.events, None
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # per-kernel times from the HIP events recorded inside the timed region
    # (the Gram is one launch per row block: per-launch figures are sums over the blocks
    # of one step divided by the launches; flops likewise)
    # (the Gram of the row blocks is ONE launch of the batched K1, dmdx_syrk_blocks_f32,
    # followed by its reduce kernel on the same stream; both are inside the event pair)
    by_name: dict[str, list[float]] = {}
    for name, shape, e0, e1 in events:
        key = name
        if name in ("syrk", "syrk_blocks"):
            key = "syrk" if shape[1] == n else "syrk_small"
        by_name.setdefault(key, []).append(e0.elapsed_time(e1))
    nblk = len(blocks)
    syrk_ms = float(np.mean(by_name["syrk"]))            # average Gram launch (all row blocks of X)
    flops = float(m) * n * (n + 1)                         # algorithmic flops of that launch
    achieved = flops / (syrk_ms * 1e-3) / 1e12

    # HBM-side traffic of the Gram kernel: PMC numbers cannot be collected from inside this
    # process; they come from the committed rocprofv3 --pmc passes of this same command
    # (profiles/r1_bench_rocprof_summary.json: FETCH_SIZE doubled per the gfx950 note of
    # MI355X_MICROARCH.md, plus WRITE_SIZE, per launch).
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r1_bench_rocprof_summary.json")) as f:
            prof = json.load(f)
        if args.workload == "cfg2":
            traffic = float(prof["traffic"]["bytes_per_launch_corrected"])
    except Exception:
        traffic = None

    out = {
        "metric": "rank-r SVD GB/s on ERA5 snapshot matrix (X resident in HBM)",
        "value": world * m * n * 4.0 * args.steps / dt / 1e9,
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": desc,
            "m_per_gpu": m, "n": n, "rank": r, "svd_type": "standard",
            "sharding": f"rows x{world}, one Gram all-reduce per step" if world > 1 else "none",
        },
        "roofline": {
            "bound": "mfma",
            "kernel": "syrk_batch_kernel (K1 Gram of all row blocks in one launch, dmdx_syrk_blocks_f32)",
            "achieved": achieved,
            "peak": PEAK_FP32_MFMA_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
            "traffic": traffic,
            "traffic_unit": "bytes/launch (L2<->fabric, PMC; profiles/r1_bench_rocprof_summary.json)",
            "flops_per_launch": flops,
            "ms_per_launch": syrk_ms,
        },
        "kernel_ms_per_step": {k: float(np.sum(v)) / args.steps for k, v in by_name.items()},
        "row_blocks": nblk,
        "svd_info": {k: (float(v) if isinstance(v, (int, float)) else v)
                     for k, v in (res.info if res is not None else {}).items()},
        "s_head": [float(x) for x in res.s[:3].cpu()] if res is not None else None,
    }
    if rank == 0 and world == 1 and not args.no_calibrate:
        out["calibration"] = calibrate(device)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(blocks[0], r)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
