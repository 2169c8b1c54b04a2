#!/usr/bin/env python3
"""Headline benchmark: rank-r SVD GB/s on the ERA5 snapshot matrix + fp32 MFMA fraction.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4|small]
                    [--spectrum planted|powerlaw] [--no-hard-spectrum] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workloads (BASELINE.json `configs`), all synthetic on the device, row-centred
with K5, held as row blocks of <= 131072 space points:
  cfg2 (default, the configuration the target is quoted on): X = A diag(sigma) B^T + eps,
        m = 721*1440 = 1 038 240 space points x n = 8760 hourly snapshots, fp32,
        sigma_i = 100*0.9^i (64 terms), eps ~ N(0, 0.01^2), Philox seed 1234 (+rank); rank-50
        "standard" SVD (method of snapshots): Gram (K1) -> top eigenpairs (fp64, K8 / K7) ->
        U = X V S^-1 (K2) -> Rayleigh-Ritz refinement.  With N > 1 every rank holds its own
        1 038 240-row shard of ONE global matrix (weak scaling).
  cfg3: the same generator, 1 946 700 rows per GPU, rank 200 -- at N = 8 this IS config 3
        (15 573 600 x 8760, rank 200, row-sharded, one Gram all-reduce per step); weak scaling.
  cfg4: 15 573 600 x 3653 in total (m / N rows per GPU: strong scaling, "1 vs 8 GPU"),
        randomized SVD with 20 oversamples and 2 power iterations, rank 50 (--rank 200).
  small: 65536 x 1024, smoke only.
The default for --workload can also be set with DMDX_BENCH_WORKLOAD (the driver passes no flags).
A step = one full SVD with X resident in HBM.

Output: ONE JSON line on rank 0 (contract in the task statement) with
  value          = rows of all ranks * n * 4 bytes * K / max-over-ranks wall      [GB/s]
  roofline       = fp32-MFMA roofline of the Gram kernel (standard workloads): algorithmic flops
                   m*n*(n+1) per launch / average launch time measured with HIP events on the
                   launch stream inside the timed region; peak 157.3 TFLOP/s
  hard_spectrum  = (N = 1, cfg2) the same step on the gap-free power-law matrix
                   (make_powerlaw_blocks), outside the timed region: ms per step and the share
                   of the eigen stage, which no longer depends on the spectrum
  cpu_baseline   = (N = 1) the reference's two CPU calls, np.linalg.svd(...)[:r]
                   (era5_svd.py:251) and sklearn randomized_svd(X, r) (era5_svd.py:258), timed
                   on a bounded sample (8760 leading rows x every 2nd column: ~26 s; all columns
                   with --cpu-baseline-full: ~200 s) of the same X on the host cores, plus the HIP
                   engine's singular values of that same sample against the CPU's (parity gate)
  world_size / devices / backend = what the ranks actually saw.
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
    # name: (m, n, rank, description)   -- m = rows per GPU (weak) or in total (strong)
    "cfg2": (721 * 1440, 8760, 50,
             "cfg2: 1038240x8760 fp32 synthetic low-rank(64)+noise, row-centred, rank-50 "
             "method-of-snapshots SVD"),
    "cfg3": (15 * 721 * 1440 // 8, 8760, 200,
             "cfg3: 1946700 rows per GPU x 8760 fp32 (15573600 x 8760 at N = 8), rank-200 "
             "method-of-snapshots SVD, row-sharded, one Gram all-reduce per step"),
    "cfg4": (15 * 721 * 1440, 3653, 50,
             "cfg4: 15573600x3653 fp32 in total (rows / N per GPU), randomized SVD, 20 oversamples, "
             "2 power iterations"),
    "small": (65536, 1024, 50, "small: 65536x1024 fp32 synthetic, rank-50 (smoke only)"),
}
WORKLOAD_KIND = {"cfg2": ("standard", "weak"), "cfg3": ("standard", "weak"), "cfg4": ("randomized", "strong"),
                 "small": ("standard", "weak")}


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
    W = B diag(sigma) H^T, B and H random orthogonal (n x n), sigma_i = 100 / i
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


def calibrate(device, kern=None, blocks=None) -> dict:
    """What this GPU sustains, next to the nominal peaks the roofline is priced against
    (SURVEY.md 8d): a register-only fp32 MFMA loop (dmdx_calib_mfma_f32, 2 waves per SIMD), a
    4 GiB device-to-device copy, and -- so that a slow box can be attributed to its clock -- the
    core clock the chip holds while the Gram kernel itself runs: one extra launch of K1 over
    `blocks` with per-workgroup s_memtime / s_memrealtime stamps (dmdx_set_clock_probe).
    Reported only; `roofline.peak` stays the nominal 157.3 at 2.4 GHz."""
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
    out = {"mfma_f32_tflops_measured": best, "mfma_f32_tflops_nominal": PEAK_FP32_MFMA_TFLOPS,
           "hbm_copy_TBps_measured": copy, "hbm_TBps_nominal": 8.0, "compute_units": cus,
           "how": "dmdx_calib_mfma_f32: 16 x 200000 v_mfma_f32_32x32x2_f32 per wave, 8 waves per CU; "
                  "torch copy of 4 GiB (read + write bytes)"}
    if kern is not None and blocks is not None and len(blocks) > 1:
        ctr = torch.zeros(3, dtype=torch.int64, device=device)
        kern.clock_probe(ctr)
        try:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            kern.syrk_blocks(blocks)
            e1.record()
            e1.synchronize()
        finally:
            kern.clock_probe(None)
        cyc, ticks, wgs = (int(v) for v in ctr.tolist())
        if ticks > 0:
            out["k1_core_clock_mhz"] = 100.0 * cyc / ticks
            out["k1_core_clock_how"] = (f"sum of s_memtime cycles / sum of 100 MHz s_memrealtime ticks over the {wgs} "
                                        f"workgroups of one probed Gram launch ({e0.elapsed_time(e1):.1f} ms with the "
                                        "reduce kernel); the roofline peak assumes 2400 MHz")
            out["k1_peak_at_held_clock_tflops"] = PEAK_FP32_MFMA_TFLOPS * out["k1_core_clock_mhz"] / 2400.0
    return out


def cpu_baseline(Xt: torch.Tensor, r: int, kern, rows: int | None = None, m_full: int | None = None,
                 full_width: bool = False) -> dict:
    """The reference's two CPU calls on a bounded sample of the same X, and the HIP engine on that
    very sample (parity gate).

    Sample: the leading `rows` space points of the first row block x every 2nd column (n_s = 4380
    of cfg2's 8760; `--cpu-baseline-full`: all n columns), rows = 2 n_s by default, copied to the
    host as the F-ordered (rows, n_s) fp32 array the reference hands to LAPACK (svd_on_era5,
    era5_svd.py:246).  Timed: oracle.svd_standard = np.linalg.svd(X, full_matrices=False) + slice
    (era5_svd.py:251-254) and sklearn's randomized_svd(X, r) with the reference's defaults
    (era5_svd.py:258).  gesdd costs ~6 rows n^2 + 20 n^3 flops: the full-width sample (17520 x
    8760) takes 204 s on the box's 128 host threads (profiles/r2_bench_cpu_full.json), the
    smallest tall full-width one (rows = n) still ~165 s, so the default run -- which has to end
    within minutes -- halves the width (~1/8 of the work per row: ~26 s) and says so;
    `extrapolated_full_seconds` scales the measured time by (m / rows) (n / n_s)^2, linear in the
    rows and quadratic in the columns: the m n^2 term of gesdd, a LOWER bound of what the
    reference's call on the whole 1 038 240 x 8760 matrix costs (and it needs > 3 |X| = 110 GB
    of host RAM there)."""
    import resource

    from oracle import era5_oracle as orc

    n_all, m = Xt.shape
    cs = 1 if (full_width or n_all < 4096) else 2
    n = len(range(0, n_all, cs))
    ms = min(m, rows or 2 * n)
    sample_dev = Xt[::cs, :ms].contiguous()
    Xs = sample_dev.cpu().numpy().T                                # (ms, n) F-order view, as the reference's X
    try:
        from threadpoolctl import threadpool_info

        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    U, s, V = orc.svd_standard(Xs, r)
    dt = time.perf_counter() - t0
    out = {
        "value": Xs.nbytes / dt / 1e9,
        "unit": "GB/s",
        "cores": int(cores),
        "kind": "port",
        "seconds": dt,
        "sample": f"oracle.svd_standard (np.linalg.svd(X, full_matrices=False)[:r], the reference's "
                  f"era5_svd.py:251 call) on rows[0:{ms}] x {'all' if cs == 1 else f'every {cs}nd of the'} {n_all} columns "
                  f"= {ms}x{n} fp32 F-order of the same X ({Xs.nbytes / 1e9:.2f} GB)",
        "extrapolated_full_seconds": dt * ((m_full or m) / ms) * (n_all / n) ** 2,
        "extrapolation": f"measured seconds x ({m_full or m} / {ms} rows) x ({n_all} / {n} columns)^2: the m n^2 term "
                         "of gesdd only, a lower bound for the whole matrix",
        "s_head": [float(x) for x in s[:3]],
    }
    try:
        from sklearn.utils.extmath import randomized_svd

        t0 = time.perf_counter()
        _, s_r, _ = randomized_svd(Xs, r, random_state=0)
        dtr = time.perf_counter() - t0
        out["randomized"] = {"value": Xs.nbytes / dtr / 1e9, "unit": "GB/s", "seconds": dtr,
                             "call": "sklearn.utils.extmath.randomized_svd(X, r) with the reference's defaults "
                                     "(era5_svd.py:258; random_state=0 here)",
                             "max_rel_diff_s_vs_standard": float(np.max(np.abs(s_r - s) / s))}
    except Exception as e:  # sklearn missing on the box: say so, do not fail the bench
        out["randomized"] = {"error": repr(e)}
    out["peak_rss_gb"] = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6
    # parity gate: the HIP engine on the very same sample
    from dmd_era5_amd import svd as dsvd

    res = dsvd.svd_snapshots(sample_dev, r, kern=kern)
    s_gpu = res.s.cpu().numpy()
    out["parity"] = {"max_rel_err_s": float(np.max(np.abs(s_gpu - s.astype(np.float64)) / s.astype(np.float64))),
                     "min_abs_cos_u": float(np.min(np.abs(np.sum(res.Ut.cpu().numpy().T.astype(np.float64)
                                                                 * U.astype(np.float64), axis=0)))),
                     "what": f"svd_snapshots (HIP) vs np.linalg.svd (fp32 LAPACK) on the same {ms}x{n} sample, "
                             f"{r} singular values / left vectors"}
    return out


def hard_spectrum(m: int, n: int, r: int, device, kern, steps: int = 2) -> dict:
    """The standard path on the gap-free power-law matrix of the same size (N = 1 only, outside
    the timed region): ms per step and the stage split.  ERA5 anomalies look like this, not like
    the planted rank-64 + noise matrix whose eigenproblem three power steps finish."""
    from dmd_era5_amd import svd as dsvd

    blocks = make_powerlaw_blocks(m, n, 1234, device)
    for Xb in blocks:
        kern.row_center_scale_(Xb, False)
    dsvd.svd_snapshots(blocks, r, kern=kern)
    torch.cuda.synchronize()
    acc: dict[str, float] = {}
    t0 = time.perf_counter()
    for _ in range(steps):
        res = dsvd.svd_snapshots(blocks, r, kern=kern, timings=True)
        for k in ("t_gram", "t_eig", "t_project", "t_refine", "t_total"):
            acc[k] = acc.get(k, 0.0) + res.info[k]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"spectrum": "sigma_i = 100 / i over all n columns (make_powerlaw_blocks)", "steps": steps,
           "ms_per_step": dt * 1e3, "value_GBps": m * n * 4.0 / dt / 1e9,
           "gram_ms": acc["t_gram"] / steps * 1e3, "eig_ms": acc["t_eig"] / steps * 1e3,
           "project_ms": acc["t_project"] / steps * 1e3, "refine_ms": acc["t_refine"] / steps * 1e3,
           "eig_share": acc["t_eig"] / acc["t_total"],
           "eig_method": res.info.get("eig_method"), "eig_products": res.info.get("eig_products"),
           "eig_degrees": res.info.get("eig_degrees"), "eig_block": res.info.get("eig_block"),
           "sweeps": res.info.get("eig_outer_iters"),
           "s_head": [float(x) for x in res.s[:3].cpu()]}
    del blocks
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("DMDX_BENCH_WORKLOAD", "cfg2"), choices=sorted(WORKLOADS))
    ap.add_argument("--rank", type=int, default=None, help="override the workload's rank (cfg4: 50 or 200)")
    ap.add_argument("--spectrum", default="planted", choices=["planted", "powerlaw"],
                    help="powerlaw: run the timed steps themselves on the gap-free matrix")
    ap.add_argument("--no-hard-spectrum", action="store_true",
                    help="skip the extra power-law steps reported under `hard_spectrum` (N = 1, cfg2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-rows", type=int, default=None, help="rows of the CPU sample (default 2 n_s)")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="CPU sample over ALL n columns (17520 x 8760 at cfg2: ~200 s of np.linalg.svd)")
    ap.add_argument("--no-calibrate", action="store_true",
                    help="skip the ~0.3 s on-box micro-benchmarks (register-only fp32 MFMA loop, HBM copy, K1 clock)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
        args.gpus = world
    # DMDX_BENCH_DEVICE / DMDX_DIST_BACKEND: rehearsal knobs (several ranks on ONE GPU over gloo,
    # to exercise the N > 1 code path on a single-GPU box); DMDX_BENCH_FORCE_DIST=1: take the
    # torch.distributed path with ONE rank (the only way RCCL itself runs on a one-GPU box: every
    # collective of the step goes through the nccl backend).  The driver's runs use none of them.
    dev_index = int(os.environ.get("DMDX_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)

    from dmd_era5_amd import svd as dsvd
    from dmd_era5_amd.kernels import default_kernels

    kern = default_kernels()  # fails loudly if libdmdx.so is missing
    backend = None
    devices = [{"rank": 0, "device_index": dev_index, "name": torch.cuda.get_device_name(device)}]
    if world > 1 or os.environ.get("DMDX_BENCH_FORCE_DIST") == "1":
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
        mine = {"rank": rank, "device_index": dev_index, "name": torch.cuda.get_device_name(device),
                "pci_bus_id": getattr(torch.cuda.get_device_properties(device), "pci_bus_id", None)}
        devices = [None] * world
        dist.all_gather_object(devices, mine)
        backend = f"{dist.get_backend()} (torch.distributed; on ROCm 'nccl' is RCCL)"
    else:
        dist = None
        comm = dsvd.Comm()

    m, n, r, desc = WORKLOADS[args.workload]
    svd_type, scaling = WORKLOAD_KIND[args.workload]
    svd_type = os.environ.get("DMDX_BENCH_SVD_TYPE", svd_type)     # rehearsal knob (tests)
    if args.rank:
        r = args.rank
    if scaling == "strong":                       # rows of ONE matrix split over the ranks
        cuts = [m * i // world for i in range(world + 1)]
        m_total, m = m, cuts[rank + 1] - cuts[rank]
    else:
        m_total = m * world
    if dist is not None and svd_type == "standard":
        # the first LARGE all-reduce pays RCCL's channel / buffer set-up: prime it at the size of the
        # packed Gram triangle, outside the timed region whatever --warmup is
        dist.all_reduce(torch.zeros(n * (n + 1) // 2, dtype=torch.float64, device=device))
        torch.cuda.synchronize()
    gen = make_powerlaw_blocks if args.spectrum == "powerlaw" else make_snapshot_blocks
    blocks = gen(m, n, 1234 if args.workload != "cfg4" else 99, device, shard=rank)
    for Xb in blocks:
        kern.row_center_scale_(Xb, False)
    torch.cuda.synchronize()

    if svd_type == "standard":
        def step():
            return dsvd.svd_snapshots(blocks, r, comm=comm, kern=kern)
    else:
        def step():
            return dsvd.svd_randomized(blocks, r, n_oversamples=20, n_iter=2, random_state=0, comm=comm, kern=kern)

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
    # per-launch HIP events: ~20 launches per standard step (free); the randomized path issues
    # thousands of launches per step, where recording them would be the thing measured
    kern.events = [] if svd_type == "standard" else None
    issued = comm.n_collectives
    if hasattr(comm, "start_timing"):
        comm.start_timing()          # one HIP event pair per collective (6 per standard step)
    # per step: a HIP event at its start / end on the launch stream (no host synchronisation) and
    # the clock-probe counters of its stamped kernels (K1 / K2 / K3: three atomics per workgroup)
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    clk = torch.zeros((max(args.steps, 1), 3), dtype=torch.int64, device=device)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step_ev[i].record()
        kern.clock_probe(clk[i])
        res = step()
    step_ev[args.steps].record()
    kern.clock_probe(None)
    barrier()
    dt = time.perf_counter() - t0
    issued = (comm.n_collectives - issued) / max(args.steps, 1)
    coll = comm.stop_timing() if hasattr(comm, "stop_timing") else {}
    events, kern.events = (kern.events or []), None
    step_ms = [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps)]
    step_clock = [100.0 * c / t if t > 0 else None for c, t, _ in clk.tolist()]
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # per-kernel times from the HIP events recorded inside the timed region
    # (the Gram of the row blocks is ONE launch of the batched K1, dmdx_syrk_blocks_f32,
    # followed by its reduce kernel on the same stream; both are inside the event pair)
    by_name: dict[str, list[float]] = {}
    for name, shape, e0, e1 in events:
        key = name
        if name in ("syrk", "syrk_blocks"):
            key = "syrk" if shape[1] == n else "syrk_small"
        by_name.setdefault(key, []).append(e0.elapsed_time(e1))
    nblk = len(blocks)

    out = {
        "metric": "rank-r SVD GB/s on ERA5 snapshot matrix (X resident in HBM)",
        "value": m_total * n * 4.0 * args.steps / dt / 1e9,
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": desc + (" [power-law spectrum]" if args.spectrum == "powerlaw" else ""),
            "m_per_gpu": m, "m_total": m_total, "n": n, "rank": r, "svd_type": svd_type,
            "sharding": (f"rows x{world}, one packed-triangle Gram all-reduce per step" if svd_type == "standard"
                         else f"rows x{world}, n x l and l x l all-reduces per pass") if world > 1 else "none",
        },
        "world_size": world,
        "backend": backend,
        "collectives_per_step": issued,     # what this rank handed to torch.distributed inside the timed region
        # wall time of this rank's collectives inside the timed region, per step, by kind (HIP events
        # around each call on the launch stream; gloo: host clock): so that a multi-GPU line explains itself
        "collective_ms": {k: {"calls_per_step": v["calls"] / max(args.steps, 1), "ms_per_step": v["ms"] / max(args.steps, 1),
                              "bytes_per_step": v["bytes"] / max(args.steps, 1)} for k, v in coll.items()},
        "devices": devices,
        "step_ms": step_ms,                 # every timed step (HIP events on the launch stream, rank 0)
        "step_core_clock_mhz": step_clock,  # core clock held by the stamped kernels (K1 / K2 / K3) of each step
    }
    if svd_type == "standard":
        syrk_ms = float(np.mean(by_name["syrk"]))            # average Gram launch (all row blocks of X)
        flops = float(m) * n * (n + 1)                         # algorithmic flops of that launch
        achieved = flops / (syrk_ms * 1e-3) / 1e12
        # HBM-side traffic of the Gram kernel: PMC numbers cannot be collected from inside this
        # process; they come from the committed rocprofv3 --pmc passes of this same command
        # (profiles/r2_bench_rocprof_summary.json, else round 1's: FETCH_SIZE doubled per the gfx950
        # note of MI355X_MICROARCH.md, plus WRITE_SIZE, per launch).
        traffic, traffic_src = None, None
        if args.workload == "cfg2" and args.spectrum == "planted":
            for name in ("r3_bench_rocprof_summary.json", "r2_bench_rocprof_summary.json", "r1_bench_rocprof_summary.json"):
                try:
                    with open(os.path.join(ROOT, "profiles", name)) as f:
                        traffic = float(json.load(f)["traffic"]["bytes_per_launch_corrected"])
                    traffic_src = name
                    break
                except Exception:
                    continue
        per_launch = [float(x) for x in by_name["syrk"]]      # one Gram launch per timed step, in order
        fr = [flops / (x * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS for x in per_launch]
        nf = min(5, len(fr))
        clocks = [c for c in step_clock if c]
        out["roofline"] = {
            "bound": "mfma",
            "kernel": "syrk_batch_kernel (K1 Gram of all row blocks in one launch, dmdx_syrk_blocks_f32)",
            "achieved": achieved,
            "peak": PEAK_FP32_MFMA_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
            "traffic": traffic,
            "traffic_unit": f"bytes/launch (L2<->fabric, PMC; profiles/{traffic_src})" if traffic_src else None,
            "flops_per_launch": flops,
            "ms_per_launch": syrk_ms,
            # what the kernel SUSTAINS: `frac` is the mean over the timed steps; the chip warms up over
            # the first ~20 steps (it gives clock back), so a long run ends lower than it starts
            "frac_first5": float(np.mean(fr[:nf])), "frac_last5": float(np.mean(fr[-nf:])),
            "frac_min": float(min(fr)), "frac_max": float(max(fr)),
            "core_clock_mhz_first5": float(np.mean(clocks[:nf])) if clocks else None,
            "core_clock_mhz_last5": float(np.mean(clocks[-nf:])) if clocks else None,
            # the same launches priced at the clock the chip actually held (nominal: 2400 MHz)
            "frac_at_held_clock": float(np.mean([f * 2400.0 / c for f, c in zip(fr, step_clock) if c])) if clocks else None,
        }
    else:
        # randomized: 2 n_iter + 2 = 6 passes over X of 2 m n l flops each (K2 / K3 alternate); the
        # per-kernel split lives in profiles/ (rocprofv3 --kernel-trace of scripts/bench_cfg4.py)
        l = r + 20
        flops = 6 * 2.0 * m * n * l
        out["roofline"] = {"bound": "mfma", "kernel": "K2 skinny16 (Y = X Q, fused Gram) + K3 gemm_tn (Z = X^T Y), 6 passes over X",
                           "achieved": flops * args.steps / dt / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": flops * args.steps / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                           "traffic": None, "flops_per_step": flops,
                           "note": "whole-step algorithmic rate of ONE rank's shard (per GPU; no per-launch "
                                   "events in the timed region)"}
        # the phases of one more step, outside the timed region (every phase boundary synchronises)
        ph = dsvd.svd_randomized(blocks, r, n_oversamples=20, n_iter=2, random_state=0, comm=comm, kern=kern, timings=True).info
        out["stage_ms"] = {k: float(v) for k, v in ph.get("phase_ms", {}).items()}
        out["stage_ms"]["total"] = float(ph["t_total"]) * 1e3
    if svd_type == "standard":
        # stage split of one more step, outside the timed region (the stage timers synchronise)
        # (the minimum of three calls per stage: a single call once showed a 41 ms `refine` on the
        # driver's box where every other run has 2.3 ms -- a 208 MB allocation served by the driver
        # instead of the caching allocator; all three calls are listed)
        sts = [dsvd.svd_snapshots(blocks, r, comm=comm, kern=kern, timings=True).info for _ in range(3)]
        keys = ["t_gram", "t_eig", "t_project", "t_refine", "t_total"] + (["t_polish"] if "t_polish" in sts[0] else [])
        out["stage_ms"] = {k[2:]: float(min(st[k] for st in sts)) * 1e3 for k in keys}
        out["stage_ms_calls"] = [{k[2:]: float(st[k]) * 1e3 for k in keys} for st in sts]
    out["kernel_ms_per_step"] = {k: float(np.sum(v)) / args.steps for k, v in by_name.items()}
    out["row_blocks"] = nblk
    out["svd_info"] = {k: (float(v) if isinstance(v, (int, float)) else v)
                       for k, v in (res.info if res is not None else {}).items()}
    out["s_head"] = [float(x) for x in res.s[:3].cpu()] if res is not None else None
    if rank == 0 and world == 1 and not args.no_calibrate:
        out["calibration"] = calibrate(device, kern, blocks if svd_type == "standard" else None)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(blocks[0], r, kern, args.cpu_baseline_rows, m, args.cpu_baseline_full)
        # The full-width sample SURVEY.md 8(d) asks for (all n columns: 17520 x 8760 at cfg2) costs ~7.4 x
        # the half-width one (203.8 s against 27.8 s on the 128-thread box).  The default run has to
        # end within minutes and its CPU leg is meant to be 10-30 s of work, so the full-width sample
        # is taken by default only when the half-width timing predicts <= 2 minutes for it (a faster
        # host than the 128-thread boxes seen so far); `--cpu-baseline-full` (or
        # DMDX_BENCH_CPU_FULL=2) takes it regardless -- profiles/r2_bench_cpu_full.json is such a run.
        half = out["cpu_baseline"]
        if (not args.cpu_baseline_full and args.workload == "cfg2" and os.environ.get("DMDX_BENCH_CPU_FULL", "1") != "0"
                and args.cpu_baseline_rows is None
                and (7.5 * half["seconds"] <= 120.0 or os.environ.get("DMDX_BENCH_CPU_FULL") == "2")):
            full = cpu_baseline(blocks[0], r, kern, None, m, True)
            full["half_width_sample"] = {k: half[k] for k in ("value", "seconds", "sample", "parity", "randomized") if k in half}
            out["cpu_baseline"] = full
    if rank == 0 and world == 1 and not args.no_hard_spectrum and args.workload == "cfg2" and args.spectrum == "planted":
        del blocks
        kern.release_workspace()
        torch.cuda.empty_cache()
        out["hard_spectrum"] = hard_spectrum(m, n, r, device, kern)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
