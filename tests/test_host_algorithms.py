"""CPU tests of the host-side algorithms (dmd_era5_amd.svd) with the kernel layer
replaced by the CPU test double -- including the row-sharded multi-rank path over
gloo with world_size 2."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from kernel_double import CpuKernelDouble
from oracle import era5_oracle as orc
from parity_utils import col_cosines, sv_tolerance

from dmd_era5_amd import svd as dsvd

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
K = CpuKernelDouble()


def _xt(X):
    return torch.from_numpy(np.ascontiguousarray(X.T, dtype=np.float32))


def test_snapshots_matches_golden():
    g = np.load(os.path.join(GOLDEN, "lowrank_4096x192.npz"))
    X = orc.lowrank_matrix(4096, 192, 100, 0)
    for refine in (True, False):
        r = dsvd.svd_snapshots(_xt(X), 50, refine=refine, kern=K)
        assert np.all(np.abs(r.s.numpy() - g["s64"]) <= sv_tolerance(g["s64"]))
        assert col_cosines(r.Ut.numpy().T, g["U64"]).min() > 1 - 1e-5
        assert col_cosines(r.Vh.numpy().T, g["V64"].T).min() > 1 - 1e-5


def _psd_with_spectrum(n, lam, seed):
    rs = np.random.RandomState(seed)
    Qm, _ = np.linalg.qr(rs.standard_normal((n, n)))
    G = (Qm * lam) @ Qm.T
    return torch.from_numpy(0.5 * (G + G.T))


@pytest.mark.parametrize("l,power", [(20, 2.0), (62, 2.0), (40, 1.0)])
def test_top_eigh_cheb_equals_full_on_gap_free_spectra(l, power):
    """lambda_i ~ i^-2 (singular values ~ 1/i: what ERA5 anomalies look like) and the slower
    i^-1: no gap behind any block, so the power steps cannot finish and the Chebyshev-filtered
    iteration must -- with a handful of products, not by falling back to the full solver."""
    n = 1500
    G = _psd_with_spectrum(n, 1.0 / np.arange(1, n + 1) ** power, 0)
    info = {}
    lam_k, V_k = dsvd.top_eigh(G, l, method="cheb", info=info, kern=K)
    lam_f, V_f = dsvd.top_eigh(G, l, method="full")
    assert info["eig_method"] == "cheb" and info["eig_residual"] <= 1e-9
    assert info["eig_products"] <= (16 if power == 2.0 else 30) and info["eig_block"] == 2 * l
    assert torch.allclose(lam_k, lam_f, rtol=1e-10)
    # vectors: residual 1e-9 lambda_1 over a gap of ~2 lambda_l / l
    assert (V_k * V_f).sum(dim=0).abs().min() > 1 - 1e-6
    assert float((V_k.T @ V_k - torch.eye(l, dtype=torch.float64)).abs().max()) < 1e-12


def test_top_eigh_krylov_is_an_alias_of_cheb():
    rs = np.random.RandomState(0)
    A = rs.standard_normal((3000, 1200)) * (0.99 ** np.arange(1200))
    G = torch.from_numpy(A.T @ A)
    info = {}
    lam_k, V_k = dsvd.top_eigh(G, 20, method="krylov", info=info, kern=K)
    lam_f, V_f = dsvd.top_eigh(G, 20, method="full")
    assert info["eig_method"] == "cheb"
    assert torch.allclose(lam_k, lam_f, rtol=1e-10)
    assert (V_k * V_f).sum(dim=0).abs().min() > 1 - 1e-8
    with pytest.raises(ValueError):
        dsvd.top_eigh(G, 20, method="lanczos")


def test_orth_survives_blocks_beyond_choleskyqr():
    """G times a random block of a low-rank + noise matrix spans lambda_1 / lambda_noise ~ 1e12:
    the plain Gram route fails there, the shifted round must take over (not Householder QR)."""
    rs = np.random.RandomState(3)
    n, b = 600, 40
    Qm, _ = np.linalg.qr(rs.standard_normal((n, b)))
    Y = torch.from_numpy(Qm * np.logspace(0, -12, b)) @ torch.from_numpy(rs.standard_normal((b, b)))
    Q = dsvd._orth(Y)
    assert float((Q.T @ Q - torch.eye(b, dtype=torch.float64)).abs().max()) < 1e-10
    # same span as far as fp64 resolves it: the leading directions of Y are inside span(Q)
    lead = torch.from_numpy(Qm[:, :8])
    assert float((lead - Q @ (Q.T @ lead)).abs().max()) < 1e-6


def test_top_eigh_power_fast_path_on_lowrank_plus_noise():
    """A steep drop behind the block (cfg2's spectrum shape) is finished by the (b x b) power /
    Rayleigh-Ritz steps; a slowly decaying one (the tests above) goes on to the filtered iteration."""
    rs = np.random.RandomState(1)
    A = rs.standard_normal((4000, 24)) * (100 * 0.9 ** np.arange(24))   # rank 24 < block width 28
    X = A @ rs.standard_normal((24, 500)) + 1e-3 * rs.standard_normal((4000, 500))
    G = torch.from_numpy(X.T @ X)
    info = {}
    lam_p, V_p = dsvd.top_eigh(G, 20, method="cheb", info=info, kern=K)
    lam_f, V_f = dsvd.top_eigh(G, 20, method="full")
    assert info["eig_method"] == "power" and info["eig_residual"] <= 1e-11
    assert torch.allclose(lam_p, lam_f, rtol=1e-10)
    assert (V_p * V_f).sum(dim=0).abs().min() > 1 - 1e-8


def test_top_eigh_flat_spectrum_goes_to_the_full_solver_early():
    """Pure noise has no spectral gap: the Ritz values forecast far more filter steps than the
    full solver costs at this size, so the answer comes from eigh at once."""
    rs = np.random.RandomState(2)
    A = rs.standard_normal((6000, 1800))
    G = torch.from_numpy(A.T @ A)
    info = {}
    lam_a, V_a = dsvd.top_eigh(G, 20, method="cheb", info=info, kern=K)
    lam_f, V_f = dsvd.top_eigh(G, 20, method="full")
    assert info["eig_method"] == "full" and info["eig_cheb_forecast_steps"] > 18 and info["eig_products"] <= 16
    assert torch.equal(lam_a, lam_f) and torch.equal(V_a, V_f)


@pytest.mark.parametrize("shape", [(9, 2), (500, 30)])
def test_all_zero_matrix_gives_zero_singular_values(shape):
    """X = 0 (found by the sparse kind of the differential fuzz): s = 0, U and V orthonormal (round 3:
    directions below the resolution of the data get an orthonormal completion, as LAPACK's U for
    zero singular values; they were zero columns before), for both types -- the range finder's
    CholeskyQR has nothing to factor and must not raise."""
    m, n = shape
    k = min(n, 5)
    for fn in (dsvd.svd_snapshots, dsvd.svd_randomized):
        kw = {"random_state": 0} if fn is dsvd.svd_randomized else {}
        r = fn(torch.zeros((n, m)), k, kern=K, **kw)
        assert r.s.shape == (k,) and float(r.s.abs().max()) == 0.0
        assert bool(torch.isfinite(r.Ut).all()) and bool(torch.isfinite(r.Vh).all())
        assert torch.allclose(r.Ut.double() @ r.Ut.double().T, torch.eye(k, dtype=torch.float64), atol=1e-5)
        assert torch.allclose(r.Vh @ r.Vh.T, torch.eye(k, dtype=torch.float64), atol=1e-12)


def test_graded_refinement_matrix_keeps_relative_accuracy_without_the_jacobi_kernel():
    """T = S M S with s over 7 decades: beyond K7's size the eigenpairs come from the SVD of
    L^T S (M = L L^T), accurate relative to EACH eigenvalue (a library eigh of T only promises
    eps * s_1^2: rocSOLVER's syevd left U orthonormal to 1e-3 at cfg3's rank 200; host LAPACK
    happens to do better on such matrices) -- checked against a 50-digit mpmath solution."""
    mpmath = pytest.importorskip("mpmath")
    rs = np.random.RandomState(7)
    l = 36
    E = rs.standard_normal((l, l)) * 1e-3
    Mn = np.eye(l) + 0.5 * (E + E.T)
    sn = np.logspace(0, -7, l)
    mpmath.mp.dps = 50
    Tm = mpmath.matrix(l, l)
    for i in range(l):
        for j in range(l):
            Tm[i, j] = mpmath.mpf(float(sn[i])) * mpmath.mpf(float(Mn[i, j])) * mpmath.mpf(float(sn[j]))
    exact = np.array(sorted((float(x) for x in mpmath.eigsy(Tm, eigvals_only=True)), reverse=True))
    M, s0 = torch.from_numpy(Mn), torch.from_numpy(sn)
    mu_s, Z_s = dsvd._graded_eigh(s0, M, None)               # no small-eigh provider: the SVD route
    assert np.max(np.abs(mu_s.numpy() - exact) / exact) < 1e-9
    T = s0[:, None] * M * s0[None, :]
    assert float(((T @ Z_s - Z_s * mu_s).norm(dim=0) / mu_s).max()) < 1e-6     # residuals relative to EACH eigenvalue
    # directions dropped from S come back as zero eigenvalues with unit eigenvectors
    s1 = s0.clone()
    s1[-3:] = 0.0
    mu_z, Z_z = dsvd._graded_eigh(s1, M, None)
    assert torch.all(mu_z[-3:] == 0) and torch.allclose(mu_z[:-3], dsvd._graded_eigh(s0[:-3], M[:-3, :-3], None)[0])
    assert torch.allclose(Z_z.T @ Z_z, torch.eye(l, dtype=torch.float64), atol=1e-12)


def test_randomized_same_omega_as_sklearn():
    g = np.load(os.path.join(GOLDEN, "lowrank_4096x192.npz"))
    X = orc.lowrank_matrix(4096, 192, 100, 0)
    r = dsvd.svd_randomized(_xt(X), 50, random_state=0, kern=K)
    assert r.info["n_iter"] == 4 and r.info["l"] == 60
    assert np.allclose(r.s.numpy(), g["rdef_s"], rtol=2e-5)
    assert np.all(np.sum(r.Ut.numpy().T * g["rdef_U"], axis=0) > 0.9999)
    r2 = dsvd.svd_randomized(_xt(X), 50, omega=g["rdef_omega"].astype(np.float64), kern=K)
    assert torch.allclose(r2.s, r.s, rtol=1e-6)


def test_n_iter_auto_rule():
    assert dsvd.resolve_n_iter(10, 5184, 24) == 4      # k >= 0.1 * min(m, n)
    assert dsvd.resolve_n_iter(50, 1038240, 8760) == 7
    assert dsvd.resolve_n_iter(50, 100, 100, n_iter=2) == 2


@pytest.mark.parametrize("d", [2, 3])
def test_delay_embedding_paths(d):
    X = orc.lowrank_matrix(1500, 60, 30, 4)
    Xe = orc.delay_embed(X, d)
    Ue, se, Ve = orc.svd_standard(Xe.astype(np.float64), 8)
    Ue, Ve = orc.svd_flip(Ue, Ve)
    r = dsvd.svd_snapshots(_xt(X), 8, delay=d, kern=K)
    assert r.Ut.shape == (8, d * 1500) and r.Vh.shape == (8, 60 - d + 1)
    assert np.allclose(r.s.numpy(), se, rtol=1e-6)
    assert np.all(np.sum(r.Ut.numpy().T * Ue, axis=0) > 0.9999)   # row order k*m + s
    assert torch.equal(dsvd.embed_view(_xt(X), d), torch.from_numpy(np.ascontiguousarray(Xe.T)))


def test_row_blocks_do_not_change_the_answer(monkeypatch):
    X = orc.lowrank_matrix(4096, 96, 40, 2)
    base = dsvd.svd_snapshots(_xt(X), 12, delay=2, kern=K)
    monkeypatch.setattr(dsvd, "BLOCK_ROWS", 500)
    blk = dsvd.svd_snapshots(_xt(X), 12, delay=2, kern=K)
    assert blk.info["row_blocks"] == 9
    assert torch.allclose(blk.s, base.s, rtol=1e-10)
    assert torch.allclose(blk.Ut, base.Ut, atol=1e-6)
    assert dsvd.split_rows(1038240, 131072) == [(i * 129780, (i + 1) * 129780) for i in range(8)]


def test_rank_deficient_input_gives_zero_tail():
    rs = np.random.RandomState(1)
    X = (rs.standard_normal((500, 3)) @ rs.standard_normal((3, 40))).astype(np.float32)
    r = dsvd.svd_snapshots(_xt(X), 6, kern=K)
    s = r.s.numpy()
    assert np.all(s[:3] > 1) and np.all(s[3:] < 1e-3 * s[0])
    assert np.isfinite(r.Ut.numpy()).all()


# ---------------------------------------------------------------- world_size 2 over gloo
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _sharded_case(svd_type):
    """(X, k): 'steep' has s_k / s_1 ~ 1e-5 on top of 1e-7 noise, which sends svd_snapshots through
    its polish step (a K2 + K3 pass with its own all-reduces)."""
    if svd_type == "steep":
        rs = np.random.RandomState(5)
        A = np.linalg.qr(rs.standard_normal((3000, 24)))[0] * (100 * 0.55 ** np.arange(24))
        X = A @ np.linalg.qr(rs.standard_normal((96, 24)))[0].T + 1e-7 * rs.standard_normal((3000, 96))
        return X.astype(np.float32), 20
    return orc.lowrank_matrix(3000, 96, 40, 2), 10


def _worker(rank, world, port, svd_type, q):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, k = _sharded_case(svd_type)
        rows = np.array_split(np.arange(3000), world)[rank]
        comm = dsvd.TorchDistComm()
        Xt = _xt(X[rows])
        if rank == 0:   # ranks need not hold the same number of row blocks (one exchange per rank, not per block)
            Xt = [Xt[:, :400].contiguous(), Xt[:, 400:401].contiguous(), Xt[:, 401:].contiguous()]
        if svd_type in ("standard", "steep"):
            r = dsvd.svd_snapshots(Xt, k, comm=comm, kern=K)
        else:
            r = dsvd.svd_randomized(Xt, k, random_state=0, comm=comm, kern=K)
        q.put((rank, r.s.numpy(), r.Ut.numpy(), r.Vh.numpy(), bool(r.info.get("polished"))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("svd_type", ["standard", "randomized", "steep"])
def test_row_sharded_two_ranks_equal_single_rank(svd_type):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, svd_type, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X, k = _sharded_case(svd_type)
    if svd_type in ("standard", "steep"):
        ref = dsvd.svd_snapshots(_xt(X), k, kern=K)
    else:
        ref = dsvd.svd_randomized(_xt(X), k, random_state=0, kern=K)
    U = np.concatenate([o[2] for o in out], axis=1)            # shards are contiguous row ranges
    assert np.allclose(out[0][1], out[1][1])                   # s replicated
    if svd_type == "steep":
        assert out[0][4] and out[1][4] and ref.info.get("polished")
        s64 = np.linalg.svd(X.astype(np.float64), compute_uv=False)[:k]
        assert np.allclose(out[0][1], s64, rtol=0, atol=2e-6 * s64[0])
        assert np.allclose(out[0][1], ref.s.numpy(), rtol=0, atol=1e-7 * s64[0])
        good = s64 > 1e-4 * s64[0]                             # vectors above the fp32 noise of X itself
        assert np.all(np.abs(np.sum(U * ref.Ut.numpy(), axis=1))[good] > 1 - 1e-5)
        return
    assert np.allclose(out[0][1], ref.s.numpy(), rtol=1e-8)
    assert np.allclose(out[0][3], ref.Vh.numpy(), atol=1e-7)
    assert np.allclose(U, ref.Ut.numpy(), atol=1e-6)           # includes the global sign flip


def test_non_finite_input_raises_like_numpy():
    """np.linalg.svd (the reference's call) raises LinAlgError on NaN input; so does the engine,
    instead of iterating on a Gram matrix full of NaNs."""
    X = orc.lowrank_matrix(512, 40, 10, 1)
    X[17, 3] = np.nan
    with pytest.raises(np.linalg.LinAlgError, match="did not converge"):
        dsvd.svd_snapshots(_xt(X), 5, kern=K)


def test_uncentred_data_takes_the_mean_deflated_route():
    """Temperature-like data with its time mean left in (s_1 ~ 3e4 s_2): the engine detects the
    dominant mean and deflates it exactly (centred Gram + Schur complement of the mean
    direction); the result is the SVD of the UN-centred matrix (golden: numpy fp64), and the caller's X is left untouched."""
    g = np.load(os.path.join(GOLDEN, "conditioning_2048x160.npz"))
    k = int(g["k"])
    for tag, expect in (("raw", True), ("cen", False)):
        X = g[f"{tag}_X"]
        Xt = _xt(X)
        keep = Xt.clone()
        r = dsvd.svd_snapshots(Xt, k, kern=K)
        assert bool(r.info.get("mean_deflated", False)) is expect
        assert torch.allclose(Xt, keep, rtol=0, atol=1e-4)           # restored (fp32 add of the mean back)
        assert np.abs(r.s.numpy() / g[f"{tag}_s64"] - 1).max() < 2e-6
        assert col_cosines(r.Ut.numpy().T, g[f"{tag}_U64"]).min() > 1 - 1e-5
        assert col_cosines(r.Vh.numpy().T, g[f"{tag}_V64"].T).min() > 1 - 1e-5
        Urec = (r.Ut.numpy().T * r.s.numpy()) @ r.Vh.numpy()
        top = (g[f"{tag}_U64"] * g[f"{tag}_s64"]) @ g[f"{tag}_V64"]
        assert np.abs(Urec - top).max() < 1e-3 * np.abs(top).max()


@pytest.mark.parametrize("d", [2, 3])
def test_uncentred_data_with_delay_embedding(d):
    """Mean deflation under delay embedding: the embedding of un-centred data is the embedding of
    the centred data plus (tiled mean) 1^T; the ones vector is no longer in the null space of the
    centred Gram (window sums of a centred series are small, not zero), which the projected form
    P A^T A P handles exactly.  Truth: numpy fp64 SVD of the materialised embedding."""
    g = np.load(os.path.join(GOLDEN, "conditioning_2048x160.npz"))
    X = g["raw_X"][:600]
    Xe = orc.delay_embed(X, d)
    Ue, se, Ve = orc.svd_standard(Xe.astype(np.float64), 8)
    r = dsvd.svd_snapshots(_xt(X), 8, delay=d, kern=K)
    assert r.info.get("mean_deflated")
    assert np.abs(r.s.numpy() / se - 1).max() < 2e-6
    assert col_cosines(r.Ut.numpy().T, Ue).min() > 1 - 1e-5
    assert col_cosines(r.Vh.numpy().T, Ve.T).min() > 1 - 1e-5


@pytest.mark.parametrize("scale", [1e22, 1e-24])
def test_extreme_magnitudes_are_rescaled(scale):
    """|x| ~ 1e22 (squares overflow fp32) / 1e-24 (they underflow): scaled by a power of two for
    the factorisation, like LAPACK's xLASCL; the caller's matrix comes back bit-identical."""
    X = (orc.lowrank_matrix(700, 48, 20, 2).astype(np.float64) * scale).astype(np.float32)
    sref = np.linalg.svd(X.astype(np.float64), compute_uv=False)[:6]
    for fn, kw in ((dsvd.svd_snapshots, {}), (dsvd.svd_randomized, {"random_state": 0})):
        Xt = _xt(X)
        keep = Xt.clone()
        r = fn(Xt, 6, kern=K, **kw)
        assert "rescaled_by" in r.info and torch.equal(Xt, keep)
        assert np.allclose(r.s.numpy(), sref, rtol=1e-5)


def test_orth_on_exactly_rank_deficient_blocks_ends_in_householder():
    """A constant matrix (found by the GPU suite: RecursionError): every Gram, shifted or not,
    stays singular behind the shifted round -- the fallback must end, not recurse."""
    Y = torch.ones((50, 6), dtype=torch.float64)
    Q = dsvd._orth(Y)
    assert Q.shape == (50, 6) and bool(torch.isfinite(Q).all())
    one = Y[:, :1] / Y[:, :1].norm()
    assert float((one - Q @ (Q.T @ one)).abs().max()) < 1e-12          # the one direction Y has is in span(Q)
    Z = torch.zeros((40, 3), dtype=torch.float64)
    assert bool(torch.isfinite(dsvd._orth(Z)).all())


@pytest.mark.parametrize("d", [1, 2])
def test_streaming_standard_path_polishes_steep_spectra_in_one_extra_pass(d):
    """s_k / s_1 = 1e-6: below what the Gram of fp32 products resolves (1e-9 lambda_1).  The
    resident path runs a polish step (one subspace iteration on X itself); the streamed path used
    to warn and return the unresolved directions -- it now takes one more pass over the pieces and
    must agree with the resident result and with numpy's fp64 SVD of the same fp32 data."""
    rs = np.random.RandomState(5)
    m, n, k = 3000, 64, 7
    A, _ = np.linalg.qr(rs.standard_normal((m, k)))
    B, _ = np.linalg.qr(rs.standard_normal((n, k)))
    X = ((A * np.logspace(0, -6, k)) @ B.T).astype(np.float32)
    X += (1e-9 * rs.standard_normal((m, n))).astype(np.float32)
    Xt = _xt(X)                                                 # (n, m)
    cuts = [0, 700, 1500, 2300, 3000]

    def pieces():
        for a, b in zip(cuts[:-1], cuts[1:]):
            yield [Xt[:, a:(a + b) // 2].contiguous(), Xt[:, (a + b) // 2:b].contiguous()]

    if d == 1:
        ref = np.linalg.svd(X.astype(np.float64), compute_uv=False)[:k]
    else:
        ref = np.linalg.svd(orc.delay_embed(X.astype(np.float64), d), compute_uv=False)[:k]
        # the streamed pieces embed per block: compare against the resident path on the same blocks
    blocks = [b for p in pieces() for b in p]
    res = dsvd.svd_snapshots(blocks, k, delay=d, kern=K)
    Ub, s, Vh, info = dsvd.svd_snapshots_streaming(pieces, k, d * m, delay=d, kern=K)
    assert bool(info.get("polished")) == bool(res.info.get("polished"))
    assert info["passes_over_X"] == (3 if info.get("polished") else 2)
    if d == 1:
        assert info.get("polished")
    assert np.allclose(s.numpy(), res.s.numpy(), rtol=1e-6, atol=1e-9 * float(res.s[0]))
    U = torch.cat([u for p in Ub for u in p], dim=1)
    assert float((U @ U.T - torch.eye(k)).abs().max()) < 1e-4
    cos = (Vh * res.Vh).sum(dim=1).abs()
    assert float(cos.min()) > 1 - 1e-6
    if d == 1:
        assert np.abs(s.numpy() - ref).max() <= 2e-7 * ref[0]


def test_explicit_normalizer_none_is_sklearns_arithmetic():
    """`power_iteration_normalizer="none"` given explicitly is taken literally (sklearn's
    un-normalised power iterations, extmath.py:314-316 / 349-351), unlike "auto", where the engine
    keeps normalising also at n_iter <= 2.  On a gently decaying spectrum, where fp32 carries the
    un-normalised iterate, both must agree with the oracle's restatement of sklearn run the same
    way on the same Omega."""
    rs = np.random.RandomState(11)
    m, n, k = 3000, 80, 5
    A, _ = np.linalg.qr(rs.standard_normal((m, n)))
    B, _ = np.linalg.qr(rs.standard_normal((n, n)))
    X = ((A * (1.0 / (1.0 + np.arange(n)))) @ B.T).astype(np.float32)
    omega = np.random.RandomState(0).normal(size=(n, k + 5))
    Uo, so, Vo = orc.svd_randomized(X.astype(np.float64), k, n_oversamples=5, n_iter=2,
                                    power_iteration_normalizer="none", omega=omega)
    r = dsvd.svd_randomized(_xt(X), k, n_oversamples=5, n_iter=2, power_iteration_normalizer="none",
                            omega=omega, kern=K)
    assert r.info["normalizer"] == "none"
    assert np.allclose(r.s.numpy(), so, rtol=2e-5)
    assert col_cosines(r.Ut.numpy().T, Uo).min() > 1 - 1e-6
    ra = dsvd.svd_randomized(_xt(X), k, n_oversamples=5, n_iter=2, omega=omega, kern=K)   # "auto": normalised
    assert np.allclose(ra.s.numpy(), so, rtol=2e-5)                      # same subspaces in exact arithmetic


def test_powerlaw_generator_has_the_spectrum_it_claims():
    """bench.make_powerlaw_blocks (the hard-spectrum leg of bench.py and the gap-free GPU test):
    singular values sigma_i sqrt(m) within the Marchenko-Pastur edge factors, every consecutive
    ratio close to (i + 1) / i -- no gap anywhere -- and the same time factor for every shard."""
    import bench

    m, n = 6000, 48
    blocks = bench.make_powerlaw_blocks(m, n, 7, torch.device("cpu"))
    X = torch.cat(blocks, dim=1).numpy().T.astype(np.float64)           # (m, n)
    s = np.linalg.svd(X, compute_uv=False)
    expect = 100.0 / np.arange(1, n + 1) * np.sqrt(m)
    assert np.all(np.abs(s / expect - 1) < 3.5 * np.sqrt(n / m))
    assert np.all(s[:-1] / s[1:] < 2.2) and np.all(s[:-1] / s[1:] > 1.0)
    other = bench.make_powerlaw_blocks(m, n, 7, torch.device("cpu"), shard=1)
    Y = torch.cat(other, dim=1).numpy().T.astype(np.float64)
    assert not np.allclose(X, Y)
    # one global matrix: stacking the shards keeps the spectrum (x sqrt(2)), it does not double the rank
    s2 = np.linalg.svd(np.concatenate([X, Y]), compute_uv=False)
    assert np.all(np.abs(s2 / (expect * np.sqrt(2)) - 1) < 3.5 * np.sqrt(n / m))


@pytest.mark.parametrize("l,n,decades", [(20, 300, 3), (60, 500, 7), (7, 7, 2)])
def test_svd_wide_matches_lapack(l, n, decades):
    """The thin SVD of the randomized path's l x n factor B (CholeskyQR2 + one-sided Jacobi of the
    l x l factor, svd._svd_wide) against LAPACK: singular values to 1e-12 relative, Uhat S Vh = B,
    orthonormal factors; a rank-deficient B goes to the library."""
    import torch

    from dmd_era5_amd import svd as dsvd
    from kernel_double import CpuKernelDouble as NumpyKernels

    rs = np.random.RandomState(l + n)
    U0, _ = np.linalg.qr(rs.standard_normal((l, l)))
    V0, _ = np.linalg.qr(rs.standard_normal((n, l)))
    s0 = np.logspace(0, -decades, l) * 5e3
    B = torch.from_numpy((U0 * s0) @ V0.T)
    Uh, s, Vh = dsvd._svd_wide(B.clone(), NumpyKernels())
    assert np.allclose(s.numpy(), s0, rtol=1e-12)
    assert torch.linalg.norm(Uh * s @ Vh - B) <= 1e-13 * torch.linalg.norm(B)
    assert torch.linalg.norm(Uh.T @ Uh - torch.eye(l, dtype=torch.float64)) < 1e-12
    assert torch.linalg.norm(Vh @ Vh.T - torch.eye(l, dtype=torch.float64)) < 1e-12
    # rank-deficient: two equal rows
    B2 = B.clone()
    B2[-1] = B2[0]
    Uh, s, Vh = dsvd._svd_wide(B2.clone(), NumpyKernels())
    ref = np.linalg.svd(B2.numpy(), compute_uv=False)
    assert np.allclose(s.numpy()[:-1], ref[:-1], rtol=1e-10) and s[-1] <= 1e-10 * s[0]
    assert torch.linalg.norm(Uh * s @ Vh - B2) <= 1e-12 * torch.linalg.norm(B2)
