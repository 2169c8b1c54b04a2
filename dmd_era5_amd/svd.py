"""Rank-r SVD of the (space x time) snapshot matrix on MI355X.

Replaces the two CPU calls of the reference's ``svd_on_era5``
(/root/reference/src/dmd_era5/era5_svd/era5_svd.py:249-259):

* ``svd_type == "standard"``  (reference: ``np.linalg.svd`` then slice, :251-254)
  -> :func:`svd_snapshots`: method of snapshots.  One pass of X through the
  fp32-MFMA Gram kernel (K1), the small eigenproblem of the n x n Gram in
  fp64, one pass of X through the tall-skinny projection kernel (K2), and a
  Rayleigh-Ritz refinement on the projected basis (a Gram of the m x l basis,
  l ~ r, which restores the relative accuracy of the small singular values).
* ``svd_type == "randomized"`` (reference: sklearn ``randomized_svd`` with all
  defaults, :258; algorithm extmath.py:287-357,531-604) -> :func:`svd_randomized`:
  the same range finder / power iterations, with the tall LU/QR normalisers
  replaced by CholeskyQR (an l x l Gram + triangular solve) -- the iterates
  span the same subspaces, so U, s, V agree up to rounding for the same Omega.

Data layout: see :mod:`dmd_era5_amd.kernels` -- the snapshot matrix is the
``(time, space)`` fp32 tensor ``Xt``; with row sharding every rank holds
``Xt[:, rows_of_this_rank]`` and the only exchanges are sum-all-reduces of
small matrices (n x n Gram, n x l, l x l) through ``comm``.

Everything dense and small (eigh / cholesky / qr / svd of matrices with at most
n ~ 10^4 rows) runs in fp64 through torch on the same device.
"""

from __future__ import annotations

import math
import os
import time
from dataclasses import dataclass, field

import numpy as np
import torch

__all__ = [
    "Comm",
    "TorchDistComm",
    "SvdResult",
    "embed_view",
    "as_blocks",
    "split_rows",
    "top_eigh",
    "svd_snapshots",
    "svd_snapshots_streaming",
    "svd_randomized_streaming",
    "svd_randomized",
]


# ---------------------------------------------------------------------------
# communication (row shards): sum all-reduce of small dense matrices only
# ---------------------------------------------------------------------------
class Comm:
    """Single-rank communicator (no-op)."""

    world_size = 1
    rank = 0
    exchanges = False   # whether the collectives are issued (TorchDistComm: world_size > 1)
    n_collectives = 0   # collectives issued so far (bench.py reports them per step)
    timed = None        # TorchDistComm.start_timing(): list of (tag, bytes, start, stop) per collective

    def allreduce_sum_(self, t: torch.Tensor, tag: str = "allreduce") -> torch.Tensor:
        return t

    def allgather(self, t: torch.Tensor, tag: str = "allgather") -> list[torch.Tensor]:
        return [t]

    def broadcast_(self, *tensors: torch.Tensor, tag: str = "broadcast"):
        """Make rank 0's copy of small replicated results (eigenvectors, rotations)
        authoritative: every rank computes them from identical all-reduced inputs, but a
        last-bit or sign difference between devices would make the U shards inconsistent."""
        return tensors if len(tensors) != 1 else tensors[0]

    def gather_to_root(self, t: torch.Tensor, tag: str = "gather") -> list[torch.Tensor] | None:
        """Every rank's tensor (shapes may differ in any dimension) as a list of HOST tensors on
        rank 0, None elsewhere: the result assembly for the NetCDF write -- U, the row means
        and, if asked for, X leave the devices only here."""
        return [t.cpu()]


class TorchDistComm(Comm):
    """torch.distributed communicator: backend "nccl" is RCCL over xGMI on ROCm;
    "gloo" is used by the CPU tests."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self._dist = dist
        self._group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # DMDX_COMM_FORCE=1: issue every collective with ONE rank too -- the way RCCL itself gets
        # exercised on a one-GPU box (tests/test_gpu_pipeline.py); never set by the product
        self.exchanges = self.world_size > 1 or os.environ.get("DMDX_COMM_FORCE") == "1"

    # -- per-collective wall time (bench.py: `collective_ms`; so that the first multi-GPU run explains
    # itself).  Device tensors: a HIP event pair on the current stream around the call (the call
    # returns once the collective is enqueued; its completion is ordered before the second event);
    # host tensors (gloo): the host clock.
    def start_timing(self) -> None:
        self.timed = []

    def stop_timing(self) -> dict:
        """-> {tag: {"calls", "ms", "bytes"}} of everything issued since start_timing()."""
        out: dict = {}
        for tag, nbytes, t0, t1 in (self.timed or []):
            ms = t0.elapsed_time(t1) if hasattr(t0, "elapsed_time") else (t1 - t0) * 1e3
            d = out.setdefault(tag, {"calls": 0, "ms": 0.0, "bytes": 0})
            d["calls"] += 1
            d["ms"] += float(ms)
            d["bytes"] += int(nbytes)
        self.timed = None
        return out

    def _run(self, tag: str, tensors, fn):
        if self.timed is None:
            return fn()
        nbytes = sum(int(t.numel()) * t.element_size() for t in tensors)
        if tensors and tensors[0].is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn()
            e1.record()
        else:
            e0 = time.perf_counter()
            r = fn()
            e1 = time.perf_counter()
        self.timed.append((tag, nbytes, e0, e1))
        return r

    def allreduce_sum_(self, t: torch.Tensor, tag: str = "allreduce") -> torch.Tensor:
        if self.exchanges:
            self.n_collectives += 1
            self._run(tag, [t], lambda: self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self._group))
        return t

    def allgather(self, t: torch.Tensor, tag: str = "allgather") -> list[torch.Tensor]:
        if not self.exchanges:
            return [t]
        out = [torch.empty_like(t) for _ in range(self.world_size)]
        self.n_collectives += 1
        tc = t.contiguous()
        self._run(tag, [tc], lambda: self._dist.all_gather(out, tc, group=self._group))
        return out

    def broadcast_(self, *tensors: torch.Tensor, tag: str = "broadcast"):
        if self.exchanges:
            if len(tensors) > 1 and len({(t.dtype, t.device) for t in tensors}) == 1:
                # ONE collective for the lot (eigenvalues + eigenvectors, rotation + values)
                flat = torch.cat([t.reshape(-1) for t in tensors])
                self.n_collectives += 1
                self._run(tag, [flat], lambda: self._dist.broadcast(flat, src=0, group=self._group))
                off = 0
                for t in tensors:
                    t.copy_(flat[off:off + t.numel()].view(t.shape))
                    off += t.numel()
            else:
                for t in tensors:
                    self.n_collectives += 1
                    self._run(tag, [t], lambda t=t: self._dist.broadcast(t, src=0, group=self._group))
        return tensors if len(tensors) != 1 else tensors[0]

    GATHER_BUDGET = 1 << 30   # bytes of receive buffer on the root per collective

    def gather_to_root(self, t: torch.Tensor, tag: str = "gather") -> list[torch.Tensor] | None:
        """One `gather` collective per GATHER_BUDGET of receive buffer instead of world_size - 1
        serial send / recv pairs (round 3): every rank's tensor is flattened and padded to the
        longest; the root receives all pieces of a chunk at once (RCCL moves device buffers, gloo
        host buffers) and copies them into its host tensors.  U of a cfg3 shard (1.6 GB per rank) is
        13 collectives at N = 8; a `save_data_matrix` X never needs more than the budget on the
        root's device."""
        if not self.exchanges:
            return [t.cpu()]
        dist = self._dist
        on_host = dist.get_backend(self._group) == "gloo"
        t = t.contiguous()
        if t.dim() > 8:
            raise ValueError("gather_to_root: at most 8 dimensions")
        dims = torch.full((9,), -1, dtype=torch.int64, device="cpu" if on_host else t.device)
        dims[0] = t.dim()
        dims[1:1 + t.dim()] = torch.tensor(t.shape, dtype=torch.int64)
        shapes = [tuple(int(v) for v in d[1:1 + int(d[0])].tolist()) for d in self.allgather(dims, tag=tag + "_shapes")]
        counts = [int(np.prod(sh)) if len(sh) else 1 for sh in shapes]
        longest = max(counts)
        root = self.rank == 0
        host = [torch.empty(c, dtype=t.dtype) for c in counts] if root else None
        flat = t.reshape(-1).cpu() if on_host else t.reshape(-1)
        dev = flat.device
        chunk = max(1, self.GATHER_BUDGET // (self.world_size * t.element_size()))
        for off in range(0, longest, chunk):
            n = min(chunk, longest - off)
            piece = flat[off:off + n]
            if piece.numel() < n:        # shorter ranks pad (the root drops the padding)
                piece = torch.cat([piece, torch.zeros(n - piece.numel(), dtype=flat.dtype, device=dev)])
            recv = [torch.empty(n, dtype=flat.dtype, device=dev) for _ in range(self.world_size)] if root else None
            self.n_collectives += 1
            self._run(tag, [piece], lambda: dist.gather(piece.contiguous(), recv, dst=0, group=self._group))
            if root:
                for r in range(self.world_size):
                    m = min(n, counts[r] - off)
                    if m > 0:
                        host[r][off:off + m] = recv[r][:m].cpu()
        if not root:
            return None
        return [host[r].reshape(shapes[r]) for r in range(self.world_size)]


@dataclass
class SvdResult:
    Ut: torch.Tensor  # (k, M_local) fp32: row j = left singular vector j (local rows)
    s: torch.Tensor   # (k,) fp64
    Vh: torch.Tensor  # (k, n_eff) fp64: row j = right singular vector j
    info: dict = field(default_factory=dict)


def _kern(kern):
    if kern is not None:
        return kern
    from .kernels import default_kernels

    return default_kernels()


def embed_view(Xt: torch.Tensor, d: int) -> torch.Tensor:
    """Zero-copy delay embedding (reference slice_tools.py:207-211): row
    ``k*m + s`` of the embedded matrix is ``X[s, t+k]``."""
    if d == 1:
        return Xt
    n, m = Xt.shape
    if not Xt.is_contiguous():
        raise ValueError("delay embedding view needs a contiguous (time, space) tensor")
    if d > n:
        raise ValueError(f"delay embedding {d} exceeds the number of snapshots {n}")
    return Xt.as_strided((n - d + 1, d * m), (m, 1))


# ---------------------------------------------------------------------------
# small dense pieces (fp64, torch)
# ---------------------------------------------------------------------------
def _tn(kern, A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """A^T B for tall fp64 blocks: K9 (fp64 MFMA, ~20 us) when the provider has it, else the
    library GEMM (~250 us at 8760 x 124: tall-skinny shapes are not rocBLAS's case)."""
    f = getattr(kern, "gemm_tn64", None) if kern is not None else None
    return f(A, B) if f is not None else A.T @ B


def _chol(kern, A: torch.Tensor, shift=0.0, want_inv: bool = True):
    """A + shift I = L L^T for a small symmetric fp64 matrix -> (L, Linv or None, info) with ``info`` a
    3-vector of device doubles (status: 0 ok; min, max of diag(L)) that the caller reads ONCE,
    after queueing what follows.  One launch of K10 (kernels.chol_inv: Cholesky + triangular
    inverse, n <= 1024) when the provider has it -- rocSOLVER's potrf / trtri are launch-rate bound
    at these sizes and cost 0.3-0.8 s of library start-up in a CLI process --, else the library."""
    n = A.shape[0]
    f = getattr(kern, "chol_inv", None) if kern is not None else None
    if f is not None and A.is_cuda and n <= getattr(kern, "chol_max_n", 0):
        return f(A, shift=float(shift), want_inv=want_inv)
    As = A if (isinstance(shift, float) and shift == 0.0) else A + shift * torch.eye(n, dtype=A.dtype, device=A.device)
    L, err = torch.linalg.cholesky_ex(As)
    diag = torch.diagonal(L)
    fin = torch.isfinite(diag).all()
    info = torch.stack([torch.where(fin, err.to(A.dtype).reshape(()), torch.ones((), dtype=A.dtype, device=A.device)),
                        diag.min(), diag.max()])
    Linv = torch.linalg.solve_triangular(L, torch.eye(n, dtype=A.dtype, device=A.device), upper=False) if want_inv else None
    return L, Linv, info


def _chol_ok(info) -> bool:
    """One read-back of a factorisation's status (see :func:`_chol`)."""
    st, dmin, dmax = info.tolist()
    return st == 0.0 and math.isfinite(dmin) and math.isfinite(dmax) and dmin > 0.0


def _apply_nt(kern, Q: torch.Tensor, Mt: torch.Tensor) -> torch.Tensor:
    """Q Mt^T for a tall fp64 block: K11 (fp64 MFMA, one launch) when the provider has it."""
    f = getattr(kern, "gemm_nt64", None) if kern is not None else None
    if f is not None and Q.is_cuda:
        return f(Q if Q.stride(1) == 1 else Q.contiguous(), Mt if Mt.stride(1) == 1 else Mt.contiguous())
    return Q @ Mt.T


def _orth(Y: torch.Tensor, rounds: int = 2, _shifted: bool = False, kern=None) -> torch.Tensor:
    """Orthonormal basis of the columns of a tall fp64 block.  CholeskyQR2 (two rounds of
    Gram (K9) -> Cholesky + triangular inverse (K10) -> Q L^-T (K11): three launches of our own per
    round; round 2 went through rocSOLVER potrf + rocBLAS trsm, Householder QR costs ~9 ms).  A
    block too ill-conditioned for
    the plain Gram route (cond > ~1e8: G times a random block of a low-rank + noise matrix spans
    lambda_1 / lambda_noise) first gets ONE *shifted* round (Fukaya et al. 2020: factor
    Y^T Y + s I, s ~ 1e-11 |Y|^2, which caps the conditioning of what the plain rounds then see);
    if the plain rounds fail again behind it (an exactly rank-deficient block: a constant
    matrix) Householder QR takes over.  ``rounds=1``: conditioning control between the factors
    of a polynomial filter (orthogonal to ~cond^2 eps), not an orthonormal basis."""
    Q = Y
    for it in range(rounds):
        G = _tn(kern, Q, Q)
        L, Linv, info = _chol(kern, G)
        # the product is queued BEFORE the host looks at the factorisation's status (one read-back
        # of three numbers): the device works on while the host waits for it
        Qn = _apply_nt(kern, Q, Linv)
        if not _chol_ok(info):
            if it == 0 and not _shifted:
                tr = float(torch.diagonal(G).sum())
                for rel in (1e-11, 1e-8):
                    L, Linv, info = _chol(kern, G, shift=rel * tr)
                    if _chol_ok(info) and bool(torch.isfinite(Linv).all()):
                        Qs = _apply_nt(kern, Q, Linv)
                        return _orth(Qs, rounds=max(rounds, 2) + (1 if rel == 1e-8 else 0), _shifted=True, kern=kern)
            Qh, _ = torch.linalg.qr(Y, mode="reduced")
            return Qh
        Q = Qn
    return Q.contiguous()


def _jacobi_pd_eigh(C_t: torch.Tensor, kern):
    """Eigenpairs (descending) of T = C C^T from the one-sided Jacobi SVD of C (K7L), C given
    column by column (row c of ``C_t`` = column c of C): (sigma^2, Z)."""
    sig, Zt = kern.svd_jacobi(C_t)
    return sig * sig, Zt.T


def _eigh_desc(T: torch.Tensor, kern=None):
    """Eigenpairs of a small symmetric fp64 matrix, eigenvalues descending, eigenvectors in
    columns.  Up to the provider's ``eigh_small_max_n`` (96) this is ONE launch of the two-sided
    Jacobi kernel K7; up to ``svd_jacobi_max_n`` (1024) one launch of the one-sided Jacobi kernel
    K7L on the Cholesky factor of T (the Rayleigh-Ritz matrices of a PSD Gram matrix are positive
    definite up to its rounding; if the factorisation fails T is shifted by 1e-8 of its mean
    diagonal first -- the vectors do not change, the values are shifted back, which costs them
    nothing at the absolute accuracy a Rayleigh-Ritz step needs).  rocSOLVER's syevd
    (torch.linalg.eigh) is launch-rate bound at these sizes and remains the fallback."""
    n = T.shape[0]
    if kern is not None and n <= getattr(kern, "eigh_small_max_n", 0):
        try:
            return kern.eigh_small(T)
        except RuntimeError:
            pass
    elif kern is not None and n <= getattr(kern, "svd_jacobi_max_n", 0):
        shift = 0.0
        L, _, info = _chol(kern, T, want_inv=False)
        ok = _chol_ok(info)
        if not ok:
            shift = 1e-8 * float(torch.diagonal(T).abs().mean())
            L, _, info = _chol(kern, T, shift=shift, want_inv=False)
            ok = _chol_ok(info)
        if ok:
            try:
                lam, Z = _jacobi_pd_eigh(L.T.contiguous(), kern)
                return lam - shift, Z
            except RuntimeError:
                pass
    th, Z = torch.linalg.eigh(T)
    return torch.flip(th, dims=(0,)), torch.flip(Z, dims=(1,))


def _smallest_eig_estimate(T: torch.Tensor, kern, gen, iters: int = 8) -> torch.Tensor:
    """An estimate (from above: a Rayleigh quotient) of the smallest eigenvalue of the small SPD matrix
    T, as a 0-d device tensor: ``iters`` steps of inverse iteration through K10's explicit L^-1
    (0.7 ms at 188 columns where the full Jacobi solve it replaces took 3.7 ms of the rank-200 eigen
    stage).  Only the filter's cut is taken from it, and any cut between the block's smallest Ritz
    value and the wanted ones filters correctly; the library solver answers when T is not
    numerically positive definite."""
    if kern is not None and 2 <= T.shape[0] <= getattr(kern, "chol_max_n", 0):
        L, Linv, info = _chol(kern, T)
        if _chol_ok(info):
            v = torch.randn((T.shape[0], 1), dtype=T.dtype, generator=gen, device=T.device)
            for _ in range(iters):
                v = Linv.T @ (Linv @ v)
                v = v / torch.linalg.vector_norm(v).clamp_min(1e-300)
            return torch.nan_to_num((v.T @ (T @ v)).reshape(()), nan=math.inf)   # (a NaN would win torch.minimum)
    return _eigh_desc(T, kern)[0][-1]


def _svd_wide(Bm: torch.Tensor, kern=None):
    """Thin SVD of the wide l x n fp64 matrix B = Q^T X of the randomized path (extmath.py:579) ->
    (Uhat (l, l), s (l,) descending, Vh (l, n)).  The library's gesvd is launch-rate bound with host
    round trips (4 ms at 20 x 8760, of a 113 ms step); here B^T = Q_b R by CholeskyQR2 (`_orth`,
    R = Q_b^T B^T whatever Q_b's triangularity), the l x l factor goes through the one-sided
    Jacobi kernel (K7L: left singular vectors of R^T = Uhat, errors relative to each sigma) and
    Vh = S^-1 Uhat^T B, re-orthonormalised by one Cholesky round.  A numerically rank-deficient B (s_l <= 1e-13 s_1: Vh's last rows would
    be noise) and sizes beyond the kernel go to the library."""
    l, n = Bm.shape
    mx = getattr(kern, "svd_jacobi_max_n", 0) if kern is not None else 0
    if Bm.dtype == torch.float64 and 2 <= l <= mx and n >= l:
        try:
            Bt = Bm.T.contiguous()
            Qb = _orth(Bt, kern=kern)                       # (n, l), orthonormal columns
            R = _tn(kern, Qb, Bt)                           # (l, l): B^T = Q_b R
            sig, Zt = kern.svd_jacobi(R.contiguous())       # row c of R = column c of R^T
            if bool(torch.isfinite(sig).all()) and float(sig[-1]) > 1e-13 * float(sig[0]):
                # rows of S^-1 Uhat^T B are orthonormal to eps * s_1 / s_j only (2e-9 at cfg4's
                # rank 200): one Cholesky round restores 1e-15 -- triangular, so every row is
                # corrected against the rows of LARGER sigma, the accurate ones
                Vt = _orth(((Zt @ Bm) / sig[:, None]).T.contiguous(), rounds=1, kern=kern)
                return Zt.T.contiguous(), sig, Vt.T.contiguous()
        except RuntimeError:
            pass
    return torch.linalg.svd(Bm, full_matrices=False)


def _graded_eigh(s0: torch.Tensor, Mm: torch.Tensor, kern=None):
    """Eigenpairs (descending) of T = S M S, S = diag(s0) spanning many decades, M = U'^T U' close
    to the identity -- with RELATIVE accuracy for the small eigenvalues.  Up to the Jacobi kernel's
    size (K7, relative rotation criterion) T is solved directly.  Beyond it the library solver
    (syevd) is only accurate to eps * s_1^2 absolutely: at s_k / s_1 = 5e-7 (cfg3's rank 200 on
    the rank-64 + noise matrix) that is 4e-4 of the small eigenvalues, and U came out orthonormal
    to 1e-3 only.  There T = B^T B with B = L^T S, M = L L^T, and the SVD of the l x l matrix B
    gives the same eigenvectors (its right singular vectors) with errors relative to s, not s^2."""
    l = s0.numel()
    T = s0[:, None] * Mm * s0[None, :]
    T = 0.5 * (T + T.T)
    if kern is not None and l <= getattr(kern, "eigh_small_max_n", 0):
        return _eigh_desc(T, kern)
    g = torch.nonzero(s0 > 0).squeeze(1)
    lg = int(g.numel())
    if lg == 0:
        return _eigh_desc(T, kern)
    Mg = Mm[g][:, g]
    L, _, info = _chol(kern, 0.5 * (Mg + Mg.T), want_inv=False)
    if not _chol_ok(info) or not bool(torch.isfinite(L).all()):
        return _eigh_desc(T, kern)
    Zg = None
    if kern is not None and 2 <= lg <= getattr(kern, "svd_jacobi_max_n", 0):
        # K7L on C = S L (column c of C = s * L[:, c]): T_g = C C^T, eigenvectors = left singular
        # vectors, errors relative to each singular value; one launch instead of gesvd's ~40 ms
        try:
            sig, Zt = kern.svd_jacobi((L.T * s0[g][None, :]).contiguous())
            Zg = Zt.T
        except RuntimeError:
            Zg = None
    if Zg is None:
        _, sig, Vbh = torch.linalg.svd(L.T * s0[g][None, :])
        Zg = Vbh.T
    mu = torch.zeros(l, dtype=T.dtype, device=T.device)
    mu[:lg] = sig * sig
    Z = torch.zeros((l, l), dtype=T.dtype, device=T.device)
    Z[g, :lg] = Zg
    if lg < l:   # directions dropped from S: eigenvalue 0, unit eigenvectors
        rest = torch.nonzero(s0 <= 0).squeeze(1)
        Z[rest, torch.arange(lg, l, device=T.device)] = 1.0
    return mu, Z


# Cost model of the eigen stage on the MI355X (scripts/probe_powerlaw.py, fp64): a product G @ Q
# takes ~1.0 ms at n = 8760 for any block up to 312 columns (rocBLAS, tall-skinny), CholeskyQR2
# ~0.9 ms at 77 columns / 1.2 at 124 / 2.5 at 312; the library's full solver (syevd) 816 / 142 /
# 50 ms at n = 8760 / 4000 / 2000.
_FULL_EIGH_MAX_N = 2048   # above it the filtered iteration never hands over to the library's full solver (50 ms at 2048, 816 ms at 8760)


def _full_eigh_ms(n: int) -> float:
    return 816.0 * (n / 8760.0) ** 2


def _filter_step_ms(n: int, b: int) -> float:
    return max(0.25, 1.0 * (n / 8760.0) ** 2) + 0.6 + 0.006 * b


# degree forecast x safety: one degree too many costs ~0.7 ms, one too few a whole extra pass + Rayleigh-Ritz
# (~4 ms); measured on the power-law cfg2 Gram: 1.25 -> degrees [6, 2] 18.5 ms, 1.6 -> [7] 16.6 ms, 2.0 -> [8] 16.9 ms
_CHEB_SAFETY = float(os.environ.get("DMDX_CHEB_SAFETY", "1.75"))


def top_eigh(G: torch.Tensor, l: int, method: str = "auto", tol: float = 1e-9,
             max_outer: int = 40, info: dict | None = None, kern=None):
    """Largest ``l`` eigenpairs of the symmetric PSD fp64 matrix ``G`` (n x n),
    eigenvalues descending.

    ``full``: torch.linalg.eigh.  ``cheb`` (alias ``krylov``, the name of the restarted block
    Krylov solver this replaced): block power steps, then a Chebyshev-filtered subspace iteration
    -- only products G @ block, CholeskyQR2 and (b x b) Rayleigh-Ritz steps --, run until every
    wanted pair has residual <= tol * lambda_1 (default 1e-9: the level of G's own rounding error
    -- its entries are sums of fp32 products --, i.e. the pairs returned are exact for a matrix as
    close to X^T X as G itself is).  ``auto`` picks by size.

    The filter.  With the block's smallest Ritz value c as the cut, the degree-d Chebyshev
    polynomial of [0, c] damps everything G has below the block and grows like rho^d above it,
    rho = x + sqrt(x^2 - 1), x = 2 lambda / c - 1: a gap-free spectrum lambda_i ~ i^-2 (ERA5
    anomalies are power-law, not low-rank + noise) with l = 62, b = 124 gains a factor 14 per
    product on the slowest wanted pair where a power step gains 4.  The spectrum of a snapshot Gram
    spans many decades inside the block (lambda_1 / lambda_l ~ 4000), so the three-term recurrence
    cannot be run on the block as it stands -- after d steps the columns have collapsed onto the
    leading eigenvectors by (lambda_1 / lambda_l)^d --; the polynomial is applied instead as the
    product of its linear factors, Q <- orth((G - r_i I) Q) over the d Chebyshev roots r_i of
    [0, c]: the same subspace, each step a shifted power step whose block stays well inside the
    range of CholeskyQR2.  The degree of every pass is forecast from the residual and the rho of
    the slowest wanted pair, a Rayleigh-Ritz step follows each pass.  When the forecast says the
    remaining passes cost more than the library's full solver (flat spectra -- pure noise has no
    gap for any polynomial to use -- at sizes where syevd is cheap) the full solver is taken at
    once; the rule is a function of n and the residuals only, so every rank and every run takes
    the same branch.
    """
    n = G.shape[0]
    l = min(l, n)
    # auto (measured on the MI355X, l = 62: scripts/probe_eig_threshold.py): the block power steps
    # cost 3-4 ms whatever n is and finish every steep spectrum; the library solver costs 9 / 17 /
    # 23 / 36 ms at n = 384 / 768 / 1024 / 1536.  So: power steps first unless the block would be
    # most of the matrix, then the full solver up to n = 1024, the filtered iteration above.
    full_after_power = False
    if method == "krylov":
        method = "cheb"
    if method == "auto":
        if n <= 256 or 4 * l >= n:
            method = "full"
        else:
            method, full_after_power = "cheb", n <= 1024
    if method == "full":
        lam, V = torch.linalg.eigh(G)
        lam = torch.flip(lam[-l:], dims=(0,))
        V = torch.flip(V[:, -l:], dims=(1,))
        if info is not None:
            info["eig_method"] = "full"
        return lam, V.contiguous()
    if method != "cheb":
        raise ValueError(f"top_eigh: unknown method {method!r}")

    if n % 2 == 1 and n > 1024 and getattr(kern, "symm_skinny", None) is not None:
        # K8 (fp64 MFMA products G Q) needs an even order and leading dimension; an odd Gram -- delay
        # embedding d = 2 of an even number of snapshots, the reference's default config: n = 8759 --
        # would send every product of the stage to the library GEMM (1.0 instead of 0.33 ms each).
        # One zero row and column are appended (0.3 ms for 614 MB): the extra eigenvalue is 0, the
        # wanted pairs are unchanged and have a zero last component.
        Gp = torch.zeros((n + 1, n + 1), dtype=G.dtype, device=G.device)
        Gp[:n, :n] = G
        lam, Vp = top_eigh(Gp, l, method="cheb", tol=tol, max_outer=max_outer, info=info, kern=kern)
        if info is not None:
            info["eig_padded_to_even"] = True
        return lam, Vp[:n].contiguous()
    b = min(n // 3, l + max(8, l // 4) + 1) & ~1    # even widths: K8's 16-byte fragment loads
    gen = torch.Generator(device=G.device).manual_seed(1234)  # (a host draw + upload costs 7 ms)
    Q = torch.randn((n, b), dtype=torch.float64, generator=gen, device=G.device)
    k8 = getattr(kern, "symm_skinny", None)

    def gq(Qb, shift: float = 0.0):
        """G Qb - shift Qb: K8 (fp64 MFMA, one pass over G) when the provider has it."""
        if k8 is not None:
            return k8(G, Qb, shift)
        return torch.addmm(Qb, G, Qb, beta=-shift) if shift != 0.0 else G @ Qb

    Q = _orth(gq(Q), kern=kern)
    products = 1

    def ritz(S, GS):
        """Rayleigh-Ritz of G in span(S) (S orthonormal, GS = G S): Ritz values (descending), Ritz
        vectors Qn, G Qn, and the residual norms of all of them relative to theta_1."""
        T = _tn(kern, S, GS)
        T = 0.5 * (T + T.T)
        th, Z = _eigh_desc(T, kern)
        Qn = S @ Z
        GQn = GS @ Z
        res = torch.linalg.vector_norm(GQn - Qn * th, dim=0) / th[0].abs().clamp_min(1e-300)
        # Pairs at the rounding floor of G (theta_j < 8 tol theta_1: G is a sum of fp32 products,
        # its noise "eigenvalues" are of that size and have no gaps to converge in) count as
        # converged: svd_snapshots re-derives such directions from X itself (its polish step
        # triggers at lambda_k < 1e-7 lambda_1).
        res = torch.where(th > 8.0 * tol * th[0].abs(), res, torch.zeros_like(res))
        return th, Qn, GQn, res

    def done(th, Qn, res, how, it):
        if info is not None:
            info["eig_method"] = how
            info["eig_outer_iters"] = it
            info["eig_residual"] = res
            info["eig_products"] = products
            info["eig_block"] = int(Qn.shape[1])
        return th[:l].contiguous(), Qn[:, :l].contiguous()

    # Fast path: block power steps with a (b x b) Rayleigh-Ritz (K7) after the 2nd -- and, if the
    # Ritz values say one more round (two products, each gaining theta_j / theta_b on pair j) will
    # do, after the 4th.  With a steep spectrum behind the block (lambda_{b+1} << lambda_l: every
    # low-rank + noise matrix, cfg2, where the fp32 rounding of G leaves a floor of ~1e-9 lambda_1
    # and each step gains ~3 digits) this reaches the tolerance after 3 products; otherwise its
    # Ritz vectors are the start of the filtered iteration.
    for it in range(2):
        Q = _orth(gq(Q), kern=kern)
        Y = gq(Q)
        products += 2
        th, Q, GQ, resv = ritz(Q, Y)
        res = float(resv[:l].max())
        if res <= tol:
            return done(th, Q, res, "power", it + 1)
        if it == 0:
            gain = (th[:l] / th[b - 1].clamp_min(1e-300)) ** 2
            if not bool((resv[:l] <= tol * gain).all()):
                break
    if full_after_power:
        return top_eigh(G, l, method="full", info=info, kern=kern)

    # Widen the block for the filter (its growth rate per product is set by lambda_l / lambda_{b+1}):
    # 2 l columns, the new ones G-multiplied noise orthogonalised against the Ritz vectors; G Q of
    # the old ones is known from the Rayleigh-Ritz step, so this costs two products.
    b2 = min(n // 3, max(b, 2 * l)) & ~1
    if b2 > b:
        # NO Rayleigh-Ritz step on the widened block before the first filter pass (round 3: it cost
        # 12 ms of 65 at rank 200 -- a 500 x 500 eigenproblem -- to learn one number, the cut).  The
        # widened block [Q, W] is orthonormal as it stands; the wanted pairs and their residuals are
        # those of the narrow block; the cut -- the smallest Ritz value of the block, below which
        # the filter damps -- is taken from the small matrix W^T G W alone: its smallest eigenvalue
        # is >= the block's smallest Ritz value (interlacing) and far below theta_l, and any cut below
        # theta_l filters correctly (a higher cut only converges a little more slowly).
        W = gq(torch.randn((n, b2 - b), dtype=torch.float64, generator=gen, device=G.device))
        for _ in range(2):
            W = W - Q @ _tn(kern, Q, W)
        W = _orth(W, kern=kern)
        GW = gq(W)
        products += 2
        Tw = _tn(kern, W, GW)
        cut_w = torch.minimum(_smallest_eig_estimate(0.5 * (Tw + Tw.T), kern, gen), th[b - 1]).clamp_min(0.0)
        Q = torch.cat([Q, W], dim=1)
        th = torch.cat([th, cut_w.expand(b2 - b)])
        resv = torch.cat([resv, torch.zeros(b2 - b, dtype=resv.dtype, device=resv.device)])
        b = b2

    max_deg = 24
    full_ms = _full_eigh_ms(n)
    step_ms = _filter_step_ms(n, b)
    spent = 0
    capped = 0
    degrees: list[int] = []
    if info is not None:
        info["eig_degrees"] = degrees
    for it in range(max_outer):
        host = torch.cat([th, resv]).tolist()          # one transfer: Ritz values + residuals
        thl, rl = host[:b], host[b:]
        cut = max(thl[b - 1], 0.0)
        # the slowest wanted pair: the unconverged one closest to the cut
        j = max(i for i in range(l) if rl[i] > tol)
        x = 2.0 * thl[j] / cut - 1.0 if cut > 0.0 else math.inf
        rho = x + math.sqrt(max(x * x - 1.0, 0.0)) if math.isfinite(x) else math.inf
        need = math.log(3.0 * rl[j] / tol) / math.log(rho) if rho > 1.0 + 1e-12 else math.inf
        hopeless = not math.isfinite(need) or need * step_ms > full_ms or spent * step_ms > 3.0 * full_ms
        if hopeless:
            if info is not None:
                info["eig_cheb_forecast_steps"] = float(min(need, 1e9))
            if n <= _FULL_EIGH_MAX_N:
                break
            # A spectrum with no gap anywhere near the wanted rank (a numerical multiple of the
            # identity: every eigenvalue within 0.1 %) at a size where the library's full solver
            # costs most of a second (828 ms at n = 8760, more than the Gram itself): no polynomial
            # converges there, and nothing is lost by not converging -- every orthonormal block
            # captures the same energy to the spread of the spectrum.  Two more passes of full
            # degree, then the Ritz pairs are returned with their residual stated (round 3).
            if capped >= 2:
                if info is not None:
                    info["eig_warning"] = (f"flat spectrum: residual {res:.2e} lambda_1 > tol {tol:.0e} after {products} products; "
                                           "Ritz values are accurate to that residual")
                return done(th, Q, res, "cheb-capped", it)
            capped += 1
            need = float(max_deg)
        # (the first forecast rests on the Ritz values of a freshly widened block: capped lower)
        deg = int(min(max_deg if it else 10, max(2, math.ceil(_CHEB_SAFETY * need) + 1)))
        degrees.append(deg)
        roots = [0.5 * cut * (1.0 + math.cos(math.pi * (2 * i + 1) / (2 * deg))) for i in range(deg)]
        order = []
        lo, hi = 0, deg - 1
        while lo <= hi:                                  # large and small roots alternate
            order += [roots[lo]] if lo == hi else [roots[lo], roots[hi]]
            lo, hi = lo + 1, hi - 1
        for i, r_i in enumerate(order):
            # one CholeskyQR round after every second factor (conditioning control: a factor
            # stretches the block by at most (lambda_1 - r) / (theta_b - r) ~ 1e6, and what two
            # of them add along the leading eigenvectors, ~1e-16 x 1e8, the round removes), two
            # rounds before the Rayleigh-Ritz step (an orthonormal basis)
            Q = gq(Q, r_i)
            if i == deg - 1:
                Q = _orth(Q, kern=kern)
            elif i % 2 == 1:
                Q = _orth(Q, rounds=1, kern=kern)
            else:
                Q = Q / torch.linalg.vector_norm(Q, dim=0, keepdim=True).clamp_min(1e-300)
        Y = gq(Q)
        products += deg + 1
        spent += deg
        th, Q, GQ, resv = ritz(Q, Y)
        res = float(resv[:l].max())
        if res <= tol:
            return done(th, Q, res, "cheb", it + 1)
    if info is not None:
        info["eig_cheb_failed_residual"] = res
        info["eig_products"] = products
    return top_eigh(G, l, method="full", info=info, kern=kern)


# ---------------------------------------------------------------------------
# row (space) blocks
# ---------------------------------------------------------------------------
# A 1 038 240-row fp32 snapshot has a 4 MB column stride: the 256 columns a Gram
# tile touches per K-step then live in 256 different 2 MB pages and the per-CU
# TLB thrashes (measured: 32 % UTCL1 misses, -10 % Gram throughput).  The engine
# therefore keeps X in HBM as row blocks of at most BLOCK_ROWS space points, each
# a contiguous (time, rows) tensor; Gram / projections are sums / concatenations
# over the blocks.  (It is also exactly the shape of a multi-GPU row shard.)
BLOCK_ROWS = 131072


def _pitched(kern, Wt: torch.Tensor) -> torch.Tensor:
    """The small operand of K2, re-pitched ONCE for all row blocks (HipKernels.pitch)."""
    fn = getattr(kern, "pitch", None)
    return fn(Wt) if fn is not None else Wt


def split_rows(m: int, block_rows: int | None = None) -> list[tuple[int, int]]:
    """[(start, stop)] of nearly equal row blocks, each a multiple of 4 rows -- except the last one
    when m is not (the ingest pads that one with zero rows, era5_svd._upload_variable)."""
    block_rows = block_rows or BLOCK_ROWS
    nb = max(1, -(-m // block_rows))
    base = -(-(-(-m // nb)) // 4) * 4
    out, r = [], 0
    while r < m:
        out.append((r, min(m, r + base)))
        r += base
    return out


def as_blocks(Xt) -> list[torch.Tensor]:
    """Normalise the snapshot-matrix argument to a list of (time, rows) blocks.
    A single tensor wider than BLOCK_ROWS is re-blocked (device copy)."""
    if isinstance(Xt, (list, tuple)):
        return list(Xt)
    n, m = Xt.shape
    if m <= 2 * BLOCK_ROWS:
        return [Xt]
    return [Xt[:, a:b].contiguous() for a, b in split_rows(m)]


def _gram_blocks(blocks, kern, comm: Comm) -> torch.Tensor:
    """G = sum over the local row blocks and over the ranks of X_b^T X_b (fp64, both triangles).
    Across ranks only the upper triangle travels: n (n + 1) / 2 doubles, 307 MB instead of 614 MB
    at n = 8760 (the one large exchange of the row-sharded method of snapshots)."""
    G = kern.syrk_blocks(blocks) if len(blocks) > 1 else kern.syrk(blocks[0])
    n = G.shape[0]
    if comm.exchanges and n >= 64 and hasattr(kern, "pack_triu"):
        packed = comm.allreduce_sum_(kern.pack_triu(G), tag="gram_allreduce" if n >= 1024 else "small_gram_allreduce")
        return kern.unpack_triu(packed, n, out=G)
    return comm.allreduce_sum_(G, tag="gram_allreduce" if n >= 1024 else "small_gram_allreduce")


def _gemm_tn_blocks(Ablocks, Bblocks, kern, comm: Comm) -> torch.Tensor:
    if len(Ablocks) > 1:
        C = kern.gemm_tn_blocks(list(Ablocks), list(Bblocks))
    else:
        C = kern.gemm_tn(Ablocks[0], Bblocks[0])
    return comm.allreduce_sum_(C, tag="xty_allreduce")


def _project_blocks(kern, Xblocks, Wt: torch.Tensor, d: int, whole: torch.Tensor | None = None):
    """[X_b W for every row block] -> (blocks, whole).  Without delay embedding the per-block
    results are written straight into column slices of ONE (l, M) tensor (``whole``; the blocks
    are views of it), so that assembling U is not a second copy of it -- at cfg4 that copy was a
    13 GB allocation next to 227 GB of X, which made the caching allocator hand cached segments
    back to the driver (350 ms).  With d > 1 the rows of a block interleave with the other
    blocks' (order k_delay*m + s) and ``whole`` is None."""
    if d != 1 or len(Xblocks) == 1:
        return [kern.skinny(X, Wt) for X in Xblocks], None
    M = sum(int(X.shape[1]) for X in Xblocks)
    if whole is None or tuple(whole.shape) != (Wt.shape[0], M):
        whole = torch.empty((Wt.shape[0], M), dtype=torch.float32, device=Xblocks[0].device)
    views, off = [], 0
    for X in Xblocks:
        mb = int(X.shape[1])
        views.append(kern.skinny(X, Wt, out=whole[:, off:off + mb]))
        off += mb
    return views, whole


def _assemble_rows(Ublocks, d: int) -> torch.Tensor:
    """Concatenate per-block results (k, d*mb_b) into (k, d*m) in the reference's
    embedded row order k_delay*m + s (slice_tools.py:207-211)."""
    if len(Ublocks) == 1:
        return Ublocks[0]
    if d == 1:
        return torch.cat(Ublocks, dim=1)
    parts = []
    for kd in range(d):
        for U in Ublocks:
            mb = U.shape[1] // d
            parts.append(U[:, kd * mb:(kd + 1) * mb])
    return torch.cat(parts, dim=1)


def _sign_flip(Ublocks, Vh: torch.Tensor, comm: Comm, kern):
    """u-based sign convention of sklearn's svd_flip (extmath.py:935-943): the
    largest-|.| entry of every left singular vector becomes positive."""
    vals = []
    for Ut in Ublocks:
        idx = Ut.abs().argmax(dim=1, keepdim=True)
        vals.append(Ut.gather(1, idx).squeeze(1))
    allv = torch.stack(vals, dim=0)  # (blocks, k)
    pick = allv.abs().argmax(dim=0, keepdim=True)
    val = allv.gather(0, pick).squeeze(0)
    if comm.exchanges:
        # ONE exchange per rank, whatever its number of row blocks (ranks may hold different numbers
        # of blocks: bands of different height, streamed pieces)
        allv = torch.stack(comm.allgather(val.contiguous(), tag="sign_allgather"), dim=0)  # (world, k)
        pick = allv.abs().argmax(dim=0, keepdim=True)
        val = allv.gather(0, pick).squeeze(0)
    sign = torch.where(val < 0, -torch.ones_like(val), torch.ones_like(val))
    for Ut in Ublocks:
        kern.scale_columns_(Ut, sign.to(torch.float32))
    Vh = Vh * sign.to(Vh.dtype)[:, None]
    return Ublocks, Vh


class _PhaseClock:
    """Wall time per phase of a driver, for reporting only (``timings=True``: every call
    synchronises the device, so the phases no longer overlap with the host)."""

    def __init__(self, device, on: bool):
        self.dev, self.on, self.acc = device, bool(on), {}
        self.t = self._now() if self.on else 0.0

    def _now(self) -> float:
        if self.dev.type == "cuda":
            torch.cuda.synchronize(self.dev)
        return time.perf_counter()

    def __call__(self, name: str) -> None:
        if self.on:
            t = self._now()
            self.acc[name] = self.acc.get(name, 0.0) + (t - self.t)
            self.t = t


def _sync_time(device) -> float:
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    return time.perf_counter()


# ---------------------------------------------------------------------------
# "standard": method of snapshots
# ---------------------------------------------------------------------------
def _shard_stats(blocks, comm: Comm, delay: int, with_mean: bool) -> dict:
    """What the drivers need to know about the matrix as a whole, from ONE exchange: the largest
    magnitude (for the power-of-two rescaling), the energy of the per-row time mean against the
    energy around it (un-centred fields such as temperature ~ 280 K + O(10) K anomalies), both read
    from up to 2048 rows of every block, and the global number of embedded rows."""
    dev = blocks[0].device
    v = torch.zeros(4, dtype=torch.float64, device=dev)
    v[0] = torch.stack([B[:, : min(2048, B.shape[1])].abs().max() for B in blocks]).max().double()
    if with_mean:
        # (Welford in fp32 on the sample as it lies: the fp64 copies of 2048 columns of every block
        # cost 3 ms per cfg2 step, 0.5 % -- and the answer only feeds a factor-of-many comparison)
        # the sample is brought to O(1) first (data at 1e-24 would underflow fp32 squares), by the
        # device-side amax, no host round trip
        inv = (1.0 / v[0].clamp_min(1e-300)).to(torch.float32)
        usable = torch.isfinite(inv)                 # (all-zero data, fp32 denormals: leave as it is)
        inv = torch.where(usable, inv, torch.ones_like(inv))
        for B in blocks:
            var, mean = torch.var_mean(B[:, : min(2048, B.shape[1])] * inv, dim=0, unbiased=False)
            v[1] += (mean.double() * mean.double()).sum()
            v[2] += var.double().sum()
        scale2 = torch.where(usable, v[0] * v[0], torch.ones_like(v[0]))
        v[1] *= scale2          # back to the data's units (fp64 holds 1e-48 .. 1e44): ranks with
        v[2] *= scale2          # different amax must add like with like
    v[3] = float(sum(B.shape[1] for B in blocks) * delay)
    if comm.exchanges:
        allv = torch.stack(comm.allgather(v, tag="stats_allgather"))
        v = torch.cat([allv[:, :1].max(dim=0).values, allv[:, 1:].sum(dim=0)])
    amax, mean2, var, rows = v.tolist()
    return {"amax": amax, "mean2": mean2, "var": var, "rows": int(round(rows))}


def _magnitude_guard(with_mean: bool):
    """Entry-point decorator: gathers the global facts about the matrix (:func:`_shard_stats`, one
    collective) and passes them on as ``_stats``; data far outside the fp32 comfort zone
    (|x| ~ 1e20: the squares overflow; ~1e-20: they underflow) are scaled in place by a power of
    two (exact), factored, and scaled back; s is scaled accordingly.  LAPACK's gesdd does the same
    (xLASCL)."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapper(Xt, *args, **kwargs):
            blocks = as_blocks(Xt)
            comm = kwargs.get("comm") or Comm()
            delay = kwargs.get("delay", args[1] if len(args) > 1 else 1)
            want_mean = with_mean and kwargs.get("deflate_mean") is None
            stats = _shard_stats(blocks, comm, int(delay), want_mean)
            kwargs["_stats"] = stats
            a = stats["amax"]
            if not (math.isfinite(a) and a > 0.0) or 2.0 ** -40 <= a <= 2.0 ** 40:
                return fn(blocks, *args, **kwargs)
            c = 2.0 ** (-round(math.log2(a)))
            for B in blocks:
                B *= c
            try:
                res = fn(blocks, *args, **kwargs)
            finally:
                for B in blocks:
                    B *= 1.0 / c
            res.s = res.s / c
            res.info["rescaled_by"] = c
            return res

        return wrapper

    return deco


def _eig_mean_deflated(G, w, musq, delay: int, l: int, eig_method: str, info: dict, kern):
    """Top-l eigenpairs of E^T E = G + 1 w^T + w 1^T + d |mu|^2 1 1^T from the Gram G = A^T A of the
    row-centred matrix, w = A^T mu~ and |mu|^2 (see :func:`svd_snapshots`): the dominant pair is
    split off exactly (Schur complement of the direction q = 1 / sqrt(nd)), the rest comes from
    the small-scale matrix Gd, and a Rayleigh-Ritz step in span{q, z, V_d} restores the coupling.
    Returns (lam, V, lam_d, l)."""
    nd, dev = G.shape[0], G.device
    rootn = math.sqrt(nd)
    q = torch.full((nd,), 1.0 / rootn, dtype=torch.float64, device=dev)
    h = G @ q
    qh = torch.dot(q, h)
    alpha = qh + 2.0 * rootn * torch.dot(w, q) + delay * musq[0] * nd
    zf = h + rootn * w
    z = zf - q * torch.dot(q, zf)
    Gd = G - torch.outer(q, h) - torch.outer(h, q) + qh * torch.outer(q, q) - torch.outer(z, z) / alpha
    Gd = 0.5 * (Gd + Gd.T)
    lam_d, Vd = top_eigh(Gd, max(1, l - 1), method=eig_method, info=info, kern=kern)
    # Basis of the complement: z itself (G q = alpha q + z: with it the dominant pair is
    # exact up to lambda_2 / alpha, whatever part of z lies outside the kept modes) and the
    # leading eigenvectors of Gd, projected off q (a rank-deficient Gd returns arbitrary
    # null vectors, q among them); the projected matrix comes from products, not from lam_d.
    Bs = torch.cat([z[:, None] / torch.linalg.vector_norm(z).clamp_min(1e-300), Vd], dim=1)
    Bs = _orth(Bs - torch.outer(q, q @ Bs), kern=kern)
    c = Bs.T @ z
    T11 = Bs.T @ (Gd @ Bs) + torch.outer(c, c) / alpha
    T = torch.zeros((Bs.shape[1] + 1,) * 2, dtype=torch.float64, device=dev)
    T[0, 0] = alpha
    T[0, 1:] = c
    T[1:, 0] = c
    T[1:, 1:] = 0.5 * (T11 + T11.T)
    lam, Z0 = _eigh_desc(T, kern)
    l = min(l, lam.numel())
    lam, Z0 = lam[:l].contiguous(), Z0[:, :l]
    V = torch.cat([q[:, None], Bs], dim=1) @ Z0
    info["mean_deflated"] = True
    return lam, V, lam_d, l


def _complete_unit_rows(Ub, ok: torch.Tensor, comm: Comm) -> None:
    """Left singular vectors whose singular value lies below the resolution of the data (s_j <=
    1e-7 s_1 in fp32: their rows of the (k, M) factor came out as zeros) are replaced IN PLACE by
    unit vectors orthogonal to the resolved ones and to each other -- what LAPACK returns for a
    zero singular value (np.linalg.svd, era5_svd.py:251: any orthonormal completion; round 3: the
    factor was left with zero columns before).  Random directions (seeded per rank: every rank
    fills its own rows), two projections against the resolved rows, CholeskyQR2 among themselves;
    the inner products are summed over the row blocks and the ranks in fp64.  Only ever runs on
    (numerically) rank-deficient input, so plain torch products on the few rows involved."""
    bad = torch.nonzero(~ok).squeeze(1)
    nbad = int(bad.numel())
    if nbad == 0:
        return
    good = torch.nonzero(ok).squeeze(1)
    dev = Ub[0].device
    gen = torch.Generator(device=dev).manual_seed(4321 + 7919 * int(comm.rank))
    R = [torch.randn((nbad, U.shape[1]), generator=gen, device=dev, dtype=torch.float32) for U in Ub]
    if good.numel():
        Ug = [U[good] for U in Ub]
        for _ in range(2):
            C = torch.zeros((nbad, int(good.numel())), dtype=torch.float64, device=dev)
            for Rb, Gb in zip(R, Ug):
                C += Rb.double() @ Gb.double().T
            comm.allreduce_sum_(C, tag="completion_allreduce")
            for Rb, Gb in zip(R, Ug):
                Rb -= (C @ Gb.double()).to(torch.float32)
    for _ in range(2):
        G = torch.zeros((nbad, nbad), dtype=torch.float64, device=dev)
        for Rb in R:
            G += Rb.double() @ Rb.double().T
        comm.allreduce_sum_(G, tag="completion_allreduce")
        L = torch.linalg.cholesky(0.5 * (G + G.T))
        Linv = torch.linalg.solve_triangular(L, torch.eye(nbad, dtype=torch.float64, device=dev), upper=False)
        for i, Rb in enumerate(R):
            R[i] = (Linv @ Rb.double()).to(torch.float32)
    for U, Rb in zip(Ub, R):
        U[bad] = Rb


@_magnitude_guard(with_mean=True)
def svd_snapshots(Xt, n_components: int, delay: int = 1, oversample: int | None = None,
                  refine: bool = True, flip_sign: bool = True, comm: Comm | None = None,
                  kern=None, eig_method: str = "auto", timings: bool = False,
                  deflate_mean: bool | None = None, _stats: dict | None = None) -> SvdResult:
    """Rank-k SVD of the (delay-embedded) snapshot matrix by the Gram route.

    Xt: (n, m_local) fp32 device tensor -- or a list of such row blocks --
    already pre-processed (centred/scaled).  Returns local rows of U; s and V
    are replicated on every rank.

    ``deflate_mean``: None = detect a dominant per-row time mean (un-centred data) on a sample
    of rows; True / False force / forbid its exact deflation.  The Gram route squares the
    condition number: with a dominant time mean (s_1 ~ 1e4 s_2 for un-centred temperature) the
    wanted trailing eigenvalues of E^T E drown in the rounding of its fp32 products.  With
    E = E(Xc) + mu~ 1^T (Xc row-centred, mu~ the time means repeated for every delay):
        E^T E = A^T A + 1 w^T + w 1^T + |mu~|^2 1 1^T,     A = E(Xc),  w = A^T mu~,
    and in the coordinates (q = 1/sqrt(nd), its complement P = I - q q^T) the mean terms vanish
    from the complement:  P E^T E P = P A^T A P.  So the blocks are centred in place (K5), the
    accurate small-scale Gram A^T A is computed as usual, the non-dominant eigenpairs come from
    Gd = P A^T A P - z z^T / alpha  (alpha = q^T E^T E q, z = P E^T E q: the Schur complement of
    the dominant direction, exact up to lambda / alpha ~ 1e-9), the (l x l) Rayleigh-Ritz in
    span{q, V} restores the exact coupling, U' = E V S^-1 is formed as A V S^-1 + mu~ (1^T V) S^-1,
    and the blocks are un-centred again before returning.
    """
    kern = _kern(kern)
    comm = comm or Comm()
    info: dict = {}
    blocks = as_blocks(Xt)
    dev = blocks[0].device
    if _stats is None:
        _stats = _shard_stats(blocks, comm, delay, deflate_mean is None)
    if deflate_mean is None:
        # > 99 % of the energy in the per-row time mean: below that ratio s_1 / s_2 stays in the
        # hundreds, which the plain Gram route resolves (and the deflation's Schur complement is
        # exact only up to lambda_2 / lambda_1)
        deflate_mean = refine and blocks[0].shape[0] - delay + 1 > 2 and _stats["mean2"] > 100.0 * _stats["var"]
    t0 = _sync_time(dev) if timings else 0.0
    mus = [kern.row_center_scale_(B, False)[0] for B in blocks] if deflate_mean else None
    try:
        G = _gram_blocks(blocks, kern, comm)
        if not bool(torch.isfinite(torch.diagonal(G)).all()):
            # NaN / Inf in X: what np.linalg.svd (the reference's call, era5_svd.py:251) reports
            raise np.linalg.LinAlgError("SVD did not converge")
        if mus is not None:
            # w = Xc^T mu (n,), |mu|^2: one K3 pass with a single column
            w = _gemm_tn_blocks([mu[None, :].contiguous() for mu in mus], blocks, kern, comm).reshape(-1)
            musq = torch.stack([(mu.double() ** 2).sum() for mu in mus]).sum().reshape(1)
            comm.allreduce_sum_(musq)
        if delay > 1:
            G = kern.delay_shift_sum(G, delay)
            if mus is not None:   # w_E[t] = sum_k w[t + k]
                ndw = w.numel() - delay + 1
                w = torch.stack([w[kd:kd + ndw] for kd in range(delay)]).sum(dim=0)
        nd = G.shape[0]
        t1 = _sync_time(dev) if timings else 0.0

        Mg = _stats["rows"]
        k = min(n_components, nd, Mg)  # np.linalg.svd(full_matrices=False)[:k]
        p = oversample if oversample is not None else max(8, k // 4)
        l = min(nd, k + p) if refine else k
        lam_d = None
        if mus is None:
            lam, V = top_eigh(G, l, method=eig_method, info=info, kern=kern)
        else:
            lam, V, lam_d, l = _eig_mean_deflated(G, w, musq, delay, l, eig_method, info, kern)
        comm.broadcast_(lam, V, tag="eig_broadcast")
        lam1 = lam[0].clamp_min(1e-300)
        ref = lam[1] if (mus is not None and lam.numel() > 1) else lam1   # the deflated scale
        steep = float(lam[min(k, lam.numel()) - 1]) < 1e-7 * float(ref)
        polish = steep and refine
        tp0 = _sync_time(dev) if (timings and polish) else 0.0
        if polish:
            # Steep spectrum (s_k < 3e-4 s_1): G (sums of fp32 products) resolves eigenvalues down
            # to ~1e-9 lambda_1 only, so the trailing wanted directions are poorly determined by
            # it.  One subspace iteration on X itself, V <- orth(E^T (E V)) (a K2 and a K3 pass),
            # fixes them up to the rounding of those products (~1e-7 s_1 / s_j relative); the
            # Rayleigh-Ritz below then runs on the un-normalised E V.
            Eb = [embed_view(B, delay) for B in blocks]
            Vt32 = _pitched(kern, V.T.contiguous().to(torch.float32))
            Yb = [kern.skinny(E, Vt32) for E in Eb]
            if mus is not None:   # E = A + mu~ 1^T on the centred blocks: Y += mu~ (1^T V)
                ones_v = V.sum(dim=0).to(torch.float32)
                for Y, mu in zip(Yb, mus):
                    Y.addmm_(ones_v[:, None], mu.repeat(delay)[None, :])
            Zt = _gemm_tn_blocks(Eb, Yb, kern, comm)                  # (l, nd) = (A^T Y)^T
            if mus is not None:   # ... and E^T Y = A^T Y + 1 (mu~^T Y)
                my = torch.zeros(Zt.shape[0], dtype=torch.float64, device=dev)
                for Y, mu in zip(Yb, mus):
                    my += Y.double() @ mu.double().repeat(delay)
                comm.allreduce_sum_(my)
                Zt = Zt + my[:, None]
            Zn = Zt / torch.linalg.vector_norm(Zt, dim=1, keepdim=True).clamp_min(1e-300)
            V = _orth(Zn.T.contiguous(), kern=kern)   # (columns span ~s_j^2: normalised first, CholeskyQR is not scale invariant)
            comm.broadcast_(V)
            lam = torch.ones_like(lam)                                # no S^-1 scaling of E V below
            info["polished"] = True
            if timings:
                info["t_polish"] = _sync_time(dev) - tp0              # (part of t_eig: two passes over X)
        elif steep:
            info["warning"] = ("s_k < 3e-4 s_1 and refine=False: the Gram matrix of fp32 products resolves "
                               "eigenvalues down to ~1e-9 lambda_1 only")
        good = lam > lam1 * 1e-14
        if polish:
            good = torch.ones_like(lam, dtype=torch.bool)
            ref = lam1 = lam[0]
        if mus is not None and not polish:
            # resolvable: above the rounding of the deflated Gram (scale lam_d[0]) and above the
            # rounding the dominant entry alpha leaves in the small Rayleigh-Ritz problem
            good = lam > torch.maximum(lam_d[0].abs() * 1e-13, lam1 * 1e-15)
        s0 = torch.sqrt(torch.where(good, lam, torch.ones_like(lam)))
        inv_s0 = torch.where(good, 1.0 / s0, torch.zeros_like(s0))
        s0 = torch.where(good, s0, torch.zeros_like(s0))
        t2 = _sync_time(dev) if timings else 0.0

        Wt = _pitched(kern, (V * inv_s0).T.contiguous().to(torch.float32))  # (l, nd)
        Up = [kern.skinny(embed_view(B, delay), Wt) for B in blocks]  # (l, d*mb): U' = X V S^-1
        if mus is not None:   # + mu~ (1^T V) S^-1: the part of E the centred blocks no longer hold
            cvec = (V.sum(dim=0) * inv_s0).to(torch.float32)
            for U, mu in zip(Up, mus):
                U.addmm_(cvec[:, None], mu.repeat(delay)[None, :])
        t3 = _sync_time(dev) if timings else 0.0

        if refine:
            # Rayleigh-Ritz in span(V): (XV)^T (XV) = S (U'^T U') S, graded by S so the
            # small singular values keep their relative accuracy.
            Mm = _gram_blocks(Up, kern, comm)                         # (l, l) fp64
            mu_, Z = _graded_eigh(s0, Mm, kern)
            mu_ = mu_[:k].contiguous()
            Z = Z[:, :k].contiguous()
            comm.broadcast_(mu_, Z)
            s = torch.sqrt(mu_.clamp_min(0.0))
            ok = s > s0[0] * 1e-7 if (mus is None and not polish) else s > s[0] * 1e-7
            inv_s = torch.where(ok, 1.0 / torch.where(ok, s, torch.ones_like(s)), torch.zeros_like(s))
            Rm = (s0[:, None] * Z) * inv_s[None, :]                   # (l, k): U = U' R
            Rt = _pitched(kern, Rm.T.contiguous().to(torch.float32))
            Ub, whole = _project_blocks(kern, Up, Rt, delay)          # (k, d*mb)
            Vh = (V @ Z).T.contiguous()
            if not bool(ok.all()):
                _complete_unit_rows(Ub, ok, comm)
                info["completed_directions"] = int((~ok).sum())
        else:
            s = s0[:k]
            Ub, whole = [U[:k] for U in Up], None
            Vh = V[:, :k].T.contiguous()
            if not bool(good[:k].all()):
                _complete_unit_rows(Ub, good[:k], comm)
                info["completed_directions"] = int((~good[:k]).sum())
    finally:
        if mus is not None:
            for B, mu in zip(blocks, mus):
                B += mu                                              # un-centre: the caller's X is intact again
    if flip_sign:
        Ub, Vh = _sign_flip(Ub, Vh, comm, kern)
    Ut = whole if whole is not None else _assemble_rows(Ub, delay)
    if timings:
        t4 = _sync_time(dev)
        info.update(t_gram=t1 - t0, t_eig=t2 - t1, t_project=t3 - t2, t_refine=t4 - t3,
                    t_total=t4 - t0)
    info.update(l=l, k=k, nd=nd, row_blocks=len(blocks))
    return SvdResult(Ut=Ut, s=s, Vh=Vh, info=info)


# ---------------------------------------------------------------------------
# "standard", snapshot matrix larger than the HBM: two streaming passes
# ---------------------------------------------------------------------------
def svd_snapshots_streaming(pieces, n_components: int, rows_global: int, delay: int = 1,
                            oversample: int | None = None, flip_sign: bool = True,
                            comm: Comm | None = None, kern=None, deflate_mean: bool | None = None) -> tuple:
    """Method of snapshots for an X that does not fit the HBM: ``pieces()`` is called once per pass
    and yields, piece after piece, lists of (n, mb) row blocks (pre-processed as the configuration
    asks; resident only until the next piece is asked for, so they may be modified in place).
    Pass 1 accumulates the Gram, the last pass projects: only the m x l basis U' stays resident
    (l = k + max(8, k/4) columns against X's n).

    Same arithmetic as :func:`svd_snapshots` on the concatenation of all pieces -- Gram in fp64
    across blocks, top-l eigenpairs, U' = E V S^-1, Rayleigh-Ritz refinement, sign convention --
    including its two refinements, each at the price of what it needs here:
    * a dominant per-row time mean (un-centred data; ``deflate_mean`` None = detected on the first
      piece of every rank, one exchange) is deflated exactly: the pieces are row-centred in place
      as they pass (K5 in pass 1, which also yields w = A^T mu and |mu|^2 with one extra
      single-column product per piece; the stored means -- 4 bytes per space point -- are
      subtracted again in the later passes), no extra pass;
    * a spectrum steeper than the Gram resolves (lambda_k < 1e-7 lambda_1) gets the polish step
      V <- orth(E^T (E V)) as ONE extra pass (both products while a piece is resident).
    ``rows_global``: rows of the embedded matrix over all ranks (caps k like numpy's slicing).
    Returns (Ublocks, s, Vh, info): ``Ublocks[i]`` = list of (k, d*mb) tensors of piece i."""
    kern = _kern(kern)
    comm = comm or Comm()
    info: dict = {"streaming": True, "passes_over_X": 2}
    G = None
    mus: list | None = None          # per block, in the order the pieces yield them
    w = None
    musq = None
    for pi, blocks in enumerate(pieces()):
        blocks = list(blocks)
        if pi == 0:
            if deflate_mean is None:
                st = _shard_stats(blocks, comm, delay, True)
                deflate_mean = blocks[0].shape[0] - delay + 1 > 2 and st["mean2"] > 100.0 * st["var"]
            if deflate_mean:
                mus = []
        if mus is not None:
            mp = [kern.row_center_scale_(B, False)[0] for B in blocks]
            wp = _gemm_tn_blocks([mu[None, :].contiguous() for mu in mp], blocks, kern, Comm()).reshape(-1)
            w = wp if w is None else w + wp
            sq = torch.stack([(mu.double() ** 2).sum() for mu in mp]).sum().reshape(1)
            musq = sq if musq is None else musq + sq
            mus.extend(mp)
        G = kern.syrk_blocks(blocks, out=G) if (G is not None or len(blocks) > 1) else kern.syrk(blocks[0])
    if G is None:
        raise ValueError("svd_snapshots_streaming: no pieces")
    n_t = G.shape[0]
    if comm.exchanges and n_t >= 64 and hasattr(kern, "pack_triu"):
        G = kern.unpack_triu(comm.allreduce_sum_(kern.pack_triu(G)), n_t, out=G)
    else:
        comm.allreduce_sum_(G)
    if mus is not None:
        comm.allreduce_sum_(w)
        comm.allreduce_sum_(musq)
    if not bool(torch.isfinite(torch.diagonal(G)).all()):
        raise np.linalg.LinAlgError("SVD did not converge")
    if delay > 1:
        G = kern.delay_shift_sum(G, delay)
        if mus is not None:
            ndw = w.numel() - delay + 1
            w = torch.stack([w[kd:kd + ndw] for kd in range(delay)]).sum(dim=0)
    nd = G.shape[0]
    dev = G.device
    k = min(n_components, nd, rows_global)
    p = oversample if oversample is not None else max(8, k // 4)
    l = min(nd, k + p)
    lam_d = None
    if mus is None:
        lam, V = top_eigh(G, l, info=info, kern=kern)
    else:
        lam, V, lam_d, l = _eig_mean_deflated(G, w, musq, delay, l, "auto", info, kern)
    comm.broadcast_(lam, V)
    lam1 = lam[0].clamp_min(1e-300)
    ref = lam[1] if (mus is not None and lam.numel() > 1) else lam1

    def centred(blocks, off):
        """The piece as pass 1 saw it: the stored row means subtracted again (in place)."""
        blocks = list(blocks)
        if mus is not None:
            for j, B in enumerate(blocks):
                B -= mus[off + j]
        return blocks

    polish = float(lam[min(k, lam.numel()) - 1]) < 1e-7 * float(ref)
    if polish:
        # one subspace iteration on X itself (see svd_snapshots): Z = E^T (E V), piece by piece
        Vt32 = _pitched(kern, V.T.contiguous().to(torch.float32))
        ones_v = V.sum(dim=0).to(torch.float32)
        Zt, my, off = None, torch.zeros(V.shape[1], dtype=torch.float64, device=dev), 0
        for blocks in pieces():
            blocks = centred(blocks, off)
            Eb = [embed_view(B, delay) for B in blocks]
            Yb = [kern.skinny(E, Vt32) for E in Eb]
            if mus is not None:
                for j, Y in enumerate(Yb):
                    mu = mus[off + j]
                    Y.addmm_(ones_v[:, None], mu.repeat(delay)[None, :])
                    my += Y.double() @ mu.double().repeat(delay)
            Zt = kern.gemm_tn_blocks(Eb, Yb, out=Zt) if (Zt is not None or len(Eb) > 1) else kern.gemm_tn(Eb[0], Yb[0])
            off += len(blocks)
            del Yb
        comm.allreduce_sum_(Zt)
        if mus is not None:
            comm.allreduce_sum_(my)
            Zt = Zt + my[:, None]
        Zn = Zt / torch.linalg.vector_norm(Zt, dim=1, keepdim=True).clamp_min(1e-300)
        V = _orth(Zn.T.contiguous(), kern=kern)
        comm.broadcast_(V)
        lam = torch.ones_like(lam)
        info["polished"] = True
        info["passes_over_X"] = 3
    good = lam > lam1 * 1e-14
    if polish:
        good = torch.ones_like(lam, dtype=torch.bool)
    elif mus is not None:
        good = lam > torch.maximum(lam_d[0].abs() * 1e-13, lam1 * 1e-15)
    s0 = torch.sqrt(torch.where(good, lam, torch.ones_like(lam)))
    inv_s0 = torch.where(good, 1.0 / s0, torch.zeros_like(s0))
    s0 = torch.where(good, s0, torch.zeros_like(s0))
    Wt = _pitched(kern, (V * inv_s0).T.contiguous().to(torch.float32))
    cvec = (V.sum(dim=0) * inv_s0).to(torch.float32)
    Up, off = [], 0
    for blocks in pieces():                                   # U' = E V S^-1
        blocks = centred(blocks, off)
        piece = [kern.skinny(embed_view(B, delay), Wt) for B in blocks]
        if mus is not None:
            for j, U in enumerate(piece):
                U.addmm_(cvec[:, None], mus[off + j].repeat(delay)[None, :])
        Up.append(piece)
        off += len(blocks)
    flat = [U for piece in Up for U in piece]
    Mm = _gram_blocks(flat, kern, comm)
    mu_, Z = _graded_eigh(s0, Mm, kern)
    mu_, Z = mu_[:k].contiguous(), Z[:, :k].contiguous()
    comm.broadcast_(mu_, Z)
    s = torch.sqrt(mu_.clamp_min(0.0))
    ok = s > s0[0] * 1e-7 if (mus is None and not polish) else s > s[0] * 1e-7
    inv_s = torch.where(ok, 1.0 / torch.where(ok, s, torch.ones_like(s)), torch.zeros_like(s))
    Rt = _pitched(kern, ((s0[:, None] * Z) * inv_s[None, :]).T.contiguous().to(torch.float32))
    Ub = []
    for piece in Up:                      # U = U' R, piece by piece; U' of the piece is dropped right away
        Ub.append([kern.skinny(U, Rt) for U in piece])
        piece.clear()
    Vh = (V @ Z).T.contiguous()
    if flip_sign:
        _, Vh = _sign_flip([U for piece in Ub for U in piece], Vh, comm, kern)
    info.update(l=l, k=k, nd=nd)
    return Ub, s, Vh, info


def svd_randomized_streaming(pieces, n_components: int, rows_global: int, n_time: int, delay: int = 1,
                             n_oversamples: int = 10, n_iter="auto", omega=None, random_state=None,
                             flip_sign: bool = True, comm: Comm | None = None, kern=None) -> tuple:
    """Randomized SVD for an X that does not fit the HBM (``pieces()`` as in
    :func:`svd_snapshots_streaming`): ONE pass over X per power iteration -- while a piece is
    resident, Y_p = E_p Q and Z += E_p^T Y_p are both formed and Y_p is dropped -- plus one pass
    for the final Y = E Q (kept: m x l) and one for B = Q^T E; n_iter + 2 passes instead of the
    2 n_iter + 2 of the resident path.  The price: Y is not orthonormalised between the two half
    steps (only Q = orth(Z) once per iteration), i.e. sklearn's iterate with one normalisation per
    full step -- same subspaces in exact arithmetic, a conditioning of (s_1 / s_l)^2 per step
    instead of s_1 / s_l, which fp32 storage of Y carries for the spectra this path is for.
    Returns (Ublocks, s, Vh, info) like the streaming standard path."""
    kern = _kern(kern)
    comm = comm or Comm()
    nd = n_time - delay + 1
    k = min(n_components, nd, rows_global)
    l = min(n_components + n_oversamples, nd, rows_global)
    n_it = resolve_n_iter(n_components, rows_global, nd, n_iter)
    if omega is None:
        rs = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
        omega = rs.normal(size=(nd, n_components + n_oversamples))[:, :l]
    omega = torch.as_tensor(np.ascontiguousarray(np.asarray(omega).T, dtype=np.float32)) if not isinstance(omega, torch.Tensor) \
        else omega.T.contiguous().to(torch.float32)
    if tuple(omega.shape) != (l, nd):
        raise ValueError(f"omega must be ({nd}, {l}), got {tuple(omega.shape)[::-1]}")
    Qt, dev = None, None
    for _ in range(n_it):
        Z = None
        for blocks in pieces():
            if Qt is None:
                dev = blocks[0].device
                Qt = omega.to(dev)
            Qp = _pitched(kern, Qt)
            Eb = [embed_view(B, delay) for B in blocks]
            Yb = [kern.skinny(E, Qp) for E in Eb]
            Z = kern.gemm_tn_blocks(Eb, Yb, out=Z) if (Z is not None or len(Eb) > 1) else kern.gemm_tn(Eb[0], Yb[0])
            del Yb
        comm.allreduce_sum_(Z)
        if not bool(torch.isfinite(Z).all()):
            raise np.linalg.LinAlgError("SVD did not converge")
        Zn = Z / torch.linalg.vector_norm(Z, dim=1, keepdim=True).clamp_min(1e-300)
        Qt = comm.broadcast_(_orth(Zn.T.contiguous(), kern=kern).T.contiguous().to(torch.float32))
    Yp, bounds = [], [0]
    for blocks in pieces():
        if Qt is None:
            dev = blocks[0].device
            Qt = omega.to(dev)
        Qp = _pitched(kern, Qt)
        Yp.extend(kern.skinny(embed_view(B, delay), Qp) for B in blocks)
        bounds.append(len(Yp))
    Qm = _cholqr(Yp, comm, kern, passes=2)                    # orthonormal basis of range(Y), all pieces
    Bm = None
    for pi, blocks in enumerate(pieces()):                     # B = Q^T E, piece by piece
        Eb = [embed_view(B, delay) for B in blocks]
        Qb = Qm[bounds[pi]:bounds[pi + 1]]
        Bm = kern.gemm_tn_blocks(Eb, Qb, out=Bm) if (Bm is not None or len(Eb) > 1) else kern.gemm_tn(Eb[0], Qb[0])
    comm.allreduce_sum_(Bm)
    Uhat, s, Vh = _svd_wide(Bm, kern)
    Uhat, s, Vh = Uhat.contiguous(), s.contiguous(), Vh.contiguous()
    comm.broadcast_(Uhat, s, Vh)
    Uk = _pitched(kern, Uhat[:, :k].T.contiguous().to(torch.float32))
    Ub = [[kern.skinny(Q, Uk) for Q in Qm[bounds[pi]:bounds[pi + 1]]] for pi in range(len(bounds) - 1)]
    s, Vh = s[:k], Vh[:k].contiguous()
    if flip_sign:
        _, Vh = _sign_flip([U for piece in Ub for U in piece], Vh, comm, kern)
    return Ub, s, Vh, {"streaming": True, "l": l, "k": k, "nd": nd, "n_iter": n_it, "passes_over_X": n_it + 2}


# ---------------------------------------------------------------------------
# "randomized": sklearn's range finder with CholeskyQR normalisers
# ---------------------------------------------------------------------------
def _project_with_gram(kern, Eb, Qp, comm: Comm):
    """[E_b Q for every row block] and, where the provider can fuse it into the same launches
    (K2 with l <= 96), the Gram G = sum_b Y_b^T Y_b of the result, summed over the ranks: what the
    CholeskyQR round that follows starts from -- otherwise None (the round computes it itself, one
    more pass over the m x l matrix)."""
    l = int(Qp.shape[0])
    if l <= getattr(kern, "skinny_gram_max_l", 0):
        G0 = torch.zeros((l, l), dtype=torch.float64, device=Eb[0].device)
        Yb = [kern.skinny(E, Qp, gram=G0) for E in Eb]
        return Yb, comm.allreduce_sum_(G0, tag="small_gram_allreduce")
    return [kern.skinny(E, Qp) for E in Eb], None


def _cholqr(Yb, comm: Comm, kern, passes: int = 1, G0: torch.Tensor | None = None):
    """Orthonormalise the columns of the tall matrix Y given as row blocks
    (each (l, M_b)): G = Y^T Y (l x l, summed over blocks and ranks; ``G0``: that Gram for the
    first pass, when the launches that produced Y already formed it), G = R^T R, Y <- Y R^-1.

    CholeskyQR needs cond(Y)^2 below the accuracy of the fp32-product Gram (~1e8).  Oversampled
    range-finder blocks reach past the numerical rank of X, so when the factorisation fails (or
    its pivots collapse) the pass is redone as *shifted* CholeskyQR (Fukaya et al. 2020):
    factor G + s I with s ~ 1e-6 trace(G), which caps the conditioning of Y R^-1 at ~1e3, and one
    extra plain pass is appended to restore orthonormality.  The span of Y is unchanged."""
    todo, done = passes, 0
    while todo > 0 and done < passes + 3:
        done += 1
        todo -= 1
        G = G0 if (G0 is not None and done == 1) else _gram_blocks(Yb, kern, comm)   # G0: fused into the K2 launches
        # (the Gram of fp32 products can be indefinite by more than 1e-6 of its trace -- identical
        # values accumulate their rounding coherently -- so the shift escalates: _chol_rinv)
        fac = _chol_rinv(G, comm, kern)
        if fac is None:
            return Yb      # Y is exactly zero (X = 0): nothing to orthonormalise, s comes out 0
        Rinv, shifted, _ = fac
        if shifted:
            todo += 1
        Rt = _pitched(kern, Rinv.T.contiguous().to(torch.float32))
        # block by block, dropping each input as soon as its output exists: the peak stays at one
        # m x l matrix + one block instead of two matrices (13.7 GB each at cfg4, next to 227 GB of X)
        # (the caller's list is overwritten in place -- it is consumed by this function)
        for i in range(len(Yb)):
            Yb[i] = kern.skinny(Yb[i], Rt)
    return Yb


def _chol_rinv(G: torch.Tensor, comm: Comm, kern=None):
    """R^-1 (upper triangular, fp64) of the Cholesky factorisation G = R^T R of the Gram matrix of a
    tall block Y, plus how far it can be trusted: -> (Rinv, shifted, spread) with ``spread`` =
    min / max of R's diagonal (~ 1 / cond(Y)).  The Gram of fp32 products of an oversampled
    range-finder block can be numerically indefinite; then G + s I is factored instead with the
    smallest s of an escalating list that works (shifted CholeskyQR, Fukaya et al. 2020: Y R^-1 is
    then not orthonormal but has a condition number <= ~1e3) and ``shifted`` is True.  None if Y is
    exactly zero (X = 0).  Rank 0's factor is made authoritative (one small broadcast).
    One launch of K10 per factorisation (Cholesky + triangular inverse), one read-back of its status."""
    G = 0.5 * (G + G.T)
    L, Linv, info = _chol(kern, G)
    st, dmin, dmax = info.tolist()
    spread = dmin / max(dmax, 1e-300) if (math.isfinite(dmin) and math.isfinite(dmax)) else 0.0
    bad = st != 0.0 or not (spread >= 1e-4)
    shifted = False
    if bad:
        tr = float(torch.diagonal(G).sum())
        if tr == 0.0 and bool(torch.isfinite(G).all()):
            return None
        for rel in (1e-6, 3e-5, 1e-3, 3e-2):
            L, Linv, info = _chol(kern, G, shift=rel * tr)
            st, dmin, dmax = info.tolist()
            if st == 0.0 and math.isfinite(dmin) and math.isfinite(dmax) and dmin > 0.0:
                break
        else:
            raise np.linalg.LinAlgError("CholeskyQR: the Gram matrix of the range-finder block is not finite")
        shifted = True
        spread = dmin / max(dmax, 1e-300)
    Rinv = Linv.T.contiguous()   # R = L^T, R^-1 = (L^-1)^T
    comm.broadcast_(Rinv)
    return Rinv, shifted, float(spread)


def resolve_n_iter(n_components: int, m: int, n: int, n_iter="auto") -> int:
    """extmath.py:557-560."""
    if n_iter == "auto":
        return 7 if n_components < 0.1 * min(m, n) else 4
    return int(n_iter)


@_magnitude_guard(with_mean=False)
def svd_randomized(Xt, n_components: int, delay: int = 1, n_oversamples: int = 10,
                   n_iter="auto", power_iteration_normalizer: str = "auto",
                   omega: np.ndarray | torch.Tensor | None = None, random_state=None,
                   flip_sign: bool = True, comm: Comm | None = None, kern=None,
                   timings: bool = False, _stats: dict | None = None) -> SvdResult:
    """Randomized SVD (Halko et al.) as sklearn runs it for m >= n.

    omega: optional (n_eff, k+p) test matrix; default
    ``RandomState(random_state).normal(size=(n_eff, k+p))`` -- the very draw
    sklearn makes (extmath.py:297), so a seeded run is comparable entry by entry.
    """
    kern = _kern(kern)
    comm = comm or Comm()
    info: dict = {}
    blocks = as_blocks(Xt)
    dev = blocks[0].device
    Eb = [embed_view(B, delay) for B in blocks]
    nd = Eb[0].shape[0]
    if _stats is None:
        _stats = _shard_stats(blocks, comm, delay, False)
    Mg = _stats["rows"]
    if Mg < nd:
        raise ValueError("svd_randomized expects a tall matrix (space >= time); "
                         "transpose first (see svd_device)")
    k = min(n_components, nd)
    l = min(nd, n_components + n_oversamples)
    n_it = resolve_n_iter(n_components, Mg, nd, n_iter)
    asked_none = power_iteration_normalizer == "none"
    if power_iteration_normalizer == "auto":
        power_iteration_normalizer = "none" if n_it <= 2 else "LU"
    # The iterates are re-orthonormalised (CholeskyQR is stable only for moderately conditioned
    # blocks, and it costs two passes over an m x l matrix, nothing next to a pass over X) -- also
    # where sklearn's "auto" would skip it (n_iter <= 2): its un-normalised fp32 power iterations
    # lose the trailing directions (93 % error on sigma_i = 0.9^i data, tests/golden).  In exact
    # arithmetic this spans the same subspaces as sklearn's "none" / "LU" / "QR" choices.  Only an
    # EXPLICIT power_iteration_normalizer="none" is taken literally (sklearn's arithmetic, its
    # loss of the trailing directions included).
    normalise = not asked_none
    if omega is None:
        rs = random_state if isinstance(random_state, np.random.RandomState) else \
            np.random.RandomState(random_state)
        omega = rs.normal(size=(nd, n_components + n_oversamples))[:, :l]
    if isinstance(omega, np.ndarray):
        omega = torch.from_numpy(np.ascontiguousarray(omega.T, dtype=np.float32))
        Qt = omega.to(dev)
    else:
        Qt = omega.T.contiguous().to(device=dev, dtype=torch.float32)
    if tuple(Qt.shape) != (l, nd):
        raise ValueError(f"omega must be ({nd}, {l}), got {tuple(Qt.shape)[::-1]}")
    t0 = _sync_time(dev) if timings else 0.0
    # The (k, M) result is allocated FIRST: requested at the end, next to X and the m x l iterates
    # (cfg4: 227.6 + 13.7 GB resident, 12.5 GB wanted), the caching allocator finds its one cached
    # block of that size cut up by the 119 Y blocks of the iterations, hands cached segments back to
    # the driver and asks again -- 350 ms with the GPU idle at the end of every SVD.
    U_out = None
    if delay == 1 and len(Eb) > 1:
        U_out = torch.empty((k, sum(int(E.shape[1]) for E in Eb)), dtype=torch.float32, device=dev)

    # Normalisation of the iterates WITHOUT a pass over the m x l matrix (round 3).  sklearn
    # normalises Y = X Q before the next product (extmath.py:349-351); with CholeskyQR that is
    # Y R^-1, G = Y^T Y = R^T R.  But X^T (Y R^-1) = (X^T Y) R^-1: the triangular factor is applied to
    # the small n x l result instead (fp64), Y itself is only ever read by K3 -- the same subspace,
    # one fp32 rounding of Y less, and at cfg4 (l = 220) four passes over a 13.7 GB matrix less
    # per SVD.  G comes out of the K2 launches that produce Y (fused Gram, l <= 224).
    phase = _PhaseClock(dev, timings)
    for _ in range(n_it):
        Qp = _pitched(kern, Qt)
        if normalise:
            Yb, G0 = _project_with_gram(kern, Eb, Qp, comm)   # Y = X Q (extmath.py:350) + its Gram
            if G0 is None:
                G0 = _gram_blocks(Yb, kern, comm)
            phase("k2")
            fac = _chol_rinv(G0, comm, kern)
            phase("small")
        else:
            Yb, fac = [kern.skinny(E, Qp) for E in Eb], None
            phase("k2")
        Zt = _gemm_tn_blocks(Eb, Yb, kern, comm)  # Z = X^T Y, (l, nd) (extmath.py:351)
        del Yb
        phase("k3")
        if normalise:
            if fac is not None:
                Zt = fac[0].T @ Zt                # Z R^-1, in the (l, nd) layout
            Qt = _orth(Zt.T.contiguous(), kern=kern).T.contiguous().to(torch.float32)
        else:
            Qt = Zt.to(torch.float32)
        comm.broadcast_(Qt)
        phase("small")
    # Final basis (extmath.py:355, `Q, _ = qr(A @ Q)`): CholeskyQR2 with its LAST triangular factor
    # deferred the same way.  Y_1 = Y R_0^-1 is formed explicitly (one K2 pass over Y, its Gram fused
    # in); if Y_1 is well conditioned (it is nearly orthonormal unless R_0 came from a shifted
    # factorisation, in which case one more explicit pass follows) the second factor R_1 is never
    # applied to the tall matrix: B = Q^T X = R_1^-T (Y_1^T X) and U = Q Uhat = Y_1 (R_1^-1 Uhat).
    Qp = _pitched(kern, Qt)
    Yb, G = _project_with_gram(kern, Eb, Qp, comm)        # Y_0 = X Q
    if G is None:
        G = _gram_blocks(Yb, kern, comm)
    phase("k2")
    fused = int(G.shape[0]) <= getattr(kern, "skinny_gram_max_l", 0)
    Rlast = None
    for explicit in range(4):
        fac = _chol_rinv(G, comm, kern)
        if fac is None:                                    # Y is exactly zero (X = 0): s comes out 0
            break
        Rinv, shifted, spread = fac
        if explicit >= 1 and not shifted and spread > 0.5:
            Rlast = Rinv                                   # well conditioned: deferred
            break
        Rt = _pitched(kern, Rinv.T.contiguous().to(torch.float32))
        G = torch.zeros_like(G) if fused else None
        for i in range(len(Yb)):                           # Y_{j+1} = Y_j R_j^-1, block by block, Gram fused in
            Yb[i] = kern.skinny(Yb[i], Rt, gram=G) if fused else kern.skinny(Yb[i], Rt)
        G = comm.allreduce_sum_(G, tag="small_gram_allreduce") if fused else _gram_blocks(Yb, kern, comm)
    info["cholqr_explicit_passes"] = explicit
    phase("cholqr")
    Qmb = Yb
    Bm = _gemm_tn_blocks(Eb, Qmb, kern, comm)    # (l, nd) = Y_1^T X
    phase("k3")
    if Rlast is not None:
        Bm = Rlast.T @ Bm                        # = Q^T X    (extmath.py:577)
    Uhat, s, Vh = _svd_wide(Bm, kern)
    Uhat, s, Vh = Uhat.contiguous(), s.contiguous(), Vh.contiguous()
    comm.broadcast_(Uhat, s, Vh)
    Uk = Uhat[:, :k] if Rlast is None else Rlast @ Uhat[:, :k]
    Uk = _pitched(kern, Uk.T.contiguous().to(torch.float32))
    phase("small")
    Ub, whole = _project_blocks(kern, Qmb, Uk, delay, whole=U_out)   # U = Q Uhat = Y_1 (R_1^-1 Uhat)
    # directions below the resolution of the data (exactly rank-deficient input: the shifted
    # CholeskyQR leaves no orthonormal basis there): an orthonormal completion, as in svd_snapshots
    ok = s[:k] > 1e-7 * s[0]
    if not bool(ok.all()):
        _complete_unit_rows(Ub, ok, comm)
        info["completed_directions"] = int((~ok).sum())
    phase("project")
    if timings:
        info["phase_ms"] = {k_: v_ * 1e3 for k_, v_ in phase.acc.items()}
    s = s[:k]
    Vh = Vh[:k].contiguous()
    if flip_sign:
        Ub, Vh = _sign_flip(Ub, Vh, comm, kern)
    Ut = whole if whole is not None else _assemble_rows(Ub, delay)
    if timings:
        info["t_total"] = _sync_time(dev) - t0
    info.update(l=l, k=k, nd=nd, n_iter=n_it, normalizer=power_iteration_normalizer,
                passes_over_X=2 * n_it + 2, row_blocks=len(blocks))
    return SvdResult(Ut=Ut, s=s, Vh=Vh, info=info)
