"""Python face of the C ABI (include/dmdx.h): torch tensors in, torch tensors out.

Layout convention used everywhere in this package
-------------------------------------------------
The C ABI is column-major ``(ptr, rows, cols, ld)``.  A column-major matrix
``M`` (rows x cols) is held on the device as the torch tensor ``Mt`` of shape
``(cols, rows)`` with strides ``(ld, 1)`` -- i.e. the row-major transpose, so
``Mt[j, i] == M[i, j]``.  The snapshot matrix X (space x time) is therefore a
``(time, space)`` tensor: one snapshot per row, exactly the order an ERA5
NetCDF slice is stored in.  The delay-embedded matrix (reference
slice_tools.py:207-211) is the overlapping view
``Xt.as_strided((n-d+1, d*m), (m, 1))`` -- zero copy.

PyTorch is plumbing here (device memory, streams); the arithmetic is in
libdmdx.so.  There is no CPU fallback: every method raises if the library or
the GPU is missing.
"""

from __future__ import annotations

import torch

from . import _lib


def _ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def _check_mat(t: torch.Tensor, dtype, name: str) -> tuple[int, int, int]:
    """-> (rows, cols, ld) of the column-major matrix a (cols, rows) tensor holds."""
    if not t.is_cuda:
        raise _lib.DmdxError(f"{name}: expected a device tensor (no CPU fallback exists)")
    if t.dtype != dtype or t.dim() != 2:
        raise _lib.DmdxError(f"{name}: expected 2-D {dtype}, got {t.dtype} {tuple(t.shape)}")
    if t.shape[1] > 1 and t.stride(1) != 1:
        raise _lib.DmdxError(f"{name}: inner stride must be 1, got {t.stride()}")
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)
    return t.shape[1], t.shape[0], ld


class HipKernels:
    """The product kernel provider (libdmdx.so on the current CUDA/HIP device)."""

    name = "hip"

    def __init__(self):
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.DmdxError("no HIP device visible: the dmdx kernels have no CPU fallback")
        self._ws: dict[int, torch.Tensor] = {}
        # when set to a list, every launch is bracketed by HIP events on the launch
        # stream and (name, shape, start, stop) is appended (bench.py reads these)
        self.events: list | None = None

    # -- plumbing ---------------------------------------------------------
    def _stream(self) -> int:
        return torch.cuda.current_stream().cuda_stream

    def _timed(self, name: str, shape: tuple, fn):
        if self.events is None:
            return fn()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn()
        e1.record()
        self.events.append((name, shape, e0, e1))
        return rc

    def _workspace(self, device: torch.device, nbytes: int) -> torch.Tensor:
        # one scratch buffer per (device, stream): launches on different streams may overlap
        idx = device.index if device.index is not None else torch.cuda.current_device()
        key = (idx, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            self._ws[key] = ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        return ws

    def release_workspace(self) -> int:
        """Drop the cached partial-tile workspaces (K1 / K3: 128 KB per (row block, K-split, tile)
        of a launch -- 5.1 GB for cfg2's Gram, 9.5 GB for a cfg3 shard; they are kept between calls
        because re-allocating them costs more than the eigen stage).  Callers that are about to
        fill the HBM with something else (main() before a larger slice, the 227 GB cfg4 matrix)
        call this first.  Returns the number of bytes released."""
        n = sum(int(w.numel()) for w in self._ws.values())
        self._ws.clear()
        return n

    # -- K1 -----------------------------------------------------------------
    def syrk(self, Xt: torch.Tensor, want32: bool = False, out: torch.Tensor | None = None):
        """G = X^T X (fp64, both triangles).  Xt: (n, m) fp32.  -> G64 [, G32].

        ``out``: an (n, n) fp64 tensor to accumulate into (G64 = out += X^T X): the
        Gram of a row-blocked snapshot matrix is the sum over its blocks."""
        m, n, ld = _check_mat(Xt, torch.float32, "syrk X")
        if out is not None:
            if out.shape != (n, n) or out.dtype != torch.float64 or not out.is_contiguous():
                raise _lib.DmdxError("syrk: out must be a contiguous (n, n) fp64 tensor")
            G64 = out
        else:
            G64 = torch.empty((n, n), dtype=torch.float64, device=Xt.device)
        G32 = torch.empty((n, n), dtype=torch.float32, device=Xt.device) if want32 else None
        nbytes = self._lib.dmdx_syrk_workspace_bytes(m, n)
        ws = self._workspace(Xt.device, nbytes)
        rc = self._timed("syrk", (m, n), lambda: self._lib.dmdx_syrk_f32(
            _ptr(Xt), m, n, ld, _ptr(G64), n, _ptr(G32), n, int(out is not None), _ptr(ws),
            ws.numel(), self._stream()
        ))
        _lib.check(rc, "dmdx_syrk_f32")
        return (G64, G32) if want32 else G64

    def syrk_blocks(self, blocks, out: torch.Tensor | None = None) -> torch.Tensor:
        """G (+)= sum_j X_j^T X_j over a list of (n, m_j) fp32 row blocks, 16 blocks per launch."""
        import ctypes as C

        shapes = [_check_mat(B, torch.float32, "syrk_blocks X") for B in blocks]
        n = shapes[0][1]
        if any(s[1] != n for s in shapes):
            raise _lib.DmdxError("syrk_blocks: the blocks must have the same number of columns")
        dev = blocks[0].device
        if out is not None:
            if out.shape != (n, n) or out.dtype != torch.float64 or not out.is_contiguous():
                raise _lib.DmdxError("syrk_blocks: out must be a contiguous (n, n) fp64 tensor")
            G64 = out
        else:
            G64 = torch.empty((n, n), dtype=torch.float64, device=dev)
        nb = len(blocks)
        ptrs = (C.c_void_p * nb)(*[B.data_ptr() for B in blocks])
        ms = (C.c_int64 * nb)(*[s[0] for s in shapes])
        lds = (C.c_int64 * nb)(*[s[2] for s in shapes])
        ws = self._workspace(dev, self._lib.dmdx_syrk_blocks_workspace_bytes(ms, nb, n))
        rc = self._timed("syrk_blocks", (sum(s[0] for s in shapes), n, nb), lambda: self._lib.dmdx_syrk_blocks_f32(
            ptrs, ms, lds, nb, n, _ptr(G64), n, None, 0, int(out is not None), _ptr(ws), ws.numel(), self._stream()
        ))
        _lib.check(rc, "dmdx_syrk_blocks_f32")
        return G64

    # -- K3 -----------------------------------------------------------------
    def gemm_tn(self, At: torch.Tensor, Bt: torch.Tensor, want32: bool = False,
                out: torch.Tensor | None = None):
        """C = A^T B for K-contiguous A (K x na), B (K x nb).

        At: (na, K), Bt: (nb, K) fp32.  Returns Ct of shape (nb, na) (the
        column-major na x nb C), fp64 [and fp32]."""
        Ka, na, lda = _check_mat(At, torch.float32, "gemm_tn A")
        Kb, nb, ldb = _check_mat(Bt, torch.float32, "gemm_tn B")
        if Ka != Kb:
            raise _lib.DmdxError(f"gemm_tn: K mismatch {Ka} vs {Kb}")
        if out is not None:
            if out.shape != (nb, na) or out.dtype != torch.float64 or not out.is_contiguous():
                raise _lib.DmdxError("gemm_tn: out must be a contiguous (nb, na) fp64 tensor")
            C64 = out
        else:
            C64 = torch.empty((nb, na), dtype=torch.float64, device=At.device)
        C32 = torch.empty((nb, na), dtype=torch.float32, device=At.device) if want32 else None
        nbytes = self._lib.dmdx_gemm_tn_workspace_bytes(Ka, na, nb)
        ws = self._workspace(At.device, nbytes)
        rc = self._timed("gemm_tn", (Ka, na, nb), lambda: self._lib.dmdx_gemm_tn_f32(
            _ptr(At), lda, _ptr(Bt), ldb, Ka, na, nb, _ptr(C64), na, _ptr(C32), na,
            int(out is not None), _ptr(ws), ws.numel(), self._stream(),
        ))
        _lib.check(rc, "dmdx_gemm_tn_f32")
        return (C64, C32) if want32 else C64

    def gemm_tn_blocks(self, Ablocks, Bblocks, out: torch.Tensor | None = None) -> torch.Tensor:
        """C (+)= sum_j A_j^T B_j over lists of K-contiguous row blocks (At_j: (na, K_j), Bt_j:
        (nb, K_j) fp32), 16 blocks per launch.  Returns Ct (nb, na) fp64."""
        import ctypes as C

        if len(Ablocks) != len(Bblocks) or not Ablocks:
            raise _lib.DmdxError("gemm_tn_blocks: two equally long, non-empty lists of blocks")
        sa = [_check_mat(A, torch.float32, "gemm_tn_blocks A") for A in Ablocks]
        sb = [_check_mat(B, torch.float32, "gemm_tn_blocks B") for B in Bblocks]
        na, nb_ = sa[0][1], sb[0][1]
        if any(x[1] != na for x in sa) or any(x[1] != nb_ for x in sb) or any(x[0] != y[0] for x, y in zip(sa, sb)):
            raise _lib.DmdxError("gemm_tn_blocks: inconsistent block shapes")
        dev = Ablocks[0].device
        if out is not None:
            if out.shape != (nb_, na) or out.dtype != torch.float64 or not out.is_contiguous():
                raise _lib.DmdxError("gemm_tn_blocks: out must be a contiguous (nb, na) fp64 tensor")
            C64 = out
        else:
            C64 = torch.empty((nb_, na), dtype=torch.float64, device=dev)
        n = len(Ablocks)
        pa = (C.c_void_p * n)(*[A.data_ptr() for A in Ablocks])
        pb = (C.c_void_p * n)(*[B.data_ptr() for B in Bblocks])
        la = (C.c_int64 * n)(*[x[2] for x in sa])
        lb = (C.c_int64 * n)(*[x[2] for x in sb])
        ks = (C.c_int64 * n)(*[x[0] for x in sa])
        ws = self._workspace(dev, self._lib.dmdx_gemm_tn_blocks_workspace_bytes(ks, n, na, nb_))
        rc = self._timed("gemm_tn_blocks", (sum(x[0] for x in sa), na, nb_, n), lambda: self._lib.dmdx_gemm_tn_blocks_f32(
            pa, la, pb, lb, ks, n, na, nb_, _ptr(C64), na, None, 0, int(out is not None), _ptr(ws), ws.numel(),
            self._stream()
        ))
        _lib.check(rc, "dmdx_gemm_tn_blocks_f32")
        return C64

    # -- K2 -----------------------------------------------------------------
    @staticmethod
    def pitch(Wt: torch.Tensor) -> torch.Tensor:
        """A view of the small operand W (l, n) whose rows start on 16-byte boundaries (a padded
        copy unless it already is one).  K2's 16-byte load path needs that of X, Y *and* W; W is
        small and X is not, so W is re-pitched rather than X sent down the scalar-load path
        (n = 3653, cfg4: 1.6 -> 2.9 TB/s of X).  Callers that loop over row blocks do it once."""
        l, n = Wt.shape
        if l <= 1 or not (Wt.stride(0) % 4 or Wt.data_ptr() % 16 or Wt.stride(1) != 1):
            return Wt
        Wp = torch.zeros((l, (n + 3) // 4 * 4), dtype=Wt.dtype, device=Wt.device)
        Wp[:, :n] = Wt
        return Wp[:, :n]

    @property
    def skinny_gram_max_l(self) -> int:
        return int(self._lib.dmdx_gemm_nn_skinny_gram_max_l())

    def skinny(self, Xt: torch.Tensor, Wt: torch.Tensor, out: torch.Tensor | None = None,
               gram: torch.Tensor | None = None) -> torch.Tensor:
        """Y = X W.  Xt: (n, m), Wt: (l, n) fp32 -> Yt: (l, m) fp32.

        ``out``: an (l, m) fp32 view to write into (inner stride 1, any row stride >= m) -- a
        column slice of the (l, M) result of all row blocks, so that no concatenation follows.
        ``gram``: an (l, l) fp64 tensor (l <= skinny_gram_max_l) that Y^T Y is ADDED to, formed from
        the accumulators inside the same launch (the Gram of the CholeskyQR round that follows)."""
        m, n, ldx = _check_mat(Xt, torch.float32, "skinny X")
        nw, l, ldw = _check_mat(Wt, torch.float32, "skinny W")
        if nw != n:
            raise _lib.DmdxError(f"skinny: W has {nw} rows, X has {n} columns")
        Wt = self.pitch(Wt)
        ldw = _check_mat(Wt, torch.float32, "skinny W")[2]
        if out is not None:
            mo, lo, ldy = _check_mat(out, torch.float32, "skinny out")
            if (mo, lo) != (m, l) or out.device != Xt.device:
                raise _lib.DmdxError(f"skinny: out must be ({l}, {m}) on {Xt.device}, got {tuple(out.shape)}")
            Yt = out
        else:
            Yt, ldy = torch.empty((l, m), dtype=torch.float32, device=Xt.device), m
        if gram is not None:
            if gram.shape != (l, l) or gram.dtype != torch.float64 or not gram.is_contiguous() or gram.device != Xt.device \
                    or l > self.skinny_gram_max_l:
                raise _lib.DmdxError(f"skinny: gram must be a contiguous ({l}, {l}) fp64 tensor on {Xt.device}, "
                                     f"l <= {self.skinny_gram_max_l}")
            ws = self._workspace(Xt.device, self._lib.dmdx_gemm_nn_skinny_gram_workspace_bytes(m, l))
            rc = self._timed("skinny_gram", (m, n, l), lambda: self._lib.dmdx_gemm_nn_skinny_gram_f32(
                _ptr(Xt), m, n, ldx, _ptr(Wt), ldw, l, _ptr(Yt), ldy, _ptr(gram), l, 1, _ptr(ws), ws.numel(),
                self._stream()
            ))
            _lib.check(rc, "dmdx_gemm_nn_skinny_gram_f32")
            return Yt
        rc = self._timed("skinny", (m, n, l), lambda: self._lib.dmdx_gemm_nn_skinny_f32(
            _ptr(Xt), m, n, ldx, _ptr(Wt), ldw, l, _ptr(Yt), ldy, self._stream()
        ))
        _lib.check(rc, "dmdx_gemm_nn_skinny_f32")
        return Yt

    # -- K5 -----------------------------------------------------------------
    def row_center_scale_(self, Xt: torch.Tensor, scale: bool):
        """In place: subtract the per-space-point mean over time (and divide by the
        std, ddof 0).  Xt: (n, m).  -> (mean (m,), std (m,) or None)."""
        m, n, ldx = _check_mat(Xt, torch.float32, "row_center_scale X")
        mean = torch.empty(m, dtype=torch.float32, device=Xt.device)
        std = torch.empty(m, dtype=torch.float32, device=Xt.device) if scale else None
        rc = self._timed("row_center_scale", (m, n), lambda: self._lib.dmdx_row_center_scale_f32(
            _ptr(Xt), m, n, ldx, _ptr(mean), _ptr(std), int(bool(scale)), self._stream()
        ))
        _lib.check(rc, "dmdx_row_center_scale_f32")
        return mean, std

    # -- K6 -----------------------------------------------------------------
    def delay_shift_sum(self, G64: torch.Tensor, d: int, want32: bool = False):
        """Gd[i, j] = sum_{k<d} G[i+k, j+k]."""
        if G64.dtype != torch.float64 or G64.dim() != 2 or G64.shape[0] != G64.shape[1]:
            raise _lib.DmdxError("delay_shift_sum: expected a square fp64 matrix")
        if not G64.is_cuda or not G64.is_contiguous():
            raise _lib.DmdxError("delay_shift_sum: expected a contiguous device tensor")
        n = G64.shape[0]
        nd = n - d + 1
        Gd = torch.empty((nd, nd), dtype=torch.float64, device=G64.device)
        Gd32 = torch.empty((nd, nd), dtype=torch.float32, device=G64.device) if want32 else None
        rc = self._lib.dmdx_delay_shift_sum_f64(
            _ptr(G64), n, n, int(d), _ptr(Gd), nd, _ptr(Gd32), nd, self._stream()
        )
        _lib.check(rc, "dmdx_delay_shift_sum_f64")
        return (Gd, Gd32) if want32 else Gd

    # -- helper -------------------------------------------------------------
    def scale_columns_(self, Yt: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
        """Y[:, j] *= alpha[j]  (Yt: (l, m), alpha: (l,) fp32)."""
        m, l, ldy = _check_mat(Yt, torch.float32, "scale_columns Y")
        if alpha.dtype != torch.float32 or alpha.numel() != l or not alpha.is_cuda:
            raise _lib.DmdxError("scale_columns: alpha must be a device fp32 vector of length l")
        rc = self._lib.dmdx_scale_columns_f32(
            _ptr(Yt), m, l, ldy, _ptr(alpha.contiguous()), self._stream()
        )
        _lib.check(rc, "dmdx_scale_columns_f32")
        return Yt


    # -- K7 -----------------------------------------------------------------
    @property
    def eigh_small_max_n(self) -> int:
        return int(self._lib.dmdx_eigh_small_max_n())

    def eigh_small(self, T: torch.Tensor):
        """Eigenpairs of a small symmetric fp64 device matrix (n <= eigh_small_max_n), one
        launch: -> (w (n,) descending, V (n, n) with the eigenvectors in its columns)."""
        if T.dim() != 2 or T.shape[0] != T.shape[1] or T.dtype != torch.float64 or not T.is_cuda:
            raise _lib.DmdxError("eigh_small: T must be a square fp64 device matrix")
        if T.stride(1) != 1:
            T = T.contiguous()
        n = T.shape[0]
        w = torch.empty(n, dtype=torch.float64, device=T.device)
        V = torch.empty((n, n), dtype=torch.float64, device=T.device)
        info = torch.zeros(1, dtype=torch.int32, device=T.device)
        rc = self._timed("eigh_small", (n,), lambda: self._lib.dmdx_eigh_small_f64(
            _ptr(T), n, T.stride(0), _ptr(w), _ptr(V), n, _ptr(info), self._stream()
        ))
        _lib.check(rc, "dmdx_eigh_small_f64")
        # the kernel stops silently at its sweep limit (30): eigenpairs of a run that did not
        # converge must not be used as exact by the Rayleigh-Ritz steps
        self.last_eigh_sweeps = int(info.item())
        if self.last_eigh_sweeps > 30:      # (31 = no rotation-free sweep within the limit of 30)
            raise _lib.DmdxError(f"eigh_small: no convergence in {self.last_eigh_sweeps} Jacobi sweeps (n = {n})")
        return w, V


    # -- K7L ----------------------------------------------------------------
    @property
    def svd_jacobi_max_n(self) -> int:
        return int(self._lib.dmdx_svd_jacobi_max_n())

    def svd_jacobi(self, Ct: torch.Tensor):
        """One-sided Jacobi SVD of a square fp64 device matrix C, given as ``Ct`` with row c =
        column c of C (2 <= n <= svd_jacobi_max_n), one launch: -> (sigma (n,) descending,
        Zt (n, n) with row j = left singular vector j).  ``Ct`` is overwritten.  Raises if the
        workgroups of the launch could not synchronise or the sweeps did not converge."""
        if Ct.dim() != 2 or Ct.shape[0] != Ct.shape[1] or Ct.dtype != torch.float64 or not Ct.is_cuda \
                or Ct.stride(1) != 1:
            raise _lib.DmdxError("svd_jacobi: Ct must be a square fp64 device matrix with inner stride 1")
        n = Ct.shape[0]
        sigma = torch.empty(n, dtype=torch.float64, device=Ct.device)
        Zt = torch.empty((n, n), dtype=torch.float64, device=Ct.device)
        info = torch.zeros(1, dtype=torch.int32, device=Ct.device)
        ws = torch.empty(self._lib.dmdx_svd_jacobi_workspace_bytes(n), dtype=torch.uint8, device=Ct.device)
        rc = self._timed("svd_jacobi", (n,), lambda: self._lib.dmdx_svd_jacobi_f64(
            _ptr(Ct), n, Ct.stride(0), _ptr(sigma), _ptr(Zt), n, _ptr(info), _ptr(ws), ws.numel(), self._stream()
        ))
        _lib.check(rc, "dmdx_svd_jacobi_f64")
        sweeps = int(info.item())
        if sweeps < 0:
            raise _lib.DmdxError("svd_jacobi: the workgroups of the launch could not synchronise "
                                 "(the device is occupied by another kernel that does not end)")
        if sweeps > 40:                     # (41 = no rotation-free sweep within the limit of 40)
            raise _lib.DmdxError(f"svd_jacobi: no convergence in {sweeps} sweeps (n = {n})")
        self.last_jacobi_sweeps = sweeps
        return sigma, Zt

    # -- K8 -----------------------------------------------------------------
    def symm_skinny(self, G: torch.Tensor, Q: torch.Tensor, shift: float = 0.0,
                    out: torch.Tensor | None = None) -> torch.Tensor:
        """Y = G Q - shift Q for a SYMMETRIC fp64 device matrix G (n, n) and a block Q (n, b):
        the products of the top-eigenpair solver (fp64 MFMA, one pass over G).  Shapes the
        kernel's 16-byte fragment loads cannot take (odd n or b, unaligned views) go through
        the library GEMM -- same result, ~1 ms at n = 8760."""
        if G.dtype != torch.float64 or Q.dtype != torch.float64 or G.dim() != 2 or Q.dim() != 2 \
                or G.shape[0] != G.shape[1] or Q.shape[0] != G.shape[0] or not (G.is_cuda and Q.is_cuda):
            raise _lib.DmdxError("symm_skinny: G (n, n) and Q (n, b) must be fp64 device matrices")
        n, b = Q.shape
        ok = (n >= 2 and b >= 2 and n % 2 == 0 and b % 2 == 0 and G.stride(1) == 1 and Q.stride(1) == 1
              and G.stride(0) % 2 == 0 and Q.stride(0) % 2 == 0 and G.data_ptr() % 16 == 0
              and Q.data_ptr() % 16 == 0 and b <= 4096)
        if not ok:
            Y = torch.addmm(Q, G, Q, beta=-float(shift)) if shift != 0.0 else G @ Q
            if out is not None:
                out.copy_(Y)
                return out
            return Y
        Y = out if out is not None else torch.empty((n, b), dtype=torch.float64, device=G.device)
        if Y.shape != (n, b) or Y.dtype != torch.float64 or Y.stride(1) != 1 or Y.data_ptr() == Q.data_ptr():
            raise _lib.DmdxError("symm_skinny: out must be an (n, b) fp64 tensor that does not alias Q")
        ws = self._workspace(G.device, self._lib.dmdx_symm_skinny_workspace_bytes(n, b))
        rc = self._timed("symm_skinny", (n, b), lambda: self._lib.dmdx_symm_skinny_f64(
            _ptr(G), n, G.stride(0), _ptr(Q), Q.stride(0), b, float(shift), _ptr(Y), Y.stride(0),
            _ptr(ws), ws.numel(), self._stream()
        ))
        _lib.check(rc, "dmdx_symm_skinny_f64")
        return Y

    # -- K9 -----------------------------------------------------------------
    def gemm_tn64(self, A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
        """C = A^T B for tall fp64 device blocks A (n, b1), B (n, b2): the small Gram-type products
        of the eigen stage (fp64 MFMA, one launch + a reduce).  Odd widths / unaligned views go
        through the library GEMM."""
        if A.dtype != torch.float64 or B.dtype != torch.float64 or A.dim() != 2 or B.dim() != 2 \
                or A.shape[0] != B.shape[0] or not (A.is_cuda and B.is_cuda):
            raise _lib.DmdxError("gemm_tn64: A (n, b1) and B (n, b2) must be fp64 device matrices")
        n, b1 = A.shape
        b2 = B.shape[1]
        ok = (b1 >= 2 and b2 >= 2 and b1 % 2 == 0 and b2 % 2 == 0 and A.stride(1) == 1 and B.stride(1) == 1
              and A.stride(0) % 2 == 0 and B.stride(0) % 2 == 0 and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0
              and n >= 256)
        if not ok:
            return A.T @ B
        Cm = torch.empty((b1, b2), dtype=torch.float64, device=A.device)
        ws = self._workspace(A.device, self._lib.dmdx_gemm_tn_f64_workspace_bytes(n, b1, b2))
        rc = self._timed("gemm_tn64", (n, b1, b2), lambda: self._lib.dmdx_gemm_tn_f64(
            _ptr(A), A.stride(0), _ptr(B), B.stride(0), n, b1, b2, _ptr(Cm), b2, _ptr(ws), ws.numel(), self._stream()
        ))
        _lib.check(rc, "dmdx_gemm_tn_f64")
        return Cm

    # -- K10 / K11 --------------------------------------------------------------
    @property
    def chol_max_n(self) -> int:
        return int(self._lib.dmdx_potrf_trtri_max_n())

    def chol_inv(self, A: torch.Tensor, shift: float = 0.0, want_inv: bool = True):
        """A + shift I = L L^T for a symmetric fp64 device matrix (n <= chol_max_n), one launch:
        -> (L (n, n) lower, Linv (n, n) lower or None, info) with ``info`` a 3-vector of device
        doubles: status (0 ok; j + 1 = first bad pivot, outputs finite but meaningless; -1 = the
        launch could not synchronise), min and max of diag(L).  Nothing is read back here: the
        caller queues what follows and looks at ``info`` once."""
        if A.dim() != 2 or A.shape[0] != A.shape[1] or A.dtype != torch.float64 or not A.is_cuda:
            raise _lib.DmdxError("chol_inv: A must be a square fp64 device matrix")
        if A.stride(1) != 1:
            A = A.contiguous()
        n = A.shape[0]
        L = torch.empty((n, n), dtype=torch.float64, device=A.device)
        Linv = torch.empty((n, n), dtype=torch.float64, device=A.device) if want_inv else None
        info = torch.empty(3, dtype=torch.float64, device=A.device)
        ws = torch.empty(self._lib.dmdx_potrf_trtri_workspace_bytes(n), dtype=torch.uint8, device=A.device)
        rc = self._timed("chol_inv", (n,), lambda: self._lib.dmdx_potrf_trtri_f64(
            _ptr(A), n, A.stride(0), float(shift), _ptr(L), n, _ptr(Linv), n, _ptr(info), _ptr(ws), ws.numel(),
            self._stream()
        ))
        _lib.check(rc, "dmdx_potrf_trtri_f64")
        return L, Linv, info

    def gemm_nt64(self, Q: torch.Tensor, Mt: torch.Tensor) -> torch.Tensor:
        """Y = Q Mt^T for a tall fp64 device block Q (n, b1) and a small Mt (b2, b1): fp64 MFMA, one
        launch.  Odd b1 / unaligned views go through the library GEMM."""
        if Q.dtype != torch.float64 or Mt.dtype != torch.float64 or Q.dim() != 2 or Mt.dim() != 2 \
                or Q.shape[1] != Mt.shape[1] or not (Q.is_cuda and Mt.is_cuda):
            raise _lib.DmdxError("gemm_nt64: Q (n, b1) and Mt (b2, b1) must be fp64 device matrices")
        n, b1 = Q.shape
        b2 = Mt.shape[0]
        ok = (b1 >= 2 and b1 % 2 == 0 and Q.stride(1) == 1 and Mt.stride(1) == 1 and Q.stride(0) % 2 == 0
              and Mt.stride(0) % 2 == 0 and Q.data_ptr() % 16 == 0 and Mt.data_ptr() % 16 == 0 and n >= 1 and b2 >= 1)
        if not ok:
            return Q @ Mt.T
        Y = torch.empty((n, b2), dtype=torch.float64, device=Q.device)
        rc = self._timed("gemm_nt64", (n, b1, b2), lambda: self._lib.dmdx_gemm_nt_f64(
            _ptr(Q), Q.stride(0), n, b1, _ptr(Mt), Mt.stride(0), b2, _ptr(Y), b2, self._stream()
        ))
        _lib.check(rc, "dmdx_gemm_nt_f64")
        return Y

    # -- packed upper triangle (the Gram all-reduce of the row-sharded path) -----
    def pack_triu(self, A: torch.Tensor) -> torch.Tensor:
        """Upper triangle of a square fp64 device matrix, row by row: n (n + 1) / 2 doubles."""
        if A.dtype != torch.float64 or A.dim() != 2 or A.shape[0] != A.shape[1] or not A.is_cuda or A.stride(1) != 1:
            raise _lib.DmdxError("pack_triu: expected a square fp64 device matrix with inner stride 1")
        n = A.shape[0]
        packed = torch.empty(n * (n + 1) // 2, dtype=torch.float64, device=A.device)
        _lib.check(self._lib.dmdx_pack_triu_f64(_ptr(A), n, A.stride(0), _ptr(packed), self._stream()),
                   "dmdx_pack_triu_f64")
        return packed

    def unpack_triu(self, packed: torch.Tensor, n: int, out: torch.Tensor | None = None) -> torch.Tensor:
        """The symmetric (n, n) matrix whose upper triangle ``packed`` holds (both triangles written)."""
        if packed.dtype != torch.float64 or packed.numel() != n * (n + 1) // 2 or not packed.is_cuda:
            raise _lib.DmdxError("unpack_triu: packed must hold n (n + 1) / 2 fp64 device values")
        A = out if out is not None else torch.empty((n, n), dtype=torch.float64, device=packed.device)
        _lib.check(self._lib.dmdx_unpack_triu_f64(_ptr(packed.contiguous()), n, _ptr(A), A.stride(0), self._stream()),
                   "dmdx_unpack_triu_f64")
        return A

    # -- optimized DMD ---------------------------------------------------------
    def exp_basis(self, alpha: torch.Tensor, t: torch.Tensor, dtype: torch.dtype, want_w: bool = True):
        """Phi = exp(t alpha^T) (n, r) and W = diag(t) Phi in ``dtype`` (complex64 / complex128) from
        complex128 alpha (r,) and fp64 t (n,) on the device: one launch, exponent in fp64."""
        if alpha.dtype != torch.complex128 or t.dtype != torch.float64 or not (alpha.is_cuda and t.is_cuda):
            raise _lib.DmdxError("exp_basis: alpha complex128 and t float64 device tensors expected")
        if dtype not in (torch.complex64, torch.complex128):
            raise _lib.DmdxError("exp_basis: dtype must be complex64 or complex128")
        n, r = t.numel(), alpha.numel()
        a = torch.view_as_real(alpha.contiguous()).contiguous()
        Phi = torch.empty((n, r), dtype=dtype, device=t.device)
        W = torch.empty((n, r), dtype=dtype, device=t.device) if want_w else None
        rc = self._lib.dmdx_exp_basis(_ptr(a), _ptr(t.contiguous()), n, r, _ptr(Phi), _ptr(W),
                                      int(dtype == torch.complex64), self._stream())
        _lib.check(rc, "dmdx_exp_basis")
        return Phi, W

    # -- measurement aid ------------------------------------------------------
    def clock_probe(self, counters: torch.Tensor | None) -> None:
        """Switch the per-workgroup clock stamps of the batched Gram launch on (3 zeroed device
        uint64 / int64) or off (None): bench.py's calibration block only."""
        _lib.check(self._lib.dmdx_set_clock_probe(_ptr(counters)), "dmdx_set_clock_probe")


_default: HipKernels | None = None


def default_kernels() -> HipKernels:
    """The process-wide HIP kernel provider (raises if library / GPU is missing)."""
    global _default
    if _default is None:
        _default = HipKernels()
    return _default


def release_cached_workspaces() -> int:
    """Free the partial-tile workspaces of the process-wide provider, if one exists (main() does
    after every run: 5-10 GB at cfg2 / cfg3 that the next, possibly larger, slice may need)."""
    return _default.release_workspace() if _default is not None else 0
