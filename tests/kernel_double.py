"""CPU test double for the kernel provider of dmd_era5_amd.svd.

TEST INFRASTRUCTURE: it lets the host-side algorithms (method of snapshots,
randomized range finder, row-shard all-reduces over gloo) run in this GPU-less
container.  It is never importable from the product package; the product's
default provider (dmd_era5_amd.kernels.HipKernels) has no CPU path.
Arithmetic: float64 accumulate on fp32 inputs, like the oracle would.
"""
import torch


class CpuKernelDouble:
    name = "cpu-double"

    def syrk(self, Xt, want32=False, out=None):
        X = Xt.to(torch.float64)
        G = X @ X.T
        if out is not None:
            out += G
            G = out
        return (G, G.float()) if want32 else G

    def syrk_blocks(self, blocks, out=None):
        G = out
        for B in blocks:
            G = self.syrk(B) if G is None else self.syrk(B, out=G)
        return G

    def gemm_tn(self, At, Bt, want32=False, out=None):
        C = Bt.to(torch.float64) @ At.to(torch.float64).T
        if out is not None:
            out += C
            C = out
        return (C, C.float()) if want32 else C

    def gemm_tn_blocks(self, Ablocks, Bblocks, out=None):
        Cm = out
        for A, B in zip(Ablocks, Bblocks):
            Cm = self.gemm_tn(A, B) if Cm is None else self.gemm_tn(A, B, out=Cm)
        return Cm

    skinny_gram_max_l = 224

    def skinny(self, Xt, Wt, out=None, gram=None):
        Y = (Wt.to(torch.float64) @ Xt.to(torch.float64)).float()
        if gram is not None:
            gram += Y.double() @ Y.double().T
        if out is None:
            return Y
        out.copy_(Y)
        return out

    def row_center_scale_(self, Xt, scale):
        mean = Xt.to(torch.float64).mean(dim=0).float()
        Xt -= mean
        std = None
        if scale:
            std = Xt.to(torch.float64).std(dim=0, unbiased=False).float()
            Xt /= std
        return mean, std

    def delay_shift_sum(self, G, d, want32=False):
        n = G.shape[0]
        nd = n - d + 1
        Gd = sum(G[k:k + nd, k:k + nd] for k in range(d)).contiguous()
        return (Gd, Gd.float()) if want32 else Gd

    def scale_columns_(self, Yt, alpha):
        Yt *= alpha[:, None]
        return Yt

    def symm_skinny(self, G, Q, shift=0.0, out=None):
        Y = G @ Q - shift * Q
        if out is not None:
            out.copy_(Y)
            return out
        return Y

    def gemm_tn64(self, A, B):
        return A.T @ B

    def pack_triu(self, A):
        i, j = torch.triu_indices(A.shape[0], A.shape[0])
        return A[i, j].contiguous()

    def unpack_triu(self, packed, n, out=None):
        A = out if out is not None else torch.empty((n, n), dtype=packed.dtype)
        i, j = torch.triu_indices(n, n)
        A[i, j] = packed
        A[j, i] = packed
        return A

    svd_jacobi_max_n = 1024

    def svd_jacobi(self, Ct):
        U, s, _ = torch.linalg.svd(Ct.T)
        return s, U.T.contiguous()

    eigh_small_max_n = 96

    def eigh_small(self, T):
        w, V = torch.linalg.eigh(0.5 * (T + T.T))
        return torch.flip(w, dims=(0,)), torch.flip(V, dims=(1,))
