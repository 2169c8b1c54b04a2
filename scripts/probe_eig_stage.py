"""Scratch: where the ~26 ms of svd.top_eigh (n = 8760, l = 62) go."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dmd_era5_amd import svd as S
from dmd_era5_amd.kernels import default_kernels
kern = default_kernels()
n, m = 8760, 129780
blocks = bench.make_snapshot_blocks(m, n, 1234, torch.device("cuda"))
for B in blocks: kern.row_center_scale_(B, False)
G = S._gram_blocks(blocks, kern, S.Comm())
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("top_eigh: %.2f ms" % t(lambda: S.top_eigh(G, 62)))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    S.top_eigh(G, 62); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
b = 77
Q = torch.randn(n, b, dtype=torch.float64, device="cuda")
print("G@Q %.2f | orth(77) %.2f | cat+orth(231) %.2f" % (t(lambda: G @ Q), t(lambda: S._orth(Q)), t(lambda: S._orth(torch.cat([Q, Q.roll(1, 0), Q.roll(2, 0)], 1)))))
# residual history of the power / Rayleigh-Ritz steps
l = 62
gen = torch.Generator(device="cuda").manual_seed(1234)
Q = torch.randn((n, b), dtype=torch.float64, generator=gen, device="cuda")
Q = S._orth(G @ Q)
for it in range(4):
    Y = G @ Q; T = Q.T @ Y; T = 0.5 * (T + T.T); th, Z = torch.linalg.eigh(T); th = th.flip(0); Z = Z.flip(1)
    Qn = Q @ Z; R = Y @ Z[:, :l] - Qn[:, :l] * th[:l]
    rr = (torch.linalg.vector_norm(R, dim=0) / th[0])
    L, err = torch.linalg.cholesky_ex(Y.T @ Y)
    print(it, "res max %.3e at %d" % (float(rr.max()), int(rr.argmax())), "th[0,61,63,64,76]", th[[0, 61, 63, 64, 76]].tolist(), "chol err", int(err), "orthdef %.2e" % float((Q.T @ Q - torch.eye(b, dtype=torch.float64, device="cuda")).abs().max()))
    Q = S._orth(Y)
