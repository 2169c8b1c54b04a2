import numpy as np, sys, traceback, torch
sys.path.insert(0, "/root/repo")
from dmd_era5_amd import svd as dsvd
from dmd_era5_amd.kernels import default_kernels
K = default_kernels()
m, n, k = 20000, 300, 6
X = np.full((m, n), 3.5, dtype=np.float32)
Xt = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
orig = dsvd._cholqr
def traced(Yb, comm, kern, passes=1):
    G = dsvd._gram_blocks(Yb, kern, comm)
    ev = torch.linalg.eigvalsh(0.5 * (G + G.T))
    print("cholqr in: l", G.shape[0], "trace %.3e" % float(torch.diagonal(G).sum()), "eig min/max %.3e %.3e" % (float(ev[0]), float(ev[-1])), "finite", bool(torch.isfinite(G).all()), flush=True)
    return orig(Yb, comm, kern, passes)
dsvd._cholqr = traced
try:
    r = dsvd.svd_randomized(Xt, k, random_state=0, kern=K); print(r.s)
except Exception:
    traceback.print_exc()
