import numpy as np, torch, sys
sys.path.insert(0, "/root/repo")
from dmd_era5_amd.engine import svd_numpy
g = np.load("/root/repo/tests/golden/conditioning_2048x160.npz")
for tag in ("raw", "cen"):
    X = g[f"{tag}_X"]; k = int(g["k"])
    for typ in ("standard", "randomized"):
        U, s, V = svd_numpy(X, typ, k, device="cuda:0", **({"random_state": 0} if typ == "randomized" else {}))
        rel = np.abs(s / g[f"{tag}_s64"] - 1)
        print(tag, typ, "max rel err of s", rel.max(), "numpy fp32:", np.abs(g[f"{tag}_s32"] / g[f"{tag}_s64"] - 1).max(), "s[:3]", s[:3])
