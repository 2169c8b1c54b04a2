"""Scratch: a longer differential sweep (same checks as tests/test_gpu_parity.py::test_differential_sweep)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.engine import svd_numpy
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
WIDE = len(sys.argv) > 2 and sys.argv[2] == "wide"      # ranks 97..220: beyond K7's size, K2 column groups, 96-/128-row K3 tiles
rs0 = np.random.RandomState((78 if WIDE else 77) + 1000 * int(os.environ.get("DMDX_FUZZ_SEED", "0")))
bad = 0
for i in range(N):
    m = int(rs0.choice([3, 9, 17, 64, 257, 1000, 4099, 20000, 70000, 300001, 524288])); n = int(rs0.choice([2, 3, 5, 24, 96, 130, 300, 700]))
    if m > 100000 and n > 300: n = 300
    k = int(rs0.randint(1, min(m, n, 60) + 1))
    if WIDE:
        m = int(rs0.choice([1000, 4099, 20000, 70000])); n = int(rs0.choice([300, 500, 700])); k = int(rs0.randint(97, min(n, 220) + 1))
    kind = ["gauss", "lowrank", "deficient", "offset", "graded", "offsetlow", "const", "huge", "tiny", "sparse", "dup"][i % 11]
    typ = "standard" if i % 3 else "randomized"
    rs = np.random.RandomState(5000 + i + 100000 * int(os.environ.get("DMDX_FUZZ_SEED", "0")))
    if kind == "gauss": X = rs.standard_normal((m, n))
    elif kind == "lowrank":
        r = max(1, min(m, n) // 3); X = rs.standard_normal((m, r)) @ (rs.standard_normal((r, n)) * (0.8 ** np.arange(r))[:, None]) + 1e-3 * rs.standard_normal((m, n))
    elif kind == "deficient":
        r = max(1, min(m, n, k) // 2); X = rs.standard_normal((m, r)) @ rs.standard_normal((r, n))
    elif kind == "offset": X = 300.0 + rs.standard_normal((m, 1)) * 5 + rs.standard_normal((m, n))
    elif kind == "offsetlow":
        r = max(1, min(m, n) // 4); X = 250.0 + 10 * rs.standard_normal((m, 1)) + rs.standard_normal((m, r)) @ (rs.standard_normal((r, n)) * (0.7 ** np.arange(r))[:, None])
    elif kind == "const": X = np.full((m, n), 3.5) + (1e-3 * rs.standard_normal((m, n)) if i % 2 else 0)
    elif kind == "huge": X = 1e22 * rs.standard_normal((m, min(n, 4))) @ rs.standard_normal((min(n, 4), n))
    elif kind == "tiny": X = 1e-24 * rs.standard_normal((m, min(n, 4))) @ rs.standard_normal((min(n, 4), n))
    elif kind == "sparse": X = rs.standard_normal((m, n)) * (rs.rand(m, n) < 0.05)
    elif kind == "dup":
        X = rs.standard_normal((m, n)); X[:, n // 2:] = X[:, : n - n // 2]
    else: X = rs.standard_normal((m, n)) * (0.7 ** np.arange(n))
    X = X.astype(np.float32)
    try:
        U, s, V = svd_numpy(X, typ, k, device="cuda:0", **({"random_state": 0} if typ == "randomized" else {}))
    except Exception as e:
        print("EXC", i, m, n, k, kind, typ, repr(e)[:200]); bad += 1; continue
    X64 = X.astype(np.float64); sref = np.linalg.svd(X64, compute_uv=False); kk = min(k, m, n)
    exact = typ == "standard" or kind in ("lowrank", "deficient", "offsetlow", "const", "huge", "tiny")
    ds = np.abs(s - sref[:kk]).max() / sref[0]
    err = np.linalg.norm(X64 - (U.astype(np.float64) * s) @ V.astype(np.float64)); opt = np.sqrt((sref[kk:] ** 2).sum())
    live = s > float(os.environ.get("DMDX_FUZZ_LIVE", "1e-6")) * s[0]; Ul = U[:, live].astype(np.float64)
    orth = np.abs(Ul.T @ Ul - np.eye(live.sum())).max() if live.any() else 0
    flag = (exact and (ds > 2e-5 or err > 1.001 * opt + 5e-5 * np.linalg.norm(X64))) or orth > 5e-4 or not np.all(np.isfinite(s))
    if i % 50 == 49:
        print("...", i + 1, "cases,", bad, "flagged so far", flush=True)
    if flag:
        bad += 1
        print("BAD", i, m, n, k, kind, typ, "ds/s1 %.2e err %.3e opt %.3e orth %.1e" % (ds, err, opt, orth), flush=True)
print("done", N, "cases,", bad, "flagged")
