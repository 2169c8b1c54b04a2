"""Worker of tests/test_gpu_pipeline.py::test_main_sharded_over_two_ranks: one rank of
``main(config, write_to_netcdf=True)`` under torch.distributed.run (config as JSON in DMDX_TEST_CONFIG)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from dmd_era5_amd.era5_svd import main  # noqa: E402

calls = []
if os.environ.get("DMDX_TEST_EXPECT_BACKEND"):
    # count what actually reaches torch.distributed (the one-rank RCCL rehearsal must not be a no-op)
    import torch.distributed as dist

    for name in ("all_reduce", "all_gather", "broadcast"):
        def counted(*a, _f=getattr(dist, name), _n=name, **k):
            calls.append((_n, dist.get_backend(), str(a[0][0].device if isinstance(a[0], list) else a[0].device)))
            return _f(*a, **k)
        setattr(dist, name, counted)

res, _, _ = main(json.loads(os.environ["DMDX_TEST_CONFIG"]), write_to_netcdf=True)
if os.environ.get("DMDX_TEST_EXPECT_BACKEND"):
    want = os.environ["DMDX_TEST_EXPECT_BACKEND"]
    assert {c[0] for c in calls} >= {"all_reduce", "all_gather", "broadcast"}, calls
    assert all(c[1] == want and c[2].startswith("cuda") for c in calls), calls
    print(f"backend {want}: {len(calls)} collectives issued", flush=True)
rank = int(os.environ.get("RANK", "0"))
assert (res is not None) == (rank == 0), "only rank 0 assembles the results"
print(f"rank {rank} done", flush=True)
