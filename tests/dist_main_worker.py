"""Worker of tests/test_gpu_pipeline.py::test_main_sharded_over_two_ranks: one rank of
``main(config, write_to_netcdf=True)`` under torch.distributed.run (config as JSON in DMDX_TEST_CONFIG)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from dmd_era5_amd.era5_svd import main  # noqa: E402

res, _, _ = main(json.loads(os.environ["DMDX_TEST_CONFIG"]), write_to_netcdf=True)
rank = int(os.environ.get("RANK", "0"))
assert (res is not None) == (rank == 0), "only rank 0 assembles the results"
print(f"rank {rank} done", flush=True)
