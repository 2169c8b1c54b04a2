"""``python -m dmd_era5.era5_svd.era5_svd`` -> dmd_era5_amd.era5_svd.main(write_to_netcdf=True)."""
from dmd_era5_amd.era5_svd import *  # noqa: F401,F403
from dmd_era5_amd.era5_svd import log_and_print, logger, main

if __name__ == "__main__":
    log_and_print(logger, "Not a Data Version Control (DVC) repository. Will not use DVC.", level="warning")
    main(write_to_netcdf=True)
