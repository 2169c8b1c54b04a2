"""Alias package: ``python -m dmd_era5.era5_svd.era5_svd`` and ``from dmd_era5.era5_svd import
main, svd_on_era5, ...`` resolve to the MI355X engine in :mod:`dmd_era5_amd` (drop-in for the
reference's entry point, ref: src/dmd_era5/era5_svd/era5_svd.py:456-478)."""
from dmd_era5_amd.config_parser import config_parser  # noqa: F401
from dmd_era5_amd.config_reader import config_reader  # noqa: F401
from dmd_era5_amd.logger import log_and_print, setup_logger  # noqa: F401
