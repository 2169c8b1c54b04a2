from dmd_era5_amd.era5_svd import (  # noqa: F401
    add_config_attributes,
    combine_svd_results,
    main,
    retrieve_era5_slice,
    retrieve_svd_results,
    svd_on_era5,
)

__all__ = ["svd_on_era5", "combine_svd_results", "retrieve_era5_slice", "retrieve_svd_results",
           "add_config_attributes", "main"]
