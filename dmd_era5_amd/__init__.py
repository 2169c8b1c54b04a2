"""dmd_era5_amd -- MI355X-native engine for the ERA5 snapshot-matrix SVD hot path
of ClimeTrend/DMD-ERA5 (drop-in for ``dmd_era5.era5_svd``).  See DESIGN.md."""

__version__ = "0.1.0"
