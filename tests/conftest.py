import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the mirrored reference modules resolve config.ini / logs / data under the project root
# (pyprojroot.here() in the reference); tests work in a throw-away root
import tempfile  # noqa: E402

os.environ.setdefault("DMD_ERA5_ROOT", tempfile.mkdtemp(prefix="dmd_era5_root_"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def project_root(tmp_path, monkeypatch):
    """A fresh project root (data/, logs/, config.ini live under it)."""
    monkeypatch.setenv("DMD_ERA5_ROOT", str(tmp_path))
    return tmp_path


@pytest.fixture
def svd_base_config():
    # same values as the reference's tests/test_03_era5_svd.py:22-37 base_config
    return {
        "source_path": "gs://gcp-public-data-arco-era5/ar/1959-2022-full_37-1h-0p25deg-chunk-1.zarr-v2",
        "variables": "temperature",
        "levels": "1000",
        "svd_type": "randomized",
        "delay_embedding": 2,
        "mean_center": False,
        "scale": False,
        "start_datetime": "2019-01-01T06",
        "end_datetime": "2020-01-01T12",
        "delta_time": "1h",
        "n_components": 10,
        "save_data_matrix": True,
    }
