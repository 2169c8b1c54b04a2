"""Project-root discovery.  The reference resolves every path with
``pyprojroot.here()`` (config.ini, logs/, data/era5_download, data/era5_svd;
ref: src/dmd_era5/config_parser.py:198-216, logger.py:16, config_reader.py:13).
pyprojroot is not installed here, so this is a small restatement of its rule:
walk up from the working directory to the first directory holding a project
marker.  ``DMD_ERA5_ROOT`` overrides (used by the tests)."""
from __future__ import annotations

import os

_MARKERS = (".here", ".git", "pyproject.toml", "setup.py", "config.ini", ".dvc")


def here(*parts: str) -> str:
    root = os.environ.get("DMD_ERA5_ROOT")
    if not root:
        d = os.path.abspath(os.getcwd())
        root = d
        while True:
            if any(os.path.exists(os.path.join(d, mk)) for mk in _MARKERS):
                root = d
                break
            parent = os.path.dirname(d)
            if parent == d:
                break
            d = parent
    return os.path.join(root, *parts)
