"""CPU oracle (test infrastructure only -- see era5_oracle.py header)."""
from .era5_oracle import *  # noqa: F401,F403
