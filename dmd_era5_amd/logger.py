"""File + stdout logging with the reference's logger names and line format
(ref: src/dmd_era5/logger.py:7-46)."""
from __future__ import annotations

import logging
import os

from .paths import here

_FORMAT = "%(asctime)s - %(name)s - %(levelname)s - %(message)s"


def setup_logger(name: str, log_file: str, level=logging.INFO) -> logging.Logger:
    """Logger ``name`` writing to ``<project root>/logs/<log_file>`` (handlers of a
    previous setup are dropped, like the reference does)."""
    log_dir = here("logs")
    os.makedirs(log_dir, exist_ok=True)
    handler = logging.FileHandler(os.path.join(log_dir, log_file))
    handler.setFormatter(logging.Formatter(_FORMAT))
    logger = logging.getLogger(name)
    logger.setLevel(level)
    for h in list(logger.handlers):
        logger.removeHandler(h)
    logger.addHandler(handler)
    return logger


def log_and_print(logger: logging.Logger, message: str, level: str = "info") -> None:
    getattr(logger, level.lower())(message)
    print(message)
