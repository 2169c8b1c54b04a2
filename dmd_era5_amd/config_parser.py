"""Validate a config section and derive paths (ref: src/dmd_era5/config_parser.py:14-290).

Same keys in, same keys out, same exception types and message substrings (the
reference's tests match on them: "Missing required field in config: {f}",
"Invalid datetime", "Error parsing delta_time", "Invalid SVD type in config",
"Invalid delay embedding in config", "Invalid number of components in config")."""
from __future__ import annotations

from datetime import datetime, timedelta
from logging import Logger

from .constants import (
    ERA5_PRESSURE_LEVEL_VARIABLES,
    ERA5_PRESSURE_LEVELS,
    ERA5_SINGLE_LEVEL_VARIABLES,
)
from .paths import here

_COMMON = ["source_path", "start_datetime", "end_datetime", "delta_time", "variables", "levels"]
REQUIRED = {
    "era5-download": _COMMON,
    "era5-svd": ["source_path", "variables", "levels", "svd_type", "delay_embedding", "mean_center",
                 "scale", "start_datetime", "end_datetime", "delta_time", "n_components",
                 "save_data_matrix"],
}
SUPPORTED_SVD_TYPES = ["standard", "randomized"]

# "1h", "2d", "1w"; a month is 365//12 days and a year 365 days (config_parser.py:122-128)
_DELTA_UNITS = {
    "h": lambda k: timedelta(hours=k),
    "d": lambda k: timedelta(days=k),
    "w": lambda k: timedelta(weeks=k),
    "m": lambda k: timedelta(days=k * 365 // 12),
    "y": lambda k: timedelta(days=k * 365),
}


def _fail(msg: str, logger: Logger | None, cause: Exception | None = None):
    if logger is not None:
        logger.error(msg)
    if cause is not None:
        raise ValueError(msg) from cause
    raise ValueError(msg)


def validate_time_parameters(parsed: dict) -> None:
    start, end, step = parsed["start_datetime"], parsed["end_datetime"], parsed["delta_time"]
    if end <= start:
        raise ValueError("End datetime must be after start datetime")
    if end - start < step:
        raise ValueError(f"Time range must be at least as long as delta_time.\n{end} - {start} < {step}")
    if step <= timedelta(0):
        raise ValueError("delta_time must be positive.")
    if start > datetime.now():
        raise ValueError("Start date cannot be in the future.")


def parse_delta_time(text: str) -> timedelta:
    unit, count = text[-1].lower(), int(text[:-1])
    if unit not in _DELTA_UNITS:
        raise ValueError(f"Unsupported delta_time format in config: {text}")
    return _DELTA_UNITS[unit](count)


def config_parser(config: dict, section: str, logger: Logger | None = None) -> dict:
    if section not in REQUIRED:
        raise ValueError(f"Section {section} is not currently supported.")
    for name in REQUIRED[section]:
        if name not in config:
            _fail(f"Missing required field in config: {name}", logger)

    out: dict = {"source_path": config["source_path"]}
    try:
        out["start_datetime"] = datetime.fromisoformat(config["start_datetime"])
        out["end_datetime"] = datetime.fromisoformat(config["end_datetime"])
    except ValueError as e:
        _fail(f"Invalid datetime format in config: {e}", logger, e)
    try:
        out["delta_time"] = parse_delta_time(config["delta_time"])
    except ValueError as e:
        _fail(f"Error parsing delta_time from config: {e}", logger, e)
    validate_time_parameters(out)

    try:
        spec = config["variables"]
        if spec == "all_pressure_level_vars":
            out["variables"] = list(ERA5_PRESSURE_LEVEL_VARIABLES)
        elif spec == "all_single_level_vars":
            raise ValueError("Single level variables not currently supported.")
        else:
            out["variables"] = [v.strip() for v in spec.split(",")]
            for v in out["variables"]:
                if v in ERA5_SINGLE_LEVEL_VARIABLES:
                    raise ValueError(f"Single level variables not currently supported: {v}")
                if v not in ERA5_PRESSURE_LEVEL_VARIABLES:
                    raise ValueError(f"Unsupported variable in config: {v}")
    except ValueError as e:
        _fail(f"Error parsing variables from config: {e}", logger, e)
    try:
        if config["levels"] == "all":
            out["levels"] = list(ERA5_PRESSURE_LEVELS)
        else:
            out["levels"] = [int(x) for x in config["levels"].split(",")]
            for lev in out["levels"]:
                if lev not in ERA5_PRESSURE_LEVELS:
                    raise ValueError(f"Unsupported level in config: {lev}")
    except ValueError as e:
        _fail(f"Error parsing levels from config: {e}", logger, e)

    # "{start:%Y-%m-%dT%H}_{end:%Y-%m-%dT%H}_{delta_time as written}.nc" (config_parser.py:201-207)
    stamp = "%Y-%m-%dT%H"
    out["save_name"] = (f"{out['start_datetime'].strftime(stamp)}_"
                        f"{out['end_datetime'].strftime(stamp)}_{config['delta_time']}.nc")
    out_dir = "era5_download" if section == "era5-download" else "era5_svd"
    out["save_path"] = here("data", out_dir, out["save_name"])
    out["era5_slice_path"] = here("data", "era5_download", out["save_name"])
    if section != "era5-svd":
        return out

    out["era5_svd_path"] = here("data", "era5_svd", out["save_name"])
    out["svd_type"] = config["svd_type"]
    if out["svd_type"] not in SUPPORTED_SVD_TYPES:
        _fail(f"Invalid SVD type in config: {out['svd_type']}. "
              f"Supported types: {SUPPORTED_SVD_TYPES}.", logger)

    def _positive_int(key: str, label: str, what: str):
        val = config[key]
        if not isinstance(val, int) or val < 1:  # (bool is an int in Python, as in the reference)
            _fail(f"Invalid {label} in config: {val}. {what} must be an integer greater than 0.", logger)
        out[key] = val

    def _boolean(key: str, label: str):
        val = config[key]
        if not isinstance(val, bool):
            _fail(f"Invalid {label} in config: {val}. It must be a boolean value.", logger)
        out[key] = val

    _positive_int("delay_embedding", "delay embedding", "Delay embedding")
    _boolean("mean_center", "mean centering")
    _boolean("scale", "scaling")
    _positive_int("n_components", "number of components", "Number of components")
    _boolean("save_data_matrix", "save_data_matrix")
    return out
