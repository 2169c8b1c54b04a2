"""config.ini -> dict of Python literals (ref: src/dmd_era5/config_reader.py:16-62).

Every value of the section is passed through ``ast.literal_eval`` (strings must be
quoted in the file, booleans/ints come back typed); a missing section raises
``Exception("Section ... not found in the ... file")``."""
from __future__ import annotations

import ast
from configparser import ConfigParser

from .paths import here


def default_config_path() -> str:
    return here("config.ini")


def config_reader(section: str, config_path: str | None = None) -> dict:
    path = config_path or default_config_path()
    parser = ConfigParser()
    parser.read(path, encoding="utf-8-sig")
    if not parser.has_section(section):
        raise Exception(f"Section {section} not found in the {path} file")
    out = {}
    for key, raw in parser.items(section):
        try:
            out[key] = ast.literal_eval(raw)
        except Exception as e:
            print(f"Error while parsing {key} from {section} section in the config file: {e}")
            raise
    return out
