"""`verl.utils.py_functional` — the small dict helpers the reference's trainer, workers and loggers import
(verl/utils/py_functional.py:50-103)."""
from __future__ import annotations

import importlib.util
from functools import lru_cache
from typing import Any, Dict, List

__all__ = ["append_to_dict", "convert_dict_to_str", "flatten_dict", "is_package_available", "unflatten_dict", "union_two_dict"]


@lru_cache(maxsize=None)
def is_package_available(name: str) -> bool:
    try:
        return importlib.util.find_spec(name) is not None
    except (ImportError, ValueError):
        return False


def union_two_dict(dict1: Dict[str, Any], dict2: Dict[str, Any]) -> Dict[str, Any]:
    """dict1 updated with dict2; a key present in both must hold equal values (AssertionError otherwise, as the reference)."""
    for key, value in dict2.items():
        if key in dict1:
            assert dict1[key] == value, f"{key} in dict1 and dict2 are not the same object"
        dict1[key] = value
    return dict1


def append_to_dict(data: Dict[str, List[Any]], new_data: Dict[str, Any]) -> None:
    for key, value in new_data.items():
        data.setdefault(key, []).append(value)


def unflatten_dict(data: Dict[str, Any], sep: str = "/") -> Dict[str, Any]:
    """{"a/b": 1, "a/c": 2} -> {"a": {"b": 1, "c": 2}}"""
    out: Dict[str, Any] = {}
    for key, value in data.items():
        *parents, leaf = key.split(sep)
        node = out
        for p in parents:
            node = node.setdefault(p, {})
        node[leaf] = value
    return out


def flatten_dict(data: Dict[str, Any], parent_key: str = "", sep: str = "/") -> Dict[str, Any]:
    out: Dict[str, Any] = {}
    for key, value in data.items():
        name = f"{parent_key}{sep}{key}" if parent_key else key
        if isinstance(value, dict):
            out.update(flatten_dict(value, name, sep=sep))
        else:
            out[name] = value
    return out


def convert_dict_to_str(data: Dict[str, Any]) -> str:
    """YAML text of a (nested) dict, floats rounded to 3 decimals unless in scientific notation (the reference registers that float
    representer globally on yaml; here it is local to this dump)."""
    import yaml

    class _Dumper(yaml.Dumper):
        pass

    def _float(dumper, number):
        text = str(number)
        if "e" in text or "E" in text:
            if "." not in text:
                text = text.replace("e", ".0e", 1)
        else:
            text = str(round(float(number), 3))
        return dumper.represent_scalar("tag:yaml.org,2002:float", text)

    _Dumper.add_representer(float, _float)
    try:
        import numpy as np
        _Dumper.add_representer(np.float32, _float)
        _Dumper.add_representer(np.float64, _float)
    except ImportError:                                    # pragma: no cover
        pass
    return yaml.dump(data, Dumper=_Dumper, indent=2)
