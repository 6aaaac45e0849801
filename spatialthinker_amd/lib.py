"""ctypes binding of libst_hip.so — the C-ABI boundary declared in include/st_hip.h.

The prototypes are parsed from the header itself, so the binding cannot drift from the
declared ABI.  There is NO fallback: if the library is missing or a symbol is not exported,
import fails loudly (the product path never routes through oracle/ or a CPU path).
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(ROOT, "include", "st_hip.h")
LIB_PATH = os.path.join(_HERE, "libst_hip.so")

_SCALARS = {
    "int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64,
    "float": ctypes.c_float, "double": ctypes.c_double, "st_stream_t": ctypes.c_void_p,
}


def parse_header(path: str = HEADER):
    """Return {name: (restype, [argtypes], [argnames])} for every `st_*` function declared."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    out = {}
    for m in re.finditer(r"\b(int64_t|int|const char\*)\s+(st_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes, argnames = [], []
        args = args.strip()
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                    argnames.append(a.split("*")[-1].strip())
                else:
                    parts = a.replace("const ", "").split()
                    argtypes.append(_SCALARS[parts[0]])
                    argnames.append(parts[-1])
        out[name] = ({"int": ctypes.c_int, "int64_t": ctypes.c_int64}.get(ret, ctypes.c_char_p), argtypes, argnames)
    return out


class StError(RuntimeError):
    pass


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP kernels first (python -c 'import __graft_entry__ as g; g.build()' "
                "or make -C spatialthinker_amd/csrc). There is no CPU fallback.")
        self._dll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (res, argtypes, _names) in self.protos.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError as e:                                   # declared but not exported
                raise ImportError(f"libst_hip.so does not export {name} declared in include/st_hip.h") from e
            fn.restype = res
            fn.argtypes = argtypes
            if res is ctypes.c_int and name not in ("st_version",):
                setattr(self, name, self._checked(name, fn))
            else:
                setattr(self, name, fn)

    @staticmethod
    def _checked(name, fn):
        def call(*args):
            rc = fn(*args)
            if rc != 0:
                raise StError(f"{name} failed with code {rc}" + (" (ST_EINVAL: bad arguments)" if rc == -22 else " (hipError_t)"))
            return rc
        call.__name__ = name
        call.raw = fn
        return call


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
