"""`verl.utils.torch_dtypes.PrecisionType` — precision names <-> torch dtypes (reference: verl/utils/torch_dtypes.py:18-70; used by its
workers for `fsdp.torch_dtype` / `mp_param_dtype` strings)."""
import torch

HALF_LIST = [16, "16", "fp16", "float16"]
FLOAT_LIST = [32, "32", "fp32", "float32"]
BFLOAT_LIST = ["bf16", "bfloat16"]

_TO_DTYPE = [(HALF_LIST, torch.float16), (FLOAT_LIST, torch.float32), (BFLOAT_LIST, torch.bfloat16)]
_TO_STR = {torch.float16: "float16", torch.float32: "float32", torch.bfloat16: "bfloat16"}


class PrecisionType:
    HALF, FLOAT, FULL, BFLOAT, MIXED = "16", "32", "64", "bf16", "mixed"

    @staticmethod
    def is_fp16(precision) -> bool:
        return precision in HALF_LIST

    @staticmethod
    def is_fp32(precision) -> bool:
        return precision in FLOAT_LIST

    @staticmethod
    def is_bf16(precision) -> bool:
        return precision in BFLOAT_LIST

    @staticmethod
    def to_dtype(precision) -> torch.dtype:
        for names, dtype in _TO_DTYPE:
            if precision in names:
                return dtype
        raise RuntimeError(f"unexpected precision: {precision}")

    @staticmethod
    def to_str(precision: torch.dtype) -> str:
        if precision not in _TO_STR:
            raise RuntimeError(f"unexpected precision: {precision}")
        return _TO_STR[precision]
