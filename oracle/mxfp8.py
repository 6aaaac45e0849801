"""Oracle (test infrastructure): MX-fp8 block quantisation and the block-scaled GEMM it feeds, in numpy.

Not a restatement of reference code — the reference (hunarbatra/SpatialThinker) has no fp8 path; BASELINE.json config #5 asks for
one on MI355X.  This file restates the PUBLISHED format instead: OCP Microscaling Formats (MX) v1.0 — element type FP8 E4M3
(OCP 8-bit Floating Point Specification: bias 7, max normal 448, no infinities, S.1111.111 = NaN), shared scale type E8M0
(2^(byte - 127)), block size 32, shared exponent = floor(log2(max|x|)) - emax_elem with emax_elem = 8, elements rounded to nearest
even and saturated.  The HIP kernels (csrc/gemm_fp8.hip) are checked against it: quantiser bit for bit, GEMM against the fp32
product of the de-quantised operands."""
from __future__ import annotations

import numpy as np


def e4m3_decode_table() -> np.ndarray:
    t = np.zeros(256, dtype=np.float32)
    for b in range(256):
        s, e, m = b >> 7, (b >> 3) & 15, b & 7
        if e == 15 and m == 7:
            v = np.nan
        elif e == 0:
            v = m / 8.0 * 2.0 ** -6
        else:
            v = (1 + m / 8.0) * 2.0 ** (e - 7)
        t[b] = -v if s else v
    return t


def e4m3_encode(v: np.ndarray) -> np.ndarray:
    """fp32 -> e4m3fn bytes, round to nearest even, |v| <= 448 assumed (callers saturate first)."""
    v = np.asarray(v, dtype=np.float32)
    a = np.abs(v).astype(np.float64)
    sign = (np.signbit(v)).astype(np.uint8) << 7
    out = np.zeros(v.shape, dtype=np.uint8)
    sub = a < 2.0 ** -6
    qs = np.rint(a / 2.0 ** -9)                                  # numpy rint = round half to even
    e = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    mant = np.rint((a / 2.0 ** e - 1.0) * 8.0)
    carry = mant == 8
    e = np.where(carry, e + 1, e)
    mant = np.where(carry, 0, mant)
    norm = ((e + 7).astype(np.int64) << 3) | mant.astype(np.int64)
    norm = np.minimum(norm, 0x7E)                                # 448 = S.1111.110
    out = np.where(sub, qs.astype(np.int64), norm).astype(np.uint8)   # qs == 8 is exactly the smallest normal's code 0x08
    return out | sign


def quantize(x: np.ndarray):
    """x (R, K) float32 (bf16-representable) -> (q (R, K) uint8, scale_bytes (R, K/32) uint8)."""
    x = np.asarray(x, dtype=np.float32)
    R, K = x.shape
    blk = x.reshape(R, K // 32, 32)
    amax = np.abs(blk).max(-1)
    E = (amax.view(np.uint32) >> 23) & 0xFF                       # biased exponent = floor(log2 amax) + 127 for normals
    sb = np.clip(E.astype(np.int64) - 8, 0, 254).astype(np.uint8)
    inv = ((254 - sb.astype(np.uint32)) << 23).view(np.float32)   # 2^-(sb - 127)
    scaled = np.clip(blk * inv[..., None], -448.0, 448.0).astype(np.float32)
    return e4m3_encode(scaled).reshape(R, K), sb


def pack_scales(sb: np.ndarray, rows_pad: int) -> np.ndarray:
    """(R, K/32) scale bytes -> the kernels' K-tile-major layout (K/128, rows_pad) dwords (byte j = block 4*kt + j)."""
    R, nb = sb.shape
    out = np.zeros((nb // 4, rows_pad), dtype=np.uint32)
    s4 = sb.reshape(R, nb // 4, 4).astype(np.uint32)
    out[:, :R] = (s4[..., 0] | (s4[..., 1] << 8) | (s4[..., 2] << 16) | (s4[..., 3] << 24)).T
    return out


def dequantize(q: np.ndarray, sb: np.ndarray) -> np.ndarray:
    R, K = q.shape
    vals = e4m3_decode_table()[q].reshape(R, K // 32, 32)
    scale = ((sb.astype(np.uint32)) << 23).view(np.float32)        # 2^(sb - 127); sb = 0 -> 0.0 * (all-zero block) = 0
    scale = np.where(sb == 0, np.float32(2.0 ** -126) * np.float32(0.5), scale)
    return (vals * scale[..., None]).reshape(R, K).astype(np.float32)
