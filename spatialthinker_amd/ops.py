"""Thin torch-tensor front-end over the C ABI (include/st_hip.h).

torch is plumbing here: device memory, streams.  Every function enqueues exactly the HIP
kernels of libst_hip.so on torch's current stream and raises if a tensor is not on a GPU.
"""
from __future__ import annotations

import math

import numpy as np
import os
from typing import Optional

import torch

from .lib import lib

BF16, F32, I32, I64 = torch.bfloat16, torch.float32, torch.int32, torch.int64
KL_KINDS = {"kl": 0, "abs": 1, "mse": 2, "low_var_kl": 3, "chi2": 4}
K_GEMM, K_ATTN_FWD, K_ATTN_BWD, K_LOGPROB, K_ADAMW, K_RMSNORM, K_VIT_ATTN, K_DECODE_ATTN, K_GEMM_FP8, K_VIT_WIN = range(10)


def _s() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("spatialthinker_amd.ops: tensor is not on a GPU (there is no CPU path)")
    return t.data_ptr()


def _chk(t, dtype, what):
    if t.dtype != dtype:
        raise TypeError(f"{what}: expected {dtype}, got {t.dtype}")
    if t.dim() >= 1 and t.stride(-1) != 1:
        raise ValueError(f"{what}: last dim must be contiguous")
    return t


# ------------------------------------------------------------------ log-prob / GRPO
def logprob_fwd(logits: torch.Tensor, labels: torch.Tensor, temperature: float = 1.0):
    """logits (T,V) bf16, labels (T,) int64 -> (logp (T,) f32, lse (T,) f32)."""
    _chk(logits, BF16, "logits"); _chk(labels, I64, "labels")
    T, V = logits.shape
    logp = torch.empty(T, dtype=F32, device=logits.device)
    lse = torch.empty(T, dtype=F32, device=logits.device)
    lib().st_logprob_fwd(_p(logits), logits.stride(0), _p(labels), 1.0 / temperature, _p(logp), _p(lse), T, V, _s())
    return logp, lse


def logprob_bwd_(logits: torch.Tensor, labels: torch.Tensor, lse: torch.Tensor, g: torch.Tensor, temperature: float = 1.0):
    """In place: logits <- d(sum g*logp)/dlogits."""
    _chk(logits, BF16, "logits"); _chk(g, F32, "g"); _chk(lse, F32, "lse")
    T, V = logits.shape
    lib().st_logprob_bwd(_p(logits), logits.stride(0), _p(labels), _p(lse), _p(g), 1.0 / temperature, T, V, _s())
    return logits


def grpo_loss(logp, old_logp, ref_logp, adv, mask, *, clip_low=0.2, clip_high=0.3, clip_dual=3.0,
              kl_kind="low_var_kl", kl_coef=1e-2, grad_accum=1.0):
    """Flattened (n,) fp32 inputs, mask int64 -> (g (n,) f32, metrics (8,) f32 on device)."""
    for t, nm in ((logp, "logp"), (old_logp, "old"), (adv, "adv")):
        _chk(t, F32, nm)
    _chk(mask, I64, "mask")
    n = logp.numel()
    g = torch.empty(n, dtype=F32, device=logp.device)
    met = torch.empty(8, dtype=F32, device=logp.device)
    lib().st_grpo_loss(_p(logp), _p(old_logp), _p(ref_logp), _p(adv), _p(mask), n, clip_low, clip_high, clip_dual,
                       KL_KINDS[kl_kind], kl_coef, grad_accum, _p(g), _p(met), _s())
    return g, met


def value_loss(vpreds, returns, values, mask, *, cliprange_value=0.5, grad_accum=1.0):
    """Clipped value loss of the critic (core_algos.compute_value_loss) + gradient: flattened (n,) fp32 inputs, mask int64 ->
    (g (n,) f32 = d(vf_loss / grad_accum)/dvpreds, metrics (4,) f32 on device = [vf_loss, vf_clipfrac, masked_mean(vpreds), sum(mask)])."""
    for t, nm in ((vpreds, "vpreds"), (returns, "returns"), (values, "values")):
        _chk(t, F32, nm)
    _chk(mask, I64, "mask")
    n = vpreds.numel()
    g = torch.empty(n, dtype=F32, device=vpreds.device)
    met = torch.empty(4, dtype=F32, device=vpreds.device)
    lib().st_value_loss(_p(vpreds), _p(returns), _p(values), _p(mask), n, float(cliprange_value), float(grad_accum), _p(g), _p(met), _s())
    return g, met


def value_head_fwd(hn, w, bias=None):
    """v (T,) fp32 = bf16(hn @ w + bias): the token-classification score head (nn.Linear(H, 1)) of the critic."""
    T, H = hn.shape
    out = torch.empty(T, dtype=F32, device=hn.device)
    lib().st_value_head_fwd(_p(hn), hn.stride(0), _p(w), _p(bias), _p(out), T, H, _s())
    return out


def value_head_bwd(hn, w, dv, dw_accum, db_accum=None):
    """dhn (T, H) bf16 = dv (x) w;  dw_accum (H,) f32 += dv^T hn;  db_accum (>= 1,) f32 [0] += sum(dv)."""
    T, H = hn.shape
    _chk(dv, F32, "dv")
    dhn = torch.empty(T, H, dtype=BF16, device=hn.device)
    lib().st_value_head_bwd(_p(hn), hn.stride(0), _p(w), _p(dv), _p(dhn), dhn.stride(0), _p(dw_accum), _p(db_accum), T, H, _s())
    return dhn


def grpo_advantage(rewards, mask, group, n_groups: int, eps: float = 1e-6):
    _chk(rewards, F32, "rewards"); _chk(mask, I64, "mask"); _chk(group, I32, "group")
    N, R = rewards.shape
    adv = torch.empty_like(rewards)
    scratch = torch.empty(N + 2 * n_groups, dtype=F32, device=rewards.device)
    status = torch.zeros(1, dtype=I32, device=rewards.device)
    lib().st_grpo_advantage(_p(rewards), _p(mask), _p(group), N, R, n_groups, eps, _p(adv), _p(scratch), _p(status), _s())
    return adv, status


# ------------------------------------------------------------------ norms / rope / activations
def rmsnorm_fwd(x, w, eps, want_rstd=True, out=None):
    _chk(x, BF16, "x"); _chk(w, BF16, "w")
    T, H = x.shape
    y = torch.empty(T, H, dtype=BF16, device=x.device) if out is None else out
    rstd = torch.empty(T, dtype=F32, device=x.device) if want_rstd else None
    lib().st_rmsnorm_fwd(_p(x), x.stride(0), _p(w), eps, _p(y), y.stride(0), _p(rstd), T, H, _s())
    return y, rstd


RMSNORM_BWD_FUSED = os.environ.get("ST_RMSNORM_BWD_FUSED", "1") != "0"      # one-pass deterministic backward (round 6); 0: the two-kernel form


def rmsnorm_bwd(x, w, rstd, dy, dres=None, dw_accum=None, out=None):
    T, H = x.shape
    dx = torch.empty(T, H, dtype=BF16, device=x.device) if out is None else out
    if RMSNORM_BWD_FUSED and H <= 4096 and not any(t_ is not None and (t_.data_ptr() & 15 or (t_.dim() == 2 and t_.stride(0) & 7)) for t_ in (x, dy, dx, w, dres)):
        # one pass, deterministic dw (st_rmsnorm_bwd_fused); the scratch comes from torch's stream-ordered allocator
        ws = torch.empty(lib().st_rmsnorm_bwd_workspace_bytes(T, H) // 4, dtype=F32, device=x.device) if dw_accum is not None else None
        lib().st_rmsnorm_bwd_fused(_p(x), x.stride(0), _p(w), _p(rstd), _p(dy), dy.stride(0), _p(dres), dres.stride(0) if dres is not None else 0,
                                   _p(dx), dx.stride(0), _p(dw_accum), _p(ws), ws.numel() * 4 if ws is not None else 0, T, H, _s())
        return dx
    lib().st_rmsnorm_bwd(_p(x), x.stride(0), _p(w), _p(rstd), _p(dy), dy.stride(0), _p(dres),
                         dres.stride(0) if dres is not None else 0, _p(dx), dx.stride(0), _p(dw_accum), T, H, _s())
    return dx


def mrope_table(pos_3T: torch.Tensor, inv_freq: torch.Tensor, D: int, section):
    _chk(pos_3T, I32, "pos"); _chk(inv_freq, F32, "inv_freq")
    T = pos_3T.shape[1]
    cos = torch.empty(T, D // 2, dtype=F32, device=pos_3T.device)
    sin = torch.empty_like(cos)
    lib().st_mrope_table(_p(pos_3T), _p(inv_freq), T, D, section[0], section[1], section[2], _p(cos), _p(sin), _s())
    return cos, sin


def rope_apply_(x, cos, sin, n_rot_heads: int, D: int, inverse: bool = False):
    _chk(x, BF16, "x"); _chk(cos, F32, "cos")
    lib().st_rope_apply(_p(x), x.stride(0), _p(cos), _p(sin), x.shape[0], n_rot_heads, D, int(inverse), _s())
    return x


def swiglu_fwd(gu, out=None):
    T, I2 = gu.shape
    o = torch.empty(T, I2 // 2, dtype=BF16, device=gu.device) if out is None else out
    lib().st_swiglu_fwd(_p(gu), gu.stride(0), _p(o), o.stride(0), T, I2 // 2, _s())
    return o


def swiglu_bwd(gu, dout, out=None, want_m=False):
    """dgu of the SwiGLU; want_m: also the forward's result m = swiglu_fwd(gu) (bit-identical), recomputed in the same pass -> (dgu, m)."""
    T, I2 = gu.shape
    d = torch.empty_like(gu) if out is None else out
    if not want_m:
        lib().st_swiglu_bwd(_p(gu), gu.stride(0), _p(dout), dout.stride(0), _p(d), d.stride(0), T, I2 // 2, _s())
        return d
    m = torch.empty(T, I2 // 2, dtype=BF16, device=gu.device)
    lib().st_swiglu_bwd_m(_p(gu), gu.stride(0), _p(dout), dout.stride(0), _p(d), d.stride(0), _p(m), m.stride(0), T, I2 // 2, _s())
    return d, m


def gelu_fwd(x):
    y = torch.empty_like(x)
    lib().st_gelu_fwd(_p(x), _p(y), x.numel(), _s())
    return y


def gelu_bwd(x, dy):
    dx = torch.empty_like(x)
    lib().st_gelu_bwd(_p(x), _p(dy), _p(dx), x.numel(), _s())
    return dx


# ------------------------------------------------------------------ GEMM
DECODE_MAX_ROWS = 512           # ST_DECODE_MAX_ROWS of the library
_SCRATCH_ELEMS = 16 * DECODE_MAX_ROWS * 4608
_scratch = {}


_scratch_slot = [0]


class scratch_slot:
    """`with ops.scratch_slot(1):` — the decode-shaped GEMMs issued inside use their OWN fp32 slab buffer.  The decode tail of a rollout
    (a replayed hipGraph whose slab buffer is slot 0, fixed at capture) and the log-prob pass of its finished samples run side by side on
    two CU-partitioned streams (round 5); the pass reaches the same entry with its M <= 256 GEMMs and must not share the slabs."""

    def __init__(self, slot: int):
        self.slot = int(slot)

    def __enter__(self):
        self.prev, _scratch_slot[0] = _scratch_slot[0], self.slot

    def __exit__(self, *exc):
        _scratch_slot[0] = self.prev


def _skinny_scratch(device):
    """fp32 split-K slab buffer [split][M][N] of the decode-shaped GEMM (contents irrelevant between calls); one per scratch_slot."""
    key = (device.type, device.index, _scratch_slot[0])
    if key not in _scratch:
        assert not torch.cuda.is_current_stream_capturing(), "the slab buffer must exist before a hipGraph capture (run one eager iteration first)"
        _scratch[key] = torch.empty(_SCRATCH_ELEMS, dtype=F32, device=device)
    return _scratch[key]


_cu_streams = {}


def h2d(a, dtype, device) -> torch.Tensor:
    """Host array / tensor -> device tensor WITHOUT holding the host: the bytes go through a pinned buffer (torch's caching host allocator
    keeps it alive until the copy has run) and the copy is queued on the current stream.  A pageable source makes the copy synchronous —
    the host then waits for every kernel already queued and cannot prepare the next pass meanwhile (round 6; under a profiler, whose
    per-launch cost makes the host the slower side, that showed as ~50 ms of idle GPU per packed pass; un-profiled the step time did not
    move: profiles/r06_notes.md §3)."""
    src = a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))
    if src.is_cuda or torch.device(device).type != "cuda":           # (CPU targets: the gloo tests' stub engines)
        return src.to(device=device, dtype=dtype, non_blocking=True)
    if src.dtype != dtype:
        src = src.to(dtype)
    src = src.contiguous()
    if src.numel() == 0:
        return torch.empty(src.shape, dtype=dtype, device=device)
    return src.pin_memory().to(device=device, non_blocking=True)


def cu_range_stream(first_cu: int, n_cus: int) -> "torch.cuda.Stream":
    """A stream whose kernels run on compute units [first_cu, first_cu + n_cus) only (st_stream_create_cu_range; on MI355X multiples of 8
    take the same share of every XCD).  Cached per range: the handle lives as long as the process."""
    import ctypes
    key = (torch.cuda.current_device(), int(first_cu), int(n_cus))
    if key not in _cu_streams:
        h = ctypes.c_void_p(0)
        lib().st_stream_create_cu_range(int(first_cu), int(n_cus), ctypes.byref(h))
        _cu_streams[key] = torch.cuda.ExternalStream(h.value)
    return _cu_streams[key]


# (M, N, K) -> (tile variant, split-K count) chosen by autotune_decode_gemm; empty = library defaults everywhere
_decode_plans = {}
_DECODE_CANDIDATES = {64: (10, 11, 12, 21), 128: (13, 14, 19, 20), 256: (13, 14, 16, 18)}


def autotune_decode_gemm(M: int, weights, reps: int = 2):
    """Time the decode-shaped GEMM out[M,N] = x[M,K] @ weight[N,K]^T for every (tile variant, split-K) candidate and pin the
    fastest for this (M, N, K).  Launches are replayed from a hipGraph so the timing has no host launch gaps (the decode loop
    replays a graph as well).  Must not be called while a stream capture is active.  Opt-in: the choice depends on measured
    times, so two processes may pick different split counts (different fp32 summation order in the last bits)."""
    weights = [weights] if torch.is_tensor(weights) else list(weights)
    N, K = weights[0].shape
    key = (M, N, K)
    if key in _decode_plans or M > DECODE_MAX_ROWS:
        return _decode_plans.get(key)
    dev = weights[0].device
    if len(weights) * N * K * 2 < (1 << 30):                 # too small a footprint to defeat the cache: keep the defaults
        return None
    x = (torch.randn(M, K, device=dev) * 0.1).to(BF16)
    out = torch.empty(M, N, dtype=BF16, device=dev)
    scratch = _skinny_scratch(dev)
    bm = 64 if M <= 64 else (128 if M <= 128 else 256)

    def timed(fn):
        fn(weights[0])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            with torch.cuda.graph(g, stream=st):
                for _ in range(reps):
                    for wt in weights:
                        fn(wt)
        g.replay()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        g.replay(); g.replay()
        ev1.record()
        torch.cuda.synchronize()
        return ev0.elapsed_time(ev1) / (2 * reps * len(weights))

    def run_default(weight):
        lib().st_gemm_nt_skinny(_p(x), x.stride(0), _p(weight), weight.stride(0), None, None, 0, _p(out), out.stride(0), _p(scratch),
                                scratch.numel(), M, N, K, _s())

    best_t, best = timed(run_default), None
    for v in _DECODE_CANDIDATES[bm]:
        for sp in (1, 2, 4, 8):
            if sp > 1 and (sp * M * N > scratch.numel() or N >= 16384 or K // 64 < 4 * sp):
                continue

            def run(weight, v=v, sp=sp):
                lib().st_gemm_nt_decode_variant(v, sp, _p(x), x.stride(0), _p(weight), weight.stride(0), None, None, 0, _p(out),
                                                out.stride(0), _p(scratch), scratch.numel(), M, N, K, _s())
            t = timed(run)
            if t < 0.97 * best_t:                           # keep the default unless clearly beaten
                best_t, best = t, (v, sp)
    if best is not None:
        _decode_plans[key] = best
    return best


_gemm_ws = {}
# algorithmic HBM bytes (operands once + result once) and launches of the training-shape GEMM entries while counting is on
# (bench.py: the denominator of roofline.traffic_over_algorithmic)
gemm_bytes = {"on": False, "bytes": 0.0, "launches": 0}


def _count_gemm(M, N, K, out_bytes_per_elem, extra=0.0):
    if gemm_bytes["on"]:
        gemm_bytes["bytes"] += 2.0 * (M * K + N * K) + out_bytes_per_elem * M * N + extra
        gemm_bytes["launches"] += 1



def _gemm_workspace(device, nbytes: int = 128 << 20):
    """Registers (once) the tail-split workspace of st_gemm_nt (st_gemm_set_workspace); ST_GEMM_TAIL_SPLIT=0 leaves it off."""
    if "ws" not in _gemm_ws:
        import os
        ws = None
        if os.environ.get("ST_GEMM_TAIL_SPLIT", "1") != "0":
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
            lib().st_gemm_set_workspace(_p(ws), nbytes)
        _gemm_ws["ws"] = ws
    return _gemm_ws["ws"]


def gemm_tail_split(enable: bool, device="cuda"):
    """Switch the tail split of the 256x256-tile GEMMs on/off at run time (tests, A/B timing)."""
    ws = _gemm_workspace(device)
    if ws is None and enable:
        ws = _gemm_ws["ws"] = torch.empty(128 << 20, dtype=torch.uint8, device=device)
    lib().st_gemm_set_workspace(_p(ws) if enable else None, ws.numel() if (enable and ws is not None) else 0)


def gemm_nt(a, b, *, bias=None, residual=None, out=None, out_f32=None, accumulate=False, decode=False):
    """C[M,N] = A[M,K] @ B[N,K]^T (+bias)(+residual).  Returns bf16 `out` (allocated if needed) unless `out_f32` given.
    decode=True (the rollout generator's call sites only): 257..512-row GEMMs also take the weight-streaming decode plans; every
    other caller gets st_gemm_nt above 256 rows, whatever ran before in the process."""
    _chk(a, BF16, "A"); _chk(b, BF16, "B")
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K, (a.shape, b.shape)
    if out_f32 is None and out is None:
        out = torch.empty(M, N, dtype=BF16, device=a.device)
    if out_f32 is None and (M <= 256 or (decode and M <= DECODE_MAX_ROWS)):   # decode-shaped: weight-streaming tiles (+ split-K slabs)
        scratch = _skinny_scratch(a.device)
        plan = _decode_plans.get((M, N, K))
        ldr = residual.stride(0) if residual is not None else 0
        if plan is None:                                    # library default (heuristic table in gemm.hip)
            lib().st_gemm_nt_skinny(_p(a), a.stride(0), _p(b), b.stride(0), _p(bias), _p(residual), ldr, _p(out), out.stride(0),
                                    _p(scratch), scratch.numel(), M, N, K, _s())
        else:
            lib().st_gemm_nt_decode_variant(plan[0], plan[1], _p(a), a.stride(0), _p(b), b.stride(0), _p(bias), _p(residual), ldr,
                                            _p(out), out.stride(0), _p(scratch), scratch.numel(), M, N, K, _s())
        return out
    c = out if out_f32 is None else out_f32
    _gemm_workspace(a.device)
    _count_gemm(M, N, K, 2 if out_f32 is None else (8 if accumulate else 4), (2.0 * M * N if residual is not None else 0.0))
    lib().st_gemm_nt(_p(a), a.stride(0), _p(b), b.stride(0), _p(bias), _p(residual),
                     residual.stride(0) if residual is not None else 0, _p(out) if out_f32 is None else None,
                     _p(out_f32), c.stride(0), int(accumulate), M, N, K, _s())
    return c


def gemm_nn(a, b_kn, out=None):
    """out[M,N] bf16 = a[M,K] @ b_kn[K,N] (b contraction-major, e.g. dX = dY @ W with W stored (out, in)): st_gemm_nn."""
    _chk(a, BF16, "A"); _chk(b_kn, BF16, "B")
    M, K = a.shape
    N = b_kn.shape[1]
    assert b_kn.shape[0] == K, (a.shape, b_kn.shape)
    out = torch.empty(M, N, dtype=BF16, device=a.device) if out is None else out
    _gemm_workspace(a.device)
    _count_gemm(M, N, K, 2)
    lib().st_gemm_nn(_p(a), a.stride(0), _p(b_kn), b_kn.stride(0), _p(out), out.stride(0), M, N, K, _s())
    return out


def gemm_tn(a_km, b_kn, out_f32, accumulate=False):
    """out_f32[M,N] (+)= a_km[K,M]^T @ b_kn[K,N] (both contraction-major, e.g. dW = dY^T @ X): st_gemm_tn."""
    _chk(a_km, BF16, "A"); _chk(b_kn, BF16, "B"); _chk(out_f32, F32, "out")
    K, M = a_km.shape
    N = b_kn.shape[1]
    assert b_kn.shape[0] == K and out_f32.shape == (M, N), (a_km.shape, b_kn.shape, out_f32.shape)
    _gemm_workspace(a_km.device)
    _count_gemm(M, N, K, 8 if accumulate else 4)
    lib().st_gemm_tn(_p(a_km), a_km.stride(0), _p(b_kn), b_kn.stride(0), _p(out_f32), out_f32.stride(0), int(accumulate), M, N, K, _s())
    return out_f32


def decode_plan(M: int, N: int, K: int):
    """(tile variant, split-K slices) the library picks for a decode-shaped GEMM (st_gemm_decode_plan; inspection only)."""
    import ctypes
    v, sp = ctypes.c_int(0), ctypes.c_int(0)
    lib().st_gemm_decode_plan(M, N, K, _SCRATCH_ELEMS, ctypes.addressof(v), ctypes.addressof(sp))
    return v.value, sp.value


def swiglu_decode_plan(M: int, I: int) -> int:
    import ctypes
    v = ctypes.c_int(0)
    lib().st_gemm_swiglu_decode_plan(M, I, ctypes.addressof(v))
    return v.value


def layout_gemm_ok(M: int, N: int, K: int) -> bool:
    """The contraction-major GEMM forms run on the 256x256 tile only: worth it from half a round of tiles on 256 CUs."""
    return K % 64 == 0 and M % 8 == 0 and N % 8 == 0 and M >= 8 and N >= 8 and (-(-M // 256)) * (-(-N // 256)) >= 128


# ------------------------------------------------------------------ fused decode epilogues
def gemm_nt_decode_slabs(a, b):
    """Decode-shaped GEMM (M <= 256) that leaves fp32 split-K slabs in the shared scratch; returns (scratch, splits)."""
    import ctypes
    M, K = a.shape
    N = b.shape[0]
    scratch = _skinny_scratch(a.device)
    sp = ctypes.c_int(0)
    lib().st_gemm_nt_decode_slabs(_p(a), a.stride(0), _p(b), b.stride(0), _p(scratch), scratch.numel(), M, N, K, ctypes.addressof(sp), _s())
    return scratch, sp.value


def decode_finish_norm(slabs, splits, M, N, *, residual, x_out, norm_w=None, eps=1e-6, h_out=None):
    lib().st_decode_finish_norm(_p(slabs), splits, _p(residual), residual.stride(0) if residual is not None else 0, _p(x_out),
                                x_out.stride(0), _p(norm_w), eps, _p(h_out), h_out.stride(0) if h_out is not None else 0, M, N, _s())


def decode_finish_qkv(slabs, splits, M, bias, cos, sin, q_out, kg, vg, gen_len, B, n_q, n_kv, D, row_map=None):
    lib().st_decode_finish_qkv(_p(slabs), splits, _p(bias), _p(cos), _p(sin), _p(q_out), q_out.stride(0), _p(kg), _p(vg), kg.stride(0),
                               _p(gen_len), _p(row_map), B, M, n_q, n_kv, D, _s())


def gemm_swiglu_decode(a, gate_up_w, out=None):
    """out[M, I] = silu(a @ gate_w^T) * (a @ up_w^T), gate_up_w = [gate_w ; up_w] (2I, K); M <= 256."""
    M, K = a.shape
    I = gate_up_w.shape[0] // 2
    out = torch.empty(M, I, dtype=BF16, device=a.device) if out is None else out
    if M > 256:
        _gemm_workspace(a.device)                           # 257..512 rows run the training tile with its K-split tail (st_gemm_set_workspace)
    lib().st_gemm_swiglu_decode(_p(a), a.stride(0), _p(gate_up_w), gate_up_w.stride(0), _p(out), out.stride(0), M, I, K, _s())
    return out


def gemm_swiglu(a, gate_up_w, want_gu=True):
    """(gu, m): m[M, I] = silu(a @ gate_w^T) * (a @ up_w^T) with the SwiGLU in the GEMM epilogue; gu [M, 2I] (bf16 gate|up) only
    when the backward needs it."""
    M, K = a.shape
    I = gate_up_w.shape[0] // 2
    m = torch.empty(M, I, dtype=BF16, device=a.device)
    gu = torch.empty(M, 2 * I, dtype=BF16, device=a.device) if want_gu else None
    _count_gemm(M, 2 * I, K, 0, 2.0 * M * I * (3 if want_gu else 1))
    lib().st_gemm_swiglu(_p(a), a.stride(0), _p(gate_up_w), gate_up_w.stride(0), _p(gu), gu.stride(0) if gu is not None else 0,
                         _p(m), m.stride(0), M, I, K, _s())
    return gu, m


# ------------------------------------------------------------------ block-scaled fp8 (MX-fp8) GEMM
def mxfp8_quantize(x):
    """x (R, K) bf16 -> (q (R, K) uint8 e4m3 bytes, scales (K/128, R_pad4) int32: 4 e8m0 bytes per row and K-tile)."""
    _chk(x, BF16, "x")
    R, K = x.shape
    q = torch.empty(R, K, dtype=torch.uint8, device=x.device)
    rows = (R + 3) // 4 * 4
    sc = torch.zeros(K // 128, rows, dtype=torch.int32, device=x.device)
    lib().st_mxfp8_quantize(_p(x), x.stride(0), _p(q), q.stride(0), _p(sc), rows, R, K, _s())
    return q, sc


def mxfp8_quantize_t(x):
    """MX-fp8 of x^T without materialising it: x (R, C) bf16, R % 128 == 0 -> (qT (C, R) uint8, scales (R/128, C_pad4) int32): MX blocks run
    along the ROWS of x (the token index) — the operands of the weight-gradient products."""
    _chk(x, BF16, "x")
    R, C = x.shape
    qT = torch.empty(C, R, dtype=torch.uint8, device=x.device)
    rows = (C + 3) // 4 * 4
    sc = torch.zeros(R // 128, rows, dtype=torch.int32, device=x.device)
    lib().st_mxfp8_quantize_t(_p(x), x.stride(0), _p(qT), qT.stride(0), _p(sc), rows, R, C, _s())
    return qT, sc


def mxfp8_quantize_both(x):
    """((q, scales), (qT, scales_t)) = (mxfp8_quantize(x), mxfp8_quantize_t(x)) in one pass over x (R % 128 == 0, C % 128 == 0)."""
    _chk(x, BF16, "x")
    R, C = x.shape
    q, sc = _mx_buffers(R, C, x.device)
    qT = torch.empty(C, R, dtype=torch.uint8, device=x.device)
    sct = torch.zeros(R // 128, (C + 3) // 4 * 4, dtype=torch.int32, device=x.device)
    lib().st_mxfp8_quantize_both(_p(x), x.stride(0), _p(q), q.stride(0), _p(sc), sc.shape[1], _p(qT), qT.stride(0), _p(sct), sct.shape[1], R, C, _s())
    return (q, sc), (qT, sct)


def gemm_mxfp8_nt_f32(aq, sa, bq, sb, out_f32, accumulate=False):
    """out_f32[M,N] = or += dequant(aq, sa) @ dequant(bq, sb)^T (fp32 result: the weight gradients)."""
    M, K = aq.shape
    N = bq.shape[0]
    assert bq.shape[1] == K and sa.shape[0] == K // 128 and sb.shape[0] == K // 128 and out_f32.shape == (M, N)
    lib().st_gemm_mxfp8_nt_f32(_p(aq), aq.stride(0), _p(sa), sa.shape[1], _p(bq), bq.stride(0), _p(sb), sb.shape[1], _p(out_f32),
                               out_f32.stride(0), int(accumulate), M, N, K, _s())
    return out_f32


_fp8_tile = 8 if os.environ.get("ST_FP8_TILE", "") == "8" else 4


def gemm_mxfp8_select(waves: int) -> int:
    """Tile behind gemm_mxfp8_nt: 4 = the hand-scheduled 4-wave tile (gemm_mx4.hip, default), 8 = the 8-wave tile; returns the previous."""
    global _fp8_tile
    prev = _fp8_tile
    lib().st_gemm_mxfp8_select(int(waves))
    _fp8_tile = int(waves)
    return prev


def gemm_mxfp8_nt(aq, sa, bq, sb, *, bias=None, residual=None, out=None):
    """out[M,N] bf16 = dequant(aq, sa) @ dequant(bq, sb)^T (+bias)(+residual); operands from mxfp8_quantize."""
    M, K = aq.shape
    N = bq.shape[0]
    assert bq.shape[1] == K and sa.shape[0] == K // 128 and sb.shape[0] == K // 128
    out = torch.empty(M, N, dtype=BF16, device=aq.device) if out is None else out
    lib().st_gemm_mxfp8_nt(_p(aq), aq.stride(0), _p(sa), sa.shape[1], _p(bq), bq.stride(0), _p(sb), sb.shape[1], _p(bias), _p(residual),
                           residual.stride(0) if residual is not None else 0, _p(out), out.stride(0), M, N, K, _s())
    return out


def gemm_mxfp8_swiglu(aq, sa, bq, sb, *, out=None):
    """out[M, I] bf16 = silu(gate) * up with [gate | up] = dequant(aq, sa) @ dequant(bq, sb)^T, bq = (2I, K): SwiGLU in the fp8 tile's epilogue."""
    M, K = aq.shape
    I = bq.shape[0] // 2
    assert bq.shape[1] == K and bq.shape[0] == 2 * I and sa.shape[0] == K // 128 and sb.shape[0] == K // 128
    out = torch.empty(M, I, dtype=BF16, device=aq.device) if out is None else out
    lib().st_gemm_mxfp8_swiglu(_p(aq), aq.stride(0), _p(sa), sa.shape[1], _p(bq), bq.stride(0), _p(sb), sb.shape[1], _p(out), out.stride(0),
                               M, I, K, _s())
    return out


def _mx_buffers(R, K, device):
    rows = (R + 3) // 4 * 4
    return torch.empty(R, K, dtype=torch.uint8, device=device), torch.zeros(K // 128, rows, dtype=torch.int32, device=device)


def gemm_mxfp8_swiglu_q(aq, sa, bq, sb):
    """MX-fp8 (q, scales) of silu(gate) * up — gemm_mxfp8_swiglu + mxfp8_quantize in the tile's epilogue; the bf16 activation is never stored."""
    M, K = aq.shape
    I = bq.shape[0] // 2
    assert bq.shape[1] == K and bq.shape[0] == 2 * I and I % 128 == 0
    q, sc = _mx_buffers(M, I, aq.device)
    lib().st_gemm_mxfp8_swiglu_q(_p(aq), aq.stride(0), _p(sa), sa.shape[1], _p(bq), bq.stride(0), _p(sb), sb.shape[1], _p(q), q.stride(0),
                                 _p(sc), sc.shape[1], M, I, K, _s())
    return q, sc


def rmsnorm_mxfp8(x, w, eps, want_y=True, want_rstd=True):
    """RMSNorm forward + MX-fp8 quantisation of its bf16 result in one pass: (y | None, rstd | None, (q, scales))."""
    T, H = x.shape
    y = torch.empty(T, H, dtype=BF16, device=x.device) if want_y else None
    rstd = torch.empty(T, dtype=F32, device=x.device) if want_rstd else None
    q, sc = _mx_buffers(T, H, x.device)
    lib().st_rmsnorm_mxfp8(_p(x), x.stride(0), _p(w), float(eps), _p(y), y.stride(0) if want_y else 0, _p(q), q.stride(0), _p(sc), sc.shape[1],
                           _p(rstd), T, H, _s())
    return y, rstd, (q, sc)


def swiglu_mxfp8(gu, want_out=True):
    """SwiGLU forward + MX-fp8 quantisation of its bf16 result in one pass: (out | None, (q, scales))."""
    T, I2 = gu.shape
    I = I2 // 2
    out = torch.empty(T, I, dtype=BF16, device=gu.device) if want_out else None
    q, sc = _mx_buffers(T, I, gu.device)
    lib().st_swiglu_mxfp8(_p(gu), gu.stride(0), _p(out), out.stride(0) if want_out else 0, _p(q), q.stride(0), _p(sc), sc.shape[1], T, I, _s())
    return out, (q, sc)


def gemm_select(variant: int):
    """Production tile of the training-shape GEMMs (st_gemm_select): 23 = 8-wave tile, 40 = 4-wave tile with the hand-scheduled loop."""
    lib().st_gemm_select(int(variant))


def gemm_nt_variant(variant, a, b, out=None, out_f32=None, accumulate=False, bias=None, residual=None):
    M, K = a.shape
    N = b.shape[0]
    if out is None and out_f32 is None:
        out = torch.empty(M, N, dtype=BF16, device=a.device)
    c = out if out_f32 is None else out_f32
    lib().st_gemm_nt_variant(int(variant), _p(a), a.stride(0), _p(b), b.stride(0), _p(bias), _p(residual),
                             residual.stride(0) if residual is not None else 0, _p(out) if out_f32 is None else None, _p(out_f32),
                             c.stride(0), int(accumulate), M, N, K, _s())
    return c


def transpose(x, out=None):
    R, C = x.shape
    o = torch.empty(C, R, dtype=BF16, device=x.device) if out is None else out
    lib().st_transpose(_p(x), x.stride(0), _p(o), o.stride(0), R, C, _s())
    return o


def colsum(x, out_f32=None, accumulate=False):
    R, C = x.shape
    o = torch.empty(C, dtype=F32, device=x.device) if out_f32 is None else out_f32
    lib().st_colsum(_p(x), x.stride(0), _p(o), int(accumulate and out_f32 is not None), R, C, _s())
    return o


# ------------------------------------------------------------------ attention
_VIT_WIN = os.environ.get("ST_VIT_WIN", "1")[:1] != "0"


def _is_vit_window(max_seqlen, n_q, n_kv, D, causal) -> bool:
    """the launches st_attn_fwd / st_attn_bwd hand to attention_win.hip (same rule as csrc/attention.hip)"""
    return _VIT_WIN and D == 80 and not causal and n_q == n_kv and 0 < max_seqlen <= 64


def attn_fwd(q, k, v, cu_seqlens, max_seqlen, n_q, n_kv, D, scale, causal, out=None, pairs=None, tokens=None):
    """q (T, >=n_q*D) view, k/v (T, >=n_kv*D) views (row strides may exceed the width: qkv buffer slices).
    pairs: number of (query, key) pairs the launch evaluates, tokens: rows inside the sequences (host-side knowledge, only used for
    the profiling hooks: flops of the MFMA-bound kernels, bytes of the ViT window kernels)."""
    T = q.shape[0]
    if _is_vit_window(max_seqlen, n_q, n_kv, D, causal):
        if tokens is not None:
            prof_hint(K_VIT_WIN, tokens * n_q * (4.0 * D * 2 + 4))            # q, k, v in, o out, lse out
    elif pairs is not None:
        prof_hint(K_ATTN_FWD if D == 128 else K_VIT_ATTN, 4.0 * D * n_q * pairs)
    o = torch.empty(T, n_q * D, dtype=BF16, device=q.device) if out is None else out
    lse = torch.empty(n_q, T, dtype=F32, device=q.device)
    lib().st_attn_fwd(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(cu_seqlens), cu_seqlens.numel() - 1,
                      T, n_q, n_kv, D, scale, int(causal), _p(o), o.stride(0), _p(lse), int(max_seqlen), _s())
    return o, lse


_attn_ws = {}


def attn_bwd(q, k, v, o, do, lse, cu_seqlens, max_seqlen, n_q, n_kv, D, scale, causal, dq, dk, dv, pairs=None, tokens=None):
    T = q.shape[0]
    if _is_vit_window(max_seqlen, n_q, n_kv, D, causal):
        if tokens is not None:
            prof_hint(K_VIT_WIN, tokens * n_q * (7.0 * D * 2 + 8))            # q, k, v, dO in, dq, dk, dv out, lse in, delta out
    elif pairs is not None:
        prof_hint(K_ATTN_BWD if D == 128 else K_VIT_ATTN, 10.0 * D * n_q * pairs)
    delta = torch.empty(n_q, T, dtype=F32, device=q.device)
    need = int(lib().st_attn_bwd_workspace_bytes(T, n_q, D))
    ws = _attn_ws.get(q.device)
    if need and (ws is None or ws.numel() < need):                       # grow-only scratch for the per-head dK/dV partials
        ws = _attn_ws[q.device] = torch.empty(need, dtype=torch.uint8, device=q.device)
    lib().st_attn_bwd(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), o.stride(0), _p(do), do.stride(0),
                      _p(lse), _p(cu_seqlens), cu_seqlens.numel() - 1, T, n_q, n_kv, D, scale, int(causal),
                      _p(dq), dq.stride(0), _p(dk), dk.stride(0), _p(dv), dv.stride(0), _p(delta),
                      _p(ws) if need else None, need, int(max_seqlen), _s())
    return dq, dk, dv


def attn_fwd_seg(q, k, v, seg_b, seg_e, pre_b, pre_e, max_seg, n_q, n_kv, D, scale, out=None, k_pre=None, v_pre=None, pairs=None):
    """Shared-prefix causal attention over packed segments (st_attn_fwd_seg); k_pre / v_pre: external prefix K/V tensors."""
    T = q.shape[0]
    if pairs is not None:
        prof_hint(K_ATTN_FWD, 4.0 * D * n_q * pairs)
    o = torch.empty(T, n_q * D, dtype=BF16, device=q.device) if out is None else out
    lse = torch.empty(n_q, T, dtype=F32, device=q.device)
    lib().st_attn_fwd_seg(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(seg_b), _p(seg_e), _p(pre_b), _p(pre_e),
                          _p(k_pre), k_pre.stride(0) if k_pre is not None else 0, _p(v_pre), v_pre.stride(0) if v_pre is not None else 0,
                          seg_b.numel(), T, n_q, n_kv, D, scale, _p(o), o.stride(0), _p(lse), int(max_seg), _s())
    return o, lse


def attn_bwd_seg(q, k, v, o, do, lse, seg_b, seg_e, pre_b, pre_e, dep_e, T_valid, max_seg, n_q, n_kv, D, scale, dq, dk, dv, pairs=None):
    T = q.shape[0]
    if pairs is not None:
        prof_hint(K_ATTN_BWD, 10.0 * D * n_q * pairs)
    delta = torch.empty(n_q, T, dtype=F32, device=q.device)
    need = int(lib().st_attn_bwd_workspace_bytes(T, n_q, D))
    ws = _attn_ws.get(q.device)
    if ws is None or ws.numel() < need:
        ws = _attn_ws[q.device] = torch.empty(need, dtype=torch.uint8, device=q.device)
    lib().st_attn_bwd_seg(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(o), o.stride(0), _p(do), do.stride(0),
                          _p(lse), _p(seg_b), _p(seg_e), _p(pre_b), _p(pre_e), _p(dep_e), seg_b.numel(), T, int(T_valid), n_q, n_kv, D,
                          scale, _p(dq), dq.stride(0), _p(dk), dk.stride(0), _p(dv), dv.stride(0), _p(delta), _p(ws), need,
                          int(max_seg), _s())
    return dq, dk, dv


# ------------------------------------------------------------------ optimizer
def adamw_scalars(t: int, lr: float, beta1: float, beta2: float):
    """step_size and denom-correction exactly as torch forms them on a float32 0-d `step` tensor
    (verl/utils/torch_functional.py:306-309)."""
    step = torch.tensor(float(t), dtype=torch.float32)
    bc1 = 1 - beta1 ** step
    return float(lr / bc1), float((1 - beta2 ** step) ** 0.5)


def adamw_kahan_step_(p, grad_f32, m, v, c, *, t: int, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=None):
    _chk(p, BF16, "p"); _chk(grad_f32, F32, "grad")
    step_size, dc = adamw_scalars(t, lr, betas[0], betas[1])
    lib().st_adamw_kahan_step(_p(p), _p(grad_f32), _p(m), _p(v), _p(c), p.numel(), lr, betas[0], betas[1], eps, weight_decay,
                              step_size, dc, _p(grad_scale), _s())


def adamw_step_(p, grad_f32, m, v, *, t: int, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=None):
    """torch.optim.AdamW(fused=True) semantics on bf16 p / m / v (optim.strategy=adamw): bias corrections as torch's
    _fused_adamw forms them (python doubles, cast once to the fp32 opmath type)."""
    _chk(p, BF16, "p"); _chk(grad_f32, F32, "grad")
    bc1 = 1.0 - betas[0] ** t
    bc2_sqrt = (1.0 - betas[1] ** t) ** 0.5
    lib().st_adamw_step(_p(p), _p(grad_f32), _p(m), _p(v), p.numel(), lr, betas[0], betas[1], eps, weight_decay, bc1, bc2_sqrt,
                        _p(grad_scale), _s())


def adamw_master_step_(master_f32, p_bf16, grad_f32, m_f32, v_f32, *, t: int, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=None):
    """torch.optim.AdamW(fused=True) on fp32 master weights and fp32 moments; p_bf16 = bf16 rounding of the new master (st_adamw_master_step)."""
    _chk(master_f32, F32, "master"); _chk(p_bf16, BF16, "p"); _chk(grad_f32, F32, "grad"); _chk(m_f32, F32, "m"); _chk(v_f32, F32, "v")
    bc1 = 1.0 - betas[0] ** t
    bc2_sqrt = (1.0 - betas[1] ** t) ** 0.5
    lib().st_adamw_master_step(_p(master_f32), _p(p_bf16), _p(grad_f32), _p(m_f32), _p(v_f32), p_bf16.numel(), lr, betas[0], betas[1], eps,
                               weight_decay, bc1, bc2_sqrt, _p(grad_scale), _s())


def sumsq(x_f32, out=None, accumulate=False):
    scratch = torch.empty(1024, dtype=F32, device=x_f32.device)
    o = torch.zeros(1, dtype=F32, device=x_f32.device) if out is None else out
    lib().st_sumsq_f32(_p(x_f32), x_f32.numel(), _p(scratch), _p(o), int(accumulate), _s())
    return o


# ------------------------------------------------------------------ gather / scatter
def embed_gather(table, ids_i32, out=None):
    T, H = ids_i32.numel(), table.shape[1]
    o = torch.empty(T, H, dtype=BF16, device=table.device) if out is None else out
    lib().st_embed_gather(_p(table), table.stride(0), _p(ids_i32), _p(o), o.stride(0), T, H, _s())
    return o


def rows_gather(src, rows_i32, out=None):
    n, H = rows_i32.numel(), src.shape[1]
    o = torch.empty(n, H, dtype=BF16, device=src.device) if out is None else out
    lib().st_rows_gather(_p(src), src.stride(0), _p(rows_i32), _p(o), o.stride(0), n, H, _s())
    return o


def rows_scatter_(dst, rows_i32, src, add=False):
    lib().st_rows_scatter(_p(src), src.stride(0), _p(rows_i32), _p(dst), dst.stride(0), rows_i32.numel(), src.shape[1], int(add), _s())
    return dst


def rows_gather_sum(src, idx2d_i32, out=None):
    """out[u] = sum_r src[idx2d[u, r]] (fp32 accumulate, fixed order; -1 entries skipped)."""
    n_out, k = idx2d_i32.shape
    H = src.shape[1]
    o = torch.empty(n_out, H, dtype=BF16, device=src.device) if out is None else out
    lib().st_rows_gather_sum(_p(src), src.stride(0), _p(idx2d_i32), k, _p(o), o.stride(0), n_out, H, _s())
    return o


def embed_grad_(dtable_f32, ids_i32, dx):
    lib().st_embed_grad(_p(dx), dx.stride(0), _p(ids_i32), _p(dtable_f32), dtable_f32.stride(0), ids_i32.numel(), dx.shape[1], _s())


def cast_pad(x_f32, c_out: int):
    R, C = x_f32.shape
    o = torch.empty(R, c_out, dtype=BF16, device=x_f32.device)
    lib().st_cast_pad_f32_bf16(_p(x_f32), x_f32.stride(0), _p(o), o.stride(0), R, C, c_out, _s())
    return o


def add_(a, b, out=None):
    o = torch.empty_like(a) if out is None else out
    lib().st_add_bf16(_p(a), _p(b), _p(o), a.numel(), _s())
    return o


# ------------------------------------------------------------------ profiling hooks
_prof_on = [False] * 10


def clock_probe(out: torch.Tensor, slot: int, stream: "torch.cuda.Stream", n_blocks: int = 8, spin_us: int = 20):
    """One shader-clock sample into out[slot] ((n_slots, n_blocks, 2) int64: {shader cycles, 100-MHz reference ticks} per block) on
    `stream` (st_clock_probe) — a side stream, so that the sample is taken WHILE the compute stream's kernels run."""
    assert out.dtype == torch.int64 and out.is_cuda and out.shape[1:] == (n_blocks, 2) and out.is_contiguous()
    lib().st_clock_probe(out[slot].data_ptr(), n_blocks, spin_us, stream.cuda_stream)


def prof_enable(klass: int, max_events: int = 200000, stride: int = 1):
    """Event-time every `stride`-th launch of the class on its launch stream (hipGraph-captured launches are never timed)."""
    lib().st_prof_enable(klass, max_events)
    lib().st_prof_set_stride(klass, stride)
    _prof_on[klass] = True


def prof_read(klass: int):
    """(sampled launches, their summed ms, their summed algorithmic units)"""
    import ctypes
    n, ms, units = ctypes.c_int(0), ctypes.c_double(0.0), ctypes.c_double(0.0)
    lib().st_prof_read(klass, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(units))
    return n.value, ms.value, units.value


def prof_read_events(klass: int, max_events: int = 400000):
    """Per-launch view of the sampled launches (st_prof_read_events): (ms float32[n], units float64[n], tags uint64[n]); resets the class.
    GEMM-class tags decode with gemm_tag_decode()."""
    import ctypes
    ms = np.zeros(max_events, dtype=np.float32); units = np.zeros(max_events, dtype=np.float64); tags = np.zeros(max_events, dtype=np.uint64)
    n = ctypes.c_int(0)
    lib().st_prof_read_events(klass, max_events, ms.ctypes.data_as(ctypes.c_void_p), units.ctypes.data_as(ctypes.c_void_p),
                              tags.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n))
    k = min(n.value, max_events)
    return ms[:k], units[:k], tags[:k]


def gemm_tag_decode(tag: int) -> dict:
    """st_prof_tag of a GEMM launch -> {form, flags, M, N, K}"""
    tag = int(tag)
    form = ("nt", "nn", "tn", "swiglu")[(tag >> 60) & 15] if ((tag >> 60) & 15) < 4 else "?"
    fl = (tag >> 56) & 15
    flags = "+".join(n for b, n in ((1, "gu_kept" if form == "swiglu" else "bias"), (2, "residual"), (4, "f32out"), (8, "accumulate")) if fl & b)
    return {"form": form, "flags": flags, "M": tag & 0x3ffff, "N": (tag >> 18) & 0x3ffff, "K": (tag >> 36) & 0x3ffff}


def prof_seen(klass: int) -> int:
    return int(lib().st_prof_seen(klass))


def prof_hint(klass: int, units: float):
    if _prof_on[klass]:
        lib().st_prof_hint_units(klass, float(units))


# ------------------------------------------------------------------ rollout (decode) kernels
def attn_fwd_ranges(q, k, v, q_beg, q_end, k_beg, k_end, max_q, n_q, n_kv, D, scale, out, lse, o_beg=None, q_group=0,
                    pre_beg=None, pre_end=None, k_pre=None, v_pre=None):
    """out: (T_out, n_q*D) bf16 slab buffer, lse: (n_q, T_out) fp32 — both caller-owned (several launches fill disjoint slabs).
    pre_*/k_pre/v_pre: optional second key range per sequence in other tensors (see st_attn_fwd_ranges)."""
    lib().st_attn_fwd_ranges(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(q_beg), _p(q_end), _p(k_beg), _p(k_end),
                             _p(o_beg), int(q_group), q_beg.numel(), out.shape[0], n_q, n_kv, D, scale, _p(out), out.stride(0), _p(lse), int(max_q),
                             _p(pre_beg), _p(pre_end), _p(k_pre), k_pre.stride(0) if k_pre is not None else 0,
                             _p(v_pre), v_pre.stride(0) if v_pre is not None else 0, _s())
    return out, lse


def attn_decode_rows(q, k, v, q_beg, q_end, k_beg, k_end, max_q, n_heads, D, scale, out, lse, o_beg=None, q_group=0, slots=2):
    """Per-sample decode partials, one wave per (item, head) (st_attn_decode_rows): items of <= 32 query rows against their own keys."""
    lib().st_attn_decode_rows(_p(q), q.stride(0), _p(k), k.stride(0), _p(v), v.stride(0), _p(q_beg), _p(q_end), _p(k_beg), _p(k_end), _p(o_beg),
                              int(q_group), q_beg.numel(), out.shape[0], n_heads, D, scale, _p(out), out.stride(0), _p(lse), int(max_q), int(slots), _s())
    return out, lse


def decode_attn_select(persistent):
    """Kernel behind attn_fwd_ranges for decode-shaped launches: 0 / False = one workgroup per item (default), 1 / True = the persistent
    kernel with one workgroup per CU (4-slot ring), 2 = persistent with two workgroups per CU (2-slot rings); all bit-identical
    (st_decode_attn_select; A/B runs and the bit-identity test)."""
    lib().st_decode_attn_select(int(persistent))


def attn_merge(parts, lse, n_parts, heads, D, out=None, q_group=0):
    rows = parts.shape[0] // n_parts
    o = torch.empty(rows, heads * D, dtype=BF16, device=parts.device) if out is None else out
    lib().st_attn_merge(_p(parts), parts.stride(0), _p(lse), n_parts, _p(o), o.stride(0), rows, heads, D, int(q_group), _s())
    return o


def kv_append_(qkv, col_k, col_v, width, kg, vg, gen_len, active=None, increment=False):
    B = qkv.shape[0]
    lib().st_kv_append(_p(qkv), qkv.stride(0), col_k, col_v, width, _p(kg), _p(vg), kg.stride(0), _p(gen_len), _p(active), B,
                       int(increment), _s())


def sample(logits, temperature, seed, step=0, forced=None, top_k=-1, top_p=1.0, step_dev=None, out=None, row_ids=None, row_steps=None):
    B, V = logits.shape
    out = torch.empty(B, dtype=I32, device=logits.device) if out is None else out
    scratch = torch.empty(B * 33, dtype=F32, device=logits.device)
    lib().st_sample(_p(logits), logits.stride(0), B, V, float(temperature), int(top_k), float(top_p), int(seed), int(step), _p(step_dev),
                    _p(forced), _p(row_ids), _p(row_steps), _p(out), _p(scratch), _s())
    return out


def sample_partials(logits, temperature, seed, scratch, step=0, top_k=-1, top_p=1.0, row_ids=None, row_steps=None, lse_partials=None):
    """The sampler up to its 16 partial (value, index) pairs per row (left in `scratch`, B*33 floats); decode_step finishes them.
    lse_partials (B*32 floats, optional): per-split (max, sum exp) of logits / T for the rollout's own log-probs (decode_step logp_out)."""
    B, V = logits.shape
    lib().st_sample_partials(_p(logits), logits.stride(0), B, V, float(temperature), int(top_k), float(top_p), int(seed), int(step), None,
                             _p(row_ids), _p(row_steps), _p(scratch), _p(lse_partials), _s())


def decode_step(scratch, *, forced_len, forced_token, eos_ids, ignore_eos, gen_len, active, out_tokens, tok_out, slot_out, k_base, kb_gen, ke_gen,
                n_chunks, chunk_keys, pos, inv_freq, D, section, cos_out, sin_out, embed, x_out, lse_partials=None, logits=None, temperature=1.0,
                logp_out=None):
    """One launch between two decode forwards (st_decode_step): sampler finish + forced EOS, token record, live flags, response index,
    cache slot, generated-key range ends, the rows' M-RoPE cos/sin + position advance, embedding gather."""
    B, R = out_tokens.shape
    lib().st_decode_step(_p(scratch), _p(forced_len), int(forced_token), _p(eos_ids), 0 if eos_ids is None else eos_ids.numel(), int(bool(ignore_eos)),
                         _p(gen_len), _p(active), _p(out_tokens), R, _p(tok_out), _p(slot_out), _p(k_base), _p(kb_gen), _p(ke_gen), int(n_chunks),
                         int(chunk_keys), _p(pos), _p(inv_freq), D, section[0], section[1], section[2], _p(cos_out), _p(sin_out), _p(embed),
                         embed.stride(0), _p(x_out), x_out.stride(0), embed.shape[1], B, _p(lse_partials), _p(logits),
                         logits.stride(0) if logits is not None else 0, float(temperature), _p(logp_out), _s())


def prof_disable(klass: int):
    lib().st_prof_disable(klass)
    _prof_on[klass] = False
