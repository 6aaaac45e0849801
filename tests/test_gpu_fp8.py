"""MX-fp8 path (BASELINE config #5): quantiser bit-exact against the numpy oracle of the OCP MX rule; block-scaled
GEMM against the fp32 product of the de-quantised operands (the products of two e4m3 values and a power-of-two scale are exact in
fp32, so only the fp32 accumulation order and the final bf16 rounding differ: tolerance 2^-7 of the output scale); and the
quantisation error itself against the bf16 GEMM, which is the number DESIGN.md quotes for the fp8 tolerance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mxfp8 as MX  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    from spatialthinker_amd import ops as o
    return o


def _bf(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).bfloat16()


@pytest.mark.parametrize("shape", [(5, 128), (300, 1024), (64, 3584)])
def test_quantiser_bit_exact_vs_oracle(ops, shape):
    R, K = shape
    rs = np.random.RandomState(R + K)
    x = rs.standard_normal((R, K)).astype(np.float32) * np.exp(rs.uniform(-8, 6, (R, 1))).astype(np.float32)
    x[0, :32] = 0.0                                   # an all-zero block
    x[1 % R, 5] = 3.0e4                               # an outlier that dominates its block
    x[2 % R, 40:44] = [448.0, -448.0, 447.0, 1e-30]
    xb = _bf(x)
    q, sc = ops.mxfp8_quantize(xb.cuda())
    qo, sbo = MX.quantize(xb.float().numpy())
    assert np.array_equal(q.cpu().numpy(), qo)
    want_sc = MX.pack_scales(sbo, sc.shape[1])
    assert np.array_equal(sc.cpu().numpy().view(np.uint32)[:, :R], want_sc[:, :R])
    # round trip error of the format relative to the block maximum: half a step of the top binade (2^-4) for values below 448 * scale,
    # up to 2^-3 where a block maximum in (448, 512) * scale saturates (the OCP MX rule clamps, it does not widen the scale)
    dq = MX.dequantize(qo, sbo)
    blk = np.abs(xb.float().numpy()).reshape(R, K // 32, 32).max(-1, keepdims=True)
    err = np.abs(dq - xb.float().numpy()).reshape(R, K // 32, 32)
    assert float((err / np.maximum(blk, 1e-30)).max()) <= 2.0 ** -3 + 1e-6


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("shape", [(256, 256, 128), (512, 256, 256), (300, 520, 384), (1000, 3584, 1024), (4096, 4608, 3584), (515, 3584, 18944),
                                   (257, 37888, 3584)])
def test_gemm_vs_fp32_product_of_dequantised_operands(ops, shape, waves):
    """Both tiles: the 4-wave hand-scheduled one (gemm_mx4.hip; K-tile counts 1, 2, 3 exercise its prologue / re-fetch paths, the ragged
    shapes its clamped rows and the generic epilogue) and the 8-wave one."""
    M, N, K = shape
    prev = ops.gemm_mxfp8_select(waves)
    try:
        _gemm_case(ops, M, N, K)
    finally:
        ops.gemm_mxfp8_select(prev)


def _gemm_case(ops, M, N, K):
    rs = np.random.RandomState(M + N + K)
    # every (row, 32-k block) gets its own power-of-two magnitude: a scale taken from the wrong row or block shows up at once
    blk = lambda R: np.repeat(2.0 ** rs.randint(-3, 4, (R, K // 32)), 32, axis=1)
    a = _bf(rs.standard_normal((M, K)) * blk(M)).cuda()
    b = _bf(rs.standard_normal((N, K)) * blk(N) * (1 + np.arange(N)[:, None] / N)).cuda()       # asymmetric: catches a transposed C
    aq, sa = ops.mxfp8_quantize(a)
    bq, sb = ops.mxfp8_quantize(b)
    qa, sba = MX.quantize(a.float().cpu().numpy())
    qb, sbb = MX.quantize(b.float().cpu().numpy())
    want = torch.from_numpy(MX.dequantize(qa, sba)).cuda() @ torch.from_numpy(MX.dequantize(qb, sbb)).cuda().t()
    scale = float(want.abs().max())
    out = ops.gemm_mxfp8_nt(aq, sa, bq, sb)
    assert float((out.float() - want).abs().max()) < scale * 2 ** -7
    bias, res = _bf(rs.standard_normal(N)).cuda(), _bf(rs.standard_normal((M, N))).cuda()
    out2 = ops.gemm_mxfp8_nt(aq, sa, bq, sb, bias=bias, residual=res)
    assert float((out2.float() - (want + bias.float() + res.float())).abs().max()) < scale * 2 ** -6
    # quantisation error against the bf16 GEMM of the original operands (reported in DESIGN.md): a few percent of the output RMS
    ref = a.float() @ b.float().t()
    rel = float((out.float() - ref).norm() / ref.norm())
    print(f"MX-fp8 GEMM {M}x{N}x{K}: relative L2 error vs the bf16-operand product = {rel:.4f}")
    assert rel < 0.05


@pytest.mark.parametrize("shape", [(256, 128, 128), (300, 200, 384), (1000, 18944, 3584)])
def test_swiglu_epilogue_of_the_fp8_tile_matches_the_unfused_pair(ops, shape):
    """st_gemm_mxfp8_swiglu = st_gemm_mxfp8_nt on [gate | up] followed by st_swiglu_fwd, bit for bit (same accumulation order per output
    element, same roundings), incl. ragged rows / columns."""
    M, I, K = shape
    rs = np.random.RandomState(M + I + K)
    blk = lambda R: np.repeat(2.0 ** rs.randint(-2, 3, (R, K // 32)), 32, axis=1)
    a = _bf(rs.standard_normal((M, K)) * blk(M)).cuda()
    b = _bf(rs.standard_normal((2 * I, K)) * blk(2 * I) * K ** -0.5).cuda()
    aq, sa = ops.mxfp8_quantize(a)
    bq, sb = ops.mxfp8_quantize(b)
    prev = ops.gemm_mxfp8_select(4)
    try:
        want = ops.swiglu_fwd(ops.gemm_mxfp8_nt(aq, sa, bq, sb))
        got = ops.gemm_mxfp8_swiglu(aq, sa, bq, sb)
    finally:
        ops.gemm_mxfp8_select(prev)
    assert float(want.float().abs().max()) > 0.1
    assert torch.equal(got, want)


@pytest.mark.parametrize("shape", [(128, 64), (384, 200), (1024, 3584), (256, 37888)])
def test_transposing_quantiser_matches_quantiser_of_the_transpose(ops, shape):
    R, C = shape
    rs = np.random.RandomState(R + C)
    x = _bf(rs.standard_normal((R, C)) * np.exp(rs.uniform(-6, 4, (R, 1))) * np.exp(rs.uniform(-2, 2, (1, C)))).cuda()
    x[R // 2:R // 2 + 32, 3] = 0                                  # an all-zero block
    want = ops.mxfp8_quantize(x.t().contiguous())
    got = ops.mxfp8_quantize_t(x)
    assert torch.equal(got[0], want[0])
    assert torch.equal(got[1][:, :C], want[1][:, :C])


@pytest.mark.parametrize("shape", [(128, 128), (384, 256), (1024, 3584), (256, 37888)])
def test_one_pass_quantiser_for_both_gradient_gemms(ops, shape):
    R, C = shape
    rs = np.random.RandomState(R + C + 1)
    x = _bf(rs.standard_normal((R, C)) * np.exp(rs.uniform(-6, 4, (R, 1))) * np.exp(rs.uniform(-2, 2, (1, C)))).cuda()
    x[R // 2, :32] = 0
    (q, s), (qt, st) = ops.mxfp8_quantize_both(x)
    wq, ws = ops.mxfp8_quantize(x)
    wt, wts = ops.mxfp8_quantize_t(x)
    assert torch.equal(q, wq) and torch.equal(s[:, :R], ws[:, :R])
    assert torch.equal(qt, wt) and torch.equal(st[:, :C], wts[:, :C])


@pytest.mark.parametrize("shape", [(256, 256, 128), (300, 520, 384), (3584, 4608, 2048)])
def test_fp32_accumulating_gemm_of_the_fp8_tile(ops, shape):
    """st_gemm_mxfp8_nt_f32 (the weight-gradient form: fp32 result, = or +=) against the fp32 product of the de-quantised operands."""
    M, N, K = shape
    rs = np.random.RandomState(M + N + K + 7)
    blk = lambda R: np.repeat(2.0 ** rs.randint(-3, 4, (R, K // 32)), 32, axis=1)
    a = _bf(rs.standard_normal((M, K)) * blk(M)).cuda()
    b = _bf(rs.standard_normal((N, K)) * blk(N) * (1 + np.arange(N)[:, None] / N)).cuda()
    aq, sa = ops.mxfp8_quantize(a)
    bq, sb = ops.mxfp8_quantize(b)
    qa, sba = MX.quantize(a.float().cpu().numpy())
    qb, sbb = MX.quantize(b.float().cpu().numpy())
    want = torch.from_numpy(MX.dequantize(qa, sba)).cuda() @ torch.from_numpy(MX.dequantize(qb, sbb)).cuda().t()
    scale = float(want.abs().max())
    out = torch.full((M, N), 7.0, device="cuda")
    ops.gemm_mxfp8_nt_f32(aq, sa, bq, sb, out, accumulate=False)
    assert float((out - want).abs().max()) < scale * 2 ** -12
    base = torch.from_numpy(rs.standard_normal((M, N)).astype(np.float32)).cuda() * scale
    out2 = base.clone()
    ops.gemm_mxfp8_nt_f32(aq, sa, bq, sb, out2, accumulate=True)
    assert float((out2 - (base + want)).abs().max()) < scale * 2 ** -12


def _same_mx(got, want, R):
    assert torch.equal(got[0], want[0])
    assert torch.equal(got[1][:, :R], want[1][:, :R])


@pytest.mark.parametrize("shape", [(5, 128), (301, 3584), (1100, 2048), (2000, 3584)])
def test_rmsnorm_with_fp8_output_matches_rmsnorm_then_quantiser(ops, shape):
    """Bit-identical to st_rmsnorm_fwd + st_mxfp8_quantize where st_rmsnorm_fwd runs its wave-per-row kernel (T > 1024 or H > 4096: the
    same summation order); below that st_rmsnorm_fwd sums a row with a whole workgroup, rstd may differ in the last bit and y by one bf16
    step — there the quantised output must still be exactly the quantisation of the kernel's own y."""
    T, H = shape
    rs = np.random.RandomState(T + H)
    x = _bf(rs.standard_normal((T, H)) * np.exp(rs.uniform(-3, 3, (T, 1)))).cuda()
    w = _bf(1.0 + 0.3 * rs.standard_normal(H)).cuda()
    y0, r0 = ops.rmsnorm_fwd(x, w, 1e-6)
    y1, r1, got = ops.rmsnorm_mxfp8(x, w, 1e-6)
    if T > 1024 or H > 4096:
        assert torch.equal(y1, y0) and torch.equal(r1, r0)
    else:
        assert float(((r1 - r0).abs() / r0).max()) < 1e-6
        assert float(((y1.float() - y0.float()).abs() / y0.float().abs().clamp_min(1e-30)).max()) <= 2.0 ** -7
    want = ops.mxfp8_quantize(y1)
    _same_mx(got, want, T)
    y2, r2, got2 = ops.rmsnorm_mxfp8(x, w, 1e-6, want_y=False, want_rstd=False)
    assert y2 is None and r2 is None
    _same_mx(got2, want, T)


@pytest.mark.parametrize("shape", [(3, 128), (301, 18944), (700, 1408)])
def test_swiglu_with_fp8_output_matches_swiglu_then_quantiser(ops, shape):
    T, I = shape
    rs = np.random.RandomState(T + I)
    gu = _bf(rs.standard_normal((T, 2 * I)) * 2.0).cuda()
    m0 = ops.swiglu_fwd(gu)
    want = ops.mxfp8_quantize(m0)
    m1, got = ops.swiglu_mxfp8(gu)
    assert torch.equal(m1, m0)
    _same_mx(got, want, T)
    m2, got2 = ops.swiglu_mxfp8(gu, want_out=False)
    assert m2 is None
    _same_mx(got2, want, T)


@pytest.mark.parametrize("shape", [(256, 128, 128), (300, 256, 384), (1000, 18944, 3584)])
def test_swiglu_epilogue_with_fp8_output_matches_epilogue_then_quantiser(ops, shape):
    M, I, K = shape
    rs = np.random.RandomState(M + I + K + 1)
    blk = lambda R: np.repeat(2.0 ** rs.randint(-2, 3, (R, K // 32)), 32, axis=1)
    aq, sa = ops.mxfp8_quantize(_bf(rs.standard_normal((M, K)) * blk(M)).cuda())
    bq, sb = ops.mxfp8_quantize(_bf(rs.standard_normal((2 * I, K)) * blk(2 * I) * K ** -0.5).cuda())
    want = ops.mxfp8_quantize(ops.gemm_mxfp8_swiglu(aq, sa, bq, sb))
    got = ops.gemm_mxfp8_swiglu_q(aq, sa, bq, sb)
    assert int((want[0] != 0).sum()) > M * I // 2
    _same_mx(got, want, M)


def test_7b_dimension_layer_fp8_forward_vs_bf16_and_oracle():
    """Qwen25VL.enable_fp8 at the real 7B layer widths (1 LM layer + 2 ViT blocks, shared-prompt group of 6 rollouts): response
    log-probs with the four LM projections on the MX-fp8 path vs the bf16 engine and vs the fp32 oracle, and the straight-through
    backward.  fp8 is NOT a parity mode: DESIGN.md states the measured deviation (max |dlogp| 0.21, gradient relative error
    24-37 % on this random-init layer) and this test pins it at 1.3x."""
    import test_gpu_fullsize as F
    from oracle import qwen25vl as Q
    from oracle import rl_math as RM
    from spatialthinker_amd import model as mdl
    rs = np.random.RandomState(17)
    params = F._params()
    cfg = mdl.VLConfig(**F.FULL)
    store = mdl.ParamStore(cfg, trainable=True)
    store.load_hf_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    eng = mdl.Qwen25VL(cfg, store)
    ids, mask, pos, px, g, R = F._group_batch(rs, n_roll=6)
    k = ids.shape[0]
    rmask = mask[:, -R:]
    m = rmask.astype(bool)
    p32 = {n_: torch.from_numpy(v).clone().requires_grad_(n_.endswith(("q_proj.weight", "gate_proj.weight"))) for n_, v in params.items()}
    lp = Q.response_log_probs(p32, Q.VLConfig(**F.FULL), torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(pos), R, 1.0,
                              torch.from_numpy(np.concatenate([px] * k, 0)), np.concatenate([g] * k, 0))
    old = (lp.detach().numpy() + 0.2 * rs.standard_normal((k, R))).astype(np.float32)
    adv = rs.standard_normal((k, 1)).astype(np.float32).repeat(R, 1) * rmask
    _, gl = RM.actor_micro_batch_loss(lp.detach().numpy(), old, old, adv, rmask, kl_kind="low_var_kl", kl_coef=1e-2, grad_accum=1)
    lp.backward(torch.from_numpy(gl))
    b = eng.stage(ids, mask, pos, R, px, g, groups=[0] * k)
    assert b.pk.T_pad > 256                                    # the fp8 tile path is taken
    dv = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dt)
    li = dict(old_log_probs=dv(old), ref_log_probs=dv(old), advantages=dv(adv), response_mask=dv(rmask, torch.int64))
    kw = dict(clip_low=0.2, clip_high=0.3, clip_dual=3.0, kl_kind="low_var_kl", kl_coef=1e-2, grad_accum=1.0)
    res = {}
    for mode in ("bf16", "fp8", "fp8+dgrad", "fp8+dgrad+wgrad"):
        eng.enable_fp8(mode != "bf16", dgrad="dgrad" in mode, wgrad="wgrad" in mode)
        store.grad.zero_()
        lp_e, _ = eng.forward_backward(b, li, 1.0, **kw)
        grads = store.export_hf(store.g)
        err = float(np.abs(lp_e.cpu().numpy()[m] - lp.detach().numpy()[m]).max())
        rel = {n_: float(np.linalg.norm(grads[n_].float().cpu().numpy() - t.grad.numpy()) / np.linalg.norm(t.grad.numpy()))
               for n_, t in p32.items() if t.grad is not None}
        res[mode] = (err, rel, lp_e.clone())
        print(f"{mode}: max|dlogp| vs fp32 oracle {err:.4f}; grad rel err {rel}")
    eng.enable_fp8(False)
    assert store.wq is None
    assert not torch.equal(res["fp8"][2], res["bf16"][2])      # the fp8 path really ran
    # measured (round 2): bf16 0.0140 / 2.3-5.1 %, fp8 0.209 / 24-37 % — each projection carries ~4.2 % relative L2 quantisation error
    # (3 mantissa bits on both operands), which a random-init layer passes on undamped; bounds at 1.3x the measurement.  Round 4 (4-wave
    # tile: another fp32 summation order, so other elements sit on e4m3 rounding boundaries): fp8 0.263 / 26-41 %, fp8 + dgrad 27-44 %.
    assert res["fp8"][0] <= 0.28
    for n_ in res["bf16"][1]:
        assert res["fp8"][1][n_] <= 0.49, n_
    # fp8 input gradients on top (round 4): the forward is the same (same log-probs), the gradients pass through four more quantised
    # products per layer
    assert torch.equal(res["fp8+dgrad"][2], res["fp8"][2])
    assert any(res["fp8+dgrad"][1][n_] != res["fp8"][1][n_] for n_ in res["fp8"][1])       # the fp8 dgrad path really ran
    for n_ in res["bf16"][1]:
        assert res["fp8+dgrad"][1][n_] <= 0.60, n_
    # ... and fp8 weight gradients (token-minor MX blocks): only the LM's weight gradients change
    assert torch.equal(res["fp8+dgrad+wgrad"][2], res["fp8"][2])
    lm = [n_ for n_ in res["fp8"][1] if "language_model" in n_]
    assert lm and all(res["fp8+dgrad+wgrad"][1][n_] != res["fp8+dgrad"][1][n_] for n_ in lm)
    for n_ in res["bf16"][1]:
        assert res["fp8+dgrad+wgrad"][1][n_] <= 0.70, n_
