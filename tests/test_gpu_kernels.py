"""GPU parity tests of the individual HIP kernels, called through the C ABI (spatialthinker_amd.ops ->
libst_hip.so) and checked against the CPU oracle on the same seeded inputs.

Tolerances: kernels accumulate in fp32 and round once to bf16, so outputs are
compared with the fp32 oracle evaluated on identical bf16-rounded inputs at <= 1e-3 relative (bf16 outputs:
<= 1 bf16 ulp = 2^-8 relative); integer / index results are bit-exact.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import rl_math as M  # noqa: E402
from oracle import qwen25vl as Q  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from spatialthinker_amd import ops as _ops
    return _ops


def dev(x, dtype=None):
    t = torch.as_tensor(x)
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def bf(x):
    return torch.as_tensor(x, dtype=torch.float32).bfloat16()


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


# ------------------------------------------------------------------ log-prob
@pytest.mark.parametrize("T,V,temp", [(33, 512, 1.0), (7, 1000, 0.5), (5, 152064, 1.0), (3, 151936, 0.7), (4, 1003, 1.0)])
def test_logprob_fwd_bwd(ops, T, V, temp):
    rs = np.random.RandomState(V + T)
    z = bf(rs.standard_normal((T, V)) * 3)
    lab = torch.from_numpy(rs.randint(0, V, size=T))
    want = M.log_probs_from_logits(z.float().numpy() / temp, lab.numpy())
    zd = z.cuda()
    logp, lse = ops.logprob_fwd(zd, lab.cuda(), temp)
    np.testing.assert_allclose(logp.cpu().numpy(), want, rtol=0, atol=2e-5 * max(1.0, 1 / temp))
    g = rs.standard_normal(T).astype(np.float32)
    g[0] = 0.0                                                      # masked token: row must come back exactly zero
    want_g = M.log_probs_grad(z.float().numpy() / temp, lab.numpy(), g) / temp
    ops.logprob_bwd_(zd, lab.cuda(), lse, dev(g), temp)
    got = zd.float().cpu().numpy()
    assert np.all(got[0] == 0)
    np.testing.assert_allclose(got, want_g, rtol=2 ** -7, atol=1e-9)


def test_logprob_golden(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "rl_math.npz"))
    z = torch.from_numpy(g["lp512_logits_bf16_bits"]).view(torch.bfloat16)
    logp, _ = ops.logprob_fwd(z.cuda(), dev(g["lp512_labels"]), 1.0)
    np.testing.assert_allclose(logp.cpu().numpy(), g["lp512_logp"], rtol=0, atol=5e-6)
    V = 152064
    rs = np.random.RandomState(V)
    zz = bf(rs.standard_normal((5, V)) * 3)
    lab = rs.randint(0, V, size=5)
    logp, _ = ops.logprob_fwd(zz.cuda(), dev(lab), 1.0)
    np.testing.assert_allclose(logp.cpu().numpy(), g["lp152064_logp"], rtol=0, atol=2e-5)


def test_logprob_strided_and_bad_args(ops):
    from spatialthinker_amd.lib import StError
    z = bf(np.random.RandomState(0).standard_normal((6, 520))).cuda()
    view = z[:, :512]                                                # ld 520 > V 512
    lab = dev(np.arange(6) * 7)
    logp, _ = ops.logprob_fwd(view, lab, 1.0)
    want = M.log_probs_from_logits(view.float().cpu().numpy(), lab.cpu().numpy())
    np.testing.assert_allclose(logp.cpu().numpy(), want, atol=1e-5)
    with pytest.raises(StError):
        ops.logprob_fwd(view, lab, -1.0)
    with pytest.raises(RuntimeError):
        ops.logprob_fwd(view.cpu(), lab.cpu(), 1.0)                   # no CPU path


# ------------------------------------------------------------------ GRPO
def test_grpo_loss_golden(ops, golden_dir):
    g = np.load(os.path.join(golden_dir, "rl_math.npz"))
    f = lambda k: dev(g[k].reshape(-1))
    for ref, kind in ((f("pl_ref"), "low_var_kl"), (f("pl_ref"), "chi2"), (f("pl_ref"), "kl"), (f("pl_ref"), "abs"), (f("pl_ref"), "mse"), (None, "kl")):
        grad, met = ops.grpo_loss(f("pl_new"), f("pl_old"), ref, f("pl_adv"), f("pl_mask"), kl_kind=kind, kl_coef=1e-2, grad_accum=4.0)
        want_met, want_g = M.actor_micro_batch_loss(g["pl_new"], g["pl_old"], g["pl_ref"] if ref is not None else None, g["pl_adv"],
                                                    g["pl_mask"], kl_kind=kind, kl_coef=1e-2, grad_accum=4)
        met = met.cpu().numpy()
        np.testing.assert_allclose(met[0], want_met["pg_loss"], rtol=2e-5)
        np.testing.assert_allclose([met[1], met[2], met[3], met[4]],
                                   [want_met["pg_clipfrac_higher"], want_met["pg_clipfrac_lower"], want_met["ppo_kl"], want_met["entropy_loss"]], rtol=2e-5, atol=1e-7)
        if ref is not None:
            np.testing.assert_allclose(met[5], want_met["kl_loss"], rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(grad.cpu().numpy().reshape(g["pl_new"].shape), want_g, rtol=2e-5, atol=2e-9)
    # the reference's own numbers (torch autograd) for the shipped configuration
    grad, met = ops.grpo_loss(f("pl_new"), f("pl_old"), f("pl_ref"), f("pl_adv"), f("pl_mask"), kl_kind="low_var_kl", kl_coef=1e-2, grad_accum=4.0)
    np.testing.assert_allclose(met.cpu().numpy()[[0, 5, 4]], g["mb_metrics"], rtol=2e-5)
    np.testing.assert_allclose(grad.cpu().numpy().reshape(g["mb_grad"].shape), g["mb_grad"], rtol=2e-5, atol=2e-9)


@pytest.mark.parametrize("G", [4, 8, 16])
def test_grpo_advantage_golden(ops, golden_dir, G):
    g = np.load(os.path.join(golden_dir, "rl_math.npz"))
    uid = g[f"grpo{G}_uid"]
    _, dense = np.unique(uid, return_inverse=True)
    adv, status = ops.grpo_advantage(dev(g[f"grpo{G}_rewards"]), dev(g[f"grpo{G}_mask"]), dev(dense.astype(np.int32)), int(dense.max()) + 1)
    assert int(status.item()) == 0
    got, want = adv.cpu().numpy(), g[f"grpo{G}_adv"]
    live = uid != 0                                                 # group 0 is the zero-variance group: rounding noise / 1e-6
    if G <= 8:
        np.testing.assert_array_equal(got[live], want[live])        # bit-exact: same summation order as torch for n <= 8
    else:
        np.testing.assert_allclose(got[live], want[live], rtol=2e-6, atol=1e-6)
    assert np.all(np.abs(got[~live]) < 0.2)


def test_grpo_advantage_singleton_group_flags_error(ops):
    r = torch.zeros(3, 4); r[:, 1] = torch.tensor([1.0, 2.0, 3.0])
    adv, status = ops.grpo_advantage(r.cuda(), torch.ones(3, 4, dtype=torch.int64).cuda(), dev(np.array([0, 0, 1], np.int32)), 2)
    assert int(status.item()) == -1                                  # "GRPO needs rollout.n > 1" (core_algos.py:166)


# ------------------------------------------------------------------ RMSNorm / RoPE / activations
@pytest.mark.parametrize("T,H", [(1, 256), (37, 3584), (130, 1280), (9, 2048)])
def test_rmsnorm(ops, T, H):
    rs = np.random.RandomState(T * H)
    x, w = bf(rs.standard_normal((T, H)) * 2), bf(1 + 0.1 * rs.standard_normal(H))
    y, rstd = ops.rmsnorm_fwd(x.cuda(), w.cuda(), 1e-6)
    xf = x.float()
    r = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)
    want = (w.float() * (xf * r).bfloat16().float())
    np.testing.assert_allclose(rstd.cpu().numpy(), r[:, 0].numpy(), rtol=1e-5)
    assert rel_err(y.float().cpu().numpy(), want.numpy()) < 2 ** -7
    # backward vs autograd of the fp32 oracle
    xg = xf.clone().requires_grad_(True)
    wg = w.float().clone().requires_grad_(True)
    dy = bf(rs.standard_normal((T, H)))
    dres = bf(rs.standard_normal((T, H)))
    Q.rms_norm(xg, wg, 1e-6).backward(dy.float())
    dw = torch.zeros(H, dtype=torch.float32, device="cuda")
    dx = ops.rmsnorm_bwd(x.cuda(), w.cuda(), rstd, dy.cuda(), dres=dres.cuda(), dw_accum=dw)
    assert rel_err(dx.float().cpu().numpy(), (xg.grad + dres.float()).numpy()) < 1e-2
    assert rel_err(dw.cpu().numpy(), wg.grad.numpy()) < 1e-2


@pytest.mark.parametrize("T,H", [(7, 256), (130, 320), (1000, 1280), (513, 2048), (2500, 3584), (64, 4096)])
def test_rmsnorm_backward_one_pass_is_bit_identical_in_dx_and_deterministic_in_dw(ops, T, H):
    """st_rmsnorm_bwd_fused (round 6): one pass over x / dy, per-workgroup fp32 partial rows of dw added in a fixed order.  dx must equal the
    two-kernel form bit for bit (same per-row arithmetic in the same order), dw must equal the fp64 column sums to fp32 accumulation error,
    be the same bits on every run (the two-kernel form adds with atomics), honour dx aliasing dy and a NULL dw_accum."""
    import spatialthinker_amd.ops as O
    rs = np.random.RandomState(T + H)
    x, dy, dres = (bf(rs.standard_normal((T, H))).cuda() for _ in range(3))
    w = bf(1.0 + 0.1 * rs.standard_normal(H)).cuda()
    _, rstd = ops.rmsnorm_fwd(x, w, 1e-6)
    res = {}
    for fused in (False, True):
        O.RMSNORM_BWD_FUSED = fused
        try:
            dw = torch.full((H,), 0.5, dtype=torch.float32, device="cuda")                 # accumulates INTO dw_accum
            dx = ops.rmsnorm_bwd(x, w, rstd, dy, dres=dres, dw_accum=dw)
            dw2 = torch.full((H,), 0.5, dtype=torch.float32, device="cuda")
            dx2 = ops.rmsnorm_bwd(x, w, rstd, dy, dres=dres, dw_accum=dw2)
            dx_nores = ops.rmsnorm_bwd(x, w, rstd, dy)                                    # no residual gradient, no dw
            alias = dy.clone()
            ops.rmsnorm_bwd(x, w, rstd, alias, dres=dres, dw_accum=torch.zeros(H, dtype=torch.float32, device="cuda"), out=alias)
            torch.cuda.synchronize()
            res[fused] = (dx.clone(), dw.clone(), dw2.clone(), dx_nores.clone(), alias.clone(), dx2.clone())
        finally:
            O.RMSNORM_BWD_FUSED = True
    a, b_ = res[False], res[True]
    assert torch.equal(a[0].view(torch.int16), b_[0].view(torch.int16)) and torch.equal(a[3].view(torch.int16), b_[3].view(torch.int16))
    assert torch.equal(b_[0].view(torch.int16), b_[4].view(torch.int16)) and torch.equal(b_[0].view(torch.int16), b_[5].view(torch.int16))
    assert torch.equal(b_[1], b_[2])                                                          # the same bits on every run
    xhat = (x.float() * rstd[:, None]).bfloat16().double()                  # the kernel's own rounding points
    want = 0.5 + (dy.double() * xhat).sum(0)
    scale = float((dy.double() * xhat).abs().sum(0).max())
    assert float((b_[1].double() - want).abs().max()) < 2e-6 * scale + 1e-5
    assert float((a[1].double() - want).abs().max()) < 2e-6 * scale + 1e-5


def test_mrope_table_and_apply(ops):
    rs = np.random.RandomState(2)
    T, D, nq, nkv = 50, 128, 3, 2
    pos = rs.randint(0, 5000, size=(3, T)).astype(np.int32)
    inv = (1.0 / (1e6 ** (torch.arange(0, D, 2, dtype=torch.float32) / D)))
    cos, sin = ops.mrope_table(dev(pos), inv.cuda(), D, [16, 24, 24])
    wc, ws = Q.mrope_cos_sin(torch.from_numpy(pos.astype(np.int64)), D, 1e6, [16, 24, 24])
    np.testing.assert_allclose(cos.cpu().numpy(), wc[:, :D // 2].numpy(), atol=2e-4)     # fp32 sincos of angles up to 5e3
    np.testing.assert_allclose(sin.cpu().numpy(), ws[:, :D // 2].numpy(), atol=2e-4)
    qkv = bf(rs.standard_normal((T, (nq + 2 * nkv) * D)))
    x = qkv.cuda().clone()
    ops.rope_apply_(x, cos, sin, nq + nkv, D)
    xr = qkv.float().reshape(T, nq + 2 * nkv, D)
    c, s = torch.cat([cos.cpu(), cos.cpu()], -1)[:, None, :], torch.cat([sin.cpu(), sin.cpu()], -1)[:, None, :]
    want = xr.clone()
    want[:, :nq + nkv] = xr[:, :nq + nkv] * c + Q.rotate_half(xr[:, :nq + nkv]) * s
    got = x.float().cpu().reshape(T, nq + 2 * nkv, D)
    assert rel_err(got.numpy(), want.numpy()) < 2 ** -7
    assert torch.equal(got[:, nq + nkv:], xr[:, nq + nkv:])                                 # v heads untouched
    # inverse rotation is the transpose: <R x, y> == <x, R^T y>
    y = bf(rs.standard_normal((T, (nq + 2 * nkv) * D))).cuda()
    yt = y.clone()
    ops.rope_apply_(yt, cos, sin, nq + nkv, D, inverse=True)
    lhs = (x.float() * y.float()).sum().item()
    rhs = (qkv.cuda().float() * yt.float()).sum().item()
    assert abs(lhs - rhs) < 2e-2 * (abs(lhs) + 1)


def test_vision_rope_d80(ops):
    rs = np.random.RandomState(3)
    N, heads, D = 24, 4, 80
    ang = torch.from_numpy(rs.rand(N, D // 2).astype(np.float32) * 6)
    cos, sin = ang.cos(), ang.sin()
    qkv = bf(rs.standard_normal((N, 3 * heads * D)))
    x = qkv.cuda().clone()
    ops.rope_apply_(x, cos.cuda(), sin.cuda(), 2 * heads, D)
    xr = qkv.float().reshape(N, 3 * heads, D)
    c, s = torch.cat([cos, cos], -1)[:, None, :], torch.cat([sin, sin], -1)[:, None, :]
    want = xr.clone()
    want[:, :2 * heads] = xr[:, :2 * heads] * c + Q.rotate_half(xr[:, :2 * heads]) * s
    assert rel_err(x.float().cpu().reshape(N, 3 * heads, D).numpy(), want.numpy()) < 2 ** -7


def test_swiglu_gelu(ops):
    rs = np.random.RandomState(4)
    T, I = 19, 512
    gu = bf(rs.standard_normal((T, 2 * I)) * 2)
    out = ops.swiglu_fwd(gu.cuda())
    g, u = gu.float()[:, :I].clone().requires_grad_(True), gu.float()[:, I:].clone().requires_grad_(True)
    want = torch.nn.functional.silu(g) * u
    assert rel_err(out.float().cpu().numpy(), want.detach().numpy()) < 2 ** -6
    do = bf(rs.standard_normal((T, I)))
    want.backward(do.float())
    dgu = ops.swiglu_bwd(gu.cuda(), do.cuda())
    assert rel_err(dgu.float().cpu().numpy(), torch.cat([g.grad, u.grad], -1).numpy()) < 2 ** -6
    # the backward that recomputes the forward's result in the same pass (recompute_light): both outputs bit-identical to the two kernels
    dgu2, m2 = ops.swiglu_bwd(gu.cuda(), do.cuda(), want_m=True)
    assert torch.equal(dgu2, dgu) and torch.equal(m2, out)
    gu_inplace = gu.cuda().clone()
    dgu3, m3 = ops.swiglu_bwd(gu_inplace, do.cuda(), out=gu_inplace, want_m=True)          # dgu may alias gu
    assert torch.equal(dgu3, dgu) and torch.equal(m3, out)
    x = bf(rs.standard_normal(4096) * 2)
    xg = x.float().clone().requires_grad_(True)
    y = torch.nn.functional.gelu(xg)
    assert rel_err(ops.gelu_fwd(x.cuda()).float().cpu().numpy(), y.detach().numpy()) < 2 ** -7
    dy = bf(rs.standard_normal(4096))
    y.backward(dy.float())
    assert rel_err(ops.gelu_bwd(x.cuda(), dy.cuda()).float().cpu().numpy(), xg.grad.numpy()) < 2 ** -6


# ------------------------------------------------------------------ AdamW
def _torch_reference_step(p, g, st, t, lr, b1=0.9, b2=0.999, eps=1e-8, wd=1e-2):
    """The op sequence of verl/utils/torch_functional.py:296-320, executed by torch itself on the GPU
    (bf16 tensors): the ground truth for the reference's on-device rounding behaviour."""
    step = torch.tensor(float(t))
    if wd:
        p.mul_(1 - lr * wd)
    st["m"].mul_(b1).add_(g, alpha=1 - b1)
    st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    step_size = lr / bc1
    dc = (1 - b2 ** step) ** 0.5
    cv = (st["v"].sqrt() / dc).add_(eps, alpha=1)
    st["c"].addcdiv_(st["m"], cv, value=-step_size)
    tmp = p.detach().clone()
    p.add_(st["c"])
    st["c"].add_(tmp.sub_(p))


@pytest.mark.parametrize("lr", [1e-6, 1e-3])
def test_adamw_kahan_bit_exact_vs_torch_on_device_and_oracle(ops, lr):
    rs = np.random.RandomState(9)
    n = 8 * 1000 + 5                                                  # exercises the scalar tail
    p0 = bf(rs.standard_normal(n) * 0.02)
    pt = p0.cuda().clone()
    st = {k: torch.zeros(n, dtype=torch.bfloat16, device="cuda") for k in "mvc"}
    pk = p0.cuda().clone()
    sk = {k: torch.zeros(n, dtype=torch.bfloat16, device="cuda") for k in "mvc"}
    orc = M.AdamWKahanBF16(lr=lr, scalar_mode="gpu")
    po = p0.float().numpy()
    for t in range(1, 5):
        g = bf(rs.standard_normal(n) * (10.0 ** -rs.randint(1, 4)))
        _torch_reference_step(pt, g.cuda(), st, t, lr)
        ops.adamw_kahan_step_(pk, g.float().cuda(), sk["m"], sk["v"], sk["c"], t=t, lr=lr)
        po = orc.step(po, g.float().numpy(), lr=lr)
        for name, a, b in (("p", pk, pt), ("m", sk["m"], st["m"]), ("v", sk["v"], st["v"]), ("c", sk["c"], st["c"])):
            assert torch.equal(a, b), f"{name} differs from torch-on-device at step {t}: {(a != b).sum().item()} / {n}"
        np.testing.assert_array_equal(pk.float().cpu().numpy(), po, err_msg=f"oracle(gpu mode) step {t}")
    assert not torch.equal(pk, p0.cuda())


def test_sumsq(ops):
    x = torch.randn(1_000_003, generator=torch.Generator().manual_seed(1))
    got = ops.sumsq(x.cuda()).item()
    assert abs(got - float((x.double() ** 2).sum())) / got < 1e-6


# ------------------------------------------------------------------ GEMM & layout helpers
@pytest.mark.parametrize("M_,N,K", [(128, 128, 64), (256, 384, 512), (200, 1000, 256), (1, 136, 128), (517, 264, 3584), (64, 37888 // 8, 320), (2300, 12900, 128)])
def test_gemm_nt(ops, M_, N, K):
    rs = np.random.RandomState(M_ + N + K)
    a = bf(rs.standard_normal((M_, K)))
    b = bf(rs.standard_normal((N, K)) * (1 + np.arange(N)[:, None] / N))     # asymmetric: catches transposed C
    bias, res = bf(rs.standard_normal(N)), bf(rs.standard_normal((M_, N)))
    want = a.float() @ b.float().t()
    scale = np.abs(want.numpy()).max()
    out = ops.gemm_nt(a.cuda(), b.cuda())
    assert np.abs(out.float().cpu().numpy() - want.numpy()).max() < scale * 2 ** -7
    out = ops.gemm_nt(a.cuda(), b.cuda(), bias=bias.cuda(), residual=res.cuda())
    assert np.abs(out.float().cpu().numpy() - (want + bias.float() + res.float()).numpy()).max() < scale * 2 ** -7
    acc = torch.full((M_, N), 0.5, dtype=torch.float32, device="cuda")
    ops.gemm_nt(a.cuda(), b.cuda(), out_f32=acc, accumulate=True)
    assert np.abs(acc.cpu().numpy() - (want + 0.5).numpy()).max() < scale * 1e-5 + 1e-4
    ops.gemm_nt(a.cuda(), b.cuda(), out_f32=acc, accumulate=False)
    assert np.abs(acc.cpu().numpy() - want.numpy()).max() < scale * 1e-5 + 1e-4


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 23, 40])
@pytest.mark.parametrize("K", [64, 128, 192, 256, 320, 448])
def test_gemm_tile_variants_all_pipeline_lengths(ops, variant, K):
    """Every tile/pipeline variant (2- and 3-slot rings, the mid-tile-barrier schedule) at 1, 2, 3 and 5 K-tiles — the
    prologue / steady state / peeled tail paths — with ragged M, N; bf16 and fp32-accumulate epilogues."""
    rs = np.random.RandomState(variant * 100 + K)
    M_, N = 300, 520
    a = bf(rs.standard_normal((M_, K))).cuda()
    b = bf(rs.standard_normal((N, K)) * (1 + np.arange(N)[:, None] / N)).cuda()
    want = a.float() @ b.float().t()
    scale = float(want.abs().max())
    out = ops.gemm_nt_variant(variant, a, b)
    assert float((out.float() - want).abs().max()) < scale * 2 ** -7
    acc = torch.full((M_, N), 0.25, dtype=torch.float32, device="cuda")
    ops.gemm_nt_variant(variant, a, b, out_f32=acc, accumulate=True)
    assert float((acc - want - 0.25).abs().max()) < scale * 1e-5 + 1e-4


@pytest.mark.parametrize("variant", [6, 8, 23, 40])
@pytest.mark.parametrize("shape", [(4200, 4096, 4096), (3000, 5700, 3200), (70000, 300, 4096)])
def test_gemm_tail_split_matches_unsplit(ops, variant, shape):
    """st_gemm_set_workspace: the tiles beyond whole rounds of the CUs are cut into K-slices (fp32 partials + a finish launch
    that runs the epilogue).  Every epilogue kind, ragged edges, against the unsplit launch and the fp32 product."""
    M_, N, K = shape
    rs = np.random.RandomState(M_ % 97)
    a = torch.from_numpy(rs.standard_normal((M_, K)).astype(np.float32)).cuda().bfloat16()
    b = torch.from_numpy(rs.standard_normal((N, K)).astype(np.float32)).cuda().bfloat16()
    bias = torch.from_numpy(rs.standard_normal(N).astype(np.float32)).cuda().bfloat16()
    res = torch.from_numpy(rs.standard_normal((M_, N)).astype(np.float32)).cuda().bfloat16()
    want = a.float() @ b.float().t()
    scale = float(want.abs().max())
    outs = {}
    try:
        for on in (False, True):
            ops.gemm_tail_split(on)
            c = ops.gemm_nt_variant(variant, a, b)
            c2 = ops.gemm_nt_variant(variant, a, b, bias=bias, residual=res)
            f = torch.full((M_, N), 0.5, dtype=torch.float32, device="cuda")
            ops.gemm_nt_variant(variant, a, b, out_f32=f, accumulate=True)
            outs[on] = (c, c2, f)
    finally:
        ops.gemm_tail_split(True)
    tiles = -(-M_ // 256) * -(-N // 256)
    assert tiles > 256 and tiles % 256 != 0                  # the shapes do leave a tail on a 256-CU part
    for on in (False, True):
        c, c2, f = outs[on]
        assert float((c.float() - want).abs().max()) < scale * 2 ** -7, on
        assert float((c2.float() - (want + bias.float() + res.float())).abs().max()) < scale * 2 ** -6, on
        assert float((f - want - 0.5).abs().max()) < scale * 1e-5 + 1e-3, on
    assert float((outs[True][2] - outs[False][2]).abs().max()) < scale * 1e-5 + 1e-3
    assert not torch.equal(outs[True][2], outs[False][2])     # the split really ran (different fp32 summation order somewhere)


@pytest.mark.parametrize("shape", [(300, 520, 192), (1000, 777 * 8, 256), (513, 1016, 128)])
def test_gemm_lds_staged_epilogue_is_bit_identical_to_direct(ops, shape):
    """Variant 23 = variant 6 with the LDS-staged epilogue (accumulators -> LDS -> row-contiguous 16-byte vectors): same MFMA
    order, same fp32 epilogue arithmetic, so every output kind must equal variant 6 bit for bit — ragged M, N not a multiple of the
    tile, and column counts that leave partial 8-column vectors at the edge."""
    M_, N, K = shape
    rs = np.random.RandomState(M_ + N)
    a = bf(rs.standard_normal((M_, K))).cuda()
    b = bf(rs.standard_normal((N, K)) * (1 + np.arange(N)[:, None] / N)).cuda()
    bias, res = bf(rs.standard_normal(N)).cuda(), bf(rs.standard_normal((M_, N))).cuda()
    for kw in ({}, {"bias": bias}, {"residual": res}, {"bias": bias, "residual": res}):
        assert torch.equal(ops.gemm_nt_variant(23, a, b, **kw), ops.gemm_nt_variant(6, a, b, **kw)), list(kw)
    for accumulate in (True, False):
        f6 = torch.full((M_, N), 0.25, dtype=torch.float32, device="cuda"); f23 = f6.clone()
        ops.gemm_nt_variant(6, a, b, out_f32=f6, accumulate=accumulate)
        ops.gemm_nt_variant(23, a, b, out_f32=f23, accumulate=accumulate)
        assert torch.equal(f6, f23), accumulate
    # a strided output / residual view (row pitch > N), as the engine's qkv and gate|up buffers have
    big = torch.zeros(M_, N + 64, dtype=torch.bfloat16, device="cuda")
    resb = torch.zeros(M_, N + 64, dtype=torch.bfloat16, device="cuda"); resb[:, 8:N + 8] = res
    ops.gemm_nt_variant(23, a, b, out=big[:, 8:N + 8], residual=resb[:, 8:N + 8])
    assert torch.equal(big[:, 8:N + 8], ops.gemm_nt_variant(6, a, b, residual=res)) and float(big[:, :8].abs().max()) == 0


def test_gemm_identity_layout(ops):
    """A = I (padded) against an asymmetric B: C must equal B^T block exactly."""
    K = 128
    a = torch.eye(K).bfloat16()
    b = bf(np.arange(96 * K).reshape(96, K) % 251 - 125)
    out = ops.gemm_nt(a.cuda(), b.cuda())
    assert torch.equal(out.float().cpu(), b.float().t())


def test_transpose_colsum_gather(ops):
    rs = np.random.RandomState(6)
    x = bf(rs.standard_normal((130, 200)))
    assert torch.equal(ops.transpose(x.cuda()).cpu(), x.t().contiguous())
    np.testing.assert_allclose(ops.colsum(x.cuda()).cpu().numpy(), x.float().sum(0).numpy(), rtol=1e-5, atol=1e-4)
    rows = torch.from_numpy(rs.permutation(130)[:50].astype(np.int32))
    got = ops.rows_gather(x.cuda()[:, :192], rows.cuda())
    assert torch.equal(got.cpu(), x[rows.long(), :192])
    dst = torch.zeros(130, 192, dtype=torch.bfloat16, device="cuda")
    ops.rows_scatter_(dst, rows.cuda(), got)
    assert torch.equal(dst.cpu()[rows.long()], x[rows.long(), :192])
    ops.rows_scatter_(dst, rows.cuda(), got, add=True)
    assert torch.equal(dst.cpu()[rows.long()], (x[rows.long(), :192].float() * 2).bfloat16())
    pv = torch.from_numpy(rs.standard_normal((5, 1176)).astype(np.float32))
    cp = ops.cast_pad(pv.cuda(), 1216).cpu()
    assert torch.equal(cp[:, :1176], pv.bfloat16()) and torch.all(cp[:, 1176:] == 0)
    ids = torch.tensor([3, -1, 3, 7], dtype=torch.int32)
    dx = bf(rs.standard_normal((4, 64)))
    dt = torch.zeros(10, 64, dtype=torch.float32, device="cuda")
    ops.embed_grad_(dt, ids.cuda(), dx.cuda())
    want = torch.zeros(10, 64)
    want[3] = dx[0].float() + dx[2].float(); want[7] = dx[3].float()
    np.testing.assert_allclose(dt.cpu().numpy(), want.numpy(), atol=1e-6)


# ------------------------------------------------------------------ attention
def _attn_case(rs, lens, n_q, n_kv, D):
    T = sum(lens)
    width = (n_q + 2 * n_kv) * D
    qkv = bf(rs.standard_normal((T, width)))
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    return qkv, cu


def test_vit_full_attention_through_the_d128_kernels_on_padded_heads(ops, measured):
    """Qwen25VL._vit_attn_fwd / _vit_attn_bwd: head dim 80 zero-padded to 128 so that whole-image ViT attention (1344 patches at the STVQA
    shape) runs on the D = 128 kernels.  Forward output and dq / dk / dv against the fp32 dense reference (same bounds as the D = 80
    kernels' own tests) and against the D = 80 kernels."""
    import types
    from spatialthinker_amd.model import Qwen25VL
    heads, hd, lens = 16, 80, [1344, 700]
    rs = np.random.RandomState(80)
    qkv_h, cu = _attn_case(rs, lens, heads, heads, hd)
    T = qkv_h.shape[0]
    qkv = qkv_h.cuda()
    me = types.SimpleNamespace(cfg=types.SimpleNamespace(v_heads=heads, v_head_dim=hd, v_hidden=heads * hd), v_scale=hd ** -0.5,
                               VIT_PAD_MIN_SEQ=512, VIT_PAD_MIN_SEQ_BWD=512, _vit_pad=lambda x: Qwen25VL._vit_pad(None, x))
    a = torch.zeros(T, heads * hd, dtype=torch.bfloat16, device="cuda")
    lse = Qwen25VL._vit_attn_fwd(me, qkv, dev(cu), max(lens), a, None)
    xf = qkv_h.float()
    W = heads * hd
    want = Q.dense_attention(xf[:, :W].reshape(T, heads, hd), xf[:, W:2 * W].reshape(T, heads, hd), xf[:, 2 * W:].reshape(T, heads, hd), cu, False).reshape(T, W)
    err = float(np.abs(a.float().cpu().numpy() - want.numpy()).max())
    a80, lse80 = ops.attn_fwd(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], dev(cu), max(lens), heads, heads, hd, hd ** -0.5, False)
    measured("vit_padded_d128_fwd_abs", err)
    assert err < 1.05e-2, err
    assert float((a.float() - a80.float()).abs().max()) < 1.6e-2 and float((lse - lse80).abs().max()) < 2e-3
    do = bf(rs.standard_normal((T, W)) * 0.5).cuda()
    dqkv, dqkv80 = torch.zeros_like(qkv), torch.zeros_like(qkv)
    Qwen25VL._vit_attn_bwd(me, qkv, a, do, lse, dev(cu), max(lens), dqkv, None)
    ops.attn_bwd(qkv[:, :W], qkv[:, W:2 * W], qkv[:, 2 * W:], a80, do, lse80, dev(cu), max(lens), heads, heads, hd, hd ** -0.5, False,
                 dqkv80[:, :W], dqkv80[:, W:2 * W], dqkv80[:, 2 * W:])
    # fp32 autograd reference
    qf = xf.clone().cuda().requires_grad_(True)
    qq, kk, vv = qf[:, :W].reshape(T, heads, hd), qf[:, W:2 * W].reshape(T, heads, hd), qf[:, 2 * W:].reshape(T, heads, hd)
    outs = []
    for b0, b1 in zip(cu[:-1], cu[1:]):
        p_ = torch.softmax(torch.einsum("qhd,khd->hqk", qq[b0:b1], kk[b0:b1]) * hd ** -0.5, -1)
        outs.append(torch.einsum("hqk,khd->qhd", p_, vv[b0:b1]).reshape(b1 - b0, W))
    (torch.cat(outs) * do.float()).sum().backward()
    ref = qf.grad
    for name, sl in (("dq", slice(0, W)), ("dk", slice(W, 2 * W)), ("dv", slice(2 * W, 3 * W))):
        e_pad = float((dqkv[:, sl].float() - ref[:, sl]).abs().max() / ref[:, sl].abs().max())
        e_80 = float((dqkv80[:, sl].float() - ref[:, sl]).abs().max() / ref[:, sl].abs().max())
        measured(f"vit_padded_d128_{name}_rel", e_pad)
        assert e_pad < 8.0e-3 and e_80 < 8.0e-3, (name, e_pad, e_80)


@pytest.mark.parametrize("lens,n_q,n_kv,D,causal", [
    ([5], 2, 1, 128, True), ([130, 64, 1, 257], 4, 2, 128, True), ([200, 77], 7, 1, 128, True),
    ([64, 64, 16, 48], 4, 4, 80, False), ([300], 2, 2, 80, False), ([100, 33], 2, 1, 128, False),
    ([700, 129, 383], 7, 1, 128, True), ([450, 65], 2, 2, 128, False)])      # many K/V tiles: the staged pipelines wrap their rings
def test_attn_fwd(ops, measured, lens, n_q, n_kv, D, causal):
    rs = np.random.RandomState(sum(lens) + D)
    qkv, cu = _attn_case(rs, lens, n_q, n_kv, D)
    T = qkv.shape[0]
    x = qkv.cuda()
    q, k, v = x[:, :n_q * D], x[:, n_q * D:(n_q + n_kv) * D], x[:, (n_q + n_kv) * D:]
    scale = D ** -0.5
    out, lse = ops.attn_fwd(q, k, v, dev(cu), max(lens), n_q, n_kv, D, scale, causal)
    xf = qkv.float()
    want = Q.dense_attention(xf[:, :n_q * D].reshape(T, n_q, D), xf[:, n_q * D:(n_q + n_kv) * D].reshape(T, n_kv, D),
                             xf[:, (n_q + n_kv) * D:].reshape(T, n_kv, D), cu, causal).reshape(T, n_q * D)
    err = np.abs(out.float().cpu().numpy() - want.numpy()).max()
    measured(f"attn_fwd_D{D}_{'c' if causal else 'f'}_{sum(lens)}", err)
    assert err < 1.05e-2, err                                        # measured <= 0.0080 over the cases (1.3x): P is rounded to bf16 before PV (as
                                                                     # flash-attn does) and O(1) outputs land on the bf16 grid (half an ulp at 2..4 = 0.0078)
    # spiked key forces the online-softmax rescale branch on a late tile
    qkv2 = qkv.clone().float()
    if lens[0] > 70:
        qkv2[69, n_q * D:(n_q + n_kv) * D] = qkv2[lens[0] - 1, :D].repeat(n_kv) * 4
    x2 = qkv2.bfloat16().cuda()
    out2, _ = ops.attn_fwd(x2[:, :n_q * D], x2[:, n_q * D:(n_q + n_kv) * D], x2[:, (n_q + n_kv) * D:], dev(cu), max(lens), n_q, n_kv, D, scale, causal)
    xf2 = x2.float().cpu()
    want2 = Q.dense_attention(xf2[:, :n_q * D].reshape(T, n_q, D), xf2[:, n_q * D:(n_q + n_kv) * D].reshape(T, n_kv, D),
                              xf2[:, (n_q + n_kv) * D:].reshape(T, n_kv, D), cu, causal).reshape(T, n_q * D)
    err2 = np.abs(out2.float().cpu().numpy() - want2.numpy()).max()
    measured(f"attn_fwd_spiked_D{D}_{'c' if causal else 'f'}_{sum(lens)}", err2)
    assert err2 < 1.05e-2, err2                                      # measured <= 0.0080


@pytest.mark.parametrize("lens,n_q,n_kv,D,causal", [
    ([5], 2, 1, 128, True), ([130, 64, 1, 257], 4, 2, 128, True), ([200, 77], 7, 1, 128, True),
    ([64, 64, 16, 48], 4, 4, 80, False), ([300], 2, 2, 80, False), ([100, 33], 2, 1, 128, False),
    ([700, 129, 383], 7, 1, 128, True), ([450, 65], 2, 2, 128, False)])      # many K/V tiles: the staged pipelines wrap their rings
def test_attn_bwd(ops, measured, lens, n_q, n_kv, D, causal):
    rs = np.random.RandomState(sum(lens) + D + 1)
    qkv, cu = _attn_case(rs, lens, n_q, n_kv, D)
    T = qkv.shape[0]
    x = qkv.cuda()
    q, k, v = x[:, :n_q * D], x[:, n_q * D:(n_q + n_kv) * D], x[:, (n_q + n_kv) * D:]
    scale = D ** -0.5
    out, lse = ops.attn_fwd(q, k, v, dev(cu), max(lens), n_q, n_kv, D, scale, causal)
    do = bf(rs.standard_normal((T, n_q * D)))
    dqkv = torch.zeros_like(x)
    dq, dk, dv = dqkv[:, :n_q * D], dqkv[:, n_q * D:(n_q + n_kv) * D], dqkv[:, (n_q + n_kv) * D:]
    ops.attn_bwd(q, k, v, out, do.cuda(), lse, dev(cu), max(lens), n_q, n_kv, D, scale, causal, dq, dk, dv)
    xf = qkv.float().clone().requires_grad_(True)
    o = Q.dense_attention(xf[:, :n_q * D].reshape(T, n_q, D), xf[:, n_q * D:(n_q + n_kv) * D].reshape(T, n_kv, D),
                          xf[:, (n_q + n_kv) * D:].reshape(T, n_kv, D), cu, causal).reshape(T, n_q * D)
    o.backward(do.float())
    got, want = dqkv.float().cpu().numpy(), xf.grad.numpy()
    for name, sl in (("dq", slice(0, n_q * D)), ("dk", slice(n_q * D, (n_q + n_kv) * D)), ("dv", slice((n_q + n_kv) * D, None))):
        err = np.abs(got[:, sl] - want[:, sl]).max() / (np.abs(want[:, sl]).max() + 1e-9)
        measured(f"attn_bwd_{name}_D{D}_{'c' if causal else 'f'}_{sum(lens)}", err)
        assert err < 8.0e-3, (name, err)                            # measured <= 0.0061 of the gradient's max (1.3x)


def test_vit_window_attention_one_launch_kernels(ops, measured):
    """D = 80 bidirectional sequences of <= 64 tokens (the ViT's windows) take attention_win.hip: one (window, head) pair per two-wave
    workgroup, the backward in ONE launch.  Forward bit-identical to the generic D = 80 kernel (forced by passing max_seqlen = 65, which
    only sizes its grid); backward against the fp32 autograd reference at the generic kernels' bound and next to the generic kernels."""
    rs = np.random.RandomState(80)
    lens = [64, 64, 33, 32, 31, 1, 17, 48, 64, 2, 63, 64, 8, 40, 64, 64]
    n_q, D = 16, 80
    qkv, cu = _attn_case(rs, lens, n_q, n_q, D)
    T, W = qkv.shape[0], n_q * D
    x = qkv.cuda()
    q, k, v = x[:, :W], x[:, W:2 * W], x[:, 2 * W:]
    scale = D ** -0.5
    out, lse = ops.attn_fwd(q, k, v, dev(cu), 64, n_q, n_q, D, scale, False)
    out_g, lse_g = ops.attn_fwd(q, k, v, dev(cu), 65, n_q, n_q, D, scale, False)
    assert torch.equal(out, out_g) and torch.equal(lse, lse_g)
    do = bf(rs.standard_normal((T, W))).cuda()
    dqkv, dqkv_g = torch.zeros_like(x), torch.zeros_like(x)
    ops.attn_bwd(q, k, v, out, do, lse, dev(cu), 64, n_q, n_q, D, scale, False, dqkv[:, :W], dqkv[:, W:2 * W], dqkv[:, 2 * W:])
    ops.attn_bwd(q, k, v, out, do, lse, dev(cu), 65, n_q, n_q, D, scale, False, dqkv_g[:, :W], dqkv_g[:, W:2 * W], dqkv_g[:, 2 * W:])
    xf = qkv.float().clone().requires_grad_(True)
    o = Q.dense_attention(xf[:, :W].reshape(T, n_q, D), xf[:, W:2 * W].reshape(T, n_q, D), xf[:, 2 * W:].reshape(T, n_q, D), cu, False).reshape(T, W)
    o.backward(do.float().cpu())
    got, gen, want = dqkv.float().cpu().numpy(), dqkv_g.float().cpu().numpy(), xf.grad.numpy()
    for name, sl in (("dq", slice(0, W)), ("dk", slice(W, 2 * W)), ("dv", slice(2 * W, None))):
        den = np.abs(want[:, sl]).max() + 1e-9
        err, err_g = np.abs(got[:, sl] - want[:, sl]).max() / den, np.abs(gen[:, sl] - want[:, sl]).max() / den
        measured(f"vit_window_bwd_{name}_rel", err)
        assert err < 8.0e-3 and err < 1.5 * err_g + 1e-3, (name, err, err_g)


# ------------------------------------------------------------------ fused decode epilogues vs the unfused launch chain
@pytest.mark.parametrize("M,B", [(32, 24), (64, 64), (160, 150)])
def test_decode_fused_ops_bit_identical(ops, M, B):
    rs = np.random.RandomState(M)
    n_q, n_kv, D, H, I = 4, 2, 128, 512, 8200          # 2I >= 16384: the gate/up GEMM takes the no-split plan like the real MLP
    N = (n_q + 2 * n_kv) * D
    dev_ = lambda a, dt=torch.bfloat16: torch.from_numpy(a).to("cuda").to(dt)
    h = dev_(rs.randn(M, H).astype(np.float32))
    wqkv, bqkv = dev_(rs.randn(N, H).astype(np.float32) * 0.05), dev_(rs.randn(N).astype(np.float32))
    cos = torch.from_numpy(np.cos(rs.rand(B, D // 2) * 6).astype(np.float32)).cuda()
    sin = torch.from_numpy(np.sin(rs.rand(B, D // 2) * 6).astype(np.float32)).cuda()
    R, width = 5, n_kv * D
    gen_len = torch.from_numpy(rs.randint(0, R, B).astype(np.int32)).cuda()
    # --- unfused: GEMM(+bias) -> RoPE -> KV append
    qkv = ops.gemm_nt(h, wqkv, bias=bqkv)
    ops.rope_apply_(qkv[:B], cos, sin, n_q + n_kv, D)
    kg0, vg0 = torch.zeros(B, R, width, dtype=torch.bfloat16, device="cuda"), torch.zeros(B, R, width, dtype=torch.bfloat16, device="cuda")
    ops.kv_append_(qkv[:B], n_q * D, n_q * D + width, width, kg0, vg0, gen_len)
    # --- fused
    slabs, sp = ops.gemm_nt_decode_slabs(h, wqkv)
    qb = torch.zeros(M, n_q * D, dtype=torch.bfloat16, device="cuda")
    kg1, vg1 = torch.zeros_like(kg0), torch.zeros_like(vg0)
    ops.decode_finish_qkv(slabs, sp, M, bqkv, cos, sin, qb, kg1, vg1, gen_len, B, n_q, n_kv, D)
    assert torch.equal(qb[:B], qkv[:B, :n_q * D]) and torch.equal(kg0, kg1) and torch.equal(vg0, vg1)

    # --- finish + residual + RMSNorm
    a = dev_(rs.randn(M, n_q * D).astype(np.float32))
    wo = dev_(rs.randn(H, n_q * D).astype(np.float32) * 0.05)
    res = dev_(rs.randn(M, H).astype(np.float32))
    nw = dev_(1.0 + 0.1 * rs.randn(H).astype(np.float32))
    x_ref = ops.gemm_nt(a, wo, residual=res)
    h_ref, _ = ops.rmsnorm_fwd(x_ref, nw, 1e-6, want_rstd=False)
    slabs, sp = ops.gemm_nt_decode_slabs(a, wo)
    x1, h1 = torch.empty_like(x_ref), torch.empty_like(h_ref)
    ops.decode_finish_norm(slabs, sp, M, H, residual=res, x_out=x1, norm_w=nw, eps=1e-6, h_out=h1)
    assert torch.equal(x1, x_ref) and torch.equal(h1, h_ref)

    # --- SwiGLU in the gate/up epilogue
    wgu = dev_(rs.randn(2 * I, H).astype(np.float32) * 0.05)
    ref = ops.swiglu_fwd(ops.gemm_nt(h, wgu))
    got = ops.gemm_swiglu_decode(h, wgu)
    assert torch.equal(ref, got)


# ------------------------------------------------------------------ shared-prefix (segment) attention
@pytest.mark.parametrize("groups", [[(70, [33, 1, 64])], [(200, [150, 129]), (65, [5]), (0, [90])], [(333, [257, 64, 100, 31])]])
def test_attn_seg_shared_prefix_fwd_bwd(ops, measured, groups):
    """Packed [prompt][resp_1]..[resp_k] with the prompt stored once vs dense fp32 attention over every full sequence
    (autograd sums the prompt gradients of the k rollouts)."""
    n_q, n_kv, D = 4, 2, 128
    g = n_q // n_kv
    rs = np.random.RandomState(sum(p for p, _ in groups))
    segs = []          # (b, e, pre_b, pre_e, dep_e)
    row = 0
    for P, resps in groups:
        pb, pe = row, row + P
        gend = pe + sum(resps)
        if P:
            segs.append((pb, pe, 0, 0, gend))
        row = pe
        for R in resps:
            segs.append((row, row + R, pb, pe, row + R))
            row += R
    T = row
    T_pad = (T + 127) // 128 * 128
    W = (n_q + 2 * n_kv) * D
    qkv_np = rs.standard_normal((T_pad, W)).astype(np.float32)
    qkv = bf(qkv_np).cuda()
    do = bf(rs.standard_normal((T_pad, n_q * D))).cuda()
    q, k, v = qkv[:, :n_q * D], qkv[:, n_q * D:(n_q + n_kv) * D], qkv[:, (n_q + n_kv) * D:]
    ti = lambda i: torch.tensor([s_[i] for s_ in segs], dtype=torch.int32, device="cuda")
    seg_b, seg_e, pre_b, pre_e, dep_e = (ti(i) for i in range(5))
    max_seg = max(e - b for b, e, *_ in segs)
    scale = D ** -0.5
    out, lse = ops.attn_fwd_seg(q, k, v, seg_b, seg_e, pre_b, pre_e, max_seg, n_q, n_kv, D, scale)
    dq = torch.zeros(T_pad, n_q * D, dtype=torch.bfloat16, device="cuda")
    dk = torch.zeros(T_pad, n_kv * D, dtype=torch.bfloat16, device="cuda"); dv = torch.zeros_like(dk)
    ops.attn_bwd_seg(q, k, v, out, do, lse, seg_b, seg_e, pre_b, pre_e, dep_e, T, max_seg, n_q, n_kv, D, scale, dq, dk, dv)
    # dense reference with autograd
    leaf = qkv.float().cpu().clone().requires_grad_(True)
    ref_out = torch.zeros(T_pad, n_q * D)
    loss = 0.0
    for (b, e, pb, pe, _) in segs:
        rows = list(range(pb, pe)) + list(range(b, e))
        x = leaf[rows]
        L = len(rows)
        qq = x[:, :n_q * D].view(L, n_q, D).transpose(0, 1)
        kk = x[:, n_q * D:(n_q + n_kv) * D].view(L, n_kv, D).transpose(0, 1).repeat_interleave(g, 0)
        vv = x[:, (n_q + n_kv) * D:].view(L, n_kv, D).transpose(0, 1).repeat_interleave(g, 0)
        sc = (qq @ kk.transpose(1, 2)) * scale
        sc = sc.masked_fill(~torch.ones(L, L, dtype=torch.bool).tril(), float("-inf"))
        o = (sc.softmax(-1) @ vv).transpose(0, 1).reshape(L, n_q * D)
        own = o[pe - pb:]                                       # only the segment's own rows are outputs of this segment
        ref_out[b:e] = own.detach()
        loss = loss + (own * do.float().cpu()[b:e]).sum()
    loss.backward()
    gr = leaf.grad
    e_f = float((out.float().cpu()[:T] - ref_out[:T]).abs().max())
    measured(f"attn_seg_fwd_{T}", e_f)
    assert e_f < 1.02e-2, e_f                                         # measured <= 0.0078
    for name, got, want in (("dq", dq, gr[:, :n_q * D]), ("dk", dk, gr[:, n_q * D:(n_q + n_kv) * D]), ("dv", dv, gr[:, (n_q + n_kv) * D:])):
        err = float((got.float().cpu()[:T] - want[:T]).abs().max())
        measured(f"attn_seg_{name}_{T}_rel", err / float(want.abs().max()))
        assert err < 7.0e-3 * float(want.abs().max()), (name, err, float(want.abs().max()))      # measured <= 0.0054 (1.3x)


@pytest.mark.parametrize("M_,I,K", [(517, 1216, 256), (300, 200, 128), (1024, 4096, 512)])
def test_gemm_swiglu_fused_epilogue_bit_identical(ops, M_, I, K):
    """gate/up GEMM with the SwiGLU in the epilogue (training forward): m and the kept gate|up equal st_gemm_nt + st_swiglu_fwd."""
    rs = np.random.RandomState(M_ + I)
    a = bf(rs.standard_normal((M_, K))).cuda()
    w = bf(rs.standard_normal((2 * I, K)) * 0.1).cuda()
    gu_ref = ops.gemm_nt_variant(6, a, w) if M_ > 256 else ops.gemm_nt_variant(0, a, w)
    m_ref = ops.swiglu_fwd(gu_ref)
    gu, m = ops.gemm_swiglu(a, w, want_gu=True)
    assert torch.equal(gu, gu_ref) and torch.equal(m, m_ref)
    gu2, m2 = ops.gemm_swiglu(a, w, want_gu=False)
    assert gu2 is None and torch.equal(m2, m_ref)


@pytest.mark.parametrize("variant", [40, 23])
@pytest.mark.parametrize("shape", [(256, 256, 64), (304, 520, 192), (1000, 3592, 256), (4104, 4096, 1024), (520, 264, 64 * 37), (264, 8, 128), (2048, 1536, 320)])
def test_gemm_nn_tn_contraction_major_operands(ops, shape, variant):
    """st_gemm_nn (dX = dY W with W as stored) and st_gemm_tn (dW = dY^T X) against fp32 torch and, bit for bit, against the NT
    kernel fed with explicitly transposed operands (same tile, same MFMA order): ragged edges, 1..16 K-tiles, views with a row pitch
    larger than the width (qkv / gate|up slices), with and without the tail split, accumulate on and off."""
    M_, N, K = shape
    rs = np.random.RandomState(M_ + N + K)
    a = bf(rs.standard_normal((M_, K))).cuda()
    w_kn = bf(rs.standard_normal((K, N)) * (1 + np.arange(N)[None, :] / N)).cuda()          # B[k][n], asymmetric
    want = a.float() @ w_kn.float()
    scale = float(want.abs().max())
    ops.gemm_select(variant)                         # 40: the 4-wave hand-scheduled tile's contraction-major forms; 23: the 8-wave tile's
    for split in (True, False):
        ops.gemm_tail_split(split)
        try:
            out = ops.gemm_nn(a, w_kn)
            assert float((out.float() - want).abs().max()) < scale * 2 ** -7, split
            ref = ops.gemm_nt_variant(variant if M_ > 256 else 23, a, ops.transpose(w_kn))
            assert torch.equal(out, ref), split
            # dW form: A[k][m], B[k][n]; here "k" is the token index
            x_km = bf(rs.standard_normal((K, M_))).cuda()
            want_tn = x_km.float().t() @ w_kn.float()
            f = torch.full((M_, N), 0.5, dtype=torch.float32, device="cuda")
            ops.gemm_tn(x_km, w_kn, f, accumulate=True)
            assert float((f - want_tn - 0.5).abs().max()) < float(want_tn.abs().max()) * 1e-5 + 1e-3, split
            f2 = torch.full((M_, N), 0.5, dtype=torch.float32, device="cuda")
            ops.gemm_nt_variant(variant, ops.transpose(x_km), ops.transpose(w_kn), out_f32=f2, accumulate=True)
            assert torch.equal(f, f2), split
            ops.gemm_tn(x_km, w_kn, f, accumulate=False)
            assert float((f - want_tn).abs().max()) < float(want_tn.abs().max()) * 1e-5 + 1e-3
        finally:
            ops.gemm_tail_split(True)
            ops.gemm_select(40)
    # strided views: operands that are column slices of wider buffers
    wide_a = torch.zeros(M_, K + 128, dtype=torch.bfloat16, device="cuda"); wide_a[:, 64:64 + K] = a
    wide_w = torch.zeros(K, N + 256, dtype=torch.bfloat16, device="cuda"); wide_w[:, 128:128 + N] = w_kn
    ops.gemm_select(variant)
    try:
        assert torch.equal(ops.gemm_nn(wide_a[:, 64:64 + K], wide_w[:, 128:128 + N]), ops.gemm_nn(a, w_kn))
        wide_x = torch.zeros(K, M_ + 64, dtype=torch.bfloat16, device="cuda"); wide_x[:, 32:32 + M_] = x_km
        f3, f4 = torch.zeros(M_, N, dtype=torch.float32, device="cuda"), torch.zeros(M_, N, dtype=torch.float32, device="cuda")
        ops.gemm_tn(wide_x[:, 32:32 + M_], wide_w[:, 128:128 + N], f3)
        ops.gemm_tn(x_km, w_kn, f4)
        assert torch.equal(f3, f4)
    finally:
        ops.gemm_select(40)


# ------------------------------------------------------------------ the 4-wave tile with the hand-scheduled K loop (gemm_asm4.hip)
@pytest.mark.parametrize("shape", [(300, 520, 192), (1000, 777 * 8, 256), (513, 1016, 64), (4200, 4104, 1024), (256, 256, 3584)])
def test_gemm_asm4_every_epilogue_vs_fp32_and_8wave_tile(ops, shape):
    """Variant 40 (4 waves x 128x128, every K-loop instruction an asm statement, operands through buffer_load ... lds with the rows
    beyond M / N cut off by the buffer's num_records) against the fp32 product and against the 8-wave production tile (same MFMA
    shape and fp32 summation order per output element -> bit-identical bf16 outputs): ragged M / N, 1..16 K-tiles (every prologue /
    steady-state / tail path of the 2-slot pipeline), bias / residual / fp32 (accumulate) epilogues, strided views."""
    M_, N, K = shape
    rs = np.random.RandomState(M_ + N + K)
    a = bf(rs.standard_normal((M_, K))).cuda()
    b = bf(rs.standard_normal((N, K)) * (1 + np.arange(N)[:, None] / N)).cuda()
    bias, res = bf(rs.standard_normal(N)).cuda(), bf(rs.standard_normal((M_, N))).cuda()
    want = a.float() @ b.float().t()
    scale = float(want.abs().max())
    ops.gemm_tail_split(False)                                   # compare whole-tile launches (the K-sliced tail sums in another order)
    for kw in ({}, {"bias": bias}, {"residual": res}, {"bias": bias, "residual": res}):
        got = ops.gemm_nt_variant(40, a, b, **kw)
        ref = want + (bias.float() if "bias" in kw else 0.0) + (res.float() if "residual" in kw else 0.0)
        assert float((got.float() - ref).abs().max()) < scale * 2 ** -7, list(kw)
        assert torch.equal(got, ops.gemm_nt_variant(6, a, b, **kw)), list(kw)
    for accumulate in (True, False):
        f = torch.full((M_, N), 0.25, dtype=torch.float32, device="cuda"); f6 = f.clone()
        ops.gemm_nt_variant(40, a, b, out_f32=f, accumulate=accumulate)
        ops.gemm_nt_variant(6, a, b, out_f32=f6, accumulate=accumulate)
        assert float((f - want - (0.25 if accumulate else 0.0)).abs().max()) < scale * 1e-5 + 1e-4, accumulate
        assert torch.equal(f, f6), accumulate
    big = torch.zeros(M_, N + 64, dtype=torch.bfloat16, device="cuda")
    resb = torch.zeros(M_, N + 64, dtype=torch.bfloat16, device="cuda"); resb[:, 8:N + 8] = res
    wide_a = torch.zeros(M_, K + 128, dtype=torch.bfloat16, device="cuda"); wide_a[:, 64:64 + K] = a
    ops.gemm_nt_variant(40, wide_a[:, 64:64 + K], b, out=big[:, 8:N + 8], residual=resb[:, 8:N + 8])
    assert torch.equal(big[:, 8:N + 8], ops.gemm_nt_variant(6, a, b, residual=res)) and float(big[:, :8].abs().max()) == 0
    ops.gemm_tail_split(True)


@pytest.mark.parametrize("M_,I,K", [(517, 1216, 256), (300, 200, 128), (4100, 4104, 512), (33000, 1024, 192)])
def test_gemm_asm4_swiglu_epilogue_bit_identical(ops, M_, I, K):
    """st_gemm_swiglu on the 4-wave tile (st_gemm_select(40)): m and the kept gate|up equal st_gemm_nt + st_swiglu_fwd bit for bit."""
    rs = np.random.RandomState(M_ + I)
    a = bf(rs.standard_normal((M_, K))).cuda()
    w = bf(rs.standard_normal((2 * I, K)) * 0.1).cuda()
    gu_ref = ops.gemm_nt_variant(6, a, w)
    m_ref = ops.swiglu_fwd(gu_ref)
    try:
        ops.gemm_select(40)
        gu, m = ops.gemm_swiglu(a, w, want_gu=True)
        gu2, m2 = ops.gemm_swiglu(a, w, want_gu=False)
        ops.gemm_select(23)                                      # ... and the 8-wave tile's fused epilogue gives the same bits
        gu3, m3 = ops.gemm_swiglu(a, w, want_gu=True)
        assert torch.equal(gu3, gu_ref) and torch.equal(m3, m_ref)
    finally:
        ops.gemm_select(40)
    if -(-M_ // 256) * -(-I // 128) >= 128:                      # below that the launcher keeps the 8-wave tile
        pass
    assert torch.equal(gu, gu_ref) and torch.equal(m, m_ref)
    assert gu2 is None and torch.equal(m2, m_ref)


@pytest.mark.parametrize("case", ["mixed", "one_tile_items", "few_items"])
def test_decode_attention_persistent_kernel_is_bit_identical_to_one_item_per_workgroup_and_matches_fp32(ops, measured, case):
    """st_attn_fwd_ranges on decode-shaped item lists (the rollout's per-layer launch: shared-prompt partials with 56 query rows and a
    prefix key range in the prompt cache + per-sample partials with 7 query rows over their own cache chunks, many of them empty): the
    persistent kernel (attn_decode128_kernel — one workgroup per CU streaming the tiles of all its items through a 4-slot LDS ring with
    counted waits) must give BIT-IDENTICAL partial outputs and lse to the one-item-per-workgroup kernel, -inf lse for empty items, and the
    merged result must match dense fp32 attention over the union of every row's keys."""
    import math
    rs = np.random.RandomState({"mixed": 0, "one_tile_items": 1, "few_items": 2}[case])
    nkv, g, D = 4, 7, 128
    width = nkv * D
    if case == "mixed":
        n_prompts, n, CK, CKG, R = 9, 8, 576, 512, 1100
        plens = rs.randint(500, 1250, n_prompts)
        glens = rs.randint(0, R, n_prompts * n)
        glens[:5] = [0, 1, 63, 64, 65]
    elif case == "one_tile_items":
        n_prompts, n, CK, CKG, R = 40, 8, 576, 512, 600
        plens = rs.randint(20, 64, n_prompts)                    # every item a single (partial) tile: the producer runs one item ahead only
        glens = rs.randint(0, 64, n_prompts * n)
    else:
        n_prompts, n, CK, CKG, R = 2, 3, 256, 128, 300          # fewer work items than CUs
        plens = np.array([300, 77])
        glens = np.array([5, 140, 0, 299, 128, 1])
    B = n_prompts * n
    p_off = np.concatenate([[0], np.cumsum(plens)]).astype(np.int64)
    kp = bf(rs.standard_normal((int(p_off[-1]) + 64, width))).cuda(); vp = bf(rs.standard_normal((int(p_off[-1]) + 64, width))).cuda()
    kg = bf(rs.standard_normal((B * R, width))).cuda(); vg = bf(rs.standard_normal((B * R, width))).cuda()
    q = bf(rs.standard_normal((B, nkv * g * D)) * 0.5).cuda()
    C, Cg = int(-(-plens.max() // CK)), -(-R // CKG)
    NP, rows_all = C + Cg, B * g
    i32 = lambda a_: torch.from_numpy(np.ascontiguousarray(a_).astype(np.int32)).cuda()
    first = np.arange(n_prompts) * n
    kb1 = np.concatenate([np.minimum(p_off[:-1] + c * CK, p_off[1:]) for c in range(C)]); ke1 = np.concatenate([np.minimum(p_off[:-1] + (c + 1) * CK, p_off[1:]) for c in range(C)])
    qb1 = np.tile(first * g, C); qe1 = np.tile((first + n) * g, C)
    ob1 = np.concatenate([c * rows_all + first * g for c in range(C)])
    ar = np.arange(B)
    qb2 = np.tile(ar * g, Cg); qe2 = qb2 + g
    cidx = np.repeat(np.arange(Cg), B)
    kb2 = np.tile(ar * R, Cg) + cidx * CKG
    ke2 = np.maximum(np.minimum(kb2 + CKG, np.tile(ar * R + glens, Cg)), kb2)
    ob2 = (C + cidx) * rows_all + np.tile(ar * g, Cg)
    z1, z2 = np.zeros_like(kb1), np.zeros_like(kb2)
    args = dict(q_beg=i32(np.concatenate([qb1, qb2])), q_end=i32(np.concatenate([qe1, qe2])), k_beg=i32(np.concatenate([z1, kb2])),
                k_end=i32(np.concatenate([z1, ke2])), o_beg=i32(np.concatenate([ob1, ob2])), pre_beg=i32(np.concatenate([kb1, z2])),
                pre_end=i32(np.concatenate([ke1, z2])))
    scale = 1.0 / math.sqrt(D)
    res = {}
    try:
        for mode in (0, 1, 2):                                   # one workgroup per item / persistent, one per CU / persistent, two per CU
            ops.decode_attn_select(mode)
            parts = torch.full((NP * rows_all, width), float("nan"), dtype=torch.bfloat16, device="cuda")
            lse = torch.full((nkv, NP * rows_all), float("nan"), dtype=torch.float32, device="cuda")
            for _ in range(2):                                   # twice: the second launch starts with warm caches and stale LDS
                ops.attn_fwd_ranges(q, kg, vg, args["q_beg"], args["q_end"], args["k_beg"], args["k_end"], n * g, nkv, nkv, D, scale, parts, lse,
                                    o_beg=args["o_beg"], q_group=g, pre_beg=args["pre_beg"], pre_end=args["pre_end"], k_pre=kp, v_pre=vp)
            merged = ops.attn_merge(parts, lse, NP, nkv, D, out=torch.zeros(B, nkv * g * D, dtype=torch.bfloat16, device="cuda"), q_group=g)
            torch.cuda.synchronize()
            res[mode] = (parts.clone(), lse.clone(), merged.clone())
    finally:
        ops.decode_attn_select(0)                                 # the default
    (p0, l0, m0) = res[0]
    live = torch.isfinite(l0)                                     # (heads, slabs*rows): rows of items that had keys
    assert int(live.sum()) > 0 and int((l0 == float("-inf")).sum()) > 0
    rows_live = live.any(0)
    for mode in (1, 2):
        p1, l1, m1 = res[mode]
        assert torch.equal(l0, l1) or torch.equal(torch.nan_to_num(l0, nan=7.0), torch.nan_to_num(l1, nan=7.0)), mode
        assert torch.equal(p0[rows_live].view(torch.int16), p1[rows_live].view(torch.int16)), mode
        assert torch.equal(m0.view(torch.int16), m1.view(torch.int16)), mode
    # dense fp32 reference of the merged attention for a sample of rows
    worst = 0.0
    for b in rs.choice(B, size=min(B, 6), replace=False):
        pr = b // n
        K = torch.cat([kp[p_off[pr]:p_off[pr + 1]], kg[b * R:b * R + int(glens[b])]]).float()
        V = torch.cat([vp[p_off[pr]:p_off[pr + 1]], vg[b * R:b * R + int(glens[b])]]).float()
        for h in range(nkv):
            qq = q[b].float().view(nkv * g, D)[h * g:(h + 1) * g]
            Kh, Vh = K[:, h * D:(h + 1) * D], V[:, h * D:(h + 1) * D]
            want = torch.softmax(qq @ Kh.t() * scale, -1) @ Vh
            got = m1[b].float().view(nkv * g, D)[h * g:(h + 1) * g]
            worst = max(worst, float((got - want).abs().max()))
    measured(f"decode_attention_persistent_{case}_max_abs_vs_fp32", worst)
    assert worst < 0.02                                           # bf16 partials + bf16 output, |v| ~ 1 (measured ~0.008)
    # ---- round 6: the per-sample items on the one-wave-per-item kernel (st_attn_decode_rows: 32-key tiles in a private ring of 2..4 slots,
    # counted waits, no barriers), the prompt items on the workgroup kernel as before — the split the rollout's decode step launches.
    # Same quantities up to the fp32 rounding of an online softmax that advances in 32-key steps: lse to 1e-5, partials to one bf16 step,
    # -inf exactly where the workgroup kernel puts it, and the merged result against the dense fp32 attention as above.
    n1 = len(kb1)
    for slots in (2, 3, 4):
        parts = torch.full((NP * rows_all, width), float("nan"), dtype=torch.bfloat16, device="cuda")
        lse = torch.full((nkv, NP * rows_all), float("nan"), dtype=torch.float32, device="cuda")
        for _ in range(2):
            ops.attn_fwd_ranges(q, kg, vg, args["q_beg"][:n1], args["q_end"][:n1], args["k_beg"][:n1], args["k_end"][:n1], n * g, nkv, nkv, D, scale, parts, lse,
                                o_beg=args["o_beg"][:n1], q_group=g, pre_beg=args["pre_beg"][:n1], pre_end=args["pre_end"][:n1], k_pre=kp, v_pre=vp)
            ops.attn_decode_rows(q, kg, vg, args["q_beg"][n1:], args["q_end"][n1:], args["k_beg"][n1:], args["k_end"][n1:], g, nkv, D, scale, parts, lse,
                                 o_beg=args["o_beg"][n1:], q_group=g, slots=slots)
        mr = ops.attn_merge(parts, lse, NP, nkv, D, out=torch.zeros(B, nkv * g * D, dtype=torch.bfloat16, device="cuda"), q_group=g)
        torch.cuda.synchronize()
        assert torch.equal(torch.isfinite(lse), live) and torch.equal(lse == float("-inf"), l0 == float("-inf")), slots
        assert float((lse[live] - l0[live]).abs().max()) < 2e-5, slots
        dp = (parts[rows_live].float() - p0[rows_live].float()).abs()
        dp = torch.nan_to_num(dp, nan=0.0)                           # (heads without keys inside a live row keep their NaN fill in both)
        assert float(dp.max()) <= 2.0 ** -7 * max(1.0, float(torch.nan_to_num(p0[rows_live].float(), nan=0.0).abs().max())), (slots, float(dp.max()))
        assert float((mr.float() - m0.float()).abs().max()) <= 2.0 ** -6, slots
        worst_r = 0.0
        for b in range(min(B, 6)):
            pr = b // n
            K = torch.cat([kp[p_off[pr]:p_off[pr + 1]], kg[b * R:b * R + int(glens[b])]]).float()
            V = torch.cat([vp[p_off[pr]:p_off[pr + 1]], vg[b * R:b * R + int(glens[b])]]).float()
            for h in range(nkv):
                qq = q[b].float().view(nkv * g, D)[h * g:(h + 1) * g]
                want = torch.softmax(qq @ K[:, h * D:(h + 1) * D].t() * scale, -1) @ V[:, h * D:(h + 1) * D]
                worst_r = max(worst_r, float((mr[b].float().view(nkv * g, D)[h * g:(h + 1) * g] - want).abs().max()))
        measured(f"decode_attention_rows_kernel_{case}_slots{slots}_max_abs_vs_fp32", worst_r)
        assert worst_r < 0.02


@pytest.mark.parametrize("M_,I,K", [(257, 1000, 128), (300, 80, 64), (512, 81, 192), (384, 2000, 3584), (512, 18944, 3584), (448, 11008, 2048), (1, 160, 64), (64, 240, 128)])
def test_one_pass_512_row_swiglu_tile_is_bit_identical_to_the_two_round_tile(ops, M_, I, K):
    """gemm_swiglu512.hip (round 5; plan 512 of the decode gate/up + SwiGLU at 257..512 rows: all rows x 80 output columns per workgroup, K-steps
    of 32, ping-pong wave groups) against the 256x160 tile it replaces (plan 1): the same K summation order and the same epilogue
    roundings — bit-identical, incl. ragged row counts (waves without valid rows skip their work), a ragged last column tile and K = 64;
    and within the decode GEMM bound of the fp32 product."""
    from spatialthinker_amd.lib import lib
    torch.manual_seed(M_ + I)
    a = (torch.randn(M_, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(2 * I, K, device="cuda") * 0.05).bfloat16()
    o1 = torch.full((M_, I), 7.0, device="cuda", dtype=torch.bfloat16)
    o5 = torch.full((M_, I), -7.0, device="cuda", dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    for v, o in ((1, o1), (512, o5)):
        lib().st_gemm_swiglu_decode_variant(v, a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), o.data_ptr(), o.stride(0), M_, I, K, st)
    torch.cuda.synchronize()
    assert torch.equal(o1, o5)
    want = a.float() @ w.float().t()
    g_, u_ = want[:, :I].bfloat16().float(), want[:, I:].bfloat16().float()
    ref = (g_ * torch.sigmoid(g_)).bfloat16().float() * u_
    assert float((o5.float() - ref).abs().max() / ref.abs().max()) < 9.1e-3
    if M_ > 256:
        assert ops.swiglu_decode_plan(M_, I) == 512 and torch.equal(ops.gemm_swiglu_decode(a, w), o5)       # the decode entry takes it


@pytest.mark.parametrize("M_,I,K", [(17, 1000, 128), (64, 80, 64), (33, 81, 192), (64, 18944, 3584), (128, 18944, 3584), (100, 11008, 2048), (1, 160, 64)])
def test_one_tile_per_cu_small_swiglu_tiles_are_bit_identical(ops, M_, I, K):
    """gemm_tiles_swiglu_small.hip (round 5): the <= 128-row gate/up + SwiGLU on 160-weight-row tiles (237 workgroups for the 7B MLP, one per
    CU; plans 8 / 9) — same K order and epilogue roundings as the 256x160 tile (plan 1): bit-identical."""
    from spatialthinker_amd.lib import lib
    torch.manual_seed(M_ + I)
    a = (torch.randn(M_, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(2 * I, K, device="cuda") * 0.05).bfloat16()
    st = torch.cuda.current_stream().cuda_stream
    o1 = torch.full((M_, I), 7.0, device="cuda", dtype=torch.bfloat16)
    lib().st_gemm_swiglu_decode_variant(1, a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), o1.data_ptr(), o1.stride(0), M_, I, K, st)
    for v in ((8, 9) if M_ <= 64 else (9,)):
        o = torch.full((M_, I), -7.0, device="cuda", dtype=torch.bfloat16)
        lib().st_gemm_swiglu_decode_variant(v, a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), o.data_ptr(), o.stride(0), M_, I, K, st)
        torch.cuda.synchronize()
        assert torch.equal(o1, o), v
    assert torch.equal(ops.gemm_swiglu_decode(a, w), o1)                    # whatever plan the decode entry takes
