"""GPU: each HIP kernel (called through the C ABI) against a plain fp32 torch restatement / the oracle."""
import math
import os

import numpy as np
import pytest
import torch
from conftest import measured

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _rel(a, b):
    a, b = a.float(), b.float()
    return measured(((a - b).norm() / (b.norm() + 1e-12)).item())     # `< tol` records (measured, tol) in the parity report


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(DEV)


@pytest.fixture(scope="module")
def K():
    from tokensgen_amd import kernels
    return kernels


def _table(K, B, rows, D, ngroups, tokens, seed):
    from tokensgen_amd import kernels
    mod = _rand(B, rows, 3 * D * ngroups, seed=seed, scale=0.5)
    g = torch.Generator().manual_seed(seed + 1)
    tok_group = torch.randint(0, ngroups, (tokens,), generator=g, dtype=torch.uint8).to(DEV)
    r = [int(x) for x in torch.randint(0, rows, (ngroups,), generator=g)]
    sh = [3 * D * i for i in range(ngroups)]
    sc = [3 * D * i + D for i in range(ngroups)]
    ga = [3 * D * i + 2 * D for i in range(ngroups)]
    return kernels.GroupTable(mod, tok_group, r, sh, sc, ga), mod, tok_group, r, sh, sc, ga


def _gather(mod, tok_group, rows, cols, D):
    """[B, tokens, D] rows of the modulation table selected per token."""
    tg = tok_group.long()
    r = torch.tensor(rows, device=DEV)[tg]
    c = torch.tensor(cols, device=DEV)[tg]
    idx = c[:, None] + torch.arange(D, device=DEV)[None]
    return mod.float()[:, r[:, None], idx]


@pytest.mark.parametrize("M,N,K_,B", [(300, 256, 192, 2), (128, 128, 64, 1), (26, 384, 128, 1), (1031, 512, 320, 3),
                                      (2050, 768, 1024, 2), (1024, 256, 64, 1), (1500, 384, 128, 1)])
def test_gemm_bias(K, M, N, K_, B):
    from tokensgen_amd import lib as L
    a_full = _rand(B, M + 5, K_ + 8, seed=1)            # strided views: row stride and batch stride differ from M,K
    a = a_full[:, 2:2 + M, :K_]
    w = _rand(N, K_, seed=2, scale=0.1)
    bias = _rand(N, seed=3)
    out_full = torch.zeros(B, M, N + 16, dtype=torch.bfloat16, device=DEV)
    out = out_full[:, :, 8:8 + N]
    K.gemm(a, w, bias, out, L.EPI_BIAS)
    ref = a.float() @ w.float().T + bias.float()
    assert _rel(out, ref) < 4e-3
    assert (out_full[:, :, :8] == 0).all() and (out_full[:, :, 8 + N:] == 0).all()


def test_gemm_transpose_detecting(K):
    """A = I (asymmetric W): C must equal W^T exactly (catches row/col swaps in the MFMA C layout)."""
    from tokensgen_amd import lib as L
    n = 128
    a = torch.eye(n, dtype=torch.bfloat16, device=DEV)
    w = _rand(n, n, seed=5)
    out = torch.empty(n, n, dtype=torch.bfloat16, device=DEV)
    K.gemm(a, w, None, out, L.EPI_BIAS)
    assert torch.equal(out, w.T.contiguous())


def test_gemm_gelu_silu(K):
    from tokensgen_amd import lib as L
    a, w, bias = _rand(200, 256, seed=1), _rand(384, 256, seed=2, scale=0.1), _rand(384, seed=3)
    pre = (a.float() @ w.float().T + bias.float()).to(torch.bfloat16).float()
    out = torch.empty(200, 384, dtype=torch.bfloat16, device=DEV)
    K.gemm(a, w, bias, out, L.EPI_BIAS_GELU)
    assert _rel(out, torch.nn.functional.gelu(pre, approximate="tanh")) < 6e-3
    K.gemm(a, w, bias, out, L.EPI_BIAS_SILU)
    assert _rel(out, torch.nn.functional.silu(pre)) < 6e-3


@pytest.mark.parametrize("M", [333, 1111])     # 128^2 kernel / 256^2 ping-pong kernel
def test_gemm_gate_residual_inplace(K, M):
    from tokensgen_amd import lib as L
    B, N, K_ = 2, 256, 128
    a, w, bias = _rand(B, M, K_, seed=1), _rand(N, K_, seed=2, scale=0.1), _rand(N, seed=3)
    x = _rand(B, M, N, seed=4)
    tab, mod, tg, r, sh, sc, ga = _table(K, B, 7, N, 5, M, seed=9)
    ref = x.float() + _gather(mod, tg, r, ga, N) * (a.float() @ w.float().T + bias.float())
    K.gemm(a, w, bias, x, L.EPI_BIAS_GATE_RES, residual=x, gate=tab)
    assert _rel(x, ref) < 4e-3


@pytest.mark.parametrize("D", [128, 3072])
def test_adaln_modulate(K, D):
    B, T = 2, 77
    x = _rand(B, T + 3, D, seed=1, scale=2.0)[:, 1:1 + T]
    w, b = _rand(D, seed=2, scale=0.1) + 1, _rand(D, seed=3, scale=0.1)
    tab, mod, tg, r, sh, sc, ga = _table(K, B, 13, D, 4, T, seed=5)
    out = torch.empty(B, T, D, dtype=torch.bfloat16, device=DEV)
    K.adaln_modulate(x, out, w, b, 1e-5, tab)
    ln = torch.nn.functional.layer_norm(x.float(), (D,), w.float(), b.float(), 1e-5)
    ref = ln * (1 + _gather(mod, tg, r, sc, D)) + _gather(mod, tg, r, sh, D)
    assert _rel(out, ref) < 4e-3
    K.adaln_modulate(x, out, w, b, 1e-5, None)
    assert _rel(out, ln) < 4e-3


def test_qk_layernorm_rope(K):
    from oracle import dit_ref as O
    B, T, H = 2, 50, 3
    buf = _rand(B, T, 3 * H * 64, seed=1)
    orig = buf.clone()
    w, b = _rand(64, seed=2, scale=0.1) + 1, _rand(64, seed=3, scale=0.1)
    f32 = np.float32
    c0 = O.rope_3d(64, np.arange(2, dtype=f32), np.arange(3, dtype=f32), np.arange(4, dtype=f32))     # 24 tokens
    c1 = O.rope_3d(64, np.linspace(1000, 1003, 2, dtype=f32), np.arange(2, dtype=f32), np.arange(3, dtype=f32))  # 12
    c0d = tuple(t.to(DEV).contiguous() for t in c0)
    c1d = tuple(t.to(DEV).contiguous() for t in c1)
    q = buf[:, :, H * 64:2 * H * 64]                       # the "k" columns of a fused buffer
    K.qk_layernorm_rope(q, H, w, b, 1e-6, (8, c0d), (34, c1d))
    x = orig[:, :, H * 64:2 * H * 64].float().cpu().view(B, T, H, 64).transpose(1, 2)
    ln = torch.nn.functional.layer_norm(x, (64,), w.float().cpu(), b.float().cpu(), 1e-6).to(torch.bfloat16)
    ln[:, :, 8:32] = O.apply_rope(ln[:, :, 8:32], c0)
    ln[:, :, 34:46] = O.apply_rope(ln[:, :, 34:46], c1)
    got = q.cpu().view(B, T, H, 64).transpose(1, 2)
    assert _rel(got, ln) < 3e-3
    assert torch.equal(buf[:, :, :H * 64], orig[:, :, :H * 64]) and torch.equal(buf[:, :, 2 * H * 64:], orig[:, :, 2 * H * 64:])


def test_transpose_v(K):
    B, T, H = 2, 150, 3
    buf = _rand(B, T, 3 * H * 64, seed=1)
    v = buf[:, :, 2 * H * 64:]
    vt = torch.full((B, H, 64, 128), 7.0, dtype=torch.bfloat16, device=DEV)
    K.transpose_v(v, H, 30, 100, vt)
    ref = v[:, 30:130].reshape(B, 100, H, 64).permute(0, 2, 3, 1)
    assert torch.equal(vt[..., :100], ref) and (vt[..., 100:] == 0).all()


def _sdpa_ref(q, k, v, H, scale):
    B = q.shape[0]
    qh, kh, vh = (t.float().reshape(B, t.shape[1], H, 64).transpose(1, 2) for t in (q, k, v))
    p = torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(B, q.shape[1], H * 64)


@pytest.mark.parametrize("nq,nk1,nk2", [(200, 333, 100), (128, 64, 0), (77, 1000, 30), (513, 4097, 480)])
def test_attention_two_segments(K, nq, nk1, nk2):
    B, H = 2, 4
    qkv1 = _rand(B, max(nq, nk1), 3 * H * 64, seed=1)
    qkv2 = _rand(B, max(nq, nk2, 1), 3 * H * 64, seed=2)
    q1, k1, v1 = qkv1[:, :nq, :H * 64], qkv1[:, :nk1, H * 64:2 * H * 64], qkv1[:, :nk1, 2 * H * 64:]
    pad = lambda n: (n + 63) // 64 * 64
    vt1 = torch.empty(B, H, 64, pad(nk1), dtype=torch.bfloat16, device=DEV)
    K.transpose_v(v1, H, 0, nk1, vt1)
    out = torch.empty(B, nq, H * 64, dtype=torch.bfloat16, device=DEV)
    ref = _sdpa_ref(q1, k1, v1, H, 0.125)
    if nk2:
        q2, k2, v2 = qkv2[:, :nq, :H * 64], qkv2[:, :nk2, H * 64:2 * H * 64], qkv2[:, :nk2, 2 * H * 64:]
        vt2 = torch.empty(B, H, 64, pad(nk2), dtype=torch.bfloat16, device=DEV)
        K.transpose_v(v2, H, 0, nk2, vt2)
        K.attention(q1, k1, vt1, nk1, out, H, 0.125, q2, k2, vt2, nk2, 0.6)
        ref = ref + 0.6 * _sdpa_ref(q2, k2, v2, H, 0.125)
    else:
        K.attention(q1, k1, vt1, nk1, out, H, 0.125)
    assert _rel(out, ref) < 6e-3


@pytest.mark.parametrize("nq,nk1,nk2", [(200, 333, 100), (513, 1500, 480)])
def test_attention_prescaled_keys(K, nq, nk1, nk2):
    """K carries softmax_scale*log2(e) (tg_qk_layernorm_rope out_scale) and the kernel runs in the log2 domain directly."""
    B, H = 2, 4
    c = 0.125 * 1.4426950408889634
    qkv1, qkv2 = _rand(B, max(nq, nk1), 3 * H * 64, seed=1), _rand(B, max(nq, nk2), 3 * H * 64, seed=2)
    q1, k1, v1 = qkv1[:, :nq, :H * 64], qkv1[:, :nk1, H * 64:2 * H * 64], qkv1[:, :nk1, 2 * H * 64:]
    q2, k2, v2 = qkv2[:, :nq, :H * 64], qkv2[:, :nk2, H * 64:2 * H * 64], qkv2[:, :nk2, 2 * H * 64:]
    ref = _sdpa_ref(q1, k1, v1, H, 0.125) + 0.6 * _sdpa_ref(q2, k2, v2, H, 0.125)
    pad = lambda n: (n + 63) // 64 * 64
    vt1 = torch.empty(B, H, 64, pad(nk1), dtype=torch.bfloat16, device=DEV)
    vt2 = torch.empty(B, H, 64, pad(nk2), dtype=torch.bfloat16, device=DEV)
    K.transpose_v(v1, H, 0, nk1, vt1); K.transpose_v(v2, H, 0, nk2, vt2)
    k1s, k2s = (k1.float() * c).to(torch.bfloat16), (k2.float() * c).to(torch.bfloat16)
    out = torch.empty(B, nq, H * 64, dtype=torch.bfloat16, device=DEV)
    K.attention(q1, k1s, vt1, nk1, out, H, 0.125, q2, k2s, vt2, nk2, 0.6, k_prescaled=True)
    assert _rel(out, ref) < 8e-3


def test_qk_layernorm_out_scale(K):
    x = _rand(1, 40, 2 * 64, seed=1)
    w, b = _rand(64, seed=2, scale=0.1) + 1, _rand(64, seed=3, scale=0.1)
    a, bb = x.clone(), x.clone()
    K.qk_layernorm_rope(a, 2, w, b, 1e-6)
    K.qk_layernorm_rope(bb, 2, w, b, 1e-6, out_scale=0.18033688)
    assert _rel(bb, a.float() * 0.18033688) < 4e-3


def test_attention_forced_rescale(K):
    """Spike one key late in the sequence so the running max jumps mid-stream (online-softmax rescale path)."""
    B, H, nq, nk = 1, 1, 128, 512
    q, k, v = _rand(B, nq, 64, seed=1), _rand(B, nk, 64, seed=2), _rand(B, nk, 64, seed=3)
    k[:, 300] = q[:, 5] * 4.0
    k[:, 3] = q[:, 70] * 3.0
    vt = torch.empty(B, H, 64, nk, dtype=torch.bfloat16, device=DEV)
    K.transpose_v(v, H, 0, nk, vt)
    out = torch.empty(B, nq, 64, dtype=torch.bfloat16, device=DEV)
    K.attention(q, k, vt, nk, out, H, 0.125)
    ref = _sdpa_ref(q, k, v, H, 0.125)
    assert (out.float() - ref).abs().max().item() < 3e-2
    assert _rel(out, ref) < 6e-3


def test_timestep_sinusoid(K):
    from oracle import dit_ref as O
    t = torch.tensor([0, 1, 18, 500, 999, 37], dtype=torch.int64)
    out = torch.empty(6, 256, dtype=torch.bfloat16, device=DEV)
    K.timestep_sinusoid(t.to(DEV), 256, out)
    ref = O.timestep_sinusoid(t, 256).to(torch.bfloat16)
    assert (out.cpu().float() - ref.float()).abs().max().item() <= 2 ** -7


def test_patchify_roundtrip(K):
    lat = _rand(6, 16, 8, 12, seed=1)
    cols = torch.empty(6 * 4 * 6, 64, dtype=torch.bfloat16, device=DEV)
    K.patchify(lat, cols)
    ref = lat.reshape(6, 16, 4, 2, 6, 2).permute(0, 2, 4, 1, 3, 5).reshape(6 * 24, 64)
    assert torch.equal(cols, ref)
    back = torch.empty_like(lat)
    K.unpatchify(cols, back)
    assert torch.equal(back, lat)


def test_cfg_dpm_step_matches_oracle(K):
    from oracle import scheduler_ref as S
    _, ac = S.alphas_cumprod()
    ts = S.trailing_timesteps(52)
    frames, E = 5, 16 * 4 * 6
    mo, x, old, noise = _rand(2, frames, E, seed=1), _rand(frames, E, seed=2), _rand(frames, E, seed=3), _rand(frames, 2, E, seed=4)
    cases = [(999, 980, None, False), (980, 961, 999, True), (499, 480, 518, True), (18, -1, 37, True), (37, 18, 57, False)]
    from tokensgen_amd.scheduler import dpm_coef_row
    coef = torch.tensor([dpm_coef_row(ac, t, p, tb, ho) for (t, p, tb, ho) in cases], dtype=torch.float32, device=DEV)
    xo, x0o = torch.empty_like(x), torch.empty_like(x)
    K.cfg_dpm_step(mo, x, old, noise, coef, 6.0, xo, x0o)
    for f, (t, p, tb, ho) in enumerate(cases):
        v = S.cfg_combine(mo[:, f].cpu(), 6.0).float()
        it = iter([noise[f, 0].cpu().float(), noise[f, 1].cpu().float()])
        prev, x0 = S.dpm_step(ac, v, old[f].cpu().float() if ho else None, t, p, tb, x[f].cpu().float(), lambda: next(it))
        assert (x0o[f].cpu().float() - x0).abs().max().item() <= 2e-2 * x0.abs().max().item() + 1e-3, (t, p)
        assert (xo[f].cpu().float() - prev).abs().max().item() <= 2e-2 * prev.abs().max().item() + 1e-3, (t, p)


@pytest.mark.parametrize("pred", ["v_prediction", "epsilon", "sample"])
@pytest.mark.parametrize("branches,dynamic", [(2, False), (3, False), (2, True), (3, True)])
def test_cfg_dpm_step_ex_variants_match_oracle(K, branches, dynamic, pred):
    """tg_cfg_dpm_step_ex against the oracle's worker arithmetic run on bf16 tensors (torch's own promotion rules = the reference's): 2-way and
    3-way (`use_separate_guidance`) guidance, static Python-float scales (bf16 result) and the dynamic per-frame fp32 guidance tensor (promoted
    fp32 result -> the solver's f32 arithmetic), all three prediction types (cogvideo_sampling_mp_fifo.py:519-550,
    scheduling_dpm_cogvideox.py:424-463); pinned to reference runs by tests/test_oracle_golden.py::test_fifo_worker_guidance_variants_match_reference."""
    from oracle import scheduler_ref as S
    from tokensgen_amd.scheduler import dpm_coef_row
    _, ac = S.alphas_cumprod()
    frames, E = 5, 16 * 4 * 6
    mo, x, old, noise = _rand(branches, frames, E, seed=11), _rand(frames, E, seed=12), _rand(frames, E, seed=13), _rand(frames, 2, E, seed=14)
    cases = [(980, 961, 999, True), (499, 480, 518, True), (18, -1, 37, True), (37, 18, 57, False), (961, 941, 980, True)]
    if pred == "v_prediction":
        cases[0] = (999, 980, None, False)                    # alphas_cumprod = 0: only v-prediction is finite there
    coef = torch.tensor([dpm_coef_row(ac, t, p, tb, ho) for (t, p, tb, ho) in cases], dtype=torch.float32, device=DEV)
    tv = torch.tensor([c[0] for c in cases])
    g, gi = 6.0, 4.0
    gpf = None
    if dynamic:
        gt_, gti = S.dynamic_guidance(g, tv, 52), S.dynamic_guidance(gi, tv, 52)       # [1, F, 1, 1, 1] fp32
        gpf = torch.stack([gt_.flatten(), gti.flatten()], dim=1).float().contiguous().to(DEV)
    xo, x0o = torch.empty_like(x), torch.empty_like(x)
    K.cfg_dpm_step_ex(mo, x, old, noise, coef, g, xo, x0o, guidance_img=gi, guidance_per_frame=gpf, f32_math=dynamic, prediction_type=pred)
    moc = mo.cpu().view(branches, frames, 1, 1, E)                                       # [B, F, C, H, W]-shaped for the oracle's broadcasting
    gg, ggi = (gt_, gti) if dynamic else (g, gi)
    v = S.cfg_combine_separate(moc, gg, ggi) if branches == 3 else S.cfg_combine(moc, gg)
    assert v.dtype == (torch.float32 if dynamic else torch.bfloat16)
    for f, (t, p, tb, ho) in enumerate(cases):
        it = iter([noise[f, 0].cpu().view(1, 1, 1, 1, E), noise[f, 1].cpu().view(1, 1, 1, 1, E)])
        prev, x0 = S.dpm_step(ac, v[:, [f]], old[f].cpu().view(1, 1, 1, 1, E) if ho else None, t, p, tb, x[f].cpu().view(1, 1, 1, 1, E), lambda: next(it),
                              prediction_type=pred)
        prev, x0 = prev.to(torch.bfloat16).flatten(), x0.to(torch.bfloat16).flatten()     # the worker casts both back (:549-550)
        # static: the reference rounds every op of the step to bf16 (measured up to 6.1e-3 against the kernel's single rounding); dynamic: the
        # kernel follows the fp32 arithmetic exactly (measured 0)
        tol = 1e-3 if dynamic else 9e-3
        assert _rel(x0o[f].cpu(), x0) < tol, (t, p, "x0")
        assert _rel(xo[f].cpu(), prev) < tol, (t, p, "prev")


def test_attention_cases_again_on_the_pingpong_kernel():
    """tg_attention_fwd picks the 8-wave ping-pong kernel only for long query ranges (>= 1024 workgroups of 512 rows); the cases above
    are too small for it.  TG_ATTN_PP_MIN_WG is read once per process, so re-run them in a child process with the threshold at 1:
    every shape (ragged tiles, nq not a multiple of 512, second segment, prescaled keys) then goes through attn_fwd_pp_kernel."""
    import subprocess
    import sys
    if os.environ.get("TG_ATTN_PP_MIN_WG") == "1":
        pytest.skip("already inside the forced run")
    env = dict(os.environ, TG_ATTN_PP_MIN_WG="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-k", "attention and not pingpong", "-x"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_attention_multi_rider_matches_separate_launches(K):
    BF = torch.bfloat16
    """tg_attention_fwd_multi: the second problem's workgroups ride in the first problem's launch (ping-pong kernel) or run as a launch
    of their own (short query ranges) — either way bitwise what two tg_attention_fwd calls give."""
    torch.manual_seed(3)
    B, H, N1, NP = 2, 2, 700, 70
    D, N = H * 64, N1 + NP
    qkv = (torch.randn(B, N1, 3 * D, device=DEV) * 0.5).to(BF)
    qkvv = (torch.randn(B, N, 3 * D, device=DEV) * 0.5).to(BF)
    pad = lambda n: (n + 63) // 64 * 64
    vt1 = torch.zeros(B, H, 64, pad(N1), dtype=BF, device=DEV); K.transpose_v(qkv[:, :, 2 * D:], H, 0, N1, vt1)
    vt2 = torch.zeros(B, H, 64, pad(NP), dtype=BF, device=DEV); K.transpose_v(qkvv[:, :, 2 * D:], H, N1, NP, vt2)
    vt3 = torch.zeros(B, H, 64, pad(N), dtype=BF, device=DEV); K.transpose_v(qkvv[:, :, 2 * D:], H, 0, N, vt3)
    for pre in (False, True):
        a = torch.zeros(B, N, D, dtype=BF, device=DEV); b = torch.zeros(B, N, D, dtype=BF, device=DEV)
        K.attention(qkv[:, :, :D], qkv[:, :, D:2 * D], vt1, N1, a[:, :N1], H, 0.125, qkvv[:, :N1, :D], qkvv[:, N1:, D:2 * D], vt2, NP, 0.6, k_prescaled=pre)
        K.attention(qkvv[:, N1:, :D], qkvv[:, :, D:2 * D], vt3, N, a[:, N1:], H, 0.125, k_prescaled=pre)
        K.attention_multi(dict(q1=qkv[:, :, :D], k1=qkv[:, :, D:2 * D], vt1=vt1, nk1=N1, out=b[:, :N1], q2=qkvv[:, :N1, :D],
                               k2=qkvv[:, N1:, D:2 * D], vt2=vt2, nk2=NP, seg2_scale=0.6),
                          dict(q1=qkvv[:, N1:, :D], k1=qkvv[:, :, D:2 * D], vt1=vt3, nk1=N, out=b[:, N1:]), H, 0.125, k_prescaled=pre)
        assert torch.isfinite(b).all() and torch.equal(a, b)


@pytest.mark.parametrize("shift", [-400.0, -60.0, 90.0])
def test_attention_prescaled_extreme_score_offsets(K, shift):
    """Softmax is shift invariant: rows whose scores all sit far below (or above) zero in the log2 domain must come out like the
    un-shifted rows.  -400 would underflow every weight if the running reference stayed at its initial 0; +90 overflows exp2 without
    the max subtraction; a late spike exercises the upward rescale on top."""
    B, H, nq, nk = 1, 8, 96, 640
    torch.manual_seed(11)
    q = torch.randn(B, nq, H * 64, device=DEV).to(torch.bfloat16)
    k = (torch.randn(B, nk, H * 64, device=DEV) * 0.2).to(torch.bfloat16)
    v = torch.randn(B, nk, H * 64, device=DEV).to(torch.bfloat16)
    # add a component along a direction u that q carries with unit weight: every score of the row moves by `shift`
    u = torch.zeros(64, device=DEV); u[7] = 1.0
    qh = q.view(B, nq, H, 64).float(); qh[..., 7] = 1.0
    kh = k.view(B, nk, H, 64).float(); kh[..., 7] = shift
    kh[:, 500, :, :] += qh[:, 3, :, :] * 0.5                      # a late spike for one query
    q, k = qh.view(B, nq, -1).to(torch.bfloat16), kh.view(B, nk, -1).to(torch.bfloat16)
    ref = _sdpa_ref(q, k, v, H, 1.0 / 1.4426950408889634)         # scores are already in log2 units: softmax over s*ln2
    vt = torch.empty(B, H, 64, nk, dtype=torch.bfloat16, device=DEV)
    K.transpose_v(v, H, 0, nk, vt)
    out = torch.empty(B, nq, H * 64, dtype=torch.bfloat16, device=DEV)
    K.attention(q, k, vt, nk, out, H, 0.125, k_prescaled=True)
    assert torch.isfinite(out).all()
    assert _rel(out, ref) < 8e-3


def test_attention_repeatable_bitwise(K):
    """Race screen: the kernels stage K/V with LDS-DMA ordered only by counted waits + barriers; a read that runs ahead of its data
    shows up as run-to-run differences.  Same inputs, 6 launches, bitwise identical outputs (also run through the ping-pong kernel
    by the forced child process)."""
    B, H, nq, nk1, nk2 = 2, 8, 1100, 2100, 200
    qkv1, qkv2 = _rand(B, max(nq, nk1), 3 * H * 64, seed=5), _rand(B, max(nq, nk2), 3 * H * 64, seed=6)
    q1, k1, v1 = qkv1[:, :nq, :H * 64], qkv1[:, :nk1, H * 64:2 * H * 64], qkv1[:, :nk1, 2 * H * 64:]
    q2, k2, v2 = qkv2[:, :nq, :H * 64], qkv2[:, :nk2, H * 64:2 * H * 64], qkv2[:, :nk2, 2 * H * 64:]
    pad = lambda n: (n + 63) // 64 * 64
    vt1 = torch.zeros(B, H, 64, pad(nk1), dtype=torch.bfloat16, device=DEV)
    vt2 = torch.zeros(B, H, 64, pad(nk2), dtype=torch.bfloat16, device=DEV)
    K.transpose_v(v1, H, 0, nk1, vt1); K.transpose_v(v2, H, 0, nk2, vt2)
    outs = []
    for _ in range(6):
        out = torch.empty(B, nq, H * 64, dtype=torch.bfloat16, device=DEV)
        K.attention(q1, k1, vt1, nk1, out, H, 0.125, q2, k2, vt2, nk2, 0.6)
        outs.append(out)
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    ref = _sdpa_ref(q1, k1, v1, H, 0.125) + 0.6 * _sdpa_ref(q2, k2, v2, H, 0.125)
    assert _rel(outs[0], ref) < 8e-3


def test_gemm256_repeatable_bitwise(K):
    """Race screen for the 256x256 ping-pong GEMM (4-deep LDS-DMA ring, counted vmcnt): several tiles per CU, ragged last m-tile,
    K long enough for the steady-state loop; 5 launches, bitwise identical, and right against fp32 torch."""
    from tokensgen_amd import lib as L
    B, M, N, Kd = 2, 2300, 512, 1536
    a, w, bias = _rand(B, M, Kd, seed=21), _rand(N, Kd, seed=22, scale=0.05), _rand(N, seed=23)
    outs = []
    for _ in range(5):
        out = torch.empty(B, M, N, dtype=torch.bfloat16, device=DEV)
        K.gemm(a, w, bias, out, L.EPI_BIAS_GELU)
        outs.append(out)
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    ref = torch.nn.functional.gelu((a.float() @ w.float().t() + bias.float()).to(torch.bfloat16).float(), approximate="tanh")
    assert _rel(outs[0], ref) < 6e-3


def test_qk_layernorm_rope_pair_equals_two_single_launches(K):
    """tg_qk_layernorm_rope_pair (q and k columns of the fused buffer in one launch, tables read once) is bitwise the two single calls."""
    from oracle import dit_ref as O
    B, T, H = 2, 50, 3
    buf = _rand(B, T, 3 * H * 64, seed=1)
    wq, bq = _rand(64, seed=2, scale=0.1) + 1, _rand(64, seed=3, scale=0.1)
    wk, bk = _rand(64, seed=4, scale=0.1) + 1, _rand(64, seed=5, scale=0.1)
    f32 = np.float32
    c0 = tuple(t.to(DEV).contiguous() for t in O.rope_3d(64, np.arange(2, dtype=f32), np.arange(3, dtype=f32), np.arange(4, dtype=f32)))
    c1 = tuple(t.to(DEV).contiguous() for t in O.rope_3d(64, np.linspace(1000, 1003, 2, dtype=f32), np.arange(2, dtype=f32), np.arange(3, dtype=f32)))
    a, b = buf.clone(), buf.clone()
    K.qk_layernorm_rope(a[:, :, :H * 64], H, wq, bq, 1e-6, (8, c0), (34, c1))
    K.qk_layernorm_rope(a[:, :, H * 64:2 * H * 64], H, wk, bk, 1e-6, (8, c0), (34, c1), out_scale=0.18033688)
    K.qk_layernorm_rope_pair(b[:, :, :H * 64], b[:, :, H * 64:2 * H * 64], H, wq, bq, wk, bk, 1e-6, (8, c0), (34, c1), k_scale=0.18033688)
    assert torch.equal(a, b) and not torch.equal(a[:, :, :2 * H * 64], buf[:, :, :2 * H * 64])
    # OUT OF PLACE (tg_qk_layernorm_rope_pair_out, the training forward: the pre-norm projection stays for the backward): the same rows into a buffer of another row stride,
    # the source untouched — without and with the key-norm bound of the constant-shift attention (same numbers as the in-place launch's)
    src = buf.clone()
    dst = torch.full((B, T, 2 * H * 64 + 64), 7.0, dtype=torch.bfloat16, device=DEV)
    K.qk_layernorm_rope_pair(src[:, :, :H * 64], src[:, :, H * 64:2 * H * 64], H, wq, bq, wk, bk, 1e-6, (8, c0), (34, c1), k_scale=0.18033688,
                             out=(dst[:, :, :H * 64], dst[:, :, H * 64:2 * H * 64]))
    assert torch.equal(src, buf) and torch.equal(dst[:, :, :2 * H * 64], a[:, :, :2 * H * 64]) and (dst[:, :, 2 * H * 64:] == 7.0).all()
    km_in, km_out = (torch.zeros(B, H, dtype=torch.float32, device=DEV) for _ in range(2))
    kws = K.kmax_workspace(T, H, B, DEV)
    c = buf.clone()
    K.qk_layernorm_rope_pair(c[:, :, :H * 64], c[:, :, H * 64:2 * H * 64], H, wq, bq, wk, bk, 1e-6, (8, c0), (34, c1), k_scale=0.18033688, kmax=km_in, kmax_ws=kws)
    dst.fill_(7.0)
    K.qk_layernorm_rope_pair(src[:, :, :H * 64], src[:, :, H * 64:2 * H * 64], H, wq, bq, wk, bk, 1e-6, (8, c0), (34, c1), k_scale=0.18033688, kmax=km_out, kmax_ws=kws,
                             out=(dst[:, :, :H * 64], dst[:, :, H * 64:2 * H * 64]))
    assert torch.equal(src, buf) and torch.equal(dst[:, :, :2 * H * 64], c[:, :, :2 * H * 64]) and torch.equal(km_in, km_out) and float(km_out.min()) > 0


def test_rope_tables_on_device_match_host():
    """tg_rope_table_3d against the host tables (which are bit-exact to the reference's, tests/test_oracle_golden.py): FIFO-style grids
    incl. the condensed-token positions around 1000 and the T2To 52/6/6 split; only the device cosf/sinf differ."""
    from tokensgen_amd import rope as R
    f32 = np.float32
    cases = [(64, np.arange(13, dtype=f32) + 39, np.arange(30, dtype=f32), np.arange(45, dtype=f32), None),
             (64, np.linspace(1000, 1016.25, 5, dtype=f32), np.linspace(0, 30, 8, endpoint=False, dtype=f32), np.linspace(0, 45, 12, endpoint=False, dtype=f32), None),
             (64, np.arange(96, dtype=f32), np.arange(8, dtype=f32), np.arange(12, dtype=f32), (52, 6, 6))]
    for hd, gt, gh, gw, dims in cases:
        kw = {} if dims is None else dict(dim_t=dims[0], dim_h=dims[1], dim_w=dims[2])
        hc, hs = R.rope_3d(hd, gt, gh, gw, **kw)
        dc, ds = R.rope_3d(hd, gt, gh, gw, device=DEV, **kw)
        assert dc.shape == hc.shape and dc.is_cuda
        assert (dc.cpu() - hc).abs().max().item() < 2e-6 and (ds.cpu() - hs).abs().max().item() < 2e-6


def test_gemm_pair_equals_two_launches(K):
    """tg_gemm_bf16_pair: the second problem's tiles ride in the first one's persistent launch — bitwise the two separate launches, for
    row counts that leave ragged last m-tiles and with the second problem's rows a superset of the first's (the To2V QKV case)."""
    from tokensgen_amd import lib as L
    B, M1, M2, N, Kd = 2, 1100, 1230, 768, 512
    x = _rand(B, M2, Kd, seed=31)
    w1, w2, b1, b2 = _rand(N, Kd, seed=32, scale=0.05), _rand(N, Kd, seed=33, scale=0.05), _rand(N, seed=34), _rand(N, seed=35)
    for epi in (L.EPI_BIAS, L.EPI_BIAS_GELU):
        a1, a2 = torch.zeros(B, M1, N, dtype=torch.bfloat16, device=DEV), torch.zeros(B, M2, N, dtype=torch.bfloat16, device=DEV)
        c1, c2 = torch.zeros_like(a1), torch.zeros_like(a2)
        K.gemm(x[:, :M1], w1, b1, a1, epi)
        K.gemm(x, w2, b2, a2, epi)
        K.gemm_pair(x[:, :M1], w1, b1, c1, x, w2, b2, c2, epi)
        assert torch.equal(a1, c1) and torch.equal(a2, c2) and a1.abs().sum() > 0


@pytest.mark.parametrize("two", [False, True])
def test_gemm_qkv_vt_equals_gemm_plus_transpose(K, two):
    """tg_gemm_bf16_qkv: the V third of a QKV projection written transposed by the GEMM epilogue (operands exchanged for those tiles)
    is bitwise tg_gemm_bf16 + tg_transpose_v, including the zero padding of the key axis out to a multiple of 64; the q/k columns
    are bitwise the plain GEMM's; the V columns of the C buffer are left untouched.  Ragged last m-tiles, two batch items, one and
    two problems per launch, several tiles per workgroup."""
    from tokensgen_amd import lib as L
    B, H, Kd = 2, 12, 512
    D = H * 64
    N = 3 * D
    M1, M2 = 1100, 1230
    pad = lambda n: (n + 63) // 64 * 64
    x = _rand(B, M2, Kd, seed=41)
    w1, w2, b1, b2 = _rand(N, Kd, seed=42, scale=0.05), _rand(N, Kd, seed=43, scale=0.05), _rand(N, seed=44), _rand(N, seed=45)
    ref1, ref2 = torch.zeros(B, M1, N, dtype=torch.bfloat16, device=DEV), torch.zeros(B, M2, N, dtype=torch.bfloat16, device=DEV)
    K.gemm(x[:, :M1], w1, b1, ref1, L.EPI_BIAS)
    K.gemm(x, w2, b2, ref2, L.EPI_BIAS)
    rvt1 = torch.full((B, H, 64, pad(M1)), 7.0, dtype=torch.bfloat16, device=DEV)
    rvt2 = torch.full((B, H, 64, pad(M2)), 7.0, dtype=torch.bfloat16, device=DEV)
    K.transpose_v(ref1[:, :, 2 * D:], H, 0, M1, rvt1)
    K.transpose_v(ref2[:, :, 2 * D:], H, 0, M2, rvt2)
    c1, c2 = torch.full_like(ref1, 3.0), torch.full_like(ref2, 3.0)
    vt1, vt2 = torch.full_like(rvt1, 5.0), torch.full_like(rvt2, 5.0)
    if two:
        K.gemm_qkv(x[:, :M1], w1, b1, c1, vt1, x, w2, b2, c2, vt2)
    else:
        K.gemm_qkv(x[:, :M1], w1, b1, c1, vt1)
        K.gemm_qkv(x, w2, b2, c2, vt2)
    for c, ref, vt, rvt, M in ((c1, ref1, vt1, rvt1, M1), (c2, ref2, vt2, rvt2, M2)):
        assert torch.equal(c[:, :, :2 * D], ref[:, :, :2 * D])
        assert (c[:, :, 2 * D:] == 3.0).all()                      # V columns of C: never written
        assert torch.equal(vt, rvt) and (vt[:, :, :, M:] == 0).all() and vt[:, :, :, :M].abs().sum() > 0


def test_gemm_4wave_kernel_matches_8wave_kernel(tmp_path):
    """The 256x256x64 4-wave kernel (default for M >= 1024, K >= 256; 16x16x32 MFMA) against the 8-wave 256x256x32 kernel
    (TG_GEMM_W4=0; 32x32x16 MFMA) and fp32 torch — bias / GELU / SiLU / gated-residual epilogues, M edges, strided views, several
    tiles per workgroup (the DMA cursor crossing tile boundaries), K = 12288.  The knob is read once per process: tools/w4_check.py
    runs once per mode, checks each mode against fp32 torch and the second run against the first run's outputs (equal up to the
    bf16 rounding of differently chunked fp32 sums)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("0", "1"):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "w4_check.py"), str(tmp_path)], env=dict(os.environ, TG_GEMM_W4=mode),
                           capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "vs other mode" in r.stdout


def test_gemm_4wave_random_shapes(K):
    """Seeded sweep of the 4-wave GEMM's corner parameters against fp32 torch: odd and even numbers of K stages (the two LDS buffers
    alternate across tile boundaries), M one row short of / past a tile edge, more tiles than CUs (several tiles per workgroup, the DMA
    cursor crossing tiles and batch items), strided A and C views, every epilogue."""
    from tokensgen_amd import lib as L
    rng = np.random.RandomState(7)
    cases = [(1024, 256, 256, 1), (1025, 256, 320, 1), (1279, 512, 448, 2), (1281, 256, 832, 3), (4095, 768, 256, 1), (9000, 2560, 320, 1),
             (2304, 1024, 1088, 2)]
    rng = np.random.RandomState(7)
    for i, (M, N, Kd, B) in enumerate(cases):
        epi = [L.EPI_BIAS, L.EPI_BIAS_GELU, L.EPI_BIAS_SILU][i % 3]
        a_full = _rand(B, M + 3, Kd + 16, seed=100 + i)
        a = a_full[:, 1:1 + M, 8:8 + Kd]
        w, bias = _rand(N, Kd, seed=200 + i, scale=0.05), _rand(N, seed=300 + i)
        use_bias = rng.rand() < 0.8
        out_full = torch.full((B, M + 2, N + 8), 9.0, dtype=torch.bfloat16, device=DEV)
        out = out_full[:, 1:1 + M, :N]
        K.gemm(a, w, bias if use_bias else None, out, epi)
        pre = (a.float() @ w.float().T + (bias.float() if use_bias else 0.0)).to(torch.bfloat16).float()
        ref = pre if epi == L.EPI_BIAS else torch.nn.functional.gelu(pre, approximate="tanh") if epi == L.EPI_BIAS_GELU else torch.nn.functional.silu(pre)
        assert _rel(out, ref) < 6e-3, (M, N, Kd, B, epi)
        assert (out_full[:, 0] == 9.0).all() and (out_full[:, M + 1] == 9.0).all() and (out_full[:, :, N:] == 9.0).all(), (M, N, Kd, B)


def test_attention_ragged_tile_split_at_launch_scale(K):
    """At launch scale (>= 1024 ping-pong workgroups) a single-problem call whose ragged last query tile would cost a whole extra round
    of workgroups hands that tile to the 4-wave kernel behind the ping-pong launch (attention.hip, attention_launch): 48 heads x 2, nq =
    13 x 512 + 150 (1248 -> 5 rounds instead of 1344 -> 6).  Both parts against fp32 torch on a few heads: rows of full tiles, the
    ragged rows, two key segments."""
    B, H, nq, nk2 = 2, 48, 13 * 512 + 150, 200
    D = H * 64
    qkv = _rand(B, nq, 3 * D, seed=61, scale=0.6)
    qkv2 = _rand(B, nq, 3 * D, seed=62, scale=0.6)
    pad = lambda n: (n + 63) // 64 * 64
    vt1 = torch.empty(B, H, 64, pad(nq), dtype=torch.bfloat16, device=DEV)
    vt2 = torch.empty(B, H, 64, pad(nk2), dtype=torch.bfloat16, device=DEV)
    K.transpose_v(qkv[:, :, 2 * D:], H, 0, nq, vt1)
    K.transpose_v(qkv2[:, :, 2 * D:], H, 0, nk2, vt2)
    out = torch.zeros(B, nq, D, dtype=torch.bfloat16, device=DEV)
    K.attention(qkv[:, :, :D], qkv[:, :, D:2 * D], vt1, nq, out, H, 0.125, qkv2[:, :, :D], qkv2[:, :nk2, D:2 * D], vt2, nk2, 0.6)
    rows = torch.cat([torch.arange(0, 40), torch.arange(6000, 6040), torch.arange(13 * 512 - 20, nq)]).to(DEV)
    for h in (0, 17, 47):
        sl = slice(h * 64, h * 64 + 64)
        q1, k1, v1 = qkv[:, rows, sl].float(), qkv[:, :, D + h * 64:D + h * 64 + 64].float(), qkv[:, :, 2 * D + h * 64:2 * D + h * 64 + 64].float()
        q2, k2, v2 = qkv2[:, rows, sl].float(), qkv2[:, :nk2, D + h * 64:D + h * 64 + 64].float(), qkv2[:, :nk2, 2 * D + h * 64:2 * D + h * 64 + 64].float()
        ref = torch.softmax(q1 @ k1.transpose(1, 2) * 0.125, -1) @ v1 + 0.6 * (torch.softmax(q2 @ k2.transpose(1, 2) * 0.125, -1) @ v2)
        assert _rel(out[:, rows, sl], ref) < 8e-3, h


@pytest.mark.parametrize("gain", [1.0, 3.0, 8.0])
@pytest.mark.parametrize("B,H,N1,NP", [(2, 2, 700, 70), (2, 48, 7 * 512 + 150, 480), (2, 48, 2 * 512 + 150, 480)])
def test_attention_constant_shift_softmax(K, B, H, N1, NP, gain):
    """tg_attn_segment.k_norm2_max + retry workspace: the 512-row kernel subtracts a per-row CONSTANT c = max(0, ||q|| max||k|| - 64) instead of
    a running row maximum, for ANY LayerNorm weights (the round-2 bound came from the weights and silently fell back for
    gains above ~1.7).  q / k come from the real K-norm kernel (tg_qk_layernorm_rope_pair_kmax) with gains `gain` * (1 +- 10 %):
      gain 1: B ~ 12, c = 0 — safe by construction;  gain 3: B ~ 105, c ~ 40 — the fast path still engages, verified per row, nothing retried;
      gain 8: B ~ 740 — every row's scores sit hundreds of units below c, the verification fails everywhere and the retry launch recomputes
      every workgroup with the running maximum: bitwise the running-max result.
    Main problem with two key segments + the rider.  The second shape is at launch scale (>= 1024 workgroups of 512 rows: the kernel really
    runs); the first and the third reach it in the forced child run of test_attention_cases_again_on_the_pingpong_kernel.  The third shape
    (288 + 96 = 256 + 128 workgroups) has 32 TWO-SEGMENT main workgroups in its split tail (see the end of the test)."""
    BF = torch.bfloat16
    D, N = H * 64, N1 + NP
    kscale = 0.125 * 1.4426950408889634
    g = torch.Generator().manual_seed(5)
    qkv = _rand(B, N1, 3 * D, seed=71)
    qkvv = _rand(B, N, 3 * D, seed=72)
    w = lambda: ((1.0 + 0.1 * torch.randn(64, generator=g)) * gain).to(BF).to(DEV)
    bb = lambda: (0.1 * torch.randn(64, generator=g)).to(BF).to(DEV)
    km1 = torch.zeros(B, H, dtype=torch.float32, device=DEV)
    km2 = torch.zeros(B, H, dtype=torch.float32, device=DEV)
    kws = K.kmax_workspace(N, H, B, DEV)
    K.qk_layernorm_rope_pair(qkv[:, :, :D], qkv[:, :, D:2 * D], H, w(), bb(), w(), bb(), 1e-6, k_scale=kscale, kmax=km1, kmax_ws=kws)
    K.qk_layernorm_rope_pair(qkvv[:, :, :D], qkvv[:, :, D:2 * D], H, w(), bb(), w(), bb(), 1e-6, k_scale=kscale, kmax=km2, kmax_ws=kws)
    n2 = lambda t: t.float().reshape(t.shape[0], t.shape[1], H, 64).pow(2).sum(-1).amax(dim=1)       # [B, H] max_t ||row||^2 of the stored rows
    assert torch.allclose(km1, n2(qkv[:, :, D:2 * D]), rtol=1e-5) and torch.allclose(km2, n2(qkvv[:, :, D:2 * D]), rtol=1e-5)
    pad = lambda n: (n + 63) // 64 * 64
    vt1 = torch.zeros(B, H, 64, pad(N1), dtype=BF, device=DEV); K.transpose_v(qkv[:, :, 2 * D:], H, 0, N1, vt1)
    vt2 = torch.zeros(B, H, 64, pad(NP), dtype=BF, device=DEV); K.transpose_v(qkvv[:, :, 2 * D:], H, N1, NP, vt2)
    vt3 = torch.zeros(B, H, 64, pad(N), dtype=BF, device=DEV); K.transpose_v(qkvv[:, :, 2 * D:], H, 0, N, vt3)
    retry = K.AttnRetry(N1, NP, H, B, DEV)

    def run(fast, split=None):
        a = torch.zeros(B, N, D, dtype=BF, device=DEV)
        K.attention_multi(dict(q1=qkv[:, :, :D], k1=qkv[:, :, D:2 * D], vt1=vt1, nk1=N1, out=a[:, :N1], q2=qkvv[:, :N1, :D], k2=qkvv[:, N1:, D:2 * D],
                               vt2=vt2, nk2=NP, seg2_scale=0.6, kmax1=km1 if fast else None, kmax2=km2 if fast else None),
                          dict(q1=qkvv[:, N1:, :D], k1=qkvv[:, :, D:2 * D], vt1=vt3, nk1=N, out=a[:, N1:], kmax1=km2 if fast else None), H, 0.125,
                          k_prescaled=True, retry=retry if fast else None, split=split)
        return a
    exact, fast = run(False), run(True)
    wgs = retry.ints - 1
    on_pp = (N1 + 511) // 512 * H * B >= int(os.environ.get("TG_ATTN_PP_MIN_WG", "1024")) and os.environ.get("TG_ATTN_FIXEDM", "1") != "0"
    assert int(retry.buf[1:].abs().sum().item()) == 0          # every raised flag was consumed by the retry launch
    if not on_pp:
        assert retry.count() == 0 and torch.equal(fast, exact)   # short query ranges: the 4-wave running-max kernel, bounds ignored
    elif gain < 5:
        assert retry.count() == 0                                # the constant-shift pass stood everywhere
        assert not torch.equal(fast, exact)                      # ... and it IS a different kernel
        assert _rel(fast, exact) < 4e-3
    else:
        assert retry.count() == wgs                              # verification failed everywhere -> recomputed with the running maximum
        assert torch.equal(fast, exact)
    rows = torch.cat([torch.arange(0, 40), torch.arange(N1 - 40, N1)]).to(DEV)
    for h in sorted({0, H // 2, H - 1}):
        sl, ks, vs = slice(h * 64, h * 64 + 64), slice(D + h * 64, D + h * 64 + 64), slice(2 * D + h * 64, 2 * D + h * 64 + 64)
        sm = lambda q, k, v: torch.softmax(q.float() @ k.float().transpose(1, 2) * math.log(2.0), -1) @ v.float()
        ref = sm(qkv[:, rows, sl], qkv[:, :, ks], qkv[:, :, vs]) + 0.6 * sm(qkvv[:, rows, sl], qkvv[:, N1:, ks], qkvv[:, N1:, vs])
        assert _rel(fast[:, rows, sl], ref) < 8e-3, h
        refv = sm(qkvv[:, N1:, sl], qkvv[:, :, ks], qkvv[:, :, vs])
        assert _rel(fast[:, N1:, sl], refv) < 8e-3, h
    # key-axis split of the launch's half-round tail (tg_attn_workspace.split): at the launch-scale shape 768 + 96 = 864 = 3 x 256 + 96
    # workgroups -> the last 96 workgroups of the MAIN problem (two key segments each) run as 192 half-length workgroups + the combine launch.
    # Same function: equal to the unsplit launch up to the rounding of the fp32 partial sums; the riders are never split (bitwise unchanged).
    if retry.split is not None and on_pp:
        n0 = retry.count()
        for fastpath, base in ((True, fast), (False, exact)):
            sp = run(fastpath, split=retry.split)
            assert torch.equal(sp[:, N1:], base[:, N1:])                     # riders: whole workgroups
            assert not torch.equal(sp[:, :N1], base[:, :N1]) or (gain >= 5 and fastpath)   # main tail: two halves (gain 8, fast path: recomputed unsplit by the retry)
            assert _rel(sp, base) < 2e-3
        assert int(retry.buf[1:].abs().sum().item()) == 0
        assert retry.count() - n0 == (wgs if gain >= 5 else 0)
    else:
        assert retry.split is None or torch.equal(run(True, split=retry.split), fast)      # a launch that does not qualify ignores the workspace
