"""GPU: the training step's first operator (BASELINE config 5) — flash-attention BACKWARD on MFMA (tg_attention_bwd) —
against torch.autograd of a plain fp32 restatement of the same op on the same bf16-rounded inputs.  Covers the single attention call
(ragged tiles, several heads / batch items, strided fused-QKV views) and the To2V processor's three-call composition
(attention_processor.py:2066-2135: O = cat(sdpa(q,k,v) + s * sdpa(qx,kv,vv), sdpa(qv, cat(kx,kv), cat(vx,vv)))) where K / V tensors are
shared between calls and their gradients accumulate."""
import math

import pytest
import torch
from conftest import measured

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return measured(((a - b).norm() / (b.norm() + 1e-12)).item())


def _sdpa(q, k, v, heads, scale):
    """[B, n, heads*64] fp32 tensors -> softmax(scale q k^T) v per head, merged back."""
    B, nq, _ = q.shape
    sp = lambda t: t.view(B, t.shape[1], heads, 64).transpose(1, 2)
    p = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * scale, dim=-1)
    return (p @ sp(v)).transpose(1, 2).reshape(B, nq, heads * 64)


def _rand(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


@pytest.mark.parametrize("B,H,nq,nk", [(1, 2, 50, 70), (2, 3, 513, 1500), (1, 1, 64, 64), (1, 4, 200, 33), (1, 8, 1100, 700), (2, 4, 2500, 300),
                                       (2, 4, 1030, 257), (1, 8, 4100, 1000), (1, 8, 129, 5)])
def test_attention_bwd_vs_autograd(B, H, nq, nk):
    """(the shapes from the fifth on — 8 (batch, head) pairs, >= 4 query tiles per key block — take the one-kernel form: ragged last query tile, a key
    block with a single key, 4 key blocks chained, the smallest legal call; TG_ATTN_BWD_FUSED=0 in the child run below sends them through the two-kernel
    form as well)"""
    from tokensgen_amd import kernels as K
    scale = 1.0 / math.sqrt(64)
    # q|k|v as column slices of one fused buffer (what the QKV GEMM writes): exercises the row / batch strides
    nmax = max(nq, nk)
    fused = _rand(B, nmax, 3 * H * 64, seed=1, scale=1.5)
    q, k, v = fused[:, :nq, :H * 64], fused[:, :nk, H * 64:2 * H * 64], fused[:, :nk, 2 * H * 64:]
    g = _rand(B, nq, H * 64, seed=2)
    qf, kf, vf = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    o = _sdpa(qf, kf, vf, H, scale)
    (o * g.float()).sum().backward()
    fd = fused.to(DEV)
    qd, kd, vd = fd[:, :nq, :H * 64], fd[:, :nk, H * 64:2 * H * 64], fd[:, :nk, 2 * H * 64:]
    dq, dk, dv = K.attention_bwd(qd, kd, vd, o.detach().to(BF).to(DEV), g.to(DEV), H, scale)
    assert dq.shape == (B, nq, H * 64) and dk.shape == dv.shape == (B, nk, H * 64) and dq.dtype == torch.float32
    assert _rel(dv, vf.grad) < 4.5e-3          # tolerances = 2x the values measured on MI355X (profiles/r2_parity_report.json: 2.3e-3 / 2.7e-3)
    assert _rel(dq, qf.grad) < 5.5e-3
    assert _rel(dk, kf.grad) < 5.5e-3
    # deterministic (no atomics on the data) and accumulate adds
    dq2, dk2, dv2 = K.attention_bwd(qd, kd, vd, o.detach().to(BF).to(DEV), g.to(DEV), H, scale)
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)
    K.attention_bwd_check(DEV)                                   # no poll timed out, every head's key blocks sat on one XCD
    # bf16 dV straight from the epilogue (tg_attn_bwd_problem.dv_bf16) into the V third of a fused projection-gradient buffer: bitwise the conversion pass's values,
    # alone (no fp32 dv is produced) and beside the fp32 tensor; nothing outside the third is touched
    dfused = torch.full((B, nk + 3, 3 * H * 64), 7.0, dtype=BF, device=DEV)
    dq4, dk4, dv4 = K.attention_bwd(qd, kd, vd, o.detach().to(BF).to(DEV), g.to(DEV), H, scale, dv_bf16=dfused[:, :nk, 2 * H * 64:])
    assert dv4 is None and torch.equal(dq4, dq) and torch.equal(dk4, dk) and torch.equal(dfused[:, :nk, 2 * H * 64:], dv.to(BF))
    assert bool((dfused[:, nk:] == 7.0).all()) and bool((dfused[:, :, :2 * H * 64] == 7.0).all())
    dfused2 = torch.zeros_like(dfused)
    dv5 = torch.empty_like(dv)
    K.attention_bwd(qd, kd, vd, o.detach().to(BF).to(DEV), g.to(DEV), H, scale, dv=dv5, dv_bf16=dfused2[:, :nk, 2 * H * 64:])
    assert torch.equal(dv5, dv) and torch.equal(dfused2[:, :nk, 2 * H * 64:], dv.to(BF))
    K.attention_bwd(qd, kd, vd, o.detach().to(BF).to(DEV), g.to(DEV), H, scale, dq=dq2, dk=dk2, dv=dv2, accumulate=True)
    # (the one-kernel form adds its key blocks' dQ contributions to what is there one after the other: equal up to fp32 summation order)
    assert torch.allclose(dq2, 2 * dq, rtol=1e-4, atol=1e-5) and torch.allclose(dk2, 2 * dk) and torch.allclose(dv2, 2 * dv)
    # accumulate into dk / dv with the bf16 copy: bf16 of the SUM the fp32 tensor now holds; without the fp32 tensor the call is refused
    dv6 = dv.clone()
    K.attention_bwd(qd, kd, vd, o.detach().to(BF).to(DEV), g.to(DEV), H, scale, dk=dk.clone(), dv=dv6, accumulate=2, dv_bf16=dfused2[:, :nk, 2 * H * 64:])
    assert torch.equal(dv6, dv2) and torch.equal(dfused2[:, :nk, 2 * H * 64:], dv2.to(BF))
    with pytest.raises(AssertionError, match="accumulate into dv"):
        K.attention_bwd(qd, kd, vd, o.detach().to(BF).to(DEV), g.to(DEV), H, scale, accumulate=2, dv_bf16=dfused2[:, :nk, 2 * H * 64:])
    # the forward's own log-sum-exp (tg_attention_fwd_lse) instead of the statistics pass: same row statistics up to fp32 rounding
    pad = (nk + 63) // 64 * 64
    vt = K.transpose_v(vd, H, 0, nk, torch.zeros(B, H, 64, pad, dtype=BF, device=DEV))
    out = torch.empty(B, nq, H * 64, dtype=BF, device=DEV)
    _, lse = K.attention_lse(qd, kd, vt, nk, out, H, scale)
    sp = lambda t: t.view(B, t.shape[1], H, 64).transpose(1, 2)
    want = torch.logsumexp(sp(qf.detach()) @ sp(kf.detach()).transpose(-1, -2) * scale, dim=-1) / math.log(2.0)
    assert (lse.cpu() - want).abs().max().item() < 2e-3
    assert _rel(out, o.detach()) < 8e-3
    dq3, dk3, dv3 = K.attention_bwd(qd, kd, vd, out, g.to(DEV), H, scale, lse=lse)
    assert _rel(dq3, qf.grad) < 5.5e-3 and _rel(dk3, kf.grad) < 5.5e-3 and _rel(dv3, vf.grad) < 4.5e-3


def test_attention_lse_on_the_constant_shift_kernel_and_its_backward():
    """tg_attention_fwd_lse_ex (the training forward of the 17776^2 call): K rows prescaled by scale * log2(e) in the real norm kernel, the key-norm
    bound from the same launch, the verified constant-shift 512-row kernel WITH the log-sum-exp store (launch scale: 2 x 48 x 11 = 1056 workgroups
    + a ragged tail of 68 rows on the 4-wave kernel) against the running-max kernels on the same operands and torch on sampled rows; then
    tg_attention_bwd on the scaled K with scale = ln 2 against autograd of softmax(ln2 q k'^T) v on sampled heads."""
    from tokensgen_amd import kernels as K
    B, H, N = 2, 48, 11 * 512 + 68
    D = H * 64
    kscale = 0.125 * 1.4426950408889634
    qkv = _rand(B, N, 3 * D, seed=91).to(DEV)
    g = torch.Generator().manual_seed(9)
    w = lambda: (1.0 + 0.1 * torch.randn(64, generator=g)).to(BF).to(DEV)
    bb = lambda: (0.1 * torch.randn(64, generator=g)).to(BF).to(DEV)
    km, kws = torch.zeros(B, H, dtype=torch.float32, device=DEV), K.kmax_workspace(N, H, B, DEV)
    K.qk_layernorm_rope_pair(qkv[:, :, :D], qkv[:, :, D:2 * D], H, w(), bb(), w(), bb(), 1e-6, k_scale=kscale, kmax=km, kmax_ws=kws)
    q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
    vt = K.transpose_v(v, H, 0, N, torch.zeros(B, H, 64, (N + 63) // 64 * 64, dtype=BF, device=DEV))
    retry = K.AttnRetry(N, 0, H, B, DEV)
    o_fast, o_run = torch.zeros(B, N, D, dtype=BF, device=DEV), torch.zeros(B, N, D, dtype=BF, device=DEV)
    _, lse_fast = K.attention_lse(q, k, vt, N, o_fast, H, 0.125, k_prescaled=True, kmax=km, retry=retry)
    _, lse_run = K.attention_lse(q, k, vt, N, o_run, H, 0.125, k_prescaled=True)
    assert retry.count() == 0 and int(retry.buf[1:].abs().sum().item()) == 0          # the constant shift stood everywhere
    assert not torch.equal(o_fast, o_run)                                              # ... and it is a different kernel
    assert _rel(o_fast, o_run) < 4e-3 and (lse_fast - lse_run).abs().max().item() < 2e-3
    rows = torch.cat([torch.arange(0, 40), torch.arange(N - 40, N)]).to(DEV)
    for h in (0, 23, 47):
        sl = slice(h * 64, h * 64 + 64)
        sc = q[:, rows, sl].float() @ k[:, :, sl].float().transpose(1, 2)              # log2-domain scores
        assert (lse_fast[:, h, rows] - torch.logsumexp(sc * math.log(2.0), -1) / math.log(2.0)).abs().max().item() < 2e-3, h
        assert _rel(o_fast[:, rows, sl], torch.softmax(sc * math.log(2.0), -1) @ v[:, :, sl].float()) < 8e-3, h
    # backward on the scaled rows: scale = ln 2, dk = gradient of the scaled K
    do = _rand(B, N, D, seed=92).to(DEV)
    dq, dk, dv = K.attention_bwd(q, k, v, o_fast, do, H, math.log(2.0), lse=lse_fast)
    for h in (0, 47):
        sl = slice(h * 64, h * 64 + 64)
        qf, kf, vf = (t[:1, :, sl].float().clone().requires_grad_(True) for t in (q, k, v))
        (((torch.softmax(qf @ kf.transpose(1, 2) * math.log(2.0), -1) @ vf)) * do[:1, :, sl].float()).sum().backward()
        assert _rel(dq[:1, :, sl], qf.grad) < 5.5e-3 and _rel(dk[:1, :, sl], kf.grad) < 5.5e-3 and _rel(dv[:1, :, sl], vf.grad) < 4.5e-3, h



@pytest.mark.parametrize("B,H,nq,nk", [(1, 2, 200, 333), (2, 4, 1100, 260)])
def test_attention_bwd_product_kernels_vs_the_independent_cross_check_library(B, H, nq, nk):
    """tests/libtg_crosscheck.so (tests/csrc/attention_bwd_crosscheck.hip: correct-first kernels — LDS-staged 64 x 64 tiles, explicit transposes, a
    one-thread-per-query statistics pass; TEST-ONLY, never in the product library) is an independent implementation of the same mathematics: it must pass
    the same autograd comparison, and the product kernels must agree with it to bf16-operand rounding.  Both are called through their C ABIs."""
    import ctypes as C
    import os
    from tokensgen_amd import kernels as K
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtg_crosscheck.so")
    assert os.path.exists(so), "build it: python -c 'import __graft_entry__ as g; g.build()' (make -C tests/csrc)"
    x = C.CDLL(so)
    vp, l, i, f = C.c_void_p, C.c_long, C.c_int, C.c_float
    x.tgx_attention_bwd.argtypes = [vp, l, l] * 5 + [vp, l, l] * 3 + [i, i, i, i, f, i, vp, vp, vp]
    x.tgx_attention_bwd.restype = C.c_int
    x.tgx_last_error_string.restype = C.c_char_p
    scale = 1.0 / math.sqrt(64)
    D = H * 64
    fused = _rand(B, max(nq, nk), 3 * D, seed=11, scale=1.5)
    q, k, v = fused[:, :nq, :D], fused[:, :nk, D:2 * D], fused[:, :nk, 2 * D:]
    g = _rand(B, nq, D, seed=12)
    qf, kf, vf = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    o = _sdpa(qf, kf, vf, H, scale)
    (o * g.float()).sum().backward()
    fd, od, gd = fused.to(DEV), o.detach().to(BF).to(DEV), g.to(DEV)
    qd, kd, vd = fd[:, :nq, :D], fd[:, :nk, D:2 * D], fd[:, :nk, 2 * D:]
    dq, dk, dv = (torch.full((B, n, D), float("nan"), dtype=torch.float32, device=DEV) for n in (nq, nk, nk))
    ws = torch.empty(2 * B * H * nq, dtype=torch.float32, device=DEV)
    st = lambda t: (t.data_ptr(), t.stride(1), t.stride(0))
    rc = x.tgx_attention_bwd(*st(qd), *st(kd), *st(vd), *st(od), *st(gd), *st(dq), *st(dk), *st(dv), nq, nk, H, B, scale, 0, None, ws.data_ptr(),
                             torch.cuda.current_stream().cuda_stream)
    assert rc == 0, x.tgx_last_error_string()
    torch.cuda.synchronize()
    assert _rel(dv, vf.grad) < 4.5e-3 and _rel(dq, qf.grad) < 5.5e-3 and _rel(dk, kf.grad) < 5.5e-3
    pq, pk, pv = K.attention_bwd(qd, kd, vd, od, gd, H, scale)
    K.attention_bwd_check(DEV)
    assert _rel(pq, dq) < 5e-3 and _rel(pk, dk) < 5e-3 and _rel(pv, dv) < 4e-3


def test_attention_bwd_two_problems_in_one_launch_equal_two_calls():
    """tg_attention_bwd_multi: the To2V processor's main call and its vip-key call (attention_processor.py:2066-2069, 2117-2125; same queries' length) handed over
    together — when both take the one-kernel form the second rides in the first's launch.  Bitwise equal to the two separate calls (each head's key blocks still add
    their dQ shares in key-block order), accumulate flags per problem honoured, status words clean; a pair whose second problem takes the two-launch form works too."""
    from tokensgen_amd import kernels as K
    B, H, nq, nk1, nk2 = 1, 8, 4100, 1000, 300
    D = H * 64
    mk = lambda n, seed, sc=1.2: _rand(B, n, D, seed=seed, scale=sc).to(DEV)
    q1, k1, v1, o1, g1 = mk(nq, 31), mk(nk1, 32), mk(nk1, 33), mk(nq, 34, 0.3), mk(nq, 35, 0.3)
    q2, k2, v2, o2, g2 = mk(nq, 36), mk(nk2, 37), mk(nk2, 38), mk(nq, 39, 0.3), mk(nq, 40, 0.3)
    base = torch.randn(B, nk2, D, generator=torch.Generator().manual_seed(41)).to(DEV)
    for kk2, nkk2 in ((k2, nk2), (mk(1500, 42), 1500)):            # second case: 6 key blocks for 129 query tiles -> still one kernel; then a two-launch second problem below
        vv2 = v2 if nkk2 == nk2 else mk(nkk2, 43)
        b0 = base if nkk2 == nk2 else torch.randn(B, nkk2, D, generator=torch.Generator().manual_seed(44)).to(DEV)
        a1 = K.attention_bwd(q1, k1, v1, o1, g1, H, math.log(2.0))
        dk_s, dv_s = b0.clone(), b0.clone() * 2
        a2 = K.attention_bwd(q2, kk2, vv2, o2, g2, H, 0.125, dk=dk_s, dv=dv_s, accumulate=2)
        dk_m, dv_m = b0.clone(), b0.clone() * 2
        m1, m2 = K.attention_bwd_multi([dict(q=q1, k=k1, v=v1, o=o1, dout=g1, scale=math.log(2.0)),
                                        dict(q=q2, k=kk2, v=vv2, o=o2, dout=g2, scale=0.125, dk=dk_m, dv=dv_m, accumulate=2)], H)
        K.attention_bwd_check(DEV)
        for x, y in zip(a1 + a2, m1 + m2):
            assert torch.equal(x, y)
    # a second problem with few query tiles per key block (two-launch form) beside a one-kernel first problem
    q3, o3, g3 = mk(200, 45), mk(200, 46, 0.3), mk(200, 47, 0.3)
    s3 = K.attention_bwd(q3, k1, v1, o3, g3, H, 0.125)
    m1, m3 = K.attention_bwd_multi([dict(q=q1, k=k1, v=v1, o=o1, dout=g1, scale=math.log(2.0)), dict(q=q3, k=k1, v=v1, o=o3, dout=g3, scale=0.125)], H)
    K.attention_bwd_check(DEV)
    for x, y in zip(a1 + s3, m1 + m3):
        assert torch.equal(x, y)


def test_attention_bwd_poll_timeout_is_reported_not_silent():
    """The one-kernel backward's ordered dQ exchange is bounded (a key block that never sees its predecessor's signal goes on instead of hanging the
    GPU) — and that MUST be visible: the poll limit is forced to 1 through the caller-owned status words (include/tokensgen_hip.h, tg_attention_bwd_ex
    status[1]), so key blocks overtake each other; the launch returns 0 (the condition exists on the device only), the sticky word status[0] counts the
    polls that gave up, and kernels.attention_bwd_check() — what To2VTrainStep.micro_step calls before the optimizer may run — raises.  Afterwards the
    word is clear and an ordinary call is exact again."""
    import os
    from tokensgen_amd import kernels as K
    if os.environ.get("TG_ATTN_BWD_FUSED") == "0":
        pytest.skip("the one-kernel form is switched off in this run")
    st = K.BwdDeviceState.get(DEV)
    assert st.one_kernel, "tg_attention_bwd_probe failed on this device: the one-kernel backward would never be selected"
    B, H, nq, nk = 1, 8, 1100, 700
    scale = 0.125
    fused = _rand(B, nq, 3 * H * 64, seed=11, scale=1.5).to(DEV)
    q, k, v = fused[:, :, :H * 64], fused[:, :nk, H * 64:2 * H * 64], fused[:, :nk, 2 * H * 64:]
    g = _rand(B, nq, H * 64, seed=12).to(DEV)
    o = _sdpa(q.float(), k.float(), v.float(), H, scale).to(BF)
    K.attention_bwd_check(DEV)                                       # clean before
    good = K.attention_bwd(q, k, v, o, g, H, scale)
    K.attention_bwd_check(DEV)
    st.status[1] = 1                                                 # poll limit 1: every wait that is not already satisfied gives up
    try:
        K.attention_bwd(q, k, v, o, g, H, scale)
    finally:
        st.status[1] = 0
    assert int(st.status[0].item()) > 0
    with pytest.raises(RuntimeError, match="timed out"):
        K.attention_bwd_check(DEV)
    K.attention_bwd_check(DEV)                                       # reported once, then clear
    again = K.attention_bwd(q, k, v, o, g, H, scale)
    K.attention_bwd_check(DEV)
    assert all(torch.equal(a, b) for a, b in zip(good, again))


def test_attention_bwd_without_the_probe_flag_takes_two_launches():
    """The plain entry point tg_attention_bwd (no flags, no status words) never selects the one-kernel form; its results equal tg_attention_bwd_ex's
    dK / dV bit for bit is not promised (different kernels) — both must match autograd's."""
    import ctypes as C
    from tokensgen_amd import kernels as K
    from tokensgen_amd import lib as L
    B, H, nq, nk = 1, 8, 1100, 700
    scale = 0.125
    fused = _rand(B, nq, 3 * H * 64, seed=21, scale=1.5)
    q, k, v = fused[:, :, :H * 64], fused[:, :nk, H * 64:2 * H * 64], fused[:, :nk, 2 * H * 64:]
    g = _rand(B, nq, H * 64, seed=22)
    qf, kf, vf = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    o = _sdpa(qf, kf, vf, H, scale)
    (o * g.float()).sum().backward()
    fd, gd, od = fused.to(DEV), g.to(DEV), o.detach().to(BF).to(DEV)
    qd, kd, vd = fd[:, :, :H * 64], fd[:, :nk, H * 64:2 * H * 64], fd[:, :nk, 2 * H * 64:]
    lib = L.load()
    f32 = torch.float32
    dq, dk, dv = (torch.empty(B, n, H * 64, dtype=f32, device=DEV) for n in (nq, nk, nk))
    ws = torch.empty(lib.tg_attention_bwd_ws_floats(nq, nk, H, B), dtype=f32, device=DEV)
    L.check(lib.tg_attention_bwd(qd.data_ptr(), qd.stride(1), qd.stride(0), kd.data_ptr(), kd.stride(1), kd.stride(0), vd.data_ptr(), vd.stride(1), vd.stride(0),
                                 od.data_ptr(), od.stride(1), od.stride(0), gd.data_ptr(), gd.stride(1), gd.stride(0), dq.data_ptr(), dq.stride(1), dq.stride(0),
                                 dk.data_ptr(), dk.stride(1), dk.stride(0), dv.data_ptr(), dv.stride(1), dv.stride(0), nq, nk, H, B, scale, 0, None,
                                 ws.data_ptr(), K._stream()), "tg_attention_bwd")
    assert _rel(dq, qf.grad) < 5.5e-3 and _rel(dk, kf.grad) < 5.5e-3 and _rel(dv, vf.grad) < 4.5e-3
    # TG_BWD_ONE_KERNEL without status words is an argument error, not a crash
    code = lib.tg_attention_bwd_ex(qd.data_ptr(), qd.stride(1), qd.stride(0), kd.data_ptr(), kd.stride(1), kd.stride(0), vd.data_ptr(), vd.stride(1), vd.stride(0),
                                   od.data_ptr(), od.stride(1), od.stride(0), gd.data_ptr(), gd.stride(1), gd.stride(0), dq.data_ptr(), dq.stride(1), dq.stride(0),
                                   dk.data_ptr(), dk.stride(1), dk.stride(0), dv.data_ptr(), dv.stride(1), dv.stride(0), nq, nk, H, B, scale, 0, None,
                                   ws.data_ptr(), L.TG_BWD_ONE_KERNEL, None, K._stream())
    assert code == -1 and b"status" in lib.tg_last_error_string()


def test_attention_bwd_one_kernel_form_after_multi_stream_graph_work_in_a_child_process(golden_dir):
    """bench.py's default run is window -> VAE (three tile streams + captured HIP graphs) -> training in ONE process, and after that VAE phase the
    dispatcher no longer puts workgroup b on XCD b % 8 (all 4096 probe workgroups were off it; residue classes still share an XCD).  The probe used to
    demand the identity and silently sent the training record to the two-launch form.  Child process: tiny tiled VAE decode / encode three times (eager,
    capture, replay), THEN the first backward call of the process — the device probe must pass, the one-kernel form must run, the gradients must match
    autograd, and the placement / poll status words must be clean."""
    import os
    import subprocess
    import sys
    if os.environ.get("TG_ATTN_BWD_FUSED") == "0":
        pytest.skip("the one-kernel form is switched off in this run")
    if os.environ.get("TG_TEST_AFTER_VAE") != "1":
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-k", "after_multi_stream_graph_work", "-x"],
                           env=dict(os.environ, TG_TEST_AFTER_VAE="1"), capture_output=True, text=True, timeout=600,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
        return
    from oracle import vae_ref as V
    from tokensgen_amd import kernels as K
    from tokensgen_amd.vae import AutoencoderKLCogVideoX
    assert not K.BwdDeviceState._by_device, "the probe must not have run yet in this process"
    g = torch.load(os.path.join(golden_dir, "vae_tiny.pt"), weights_only=False)
    cfg = g["cfg"]
    vae = AutoencoderKLCogVideoX(block_out_channels=cfg["block_out_channels"], layers_per_block=cfg["layers_per_block"], sample_height=64, sample_width=96,
                                 device=DEV)
    vae.load_state_dict(V.make_state_dict(cfg, seed=g["weight_seed"]))
    vae.enable_tiling(); vae.enable_slicing()
    gen = torch.Generator().manual_seed(5)
    z = torch.randn(1, 16, 13, 8, 12, generator=gen).to(DEV, BF)
    x = (torch.rand(1, 3, 17, 64, 96, generator=gen) * 2 - 1).to(DEV, BF)
    for _ in range(3):
        vae.decode(z).sample
        vae.encode(x).latent_dist.parameters
    torch.cuda.synchronize()
    B, H, nq, nk = 1, 8, 1100, 700
    scale = 0.125
    fused = _rand(B, nq, 3 * H * 64, seed=41, scale=1.5)
    q, k, v = fused[:, :, :H * 64], fused[:, :nk, H * 64:2 * H * 64], fused[:, :nk, 2 * H * 64:]
    gq = _rand(B, nq, H * 64, seed=42)
    qf, kf, vf = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    o = _sdpa(qf, kf, vf, H, scale)
    (o * gq.float()).sum().backward()
    fd = fused.to(DEV)
    dq, dk, dv = K.attention_bwd(fd[:, :, :H * 64], fd[:, :nk, H * 64:2 * H * 64], fd[:, :nk, 2 * H * 64:], o.detach().to(BF).to(DEV), gq.to(DEV), H, scale)
    st = K.BwdDeviceState.get(DEV)
    assert st.one_kernel and st.probe["flagged"] == 0 and st.probe["sums_exact"], st.probe
    K.attention_bwd_check(DEV)
    assert _rel(dq, qf.grad) < 5.5e-3 and _rel(dk, kf.grad) < 5.5e-3 and _rel(dv, vf.grad) < 4.5e-3


def test_attention_bwd_two_kernel_form_in_a_child_process():
    """TG_ATTN_BWD_FUSED=0: the dK/dV + dQ launches for the shapes the one-kernel form would take (same autograd comparison, same determinism check)."""
    import os
    import subprocess
    import sys
    if os.environ.get("TG_ATTN_BWD_FUSED") == "0":
        pytest.skip("already inside a cross-check run")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-k", "test_attention_bwd_vs_autograd", "-x"],
                       env=dict(os.environ, TG_ATTN_BWD_FUSED="0"), capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout



def test_to2v_processor_attention_gradients():
    """The three SDPA calls of VideoIPAdapterCogVideoXAttnProcessor2_0 (func_type "1") composed from tg_attention_bwd calls: gradients of
    every q / k / v tensor of the base and the vip projections (the vip ones are the TRAINABLE path, cogvideox_transformer_3d.py:207-218)
    against autograd of the fp32 restatement."""
    from tokensgen_amd import kernels as K
    B, H, N1, Np, s = 2, 2, 150, 40, 0.6
    scale = 1.0 / math.sqrt(64)
    names = ("q", "k", "v", "qx", "kx", "vx", "qv", "kv", "vv")
    shapes = dict(q=N1, k=N1, v=N1, qx=N1, kx=N1, vx=N1, qv=Np, kv=Np, vv=Np)
    t = {n: _rand(B, shapes[n], H * 64, seed=10 + i, scale=1.2) for i, n in enumerate(names)}
    g = _rand(B, N1 + Np, H * 64, seed=30)
    f = {n: x.float().clone().requires_grad_(True) for n, x in t.items()}
    o1 = _sdpa(f["q"], f["k"], f["v"], H, scale)
    o2 = _sdpa(f["qx"], f["kv"], f["vv"], H, scale)
    o3 = _sdpa(f["qv"], torch.cat([f["kx"], f["kv"]], 1), torch.cat([f["vx"], f["vv"]], 1), H, scale)
    out = torch.cat([o1 + s * o2, o3], dim=1)
    (out * g.float()).sum().backward()
    d = {n: x.to(DEV) for n, x in t.items()}
    gd = g.to(DEV)
    bf = lambda x: x.detach().to(BF).to(DEV)
    f32 = torch.float32
    # SDPA #1
    dq, dk, dv = K.attention_bwd(d["q"], d["k"], d["v"], bf(o1), gd[:, :N1], H, scale)
    # SDPA #2: dO2 = s * dO (bf16, like the forward's `scale * O2` tensor)
    g2 = (gd[:, :N1].float() * s).to(BF)
    dqx, dkv, dvv = K.attention_bwd(d["qx"], d["kv"], d["vv"], bf(o2), g2, H, scale)
    # SDPA #3: keys / values = cat(x-part, vip-part): gradients land in one [N1 + Np] buffer, whose tail ADDS to SDPA #2's kv / vv gradients
    kcat, vcat = torch.cat([d["kx"], d["kv"]], 1), torch.cat([d["vx"], d["vv"]], 1)
    dkcat = torch.zeros(B, N1 + Np, H * 64, dtype=f32, device=DEV)
    dvcat = torch.zeros_like(dkcat)
    dkcat[:, N1:], dvcat[:, N1:] = dkv, dvv
    dqv, _, _ = K.attention_bwd(d["qv"], kcat, vcat, bf(o3), gd[:, N1:], H, scale, dk=dkcat, dv=dvcat, accumulate=True,
                                dq=torch.zeros(B, Np, H * 64, dtype=f32, device=DEV))
    got = dict(q=dq, k=dk, v=dv, qx=dqx, kx=dkcat[:, :N1], vx=dvcat[:, :N1], qv=dqv, kv=dkcat[:, N1:], vv=dvcat[:, N1:])
    for n in names:
        assert _rel(got[n], f[n].grad) < 5e-3, n          # measured 2.4e-3


def test_vpred_loss_and_gradient_vs_autograd():
    """tg_vpred_loss_grad against autograd of the oracle's restatement of train_cogvideo_to2v.py:1990-2010 (bf16 tensors, per-frame timesteps)."""
    from oracle import scheduler_ref as S
    from oracle import train_ref as T
    from tokensgen_amd import train
    _, ac = S.alphas_cumprod()
    ac = torch.as_tensor(ac, dtype=torch.float32)
    B, F, C, H, W = 2, 13, 16, 6, 10
    g = torch.Generator().manual_seed(5)
    out, noisy, x0 = (torch.randn(B, F, C, H, W, generator=g).to(BF) for _ in range(3))
    ts = torch.randint(20, 980, (B, F), generator=g)
    o = out.clone().requires_grad_(True)
    loss, per_item = T.vpred_loss(ac, o, noisy, x0, ts)
    loss.backward()
    l2, p2, grad = train.vpred_loss_and_grad(out.to(DEV), noisy.to(DEV), x0.to(DEV), ts, ac)
    assert grad.shape == out.shape and grad.dtype == BF
    assert abs(l2.item() - loss.item()) < 2e-3 * abs(loss.item())
    assert torch.allclose(p2.cpu(), per_item.detach().float(), rtol=2e-3)
    assert _rel(grad, o.grad) < 7.5e-3          # measured 3.7e-3 (the gradient is a bf16 tensor)
    # scalar timestep per batch item
    ts1 = torch.tensor([100, 900])
    o = out.clone().requires_grad_(True)
    loss, _ = T.vpred_loss(ac, o, noisy, x0, ts1)
    loss.backward()
    l3, _, grad3 = train.vpred_loss_and_grad(out.to(DEV), noisy.to(DEV), x0.to(DEV), ts1, ac)
    assert abs(l3.item() - loss.item()) < 2e-3 * abs(loss.item()) and _rel(grad3, o.grad) < 7.5e-3


def _rope_tables(n, seed):
    g = torch.Generator().manual_seed(seed)
    ang = torch.rand(n, 32, generator=g) * 6.28
    return ang.cos().repeat_interleave(2, dim=1).float().contiguous(), ang.sin().repeat_interleave(2, dim=1).float().contiguous()


def test_qk_layernorm_rope_backward_and_linear_backward_vs_autograd():
    """tg_qk_layernorm_rope_bwd (per-head LayerNorm + interleaved-pair RoPE, two rotary segments + un-rotated text rows, output scale) and the
    GEMM weight / bias / input gradients (tg_transpose_2d + tg_gemm_bf16 + tg_colsum) against autograd of the fp32 restatement."""
    from oracle import dit_ref as O
    from tokensgen_amd import train
    B, T, H, Nt, n0 = 2, 90, 3, 11, 50
    D = H * 64
    x = _rand(B, T, D, seed=1, scale=1.3)
    dy = _rand(B, T, D, seed=2).float()
    gw, gb = (_rand(64, seed=3, scale=0.2).float() + 1).to(BF), _rand(64, seed=4, scale=0.2)
    r0, r1 = _rope_tables(n0, 5), _rope_tables(T - Nt - n0, 6)
    xf, wf, bfv = x.float().clone().requires_grad_(True), gw.float().clone().requires_grad_(True), gb.float().clone().requires_grad_(True)
    xh = xf.view(B, T, H, 64).transpose(1, 2)
    ln = torch.nn.functional.layer_norm(xh, (64,), wf, bfv, 1e-6)
    y = torch.cat([ln[:, :, :Nt], O.apply_rope(ln[:, :, Nt:Nt + n0], r0), O.apply_rope(ln[:, :, Nt + n0:], r1)], dim=2) * 0.37
    (y.transpose(1, 2).reshape(B, T, D) * dy).sum().backward()
    dx, dg, db = train.qk_layernorm_rope_backward(x.to(DEV), dy.to(DEV), H, gw.to(DEV), 1e-6, (Nt, tuple(t.to(DEV) for t in r0)),
                                                 (Nt + n0, tuple(t.to(DEV) for t in r1)), out_scale=0.37)
    assert _rel(dx, xf.grad) < 4e-3 and _rel(dg, wf.grad) < 2e-3 and _rel(db, bfv.grad) < 2e-3
    # linear backward
    M, cin, cout = 333, 192, 256
    xi, dyo, w = _rand(M, cin, seed=7), _rand(M, cout, seed=8), _rand(cout, cin, seed=9, scale=0.1)
    dW, dbias, dxi = train.linear_backward(xi.to(DEV), dyo.to(DEV), w.to(DEV), need_dx=True)
    assert _rel(dW, dyo.float().t() @ xi.float()) < 4e-3
    assert _rel(dbias, dyo.float().sum(0)) < 1e-3
    assert _rel(dxi, dyo.float() @ w.float()) < 4e-3


@pytest.mark.parametrize("R,C,ld,pad", [(200, 128, 128, 256), (36, 64, 192, 64), (1000, 3072, 9216, 1024), (130, 72, 72, 136), (50, 30, 30, 64), (77, 64, 64, 77)])
def test_transpose_2d_both_forms_are_exact(R, C, ld, pad):
    """tg_transpose_2d: dst[c][r] = src[r][c], zero columns rows..rows_pad; the 16-byte form (cols / strides / rows_pad multiples of 8) and the scalar
    form move the same bits (strided sources: a column slice of a fused buffer)."""
    from tokensgen_amd import train
    src = _rand(R, ld, seed=R + C).to(DEV)[:, :C]
    got = train.transpose_2d(src, pad)
    want = torch.zeros(C, pad, dtype=BF, device=DEV)
    want[:, :R] = src.t()
    assert got.shape == (C, pad) and torch.equal(got, want)



def test_vip_processor_trainable_parameter_gradients_vs_autograd_of_the_oracle(parity):
    """End of the chain for the attention sub-block: from the gradient of the processor's pre-`to_out` output to the gradients of every TRAINABLE
    processor parameter (vip_to_{q,k,v}.{weight,bias}, vip_norm_{q,k}.{weight,bias}; cogvideox_transformer_3d.py:207-218, train_cogvideo_to2v.py:
    1456-1481) — HIP forward pieces (projection GEMM, QK-norm+RoPE, attention) + tg_attention_bwd + tg_qk_layernorm_rope_bwd + GEMM wgrad —
    against autograd through oracle.dit_ref.vip_attention's arithmetic in fp32 on the same bf16-rounded weights and inputs."""
    import numpy as np
    from oracle import dit_ref as O
    from tokensgen_amd import kernels as K
    from tokensgen_amd import lib as L
    from tokensgen_amd import train
    B, H, Nt, Nv, Np = 2, 2, 9, 120, 30
    D, N1, N = H * 64, 9 + 120, 9 + 120 + 30
    f32 = np.float32
    cfg = dict(num_attention_heads=H, attention_head_dim=64, num_layers=1, patch_size=2, time_embed_dim=128, text_embed_dim=64, in_channels=16, out_channels=16)
    sd = {k: v.to(BF).float() for k, v in O.make_state_dict(cfg, n_vip_dim=128, seed=51, std=0.08).items() if k.startswith("transformer_blocks.0.attn1.")}
    pre = "transformer_blocks.0.attn1"
    P = pre + ".processor"
    train_keys = [f"{P}.vip_to_{n}.{wb}" for n in "qkv" for wb in ("weight", "bias")] + [f"{P}.vip_norm_{n}.{wb}" for n in "qk" for wb in ("weight", "bias")]
    for k in train_keys:
        sd[k] = sd[k].clone().requires_grad_(True)
    hidden, enc = _rand(B, Nv, D, seed=52), _rand(B, Nt + Np, D, seed=53)
    rope = O.rope_3d(64, np.arange(4, dtype=f32), np.arange(5, dtype=f32), np.arange(6, dtype=f32))
    vrope = O.rope_3d(64, np.arange(4, dtype=f32) + f32(3), np.arange(5, dtype=f32), np.arange(6, dtype=f32))
    crope = O.rope_3d(64, np.linspace(1000, 1016.25, 5, dtype=f32), np.arange(2, dtype=f32), np.arange(3, dtype=f32))
    taps = {}
    # the oracle's processor without its final to_out: re-run its arithmetic up to the concatenated attention output
    sd_no_out = dict(sd)
    sd_no_out[pre + ".to_out.0.weight"], sd_no_out[pre + ".to_out.0.bias"] = torch.eye(D), torch.zeros(D)
    oh, oe = O.vip_attention(sd_no_out, pre, hidden.float(), enc.float(), H, Np, [0.6], rope, vrope, crope, taps=taps)
    ao_ref = torch.cat([oe[:, :Nt], oh, oe[:, Nt:]], dim=1)                      # text | video | vip rows
    G = _rand(B, N, D, seed=54)
    (ao_ref * G.float()).sum().backward()
    # noise floor: the same arithmetic in bf16 autograd (what the reference's own bf16 training computes) against the fp32 gradients
    sd16 = {k: v.detach().to(BF).requires_grad_(k in train_keys) for k, v in sd_no_out.items()}
    oh16, oe16 = O.vip_attention(sd16, pre, hidden, enc, H, Np, [0.6], rope, vrope, crope)
    (torch.cat([oe16[:, :Nt], oh16, oe16[:, Nt:]], dim=1) * G).sum().backward()
    parity(max(float(_rel(sd16[k].grad, sd[k].grad)) for k in train_keys), 1.0, "noise floor: bf16 autograd of the oracle vs fp32 autograd, worst tensor (informative)")
    # ---- HIP: forward pieces with the pre-norm projections kept, then the backward chain ----
    dev = lambda t: t.detach().to(BF).to(DEV).contiguous()
    xn = torch.cat([enc[:, :Nt], hidden, enc[:, Nt:]], dim=1).to(DEV)          # the processor's inputs in residual-stream row order
    Wv = torch.cat([dev(sd[f"{P}.vip_to_{n}.weight"]) for n in "qkv"]); bv = torch.cat([dev(sd[f"{P}.vip_to_{n}.bias"]) for n in "qkv"])
    Wb = torch.cat([dev(sd[f"{pre}.to_{n}.weight"]) for n in "qkv"]); bb = torch.cat([dev(sd[f"{pre}.to_{n}.bias"]) for n in "qkv"])
    qkvv_pre = torch.empty(B, N, 3 * D, dtype=BF, device=DEV); K.gemm(xn, Wv, bv, qkvv_pre, L.EPI_BIAS)
    qkv = torch.empty(B, N1, 3 * D, dtype=BF, device=DEV); K.gemm(xn[:, :N1].contiguous(), Wb, bb, qkv, L.EPI_BIAS)
    qkvv = qkvv_pre.clone()
    tab = lambda r: tuple(t.to(DEV) for t in r)
    K.qk_layernorm_rope(qkv[:, :, :D], H, dev(sd[pre + ".norm_q.weight"]), dev(sd[pre + ".norm_q.bias"]), 1e-6, (Nt, tab(rope)))
    K.qk_layernorm_rope(qkv[:, :, D:2 * D], H, dev(sd[pre + ".norm_k.weight"]), dev(sd[pre + ".norm_k.bias"]), 1e-6, (Nt, tab(rope)))
    gq, bq_, gk, bk_ = (dev(sd[f"{P}.vip_norm_{n}.{wb}"]) for n in "qk" for wb in ("weight", "bias"))
    K.qk_layernorm_rope(qkvv[:, :, :D], H, gq, bq_, 1e-6, (Nt, tab(vrope)), (N1, tab(crope)))
    K.qk_layernorm_rope(qkvv[:, :, D:2 * D], H, gk, bk_, 1e-6, (Nt, tab(vrope)), (N1, tab(crope)))
    sl = lambda t, a, b_, c: t[:, a:b_, c * D:(c + 1) * D]
    q, k, v = (sl(qkv, 0, N1, c) for c in range(3))
    qx, kx, vx = (sl(qkvv, 0, N1, c) for c in range(3))
    qv, kv, vv = (sl(qkvv, N1, N, c) for c in range(3))
    merge = lambda o: o.transpose(1, 2).reshape(B, -1, D)
    o1, o2, o3 = (dev(merge(taps[n])) for n in ("o1", "o2", "o3"))               # the saved forward outputs (bf16)
    Gd = G.to(DEV)
    d_out = torch.cat([Gd[:, :N1], Gd[:, N1:]], dim=1)                          # gradient of cat(O1 + s O2, O3) in row order text|video | vip
    grads = train.to2v_attention_backward(q, k, v, qx, kx, vx, qv, kv, vv, o1, o2, o3, d_out, H, 1.0 / math.sqrt(64), 0.6)
    got = train.vip_projection_backward(xn, qkvv_pre, grads, H, Nt, N1, gq, gk, tab(vrope), tab(crope))
    parity(max(float(_rel(g_, sd[f"{P}.{name}"].grad)) for name, g_ in got.items()), 3e-2,
           "worst trainable processor tensor, HIP vs fp32 autograd (r2 measured 2.3e-2; compare the floor above)")


@pytest.mark.parametrize("D", [3072, 4096, 4608])
def test_adaln_backward_row_forms_vs_autograd(D):
    """tg_adaln_modulate_bwd at the widths that pick its three forms (row kept in registers as 6 or 8 chunks per lane; rows re-read beyond 4096): plain affine LayerNorm
    backward against autograd of the fp32 formula, with and without the product tensors / the residual add."""
    import torch.nn.functional as F
    from tokensgen_amd import train
    B, T = 2, 37
    x, dy, res = _rand(B, T, D, seed=171), _rand(B, T, D, seed=172), _rand(B, T, D, seed=173)
    w, bvec = (1 + 0.2 * _rand(D, seed=174).float()).to(BF), _rand(D, seed=175, scale=0.2)
    xf, wf, bf_ = x.float().requires_grad_(True), w.float().requires_grad_(True), bvec.float().requires_grad_(True)
    (F.layer_norm(xf, (D,), wf, bf_, 1e-5) * dy.float()).sum().backward()
    dx = torch.empty(B, T, D, dtype=BF, device=DEV)
    t_dln, t_dlnx, _ = train._adaln_bwd(x.to(DEV), dy.to(DEV), dx, w.to(DEV), bvec.to(DEV), 1e-5, None)
    assert _rel(dx, xf.grad) < 4e-3
    assert _rel(train._colsum_f32(t_dln), bf_.grad) < 1e-5 and _rel(train._colsum_f32(t_dlnx), wf.grad) < 1e-5
    dx2 = torch.empty(B, T, D, dtype=BF, device=DEV)
    train._adaln_bwd(x.to(DEV), dy.to(DEV), dx2, w.to(DEV), bvec.to(DEV), 1e-5, None, products=False, add=res.to(DEV))
    assert torch.equal(dx2, (dx.float() + res.to(DEV).float()).to(BF))


def test_adaln_gate_and_activation_backward_kernels_vs_autograd():
    """tg_adaln_modulate_bwd / tg_gate_residual_bwd / tg_act / tg_colsum_f32 one by one against autograd of the fp32 formulas
    (normalization.py:441-488, cogvideox_transformer_3d.py:290-324, FeedForward gelu-approximate)."""
    import torch.nn.functional as F
    from tokensgen_amd import kernels as K
    from tokensgen_amd import lib as L
    from tokensgen_amd import train
    B, T, D, Fr = 2, 40, 128, 4
    x, dy = _rand(B, T, D, seed=71), _rand(B, T, D, seed=72)
    w, bvec = (1 + 0.2 * _rand(D, seed=73).float()).to(BF), _rand(D, seed=74, scale=0.2)
    mod = _rand(B, Fr, 6 * D, seed=75, scale=0.3)
    tg = torch.empty(T, dtype=torch.uint8)
    tg[:8], tg[8:] = Fr, (torch.arange(T - 8) // 8).to(torch.uint8)
    table = K.GroupTable(mod.to(DEV), tg.to(DEV), list(range(Fr)) + [0], [0] * Fr + [3 * D], [D] * Fr + [4 * D], [2 * D] * Fr + [5 * D])
    # fp32 reference with per-token modulation rows
    rows = torch.tensor([0] * 8 + [int(i) for i in tg[8:]])
    base = torch.tensor([3 * D] * 8 + [0] * (T - 8))
    pick = lambda m, col: torch.stack([torch.stack([m[b, rows[t], base[t] + col: base[t] + col + D] for t in range(T)]) for b in range(B)])
    xf, wf, bf_, mf = x.float().requires_grad_(True), w.float().requires_grad_(True), bvec.float().requires_grad_(True), mod.float().requires_grad_(True)
    ln = F.layer_norm(xf, (D,), wf, bf_, 1e-5)
    y = ln * (1 + pick(mf, D)) + pick(mf, 0)
    (y * dy.float()).sum().backward()
    dx = torch.empty(B, T, D, dtype=BF, device=DEV)
    t_dln, t_dlnx, t_dyln = train._adaln_bwd(x.to(DEV), dy.to(DEV), dx, w.to(DEV), bvec.to(DEV), 1e-5, table)
    assert _rel(dx, xf.grad) < 4e-3
    assert _rel(train._colsum_f32(t_dln), bf_.grad) < 1e-5 and _rel(train._colsum_f32(t_dlnx), wf.grad) < 1e-5
    # d scale of frame 1, batch 1: its rows are 16..24
    got = train._colsum_f32(t_dyln.view(B, T, D)[1, 16:24])
    assert _rel(got, mf.grad[1, 1, D:2 * D]) < 4e-3
    # gated residual
    yv, dout = _rand(B, T, D, seed=76), _rand(B, T, D, seed=77)
    mg = mod.float().requires_grad_(True)
    yf = yv.float().requires_grad_(True)
    ((pick(mg, 2 * D) * yf) * dout.float()).sum().backward()
    dyk, tgate = train._gate_res_bwd(dout.to(DEV), yv.to(DEV), table)
    assert _rel(dyk, yf.grad) < 4e-3
    assert _rel(train._colsum_f32(tgate.view(B, T, D)[0, :8]), mg.grad[0, 0, 5 * D:6 * D]) < 1e-5
    dyk2, tg_tail = train._gate_res_bwd(dout.to(DEV), yv.to(DEV), table, row0=24)        # products only for the trailing rows (the vip group in the block)
    assert torch.equal(dyk2, dyk) and tg_tail.shape == (B, T - 24, D) and torch.equal(tg_tail, tgate.view(B, T, D)[:, 24:])
    dyk3, tg_tail3 = train._gate_res_bwd(dout.to(DEV), yv[:, 24:].contiguous().to(DEV), table, row0=24)   # the caller kept only the rows that are read
    assert torch.equal(dyk3, dyk) and torch.equal(tg_tail3, tg_tail)
    # column sums with few rows per block (short matrices) and with 256 (tall ones): same sums as torch, fixed order run to run
    for R_, C_ in ((480, 3072), (40, 128), (5000, 640), (33, 100)):
        m = torch.randn(R_, C_, generator=torch.Generator().manual_seed(R_), dtype=torch.float32)
        got = train._colsum_f32(m.to(DEV))
        assert _rel(got, m.double().sum(0).float()) < 1e-6 and torch.equal(got, train._colsum_f32(m.to(DEV)))
        assert _rel(train.colsum(m.to(BF).to(DEV)), m.to(BF).double().sum(0).float()) < 1e-6
    # the frozen-norm form: no products, the residual gradient summed in the same pass (bf16 + bf16 like autograd on bf16 tensors)
    res = _rand(B, T, D, seed=80)
    dx2 = torch.empty(B, T, D, dtype=BF, device=DEV)
    none3 = train._adaln_bwd(x.to(DEV), dy.to(DEV), dx2, w.to(DEV), bvec.to(DEV), 1e-5, table, products=False, add=res.to(DEV))
    assert none3 == [None, None, None] and torch.equal(dx2, (dx.float() + res.to(DEV).float()).to(BF))
    # activations
    v = _rand(3000, seed=78, scale=2.0)
    g_ = _rand(3000, seed=79)
    vf = v.float().requires_grad_(True)
    (F.gelu(vf, approximate="tanh") * g_.float()).sum().backward()
    assert _rel(train._act(v.to(DEV), g_.to(DEV)), vf.grad) < 4e-3
    assert _rel(train._act(v.to(DEV)), F.silu(v.float())) < 4e-3
    # gelu forward as a pass of its own == the GEMM's GELU epilogue on the stored pre-activation, bit for bit (the training forward relies on it)
    a_, w_, b_ = _rand(2, 300, 256, seed=81), _rand(384, 256, seed=82, scale=0.1), _rand(384, seed=83)
    pre_, fused = torch.empty(2, 300, 384, dtype=BF, device=DEV), torch.empty(2, 300, 384, dtype=BF, device=DEV)
    K.gemm(a_.to(DEV), w_.to(DEV), b_.to(DEV), pre_, L.EPI_BIAS)
    K.gemm(a_.to(DEV), w_.to(DEV), b_.to(DEV), fused, L.EPI_BIAS_GELU)
    assert torch.equal(train._act(pre_, gelu=True), fused)
    assert _rel(train._act(v.to(DEV), gelu=True), F.gelu(v.float(), approximate="tanh")) < 4e-3
    assert torch.equal(train._act(v[:2995].contiguous().to(DEV), gelu=True), train._act(v.to(DEV), gelu=True)[:2995])
    # element counts that are not a multiple of 8 take the scalar form of the kernel: same values
    assert torch.equal(train._act(v[:2995].contiguous().to(DEV), g_[:2995].contiguous().to(DEV)), train._act(v.to(DEV), g_.to(DEV))[:2995])


def test_gemm_activation_epilogues_equal_gemm_plus_act_pass():
    """The training step's two GEMM epilogues (4-wave kernel): TG_EPI_BIAS_KEEP_GELU writes the pre-activation and its GELU in one launch, TG_EPI_BIAS_MUL_GELU_GRAD
    multiplies the dgrad by gelu'(kept pre-activation) — bitwise tg_gemm_bf16(TG_EPI_BIAS) followed by tg_act (modes 2 / 1), ragged last m-tile, two batch items,
    strided outputs; shapes without the 4-wave kernel are refused (the caller keeps the two-step form)."""
    from tokensgen_amd import kernels as K
    from tokensgen_amd import lib as L
    from tokensgen_amd import train
    B, M, Kd, N = 2, 1100, 256, 512
    a_, w_, b_ = _rand(B, M, Kd, seed=181), _rand(N, Kd, seed=182, scale=0.1), _rand(N, seed=183)
    a_, w_, b_ = a_.to(DEV), w_.to(DEV), b_.to(DEV)
    assert K.gemm_act_supported(M, N, Kd) and not K.gemm_act_supported(300, N, Kd)
    pre_ref = torch.empty(B, M, N, dtype=BF, device=DEV)
    K.gemm(a_, w_, b_, pre_ref, L.EPI_BIAS)
    act_ref = train._act(pre_ref, gelu=True)
    big = torch.full((B, M, 2 * N), 9.0, dtype=BF, device=DEV)                   # both outputs as column halves of one buffer (strided rows)
    K.gemm(a_, w_, b_, big[:, :, :N], L.EPI_BIAS_KEEP_GELU, residual=big[:, :, N:])
    assert torch.equal(big[:, :, :N], pre_ref) and torch.equal(big[:, :, N:], act_ref) and float(act_ref.float().abs().sum()) > 0
    # dgrad: dy [M2, N] x W [N, Kin] with the kept pre-activation [M2, Kin]
    M2, Kin = B * M, 768
    dy, w2, pre = _rand(M2, N, seed=184).to(DEV), _rand(N, Kin, seed=185, scale=0.1).to(DEV), _rand(M2, Kin, seed=186, scale=2.0).to(DEV)
    ref = train._act(pre, train.linear_backward_dx(dy, w2).contiguous())
    got = train.linear_backward_dx(dy, w2, gelu_pre=pre)
    assert got.shape == ref.shape and torch.equal(got, ref) and float(ref.float().abs().sum()) > 0
    small = train.linear_backward_dx(dy[:300].contiguous(), w2, gelu_pre=pre[:300].contiguous())        # no 4-wave kernel at M = 300: the two-step form, same values
    assert torch.equal(small, ref[:300])
    with pytest.raises(RuntimeError):
        K.gemm(a_[:, :300], w_, b_, pre_ref[:, :300], L.EPI_BIAS_KEEP_GELU, residual=act_ref[:, :300])


def test_to2v_block_backward_vs_autograd_of_the_oracle_block():
    """one whole CogVideoXBlock with the vip branch.  HIP forward (the product kernels, intermediates kept) + HIP backward
    (train.To2VBlockTrainer) against torch.autograd through oracle.dit_ref.block_forward in fp32 on the same bf16-rounded weights and inputs:
    the gradient of every trainable parameter of the block (vip_norm1 / vip_norm2 linear + LayerNorm, processor.vip_to_{q,k,v},
    processor.vip_norm_{q,k}; train_cogvideo_to2v.py:1456-1481) and the gradients handed to the previous block (hidden, text | vip rows)."""
    import numpy as np
    from oracle import dit_ref as O
    from tokensgen_amd import train
    B, H, Nt, Fr, hw, Np = 2, 2, 9, 4, 30, 30
    Nv, D = Fr * hw, H * 64
    f32 = np.float32
    cfg = dict(num_attention_heads=H, attention_head_dim=64, num_layers=1, patch_size=2, time_embed_dim=128, text_embed_dim=64, in_channels=16, out_channels=16)
    pre = "transformer_blocks.0"
    sd = {k: v.to(BF).float() for k, v in O.make_state_dict(cfg, n_vip_dim=128, seed=81, std=0.08).items() if k.startswith(pre + ".")}
    train_keys = [k for k in sd if "vip_" in k]
    assert len(train_keys) == 8 + 10
    for k in train_keys:
        sd[k] = sd[k].clone().requires_grad_(True)
    hidden, enc, temb = _rand(B, Nv, D, seed=82), _rand(B, Nt + Np, D, seed=83), _rand(B, Fr, 128, seed=84)
    rope = O.rope_3d(64, np.arange(4, dtype=f32), np.arange(5, dtype=f32), np.arange(6, dtype=f32))
    vrope = O.rope_3d(64, np.arange(4, dtype=f32) + f32(3), np.arange(5, dtype=f32), np.arange(6, dtype=f32))
    crope = O.rope_3d(64, np.linspace(1000, 1016.25, 5, dtype=f32), np.arange(2, dtype=f32), np.arange(3, dtype=f32))
    hf, ef = hidden.float().requires_grad_(True), enc.float().requires_grad_(True)
    oh, oe = O.block_forward(sd, pre, hf, ef, temb.float(), H, Np, [0.6], rope, vrope, crope)
    Gh, Ge = _rand(B, Nv, D, seed=85), _rand(B, Nt + Np, D, seed=86)
    ((oh * Gh.float()).sum() + (oe * Ge.float()).sum()).backward()
    sd_dev = {k: v.detach().to(BF).to(DEV).contiguous() for k, v in sd.items()}
    blk = train.To2VBlockTrainer(sd_dev, pre, H, Nt, Np, Fr, 0.6)
    gh, ge = blk.forward(hidden.to(DEV), enc.to(DEV), temb.to(DEV), rope, vrope, crope)
    assert _rel(gh, oh.detach()) < 6e-3 and _rel(ge, oe.detach()) < 6e-3
    grads, dh, de = blk.backward(Gh.to(DEV), Ge.to(DEV))
    assert set(pre + "." + k for k in grads) == set(train_keys)
    assert _rel(dh, hf.grad) < 7e-3 and _rel(de, ef.grad) < 7e-3
    for name, g_ in grads.items():
        assert _rel(g_, sd[pre + "." + name].grad) < 2e-2, name


def test_to2v_model_training_forward_backward_vs_autograd_of_the_oracle_model(parity):
    """The transformer's share of one training micro-step (train_cogvideo_to2v.py:1930-2010): embeddings -> 2 blocks with per-block checkpointing
    -> final norms / proj_out -> v-prediction loss, then backward to EVERY trainable transformer parameter (all names containing "vip_", incl.
    patch_embed.vip_proj) and to the vip tokens (the Resampler's output).  Against autograd through oracle.dit_ref.dit_forward +
    oracle.train_ref.vpred_loss in fp32 on the same bf16-rounded weights and inputs."""
    import numpy as np
    from oracle import dit_ref as O
    from oracle import scheduler_ref as S
    from oracle import train_ref as T
    from tokensgen_amd import train
    B, H, Nt, Fr, Hh, Ww = 2, 2, 9, 4, 10, 12
    f32 = np.float32
    cfg = dict(num_attention_heads=H, attention_head_dim=64, num_layers=2, patch_size=2, time_embed_dim=128, text_embed_dim=64, in_channels=16, out_channels=16)
    sd = {k: v.to(BF).float() for k, v in O.make_state_dict(cfg, n_vip_dim=128, seed=91, std=0.08).items()}
    train_keys = sorted(k for k in sd if "vip_" in k)
    assert len(train_keys) == 2 * 18 + 2
    for k in train_keys:
        sd[k] = sd[k].clone().requires_grad_(True)
    _, ac = S.alphas_cumprod()
    ac = torch.as_tensor(ac, dtype=torch.float32)
    g = torch.Generator().manual_seed(92)
    noisy, x0 = (torch.randn(B, Fr, 16, Hh, Ww, generator=g).to(BF) for _ in range(2))
    text = _rand(B, Nt, 64, seed=93)
    vip = _rand(B, 5, 128, 2, 3, seed=94)
    ts = torch.randint(20, 980, (B, Fr), generator=g)
    rope = O.rope_3d(64, np.arange(4, dtype=f32), np.arange(5, dtype=f32), np.arange(6, dtype=f32))
    vrope = O.rope_3d(64, np.arange(4, dtype=f32) + f32(3), np.arange(5, dtype=f32), np.arange(6, dtype=f32))
    crope = O.rope_3d(64, np.linspace(1000, 1016.25, 5, dtype=f32), np.arange(2, dtype=f32), np.arange(3, dtype=f32))
    vf = vip.float().requires_grad_(True)
    out_ref = O.dit_forward(sd, cfg, noisy.float(), text.float(), ts, vf, rope, vrope, crope, vip_scale=[0.7])
    loss_ref, _ = T.vpred_loss(ac, out_ref, noisy.float(), x0.float(), ts)
    loss_ref.backward()
    # the reference's own arithmetic: the same chain with bf16 parameters and activations (torch autograd rounds every op) — its distance to the
    # fp32 gradients is the noise floor the HIP figures below are read against
    sd16 = {k: v.detach().to(BF).requires_grad_(k in train_keys) for k, v in sd.items()}
    v16 = vip.clone().requires_grad_(True)
    loss16, _ = T.vpred_loss(ac, O.dit_forward(sd16, cfg, noisy, text, ts, v16, rope, vrope, crope, vip_scale=[0.7]), noisy, x0, ts)
    loss16.backward()
    floor = max(float(_rel(sd16[k].grad, sd[k].grad)) for k in train_keys)
    parity(floor, 1.0, "noise floor: bf16 autograd of the oracle vs fp32 autograd, worst trainable tensor (informative)")
    parity(_rel(v16.grad, vf.grad), 1.0, "noise floor of d(vip tokens) (informative)")
    sd_dev = {k: v.detach().to(BF).to(DEV).contiguous() for k, v in sd.items()}
    tr = train.To2VTrainer(sd_dev, H, 2, patch_size=2, vip_scale=0.7)
    assert tr.trainable == train_keys
    out = tr.forward(noisy.to(DEV), text.to(DEV), ts, vip.to(DEV), rope, vrope, crope)
    assert _rel(out, out_ref.detach()) < 1e-2
    loss, _, d_out = train.vpred_loss_and_grad(out, noisy.to(DEV), x0.to(DEV), ts, ac)
    assert abs(loss.item() - loss_ref.item()) < 2e-2 * abs(loss_ref.item())
    grads, d_vip = tr.backward(d_out)
    assert sorted(grads) == train_keys
    want_vip = vf.grad.permute(0, 1, 3, 4, 2).reshape(B, -1, 128)
    parity(_rel(d_vip, want_vip), 2e-2, "d(vip tokens), HIP vs fp32 autograd (r2 measured 1.0e-2)")
    worst = max(float(_rel(grads[name], sd[name].grad)) for name in train_keys)
    parity(worst, 3.6e-2, "worst trainable tensor, HIP vs fp32 autograd (r2 measured 1.8e-2; compare the floor above)")
    # the two activation schedules are the same computation: blocks that keep their activations (288 GB: the default when memory allows) and blocks
    # that re-run their forward inside the backward (the reference's per-block checkpointing) give bitwise the same gradients
    assert not tr._kept and not tr._ckpt
    tr.activation_budget_bytes = 0
    out0 = tr.forward(noisy.to(DEV), text.to(DEV), ts, vip.to(DEV), rope, vrope, crope)
    assert not tr._kept and torch.equal(out0, out)
    grads0, d_vip0 = tr.backward(d_out)
    assert torch.equal(d_vip0, d_vip) and all(torch.equal(grads0[k], grads[k]) for k in train_keys)
    tr.activation_budget_bytes = None
    tr.forward(noisy.to(DEV), text.to(DEV), ts, vip.to(DEV), rope, vrope, crope)
    assert sorted(tr._kept) == [0, 1]


def test_optimizer_step_vs_torch_adamw_with_clipping():
    """optim.ParamArena + optim.AdamW (tg_grad_accumulate / tg_grad_clip_coef / tg_adamw_step) against torch.optim.AdamW + clip_grad_norm_ on fp32
    copies (train_cogvideo_to2v.py:1091-1098, 2012-2021; yaml betas 0.9 / 0.95, eps 1e-8, weight decay 1e-4, max_grad_norm 1.0): three steps with two
    accumulated micro-gradients each; the clipped prefix (the transformer's parameters) and the unclipped rest (the Resampler) as in the reference."""
    from tokensgen_amd import optim
    g = torch.Generator().manual_seed(11)
    shapes = {"transformer_blocks.0.a.vip_w": (37, 50), "transformer_blocks.0.a.vip_b": (129,), "patch_embed.vip_proj.weight": (64, 64), "resampler.latents": (1, 7, 33)}
    params = {k: (torch.randn(*s, generator=g) * 0.1).to(BF) for k, s in shapes.items()}
    order = optim.arena_order(list(params), 1)
    arena = optim.ParamArena({k: v.to(DEV) for k, v in params.items()}, order, DEV)
    n_clip = arena.prefix_elems(lambda n: not n.startswith("resampler."))
    assert 0 < n_clip < arena.numel and arena.param.data_ptr() % 128 == 0
    lr = 3e-2                                                # large enough that a bf16 parameter moves every step
    opt = optim.AdamW(arena, lr=lr, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-4, max_grad_norm=1.0, clip_elems=n_clip)
    ref = {k: torch.nn.Parameter(v.float().clone()) for k, v in params.items()}
    topt = torch.optim.AdamW(list(ref.values()), lr=lr, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-4)
    for step in range(3):
        g1 = {k: (torch.randn(*s, generator=g) * (3.0 if step == 1 else 0.005)).to(BF) for k, s in shapes.items()}     # step 1 clips, the others do not
        g2 = {k: torch.randn(*s, generator=g) * 0.005 for k, s in shapes.items()}                                      # fp32 gradients are accepted too
        arena.accumulate({k: v.to(DEV) for k, v in g1.items()}, 0.5)
        arena.accumulate({k: v.to(DEV) for k, v in g2.items()}, 0.5)
        for k in ref:
            ref[k].grad = 0.5 * g1[k].float() + 0.5 * g2[k]
        assert _rel(arena.grad_view("resampler.latents"), ref["resampler.latents"].grad) < 1e-6
        norm = torch.nn.utils.clip_grad_norm_([ref[k] for k in ref if not k.startswith("resampler.")], 1.0)
        topt.step()
        opt.step()
        assert abs(opt.coef[0].item() - norm.item()) < 1e-4 * norm.item()
        assert (opt.coef[1].item() < 1.0) == (step == 1)
        assert float(arena.grad.abs().max()) == 0.0                          # zero_grad in the same pass
        for k in ref:
            got, want = arena.views[k].float().cpu(), ref[k].detach().to(BF).float()
            ulp = want.abs().clamp_min(1e-30).log2().floor().exp2() * 2.0 ** -7
            assert ((got - want).abs() <= ulp).all(), (k, step)              # at most one bf16 ulp (fp32 operation order)
            assert measured(((got != want).float().mean()).item()) < 2e-2, (k, step)
            ref[k].data.copy_(got)                                           # the reference continues from the bf16 parameters, moments carry over


def test_multi_item_launches_of_the_training_step_are_exact():
    """Round-5 launch diet (train_cogvideo_to2v.py:1995-2021 has one autograd graph instead): tg_grad_accumulate_multi over MORE items than one kernel
    argument table holds (chunks of TG_ACCUM_MAX, bf16 and fp32 sources of very different sizes) equals the per-item tg_grad_accumulate bit for bit, and
    tg_colsum_multi (several matrices of different heights and dtypes, one launch + one fixed-order sum) matches fp64 column sums and is run-to-run bitwise."""
    from tokensgen_amd import lib as L, optim, train
    g = torch.Generator().manual_seed(21)
    n_items = L.TG_ACCUM_MAX + 7
    shapes = {f"transformer_blocks.0.p{i:02d}": ((3 + 5 * i, 17) if i % 3 else (4099 * (1 + i % 2),)) for i in range(n_items)}
    params = {k: torch.zeros(*s).to(BF) for k, s in shapes.items()}
    arena = optim.ParamArena({k: v.to(DEV) for k, v in params.items()}, optim.arena_order(list(params), 1), DEV)
    grads = {k: (torch.randn(*s, generator=g).to(BF) if i % 2 else torch.randn(*s, generator=g)).to(DEV) for i, (k, s) in enumerate(shapes.items())}
    arena.accumulate(grads, 0.25)
    arena.accumulate(grads, 0.5)
    lib = L.load()
    want = torch.zeros_like(arena.grad)
    for sc in (0.25, 0.5):
        for k, gk in grads.items():
            L.check(lib.tg_grad_accumulate(gk.data_ptr(), 1 if gk.dtype == BF else 0, want.data_ptr() + 4 * arena.offsets[k], gk.numel(), sc, 0,
                                           torch.cuda.current_stream().cuda_stream), "tg_grad_accumulate")
    assert torch.equal(arena.grad, want)
    for k, gk in grads.items():
        assert _rel(arena.grad_view(k), 0.75 * gk.float()) < 1e-6, k
    mats = [torch.randn(960, 384, generator=g).to(DEV), torch.randn(960, 384, generator=g).to(BF).to(DEV), torch.randn(2, 480, 512, generator=g).to(BF).to(DEV)[1],
            torch.randn(480, 384, generator=g).to(DEV)[:, :130], torch.randn(7, 33, generator=g).to(DEV)]
    s1, s2 = train.colsum_multi(mats), train.colsum_multi(mats)
    for a, b, m in zip(s1, s2, mats):
        assert torch.equal(a, b) and a.shape == (m.shape[1],) and a.dtype == torch.float32
        assert _rel(a, m.double().sum(dim=0)) < 2e-6


def test_training_steps_reduce_the_loss_and_update_only_trainable_parameters(tmp_path):
    """train.To2VTrainStep end to end on a 2-layer model: three optimizer steps of two micro-steps each on one fixed batch — the loss falls, the
    arena-backed views are what the next forward reads (the fused vip_to_qkv weight is a view, not a stale copy), frozen tensors are untouched."""
    import numpy as np
    from oracle import dit_ref as O
    from oracle import scheduler_ref as S
    from tokensgen_amd import optim, train
    B, H, Nt, Fr, Hh, Ww = 1, 2, 9, 4, 10, 12
    f32 = np.float32
    cfg = dict(num_attention_heads=H, attention_head_dim=64, num_layers=2, patch_size=2, time_embed_dim=128, text_embed_dim=64, in_channels=16, out_channels=16)
    sd = {k: v.to(BF).to(DEV).contiguous() for k, v in O.make_state_dict(cfg, n_vip_dim=128, seed=95, std=0.08).items()}
    frozen_before = {k: v.clone() for k, v in sd.items() if "vip_" not in k}
    tr = train.To2VTrainer(sd, H, 2, patch_size=2, vip_scale=1.0)
    arena = optim.ParamArena({k: sd[k] for k in tr.trainable}, optim.arena_order(tr.trainable, 2), DEV)
    tr.use_arena(arena)
    start = arena.param.clone()
    opt = optim.AdamW(arena, lr=2e-3, max_grad_norm=1.0)
    _, ac = S.alphas_cumprod()
    ac = torch.as_tensor(ac, dtype=torch.float32)
    step = train.To2VTrainStep(tr, arena, opt, ac, accumulation_steps=2)
    g = torch.Generator().manual_seed(96)
    x0, noise = (torch.randn(B, Fr, 16, Hh, Ww, generator=g).to(BF).to(DEV) for _ in range(2))
    text, vip = _rand(B, Nt, 64, seed=97).to(DEV), _rand(B, 5, 128, 2, 3, seed=98).to(DEV)
    ts = torch.tensor([[500, 520, 480, 510]])
    rope = O.rope_3d(64, np.arange(4, dtype=f32), np.arange(5, dtype=f32), np.arange(6, dtype=f32))
    vrope = O.rope_3d(64, np.arange(4, dtype=f32) + f32(3), np.arange(5, dtype=f32), np.arange(6, dtype=f32))
    crope = O.rope_3d(64, np.linspace(1000, 1016.25, 5, dtype=f32), np.arange(2, dtype=f32), np.arange(3, dtype=f32))
    losses, stepped = [], []
    for i in range(8):
        loss, did = step.micro_step(x0, noise, ts, text, vip, rope, vrope, crope)
        losses.append(loss.item()); stepped.append(did)
    assert stepped == [False, True] * 4 and opt.t == 4
    assert losses[0] == losses[1]                                             # same batch, no step in between: bitwise the same loss
    assert losses[2] < losses[0] and losses[6] < losses[2], losses
    assert (arena.param != start).float().mean().item() > 0.5                 # the parameters moved
    blk = tr._blocks[0]
    assert blk.Wv.data_ptr() == arena.views["transformer_blocks.0.attn1.processor.vip_to_q.weight"].data_ptr()
    for k, v in frozen_before.items():
        assert torch.equal(sd[k], v), k
    # the reference's save hook -> the inference loading contract: vip.pt written by the trainer is what set_vip_layers loads, and the
    # inference transformer with the TRAINED vip weights reproduces the trainer's forward
    from tokensgen_amd.transformer import CogVideoXTransformer3DModel
    tr.save_vip_layers(str(tmp_path))
    saved = torch.load(str(tmp_path / "vip.pt"), weights_only=True)
    assert sorted(saved) == tr.trainable and all(v.dtype == torch.float32 and v.device.type == "cpu" for v in saved.values())
    m = CogVideoXTransformer3DModel(num_attention_heads=H, attention_head_dim=64, num_layers=2, time_embed_dim=128, text_embed_dim=64,
                                    use_rotary_positional_embeddings=True, device=DEV)
    m.load_state_dict({k: v for k, v in frozen_before.items()}, strict=False)
    m.set_vip_layers(str(tmp_path), length=30, func_type="1", scale=[1.0],
                     resampler_params=dict(output_dim=128, num_height_queries=2, num_width_queries=3, num_temporal_queries=4))
    noisy = step.add_noise(x0, noise, ts).contiguous()
    y_inf = m(noisy, text, ts.to(DEV), vip_encoder_hidden_states=vip, image_rotary_emb=rope, vip_image_rotary_emb=vrope, vip_condition_rotary_emb=crope,
              return_dict=False)[0]
    y_tr = tr.forward(noisy, text, ts, vip, rope, vrope, crope)
    assert _rel(y_inf, y_tr) < 1e-2


def test_resampler_backward_vs_autograd_of_the_oracle(parity):
    """The Resampler is trainable as a whole (train_cogvideo_to2v.py:1479-1481): train.ResamplerTrainer forward + backward (LayerNorms, to_q / to_kv /
    to_out, per-head QK-norm + the two rotary segments, the 24-query attention over 96 keys, FeedForward, proj_in / proj_out / norm_out, latents) against
    autograd through oracle.resampler_ref.resampler_forward in fp32 on the same bf16-rounded weights — every parameter of a depth-2 module."""
    import numpy as np
    from oracle import dit_ref as O
    from oracle import resampler_ref as RR
    from tokensgen_amd import train
    cfg = dict(dim=128, depth=2, dim_head=64, heads=2, num_height_queries=2, num_width_queries=3, num_temporal_queries=4, embedding_dim=128, output_dim=128, ff_mult=4)
    sd = {k: v.to(BF).float().requires_grad_(True) for k, v in RR.make_state_dict(cfg, seed=61).items()}
    b = 2
    x = _rand(b, 3, 24, 128, seed=62)
    f32 = np.float32
    img = O.rope_3d(64, np.arange(3, dtype=f32), np.arange(4, dtype=f32), np.arange(6, dtype=f32))
    smp = O.rope_3d(64, np.linspace(0, 3, 4, endpoint=False, dtype=f32), np.linspace(0, 4, 2, endpoint=False, dtype=f32), np.linspace(0, 6, 3, endpoint=False, dtype=f32))
    out_ref = RR.resampler_forward(sd, cfg, x.float(), img, smp)
    tok_ref = out_ref.permute(0, 1, 3, 4, 2).reshape(b, 24, 128)
    G = _rand(b, 24, 128, seed=63)
    (tok_ref * G.float()).sum().backward()
    sd16 = {k: v.detach().to(BF).requires_grad_(True) for k, v in sd.items()}                 # noise floor: the same chain in bf16 autograd
    (RR.resampler_forward(sd16, cfg, x, img, smp).permute(0, 1, 3, 4, 2).reshape(b, 24, 128) * G).sum().backward()
    parity(max(float(_rel(sd16[k].grad, sd[k].grad)) for k in sd), 1.0, "noise floor: bf16 autograd of the oracle vs fp32 autograd, worst tensor (informative)")
    sd_dev = {k: v.detach().to(BF).to(DEV).contiguous() for k, v in sd.items()}
    rt = train.ResamplerTrainer(sd_dev, depth=2, heads=2)
    tok, ctx = rt.forward(x.to(DEV), img, smp)
    assert _rel(tok, tok_ref.detach()) < 1e-2
    grads = rt.backward(ctx, G.to(DEV))
    assert sorted(grads) == sorted("resampler." + k for k in sd)
    parity(max(float(_rel(grads["resampler." + k], sd[k].grad)) for k in sd), 4e-2, "worst Resampler tensor, HIP vs fp32 autograd (r2 measured 2.3e-2; compare the floor above)")


def test_full_micro_step_with_resampler_gradients_vs_autograd_of_the_oracle_chain(parity):
    """One whole micro-step of the reference loop body (train_cogvideo_to2v.py:1931-2010) on the HIP path: Resampler over two chunks -> five temporal
    slots per batch item -> add_noise -> transformer (checkpointed) -> v-prediction loss -> backward through the transformer AND both Resampler calls,
    accumulated into the gradient arena.  Every trainable tensor (38 transformer + 38 Resampler) against autograd through the oracle chain
    (resampler_ref -> dit_ref -> train_ref) in fp32 on the same bf16-rounded weights / inputs / noise."""
    import numpy as np
    from oracle import dit_ref as O
    from oracle import resampler_ref as RR
    from oracle import scheduler_ref as S
    from oracle import train_ref as T
    from tokensgen_amd import optim, train
    B, H, Nt, Fr, Hh, Ww = 2, 2, 9, 4, 10, 12
    f32 = np.float32
    cfg = dict(num_attention_heads=H, attention_head_dim=64, num_layers=2, patch_size=2, time_embed_dim=128, text_embed_dim=64, in_channels=16, out_channels=16)
    rcfg = dict(dim=128, depth=2, dim_head=64, heads=2, num_height_queries=2, num_width_queries=3, num_temporal_queries=4, embedding_dim=128, output_dim=128, ff_mult=4)
    sd = {k: v.to(BF).float() for k, v in O.make_state_dict(cfg, n_vip_dim=128, seed=101, std=0.08).items()}
    rsd = {k: v.to(BF).float().requires_grad_(True) for k, v in RR.make_state_dict(rcfg, seed=102).items()}
    tkeys = sorted(k for k in sd if "vip_" in k)
    for k in tkeys:
        sd[k] = sd[k].clone().requires_grad_(True)
    _, ac = S.alphas_cumprod()
    ac = torch.as_tensor(ac, dtype=torch.float32)
    g = torch.Generator().manual_seed(103)
    x0, noise = (torch.randn(B, Fr, 16, Hh, Ww, generator=g).to(BF) for _ in range(2))
    text, emb = _rand(B, Nt, 64, seed=104), _rand(B, 6, 24, 128, seed=105)          # 2 chunks x 3 latent frames x 24 tokens
    ts = torch.randint(20, 980, (B, Fr), generator=g)
    start = [1, 3]
    rope = O.rope_3d(64, np.arange(4, dtype=f32), np.arange(5, dtype=f32), np.arange(6, dtype=f32))
    vrope = O.rope_3d(64, np.arange(4, dtype=f32) + f32(3), np.arange(5, dtype=f32), np.arange(6, dtype=f32))
    crope = O.rope_3d(64, np.linspace(1000, 1016.25, 5, dtype=f32), np.arange(2, dtype=f32), np.arange(3, dtype=f32))
    img = O.rope_3d(64, np.arange(3, dtype=f32), np.arange(4, dtype=f32), np.arange(6, dtype=f32))
    smp = O.rope_3d(64, np.linspace(0, 3, 4, endpoint=False, dtype=f32), np.linspace(0, 4, 2, endpoint=False, dtype=f32), np.linspace(0, 6, 3, endpoint=False, dtype=f32))
    # ---- oracle chain ----
    acb = ac.to(BF)
    sa, sb = (acb[ts] ** 0.5)[..., None, None, None], ((1 - acb[ts]) ** 0.5)[..., None, None, None]
    noisy = (sa * x0 + sb * noise)                                                    # scheduler.add_noise in bf16

    def chain(sd_, rsd_, cast):
        toks = torch.cat([RR.resampler_forward(rsd_, rcfg, cast(emb[:, c * 3:(c + 1) * 3]), img, smp) for c in range(2)], dim=1)   # [B, 8, 128, 2, 3]
        toks = cast(toks.to(BF))                                                      # the Resampler's output is a bf16 tensor (the cast passes gradients through)
        vip = torch.stack([toks[b, start[b]:start[b] + 5] for b in range(B)])
        out = O.dit_forward(sd_, cfg, cast(noisy), cast(text), ts, vip, rope, vrope, crope, vip_scale=[1.0])
        return T.vpred_loss(ac, out, cast(noisy), cast(x0), ts)[0]
    loss_ref = chain(sd, rsd, lambda t: t.float())
    loss_ref.backward()
    # noise floor: the same chain in bf16 autograd (the reference trains in bf16)
    sd16 = {k: v.detach().to(BF).requires_grad_(k in tkeys) for k, v in sd.items()}
    rsd16 = {k: v.detach().to(BF).requires_grad_(True) for k, v in rsd.items()}
    chain(sd16, rsd16, lambda t: t).backward()
    parity(max([float(_rel(sd16[k].grad, sd[k].grad)) for k in tkeys] + [float(_rel(rsd16[k].grad, rsd[k].grad)) for k in rsd]), 1.0,
           "noise floor: bf16 autograd of the oracle chain vs fp32 autograd, worst tensor (informative)")
    # ---- HIP path ----
    sd_dev = {k: v.detach().to(BF).to(DEV).contiguous() for k, v in sd.items()}
    rsd_dev = {k: v.detach().to(BF).to(DEV).contiguous() for k, v in rsd.items()}
    tr = train.To2VTrainer(sd_dev, H, 2, patch_size=2, vip_scale=1.0)
    rt = train.ResamplerTrainer(rsd_dev, depth=2, heads=2)
    params = {k: sd_dev[k] for k in tr.trainable}
    params.update({"resampler." + k: v for k, v in rsd_dev.items()})
    arena = optim.ParamArena(params, optim.arena_order(list(params), 2), DEV)
    tr.use_arena(arena); rt.use_arena(arena)
    n_clip = arena.prefix_elems(lambda n: not n.startswith("resampler."))
    opt = optim.AdamW(arena, lr=1e-3, clip_elems=n_clip)
    step = train.To2VTrainStep(tr, arena, opt, ac, accumulation_steps=2, resampler=rt)
    step.latent_frames_per_chunk = 3
    loss, did = step.micro_step(x0.to(DEV), noise.to(DEV), ts, text.to(DEV), None, rope, vrope, crope, image_embeddings=emb.to(DEV), emb_start_idx=start,
                                resampler_ropes=(img, smp), vip_frames=5)
    assert not did and abs(loss.item() - loss_ref.item()) < 2e-2 * abs(loss_ref.item())
    worst = max([float(_rel(arena.grad_view(k) * 2.0, sd[k].grad)) for k in tkeys] +                # the arena holds grad / accumulation_steps
                [float(_rel(arena.grad_view("resampler." + k) * 2.0, rsd[k].grad)) for k in rsd])
    parity(worst, 5e-2, "worst of the 76 trainable tensors, HIP micro-step vs fp32 autograd (r2 measured 3.7e-2; compare the floor above)")
    before = arena.param.clone()
    _, did = step.micro_step(x0.to(DEV), noise.to(DEV), ts, text.to(DEV), None, rope, vrope, crope, image_embeddings=emb.to(DEV), emb_start_idx=start,
                             resampler_ropes=(img, smp), vip_frames=5)
    assert did and opt.t == 1 and float(arena.grad.abs().max()) == 0.0
    moved = (arena.param != before)
    assert moved[:n_clip].float().mean().item() > 0.3 and moved[n_clip:].float().mean().item() > 0.3      # transformer and Resampler parameters both stepped


@pytest.mark.timeout(900)
def test_full_width_block_backward_vs_autograd():
    """The block backward at the REAL width (48 heads x 64 = 3072 channels, FeedForward 12288: the 256^2 GEMM tiles, the 8-wave backward attention
    kernels, the 16-byte elementwise kernels) on a short stream (16 text + 2 x 192 video + 64 vip tokens), against autograd through the oracle block
    in fp32 — run on the GPU for speed, it is still the checker — on the same bf16-rounded weights and inputs."""
    import numpy as np
    from oracle import dit_ref as O
    from tokensgen_amd import train
    B, H, Nt, Fr, hw, Np = 1, 48, 16, 2, 192, 64
    Nv, D = Fr * hw, H * 64
    f32 = np.float32
    cfg = dict(num_attention_heads=H, attention_head_dim=64, num_layers=1, patch_size=2, time_embed_dim=512, text_embed_dim=64, in_channels=16, out_channels=16)
    pre = "transformer_blocks.0"
    sd = {k: v.to(BF).float().to(DEV) for k, v in O.make_state_dict(cfg, n_vip_dim=128, seed=111, std=0.02).items() if k.startswith(pre + ".")}
    train_keys = [k for k in sd if "vip_" in k]
    for k in train_keys:
        sd[k] = sd[k].clone().requires_grad_(True)
    hidden, enc, temb = _rand(B, Nv, D, seed=112).to(DEV), _rand(B, Nt + Np, D, seed=113).to(DEV), _rand(B, Fr, 512, seed=114).to(DEV)
    dev = lambda r: tuple(t.to(DEV) for t in r)
    rope = O.rope_3d(64, np.arange(2, dtype=f32), np.arange(12, dtype=f32), np.arange(16, dtype=f32))
    vrope = O.rope_3d(64, np.arange(2, dtype=f32) + f32(3), np.arange(12, dtype=f32), np.arange(16, dtype=f32))
    crope = O.rope_3d(64, np.linspace(1000, 1016.25, 4, dtype=f32), np.arange(4, dtype=f32), np.arange(4, dtype=f32))
    hf, ef = hidden.float().requires_grad_(True), enc.float().requires_grad_(True)
    oh, oe = O.block_forward(sd, pre, hf, ef, temb.float(), H, Np, [1.0], dev(rope), dev(vrope), dev(crope))
    Gh, Ge = _rand(B, Nv, D, seed=115).to(DEV), _rand(B, Nt + Np, D, seed=116).to(DEV)
    ((oh * Gh.float()).sum() + (oe * Ge.float()).sum()).backward()
    sd_dev = {k: v.detach().to(BF).contiguous() for k, v in sd.items()}
    blk = train.To2VBlockTrainer(sd_dev, pre, H, Nt, Np, Fr, 1.0)
    gh, ge = blk.forward(hidden, enc, temb, rope, vrope, crope)
    assert _rel(gh, oh.detach()) < 1e-2 and _rel(ge, oe.detach()) < 1e-2
    grads, dh, de = blk.backward(Gh, Ge)
    assert _rel(dh, hf.grad) < 1.3e-2 and _rel(de, ef.grad) < 1.3e-2          # measured 5.5e-3 / 6.3e-3
    for name, g_ in grads.items():
        assert _rel(g_, sd[pre + "." + name].grad) < 3.5e-2, name               # measured <= 1.7e-2
