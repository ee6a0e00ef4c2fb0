"""torch-tensor front end of the C ABI (include/tokensgen_hip.h).  Plumbing only: device memory comes from
torch, launches go on torch's current HIP stream.  Every function requires bf16 CUDA(HIP) tensors; there is
no CPU path."""
import ctypes as C
import os

import torch

from . import lib as L

BF16 = torch.bfloat16


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name, dtype=BF16):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (tokensgen_amd has no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost dimension must be contiguous")
    return t


def _p(t):
    return 0 if t is None else t.data_ptr()


# Optional per-launch timing with HIP events recorded on the launch stream (bench.py's roofline leg).
PROFILE_ON = [False]
PROFILE = {}


PROFILE_FILTER = [None]      # None: time every launch; a set of names: only those (fewer event markers in a timed region)


def _launch(name, fn, *args):
    if not PROFILE_ON[0] or (PROFILE_FILTER[0] is not None and name not in PROFILE_FILTER[0]):
        return fn(*args)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    code = fn(*args)
    e.record()
    PROFILE.setdefault(name, []).append((s, e))
    return code


def profile_summary():
    """{kernel name: {"ms": mean launch duration, "n": launches, "total_ms": sum}} from the recorded events."""
    torch.cuda.synchronize()
    out = {}
    for name, evs in PROFILE.items():
        ms = [s.elapsed_time(e) for s, e in evs]
        out[name] = {"ms": sum(ms) / len(ms), "n": len(ms), "total_ms": sum(ms)}
    return out


class GroupTable:
    """Host mirror of tg_group_table: token -> (modulation row, shift/scale/gate column)."""

    def __init__(self, mod, tok_group, rows, shift_cols, scale_cols, gate_cols, tok_offset=0):
        _chk(mod, "mod")
        assert mod.dim() == 3, "mod must be [batch, rows, cols]"
        _chk(tok_group, "tok_group", torch.uint8)
        self.mod, self.tok_group = mod, tok_group
        self.c = L.GroupTable()
        self.c.mod = mod.data_ptr()
        self.c.mod_ld = mod.stride(1)
        self.c.mod_batch_stride = mod.stride(0)
        self.c.tok_group = tok_group.data_ptr() + tok_offset
        n = len(rows)
        assert n <= L.TG_MAX_GROUPS
        for i in range(n):
            self.c.row[i], self.c.shift_col[i], self.c.scale_col[i], self.c.gate_col[i] = (
                int(rows[i]), int(shift_cols[i]), int(scale_cols[i]), int(gate_cols[i]))
        self.rows, self.shift_cols, self.scale_cols, self.gate_cols = list(rows), list(shift_cols), list(scale_cols), list(gate_cols)

    def offset(self, tok_offset):
        """Same table, token indices shifted (for launches that start at a later token)."""
        return GroupTable(self.mod, self.tok_group, self.rows, self.shift_cols, self.scale_cols, self.gate_cols, tok_offset)

    def ref(self):
        return C.byref(self.c)


def _bmk(t):
    """View a [M,K] or [B,M,K] tensor as (batch, M, K, ld, batch_stride)."""
    if t.dim() == 2:
        return 1, t.shape[0], t.shape[1], t.stride(0), 0
    if t.dim() == 3:
        return t.shape[0], t.shape[1], t.shape[2], t.stride(1), t.stride(0)
    raise ValueError("expected a 2-D or 3-D tensor")


def gemm(a, w, bias, out, epilogue=L.EPI_BIAS, residual=None, gate=None):
    """out = epilogue(a @ w.T + bias); a [..,M,K], w [N,K] (nn.Linear layout), out [..,M,N] (views allowed)."""
    _chk(a, "a"); _chk(w, "w"); _chk(out, "out")
    B, M, K, lda, sa = _bmk(a)
    Bo, Mo, N, ldc, sc = _bmk(out)
    if (B, M) != (Bo, Mo) or w.shape != (N, K):
        raise ValueError(f"gemm: shape mismatch a{tuple(a.shape)} w{tuple(w.shape)} out{tuple(out.shape)}")
    if bias is not None:
        _chk(bias, "bias")
    ldr = sr = 0
    if residual is not None:
        _chk(residual, "residual")
        _, _, _, ldr, sr = _bmk(residual)
    L.check(_launch(f"gemm_M{M}_N{N}_K{K}_epi{epilogue}", L.load().tg_gemm_bf16, _p(a), lda, sa, _p(w), w.stride(0), _p(bias), _p(out), ldc, sc, M, N, K, B, epilogue,
                                  _p(residual), ldr, sr, gate.ref() if gate is not None else None, _stream()), "tg_gemm_bf16")
    return out


def gemm_act_supported(M, N, K, lda=0, ldw=0):
    """Shapes that have the training step's activation epilogues (EPI_BIAS_KEEP_GELU: `residual` is the second OUTPUT, gelu of the kept pre-activation;
    EPI_BIAS_MUL_GELU_GRAD: `residual` is the kept pre-activation): the 4-wave GEMM kernel's.  Elsewhere: gemm(EPI_BIAS) + tg_act."""
    return (M >= 1024 and N % 256 == 0 and K % 64 == 0 and K >= 256 and max(lda, ldw, K) < (1 << 21)      # csrc/gemm.hip: 32-bit buffer offsets in the 4-wave kernel
            and L.debug_get("TG_GEMM_W4") != 0)                                                            # the knob the library dispatches on, not the environment


def gemm_pair(a1, w1, bias1, out1, a2, w2, bias2, out2, epilogue=L.EPI_BIAS):
    """out1 = epi(a1 @ w1^T + bias1) and out2 = epi(a2 @ w2^T + bias2) in one launch (tg_gemm_bf16_pair): same N, K, batch, leading
    dimensions; [B, M, K] activations with M >= 1024."""
    for n, t in (("a1", a1), ("w1", w1), ("out1", out1), ("a2", a2), ("w2", w2), ("out2", out2)):
        _chk(t, n)
    B, M1, Kd, lda, sa1 = _bmk(a1)
    B2, M2, Kd2, lda2, sa2 = _bmk(a2)
    _, _, N, ldc, sc1 = _bmk(out1)
    _, _, N2, ldc2, sc2 = _bmk(out2)
    assert (B, Kd, lda, N, ldc) == (B2, Kd2, lda2, N2, ldc2) and w1.shape == w2.shape == (N, Kd) and w1.stride(0) == w2.stride(0)
    L.check(_launch(f"gemm_pair_M{M1}+{M2}_N{N}_K{Kd}_epi{epilogue}", L.load().tg_gemm_bf16_pair, _p(a1), sa1, _p(w1), _p(bias1), _p(out1), sc1, M1,
                    _p(a2), sa2, _p(w2), _p(bias2), _p(out2), sc2, M2, lda, w1.stride(0), ldc, N, Kd, B, epilogue, _stream()), "tg_gemm_bf16_pair")
    return out1, out2


def gemm_qkv_supported(M, N, K, v_col0):
    """Shapes the fused QKV + V^T launch (tg_gemm_bf16_qkv) takes; TG_GEMM_W4=0 (the cross-check tests' switch to the 8-wave GEMM) disables it too."""
    return M >= 1024 and N % 256 == 0 and K % 64 == 0 and K >= 256 and v_col0 % 256 == 0 and L.debug_get("TG_GEMM_W4") != 0


def gemm_qkv(a1, w1, bias1, out1, vt1, a2=None, w2=None, bias2=None, out2=None, vt2=None, v_col0=None):
    """QKV projection(s) with the V third (columns >= v_col0, default 2N/3) written transposed into vt [B, H, 64, ld] instead of out
    (tg_gemm_bf16_qkv): out[:, :, :v_col0] = a @ w[:v_col0]^T + bias, vt[b, h, d, m] = (a @ w^T + bias)[b, m, v_col0 + 64 h + d], zero for
    m >= M.  One or two problems (same N, K, batch, leading dimensions) in one launch."""
    two = a2 is not None
    for n, t in (("a1", a1), ("w1", w1), ("out1", out1), ("vt1", vt1)) + ((("a2", a2), ("w2", w2), ("out2", out2), ("vt2", vt2)) if two else ()):
        _chk(t, n)
    B, M1, Kd, lda, sa1 = _bmk(a1)
    _, _, N, ldc, sc1 = _bmk(out1)
    v_col0 = 2 * N // 3 if v_col0 is None else v_col0
    assert w1.shape == (N, Kd)

    def vt_ld(vt, M):
        assert vt.dim() == 4 and vt.shape[0] == B and vt.shape[1] * vt.shape[2] == N - v_col0 and vt.is_contiguous() and vt.shape[3] >= M
        return vt.shape[3]
    ld1 = vt_ld(vt1, M1)
    M2 = sa2 = sc2 = ld2 = 0
    if two:
        B2, M2, Kd2, lda2, sa2 = _bmk(a2)
        _, _, N2, ldc2, sc2 = _bmk(out2)
        assert (B, Kd, lda, N, ldc) == (B2, Kd2, lda2, N2, ldc2) and w2.shape == (N, Kd) and w1.stride(0) == w2.stride(0)
        ld2 = vt_ld(vt2, M2)
    name = f"gemm_qkv_M{M1}+{M2}_N{N}_K{Kd}" if two else f"gemm_qkv_M{M1}_N{N}_K{Kd}"
    L.check(_launch(name, L.load().tg_gemm_bf16_qkv, _p(a1), sa1, _p(w1), _p(bias1), _p(out1), sc1, M1, _p(vt1), ld1,
                    _p(a2) if two else None, sa2, _p(w2) if two else None, _p(bias2) if two else None, _p(out2) if two else None, sc2, M2,
                    _p(vt2) if two else None, ld2, lda, w1.stride(0), ldc, N, Kd, B, v_col0, _stream()), "tg_gemm_bf16_qkv")
    return out1, vt1


def adaln_modulate(x, out, ln_weight, ln_bias, eps, table=None):
    """out = LN(x)*(1+scale[g])+shift[g] (table given) or plain affine LN (table None). x/out [B,T,D] views."""
    _chk(x, "x"); _chk(out, "out")
    B, T, D, ldx, sx = _bmk(x)
    _, _, _, ldy, sy = _bmk(out)
    L.check(_launch("adaln_modulate", L.load().tg_adaln_modulate, _p(x), ldx, sx, _p(out), ldy, sy, _p(ln_weight), _p(ln_bias), float(eps), T, D, B,
                                       1 if table is not None else 0, table.ref() if table is not None else None,
                                       _stream()), "tg_adaln_modulate")
    return out


def qk_layernorm_rope(x, heads, ln_weight, ln_bias, eps, seg0=None, seg1=None, out_scale=1.0):
    """In place on x [B,T,heads*64] (a column slice of the fused QKV buffer). seg = (start, (cos, sin))."""
    _chk(x, "x")
    B, T, HD, ld, sb = _bmk(x)
    assert HD == heads * 64

    def unpack(seg):
        if seg is None:
            return 0, 0, None, None
        start, (cos, sin) = seg
        _chk(cos, "cos", torch.float32); _chk(sin, "sin", torch.float32)
        assert cos.is_contiguous() and sin.is_contiguous() and cos.shape[-1] == 64
        return int(start), int(cos.shape[0]), cos, sin
    s0, l0, c0, n0 = unpack(seg0)
    s1, l1, c1, n1 = unpack(seg1)
    L.check(_launch("qk_layernorm_rope", L.load().tg_qk_layernorm_rope, _p(x), ld, sb, T, heads, B, _p(ln_weight), _p(ln_bias), float(eps), s0, l0, _p(c0),
                                          _p(n0), s1, l1, _p(c1), _p(n1), float(out_scale), _stream()), "tg_qk_layernorm_rope")
    return x


def kmax_workspace(tokens, heads, batch, device):
    """Scratch for qk_layernorm_rope_pair(kmax=...): fp32 [tg_qk_kmax_ws_floats]."""
    return torch.empty(L.load().tg_qk_kmax_ws_floats(tokens, heads, batch), dtype=torch.float32, device=device)


def qk_layernorm_rope_pair(xq, xk, heads, q_weight, q_bias, k_weight, k_bias, eps, seg0=None, seg1=None, q_scale=1.0, k_scale=1.0,
                           kmax=None, kmax_ws=None, out=None):
    """qk_layernorm_rope on the q and the k column slices of the same fused buffer in one launch (rotary tables read once).
    kmax (fp32 [B, heads], with kmax_ws from kmax_workspace): also receives max_t ||k_t||^2 of the stored K rows — the key-side range
    bound of the constant-shift attention (tg_qk_layernorm_rope_pair_kmax).  out = (yq, yk): OUT OF PLACE (tg_qk_layernorm_rope_pair_out) — xq / xk stay as they are."""
    _chk(xq, "xq"); _chk(xk, "xk")
    B, T, HD, ld, sb = _bmk(xq)
    assert HD == heads * 64 and _bmk(xk) == (B, T, HD, ld, sb)
    if out is not None:
        yq, yk = out
        _chk(yq, "yq"); _chk(yk, "yk")
        _, _, _, yld, ysb = _bmk(yq)
        assert _bmk(yk) == (B, T, HD, yld, ysb)

    def unpack(seg):
        if seg is None:
            return 0, 0, None, None
        start, (cos, sin) = seg
        _chk(cos, "cos", torch.float32); _chk(sin, "sin", torch.float32)
        assert cos.is_contiguous() and sin.is_contiguous() and cos.shape[-1] == 64
        return int(start), int(cos.shape[0]), cos, sin
    s0, l0, c0, n0 = unpack(seg0)
    s1, l1, c1, n1 = unpack(seg1)
    if kmax is not None:
        _chk(kmax, "kmax", torch.float32); _chk(kmax_ws, "kmax_ws", torch.float32)
        assert kmax.is_contiguous() and kmax.shape == (B, heads) and kmax_ws.numel() >= L.load().tg_qk_kmax_ws_floats(T, heads, B)
    if out is not None:
        L.check(_launch("qk_layernorm_rope_pair", L.load().tg_qk_layernorm_rope_pair_out, _p(xq), _p(xk), ld, sb, _p(yq), _p(yk), yld, ysb, T, heads, B, _p(q_weight),
                        _p(q_bias), _p(k_weight), _p(k_bias), float(eps), s0, l0, _p(c0), _p(n0), s1, l1, _p(c1), _p(n1), float(q_scale), float(k_scale),
                        _p(kmax), _p(kmax_ws) if kmax is not None else None, _stream()), "tg_qk_layernorm_rope_pair_out")
        return yq, yk
    if kmax is not None:
        L.check(_launch("qk_layernorm_rope_pair", L.load().tg_qk_layernorm_rope_pair_kmax, _p(xq), _p(xk), ld, sb, T, heads, B, _p(q_weight), _p(q_bias),
                        _p(k_weight), _p(k_bias), float(eps), s0, l0, _p(c0), _p(n0), s1, l1, _p(c1), _p(n1), float(q_scale), float(k_scale),
                        _p(kmax), _p(kmax_ws), _stream()), "tg_qk_layernorm_rope_pair_kmax")
        return xq, xk
    L.check(_launch("qk_layernorm_rope_pair", L.load().tg_qk_layernorm_rope_pair, _p(xq), _p(xk), ld, sb, T, heads, B, _p(q_weight), _p(q_bias),
                    _p(k_weight), _p(k_bias), float(eps), s0, l0, _p(c0), _p(n0), s1, l1, _p(c1), _p(n1), float(q_scale), float(k_scale),
                    _stream()), "tg_qk_layernorm_rope_pair")
    return xq, xk


def transpose_v(v, heads, key_start, n_keys, vt):
    """vt [B,heads,64,ldvt] <- v[:, key_start:key_start+n_keys] (v is a [B,T,heads*64] column slice)."""
    _chk(v, "v"); _chk(vt, "vt")
    B, T, HD, ld, sb = _bmk(v)
    assert vt.is_contiguous() and vt.shape[:3] == (B, heads, 64)
    L.check(_launch("transpose_v", L.load().tg_transpose_v, _p(v), ld, sb, key_start, n_keys, heads, B, _p(vt), vt.shape[3], _stream()), "tg_transpose_v")
    return vt


def attention(q1, k1, vt1, nk1, out, heads, scale, q2=None, k2=None, vt2=None, nk2=0, seg2_scale=0.0, k_prescaled=False):
    """out[B,nq,heads*64] = softmax(q1 k1^T) v1 + seg2_scale*softmax(q2 k2^T) v2 (segment 2 optional)."""
    _chk(q1, "q1"); _chk(k1, "k1"); _chk(vt1, "vt1"); _chk(out, "out")
    B, nq, _, qld, qsb = _bmk(q1)
    _, _, _, kld, ksb = _bmk(k1)
    _, _, _, old, osb = _bmk(out)
    a2 = (0, 0, 0, 0, 0, 0, 0, 0, 0)
    if q2 is not None:
        _chk(q2, "q2"); _chk(k2, "k2"); _chk(vt2, "vt2")
        _, _, _, q2ld, q2sb = _bmk(q2)
        _, _, _, k2ld, k2sb = _bmk(k2)
        a2 = (_p(q2), q2ld, q2sb, _p(k2), k2ld, k2sb, _p(vt2), vt2.stride(2), nk2)
    L.check(_launch("attention_2seg" if q2 is not None else f"attention_1seg_nq{nq}", L.load().tg_attention_fwd, _p(q1), qld, qsb, _p(k1), kld, ksb, _p(vt1), vt1.stride(2), nk1, *a2, float(seg2_scale),
                                      _p(out), old, osb, nq, heads, B, float(scale), 1 if k_prescaled else 0, _stream()), "tg_attention_fwd")
    return out


def attention_lse(q, k, vt, nk, out, heads, scale, k_prescaled=False, kmax=None, retry=None):
    """Training forward of one attention call: out = softmax(scale q k^T) v AND the per-row log-sum-exp (log2 domain, fp32 [B, heads, nq]) that
    attention_bwd takes instead of recomputing it (tg_attention_fwd_lse).  k_prescaled: k already carries scale * log2(e)
    (qk_layernorm_rope_pair k_scale) — attention_bwd then takes the same k with scale = ln 2; kmax (from qk_layernorm_rope_pair) + retry
    (AttnRetry): the verified constant-shift softmax of the inference path (tg_attention_fwd_lse_ex)."""
    _chk(q, "q"); _chk(k, "k"); _chk(vt, "vt"); _chk(out, "out")
    B, nq, _, qld, qsb = _bmk(q)
    _, _, _, kld, ksb = _bmk(k)
    _, _, _, old, osb = _bmk(out)
    lse = torch.empty(B, heads, nq, dtype=torch.float32, device=q.device)
    if k_prescaled or kmax is not None or retry is not None:
        if kmax is not None:
            _chk(kmax, "kmax", torch.float32)
            assert kmax.is_contiguous() and kmax.shape == (B, heads)
        assert retry is None or retry.fits(nq, 0, heads, B)
        L.check(_launch(f"attention_lse_nq{nq}", L.load().tg_attention_fwd_lse_ex, _p(q), qld, qsb, _p(k), kld, ksb, _p(vt), vt.stride(2), nk, _p(out), old, osb,
                        nq, heads, B, float(scale), 1 if k_prescaled else 0, _p(kmax), _p(retry.buf) if retry is not None else None,
                        retry.ints if retry is not None else 0, _p(lse), _stream()), "tg_attention_fwd_lse_ex")
        return out, lse
    L.check(_launch(f"attention_lse_nq{nq}", L.load().tg_attention_fwd_lse, _p(q), qld, qsb, _p(k), kld, ksb, _p(vt), vt.stride(2), nk, _p(out), old, osb, nq, heads, B,
                    float(scale), _p(lse), _stream()), "tg_attention_fwd_lse")
    return out, lse


class BwdDeviceState:
    """What the one-kernel attention backward needs per DEVICE, owned by the caller of the C ABI (nothing process-wide lives in the library):
    `status` = the int32[4] words tg_attention_bwd_ex reports through ([0] sticky count of exchange polls that timed out, [1] poll-limit override),
    `one_kernel` = verdict of the device probe (tg_attention_bwd_probe on a buffer of ours; one synchronisation, here, outside the ABI)."""

    _by_device = {}

    def __init__(self, device):
        lib = L.load()
        self.status = torch.zeros(4, dtype=torch.int32, device=device)
        self.one_kernel = False
        self.probe = None
        if os.environ.get("TG_ATTN_BWD_FUSED", "1") != "0":
            nb = lib.tg_attention_bwd_probe_bytes()
            buf = torch.empty(nb, dtype=torch.uint8, device=device)
            L.check(lib.tg_attention_bwd_probe(_p(buf), nb, _stream()), "tg_attention_bwd_probe")
            host = buf.cpu()                                   # (synchronises this stream, once per device and process)
            self.one_kernel = bool(lib.tg_attention_bwd_probe_verdict(host.data_ptr(), nb))
            # what the probe saw, for the record (bench.py's train sub-record): workgroups off their XCD / polls that gave up, and the chain's sums
            hi = host[:16].view(torch.int32)
            self.probe = {"flagged": int(hi[0]), "workgroups_not_on_xcd_b_mod_8": int(hi[2]),
                          "sums_exact": bool((host[4 * 2052:].view(torch.float32) == 528.0).all())}

    def flags(self):
        return L.TG_BWD_ONE_KERNEL if self.one_kernel else 0

    @classmethod
    def get(cls, device):
        device = torch.device(device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        st = cls._by_device.get(idx)
        if st is None:
            with torch.cuda.device(idx):
                st = cls._by_device[idx] = cls(torch.device("cuda", idx))
        return st


def attention_bwd_status(device=None, clear=True):
    """(timed-out exchange polls, workgroups off their head's XCD) summed over the devices this process has used since the last call — the sticky status
    words of the one-kernel backward (one synchronisation).  `clear` resets them, so that a discarded step can be attempted again."""
    devs = [torch.device(device).index] if device is not None else list(BwdDeviceState._by_device)
    polls = xcd = 0
    for idx in devs:
        st = BwdDeviceState._by_device.get(idx if idx is not None else torch.cuda.current_device())
        if st is None:
            continue
        n, _, x, _ = st.status.tolist()
        polls, xcd = polls + n, xcd + x
        if clear and (n or x):
            st.status[0].zero_(); st.status[2].zero_()
    return polls, xcd


def attention_bwd_error(polls, xcd, where="this process"):
    return RuntimeError(f"tg_attention_bwd_ex ({where}): {polls} ordered dQ exchange poll(s) timed out, {xcd} workgroup(s) found their head's key blocks on "
                        "more than one XCD — the gradients of this step are invalid; discard the step. TG_ATTN_BWD_FUSED=0 selects the two-launch form.")


def attention_bwd_check(device=None):
    """Read the sticky status word of the one-kernel backward on `device` (synchronises) and raise if any ordered-exchange poll timed out since the
    last check: the dq of that launch is invalid, the step must not be applied.  Single-process callers; the training step uses attention_bwd_status
    and makes the verdict collective before anything is applied (train.To2VTrainStep.micro_step)."""
    polls, xcd = attention_bwd_status(device)
    if polls or xcd:
        raise attention_bwd_error(polls, xcd, f"cuda:{torch.device(device).index}" if device is not None else "this process")


def _bwd_problem(q, k, v, o, dout, heads, scale, dq=None, dk=None, dv=None, accumulate=False, lse=None, dv_bf16=None):
    """One tg_attn_bwd_problem + the tensors that must outlive the launch call: (struct, (dq, dk, dv), keep).  dv_bf16 (bf16 [B, nk, heads*64] view, e.g. the V third of a fused
    projection gradient): receives bf16(dv) from the kernel's epilogue; the fp32 dv is then written only when `dv` is given too (else the tuple's dv is None)."""
    for n, t in (("q", q), ("k", k), ("v", v), ("o", o), ("dout", dout)):
        _chk(t, n)
    if lse is not None:
        _chk(lse, "lse", torch.float32)
        assert lse.is_contiguous() and lse.shape == (q.shape[0], heads, q.shape[1])
    B, nq, HD, qld, qsb = _bmk(q)
    _, nk, _, kld, ksb = _bmk(k)
    _, _, _, vld, vsb = _bmk(v)
    _, _, _, old, osb = _bmk(o)
    _, _, _, gld, gsb = _bmk(dout)
    assert HD == heads * 64 and o.shape == q.shape == dout.shape and v.shape == k.shape
    f32 = torch.float32
    acc = 3 if accumulate is True else int(accumulate)     # True: all three; 2: dk / dv only (dq overwritten); 0 / False: overwrite
    new = torch.zeros if acc else torch.empty              # accumulate adds to the buffer: a fresh one must start at zero
    dq = new(B, nq, HD, dtype=f32, device=q.device) if dq is None else _chk(dq, "dq", f32)
    dk = new(B, nk, HD, dtype=f32, device=q.device) if dk is None else _chk(dk, "dk", f32)
    if dv_bf16 is not None:
        _chk(dv_bf16, "dv_bf16")
        assert dv_bf16.shape == (B, nk, HD) and dv_bf16.stride(2) == 1 and (dv is not None or not (acc & 2)), "accumulate into dv needs the fp32 dv"
    if dv is not None or dv_bf16 is None:
        dv = new(B, nk, HD, dtype=f32, device=q.device) if dv is None else _chk(dv, "dv", f32)
    ws = torch.empty(L.load().tg_attention_bwd_ws_floats(nq, nk, heads, B), dtype=f32, device=q.device)
    pr = L.AttnBwdProblem()
    (pr.q, pr.q_ld, pr.q_sb, pr.k, pr.k_ld, pr.k_sb, pr.v, pr.v_ld, pr.v_sb, pr.o, pr.o_ld, pr.o_sb, pr.dout, pr.do_ld, pr.do_sb) = (
        _p(q), qld, qsb, _p(k), kld, ksb, _p(v), vld, vsb, _p(o), old, osb, _p(dout), gld, gsb)
    (pr.dq, pr.dq_ld, pr.dq_sb, pr.dk, pr.dk_ld, pr.dk_sb, pr.dv, pr.dv_ld, pr.dv_sb) = (
        _p(dq), dq.stride(1), dq.stride(0), _p(dk), dk.stride(1), dk.stride(0), _p(dv) if dv is not None else None, dv.stride(1) if dv is not None else 0,
        dv.stride(0) if dv is not None else 0)
    if dv_bf16 is not None:
        pr.dv_bf16, pr.dv_bf16_ld, pr.dv_bf16_sb = _p(dv_bf16), dv_bf16.stride(1), dv_bf16.stride(0)
    pr.nq, pr.nk, pr.scale, pr.accumulate, pr.lse, pr.ws = nq, nk, float(scale), acc, _p(lse), _p(ws)
    return pr, (dq, dk, dv), (ws, B)


def attention_bwd_multi(problems, heads):
    """One or two attention backward problems in ONE call (tg_attention_bwd_multi): the same as attention_bwd on each in order, except that two problems that both
    take the one-kernel form share a launch (the second rides in the first's last round of CUs).  problems: dicts of attention_bwd's arguments (without `heads`).
    Returns the list of (dq, dk, dv)."""
    assert 1 <= len(problems) <= 2
    built = [_bwd_problem(heads=heads, **pr) for pr in problems]
    arr = (L.AttnBwdProblem * len(built))(*[b[0] for b in built])
    B = built[0][2][1]
    dev = problems[0]["q"].device
    st = BwdDeviceState.get(dev)
    L.check(_launch("attention_bwd", L.load().tg_attention_bwd_multi, arr, len(built), heads, B, st.flags(), _p(st.status), _stream()), "tg_attention_bwd_multi")
    return [b[1] for b in built]


def attention_bwd(q, k, v, o, dout, heads, scale, dq=None, dk=None, dv=None, accumulate=False, lse=None, dv_bf16=None):
    """Gradients of o = softmax(scale q k^T) v per head (tg_attention_bwd_multi with one problem).  q/o/dout [B,nq,heads*64], k/v [B,nk,heads*64] bf16 views;
    returns fp32 (dq, dk, dv) shaped like q, k, v (given tensors are written, or added to with accumulate=True).  lse: the forward's
    log-sum-exp from attention_lse (optional; recomputed when None).  The one-kernel form is offered to the library when the device probe
    passed (BwdDeviceState); a poll time-out inside it is reported by attention_bwd_status() / attention_bwd_check(), never silently."""
    return attention_bwd_multi([dict(q=q, k=k, v=v, o=o, dout=dout, scale=scale, dq=dq, dk=dk, dv=dv, accumulate=accumulate, lse=lse, dv_bf16=dv_bf16)], heads)[0]


def _attn_problem(q1, k1, vt1, nk1, out, q2=None, k2=None, vt2=None, nk2=0, seg2_scale=0.0, kmax1=None, kmax2=None, seg2_scale_batch=None):
    for n, t in (("q1", q1), ("k1", k1), ("vt1", vt1), ("out", out)):
        _chk(t, n)
    pr = L.AttnProblem()
    B, nq, _, qld, qsb = _bmk(q1)
    _, _, _, kld, ksb = _bmk(k1)
    _, _, _, old, osb = _bmk(out)
    for km in (kmax1, kmax2):
        if km is not None:
            _chk(km, "kmax", torch.float32)
            assert km.is_contiguous() and km.shape[0] == B
    pr.seg[0] = L.AttnSegment(_p(q1), qld, qsb, _p(k1), kld, ksb, _p(vt1), vt1.stride(2), nk1, _p(kmax1) or None)
    pr.nseg = 1
    if q2 is not None:
        _chk(q2, "q2"); _chk(k2, "k2"); _chk(vt2, "vt2")
        _, _, _, q2ld, q2sb = _bmk(q2)
        _, _, _, k2ld, k2sb = _bmk(k2)
        pr.seg[1] = L.AttnSegment(_p(q2), q2ld, q2sb, _p(k2), k2ld, k2sb, _p(vt2), vt2.stride(2), nk2, _p(kmax2) or None)
        pr.nseg = 2
    pr.seg2_scale = float(seg2_scale)
    pr.out, pr.out_ld, pr.out_strideB, pr.nq = _p(out), old, osb, nq
    if seg2_scale_batch is not None:            # host list of per-batch-item weights; the ctypes array must outlive the call: kept on the struct
        assert len(seg2_scale_batch) == B
        pr._scales = (C.c_float * B)(*[float(v) for v in seg2_scale_batch])
        pr.seg2_scale_batch = C.cast(pr._scales, C.POINTER(C.c_float))
    return pr, B


class AttnRetry:
    """Retry workspace of the constant-shift attention path (tg_attention_fwd_multi retry_ws): one int32 buffer per model, sized for its
    largest launch, zeroed once.  `count()` = workgroups recomputed with the running maximum so far (a device read: diagnostics / tests);
    `poll()` is the non-blocking form the model uses to notice weights for which the shift estimate keeps failing."""

    def __init__(self, nq0, nq1, heads, batch, device):
        self.ints = int(L.load().tg_attention_retry_ints(nq0, nq1, heads, batch))
        self.buf = torch.zeros(self.ints, dtype=torch.int32, device=device)
        self._host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._event = None
        # the fast-path launches issued against THIS buffer and (launches, retries) at the last landed poll: the counter lives in this buffer, so
        # the bookkeeping that interprets it does too (a model owns one AttnRetry per workspace shape; mixing their deltas would be meaningless)
        self.launches, self.mark = 0, (0, 0)
        # the key-split scratch of the same launch shape (tg_attn_workspace.split; 0 floats when the shape leaves no half-round tail)
        nsf = int(L.load().tg_attention_split_floats(nq0, nq1, heads, batch))
        self.split = torch.empty(nsf, dtype=torch.float32, device=device) if nsf else None

    def fits(self, nq0, nq1, heads, batch):
        return L.load().tg_attention_retry_ints(nq0, nq1, heads, batch) <= self.ints

    def count(self):
        return int(self.buf[0].item())

    def poll(self, tag):
        """Non-blocking read of the counter: returns (value, tag given when that copy was queued) once a queued copy has landed, else
        None; queues the next copy on the current stream when none is pending."""
        val = None
        if self._event is not None and self._event.query():
            val = (int(self._host[0]), self._tag)
            self._event = None
        if self._event is None:
            self._host.copy_(self.buf[:1], non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()
            self._tag = tag
        return val


def attention_multi(main, rider, heads, scale, k_prescaled=False, retry=None, split=None):
    """One launch for one or two attention problems of the same heads/batch (tg_attention_fwd_multi).  main / rider: dicts of the
    keyword arguments of `attention` (q1, k1, vt1, nk1, out [, q2, k2, vt2, nk2, seg2_scale, kmax1, kmax2]); the rider (may be None) has one
    key segment.  kmax1 / kmax2 (fp32 [B, heads], from qk_layernorm_rope_pair(kmax=...)) on every segment + retry (AttnRetry): the
    constant-shift softmax path (tg_attn_segment.k_norm2_max).  split (fp32 scratch tensor, e.g. AttnRetry.split): lets the launch split its
    half-round tail over the key axis (tg_attn_workspace.split)."""
    import ctypes
    pa, B = _attn_problem(**main)
    n = 1
    if rider is not None:
        pb, B2 = _attn_problem(**rider)
        assert B == B2
        arr = (L.AttnProblem * 2)(pa, pb)
        n = 2
    else:
        arr = (L.AttnProblem * 1)(pa)
    if retry is not None:
        assert retry.fits(main["q1"].shape[1], rider["q1"].shape[1] if rider is not None else 0, heads, B)
    ws = L.AttnWorkspace(_p(retry.buf) if retry is not None else None, retry.ints if retry is not None else 0,
                         _p(split) if split is not None else None, split.numel() if split is not None else 0)
    L.check(_launch("attention_2seg+rider" if n == 2 else f"attention_multi1_nq{pa.nq}", L.load().tg_attention_fwd_multi, ctypes.addressof(arr), n, heads, B,
                    float(scale), 1 if k_prescaled else 0, ctypes.addressof(ws), _stream()), "tg_attention_fwd_multi")
    return main["out"], (rider["out"] if rider is not None else None)


def timestep_sinusoid(t, dim, out):
    _chk(t, "t", torch.int64); _chk(out, "out")
    L.check(L.load().tg_timestep_sinusoid(_p(t), t.numel(), dim, _p(out), _stream()), "tg_timestep_sinusoid")
    return out


def patchify(lat, out, p=2):
    """lat [BF,C,H,W] contiguous -> out [BF*(H/p)*(W/p), ld >= p*p*C] (pad columns are left untouched)"""
    _chk(lat, "lat"); _chk(out, "out")
    assert lat.is_contiguous() and out.stride(1) == 1
    BF, Cc, H, W = lat.shape
    L.check(L.load().tg_patchify(_p(lat), _p(out), out.stride(0), BF, Cc, H, W, p, _stream()), "tg_patchify")
    return out


def unpatchify(x, lat, p=2):
    """x [BF*(H/p)*(W/p), ld >= p*p*C] rows -> lat [BF,C,H,W] contiguous"""
    _chk(x, "x"); _chk(lat, "lat")
    assert lat.is_contiguous()
    BF, Cc, H, W = lat.shape
    L.check(L.load().tg_unpatchify(_p(x), x.stride(0), _p(lat), BF, Cc, H, W, p, _stream()), "tg_unpatchify")
    return lat


def cfg_dpm_step(model_out, x, old_x0, noise, coef, guidance, x_out, x0_out):
    """model_out [2,F,E]; x/old_x0/x_out/x0_out [F,E]; noise [F,2,E]; coef fp32 [F,8] (device)."""
    for n, t in (("model_out", model_out), ("x", x), ("old_x0", old_x0), ("noise", noise), ("x_out", x_out), ("x0_out", x0_out)):
        _chk(t, n)
        assert t.is_contiguous()
    _chk(coef, "coef", torch.float32)
    F_, E = x.shape[0], x[0].numel()
    L.check(L.load().tg_cfg_dpm_step(_p(model_out), _p(x), _p(old_x0), _p(noise), _p(coef), float(guidance), _p(x_out),
                                     _p(x0_out), F_, E, _stream()), "tg_cfg_dpm_step")
    return x_out, x0_out


def cfg_dpm_step_f32(model_out, x, old_x0, noise, coef, guidance, x_out, x0_out):
    """Pipeline-loop variant: model_out [2,F,E] / x / x_out / noise [F,2,E] bf16; old_x0, x0_out [F,E] fp32."""
    for n, t in (("model_out", model_out), ("x", x), ("x_out", x_out), ("noise", noise)):
        _chk(t, n)
        assert t.is_contiguous()
    for n, t in (("old_x0", old_x0), ("x0_out", x0_out), ("coef", coef)):
        _chk(t, n, torch.float32)
        assert t.is_contiguous()
    F_, E = x.shape[0], x[0].numel()
    L.check(L.load().tg_cfg_dpm_step_f32(_p(model_out), _p(x), _p(old_x0), _p(noise), _p(coef), float(guidance), _p(x_out),
                                         _p(x0_out), F_, E, _stream()), "tg_cfg_dpm_step_f32")
    return x_out, x0_out


PRED_TYPES = {"v_prediction": 0, "epsilon": 1, "sample": 2}


def cfg_dpm_step_ex(model_out, x, old_x0, noise, coef, guidance, x_out, x0_out, guidance_img=0.0, guidance_per_frame=None, f32_math=False,
                    prediction_type="v_prediction"):
    """tg_cfg_dpm_step_ex: model_out [1 (no guidance), 2 or 3, F, E] bf16; x / x_out / noise [F,2,E] bf16; old_x0 / x0_out [F,E] bf16 or fp32 (fp32: the pipelines'
    solver state, implies f32_math); guidance_per_frame: fp32 [F, 2] device tensor {g, g_img} (the worker's dynamic cfg) or None."""
    for n, t in (("model_out", model_out), ("x", x), ("x_out", x_out), ("noise", noise)):
        _chk(t, n)
        assert t.is_contiguous()
    f32_state = old_x0.dtype == torch.float32
    for n, t in (("old_x0", old_x0), ("x0_out", x0_out)):
        _chk(t, n, torch.float32 if f32_state else BF16)
        assert t.is_contiguous()
    _chk(coef, "coef", torch.float32)
    F_, E = x.shape[0], x[0].numel()
    br = model_out.shape[0]
    assert br in (1, 2, 3) and model_out.numel() == br * F_ * E and coef.shape == (F_, 8) and coef.is_contiguous()
    if guidance_per_frame is not None:
        _chk(guidance_per_frame, "guidance_per_frame", torch.float32)
        assert guidance_per_frame.is_contiguous() and guidance_per_frame.shape == (F_, 2)
    L.check(L.load().tg_cfg_dpm_step_ex(_p(model_out), br, _p(x), _p(old_x0), _p(noise), _p(coef), float(guidance), float(guidance_img),
                                        _p(guidance_per_frame) or None, 1 if (f32_math or f32_state) else 0, 1 if f32_state else 0,
                                        PRED_TYPES[prediction_type], _p(x_out), _p(x0_out), F_, E, _stream()), "tg_cfg_dpm_step_ex")
    return x_out, x0_out


def pca_inverse(lat, std, mean, comp, pmean, out):
    """lat bf16 [F,16,h,w]; std/mean fp32 [16]; comp fp32 [16,C]; pmean fp32 [C] -> out bf16 [F,C,h,w]."""
    _chk(lat, "lat"); _chk(out, "out")
    for n, t in (("std", std), ("mean", mean), ("comp", comp), ("pmean", pmean)):
        _chk(t, n, torch.float32)
        assert t.is_contiguous()
    assert lat.is_contiguous() and out.is_contiguous()
    F_, nc, h, w = lat.shape
    Cc = out.shape[1]
    assert comp.shape == (nc, Cc) and pmean.numel() == Cc and std.numel() == nc and mean.numel() == nc and out.shape == (F_, Cc, h, w)
    L.check(L.load().tg_pca_inverse(_p(lat), _p(std), _p(mean), _p(comp), _p(pmean), _p(out), F_, nc, h * w, Cc, _stream()), "tg_pca_inverse")
    return out


def pca_lowrank_filter(x, comp, mean, out=None):
    """x bf16 [rows, D] (rows may be strided) -> bf16 [rows, D]: project onto the first comp.shape[0] (<= 16) PCA components and back, fp32."""
    _chk(x, "x"); _chk(comp, "comp", torch.float32); _chk(mean, "mean", torch.float32)
    assert x.dim() == 2 and comp.is_contiguous() and mean.is_contiguous() and comp.shape[1] == x.shape[1] and mean.numel() == x.shape[1]
    out = torch.empty_like(x) if out is None else _chk(out, "out")
    L.check(_launch("pca_lowrank_filter", L.load().tg_pca_lowrank_filter, _p(x), x.stride(0), _p(comp), _p(mean), _p(out), out.stride(0), x.shape[0],
                    x.shape[1], comp.shape[0], _stream()), "tg_pca_lowrank_filter")
    return out


# ---------------------------------------------------------------------------------------------------------
# VAE (channels-last bf16 activations)
# ---------------------------------------------------------------------------------------------------------
_ZEROS = {}


def zero_page(device):
    z = _ZEROS.get(device)
    if z is None:
        z = _ZEROS[device] = torch.zeros(8192, dtype=BF16, device=device)
    return z


_SPLITK = {}


def _splitk_floats(*key):
    n = _SPLITK.get(key)
    if n is None:
        n = _SPLITK[key] = int(L.load().tg_conv3d_splitk_floats(*key))
    return n


def conv3d_cl(x, w_packed, bias, cout, kt, kh, kw, cache=None, stride=1, pad=1, up=1, t_map=None, residual=None, out_dims=None,
              gn_stats_eps=None):
    """x [T,H,W,Cin] channels-last; w_packed [Cout_pad, kt*kh*kw, Cin]; returns y [To,Ho,Wo,cout].
    gn_stats_eps: also produce the GroupNorm(32) sums of y in the epilogue (cout % 128 == 0): attached as y.gn_sums (GnSums) for the norm that
    consumes y, which finalises them itself — no statistics launch; y.gn_sums.stats() gives the [32, 2] (mean, rstd) tensor on request."""
    _chk(x, "x"); _chk(w_packed, "w")
    assert x.is_contiguous() and w_packed.is_contiguous()
    T, H, W, Cin = x.shape
    To, Ho, Wo = out_dims if out_dims is not None else (T, H, W)
    y = torch.empty(To, Ho, Wo, cout, dtype=BF16, device=x.device)
    if cache is not None:
        _chk(cache, "cache"); assert cache.is_contiguous() and cache.shape == (kt - 1, H, W, Cin)
    if residual is not None:
        _chk(residual, "residual"); assert residual.is_contiguous() and residual.shape == y.shape
    if t_map is not None:
        _chk(t_map, "t_map", torch.int32)
    fuse = gn_stats_eps is not None and cout % 128 == 0 and w_packed.shape[0] == cout
    partial = torch.empty(L.load().tg_conv3d_gn_partial_floats(To, Ho, Wo), dtype=torch.float32, device=x.device) if fuse else None
    nws = _splitk_floats(Cin, cout, w_packed.shape[0], kt, kh, kw, To, Ho, Wo)
    ws = torch.empty(nws, dtype=torch.float32, device=x.device) if nws else None
    L.check(_launch(f"conv3d_cl_Cin{Cin}_Cout{cout}_k{kt}{kh}{kw}_s{stride}_u{up}", L.load().tg_conv3d_cl, _p(x), T, H, W, Cin, _p(cache),
                    _p(w_packed), _p(bias), cout, w_packed.shape[0], kt, kh, kw, stride, pad, up, _p(t_map), _p(residual), _p(y), cout, To, Ho,
                    Wo, _p(zero_page(x.device)), _p(partial), _p(ws), _stream()), "tg_conv3d_cl")
    if fuse:
        y.gn_sums = GnSums(partial, To * Ho * Wo, cout, float(gn_stats_eps))
    return y


def conv3d_up2_subpixel_ok(T, H, W, Cin, cout):
    """Does tg_conv3d_up2_subpixel take this shape (the 256 x 256 convolution kernel's range)?"""
    return bool(L.load().tg_conv3d_up2_subpixel_ok(T, H, W, Cin, cout))


def conv3d_up2_subpixel(x, w_phases, bias, cout, gn_stats_eps=None, time_x2=False):
    """Nearest x2 spatial upsampling + Conv2d 3x3 (pad 1) as four 2x2 phase convolutions on the low-resolution input (tg_conv3d_up2_subpixel; the deviation — pre-summed
    bf16 weights — is stated in the header).  x [T, H, W, Cin] channels-last; w_phases [4, cout, 4, Cin] (vae.pack_up2_phases); returns y [To, 2H, 2W, cout], with
    y.gn_sums as conv3d_cl leaves them.  time_x2: the layer's nearest x2 in TIME as well (each frame convolved once, stored twice; To = 2T, or 2T - 1 for odd T > 1)."""
    _chk(x, "x"); _chk(w_phases, "w_phases")
    assert x.is_contiguous() and w_phases.is_contiguous()
    T, H, W, Cin = x.shape
    assert tuple(w_phases.shape) == (4, cout, 4, Cin)
    To = T if not (time_x2 and T > 1) else (2 * T - 1 if T % 2 else 2 * T)        # time_x2: every frame twice, except the first of an odd count (CogVideoXUpsample3D)
    y = torch.empty(To, 2 * H, 2 * W, cout, dtype=BF16, device=x.device)
    fuse = gn_stats_eps is not None
    partial = torch.empty(L.load().tg_conv3d_up2_subpixel_gn_floats(T, H, W), dtype=torch.float32, device=x.device) if fuse else None
    L.check(_launch(f"conv3d_up2_subpixel_Cin{Cin}_Cout{cout}_t{int(bool(time_x2))}", L.load().tg_conv3d_up2_subpixel, _p(x), T, H, W, Cin, _p(w_phases), _p(bias), cout, _p(y), cout,
                    1 if time_x2 else 0, _p(zero_page(x.device)), _p(partial), _stream()), "tg_conv3d_up2_subpixel")
    if fuse:
        y.gn_sums = GnSums(partial, To * 4 * H * W, cout, float(gn_stats_eps))
    return y


class GnSums:
    """The GroupNorm(32) sums a convolution's epilogue left for the norm that reads its output: rows of [2][32] (sum, sum of squares per group), one per
    128 voxels.  The norm pass turns <= 64 rows into (mean, rstd) in its own prologue (tg_groupnorm_silu_ex / tg_spatialnorm_silu_ex); longer lists are
    cut to <= 64 fp64 rows by one tg_groupnorm_reduce launch first.  `.stats()` gives the classic [32, 2] (mean, rstd) tensor through
    tg_groupnorm_finalize (tests / callers that want the numbers)."""

    def __init__(self, partial, V, C, eps):
        self.partial, self.V, self.C, self.eps = partial, V, C, eps
        self._rows = None

    def rows(self):
        """(tensor, row count, is_f64) as the *_ex norm entry points take them."""
        if self._rows is None:
            n = self.partial.numel() // 64
            r = L.load().tg_groupnorm_reduce_rows(n)
            if r == 0:
                self._rows = (self.partial, n, 0)
            else:
                out = torch.empty(r, 64, dtype=torch.float64, device=self.partial.device)
                L.check(_launch("groupnorm_reduce", L.load().tg_groupnorm_reduce, _p(self.partial), n, _p(out), _stream()), "tg_groupnorm_reduce")
                self._rows = (out, r, 1)
        return self._rows

    def stats(self):
        if self.partial.numel() // 64 != (self.V + 127) // 128:
            # tg_groupnorm_finalize walks ceil(V / 128) rows: the phase launches of conv3d_up2_subpixel leave 4 x ceil(V / 4 / 128) — the norm passes take any row count (rows())
            raise ValueError("GnSums.stats(): this buffer has one row list per phase launch; use rows() (what the norm passes read)")
        stats = torch.empty(32, 2, dtype=torch.float32, device=self.partial.device)
        L.check(_launch("groupnorm_finalize", L.load().tg_groupnorm_finalize, _p(self.partial), self.V, self.C, self.eps, _p(stats), _stream()),
                "tg_groupnorm_finalize")
        return stats


def groupnorm_stats(x2d, eps=1e-6):
    """x2d [V, C] -> stats fp32 [32, 2] (mean, rstd)."""
    _chk(x2d, "x"); assert x2d.is_contiguous()
    V, C = x2d.shape
    n = L.load().tg_groupnorm_partial_floats(V, C)
    partial = torch.empty(n, dtype=torch.float32, device=x2d.device)
    stats = torch.empty(32, 2, dtype=torch.float32, device=x2d.device)
    L.check(_launch("groupnorm_stats", L.load().tg_groupnorm_stats, _p(x2d), V, C, float(eps), _p(partial), _p(stats), _stream()), "tg_groupnorm_stats")
    return stats


def _stats_args(stats):
    """stats: a [32, 2] (mean, rstd) tensor, or a GnSums -> (stats ptr, sums ptr, rows, is_f64, eps) of the *_ex norm entry points."""
    if isinstance(stats, GnSums):
        t, n, f64 = stats.rows()
        return None, _p(t), n, f64, stats.eps
    return _p(stats), None, 0, 0, 0.0


def groupnorm_silu(x, stats, gamma, beta, silu=True):
    """stats: [32, 2] (mean, rstd) from groupnorm_stats, or the GnSums a convolution attached to x (no statistics launch)."""
    _chk(x, "x"); assert x.is_contiguous()
    C = x.shape[-1]
    y = torch.empty_like(x)
    L.check(_launch("groupnorm_silu", L.load().tg_groupnorm_silu_ex, _p(x), x.numel() // C, C, *_stats_args(stats), _p(gamma), _p(beta), _p(y),
                    1 if silu else 0, _stream()), "tg_groupnorm_silu")
    return y


def spatialnorm_silu(f, stats, gamma, beta, yz, bz, zdims, silu=True):
    """f [T,H,W,C]; yz/bz [Tz*Hz*Wz, >=C] = conv_y(z)/conv_b(z) per latent voxel (row stride may exceed C); zdims=(Tz,Hz,Wz)."""
    _chk(f, "f"); _chk(yz, "yz"); _chk(bz, "bz"); assert f.is_contiguous()
    T, H, W, C = f.shape
    Tz, Hz, Wz = zdims
    assert yz.shape[0] == Tz * Hz * Wz and yz.stride(0) == bz.stride(0)
    y = torch.empty_like(f)
    L.check(_launch("spatialnorm_silu", L.load().tg_spatialnorm_silu_ex, _p(f), T, H, W, C, *_stats_args(stats), _p(gamma), _p(beta), _p(yz), _p(bz),
                    yz.stride(0), Tz, Hz, Wz, _p(y), 1 if silu else 0, _stream()), "tg_spatialnorm_silu")
    return y


def avgpool_time(x):
    _chk(x, "x"); assert x.is_contiguous()
    T, H, W, C = x.shape
    To = 1 + (T - 1) // 2 if T % 2 else T // 2
    y = torch.empty(To, H, W, C, dtype=BF16, device=x.device)
    L.check(_launch("avgpool_time", L.load().tg_avgpool_time, _p(x), T, H * W, C, _p(y), _stream()), "tg_avgpool_time")
    return y


def ncdhw_to_cl(src, t0, Tc, h0, Hc, w0, Wc, Cpad, scale=1.0):
    """src [C,T,H,W] (fp32 or bf16, contiguous) window -> [Tc,Hc,Wc,Cpad] bf16."""
    if not src.is_cuda:
        raise RuntimeError("ncdhw_to_cl: expected a GPU tensor (tokensgen_amd has no CPU fallback)")
    assert src.is_contiguous() and src.dtype in (torch.float32, BF16)
    C, Tt, Ht, Wt = src.shape
    dst = torch.empty(Tc, Hc, Wc, Cpad, dtype=BF16, device=src.device)
    L.check(L.load().tg_ncdhw_to_cl(_p(src), 1 if src.dtype == torch.float32 else 0, C, Tt, Ht, Wt, t0, Tc, h0, Hc, w0, Wc, float(scale), _p(dst),
                                    Cpad, _stream()), "tg_ncdhw_to_cl")
    return dst


def cl_to_ncdhw(src, dst, t0=0, h0=0, w0=0):
    """src [T,H,W,C] channels-last bf16 -> window of dst [C,Tt,Ht,Wt] (fp32 or bf16)."""
    _chk(src, "src"); assert src.is_contiguous() and dst.is_contiguous() and dst.is_cuda
    T, H, W, C = src.shape
    Cd, Tt, Ht, Wt = dst.shape
    assert Cd == C
    L.check(L.load().tg_cl_to_ncdhw(_p(src), C, C, T, H, W, _p(dst), 1 if dst.dtype == torch.float32 else 0, Tt, Ht, Wt, t0, h0, w0, _stream()),
            "tg_cl_to_ncdhw")
    return dst


def tile_blend(a, b, axis, extent):
    """In place on b: NCDHW [C,T,H,W] tiles (same dtype, fp32 or bf16); axis 3 = height (blend_v), 4 = width (blend_h)."""
    assert a.is_cuda and b.is_cuda and a.dtype == b.dtype and a.is_contiguous() and b.is_contiguous()
    C, T, Ha, Wa = a.shape
    _, _, Hb, Wb = b.shape
    extent = min(a.shape[axis - 1], b.shape[axis - 1], extent)
    if extent <= 0:
        return b
    L.check(L.load().tg_tile_blend(_p(a), _p(b), 1 if a.dtype == torch.float32 else 0, C, T, Ha, Wa, Hb, Wb, axis, extent, _stream()), "tg_tile_blend")
    return b
