"""Timing-only variants of attn_bwd_fused_kernel: the streamed tensors addressed HEAD-MAJOR ([B][H][N][64] inside the same buffers) to see how much of the
L2-miss traffic is set conflicts of the 6 KB / 12 KB row strides.  Results are garbage; only time and PMC counters mean anything."""


def patch(s, dq=True, qdo=True):
    i = s.index("void attn_bwd_fused_kernel(FusedParams fp)")
    a, b = s[:i], s[i:]
    if dq:
        b = b.replace("float* const DQb = p.dq + (long)b * p.dq_sb + h * HD + ec;", "float* const DQb = p.dq + (long)b * p.dq_sb + (long)h * p.nq * HD + ec;", 1)
        b = b.replace("float* pRq = DQb + (long)min(er, p.nq - 1) * p.dq_ld;", "float* pRq = DQb + (long)min(er, p.nq - 1) * HD;", 1)
        b = b.replace("const long stepDQ = (long)BT * p.dq_ld;", "const long stepDQ = (long)BT * HD;", 1)
        b = b.replace("float* const pDQLast = DQb + (long)min(qlast + er, p.nq - 1) * p.dq_ld;", "float* const pDQLast = DQb + (long)min(qlast + er, p.nq - 1) * HD;", 1)
    if qdo:
        b = b.replace("const bf16_t* Q = p.q + (long)b * p.q_sb + h * HD;", "const bf16_t* Q = p.q + (long)b * p.q_sb + (long)h * p.nq * HD;", 1)
        b = b.replace("const bf16_t* dO = p.dout + (long)b * p.do_sb + h * HD;", "const bf16_t* dO = p.dout + (long)b * p.do_sb + (long)h * p.nq * HD;", 1)
        b = b.replace("const long ldR = half ? p.do_ld : p.q_ld;", "const long ldR = HD;", 1)
    return a + b
