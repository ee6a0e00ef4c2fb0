"""Timing-only ablations of conv3d_halo2_kernel's stage loop (results are garbage): which part of a stage costs what.
  nodma: no LDS-DMA in the loop;  nohalo / now: no halo / no weight pieces;  noreads: no fragment reads in the loop;  nobar: no s_barrier;  samesrc: every piece reads one
  fixed L2-resident KiB (the DMA instructions stay, their address pattern goes)"""


def patch(s, what):
    i = s.index("__global__ __launch_bounds__(64 * NW) void conv3d_halo2_kernel(ConvParams p)")
    j = s.index("// Split-K epilogue", i) if "// conv_out of the decoder" not in s[i:] else s.index("// conv_out of the decoder", i)
    a, b, c = s[:i], s[i:j], s[j:]
    k0 = b.index("    for (int g0 = 0; g0 < ngroups; g0 += 4) {")
    k1 = b.index("    // ---- epilogue: bias, bf16 rounding before the residual add")
    pro, loop, epi = b[:k0], b[k0:k1], b[k1:]
    if what in ("nodma", "now"):
        loop = loop.replace("                for (int i = 0; i < WP; ++i) dma_w((k + 3) % H2_RING, wk_use, i);", "                for (int i = 0; i < WP; ++i) (void)wk_use;")
    if what in ("nodma", "nohalo"):
        loop = loop.replace("                    if (i >= hfirst && i < hfirst + hcnt) dma_halo((g + 1) & 1, hsrc, i);", "                    (void)hsrc;")
    if what == "noreads":
        loop = loop.replace("                if constexpr (ni + WD < 8) read_w(", "                if constexpr (false) read_w(").replace("                else read_w(", "                else if constexpr (false) read_w(")
        loop = loop.replace("                if constexpr (ni % ASTEP == 0) read_a(", "                if constexpr (false) read_a(")
    if what == "nobar":
        loop = loop.replace("            __builtin_amdgcn_s_barrier();", "")
    if what in ("nodma",):
        loop = loop.replace('            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WP + hcnt) : "memory");', "")
    if what == "mfmaonly":
        loop = loop.replace("                for (int i = 0; i < WP; ++i) dma_w((k + 3) % H2_RING, wk_use, i);", "                for (int i = 0; i < WP; ++i) (void)wk_use;")
        loop = loop.replace("                    if (i >= hfirst && i < hfirst + hcnt) dma_halo((g + 1) & 1, hsrc, i);", "                    (void)hsrc;")
        loop = loop.replace('            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WP + hcnt) : "memory");', "")
        loop = loop.replace("                if constexpr (ni + WD < 8) read_w(", "                if constexpr (false) read_w(").replace("                else read_w(", "                else if constexpr (false) read_w(")
        loop = loop.replace("                if constexpr (ni % ASTEP == 0) read_a(", "                if constexpr (false) read_a(")
        loop = loop.replace("            __builtin_amdgcn_s_barrier();", "")
    if what == "samesrc":
        pro = pro.replace("        const bf16_t* src = hoff[i] >= 0 ? hsrc + hoff[i] : p.zeros + (lane & 7) * 8;", "        const bf16_t* src = (hoff[i] >= 0 && p.T < 0) ? hsrc + hoff[i] : p.w + lane * 8;")
        pro = pro.replace("(const __attribute__((address_space(1))) void*)(p.w + woffs[i] + koff),", "(const __attribute__((address_space(1))) void*)(p.T < 0 ? p.w + woffs[i] + koff : p.w + lane * 8),")
    assert (pro + loop) != b[:k1], what
    return a + pro + loop + epi + c
