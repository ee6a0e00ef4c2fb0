"""Variant of the (branch-free) conv3d_halo2_kernel: the stage's LDS-DMA pieces issued INSIDE the MFMA section — the weight piece(s) behind block wblk, the halo piece behind block hblk."""


def patch(s, wblk=1, hblk=3):
    i = s.index("__global__ __launch_bounds__(64 * NW) void conv3d_halo2_kernel(ConvParams p)")
    j = s.index("// conv_out of the decoder", i)
    b = s[i:j]
    wl = "#pragma unroll\n                for (int i = 0; i < WP; ++i) dma_w((k + 3) % H2_RING, wk_use, i);\n"
    assert wl in b
    b = b.replace(wl, "")
    b = b.replace("            {\n                const int wk_use = min(wk, wk_last);\n", "            const int wk_use = min(wk, wk_last);\n            {\n")
    hl = "            if constexpr (hcnt > 0) {\n#pragma unroll\n                for (int i = 0; i < HP; ++i)\n                    if (i >= hfirst && i < hfirst + hcnt) dma_halo((g + 1) & 1, hsrc, i);\n            }\n"
    assert hl in b
    b = b.replace(hl, "")
    hook = "                if constexpr (ni % ASTEP == 0) read_a(std::integral_constant<int, nxt>{}, H2_AIMM(k + 1), ni / ASTEP);\n"
    assert hook in b
    ins = ("                __builtin_amdgcn_sched_barrier(0);\n"
           "                if constexpr (ni == %d) {\n#pragma unroll\n                    for (int i = 0; i < WP; ++i) dma_w((k + 3) %% H2_RING, wk_use, i);\n                }\n"
           "                if constexpr (ni == %d && hcnt > 0) {\n#pragma unroll\n                    for (int i = 0; i < HP; ++i)\n                        if (i >= hfirst && i < hfirst + hcnt) dma_halo((g + 1) & 1, hsrc, i);\n                }\n"
           "                __builtin_amdgcn_sched_barrier(0);\n" % (wblk, hblk))
    b = b.replace(hook, hook + ins)
    return s[:i] + b + s[j:]
