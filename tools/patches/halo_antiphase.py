"""Variant of conv3d_halo2_kernel: the upper half of the waves issues its LDS-DMA pieces BEHIND its MFMA blocks (anti-phase with its SIMD partner).
usage in tools/variant_build.sh: __import__('sys').path.insert(0,'../../tools/patches') or __import__('halo_antiphase').patch(s)"""


def patch(s, mode="half"):
    a = s.index("            if (w_iss) {\n#pragma unroll\n                for (int i = 0; i < WP; ++i) dma_w((k + 3) % H2_RING, wk, i);")
    b = s.index("            const bool more = st + 1 < nst;", a)
    body = s[a:b]
    lam = "            auto issue = [&]() {\n" + body + "            };\n"
    cond_early = {"half": "wave < NW / 2", "odd": "(wave & 1) == 0", "late": "false"}[mode]
    s = s[:a] + lam + "            if (%s) issue();\n" % cond_early + s[b:]
    tail = "            __builtin_amdgcn_sched_barrier(0);\n            {\n                bf16x8 &a0 = fa[nxt][0]"
    c = s.index(tail, a)
    s = s[:c] + "            __builtin_amdgcn_sched_barrier(0);\n            if (!(%s)) issue();\n" % cond_early + s[c:]
    return s


def spread(s, wblk=1, hblk=3):
    """All waves issue together, but INSIDE the MFMA section: the weight piece(s) behind block wblk, the halo piece behind block hblk."""
    a = s.index("            if (w_iss) {\n#pragma unroll\n                for (int i = 0; i < WP; ++i) dma_w((k + 3) % H2_RING, wk, i);")
    m = s.index("            if constexpr (tap == 0) {                       // this group's stages stage the NEXT group's halo", a)
    h = s.index("            if (h_iss) {\n#pragma unroll\n                for (int i = 0; i < HP; ++i)", m)
    b = s.index("            const bool more = st + 1 < nst;", h)
    wpart, mid, hpart = s[a:m], s[m:h], s[h:b]
    lam = ("            auto issue_w = [&]() {\n" + wpart + "            };\n" + mid + "            auto issue_h = [&]() {\n" + hpart + "            };\n")
    s = s[:a] + lam + s[b:]
    hook = "                if constexpr (ni % ASTEP == 0) {\n                    if (more) read_a("
    c = s.index(hook, a)
    ins = ("                if constexpr (ni == %d) issue_w();\n                if constexpr (ni == %d) issue_h();\n"
           "                __builtin_amdgcn_sched_barrier(0);\n" % (wblk, hblk))
    return s[:c] + ins + s[c:]
