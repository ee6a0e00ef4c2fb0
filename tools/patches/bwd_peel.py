"""attn_bwd_fused_kernel: the first two tiles (no dQ write / request / signal yet) peeled out of the steady loop, so the `it >= 1` / `it >= 2` tests leave it."""


def patch(s):
    i = s.index("void attn_bwd_fused_kernel(FusedParams fp)")
    a = s.index("    for (int it = 0; it < ntile; ++it) {\n", i)
    b = s.index("    // drain the dQ pipeline", a)
    body = s[a:b]
    assert body.rstrip().endswith("}")
    inner = body[len("    for (int it = 0; it < ntile; ++it) {\n"):body.rstrip().rfind("}")]
    assert "continue;" not in inner and "break;" not in inner
    inner = inner.replace("if (it >= 2) e_write(it - 2);", "if (!EARLY || it >= 2) e_write(it - 2);")
    inner = inner.replace("if (it >= 1) e_request(it - 1);", "if (!EARLY || it >= 1) e_request(it - 1);")
    inner = inner.replace("if (it >= 2) e_signal(it - 2);", "if (!EARLY || it >= 2) e_signal(it - 2);")
    new = ("    auto tile_iter = [&](auto earlyc, int it) {\n        constexpr bool EARLY = decltype(earlyc)::value;\n" + inner +
           "    };\n    {\n        int it = 0;\n        for (; it < min(2, ntile); ++it) tile_iter(std::true_type{}, it);\n        for (; it < ntile; ++it) tile_iter(std::false_type{}, it);\n    }\n")
    return s[:a] + new + s[b:]
