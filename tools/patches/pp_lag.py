"""Patch for tools/variant_build.sh: every key block of the one-kernel attention backward leaves s_memtime at its iterations 100 and 300 and its number of exchange polls in the
unused words 16..21 of counter line [head][tile = its block index] (tools/pp_lag_run.py reads them: distance between consecutive key blocks of a head in tile times).
   tools/variant_build.sh pplag attention_bwd "__import__('runpy').run_path('../../tools/patches/pp_lag.py', {'s': s})['s']" """
s = s  # noqa: F821  (injected)


def rep(a, b):
    global s
    assert s.count(a) == 1, a
    s = s.replace(a, b)


i0 = s.index("__global__ __launch_bounds__(512) void attn_bwd_fused_pp_kernel(FusedParams fp) {")
head, s = s[:i0], s[i0:]
rep("        // ---------------- Y(it): vector segment ----------------\n",
    "        if ((it == 100 || it == 300) && tid == 0) { long long t_ = __builtin_amdgcn_s_memrealtime(); *(long long*)(cnt_base + ((long)hb * ntile + blk) * 32 + (it == 100 ? 16 : 18)) = t_; }\n")
rep("                int spin = 0;\n", "                int spin = 0; if (lane == 0) atomicAdd(cnt_base + ((long)hb * ntile + blk) * 32 + 20, 1);\n")
s = head + s
