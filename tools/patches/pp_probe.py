"""Patch for tools/variant_build.sh: s_memtime stamps in the ping-pong one-kernel attention backward (attn_bwd_fused_pp_kernel), summed per wave over all tiles:
[0] X (matrix segment), [1] barrier after X, [2] Y top (exchange reads, vmcnt wait, stage write, fetch, exchange add / store / request), [3] Y softmax + dS^T writes,
[4] Y tail (check, sample, prefetch issue), [5] barrier after Y.
   tools/variant_build.sh ppprobe attention_bwd "__import__('runpy').run_path('../../tools/patches/pp_probe.py', {'s': s})['s']"
   TG_LIB_PATH=.../variants/ppprobe.so python tools/pp_probe_run.py        (the dV of that run is garbage: the counters land there)"""
s = s  # noqa: F821  (injected)


def rep(a, b, n=1):
    global s
    assert s.count(a) >= 1, a
    s = s.replace(a, b, n)


i0 = s.index("__global__ __launch_bounds__(512) void attn_bwd_fused_pp_kernel(FusedParams fp) {")
head, s = s[:i0], s[i0:]
T = "{ TG_SB(); const long long n_ = __builtin_amdgcn_s_memtime(); tcs[%d] += n_ - tprev; tprev = n_; TG_SB(); }"
rep("    const int nit = ntile + PP_EXTRA;\n", "    const int nit = ntile + PP_EXTRA;\n    long long tcs[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};\n    long long tprev = __builtin_amdgcn_s_memtime();\n")
rep("        __builtin_amdgcn_s_setprio(0);\n        BWD_BAR();\n        // ---------------- Y(it): vector segment ----------------\n",
    "        __builtin_amdgcn_s_setprio(0);\n        " + T % 0 + "\n        BWD_BAR();\n        " + T % 1 + "\n")
rep("        asm volatile(\"s_waitcnt vmcnt(0)\" : \"+v\"(g0), \"+v\"(gseed), \"+v\"(ldv), \"+v\"(cval));\n        // stage write of tile",
    "        " + T % 6 + "\n        asm volatile(\"s_waitcnt vmcnt(0)\" : \"+v\"(g0), \"+v\"(gseed), \"+v\"(ldv), \"+v\"(cval));\n        " + T % 7 + "\n        // stage write of tile")
rep("        wb = (wb + 1) & (RING - 1);\n        if (doW) {\n            asm volatile(\"s_waitcnt lgkmcnt(1)\" : \"+v\"(e0), \"+v\"(e1));",
    "        wb = (wb + 1) & (RING - 1);\n        " + T % 8 + "\n        if (doW) {\n            asm volatile(\"s_waitcnt lgkmcnt(1)\" : \"+v\"(e0), \"+v\"(e1));\n            " + T % 9)
rep("        // P = exp2(S), dS = P o dP -> the A operands of X(it + 1); dS^T -> LDS for X(it + 2)\n", "        " + T % 2 + "\n")
rep("        if (wg == 0) {                                      // the group's first wave keeps its chain", "        " + T % 3 + "\n        if (wg == 0) {                                      // the group's first wave keeps its chain")
rep("        xprefetch((it + 1) & 1);\n        BWD_BAR();\n    };\n", "        xprefetch((it + 1) & 1);\n        " + T % 4 + "\n        BWD_BAR();\n        " + T % 5 + "\n    };\n")
rep("            *c = (p.accumulate & 2) ? *c + vv : vv;\n        }\n    }\n",
    "            *c = (p.accumulate & 2) ? *c + vv : vv;\n        }\n    }\n    __syncthreads();\n"
    "    if (lane == 0 && (wave == 0 || wave == 1 || wave == 4 || wave == 5) && (blockIdx.x == 40 || blockIdx.x == 320 || blockIdx.x == 2400)) {\n"
    "        long long* o_ = (long long*)fp.p.dv + ((blockIdx.x == 40 ? 0 : blockIdx.x == 320 ? 1 : 2) * 4 + (wave & 1) + 2 * (wave >> 2)) * 12;\n"
    "        for (int i = 0; i < 10; ++i) o_[i] = tcs[i];\n        o_[10] = nit;\n    }\n")
s = head + s
