"""Patch for tools/variant_build.sh: cycle counters (s_memtime) around the phases of the attention-backward tile loops.
   tools/variant_build.sh bwdprobe attention_bwd "__import__('runpy').run_path('../../tools/patches/bwd_probe.py', {'s': s})['s']"
   TG_LIB_PATH=.../variants/bwdprobe.so python tools/bwd_phase_probe.py
Workgroup 0, waves 0 and 4 (dK/dV) / 0 and 2 (dQ) leave their sums in the first floats of dV / dQ (the gradients of that launch are garbage there)."""
s = s  # noqa: F821  (injected)

def rep(a, b, count=1):
    global s
    assert s.count(a) >= 1, a
    s = s.replace(a, b, count)

TCK = "#define TCK(i) do { TG_SB(); const long long n_ = __builtin_amdgcn_s_memtime(); tc[i] += n_ - tprev; tprev = n_; TG_SB(); } while (0)\n"
rep("constexpr int BT = 32;", TCK + "constexpr int BT = 32;")
# ---- dK/dV (ping-pong kernel) ----
rep("    for (int it = 0; it < ntile; ++it) {\n        // ---------------- X(it): matrix segment ----------------\n",
    "    long long tc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tprev = __builtin_amdgcn_s_memtime();\n    for (int it = 0; it < ntile; ++it) {\n        TCK(3);\n")
rep("        xseg(cb, pb, false);\n        __builtin_amdgcn_s_setprio(0);\n        BWD_BAR();",
    "        xseg(cb, pb, false);\n        __builtin_amdgcn_s_setprio(0);\n        TCK(0);\n        BWD_BAR();\n        TCK(1);")
rep("        stash(wb);  ", "        TCK(2);\n        stash(wb);  ")
rep("        fetch();\n        wb = ", "        TCK(4);\n        fetch();\n        TCK(5);\n        wb = ")
rep("            *c = (p.accumulate & 2) ? *c + vv : vv;\n        }\n    }\n}\n\n// ---- (3') dQ",
    "            *c = (p.accumulate & 2) ? *c + vv : vv;\n        }\n    }\n    __syncthreads();\n"
    "    if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) { long long* o_ = (long long*)p.dv + (wave >> 2) * 8; for (int i = 0; i < 6; ++i) o_[i] = tc[i]; o_[6] = ntile; }\n}\n\n// ---- (3') dQ")
# ---- dQ ----
rep("    fetch(0);\n    stash(0);\n    __syncthreads();\n    for (int it = 0; it < ntile; ++it) {\n        const int buf = it & 1;",
    "    fetch(0);\n    stash(0);\n    __syncthreads();\n    long long tc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long tprev = __builtin_amdgcn_s_memtime();\n    for (int it = 0; it < ntile; ++it) {\n        const int buf = it & 1;")
rep('        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bK[0][0].v), "+v"(bK[0][1].v), "+v"(bK[1][0].v), "+v"(bK[1][1].v));',
    '        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bK[0][0].v), "+v"(bK[0][1].v), "+v"(bK[1][0].v), "+v"(bK[1][1].v));\n        TCK(0);')
rep("            mfma_pair(st, dpt, aK, aV, qf[qb], of[qb]);", "            mfma_pair(st, dpt, aK, aV, qf[qb], of[qb]);\n            TCK(1);")
rep("#pragma unroll\n            for (int t = 0; t < 2; ++t)\n#pragma unroll\n                for (int db = 0; db < 2; ++db)\n                    mfma_acc(dq[qb][db], dA[t].v, bK[t][db].v);\n        }",
    "            TCK(2);\n#pragma unroll\n            for (int t = 0; t < 2; ++t)\n#pragma unroll\n                for (int db = 0; db < 2; ++db)\n                    mfma_acc(dq[qb][db], dA[t].v, bK[t][db].v);\n            TCK(3);\n        }")
rep("        if (it + 1 < ntile) stash(buf ^ 1);\n        __syncthreads();\n    }\n    asm volatile(\"s_nop 15\" ::: \"memory\");\n#pragma unroll\n    for (int qb = 0; qb < 2; ++qb)",
    "        if (it + 1 < ntile) stash(buf ^ 1);\n        TCK(4);\n        __syncthreads();\n        TCK(5);\n    }\n    asm volatile(\"s_nop 15\" ::: \"memory\");\n#pragma unroll\n    for (int qb = 0; qb < 2; ++qb)")
rep("                *a = (p.accumulate & 1) ? *a + vq : vq;\n            }\n        }\n}",
    "                *a = (p.accumulate & 1) ? *a + vq : vq;\n            }\n        }\n    __syncthreads();\n"
    "    if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 2)) { long long* o_ = (long long*)p.dq + (wave >> 1) * 8; for (int i = 0; i < 6; ++i) o_[i] = tc[i]; o_[6] = ntile; }\n}")
