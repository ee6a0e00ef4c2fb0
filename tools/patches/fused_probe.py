"""Patch for tools/variant_build.sh: s_memtime counters in the one-kernel attention backward (per tile: cycles of the dQ-exchange section at the top of an
iteration, cycles of the whole iteration, blocking polls).
   tools/variant_build.sh fprobe attention_bwd "__import__('runpy').run_path('../../tools/patches/fused_probe.py', {'s': s})['s']"
   TG_LIB_PATH=.../variants/fprobe.so python tools/fused_probe_run.py        (the gradients of that run are garbage: the counters land in dV)"""
s = s  # noqa
def rep(a, b):
    global s
    assert s.count(a) >= 1, a
    s = s.replace(a, b, 1)
rep("    constexpr int CNT_PAD = 32;", "    long long nspin = 0;\n    constexpr int CNT_PAD = 32;")
rep("    for (int it = 0; it < ntile; ++it) {\n        const int buf = it % 3, sbuf = (it + 2) % 3;",
    "    long long tcE = 0, tcAll = 0, tprev = __builtin_amdgcn_s_memtime(); const long long tstart = tprev;\n    for (int it = 0; it < ntile; ++it) {\n        const int buf = it % 3, sbuf = (it + 2) % 3;\n        { TG_SB(); tprev = __builtin_amdgcn_s_memtime(); TG_SB(); }")
rep("        if (it >= 1) e_request(it - 1);                      // (tile it - 1 was checked before the previous barrier)\n",
    "        if (it >= 1) e_request(it - 1);\n        { TG_SB(); const long long n_ = __builtin_amdgcn_s_memtime(); tcE += n_ - tprev; TG_SB(); }\n")
rep("            if (lane == 0 && cval != blk) {\n                int spin = 0;", "            if (cval != blk) ++nspin;\n            if (lane == 0 && cval != blk) {\n                int spin = 0;")
rep("    // drain the dQ pipeline (tiles", "    tcAll = __builtin_amdgcn_s_memtime() - tstart;\n    // drain the dQ pipeline (tiles")
rep("    asm volatile(\"s_nop 15\" ::: \"memory\");\n#pragma unroll\n    for (int db = 0; db < 2; ++db) {\n        float* DK = p.dk + (long)b * p.dk_sb + h * HD + db * 32 + j;\n        float* DV = p.dv + (long)b * p.dv_sb + h * HD + db * 32 + j;\n#pragma unroll\n        for (int r = 0; r < 16; ++r) {\n            const int key = kw0 + acc_row(r, hi);\n            if (key >= p.nk) continue;\n            float* a = DK + (long)key * p.dk_ld;\n            float* c = DV + (long)key * p.dv_ld;\n            const float vk = dk[db][r] * p.scale, vv = dv[db][r];\n            *a = (p.accumulate & 2) ? *a + vk : vk;\n            *c = (p.accumulate & 2) ? *c + vv : vv;\n        }\n    }\n}\n\n// probe: does workgroup",
    "    __syncthreads();\n    if (lane == 0 && (wave == 0 || wave == 5) && (blockIdx.x == 8 * 5 || blockIdx.x == 8 * 40 || blockIdx.x == 8 * 300)) { long long* o_ = (long long*)fp.p.dv + ((blockIdx.x == 40 ? 0 : blockIdx.x == 320 ? 1 : 2) * 2 + (wave ? 1 : 0)) * 4; o_[0] = tcE; o_[1] = tcAll; o_[2] = nspin; o_[3] = ntile; }\n}\n\n// probe: does workgroup")
