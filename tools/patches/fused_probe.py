"""Patch for tools/variant_build.sh: s_memtime counters in the one-kernel attention backward, per tile and wave: [0] the dQ-exchange section at the top of the
iteration, [1] the check (sample wait + polls) before the barrier, [2] the wait in front of the barrier + the barrier, [3] the whole iteration, [4] polls.
   tools/variant_build.sh fprobe attention_bwd "__import__('runpy').run_path('../../tools/patches/fused_probe.py', {'s': s})['s']"
   TG_LIB_PATH=.../variants/fprobe.so python tools/fused_probe_run.py        (the gradients of that run are garbage: the counters land in dV)"""
s = s  # noqa: F821  (injected)

def rep(a, b):
    global s
    assert s.count(a) >= 1, a
    s = s.replace(a, b, 1)

T = "{ TG_SB(); const long long n_ = __builtin_amdgcn_s_memtime(); tc[%d] += n_ - tprev; tprev = n_; TG_SB(); }"
rep("    constexpr int CNT_PAD = 32;", "    long long tc[5] = {0, 0, 0, 0, 0};\n    constexpr int CNT_PAD = 32;")
rep("    for (int it = 0; it < ntile; ++it) {\n        const int buf = it % 3, sbuf = (it + 2) % 3;",
    "    long long tprev = __builtin_amdgcn_s_memtime();\n    for (int it = 0; it < ntile; ++it) {\n        const int buf = it % 3, sbuf = (it + 2) % 3;")
rep("        if (it >= 1) e_request(it - 1);                      // (tile it - 1 was checked before the previous barrier)\n",
    "        if (it >= 1) e_request(it - 1);\n        " + T % 0 + "\n")
rep("            if (lane == 0 && cval != blk) {\n                int spin = 0;", "            if (cval != blk) ++tc[4];\n            if (lane == 0 && cval != blk) {\n                int spin = 0;")
rep("        e_check(it);                                        // wave 0: key block blk - 1 has completed tile it (needed from the top of the next iteration on)\n",
    "        " + T % 3 + "\n        e_check(it);\n        " + T % 1 + "\n")
rep("        if (it >= 2) e_signal(it - 2);\n", "        " + T % 2 + "\n        if (it >= 2) e_signal(it - 2);\n")
rep("            *(f32x4*)(sDQ + ((it & 1) * BT + 16 * qh + t16) * DQLD + 16 * dblk + 4 * g4) = acc;\n        }\n    }\n", "            *(f32x4*)(sDQ + ((it & 1) * BT + 16 * qh + t16) * DQLD + 16 * dblk + 4 * g4) = acc;\n        }\n        " + T % 3 + "\n    }\n")
rep("            *c = (p.accumulate & 2) ? *c + vv : vv;\n        }\n    }\n}\n\n// One-time probe",
    "            *c = (p.accumulate & 2) ? *c + vv : vv;\n        }\n    }\n    __syncthreads();\n"
    "    if (lane == 0 && (wave == 0 || wave == 5) && (blockIdx.x == 40 || blockIdx.x == 320 || blockIdx.x == 2400)) { long long* o_ = (long long*)fp.p.dv + ((blockIdx.x == 40 ? 0 : blockIdx.x == 320 ? 1 : 2) * 2 + (wave ? 1 : 0)) * 8; for (int i = 0; i < 5; ++i) o_[i] = tc[i]; o_[5] = ntile; }\n}\n\n// One-time probe")
