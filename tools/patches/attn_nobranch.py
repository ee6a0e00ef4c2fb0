"""attn_fwd_pp_kernel's tile loop without wave-uniform branches: the loop body specialised on the wave group (compile-time) and on "guarded" (the last two tiles of a segment:
DMA issue past the end, key mask) against "steady" (everything unconditional)."""


def patch(s, mode="both"):
    old_x = "        auto xseg = [&](int t) {\n"
    assert s.count(old_x) == 1
    s = s.replace(old_x, "        auto xseg = [&](int t, auto guardc) {\n")
    old_m = "            if ((t + 1) * KVBLK > S.nk) {\n"
    k = s.index(old_m, s.index("auto xseg = [&](int t, auto guardc)"))
    s = s[:k] + "            if constexpr (decltype(guardc)::value) if ((t + 1) * KVBLK > S.nk) {\n" + s[k + len(old_m):]
    old_d = "        auto dma_pair = [&](int u) {  // (K(u+1), V(u))\n            if (u + 1 < te) dmaK(u + 1);\n            if (u < te) dmaV(u);\n        };\n"
    assert s.count(old_d) == 1
    s = s.replace(old_d, old_d + "        auto dma_pair_all = [&](int u) { dmaK(u + 1); dmaV(u); };      // steady tiles: both exist\n")
    a = s.index("        for (int t = tb; t < te; ++t) {\n            // X(t): matrix segment\n")
    b = s.index("        pv(te - 1);                                                                // X(nt): last P.V")
    new = '''        // The tile loop carries NO wave-uniform branch: the body is instantiated per wave group (which half of the pair's DMA a wave issues, and where it waits for it) and
        // per "guarded" (the last two tiles of a range: issues past the end, key mask) / "steady".  The first version tested grp four times per tile, te twice per DMA
        // issue and nk once: 11 s_cbranch per tile and wave, most of them in the vector segment whose issue stream is the tile's critical path.
        auto tile = [&](auto gc, auto guardc, int t) {
            constexpr int G = decltype(gc)::value;
            constexpr bool GUARD = decltype(guardc)::value;
            // X(t): matrix segment
            if constexpr (G == 0) { if constexpr (GUARD) dma_pair(t); else dma_pair_all(t); }
            // the matrix segment runs at raised priority: its MFMA / ds_read issue slots are few (one per ~32 cycles) but each one the
            // partner's VALU stream delays idles the matrix pipe; measured -4..6 % (7.52 vs 7.94 ms same box); prio 1: -2 %, prio 3 = 2
            __builtin_amdgcn_s_setprio(2);
            xseg(t, guardc);
            __builtin_amdgcn_s_setprio(0);                         // (fencing this with sched_barrier(0) measured 2.5 % slower)
            if constexpr (G == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // pair t (issued one segment ago) has landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PP_BAR();
            // Y(t): vector segment
            if constexpr (G == 1) { if constexpr (GUARD) dma_pair(t + 1); else dma_pair_all(t + 1); }
            softmax();
            if constexpr (G == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // pair t has landed
            PP_BAR();
        };
        auto tiles = [&](auto gc) {
            int t = tb;
            for (; t < te - 2; ++t) tile(gc, std::false_type{}, t);
            for (; t < te; ++t) tile(gc, std::true_type{}, t);
        };
        if (grp == 0) tiles(std::integral_constant<int, 0>{});
        else tiles(std::integral_constant<int, 1>{});
'''
    if mode == "guard":      # one loop body for both groups (runtime grp tests stay), guards peeled
        new = new.replace("if constexpr (G == 0)", "if (grp == 0)").replace("if constexpr (G == 1)", "if (grp == 1)").replace("constexpr int G = decltype(gc)::value;", "")
        new = new.replace("        if (grp == 0) tiles(std::integral_constant<int, 0>{});\n        else tiles(std::integral_constant<int, 1>{});\n", "        tiles(std::integral_constant<int, 0>{});\n")
    if mode == "grp":        # specialised on the group, guards stay everywhere
        new = new.replace("for (; t < te - 2; ++t) tile(gc, std::false_type{}, t);", "")
    return s[:a] + new + s[b:]
