def patch(s, hm=False):
    a = "const int ts = it - 2 - lag;"
    assert s.count(a) == 1; s = s.replace(a, "const int ts = it - 1 - lag;")
    a = '"s"(pcnt), "i"(-2 * CNT_PAD * 4) : "memory");'
    assert s.count(a) == 1; s = s.replace(a, '"s"(pcnt), "i"(-1 * CNT_PAD * 4) : "memory");')
    a = "        xprefetch((it + 1) & 1);\n        BWD_BAR();\n    };"
    assert s.count(a) == 1
    s = s.replace(a, '        xprefetch((it + 1) & 1);\n        if (FAST && !first) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n        BWD_BAR();\n    };')
    if hm:
        for a, b in [("char* sbDQr = (char*)(p.dq + (long)b * p.dq_sb + h * HD);", "char* sbDQr = (char*)(p.dq + (long)b * p.dq_sb + (long)h * p.nq * HD);"),
                     ("const long stepDQb = (long)BT * p.dq_ld * 4;", "const long stepDQb = (long)BT * HD * 4;"),
                     ("const uint32_t voDQ = (uint32_t)(((long)er * p.dq_ld + ec) * 4);", "const uint32_t voDQ = (uint32_t)(((long)er * HD + ec) * 4);"),
                     ("- qlast) * p.dq_ld + ec) * 4);", "- qlast) * HD + ec) * 4);")]:
            assert s.count(a) == 1, a; s = s.replace(a, b)
    return s
