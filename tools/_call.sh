timeout 120 python tools/clock_probe.py attn_bwd 5 2>&1 | tail -6
timeout 120 python tools/clock_probe.py attn 4 2>&1 | tail -4
