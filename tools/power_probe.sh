#!/bin/bash
# The same launches on random-normal and on all-zero operands (same box, same process order): how much of each MFMA kernel's time is the socket power cap.
#   tools/power_probe.sh TAG -> gpurun_out/TAG_power_probe.json
tag=${1:-r4}
out=gpurun_out/${tag}_power_probe.json
{
  for z in 0 1; do
    TG_BENCH_ZERO=$z python tools/bench_kernels.py gemm attn_dit attn_bwd 2>/dev/null | grep '^{' | sed "s/^{/{\"zero_operands\": $z, /"
    TG_CONV_MICRO_ZERO=$z python tools/conv_micro.py 20 c128 c256_128 c256 2>/dev/null | grep '^{' | sed "s/^{/{\"zero_operands\": $z, /"
  done
} | python -c "
import sys, json
rows = [json.loads(l) for l in sys.stdin]
json.dump(rows, open('$out', 'w'), indent=1)
print(len(rows), 'rows ->', '$out')"
