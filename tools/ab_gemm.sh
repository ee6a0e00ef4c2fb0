#!/bin/bash
# same-box A/B of GEMM variants: tools/ab_gemm.sh NAME[:ENV=VAL,...] ...   three interleaved rounds, per-GEMM ms and the sum
for r in 1 2 3; do for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  lib=$PWD/tokensgen_amd/csrc/variants/$v.so; [ "$v" = base ] && lib=$PWD/tokensgen_amd/libtokensgen_hip.so
  echo -n "$spec: "; env $envs TG_LIB_PATH=$lib timeout 300 python tools/bench_kernels.py gemm 2>&1 | grep '"ms"' | python3 -c "
import sys, json
ms = [json.loads(l)['ms'] for l in sys.stdin]
print(' '.join(f'{m:.3f}' for m in ms), 'sum %.3f' % sum(ms))"
done; done
