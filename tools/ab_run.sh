#!/bin/bash
# same-box A/B: tools/ab_run.sh WHAT NAME...   runs tools/bench_kernels.py WHAT for every variant, three interleaved rounds
what=$1; shift
for r in 1 2 3; do for v in "$@"; do
  echo -n "$v: "; TG_LIB_PATH=$PWD/tokensgen_amd/csrc/variants/$v.so timeout 300 python tools/bench_kernels.py $what 2>&1 | grep -v amdgpu.ids | head -1
done; done
