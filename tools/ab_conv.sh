#!/bin/bash
# same-box A/B of library variants on the conv micro cases: tools/ab_conv.sh "c128 c256_128" NAME...   (three interleaved rounds)
cases=$1; shift
for r in 1 2 3; do for v in "$@"; do
  echo -n "$v "; TG_LIB_PATH=$PWD/tokensgen_amd/csrc/variants/$v.so timeout 300 python tools/conv_micro.py 20 $cases 2>/dev/null | python -c "
import sys, json
print(' '.join('%s %.4f %.6f' % (d['case'], d['ms'], d['checksum']) for d in map(json.loads, sys.stdin)))"
done; done
