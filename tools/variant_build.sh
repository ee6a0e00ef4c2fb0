#!/bin/bash
# Build a library variant from a PATCHED COPY of one kernel source (experiments stay out of the product tree):
#   tools/variant_build.sh NAME FILE 'python expression transforming s (the source text)'
# -> tokensgen_amd/csrc/variants/NAME.so; select with TG_LIB_PATH (tools/ab_run.sh WHAT NAME...).
set -e
name=$1; file=$2; expr=$3
cd "$(dirname "$0")/../tokensgen_amd/csrc"
mkdir -p variants
make >/dev/null
python3 - "$file" "$name" "$expr" <<'PY'
import sys
f, name, expr = sys.argv[1:4]
s = open(f + ".hip").read()
s2 = eval(expr, {"s": s})
assert s2 != s, "the patch did not change the source"
open("variants/%s_%s.hip" % (name, f), "w").write(s2)
PY
extra=""
[ "$file" = attention ] && extra="-mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize -I../../include -I. -Wno-unused-result $extra -c variants/${name}_$file.hip -o variants/$name.o
objs=""
for f in gemm attention attention_bwd train norm elementwise vae; do
  if [ "$f" = "$file" ]; then objs="$objs variants/$name.o"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs api.o -o variants/$name.so
rm variants/$name.o variants/${name}_$file.hip
echo variants/$name.so
