#!/bin/bash
# Build a named variant of the HIP library for same-box A/B timing:
#   [AB_FILE=attention|gemm|...] tools/ab_build.sh NAME [extra hipcc flags for that file]
# -> tokensgen_amd/csrc/variants/NAME.so (git-ignored, travels with gpurun); select it with TG_LIB_PATH.
set -e
cd "$(dirname "$0")/../tokensgen_amd/csrc"
name=$1; shift
file=${AB_FILE:-attention}
mkdir -p variants
make >/dev/null
extra=""
[ "$file" = attention ] && extra="-mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fno-slp-vectorize -I../../include -I. -Wno-unused-result $extra "$@" -c $file.hip -o variants/$name.o
objs=""
for f in gemm attention attention_bwd train norm elementwise vae; do
  if [ "$f" = "$file" ]; then objs="$objs variants/$name.o"; else objs="$objs $f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs api.o -o variants/$name.so
rm variants/$name.o
echo variants/$name.so
