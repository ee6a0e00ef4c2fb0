#!/bin/bash
# Build a named variant of the HIP library for same-box A/B timing:  tools/ab_build.sh NAME [extra hipcc flags for attention.hip]
# -> tokensgen_amd/csrc/variants/NAME.so (git-ignored, travels with gpurun); select it with TG_LIB_PATH.
set -e
cd "$(dirname "$0")/../tokensgen_amd/csrc"
name=$1; shift
mkdir -p variants
make >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -mllvm -amdgpu-mfma-vgpr-form "$@" -c attention.hip -o variants/$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC gemm.o variants/$name.o norm.o elementwise.o vae.o api.o -o variants/$name.so
rm variants/$name.o
echo variants/$name.so
