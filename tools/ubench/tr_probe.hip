// Semantics probe of ds_read_b64_tr_b16 (gfx950): a [32 rows][72-element stride] bf16 tile holds value = row * 256 + col; every lane issues one
// transpose read at the address the attention-backward B operand would use and prints what it got.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/tr_probe.hip -o tools/ubench/tr_probe && tools/ubench/tr_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void probe(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t tile[32 * 72];
    for (int i = threadIdx.x; i < 32 * 72; i += 64) tile[i] = (uint16_t)((i / 72) * 256 + (i % 72));
    __syncthreads();
    const int lane = threadIdx.x, a = lane & 15, hi = lane >> 5, dhalf = (lane >> 4) & 1;
    // 16-lane group: rows q0 + (a >> 2), 4-element chunk (a & 3) of the 16 columns [dhalf * 16, +16); q0 = 4 * hi
    const uint32_t addr = (uint32_t)(uintptr_t)tile + ((4 * hi + (a >> 2)) * 72 + dhalf * 16 + (a & 3) * 4) * 2;
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[lane * 4 + 0] = r.x & 0xffff; out[lane * 4 + 1] = r.x >> 16; out[lane * 4 + 2] = r.y & 0xffff; out[lane * 4 + 3] = r.y >> 16;
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[l * 4 + e] / 256, h[l * 4 + e] % 256); printf("\n"); }
    return 0;
}
