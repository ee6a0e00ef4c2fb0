// Micro-benchmark: issue cost (shader cycles per wave-instruction) of the VALU ops the attention softmax is made of,
// with 1 and 2 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ void k(float* out, long long* cyc, int iters) {
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 0.5f, b1 = 0.25f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_exp_f32 %0, %1\n v_exp_f32 %2, %3" : "=v"(a0), "=v"(a1) : "v"(a2), "v"(a3));) }
        if (OP == 1) { REP16(asm volatile("v_add_f32 %0, %1, %2\n v_add_f32 %3, %4, %2" : "=v"(a0), "=v"(a1) : "v"(b0), "v"(a2), "v"(a3));) }
        if (OP == 2) { REP16(asm volatile("v_pk_add_f32 %0, %1, %2\n v_pk_add_f32 %3, %4, %2" : "=v"(*(double*)&a0), "=v"(*(double*)&a2) : "v"(*(double*)&a4), "v"(*(double*)&a6), "v"(*(double*)&b0));) }
        if (OP == 3) { REP16(asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %3, %4, %2" : "=v"(a0), "=v"(a1) : "v"(b0), "v"(a2), "v"(a3));) }
        if (OP == 4) { REP16(asm volatile("v_max3_f32 %0, %1, %2, %3\n v_max3_f32 %4, %1, %2, %5" : "=v"(a0), "=v"(a1) : "v"(b0), "v"(a2), "v"(a3), "v"(a4));) }
        if (OP == 5) { REP16(asm volatile("v_fma_f32 %0, %1, %2, %3\n v_fma_f32 %4, %1, %2, %5" : "=v"(a0), "=v"(a1) : "v"(b0), "v"(a2), "v"(a3), "v"(a4));) }
        if (OP == 6) { REP16(asm volatile("v_pk_fma_f32 %0, %1, %2, %2\n v_pk_fma_f32 %3, %4, %2, %2" : "=v"(*(double*)&a0), "=v"(*(double*)&a2) : "v"(*(double*)&a4), "v"(*(double*)&a6), "v"(*(double*)&b0));) }
        if (OP == 7) { REP16(asm volatile("v_exp_f32 %0, %1\n v_add_f32 %2, %3, %3" : "=v"(a0), "=v"(a1) : "v"(a2), "v"(a3));) }   // trans + plain alternating
        if (OP == 8) { REP16(asm volatile("v_exp_f32 %0, %1\n v_add_f32 %2, %3, %3\n v_add_f32 %4, %3, %3\n v_add_f32 %5, %3, %3" : "=v"(a0), "=v"(a1), "=v"(a4), "=v"(a5) : "v"(a2), "v"(a3));) }
        if (OP == 9) { REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %2, %3" : "=v"(a0), "=v"(a1) : "v"(a2), "v"(a3));) }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b1;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP>
void run(const char* name, int per_iter) {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 8);
    for (int threads : {256, 512}) {
        const int iters = 2000;
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-28s waves/SIMD=%d  cycles per wave-instruction = %.2f  (per SIMD: %.2f)\n", name, threads / 256,
               (double)c / (iters * 16.0 * per_iter), (double)c / (iters * 16.0 * per_iter) / (threads / 256));
    }
}

int main() {
    run<0>("v_exp_f32", 2); run<1>("v_add_f32", 2); run<2>("v_pk_add_f32", 2); run<3>("v_cvt_pk_bf16_f32", 2);
    run<4>("v_max3_f32", 2); run<5>("v_fma_f32", 2); run<6>("v_pk_fma_f32", 2); run<7>("exp+add pair", 2); run<8>("exp+3add quad", 4);
    run<9>("v_mov_b32", 2);
    return 0;
}
