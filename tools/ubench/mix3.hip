// Ceiling probe for an attention-forward restructuring: how fast can a SIMD alternate matrix and softmax-like vector work when 2 or 3 waves share it
// and nothing else (no LDS, no barriers) is in the way?  Per loop iteration a wave issues the MFMAs and the VALU mix one KV tile costs it.
//   A: 8 waves / workgroup (2 per SIMD), 64 query rows per wave: 36 x v_mfma_f32_32x32x16_bf16 + 64 v_exp + 32 v_cvt_pk + 32 v_pk_add   (the shipped kernel's mix)
//   B: 12 waves / workgroup (3 per SIMD), 48 query rows per wave: 48 x v_mfma_f32_16x16x32_bf16 + 48 v_exp + 24 v_cvt_pk + 24 v_pk_add
//   A16: A's shape on 72 x v_mfma_f32_16x16x32_bf16 (same flops, a quarter of the accumulator registers per instruction)
//   "skewed": the second wave of every SIMD starts with the vector block, i.e. half an iteration late (what the shipped kernel's barriers enforce)
// Prints shader cycles per iteration and cycles per query row per SIMD (the shipped kernel measures ~27 with everything included).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mix3.hip -o tools/ubench/mix3 && tools/ubench/mix3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ unsigned pk(float a, float b) { unsigned r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <int MODE>
__global__ __launch_bounds__(MODE == 0 ? 512 : 768) void k(float* out, long long* cyc, int iters) {
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {2, 3, 4, 5, 6, 7, 8, 9};
    const float seed = threadIdx.x * 1e-6f;
    float lsum = 0.f;
    long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {
        f32x16 o[4], s[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) { o[i][r] = 0.f; s[i][r] = seed; }
        for (int it = 0; it < iters; ++it) {
            __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int m = 0; m < 4; ++m) {       // 16 PV + 16 S + 4 seed = 36
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, o[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, s[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, s[i], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            f32x2 acc = {0.f, 0.f};
            unsigned px = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float e0 = __builtin_amdgcn_exp2f(s[i][r] * 1e-30f), e1 = __builtin_amdgcn_exp2f(s[i][r + 1] * 1e-30f);
                    px ^= pk(e0, e1);
                    acc += f32x2{e0, e1};
                    s[i][r] = e0 * 1e-30f; s[i][r + 1] = e1 * 1e-30f;
                }
            lsum += acc[0] + acc[1];
            a[0] = (short)(px & 0x3f);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) lsum += o[i][r] + s[i][r];
    } else {
        f32x4 o[12], s[12];
        for (int i = 0; i < 12; ++i) for (int r = 0; r < 4; ++r) { o[i][r] = 0.f; s[i][r] = seed; }
        for (int it = 0; it < iters; ++it) {
            __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int m = 0; m < 2; ++m) {       // 24 PV + 24 S = 48
#pragma unroll
                for (int i = 0; i < 12; ++i) o[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, o[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 12; ++i) s[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, s[i], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            f32x2 acc = {0.f, 0.f};
            unsigned px = 0;
#pragma unroll
            for (int i = 0; i < 12; ++i)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const float e0 = __builtin_amdgcn_exp2f(s[i][r] * 1e-30f), e1 = __builtin_amdgcn_exp2f(s[i][r + 1] * 1e-30f);
                    px ^= pk(e0, e1);
                    acc += f32x2{e0, e1};
                    s[i][r] = e0 * 1e-30f; s[i][r + 1] = e1 * 1e-30f;
                }
            lsum += acc[0] + acc[1];
            a[0] = (short)(px & 0x3f);
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int i = 0; i < 12; ++i) for (int r = 0; r < 4; ++r) lsum += o[i][r] + s[i][r];
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = lsum;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// A16: 2 waves per SIMD, 64 rows per wave, 16x16x32 MFMAs; SKEW: waves 4-7 run the vector block first
template <int M16, int SKEW>
__global__ __launch_bounds__(512) void k2(float* out, long long* cyc, int iters) {
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {2, 3, 4, 5, 6, 7, 8, 9};
    const float seed = threadIdx.x * 1e-6f;
    float lsum = 0.f;
    f32x4 o[16], s[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) { o[i][r] = 0.f; s[i][r] = seed; }
    auto xs = [&]() {
        __builtin_amdgcn_s_setprio(2);
        if (M16) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {       // 32 PV + 32 S (+ 8 for the seed's share) = 72
#pragma unroll
                for (int i = 0; i < 16; ++i) o[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, o[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 16; ++i) s[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, s[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, s[i], 0, 0, 0);
        } else {
            f32x16* o16 = (f32x16*)o; f32x16* s16 = (f32x16*)s;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int i = 0; i < 4; ++i) o16[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, o16[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) s16[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, s16[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) s16[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, s16[i], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto ys = [&]() {
        f32x2 acc = {0.f, 0.f};
        unsigned px = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const float e0 = __builtin_amdgcn_exp2f(s[i][r] * 1e-30f), e1 = __builtin_amdgcn_exp2f(s[i][r + 1] * 1e-30f);
                px ^= pk(e0, e1);
                acc += f32x2{e0, e1};
                s[i][r] = e0 * 1e-30f; s[i][r + 1] = e1 * 1e-30f;
            }
        lsum += acc[0] + acc[1];
        a[0] = (short)(px & 0x3f);
        __builtin_amdgcn_sched_barrier(0);
    };
    long long t0 = __builtin_amdgcn_s_memtime();
    if (SKEW && threadIdx.x >= 256) ys();
    for (int it = 0; it < iters; ++it) { xs(); ys(); }
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) lsum += o[i][r] + s[i][r];
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = lsum;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, int rows_per_wave, int waves_per_simd) {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() {
        if (MODE < 2) hipLaunchKernelGGL(k<(MODE < 2 ? MODE : 0)>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        else hipLaunchKernelGGL((k2<(MODE & 1), ((MODE >> 1) & 1) ^ 1>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    };
    launch();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double per_it = (double)c / iters;
    printf("%-44s cycles/iteration %.0f   cycles per row per SIMD %.2f   wall ns per row per SIMD %.2f\n", name, per_it,
           per_it / (rows_per_wave * waves_per_simd), ms * 1e6 / iters / (rows_per_wave * waves_per_simd));
}

int main() {
    run<0>("A: 2 waves/SIMD, 32x32x16, 64 rows/wave", 512, 64, 2);
    run<1>("B: 3 waves/SIMD, 16x16x32, 48 rows/wave", 768, 48, 3);
    run<4>("A skewed: 32x32x16", 512, 64, 2);
    run<5>("A16 skewed: 16x16x32", 512, 64, 2);
    run<2>("A in phase (k2): 32x32x16", 512, 64, 2);
    run<3>("A16 in phase: 16x16x32", 512, 64, 2);
    run<4>("A skewed: 32x32x16 (again)", 512, 64, 2);
    run<5>("A16 skewed: 16x16x32 (again)", 512, 64, 2);
    return 0;
}
