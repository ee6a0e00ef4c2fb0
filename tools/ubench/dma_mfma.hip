// Micro-benchmark: does the issue of LDS-DMA pieces (global_load_lds, 1 KiB per wave-instruction) overlap with MFMAs of the SAME workgroup, and how does that depend on where
// the pieces are issued?  One workgroup of 8 waves (two per SIMD) per CU, the shape of the VAE's halo convolution stage: per iteration and wave 32 MFMA 16x16x32 (512 matrix
// cycles; 1024 per SIMD), NP DMA pieces, optionally 12 ds_read_b128 between the MFMAs, one s_barrier.  The source window is L2-resident (64 KiB per workgroup, re-read).
//   MODE 0: no DMA            1: all waves, pieces at the top of the iteration      2: all waves, pieces spread behind MFMA 8, 16, ...
//   3: anti-phase (waves 0-3 at the top, waves 4-7 behind their MFMAs)              4: only waves 0-3 issue (2 NP pieces each), at the top
//   5: only waves 0-3 issue, spread                                                 6: all waves at the top, but NO barrier (free running)
//   hipcc --offload-arch=gfx950 -O3 dma_mfma.hip -o dma_mfma && ./dma_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NP, bool LDSR, int SRC = 0>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)blockIdx.x * 65536 + lane * 16;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    bf16x8 fr[4] = {a, b, a, b};
    const unsigned la = (unsigned)(uintptr_t)smem + 65536 + lane * 16;
    int slot = 0;
    auto dma = [&](int i) {
        const int piece = (wave * 4 + i) & 31;
        const char* sp = SRC == 0 ? base + ((slot * 32 + piece) & 63) * 1024
                         : SRC == 1 ? src + ((size_t)blockIdx.x * 4096 + (size_t)(slot & 4095)) % 1000000 * 0 + (((size_t)slot * 256 + blockIdx.x) * 32 + piece) % (1u << 21) * 1024 + lane * 16
                         : SRC == 2 ? src + ((((size_t)slot * 256 + blockIdx.x) * 32 + piece) % (1u << 19)) * 4096 + (lane >> 2) * 256 + (lane & 3) * 16
                         : SRC == 3 ? src + (size_t)blockIdx.x * 0 + (size_t)((slot * 32 + piece) % 54) * 128 + (size_t)(lane >> 2) * 6912 + (lane & 3) * 16       // weights: 16 rows x 64 B, 6912-byte row stride, L2-resident (shared by all workgroups)
                         : SRC == 4 ? src + (size_t)blockIdx.x * 262144 + (size_t)((slot * 32 + piece) & 63) * 4096 + (lane >> 2) * 256 + (lane & 3) * 16        // 64-B rows at a 256-byte stride, L2-resident window per workgroup
                                    : src + ((((size_t)slot * 256 + blockIdx.x) * 32 + piece) % (1u << 15)) * 4096 + (lane >> 2) * 256 + (lane & 3) * 16;       // the same from a 128 MiB window (Infinity Cache)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                         (__attribute__((address_space(3))) void*)(smem + ((slot & 1) * 32 + piece) * 1024), 16, 0, 0);
    };
    const bool early = MODE == 1 || MODE == 6 || (MODE == 3 && wave < 4) || (MODE == 4 && wave < 4);
    const bool late = MODE == 3 && wave >= 4;
    const bool spread = MODE == 2 || (MODE == 5 && wave < 4);
    constexpr int NPW = (MODE == 4 || MODE == 5) ? 2 * NP : NP;       // pieces per issuing wave
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_sched_barrier(0);
        if (early)
#pragma unroll
            for (int i = 0; i < NPW; ++i) dma(i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 32; ++m) {
            acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[m & 3], fr[(m + 1) & 3], acc[m & 7], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (LDSR && (m % 3) == 1 && m < 30) {
                bf16x8& d = fr[(m / 3) & 3];
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(la), "n"((m / 3) * 1024));
            }
            if (LDSR && m == 31) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[0]), "+v"(fr[1]), "+v"(fr[2]), "+v"(fr[3]));
            if (spread && m % 8 == 7 && m / 8 < NPW) dma(m / 8);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (late)
#pragma unroll
            for (int i = 0; i < NPW; ++i) dma(i);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");     // last iteration's pieces have landed
        if (MODE != 6) __builtin_amdgcn_s_barrier();
        ++slot;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (iters < 0) sink[tid] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + acc[4][0] + acc[5][0] + acc[6][0] + acc[7][0];
}

template <int MODE, int NP, bool LDSR, int SRC = 0>
static void run(const char* src, float* sink, const char* what) {
    hipFuncSetAttribute((const void*)k<MODE, NP, LDSR, SRC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    k<MODE, NP, LDSR, SRC><<<256, 512, 144 * 1024>>>(src, 100, sink);
    hipEventRecord(e0);
    k<MODE, NP, LDSR, SRC><<<256, 512, 144 * 1024>>>(src, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d  pieces/wave %d  lds reads %d  src %d : %7.1f ns per iteration   %s\n", MODE, NP, (int)LDSR, SRC, ms * 1e6 / iters, what);
}

int main() {
    char* src; float* sink;
    hipMalloc(&src, (size_t)2 << 30); hipMemset(src, 1, (size_t)2 << 30); hipMalloc(&sink, 4096);
    run<0, 2, false>(src, sink, "no DMA");
    run<1, 2, false>(src, sink, "all waves, top");
    run<2, 2, false>(src, sink, "all waves, spread");
    run<3, 2, false>(src, sink, "anti-phase");
    run<4, 2, false>(src, sink, "waves 0-3 only, top");
    run<5, 2, false>(src, sink, "waves 0-3 only, spread");
    run<6, 2, false>(src, sink, "all waves, top, no barrier");
    run<0, 2, true>(src, sink, "no DMA, with LDS reads");
    run<1, 2, true>(src, sink, "all waves, top, with LDS reads");
    run<2, 2, true>(src, sink, "all waves, spread, with LDS reads");
    run<3, 2, true>(src, sink, "anti-phase, with LDS reads");
    run<5, 2, true>(src, sink, "waves 0-3 only, spread, with LDS reads");
    run<1, 4, false>(src, sink, "all waves, top, 4 pieces");
    run<2, 4, false>(src, sink, "all waves, spread, 4 pieces");
    run<1, 8, false>(src, sink, "all waves, top, 8 pieces (the GEMM's 64 KB per 1024 MFMA cycles/SIMD)");
    run<2, 4, true>(src, sink, "all waves, spread, 4 pieces, with LDS reads");
    run<1, 2, false, 3>(src, sink, "weight-shaped rows (64 B at 6912 B), L2-resident, top");
    run<2, 2, false, 3>(src, sink, "weight-shaped rows, spread");
    run<1, 2, false, 4>(src, sink, "64-B rows at 256 B, L2-resident window, top");
    run<2, 2, false, 4>(src, sink, "64-B rows at 256 B, L2-resident window, spread");
    run<1, 2, false, 5>(src, sink, "64-B rows at 256 B, 128 MiB window, top");
    run<2, 2, false, 5>(src, sink, "64-B rows at 256 B, 128 MiB window, spread");
    run<1, 1, false, 5>(src, sink, "64-B rows at 256 B, 128 MiB window, 1 piece per wave, top");
    run<1, 2, false, 1>(src, sink, "STREAMING source (2 GiB window), all waves, top");
    run<2, 2, false, 1>(src, sink, "STREAMING source, all waves, spread");
    run<1, 2, true, 1>(src, sink, "STREAMING source, all waves, top, with LDS reads");
    run<1, 2, false, 2>(src, sink, "STREAMING 64-byte rows at a 256-byte stride, all waves, top");
    run<2, 2, false, 2>(src, sink, "STREAMING 64-byte rows at a 256-byte stride, all waves, spread");
    run<1, 2, true, 2>(src, sink, "STREAMING 64-byte rows at a 256-byte stride, all waves, top, with LDS reads");
    run<1, 1, false, 2>(src, sink, "STREAMING 64-byte rows, 1 piece per wave");
    return 0;
}
