// Micro-benchmark: L2 -> CU fill rate per CU, all 256 CUs busy, 8 waves per CU, 8 x 1 KiB per wave in flight:
//   (a) LDS-DMA (global_load_lds_dwordx4) and (b) plain global_load_dwordx4 into registers, for row segments of 64 B (16 rows per
//   wave-instruction: the K=32 GEMM stage), 128 B (K=64) and 1024 B (contiguous), from a per-CU window of WIN bytes (64 KiB: L2/L1
//   resident; 1 MiB x 256 CUs: beyond L2).   hipcc --offload-arch=gfx950 -O3 dma_rate.hip -o dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SEG, bool DMA>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, long win, int iters, long long* cyc, float* sink) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = SEG / 16, RPP = 64 / LPR;      // lanes per row segment, rows per piece
    const long row_stride = 512;                        // rows 512 B apart: a 64-B / 128-B segment is a part of one or two lines
    const long rows = win / row_stride;
    const int r = lane / LPR, c = lane % LPR;
    const char* base = src + (long)blockIdx.x * win + c * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 va[8], vb[8];      // register variant: the destination registers must stay live until the counted wait has passed
    long long t0 = __builtin_amdgcn_s_memtime();
    auto issue = [&](int it, f32x4* v) {
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const long row = ((long)(it * 8 + p) * 8 * RPP + wave * RPP + r) % rows;
            const char* a = base + row * row_stride + ((it >> 3) & 3) * (SEG < 512 ? SEG : 0);
            if (DMA) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a,
                                                 (__attribute__((address_space(3))) void*)(smem + (p & 3) * 8192 + wave * 1024), 16, 0, 0);
            } else {
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[p]) : "v"(a) : "memory");
            }
        }
    };
    auto consume = [&](f32x4* v) {
        if (!DMA) {
#pragma unroll
            for (int p = 0; p < 8; ++p) asm volatile("" :: "v"(v[p]));
        }
    };
    issue(0, va);
    for (int it = 1; it + 1 < iters; it += 2) {
        issue(it, vb);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        consume(va);
        issue(it + 1, va);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        consume(vb);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    consume(va);
    if (iters < 0) sink[threadIdx.x] = smem[threadIdx.x] + acc[0];
}

template <int SEG, bool DMA>
void run(const char* name, const char* src, long win) {
    long long* cyc; float* sink;
    hipMalloc(&cyc, 8); hipMalloc(&sink, 4096);
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<SEG, DMA>), dim3(256), dim3(512), 32768, 0, src, win, iters, cyc, sink);
    hipError_t e = hipDeviceSynchronize();
    long long c = 1; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s %6.1f B/clk/CU  (%s)\n", name, 8.0 * 8 * 1024 * iters / (double)c, hipGetErrorString(e));
    fflush(stdout);
}

int main() {
    char* src; const size_t bytes = 512UL << 20;
    if (hipMalloc(&src, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(src, 1, bytes); hipDeviceSynchronize();
    printf("src = %p\n", (void*)src); fflush(stdout);
    run<64, true>("LDS-DMA    64-B segments, 64 KiB window per CU", src, 64 << 10);
    run<128, true>("LDS-DMA   128-B segments, 64 KiB window per CU", src, 64 << 10);
    run<1024, true>("LDS-DMA  1024-B pieces,   64 KiB window per CU", src, 64 << 10);
    run<64, false>("register   64-B segments, 64 KiB window per CU", src, 64 << 10);
    run<128, false>("register  128-B segments, 64 KiB window per CU", src, 64 << 10);
    run<1024, false>("register 1024-B pieces,   64 KiB window per CU", src, 64 << 10);
    run<64, true>("LDS-DMA    64-B segments, 1 MiB window per CU (beyond L2)", src, 1 << 20);
    run<128, true>("LDS-DMA   128-B segments, 1 MiB window per CU (beyond L2)", src, 1 << 20);
    run<64, false>("register   64-B segments, 1 MiB window per CU (beyond L2)", src, 1 << 20);
    run<128, false>("register  128-B segments, 1 MiB window per CU (beyond L2)", src, 1 << 20);
    return 0;
}
