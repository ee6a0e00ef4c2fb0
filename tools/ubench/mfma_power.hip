// Which MFMA shape sustains more bf16 FLOP/s under the socket power cap: v_mfma_f32_32x32x16_bf16 or v_mfma_f32_16x16x32_bf16?
// One wave per SIMD, 256 accumulator registers per lane (as in the 4-wave GEMM), operands from registers (random bits), no memory.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power.hip -o tools/ubench/mfma_power && tools/ubench/mfma_power [seconds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <unistd.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256) void k32(const bf16x8* in, float* out, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x * 8 + i]; b[i] = in[threadIdx.x * 8 + 4 + i]; }
    f32x16 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k16(const bf16x8* in, float* out, int iters) {
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[threadIdx.x * 16 + i]; b[i] = in[threadIdx.x * 16 + 8 + i]; }
    f32x4 acc[8][8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    std::vector<short> h(256 * 16 * 8);
    for (auto& v : h) v = (short)(0x3c00 + (rand() & 0x3ff) * ((rand() & 1) ? 1 : -1) + ((rand() & 1) << 15));   // ~ +-[0.007, 0.03] bf16 values, random mantissas
    bf16x8* din; float* dout;
    hipMalloc(&din, h.size() * 2); hipMalloc(&dout, 256 * 256 * 4);
    hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int which = 0; which < 2; ++which) {
        const int iters = 20000;       // per launch: 16 x 32x32x16 (or 64 x 16x16x32) MFMAs per iteration = 2 * 128*128*16 flop per wave
        const double flop_launch = 256.0 * 4 * iters * 2.0 * 128 * 128 * (which == 0 ? 16 : 32);
        auto run = [&] { if (which == 0) hipLaunchKernelGGL(k32, dim3(256), dim3(256), 0, 0, din, dout, iters);
                         else hipLaunchKernelGGL(k16, dim3(256), dim3(256), 0, 0, din, dout, iters); };
        run(); hipDeviceSynchronize();
        volatile bool stop = false;
        std::thread smp([&] { while (!stop) { system("rocm-smi --showclocks --showpower --csv 2>/dev/null | grep card0 | cut -d, -f6,10"); usleep(400000); } });
        auto t0 = std::chrono::steady_clock::now(); int n = 0; double dt = 0;
        do { for (int i = 0; i < 8; ++i) run(); hipDeviceSynchronize(); n += 8;
             dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } while (dt < secs);
        stop = true; smp.join();
        printf("%s: %.0f TFLOP/s sustained over %.1f s (%d launches)\n", which == 0 ? "32x32x16" : "16x16x32", flop_launch * n / dt / 1e12, dt, n);
    }
    return 0;
}
