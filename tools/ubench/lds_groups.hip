// Micro-benchmark: which lanes of a wave64 ds_read_b128 are served together (share LDS-array cycles), and what the halo convolution kernel's
// fragment-read patterns cost.  One workgroup of 16 waves per CU on every CU; every lane reads 16 bytes at a per-lane byte offset from a table,
// 64 reads per timed batch; cycles per read come from s_memtime around the loop.
//   (1) pair scan: all lanes at the conflict-free linear pattern (lane * 16), except that lane X is moved onto lane 0's BANKS in another row
//       (+1024 bytes): it costs an extra array cycle iff lane X is served in lane 0's group.
//   (2) the kernel's patterns: voxel rows at a stride of 80 / 96 / 160 bytes (lane = row l & 15, slot l >> 4); weight rows of 64 bytes with the
//       slot XOR-swizzled by (row >> 2) & 3 or by (0, 3, 2, 1)[(row >> 2) & 3].
//   hipcc --offload-arch=gfx950 -O3 lds_groups.hip -o lds_groups
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void k(const int* __restrict__ offs, int iters, long long* cyc, float* sink) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) ((float*)smem)[i] = (float)i;
    __syncthreads();
    const unsigned a = (unsigned)(uintptr_t)smem + (unsigned)offs[lane];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        f32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(a));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
    if (iters < 0) sink[threadIdx.x] = acc[0];
}

static double run(const std::vector<int>& offs) {
    static int* d = nullptr; static long long* cyc = nullptr; static float* sink = nullptr;
    if (!d) { hipMalloc(&d, 256); hipMalloc(&cyc, 8); hipMalloc(&sink, 4096); hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
    hipMemcpy(d, offs.data(), 256, hipMemcpyHostToDevice);
    const int iters = 2000;
    k<<<256, 1024, 160 * 1024>>>(d, 10, cyc, sink);
    k<<<256, 1024, 160 * 1024>>>(d, iters, cyc, sink);
    long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    return (double)c / (iters * 16.0 * 16.0);      // cycles per wave-instruction with 16 waves sharing the CU's LDS
}

int main() {
    std::vector<int> lin(64);
    for (int l = 0; l < 64; ++l) lin[l] = l * 16;
    const double base = run(lin);
    printf("linear (conflict-free): %.2f cycles per ds_read_b128 per wave (16 waves / CU)\n", base);
    printf("lanes that conflict with lane 0 when moved onto its banks (+%.2f cycles or more):", 0.5);
    for (int x = 1; x < 64; ++x) {
        std::vector<int> o = lin;
        o[x] = 1024;                                 // lane 0's banks, another row
        const double t = run(o);
        if (t > base + 0.5) printf(" %d", x);
    }
    printf("\n");
    auto vox = [&](int stride, const char* name) {
        std::vector<int> o(64);
        for (int l = 0; l < 64; ++l) o[l] = (l & 15) * stride + (l >> 4) * 16;
        printf("voxel rows, stride %3d B%s: %.2f\n", stride, name, run(o));
    };
    vox(80, " (shipped)"); vox(96, ""); vox(112, ""); vox(144, ""); vox(160, ""); vox(64, " (no pad)"); vox(128, " (no pad, 64 ch)");
    for (int variant = 0; variant < 2; ++variant) {
        std::vector<int> o(64);
        for (int l = 0; l < 64; ++l) {
            const int row = l & 15, x = (row >> 2) & 3, g = variant ? ((4 - x) & 3) : x;
            o[l] = row * 64 + (((l >> 4) ^ g) << 4);
        }
        printf("weight rows 64 B, slot ^ %s: %.2f\n", variant ? "(0,3,2,1)[row>>2]" : "(row>>2)&3 (shipped)", run(o));
    }
    {   // the 256^2 GEMM's layout: 128-byte rows, slot ^ ((row >> 1) & 7), lane = (row l & 15, slot pair l >> 4)
        std::vector<int> o(64);
        for (int l = 0; l < 64; ++l) { const int row = l & 15; o[l] = row * 128 + ((((l >> 4)) ^ ((row >> 1) & 7)) << 4); }
        printf("GEMM rows 128 B, slot ^ (row>>1)&7: %.2f\n", run(o));
    }
    return 0;
}
