// How many DIRTY bytes does an XCD's L2 keep?  Each XCD's 32 workgroups (one per CU: 64 KB of LDS requested) sweep a private window of W bytes
// with 16-byte read-add-write per lane, R times; a line is touched again one sweep later.  With a write-back L2 of 4 MB the memory-side writes
// (rocprofv3 --pmc WRITE_SIZE) should stay near W per XCD until W approaches 4 MB.  Second mode: the same lines addressed as the attention backward's
// dQ rows ([rows][12 KB stride], 256 bytes used per row).
//   hipcc --offload-arch=gfx950 -O3 -o l2_dirty l2_dirty.hip ; rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d out -- ./l2_dirty
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void sweep(float* base, long window_bytes, int reps, long xcd_stride_bytes) {
    extern __shared__ char pad[];
    const int xcd = blockIdx.x & 7, wg = blockIdx.x >> 3, nwg = gridDim.x >> 3;
    char* win = (char*)base + (long)xcd * xcd_stride_bytes;
    const long chunks = window_bytes / 4096;               // 4 KB = one workgroup-wide access (256 lanes x 16 B)
    for (int r = 0; r < reps; ++r)
        for (long c = wg; c < chunks; c += nwg) {
            long off;
            if (MODE == 0) off = c * 4096 + threadIdx.x * 16;
            else off = (c * 16 + (threadIdx.x >> 4)) * 12288L + (threadIdx.x & 15) * 16;      // 16 rows of 256 B, 12 KB apart
            f32x4* p = (f32x4*)(win + off);
            f32x4 v = *p;
            v += 1.0f;
            *p = v;
        }
    if (threadIdx.x == 1000) pad[0] = 1;
}

int main(int argc, char** argv) {
    const long maxw = 16L << 20;
    const long stride = maxw * 48;                          // mode 1 spreads 16 rows per 4 KB chunk over 192 KB
    float* buf;
    if (hipMalloc(&buf, stride * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, stride * 8);
    hipFuncSetAttribute((const void*)sweep<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10);
    hipFuncSetAttribute((const void*)sweep<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10);
    const long ws[] = {256L << 10, 512L << 10, 1L << 20, 2L << 20, 3L << 20, 4L << 20, 6L << 20, 8L << 20};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (long w : ws) {
            const int reps = (int)((1L << 30) / w);         // 1 GB of read-add-write per XCD in every case
            hipEventRecord(e0);
            if (mode == 0) sweep<0><<<256, 256, 100 << 10>>>(buf, w, reps, stride);
            else sweep<1><<<256, 256, 100 << 10>>>(buf, w, reps, stride);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d window %ld KB reps %d: %.3f ms, %.1f GB/s read+write at the L2\n", mode, w >> 10, reps, ms, 2.0 * 8 * w * reps / ms / 1e6);
        }
    return 0;
}
