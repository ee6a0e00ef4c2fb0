// Shared device helpers for the tokensgen gfx950 kernels (bf16 storage, fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define TG_OK 0
#define TG_ERR_ARG (-1)
#define TG_ERR_SHAPE (-2)
#define TG_ERR_ALIGN (-3)
#define TG_ERR_HIP (-100)

// set the thread-local last-error string (api.cpp) and return `code`
extern "C" int tg_set_error(int code, const char* fmt, ...);

#define TG_REQUIRE(cond, code, ...)              \
    do {                                         \
        if (!(cond)) return tg_set_error((code), __VA_ARGS__); \
    } while (0)

#define TG_LAUNCH_CHECK(name)                                                              \
    do {                                                                                   \
        hipError_t e__ = hipGetLastError();                                                \
        if (e__ != hipSuccess)                                                             \
            return tg_set_error(TG_ERR_HIP - (int)e__, "%s: launch failed: %s", (name), hipGetErrorString(e__)); \
    } while (0)

static inline bool tg_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- per-DEVICE launch prerequisites (api.cpp).  hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the CU count belong to a device, not to the
// process: a host that drives several GPUs from one process (the reference's own topology, infer_cogvideo_mp_fifo.py:191,211-213) must get them on
// every device it launches on.  Nothing here is keyed by anything but the CURRENT device id, and nothing changes a result.
int tg_device_cus(void);                                    // multiprocessor count of the current device (cached per device id)
// `slot` = a TgOnce the launcher owns (one per kernel instantiation), one bit per device id.  The bit is set only AFTER hipFuncSetAttribute has returned,
// so a second host thread never skips the attribute while the first is still inside the call (both may set it: idempotent).
struct TgOnce { unsigned long long mask[4]; };              // one bit per device id (< 256); zero-initialised static
bool tg_done_on_device(TgOnce& once);
void tg_mark_on_device(TgOnce& once);
#define TG_DYN_LDS(kernel, bytes)                                                                                      \
    do {                                                                                                               \
        static TgOnce once__;                                                                                          \
        if (!tg_done_on_device(once__)) {                                                                              \
            (void)hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes));     \
            tg_mark_on_device(once__);                                                                                 \
        }                                                                                                              \
    } while (0)

// ---- dispatch overrides for the cross-check tests (tg_debug_set in the header): which of two product kernels of the same op a launcher picks.
// The library reads no environment variable; the defaults are the shipped path.
enum TgKnob { TG_KNOB_ATTN_PP_MIN_WG, TG_KNOB_ATTN_FIXEDM, TG_KNOB_ATTN_SPLIT, TG_KNOB_GEMM_W4, TG_KNOB_CONV_SPLITK, TG_KNOB_CONV_HALO, TG_KNOB_CONV_W4,
              TG_KNOB_COUNT };
long tg_knob(TgKnob k);

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float bf16lo_to_f32(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16hi_to_f32(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }

// round-to-nearest-even pack of two floats into (lo | hi<<16)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ bf16_t f32_to_bf16(float v) { return (bf16_t)(pack_bf16x2(v, 0.f) & 0xffffu); }
// value of v after a round trip through bf16 (mimics the reference's per-op bf16 rounding)
__device__ __forceinline__ float round_bf16(float v) { return bf16lo_to_f32(pack_bf16x2(v, 0.f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// F.gelu(approximate="tanh"): 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3)  ==  x / (1 + exp(-2u)).
// One v_exp + one v_rcp (1 ulp each; the result is rounded to bf16 right after) instead of a correctly rounded fp32 division:
// the GELU of the FF1 epilogue is 64 Ki elements per 256x256 tile on VALUs that have nothing to overlap with.  Shared by the GEMM epilogue
// (gemm.hip) and the stand-alone activation pass of the training forward (train.hip: tg_act mode 2) so that both give the same bits.
__device__ __forceinline__ float gelu_tanh(float x) {
    const float x2 = x * x;
    const float t = x * (-2.f * 0.7978845608028654f * 1.4426950408889634f) * (1.f + 0.044715f * x2);   // -2u log2(e)
    return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(t));   // t -> +inf: x * 0; t -> -inf: x
}

// dy * d/dx gelu_tanh(x).  tanh(u) = 1 - 2 / (1 + exp(2u)): one v_exp + one v_rcp (exp -> inf gives 1, -> 0 gives -1) instead of tanhf's ~40 instructions.  Shared by the
// stand-alone pass (train.hip: tg_act mode 1) and the dgrad GEMM's epilogue (gemm.hip: TG_EPI_MUL_GELU_GRAD).
// contract(off): the function is inlined into two very different kernels and must give the same bits in both (fused multiply-adds are formed per context otherwise)
__device__ __forceinline__ float gelu_tanh_bwd(float v, float dyv) {
#pragma clang fp contract(off)
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u = k0 * (v + k1 * v * v * v), th = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.f * 1.4426950408889634f * u));
    return dyv * (0.5f * (1.f + th) + 0.5f * v * (1.f - th * th) * k0 * (1.f + 3.f * k1 * v * v));
}

// XCD-aware block remap (8 XCDs, block b lands on XCD b%8): give each XCD a contiguous chunk.
// Bijective for any nwg (cdna guide T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}
