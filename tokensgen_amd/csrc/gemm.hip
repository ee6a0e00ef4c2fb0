// bf16 GEMM for the DiT linear layers:  C[b, m, n] = epilogue( sum_k A[b, m, k] * W[n, k] + bias[n] )
//
// Replaces the reference's nn.Linear calls on the hot path (attention_processor.py:2009-2018 QKV +
// vip QKV, :2143 to_out; diffusers FeedForward at cogvideox_transformer_3d.py:316,322; the modulation
// linears normalization.py:447,483; embeddings.py:516-536,953-965).  nn.Linear weights are [N, K]
// row-major, i.e. both operands are K-contiguous, which is exactly what MFMA fragments want.
//
// gfx950 design (v1 structure: 128x128x64 tile, 4 waves as 2x2, each 64x64 = 4x4 MFMA 16x16x32):
//   * global -> LDS by `global_load_lds_dwordx4` (LDS-DMA, 1 KiB per wave instruction, no VGPR trip);
//     the LDS image is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE
//     address and again on the ds_read side (same involution: 16-B slot ^= (row>>1)&7).
//   * double-buffered LDS, tile t+1 in flight while tile t is multiplied.
//   * operands are fed swapped (W rows as the MFMA "A", activations as "B") so each lane ends up with
//     4 consecutive output columns of one row -> 8-byte bf16 stores and vector bias/gate loads.
//   * block ids are remapped XCD-aware and grouped 8 m-tiles x n so the tiles resident on one XCD share
//     A/W panels in that XCD's L2.
//   * M edge: rows >= M are clamped on load and predicated on store.  N % 128 == 0, K % 64 == 0.
#include <stdlib.h>

#include <type_traits>
#include "common.h"
#include "tokensgen_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // A + W
#ifndef TG_GROUP_M
#define TG_GROUP_M 8
#endif
constexpr int GROUP_M = TG_GROUP_M;

struct GemmParams {
    const bf16_t* A; long lda; long sAb;
    const bf16_t* W; long ldw;
    const bf16_t* bias;
    bf16_t* C; long ldc; long sCb;
    const bf16_t* R; long ldr; long sRb;          // residual (EPI_GATE_RES)
    int M, N, K, batch;
    tg_group_table g;                             // gate lookup (EPI_GATE_RES)
    // optional second problem of the same N, K, batch, leading dimensions and epilogue (256^2 kernel only): its tiles are appended to
    // the persistent tile list, so the two problems share ONE partial last round of CUs instead of paying one each
    const bf16_t* A2; const bf16_t* W2; const bf16_t* bias2; bf16_t* C2; long sAb2, sCb2; int M2;
    // optional (4-wave kernel only): output columns n >= vt_col0 are NOT written to C but transposed to Vt[b][n - vt_col0][m]
    // (row stride vt_ld elements, a multiple of 64 >= M; columns M..vt_ld-1 are written as zeros): the V third of a QKV projection
    // delivered in the [head][64][keys] layout tg_attention_fwd reads
    bf16_t* Vt; bf16_t* Vt2; long vt_ld, vt_ld2; int vt_col0;
    int tiles1;                                   // tiles of the first problem
    int group_m;                                  // m-tiles per n sweep of the 256^2 kernel's tile order (see launch())
};

// gelu_tanh(): common.h
typedef float f32x2v __attribute__((ext_vector_type(2)));
// two elements at a time: the polynomial part maps onto v_pk_mul_f32 / v_pk_fma_f32 (same operations, same results as gelu_tanh)
__device__ __forceinline__ f32x2v gelu_tanh2(f32x2v x) {
    const f32x2v x2 = x * x;
    const f32x2v t = x * (-2.f * 0.7978845608028654f * 1.4426950408889634f) * (1.f + 0.044715f * x2);
    const f32x2v d = f32x2v{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.f;
    return x * f32x2v{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- tile assignment: XCD-aware remap, then grouped (GROUP_M m-tiles per n sweep) ----
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
    const int per_batch = tiles_m * tiles_n;
    const int nwg = per_batch * p.batch;
    int t = xcd_remap(blockIdx.x, nwg);
    const int b = t / per_batch;
    t -= b * per_batch;
    const int per_group = GROUP_M * tiles_n;
    const int grp = t / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int in_g = t - grp * per_group;
    const int tm = first_m + in_g % gsz, tn = in_g / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    const bf16_t* Ab = p.A + (long)b * p.sAb;

    // ---- per-lane LDS-DMA source pointers (4 pieces of A, 4 of W per k-tile) ----
    // wave-instruction i covers tile rows [wave*32 + i*8, +8): lane -> row += lane>>3, physical slot lane&7
    const char* srcA[4];
    const char* srcW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wave * 32 + i * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ ((r >> 1) & 7);          // logical 16-B slot stored at this physical one
        const int ma = min(m0 + r, p.M - 1);
        srcA[i] = (const char*)(Ab + (long)ma * p.lda + slot * 8);
        srcW[i] = (const char*)(p.W + (long)(n0 + r) * p.ldw + slot * 8);
    }
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE_BYTES + wave * (32 * 128);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + (long)kt * (BK * 2)),
                                             (__attribute__((address_space(3))) void*)(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW[i] + (long)kt * (BK * 2)),
                                             (__attribute__((address_space(3))) void*)(base + TILE_BYTES + i * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes inside a tile) ----
    // row = w?*64 + f*16 + (lane&15); logical slot = ks*4 + (lane>>4); physical = slot ^ ((row>>1)&7)
    int offA[4][2], offW[4][2];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ra = wm * 64 + f * 16 + (lane & 15);
            const int rw = wn * 64 + f * 16 + (lane & 15);
            const int sl = ks * 4 + (lane >> 4);
            offA[f][ks] = ra * 128 + ((sl ^ ((ra >> 1) & 7)) << 4);
            offW[f][ks] = rw * 128 + ((sl ^ ((rw >> 1) & 7)) << 4);
        }
    }

    f32x4 acc[4][4];   // [ni][mi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* tA = smem + cur * STAGE_BYTES;
        const char* tW = tA + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fw[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                fa[f] = *(const bf16x8*)(tA + offA[f][ks]);
                fw[f] = *(const bf16x8*)(tW + offW[f][ks]);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[ni][mi], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane holds D[n = (lane>>4)*4 + r][m = lane&15] per fragment ----
    bf16_t* Cb = p.C + (long)b * p.sCb;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm * 64 + mi * 16 + (lane & 15);
        if (m >= p.M) continue;
        const bf16_t* gate_row = nullptr;
        if (EPI == TG_EPI_BIAS_GATE_RES) {
            const int g = p.g.tok_group[m];
            gate_row = (const bf16_t*)p.g.mod + (long)b * p.g.mod_batch_stride + (long)p.g.row[g] * p.g.mod_ld + p.g.gate_col[g];
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wn * 64 + ni * 16 + (lane >> 4) * 4;
            float v[4] = {acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]};
            if (p.bias) {
                const uint2 bb = *(const uint2*)(p.bias + n);
                v[0] += bf16lo_to_f32(bb.x); v[1] += bf16hi_to_f32(bb.x);
                v[2] += bf16lo_to_f32(bb.y); v[3] += bf16hi_to_f32(bb.y);
            }
            if (EPI == TG_EPI_BIAS_GELU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = gelu_tanh(round_bf16(v[i]));
            } else if (EPI == TG_EPI_BIAS_SILU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = silu(round_bf16(v[i]));
            } else if (EPI == TG_EPI_BIAS_GATE_RES) {
                const uint2 gg = *(const uint2*)(gate_row + n);
                const uint2 rr = *(const uint2*)(p.R + (long)b * p.sRb + (long)m * p.ldr + n);
                v[0] = bf16lo_to_f32(rr.x) + bf16lo_to_f32(gg.x) * v[0];
                v[1] = bf16hi_to_f32(rr.x) + bf16hi_to_f32(gg.x) * v[1];
                v[2] = bf16lo_to_f32(rr.y) + bf16lo_to_f32(gg.y) * v[2];
                v[3] = bf16hi_to_f32(rr.y) + bf16hi_to_f32(gg.y) * v[3];
            }
            uint2 o;
            o.x = pack_bf16x2(v[0], v[1]);
            o.y = pack_bf16x2(v[2], v[3]);
            *(uint2*)(Cb + (long)m * p.ldc + n) = o;
        }
    }
}

// ================================================================================================
// 256x256 tile, 8 waves (2 M-groups x 4 N-columns, 128x64 per wave), "ping-pong" schedule over a 4-deep LDS ring.
//
// Why this shape: a 128x128 tile needs ~39 TB/s of L2->LDS fill at full MFMA rate (more than the chip has) and every
// wave stalls together at the per-k-tile barrier.  Here the traffic per MFMA is halved and the two waves that share a
// SIMD (wave w of group 0 and wave w+4 of group 1) alternate roles every phase: while one issues its 8 MFMAs
// (v_mfma_f32_32x32x16_bf16, 4 independent accumulators), the other does its ds_read_b128 fragment loads.  Group 1 simply
// runs one barrier behind group 0, so per SIMD one wave is always in its MFMA phase.
//
// Why a 4-deep ring of K=32 stages (and not 2 x K=64): timing ablations on MI355X showed the schedule without
// its LDS-DMA runs at 1.87 PFLOP/s-equivalent while the DMA path alone takes 72 % of the full kernel's time at only 40 % of
// L2 bandwidth — the fill is bound by bytes in flight x miss latency (19 % of the pieces are compulsory L2 misses served
// by MALL/HBM, and vmcnt retires in order).  Four 32-KiB stages keep THREE stages (up to 12 KiB per wave, 96 KiB per CU)
// in flight for ~10 barrier intervals before their first reader instead of one stage for ~4.
//
//   per wave per stage (K=32): 2 phases = {LOAD ds_reads | s_barrier | 8 MFMA + 2 LDS-DMA pieces | s_barrier}
//     phase 0: reads W frags (2 n-blocks x 2 k-steps) + A frags of rows [0,64) of its half   -> 8 ds_read_b128
//     phase 1: reads A frags of rows [64,128)                                                 -> 4 ds_read_b128
//   LDS stage = A[256 rows][64 B] | W[256 rows][64 B]; 16-B slot XOR-swizzled by (row>>2)&3 (conflict-free b128 reads,
//   applied on the DMA source address since LDS-DMA writes lane-linear).  DMA piece = 16 rows x 64 B.
//   During stage s every wave issues its 4 pieces of stage s+3 (2 per MFMA phase, from inside the MFMA stream); every LOAD
//   phase ends with `s_waitcnt vmcnt(6)`: everything older than the last three issue phases has landed, i.e. a stage is
//   complete one full phase before its first reader, and the barrier that follows publishes it.  The ring slot of stage
//   s+3 is that of stage s-1, whose last reader (group 1, LOAD phase 1) finished two intervals before the first overwrite.
// ================================================================================================
constexpr int BM2 = 256, BN2 = 256, BK2 = 32, NS2 = 4;
constexpr int OPER2_BYTES = BM2 * BK2 * 2;       // 16 KiB per operand per stage
constexpr int STAGE2_BYTES = 2 * OPER2_BYTES;    // 32 KiB
constexpr int RING2_BYTES = NS2 * STAGE2_BYTES;  // 128 KiB

template <int EPI>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmParams p) {
    // PERSISTENT: one workgroup per CU walks the tile list (tile += gridDim.x).  The first three stages of the NEXT output
    // tile are put in flight before the epilogue of the current one, and the epilogue goes through LDS (ring slot 3) so that
    // every global access of the C tile is a full 128-byte row segment (16 B per lane, 8 lanes per row).
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wn = wave & 3;

    const int tiles_n = p.N / BN2;
    const int tiles_m1 = (p.M + BM2 - 1) / BM2, tiles_m2 = p.A2 ? (p.M2 + BM2 - 1) / BM2 : 0;
    const int nwg1 = tiles_m1 * tiles_n * p.batch;
    const int nwg = nwg1 + tiles_m2 * tiles_n * p.batch;

    // ---- LDS-DMA pieces of this wave: issue phase 0 -> A pieces 2*wave, 2*wave+1 ; phase 1 -> W pieces 2*wave, 2*wave+1 ----
    const char* src[2][2];
    int cb = 0, cm0 = 0, cn0 = 0, csec = 0;   // batch index, origin and problem (0/1) of the tile `src` points at
    const int prow = wave * 32 + (lane >> 2);                 // + i*16 : tile row of this lane's 16-B chunk
    auto set_tile = [&](int id) {
        csec = id >= nwg1;
        const int tiles_m = csec ? tiles_m2 : tiles_m1;
        const int per_batch = tiles_m * tiles_n;
        const int Mc = csec ? p.M2 : p.M;
        int t = csec ? xcd_remap(id - nwg1, nwg - nwg1) : xcd_remap(id, nwg1);
        cb = t / per_batch;
        t -= cb * per_batch;
        const int per_group = p.group_m * tiles_n;
        const int gi = t / per_group;
        const int first_m = gi * p.group_m;
        const int gsz = min(tiles_m - first_m, p.group_m);
        const int in_g = t - gi * per_group;
        cm0 = (first_m + in_g % gsz) * BM2;
        cn0 = (in_g / gsz) * BN2;
        const bf16_t* Ab = csec ? p.A2 + (long)cb * p.sAb2 : p.A + (long)cb * p.sAb;
        const bf16_t* Wb = csec ? p.W2 : p.W;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = prow + i * 16;
            const int slot = (lane & 3) ^ ((r >> 2) & 3);     // logical 16-B slot stored at physical slot lane&3
            src[0][i] = (const char*)(Ab + (long)min(cm0 + r, Mc - 1) * p.lda + slot * 8);
            src[1][i] = (const char*)(Wb + (long)(cn0 + r) * p.ldw + slot * 8);
        }
    };
    auto stage1 = [&](int st, int ph, int i) {               // piece i of issue phase ph for stage st
        char* dstp = smem + (st & (NS2 - 1)) * STAGE2_BYTES + ph * OPER2_BYTES + (wave * 32 + i * 16) * 64;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[ph][i] + (long)st * (BK2 * 2)),
                                         (__attribute__((address_space(3))) void*)dstp, 16, 0, 0);
    };
    auto stage_all = [&](int st) {
        stage1(st, 0, 0); stage1(st, 0, 1); stage1(st, 1, 0); stage1(st, 1, 1);
    };

    // ---- fragment offsets (32x32x16 operands: lane -> row lane&31, 16-B slot ks*2 + (lane>>5)) ----
    // row = base(multiple of 32) + (lane&31): one swizzle term per lane; ks advances the slot by 2 (byte ^ 32)
    const int j31 = lane & 31, hi = lane >> 5;
    const int sw = (j31 >> 2) & 3;
    const int offA0 = (grp * 128 + j31) * 64 + ((hi ^ sw) << 4);
    const int offW0 = (wn * 64 + j31) * 64 + ((hi ^ sw) << 4);

    f32x16 acc[4][2];   // [32-row m block][32-col n block]
    bf16x8 fa[2][2], fw[2][2];   // A: 2 m-blocks x 2 k-steps of the current 64-row half; W: 2 n-blocks x 2 k-steps
    auto loadA = [&](const char* tA, int qm) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[mb][ks] = *(const bf16x8*)(tA + ((offA0 + (qm * 64 + mb * 32) * 64) ^ (ks * 32)));
    };
    auto loadW = [&](const char* tW) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fw[nb][ks] = *(const bf16x8*)(tW + ((offW0 + (nb * 32) * 64) ^ (ks * 32)));
    };
#define TG_BAR()                                         \
    do {                                                 \
        __builtin_amdgcn_sched_barrier(0);               \
        __builtin_amdgcn_s_barrier();                    \
        __builtin_amdgcn_sched_barrier(0);               \
    } while (0)
#define TG_MFMA(QM, KS, MB, NB) \
    acc[(QM) * 2 + (MB)][NB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[NB][KS], fa[MB][KS], acc[(QM) * 2 + (MB)][NB], 0, 0, 0)
// 8 MFMAs of one 64x64 half-tile step (4 independent accumulators, each used twice 4 issues apart); the wave's 2 LDS-DMA
// pieces of issue phase PH for stage st+3 are issued from INSIDE the MFMA stream (the wave idles ~24 of every 32 cycles there)
#define TG_COMPUTE(QM, PH, MORE)                                                                            \
    do {                                                                                                    \
        __builtin_amdgcn_s_setprio(1);                                                                      \
        TG_MFMA(QM, 0, 0, 0); TG_MFMA(QM, 0, 1, 0);                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if (MORE) stage1(st + 3, PH, 0);                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        TG_MFMA(QM, 0, 0, 1); TG_MFMA(QM, 0, 1, 1); TG_MFMA(QM, 1, 0, 0);                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if (MORE) stage1(st + 3, PH, 1);                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        TG_MFMA(QM, 1, 1, 0); TG_MFMA(QM, 1, 0, 1); TG_MFMA(QM, 1, 1, 1);                                   \
        __builtin_amdgcn_s_setprio(0);                                                                      \
    } while (0)
#define TG_LOAD_END(MORE)                                                              \
    do {                                                                               \
        if (MORE) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");          \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");               \
        TG_BAR();                                                                      \
    } while (0)

    const int nst = p.K / BK2;
    int tile = blockIdx.x;
    set_tile(tile);
#pragma unroll
    for (int st = 0; st < 3; ++st)
        if (st < nst) stage_all(st);

    for (; tile < nwg; tile += gridDim.x) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // stages 0..2 landed (and the previous epilogue retired)
        TG_BAR();
        if (grp == 1) TG_BAR();   // group 1 runs one barrier (= half a phase) behind group 0

        // steady part (st + 3 < nst: every `if (more)` is compile-time true -> a straight-line loop body without the eight
        // wave-uniform branches per stage) + the last three stages through the generic body
        auto kbody = [&](int st, auto steady_c) {
            constexpr bool STEADY = decltype(steady_c)::value;
            const char* tA = smem + (st & (NS2 - 1)) * STAGE2_BYTES;
            const char* tW = tA + OPER2_BYTES;
            const bool more = STEADY || st + 3 < nst;
            // phase 0: rows [0,64) of this wave's half x all 64 columns
            loadW(tW);
            loadA(tA, 0);
            TG_LOAD_END(more);
            TG_COMPUTE(0, 0, more);
            TG_BAR();
            // phase 1: rows [64,128)
            loadA(tA, 1);
            TG_LOAD_END(more);
            TG_COMPUTE(1, 1, more);
            TG_BAR();
        };
        {
            int st = 0;
            for (; st + 3 < nst; ++st) kbody(st, std::true_type{});
            for (; st < nst; ++st) kbody(st, std::false_type{});
        }
        if (grp == 0) TG_BAR();   // every wave is now past its last LDS read of this tile

        // ---- next tile's first three stages go in flight (ring slots 0..2) under the epilogue ----
        const int eb = cb, em0 = cm0, en0 = cn0, esec = csec;
        if (tile + (int)gridDim.x < nwg) {
            set_tile(tile + gridDim.x);
#pragma unroll
            for (int st = 0; st < 3; ++st)
                if (st < nst) stage_all(st);
        }

        // ---- epilogue through LDS (ring slot 3, 4 KiB per wave, 16-B slots XOR-swizzled by row&7): MFMA layout -> full rows ----
        // per 32x32 block a lane holds D[n = 8*(r>>2) + 4*hi + (r&3)][m = lane&31]
        char* stg = smem + 3 * STAGE2_BYTES + wave * 4096;
        bf16_t* Cb = esec ? p.C2 + (long)eb * p.sCb2 : p.C + (long)eb * p.sCb;
        const bf16_t* ebias = esec ? p.bias2 : p.bias;
        const int eM = esec ? p.M2 : p.M;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int nl = nt * 32 + r4 * 8 + hi * 4;
                    float v[4] = {acc[mt][nt][r4 * 4 + 0], acc[mt][nt][r4 * 4 + 1], acc[mt][nt][r4 * 4 + 2], acc[mt][nt][r4 * 4 + 3]};
                    if (ebias) {
                        const uint2 bb = *(const uint2*)(ebias + en0 + wn * 64 + nl);
                        v[0] += bf16lo_to_f32(bb.x); v[1] += bf16hi_to_f32(bb.x);
                        v[2] += bf16lo_to_f32(bb.y); v[3] += bf16hi_to_f32(bb.y);
                    }
                    if (EPI == TG_EPI_BIAS_GELU) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = gelu_tanh(round_bf16(v[i]));
                    } else if (EPI == TG_EPI_BIAS_SILU) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = silu(round_bf16(v[i]));
                    }
                    uint2 o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
                    *(uint2*)(stg + j31 * 128 + ((((nl >> 3) ^ (j31 & 7)) << 4) | ((nl & 4) << 1))) = o;
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + (lane >> 3), ch = lane & 7;
                const uint4 val = *(const uint4*)(stg + row * 128 + ((ch ^ (row & 7)) << 4));
                const int m = em0 + grp * 128 + mt * 32 + row;
                const int n = en0 + wn * 64 + ch * 8;
                if (m < eM) {
                    uint4 o = val;
                    if (EPI == TG_EPI_BIAS_GATE_RES) {   // y = residual + gate[group(m)] * bf16(linear)
                        const int g = p.g.tok_group[m];
                        const bf16_t* gate_row = (const bf16_t*)p.g.mod + (long)eb * p.g.mod_batch_stride + (long)p.g.row[g] * p.g.mod_ld + p.g.gate_col[g];
                        const uint4 gg = *(const uint4*)(gate_row + n);
                        const uint4 rr = *(const uint4*)(p.R + (long)eb * p.sRb + (long)m * p.ldr + n);
                        const uint32_t vu[4] = {val.x, val.y, val.z, val.w}, gu[4] = {gg.x, gg.y, gg.z, gg.w}, ru[4] = {rr.x, rr.y, rr.z, rr.w};
                        uint32_t ou[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            ou[i] = pack_bf16x2(bf16lo_to_f32(ru[i]) + bf16lo_to_f32(gu[i]) * bf16lo_to_f32(vu[i]),
                                                bf16hi_to_f32(ru[i]) + bf16hi_to_f32(gu[i]) * bf16hi_to_f32(vu[i]));
                        o = uint4{ou[0], ou[1], ou[2], ou[3]};
                    }
                    *(uint4*)(Cb + (long)m * p.ldc + n) = o;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // staging reads done before the next block overwrites it
            __builtin_amdgcn_wave_barrier();
        }
    }
#undef TG_BAR
#undef TG_COMPUTE
#undef TG_MFMA
#undef TG_LOAD_END
}

// ================================================================================================
// 256x256x64 tile, FOUR waves (one per SIMD), each 128x128 = 8x8 MFMA 16x16x32 with the 256 accumulators in AGPRs.
//
// Why 128x128 per wave: LDS traffic and issue slots.  In the 8-wave kernel above a wave owns 128x64 of the tile and reads 6 KiB of
// fragments per 8 MFMAs; here it is 8 KiB per 16 (the same trade the vendor's hand-written gfx950 kernel makes: MT256x256x64,
// 4 waves, 8x8 16x16x32 blocks per wave).  There is no second wave on the SIMD to hide anything, so the wave software-pipelines
// itself: every MFMA is followed by at most one LDS-DMA piece or one ds_read_b128.
// Why 16x16x32 and not 32x32x16: POWER.  These GEMMs run against the socket power cap (1300-1400 W), so the shader clock is what
// the kernel's energy per flop allows.  tools/ubench/mfma_power.hip (registers only, 1 wave per SIMD, 256 accumulators): the
// 32x32x16 loop sustains 1.79 PFLOP/s at 1.79 GHz, the 16x16x32 loop 2.05 PFLOP/s at 2.04 GHz under the same cap — half the
// accumulator-file traffic per flop.  (tools/clock_probe.py: the vendor's 16x16x32 kernel ran FF1 at 1.90 GHz, the 32x32x16
// version of this kernel at 1.72 GHz with a HIGHER per-clock MFMA utilisation.)
//
//   LDS: 2 stages x (A[256 rows][128 B] | W[256 rows][128 B]) = 128 KiB, + 4 x 4 KiB epilogue staging + bias / group tables.
//   16-B slot of a row XOR-swizzled by (row>>1)&7 (applied on the DMA source address): ds_read_b128 is served in four 16-lane groups
//   against a 256-B (two-row) bank line; a group of the 16x16x32 fragment read is 8 rows at k-chunk c and 8 other rows at chunk c^1,
//   which this swizzle spreads over all 16 slots.  DMA piece = 8 rows x 128 B: every global access of the fill is a full 128-byte
//   row segment; `buffer_load_dwordx4 ... lds` with a per-tile resource whose num_records ends at the last valid row, so M-edge rows
//   read zeros (no clamping arithmetic).
//   stage s (buffer b = s&1) = 128 MFMA slots, k32-steps 0,1 with fragment sets 0,1 (16 ds_read_b128 each):
//     slots 0..31   (every 2nd) reads of set 1 of THIS stage (buffer b); then lgkmcnt(0) + s_barrier: buffer b has no reader left
//     slots 36..126 (every 6th) the 16 DMA pieces of stage s+2 -> buffer b
//     slot 93       vmcnt(pieces issued so far) + s_barrier: stage s+1 landed everywhere; then (every 2nd slot) reads of set 0 of s+1
//   The stage stream runs across output tiles (persistent): the last two stages of a tile fetch the first two of the next.
// ================================================================================================
constexpr int BK3 = 64;
constexpr int OPER3_BYTES = 256 * BK3 * 2;       // 32 KiB per operand per stage
constexpr int STAGE3_BYTES = 2 * OPER3_BYTES;    // 64 KiB
constexpr int W4_STG_OFF = 2 * STAGE3_BYTES;     // epilogue staging behind the two stages
constexpr int W4_BIAS_OFF = W4_STG_OFF + 4 * 4096;   // 256 B of bias per wave
constexpr int W4_TOK_OFF = W4_BIAS_OFF + 4 * 256;    // gated-residual epilogue: group id of this wave's 128 rows, one dword each
constexpr int W4_GTAB_OFF = W4_TOK_OFF + 4 * 512;    // ... and the gate-row element offset of every group (16 dwords per wave)
constexpr int W4_LDS_BYTES = W4_GTAB_OFF + 4 * 64;

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm256w4_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = p.N / BN2;
    const int tiles_m1 = (p.M + BM2 - 1) / BM2, tiles_m2 = p.A2 ? (p.M2 + BM2 - 1) / BM2 : 0;
    const int nwg1 = tiles_m1 * tiles_n * p.batch;
    const int nwg = nwg1 + tiles_m2 * tiles_n * p.batch;
    const int nst = p.K / BK3;

    struct Coord { int sec, b, m0, n0; };
    auto coords = [&](int id) {
        Coord c;
        c.sec = id >= nwg1;
        const int tiles_m = c.sec ? tiles_m2 : tiles_m1;
        const int per_batch = tiles_m * tiles_n;
        int t = c.sec ? xcd_remap(id - nwg1, nwg - nwg1) : xcd_remap(id, nwg1);
        c.b = t / per_batch;
        t -= c.b * per_batch;
        const int per_group = p.group_m * tiles_n;
        const int gi = t / per_group;
        const int first_m = gi * p.group_m;
        const int gsz = min(tiles_m - first_m, p.group_m);
        const int in_g = t - gi * per_group;
        c.m0 = (first_m + in_g % gsz) * BM2;
        c.n0 = (in_g / gsz) * BN2;
        return c;
    };

    // ---- DMA side: a cursor (tile, k-stage) that runs two stages ahead of the MFMAs ----
    // LDS image: element (row, 16-B slot s) of an operand tile lives at byte row*128 + ((s ^ ((row>>1)&7)) << 4).  A DMA piece is 8 rows,
    // written lane-linear (1 KiB): the lane at physical row pr = lane>>3, physical slot lane&7 of piece i fetches source slot
    // (lane&7) ^ ((row>>1)&7) with row = wave*64 + i*8 + pr, i.e. (lane&7) ^ (4*(i&1) + (pr>>1)): one offset per piece parity.
    int voffA[2], voffW[2];
#pragma unroll
    for (int odd = 0; odd < 2; ++odd) {
        const int row = lane >> 3, dslot = (lane & 7) ^ (odd * 4 + (row >> 1));
        voffA[odd] = (int)(((long)(wave * 64 + row) * p.lda + dslot * 8) * 2);
        voffW[odd] = (int)(((long)(wave * 64 + row) * p.ldw + dslot * 8) * 2);
    }
    const int pieceA = (int)(p.lda * 16), pieceW = (int)(p.ldw * 16);      // 8 rows, bytes
    __amdgpu_buffer_rsrc_t rA, rW;
    Coord nc;                                      // coordinates of the tile the DMA cursor is in (= the next tile once it left this one)
    int dtile = blockIdx.x, dk = 0, dbuf = 0;
    auto set_dma_tile = [&](int id) {
        const Coord c = coords(id);
        nc = c;
        const int Mc = c.sec ? p.M2 : p.M;
        const bf16_t* Ab = (c.sec ? p.A2 + (long)c.b * p.sAb2 : p.A + (long)c.b * p.sAb) + (long)c.m0 * p.lda;
        const bf16_t* Wb = (c.sec ? p.W2 : p.W) + (long)c.n0 * p.ldw;
        rA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)(min(Mc - c.m0, BM2) * p.lda * 2), 0x00020000);
        rW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, (int)(BN2 * p.ldw * 2), 0x00020000);
    };
    auto dma_piece = [&](int q) {                  // q = 0..7: A pieces, 8..15: W pieces of the cursor's stage
        char* dst = smem + dbuf * STAGE3_BYTES + (q >> 3) * OPER3_BYTES + (wave * 64 + (q & 7) * 8) * 128;
        if (q < 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)dst, 16, voffA[q & 1] + (q & 7) * pieceA, dk * (BK3 * 2), 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void*)dst, 16, voffW[q & 1] + (q & 7) * pieceW, dk * (BK3 * 2), 0, 0);
    };
    auto dma_advance = [&]() { dbuf ^= 1; ++dk; };
    auto dma_next_tile = [&]() {                   // the cursor leaves the current tile (called once per tile, before its last two stages)
        dk = 0;
        dtile += gridDim.x;
        if (dtile < nwg) set_dma_tile(dtile);
    };

    // ---- fragment addresses (16x16x32 operands: lane -> row lane&15 of a 16-row block, logical 16-B slot ks*4 + (lane>>4)) ----
    const int l15 = lane & 15, ch = lane >> 4;
    const int swz = l15 * 128 + ((ch ^ ((l15 >> 1) & 7)) << 4);
    // rw: first MFMA operand (lane ends up holding 4 consecutive indices of it), ra: second operand (index lane&15).  Normally the first
    // operand is the W rows (-> 4 consecutive output columns per lane); for a V^T tile the two are exchanged (-> 4 consecutive tokens per
    // lane), which only changes WHERE the fragments are read from: the k loop and the accumulator indexing are the same code
    int ra[2], rw[2];                              // byte address of k-step ks; + block*2048 as the immediate
    int bufpar = 0;                                // 0 / STAGE3_BYTES: the stage buffer the fragment addresses point into
    auto set_frag_bases = [&](bool vtile) {
        const int aoff = wm * 128 * 128, woff = OPER3_BYTES + wn * 128 * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            ra[ks] = (((vtile ? woff : aoff) + swz) ^ (ks << 6)) ^ bufpar;
            rw[ks] = (((vtile ? aoff : woff) + swz) ^ (ks << 6)) ^ bufpar;
        }
    };
    bf16x8 fa[2][8], fw[2][8];                     // [k-step][16-row block]
    f32x4 acc[8][8];                               // [m block][n block]; lane holds D[n = 4*(lane>>4) + r][m = lane&15]

#define W4_SB() __builtin_amdgcn_sched_barrier(0)
#define W4_DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
    // fragment read r (0..15) of k-step KS: 0..7 -> W blocks, 8..15 -> A blocks
    auto frag_read = [&](auto ksc, auto rc) {
        constexpr int KS = decltype(ksc)::value, R = decltype(rc)::value;
        if constexpr (R < 8) {
            bf16x8& d = fw[KS][R];
            const int ad = rw[KS];
            W4_DSR(d, ad, R * 2048);
        } else {
            bf16x8& d = fa[KS][R - 8];
            const int ad = ra[KS];
            W4_DSR(d, ad, (R - 8) * 2048);
        }
    };
    // one stage = 128 MFMA slots (k-steps 0,1 x 64 blocks, 16 cycles each); after MFMA I the slot may carry one ds_read / one DMA
    // piece / a barrier:
    //   slots [0, 16*R1S)      every R1S-th: fragment read r of set 1 of THIS stage                        (buffer b)
    //   slot  B1               lgkmcnt(0); s_barrier -> buffer b has no reader left; fragment addresses flip to buffer b^1
    //   slots [D0, D0+16*DS)   every DS-th: DMA piece of stage s+2 -> buffer b
    //   slot  B2               vmcnt(pieces issued so far in this stage); s_barrier -> stage s+1 landed everywhere
    //   slots [R20, R20+16*R2S) every R2S-th: fragment read r of set 0 of stage s+1                        (buffer b^1)
#ifndef W4_R1S
#define W4_R1S 2
#endif
#ifndef W4_B1
#define W4_B1 35
#endif
#ifndef W4_D0
#define W4_D0 36
#endif
#ifndef W4_DS
#define W4_DS 6
#endif
#ifndef W4_B2
#define W4_B2 93
#endif
#ifndef W4_R20
#define W4_R20 94
#endif
#ifndef W4_R2S
#define W4_R2S 2
#endif
    static_assert(W4_B1 >= 16 * W4_R1S - W4_R1S && W4_B1 < 64 && W4_D0 > W4_B1 && W4_D0 + 15 * W4_DS < 128 && W4_R20 > W4_B2 &&
                  W4_R20 + 15 * W4_R2S < 128 && W4_B2 >= 64, "schedule");
    auto toggle = [&]() {
        bufpar ^= STAGE3_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { ra[ks] ^= STAGE3_BYTES; rw[ks] ^= STAGE3_BYTES; }
    };
    // STEADY: the DMA cursor stays inside the current tile and there is always something to fetch / read
    auto kstage = [&](auto steady_c, bool more, bool rd) {
        constexpr bool STEADY = decltype(steady_c)::value;
        const bool dma = STEADY || more, rdn = STEADY || rd;
        static_for<0, 128>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            constexpr int KS = I >> 6, NB = (I >> 3) & 7, MB = I & 7;
            acc[MB][NB] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[KS][NB], fa[KS][MB], acc[MB][NB], 0, 0, 0);
            W4_SB();
            if constexpr (I < 16 * W4_R1S && I % W4_R1S == 0)
                frag_read(std::integral_constant<int, 1>{}, std::integral_constant<int, I / W4_R1S>{});
            if constexpr (I == W4_B1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                W4_SB();
                toggle();
            }
            if constexpr (I >= W4_D0 && (I - W4_D0) % W4_DS == 0 && (I - W4_D0) / W4_DS < 16) {
                if (dma) dma_piece((I - W4_D0) / W4_DS);
            }
            if constexpr (I == W4_B2) {
                constexpr int ISSUED = (W4_B2 - W4_D0) / W4_DS + 1 > 16 ? 16 : (W4_B2 - W4_D0) / W4_DS + 1;
                if (dma) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ISSUED) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                W4_SB();
            }
            if constexpr (I >= W4_R20 && (I - W4_R20) % W4_R2S == 0 && (I - W4_R20) / W4_R2S < 16) {
                if (rdn) frag_read(std::integral_constant<int, 0>{}, std::integral_constant<int, (I - W4_R20) / W4_R2S>{});
            }
            W4_SB();
        });
        if (dma) dma_advance();
        W4_SB();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W4_SB();
    };

    if (EPI == TG_EPI_BIAS_GATE_RES) {             // per-wave LDS table: element offset of group g's gate row inside one batch item of `mod`
        int v = 0;
#pragma unroll
        for (int g = 0; g < TG_MAX_GROUPS; ++g) v = lane == g ? (int)(p.g.row[g] * p.g.mod_ld + p.g.gate_col[g]) : v;
        const int ga = W4_GTAB_OFF + wave * 64 + (lane & 15) * 4;
        if (lane < TG_MAX_GROUPS) asm volatile("ds_write_b32 %0, %1" ::"v"(ga), "v"(v));
    }
    int tile = blockIdx.x;
    set_dma_tile(tile);
    Coord cc = nc;
    // prologue: stages 0 and 1 in flight, fragment sets 0,1 of stage 0 in registers
    static_for<0, 16>([&](auto qc) { dma_piece(decltype(qc)::value); });
    dma_advance();
    static_for<0, 16>([&](auto qc) { dma_piece(decltype(qc)::value); });
    dma_advance();
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    W4_SB();

    for (; tile < nwg; tile += gridDim.x) {
        // fragment set 0 of this tile's stage 0 (its buffer landed: waited in the previous tile's last stage / the prologue).  Read
        // here and not under the previous epilogue: 64 live fragment registers there would be spilled, and a VMEM reload into a
        // fragment register makes the compiler guard the k loop's first ds_reads with vmcnt(1..3), i.e. wait for the DMA just issued.
        const bool vtile = p.Vt && cc.n0 >= p.vt_col0;
        set_frag_bases(vtile);
        static_for<0, 16>([&](auto rc) { frag_read(std::integral_constant<int, 0>{}, rc); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W4_SB();
        // the 128 bias values of this wave's columns go to LDS by one more DMA piece (256 B), retired by the counted waits of the first
        // stage; the epilogue then reads them with ds_read_b64.  (Global loads in the epilogue each cost a serialising vmcnt(0) that also
        // waits for the previous block's stores and the next tile's DMA; loads issued here into registers get the same vmcnt(0) at their
        // first use.)
        const bf16_t* ebias = cc.sec ? p.bias2 : p.bias;
        if (ebias)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ebias + cc.n0 + wn * 128 + lane * 2),
                                             (__attribute__((address_space(3))) void*)(smem + W4_BIAS_OFF + wave * 256), 4, 0, 0);
        if (EPI == TG_EPI_BIAS_GATE_RES) {           // group ids of the wave's 128 rows (bytes, zero-extended to one dword per lane by the DMA)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.g.tok_group + min(cc.m0 + wm * 128 + h * 64 + lane, p.M - 1)),
                                                 (__attribute__((address_space(3))) void*)(smem + W4_TOK_OFF + wave * 512 + h * 256), 1, 0, 0);
        }
        // (starting k-step 0 from a zero C operand instead of these 256 v_accvgpr_write was tried: the compiler then keeps the first
        // results in VGPRs and copies them to the AGPRs before k-step 1 — more moves, not fewer)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int jn = 0; jn < 8; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            int st = 0;
            for (; st + 2 < nst; ++st) kstage(std::true_type{}, true, true);
            const bool has_next = tile + (int)gridDim.x < nwg;
            dma_next_tile();
            kstage(std::false_type{}, has_next, true);
            kstage(std::false_type{}, has_next, false);
        }
        const Coord ec = cc;
        if (tile + (int)gridDim.x < nwg) cc = nc;

        // ---- epilogue through LDS (4 KiB per wave behind the stages, 16-B slots XOR-swizzled by row&7): MFMA layout -> full rows ----
        // per 16x16 block a lane holds D[n = 4*(lane>>4) + r][m = lane&15].  The staging accesses are asm so that the compiler's
        // LDS-DMA alias rule (vmcnt(0) before any LDS access while a DMA is in flight) does not serialise them behind the global stores
        bf16_t* Cb = ec.sec ? p.C2 + (long)ec.b * p.sCb2 : p.C + (long)ec.b * p.sCb;
        const int eM = ec.sec ? p.M2 : p.M;
        const int stg = W4_STG_OFF + wave * 4096;
        const int wbase = (stg + l15 * 128 + (ch & 1) * 8) | (((ch >> 1) ^ (l15 & 7)) << 4);
        const int rbase = stg + (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);
        // V^T tile (bias epilogue only): the operands were exchanged, so acc[i][j] is (W n-block i, activation m-block j) and a lane holds
        // D[m = 16 j + 4 (lane>>4) + r][n = 16 i + (lane&15)] — the same staging code with rows = n and columns = m; what differs is
        // tile-uniform: where the bias comes from, the zeroing of tokens >= M, and the base / stride / bounds of the store
        const bool evt = EPI == TG_EPI_BIAS && vtile;
        u32x2 bb[8];                                      // bias of this lane's 8 column quads (n = nb*16 + 4*(lane>>4) ..+3)
        float bv[8];                                      // V^T tile: bias of n = i*16 + (lane&15)
        if (ebias) {
            const int ba = W4_BIAS_OFF + wave * 256 + (evt ? l15 * 2 : ch * 8);
            if (evt) {
                uint32_t raw[8];
                static_for<0, 8>([&](auto qc) {
                    constexpr int Q = decltype(qc)::value;
                    uint32_t& d = raw[Q];
                    const int ad = ba;
                    asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"(Q * 32));
                });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                W4_SB();
#pragma unroll
                for (int q = 0; q < 8; ++q) { bv[q] = __uint_as_float(raw[q] << 16); bb[q] = u32x2{0u, 0u}; }
            } else {
                static_for<0, 8>([&](auto qc) {
                    constexpr int Q = decltype(qc)::value;
                    u32x2& d = bb[Q];
                    const int ad = ba;
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"(Q * 32));
                });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                W4_SB();
#pragma unroll
                for (int q = 0; q < 8; ++q) bv[q] = 0.f;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) { bb[q] = u32x2{0u, 0u}; bv[q] = 0.f; }
        }
        // store geometry: row = block row of the staging image (tokens, or V^T rows n - vt_col0), col = its 64-wide column block
        const long ostride = evt ? (ec.sec ? p.vt_ld2 : p.vt_ld) : p.ldc;
        bf16_t* obase = evt ? (ec.sec ? p.Vt2 : p.Vt) + (long)ec.b * (p.N - p.vt_col0) * ostride : Cb;
        const int orow0 = evt ? ec.n0 - p.vt_col0 + wn * 128 : ec.m0 + wm * 128;
        const int ocol0 = evt ? ec.m0 + wm * 128 : ec.n0 + wn * 128;
        const int orow_lim = evt ? 0x7fffffff : eM, ocol_lim = evt ? (int)ostride : 0x7fffffff;
        const int mzero = evt ? eM - (ec.m0 + wm * 128 + ch * 4) : 0x7fffffff;   // V^T: token 16 j + r of this lane is real iff < mzero
        // gated residual: y = residual + gate[group(m)] * bf16(linear).  Per lane the 16 read-back rows (mt, it) -> gate-row offsets via
        // the two LDS tables; the gate / residual chunks of block k+1 are requested before block k is converted and stored
        int goff[4][4];
        uint4 gq[2][4], rq[2][4];
        const bf16_t* gbase = (const bf16_t*)p.g.mod + (long)ec.b * p.g.mod_batch_stride + ec.n0 + wn * 128 + (lane & 7) * 8;
        const bf16_t* rbase_g = p.R + (long)ec.b * p.sRb + (long)(ec.m0 + wm * 128 + (lane >> 3)) * p.ldr + ec.n0 + wn * 128 + (lane & 7) * 8;
        const int rowclamp = eM - 1 - (ec.m0 + wm * 128 + (lane >> 3));      // rows past M re-read the last valid row (never stored)
        auto gate_issue = [&](auto blkc) {
            constexpr int BLK = decltype(blkc)::value, MT = BLK >> 1, NH = BLK & 1;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if (EPI == TG_EPI_BIAS_GATE_RES) gq[BLK & 1][it] = *(const uint4*)(gbase + goff[MT][it] + NH * 64);
                rq[BLK & 1][it] = *(const uint4*)(rbase_g + (long)min(MT * 32 + it * 8, rowclamp) * p.ldr + NH * 64);
            }
        };
        if (EPI == TG_EPI_BIAS_GATE_RES) {
            const int ta = W4_TOK_OFF + wave * 512 + (lane >> 3) * 4;
            int gid[4][4];
            static_for<0, 16>([&](auto qc) {
                constexpr int Q = decltype(qc)::value;
                int& d = gid[Q >> 2][Q & 3];
                const int ad = ta;
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"(((Q >> 2) * 32 + (Q & 3) * 8) * 4));
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            W4_SB();
            static_for<0, 16>([&](auto qc) {
                constexpr int Q = decltype(qc)::value;
                int& d = goff[Q >> 2][Q & 3];
                const int ad = W4_GTAB_OFF + wave * 64 + gid[Q >> 2][Q & 3] * 4;
                asm volatile("ds_read_b32 %0, %1" : "=v"(d) : "v"(ad));
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            W4_SB();
            gate_issue(std::integral_constant<int, 0>{});
        }
        if (EPI == TG_EPI_BIAS_MUL_GELU_GRAD) gate_issue(std::integral_constant<int, 0>{});      // (the R rows only: the pre-activation tile)
        bf16_t* const aux = EPI == TG_EPI_BIAS_KEEP_GELU ? const_cast<bf16_t*>(p.R) + (long)ec.b * p.sRb : nullptr;
        static_for<0, 8>([&](auto blkc) {
            constexpr int BLK = decltype(blkc)::value, mt = BLK >> 1, nh = BLK & 1;
            {
                if constexpr ((EPI == TG_EPI_BIAS_GATE_RES || EPI == TG_EPI_BIAS_MUL_GELU_GRAD) && BLK < 7) gate_issue(std::integral_constant<int, (BLK < 7 ? BLK + 1 : 7)>{});
                static_for<0, 8>([&](auto wc) {
                    constexpr int MB2 = decltype(wc)::value >> 2, NB4 = decltype(wc)::value & 3;
                    const f32x4 a = acc[mt * 2 + MB2][nh * 4 + NB4];
                    const u32x2 bq = bb[nh * 4 + NB4];
                    float v[4] = {a[0] + bf16lo_to_f32(bq.x), a[1] + bf16hi_to_f32(bq.x), a[2] + bf16lo_to_f32(bq.y), a[3] + bf16hi_to_f32(bq.y)};
                    if (EPI == TG_EPI_BIAS) {          // V^T tile: bb is zero, the bias is per row; tokens >= M become zeros
                        const float bs = bv[mt * 2 + MB2];
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = (nh * 4 + NB4) * 16 + i < mzero ? v[i] + bs : 0.f;
                    }
                    if (EPI == TG_EPI_BIAS_GELU) {
                        const f32x2v g0 = gelu_tanh2(f32x2v{round_bf16(v[0]), round_bf16(v[1])}), g1 = gelu_tanh2(f32x2v{round_bf16(v[2]), round_bf16(v[3])});
                        v[0] = g0.x; v[1] = g0.y; v[2] = g1.x; v[3] = g1.y;
                    } else if (EPI == TG_EPI_BIAS_SILU) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = silu(round_bf16(v[i]));
                    }
                    u32x2 o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
                    const int wa = wbase ^ (NB4 << 5);
                    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(wa), "v"(o), "n"(MB2 * 2048));
                });
                u32x4 val[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(val[it]) : "v"(rbase), "n"(it * 1024));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                W4_SB();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int m = orow0 + mt * 32 + it * 8 + (lane >> 3);          // row / column of the staging image in the output
                    const int n = ocol0 + nh * 64 + (lane & 7) * 8;
                    if (m < orow_lim && n < ocol_lim) {
                        uint4 o = uint4{val[it].x, val[it].y, val[it].z, val[it].w};
                        if (EPI == TG_EPI_BIAS_GATE_RES) {
                            const uint4 gg = gq[BLK & 1][it], rr = rq[BLK & 1][it];
                            const uint32_t vu[4] = {o.x, o.y, o.z, o.w}, gu[4] = {gg.x, gg.y, gg.z, gg.w}, ru[4] = {rr.x, rr.y, rr.z, rr.w};
                            uint32_t ou[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                ou[i] = pack_bf16x2(bf16lo_to_f32(ru[i]) + bf16lo_to_f32(gu[i]) * bf16lo_to_f32(vu[i]),
                                                    bf16hi_to_f32(ru[i]) + bf16hi_to_f32(gu[i]) * bf16hi_to_f32(vu[i]));
                            o = uint4{ou[0], ou[1], ou[2], ou[3]};
                        }
                        if (EPI == TG_EPI_BIAS_MUL_GELU_GRAD) {      // dx = bf16(dy W) * gelu'(pre-activation): tg_act mode 1 on the tile, no dhid round trip
                            const uint4 rr = rq[BLK & 1][it];
                            const uint32_t vu[4] = {o.x, o.y, o.z, o.w}, ru[4] = {rr.x, rr.y, rr.z, rr.w};
                            uint32_t ou[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                ou[i] = pack_bf16x2(gelu_tanh_bwd(bf16lo_to_f32(ru[i]), bf16lo_to_f32(vu[i])), gelu_tanh_bwd(bf16hi_to_f32(ru[i]), bf16hi_to_f32(vu[i])));
                            o = uint4{ou[0], ou[1], ou[2], ou[3]};
                        }
                        if (EPI == TG_EPI_BIAS_KEEP_GELU) {          // second output: gelu_tanh of the stored pre-activation (== TG_EPI_BIAS_GELU's values)
                            const uint32_t vu[4] = {o.x, o.y, o.z, o.w};
                            uint32_t gu[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const f32x2v g = gelu_tanh2(f32x2v{bf16lo_to_f32(vu[i]), bf16hi_to_f32(vu[i])});
                                gu[i] = pack_bf16x2(g.x, g.y);
                            }
                            *(uint4*)(aux + (long)m * p.ldr + n) = uint4{gu[0], gu[1], gu[2], gu[3]};
                        }
                        *(uint4*)(obase + (long)m * ostride + n) = o;
                    }
                }
            }
        });
    }
#undef W4_SB
#undef W4_DSR
}

template <int EPI>
int launch(GemmParams p, hipStream_t stream) {
    if (p.M >= 1024 && p.N % BN2 == 0) {   // large-M shapes: 256^2 ping-pong kernel
        const int tiles2 = (((p.M + BM2 - 1) / BM2) + (p.A2 ? (p.M2 + BM2 - 1) / BM2 : 0)) * (p.N / BN2) * p.batch;
        // tile order: groups of group_m m-tiles x all n-tiles, m fastest; the 32 tiles resident on one XCD then share group_m A panels and
        // 32/group_m W panels.  A (activations) is the big, XCD-private operand, W (weights) is shared by every XCD through the
        // Infinity Cache, so small groups win: measured sum over the four block GEMMs 7.61 (8) / 7.48 (4) / 7.53 (6) / 7.62 (2) ms,
        // and for K = 12288 (6.3 MB per A panel) a single m-tile per group is another 3 % faster (2.31 vs 2.34 vs 2.40 ms)
        p.group_m = p.K >= 8192 ? 1 : 4;
        const int n_cu = tg_device_cus();
        const dim3 grid2(tiles2 < n_cu ? tiles2 : n_cu);
        const bool w4 = tg_knob(TG_KNOB_GEMM_W4) != 0;     // 0 (cross-check tests): the 8-wave kernel for every shape
        if (w4 && p.K >= 4 * BK3 && p.lda < (1L << 21) && p.ldw < (1L << 21)) {   // 32-bit buffer offsets: 256 rows * ld * 2 B < 2^31
            TG_DYN_LDS(gemm256w4_kernel<EPI>, W4_LDS_BYTES);
            hipLaunchKernelGGL(gemm256w4_kernel<EPI>, grid2, dim3(256), W4_LDS_BYTES, stream, p);
            TG_LAUNCH_CHECK("tg_gemm_bf16(256w4)");
            return TG_OK;
        }
        TG_DYN_LDS(gemm256_kernel<EPI>, RING2_BYTES);
        hipLaunchKernelGGL(gemm256_kernel<EPI>, grid2, dim3(512), RING2_BYTES, stream, p);
        TG_LAUNCH_CHECK("tg_gemm_bf16(256)");
        return TG_OK;
    }
    const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN) * p.batch;
    TG_DYN_LDS(gemm_bf16_kernel<EPI>, 2 * STAGE_BYTES);
    hipLaunchKernelGGL(gemm_bf16_kernel<EPI>, dim3(tiles), dim3(256), 2 * STAGE_BYTES, stream, p);
    TG_LAUNCH_CHECK("tg_gemm_bf16");
    return TG_OK;
}

}  // namespace

extern "C" int tg_gemm_bf16(const void* A, long lda, long strideA, const void* W, long ldw, const void* bias,
                            void* C, long ldc, long strideC, int M, int N, int K, int batch, int epilogue,
                            const void* R, long ldr, long strideR, const tg_group_table* gate, hipStream_t stream) {
    TG_REQUIRE(A && W && C, TG_ERR_ARG, "tg_gemm_bf16: null pointer");
    TG_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, TG_ERR_SHAPE, "tg_gemm_bf16: bad dims M=%d N=%d K=%d batch=%d", M, N, K, batch);
    TG_REQUIRE(N % BN == 0 && K % BK == 0, TG_ERR_SHAPE, "tg_gemm_bf16: need N%%128==0 and K%%64==0 (N=%d K=%d)", N, K);
    TG_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && strideA % 8 == 0 && strideC % 8 == 0, TG_ERR_ALIGN,
               "tg_gemm_bf16: leading dimensions must keep 16-byte (A, W) / 8-byte (C) alignment");
    TG_REQUIRE(tg_aligned16(A) && tg_aligned16(W) && tg_aligned16(C), TG_ERR_ALIGN, "tg_gemm_bf16: unaligned base pointer");
    GemmParams p{};
    p.A = (const bf16_t*)A; p.lda = lda; p.sAb = strideA;
    p.W = (const bf16_t*)W; p.ldw = ldw;
    p.bias = (const bf16_t*)bias;
    p.C = (bf16_t*)C; p.ldc = ldc; p.sCb = strideC;
    p.R = (const bf16_t*)R; p.ldr = ldr; p.sRb = strideR;
    p.M = M; p.N = N; p.K = K; p.batch = batch;
    switch (epilogue) {
        case TG_EPI_BIAS: return launch<TG_EPI_BIAS>(p, stream);
        case TG_EPI_BIAS_GELU: return launch<TG_EPI_BIAS_GELU>(p, stream);
        case TG_EPI_BIAS_SILU: return launch<TG_EPI_BIAS_SILU>(p, stream);
        case TG_EPI_BIAS_GATE_RES:
            TG_REQUIRE(R && gate && gate->mod && gate->tok_group, TG_ERR_ARG, "tg_gemm_bf16: gate/residual epilogue needs R and a group table");
            TG_REQUIRE(ldr % 8 == 0 && strideR % 8 == 0 && tg_aligned16(R) && gate->mod_ld % 8 == 0 && gate->mod_batch_stride % 8 == 0 &&
                       tg_aligned16(gate->mod), TG_ERR_ALIGN, "tg_gemm_bf16: residual / gate table must be 16-byte aligned");
            p.g = *gate;
            return launch<TG_EPI_BIAS_GATE_RES>(p, stream);
        case TG_EPI_BIAS_KEEP_GELU:
        case TG_EPI_BIAS_MUL_GELU_GRAD: {
            TG_REQUIRE(R && ldr % 8 == 0 && strideR % 8 == 0 && tg_aligned16(R), TG_ERR_ARG, "tg_gemm_bf16: this epilogue needs R (16-byte aligned rows)");
            TG_REQUIRE(M >= 1024 && N % BN2 == 0 && K >= 4 * BK3 && lda < (1L << 21) && ldw < (1L << 21) && tg_knob(TG_KNOB_GEMM_W4) != 0, TG_ERR_SHAPE,
                       "tg_gemm_bf16: the keep-GELU / GELU-grad epilogues exist in the 4-wave kernel only (M >= 1024, N%%256 == 0, K >= 256)");
            p.group_m = p.K >= 8192 ? 1 : 4;
            const int tiles2 = ((p.M + BM2 - 1) / BM2) * (p.N / BN2) * p.batch;
            const int n_cu = tg_device_cus();
            if (epilogue == TG_EPI_BIAS_KEEP_GELU) {
                TG_DYN_LDS(gemm256w4_kernel<TG_EPI_BIAS_KEEP_GELU>, W4_LDS_BYTES);
                hipLaunchKernelGGL(gemm256w4_kernel<TG_EPI_BIAS_KEEP_GELU>, dim3(tiles2 < n_cu ? tiles2 : n_cu), dim3(256), W4_LDS_BYTES, stream, p);
            } else {
                TG_DYN_LDS(gemm256w4_kernel<TG_EPI_BIAS_MUL_GELU_GRAD>, W4_LDS_BYTES);
                hipLaunchKernelGGL(gemm256w4_kernel<TG_EPI_BIAS_MUL_GELU_GRAD>, dim3(tiles2 < n_cu ? tiles2 : n_cu), dim3(256), W4_LDS_BYTES, stream, p);
            }
            TG_LAUNCH_CHECK("tg_gemm_bf16(256w4, activation epilogue)");
            return TG_OK;
        }
        default: return tg_set_error(TG_ERR_ARG, "tg_gemm_bf16: unknown epilogue %d", epilogue);
    }
}

extern "C" int tg_gemm_bf16_pair(const void* A1, long strideA1, const void* W1, const void* bias1, void* C1, long strideC1, int M1,
                                 const void* A2, long strideA2, const void* W2, const void* bias2, void* C2, long strideC2, int M2,
                                 long lda, long ldw, long ldc, int N, int K, int batch, int epilogue, hipStream_t stream) {
    TG_REQUIRE(A1 && W1 && C1 && A2 && W2 && C2, TG_ERR_ARG, "tg_gemm_bf16_pair: null pointer");
    TG_REQUIRE(M1 >= 1024 && M2 >= 1024 && N > 0 && K > 0 && batch > 0 && N % BN2 == 0 && K % BK == 0, TG_ERR_SHAPE,
               "tg_gemm_bf16_pair: both problems must be 256^2-kernel shapes (M >= 1024, N%%256 == 0, K%%64 == 0)");
    TG_REQUIRE(epilogue == TG_EPI_BIAS || epilogue == TG_EPI_BIAS_GELU || epilogue == TG_EPI_BIAS_SILU, TG_ERR_ARG,
               "tg_gemm_bf16_pair: bias / GELU / SiLU epilogues only");
    TG_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && strideA1 % 8 == 0 && strideC1 % 8 == 0 && strideA2 % 8 == 0 && strideC2 % 8 == 0 &&
               tg_aligned16(A1) && tg_aligned16(W1) && tg_aligned16(C1) && tg_aligned16(A2) && tg_aligned16(W2) && tg_aligned16(C2), TG_ERR_ALIGN,
               "tg_gemm_bf16_pair: alignment");
    GemmParams p{};
    p.A = (const bf16_t*)A1; p.lda = lda; p.sAb = strideA1;
    p.W = (const bf16_t*)W1; p.ldw = ldw;
    p.bias = (const bf16_t*)bias1;
    p.C = (bf16_t*)C1; p.ldc = ldc; p.sCb = strideC1;
    p.M = M1; p.N = N; p.K = K; p.batch = batch;
    p.A2 = (const bf16_t*)A2; p.W2 = (const bf16_t*)W2; p.bias2 = (const bf16_t*)bias2; p.C2 = (bf16_t*)C2; p.sAb2 = strideA2; p.sCb2 = strideC2;
    p.M2 = M2;
    switch (epilogue) {
        case TG_EPI_BIAS: return launch<TG_EPI_BIAS>(p, stream);
        case TG_EPI_BIAS_GELU: return launch<TG_EPI_BIAS_GELU>(p, stream);
        default: return launch<TG_EPI_BIAS_SILU>(p, stream);
    }
}

extern "C" int tg_gemm_bf16_qkv(const void* A1, long strideA1, const void* W1, const void* bias1, void* C1, long strideC1, int M1, void* Vt1, long vt_ld1,
                                const void* A2, long strideA2, const void* W2, const void* bias2, void* C2, long strideC2, int M2, void* Vt2, long vt_ld2,
                                long lda, long ldw, long ldc, int N, int K, int batch, int v_col0, hipStream_t stream) {
    TG_REQUIRE(A1 && W1 && C1 && Vt1, TG_ERR_ARG, "tg_gemm_bf16_qkv: null pointer");
    TG_REQUIRE(!A2 || (W2 && C2 && Vt2), TG_ERR_ARG, "tg_gemm_bf16_qkv: second problem needs W2, C2 and Vt2");
    TG_REQUIRE(M1 >= 1024 && (!A2 || M2 >= 1024) && N > 0 && batch > 0 && N % BN2 == 0 && K % BK3 == 0 && K >= 4 * BK3, TG_ERR_SHAPE,
               "tg_gemm_bf16_qkv: needs the 4-wave kernel's shapes (M >= 1024, N%%256 == 0, K%%64 == 0, K >= 256)");
    TG_REQUIRE(v_col0 > 0 && v_col0 < N && v_col0 % BN2 == 0, TG_ERR_SHAPE, "tg_gemm_bf16_qkv: v_col0 must be a multiple of 256 inside (0, N)");
    TG_REQUIRE(vt_ld1 % 64 == 0 && vt_ld1 >= M1 && (!A2 || (vt_ld2 % 64 == 0 && vt_ld2 >= M2)), TG_ERR_SHAPE,
               "tg_gemm_bf16_qkv: vt_ld must be a multiple of 64 and >= M");
    TG_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && strideA1 % 8 == 0 && strideC1 % 8 == 0 && strideA2 % 8 == 0 && strideC2 % 8 == 0 &&
               tg_aligned16(A1) && tg_aligned16(W1) && tg_aligned16(C1) && tg_aligned16(Vt1) && tg_aligned16(A2) && tg_aligned16(W2) &&
               tg_aligned16(C2) && tg_aligned16(Vt2), TG_ERR_ALIGN, "tg_gemm_bf16_qkv: alignment");
    const bool w4_off = tg_knob(TG_KNOB_GEMM_W4) == 0;
    TG_REQUIRE(lda < (1L << 21) && ldw < (1L << 21), TG_ERR_SHAPE, "tg_gemm_bf16_qkv: leading dimensions must be < 2^21 elements");
    TG_REQUIRE(!w4_off, TG_ERR_ARG, "tg_gemm_bf16_qkv: only the 4-wave GEMM kernel has the V^T epilogue (TG_GEMM_W4=0 is set)");
    GemmParams p{};
    p.A = (const bf16_t*)A1; p.lda = lda; p.sAb = strideA1;
    p.W = (const bf16_t*)W1; p.ldw = ldw;
    p.bias = (const bf16_t*)bias1;
    p.C = (bf16_t*)C1; p.ldc = ldc; p.sCb = strideC1;
    p.M = M1; p.N = N; p.K = K; p.batch = batch;
    p.Vt = (bf16_t*)Vt1; p.vt_ld = vt_ld1; p.vt_col0 = v_col0;
    if (A2) {
        p.A2 = (const bf16_t*)A2; p.W2 = (const bf16_t*)W2; p.bias2 = (const bf16_t*)bias2; p.C2 = (bf16_t*)C2; p.sAb2 = strideA2; p.sCb2 = strideC2;
        p.M2 = M2; p.Vt2 = (bf16_t*)Vt2; p.vt_ld2 = vt_ld2;
    }
    return launch<TG_EPI_BIAS>(p, stream);
}
