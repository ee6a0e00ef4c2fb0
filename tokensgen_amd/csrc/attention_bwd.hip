// Flash-attention BACKWARD for head_dim 64 on gfx950 MFMA (the training step's dominant operator; the forward is
// attention.hip).  Replaces what autograd runs for F.scaled_dot_product_attention in the reference's training step
// (attention_processor.py:2066-2125 under train_cogvideo_to2v.py:1995-2010): given Q, K, V, O, dO it returns dQ, dK, dV.
//
//   P = softmax(scale * Q K^T)           recomputed tile by tile from the row log-sum-exp (never materialised)
//   dV = P^T dO        dP = dO V^T        dS = P o (dP - D),  D_i = sum_d dO_id O_id        dQ = scale dS K       dK = scale dS^T Q
//
// Three launches, no atomics (deterministic): (1) statistics — D, and the row log-sum-exp in the log2 domain unless the forward kept it
// (tg_attention_fwd_lse); (2) dK, dV: one workgroup per 256 keys walks the query tiles; (3) dQ: one workgroup per 256 queries walks the key
// tiles.  All products run on v_mfma_f32_32x32x16_bf16; gradients are fp32 (accumulate flag: the To2V processor's three attention calls share
// K / V tensors, so their gradients add up).
#include <type_traits>

#include "attention_bwd.h"
#include "tokensgen_hip.h"

namespace {
// =================================================================================================================================
// The kernels the entry point launches (the correct-first ones live in the TEST-ONLY library, tests/csrc/attention_bwd_crosscheck.hip).
//
// What changed against the correct-first version, and why:
//  * no LDS round trip for P / dS and no in-kernel transposes.  The 32x32x16 MFMA sums over 16 k values, 8 per lane half; WHICH 16 values a
//    k-step covers is free as long as both operands agree.  A lane of an S block (rows = queries, column = key j) holds queries
//    8g + 4hi + e (g, e = 0..3): accumulator elements 8t .. 8t+7 are the queries {16t + 4hi + e} u {16t + 8 + 4hi + e} — used AS the A operand
//    (row = key j) of k-step t of dV += P^T dO and dK += dS^T Q, straight from registers.  The B operand then needs dO / Q values of those
//    same queries for one head column: two 8-byte reads from a [d][q] tile.  Q^T, dO^T (and K^T for the dQ kernel) are produced once per
//    call by tg_transpose_v into the workspace, so the [d][q] tiles are plain 16-byte loads.
//  * one wave owns 64 keys (dK/dV kernel) or 64 queries (dQ, statistics): its K, V (resp. Q, dO) fragments and the 128 (resp. 64) output
//    accumulators live in registers for the whole kernel; the 32-row tiles of the other side stream through a double-buffered LDS stage that
//    all four waves share (one barrier per tile; global loads of tile i+1 are in flight while tile i is consumed).
//  * per (32 x 32) block: 16 MFMAs (dK/dV) / 12 (dQ) against 8 ds_read_b128 + 16 / 8 ds_read_b64 shared by the wave's two blocks.
// Still three launches and no atomics: gradients are bitwise reproducible.
// =================================================================================================================================
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // raw v_exp_f32: arguments are <= 0 or masked to -1e30
// Accumulate-chain MFMA with the accumulator PINNED to AGPRs ("+a"): the dK / dV / dQ blocks are touched by nothing but these MFMAs until the
// epilogue, and left to the register allocator they were shuffled between the two register halves every iteration (192 v_accvgpr moves per
// tile).  hipcc pads nothing inside an asm string: the A operand was just written by v_cvt_pk (VALU -> MFMA operand: 2 wait states, s_nop 1);
// the chain on C needs none.
__device__ __forceinline__ void mfma_acc(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// Resident fragments in the ACCUMULATOR half of the register file: an MFMA may take its B operand from AGPRs, VALU code cannot touch them, so
// the 64 registers of a wave's K | V (or Q | dO) fragments stop competing with the softmax arithmetic for the 256 architected VGPRs (left to
// the compiler they were parked in AGPRs anyway and copied back with ~130 v_accvgpr_read per tile).  Loaded straight from global memory.
__device__ __forceinline__ void load_frag_agpr(bf16x8& dst, const bf16_t* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(dst) : "v"(ptr) : "memory");
}
#define TG_WAIT_FRAGS(f) asm volatile("s_waitcnt vmcnt(0)" : "+a"(f[0][0]), "+a"(f[0][1]), "+a"(f[0][2]), "+a"(f[0][3]), "+a"(f[1][0]), "+a"(f[1][1]), "+a"(f[1][2]), "+a"(f[1][3]))
// X = A0 . B0^T and Y = A1 . B1^T over the 64 head dims (two independent 4-step chains, interleaved), A from VGPRs (LDS tile rows), B resident in
// AGPRs.  Ends with the 12 wait states an 8-pass MFMA result needs before anything but an accumulate chain reads it.
__device__ __forceinline__ void mfma_pair(f32x16& x, f32x16& y, const bf16x8 (&a0)[4], const bf16x8 (&a1)[4], const bf16x8 (&b0)[4], const bf16x8 (&b1)[4]) {
    asm("s_nop 1\n\t"
        "v_mfma_f32_32x32x16_bf16 %0, %2, %10, 0\n\t"
        "v_mfma_f32_32x32x16_bf16 %1, %6, %14, 0\n\t"
        "v_mfma_f32_32x32x16_bf16 %0, %3, %11, %0\n\t"
        "v_mfma_f32_32x32x16_bf16 %1, %7, %15, %1\n\t"
        "v_mfma_f32_32x32x16_bf16 %0, %4, %12, %0\n\t"
        "v_mfma_f32_32x32x16_bf16 %1, %8, %16, %1\n\t"
        "v_mfma_f32_32x32x16_bf16 %0, %5, %13, %0\n\t"
        "v_mfma_f32_32x32x16_bf16 %1, %9, %17, %1\n\t"
        "s_nop 11"
        : "=&v"(x), "=&v"(y)
        : "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]),
          "a"(b0[0]), "a"(b0[1]), "a"(b0[2]), "a"(b0[3]), "a"(b1[0]), "a"(b1[1]), "a"(b1[2]), "a"(b1[3]));
}
// one k-step of both chains of mfma_pair (FIRST: start from zero; LAST: append the 12 wait states): lets VALU work be placed between the steps
template <bool FIRST, bool LAST>
__device__ __forceinline__ void mfma_pair_step(f32x16& x, f32x16& y, const bf16x8& a0, const bf16x8& a1, const bf16x8& b0, const bf16x8& b1) {
    if (FIRST)
        asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %4, 0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %5, 0" : "=&v"(x), "=&v"(y) : "v"(a0), "v"(a1), "a"(b0), "a"(b1));
    else if (LAST)
        asm("v_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %5, %1\n\ts_nop 11" : "+v"(x), "+v"(y) : "v"(a0), "v"(a1), "a"(b0), "a"(b1));
    else
        asm("v_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %5, %1" : "+v"(x), "+v"(y) : "v"(a0), "v"(a1), "a"(b0), "a"(b1));
}
#define TG_SB() __builtin_amdgcn_sched_barrier(0)
// v_cvt_pk of two v_exp_f32 results: the transcendental unit's result needs a wait state before a plain VALU op reads it; hipcc pads that for
// its own instructions but not in front of an asm statement (measured: dV wrong by orders of magnitude without it once the packed-math form put
// the conversion right behind the second v_exp)
__device__ __forceinline__ uint32_t pack_bf16x2_trans(float lo, float hi) {
    uint32_t r;
    asm("s_nop 1\n\tv_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// One (batch, head) per XCD at a time, as in the forward: a 1-D grid of nx * heads * batch workgroups, workgroup w runs on XCD w % 8 and takes
// (head-batch index xcd + 8 * (slot / nx), block slot % nx) — the ~32 co-resident workgroups of an XCD then walk the SAME streamed tensor
// (Q | dO for dK/dV, K | V for dQ) nearly in step and share it in that XCD's L2.  With the plain (block, head-batch) grid the 70 key blocks of a
// head were dealt round-robin to all 8 XCDs and each XCD streamed the head's Q / dO on its own: 14.7 GB of L2 misses per dK/dV launch against
// ~1.5 GB algorithmic (profiles/r2_attention_bwd_pmc.json).
__device__ __forceinline__ void xcd_block(int w, int nx, int nhb, int& blk, int& hb) {
    if ((nhb & 7) == 0) {
        const int xcd = w & 7, slot = w >> 3;
        hb = xcd + 8 * (slot / nx);
        blk = slot % nx;
    } else {
        hb = w / nx;
        blk = w % nx;
    }
}

// B operand of dV += P^T dO / dK += dS^T Q / dQ += dS K straight from the ROW-major [row][d] tile (gfx950 LDS transpose read): a 16-lane group hands
// the hardware the sixteen 8-byte chunks of a [4 rows][16 columns] block and every lane receives ITS column's four row values — lane (j = lane & 31,
// hi) ends up with rows r0 + 4 hi + {0..3} of head column j, exactly the half k-step the P / dS accumulator fragments pair with (measured with
// tools/ubench/tr_probe.hip).  No [d][row] copies of Q / dO / K any more: half the stage writes and global fetches of the streamed side, and the three
// tg_transpose_v passes per call are gone.  The asm result is consumed after an explicit s_waitcnt (the compiler cannot see the pending LDS read).
__device__ __forceinline__ uint2 lds_tr_b64(const bf16_t* p) {
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"((uint32_t)(uintptr_t)p) : "memory");
    return r;
}

constexpr int BT = 32;                     // rows of the streamed tile
constexpr int LQ2 = 72;                    // [row][d] tile: row stride in elements (144 B)
constexpr int ROWT_EL = BT * LQ2;

struct Bwd2Params {
    BwdParams p;
    // dQ launch only: key-axis split.  With few queries (the vip-query call: 480 rows = 2 query blocks per head, 192 workgroups for 512 resident slots, each walking 571 key
    // tiles alone) the key tiles are cut into kparts ranges, one workgroup each; the partial dQ blocks go to dq_part [kparts][batch][nq][heads * 64] (scaled, fp32) and
    // attn_bwd_dq_join_kernel adds them in range order — deterministic like everything else here.
    int kparts;
    float* dq_part;
};

union Frag { bf16x8 v; uint2 u[2]; uint32_t w[4]; };

__device__ __forceinline__ uint4 ld_row16(const bf16_t* base, long ld, int row, int n, int col) {
    return row < n ? *(const uint4*)(base + (long)row * ld + col) : uint4{0, 0, 0, 0};
}
// The streamed tiles are fetched UNCONDITIONALLY from a clamped row and masked when they are written to LDS: a load under `if (row < n)` with a
// zero in the else branch makes hipcc wait for the load right there (the v_mov of the zero may not overtake it), which serialised every tile's
// global latency in front of its compute.
__device__ __forceinline__ uint4 ld_row16_clamped(const bf16_t* base, long ld, int row, int n, int col) {
    return *(const uint4*)(base + (long)min(row, n - 1) * ld + col);
}
__device__ __forceinline__ uint4 mask16(uint4 v, bool ok) { return ok ? v : uint4{0, 0, 0, 0}; }
// component-wise: `c ? a : b` on two uint4 lvalues becomes a pointer select and sends both to scratch memory
__device__ __forceinline__ uint4 sel16(bool c, uint4 a, uint4 b) { return uint4{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w}; }

// x = a + b + c to 2^-24 of x, each term the upper half of an fp32 word (= a bf16 value): the pieces of a statistic that travels through the matrix pipe
__device__ __forceinline__ void split_bf16x3(float x, uint32_t& a, uint32_t& b, uint32_t& c) {
    a = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(a);
    b = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(b);
    c = __float_as_uint(r2) & 0xffff0000u;
}
// the dK/dV kernel's per-query seed row: bf16 k-slots [l1 l2 l3 d1 d2 d3 0 0] with l = -lse / scale_log2 and d = -D in three pieces each
__device__ __forceinline__ uint4 seed_row(float lse, float dsum, float scale_log2) {
    uint32_t l1, l2, l3, d1, d2, d3;
    split_bf16x3(-lse / scale_log2, l1, l2, l3);
    split_bf16x3(-dsum, d1, d2, d3);
    return uint4{(l1 >> 16) | l2, (l3 >> 16) | d1, (d2 >> 16) | d3, 0u};
}

// ---- (1') statistics: one wave per 64 queries, keys streamed in tiles of 32; also packs the dK/dV kernel's seed rows ----
__global__ __launch_bounds__(256) void attn_bwd_stats2_kernel(BwdParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t sK[2][ROWT_EL];
    __shared__ float sDs[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hi = lane >> 5;
    const int h = blockIdx.y % p.heads, b = blockIdx.y / p.heads;
    const int q0 = blockIdx.x * 256 + wave * 64;
    const bf16_t* Q = p.q + (long)b * p.q_sb + h * HD;
    const bf16_t* Kp = p.k + (long)b * p.k_sb + h * HD;
    const long stat0 = ((long)b * p.heads + h) * p.nq;
    {   // D_i = sum_d dO_id O_id: one thread per query row
        const int r = blockIdx.x * 256 + tid;
        if (r < p.nq) {
            const uint4* o = (const uint4*)(p.o + (long)b * p.o_sb + (long)r * p.o_ld + h * HD);
            const uint4* g = (const uint4*)(p.dout + (long)b * p.do_sb + (long)r * p.do_ld + h * HD);
            float d = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint4 a = o[i], c = g[i];
                const uint32_t au[4] = {a.x, a.y, a.z, a.w}, cu[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) d += bf16lo_to_f32(au[e]) * bf16lo_to_f32(cu[e]) + bf16hi_to_f32(au[e]) * bf16hi_to_f32(cu[e]);
            }
            p.dsum[stat0 + r] = d;
            sDs[tid] = d;
            if (p.have_lse) p.seed[stat0 + r] = seed_row(p.lse[stat0 + r], d, p.scale_log2);
        }
    }
    if (p.have_lse) return;
    Frag qf[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint4 t = ld_row16(Q, p.q_ld, q0 + qb * 32 + j, p.nq, ks * 16 + hi * 8);
            qf[qb][ks].w[0] = t.x; qf[qb][ks].w[1] = t.y; qf[qb][ks].w[2] = t.z; qf[qb][ks].w[3] = t.w;
        }
    float m[2] = {-1e30f, -1e30f}, l[2] = {0.f, 0.f};
    const int row = tid >> 3, chunk = (tid & 7) * 8;
    const int ntile = (p.nk + BT - 1) / BT;
    uint4 pre = ld_row16(Kp, p.k_ld, row, p.nk, chunk);
    *(uint4*)(sK[0] + row * LQ2 + chunk) = pre;
    __syncthreads();
    for (int it = 0; it < ntile; ++it) {
        const int k0 = it * BT;
        if (it + 1 < ntile) pre = ld_row16_clamped(Kp, p.k_ld, k0 + BT + row, p.nk, chunk);        // rows past the end are masked to -1e30 below
        const bf16_t* cK = sK[it & 1];
        bf16x8 aK[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) aK[ks] = *(const bf16x8*)(cK + j * LQ2 + ks * 16 + hi * 8);
        const bool ragged = k0 + BT > p.nk;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 st = zero16();                          // rows = keys, column = query j
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aK[ks], qf[qb][ks].v, st, 0, 0, 0);
            float sv[16], mx = -1e30f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sv[r] = st[r] * p.scale_log2;
                if (ragged && k0 + acc_row(r, hi) >= p.nk) sv[r] = -1e30f;
                mx = fmaxf(mx, sv[r]);
            }
            const float mn = fmaxf(m[qb], mx);
            float add = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) add += fast_exp2(sv[r] - mn);
            l[qb] = l[qb] * fast_exp2(m[qb] - mn) + add;
            m[qb] = mn;
        }
        if (it + 1 < ntile) *(uint4*)(sK[(it + 1) & 1] + row * LQ2 + chunk) = pre;
        __syncthreads();
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const float mo = __shfl_xor(m[qb], 32, 64), lo = __shfl_xor(l[qb], 32, 64);
        const float M = fmaxf(m[qb], mo);
        const float Lsum = l[qb] * fast_exp2(m[qb] - M) + lo * fast_exp2(mo - M);
        const int q = q0 + qb * 32 + j;
        if (hi == 0 && q < p.nq) {
            const float lse = M + log2f(Lsum);
            p.lse[stat0 + q] = lse;
            p.seed[stat0 + q] = seed_row(lse, sDs[wave * 64 + qb * 32 + j], p.scale_log2);      // sDs: written before the tile loop's barriers
        }
    }
}

#define TG_WAIT_FRAGS1(f) asm volatile("s_waitcnt vmcnt(0)" : "+a"(f[0]), "+a"(f[1]), "+a"(f[2]), "+a"(f[3]))

#define BWD_BAR() do { TG_SB(); __builtin_amdgcn_s_barrier(); TG_SB(); } while (0)
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)p; }      // the low 32 bits of a flat LDS address are the LDS offset

// ---- (2') dK, dV as a two-group ping-pong (the forward's structure): 8 waves x 32 keys = 256 keys per workgroup, queries in tiles of 32 through a four-deep
// LDS ring.  The in-phase kernel this replaces (one 512-thread workgroup per CU, every wave in the same phase; round 3's first version) measured per tile and wave: S/dP MFMAs 560-860 cycles, softmax 820-1030, transposed reads 370, dV/dK
// MFMAs 250, stage 200, barrier 130-600 — the two waves of a SIMD run the same phase at the same time, so their MFMA blocks collide and their VALU
// blocks collide, and every LDS read is waited for where it is used.  Here every wave alternates
//     X(t) = { dV += P(t-1)^T dO(t-1), dK += dS(t-1)^T Q(t-1) ; S(t) = Q(t) K^T, dP(t) = dO(t) V^T }    18 MFMAs, ALL LDS reads (16 transposed b64 + 9 b128),
//                                                                                                     each read issued two MFMA pairs ahead of its use
//     Y(t) = { P = exp2(c S), dS = P o dP -> bf16 A operands ; stage write of a later tile ; next global fetch }      VALU only
// with one s_barrier after each segment and group 1 (waves 4-7) one segment behind group 0.
// The row statistics ride on the matrix pipe: S must become scale_log2 * (q.k) - lse and dP must become dP - D, both per QUERY ROW, which in this
// accumulator layout (rows = queries) costs every lane 8 ds_read_b128 and 32 VALU operations per tile.  Instead the stager splits -lse / scale_log2 and -D
// into three bf16 pieces each (exact to 2^-24) and leaves them as one 16-byte A-operand row per query: [l1 l2 l3 d1 d2 d3 0 0]; one extra k-step against the
// constant B operands [1 1 1 0 0 0 0 0] / [0 0 0 1 1 1 0 0] seeds both accumulators (2 MFMAs, 1 ds_read_b128 per wave and tile).
// Ring: tile u lives in buffer u % 4, is read in X(u) (A operands) and X(u+1) (transposed operands) of both groups = intervals 2u .. 2u+3, and is
// written by group 0 (Q + statistics) in its Y(u-2) = interval 2u-3 and by group 1 (dO) in its Y(u-3) = interval 2u-4, complete (the writer's next X ends
// with lgkmcnt(0)) one barrier before interval 2u; the tile it replaces (u-4) was last read in interval 2u-5. ----
__global__ __launch_bounds__(512) void attn_bwd_dkdv7_kernel(Bwd2Params pp) {
    const BwdParams& p = pp.p;
    constexpr int RING = 4;
    constexpr int DO_OFF = RING * ROWT_EL * 2;              // byte distance sT[0][b] -> sT[1][b]
    __shared__ __attribute__((aligned(16))) bf16_t sT[2][RING][ROWT_EL];    // [0] Q tiles, [1] dO tiles
    __shared__ __attribute__((aligned(16))) uint4 sSt[RING][64];               // seed operand rows (entries 32..63 stay zero: the hi lanes' k-slots)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int j = lane & 31, hi = lane >> 5;
    int blk, hb;
    xcd_block((int)blockIdx.x, (p.nk + 255) / 256, p.heads * p.batch, blk, hb);
    const int h = hb % p.heads, b = hb / p.heads;
    const int kw0 = blk * 256 + wave * 32;
    const bf16_t* Q = p.q + (long)b * p.q_sb + h * HD;
    const bf16_t* dO = p.dout + (long)b * p.do_sb + h * HD;
    const bf16_t* Kp = p.k + (long)b * p.k_sb + h * HD;
    const bf16_t* Vp = p.v + (long)b * p.v_sb + h * HD;
    const long stat0 = ((long)b * p.heads + h) * p.nq;
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const long r = min(kw0 + j, p.nk - 1);
        load_frag_agpr(kf[ks], Kp + r * p.k_ld + ks * 16 + hi * 8);
        load_frag_agpr(vf[ks], Vp + r * p.v_ld + ks * 16 + hi * 8);
    }
    TG_WAIT_FRAGS1(kf);
    TG_WAIT_FRAGS1(vf);
    Frag oS, oD;                                            // B operands of the seed k-step (k-slots 0..2 -> S, 3..5 -> dP; lanes hi = 0 only)
#pragma unroll
    for (int w = 0; w < 4; ++w) { oS.w[w] = 0; oD.w[w] = 0; }
    if (hi == 0) { oS.w[0] = 0x3F803F80u; oS.w[1] = 0x00003F80u; oD.w[1] = 0x3F800000u; oD.w[2] = 0x3F803F80u; }
    f32x16 dk[2], dv[2];                                   // [d block]: rows = keys, column = head dim j
#pragma unroll
    for (int c = 0; c < 2; ++c) { dk[c] = zero16(); dv[c] = zero16(); }
    // staging: group 0's threads own Q (+ wave 0, lanes 0..31: the statistics), group 1's dO; one 16-byte chunk per thread and tile
    const int t8 = tid & 255, row = t8 >> 3, chunk = (t8 & 7) * 8;
    const bf16_t* const srcR = grp ? dO : Q;
    const long ldR = grp ? p.do_ld : p.q_ld;
    bf16_t* const dstR = &sT[grp][0][0] + row * LQ2 + chunk;
    const int ntile = (p.nq + BT - 1) / BT;
    // running pointers: a tile that lies wholly inside the query range costs one load and one pointer add (and one more of each for the seed rows in
    // wave 0); clamped rows + masking for the ragged last tile only.  VALU work in Y is expensive: the partner wave's MFMA stream owns the issue port
    // (measured ~13 cycles per VALU instruction there), the address arithmetic of the first version cost 270-370 cycles per tile
    const bf16_t* pR = srcR + (long)row * ldR + chunk;
    const long stepR = (long)BT * ldR;
    const uint4* pS = p.seed + stat0 + j;
    // the last tile (and the never-read ones behind it) comes from clamped rows and is masked at the LDS write.  Loads stay unconditional and
    // branch-free (a load under a branch made hipcc wait for it on the spot: 2000 cycles per tile)
    const int qlast = (ntile - 1) * BT;
    const bf16_t* const pLast = srcR + (long)min(qlast + row, p.nq - 1) * ldR + chunk;
    const uint4* const pSLast = p.seed + stat0 + min(qlast + j, p.nq - 1);
    const bool okLast = qlast + row < p.nq, okSLast = qlast + j < p.nq;
    uint4 maskrow;                                          // seed row of a masked query: l = -1e30 (P = exp2(c (s - 1e30)) = 0), d = 0
    {
        uint32_t l1, l2, l3;
        split_bf16x3(-1e30f, l1, l2, l3);
        maskrow = uint4{(l1 >> 16) | l2, l3 >> 16, 0u, 0u};
    }
    uint4 g0, gseed = maskrow;
    bool inner = true;                                      // the fetched tile lies wholly inside the query range
    int tf = 0;                                             // tile the next fetch() loads
    auto fetch = [&]() {
        inner = tf < ntile - 1;
        g0 = *(const uint4*)(inner ? pR : pLast);
        if (wave == 0) gseed = *(inner ? pS : pSLast);
        pR += stepR; pS += BT;
        ++tf;
    };
    auto stash = [&](int buf) {
        if (inner) {
            *(uint4*)(dstR + buf * ROWT_EL) = g0;
            if (wave == 0 && hi == 0) sSt[buf][j] = gseed;
        } else {
            *(uint4*)(dstR + buf * ROWT_EL) = mask16(g0, okLast);
            if (wave == 0 && hi == 0) sSt[buf][j] = sel16(okSLast, gseed, maskrow);
        }
    };
    // prologue: Q(0), Q(1) + their statistics, dO(0..2); the buffers of "tile -1" (index RING - 1) zeroed — X(0) multiplies them by P = dS = 0
    *(uint4*)(dstR + (RING - 1) * ROWT_EL) = uint4{0, 0, 0, 0};
    if (wave == 0 && hi == 1) {
#pragma unroll
        for (int bb = 0; bb < RING; ++bb) sSt[bb][lane] = uint4{0, 0, 0, 0};
    }
    fetch();
    stash(0);
    fetch();
    stash(1);
    fetch();
    if (grp == 1) {
        stash(2);
        fetch();
    }
    Frag pA[2], dA[2];                                      // P^T / dS^T A operands, carried from Y(t) to X(t+1)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int w = 0; w < 4; ++w) { pA[t].w[w] = 0; dA[t].w[w] = 0; }
    f32x16 s, dp;                                           // rows = queries, column = key j
    // lane bases inside a tile: A rows (row j, 16-byte slot hi of every k-step), transposed 4 x 16 blocks (see lds_tr_b64), seed rows
    const uint32_t offA = (uint32_t)((j * LQ2 + hi * 8) * 2);
    const uint32_t offB = (uint32_t)(((4 * hi + ((lane & 15) >> 2)) * LQ2 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2);
    const uint32_t ldsT = lds_addr(&sT[0][0][0]), ldsS = lds_addr(&sSt[0][0]) + (uint32_t)lane * 16;
    Frag fr[4][2];
    // dV / dK products of the tile in buffer pb: fragment group i (t2 = i >> 1, db = i & 1) = 4 transposed reads
#define BWD_RD_T(dst, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(vB), "i"(off))
#define BWD_RD_A(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "i"(off))
#define BWD_WAIT(n, f) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f[0].v), "+v"(f[1].v))
    auto xseg = [&](const int cb, const int pb, const bool tail) {
        const uint32_t vB = ldsT + (uint32_t)(pb * ROWT_EL * 2) + offB;
        const uint32_t vA = ldsT + (uint32_t)(cb * ROWT_EL * 2) + offA;
        const uint32_t vS = ldsS + (uint32_t)(cb * 1024);
        // (every MFMA statement starts with s_nop 1: hipcc pads VALU -> MFMA-operand hazards only for instructions it knows, and it does move
        //  accumulator blocks between AGPR ranges with v_accvgpr_mov right in front of these statements — without the wait states the products
        //  read the accumulator's old contents)
        // fragment group i: 0..3 = transposed dO | Q blocks (t2 = i >> 1, db = i & 1) of tile pb, 4 = seed row, 5..8 = Q | dO rows of k-step i - 5 of tile cb
#define BWD_ISSUE(i)                                                                                                              \
        do {                                                                                                                      \
            Frag(&f)[2] = fr[(i) & 3];                                                                                            \
            if ((i) < 4) {                                                                                                        \
                constexpr int o = ((i) >> 1) * (16 * LQ2 * 2) + ((i) & 1) * 64;                                                   \
                BWD_RD_T(f[0].u[0], DO_OFF + o); BWD_RD_T(f[0].u[1], DO_OFF + o + 8 * LQ2 * 2);                                   \
                BWD_RD_T(f[1].u[0], o);          BWD_RD_T(f[1].u[1], o + 8 * LQ2 * 2);                                            \
            } else if ((i) == 4) {                                                                                                \
                BWD_RD_A(f[0].v, vS, 0);                                                                                          \
            } else {                                                                                                              \
                constexpr int ks = (i) >= 5 ? (i) - 5 : 0;                                                                        \
                BWD_RD_A(f[0].v, vA, ks * 32); BWD_RD_A(f[1].v, vA, DO_OFF + ks * 32);                                            \
            }                                                                                                                     \
        } while (0)
#define BWD_CONSUME(i)                                                                                                            \
        do {                                                                                                                      \
            Frag(&f)[2] = fr[(i) & 3];                                                                                            \
            if ((i) < 4) {                                                                                                        \
                constexpr int t2 = ((i) >> 1) & 1, db = (i) & 1;                                                                  \
                asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dv[db]) : "v"(pA[t2].v), "v"(f[0].v));              \
                asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dk[db]) : "v"(dA[t2].v), "v"(f[1].v));              \
            } else if ((i) == 4) {                                                                                                \
                asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %3, 0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %4, 0"                   \
                    : "=&v"(s), "=&v"(dp) : "v"(f[0].v), "a"(oS.v), "a"(oD.v));                                                   \
            } else {                                                                                                              \
                constexpr int ks = (i) >= 5 ? (i) - 5 : 0;                                                                        \
                asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %5, %1"                 \
                    : "+v"(s), "+v"(dp) : "v"(f[0].v), "v"(f[1].v), "a"(kf[ks]), "a"(vf[ks]));                                    \
            }                                                                                                                     \
        } while (0)
        // reads run three groups ahead of their MFMAs (at most 13 LDS operations in flight: lgkmcnt has 4 bits); lgkmcnt(n) = the reads of the
        // younger groups that may still be in flight (LDS returns in order).  Groups 0..2 were issued by xprefetch() BEFORE the barrier that
        // opens this segment (the tile they read was complete long ago), so the segment starts on landed data
        if (tail) {
            TG_SB(); BWD_WAIT(8, fr[0]); BWD_ISSUE(3); TG_SB(); BWD_CONSUME(0);
            TG_SB(); BWD_WAIT(8, fr[1]); BWD_CONSUME(1);
            TG_SB(); BWD_WAIT(4, fr[2]); BWD_CONSUME(2);
            TG_SB(); BWD_WAIT(0, fr[3]); BWD_CONSUME(3);
            TG_SB();
            return;
        }
        TG_SB(); BWD_WAIT(8, fr[0]); BWD_ISSUE(3); TG_SB(); BWD_CONSUME(0); TG_SB(); BWD_ISSUE(4);
        TG_SB(); BWD_WAIT(9, fr[1]); BWD_CONSUME(1); TG_SB(); BWD_ISSUE(5);
        TG_SB(); BWD_WAIT(7, fr[2]); BWD_CONSUME(2); TG_SB(); BWD_ISSUE(6);
        TG_SB(); BWD_WAIT(5, fr[3]); BWD_CONSUME(3); TG_SB(); BWD_ISSUE(7);
        TG_SB(); BWD_WAIT(6, fr[0]); BWD_CONSUME(4); TG_SB(); BWD_ISSUE(8);
        TG_SB(); BWD_WAIT(6, fr[1]); BWD_CONSUME(5);
        TG_SB(); BWD_WAIT(4, fr[2]); BWD_CONSUME(6);
        TG_SB(); BWD_WAIT(2, fr[3]); BWD_CONSUME(7);
        TG_SB(); BWD_WAIT(0, fr[0]); BWD_CONSUME(8);
        TG_SB();
        asm volatile("s_nop 11" : "+v"(s), "+v"(dp));      // an 8-pass MFMA result needs 12 wait states before VALU reads it (the barrier may be short)
    };
    auto xprefetch = [&](const int pb) {                    // fragment groups 0..2 of the next X: transposed blocks of the tile in buffer pb
        const uint32_t vB = ldsT + (uint32_t)(pb * ROWT_EL * 2) + offB, vA = 0, vS = 0;     // (vA, vS: dead branches of the macro)
        BWD_ISSUE(0); BWD_ISSUE(1); BWD_ISSUE(2);
    };
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    xprefetch(RING - 1);
    if (grp == 1) BWD_BAR();                                // group 1 falls one segment behind group 0
    int cb = 0, pb = RING - 1, wb = 2 + grp;                // current / previous tile buffer; buffer this thread stages next
    for (int it = 0; it < ntile; ++it) {
        // ---------------- X(it): matrix segment ----------------
        __builtin_amdgcn_s_setprio(2);
        xseg(cb, pb, false);
        __builtin_amdgcn_s_setprio(0);
        BWD_BAR();
        // ---------------- Y(it): vector segment ----------------
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const float p0 = fast_exp2(s[4 * g + e] * p.scale_log2), p1 = fast_exp2(s[4 * g + e + 1] * p.scale_log2);
                const float ds0 = p0 * dp[4 * g + e], ds1 = p1 * dp[4 * g + e + 1];
                pA[g >> 1].w[(g & 1) * 2 + (e >> 1)] = pack_bf16x2_trans(p0, p1);
                dA[g >> 1].w[(g & 1) * 2 + (e >> 1)] = pack_bf16x2(ds0, ds1);
            }
        stash(wb);                                          // tile it + 2 + grp (fetched one tile period ago); its LDS write is not waited for here: the
                                                            // partner group's X keeps the LDS queue full (measured 450-700 cycles), and the tile is first read two
                                                            // barriers from now — this wave's own reads in X(it + 1) return behind the write
        fetch();
        wb = (wb + 1) & (RING - 1);
        pb = cb;
        cb = (cb + 1) & (RING - 1);
        TG_SB();
        xprefetch(pb);
        BWD_BAR();
    }
    __builtin_amdgcn_s_setprio(2);
    xseg(cb, pb, true);                                     // X(ntile): the last tile's dV / dK products
    __builtin_amdgcn_s_setprio(0);
    if (grp == 0) BWD_BAR();                                // pairs with group 1's extra barrier
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        float* DK = p.dk + (long)b * p.dk_sb + h * HD + db * 32 + j;
        float* DV = p.dv + (long)b * p.dv_sb + h * HD + db * 32 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kw0 + acc_row(r, hi);
            if (key >= p.nk) continue;
            float* a = DK + (long)key * p.dk_ld;
            const float vk = dk[db][r] * p.scale;
            float vv = dv[db][r];
            *a = (p.accumulate & 2) ? *a + vk : vk;
            if (p.dv) {
                float* c = DV + (long)key * p.dv_ld;
                if (p.accumulate & 2) vv += *c;
                *c = vv;
            }
            if (p.dvb) p.dvb[(long)b * p.dvb_sb + (long)key * p.dvb_ld + h * HD + db * 32 + j] = f32_to_bf16(vv);
        }
    }
}

// ---- (3') dQ: workgroup = 256 queries (64 per wave), keys streamed in tiles of 32 ----
__global__ __launch_bounds__(256) void attn_bwd_dq2_kernel(Bwd2Params pp) {
    const BwdParams& p = pp.p;
    __shared__ __attribute__((aligned(16))) bf16_t sK[2][ROWT_EL], sV[2][ROWT_EL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hi = lane >> 5;
    int blk, hb;
    const int nqb = (p.nq + 255) / 256, per_part = nqb * p.heads * p.batch;
    const int part = pp.kparts > 1 ? (int)blockIdx.x / per_part : 0;
    xcd_block((int)blockIdx.x - part * per_part, nqb, p.heads * p.batch, blk, hb);
    const int h = hb % p.heads, b = hb / p.heads;
    const int qw0 = blk * 256 + wave * 64;
    const bf16_t* Q = p.q + (long)b * p.q_sb + h * HD;
    const bf16_t* dO = p.dout + (long)b * p.do_sb + h * HD;
    const bf16_t* Kp = p.k + (long)b * p.k_sb + h * HD;
    const bf16_t* Vp = p.v + (long)b * p.v_sb + h * HD;
    const long stat0 = ((long)b * p.heads + h) * p.nq;
    bf16x8 qf[2][4], of[2][4];                             // B operands: this wave's queries, resident in AGPRs
    float lse[2], dsum[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int q = qw0 + qb * 32 + j;
        const long r = min(q, p.nq - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            load_frag_agpr(qf[qb][ks], Q + r * p.q_ld + ks * 16 + hi * 8);
            load_frag_agpr(of[qb][ks], dO + r * p.do_ld + ks * 16 + hi * 8);
        }
        lse[qb] = q < p.nq ? p.lse[stat0 + q] : 1e30f;           // masked query rows: P = 0 (their Q / dO fragments are a clamped copy)
        dsum[qb] = q < p.nq ? p.dsum[stat0 + q] : 0.f;
    }
    TG_WAIT_FRAGS(qf);
    TG_WAIT_FRAGS(of);
    f32x16 dq[2][2];                                       // [query block][d block]: rows = queries, column = head dim j
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) dq[a][c] = zero16();
    const int row = tid >> 3, chunk = (tid & 7) * 8;
    const int trb = (4 * hi + ((lane & 15) >> 2)) * LQ2 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4;     // transpose-read base of this lane inside a [32][LQ2] tile: row 4 hi + (a >> 2), 4-element chunk (a & 3) of its 16-lane group's columns (lds_tr_b64)
    const int ntile_all = (p.nk + BT - 1) / BT;
    const int tile0 = pp.kparts > 1 ? (int)((long)part * ntile_all / pp.kparts) : 0;                  // this workgroup's key tiles [tile0, tile0 + ntile)
    const int ntile = (pp.kparts > 1 ? (int)((long)(part + 1) * ntile_all / pp.kparts) : ntile_all) - tile0;
    uint4 g0, g1;
    bool okr = false;
    // running pointers: tiles are fetched in order, so a tile that lies wholly inside the key range costs three loads and three pointer adds — the
    // clamped 64-bit index arithmetic is kept for the ragged last tile only (this kernel is VALU-bound: 61 % VALU busy against 51 % MFMA busy)
    const bf16_t* pK = Kp + ((long)tile0 * BT + row) * p.k_ld + chunk;
    const bf16_t* pV = Vp + ((long)tile0 * BT + row) * p.v_ld + chunk;
    const long stepK = (long)BT * p.k_ld, stepV = (long)BT * p.v_ld;
    auto fetch = [&](int k0) {
        if (k0 + BT <= p.nk) {
            okr = true;
            g0 = *(const uint4*)pK;
            g1 = *(const uint4*)pV;
        } else {
            okr = k0 + row < p.nk;
            g0 = ld_row16_clamped(Kp, p.k_ld, k0 + row, p.nk, chunk);
            g1 = ld_row16_clamped(Vp, p.v_ld, k0 + row, p.nk, chunk);
        }
        pK += stepK; pV += stepV;
    };
    auto stash = [&](int buf) {
        *(uint4*)(sK[buf] + row * LQ2 + chunk) = mask16(g0, okr);
        *(uint4*)(sV[buf] + row * LQ2 + chunk) = mask16(g1, okr);
    };
    fetch(tile0 * BT);
    stash(0);
    __syncthreads();
    for (int it = 0; it < ntile; ++it) {
        const int buf = it & 1;
        if (it + 1 < ntile) fetch((tile0 + it + 1) * BT);
        bf16x8 aK[4], aV[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            aK[ks] = *(const bf16x8*)(sK[buf] + j * LQ2 + ks * 16 + hi * 8);
            aV[ks] = *(const bf16x8*)(sV[buf] + j * LQ2 + ks * 16 + hi * 8);
        }
        TG_SB();
        Frag bK[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int o = trb + 16 * t * LQ2 + db * 32;
                bK[t][db].u[0] = lds_tr_b64(sK[buf] + o); bK[t][db].u[1] = lds_tr_b64(sK[buf] + o + 8 * LQ2);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bK[0][0].v), "+v"(bK[0][1].v), "+v"(bK[1][0].v), "+v"(bK[1][1].v));
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 st, dpt;                                // rows = keys, column = query j
            mfma_pair(st, dpt, aK, aV, qf[qb], of[qb]);
            Frag dA[2];
            const f32x2 lv = {lse[qb], lse[qb]}, dv2 = {dsum[qb], dsum[qb]};
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const f32x2 sv = {st[4 * g + e], st[4 * g + e + 1]}, dpv = {dpt[4 * g + e], dpt[4 * g + e + 1]};
                    // scalar fp32 arithmetic: the packed forms (v_pk_fma / add / mul) measured 3.5 % slower beside the MFMAs
                    const f32x2 pv = {fast_exp2(sv[0] * p.scale_log2 - lv[0]), fast_exp2(sv[1] * p.scale_log2 - lv[1])};
                    const f32x2 ds = {pv[0] * (dpv[0] - dv2[0]), pv[1] * (dpv[1] - dv2[1])};
                    dA[g >> 1].w[(g & 1) * 2 + (e >> 1)] = pack_bf16x2(ds[0], ds[1]);
                }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    mfma_acc(dq[qb][db], dA[t].v, bK[t][db].v);
        }
        if (it + 1 < ntile) stash(buf ^ 1);
        __syncthreads();
    }
    asm volatile("s_nop 15" ::: "memory");
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const long pld = (long)p.heads * HD;                                  // partial tensors: [part][batch][nq][heads * 64]
            float* DQ = pp.kparts > 1 ? pp.dq_part + ((long)part * p.batch + b) * p.nq * pld + h * HD + db * 32 + j : p.dq + (long)b * p.dq_sb + h * HD + db * 32 + j;
            const long qld = pp.kparts > 1 ? pld : p.dq_ld;
            const bool add = pp.kparts > 1 ? false : (p.accumulate & 1) != 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = qw0 + qb * 32 + acc_row(r, hi);
                if (q >= p.nq) continue;
                float* a = DQ + (long)q * qld;
                const float vq = dq[qb][db][r] * p.scale;
                *a = add ? *a + vq : vq;
            }
        }
}

// dq[b][q][:] = (accumulate ? dq : 0) + sum over the key ranges, in range order, of the dQ launch's partial tensors (4 floats per thread)
__global__ __launch_bounds__(256) void attn_bwd_dq_join_kernel(Bwd2Params pp) {
    const BwdParams& p = pp.p;
    const long pld = (long)p.heads * HD, per_b = (long)p.nq * pld, i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= (long)p.batch * per_b) return;
    const int b = (int)(i / per_b);
    const long rem = i - (long)b * per_b;
    const int q = (int)(rem / pld), c = (int)(rem - (long)q * pld);
    float* dst = p.dq + (long)b * p.dq_sb + (long)q * p.dq_ld + c;
    f32x4 a = (p.accumulate & 1) ? *(const f32x4*)dst : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < pp.kparts; ++k) {
        const f32x4 v = *(const f32x4*)(pp.dq_part + (long)k * p.batch * per_b + i);
        a = f32x4{a[0] + v[0], a[1] + v[1], a[2] + v[2], a[3] + v[3]};
    }
    *(f32x4*)dst = a;
}

// =================================================================================================================================
// ONE kernel for dK, dV AND dQ (5 executed GEMMs instead of 7: S and dP are formed once) — attn_bwd_fused_pp_kernel below.  What it shares with the
// two-launch form: a workgroup owns 256 keys (8 waves x 32) and walks the query tiles.  Per tile it also forms its 256 keys' contribution to the tile's
// dQ [32 q][64 d]: every wave leaves its dS^T block [32 keys][32 queries] (bf16) in LDS, 16 x 16 blocks of dQ^T are formed on v_mfma_f32_16x16x32_bf16
// (A = the workgroup's K rows, resident; B = dS through transposed reads of the [key][query] tiles; a lane ends up with 4 consecutive floats of one dQ row).
// The key blocks of a head add their blocks to the fp32 dQ tile in global memory IN KEY-BLOCK ORDER (bitwise reproducible, no atomics on the data): per
// (head, tile, row half) a counter says how many key blocks have added; key block kb waits for the counter to reach kb, every thread reads 16 bytes of the
// tile (the blocks meet in an LDS tile so that 16 lanes cover a whole 256-byte dQ row: full lines — half-line stores were not kept in this L2), adds, writes,
// and when every wave's write has been acknowledged by L2 (vmcnt(0) + a barrier) the counter is set to kb + 1.  All workgroups of a head run on ONE XCD
// (xcd_block), whose L2 is the coherence point.  fused_probe_kernel checks exactly these primitives and the workgroup -> XCD mapping on the device before a
// caller may select this form.  The wait never points at a workgroup that has not started (kb - 1 has the lower index).  The read-add-write of a tile is
// spread over several iterations, so consecutive key blocks run a few tiles apart — which is why the launcher uses the form only for calls with many more
// query tiles than key blocks.  History (profiles/NOTES.md §D, §F): the first, in-phase version of this kernel (round 3: one barrier per tile, every wave in
// the same phase) ran the 17776^2 call in 26.4 ms; the ping-pong version below in 23.7 ms on the same box.
// =================================================================================================================================
struct FusedParams {
    BwdParams p;
    int* cnt;            // [heads * batch][query tiles][32]: key blocks that have added their dQ blocks, one counter per 128-byte line (zeroed by the launcher)
    int* status;         // caller-owned int32[4]: [0] sticky count of exchange polls that timed out (dq of such a launch is INVALID), [1] poll limit (0: 2^20),
                         // [2] sticky count of workgroups that found their head's key blocks on MORE than one XCD (the exchange then is not coherent: dq INVALID)
    int* xmask;          // [heads * batch]: bit x set = a key block of this (batch, head) ran on XCD x (zeroed by the launcher)
    // "rider": a second call of the same heads / batch whose workgroups are appended to the launch (workgroup index >= main_wgs; main_wgs is a multiple of 8, so a
    // rider workgroup keeps its XCD residue).  The training step's main call leaves 6720 - 26 * 256 = 64 workgroups for its last round of 256 CUs; the vip-key call of the
    // same processor (192 workgroups of the same length: 17776 queries) rides in that round instead of costing a launch of its own.
    BwdParams r;
    int* r_cnt;
    int* r_xmask;
    int main_wgs;        // workgroups of the main call; the grid is main_wgs + (r.nq > 0 ? rider workgroups : 0)
};
constexpr int DSLD = 40;                 // dS^T tile row stride in elements (80 B: 8-byte aligned 4-query runs)
constexpr int DQLD = 68;                 // dQ tile row stride in floats (272 B: the 16 lanes of a block column land on different banks)

// The counters and the dQ blocks are exchanged between workgroups of ONE XCD, whose L2 is the coherence point.  What the kernel uses, and what
// tg_attention_bwd_probe checks on the device before a caller may select this form: dQ lines are read with PLAIN 16-byte loads behind one
// `buffer_inv sc1` per workgroup (every line is loaded once, and only after the key block before has signalled the whole tile, so the CU's L1 never holds a
// stale copy of it); they are written with plain stores, which go through the CU's L1 to the XCD's L2; the counter is written with a plain store behind
// every wave's vmcnt(0) + a barrier, and polled with sc1 loads (past the L1).  No atomics on the data or on the counters.  Measured on the way:
// agent-scope atomics compile to sc1 accesses that travel to memory behind the L2 (every iteration waited 600-1400 cycles for its signal store's
// acknowledgement); the sc0 bit alone is workgroup scope and does not bypass the L1 (stale counters, polls that never end); buffer_inv sc1 before
// every plain load is correct and 8x slower than the whole kernel.
__device__ __forceinline__ int cnt_read(int* c) {
    int v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(c) : "memory");
    return v;
}
__device__ __forceinline__ void cnt_write(int* c, int v) { asm volatile("global_store_dword %0, %1, off" :: "v"(c), "v"(v) : "memory"); }

// =================================================================================================================================
// The one-kernel backward as a TWO-GROUP PING-PONG (round 5): attn_bwd_dkdv7_kernel's structure — every wave alternates a matrix segment X (all MFMAs, all
// LDS reads, every read issued three fragment groups ahead of its use) and a vector segment Y (VALU, LDS writes, global traffic), one s_barrier after
// each, waves 4-7 one segment behind waves 0-3 — with the dQ product and the ordered dQ exchange (banner above FusedParams) laid INTO those segments.
// The in-phase one-kernel form ran at 0.78 PFLOP/s on the 17776^2 call (every wave of a SIMD in the same phase: their MFMA blocks collide, their softmax
// blocks collide, the matrix pipe idles through both softmaxes), and on all-zero operands it gained only 5.6 % where the forward gained 27 %: schedule-bound.
//   X(t) = { dQ^T partial of tile t-2 ; dV += P(t-1)^T dO(t-1), dK += dS(t-1)^T Q(t-1) ; S(t) = Q(t) K^T, dP(t) = dO(t) V^T }
//          8 MFMAs 16x16x32 + 18 MFMAs 32x32x16, 32 transposed b64 + 9 b128 LDS reads, 2 b128 LDS writes (the dQ partial)
//   Y(t) = { exchange of tile t - lag (read-add-write of 16 bytes per thread) ; stage write + fetch of a later tile ; P = exp2(S), dS = P o dP -> bf16 A operands,
//            dS^T -> LDS }                                                                                                VALU + memory only
// dQ: a GROUP (4 waves, 128 keys) forms its own partial of the tile's dQ^T [64 d][32 q] — wave w the two 16 x 16 blocks of head dims 16 w .. over both query
// halves, A = its K rows (resident, 16 AGPRs), B = the group's dS^T blocks through transposed reads — TWO tiles behind the softmax that wrote them: a
// dS^T block written in Y(t) is complete when its writer's X(t+1) has drained its LDS queue, so X(t+2) reads it with no wait anywhere (two slots).  The two
// groups' partials meet in LDS ([group][slot][32 q][64 d] fp32) and are added by the exchange: group g owns rows 16 g .. 16 g + 15 of every tile and is its own
// chain (counter word g of the tile's line): key block kb's group g adds to what key block kb - 1's group g left — in key-block order, bitwise reproducible,
// no atomics on the data; coherence through the XCD's L2 exactly as in the in-phase form (plain stores, one buffer_inv sc1 per workgroup, sc1 polls).
//   tile T: dS^T in Y(T) -> partial in X(T+2) -> [group 1: exchanged in Y(T+2); group 0: in Y(T+3)] -> acknowledged at the END of that same Y (round 6; round 5:
//   at the top of the next Y) -> signalled at the head of the next X.  The loop runs PP_EXTRA masked tiles past the last one (P = dS = 0 exactly: they add nothing to dK / dV) instead of a drain.
// scale * log2(e) == 1 (the training step hands over its main call's K prescaled by scale * log2 e, and scale = ln 2): P = exp2 of the accumulator, no multiply — a
// workgroup-uniform choice between two copies of the softmax (the main call and a rider may differ).  Measured and dropped: folding the factor into the resident K
// fragments for every caller (bf16(k * scale_log2): no multiply anywhere) — the forward had used the unrounded K, and the gradient error of such calls doubled (dV 2.3e-3 -> 4.9e-3).
// =================================================================================================================================
constexpr int PP_EXTRA = 3;
constexpr int PP_RING = 4;
constexpr int PP_ST_BYTES = 2 * PP_RING * ROWT_EL * 2;          // Q | dO tiles
constexpr int PP_SEED_BYTES = PP_RING * 64 * 16;
constexpr int PP_DS_SLOT = 256 * DSLD * 2;                      // one dS^T tile [256 keys][DSLD] bf16
constexpr int PP_DQ_SLOT = BT * DQLD * 4;                       // one dQ partial [32 q][DQLD] fp32
constexpr int PP_LDS = PP_ST_BYTES + PP_SEED_BYTES + 2 * PP_DS_SLOT + 4 * PP_DQ_SLOT;

// global accesses of the vector segment: wave-uniform 64-bit base in SGPRs + a 32-bit lane offset.  A tile step is then ONE scalar add — VALU instructions
// are the currency of Y (the partner wave's MFMA stream owns the issue port: ~13 cycles per VALU instruction there)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;      // (a native vector: HIP's uint4 struct cannot be a tied "+v" asm operand)
__device__ __forceinline__ u32x4 mask16v(u32x4 v, bool ok) { return u32x4{ok ? v[0] : 0u, ok ? v[1] : 0u, ok ? v[2] : 0u, ok ? v[3] : 0u}; }
__device__ __forceinline__ u32x4 sel16v(bool c, u32x4 a, u32x4 b) { return u32x4{c ? a[0] : b[0], c ? a[1] : b[1], c ? a[2] : b[2], c ? a[3] : b[3]}; }
__device__ __forceinline__ void gld16(u32x4& dst, const void* sbase, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void gld16f(f32x4& dst, const void* sbase, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void gst16f(const f32x4& v, void* sbase, uint32_t voff) {
    asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
}

__global__ __launch_bounds__(512) void attn_bwd_fused_pp_kernel(FusedParams fp) {
    const bool rider = (int)blockIdx.x >= fp.main_wgs;       // workgroup-uniform
    const BwdParams& p = rider ? fp.r : fp.p;
    int* const cnt_base = rider ? fp.r_cnt : fp.cnt;
    int* const xmask_base = rider ? fp.r_xmask : fp.xmask;
    constexpr int RING = PP_RING;
    constexpr int DO_OFF = RING * ROWT_EL * 2;              // byte distance Q tile b -> dO tile b
    extern __shared__ __attribute__((aligned(16))) char psm[];
    bf16_t* const sT = (bf16_t*)psm;                                                   // [2][RING][ROWT_EL]: [0] Q tiles, [1] dO tiles
    uint4* const sSt = (uint4*)(psm + PP_ST_BYTES);                                    // [RING][64] seed rows (entries 0..31 used: the seed k-step's A operand row j for
                                                                                       // BOTH lane halves — the hi = 1 half's k-slots 8..15 meet zero B operands, any finite value will do)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wg = wave & 3;
    const int j = lane & 31, hi = lane >> 5, t16 = lane & 15, g4 = lane >> 4;
    int blk, hb;
    xcd_block((int)blockIdx.x - (rider ? fp.main_wgs : 0), (p.nk + 255) / 256, p.heads * p.batch, blk, hb);
    const int h = hb % p.heads, b = hb / p.heads;
    const int kw0 = blk * 256 + wave * 32;
    const bf16_t* Q = p.q + (long)b * p.q_sb + h * HD;
    const bf16_t* dO = p.dout + (long)b * p.do_sb + h * HD;
    const bf16_t* Kp = p.k + (long)b * p.k_sb + h * HD;
    const bf16_t* Vp = p.v + (long)b * p.v_sb + h * HD;
    const long stat0 = ((long)b * p.heads + h) * p.nq;
    // Self-check of the ONE placement property the ordered exchange rests on — every key block of a head on the same XCD (its L2 is where their dQ traffic
    // meets): each workgroup ORs its XCC id into the head's mask word and looks at what was there (device-scope atomic; one per workgroup, its latency hidden
    // behind the K / V fragment loads).  A violation is counted in status[2] and reported by the host like a poll time-out — never a silently wrong dq.
    int xold = 0, xbit = 0;
    if (tid == 0) {
        xbit = 1 << (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u);           // HW_REG_XCC_ID, bits 3:0
        xold = __hip_atomic_fetch_or(xmask_base + hb, xbit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool unit = p.scale_log2 == 1.0f;                  // workgroup-uniform
    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const long r = min(kw0 + j, p.nk - 1);
        load_frag_agpr(kf[ks], Kp + r * p.k_ld + ks * 16 + hi * 8);
        load_frag_agpr(vf[ks], Vp + r * p.v_ld + ks * 16 + hi * 8);
    }
    TG_WAIT_FRAGS1(kf);
    TG_WAIT_FRAGS1(vf);
    if (tid == 0 && (xold & ~xbit)) atomicAdd(fp.status + 2, 1);
    Frag oS, oD;                                            // B operands of the seed k-step (k-slots 0..2 -> S, 3..5 -> dP; lanes hi = 0 only)
#pragma unroll
    for (int w = 0; w < 4; ++w) { oS.w[w] = 0; oD.w[w] = 0; }
    if (hi == 0) { oS.w[0] = 0x3F803F80u; oS.w[1] = 0x00003F80u; oD.w[1] = 0x3F800000u; oD.w[2] = 0x3F803F80u; }
    f32x16 dk[2], dv[2];                                   // [d block]: rows = keys, column = head dim j
#pragma unroll
    for (int c = 0; c < 2; ++c) { dk[c] = zero16(); dv[c] = zero16(); }
    // dQ^T blocks of this wave: head dims 16 wg .. 16 wg + 15 x queries (16 qh .. ), qh = 0, 1, over the GROUP's 128 keys = 4 k-steps of 32.
    // A operand (rows = head dims, k = keys): lane (t16, g4) holds K[key 8 g4 + e][16 wg + t16], e = 0..7, of the k-step's 32 keys (zero beyond nk); resident
    Frag kq[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const int k0 = blk * 256 + 128 * grp + 32 * ks + 8 * g4 + e;
            const uint32_t lo = k0 < p.nk ? Kp[(long)k0 * p.k_ld + 16 * wg + t16] : 0u;
            const uint32_t hi16 = k0 + 1 < p.nk ? Kp[(long)(k0 + 1) * p.k_ld + 16 * wg + t16] : 0u;
            kq[ks].w[e >> 1] = lo | (hi16 << 16);
        }
    // ---- staging: group 0's threads own Q (+ wave 0, lanes 0..31: the seed rows), group 1's dO; one 16-byte chunk per thread and tile ----
    const int t8 = tid & 255, row = t8 >> 3, chunk = (t8 & 7) * 8;
    const long ldR = grp ? p.do_ld : p.q_ld;
    const int ntile = (p.nq + BT - 1) / BT;
    const int qlast = (ntile - 1) * BT;
    const char* sbR = (const char*)(grp ? dO : Q);          // scalar: base of the tile the next fetch() loads
    const long stepRb = (long)BT * ldR * 2;
    const uint32_t voR = (uint32_t)(((long)row * ldR + chunk) * 2);
    const uint32_t voRLast = (uint32_t)(((long)(min(qlast + row, p.nq - 1) - qlast) * ldR + chunk) * 2);
    const char* sbS = (const char*)(p.seed + stat0);
    const uint32_t voS = (uint32_t)j * 16u, voSLast = (uint32_t)(min(qlast + j, p.nq - 1) - qlast) * 16u;
    const bool okLast = qlast + row < p.nq, okSLast = qlast + j < p.nq;
    bf16_t* const dstR = sT + (grp * RING) * ROWT_EL + row * LQ2 + chunk;
    u32x4 maskrow;                                          // seed row of a masked query: l = -1e30 (P = exp2(s - 1e30) = 0), d = 0
    {
        uint32_t l1, l2, l3;
        split_bf16x3(-1e30f, l1, l2, l3);
        maskrow = u32x4{(l1 >> 16) | l2, l3 >> 16, 0u, 0u};
    }
    u32x4 g0 = u32x4{0, 0, 0, 0}, gseed = maskrow;
    int tf = 0;                                             // tile the next fetch() loads
    int kindF = 0;                                          // of the tile held in g0: 0 wholly inside the query range, 1 the (ragged) last tile, 2 past the end (all masked)
    auto fetch = [&]() {
        kindF = tf < ntile - 1 ? 0 : (tf == ntile - 1 ? 1 : 2);
        gld16(g0, sbR, kindF == 0 ? voR : voRLast);
        if (wave == 0) gld16(gseed, sbS, kindF == 0 ? voS : voSLast);
        if (tf < ntile - 1) { sbR += stepRb; sbS += BT * 16; }       // (tiles past the end re-read the last tile's rows: in bounds, masked below)
        ++tf;
    };
    auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0), "+v"(gseed)); };
    auto stash = [&](int buf) {
        if (kindF == 0) {
            *(u32x4*)(dstR + buf * ROWT_EL) = g0;
            if (wave == 0) *(u32x4*)(sSt + buf * 64 + j) = gseed;             // (both lane halves write row j: same address, same data — no exec masking in Y)
        } else {
            *(u32x4*)(dstR + buf * ROWT_EL) = mask16v(g0, kindF == 1 && okLast);
            if (wave == 0) *(u32x4*)(sSt + buf * 64 + j) = sel16v(kindF == 1 && okSLast, gseed, maskrow);
        }
    };
    // prologue: Q(0), Q(1) + their seed rows, dO(0..2); the buffers of "tile -1" (index RING - 1) zeroed — X(0) multiplies them by P = dS = 0
    *(uint4*)(dstR + (RING - 1) * ROWT_EL) = uint4{0, 0, 0, 0};
    fetch(); landed(); stash(0);
    fetch(); landed(); stash(1);
    fetch();
    if (grp == 1) {
        landed(); stash(2);
        fetch();
    }
    Frag pA[2], dA[2];                                      // P^T / dS^T A operands, carried from Y(t) to X(t+1)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int w = 0; w < 4; ++w) { pA[t].w[w] = 0; dA[t].w[w] = 0; }
    f32x16 s, dp;                                           // rows = queries, column = key j
    // ---- LDS lane bases ----
    const uint32_t lds0 = lds_addr(psm);
    const uint32_t offA = (uint32_t)((j * LQ2 + hi * 8) * 2);
    const uint32_t offB = (uint32_t)(((4 * hi + ((lane & 15) >> 2)) * LQ2 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2);
    const uint32_t ldsT = lds0, ldsS = lds0 + PP_ST_BYTES + (uint32_t)j * 16;
    const uint32_t ldsDS = lds0 + PP_ST_BYTES + PP_SEED_BYTES, ldsDQ = ldsDS + 2 * PP_DS_SLOT;
    // dS^T [256 keys][DSLD]: this lane writes keys row (wave * 32 + j), queries 8 g + 4 hi .. (Y); reads the group's rows through transposed 4 x 16 blocks (X)
    const uint32_t dsW = ldsDS + (uint32_t)(((wave * 32 + j) * DSLD + 4 * hi) * 2);
    const uint32_t dsR = ldsDS + (uint32_t)(((128 * grp + 8 * g4 + (t16 >> 2)) * DSLD + (t16 & 3) * 4) * 2);
    // dQ partial [group][slot][32 q][DQLD]: this wave's two blocks land at rows (16 qh + t16), columns 16 wg + 4 g4 ..
    const uint32_t dqW = ldsDQ + (uint32_t)(grp * 2 * PP_DQ_SLOT) + (uint32_t)((t16 * DQLD + 16 * wg + 4 * g4) * 4);
    // exchange: group g owns rows 16 g + (t8 >> 4) of every tile, 16 lanes cover one whole 256-byte dQ row (full lines, as in the in-phase form)
    const int er = 16 * grp + (t8 >> 4), ec = (t8 & 15) * 4;
    const uint32_t dqR = ldsDQ + (uint32_t)((er * DQLD + ec) * 4);
    char* sbDQr = (char*)(p.dq + (long)b * p.dq_sb + h * HD);                    // scalar: dQ rows of the tile the next request loads
    char* sbDQw = sbDQr;                                                         //         ... the next write stores
    const long stepDQb = (long)BT * p.dq_ld * 4;
    const uint32_t voDQ = (uint32_t)(((long)er * p.dq_ld + ec) * 4);
    const uint32_t voDQLast = (uint32_t)(((long)(min(qlast + er, p.nq - 1) - qlast) * p.dq_ld + ec) * 4);
    const bool okDQLast = qlast + er < p.nq;
    const bool first = blk == 0 && !(p.accumulate & 1);      // nothing to read: this workgroup's block starts the sum
    constexpr int CNT_PAD = 32;                               // one 128-byte line per (head, tile); word g = group g's chain
    int* const cntw = cnt_base + (long)hb * ntile * CNT_PAD + grp;
    const int lag = 3 - grp;                                 // Y(u) exchanges tile u - lag (both groups' partials of it are complete and behind a barrier)
    f32x4 ldv = f32x4{0.f, 0.f, 0.f, 0.f};
    int cval = 0;                                           // (wave 4 g) counter of the next tile to check, sampled one Y ahead
    asm volatile("buffer_inv sc1" ::: "memory");            // ONCE per workgroup: whatever earlier workgroups on this CU left in its L1 is gone; from here on
                                                            // every dQ line is loaded once, and only after the key block before has completed the whole tile
    Frag fr[4][2];
    f32x4 qa0, qa1;                                         // this wave's dQ^T blocks (query halves 0, 1)
#define PP_RD_T(dst, base, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "i"(off))
#define PP_RD_A(dst, base, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "i"(off))
#define PP_WAIT(n, f) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(f[0].v), "+v"(f[1].v))
    // fragment group i of X: 0..3 = dS^T blocks of k-step i (query halves 0 | 1) of tile t - 2; 4..7 = transposed dO | Q blocks (t2 = (i - 4) >> 1, db = (i - 4) & 1) of
    // tile t - 1; 8 = seed row of tile t; 9..12 = Q | dO rows of k-step i - 9 of tile t
#define PP_ISSUE(i)                                                                                                               \
        do {                                                                                                                      \
            Frag(&f)[2] = fr[(i) & 3];                                                                                            \
            if ((i) < 4) {                                                                                                        \
                constexpr int o = ((i) & 3) * (32 * DSLD * 2);                                                                    \
                PP_RD_T(f[0].u[0], vD, o);      PP_RD_T(f[0].u[1], vD, o + 4 * DSLD * 2);                                         \
                PP_RD_T(f[1].u[0], vD, o + 32); PP_RD_T(f[1].u[1], vD, o + 4 * DSLD * 2 + 32);                                    \
            } else if ((i) < 8) {                                                                                                 \
                constexpr int o = ((((i) - 4) >> 1) & 1) * (16 * LQ2 * 2) + (((i) - 4) & 1) * 64;                                 \
                PP_RD_T(f[0].u[0], vB, DO_OFF + o); PP_RD_T(f[0].u[1], vB, DO_OFF + o + 8 * LQ2 * 2);                             \
                PP_RD_T(f[1].u[0], vB, o);          PP_RD_T(f[1].u[1], vB, o + 8 * LQ2 * 2);                                      \
            } else if ((i) == 8) {                                                                                                \
                PP_RD_A(f[0].v, vS, 0);                                                                                           \
            } else {                                                                                                              \
                constexpr int ks = (i) >= 9 ? ((i) - 9) & 3 : 0;                                                                  \
                PP_RD_A(f[0].v, vA, ks * 32); PP_RD_A(f[1].v, vA, DO_OFF + ks * 32);                                              \
            }                                                                                                                     \
        } while (0)
#define PP_CONSUME(i)                                                                                                             \
        do {                                                                                                                      \
            Frag(&f)[2] = fr[(i) & 3];                                                                                            \
            if ((i) == 0) {                                                                                                       \
                asm("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %3, 0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %4, 0"                   \
                    : "=&v"(qa0), "=&v"(qa1) : "a"(kq[0].v), "v"(f[0].v), "v"(f[1].v));                                           \
            } else if ((i) < 4) {                                                                                                 \
                asm("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %4, %1"                 \
                    : "+v"(qa0), "+v"(qa1) : "a"(kq[(i) & 3].v), "v"(f[0].v), "v"(f[1].v));                                       \
            } else if ((i) < 8) {                                                                                                 \
                constexpr int t2 = (((i) - 4) >> 1) & 1, db = ((i) - 4) & 1;                                                      \
                asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dv[db]) : "v"(pA[t2].v), "v"(f[0].v));              \
                asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dk[db]) : "v"(dA[t2].v), "v"(f[1].v));              \
            } else if ((i) == 8) {                                                                                                \
                asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %3, 0\n\tv_mfma_f32_32x32x16_bf16 %1, %2, %4, 0"                   \
                    : "=&v"(s), "=&v"(dp) : "v"(f[0].v), "a"(oS.v), "a"(oD.v));                                                   \
            } else {                                                                                                              \
                constexpr int ks = (i) >= 9 ? ((i) - 9) & 3 : 0;                                                                  \
                asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %3, %5, %1"                 \
                    : "+v"(s), "+v"(dp) : "v"(f[0].v), "v"(f[1].v), "a"(kf[ks]), "a"(vf[ks]));                                    \
            }                                                                                                                     \
        } while (0)
    // X(t): cb = buffer of tile t, pb = of tile t - 1, slot = (t & 1) = dS^T / dQ-partial slot of tile t - 2.
    // LDS operations in flight (lgkmcnt, 4 bits: <= 15): groups 0..2 were issued by xprefetch() before the barrier that opens the segment.  The two b128
    // writes of the dQ partial go out behind group 4's MFMAs (the blocks' last MFMA is three issue groups old by then) and are complete — LDS returns in
    // order — when group 8 is waited for: the partial is in LDS long before the segment's closing barrier.
    auto xseg = [&](const int cb, const int pb, const int slot) {
        const uint32_t vB = ldsT + (uint32_t)(pb * ROWT_EL * 2) + offB;
        const uint32_t vA = ldsT + (uint32_t)(cb * ROWT_EL * 2) + offA;
        const uint32_t vS = ldsS + (uint32_t)(cb * 1024);
        const uint32_t vD = dsR + (uint32_t)(slot * PP_DS_SLOT);
        const uint32_t vW = dqW + (uint32_t)(slot * PP_DQ_SLOT);
        TG_SB(); PP_WAIT(8, fr[0]); PP_ISSUE(3); TG_SB(); PP_CONSUME(0);                       // in flight behind the wait: 1 2 | after the issue: 1 2 3 (12)
        TG_SB(); PP_WAIT(8, fr[1]); PP_ISSUE(4); TG_SB(); PP_CONSUME(1);                       // 2 3 4 (12)
        TG_SB(); PP_WAIT(8, fr[2]); PP_ISSUE(5); TG_SB(); PP_CONSUME(2);                       // 3 4 5 (12)
        TG_SB(); PP_WAIT(8, fr[3]); PP_ISSUE(6); TG_SB(); PP_CONSUME(3);                       // 4 5 6 (12)
        TG_SB(); PP_WAIT(8, fr[0]); PP_ISSUE(7); TG_SB(); PP_CONSUME(4);                       // 5 6 7 (12)
        TG_SB();
        asm volatile("s_nop 7\n\tds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:%3" :: "v"(vW), "v"(qa0), "v"(qa1), "i"(16 * DQLD * 4) : "memory");   // 5 6 7 W (14)
        TG_SB(); PP_WAIT(10, fr[1]); PP_ISSUE(8); TG_SB(); PP_CONSUME(5);                      // 6 7 W 8 (11)
        TG_SB(); PP_WAIT(7, fr[2]); PP_ISSUE(9); TG_SB(); PP_CONSUME(6);                       // 7 W 8 9 (9)
        TG_SB(); PP_WAIT(5, fr[3]); PP_ISSUE(10); TG_SB(); PP_CONSUME(7);                      // W 8 9 10 (7)
        TG_SB(); PP_WAIT(4, fr[0]); PP_ISSUE(11); TG_SB(); PP_CONSUME(8);                      // 9 10 11 (6)
        TG_SB(); PP_WAIT(4, fr[1]); PP_ISSUE(12); TG_SB(); PP_CONSUME(9);                      // 10 11 12 (6)
        TG_SB(); PP_WAIT(4, fr[2]); PP_CONSUME(10);
        TG_SB(); PP_WAIT(2, fr[3]); PP_CONSUME(11);
        TG_SB(); PP_WAIT(0, fr[0]); PP_CONSUME(12);
        TG_SB();
        asm volatile("s_nop 11" : "+v"(s), "+v"(dp));      // an 8-pass MFMA result needs 12 wait states before VALU reads it (the barrier may be short)
    };
    auto xprefetch = [&](const int slot) {                  // fragment groups 0..2 of the next X: dS^T blocks of a tile whose writes completed an X ago
        const uint32_t vD = dsR + (uint32_t)(slot * PP_DS_SLOT), vB = 0, vA = 0, vS = 0;     // (vB, vA, vS: dead branches of the macro)
        PP_ISSUE(0); PP_ISSUE(1); PP_ISSUE(2);
    };
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    xprefetch(0);
    if (grp == 1) BWD_BAR();                                // group 1 falls one segment behind group 0
    int cb = 0, pb = RING - 1, wb = 2 + grp;                // current / previous tile buffer; buffer this thread stages next
    const int nit = ntile + PP_EXTRA;
    // One iteration = X(it) | barrier | Y(it) | barrier.  Every instruction of Y costs ~13 cycles of issue (the partner wave's MFMA stream has the priority), scalar ones
    // included: the first version's Y carried ~70 scalar instructions and 27 branches of range tests per tile (is the tile to exchange / request / check / signal / fetch
    // inside the range? is it the ragged last one?) and took ~1900 cycles beside an X of 1300.  FAST = the iterations in which every such test is known to be true and every
    // tile involved lies wholly inside the query range: no tests, unconditional cursor steps.  The head and tail iterations run the general body (same state, same order).
    const char* pcnt = (const char*)cntw - (long)lag * (CNT_PAD * 4);      // scalar cursor: counter line of tile (it - lag); +-k lines through the instruction offset
    auto body = [&](auto fast_c, const int it) {
        constexpr bool FAST = decltype(fast_c)::value;
        // ---------------- X(it): matrix segment ----------------
        {   // signal: the stores of tile it - 1 - lag went out in Y(it - 1), every wave of the group waited for their acknowledgement at the END of that same Y (the
            // vmcnt wait in front of its closing barrier), and that barrier lies behind us.  Round 5 signalled a tile later (acknowledged at the top of the NEXT Y):
            // bitwise the same gradients, the same 22.7 ms for the 17776^2 call, but the follower key block then re-read a dQ line one tile time later — and 26.3 GB
            // instead of 13.8 GB of the launch's 30.6 GB of dQ stores had left the XCD's L2 by then (profiles/NOTES.md F / G; r5 -> r6_attention_bwd_pmc.json)
            const int ts = it - 1 - lag;
            if ((tid & 255) == 0 && (FAST || (ts >= 0 && ts < ntile)))
                asm volatile("global_store_dword %0, %1, %2 offset:%3" :: "v"(0u), "v"(blk + 1), "s"(pcnt), "i"(-1 * CNT_PAD * 4) : "memory");
        }
        __builtin_amdgcn_s_setprio(2);
        xseg(cb, pb, it & 1);
        __builtin_amdgcn_s_setprio(0);
        BWD_BAR();
        // ---------------- Y(it): vector segment ----------------
        const int tw = it - lag;                            // tile this group exchanges now; tw + 1 is requested, tw + 2 checked, tw + 3 sampled
        const bool doW = FAST || (tw >= 0 && tw < ntile);
        f32x4 e0, e1;                                       // (written and read under doW only)
        if (doW) {
            const uint32_t a = dqR + (uint32_t)((tw & 1) * PP_DQ_SLOT);
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3" : "=&v"(e0), "=&v"(e1) : "v"(a), "i"(2 * PP_DQ_SLOT) : "memory");
        }
        // everything this wave left in flight in Y(it - 1) — the fetched tile, the requested dQ rows, the dQ store, the counter sample — has had a whole X to land
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(g0), "+v"(gseed), "+v"(ldv), "+v"(cval));
        // stage write of tile it + 2 + grp (not waited for: first read two barriers from now, behind this wave's own X(it + 1)), fetch of tile it + 3 + grp
        if (FAST) {
            *(u32x4*)(dstR + wb * ROWT_EL) = g0;
            if (wave == 0) *(u32x4*)(sSt + wb * 64 + j) = gseed;
            gld16(g0, sbR, voR);
            if (wave == 0) gld16(gseed, sbS, voS);
            sbR += stepRb; sbS += BT * 16;
            ++tf;
        } else {
            stash(wb);
            fetch();
        }
        wb = (wb + 1) & (RING - 1);
        if (doW) {
            asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(e0), "+v"(e1));       // (younger: the stage write; wave 0 also waits for its first of two)
            // (ldv stays zero in the workgroup that starts the sum: it never requests)
            const f32x4 v = f32x4{(e0[0] + e1[0]) * p.scale + ldv[0], (e0[1] + e1[1]) * p.scale + ldv[1], (e0[2] + e1[2]) * p.scale + ldv[2],
                                  (e0[3] + e1[3]) * p.scale + ldv[3]};
            if (FAST || tw < ntile - 1) gst16f(v, sbDQw, voDQ);
            else if (okDQLast) gst16f(v, sbDQw, voDQLast);
            if (FAST || tw < ntile - 1) sbDQw += stepDQb;
        }
        if (!first && (FAST || (tw + 1 >= 0 && tw + 1 < ntile))) {      // (tile tw + 1 was checked before the barrier that closed Y(it - 1))
            gld16f(ldv, sbDQr, (FAST || tw + 1 < ntile - 1) ? voDQ : voDQLast);
            if (FAST || tw + 1 < ntile - 1) sbDQr += stepDQb;
        }
        // P = exp2(S), dS = P o dP -> the A operands of X(it + 1); dS^T -> LDS for X(it + 2)
        {
            const uint32_t dsw = dsW + (uint32_t)((it & 1) * PP_DS_SLOT);
#define PP_SOFTMAX(g, SC)                                                                                                         \
            do {                                                                                                                  \
                uint2 w01;                                                                                                        \
                {                                                                                                                 \
                    const float p0 = fast_exp2(SC(s[4 * (g)])), p1 = fast_exp2(SC(s[4 * (g) + 1]));                               \
                    pA[(g) >> 1].w[((g) & 1) * 2] = pack_bf16x2_trans(p0, p1);                                                    \
                    w01.x = pack_bf16x2(p0 * dp[4 * (g)], p1 * dp[4 * (g) + 1]);                                                  \
                    dA[(g) >> 1].w[((g) & 1) * 2] = w01.x;                                                                        \
                }                                                                                                                 \
                {                                                                                                                 \
                    const float p0 = fast_exp2(SC(s[4 * (g) + 2])), p1 = fast_exp2(SC(s[4 * (g) + 3]));                           \
                    pA[(g) >> 1].w[((g) & 1) * 2 + 1] = pack_bf16x2_trans(p0, p1);                                                \
                    w01.y = pack_bf16x2(p0 * dp[4 * (g) + 2], p1 * dp[4 * (g) + 3]);                                              \
                    dA[(g) >> 1].w[((g) & 1) * 2 + 1] = w01.y;                                                                    \
                }                                                                                                                 \
                asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(dsw), "v"(w01), "i"(16 * (g)) : "memory");                    \
            } while (0)
#define PP_SC_UNIT(x) (x)
#define PP_SC_MUL(x) ((x) * p.scale_log2)
            if (unit) { PP_SOFTMAX(0, PP_SC_UNIT); PP_SOFTMAX(1, PP_SC_UNIT); PP_SOFTMAX(2, PP_SC_UNIT); PP_SOFTMAX(3, PP_SC_UNIT); }
            else { PP_SOFTMAX(0, PP_SC_MUL); PP_SOFTMAX(1, PP_SC_MUL); PP_SOFTMAX(2, PP_SC_MUL); PP_SOFTMAX(3, PP_SC_MUL); }
#undef PP_SC_UNIT
#undef PP_SC_MUL
#undef PP_SOFTMAX
        }
        if (wg == 0) {                                      // the group's first wave keeps its chain: tile tw + 2 must be complete before anybody requests it in Y(it + 1)
            const int tc = tw + 2;
            if (blk > 0 && (FAST || (tc >= 0 && tc < ntile)) && lane == 0 && cval != blk) {
                // bounded: a workgroup that never sees its turn goes on instead of hanging the GPU — and SAYS so: status[0] counts the polls that gave up, the dq of this launch
                // is then invalid and the host must not use it (kernels.attention_bwd_status); status[1] != 0 overrides the limit (tests force the path with 1)
                const int lim = fp.status[1] > 0 ? fp.status[1] : (1 << 20);
                int spin = 0;
                while (cnt_read(cntw + (long)tc * CNT_PAD) != blk && ++spin < lim) __builtin_amdgcn_s_sleep(2);
                if (spin >= lim) atomicAdd(fp.status, 1);
            }
            if (FAST) {
                asm volatile("global_load_dword %0, %1, %2 offset:%3 sc1" : "=v"(cval) : "v"(0u), "s"(pcnt), "i"(3 * CNT_PAD * 4) : "memory");
            } else {
                const int tn = min(max(tc + 1, 0), ntile - 1);
                asm volatile("global_load_dword %0, %1, off sc1" : "=v"(cval) : "v"(cntw + (long)tn * CNT_PAD) : "memory");
            }
        }
        pcnt += CNT_PAD * 4;
        pb = cb;
        cb = (cb + 1) & (RING - 1);
        TG_SB();
        xprefetch((it + 1) & 1);
        // the dQ stores of this Y acknowledged before the barrier, so that the head of the next X may signal them (the one load still in flight in the steady
        // state is the counter sample issued after them; loads and stores retire in order)
        if (FAST && !first) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BWD_BAR();
    };
    // FAST iterations: the signalled tile it - 5 (group 0) exists, and the fetched tile it + 4 (group 1) still lies wholly inside the query range
    const int itLo = min(5, nit), itHi = max(itLo, ntile - 5);
    int it = 0;
    for (; it < itLo; ++it) body(std::false_type{}, it);
    for (; it < itHi; ++it) body(std::true_type{}, it);
    for (; it < nit; ++it) body(std::false_type{}, it);
    // The last Y prefetched the fragments of an X that never runs: their LDS reads are still IN FLIGHT towards fr[0..2].  The wait must name those registers —
    // to the compiler the asm outputs were written when the reads were issued and are dead here, so it would hand the same VGPRs to the epilogue's
    // v_accvgpr_read of dK / dV at once, and the late returns would overwrite them (found on hardware: run-to-run differences in accumulator rows 0..2 of one wave).
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(fr[0][0].v), "+v"(fr[0][1].v), "+v"(fr[1][0].v), "+v"(fr[1][1].v), "+v"(fr[2][0].v), "+v"(fr[2][1].v), "+v"(g0), "+v"(gseed), "+v"(ldv), "+v"(cval)
                 :: "memory");
    if (grp == 0) BWD_BAR();                                // pairs with group 1's extra barrier
    // the last tiles' stores: acknowledged (vmcnt, above), seen by the whole group (barrier), then signalled (tile nit - 2 - lag was signalled by the last X already:
    // writing the same value again is harmless)
    __syncthreads();
    if ((tid & 255) == 0)
        for (int ts = max(nit - 2 - lag, 0); ts < ntile; ++ts) cnt_write(cntw + (long)ts * CNT_PAD, blk + 1);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        float* DK = p.dk + (long)b * p.dk_sb + h * HD + db * 32 + j;
        float* DV = p.dv + (long)b * p.dv_sb + h * HD + db * 32 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kw0 + acc_row(r, hi);
            if (key >= p.nk) continue;
            float* a = DK + (long)key * p.dk_ld;
            const float vk = dk[db][r] * p.scale;
            float vv = dv[db][r];
            *a = (p.accumulate & 2) ? *a + vk : vk;
            if (p.dv) {
                float* c = DV + (long)key * p.dv_ld;
                if (p.accumulate & 2) vv += *c;
                *c = vv;
            }
            if (p.dvb) p.dvb[(long)b * p.dvb_sb + (long)key * p.dvb_ld + h * HD + db * 32 + j] = f32_to_bf16(vv);
        }
    }
#undef PP_RD_T
#undef PP_RD_A
#undef PP_WAIT
#undef PP_ISSUE
#undef PP_CONSUME
}

// One-time probe of what the one-kernel backward relies on, on THIS device: (1) workgroup w of a 1-D launch runs on XCD w % 8; (2) the exchange protocol
// itself — a chain of 32 workgroups of XCD 0, each adding its number to 64 tiles of a buffer in chain order with exactly the kernel's primitives (L1
// invalidated once per workgroup; sc1 poll of a counter; plain 16-byte load; plain store; vmcnt(0); plain-store signal).  Any stale read, lost update or
// poll that does not end shows up as a wrong sum or a raised flag, and the launcher then keeps the two-kernel form.
__global__ __launch_bounds__(64) void fused_probe_kernel(int* bad, int* cnt, float* data) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;        // HW_REG_XCC_ID, bits 3:0
    // what the kernels need is that workgroups of ONE residue class mod 8 share an XCD (which XCD a class lands on may differ from launch to launch: the
    // dispatcher's round-robin does not always start at XCD 0): bad[0] counts workgroups that found another XCD's bit in their class's mask (the last 8
    // ints of the counter area: the chain's 64 tile counters sit at cnt[32 T], the last of them at 2016); identity (xcc == class) only goes on the
    // record, in bad[2]
    if (threadIdx.x == 0) {
        const int old = __hip_atomic_fetch_or(bad + 4 + 64 * 32 - 8 + (blockIdx.x & 7u), 1 << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old & ~(1 << xcc)) atomicAdd(bad, 1);
        if (xcc != (blockIdx.x & 7u)) atomicAdd(bad + 2, 1);
    }
    if ((blockIdx.x & 7u) != 0 || (blockIdx.x >> 3) >= 32u) return;
    const int kb = (int)(blockIdx.x >> 3), lane = (int)threadIdx.x;
    asm volatile("buffer_inv sc1" ::: "memory");
    for (int T = 0; T < 64; ++T) {
        if (kb > 0 && lane == 0) {
            int spin = 0;
            while (cnt_read(cnt + T * 32) != kb && ++spin < (1 << 20)) __builtin_amdgcn_s_sleep(2);
            if (spin >= (1 << 20)) atomicAdd(bad, 1);
        }
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        float* ptr = data + (T * 64 + lane) * 4;
        if (kb > 0) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(ptr) : "memory");
        const float add = (float)(kb + 1);
        *(f32x4*)ptr = f32x4{v[0] + add, v[1] + add, v[2] + add, v[3] + add};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) cnt_write(cnt + T * 32, kb + 1);
    }
}

}  // namespace

// statistics: 2 floats per query row (log-sum-exp unless the forward kept it, and D) + the 16-byte seed row of the dK/dV kernel
// key-axis split of the dQ launch: only for calls with so few query blocks that one workgroup per block leaves most of the chip idle
constexpr int DQ_SPLIT_MAX = 8;
static inline bool dq_split_shape(int nq, int heads, int batch) { return (long)((nq + 255) / 256) * heads * batch <= 256; }
static inline long dq_split_floats(int nq, int heads, int batch) { return dq_split_shape(nq, heads, batch) ? (long)DQ_SPLIT_MAX * batch * nq * heads * HD + 4 : 0; }

extern "C" long tg_attention_bwd_ws_floats(int nq, int nk, int heads, int batch) {
    (void)nk;
    return 6L * batch * heads * nq + 8 + (long)batch * heads * ((nq + BT - 1) / BT) * 32 + 8 + (long)batch * heads     // + the one-kernel form's dQ counters and per-head XCD masks
           + dq_split_floats(nq, heads, batch);                                                                           // + the dQ launch's key-range partials (few queries only)
}

// The device probe of the one-kernel form (fused_probe_kernel) as explicit entry points on CALLER-owned memory: nothing is allocated, freed or
// synchronised inside the ABI.  The caller zero-fills nothing (the call clears the buffer itself, asynchronously), copies the buffer to the host once the
// stream has passed the launch, and hands the copy to tg_attention_bwd_probe_verdict.
constexpr long PROBE_INTS = 4 + 64 * 32, PROBE_FLOATS = 64 * 64 * 4;
extern "C" long tg_attention_bwd_probe_bytes(void) { return (long)sizeof(int) * PROBE_INTS + (long)sizeof(float) * PROBE_FLOATS; }

extern "C" int tg_attention_bwd_probe(void* buf, long nbytes, hipStream_t stream) {
    TG_REQUIRE(buf && tg_aligned16(buf), TG_ERR_ARG, "tg_attention_bwd_probe: buf must be a 16-byte aligned device pointer");
    TG_REQUIRE(nbytes >= tg_attention_bwd_probe_bytes(), TG_ERR_SHAPE, "tg_attention_bwd_probe: buffer of %ld bytes, need %ld", nbytes, tg_attention_bwd_probe_bytes());
    const hipError_t e = hipMemsetAsync(buf, 0, (size_t)tg_attention_bwd_probe_bytes(), stream);
    if (e != hipSuccess) return tg_set_error(TG_ERR_HIP - (int)e, "tg_attention_bwd_probe: memset failed: %s", hipGetErrorString(e));
    int* bad = (int*)buf;
    int* cnt = bad + 4;
    float* data = (float*)(cnt + 64 * 32);
    hipLaunchKernelGGL(fused_probe_kernel, dim3(4096), dim3(64), 0, stream, bad, cnt, data);
    TG_LAUNCH_CHECK("tg_attention_bwd_probe");
    return TG_OK;
}

extern "C" int tg_attention_bwd_probe_verdict(const void* host_copy, long nbytes) {
    if (!host_copy || nbytes < tg_attention_bwd_probe_bytes()) return 0;
    if (*(const int*)host_copy != 0) return 0;                                          // a workgroup off its XCD, or a poll that did not end
    const float* hd = (const float*)((const char*)host_copy + sizeof(int) * PROBE_INTS);
    for (long i = 0; i < PROBE_FLOATS; ++i)
        if (hd[i] != 528.0f) return 0;                                                  // 1 + 2 + ... + 32, every element, exactly
    return 1;
}

extern "C" int tg_attention_bwd(const void* q, long q_ld, long q_sb, const void* k, long k_ld, long k_sb, const void* v, long v_ld, long v_sb,
                                 const void* o, long o_ld, long o_sb, const void* dout, long do_ld, long do_sb,
                                 float* dq, long dq_ld, long dq_sb, float* dk, long dk_ld, long dk_sb, float* dv, long dv_ld, long dv_sb,
                                 int nq, int nk, int heads, int batch, float scale, int accumulate, const float* lse, float* ws, hipStream_t stream) {
    return tg_attention_bwd_ex(q, q_ld, q_sb, k, k_ld, k_sb, v, v_ld, v_sb, o, o_ld, o_sb, dout, do_ld, do_sb, dq, dq_ld, dq_sb, dk, dk_ld, dk_sb, dv, dv_ld,
                               dv_sb, nq, nk, heads, batch, scale, accumulate, lse, ws, 0, nullptr, stream);
}

extern "C" int tg_attention_bwd_ex(const void* q, long q_ld, long q_sb, const void* k, long k_ld, long k_sb, const void* v, long v_ld, long v_sb,
                                    const void* o, long o_ld, long o_sb, const void* dout, long do_ld, long do_sb,
                                    float* dq, long dq_ld, long dq_sb, float* dk, long dk_ld, long dk_sb, float* dv, long dv_ld, long dv_sb,
                                    int nq, int nk, int heads, int batch, float scale, int accumulate, const float* lse, float* ws, int flags, int* status,
                                    hipStream_t stream) {
    const tg_attn_bwd_problem pr{q, q_ld, q_sb, k, k_ld, k_sb, v, v_ld, v_sb, o, o_ld, o_sb, dout, do_ld, do_sb, dq, dq_ld, dq_sb, dk, dk_ld, dk_sb, dv, dv_ld, dv_sb,
                                 nq, nk, scale, accumulate, lse, ws};
    return tg_attention_bwd_multi(&pr, 1, heads, batch, flags, status, stream);
}

namespace {
struct Prepared {
    Bwd2Params pp;
    dim3 gq, gk;
    bool one_kernel;
    int* cnt;
    long ncnt;
};
// validation + parameter block of one problem; decides the form (one kernel / two launches) exactly as documented for tg_attention_bwd_ex
int bwd_prepare(const tg_attn_bwd_problem& a, int heads, int batch, int flags, Prepared& out) {
    const int accumulate = a.accumulate == 1 ? 3 : (a.accumulate & 3);      // bit 0: dq, bit 1: dk and dv; 1 = all three (the original meaning of the flag)
    TG_REQUIRE(a.q && a.k && a.v && a.o && a.dout && a.dq && a.dk && (a.dv || a.dv_bf16) && a.ws, TG_ERR_ARG, "tg_attention_bwd: null pointer");
    TG_REQUIRE(a.dv || !(accumulate & 2), TG_ERR_ARG, "tg_attention_bwd: accumulate into dv needs the fp32 dv");
    TG_REQUIRE(!a.dv_bf16 || (a.dv_bf16_ld >= (long)heads * HD && a.dv_bf16_sb >= 0), TG_ERR_SHAPE, "tg_attention_bwd: dv_bf16 row stride %ld below heads * 64", a.dv_bf16_ld);
    TG_REQUIRE(a.nq > 0 && a.nk > 0 && heads > 0 && batch > 0, TG_ERR_SHAPE, "tg_attention_bwd: bad shape nq=%d nk=%d heads=%d batch=%d", a.nq, a.nk, heads, batch);
    TG_REQUIRE(tg_aligned16(a.q) && tg_aligned16(a.k) && tg_aligned16(a.v) && tg_aligned16(a.o) && tg_aligned16(a.dout) && tg_aligned16(a.ws) && a.q_ld % 8 == 0 &&
               a.k_ld % 8 == 0 && a.v_ld % 8 == 0 && a.o_ld % 8 == 0 && a.do_ld % 8 == 0 && a.q_sb % 8 == 0 && a.k_sb % 8 == 0 && a.v_sb % 8 == 0 && a.o_sb % 8 == 0 &&
               a.do_sb % 8 == 0, TG_ERR_ALIGN, "tg_attention_bwd: q/k/v/o/dO need 16-byte aligned rows");
    const int nq = a.nq, nk = a.nk;
    const long nrow = (long)batch * heads * nq;               // workspace: seed rows (16 B each, first: alignment) | log-sum-exp | D
    out.pp.p = BwdParams{(const bf16_t*)a.q, (const bf16_t*)a.k, (const bf16_t*)a.v, (const bf16_t*)a.o, (const bf16_t*)a.dout, a.q_ld, a.q_sb, a.k_ld, a.k_sb, a.v_ld,
                         a.v_sb, a.o_ld, a.o_sb, a.do_ld, a.do_sb, a.dq, a.dk, a.dv, a.dq_ld, a.dq_sb, a.dk_ld, a.dk_sb, a.dv_ld, a.dv_sb, a.ws + 4 * nrow, a.ws + 5 * nrow,
                         (uint4*)a.ws, nq, nk, heads, batch, a.scale * 1.4426950408889634f, a.scale, accumulate, a.lse ? 1 : 0, (bf16_t*)a.dv_bf16, a.dv_bf16_ld, a.dv_bf16_sb};
    if (a.lse) out.pp.p.lse = const_cast<float*>(a.lse);
    out.pp.kparts = 1;
    out.pp.dq_part = nullptr;
    if (fabsf(out.pp.p.scale_log2 - 1.0f) < 4e-7f) out.pp.p.scale_log2 = 1.0f;      // ln 2 * log2 e: exactly one
    out.gq = dim3((unsigned)((nq + 255) / 256), (unsigned)(batch * heads));
    out.gk = dim3((unsigned)((nk + 255) / 256), (unsigned)(batch * heads));
    // the ordered dQ accumulation runs the key blocks of a head as a chain a few tiles apart: worth it only when there are many more query tiles than
    // key blocks (the 17776^2 call: 556 tiles, 70 blocks; the vip queries' call with 15 tiles and 72 blocks would serialise)
    const bool chain_ok = (long)((nq + BT - 1) / BT) >= 4L * out.gk.x && a.dq_ld % 4 == 0 && a.dq_sb % 4 == 0 && tg_aligned16(a.dq);
    out.one_kernel = (flags & TG_BWD_ONE_KERNEL) && chain_ok && ((heads * batch) & 7) == 0;
    out.cnt = (int*)(a.ws + ((6 * nrow + 8 + 3) & ~3L));
    out.ncnt = (long)batch * heads * ((nq + BT - 1) / BT) * 32;       // the exchange counters; the per-head XCD masks sit behind them
    return TG_OK;
}
}  // namespace

extern "C" int tg_attention_bwd_multi(const tg_attn_bwd_problem* problems, int count, int heads, int batch, int flags, int* status, hipStream_t stream) {
    TG_REQUIRE(problems && count >= 1 && count <= 2, TG_ERR_ARG, "tg_attention_bwd_multi: one or two problems");
    TG_REQUIRE(!(flags & TG_BWD_ONE_KERNEL) || status, TG_ERR_ARG, "tg_attention_bwd_ex: TG_BWD_ONE_KERNEL needs the status words");
    Prepared P[2];
    for (int i = 0; i < count; ++i) {
        const int rc = bwd_prepare(problems[i], heads, batch, flags, P[i]);
        if (rc) return rc;
    }
    for (int i = 0; i < count; ++i) hipLaunchKernelGGL(attn_bwd_stats2_kernel, P[i].gq, dim3(256), 0, stream, P[i].pp.p);
    // one-kernel problems: the first is the main call, a second one rides in the same launch (its workgroups behind the main call's: they fill the last round)
    int main_i = -1, rider_i = -1;
    for (int i = 0; i < count; ++i)
        if (P[i].one_kernel) { if (main_i < 0) main_i = i; else rider_i = i; }
    if (main_i >= 0) {
        FusedParams fp{};
        fp.p = P[main_i].pp.p; fp.cnt = P[main_i].cnt; fp.status = status; fp.xmask = P[main_i].cnt + P[main_i].ncnt;
        fp.main_wgs = (int)(P[main_i].gk.x * P[main_i].gk.y);
        unsigned grid = (unsigned)fp.main_wgs;
        hipError_t e = hipMemsetAsync(fp.cnt, 0, (size_t)(P[main_i].ncnt + (long)batch * heads) * sizeof(int), stream);
        if (e == hipSuccess && rider_i >= 0) {
            fp.r = P[rider_i].pp.p; fp.r_cnt = P[rider_i].cnt; fp.r_xmask = P[rider_i].cnt + P[rider_i].ncnt;
            grid += P[rider_i].gk.x * P[rider_i].gk.y;
            e = hipMemsetAsync(fp.r_cnt, 0, (size_t)(P[rider_i].ncnt + (long)batch * heads) * sizeof(int), stream);
        }
        if (e != hipSuccess) return tg_set_error(TG_ERR_HIP - (int)e, "tg_attention_bwd_ex: %s", hipGetErrorString(e));
        TG_DYN_LDS(attn_bwd_fused_pp_kernel, PP_LDS);
        hipLaunchKernelGGL(attn_bwd_fused_pp_kernel, dim3(grid), dim3(512), PP_LDS, stream, fp);
        TG_LAUNCH_CHECK("tg_attention_bwd(one kernel)");
    }
    for (int i = 0; i < count; ++i) {
        if (P[i].one_kernel) continue;
        hipLaunchKernelGGL(attn_bwd_dkdv7_kernel, dim3(P[i].gk.x * P[i].gk.y), dim3(512), 0, stream, P[i].pp);
        // dQ: one workgroup per 256 queries and head — or, with few queries, per (256 queries, head, key range): ~3 rounds of the chip's resident slots (2 per CU)
        const BwdParams& bp = P[i].pp.p;
        const long wgs = (long)P[i].gq.x * P[i].gq.y, ntile = (bp.nk + BT - 1) / BT;
        int kparts = 1;
        if (dq_split_shape(bp.nq, heads, batch) && bp.dq_ld % 4 == 0 && bp.dq_sb % 4 == 0 && tg_aligned16(bp.dq)) {
            kparts = (int)((6L * tg_device_cus()) / wgs);
            if (kparts > DQ_SPLIT_MAX) kparts = DQ_SPLIT_MAX;
            if (kparts > ntile / 16) kparts = (int)(ntile / 16);               // at least 16 key tiles per range
            if (kparts < 2) kparts = 1;
        }
        P[i].pp.kparts = kparts;
        P[i].pp.dq_part = (float*)(((uintptr_t)(P[i].cnt + P[i].ncnt + (long)batch * heads) + 15) & ~(uintptr_t)15);
        hipLaunchKernelGGL(attn_bwd_dq2_kernel, dim3((unsigned)(wgs * kparts)), dim3(256), 0, stream, P[i].pp);
        if (kparts > 1) {
            const long n4 = (long)batch * bp.nq * heads * HD / 4;
            hipLaunchKernelGGL(attn_bwd_dq_join_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, P[i].pp);
        }
        TG_LAUNCH_CHECK("tg_attention_bwd");
    }
    return TG_OK;
}
