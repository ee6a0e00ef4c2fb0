// Kernels of the 3-D causal VAE (AutoencoderKLCogVideoX; reference twin longvgen/models/autoencoder_kl_cogvideox.py).
//
// Layout: activations are channels-last bf16  x[t][h][w][c].  A convolution is then a GEMM whose A rows are output
// voxels and whose K index is (tap, channel): for a fixed tap the K slice of a row is a CONTIGUOUS run of channels of one
// input voxel, so the LDS-DMA loader of the GEMM (global_load_lds_dwordx4, 16 B per lane) applies unchanged — only the
// per-lane source address is computed from (to, ho, wo, tap).  Consequences:
//   * no cat(conv_cache, x): frames before the window are read straight from the cache tensor (or frame 0, replicated);
//   * no F.pad: out-of-range taps read a zero page;
//   * no materialised F.interpolate: the nearest-neighbour x2 upsample (spatial, and temporal through t_map) is index math.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include "common.h"
#include "tokensgen_hip.h"

namespace {

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int GN_GROUPS = 32;
constexpr int TILE_BYTES = BM * BK * 2;
constexpr int STAGE_BYTES = 2 * TILE_BYTES;

struct ConvParams {
    const bf16_t* x; int T, H, W, Cin;
    const bf16_t* cache;
    const bf16_t* w; const bf16_t* bias;
    int cout, cout_pad, kt, kh, kw, stride, pad, up;
    const int32_t* t_map;
    const bf16_t* residual;
    bf16_t* y; long ldy;
    int To, Ho, Wo;
    const bf16_t* zeros;
    float* gn_partial;    // optional [tiles_m][2][32]: per-tile GroupNorm(32) sums / sums of squares of the STORED bf16 values
    int ksplit;           // > 1: the (tap, channel) reduction is cut into ksplit ranges, one workgroup each; raw fp32 sums go to kpart
    float* kpart;         // [ksplit][M][cout_pad] fp32 (conv_splitk_reduce_kernel adds them in a fixed order and runs the epilogue)
    // ---- phase launches of tg_conv3d_up2_subpixel (conv3d_w4_kernel only; all zero = an ordinary convolution) ----
    int pad_w_off;        // the W axis pads with pad + pad_w_off (the H axis with pad): the 2x2 phase kernels pad on one side only
    int o_up;             // 2: output voxel (to, ho, wo) of this launch is stored at (to, 2 ho + o_py, 2 wo + o_px) of a [To][2 Ho][2 Wo] tensor
    int o_py, o_px;
    int o_tdup;           // with o_up == 2: 0 = frame t -> frame t; 1 = the nearest x2 of an ODD frame count (frame 0 -> 0; t >= 1 -> 2t - 1 and 2t); 2 = of an even one (t -> 2t, 2t + 1)
    int o_phases;         // 4: ONE launch runs all four phases — virtual tile v = phase * tiles + tile; the phase sets pad / pad_w_off / o_py / o_px, its weights follow
                          // each other in `w` ([4][cout][4][Cin]) and its GroupNorm rows in gn_partial ([4][ceil(M / 128)][64])
};

// WN x (4 / WN) waves; a wave owns FM x FN MFMA 16x16 blocks: <2, 4, 4> = the 128 x 128 tile, <1, 2, 1> = 128 voxels x 16 output channels
// (conv_out: Cout = 3 / 32 — with N padded to 128 the launch spent 42x / 4x the MFMA work and 8x the weight traffic it needed)
template <int WN, int FM, int FN>
__global__ __launch_bounds__(256) void conv3d_cl_kernel(ConvParams p) {
    constexpr int WM = 4 / WN;
    constexpr int BN = WN * FN * 16;
    static_assert(WM * FM * 16 == BM, "the tile is 128 voxels");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const long M = (long)p.To * p.Ho * p.Wo;
    const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = p.cout_pad / BN;
    const int nwg = tiles_m * tiles_n;
    const int split = __builtin_amdgcn_readfirstlane((int)blockIdx.x / nwg);      // 0 unless ksplit > 1
    const int t = xcd_remap((int)blockIdx.x - split * nwg, nwg);
    const int tn = t % tiles_n, tm = t / tiles_n;     // n fastest: the tiles_n blocks sharing an A tile run on one XCD
    const long m0 = (long)tm * BM;
    const int n0 = tn * BN;
    const int Kw = p.kt * p.kh * p.kw * p.Cin;        // row length of the packed weights

    // ---- per-lane rows: wave-instruction i covers tile rows [wave*32 + i*8, +8) ----
    int vt[4], vh[4], vw[4], slotA[4];
    const char* srcW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wave * 32 + i * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ ((r >> 1) & 7);
        slotA[i] = slot * 8;
        long m = m0 + r;
        if (m >= M) m = M - 1;
        const int wo = (int)(m % p.Wo);
        const long q = m / p.Wo;
        vw[i] = wo; vh[i] = (int)(q % p.Ho); vt[i] = (int)(q / p.Ho);
        srcW[i] = (const char*)(p.w + (long)(n0 + (r < BN ? r : 0)) * Kw + slot * 8);      // (advanced to the first k-tile of this workgroup's range below)
    }
    const int Hv = p.H * p.up, Wv = p.W * p.up;
    const long frame = (long)p.H * p.W * p.Cin;

    // source address of row i for tap (dt, dh, dw); returns the zero page for padded taps
    auto tap_src = [&](int i, int dt, int dh, int dw) -> const char* {
        const int hv = vh[i] * p.stride + dh - p.pad, wv = vw[i] * p.stride + dw - p.pad;
        if (hv < 0 || hv >= Hv || wv < 0 || wv >= Wv) return (const char*)(p.zeros + slotA[i]);
        int tv = vt[i] + dt - (p.kt - 1);
        const bf16_t* base = p.x;
        if (tv < 0) {
            if (p.cache) { base = p.cache; tv += p.kt - 1; } else tv = 0;   // cached frames, or replicate the first frame
        } else if (p.t_map) tv = p.t_map[tv];
        const int h = (p.up == 2) ? (hv >> 1) : hv, w = (p.up == 2) ? (wv >> 1) : wv;
        return (const char*)(base + (long)tv * frame + ((long)h * p.W + w) * p.Cin + slotA[i]);
    };

    const int cpt = p.Cin / BK;                      // k-tiles per tap
    const int ntaps = p.kt * p.kh * p.kw;
    const int nk = ntaps * cpt;
    const char* srcA[4];
    const int kt_begin = (int)((long)nk * split / p.ksplit), kt_end = (int)((long)nk * (split + 1) / p.ksplit);
    // Staging cursor: stages are issued in order kt_begin, kt_begin + 1, ..., so (tap -> dt, dh, dw; channel block cc) advance by increments.  The
    // first version divided per stage (kt / cpt, and tap % kw, / kw, % kh, / kh at every tap change): runtime integer divisions on the scalar unit,
    // which the halo kernel's stamps priced at hundreds of cycles per stage — against 512 cycles of MFMA per stage and wave here.
    int s_cc, s_dw, s_dh, s_dt;
    {
        const int tap0 = kt_begin / cpt;
        s_cc = kt_begin - tap0 * cpt;
        s_dw = tap0 % p.kw;
        const int q = tap0 / p.kw;
        s_dh = q % p.kh; s_dt = q / p.kh;
    }
    bool s_first = true;
    auto stage = [&](int buf) {                      // stages the cursor's k-tile into buffer buf and advances the cursor
        if (s_cc == 0 || s_first) {
#pragma unroll
            for (int i = 0; i < 4; ++i) srcA[i] = tap_src(i, s_dt, s_dh, s_dw);
            s_first = false;
        }
        char* base = smem + buf * STAGE_BYTES + wave * (32 * 128);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + s_cc * (BK * 2)),
                                             (__attribute__((address_space(3))) void*)(base + i * 1024), 16, 0, 0);
            if (wave * 32 + i * 8 < BN) {     // wave-uniform: the W tile has BN rows
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcW[i],
                                                 (__attribute__((address_space(3))) void*)(base + TILE_BYTES + i * 1024), 16, 0, 0);
                srcW[i] += BK * 2;
            }
        }
        if (++s_cc == cpt) {
            s_cc = 0;
            if (++s_dw == p.kw) { s_dw = 0; if (++s_dh == p.kh) { s_dh = 0; ++s_dt; } }
        }
    };

    int offA[FM][2], offW[FN][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int sl = ks * 4 + (lane >> 4);
#pragma unroll
        for (int f = 0; f < FM; ++f) {
            const int ra = wm * (FM * 16) + f * 16 + (lane & 15);
            offA[f][ks] = ra * 128 + ((sl ^ ((ra >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int f = 0; f < FN; ++f) {
            const int rw = wn * (FN * 16) + f * 16 + (lane & 15);
            offW[f][ks] = rw * 128 + ((sl ^ ((rw >> 1) & 7)) << 4);
        }
    }

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int i = 0; i < 4; ++i) srcW[i] += (long)kt_begin * (BK * 2);
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt_ = kt_begin; kt_ < kt_end; ++kt_) {
        const int cur = (kt_ - kt_begin) & 1;
        if (kt_ + 1 < kt_end) stage(cur ^ 1);
        const char* tA = smem + cur * STAGE_BYTES;
        const char* tW = tA + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[FM], fw[FN];
#pragma unroll
            for (int f = 0; f < FM; ++f) fa[f] = *(const bf16x8*)(tA + offA[f][ks]);
#pragma unroll
            for (int f = 0; f < FN; ++f) fw[f] = *(const bf16x8*)(tW + offW[f][ks]);
#pragma unroll
            for (int ni = 0; ni < FN; ++ni)
#pragma unroll
                for (int mi = 0; mi < FM; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ni], fa[mi], acc[ni][mi], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if (p.ksplit > 1) {      // raw fp32 sums of this K range; bias / residual / GroupNorm sums happen in conv_splitk_reduce_kernel
        float* part = p.kpart + (long)split * M * p.cout_pad;
#pragma unroll
        for (int mi = 0; mi < FM; ++mi) {
            const long m = m0 + wm * (FM * 16) + mi * 16 + (lane & 15);
            if (m >= M) continue;
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) {
                const int n = n0 + wn * (FN * 16) + ni * 16 + (lane >> 4) * 4;
                *(f32x4*)(part + m * p.cout_pad + n) = acc[ni][mi];
            }
        }
        return;
    }

    // ---- epilogue: bias (+ residual), only columns < cout are stored ----
    // All bias / residual chunks of the lane are requested FIRST (one batch of loads in flight), then converted and stored: written the
    // naive way (load, use, store per chunk) the stores order the next chunk's loads behind them (they may alias) and every chunk pays
    // a full memory round trip behind a vmcnt(0).
    float gs[FN], gq[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) gs[i] = gq[i] = 0.f;   // GroupNorm partials of this lane's channel quad per ni
    uint2 bq[FN], rq[FM][FN];
#pragma unroll
    for (int ni = 0; ni < FN; ++ni) {
        const int n = n0 + wn * (FN * 16) + ni * 16 + (lane >> 4) * 4;
        bq[ni] = (p.bias && n + 4 <= p.cout) ? *(const uint2*)(p.bias + n) : uint2{0u, 0u};
    }
#pragma unroll
    for (int mi = 0; mi < FM; ++mi) {
        const long m = m0 + wm * (FM * 16) + mi * 16 + (lane & 15);
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) {
            const int n = n0 + wn * (FN * 16) + ni * 16 + (lane >> 4) * 4;
            rq[mi][ni] = (p.residual && m < M && n + 4 <= p.cout) ? *(const uint2*)(p.residual + m * p.ldy + n) : uint2{0u, 0u};
        }
    }
#pragma unroll
    for (int mi = 0; mi < FM; ++mi) {
        const long m = m0 + wm * (FM * 16) + mi * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) {
            const int n = n0 + wn * (FN * 16) + ni * 16 + (lane >> 4) * 4;
            if (n >= p.cout) continue;
            float v[4] = {acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]};
            bf16_t* dst = p.y + m * p.ldy + n;
            const bf16_t* res = p.residual ? p.residual + m * p.ldy + n : nullptr;
            if (n + 4 <= p.cout) {
                if (p.bias) {
                    const uint2 bb = bq[ni];
                    v[0] += bf16lo_to_f32(bb.x); v[1] += bf16hi_to_f32(bb.x);
                    v[2] += bf16lo_to_f32(bb.y); v[3] += bf16hi_to_f32(bb.y);
                }
                if (res) {   // conv output is a bf16 tensor in the reference before `hidden_states + inputs`
                    const uint2 rr = rq[mi][ni];
                    v[0] = round_bf16(v[0]) + bf16lo_to_f32(rr.x); v[1] = round_bf16(v[1]) + bf16hi_to_f32(rr.x);
                    v[2] = round_bf16(v[2]) + bf16lo_to_f32(rr.y); v[3] = round_bf16(v[3]) + bf16hi_to_f32(rr.y);
                }
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *(uint2*)dst = o;
                if (p.gn_partial) {      // statistics of what the next GroupNorm will read: the bf16 values just stored
                    const float r0 = bf16lo_to_f32(o.x), r1 = bf16hi_to_f32(o.x), r2 = bf16lo_to_f32(o.y), r3 = bf16hi_to_f32(o.y);
                    gs[ni] += (r0 + r1) + (r2 + r3);
                    gq[ni] += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
                }
            } else {
                for (int e = 0; e < 4 && n + e < p.cout; ++e) {
                    float u = v[e] + (p.bias ? bf16_to_f32(p.bias[n + e]) : 0.f);
                    if (res) u = round_bf16(u) + bf16_to_f32(res[e]);
                    dst[e] = f32_to_bf16(u);
                }
            }
        }
    }
    // ---- GroupNorm partial sums of this tile, summed in a fixed order (no atomics): lanes of a 16-lane row group -> per
    // (wave, ni, channel quad) in LDS -> one thread per (statistic, group) walks the quads of its group and both row halves
    if constexpr (BN == 128) if (p.gn_partial) {
#pragma unroll
        for (int ni = 0; ni < FN; ++ni) {
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                gs[ni] += __shfl_xor(gs[ni], off, 64);
                gq[ni] += __shfl_xor(gq[ni], off, 64);
            }
        }
        float* red = (float*)smem;                          // the stage buffers are idle: the k loop ended with a __syncthreads
        if ((lane & 15) == 0) {
#pragma unroll
            for (int ni = 0; ni < FN; ++ni) {
                red[((wave * 4 + ni) * 4 + (lane >> 4)) * 2 + 0] = gs[ni];
                red[((wave * 4 + ni) * 4 + (lane >> 4)) * 2 + 1] = gq[ni];
            }
        }
        __syncthreads();
        const int cg = p.cout / GN_GROUPS, qpg = cg >> 2, gpt = BN / cg;     // channels per group, quads per group, groups per tile
        if (tid < 2 * gpt) {
            const int stat = tid / gpt, gl = tid % gpt;
            float a = 0.f;
            for (int qd = 0; qd < qpg; ++qd) {
                const int cq = gl * qpg + qd;                // tile-local channel quad 0..31 = wn*16 + ni*4 + (lane>>4)
                const int wn_ = cq >> 4, ni_ = (cq >> 2) & 3, quad_ = cq & 3;
                for (int wm_ = 0; wm_ < 2; ++wm_) a += red[(((wm_ * 2 + wn_) * 4 + ni_) * 4 + quad_) * 2 + stat];
            }
            p.gn_partial[(long)tm * 2 * GN_GROUPS + stat * GN_GROUPS + n0 / cg + gl] = a;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Halo-tiled convolution for the Cout = 128 layers.  The GEMM-shaped kernels above fetch a fresh [voxels x 64 channels] A tile from L2 for
// EVERY (tap, channel chunk); here a workgroup owns a 16 x 32 patch of ONE output frame, and for each (temporal tap dt, 32-channel chunk) the
// 18 x 34 input halo is brought into LDS ONCE and all nine (dh, dw) taps read it shifted: A traffic / 9, and a tap change costs no VALU at all
// (the source frame of a temporal tap is uniform per workgroup: x, the cache tensor, or frame 0 replicated).
// 128 voxels x 128 channels PER WAVE (4 waves x 4 patch rows), stages of 32 input channels (one MFMA 16x16x32 k-step): 16 KB of fragment reads
// per wave per 1024 MFMA cycles = 49 % of the LDS pipe at the full matrix rate (a first version with 64-voxel waves needed 73 %), and the reads
// of stage s+1 are issued BETWEEN the MFMAs of stage s into a second fragment set, so the matrix pipe never waits for LDS.  Weights ride a ring of
// four stage buffers (the DMA of stage s+3 is issued at the top of stage s; stage s+1 is resident while stage s computes, so its fragments can
// be prefetched); the 18 x 34 halo of the next (temporal tap, 32-channel chunk) group arrives in slices during stages 0..6 of the current group.
// Voxel rows (64 B of channels) and weight rows sit at an 80-byte stride: 16 consecutive rows x one 16-byte slot = 64 distinct banks, and the
// address stays linear in the voxel index, so a tap is still a uniform offset (the immediate offset field of the read).
// ------------------------------------------------------------------------------------------------
constexpr int H2_PH = 16, H2_PW = 32, H2_LW = H2_PW + 2, H2_ROWS = (H2_PH + 2) * H2_LW;      // 612 halo voxels
constexpr int H2_STRIDE = 80;
constexpr int H2_HALO_PIECES = (H2_ROWS * H2_STRIDE + 1023) / 1024;                            // 48
constexpr int H2_HALO_BYTES = H2_HALO_PIECES * 1024;                                           // 49152
constexpr int H2_W_BYTES = 128 * 64;                  // ring slot: 128 output channels x 32 k, 64-byte rows (16-byte slots XOR-swizzled by (row >> 2) & 3): 8 pieces, 2 per wave
constexpr int H2_RING = 4;                            // weight stages in flight: the DMA of stage s+3 is issued at the top of stage s
constexpr int H2_LDS = 2 * H2_HALO_BYTES + H2_RING * H2_W_BYTES;                               // 147456


// NW waves per workgroup: 4 (one per SIMD, 128 voxels x 128 channels each) or 8 (two per SIMD, 64 voxels x 128 channels each: what one wave cannot
// overlap with its own MFMAs — the blocked issue of its LDS-DMA pieces, its fragment-read waits — is covered by the other wave of the SIMD; twice the
// weight-fragment reads, 37 % of the LDS pipe at the full matrix rate).  Output channels beyond 128 are further 128-channel SLABS of the same patch:
// virtual workgroup v = patch * nslab + slab, consecutive on one XCD, so the slabs of a patch find its halo in that XCD's L2.
template <int NW>
__global__ __launch_bounds__(64 * NW) void conv3d_halo2_kernel(ConvParams p) {
    constexpr int HP = 48 / NW;                           // halo pieces per wave and (temporal tap, 32-channel chunk) group
    constexpr int WP = 8 / NW;                            // weight pieces per wave and stage
    constexpr int RW = H2_PH / NW;                        // patch rows per wave
    constexpr int MI = 2 * RW;                            // 16-voxel A fragments per wave
    constexpr int ASTEP = 8 / MI;                         // an A fragment of the next stage is read every ASTEP-th block
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sH = smem;
    char* const sW = smem + 2 * H2_HALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (p.Wo + H2_PW - 1) / H2_PW, tiles_y = (p.Ho + H2_PH - 1) / H2_PH;
    const int npatch = p.To * tiles_y * tiles_x;
    const int nslab = p.cout / 128;
    const int vwg = xcd_remap(blockIdx.x, npatch * nslab);
    const int tile = vwg / nslab, slab = vwg - tile * nslab;
    const int ntiles = npatch;
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, tf = tile / (tiles_x * tiles_y);
    const int x0 = tx * H2_PW, y0 = ty * H2_PH;
    const int Kw = p.kt * 9 * p.Cin;
    const int nc32 = p.Cin / 32;
    const long frame = (long)p.H * p.W * p.Cin;
    const int ngroups = p.kt * nc32, nst = ngroups * 9;

    // ---- DMA maps: lane -> (row, slot) of a 1 KiB piece under the 80-byte row stride; slot 4 is the pad ----
    int hoff[HP], woffs[WP];
#pragma unroll
    for (int i = 0; i < HP; ++i) {
        const int o = (wave * HP + i) * 1024 + lane * 16;
        const int hr = o / H2_STRIDE, slot = (o - hr * H2_STRIDE) >> 4;
        const int hy = hr / H2_LW, hx = hr - hy * H2_LW;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        hoff[i] = (slot < 4 && hr < H2_ROWS && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) ? (y * p.W + x) * p.Cin + slot * 8 : -1;
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {                         // weight pieces WP * wave + i: 16 rows x 64 B; the lane's physical slot holds logical slot ^ swizzle
        const int row = (WP * wave + i) * 16 + (lane >> 2);
        woffs[i] = (slab * 128 + row) * Kw + (((lane & 3) ^ ((row >> 2) & 3)) * 8);
    }
    auto frame_base = [&](int dt) -> const bf16_t* {
        const int tv = tf + dt - (p.kt - 1);
        if (tv >= 0) return p.x + (long)tv * frame;
        return p.cache ? p.cache + (long)(tv + p.kt - 1) * frame : p.x;
    };
    // The DMA sources are RUNNING values, advanced by wave-uniform adds: in-kernel stamps (profiles/NOTES.md, round 4) showed the two DMA instructions of
    // a stage taking 460-830 cycles per wave — not the DMA, but its address arithmetic: stage -> (group, tap) -> (dt, c32) through two integer divisions by a
    // runtime value, a 64-bit multiply and the frame_base() branches, all on the scalar unit beside the partner wave's MFMA stream.
    //   hsrc: source of the halo of the group being staged = frame_base(dt) + 32 c32 (one pointer per group, set when the group before it starts)
    //   wk:   element offset (dt * 9 + tap) * Cin + 32 c32 of the next weight stage to issue; stages follow each other as tap 0..8 within a group, groups as
    //         c32 = 0..nc32-1 within a temporal tap: +Cin per tap; at a group change +32 - 8 Cin, which is +32 when c32 wraps as well (32 nc32 = Cin)
    auto dma_halo = [&](int buf, const bf16_t* hsrc, int i) {      // piece i (0..HP-1) of this wave
        const bf16_t* src = hoff[i] >= 0 ? hsrc + hoff[i] : p.zeros + (lane & 7) * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sH + buf * H2_HALO_BYTES + (wave * HP + i) * 1024), 16, 0, 0);
    };
    auto dma_w = [&](int slot3, int koff, int i) {       // weight piece i (< WP) of this wave for the stage at element offset koff into ring slot slot3
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.w + woffs[i] + koff),
                                         (__attribute__((address_space(3))) void*)(sW + slot3 * H2_W_BYTES + (WP * wave + i) * 1024), 16, 0, 0);
    };
    const uint32_t ldsH = (uint32_t)(uintptr_t)sH, ldsW = (uint32_t)(uintptr_t)sW;
    uint32_t aoff[MI], woff[8];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
        aoff[mi] = (uint32_t)(((RW * wave + (mi >> 1)) * H2_LW + (mi & 1) * 16 + (lane & 15)) * H2_STRIDE + (lane >> 4) * 16);
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) {
        const int rw = ni * 16 + (lane & 15);
        woff[ni] = (uint32_t)(rw * 64 + (((lane >> 4) ^ ((rw >> 2) & 3)) << 4));
    }

    f32x4 acc[8][MI];
#pragma unroll
    for (int ni = 0; ni < 8; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Fragment registers: the voxel (A) fragments are needed by every MFMA block of a stage -> two sets by stage parity, the next set filled
    // during the current stage; the weight (W) fragments are needed one block (8 MFMAs) at a time -> a ring of three, W fragment q = stage * 8 + ni
    // in slot q % 3, read two blocks ahead (the full double set of both spilled: 512 VGPRs + 81 to scratch, 341 vs 182 ms)
    constexpr int WD = 5;                                  // W fragments are read WD blocks (8 MFMAs = 128 cycles each) ahead of their use; ring of WD + 1
    bf16x8 fa[2][MI], fwr[WD + 1];
    // The LDS byte address of a fragment = a per-lane base (runtime, 8 + 8 registers) + a COMPILE-TIME immediate (halo buffer, tap offset /
    // weight ring slot) in the instruction's 16-bit offset field: no address arithmetic per read — and nothing for the optimizer to hoist
    // (with the addresses computed in C++ it pre-computed all 144 (stage, fragment) addresses of the unrolled loop into VGPRs and spilled)
    uint32_t abase[MI], wbase[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) wbase[i] = ldsW + woff[i];
#pragma unroll
    for (int i = 0; i < MI; ++i) abase[i] = ldsH + aoff[i];
    auto read_a = [&](auto setc, auto immc, int mi) {       // immc: (group parity) * H2_HALO_BYTES + ((tap / 3) * H2_LW + tap % 3) * H2_STRIDE
        constexpr int set = decltype(setc)::value, imm = decltype(immc)::value;
        static_assert(imm >= 0 && imm < 65536, "ds_read offset field");
        bf16x8& dst = fa[set][mi];
        const uint32_t a = abase[mi];
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(a), "n"(imm));
    };
    auto read_w = [&](auto slotc, auto immc, int ni) {      // slotc: register ring slot; immc: LDS ring slot * H2_W_BYTES
        constexpr int rs = decltype(slotc)::value, imm = decltype(immc)::value;
        bf16x8& dst = fwr[rs];
        const uint32_t a = wbase[ni];
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(a), "n"(imm));
    };
#define H2_AIMM(kk) std::integral_constant<int, (((kk) / 9) & 1) * H2_HALO_BYTES + ((((kk) % 9) / 3) * H2_LW + ((kk) % 9) % 3) * H2_STRIDE>{}
#define H2_WIMM(kk) std::integral_constant<int, ((kk) % H2_RING) * H2_W_BYTES>{}

    // ---- prologue: halo of group 0, weights of stages 0 and 1; A fragments of stage 0, W fragments 0 and 1 ----
    {
        const bf16_t* const h0 = frame_base(0);
#pragma unroll
        for (int i = 0; i < HP; ++i) dma_halo(0, h0, i);
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) { dma_w(0, 0, i); dma_w(1, p.Cin, i); dma_w(2, 2L * p.Cin, i); }       // stages 0..2 = taps 0..2 of group 0
    int wk = 3 * p.Cin;                                     // stage 3: tap 3 of group 0
    const int wk_last = ((p.kt - 1) * 9 + 8) * p.Cin + (nc32 - 1) * 32;   // the last stage's offset: issues past the end re-fetch it (see the stage loop)
    int wc32 = 0;                                           // c32 of the group the next weight stage belongs to
    int hdt = nc32 > 1 ? 0 : 1, hc32 = nc32 > 1 ? 1 : 0;    // (dt, c32) of group 1: the first group whose halo the stage loop stages
    const bf16_t* hsrc = p.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MI; ++i) read_a(std::integral_constant<int, 0>{}, H2_AIMM(0), i);
    static_for<0, WD>([&](auto ic) { read_w(ic, H2_WIMM(0), decltype(ic)::value); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // Four groups = 36 stages per loop iteration: the fragment-set parity (36 % 2), the halo buffer (group parity) and the weight ring slot
    // (36 % 4) of every stage are compile-time constants; stages past the end (ngroups % 4 != 0) are skipped by a uniform test.
    // DMA budget: everything issued in stage s-1 or earlier has landed when stage s ends (counted vmcnt: only the pieces of stage s itself may
    // still be in flight), so a piece has a full stage beyond its own to arrive — with a plain vmcnt(0) per stage the stage time WAS the DMA
    // latency (2200 cycles for 1024 cycles of MFMA, whatever the LDS schedule: the first two versions and the 4-wave kernel all ran alike).
    for (int g0 = 0; g0 < ngroups; g0 += 4) {
        static_for<0, 36>([&](auto kc) {
            constexpr int k = decltype(kc)::value;          // stage within the four groups
            constexpr int cur = k & 1, nxt = cur ^ 1;
            constexpr int tap = k % 9;
            const int g = g0 + k / 9;
            const int st = g0 * 9 + k;
            if (st >= nst) return;                          // workgroup-uniform
            __builtin_amdgcn_sched_barrier(0);
            // the wave's halo pieces of the next group go out over the first taps: 12 pieces (4 waves) as 2,2,2,2,2,1,1; 6 pieces (8 waves) as 1,1,1,1,1,1
            constexpr int hfirst = NW == 4 ? (tap < 5 ? 2 * tap : tap + 5) : tap;
            constexpr int hcnt = NW == 4 ? (tap < 5 ? 2 : (tap < 7 ? 1 : 0)) : (tap < 6 ? 1 : 0);
            // NO wave-uniform guards inside the stage: the first version skipped the DMA of stages past the end (w_iss, h_iss), the next-stage fragment reads of the
            // last stage (`more`) and picked one of four vmcnt waits — ~29 s_cbranch per stage and wave between the MFMAs, and a stage took ~1 750 cycles for 1 024 of
            // MFMA with the DMA, the fragment reads or the barrier removed one at a time changing almost nothing (round 4 ablations; tools/ubench/dma_mfma.hip runs
            // the same MFMA / DMA / read mix at 1 100).  Everything past the end is now simply done and harmless: the weight issue re-fetches the last stage into a
            // ring slot nobody reads again, the halo issue re-reads the last group's source into the idle buffer, the fragment reads fetch stale LDS into
            // registers nobody uses; the vmcnt count is then a compile-time constant of the stage.
            // The stage's DMA pieces (2 weight pieces of stage s+3, up to 2 halo pieces of the next group) go out together at the top of the stage.
            // In-kernel timers (instrumented lab builds): a pure MFMA stage is 1021 cycles (= 64 x 16); the fragment reads add ~250; each LDS-DMA
            // piece blocks its wave's issue for ~128 cycles when all four waves issue together — and ~300 when a wave issues alone between its
            // MFMAs (a staggered one-wave-at-a-time schedule was 20 % SLOWER).  On this part the L2 -> LDS fill of a wave does not overlap with that
            // wave's MFMAs, and with one wave per SIMD nobody else fills the gap: stage time = MFMA + fill / (~32 B/clk/CU) + LDS.  That one
            // relation reproduces every kernel here: this one, the GEMM-shaped 512 x 128 convolution (80 KB per 2048 MFMA cycles: 0.83 PFLOP/s)
            // and the DiT's 256 x 256 GEMM (64 KB per 2048: 1.4 PFLOP/s).  Hence: as few fill bytes per MFMA as the tile allows.
            {
                const int wk_use = min(wk, wk_last);
#pragma unroll
                for (int i = 0; i < WP; ++i) dma_w((k + 3) % H2_RING, wk_use, i);
                constexpr int wtap = (k + 3) % 9;           // tap of the stage just issued (the 36-stage body starts at a group boundary)
                if constexpr (wtap < 8) wk += p.Cin;
                else {                                      // next stage: tap 0 of the following group
                    const bool wrap = wc32 + 1 == nc32;
                    wk += wrap ? 32 : 32 - 8 * p.Cin;
                    wc32 = wrap ? 0 : wc32 + 1;
                }
            }
            if constexpr (tap == 0) {                       // this group's stages stage the NEXT group's halo: its source, once
                if (g + 1 < ngroups) hsrc = frame_base(hdt) + hc32 * 32;
                if (hc32 + 1 == nc32) { hc32 = 0; ++hdt; } else ++hc32;
            }
            if constexpr (hcnt > 0) {
#pragma unroll
                for (int i = 0; i < HP; ++i)
                    if (i >= hfirst && i < hfirst + hcnt) dma_halo((g + 1) & 1, hsrc, i);
            }
            // 8 blocks of MI MFMAs (one W fragment x the wave's MI voxel fragments).  After the first half of block ni: W fragment ni + WD (of this
            // stage, or of the next one); after the second half of every ASTEP-th block: a voxel fragment of the NEXT stage.  Reads return in order, so
            // before block ni the W fragment it needs (issued WD blocks earlier) is complete once at most `younger` reads are outstanding:
            // the WD - 1 W reads behind it + the A reads of blocks ni - WD .. ni - 1.
            static_for<0, 8>([&](auto nc) {
                constexpr int ni = decltype(nc)::value;
                constexpr int q = k * 8 + ni;               // W fragment sequence number within the 36-stage iteration (288 % (WD + 1) == 0)
                static_assert(288 % (WD + 1) == 0 && 2 * WD - 1 <= 15, "ring slot must be static; lgkmcnt is a 4-bit counter");
                __builtin_amdgcn_sched_barrier(0);
                {
                    // fragment ni was issued WD blocks ago; since then 2 WD - 1 younger reads went out (LDS returns in order).  With WD = 2 the
                    // fragment was 256 cycles old — less than the LDS latency under this kernel's own 60 % LDS load — and every block stalled
                    bf16x8& need = fwr[q % (WD + 1)];
                    constexpr int a_reads = (ni - 1) / ASTEP - (ni - WD + ASTEP - 1) / ASTEP + 1;     // multiples of ASTEP in [ni - WD, ni - 1] (ni >= WD)
                    constexpr int younger = WD - 1 + a_reads;
                    static_assert(younger <= 15, "lgkmcnt is a 4-bit counter");
                    if constexpr (ni >= WD) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(need) : "n"(younger));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = 0; mi < MI / 2; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwr[q % (WD + 1)], fa[cur][mi], acc[ni][mi], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ni + WD < 8) read_w(std::integral_constant<int, (q + WD) % (WD + 1)>{}, H2_WIMM(k), ni + WD);
                else read_w(std::integral_constant<int, (q + WD) % (WD + 1)>{}, H2_WIMM(k + 1), ni + WD - 8);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mi = MI / 2; mi < MI; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwr[q % (WD + 1)], fa[cur][mi], acc[ni][mi], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ni % ASTEP == 0) read_a(std::integral_constant<int, nxt>{}, H2_AIMM(k + 1), ni / ASTEP);
            });
            __builtin_amdgcn_sched_barrier(0);
            {
                bf16x8 &a0 = fa[nxt][0], &a1 = fa[nxt][1], &a2 = fa[nxt][2], &a3 = fa[nxt][3], &a4 = fa[nxt][MI - 4], &a5 = fa[nxt][MI - 3], &a6 = fa[nxt][MI - 2],
                       &a7 = fa[nxt][MI - 1], &w0 = fwr[0], &w1 = fwr[1], &w2 = fwr[2], &w3 = fwr[3], &w4 = fwr[4], &w5 = fwr[5];
                if constexpr (MI == 8)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(w0), "+v"(w1), "+v"(w2),
                                 "+v"(w3), "+v"(w4), "+v"(w5));
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5));
            }
            // allowed in flight: this stage's own pieces (WP weight pieces, hcnt halo pieces) — all wave-uniform, compile-time counts
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WP + hcnt) : "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        });
    }

    // ---- epilogue: bias, bf16 rounding before the residual add, bf16 store, GroupNorm sums per channel quad (a group is cout / 128 quads) ----
    float gs[8], gq[8];
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) gs[ni] = gq[ni] = 0.f;
    uint2 bq[8];
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) bq[ni] = p.bias ? *(const uint2*)(p.bias + slab * 128 + ni * 16 + (lane >> 4) * 4) : uint2{0u, 0u};
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int y = y0 + RW * wave + (mi >> 1), x = x0 + (mi & 1) * 16 + (lane & 15);
        if (y >= p.Ho || x >= p.Wo) continue;
        const long m = ((long)tf * p.Ho + y) * p.Wo + x;
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
            const int n = slab * 128 + ni * 16 + (lane >> 4) * 4;
            float v[4] = {acc[ni][mi][0] + bf16lo_to_f32(bq[ni].x), acc[ni][mi][1] + bf16hi_to_f32(bq[ni].x),
                          acc[ni][mi][2] + bf16lo_to_f32(bq[ni].y), acc[ni][mi][3] + bf16hi_to_f32(bq[ni].y)};
            if (p.residual) {
                const uint2 rr = *(const uint2*)(p.residual + m * p.ldy + n);
                v[0] = round_bf16(v[0]) + bf16lo_to_f32(rr.x); v[1] = round_bf16(v[1]) + bf16hi_to_f32(rr.x);
                v[2] = round_bf16(v[2]) + bf16lo_to_f32(rr.y); v[3] = round_bf16(v[3]) + bf16hi_to_f32(rr.y);
            }
            uint2 o;
            o.x = pack_bf16x2(v[0], v[1]);
            o.y = pack_bf16x2(v[2], v[3]);
            *(uint2*)(p.y + m * p.ldy + n) = o;
            const float r0 = bf16lo_to_f32(o.x), r1 = bf16hi_to_f32(o.x), r2 = bf16lo_to_f32(o.y), r3 = bf16hi_to_f32(o.y);
            gs[ni] += (r0 + r1) + (r2 + r3);
            gq[ni] += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
        }
    }
    if (p.gn_partial) {
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                gs[ni] += __shfl_xor(gs[ni], off, 64);
                gq[ni] += __shfl_xor(gq[ni], off, 64);
            }
        }
        // The stage loop leaves the weight pieces of its last (past-the-end, never read) stages in flight — its final wait is vmcnt(WP) — and they land in the
        // weight ring; `red` reuses the head of the halo area.  Drain them before LDS is reused at all, so that no ring / `red` placement can ever race.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float* red = (float*)smem;
        if ((lane & 15) == 0) {
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                red[(wave * 32 + ni * 4 + (lane >> 4)) * 2 + 0] = gs[ni];
                red[(wave * 32 + ni * 4 + (lane >> 4)) * 2 + 1] = gq[ni];
            }
        }
        __syncthreads();
        const int qpg = nslab, gps = 32 / nslab;          // channel quads per GroupNorm(32) group (cout / 128), groups per 128-channel slab
        if (tid < 64 && (tid & 31) < gps) {
            // patch i owns row i of the partial buffer and zeroes rows i + ntiles, i + 2 ntiles, i + 3 ntiles (ntiles <= rows <= 4 ntiles), each slab
            // its own group columns: tg_groupnorm_finalize sums all ceil(V / 128) rows.  Fixed order: waves, then the group's quads.
            const int stat = tid >> 5, gl = tid & 31;
            float a = 0.f;
            for (int w_ = 0; w_ < NW; ++w_)
                for (int j = 0; j < qpg; ++j) a += red[(w_ * 32 + gl * qpg + j) * 2 + stat];
            const long rows = ((long)p.To * p.Ho * p.Wo + BM - 1) / BM;
            const int col = stat * 32 + slab * gps + gl;
            p.gn_partial[(long)tile * 64 + col] = a;
            for (long r = (long)tile + ntiles; r < rows; r += ntiles) p.gn_partial[r * 64 + col] = 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv_out of the decoder (Cin = 128 -> Cout = 3, 3x3x3, full resolution).  On the 128 x 16 GEMM-shaped tile every (tap, 64-channel chunk) fetched a fresh
// A tile through the L2 -> LDS fill path: 6.9 KB per output voxel for 10 KFLOP of useful work — the launch was fill-bound (17.7 ms per decode, and under the
// three tile streams it took the fill path from the other tiles' convolutions: skipping it moved the decode's wall time by 17 ms).  Here the halo staging of
// conv3d_halo2_kernel (a 16 x 32 patch per workgroup, the 18 x 34 halo once per (temporal tap, 32-channel chunk), all nine taps read it shifted) feeds ONE
// 16-wide MFMA column: the few real weight rows (Cout <= 4: 27 KB) stay resident in LDS for the whole workgroup, one more all-zero row serves the lanes of
// the unused output channels, and no weight DMA is left in the loop.  Per (group, tap) a wave reads 4 voxel fragments + 1 weight fragment and issues 4 MFMAs,
// the reads of tap + 1 in flight behind the MFMAs of tap; one barrier per GROUP.  LDS-read-bound (6.9 KB per voxel from LDS instead of from L2).
// ------------------------------------------------------------------------------------------------
template <int CIN, int KT>
__global__ __launch_bounds__(512) void conv3d_halo_narrow_kernel(ConvParams p) {
    constexpr int NW = 8, HP = 48 / NW, RW = H2_PH / NW, MI = 2 * RW;
    constexpr int NC32 = CIN / 32, NG = KT * NC32;
    constexpr int WROW = KT * 9 * CIN * 2 + 64;           // LDS stride of a weight row: rows 0, 1, 2, (zero row) start 64 B apart mod 256 -> 16 lanes x 16 B = all banks once
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sH = smem;
    char* const sW = smem + 2 * H2_HALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (p.Wo + H2_PW - 1) / H2_PW, tiles_y = (p.Ho + H2_PH - 1) / H2_PH;
    const int tile = xcd_remap(blockIdx.x, p.To * tiles_y * tiles_x);
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, tf = tile / (tiles_x * tiles_y);
    const int x0 = tx * H2_PW, y0 = ty * H2_PH;
    const long frame = (long)p.H * p.W * CIN;

    // ---- weights: rows [0, cout) of the packed [cout_pad][KT * 9][CIN] tensor as they are, row `cout` = zeros ----
    {
        constexpr int CH = KT * 9 * CIN * 2 / 16;         // 16-byte chunks per row
        for (int i = tid; i < (p.cout + 1) * CH; i += 64 * NW) {
            const int r = i / CH, c = i - r * CH;
            const uint4 v = r < p.cout ? *(const uint4*)(p.w + (long)r * (KT * 9 * CIN) + c * 8) : uint4{0u, 0u, 0u, 0u};
            *(uint4*)(sW + r * WROW + c * 16) = v;
        }
    }
    int hoff[HP];
#pragma unroll
    for (int i = 0; i < HP; ++i) {
        const int o = (wave * HP + i) * 1024 + lane * 16;
        const int hr = o / H2_STRIDE, slot = (o - hr * H2_STRIDE) >> 4;
        const int hy = hr / H2_LW, hx = hr - hy * H2_LW;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        hoff[i] = (slot < 4 && hr < H2_ROWS && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) ? (y * p.W + x) * CIN + slot * 8 : -1;
    }
    auto frame_base = [&](int dt) -> const bf16_t* {
        const int tv = tf + dt - (KT - 1);
        if (tv >= 0) return p.x + (long)tv * frame;
        return p.cache ? p.cache + (long)(tv + KT - 1) * frame : p.x;
    };
    auto dma_halo = [&](int buf, const bf16_t* hsrc, int i) {
        const bf16_t* src = hoff[i] >= 0 ? hsrc + hoff[i] : p.zeros + (lane & 7) * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sH + buf * H2_HALO_BYTES + (wave * HP + i) * 1024), 16, 0, 0);
    };
    uint32_t abase[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
        abase[mi] = (uint32_t)(uintptr_t)sH + (uint32_t)(((RW * wave + (mi >> 1)) * H2_LW + (mi & 1) * 16 + (lane & 15)) * H2_STRIDE + (lane >> 4) * 16);
    const uint32_t wbase = (uint32_t)(uintptr_t)sW + (uint32_t)(min(lane & 15, p.cout) * WROW + (lane >> 4) * 16);

    f32x4 acc[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[2][MI], fw[2];
    auto reads = [&](auto setc, auto gc, auto tapc) {
        constexpr int set = decltype(setc)::value, g = decltype(gc)::value, tap = decltype(tapc)::value;
        constexpr int aimm = (g & 1) * H2_HALO_BYTES + ((tap / 3) * H2_LW + tap % 3) * H2_STRIDE;
        constexpr int wimm = (((g / NC32) * 9 + tap) * CIN + (g % NC32) * 32) * 2;
        static_assert(aimm < 65536 && wimm < 65536, "ds_read offset field");
        {
            bf16x8& d = fw[set];
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(wbase), "n"(wimm));
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            bf16x8& d = fa[set][mi];
            const uint32_t a = abase[mi];
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(aimm));
        }
    };

    {
        const bf16_t* const h0 = frame_base(0);
#pragma unroll
        for (int i = 0; i < HP; ++i) dma_halo(0, h0, i);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    static_for<0, NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g + 1 < NG) {                      // the next group's halo goes out now and has the whole group to land
            const bf16_t* const hsrc = frame_base((g + 1) / NC32) + ((g + 1) % NC32) * 32;
#pragma unroll
            for (int i = 0; i < HP; ++i) dma_halo((g + 1) & 1, hsrc, i);
        }
        reads(std::integral_constant<int, 0>{}, gc, std::integral_constant<int, 0>{});
        static_for<0, 9>([&](auto tc) {
            constexpr int tap = decltype(tc)::value, cur = tap & 1;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (tap < 8) reads(std::integral_constant<int, cur ^ 1>{}, gc, std::integral_constant<int, (tap < 8 ? tap + 1 : 8)>{});
            __builtin_amdgcn_sched_barrier(0);
            {
                bf16x8 &w0 = fw[cur], &a0 = fa[cur][0], &a1 = fa[cur][1], &a2 = fa[cur][2], &a3 = fa[cur][3];
                static_assert(MI == 4, "the wait below names four voxel fragments");
                if constexpr (tap < 8) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(w0), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));   // LDS returns in order: this tap's five
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w0), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[cur], fa[cur][mi], acc[mi], 0, 0, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (g + 1 < NG) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
    });

    // ---- epilogue: lanes 0..15 hold output channels 0..3 (acc element r) of voxel (lane & 15) of each fragment ----
    if (lane < 16) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int y = y0 + RW * wave + (mi >> 1), x = x0 + (mi & 1) * 16 + lane;
            if (y >= p.Ho || x >= p.Wo) continue;
            bf16_t* dst = p.y + (((long)tf * p.Ho + y) * p.Wo + x) * p.ldy;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < p.cout) dst[e] = f32_to_bf16(acc[mi][e] + (p.bias ? bf16_to_f32(p.bias[e]) : 0.f));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// conv_in of the encoder (3 -> 128 channels, 3x3x3, full resolution).  With the input padded to the 64-channel granule of the LDS-DMA loaders the layer ran
// 54 stages of K = 32 for 81 real products per output channel (21x the MACs, 16x the input bytes).  Here the input is channels-last with EIGHT channels
// (16 B per voxel, 3 used), the reduction index is k = tap * 3 + channel (81 -> 96 = three MFMA k-steps, weights packed [128][96] by the host), and a
// workgroup's 16 x 32 patch needs its three 18 x 34 input halos (29 KB) exactly once: the A fragments are gathered from them with 2-byte LDS reads through
// a per-lane table of 24 offsets, the 24 KB of weights sit in LDS, 96 MFMAs per wave.  The kernel is bound by its 256 B per voxel of output.
// Epilogue (bias, bf16 store, GroupNorm sums per patch) as in conv3d_halo2_kernel.
// ------------------------------------------------------------------------------------------------
constexpr int CI_WROW = 208;                                   // LDS stride of a weight row: 96 k x 2 B + 16
constexpr int CI_HALO = 3 * H2_ROWS * 16;                      // three frames x 612 voxels x 8 channels
constexpr int CI_LDS = CI_HALO + 128 * CI_WROW;

__global__ __launch_bounds__(512) void conv3d_in_kernel(ConvParams p) {
    constexpr int NW = 8, RW = H2_PH / NW, MI = 2 * RW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sH = smem;
    char* const sW = smem + CI_HALO;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (p.Wo + H2_PW - 1) / H2_PW, tiles_y = (p.Ho + H2_PH - 1) / H2_PH;
    const int ntiles = p.To * tiles_y * tiles_x;
    const int tile = xcd_remap(blockIdx.x, ntiles);
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, tf = tile / (tiles_x * tiles_y);
    const int x0 = tx * H2_PW, y0 = ty * H2_PH;
    const long frame = (long)p.H * p.W * 8;

    for (int i = tid; i < 128 * 12; i += 64 * NW) {            // weights [128][96]: 12 chunks of 16 B per row
        const int r = i / 12, c = i - r * 12;
        *(uint4*)(sW + r * CI_WROW + c * 16) = *(const uint4*)(p.w + r * 96 + c * 8);
    }
    for (int i = tid; i < 3 * H2_ROWS; i += 64 * NW) {         // halos of frames tf - 2, tf - 1, tf (cache / frame 0 replicated before the window)
        const int dt = i / H2_ROWS, hr = i - dt * H2_ROWS;
        const int hy = hr / H2_LW, hx = hr - hy * H2_LW;
        const int y = y0 - 1 + hy, x = x0 - 1 + hx;
        uint4 v{0u, 0u, 0u, 0u};
        if ((unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W) {
            const int tv = tf + dt - 2;
            const bf16_t* base = tv >= 0 ? p.x + (long)tv * frame : (p.cache ? p.cache + (long)(tv + 2) * frame : p.x);
            v = *(const uint4*)(base + ((long)y * p.W + x) * 8);
        }
        *(uint4*)(sH + i * 16) = v;
    }
    // per-lane gather table: element j of k-step ks is k = 32 ks + 8 (lane >> 4) + j = tap * 3 + channel; k >= 81 reads a zero (channel 3 of the lane's own voxel)
    int off[3][8];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ks * 32 + (lane >> 4) * 8 + j;
            const int tap = k / 3, ci = k - tap * 3;
            const int dt = tap / 9, dh = (tap - dt * 9) / 3, dw = tap % 3;
            off[ks][j] = k < 81 ? ((dt * (H2_PH + 2) + dh) * H2_LW + dw) * 16 + ci * 2 : 6;
        }
    __syncthreads();

    f32x4 acc[8][MI];
#pragma unroll
    for (int ni = 0; ni < 8; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        bf16x8 fa[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const char* vb = sH + ((RW * wave + (mi >> 1)) * H2_LW + (mi & 1) * 16 + (lane & 15)) * 16;
            uint32_t u[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                u[j] = (uint32_t)*(const uint16_t*)(vb + off[ks][2 * j]) | ((uint32_t)*(const uint16_t*)(vb + off[ks][2 * j + 1]) << 16);
            const uint4 q{u[0], u[1], u[2], u[3]};
            fa[mi] = __builtin_bit_cast(bf16x8, q);
        }
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
            const bf16x8 fw = *(const bf16x8*)(sW + (ni * 16 + (lane & 15)) * CI_WROW + ks * 64 + (lane >> 4) * 16);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, fa[mi], acc[ni][mi], 0, 0, 0);
        }
    }

    // ---- epilogue (conv3d_halo2_kernel's, one 128-channel slab): bias, bf16 store, GroupNorm sums per channel quad ----
    float gs[8], gq[8];
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) gs[ni] = gq[ni] = 0.f;
    uint2 bq[8];
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) bq[ni] = p.bias ? *(const uint2*)(p.bias + ni * 16 + (lane >> 4) * 4) : uint2{0u, 0u};
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int y = y0 + RW * wave + (mi >> 1), x = x0 + (mi & 1) * 16 + (lane & 15);
        if (y >= p.Ho || x >= p.Wo) continue;
        const long m = ((long)tf * p.Ho + y) * p.Wo + x;
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
            const int n = ni * 16 + (lane >> 4) * 4;
            uint2 o;
            o.x = pack_bf16x2(acc[ni][mi][0] + bf16lo_to_f32(bq[ni].x), acc[ni][mi][1] + bf16hi_to_f32(bq[ni].x));
            o.y = pack_bf16x2(acc[ni][mi][2] + bf16lo_to_f32(bq[ni].y), acc[ni][mi][3] + bf16hi_to_f32(bq[ni].y));
            *(uint2*)(p.y + m * p.ldy + n) = o;
            const float r0 = bf16lo_to_f32(o.x), r1 = bf16hi_to_f32(o.x), r2 = bf16lo_to_f32(o.y), r3 = bf16hi_to_f32(o.y);
            gs[ni] += (r0 + r1) + (r2 + r3);
            gq[ni] += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
        }
    }
    if (p.gn_partial) {
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
#pragma unroll
            for (int o_ = 1; o_ < 16; o_ <<= 1) {
                gs[ni] += __shfl_xor(gs[ni], o_, 64);
                gq[ni] += __shfl_xor(gq[ni], o_, 64);
            }
        }
        __syncthreads();                                   // every wave is past its LDS reads
        float* red = (float*)smem;
        if ((lane & 15) == 0) {
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                red[(wave * 32 + ni * 4 + (lane >> 4)) * 2 + 0] = gs[ni];
                red[(wave * 32 + ni * 4 + (lane >> 4)) * 2 + 1] = gq[ni];
            }
        }
        __syncthreads();
        if (tid < 64) {                                    // one channel quad per GroupNorm(32) group at 128 channels; fixed order over the waves
            const int stat = tid >> 5, gl = tid & 31;
            float a = 0.f;
            for (int w_ = 0; w_ < NW; ++w_) a += red[(w_ * 32 + gl) * 2 + stat];
            const long rows = ((long)p.To * p.Ho * p.Wo + BM - 1) / BM;
            const int col = stat * 32 + gl;
            p.gn_partial[(long)tile * 64 + col] = a;
            for (long r = (long)tile + ntiles; r < rows; r += ntiles) p.gn_partial[r * 64 + col] = 0.f;
        }
    }
}

// Split-K epilogue: sum the ksplit fp32 partial tensors in a fixed order, then exactly what conv3d_cl_kernel's own epilogue does (bias, the
// reference's bf16 rounding before the residual add, bf16 store, per-128-voxel-tile GroupNorm sums of the stored values).  One workgroup per
// (128-voxel tile, 128-channel slab) — a slab holds whole GroupNorm groups (cout / 32 <= 16 channels each) — so a 30 x 45 latent tile's 512-channel
// layer runs 88-128 workgroups of 16 row passes; the first version (one workgroup per voxel tile over all channels, 64 serial passes of 8
// dependent loads) took 110 us per call on the tile's critical path.  cout == cout_pad in {128, 256, 512}.
template <int KS>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const float* __restrict__ kpart, int ksplit, long M, int cout,
                                                                 const bf16_t* __restrict__ bias, const bf16_t* __restrict__ residual,
                                                                 bf16_t* __restrict__ y, long ldy, float* __restrict__ gn_partial) {
    __shared__ float red[256][2];
    const int tid = threadIdx.x;
    const int cq = tid & 31, rl = tid >> 5;         // 32 channel quads = 128 channels, 8 rows per pass
    const int n = blockIdx.y * 128 + cq * 4;
    const long m0 = (long)blockIdx.x * BM;
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
        const uint2 bb = *(const uint2*)(bias + n);
        b4[0] = bf16lo_to_f32(bb.x); b4[1] = bf16hi_to_f32(bb.x); b4[2] = bf16lo_to_f32(bb.y); b4[3] = bf16hi_to_f32(bb.y);
    }
    const long plane = M * cout;
    float gs = 0.f, gq = 0.f;
    for (int r = rl; r < BM; r += 8) {
        const long m = m0 + r;
        if (m >= M) break;
        const float* src = kpart + m * cout + n;
        f32x4 part[KS > 0 ? KS : 1];
        f32x4 v;
        if constexpr (KS > 0) {                     // all partial loads in flight together, summed in the fixed order 0, 1, 2, ...
#pragma unroll
            for (int sp = 0; sp < KS; ++sp) part[sp] = *(const f32x4*)(src + sp * plane);
            v = part[0];
#pragma unroll
            for (int sp = 1; sp < KS; ++sp) v += part[sp];
        } else {
            v = *(const f32x4*)src;
            for (int sp = 1; sp < ksplit; ++sp) v += *(const f32x4*)(src + sp * plane);
        }
        float u[4] = {v[0] + b4[0], v[1] + b4[1], v[2] + b4[2], v[3] + b4[3]};
        if (residual) {
            const uint2 rr = *(const uint2*)(residual + m * ldy + n);
            u[0] = round_bf16(u[0]) + bf16lo_to_f32(rr.x); u[1] = round_bf16(u[1]) + bf16hi_to_f32(rr.x);
            u[2] = round_bf16(u[2]) + bf16lo_to_f32(rr.y); u[3] = round_bf16(u[3]) + bf16hi_to_f32(rr.y);
        }
        uint2 o;
        o.x = pack_bf16x2(u[0], u[1]);
        o.y = pack_bf16x2(u[2], u[3]);
        *(uint2*)(y + m * ldy + n) = o;
        const float r0 = bf16lo_to_f32(o.x), r1 = bf16hi_to_f32(o.x), r2 = bf16lo_to_f32(o.y), r3 = bf16hi_to_f32(o.y);
        gs += (r0 + r1) + (r2 + r3);
        gq += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
    }
    if (!gn_partial) return;
    red[tid][0] = gs; red[tid][1] = gq;
    __syncthreads();
    const int cg = cout / GN_GROUPS, qpg = cg >> 2, gps = 128 / cg;      // channels per group, quads per group, groups per 128-channel slab
    if (tid < 2 * gps) {                            // fixed order: quads of the group, then the 8 row lanes
        const int stat = tid / gps, gl = tid % gps;
        float a = 0.f;
        for (int qd = 0; qd < qpg; ++qd)
            for (int r = 0; r < 8; ++r) a += red[r * 32 + gl * qpg + qd][stat];
        gn_partial[(long)blockIdx.x * 2 * GN_GROUPS + stat * GN_GROUPS + blockIdx.y * gps + gl] = a;
    }
}

// ------------------------------------------------------------------------------------------------
// The same convolution on the 4-wave structure of the DiT GEMM (gemm.hip: gemm256w4_kernel): 256 voxels x 256 output channels per
// workgroup, four waves (one per SIMD) of 128x128 = 8x8 MFMA 16x16x32 with the accumulators in AGPRs, two LDS stages of K = 64
// (one tap x 64 input channels), 8 rows x 128 B per LDS-DMA piece, the wave software-pipelines itself (128 MFMA slots per stage, each
// followed by at most one ds_read_b128 or one DMA piece).  Only the A side differs from the GEMM: its pieces are fetched through
// per-lane pointers (tap_src: padding -> zero page, frames before the window -> cache tensor) that are recomputed when the tap
// changes, every Cin/64 stages.  Used for Cout % 256 == 0 without a temporal index map; everything else stays on conv3d_cl_kernel.
// ------------------------------------------------------------------------------------------------
constexpr int CW_OPER = 256 * 64 * 2;            // 32 KiB per operand per stage
constexpr int CW_STAGE = 2 * CW_OPER;            // 64 KiB
constexpr int CW_LDS = 2 * CW_STAGE;             // 128 KiB

// NT = 256: 256 voxels x 256 channels, waves 2 x 2.  NT = 128 (Cout = 128): 512 voxels x 128 channels, waves 4 x 1 - the same 128x128 per
// wave, every wave reads the whole W tile; 80 KiB per stage, i.e. all 160 KiB of LDS for the two stages.
template <int NT>
__global__ __launch_bounds__(256) void conv3d_w4_kernel(ConvParams p) {
    constexpr int TM = NT == 256 ? 256 : 512;            // voxels per tile
    constexpr int NA = TM / 32, NW = NT / 32;            // A / W pieces (8 rows each) per wave and stage
    constexpr int OPER_A = TM * 128, STAGE = OPER_A + NT * 128;
    constexpr int DS = NT == 256 ? 6 : 4;                // MFMA slots between DMA pieces: NA + NW pieces from slot 36
    constexpr int ISSUED_AT_93 = (93 - 36) / DS + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NT == 256 ? wave >> 1 : wave, wn = NT == 256 ? wave & 1 : 0;

    const long M = (long)p.To * p.Ho * p.Wo;
    const int tiles_m = (int)((M + TM - 1) / TM), tiles_n = p.cout_pad / NT;
    // tile order: contiguous chunk of tiles per XCD, n fastest (the tiles_n blocks sharing an A tile run on one XCD).  A temporal-locality order
    // (an XCD walks a spatial stripe through all frames) measured neutral — 200.2 vs 199.9 ms on the 256 -> 256 layers: the 5-6x algorithmic
    // L2-miss bytes PMC shows for these launches are absorbed behind L2 and already hidden.
    const int Kw = p.kt * p.kh * p.kw * p.Cin;        // row length of the packed weights
    int t = xcd_remap(blockIdx.x, tiles_m * tiles_n * (p.o_phases == 4 ? 4 : 1));
    // the phase of a tg_conv3d_up2_subpixel launch: workgroup-uniform padding, output offsets, weight block and GroupNorm rows
    int pad_h = p.pad, pad_w = p.pad + p.pad_w_off, o_py = p.o_py, o_px = p.o_px;
    const bf16_t* wbase = p.w;
    float* gnp = p.gn_partial;
    if (p.o_phases == 4) {
        const int per = tiles_m * tiles_n, ph = t / per;
        t -= ph * per;
        o_py = ph >> 1; o_px = ph & 1;
        pad_h = 1 - o_py; pad_w = 1 - o_px;
        wbase += (long)ph * p.cout * Kw;
        if (gnp) gnp += (long)ph * ((M + BM - 1) / BM) * 2 * GN_GROUPS;
    }
    const int tn = t % tiles_n, tm = t / tiles_n;
    const long m0 = (long)tm * TM;
    const int n0 = tn * NT;

    // ---- A side: piece i (0..7) of this wave = tile rows [wave*64 + i*8, +8); lane -> row + (lane>>3), physical slot lane&7 ----
    // Per tile and piece the lane keeps its voxel (t << 22 | h << 11 | w) and, for the plain case (stride 1, no upsampling), the element
    // offset of that voxel and a 9-bit mask of the (dh, dw) taps that stay inside the image: a tap change (every Cin/64 stages, with
    // the matrix pipe idle - there is no second wave to cover it) is then ~10 VALU per piece instead of ~28 through tap_src.
    int vthw[NA], cen[NA], vmask[NA];
    const bool plain = p.stride == 1 && p.up == 1;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        long m = m0 + wave * (NA * 8) + i * 8 + (lane >> 3);
        if (m >= M) m = M - 1;
        const int wo = (int)(m % p.Wo);
        const long q = m / p.Wo;
        const int ho = (int)(q % p.Ho), to = (int)(q / p.Ho);
        vthw[i] = (to << 22) | (ho << 11) | wo;
        cen[i] = ((to * p.H + ho) * p.W + wo) * p.Cin;
        int mk = 0;
        for (int dh = 0; dh < p.kh; ++dh)
            for (int dw = 0; dw < p.kw; ++dw)
                if ((unsigned)(ho + dh - pad_h) < (unsigned)p.H && (unsigned)(wo + dw - pad_w) < (unsigned)p.W) mk |= 1 << (dh * 3 + dw);
        vmask[i] = mk;
    }
    const int Hv = p.H * p.up, Wv = p.W * p.up;
    const long frame = (long)p.H * p.W * p.Cin;
    const int prow = lane >> 3;
    const int slotA[2] = {((lane & 7) ^ (prow >> 1)) * 8, ((lane & 7) ^ (4 + (prow >> 1))) * 8};   // logical slot by piece parity (elements)
    auto tap_src = [&](int i, int dt, int dh, int dw) -> const char* {
        const int to = vthw[i] >> 22, ho = (vthw[i] >> 11) & 2047, wo = vthw[i] & 2047;
        const int hv = ho * p.stride + dh - p.pad, wv = wo * p.stride + dw - p.pad;
        if (hv < 0 || hv >= Hv || wv < 0 || wv >= Wv) return (const char*)(p.zeros + slotA[i & 1]);
        int tv = to + dt - (p.kt - 1);
        const bf16_t* base = p.x;
        if (tv < 0) {
            if (p.cache) { base = p.cache; tv += p.kt - 1; } else tv = 0;   // cached frames, or replicate the first frame
        }
        const int h = (p.up == 2) ? (hv >> 1) : hv, w = (p.up == 2) ? (wv >> 1) : wv;
        return (const char*)(base + (long)tv * frame + ((long)h * p.W + w) * p.Cin + slotA[i & 1]);
    };
    const int cpt = p.Cin / 64;                      // stages per tap
    const int nk = p.kt * p.kh * p.kw * cpt;
    const char* srcA[NA];
    int t_dw = 0, t_dh = 0, t_dt = 0;                // (dt, dh, dw) of the tap set_tap is called for next: taps are visited in order 0, 1, 2, ...
    auto set_tap = [&](int) {
        const int dw = t_dw, dh = t_dh, dt = t_dt;     // (no tap % kw, / kw, % kh, / kh: runtime divisions on the scalar unit at every tap change)
        if (++t_dw == p.kw) { t_dw = 0; if (++t_dh == p.kh) { t_dh = 0; ++t_dt; } }
        if (plain) {
            // in-window frames: x + cen + doff.  Frames before the window (to + dt < kt-1): the cache tensor holds frames -(kt-1)..-1 at
            // indices 0.., i.e. cache + cen + doff + (kt-1)*frame; without a cache frame 0 is replicated: x + cen + doff - tv*frame
            const int doff = ((dt - (p.kt - 1)) * p.H + (dh - pad_h)) * p.W * p.Cin + (dw - pad_w) * p.Cin;
            const int bit = 1 << (dh * 3 + dw), tneed = p.kt - 1 - dt;     // the tap reads frame to - tneed
            const bf16_t* early = p.cache ? p.cache + (long)(p.kt - 1) * frame : p.x;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int to = vthw[i] >> 22;
                int off = cen[i] + doff;
                const bool before = to < tneed;
                if (before && !p.cache) off += (tneed - to) * (int)frame;   // replicate frame 0
                const bf16_t* src = (before ? early : p.x) + off + slotA[i & 1];
                srcA[i] = (const char*)((vmask[i] & bit) ? src : p.zeros + slotA[i & 1]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) srcA[i] = tap_src(i, dt, dh, dw);
        }
    };
    // ---- W side: buffer loads exactly as in the GEMM ----
    int voffW[2];
#pragma unroll
    for (int odd = 0; odd < 2; ++odd) {
        const int dslot = (lane & 7) ^ (odd * 4 + (prow >> 1));
        voffW[odd] = (int)(((long)(wave * (NW * 8) + prow) * Kw + dslot * 8) * 2);
    }
    const int pieceW = Kw * 16;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)(wbase + (long)n0 * Kw), 0, (int)((long)NT * Kw * 2), 0x00020000);
    int dk = 0, dcc = 0, dtap = 0, dbuf = 0;          // DMA cursor: stage, channel block inside the tap, tap, LDS buffer
    auto dma_piece = [&](int q) {                    // q = 0..NA-1: A pieces, NA..NA+NW-1: W pieces of the cursor's stage
        if (q < NA) {
            char* dst = smem + dbuf * STAGE + (wave * (NA * 8) + q * 8) * 128;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[q] + (long)dcc * 128),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        } else {
            const int qw = q - NA;
            char* dst = smem + dbuf * STAGE + OPER_A + (wave * (NW * 8) + qw * 8) * 128;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (__attribute__((address_space(3))) void*)dst, 16, voffW[qw & 1], dk * 128 + qw * pieceW, 0, 0);   // piece offset in the SCALAR offset: 2 VGPRs
        }
    };
    auto dma_advance = [&]() {
        dbuf ^= 1; ++dk;
        if (++dcc == cpt) { dcc = 0; ++dtap; }
    };

    // ---- fragment addresses (16x16x32 operands: lane -> row lane&15 of a 16-row block, logical 16-B slot ks*4 + (lane>>4)) ----
    const int l15 = lane & 15, ch = lane >> 4;
    const int swz = l15 * 128 + ((ch ^ ((l15 >> 1) & 7)) << 4);
    int ra[2], rw[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        ra[ks] = (wm * 128 * 128 + swz) ^ (ks << 6);
        rw[ks] = (OPER_A + wn * 128 * 128 + swz) ^ (ks << 6);
    }
    int flip = STAGE;                                // the fragment addresses alternate between the two stage buffers
    bf16x8 fa[2][8], fw[2][8];
    f32x4 acc[8][8];                                 // [m block][n block]; lane holds D[n = 4*(lane>>4) + r][m = lane&15]
#define CW_SB() __builtin_amdgcn_sched_barrier(0)
    auto frag_read = [&](auto ksc, auto rc) {        // read r (0..15) of k-step KS: 0..7 -> W blocks, 8..15 -> A blocks
        constexpr int KS = decltype(ksc)::value, R = decltype(rc)::value;
        if constexpr (R < 8) {
            bf16x8& d = fw[KS][R];
            const int ad = rw[KS];
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"(R * 2048));
        } else {
            bf16x8& d = fa[KS][R - 8];
            const int ad = ra[KS];
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"((R - 8) * 2048));
        }
    };
    // the GEMM's slot schedule (gemm.hip): reads of set 1 in slots 0..30, barrier at 35, DMA from 36 every 6th, vmcnt + barrier at 93,
    // reads of the next stage's set 0 from 94
    auto kstage = [&](auto steady_c, bool rd) {
        constexpr bool STEADY = decltype(steady_c)::value;
        if (STEADY && dcc == 0) set_tap(dtap);
        static_for<0, 128>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            constexpr int KS = I >> 6, NB = (I >> 3) & 7, MB = I & 7;
            acc[MB][NB] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[KS][NB], fa[KS][MB], acc[MB][NB], 0, 0, 0);
            CW_SB();
            if constexpr (I < 32 && I % 2 == 0) frag_read(std::integral_constant<int, 1>{}, std::integral_constant<int, I / 2>{});
            if constexpr (I == 35) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                CW_SB();
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) { ra[ks] += flip; rw[ks] += flip; }
                flip = -flip;
            }
            if constexpr (I >= 36 && (I - 36) % DS == 0 && (I - 36) / DS < NA + NW) {
                if (STEADY) dma_piece((I - 36) / DS);
            }
            if constexpr (I == 93) {
                if (STEADY) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ISSUED_AT_93) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                CW_SB();
            }
            if constexpr (I >= 94 && (I - 94) % 2 == 0 && (I - 94) / 2 < 16) {
                if (STEADY || rd) frag_read(std::integral_constant<int, 0>{}, std::integral_constant<int, (I - 94) / 2>{});
            }
            CW_SB();
        });
        if (STEADY) dma_advance();
        CW_SB();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        CW_SB();
    };

    // prologue: stages 0 and 1 in flight, fragment set 0 of stage 0 in registers
    set_tap(0);
    static_for<0, NA + NW>([&](auto qc) { dma_piece(decltype(qc)::value); });
    dma_advance();
    if (dcc == 0) set_tap(dtap);
    static_for<0, NA + NW>([&](auto qc) { dma_piece(decltype(qc)::value); });
    dma_advance();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NW) : "memory");
    __builtin_amdgcn_s_barrier();
    CW_SB();
    static_for<0, 16>([&](auto rc) { frag_read(std::integral_constant<int, 0>{}, rc); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CW_SB();
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jn = 0; jn < 8; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
        int st = 0;
        for (; st + 2 < nk; ++st) kstage(std::true_type{}, true);
        kstage(std::false_type{}, true);
        kstage(std::false_type{}, false);
    }
#undef CW_SB

    // ---- epilogue: bias (+ residual), GroupNorm partial sums of the stored bf16 values; all chunks of the lane are requested first ----
    uint2 bq[8], rq[2][8];                            // residual chunks: the next 16-row block's are in flight while this one is stored
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
        const int n = n0 + wn * 128 + nb * 16 + ch * 4;
        bq[nb] = p.bias ? *(const uint2*)(p.bias + n) : uint2{0u, 0u};
    }
    auto res_load = [&](auto mbc) {
        constexpr int MB = decltype(mbc)::value;
        const long m = m0 + wm * 128 + MB * 16 + l15;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            const int n = n0 + wn * 128 + nb * 16 + ch * 4;
            rq[MB & 1][nb] = (p.residual && m < M) ? *(const uint2*)(p.residual + m * p.ldy + n) : uint2{0u, 0u};
        }
    };
    float gs[8], gq[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) gs[nb] = gq[nb] = 0.f;
    res_load(std::integral_constant<int, 0>{});
    static_for<0, 8>([&](auto mbc) {
        constexpr int mb = decltype(mbc)::value;
        if constexpr (mb < 7) res_load(std::integral_constant<int, (mb < 7 ? mb + 1 : 7)>{});
        const long m = m0 + wm * 128 + mb * 16 + l15;
        if (m < M) {
            long mo = m, mo2 = -1;                          // where the voxel is stored: itself, or its place(s) in the 2x upsampled tensor (phase launch)
            float wgt = 1.f;                                // ... and how many output voxels it stands for in the GroupNorm sums
            if (p.o_up == 2) {
                const int wo_ = (int)(m % p.Wo);
                const long q_ = m / p.Wo;
                const int ho_ = (int)(q_ % p.Ho), to_ = (int)(q_ / p.Ho);
                int t0 = to_, t1 = -1;                      // time-duplicated frames: a 2-D convolution per frame gives duplicate outputs for duplicate inputs
                if (p.o_tdup == 1) { t0 = to_ == 0 ? 0 : 2 * to_ - 1; t1 = to_ == 0 ? -1 : 2 * to_; }
                else if (p.o_tdup == 2) { t0 = 2 * to_; t1 = 2 * to_ + 1; }
                const long inner = (long)(2 * ho_ + o_py) * (2L * p.Wo) + 2 * wo_ + o_px, fr = 4L * p.Ho * p.Wo;
                mo = t0 * fr + inner;
                if (t1 >= 0) { mo2 = t1 * fr + inner; wgt = 2.f; }
            }
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) {
                const int n = n0 + wn * 128 + nb * 16 + ch * 4;
                float v[4] = {acc[mb][nb][0], acc[mb][nb][1], acc[mb][nb][2], acc[mb][nb][3]};
                const uint2 bb = bq[nb];
                v[0] += bf16lo_to_f32(bb.x); v[1] += bf16hi_to_f32(bb.x);
                v[2] += bf16lo_to_f32(bb.y); v[3] += bf16hi_to_f32(bb.y);
                if (p.residual) {   // conv output is a bf16 tensor in the reference before `hidden_states + inputs`
                    const uint2 rr = rq[mb & 1][nb];
                    v[0] = round_bf16(v[0]) + bf16lo_to_f32(rr.x); v[1] = round_bf16(v[1]) + bf16hi_to_f32(rr.x);
                    v[2] = round_bf16(v[2]) + bf16lo_to_f32(rr.y); v[3] = round_bf16(v[3]) + bf16hi_to_f32(rr.y);
                }
                uint2 o;
                o.x = pack_bf16x2(v[0], v[1]);
                o.y = pack_bf16x2(v[2], v[3]);
                *(uint2*)(p.y + mo * p.ldy + n) = o;
                if (mo2 >= 0) *(uint2*)(p.y + mo2 * p.ldy + n) = o;
                if (gnp) {
                    const float r0 = bf16lo_to_f32(o.x), r1 = bf16hi_to_f32(o.x), r2 = bf16lo_to_f32(o.y), r3 = bf16hi_to_f32(o.y);
                    gs[nb] += wgt * ((r0 + r1) + (r2 + r3));
                    gq[nb] += wgt * ((r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3));
                }
            }
        }
    });
    // GroupNorm partial sums in the layout of the 128-row kernel: one row of [2][32] per 128 voxels = per (tile, wm half); fixed order
    if (gnp) {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {
                gs[nb] += __shfl_xor(gs[nb], off, 64);
                gq[nb] += __shfl_xor(gq[nb], off, 64);
            }
        }
        __syncthreads();                                    // every wave is past its last fragment read: the stage buffers are free
        float* red = (float*)smem;                          // [wave][nb][ch][2]
        if (l15 == 0) {
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) {
                red[((wave * 8 + nb) * 4 + ch) * 2 + 0] = gs[nb];
                red[((wave * 8 + nb) * 4 + ch) * 2 + 1] = gq[nb];
            }
        }
        __syncthreads();
        const int cg = p.cout / GN_GROUPS, qpg = cg >> 2, gpt = NT / cg;     // channels per group, quads per group, groups per tile
        constexpr int RB = TM / 128, CW = NT / 128;         // 128-voxel row blocks and 128-column waves of the tile
        if (tid < RB * 2 * gpt) {
            const int wm_ = tid / (2 * gpt), stat = (tid / gpt) & 1, gl = tid % gpt;
            float a = 0.f;
            for (int qd = 0; qd < qpg; ++qd) {
                const int cq = gl * qpg + qd;                // tile-local channel quad = wn*32 + nb*4 + ch
                a += red[(((wm_ * CW + (cq >> 5)) * 8 + ((cq >> 2) & 7)) * 4 + (cq & 3)) * 2 + stat];
            }
            const long prow_ = (long)tm * RB + wm_;
            if (prow_ * 128 < M) gnp[prow_ * 2 * GN_GROUPS + stat * GN_GROUPS + n0 / cg + gl] = a;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// GroupNorm(32) statistics: stage 1 per-block fp32 partial (sum, sumsq) per group, stage 2 fp64 finalise
// ------------------------------------------------------------------------------------------------
constexpr int GN_ROWS_PER_BLOCK = 512;

__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t* __restrict__ x, long V, int C, float* __restrict__ partial) {
    __shared__ float sh[256][17];                         // per-thread (sum, sumsq) of its 8 channels; +1 pad against bank conflicts
    const int vec_per_row = C >> 3;                       // 16-byte vectors per voxel
    const int cg = C / GN_GROUPS;                         // channels per group (>= 2)
    const long row0 = (long)blockIdx.x * GN_ROWS_PER_BLOCK;
    const long rows = min((long)GN_ROWS_PER_BLOCK, V - row0);
    const long nvec = rows * vec_per_row;
    // a thread always visits the same channel slice when blockDim % vec_per_row == 0 (C in {64..2048} -> yes)
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = q[i] = 0.f;
    for (long v = threadIdx.x; v < nvec; v += 256) {
        const uint4 raw = *(const uint4*)(x + (row0 * vec_per_row + v) * 8);
        const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = bf16lo_to_f32(u[i]), b = bf16hi_to_f32(u[i]);
            s[2 * i] += a; q[2 * i] += a * a;
            s[2 * i + 1] += b; q[2 * i + 1] += b * b;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { sh[threadIdx.x][i] = s[i]; sh[threadIdx.x][8 + i] = q[i]; }
    __syncthreads();
    // fixed summation order (no atomics): thread (stat, g) walks the group's channels and, per channel, the threads that own it
    if (threadIdx.x < 2 * GN_GROUPS) {
        const int st = threadIdx.x / GN_GROUPS, g = threadIdx.x % GN_GROUPS;
        float acc = 0.f;
        for (int c = g * cg; c < (g + 1) * cg; ++c) {
            const int cv = c >> 3, i = c & 7;
            for (int t = cv; t < 256; t += vec_per_row) acc += sh[t][st * 8 + i];
        }
        partial[(long)blockIdx.x * 2 * GN_GROUPS + threadIdx.x] = acc;
    }
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, int nblocks, long V, int C, float eps, float* __restrict__ stats) {
    // one workgroup per group: thread k sums partial blocks k, k+256, ... in fp64, then a fixed-order tree (shuffles inside a wave,
    // four wave results through LDS) — the summation order depends only on nblocks
    __shared__ double sh[2][4];
    const int g = blockIdx.x, k = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int b = k; b < nblocks; b += 256) {
        s += (double)partial[(long)b * 2 * GN_GROUPS + g];
        q += (double)partial[(long)b * 2 * GN_GROUPS + GN_GROUPS + g];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off, 64);
        q += __shfl_xor(q, off, 64);
    }
    if ((k & 63) == 0) { sh[0][k >> 6] = s; sh[1][k >> 6] = q; }
    __syncthreads();
    if (k == 0) {
        s = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        q = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
        const double n = (double)V * (C / GN_GROUPS);
        const double mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0) var = 0;
        stats[2 * g] = (float)mean;
        stats[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// GroupNorm statistics WITHOUT a finalise launch of their own: the norm pass that consumes a convolution's output turns the per-tile sums the convolution's
// epilogue left (rows of [2][32]: sum, sum of squares per group) into (mean, rstd) in its own prologue — every workgroup does the same few KB of work in
// the same fixed order (thread (part, column) adds rows part, part + 4, ... in fp64; the four parts meet in LDS), so the result does not depend on which
// workgroup computes it.  Up to GN_FOLD_ROWS rows are read as they are; longer lists are first cut to <= GN_FOLD_ROWS fp64 rows by gn_reduce_kernel (one
// coalesced pass, a few microseconds — the launch it replaces walked the rows column-wise from 32 workgroups and averaged 26 us).
constexpr int GN_FOLD_ROWS = 64;

struct GnSums {
    const void* rows;     // [n][2][32] float (is_f64 == 0: a convolution's gn_partial) or double (is_f64 == 1: gn_reduce_kernel's output); nullptr: use `stats`
    int n, is_f64;
    long V;               // voxels behind the sums (the tensor the norm reads)
    float eps;
};

__device__ __forceinline__ void gn_stats_to_lds(const GnSums g, const float* __restrict__ stats, int C, float (*sst)[2]) {
    __shared__ double sred[4][64];
    const int t = threadIdx.x;
    if (!g.rows) {                                        // statistics were computed by a launch of their own (tg_groupnorm_stats / _finalize)
        if (t < 64) sst[t >> 1][t & 1] = stats[t];
        __syncthreads();
        return;
    }
    const int col = t & 63, part = t >> 6;
    double a = 0.0;
    if (g.is_f64) {
        const double* r = (const double*)g.rows;
        for (int b = part; b < g.n; b += 4) a += r[(long)b * 64 + col];
    } else {
        const float* r = (const float*)g.rows;
        for (int b = part; b < g.n; b += 4) a += (double)r[(long)b * 64 + col];
    }
    sred[part][col] = a;
    __syncthreads();
    if (t < GN_GROUPS) {
        const double s = (sred[0][t] + sred[1][t]) + (sred[2][t] + sred[3][t]);
        const double q = (sred[0][GN_GROUPS + t] + sred[1][GN_GROUPS + t]) + (sred[2][GN_GROUPS + t] + sred[3][GN_GROUPS + t]);
        const double n = (double)g.V * (C / GN_GROUPS);
        const double mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0) var = 0;
        sst[t][0] = (float)mean;
        sst[t][1] = (float)(1.0 / sqrt(var + (double)g.eps));
    }
    __syncthreads();
}

// rows [n][64] float -> out [gridDim.x][64] double: workgroup r adds the rows of its contiguous range (thread (part, column): rows part, part + 4, ... of
// the range; parts joined in a fixed order).  Whole 256-byte rows per wave load.
__global__ __launch_bounds__(256) void gn_reduce_kernel(const float* __restrict__ rows, long n, double* __restrict__ out) {
    __shared__ double sred[4][64];
    const int t = threadIdx.x, col = t & 63, part = t >> 6;
    const long per = (n + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * per, r1 = min(n, r0 + per);
    double a = 0.0;
    long b = r0 + part;
    for (; b + 28 < r1; b += 32) {                        // eight independent loads in flight, added in row order (the first version waited for every row:
        float v[8];                                       // 39 us for a 5 400-row list, all of it load latency)
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = rows[(b + 4 * u) * 64 + col];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += (double)v[u];
    }
    for (; b < r1; b += 4) a += (double)rows[b * 64 + col];
    sred[part][col] = a;
    __syncthreads();
    if (t < 64) out[(long)blockIdx.x * 64 + t] = (sred[0][t] + sred[1][t]) + (sred[2][t] + sred[3][t]);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
// the streaming norm passes are VALU-bound before they are HBM-bound (~30 VALU per element with a true division and per-value bf16 round
// trips): pairwise rounding through one v_cvt_pk_bf16_f32, exp2 + v_rcp_f32 instead of expf + division (the result is rounded to bf16 next)
__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ void round_bf16_pair(float& a, float& b) {
    const uint32_t p = pack_bf16x2(a, b);
    a = bf16lo_to_f32(p);
    b = bf16hi_to_f32(p);
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, long V, int C, const float* __restrict__ stats_g, GnSums sums,
                                                       const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                                                       bf16_t* __restrict__ y, int apply_silu) {
    __shared__ float sst[GN_GROUPS][2];
    gn_stats_to_lds(sums, stats_g, C, sst);
    const float* const stats = &sst[0][0];
    const int vec_per_row = C >> 3, cg = C / GN_GROUPS;
    const long total = V * vec_per_row;
    if (256 % vec_per_row == 0) {
        // the grid stride is a multiple of the vectors per row, so a thread keeps ONE channel vector: gamma / beta / mean / rstd of its
        // 8 channels are loaded once, and the loop has no division (the per-vector 64-bit modulo and 8 group-index divisions were the cost)
        const int cv = threadIdx.x % vec_per_row;
        float gm[8], bt[8], mu[8], rs[8];
        const uint4 gw = *(const uint4*)(gamma + cv * 8), bw = *(const uint4*)(beta + cv * 8);
        const uint32_t gu[4] = {gw.x, gw.y, gw.z, gw.w}, bu[4] = {bw.x, bw.y, bw.z, bw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            gm[2 * i] = bf16lo_to_f32(gu[i]); gm[2 * i + 1] = bf16hi_to_f32(gu[i]);
            bt[2 * i] = bf16lo_to_f32(bu[i]); bt[2 * i + 1] = bf16hi_to_f32(bu[i]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int g = (cv * 8 + j) / cg;
            mu[j] = stats[2 * g]; rs[j] = stats[2 * g + 1];
        }
        float ga[8], gb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { ga[j] = rs[j] * gm[j]; gb[j] = bt[j] - mu[j] * ga[j]; }
        for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long)gridDim.x * 256) {
            const uint4 raw = *(const uint4*)(x + v * 8);
            const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w};
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a = fmaf(bf16lo_to_f32(u[i]), ga[2 * i], gb[2 * i]), b = fmaf(bf16hi_to_f32(u[i]), ga[2 * i + 1], gb[2 * i + 1]);
                if (apply_silu) {
                    round_bf16_pair(a, b);
                    a = silu_fast(a); b = silu_fast(b);
                }
                o[i] = pack_bf16x2(a, b);
            }
            *(uint4*)(y + v * 8) = uint4{o[0], o[1], o[2], o[3]};
        }
        return;
    }
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long)gridDim.x * 256) {
        const int cv = (int)(v % vec_per_row);
        const uint4 raw = *(const uint4*)(x + v * 8);
        const uint4 gw = *(const uint4*)(gamma + cv * 8), bw = *(const uint4*)(beta + cv * 8);
        const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w}, gu[4] = {gw.x, gw.y, gw.z, gw.w}, bu[4] = {bw.x, bw.y, bw.z, bw.w};
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g0 = (cv * 8 + 2 * i) / cg, g1 = (cv * 8 + 2 * i + 1) / cg;
            float a = round_bf16((bf16lo_to_f32(u[i]) - stats[2 * g0]) * stats[2 * g0 + 1] * bf16lo_to_f32(gu[i]) + bf16lo_to_f32(bu[i]));
            float b = round_bf16((bf16hi_to_f32(u[i]) - stats[2 * g1]) * stats[2 * g1 + 1] * bf16hi_to_f32(gu[i]) + bf16hi_to_f32(bu[i]));
            if (apply_silu) { a = silu_f(a); b = silu_f(b); }
            o[i] = pack_bf16x2(a, b);
        }
        *(uint4*)(y + v * 8) = uint4{o[0], o[1], o[2], o[3]};
    }
}

// SpatialNorm3D + SiLU.  conv_y / conv_b are 1x1x1 convolutions of the nearest-resized latent, and a 1x1x1 conv commutes
// with nearest-neighbour resizing, so they are evaluated once per LATENT voxel (yz, bz: [Tz*Hz*Wz][C] bf16, a tiny GEMM) and
// this kernel is a pure streaming pass: y = silu( GN(f) * yz[map(voxel)] + bz[map(voxel)] ).
__global__ __launch_bounds__(256) void spatialnorm_kernel(const bf16_t* __restrict__ f, int T, int H, int W, int C,
                                                          const float* __restrict__ stats_g, GnSums sums, const bf16_t* __restrict__ gamma,
                                                          const bf16_t* __restrict__ beta, const bf16_t* __restrict__ yz,
                                                          const bf16_t* __restrict__ bz, long ldz, int Tz, int Hz, int Wz,
                                                          bf16_t* __restrict__ y, int apply_silu) {
    __shared__ float sst[GN_GROUPS][2];
    gn_stats_to_lds(sums, stats_g, C, sst);
    const float* const stats = &sst[0][0];
    const int vec_per_row = C >> 3, cg = C / GN_GROUPS;
    const long V = (long)T * H * W, total = V * vec_per_row;
    const bool split_first = (T > 1) && (T & 1);
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long)gridDim.x * 256) {
        const int cv = (int)(v % vec_per_row);
        long vox = v / vec_per_row;
        const int w_ = (int)(vox % W); vox /= W;
        const int h_ = (int)(vox % H);
        const int t_ = (int)(vox / H);
        // F.interpolate(mode="nearest"): src = floor(dst * src_size / dst_size); first frame separately when T odd > 1
        int tz;
        if (split_first) tz = (t_ == 0) ? 0 : 1 + (int)(((long)(t_ - 1) * (Tz - 1)) / (T - 1));
        else tz = (int)(((long)t_ * Tz) / T);
        const int hz = (int)(((long)h_ * Hz) / H), wz = (int)(((long)w_ * Wz) / W);
        const long zrow = (((long)tz * Hz + hz) * Wz + wz) * ldz + cv * 8;
        const uint4 raw = *(const uint4*)(f + v * 8);
        const uint4 yr = *(const uint4*)(yz + zrow), br = *(const uint4*)(bz + zrow);
        const uint4 gw = *(const uint4*)(gamma + cv * 8), bw = *(const uint4*)(beta + cv * 8);
        const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w}, yu[4] = {yr.x, yr.y, yr.z, yr.w}, zu[4] = {br.x, br.y, br.z, br.w};
        const uint32_t gu[4] = {gw.x, gw.y, gw.z, gw.w}, bu[4] = {bw.x, bw.y, bw.z, bw.w};
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g0 = (cv * 8 + 2 * i) / cg, g1 = (cv * 8 + 2 * i + 1) / cg;
            const float n0 = round_bf16((bf16lo_to_f32(u[i]) - stats[2 * g0]) * stats[2 * g0 + 1] * bf16lo_to_f32(gu[i]) + bf16lo_to_f32(bu[i]));
            const float n1 = round_bf16((bf16hi_to_f32(u[i]) - stats[2 * g1]) * stats[2 * g1 + 1] * bf16hi_to_f32(gu[i]) + bf16hi_to_f32(bu[i]));
            float a = round_bf16(n0 * bf16lo_to_f32(yu[i])) + bf16lo_to_f32(zu[i]);
            float b = round_bf16(n1 * bf16hi_to_f32(yu[i])) + bf16hi_to_f32(zu[i]);
            if (apply_silu) { a = silu_f(round_bf16(a)); b = silu_f(round_bf16(b)); }
            o[i] = pack_bf16x2(a, b);
        }
        *(uint4*)(y + v * 8) = uint4{o[0], o[1], o[2], o[3]};
    }
}

// One workgroup per (t, h) row of the output: the generic kernel above spends most of its time in the 64-bit divisions that turn a
// flat vector index into (t, h, w, channel vector) and their nearest-neighbour sources, and in per-element group-index divisions and
// statistics loads (2.1 TB/s at [8,192,288,128]).  Here t, h and their sources are workgroup-uniform, w -> wz comes from a small LDS
// table filled once per workgroup, and a thread keeps ONE channel vector (256 % (C/8) == 0), so gamma / beta / mean / rstd of its 8
// channels are loaded once.
__global__ __launch_bounds__(256) void spatialnorm_row_kernel(const bf16_t* __restrict__ f, int T, int H, int W, int C,
                                                              const float* __restrict__ stats_g, GnSums sums, const bf16_t* __restrict__ gamma,
                                                              const bf16_t* __restrict__ beta, const bf16_t* __restrict__ yz,
                                                              const bf16_t* __restrict__ bz, long ldz, int Tz, int Hz, int Wz,
                                                              bf16_t* __restrict__ y, int apply_silu) {
    __shared__ int wzs[2048];
    __shared__ float sst[GN_GROUPS][2];
    gn_stats_to_lds(sums, stats_g, C, sst);
    const float* const stats = &sst[0][0];
    const int tid = threadIdx.x;
    const int h_ = blockIdx.x % H, t_ = blockIdx.x / H;
    const bool split_first = (T > 1) && (T & 1);
    int tz;
    if (split_first) tz = (t_ == 0) ? 0 : 1 + (int)(((long)(t_ - 1) * (Tz - 1)) / (T - 1));
    else tz = (int)(((long)t_ * Tz) / T);
    const int hz = (int)(((long)h_ * Hz) / H);
    for (int w = tid; w < W; w += 256) wzs[w] = (w * Wz) / W;          // W <= 2048, Wz <= W: 32-bit
    __syncthreads();
    const int vpr = C >> 3, cg = C / GN_GROUPS;
    const int cv = tid % vpr, wstep = 256 / vpr;
    float gm[8], bt[8], mu[8], rs[8];
    {
        const uint4 gw = *(const uint4*)(gamma + cv * 8), bw = *(const uint4*)(beta + cv * 8);
        const uint32_t gu[4] = {gw.x, gw.y, gw.z, gw.w}, bu[4] = {bw.x, bw.y, bw.z, bw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            gm[2 * i] = bf16lo_to_f32(gu[i]); gm[2 * i + 1] = bf16hi_to_f32(gu[i]);
            bt[2 * i] = bf16lo_to_f32(bu[i]); bt[2 * i + 1] = bf16hi_to_f32(bu[i]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int g = (cv * 8 + j) / cg;
            mu[j] = stats[2 * g]; rs[j] = stats[2 * g + 1];
        }
    }
    // GroupNorm affine folded per channel: (x - mu) rs g + b = x (rs g) + (b - mu rs g)
    float ga[8], gb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ga[j] = rs[j] * gm[j]; gb[j] = bt[j] - mu[j] * ga[j]; }
    const bf16_t* frow = f + ((long)(t_ * H + h_) * W) * C + cv * 8;
    bf16_t* yrow = y + ((long)(t_ * H + h_) * W) * C + cv * 8;
    const long zbase = ((long)tz * Hz + hz) * Wz;
    for (int w = tid / vpr; w < W; w += wstep) {
        const long zrow = (zbase + wzs[w]) * ldz + cv * 8;
        const uint4 raw = *(const uint4*)(frow + (long)w * C);
        const uint4 yr = *(const uint4*)(yz + zrow), br = *(const uint4*)(bz + zrow);
        const uint32_t u[4] = {raw.x, raw.y, raw.z, raw.w}, yu[4] = {yr.x, yr.y, yr.z, yr.w}, zu[4] = {br.x, br.y, br.z, br.w};
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float a = fmaf(bf16lo_to_f32(u[i]), ga[2 * i], gb[2 * i]), b = fmaf(bf16hi_to_f32(u[i]), ga[2 * i + 1], gb[2 * i + 1]);
            round_bf16_pair(a, b);                                   // GroupNorm output is a bf16 tensor in the reference
            a *= bf16lo_to_f32(yu[i]); b *= bf16hi_to_f32(yu[i]);
            round_bf16_pair(a, b);                                   // norm_f * conv_y(zq)
            a += bf16lo_to_f32(zu[i]); b += bf16hi_to_f32(zu[i]);
            if (apply_silu) {
                round_bf16_pair(a, b);                               // ... + conv_b(zq)
                a = silu_fast(a); b = silu_fast(b);
            }
            o[i] = pack_bf16x2(a, b);
        }
        *(uint4*)(yrow + (long)w * C) = uint4{o[0], o[1], o[2], o[3]};
    }
}

__global__ void avgpool_time_kernel(const bf16_t* __restrict__ x, int T, long HWC, bf16_t* __restrict__ y) {
    const int To = (T & 1) ? 1 + (T - 1) / 2 : T / 2;
    const long total = (long)To * HWC;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int to = (int)(i / HWC);
        const long e = i - (long)to * HWC;
        float v;
        if (T & 1) {
            if (to == 0) v = bf16_to_f32(x[e]);
            else v = 0.5f * (bf16_to_f32(x[(long)(2 * to - 1) * HWC + e]) + bf16_to_f32(x[(long)(2 * to) * HWC + e]));
        } else {
            v = 0.5f * (bf16_to_f32(x[(long)(2 * to) * HWC + e]) + bf16_to_f32(x[(long)(2 * to + 1) * HWC + e]));
        }
        y[i] = f32_to_bf16(v);
    }
}

__global__ void ncdhw_to_cl_kernel(const void* __restrict__ src, int src_fp32, int C, int Tt, int Ht, int Wt, int t0, int Tc, int h0,
                                   int Hc, int w0, int Wc, float scale, bf16_t* __restrict__ dst, int Cpad) {
    const long total = (long)Tc * Hc * Wc * Cpad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        long r = i / Cpad;
        const int w = (int)(r % Wc); r /= Wc;
        const int h = (int)(r % Hc);
        const int t = (int)(r / Hc);
        float v = 0.f;
        if (c < C) {
            const long s = (((long)c * Tt + (t0 + t)) * Ht + (h0 + h)) * Wt + (w0 + w);
            v = src_fp32 ? ((const float*)src)[s] : bf16_to_f32(((const bf16_t*)src)[s]);
            v *= scale;
        }
        dst[i] = f32_to_bf16(v);
    }
}

__global__ void cl_to_ncdhw_kernel(const bf16_t* __restrict__ src, long ld, int C, int T, int H, int W, void* __restrict__ dst,
                                   int dst_fp32, int Tt, int Ht, int Wt, int t0, int h0, int w0) {
    const long total = (long)C * T * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        long r = i / W;
        const int h = (int)(r % H); r /= H;
        const int t = (int)(r % T);
        const int c = (int)(r / T);
        const bf16_t v = src[(((long)t * H + h) * W + w) * ld + c];
        const long d = (((long)c * Tt + (t0 + t)) * Ht + (h0 + h)) * Wt + (w0 + w);
        if (dst_fp32) ((float*)dst)[d] = bf16_to_f32(v);
        else ((bf16_t*)dst)[d] = v;
    }
}

template <typename TT>
__device__ __forceinline__ float ldv(const TT* p);
template <> __device__ __forceinline__ float ldv<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldv<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
__device__ __forceinline__ void stv(float* p, float v) { *p = v; }
__device__ __forceinline__ void stv(bf16_t* p, float v) { *p = f32_to_bf16(v); }

template <typename TT>
__global__ void tile_blend_kernel(const TT* __restrict__ a, TT* __restrict__ b, int C, int T, int Ha, int Wa, int Hb, int Wb, int axis,
                                  int extent) {
    // axis 3: rows k < extent of b over width min(Wa, Wb);  axis 4: cols k < extent over height min(Ha, Hb)
    const int other = (axis == 3) ? min(Wa, Wb) : min(Ha, Hb);
    const long total = (long)C * T * extent * other;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int o = (int)(i % other);
        long r = i / other;
        const int k = (int)(r % extent); r /= extent;
        const long ct = r;                                 // c*T + t
        long ia, ib;
        if (axis == 3) { ia = (ct * Ha + (Ha - extent + k)) * Wa + o; ib = (ct * Hb + k) * Wb + o; }
        else { ia = (ct * Ha + o) * Wa + (Wa - extent + k); ib = (ct * Hb + o) * Wb + k; }
        const float wgt = (float)k / (float)extent;
        stv(b + ib, ldv(a + ia) * (1.f - wgt) + ldv(b + ib) * wgt);
    }
}

inline unsigned grid_for(long total, int block = 256) { return (unsigned)min((total + block - 1) / block, (long)256 * 16); }

// K ranges per tile for the 128 x 128 kernel: only when the plain launch would leave most CUs idle (fewer tiles than CUs) and the reduction
// is long; aims at ~2 workgroups per CU (the kernel waits on every stage: a second resident workgroup hides that), >= 8 K steps per range.
// TG_CONV_SPLITK=0 disables it (A/B runs, and the bitwise 4-wave-vs-128 test).  The w4 kernels are chosen first where they apply.
static int conv_ksplit(long M, int cout, int cout_pad, long nk, int n_cu) {
    if (!tg_knob(TG_KNOB_CONV_SPLITK) || cout != cout_pad || cout_pad % BN != 0 || cout > 512 || nk < 32) return 1;
    const long tiles = ((M + BM - 1) / BM) * (cout_pad / BN);
    if (tiles >= n_cu) return 1;
    long ks = (2L * n_cu) / tiles;            // FLOOR: tiles * ks must not exceed the 2 n_cu resident slots — the first version rounded up, and the most common
    if (ks > 8) ks = 8;                       // shape (88 tiles: 2 x 30 x 45 latent voxels x 512 channels) ran 528 workgroups = one full round + 16 stragglers
    if (ks > nk / 8) ks = nk / 8;
    return ks < 2 ? 1 : (int)ks;
}

}  // namespace

extern "C" int tg_conv3d_cl(const void* x, int T, int H, int W, int Cin, const void* cache, const void* w, const void* bias,
                            int cout, int cout_pad, int kt, int kh, int kw, int stride, int pad, int up, const int32_t* t_map,
                            const void* residual, void* y, long ldy, int To, int Ho, int Wo, const void* zeros, float* gn_partial,
                            float* splitk_ws, hipStream_t stream) {
    TG_REQUIRE(x && w && y && zeros, TG_ERR_ARG, "tg_conv3d_cl: null pointer");
    TG_REQUIRE(T > 0 && H > 0 && W > 0 && To > 0 && Ho > 0 && Wo > 0, TG_ERR_SHAPE, "tg_conv3d_cl: bad spatial shape");
    if (Cin == 8) {                    // the encoder's conv_in: 8-channel input (3 used), weights packed [128][96] with k = tap * 3 + channel
        TG_REQUIRE(cout == 128 && cout_pad == 128 && kt == 3 && kh == 3 && kw == 3 && stride == 1 && pad == 1 && up == 1 && !t_map && !residual && To == T &&
                   Ho == H && Wo == W, TG_ERR_SHAPE, "tg_conv3d_cl: Cin = 8 is the 3x3x3, stride-1, 128-output-channel input convolution only");
        TG_REQUIRE(tg_aligned16(x) && tg_aligned16(w) && (!cache || tg_aligned16(cache)) && (((uintptr_t)y) & 7) == 0 && ldy % 4 == 0, TG_ERR_ALIGN,
                   "tg_conv3d_cl: alignment");
        const long tiles_in = (long)To * ((Ho + H2_PH - 1) / H2_PH) * ((Wo + H2_PW - 1) / H2_PW), rows_in = ((long)To * Ho * Wo + BM - 1) / BM;
        TG_REQUIRE(tiles_in < (1L << 31) && (long)(T + 2) * H * W * 8 < (1L << 31), TG_ERR_SHAPE, "tg_conv3d_cl: too many tiles");
        TG_REQUIRE(!gn_partial || rows_in <= 4 * tiles_in, TG_ERR_SHAPE, "tg_conv3d_cl: GroupNorm sums need <= 4 rows of 128 voxels per 16 x 32 patch");
        // every patch writes gn_partial[patch * 64 ..]: the buffer (tg_conv3d_gn_partial_floats) has one row per 128 voxels, so there may not be more patches than rows
        TG_REQUIRE(!gn_partial || tiles_in <= rows_in, TG_ERR_SHAPE, "tg_conv3d_cl: GroupNorm sums need at least as many 128-voxel rows (%ld) as 16 x 32 patches (%ld)",
                   rows_in, tiles_in);
        ConvParams pi{(const bf16_t*)x, T, H, W, Cin, (const bf16_t*)cache, (const bf16_t*)w, (const bf16_t*)bias, cout, cout_pad, kt, kh, kw,
                      stride, pad, up, nullptr, nullptr, (bf16_t*)y, ldy, To, Ho, Wo, (const bf16_t*)zeros, gn_partial, 1, nullptr};
        hipLaunchKernelGGL(conv3d_in_kernel, dim3((unsigned)tiles_in), dim3(512), CI_LDS, stream, pi);
        TG_LAUNCH_CHECK("tg_conv3d_cl(in)");
        return TG_OK;
    }
    TG_REQUIRE(Cin % BK == 0 && (cout_pad % BN == 0 || (cout_pad < BN && cout_pad % 16 == 0)) && cout > 0 && cout <= cout_pad, TG_ERR_SHAPE,
               "tg_conv3d_cl: need Cin%%64==0 and cout_pad%%128==0 (or cout_pad in {16, 32, ..., 112}) (Cin=%d cout=%d cout_pad=%d)", Cin, cout, cout_pad);
    TG_REQUIRE(kt >= 1 && kt <= 3 && kh >= 1 && kh <= 3 && kw >= 1 && kw <= 3 && (stride == 1 || stride == 2) && (up == 1 || up == 2) &&
               pad >= 0 && pad <= 1, TG_ERR_SHAPE, "tg_conv3d_cl: unsupported kernel/stride/pad/up");
    TG_REQUIRE(tg_aligned16(x) && tg_aligned16(w) && tg_aligned16(zeros) && (!cache || tg_aligned16(cache)) && (((uintptr_t)y) & 1) == 0 &&
               (cout % 4 != 0 || ((((uintptr_t)y) & 7) == 0 && ldy % 4 == 0)), TG_ERR_ALIGN, "tg_conv3d_cl: alignment");
    ConvParams p{(const bf16_t*)x, T, H, W, Cin, (const bf16_t*)cache, (const bf16_t*)w, (const bf16_t*)bias, cout, cout_pad, kt, kh, kw,
                 stride, pad, up, t_map, (const bf16_t*)residual, (bf16_t*)y, ldy, To, Ho, Wo, (const bf16_t*)zeros, gn_partial, 1, nullptr};
    TG_REQUIRE(!gn_partial || (cout == cout_pad && cout % BN == 0 && (cout / GN_GROUPS) % 4 == 0), TG_ERR_SHAPE,
               "tg_conv3d_cl: fused GroupNorm sums need cout in {128, 256, 512, ...} (cout=%d)", cout);
    const long M = (long)To * Ho * Wo;
    const int halo_on = (int)tg_knob(TG_KNOB_CONV_HALO);    // 0 never, 1 (default) at launch scale, 2 whenever legal (cross-check tests, tg_debug_set)
    const int n_cu = tg_device_cus();
    if (cout_pad % BN != 0 && halo_on && cout <= 4 && Cin == 128 && kt == 3 && kh == 3 && kw == 3 && pad == 1 && stride == 1 && up == 1 && !t_map && !residual &&
        To == T && Ho == H && Wo == W && (long)(T + 2) * H * W * Cin < (1L << 31)) {
        // the decoder's conv_out: halo-tiled, weights resident in LDS (17.7 -> 3 ms per decode against the 128 x 16 GEMM-shaped tile below)
        const long h2tiles = (long)To * ((Ho + H2_PH - 1) / H2_PH) * ((Wo + H2_PW - 1) / H2_PW);
        if ((halo_on == 2 || h2tiles >= n_cu) && h2tiles < (1L << 31)) {
            const int lds = 2 * H2_HALO_BYTES + (cout + 1) * (27 * 128 * 2 + 64);
            TG_DYN_LDS((conv3d_halo_narrow_kernel<128, 3>), 2 * H2_HALO_BYTES + 5 * (27 * 128 * 2 + 64));
            hipLaunchKernelGGL((conv3d_halo_narrow_kernel<128, 3>), dim3((unsigned)h2tiles), dim3(512), lds, stream, p);
            TG_LAUNCH_CHECK("tg_conv3d_cl(halo narrow)");
            return TG_OK;
        }
    }
    if (cout_pad % BN != 0) {          // narrow output (conv_out): 128 voxels x 16 channels per workgroup
        const long tiles16 = ((M + BM - 1) / BM) * (cout_pad / 16);
        TG_REQUIRE(tiles16 < (1L << 31), TG_ERR_SHAPE, "tg_conv3d_cl: too many tiles");
        TG_DYN_LDS((conv3d_cl_kernel<1, 2, 1>), 2 * STAGE_BYTES);
        hipLaunchKernelGGL((conv3d_cl_kernel<1, 2, 1>), dim3((unsigned)tiles16), dim3(256), 2 * STAGE_BYTES, stream, p);
        TG_LAUNCH_CHECK("tg_conv3d_cl(n16)");
        return TG_OK;
    }
    const long tiles = ((M + BM - 1) / BM) * (cout_pad / BN);
    TG_REQUIRE(tiles < (1L << 31), TG_ERR_SHAPE, "tg_conv3d_cl: too many tiles");
    // 4-wave kernel: 256x256 tiles.  Its launch threshold was "at least 2 tiles per CU" (set from single-stream timings in round 2: a launch of 270 tiles pays
    // two rounds for 1.05); under the three tile streams a partial round is filled by the other tiles' launches, and what counts is the fill-path bytes per
    // flop — half of the 128 x 128 kernel's.  Swept in round 4 (decode / encode wall, same box): 2 n_cu 0.469 / 0.246 s, n_cu 0.461 / 0.233, n_cu/2 0.446 / 0.228,
    // n_cu/5 0.441 / 0.226, n_cu/8 0.436 / 0.220, n_cu/12 0.451 / 0.224 (there the 512-channel layers at 30 x 45 — 22 tiles — leave split-K).
    // TG_CONV_W4=0: never, 2: whenever legal
    const int w4 = (int)tg_knob(TG_KNOB_CONV_W4);
    // Cout = 128, 3x3 spatial taps, stride 1, no upsampling: the halo-tiled kernel.  Against the GEMM-shaped kernels on the 8 x 240 x 360 layers:
    // 128 -> 128: 0.70 vs 0.74 ms per launch; per clip 64 -> 128 (encoder conv_in) 13.3 vs 17.5 ms, 256 -> 128 45.6 vs 51.2 ms.  Why not more:
    // see the stage loop's comment (the fill does not overlap with the issuing wave's MFMAs).
    // TG_CONV_HALO: 0 never, 1 (default) at launch scale, 2 whenever legal (tests).
    {
        const long h2tiles = (long)To * ((Ho + H2_PH - 1) / H2_PH) * ((Wo + H2_PW - 1) / H2_PW);
        const long rows128 = (M + BM - 1) / BM;
        // Cout = 256 (two 128-channel slabs per patch) is legal but measured SLOWER than the 256 x 256 GEMM-shaped kernel (0.66 vs 0.55 ms on 256 -> 256 at
        // 8 x 120 x 180: each slab re-stages the halo and the weights dominate the fill either way): taken only when forced (TG_CONV_HALO=2, tests)
        if (halo_on && (cout == 128 || (cout == 256 && halo_on == 2)) && cout_pad == cout && kh == 3 && kw == 3 && pad == 1 && stride == 1 && up == 1 && !t_map && (kt == 1 || kt == 3) &&
            To == T && Ho == H && Wo == W && (halo_on == 2 || h2tiles >= n_cu) && h2tiles <= rows128 &&
            rows128 <= 4 * h2tiles && (long)(T + 2) * H * W * Cin < (1L << 31) && h2tiles < (1L << 31)) {
            TG_DYN_LDS((conv3d_halo2_kernel<8>), H2_LDS);
            // 8 waves (two per SIMD): 0.716 vs 0.730 ms (128 -> 128 at 8 x 240 x 360), 1.19 vs 1.26 ms (256 -> 128) against the one-wave-per-SIMD form of
            // the same kernel, same box (profiles/NOTES.md, round 4)
            hipLaunchKernelGGL(conv3d_halo2_kernel<8>, dim3((unsigned)(h2tiles * (cout / 128))), dim3(512), H2_LDS, stream, p);
            TG_LAUNCH_CHECK("tg_conv3d_cl(halo)");
            return TG_OK;
        }
    }
    if (w4 && (w4 == 2 || ((M + 255) / 256) * (cout / 256) >= n_cu / 8) && cout == cout_pad && cout % 256 == 0 && !t_map && M >= 1024 && (long)kt * kh * kw * (Cin / 64) >= 4 && H * up < 2048 && W * up < 2048 &&
        To < 512 && Ho < 2048 && Wo < 2048 && (long)kt * kh * kw * Cin < (1L << 21) && (long)(T + 2) * H * W * Cin < (1L << 31)) {
        TG_DYN_LDS((conv3d_w4_kernel<256>), CW_LDS);
        const long tiles4 = ((M + 255) / 256) * (cout / 256);
        hipLaunchKernelGGL(conv3d_w4_kernel<256>, dim3((unsigned)tiles4), dim3(256), CW_LDS, stream, p);
        TG_LAUNCH_CHECK("tg_conv3d_cl(w4)");
        return TG_OK;
    }
    // Cout = 128: the 512x128 variant (plain 3x3x3 / 1x3x3 convolutions only: 16 A pieces per wave are too many for the general address path)
    // (TG_CONV_W4 governs this variant too; measured: 128->128 layers 203 -> 187 ms per decode, 181 -> 162 ms per encode)
    if (w4 && (w4 == 2 || (M + 511) / 512 >= 2L * n_cu) && cout == 128 && cout_pad == 128 && !t_map && stride == 1 && up == 1 && M >= 2048 &&
        (long)kt * kh * kw * (Cin / 64) >= 4 && H < 2048 && W < 2048 && To < 512 && Ho < 2048 && Wo < 2048 && (long)kt * kh * kw * Cin < (1L << 21) &&
        (long)(T + 2) * H * W * Cin < (1L << 31)) {
        constexpr int LDS_N = 2 * (512 * 128 + 128 * 128);
        TG_DYN_LDS((conv3d_w4_kernel<128>), LDS_N);
        const long tiles5 = (M + 511) / 512;
        hipLaunchKernelGGL(conv3d_w4_kernel<128>, dim3((unsigned)tiles5), dim3(256), LDS_N, stream, p);
        TG_LAUNCH_CHECK("tg_conv3d_cl(w4n)");
        return TG_OK;
    }
    TG_DYN_LDS((conv3d_cl_kernel<2, 4, 4>), 2 * STAGE_BYTES);
    // split-K: the small-M layers (the 512-channel layers at 30 x 45 latent: 88 tiles for 256 CUs, each walking 216 K steps alone on its CU)
    p.ksplit = conv_ksplit(M, cout, cout_pad, (long)kt * kh * kw * (Cin / BK), n_cu);
    if (p.ksplit > 1) {
        TG_REQUIRE(splitk_ws, TG_ERR_ARG, "tg_conv3d_cl: this shape runs split-K (tg_conv3d_splitk_floats > 0) and needs the workspace");
        p.kpart = splitk_ws;
        hipLaunchKernelGGL((conv3d_cl_kernel<2, 4, 4>), dim3((unsigned)(tiles * p.ksplit)), dim3(256), 2 * STAGE_BYTES, stream, p);
        const dim3 rgrid((unsigned)((M + BM - 1) / BM), (unsigned)(cout / 128));
#define TG_SPLITK_REDUCE(KS)                                                                                                                \
    hipLaunchKernelGGL(conv_splitk_reduce_kernel<KS>, rgrid, dim3(256), 0, stream, (const float*)splitk_ws, p.ksplit, M, cout, (const bf16_t*)bias, \
                       (const bf16_t*)residual, (bf16_t*)y, ldy, gn_partial)
        switch (p.ksplit) {
            case 2: TG_SPLITK_REDUCE(2); break;
            case 4: TG_SPLITK_REDUCE(4); break;
            case 8: TG_SPLITK_REDUCE(8); break;
            default: TG_SPLITK_REDUCE(0); break;
        }
#undef TG_SPLITK_REDUCE
        TG_LAUNCH_CHECK("tg_conv3d_cl(split-K)");
        return TG_OK;
    }
    hipLaunchKernelGGL((conv3d_cl_kernel<2, 4, 4>), dim3((unsigned)tiles), dim3(256), 2 * STAGE_BYTES, stream, p);
    TG_LAUNCH_CHECK("tg_conv3d_cl");
    return TG_OK;
}

extern "C" long tg_conv3d_splitk_floats(int Cin, int cout, int cout_pad, int kt, int kh, int kw, int To, int Ho, int Wo) {
    int n_cu = 0, dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    const long M = (long)To * Ho * Wo;
    const int ks = conv_ksplit(M, cout, cout_pad, (long)kt * kh * kw * (Cin / BK), n_cu);
    return ks > 1 ? (long)ks * M * cout_pad : 0;
}

extern "C" long tg_groupnorm_partial_floats(long V, int C) {
    (void)C;
    return ((V + GN_ROWS_PER_BLOCK - 1) / GN_ROWS_PER_BLOCK) * 2 * GN_GROUPS;
}

extern "C" int tg_groupnorm_stats(const void* x, long V, int C, float eps, float* partial, float* stats, hipStream_t stream) {
    TG_REQUIRE(x && partial && stats, TG_ERR_ARG, "tg_groupnorm_stats: null pointer");
    TG_REQUIRE(V > 0 && C % 64 == 0 && C <= 2048 && 256 % (C / 8) == 0, TG_ERR_SHAPE, "tg_groupnorm_stats: need C in {64,128,256,512,1024,2048} (C=%d)", C);
    TG_REQUIRE(tg_aligned16(x), TG_ERR_ALIGN, "tg_groupnorm_stats: alignment");
    const int nblocks = (int)((V + GN_ROWS_PER_BLOCK - 1) / GN_ROWS_PER_BLOCK);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nblocks), dim3(256), 0, stream, (const bf16_t*)x, V, C, partial);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(GN_GROUPS), dim3(256), 0, stream, (const float*)partial, nblocks, V, C, eps, stats);
    TG_LAUNCH_CHECK("tg_groupnorm_stats");
    return TG_OK;
}

// ---- nearest x2 upsampling + 3x3 convolution as four 2x2 phase convolutions on the LOW-resolution input (see the header) ----
static bool up2_subpixel_shape_ok(int T, int H, int W, int Cin, int cout, int n_cu) {
    const long M = (long)T * H * W;
    return Cin % 64 == 0 && cout % 256 == 0 && M >= 1024 && 4 * ((M + 255) / 256) * (cout / 256) >= n_cu / 8 && 4L * (Cin / 64) >= 4 && H < 1024 && W < 1024 && T < 512 &&
           4L * Cin < (1L << 21) && (long)(T + 2) * H * W * Cin < (1L << 31) && 4L * M * cout < (1L << 40);
}

extern "C" long tg_conv3d_up2_subpixel_ok(int T, int H, int W, int Cin, int cout) {
    return up2_subpixel_shape_ok(T, H, W, Cin, cout, tg_device_cus()) && tg_knob(TG_KNOB_CONV_W4) != 0 ? 1 : 0;
}

extern "C" long tg_conv3d_up2_subpixel_gn_floats(int T, int H, int W) { return 4 * ((((long)T * H * W) + BM - 1) / BM) * 2 * GN_GROUPS; }

extern "C" int tg_conv3d_up2_subpixel(const void* x, int T, int H, int W, int Cin, const void* w_phases, const void* bias, int cout, void* y, long ldy,
                                      int time_x2, const void* zeros, float* gn_partial, hipStream_t stream) {
    TG_REQUIRE(time_x2 == 0 || time_x2 == 1, TG_ERR_ARG, "tg_conv3d_up2_subpixel: time_x2 is 0 or 1");
    TG_REQUIRE(x && w_phases && y && zeros, TG_ERR_ARG, "tg_conv3d_up2_subpixel: null pointer");
    TG_REQUIRE(T > 0 && H > 0 && W > 0, TG_ERR_SHAPE, "tg_conv3d_up2_subpixel: bad spatial shape");
    TG_REQUIRE(up2_subpixel_shape_ok(T, H, W, Cin, cout, tg_device_cus()) && tg_knob(TG_KNOB_CONV_W4) != 0, TG_ERR_SHAPE,
               "tg_conv3d_up2_subpixel: shape outside the 256 x 256 kernel's range (T=%d H=%d W=%d Cin=%d cout=%d): ask tg_conv3d_up2_subpixel_ok first", T, H, W, Cin, cout);
    TG_REQUIRE(tg_aligned16(x) && tg_aligned16(w_phases) && tg_aligned16(zeros) && (((uintptr_t)y) & 7) == 0 && ldy % 4 == 0 && ldy >= cout, TG_ERR_ALIGN,
               "tg_conv3d_up2_subpixel: alignment");
    TG_REQUIRE(!gn_partial || (cout / GN_GROUPS) % 4 == 0, TG_ERR_SHAPE, "tg_conv3d_up2_subpixel: fused GroupNorm sums need cout in {256, 512, ...}");
    const long M = (long)T * H * W, tiles = ((M + 255) / 256) * (cout / 256);
    TG_DYN_LDS((conv3d_w4_kernel<256>), CW_LDS);
    // ONE launch for the four phases (virtual tile = phase * tiles + tile; the kernel derives the phase's padding — rows yl + a - (1 - py): phase 0 reads (yl - 1, yl),
    // phase 1 (yl, yl + 1), columns likewise — its output offsets, weight block and GroupNorm rows): four launches of 675 tiles were four 2.64-round launches with 16-stage
    // tiles; stand-alone at 8 x 120 x 180 x 256 567 us against 877 for the 9-tap kernel, merged see profiles/NOTES.md G
    ConvParams p{(const bf16_t*)x, T, H, W, Cin, nullptr, (const bf16_t*)w_phases, (const bf16_t*)bias, cout, cout, 1, 2, 2,
                 1, 1, 1, nullptr, nullptr, (bf16_t*)y, ldy, T, H, W, (const bf16_t*)zeros, gn_partial, 1, nullptr,
                 0, 2, 0, 0, time_x2 && T > 1 ? (T % 2 ? 1 : 2) : 0, 4};
    hipLaunchKernelGGL(conv3d_w4_kernel<256>, dim3((unsigned)(4 * tiles)), dim3(256), CW_LDS, stream, p);
    TG_LAUNCH_CHECK("tg_conv3d_up2_subpixel");
    return TG_OK;
}

extern "C" long tg_conv3d_gn_partial_floats(int To, int Ho, int Wo) {
    return (((long)To * Ho * Wo + BM - 1) / BM) * 2 * GN_GROUPS;
}

extern "C" int tg_groupnorm_finalize(const float* partial, long V, int C, float eps, float* stats, hipStream_t stream) {
    TG_REQUIRE(partial && stats, TG_ERR_ARG, "tg_groupnorm_finalize: null pointer");
    TG_REQUIRE(V > 0 && C % GN_GROUPS == 0, TG_ERR_SHAPE, "tg_groupnorm_finalize: bad shape");
    const int nblocks = (int)((V + BM - 1) / BM);           // one block of sums per 128-voxel conv tile row
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(GN_GROUPS), dim3(256), 0, stream, partial, nblocks, V, C, eps, stats);
    TG_LAUNCH_CHECK("tg_groupnorm_finalize");
    return TG_OK;
}

// the statistics source of a norm pass: ready-made (mean, rstd) pairs, or the sums a convolution's epilogue (or tg_groupnorm_reduce) left
static int gn_sums_arg(const char* who, const float* stats, const void* sums, long nsum, int sums_f64, long V, float eps, GnSums& g) {
    TG_REQUIRE((stats != nullptr) != (sums != nullptr), TG_ERR_ARG, "%s: exactly one of stats / sums", who);
    TG_REQUIRE(!sums || (nsum > 0 && nsum <= GN_FOLD_ROWS && tg_aligned16(sums)), TG_ERR_SHAPE, "%s: 1..%d rows of sums (got %ld)", who, GN_FOLD_ROWS, nsum);
    g = GnSums{sums, (int)nsum, sums_f64, V, eps};
    return TG_OK;
}

extern "C" long tg_groupnorm_reduce_rows(long nrows) { return nrows <= GN_FOLD_ROWS ? 0 : (nrows < 64L * GN_FOLD_ROWS ? (nrows + GN_FOLD_ROWS - 1) / GN_FOLD_ROWS : 64); }

extern "C" int tg_groupnorm_reduce(const float* partial, long nrows, double* out, hipStream_t stream) {
    TG_REQUIRE(partial && out, TG_ERR_ARG, "tg_groupnorm_reduce: null pointer");
    const long r = tg_groupnorm_reduce_rows(nrows);
    TG_REQUIRE(r > 0, TG_ERR_SHAPE, "tg_groupnorm_reduce: %ld rows need no reduction (<= %d are read by the norm pass itself)", nrows, GN_FOLD_ROWS);
    hipLaunchKernelGGL(gn_reduce_kernel, dim3((unsigned)r), dim3(256), 0, stream, partial, nrows, out);
    TG_LAUNCH_CHECK("tg_groupnorm_reduce");
    return TG_OK;
}

extern "C" int tg_groupnorm_silu_ex(const void* x, long V, int C, const float* stats, const void* sums, long nsum, int sums_f64, float eps,
                                    const void* gamma, const void* beta, void* y, int apply_silu, hipStream_t stream) {
    TG_REQUIRE(x && gamma && beta && y, TG_ERR_ARG, "tg_groupnorm_silu: null pointer");
    TG_REQUIRE(V > 0 && C % 64 == 0, TG_ERR_SHAPE, "tg_groupnorm_silu: bad shape");
    TG_REQUIRE(tg_aligned16(x) && tg_aligned16(y) && tg_aligned16(gamma) && tg_aligned16(beta), TG_ERR_ALIGN, "tg_groupnorm_silu: alignment");
    GnSums g;
    if (const int rc = gn_sums_arg("tg_groupnorm_silu", stats, sums, nsum, sums_f64, V, eps, g)) return rc;
    hipLaunchKernelGGL(gn_apply_kernel, dim3(grid_for(V * (C / 8))), dim3(256), 0, stream, (const bf16_t*)x, V, C, stats, g, (const bf16_t*)gamma,
                       (const bf16_t*)beta, (bf16_t*)y, apply_silu);
    TG_LAUNCH_CHECK("tg_groupnorm_silu");
    return TG_OK;
}

extern "C" int tg_groupnorm_silu(const void* x, long V, int C, const float* stats, const void* gamma, const void* beta, void* y,
                                 int apply_silu, hipStream_t stream) {
    return tg_groupnorm_silu_ex(x, V, C, stats, nullptr, 0, 0, 0.f, gamma, beta, y, apply_silu, stream);
}

extern "C" int tg_spatialnorm_silu_ex(const void* f, int T, int H, int W, int C, const float* stats, const void* sums, long nsum, int sums_f64, float eps,
                                      const void* gamma, const void* beta, const void* yz, const void* bz, long ldz, int Tz, int Hz, int Wz, void* y,
                                      int apply_silu, hipStream_t stream) {
    TG_REQUIRE(f && gamma && beta && yz && bz && y, TG_ERR_ARG, "tg_spatialnorm_silu: null pointer");
    TG_REQUIRE(T > 0 && H > 0 && W > 0 && C % 64 == 0 && Tz > 0 && Hz > 0 && Wz > 0 && ldz >= C && ldz % 8 == 0, TG_ERR_SHAPE,
               "tg_spatialnorm_silu: bad shape");
    TG_REQUIRE(tg_aligned16(f) && tg_aligned16(y) && tg_aligned16(yz) && tg_aligned16(bz), TG_ERR_ALIGN, "tg_spatialnorm_silu: alignment");
    GnSums g;
    if (const int rc = gn_sums_arg("tg_spatialnorm_silu", stats, sums, nsum, sums_f64, (long)T * H * W, eps, g)) return rc;
    const long total = (long)T * H * W * (C / 8);
    if (256 % (C / 8) == 0 && W <= 2048 && (long)T * H < (1L << 30) && T * H >= 512)   // one workgroup per row: needs >= 2 rows per CU
        hipLaunchKernelGGL(spatialnorm_row_kernel, dim3((unsigned)(T * H)), dim3(256), 0, stream, (const bf16_t*)f, T, H, W, C, stats, g,
                           (const bf16_t*)gamma, (const bf16_t*)beta, (const bf16_t*)yz, (const bf16_t*)bz, ldz, Tz, Hz, Wz, (bf16_t*)y, apply_silu);
    else
        hipLaunchKernelGGL(spatialnorm_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16_t*)f, T, H, W, C, stats, g,
                           (const bf16_t*)gamma, (const bf16_t*)beta, (const bf16_t*)yz, (const bf16_t*)bz, ldz, Tz, Hz, Wz, (bf16_t*)y, apply_silu);
    TG_LAUNCH_CHECK("tg_spatialnorm_silu");
    return TG_OK;
}

extern "C" int tg_spatialnorm_silu(const void* f, int T, int H, int W, int C, const float* stats, const void* gamma, const void* beta,
                                   const void* yz, const void* bz, long ldz, int Tz, int Hz, int Wz, void* y, int apply_silu,
                                   hipStream_t stream) {
    return tg_spatialnorm_silu_ex(f, T, H, W, C, stats, nullptr, 0, 0, 0.f, gamma, beta, yz, bz, ldz, Tz, Hz, Wz, y, apply_silu, stream);
}

extern "C" int tg_avgpool_time(const void* x, int T, long HW, int C, void* y, hipStream_t stream) {
    TG_REQUIRE(x && y, TG_ERR_ARG, "tg_avgpool_time: null pointer");
    TG_REQUIRE(T > 0 && HW > 0 && C > 0, TG_ERR_SHAPE, "tg_avgpool_time: bad shape");
    const int To = (T & 1) ? 1 + (T - 1) / 2 : T / 2;
    hipLaunchKernelGGL(avgpool_time_kernel, dim3(grid_for((long)To * HW * C)), dim3(256), 0, stream, (const bf16_t*)x, T, HW * C, (bf16_t*)y);
    TG_LAUNCH_CHECK("tg_avgpool_time");
    return TG_OK;
}

extern "C" int tg_ncdhw_to_cl(const void* src, int src_fp32, int C, int Tt, int Ht, int Wt, int t0, int Tc, int h0, int Hc, int w0, int Wc,
                              float scale, void* dst, int Cpad, hipStream_t stream) {
    TG_REQUIRE(src && dst, TG_ERR_ARG, "tg_ncdhw_to_cl: null pointer");
    TG_REQUIRE(C > 0 && Cpad >= C && t0 >= 0 && h0 >= 0 && w0 >= 0 && t0 + Tc <= Tt && h0 + Hc <= Ht && w0 + Wc <= Wt && Tc > 0 && Hc > 0 && Wc > 0,
               TG_ERR_SHAPE, "tg_ncdhw_to_cl: window outside the source tensor");
    hipLaunchKernelGGL(ncdhw_to_cl_kernel, dim3(grid_for((long)Tc * Hc * Wc * Cpad)), dim3(256), 0, stream, src, src_fp32, C, Tt, Ht, Wt, t0, Tc,
                       h0, Hc, w0, Wc, scale, (bf16_t*)dst, Cpad);
    TG_LAUNCH_CHECK("tg_ncdhw_to_cl");
    return TG_OK;
}

extern "C" int tg_cl_to_ncdhw(const void* src, long ld, int C, int T, int H, int W, void* dst, int dst_fp32, int Tt, int Ht, int Wt,
                              int t0, int h0, int w0, hipStream_t stream) {
    TG_REQUIRE(src && dst, TG_ERR_ARG, "tg_cl_to_ncdhw: null pointer");
    TG_REQUIRE(C > 0 && ld >= C && t0 >= 0 && h0 >= 0 && w0 >= 0 && t0 + T <= Tt && h0 + H <= Ht && w0 + W <= Wt, TG_ERR_SHAPE,
               "tg_cl_to_ncdhw: window outside the destination tensor");
    hipLaunchKernelGGL(cl_to_ncdhw_kernel, dim3(grid_for((long)C * T * H * W)), dim3(256), 0, stream, (const bf16_t*)src, ld, C, T, H, W, dst,
                       dst_fp32, Tt, Ht, Wt, t0, h0, w0);
    TG_LAUNCH_CHECK("tg_cl_to_ncdhw");
    return TG_OK;
}

extern "C" int tg_tile_blend(const void* a, void* b, int is_fp32, int C, int T, int Ha, int Wa, int Hb, int Wb, int axis, int extent,
                             hipStream_t stream) {
    TG_REQUIRE(a && b, TG_ERR_ARG, "tg_tile_blend: null pointer");
    TG_REQUIRE((axis == 3 || axis == 4) && extent > 0 && extent <= (axis == 3 ? min(Ha, Hb) : min(Wa, Wb)), TG_ERR_SHAPE,
               "tg_tile_blend: bad axis/extent");
    const long total = (long)C * T * extent * (axis == 3 ? min(Wa, Wb) : min(Ha, Hb));
    if (is_fp32)
        hipLaunchKernelGGL(tile_blend_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream, (const float*)a, (float*)b, C, T, Ha, Wa, Hb, Wb, axis, extent);
    else
        hipLaunchKernelGGL(tile_blend_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, stream, (const bf16_t*)a, (bf16_t*)b, C, T, Ha, Wa, Hb, Wb, axis, extent);
    TG_LAUNCH_CHECK("tg_tile_blend");
    return TG_OK;
}
