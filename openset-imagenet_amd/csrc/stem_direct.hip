// Direct kernels for the ResNet stem (conv1: 7x7 / stride 2 / pad 3, 3 -> 64 channels; torchvision.models.resnet50 under reference
// openset_imagenet/model.py:17, differentiated at train.py:138): forward, weight gradient, and the weight gradient with the backward of
// bn1 -> ReLU -> max-pool fused into its operand loader. Split from conv_igemm.hip in round 3; shares its helpers through conv_common.h.
#include "conv_common.h"

using namespace osi_conv;

namespace {

// ======================================================================================================
// Stem forward, direct form: 7x7 / stride 2 / pad 3 convolution of a 3-channel image (Cout = 64).
// The implicit-GEMM kernel above spends 224 K slots on 147 real taps (4th channel + 7 padded taps: 34 % of its MFMAs multiply zeros)
// and re-derives the tap geometry of every gathered float4. With only 3 input channels the whole receptive field of an output
// tile is small: a workgroup owns 8 x 16 output pixels, stages their 21 x 37 x 3 input patch (9.3 KB) and the complete weight
// matrix W^T[148][64] (37 KB, staged once per workgroup and reused for `tiles_per_wg` tiles) in LDS, and then runs 74 K steps of
// v_mfma_f32_32x32x2_f32 whose A operand is a strided ds_read_b32 gather straight out of the patch (per-lane pixel base + a
// compile-time tap offset) and whose B operand is a row of W^T: no global loads, no barriers and no address arithmetic inside the
// K loop, K = 148 instead of 224. Same exact-fp32 FMA chains, another summation order than the implicit-GEMM form.
// The BatchNorm partials are one (mean, M2) pair per tile of 128 pixels, the contract osi_bn_finalize_stats expects with
// rows_per_block = 128 (every tile is full: the launcher only takes this path when Ho % 8 == 0 and Wo % 16 == 0).
// ======================================================================================================
constexpr int SD_TH = 8, SD_TW = 16, SD_PH = 2 * SD_TH + 5, SD_PW = 2 * SD_TW + 5, SD_K = 148;
__host__ __device__ constexpr int sd_off(int k) {          // patch offset (floats) of tap index k = (r * 7 + s) * 3 + c
    const int kk = k < 147 ? k : 146;                       // K padding: any valid address, its weight row is zero
    return (kk / 21) * SD_PW * 3 + kk % 21;
}
__global__ __launch_bounds__(256, 3) void k_stem_fwd_direct(const float* __restrict__ x4, const float* __restrict__ wpacked,
                                                           float* __restrict__ y, float* __restrict__ pmean, float* __restrict__ pm2,
                                                           int B, int H, int W, int Ho, int Wo, int tiles_x, int tiles_y, int ntiles,
                                                           int tiles_per_wg) {
    __shared__ float sW[SD_K * 64];
    __shared__ float sP[SD_PH * SD_PW * 3 + 1];
    __shared__ float sS[2 * 64 * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l31 = lane & 31;
    // W^T[k][cout] from the packed [64][56 taps][4] weights: lanes walk the couts (conflict-free LDS stores)
    for (int i = tid; i < 64 * 50; i += 256) {
        const int cout = i & 63, tap = i >> 6;              // taps 49 holds the zero row k = 147 (tap 49, c = 0)
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (tap < 49) v = ld4(wpacked + cout * 224 + tap * 4);
        if (tap < 49) { sW[(tap * 3) * 64 + cout] = v[0]; sW[(tap * 3 + 1) * 64 + cout] = v[1]; sW[(tap * 3 + 2) * 64 + cout] = v[2]; }
        else sW[147 * 64 + cout] = 0.f;
    }
    // A operand address of K step ks = pixel base + sd_off(2 ks) + h * (sd_off(2 ks + 1) - sd_off(2 ks)); the difference is 1 inside a
    // patch row, SD_PW * 3 - 20 across a row wrap and 0 on the padded last step: three per-lane bases, every step's tap offset is an
    // immediate of its ds_read (nothing per step for the compiler to precompute and spill)
    const float* pA[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pix = wm * 64 + i * 32 + l31, py = pix >> 4, px = pix & 15;
        const int base = (2 * py * SD_PW + 2 * px) * 3;
        pA[i][0] = sP + base; pA[i][1] = sP + base + h; pA[i][2] = sP + base + h * (SD_PW * 3 - 20);
    }
    const float* pB = sW + wn * 32 + l31 + h * 64;
    const int tile_end = min(ntiles, (int)(blockIdx.x + 1) * tiles_per_wg);
    // input patch of a tile: branch-free buffer loads into registers (out-of-image pixels: sentinel offset -> zeros = the padding);
    // the NEXT tile's loads are issued before the K loop and stored to LDS behind it
    constexpr int NPX = (SD_PH * SD_PW + 255) / 256;
    const __amdgpu_buffer_rsrc_t rxb = make_rsrc(x4, (int)((size_t)B * H * W * 16));
    f32x3 rpx[NPX];      // 12-byte loads (see k_stem_wgrad_direct)
    uint32_t prel[NPX];
    int ppy[NPX], ppx[NPX];
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        const int i = tid + j * 256;
        ppy[j] = i / SD_PW; ppx[j] = i - ppy[j] * SD_PW;
        prel[j] = (uint32_t)((ppy[j] * W + ppx[j]) * 16);
        if (i >= SD_PH * SD_PW) ppy[j] = 1 << 20;          // never inside the image
    }
    auto gload = [&](int tile) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int iy0 = 2 * ty * SD_TH - 3, ix0 = 2 * tx * SD_TW - 3;
        const int pbase = __builtin_amdgcn_readfirstlane(((b * H + iy0) * W + ix0) * 16);   // may be negative at the image border: only used where the pixel is valid
        uint32_t poff[NPX];
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int ok = ((unsigned)(iy0 + ppy[j]) < (unsigned)H) & ((unsigned)(ix0 + ppx[j]) < (unsigned)W);
            poff[j] = ok ? (uint32_t)(pbase + (int)prel[j]) : OOB;
        }
#pragma unroll
        for (int j = 0; j < NPX; ++j) rpx[j] = bld3(rxb, poff[j], 0);
    };
    auto sstore = [&]() {
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int i = tid + j * 256;
            if (i < SD_PH * SD_PW) { sP[i * 3] = rpx[j][0]; sP[i * 3 + 1] = rpx[j][1]; sP[i * 3 + 2] = rpx[j][2]; }
        }
    };
    if ((int)blockIdx.x * tiles_per_wg < tile_end) gload(blockIdx.x * tiles_per_wg);
    for (int tile = blockIdx.x * tiles_per_wg; tile < tile_end; ++tile) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int oy0 = ty * SD_TH, ox0 = tx * SD_TW;
        __syncthreads();                                    // every wave is done with the previous tile's patch (and sW is complete)
        sstore();
        __syncthreads();
        if (tile + 1 < tile_end) gload(tile + 1);
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        // 74 K steps in groups of SD_G, software-pipelined by hand: the LDS reads of group g + 1 are issued before the MFMAs of group g
        // (two register sets); the scheduling barriers keep the compiler from hoisting every read of the tile to the top (it does,
        // and spills the accumulators) or sinking them next to their MFMA (exposing the LDS latency 74 times).
        constexpr int SD_G = 4, SD_NG = (SD_K / 2 + SD_G - 1) / SD_G;
        float ca0[SD_G], ca1[SD_G], cb[SD_G];
        auto lds_group = [&](int g, float (&a0)[SD_G], float (&a1)[SD_G], float (&bv)[SD_G]) {
#pragma unroll
            for (int j = 0; j < SD_G; ++j) {
                const int ks = g * SD_G + j;
                if (ks < SD_K / 2) {
                    const int o0 = sd_off(2 * ks), dl = sd_off(2 * ks + 1) - o0;
                    const int sel = dl == 1 ? 1 : (dl == 0 ? 0 : 2);
                    a0[j] = pA[0][sel][o0]; a1[j] = pA[1][sel][o0];
                    bv[j] = pB[ks * 128];
                }
            }
        };
        lds_group(0, ca0, ca1, cb);
#pragma unroll
        for (int g = 0; g < SD_NG; ++g) {
            float na0[SD_G], na1[SD_G], nb[SD_G];
            if (g + 1 < SD_NG) lds_group(g + 1, na0, na1, nb);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < SD_G; ++j) {
                if (g * SD_G + j < SD_K / 2) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca0[j], cb[j], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca1[j], cb[j], acc[1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < SD_NG) {
#pragma unroll
                for (int j = 0; j < SD_G; ++j) { ca0[j] = na0[j]; ca1[j] = na1[j]; cb[j] = nb[j]; }
            }
        }
        // output: 32 lanes of a half-wave write 128 contiguous bytes of one pixel
        const int col = wn * 32 + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int pix = wm * 64 + i * 32 + acc_row(rr, lane);
                y[(((size_t)b * Ho + oy0 + (pix >> 4)) * Wo + ox0 + (pix & 15)) * 64 + col] = acc[i][rr];
            }
        if (pmean) {   // (mean, M2) of the tile's 128 pixels per channel: this wave's 64 rows, then the two wave rows merged
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) s += acc[i][rr];
            s += __shfl_xor(s, 32, 64);
            const float mu = s * (1.f / 64.f);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) { const float dlt = acc[i][rr] - mu; q += dlt * dlt; }
            q += __shfl_xor(q, 32, 64);
            if (lane < 32) { sS[(wm * 64 + col) * 2] = mu; sS[(wm * 64 + col) * 2 + 1] = q; }
            __syncthreads();
            if (tid < 64) {
                float cn = 64.f, cm = sS[tid * 2], cs = sS[tid * 2 + 1];
                chan_merge(cn, cm, cs, 64.f, sS[(64 + tid) * 2], sS[(64 + tid) * 2 + 1]);
                pmean[(size_t)tile * 64 + tid] = cm;
                pm2[(size_t)tile * 64 + tid] = cs;
            }
        }
    }
}

// ======================================================================================================
// Stem weight gradient, direct form: dW[c][k] = sum over pixels dY[pix][c] * patch(pix)[k]  (c < 64 channels, k = (r * 7 + s) * 3 + ch < 147).
// The implicit-GEMM weight-gradient kernel walks K = pixels with a 224-wide padded tap axis (147 real columns) and streams dY once
// per column tile (4x). Here a workgroup of four waves owns the whole 64 x 160 tile in registers (wave w = channels 16 w .. 16 w + 15,
// ten 16 x 16 accumulator tiles of v_mfma_f32_16x16x4_f32), loops over its share of 8 x 16-pixel output tiles, and per tile stages the
// A rows (32 KB: the tile's rows are contiguous in NHWC) and the 21 x 37 x 3 input patch in LDS. K step = 4 pixels: A = one
// ds_read_b32, B = ten gathers patch[pixel base + tap offset of the lane's column] with the pixel base an IMMEDIATE (the pixels of a K
// step are neighbours in a tile row): no address arithmetic, no global load and no barrier inside the 32 K steps of a tile; the next
// tile's global loads are issued before them (branch-free buffer loads into registers) and land in LDS behind them.
// Each workgroup writes ONE partial [64][147] slab; k_slab_reduce adds them in workgroup order (fixed: bitwise reproducible).
//   SRC 0:  dY is a tensor [B][Ho][Wo][64] in memory.
//   SRC 2:  dY is never written to memory: it is the BatchNorm backward (g - c1 - xhat c2) gamma invstd of the max-pool backward g of
//           the pooled gradient (g[pix][c] = sum of the pooled gradients of the windows whose stored arg-max is pix and whose maximum
//           was positive), built per tile from the pooled gradient + arg-max bytes staged in LDS and the prefetched rows of the
//           stem's conv output y — the stem tail bn1 -> ReLU -> max-pool differentiated inside the operand loader.
// (A third form — conv1's gradient assembled from the moment matrices sum g (x) patch, sum y (x) patch, sum patch, so that the matrix
// work runs beside the BatchNorm reductions — was built, exact (5e-7 .. 2e-6 of fp64) and NOT faster: the forward-time moment GEMM
// cost the forward pass what the backward gained, 34.87 vs 34.96 ms. Removed; profiles/NOTES_r03.md.)
// ======================================================================================================
struct StemOuterP {
    const float* a;            // SRC 0: dY, SRC 2: the stem's conv output y; [B][Ho][Wo][64]
    const float* x4;           // [B][H][W][4]
    float* slab;               // [groups][64][147]
    const float* pg;           // SRC 2: pooled gradient [B][Hp][Wp][64]
    const uint32_t* pidx;      // SRC 2: arg-max bytes   [B][Hp][Wp][16] (bit 7 = ReLU gate, osi_bn_relu_maxpool_fwd)
    const float *gamma, *mean, *invstd, *dgamma, *dbeta;   // SRC 2: the stem BatchNorm's parameters, statistics and reductions
    float inv_m;               // SRC 2: 1 / (B * Ho * Wo)
    int B, H, W, Ho, Wo, Hp, Wp, tiles_x, tiles_y, ntiles, tiles_per_wg;
};
constexpr int SO_QH = SD_TH / 2 + 1, SO_QW = SD_TW / 2 + 1;      // pooled windows that reach an 8 x 16 tile: 5 x 9
template <int SRC>
// SRC 2 holds 32 more prefetch registers (the rows of y): two waves per SIMD without spills measure the same as three with (396 vs 400 us)
__global__ __launch_bounds__(256, SRC == 2 ? 2 : 3) void k_stem_outer(StemOuterP p) {
    __shared__ __attribute__((aligned(16))) float sA[128 * 64];
    // The patch (read by the K loop) and the staged pooled tile (read only while the A tile is being built) share their memory:
    // 46 KB per workgroup instead of 56, i.e. three resident workgroups per CU instead of two; the patch is stored after build_a.
    constexpr int SO_PFLOATS = SD_PH * SD_PW * 3 + 3, SO_GFLOATS = SO_QH * SO_QW * (64 + 16);
    __shared__ __attribute__((aligned(16))) float sU[SRC != 0 ? (SO_PFLOATS > SO_GFLOATS ? SO_PFLOATS : SO_GFLOATS) : SO_PFLOATS];
    float* const sP = sU;
    float* const sG = sU;
    uint32_t* const sI = reinterpret_cast<uint32_t*>(sU + SO_QH * SO_QW * 64);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lk = lane >> 4;
    // scalars of the parameter block copied to locals: the lambdas below would otherwise pin the whole struct in scratch memory
    const int H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo, Hp = p.Hp, Wp = p.Wp, tiles_x = p.tiles_x, tiles_y = p.tiles_y;
    const float* pA = sA + lk * 64 + wave * 16 + l15;
    const float* pB[10];
#pragma unroll
    for (int nt = 0; nt < 10; ++nt) pB[nt] = sP + sd_off(nt * 16 + l15) + lk * 6;   // pixel 4 ks + lk: lk pixels to the right = 2 lk input pixels
    constexpr int NPX = (SD_PH * SD_PW + 255) / 256;       // patch pixels per thread (4)
    constexpr int NQ = (SO_QH * SO_QW * 16 + 255) / 256;   // SRC 1: pooled float4 / index words per thread (3)
    f32x4 ra[8];                     // the tile's rows of dY (SRC 0) / of the stem's conv output y (SRC 2)
    f32x4 rq[SRC != 0 ? NQ : 1];     // SRC 2: pooled gradient float4
    uint32_t ri[SRC != 0 ? NQ : 1];  //        and arg-max words of the windows that reach the tile
    f32x3 rpx[NPX];      // 12-byte loads: the 4th channel of the NHWC4 image is padding, and a dead 4th register would be reused by the
                         // allocator while the load is in flight (a vmcnt wait in front of the K loop)
    const __amdgpu_buffer_rsrc_t rab = make_rsrc(p.a, (int)((size_t)p.B * Ho * Wo * 64 * 4));
    const __amdgpu_buffer_rsrc_t rqb = make_rsrc(SRC != 0 ? p.pg : p.x4, SRC != 0 ? (int)((size_t)p.B * Hp * Wp * 64 * 4) : 16);
    const __amdgpu_buffer_rsrc_t rib = make_rsrc(SRC != 0 ? (const float*)p.pidx : p.x4, SRC != 0 ? (int)((size_t)p.B * Hp * Wp * 64) : 16);
    const __amdgpu_buffer_rsrc_t rxb = make_rsrc(p.x4, (int)((size_t)p.B * H * W * 16));
    // per-thread offsets relative to the tile origin, computed once: the per-tile part is scalar
    uint32_t prel[NPX];
    int ppy[NPX], ppx[NPX];
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        const int i = tid + j * 256;
        ppy[j] = i / SD_PW; ppx[j] = i - ppy[j] * SD_PW;
        prel[j] = (uint32_t)((ppy[j] * W + ppx[j]) * 16);
        if (i >= SD_PH * SD_PW) ppy[j] = 1 << 20;          // never inside the image
    }
    int qy[NQ], qx[NQ];                                    // SRC 2: pooled pixel of this thread's float4 (relative to the tile's first window)
    if (SRC != 0) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int i = tid + j * 256, q = i >> 4;
            qy[j] = q / SO_QW; qx[j] = q - qy[j] * SO_QW;
            if (i >= SO_QH * SO_QW * 16) qy[j] = 1 << 20;
        }
    }
    // SRC 2: BatchNorm-backward coefficients of this thread's four channels (c4 = tid & 15):
    //   A = (g - c1 - xhat c2) gamma invstd,  xhat = (y - mean) invstd,  c1 = dbeta / M,  c2 = dgamma / M
    f32x4 bc1 = {0, 0, 0, 0}, bc2 = bc1, bmu = bc1, bis = bc1, bgs = bc1;
    if (SRC == 2) {
        const int c = (tid & 15) * 4;
        bc1 = ld4(p.dbeta + c) * p.inv_m; bc2 = ld4(p.dgamma + c) * p.inv_m;
        bmu = ld4(p.mean + c); bis = ld4(p.invstd + c); bgs = ld4(p.gamma + c) * bis;
    }
    auto gload = [&](int tile) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int oy0 = ty * SD_TH, ox0 = tx * SD_TW, iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
        // wave-uniform by construction (functions of the tile index); said explicitly, or hipcc wraps every load that takes the scalar
        // offset in a waterfall loop
        const int pbase = __builtin_amdgcn_readfirstlane(((b * H + iy0) * W + ix0) * 16);   // may be negative at the border: only used where valid
        uint32_t poff[NPX];
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int ok = ((unsigned)(iy0 + ppy[j]) < (unsigned)H) & ((unsigned)(ix0 + ppx[j]) < (unsigned)W);
            poff[j] = ok ? (uint32_t)(pbase + (int)prel[j]) : OOB;
        }
        {
            const uint32_t abase = (uint32_t)__builtin_amdgcn_readfirstlane(((b * Ho + oy0) * Wo + ox0) * 64 * 4);
#pragma unroll
            for (int j = 0; j < 8; ++j)      // row j of the tile: 16 pixels x 16 float4 = this thread's float4 tid of a 4 KB row
                ra[j] = bld4(rab, (uint32_t)(tid * 16), abase + (uint32_t)(j * Wo * 256));
        }
        if (SRC != 0) {
            // windows ho = oy0 / 2 .. oy0 / 2 + 4, wo = ox0 / 2 .. ox0 / 2 + 8 (the last row / column may lie outside the pooled grid)
            const int q0y = oy0 >> 1, q0x = ox0 >> 1;
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int ok = ((unsigned)(q0y + qy[j]) < (unsigned)Hp) & ((unsigned)(q0x + qx[j]) < (unsigned)Wp);
                const uint32_t qpix = (uint32_t)((b * Hp + q0y + qy[j]) * Wp + q0x + qx[j]);
                rq[j] = bld4(rqb, ok ? (qpix * 64 + (uint32_t)(tid & 15) * 4) * 4 : OOB, 0);
                ri[j] = __builtin_amdgcn_raw_buffer_load_b32(rib, ok ? (qpix * 16 + (uint32_t)(tid & 15)) * 4 : OOB, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < NPX; ++j) rpx[j] = bld3(rxb, poff[j], 0);
    };
    auto sstore_patch = [&]() {
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int i = tid + j * 256;
            if (i < SD_PH * SD_PW) { sP[i * 3] = rpx[j][0]; sP[i * 3 + 1] = rpx[j][1]; sP[i * 3 + 2] = rpx[j][2]; }
        }
    };
    auto sstore_a = [&](int tile) {
        if (SRC == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(sA + (j * 256 + tid) * 4) = ra[j];
        } else {
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int i = tid + j * 256;
                if (i < SO_QH * SO_QW * 16) { *reinterpret_cast<f32x4*>(sG + i * 4) = rq[j]; sI[i] = ri[j]; }
            }
        }
    };
    // SRC 2: A tile from the staged pooled tile. Thread = (pixel column px = tid >> 4, channel quad c4 = tid & 15), rows py = 0..7.
    // Pixel (h, w) belongs to windows ho in {h >> 1, (h + 1) >> 1}, wo likewise, at window position r = h - (2 ho - 1), s = w - (2 wo - 1);
    // same visiting order as k_maxpool_bwd / pool_gather (bn.hip), so the sums are the ones the unfused route produces.
    auto build_a = [&](int tile) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y;
        const int oy0 = ty * SD_TH, ox0 = tx * SD_TW, q0y = oy0 >> 1, q0x = ox0 >> 1;
        const int px = tid >> 4, c4 = tid & 15, w = ox0 + px;
#pragma unroll
        for (int py = 0; py < 8; ++py) {
            const int hh = oy0 + py;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const int ho = dy == 0 ? hh >> 1 : (hh + 1) >> 1;
                if (dy == 1 && ho == (hh >> 1)) continue;
                if (ho >= Hp) continue;
                const int r = hh - (2 * ho - 1);
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int wo = dx == 0 ? w >> 1 : (w + 1) >> 1;
                    if (dx == 1 && wo == (w >> 1)) continue;
                    if (wo >= Wp) continue;
                    const int sx = w - (2 * wo - 1);
                    const int qi = ((ho - q0y) * SO_QW + (wo - q0x)) * 16 + c4;
                    const uint32_t id = sI[qi];
                    const f32x4 g = *reinterpret_cast<const f32x4*>(sG + qi * 4);
                    const uint32_t me = (uint32_t)(r * 3 + sx) | 0x80u;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (((id >> (8 * k)) & 0xffu) == me) acc[k] += g[k];
                }
            }
            if (SRC == 2) {     // BatchNorm backward of the scattered gradient, with the y row this thread prefetched
                const f32x4 xh = (ra[py] - bmu) * bis;
                acc = (acc - bc1 - xh * bc2) * bgs;
            }
            *reinterpret_cast<f32x4*>(sA + ((py * 16 + px) * 16 + c4) * 4) = acc;
        }
    };
    f32x4acc acc[10];
#pragma unroll
    for (int nt = 0; nt < 10; ++nt) acc[nt] = f32x4acc{0.f, 0.f, 0.f, 0.f};
    const int tile0 = blockIdx.x * p.tiles_per_wg, tile_end = min(p.ntiles, tile0 + p.tiles_per_wg);
    if (tile0 < tile_end) {
        gload(tile0);
        if (SRC == 0) { sstore_patch(); sstore_a(tile0); __syncthreads(); }
        else { sstore_a(tile0); __syncthreads(); build_a(tile0); __syncthreads(); sstore_patch(); __syncthreads(); }
        for (int tile = tile0; tile < tile_end; ++tile) {
            if (tile + 1 < tile_end) gload(tile + 1);
            // 32 K steps (4 pixels each), the 11 LDS reads of step ks + 1 issued before the 10 MFMAs of step ks (two register sets)
            struct Ops { float a, b[10]; };
            auto lds_ops = [&](int ks, Ops& o) {
                o.a = pA[ks * 256];
                const int cb = (2 * (ks >> 2) * SD_PW + 8 * (ks & 3)) * 3;
#pragma unroll
                for (int nt = 0; nt < 10; ++nt) o.b[nt] = pB[nt][cb];
            };
            auto mma_ops = [&](const Ops& o) {
#pragma unroll
                for (int nt = 0; nt < 10; ++nt) {
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a, o.b[nt], acc[nt], 0, 0, 0);
                }
            };
            Ops o0, o1;
            lds_ops(0, o0);
#pragma unroll
            for (int ks = 0; ks < 32; ks += 2) {
                lds_ops(ks + 1, o1);
                __builtin_amdgcn_sched_barrier(0);
                mma_ops(o0);
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 2 < 32) lds_ops(ks + 2, o0);
                __builtin_amdgcn_sched_barrier(0);
                mma_ops(o1);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();                                // every wave is done reading this tile
            if (tile + 1 < tile_end) {
                if (SRC == 0) { sstore_patch(); sstore_a(tile + 1); __syncthreads(); }
                else { sstore_a(tile + 1); __syncthreads(); build_a(tile + 1); __syncthreads(); sstore_patch(); __syncthreads(); }
            }
        }
    }
    // partial of this workgroup. C/D of 16x16x4: column (tap) = lane & 15, row (channel) = (lane >> 4) * 4 + reg
    float* out = p.slab + (size_t)blockIdx.x * (64 * 147);
#pragma unroll
    for (int nt = 0; nt < 10; ++nt) {
        const int col = nt * 16 + l15;
        if (col < 147) {
#pragma unroll
            for (int e = 0; e < 4; ++e) out[(wave * 16 + lk * 4 + e) * 147 + col] = acc[nt][e];
        }
    }
}

}  // namespace

namespace osi_conv {

bool stem_direct_geometry(const osi_conv_desc* d) {
    return g_osi_tuning.stem_direct && conv_desc_ok(d) && conv_is_stem(d) && d->Cout == 64 && d->stride == 2 && d->pad == 3 &&
           d->Ho % SD_TH == 0 && d->Wo % SD_TW == 0;
}

int launch_stem_fwd_direct(const osi_conv_desc* d, const float* x4, const float* wpacked, float* y, float* pmean, float* pm2, int ntiles,
                           hipStream_t st) {
    const int per = 7;      // tiles per workgroup: the weight matrix is staged once per workgroup
    hipLaunchKernelGGL(k_stem_fwd_direct, dim3(osi_cdiv(ntiles, per)), dim3(256), 0, st, x4, wpacked, y, pmean, pm2, d->B, d->H, d->W, d->Ho,
                       d->Wo, d->Wo / SD_TW, d->Ho / SD_TH, ntiles, per);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

}  // namespace osi_conv

namespace {
static bool stem_wgrad_direct_ok(const osi_conv_desc* d) { return stem_direct_geometry(d); }
// direct stem weight gradient: split plan, workspace
static void stem_wgrad_plan(const osi_conv_desc* d, int& ntiles, int& per, int& groups) {
    ntiles = d->B * (d->Ho / SD_TH) * (d->Wo / SD_TW);
    // persistent workgroups, three per CU (41 KB of LDS each): one partial slab per workgroup
    const int want = 3 * hw_cus();     // the hardware's CUs, not the fwd / dgrad plan's ("tail_cus" must not resize this grid or its slab)
    per = osi_cdiv(ntiles, want);
    if (per < 1) per = 1;
    groups = osi_cdiv(ntiles, per);
}
static StemOuterP stem_outer_params(const osi_conv_desc* d, const float* a, const float* x4, float* slab, int ntiles, int per) {
    StemOuterP q{};
    q.a = a; q.x4 = x4; q.slab = slab;
    q.B = d->B; q.H = d->H; q.W = d->W; q.Ho = d->Ho; q.Wo = d->Wo;
    q.Hp = (d->Ho + 2 - 3) / 2 + 1; q.Wp = (d->Wo + 2 - 3) / 2 + 1;
    q.tiles_x = d->Wo / SD_TW; q.tiles_y = d->Ho / SD_TH; q.ntiles = ntiles; q.tiles_per_wg = per;
    return q;
}
}  // namespace

extern "C" {

size_t osi_stem_wgrad_direct_workspace(const osi_conv_desc* d) {
    if (!d || !stem_wgrad_direct_ok(d)) return 0;
    int ntiles, per, groups;
    stem_wgrad_plan(d, ntiles, per, groups);
    return (size_t)groups * 64 * 147 * sizeof(float);
}
int osi_stem_wgrad_direct(const osi_conv_desc* d, const float* dy, const float* x4, float* dw_krsc3, void* ws, size_t ws_bytes,
                          osi_stream_t stream) {
    OSI_REQUIRE(d && dy && x4 && dw_krsc3 && ws);
    if (!stem_wgrad_direct_ok(d)) return OSI_ERR_ARG;
    int ntiles, per, groups;
    stem_wgrad_plan(d, ntiles, per, groups);
    OSI_REQUIRE(ws_bytes >= (size_t)groups * 64 * 147 * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    const StemOuterP q = stem_outer_params(d, dy, x4, (float*)ws, ntiles, per);
    hipLaunchKernelGGL((k_stem_outer<0>), dim3(groups), dim3(256), 0, st, q);
    OSI_LAUNCH_CHECK();
    const size_t n4 = 64 * 147 / 4;
    return launch_slab_reduce((const float*)ws, dw_krsc3, n4, n4, groups, st);
}

// Stem weight gradient with the BatchNorm + ReLU + max-pool backward fused into its operand loader (k_stem_outer<2>): the
// 112 x 112 x 64 gradient dY is built per tile in LDS from the pooled gradient, the arg-max bytes and the stem's conv output and is
// never written to memory — no apply pass (0.8 GB of traffic) between the BatchNorm reductions and the weight gradient.
size_t osi_stem_wgrad_fused_workspace(const osi_conv_desc* d) { return osi_stem_wgrad_direct_workspace(d); }
int osi_stem_wgrad_fused(const osi_conv_desc* d, const float* gpool, const void* pool_idx, const float* y, const float* x4,
                         const float* gamma, const float* mean, const float* invstd, const float* dgamma, const float* dbeta,
                         float* dw_krsc3, void* ws, size_t ws_bytes, osi_stream_t stream) {
    OSI_REQUIRE(d && gpool && pool_idx && y && x4 && gamma && mean && invstd && dgamma && dbeta && dw_krsc3 && ws);
    if (!stem_wgrad_direct_ok(d)) return OSI_ERR_ARG;
    int ntiles, per, groups;
    stem_wgrad_plan(d, ntiles, per, groups);
    OSI_REQUIRE(ws_bytes >= (size_t)groups * 64 * 147 * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    StemOuterP q = stem_outer_params(d, y, x4, (float*)ws, ntiles, per);
    q.pg = gpool; q.pidx = (const uint32_t*)pool_idx;
    q.gamma = gamma; q.mean = mean; q.invstd = invstd; q.dgamma = dgamma; q.dbeta = dbeta;
    q.inv_m = 1.0f / (float)((size_t)d->B * d->Ho * d->Wo);
    hipLaunchKernelGGL((k_stem_outer<2>), dim3(groups), dim3(256), 0, st, q);
    OSI_LAUNCH_CHECK();
    const size_t n4 = 64 * 147 / 4;
    return launch_slab_reduce((const float*)ws, dw_krsc3, n4, n4, groups, st);
}

}  // extern "C"
