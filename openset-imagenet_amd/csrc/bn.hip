// BatchNorm2d (training mode) on NHWC fp32 activations, fused with ReLU / residual add / ReLU-mask.
//
// Replaces the 53 torch.nn.BatchNorm2d + 49 ReLU + 16 residual adds that torchvision's ResNet-50 runs under
// openset_imagenet/model.py:37 with model.train() set at openset_imagenet/train.py:125 (reference).
// Semantics kept: normalise with the biased batch variance, update running_var with the unbiased one,
// momentum 0.1 convention running = (1-m)*running + m*batch, eps inside the sqrt.
//
// All kernels are HBM-bound streams over [M = B*H*W][C] with C contiguous: 16-byte accesses, each lane owns
// 4 consecutive channels, row-lanes of a workgroup walk the rows, cross-row-lane reduction through LDS.
// Statistics use shifted sums per workgroup (shift = first row of the chunk) and a weighted Chan merge of
// the per-workgroup (mean, M2), so there is no E[x^2]-E[x]^2 cancellation in fp32.
#include "osi_common.h"

namespace {

constexpr int NT = 256;
constexpr int OSI_BN_GROUPS = 32;  // level-1 groups of the two-level statistics finalisation

struct RowSplit { int CV, RL, G; };  // float4 columns per pass, row lanes, column groups
__host__ __device__ inline RowSplit row_split(int C) {
    RowSplit s;
    int c4 = C / 4;
    s.CV = c4 < NT ? c4 : NT;
    s.RL = NT / s.CV;
    s.G = (c4 + s.CV - 1) / s.CV;
    return s;
}

// ---- statistics -------------------------------------------------------------------------------------
// partials are stored channel-major ([C][P]) so that the finalising wave of a channel reads them coalesced
__global__ __launch_bounds__(NT) void k_bn_stats_partial(const float* __restrict__ y, int M, int C, int rows_per_blk,
                                                        float* __restrict__ pmean, float* __restrict__ pm2) {
    const int P = gridDim.x;
    __shared__ f32x4 red[2][NT];
    const RowSplit sp = row_split(C);
    const int tid = threadIdx.x;
    const int cv = tid % sp.CV, rl = tid / sp.CV;
    const int r0 = blockIdx.x * rows_per_blk;
    const int r1 = min(M, r0 + rows_per_blk);
    const float n = (float)(r1 - r0);
    const bool active = rl < sp.RL;
    for (int g = 0; g < sp.G; ++g) {
        const int c4 = g * sp.CV + cv;
        const bool colok = active && c4 * 4 < C;
        f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0}, shift = {0, 0, 0, 0};
        if (colok) {
            const f32x4* base = reinterpret_cast<const f32x4*>(y) + c4;
            const size_t ld = C / 4;
            shift = base[(size_t)r0 * ld];
            int r = r0 + rl;
            for (; r + 3 * sp.RL < r1; r += 4 * sp.RL) {
                f32x4 v0 = base[(size_t)r * ld], v1 = base[(size_t)(r + sp.RL) * ld];
                f32x4 v2 = base[(size_t)(r + 2 * sp.RL) * ld], v3 = base[(size_t)(r + 3 * sp.RL) * ld];
                v0 -= shift; v1 -= shift; v2 -= shift; v3 -= shift;
                s1 += (v0 + v1) + (v2 + v3);
                s2 += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
            }
            for (; r < r1; r += sp.RL) {
                f32x4 v = base[(size_t)r * ld] - shift;
                s1 += v; s2 += v * v;
            }
        }
        red[0][tid] = s1; red[1][tid] = s2;
        __syncthreads();
        if (colok && rl == 0) {
            for (int k = 1; k < sp.RL; ++k) { s1 += red[0][k * sp.CV + cv]; s2 += red[1][k * sp.CV + cv]; }
            f32x4 mean = shift + s1 / n;
            f32x4 m2 = s2 - s1 * s1 / n;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pmean[(size_t)(c4 * 4 + k) * P + blockIdx.x] = mean[k];
                pm2[(size_t)(c4 * 4 + k) * P + blockIdx.x] = m2[k];
            }
        }
        __syncthreads();
    }
}

// One wave per channel: every lane Chan-merges its share of the P per-workgroup (n, mean, M2) triples in a fixed order,
// then a fixed xor-shuffle tree merges the 64 lanes. Deterministic, single pass over the partials.
__device__ __forceinline__ void bn_finish(int c, float mean, float m2, int M, const float* gamma, const float* beta, float eps,
                                          float momentum, float* running_mean, float* running_var, float* mean_out,
                                          float* invstd_out, float* scale_out, float* shift_out) {
    float var = m2 / (float)M;
    float invstd = 1.0f / sqrtf(var + eps);
    mean_out[c] = mean; invstd_out[c] = invstd;
    float sc = invstd * gamma[c];
    scale_out[c] = sc; shift_out[c] = beta[c] - mean * sc;
    if (running_mean) {
        float unb = M > 1 ? m2 / (float)(M - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
    }
}
// Row-major partials [P][C] as written by the conv-forward epilogue (osi_conv_fwd_bnstats): one workgroup = 16 channels x 16
// partial lanes; every lane Chan-merges partials lane, lane+16, ... in order, lane 0 then merges the 16 lane results in order.
__global__ __launch_bounds__(NT) void k_bn_stats_final_rows(const float* __restrict__ pmean, const float* __restrict__ pm2, int P,
                                                           int rows_per_blk, int M, int C, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float momentum,
                                                           float* running_mean, float* running_var, float* __restrict__ mean_out,
                                                           float* __restrict__ invstd_out, float* __restrict__ scale_out,
                                                           float* __restrict__ shift_out) {
    // division-free weighted form of the Chan merge: mean = sum n_b mean_b / M ; M2 = sum [ M2_b + n_b (mean_b - mean)^2 ]
    __shared__ float red[16][16];
    __shared__ float smean[16];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const bool ok = c < C;
    const int last = P - 1;
    const float nfull = (float)rows_per_blk, nlast = (float)(M - last * rows_per_blk);
    float a0 = 0.f, a1 = 0.f;
    if (ok) {
        int b = pl;
        for (; b + 16 < P; b += 32) {
            a0 += (b == last ? nlast : nfull) * pmean[(size_t)b * C + c];
            a1 += (b + 16 == last ? nlast : nfull) * pmean[(size_t)(b + 16) * C + c];
        }
        if (b < P) a0 += (b == last ? nlast : nfull) * pmean[(size_t)b * C + c];
    }
    red[pl][cl] = a0 + a1;
    __syncthreads();
    if (pl == 0) {
        float s = 0.f;
        for (int k = 0; k < 16; ++k) s += red[k][cl];
        smean[cl] = s / (float)M;
    }
    __syncthreads();
    const float mean = smean[cl];
    a0 = 0.f; a1 = 0.f;
    if (ok) {
        int b = pl;
        for (; b + 16 < P; b += 32) {
            const float d0 = pmean[(size_t)b * C + c] - mean, d1 = pmean[(size_t)(b + 16) * C + c] - mean;
            a0 += pm2[(size_t)b * C + c] + (b == last ? nlast : nfull) * d0 * d0;
            a1 += pm2[(size_t)(b + 16) * C + c] + (b + 16 == last ? nlast : nfull) * d1 * d1;
        }
        if (b < P) { const float d0 = pmean[(size_t)b * C + c] - mean; a0 += pm2[(size_t)b * C + c] + (b == last ? nlast : nfull) * d0 * d0; }
    }
    __syncthreads();
    red[pl][cl] = a0 + a1;
    __syncthreads();
    if (pl == 0 && ok) {
        float m2 = 0.f;
        for (int k = 0; k < 16; ++k) m2 += red[k][cl];
        bn_finish(c, mean, m2, M, gamma, beta, eps, momentum, running_mean, running_var, mean_out, invstd_out, scale_out, shift_out);
    }
}
// Single-launch finalisation for mid-sized P (round 3): 16 channels x 64 partial lanes per workgroup, ONE pass over the partials with
// four independent loads in flight per lane. Shifted sums around a pivot pv close to the batch mean (the average of 64 tile means, so
// that S2 - S1^2 / M does not cancel):  S1 = sum n_b (mean_b - pv),  S2 = sum [ M2_b + n_b (mean_b - pv)^2 ],
// mean = pv + S1 / M,  M2 = S2 - S1^2 / M. Fixed summation order (lane stride, then lanes in order): bitwise reproducible.
constexpr int WPL = 64;   // partial lanes of the wide finalisers
__device__ __forceinline__ float wide_reduce(float (*red)[16], float v, int pl, int cl) {
    // 64 lane values of channel cl -> their sum in lane order, returned to pl == 0
    red[pl][cl] = v;
    __syncthreads();
    float t = 0.f;
    if (pl < 4) {
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[pl * 16 + k][cl];
    }
    __syncthreads();
    if (pl < 4) red[pl][cl] = t;
    __syncthreads();
    if (pl == 0) t = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    __syncthreads();
    return t;
}
__global__ __launch_bounds__(16 * WPL) void k_bn_stats_final_wide(const float* __restrict__ pmean, const float* __restrict__ pm2, int P,
                                                                 int rows_per_blk, int M, int C, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float eps, float momentum,
                                                                 float* running_mean, float* running_var, float* __restrict__ mean_out,
                                                                 float* __restrict__ invstd_out, float* __restrict__ scale_out,
                                                                 float* __restrict__ shift_out) {
    __shared__ float red[WPL][16];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = min(blockIdx.x * 16 + cl, C - 1);   // lanes past C repeat the last channel and do not write
    const bool ok = blockIdx.x * 16 + cl < C;
    const int last = P - 1;
    const float nfull = (float)rows_per_blk, nlast = (float)(M - last * rows_per_blk);
    const float* pa = pmean + c;
    const float* pb = pm2 + c;
    // pivot = the average of the first 64 tile means (one per partial lane; lanes past P repeat the last tile). A single tile's mean —
    // the first tile is the top-left corner of image 0: border pixels — can sit many sigma from the batch mean, and S2 - S1^2 / M then
    // cancels (relative error ~ eps * (pivot offset / sigma)^2); the average of 64 tiles is within a fraction of sigma of it whatever
    // one tile does. Costs one more reduction of values the lanes load anyway.
    const float pv = wide_reduce(red, pa[(size_t)min(pl, last) * C], pl, cl) * (1.f / WPL);
    if (pl == 0) red[0][cl] = pv;          // wide_reduce returns the sum to pl == 0 only: hand it to the channel's other lanes
    __syncthreads();
    const float pvv = red[0][cl];
    __syncthreads();
    float s1a = 0.f, s1b = 0.f, s2a = 0.f, s2b = 0.f;
    int b = pl;
    for (; b + 3 * WPL < P; b += 4 * WPL) {
        const float m0 = pa[(size_t)b * C], m1 = pa[(size_t)(b + WPL) * C], m2_ = pa[(size_t)(b + 2 * WPL) * C], m3 = pa[(size_t)(b + 3 * WPL) * C];
        const float q0 = pb[(size_t)b * C], q1 = pb[(size_t)(b + WPL) * C], q2 = pb[(size_t)(b + 2 * WPL) * C], q3 = pb[(size_t)(b + 3 * WPL) * C];
        const float d0 = m0 - pvv, d1 = m1 - pvv, d2 = m2_ - pvv, d3 = m3 - pvv;
        const float n3 = b + 3 * WPL == last ? nlast : nfull;     // only the last of the four can be the ragged tile
        s1a += nfull * d0; s1b += nfull * d1; s1a += nfull * d2; s1b += n3 * d3;
        s2a += q0 + nfull * d0 * d0; s2b += q1 + nfull * d1 * d1; s2a += q2 + nfull * d2 * d2; s2b += q3 + n3 * d3 * d3;
    }
    for (; b < P; b += WPL) {
        const float d = pa[(size_t)b * C] - pvv, n = b == last ? nlast : nfull;
        s1a += n * d; s2a += pb[(size_t)b * C] + n * d * d;
    }
    const float S1 = wide_reduce(red, s1a + s1b, pl, cl);
    const float S2 = wide_reduce(red, s2a + s2b, pl, cl);
    if (pl == 0 && ok) {
        const float mean = pvv + S1 / (float)M;
        const float m2 = fmaxf(S2 - S1 * S1 / (float)M, 0.f);
        bn_finish(c, mean, m2, M, gamma, beta, eps, momentum, running_mean, running_var, mean_out, invstd_out, scale_out, shift_out);
    }
}
// Level 1 of the two-level finalisation: group s (blockIdx.y) reduces row tiles [s*Pc, (s+1)*Pc) of 16 channels to one
// (mean, M2) pair per channel, division-free, written channel-major [C][S] for k_bn_stats_final.
__global__ __launch_bounds__(NT) void k_bn_stats_group(const float* __restrict__ pmean, const float* __restrict__ pm2, int P, int Pc,
                                                      int rows_per_blk, int M, int C, int S, float* __restrict__ gmean,
                                                      float* __restrict__ gm2) {
    // one pass of shifted sums, four row tiles in flight per lane; see k_bn_stats_final_wide
    __shared__ float red[2][16][16];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = min(blockIdx.x * 16 + cl, C - 1), s = blockIdx.y;
    const bool ok = blockIdx.x * 16 + cl < C;
    const int b0 = s * Pc, b1 = min(P, b0 + Pc), last = P - 1;
    const float nfull = (float)rows_per_blk, nlast = (float)(M - last * rows_per_blk);
    const float ng = (float)(min(M, b1 * rows_per_blk) - b0 * rows_per_blk);
    const float* pa = pmean + c;
    const float* pb = pm2 + c;
    // pivot = the average of the group's first 16 tile means (see k_bn_stats_final_wide)
    red[0][pl][cl] = pa[(size_t)min(b0 + pl, b1 - 1) * C];
    __syncthreads();
    float pv = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) pv += red[0][k][cl];
    pv *= 1.f / 16.f;
    __syncthreads();
    float s1a = 0.f, s1b = 0.f, s2a = 0.f, s2b = 0.f;
    int b = b0 + pl;
    for (; b + 48 < b1; b += 64) {
        const float m0 = pa[(size_t)b * C], m1 = pa[(size_t)(b + 16) * C], m2_ = pa[(size_t)(b + 32) * C], m3 = pa[(size_t)(b + 48) * C];
        const float q0 = pb[(size_t)b * C], q1 = pb[(size_t)(b + 16) * C], q2 = pb[(size_t)(b + 32) * C], q3 = pb[(size_t)(b + 48) * C];
        const float d0 = m0 - pv, d1 = m1 - pv, d2 = m2_ - pv, d3 = m3 - pv;
        const float n3 = b + 48 == last ? nlast : nfull;
        s1a += nfull * d0; s1b += nfull * d1; s1a += nfull * d2; s1b += n3 * d3;
        s2a += q0 + nfull * d0 * d0; s2b += q1 + nfull * d1 * d1; s2a += q2 + nfull * d2 * d2; s2b += q3 + n3 * d3 * d3;
    }
    for (; b < b1; b += 16) {
        const float d = pa[(size_t)b * C] - pv, n = b == last ? nlast : nfull;
        s1a += n * d; s2a += pb[(size_t)b * C] + n * d * d;
    }
    red[0][pl][cl] = s1a + s1b; red[1][pl][cl] = s2a + s2b;
    __syncthreads();
    if (pl == 0 && ok) {
        float S1 = 0.f, S2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { S1 += red[0][k][cl]; S2 += red[1][k][cl]; }
        gmean[(size_t)c * S + s] = pv + S1 / ng;
        gm2[(size_t)c * S + s] = fmaxf(S2 - S1 * S1 / ng, 0.f);
    }
}
__global__ __launch_bounds__(NT) void k_bn_stats_final(const float* __restrict__ pmean, const float* __restrict__ pm2, int P,
                                                      int rows_per_blk, int M, int C, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, float momentum,
                                                      float* running_mean, float* running_var, float* __restrict__ mean_out,
                                                      float* __restrict__ invstd_out, float* __restrict__ scale_out,
                                                      float* __restrict__ shift_out) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (c >= C) return;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int b = lane; b < P; b += 64) {
        const float nb = (float)min(rows_per_blk, M - b * rows_per_blk);
        chan_merge(n, mean, m2, nb, pmean[(size_t)c * P + b], pm2[(size_t)c * P + b]);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float nb = __shfl_xor(n, o, 64), mb = __shfl_xor(mean, o, 64), sb = __shfl_xor(m2, o, 64);
        // both partners must compute the identical merge: order the pair by lane parity at this level
        if (lane & o) { float na = nb, ma = mb, sa = sb; chan_merge(na, ma, sa, n, mean, m2); n = na; mean = ma; m2 = sa; }
        else chan_merge(n, mean, m2, nb, mb, sb);
    }
    if (lane == 0)
        bn_finish(c, mean, m2, M, gamma, beta, eps, momentum, running_mean, running_var, mean_out, invstd_out, scale_out, shift_out);
}

// eval mode: scale/shift from running statistics
__global__ void k_bn_eval_coeffs(const float* rm, const float* rv, const float* gamma, const float* beta, float eps, int C,
                                 float* scale, float* shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float sc = gamma[c] * (1.0f / sqrtf(rv[c] + eps));
    scale[c] = sc; shift[c] = beta[c] - rm[c] * sc;
}

// the same for up to OSI_BN_MULTI_MAX layers in ONE launch (the inference forward of the executor: 53 BatchNorms, 53 tiny launches
// otherwise): blockIdx.y = layer, the layer's six pointers ride in the kernel arguments
struct BnEvalTable { osi_bn_eval_layer l[OSI_BN_MULTI_MAX]; };
__global__ void k_bn_eval_coeffs_multi(BnEvalTable t, float eps) {
    const osi_bn_eval_layer& L = t.l[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= L.C) return;
    const float sc = L.gamma[c] * (1.0f / sqrtf(L.running_var[c] + eps));      // the expression of k_bn_eval_coeffs
    L.scale[c] = sc; L.shift[c] = L.beta[c] - L.running_mean[c] * sc;
}

// ---- apply: out = [relu]( y*scale + shift [+ res] ) ---------------------------------------------------
// ReLU bitmask: bit (i & 63) of word [(i >> 6) * 4 + c] = (component c of float4 #i is > 0 after BN(+res)). One bit per
// element (1/32 of the tensor) lets both BatchNorm-backward passes skip re-reading the activation just to rebuild the mask.
typedef unsigned long long u64;
// Traversal direction of the two streaming apply passes (bit 0: forward block-output pass, bit 1: backward apply): descending, so
// that a pass starts on the rows its producer — a convolution walking its row tiles upwards — wrote LAST, which the 256 MiB
// Infinity Cache still holds, and ends on the low rows its consumers start with. Element-wise, so the results do not depend on it.
// Measured in the step (profiles/r05_ab_bn_reverse.txt): -0.08 ms forward, -0.12 ms both.
#ifndef OSI_BN_REVERSE
#define OSI_BN_REVERSE 3
#endif
// RES: 0 = none, 1 = add `res`, 2 = add res * rscale + rshift (the projection shortcut's own BatchNorm applied on the fly)
template <int RES, bool RELU, bool BITS>
__global__ __launch_bounds__(NT) void k_bn_apply(const f32x4* __restrict__ y, const f32x4* __restrict__ res,
                                                const f32x4* __restrict__ scale, const f32x4* __restrict__ shift,
                                                const f32x4* __restrict__ rscale, const f32x4* __restrict__ rshift,
                                                f32x4* __restrict__ out, u64* __restrict__ bits, size_t n4, int c4n) {
    const size_t nq = (n4 + NT - 1) / NT;
    for (size_t q = blockIdx.x; q < nq; q += gridDim.x) {
        const size_t i = ((OSI_BN_REVERSE & 1) ? nq - 1 - q : q) * NT + threadIdx.x;
        if (i >= n4) continue;
        int c4 = (int)(i % (size_t)c4n);
        const f32x4 yv = __builtin_nontemporal_load(y + i), sc = scale[c4], sh = shift[c4];
        f32x4 v;   // one fma per element: the same expression the fused conv loaders and the dgrad gate evaluate
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(yv[e], sc[e], sh[e]);
        if (RES == 1) v += __builtin_nontemporal_load(res + i);
        if (RES == 2) {
            const f32x4 rv = __builtin_nontemporal_load(res + i), rs = rscale[c4], rh = rshift[c4];
            f32x4 t;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = __builtin_fmaf(rv[e], rs[e], rh[e]);
            v += t;
        }
        if (BITS) {
            const u64 b0 = __ballot(v.x > 0.f), b1 = __ballot(v.y > 0.f), b2 = __ballot(v.z > 0.f), b3 = __ballot(v.w > 0.f);
            if ((threadIdx.x & 63) == 0) {  // lane 0 carries the smallest (64-aligned) index of the wave
                u64* w = bits + (i >> 6) * 4;
                w[0] = b0; w[1] = b1; w[2] = b2; w[3] = b3;
            }
        }
        if (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        out[i] = v;
    }
}
__device__ __forceinline__ f32x4 mask_by_bits(f32x4 g, const u64* __restrict__ bits, size_t i) {
    const u64* w = bits + (i >> 6) * 4;
    const int b = (int)(i & 63);
    g.x = (w[0] >> b) & 1 ? g.x : 0.f; g.y = (w[1] >> b) & 1 ? g.y : 0.f;
    g.z = (w[2] >> b) & 1 ? g.z : 0.f; g.w = (w[3] >> b) & 1 ? g.w : 0.f;
    return g;
}

// ---- backward ---------------------------------------------------------------------------------------
// g = dA * mask; mask from the activation (MODE 1: act > 0), from the forward's bitmask (MODE 2) or none (MODE 0).
// Partial sums of g and g*xhat per workgroup.
template <int MODE>
__device__ __forceinline__ f32x4 masked(f32x4 gv, const void* m, size_t i) {
    if (MODE == 1) {
        f32x4 a = reinterpret_cast<const f32x4*>(m)[i];
        gv.x = a.x > 0.f ? gv.x : 0.f; gv.y = a.y > 0.f ? gv.y : 0.f;
        gv.z = a.z > 0.f ? gv.z : 0.f; gv.w = a.w > 0.f ? gv.w : 0.f;
    } else if (MODE == 2) {
        gv = mask_by_bits(gv, reinterpret_cast<const u64*>(m), i);
    }
    return gv;
}
// MODE 3: the upstream gradient is not stored per element — it is gathered from the gradient of a 3x3 / stride 2 / pad 1 max
// pool that followed BN + ReLU (the ResNet stem, osi_bn_relu_maxpool_fwd): pixel (h, w) receives the pooled gradient of every
// window whose stored argmax is (h, w) and whose maximum was positive (bit 7 of the index byte = ReLU gate). Same window order
// as k_maxpool_bwd, so the sums are the ones the unfused max-pool backward + ReLU mask would produce.
struct PoolSrc {
    const uint32_t* idx;   // [B][Ho][Wo][C/4] packed index bytes
    FastDiv dW, dH;        // pixel -> (b, h, w)
    int H, W, Ho, Wo;
};
__device__ __forceinline__ f32x4 pool_gather(const f32x4* __restrict__ gp, const PoolSrc& ps, uint32_t pix, int c4, int c4n) {
    const uint32_t row = fdiv(pix, ps.dW);
    const int w = (int)(pix - row * ps.dW.d);
    const uint32_t b = fdiv(row, ps.dH);
    const int h = (int)(row - b * ps.dH.d);
    f32x4 acc = {0, 0, 0, 0};
    for (int ho = h >> 1; ho <= (h + 1) >> 1; ++ho) {
        if (ho >= ps.Ho) continue;
        const int r = h - (ho * 2 - 1);
        for (int wo = w >> 1; wo <= (w + 1) >> 1; ++wo) {
            if (wo >= ps.Wo) continue;
            const int s = w - (wo * 2 - 1);
            const size_t o = ((size_t)(b * ps.Ho + ho) * ps.Wo + wo) * c4n + c4;
            const uint32_t id = ps.idx[o];
            const f32x4 g = gp[o];
            const uint32_t me = (uint32_t)(r * 3 + s) | 0x80u;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (((id >> (8 * k)) & 0xffu) == me) acc[k] += g[k];
        }
    }
    return acc;
}

template <int MODE>
__global__ __launch_bounds__(NT) void k_bn_bwd_partial(const float* __restrict__ dA, const void* __restrict__ msk,
                                                      const float* __restrict__ y, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, int M, int C, int rows_per_blk,
                                                      float* __restrict__ pdb, float* __restrict__ pdg, PoolSrc ps) {
    __shared__ f32x4 red[2][NT];
    const int P = gridDim.x;
    const RowSplit sp = row_split(C);
    const int tid = threadIdx.x;
    const int cv = tid % sp.CV, rl = tid / sp.CV;
    const int r0 = blockIdx.x * rows_per_blk;
    const int r1 = min(M, r0 + rows_per_blk);
    const bool active = rl < sp.RL;
    const size_t ld = C / 4;
    for (int g = 0; g < sp.G; ++g) {
        const int c4 = g * sp.CV + cv;
        const bool colok = active && c4 * 4 < C;
        f32x4 sb = {0, 0, 0, 0}, sg = {0, 0, 0, 0}, sb2 = {0, 0, 0, 0}, sg2 = {0, 0, 0, 0};
        if (colok) {
            const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[c4];
            const f32x4 is = reinterpret_cast<const f32x4*>(invstd)[c4];
            const f32x4* pd = reinterpret_cast<const f32x4*>(dA);
            const f32x4* py = reinterpret_cast<const f32x4*>(y);
            int r = r0 + rl;
            for (; r + sp.RL < r1; r += 2 * sp.RL) {  // two independent rows in flight
                const size_t i0 = (size_t)r * ld + c4, i1 = (size_t)(r + sp.RL) * ld + c4;
                f32x4 g0, g1, y0 = py[i0], y1 = py[i1];
                if (MODE == 3) {
                    g0 = pool_gather(pd, ps, (uint32_t)r, c4, (int)ld); g1 = pool_gather(pd, ps, (uint32_t)(r + sp.RL), c4, (int)ld);
                } else {
                    g0 = masked<MODE>(pd[i0], msk, i0); g1 = masked<MODE>(pd[i1], msk, i1);
                }
                sb += g0; sg += g0 * ((y0 - mu) * is);
                sb2 += g1; sg2 += g1 * ((y1 - mu) * is);
            }
            for (; r < r1; r += sp.RL) {
                const size_t i0 = (size_t)r * ld + c4;
                f32x4 g0 = MODE == 3 ? pool_gather(pd, ps, (uint32_t)r, c4, (int)ld) : masked<MODE>(pd[i0], msk, i0);
                sb += g0; sg += g0 * ((py[i0] - mu) * is);
            }
            sb += sb2; sg += sg2;
        }
        red[0][tid] = sb; red[1][tid] = sg;
        __syncthreads();
        if (colok && rl == 0) {
            for (int k = 1; k < sp.RL; ++k) { sb += red[0][k * sp.CV + cv]; sg += red[1][k * sp.CV + cv]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pdb[(size_t)(c4 * 4 + k) * P + blockIdx.x] = sb[k];
                pdg[(size_t)(c4 * 4 + k) * P + blockIdx.x] = sg[k];
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(NT) void k_bn_bwd_final(const float* __restrict__ pdb, const float* __restrict__ pdg, int P, int M,
                                                    int C, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                    float* __restrict__ c1, float* __restrict__ c2) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (c >= C) return;
    float ab = 0.f, ag = 0.f;
    for (int b = lane; b < P; b += 64) { ab += pdb[(size_t)c * P + b]; ag += pdg[(size_t)c * P + b]; }
    ab = wave_sum(ab); ag = wave_sum(ag);
    if (lane == 0) {
        dbeta[c] = ab; dgamma[c] = ag;
        c1[c] = ab / (float)M; c2[c] = ag / (float)M;
    }
}

// dy = gamma*invstd * ( g - c1 - xhat*c2 ) ; optionally also emits g (the masked upstream gradient) for the skip path
// Column sums of two row-major [P][C] partial matrices over row group s (blockIdx.y) -> channel-major [C][S] for k_bn_bwd_final.
__global__ __launch_bounds__(NT) void k_colsum2_group(const float* __restrict__ a, const float* __restrict__ b, int P, int Pc, int C,
                                                     int S, float* __restrict__ ga, float* __restrict__ gb) {
    __shared__ float ra[16][16], rb[16][16];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = min(blockIdx.x * 16 + cl, C - 1), s = blockIdx.y;
    const bool ok = blockIdx.x * 16 + cl < C;
    const int b0 = s * Pc, b1 = min(P, b0 + Pc);
    const float* pa = a + c;
    const float* pb = b + c;
    float xa0 = 0.f, xa1 = 0.f, xb0 = 0.f, xb1 = 0.f;
    int r = b0 + pl;
    for (; r + 48 < b1; r += 64) {     // four rows in flight per lane
        const float a0 = pa[(size_t)r * C], a1 = pa[(size_t)(r + 16) * C], a2 = pa[(size_t)(r + 32) * C], a3 = pa[(size_t)(r + 48) * C];
        const float v0 = pb[(size_t)r * C], v1 = pb[(size_t)(r + 16) * C], v2 = pb[(size_t)(r + 32) * C], v3 = pb[(size_t)(r + 48) * C];
        xa0 += a0; xa1 += a1; xa0 += a2; xa1 += a3;
        xb0 += v0; xb1 += v1; xb0 += v2; xb1 += v3;
    }
    for (; r < b1; r += 16) { xa0 += pa[(size_t)r * C]; xb0 += pb[(size_t)r * C]; }
    float xa = xa0 + xa1, xb = xb0 + xb1;
    ra[pl][cl] = xa; rb[pl][cl] = xb;
    __syncthreads();
    if (pl == 0 && ok) {
        for (int k = 1; k < 16; ++k) { xa += ra[k][cl]; xb += rb[k][cl]; }
        ga[(size_t)c * S + s] = xa; gb[(size_t)c * S + s] = xb;
    }
}

// Single-launch form of k_colsum2_group + k_bn_bwd_final for mid-sized P: 16 channels x 64 row lanes, four loads in flight per lane.
__global__ __launch_bounds__(16 * WPL) void k_bn_bwd_final_wide(const float* __restrict__ a, const float* __restrict__ b, int P, int M, int C,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ c1, float* __restrict__ c2) {
    __shared__ float red[WPL][16];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = min(blockIdx.x * 16 + cl, C - 1);
    const bool ok = blockIdx.x * 16 + cl < C;
    const float* pa = a + c;
    const float* pb = b + c;
    float xa0 = 0.f, xa1 = 0.f, xb0 = 0.f, xb1 = 0.f;
    int r = pl;
    for (; r + 3 * WPL < P; r += 4 * WPL) {
        const float a0 = pa[(size_t)r * C], a1 = pa[(size_t)(r + WPL) * C], a2 = pa[(size_t)(r + 2 * WPL) * C], a3 = pa[(size_t)(r + 3 * WPL) * C];
        const float b0 = pb[(size_t)r * C], b1 = pb[(size_t)(r + WPL) * C], b2 = pb[(size_t)(r + 2 * WPL) * C], b3 = pb[(size_t)(r + 3 * WPL) * C];
        xa0 += a0; xa1 += a1; xa0 += a2; xa1 += a3;
        xb0 += b0; xb1 += b1; xb0 += b2; xb1 += b3;
    }
    for (; r < P; r += WPL) { xa0 += pa[(size_t)r * C]; xb0 += pb[(size_t)r * C]; }
    const float sa = wide_reduce(red, xa0 + xa1, pl, cl);
    const float sb = wide_reduce(red, xb0 + xb1, pl, cl);
    if (pl == 0 && ok) {
        dbeta[c] = sa; dgamma[c] = sb;
        c1[c] = sa / (float)M; c2[c] = sb / (float)M;
    }
}

template <int MODE, bool EMITG>
__global__ __launch_bounds__(NT) void k_bn_bwd_apply(const f32x4* dA /* may alias dy */, const void* __restrict__ msk,
                                                    const f32x4* __restrict__ y, const f32x4* __restrict__ mean,
                                                    const f32x4* __restrict__ invstd, const f32x4* __restrict__ gamma,
                                                    const f32x4* __restrict__ c1, const f32x4* __restrict__ c2,
                                                    f32x4* dy, f32x4* __restrict__ gout, size_t n4, int c4n, PoolSrc ps) {
    const size_t nq = (n4 + NT - 1) / NT;
    for (size_t q = blockIdx.x; q < nq; q += gridDim.x) {
        const size_t i = ((OSI_BN_REVERSE & 2) ? nq - 1 - q : q) * NT + threadIdx.x;
        if (i >= n4) continue;
        int c4 = (int)(i % (size_t)c4n);
        f32x4 gv = MODE == 3 ? pool_gather(dA, ps, (uint32_t)(i / (size_t)c4n), c4, c4n) : masked<MODE>(__builtin_nontemporal_load(dA + i), msk, i);
        f32x4 is = invstd[c4];
        f32x4 xh = (__builtin_nontemporal_load(y + i) - mean[c4]) * is;
        f32x4 r = (gv - c1[c4] - xh * c2[c4]) * (gamma[c4] * is);
        if (EMITG) gout[i] = gv;
        dy[i] = r;
    }
}

static int rows_per_block(int M, int C, int& P) {
    // ~256 KiB of activations per workgroup, at most 1024 workgroups
    long rows = (256l * 1024) / ((long)C * 4);
    if (rows < 8) rows = 8;
    long minrows = (M + 1023) / 1024;
    if (rows < minrows) rows = minrows;
    if (rows > M) rows = M;
    P = (int)((M + rows - 1) / rows);
    return (int)rows;
}
// 4 workgroups (16 waves) per CU stream at the HBM rate and leave half of every CU's wave slots, and all of its LDS, to the
// weight-gradient kernels that run concurrently on the executor's side stream.
static int stream_grid(size_t n4, bool backward = false) {
    const size_t cap = (size_t)(backward ? g_osi_tuning.bn_grid_bwd : g_osi_tuning.bn_grid);
    size_t g = (n4 + NT - 1) / NT;
    return (int)(g > cap ? cap : g);
}

}  // namespace

extern "C" {

size_t osi_bn_workspace(int M, int C) {
    if (M <= 0 || C <= 0) return 0;
    int P;
    rows_per_block(M, C, P);
    return (size_t)2 * P * C * sizeof(float);
}

int osi_bn_train_stats(const float* y, int M, int C, const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var, float* mean, float* invstd, float* scale, float* shift,
                       void* ws, size_t ws_bytes, osi_stream_t stream) {
    OSI_REQUIRE(y && gamma && beta && mean && invstd && scale && shift && ws);
    OSI_REQUIRE(M > 0 && C > 0 && C % 4 == 0);
    OSI_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    int P;
    int rpb = rows_per_block(M, C, P);
    OSI_REQUIRE(ws_bytes >= (size_t)2 * P * C * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    float* pmean = (float*)ws;
    float* pm2 = pmean + (size_t)P * C;
    hipLaunchKernelGGL(k_bn_stats_partial, dim3(P), dim3(NT), 0, st, y, M, C, rpb, pmean, pm2);
    OSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bn_stats_final, dim3(osi_cdiv(C, NT / 64)), dim3(NT), 0, st, pmean, pm2, P, rpb, M, C, gamma, beta, eps,
                       momentum, running_mean, running_var, mean, invstd, scale, shift);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

int osi_bn_finalize_stats(float* pstats, size_t pstats_bytes, int P, int rows_per_block, int M, int C, const float* gamma,
                          const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* mean,
                          float* invstd, float* scale, float* shift, osi_stream_t stream) {
    OSI_REQUIRE(pstats && gamma && beta && mean && invstd && scale && shift);
    OSI_REQUIRE(P > 0 && rows_per_block > 0 && M > 0 && C > 0 && (long)(P - 1) * rows_per_block < M && (long)P * rows_per_block >= M);
    OSI_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    OSI_REQUIRE(pstats_bytes >= ((size_t)2 * P * C + 2 * (size_t)OSI_BN_GROUPS * C) * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    const float* pmean = pstats;
    const float* pm2 = pstats + (size_t)P * C;
    if (P <= g_osi_tuning.bn_single_p) {   // one launch merges up to this many row-tile partials per channel (measured knob)
        hipLaunchKernelGGL(k_bn_stats_final_rows, dim3(osi_cdiv(C, 16)), dim3(NT), 0, st, pmean, pm2, P, rows_per_block, M, C, gamma,
                           beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
        OSI_LAUNCH_CHECK();
        return OSI_OK;
    }
    if (P <= g_osi_tuning.bn_wide_p) {     // mid-sized P: still one launch, 64 partial lanes per channel
        hipLaunchKernelGGL(k_bn_stats_final_wide, dim3(osi_cdiv(C, 16)), dim3(16 * WPL), 0, st, pmean, pm2, P, rows_per_block, M, C, gamma,
                           beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
        OSI_LAUNCH_CHECK();
        return OSI_OK;
    }
    // two levels: S groups of Pc consecutive row tiles are reduced by S x C/16 workgroups, then one wave per channel merges them
    int S = osi_cdiv(P, 64);
    if (S > OSI_BN_GROUPS) S = OSI_BN_GROUPS;
    const int Pc = osi_cdiv(P, S);
    S = osi_cdiv(P, Pc);
    float* gmean = pstats + (size_t)2 * P * C;
    float* gm2 = gmean + (size_t)S * C;
    hipLaunchKernelGGL(k_bn_stats_group, dim3(osi_cdiv(C, 16), S), dim3(NT), 0, st, pmean, pm2, P, Pc, rows_per_block, M, C, S, gmean, gm2);
    OSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bn_stats_final, dim3(osi_cdiv(C, NT / 64)), dim3(NT), 0, st, (const float*)gmean, (const float*)gm2, S,
                       Pc * rows_per_block, M, C, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

int osi_bn_eval_coeffs(const float* running_mean, const float* running_var, const float* gamma, const float* beta, float eps,
                       int C, float* scale, float* shift, osi_stream_t stream) {
    OSI_REQUIRE(running_mean && running_var && gamma && beta && scale && shift && C > 0);
    hipLaunchKernelGGL(k_bn_eval_coeffs, dim3(osi_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, running_mean, running_var,
                       gamma, beta, eps, C, scale, shift);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

int osi_bn_eval_coeffs_multi(const osi_bn_eval_layer* layers, int n, float eps, osi_stream_t stream) {
    OSI_REQUIRE(layers && n > 0 && n <= OSI_BN_MULTI_MAX);
    BnEvalTable t{};
    int cmax = 0;
    for (int i = 0; i < n; ++i) {
        const osi_bn_eval_layer& L = layers[i];
        OSI_REQUIRE(L.running_mean && L.running_var && L.gamma && L.beta && L.scale && L.shift && L.C > 0);
        t.l[i] = L;
        if (L.C > cmax) cmax = L.C;
    }
    hipLaunchKernelGGL(k_bn_eval_coeffs_multi, dim3(osi_cdiv(cmax, 256), n), dim3(256), 0, (hipStream_t)stream, t, eps);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

static int bn_apply_impl(const float* y, const float* residual, const float* scale, const float* shift, float* out, void* bits,
                         int M, int C, int relu, hipStream_t st, const float* rscale = nullptr, const float* rshift = nullptr) {
    OSI_REQUIRE(y && scale && shift && out && M > 0 && C > 0 && C % 4 == 0);
    OSI_REQUIRE(!bits || relu);
    OSI_REQUIRE((rscale == nullptr) == (rshift == nullptr) && (!rscale || (residual && bits)));
    const size_t n4 = (size_t)M * C / 4;
    const int grid = stream_grid(n4), c4n = C / 4;
    auto Y = (const f32x4*)y; auto R = (const f32x4*)residual; auto S = (const f32x4*)scale; auto H = (const f32x4*)shift;
    auto RS = (const f32x4*)rscale; auto RH = (const f32x4*)rshift;
    auto O = (f32x4*)out; auto B = (u64*)bits;
#define OSI_APPLY(RES_, RELU_, BITS_) hipLaunchKernelGGL((k_bn_apply<RES_, RELU_, BITS_>), dim3(grid), dim3(NT), 0, st, Y, R, S, H, RS, RH, O, B, n4, c4n)
    if (rscale) OSI_APPLY(2, true, true);
    else if (bits && residual) OSI_APPLY(1, true, true);
    else if (bits) OSI_APPLY(0, true, true);
    else if (residual && relu) OSI_APPLY(1, true, false);
    else if (residual) OSI_APPLY(1, false, false);
    else if (relu) OSI_APPLY(0, true, false);
    else OSI_APPLY(0, false, false);
#undef OSI_APPLY
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

int osi_bn_apply(const float* y, const float* residual, const float* scale, const float* shift, float* out, int M, int C,
                 int relu, osi_stream_t stream) {
    return bn_apply_impl(y, residual, scale, shift, out, nullptr, M, C, relu, (hipStream_t)stream);
}

size_t osi_bn_relu_mask_bytes(int M, int C) {
    if (M <= 0 || C <= 0) return 0;
    const size_t n4 = (size_t)M * C / 4;
    return (n4 + 63) / 64 * 4 * sizeof(u64);
}

int osi_bn_apply_relu_mask(const float* y, const float* residual, const float* scale, const float* shift, float* out,
                           void* relu_mask, int M, int C, osi_stream_t stream) {
    OSI_REQUIRE(relu_mask);
    return bn_apply_impl(y, residual, scale, shift, out, relu_mask, M, C, 1, (hipStream_t)stream);
}

int osi_bn_apply_relu_mask2(const float* y, const float* scale, const float* shift, const float* res_y, const float* res_scale,
                            const float* res_shift, float* out, void* relu_mask, int M, int C, osi_stream_t stream) {
    OSI_REQUIRE(relu_mask && res_y && res_scale && res_shift);
    return bn_apply_impl(y, res_y, scale, shift, out, relu_mask, M, C, 1, (hipStream_t)stream, res_scale, res_shift);
}

static int bn_backward_impl(const float* dout, const void* msk, int mode, const float* y, const float* mean, const float* invstd,
                            const float* gamma, float* dy, float* gmasked, float* dgamma, float* dbeta, int M, int C, void* ws,
                            size_t ws_bytes, hipStream_t st, PoolSrc ps = PoolSrc{}) {
    OSI_REQUIRE(dout && y && mean && invstd && gamma && (dy || mode == 3) && dgamma && dbeta && ws);
    OSI_REQUIRE(M > 0 && C > 0 && C % 4 == 0);
    int P;
    int rpb = rows_per_block(M, C, P);
    // partial sums + the two per-channel coefficient vectors
    OSI_REQUIRE(ws_bytes >= ((size_t)2 * P * C + 2 * (size_t)C) * sizeof(float));
    float* pdb = (float*)ws;
    float* pdg = pdb + (size_t)P * C;
    float* c1 = pdg + (size_t)P * C;
    float* c2 = c1 + C;
    if (mode == 3) hipLaunchKernelGGL(k_bn_bwd_partial<3>, dim3(P), dim3(NT), 0, st, dout, msk, y, mean, invstd, M, C, rpb, pdb, pdg, ps);
    else if (mode == 2) hipLaunchKernelGGL(k_bn_bwd_partial<2>, dim3(P), dim3(NT), 0, st, dout, msk, y, mean, invstd, M, C, rpb, pdb, pdg, ps);
    else if (mode == 1) hipLaunchKernelGGL(k_bn_bwd_partial<1>, dim3(P), dim3(NT), 0, st, dout, msk, y, mean, invstd, M, C, rpb, pdb, pdg, ps);
    else hipLaunchKernelGGL(k_bn_bwd_partial<0>, dim3(P), dim3(NT), 0, st, dout, msk, y, mean, invstd, M, C, rpb, pdb, pdg, ps);
    OSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bn_bwd_final, dim3(osi_cdiv(C, NT / 64)), dim3(NT), 0, st, pdb, pdg, P, M, C, dgamma, dbeta, c1, c2);
    OSI_LAUNCH_CHECK();
    if (!dy) return OSI_OK;     // reductions only (dgamma, dbeta): the caller assembles the consumer's gradient another way
    const size_t n4 = (size_t)M * C / 4;
    const int grid = stream_grid(n4, true), c4n = C / 4;
    auto D = (const f32x4*)dout; auto Y = (const f32x4*)y;
    auto MU = (const f32x4*)mean; auto IS = (const f32x4*)invstd; auto G = (const f32x4*)gamma;
    auto C1 = (const f32x4*)c1; auto C2 = (const f32x4*)c2;
    auto DY = (f32x4*)dy; auto GO = (f32x4*)gmasked;
#define OSI_BWD_APPLY(MODE_, EMIT_) hipLaunchKernelGGL((k_bn_bwd_apply<MODE_, EMIT_>), dim3(grid), dim3(NT), 0, st, D, msk, Y, MU, IS, G, C1, C2, DY, GO, n4, c4n, ps)
    if (mode == 3) OSI_BWD_APPLY(3, false);
    else if (mode == 2) { if (gmasked) OSI_BWD_APPLY(2, true); else OSI_BWD_APPLY(2, false); }
    else if (mode == 1) { if (gmasked) OSI_BWD_APPLY(1, true); else OSI_BWD_APPLY(1, false); }
    else { if (gmasked) OSI_BWD_APPLY(0, true); else OSI_BWD_APPLY(0, false); }
#undef OSI_BWD_APPLY
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

// (psum_g, psum_gx)[P][C] from a dgrad epilogue -> dgamma, dbeta and the two apply coefficients: one launch for P <= bn_wide_p, else two
static int bwd_reduce_partials(const float* psum_g, const float* psum_gx, int P, int M, int C, float* gb, float* gg, int S, int Pc,
                               float* dgamma, float* dbeta, float* c1, float* c2, hipStream_t st) {
    if (P <= g_osi_tuning.bn_wide_p) {
        hipLaunchKernelGGL(k_bn_bwd_final_wide, dim3(osi_cdiv(C, 16)), dim3(16 * WPL), 0, st, psum_g, psum_gx, P, M, C, dgamma, dbeta, c1, c2);
        OSI_LAUNCH_CHECK();
        return OSI_OK;
    }
    hipLaunchKernelGGL(k_colsum2_group, dim3(osi_cdiv(C, 16), S), dim3(NT), 0, st, psum_g, psum_gx, P, Pc, C, S, gb, gg);
    OSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bn_bwd_final, dim3(osi_cdiv(C, NT / 64)), dim3(NT), 0, st, (const float*)gb, (const float*)gg, S, M, C, dgamma,
                       dbeta, c1, c2);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

int osi_bn_backward_fused(const float* g, const float* y, const float* mean, const float* invstd, const float* gamma,
                          const float* psum_g, const float* psum_gx, int P, float* dy, float* dgamma, float* dbeta, int M, int C,
                          void* ws, size_t ws_bytes, osi_stream_t stream) {
    OSI_REQUIRE(g && y && mean && invstd && gamma && psum_g && psum_gx && dy && dgamma && dbeta && ws);
    OSI_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && P > 0);
    int S = osi_cdiv(P, 64);
    if (S > OSI_BN_GROUPS) S = OSI_BN_GROUPS;
    const int Pc = osi_cdiv(P, S);
    S = osi_cdiv(P, Pc);
    OSI_REQUIRE(ws_bytes >= ((size_t)2 * S * C + 2 * (size_t)C) * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    float* gb = (float*)ws;               // [C][S] group sums of g
    float* gg = gb + (size_t)S * C;       // [C][S] group sums of g*xhat
    float* c1 = gg + (size_t)S * C;
    float* c2 = c1 + C;
    if (int e = bwd_reduce_partials(psum_g, psum_gx, P, M, C, gb, gg, S, Pc, dgamma, dbeta, c1, c2, st)) return e;
    const size_t n4 = (size_t)M * C / 4;
    hipLaunchKernelGGL((k_bn_bwd_apply<0, false>), dim3(stream_grid(n4, true)), dim3(NT), 0, st, (const f32x4*)g, (const void*)nullptr,
                       (const f32x4*)y, (const f32x4*)mean, (const f32x4*)invstd, (const f32x4*)gamma, (const f32x4*)c1,
                       (const f32x4*)c2, (f32x4*)dy, (f32x4*)nullptr, n4, C / 4, PoolSrc{});
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

int osi_bn_backward_reduce(const float* psum_g, const float* psum_gx, int P, float* dgamma, float* dbeta, int M, int C, void* ws,
                           size_t ws_bytes, osi_stream_t stream) {
    OSI_REQUIRE(psum_g && psum_gx && dgamma && dbeta && ws && M > 0 && C > 0 && P > 0);
    int S = osi_cdiv(P, 64);
    if (S > OSI_BN_GROUPS) S = OSI_BN_GROUPS;
    const int Pc = osi_cdiv(P, S);
    S = osi_cdiv(P, Pc);
    OSI_REQUIRE(ws_bytes >= ((size_t)2 * S * C + 2 * (size_t)C) * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    float* gb = (float*)ws;
    float* gg = gb + (size_t)S * C;
    float* c1 = gg + (size_t)S * C;
    float* c2 = c1 + C;
    if (int e = bwd_reduce_partials(psum_g, psum_gx, P, M, C, gb, gg, S, Pc, dgamma, dbeta, c1, c2, st)) return e;
    return OSI_OK;
}

int osi_bn_relu_maxpool_bwd(const float* gpool, const void* idx, const float* y, const float* mean, const float* invstd,
                            const float* gamma, float* dy, float* dgamma, float* dbeta, int B, int H, int W, int C, void* ws,
                            size_t ws_bytes, osi_stream_t stream) {
    OSI_REQUIRE(idx && B > 0 && H > 0 && W > 0 && (long)B * H * W < (1l << 31));
    PoolSrc ps;
    ps.idx = (const uint32_t*)idx;
    ps.dW = make_fastdiv((uint32_t)W); ps.dH = make_fastdiv((uint32_t)H);
    ps.H = H; ps.W = W; ps.Ho = (H + 2 - 3) / 2 + 1; ps.Wo = (W + 2 - 3) / 2 + 1;
    return bn_backward_impl(gpool, nullptr, 3, y, mean, invstd, gamma, dy, nullptr, dgamma, dbeta, B * H * W, C, ws, ws_bytes,
                            (hipStream_t)stream, ps);
}

int osi_bn_backward(const float* dout, const float* act, const float* y, const float* mean, const float* invstd,
                    const float* gamma, float* dy, float* gmasked, float* dgamma, float* dbeta, int M, int C, void* ws,
                    size_t ws_bytes, osi_stream_t stream) {
    return bn_backward_impl(dout, act, act ? 1 : 0, y, mean, invstd, gamma, dy, gmasked, dgamma, dbeta, M, C, ws, ws_bytes,
                            (hipStream_t)stream);
}

int osi_bn_backward_relu_mask(const float* dout, const void* relu_mask, const float* y, const float* mean, const float* invstd,
                              const float* gamma, float* dy, float* gmasked, float* dgamma, float* dbeta, int M, int C, void* ws,
                              size_t ws_bytes, osi_stream_t stream) {
    OSI_REQUIRE(relu_mask);
    return bn_backward_impl(dout, relu_mask, 2, y, mean, invstd, gamma, dy, gmasked, dgamma, dbeta, M, C, ws, ws_bytes,
                            (hipStream_t)stream);
}

size_t osi_bn_backward_workspace(int M, int C) {
    if (M <= 0 || C <= 0) return 0;
    int P;
    rows_per_block(M, C, P);
    return ((size_t)2 * P * C + 2 * (size_t)C) * sizeof(float);
}

}  // extern "C"
