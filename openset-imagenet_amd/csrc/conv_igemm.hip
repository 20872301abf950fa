// Implicit-GEMM convolutions on the exact-fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces the 53 bias-free conv2d forward / input-gradient / weight-gradient calls that
// torchvision's ResNet-50 issues under openset_imagenet/model.py:17,37 (reference) — see
// SURVEY.md Appendix A for the 23 unique shapes.
//
// Data layout (all fp32):
//   activations  NHWC   [B][H][W][C]            (C contiguous -> the GEMM K axis of fwd/dgrad is contiguous)
//   weights      KRSC   [Cout][R][S][Cin]       (= torch OIHW tensor in channels_last strides)
//
// GEMM views
//   fwd    Y[m, n]  = sum_k  im2col(X)[m, k] * W[n, k]          m=(b,ho,wo)  k=(r,s,c)  n=cout
//   dgrad  dX[m, c] = sum_k  im2col'(dY)[m, k] * W[(n), (r,s), c] m=(b,h,w)   k=(r,s,n)
//   wgrad  dW[n, (r,s,c)] = sum_m dY[m, n] * im2col(X)[m, (r,s,c)]            k=m (split across blocks)
//
// One workgroup = 4 waves in a 2x2 grid; each wave owns WM x WN MFMA tiles of 32x32, K-step 32 per
// LDS stage, register-staged global->LDS copies through buffer (SRSRC) loads whose range check is the im2col
// zero padding. NST LDS stages: 1 (default: single buffered, two barriers per K tile, twice the resident
// workgroups) or 2 (double buffered, one barrier per K tile).
//
// LDS operand images
//   "R" image  [rows][32 + 4]  k contiguous, read with ds_read_b128 (conflict-free at a 36-float row stride)
//   "C" image  [32][cols]      k major, read with ds_read_b32 (lanes walk 32 consecutive floats)
// The K order inside a stage is permuted identically for both operands (lane half h of MFMA (j,e) consumes
// k = 8j + 4h + e), which a dot product does not care about.
#include "conv_common.h"
#include <algorithm>
#include <type_traits>

using namespace osi_conv;

namespace {

constexpr int BK = 32;
constexpr int LDR = BK + 4;  // row stride of an "R" image, floats

struct ConvP {
    const float* x;   // fwd: input, dgrad: dY, wgrad: X
    const float* w;   // fwd/dgrad: weights, wgrad: dY
    float* y;         // fwd: Y, dgrad: dX, wgrad: dW or slab
    int B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad;
    int M;            // GEMM M (fwd: B*Ho*Wo)
    int Ktot;         // R*S*Cin (row length of W)
    int MT, NT;       // tile counts
    int accumulate;   // epilogue adds into y
    int x_bytes, w_bytes;  // sizes of p.x / p.w for the buffer descriptors
    FastDiv dHoWo, dWo;  // fwd/wgrad: divide by Ho*Wo, Wo
    FastDiv cHW[4], cW[4];  // dgrad: Hc*Wc and Wc of each parity class grid (stride <= 2)
    int unit;             // dgrad: 1x1 stride-1 conv -> the im2col row of pixel m is row m of dY
    float* pmean;         // fwd: optional BatchNorm partials [MT][Cout] (mean, M2) written by the epilogue
    float* pm2;
    // dgrad epilogue fusion (all optional): dx = mask(ebits) . (acc + addend); esum[0..2][eP][Cin] = per row tile column sums of
    // g, g*ey0, g*ey1 for the BatchNorm backward of the layer(s) that consume this gradient
    const float* addend;
    const unsigned long long* ebits;
    const float* ey0;
    const float* ey1;
    const float *emean0, *einv0, *emean1, *einv1;  // batch mean / invstd of the consumer BatchNorm(s): xhat = (y - mean) * invstd
    const float *escale0, *eshift0;                // ReLU gate recomputed from ey0 (no bitmask): on where ey0 * scale + shift > 0
    int skip_empty;                                // dgrad: parity classes no filter tap reaches write nothing (osi_conv_dgrad accumulate = 2)
    int eadd_even;                                 // dgrad epilogue: the addend exists only at pixels with even h and even w
    const uint32_t* epool;                         // pool mode: arg-max bytes of the max-pool whose output this conv reads; ey0 is the
    int epH, epW;                                  // pre-pool tensor [B][epH][epW][Cin] (osi_dgrad_fusion.pool_idx)
    float* esum;
    int eP;
    // wgrad only
    int kchunk;       // pixels per split
    int wtbl;         // generic loader: rolling table of input byte offsets in LDS instead of per-row divisions
    int wgroup, splits;  // XCD-aware block mapping on a 1-D grid; number of K splits
    int gkind, gkeys, ginner;  // what an XCD's consecutive slots share (see k_conv_wgrad), number of keys, workgroups per key
    // fused input activation (fwd A operand / wgrad X operand): the operand is relu(x * in_scale[c] + in_shift[c]) — the BatchNorm +
    // ReLU of the producer layer applied in the loader, so the activation tensor is never materialised. Zero padding / rows past
    // the end stay exact zeros (the select runs AFTER the activation).
    const float* in_scale;
    const float* in_shift;
    const float* res;     // fwd XF = 2: shortcut tensor added before the ReLU (same shape as x)
    // fwd OE (inference): the OUTPUT is [relu](fma(acc, osc[n], osh[n]) [+ ores[m][n]]) — the conv's own BatchNorm (eval-mode coefficients),
    // the shortcut and the ReLU applied in the epilogue; the pre-BN tensor is never written.
    const float* osc;
    const float* osh;
    const float* ores;    // [M][Cout] or NULL (read through a zero-record descriptor: no branch around the load)
    int orelu;            // a select, not a branch
    size_t slab_stride;
    // fwd / dgrad, balanced remainder (see plan_tail_split): row tiles mt < MT1 are computed at full K by blocks [0, g1); the last
    // MT - MT1 row tiles are split ks_S ways along K by the blocks behind them, each writing its raw 64x64 accumulator tile to
    // ks_slab[((mt - MT1) * NT + nt) * ks_S + split]; k_conv_tail_fixup adds the splits in fixed order and finishes the epilogue
    int MT1, g1, ks_S, ks_T;
    float* ks_slab;
#ifdef OSI_STAMPS
    unsigned long long* stamps;   // diagnostic build only (make stamps): 8 words per workgroup, see tools/wg_timeline.py
#endif
};

// Diagnostic build (-DOSI_STAMPS -> libosi_hip_stamps.so, never loaded by the product): thread 0 of every workgroup stores the
// 100 MHz real-time counter at four points — [0] kernel entry, [1] K loop entered (prologue + first tile staged), [2] K loop done,
// [3] epilogue done — plus HW_ID [4] and XCC_ID [5], into a buffer no other code reads (MI355X_MICROARCH.md, DVFS give-back item 6;
// cdna_hip_programming.md section 7, In-kernel stamps). The product build compiles these macros to nothing.
#ifdef OSI_STAMPS
#define OSI_STAMP(p, idx, slot)                                                                                        \
    do {                                                                                                               \
        if (threadIdx.x == 0 && (p).stamps) (p).stamps[(size_t)(idx) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define OSI_STAMP_ID(p, idx)                                                                   \
    do {                                                                                       \
        if (threadIdx.x == 0 && (p).stamps) {                                                  \
            (p).stamps[(size_t)(idx) * 8 + 4] = __builtin_amdgcn_s_getreg(63492); /* HW_ID */  \
            (p).stamps[(size_t)(idx) * 8 + 5] = __builtin_amdgcn_s_getreg(63508); /* XCC_ID */ \
        }                                                                                      \
    } while (0)
unsigned long long* g_osi_stamps = nullptr;
#else
#define OSI_STAMP(p, idx, slot) do {} while (0)
#define OSI_STAMP_ID(p, idx) do {} while (0)
#endif

// XCD-aware tile mapping: workgroups b and b+8 share an XCD (and its L2). All NT column tiles of one row
// tile are placed on one XCD in consecutive dispatch slots, so the im2col rows are fetched from HBM once.
__device__ __forceinline__ bool tile_of_block(int bid, int MT, int NT, int& mt, int& nt) {
    int xcd = bid & 7, slot = bid >> 3;
    nt = slot % NT;
    mt = (slot / NT) * 8 + xcd;
    return mt < MT;
}

// Balanced remainder. A launch of V = MT x NT tiles on a chip of N CUs runs floor(V / N) full rounds plus a ragged one in which
// V mod N CUs hold one workgroup more than the rest — at 3-6 workgroups per CU (the 14x14 / 7x7 layers at batch 128) that is
// 12-25 % of the launch. The tiles of the ragged round (the last row tiles) are therefore split along K into S = floor(N / tiles)
// short workgroups each, so that every CU ends up with the same number of full-K workgroups plus (at most) one short one. Only those
// few tiles go through a slab + fix-up pass (a few MB), everything else keeps its single-pass epilogue. Deterministic: the splits
// are summed in fixed order. Returns false for a block that has nothing to do.
__device__ __forceinline__ bool tile_of_block_split(const ConvP& p, int bid, int& mt, int& nt, int& ks) {
    ks = -1;
    if (p.ks_S <= 1) return tile_of_block(bid, p.MT, p.NT, mt, nt);
    if (bid < p.g1) return tile_of_block(bid, p.MT1, p.NT, mt, nt);
    const int pb = bid - p.g1;
    const int xcd = pb & 7, slot = pb >> 3;
    nt = slot % p.NT;
    const int key = (slot / p.NT) * 8 + xcd;       // (remainder row tile, split): the NT column tiles of a key share the A slice on one XCD
    const int mt2 = key / p.ks_S;
    ks = key - mt2 * p.ks_S;
    mt = p.MT1 + mt2;
    return mt < p.MT;
}

// Diagnostic builds only (make ablate -> tools/probes/bin/libosi_hip_abl<bits>.so; tools/probes/ablate.sh): -DOSI_ABLATE=<bits> compiles
// parts of the four MFMA kernels' K loops out — 1 no global loads inside the loop, 2 no register-side staging / LDS stores, 4 no barriers,
// 8 no tap mask (all-taps weight gradient), 16 no LDS operand reads (MFMAs from staging registers), 32 no epilogue (forward, input
// gradient), 64 forward: the activation rows of a 3x3 layer loaded and transformed for the first tap only (ceiling of a window kernel). Wrong results, right timing: how the tap mask (17 - 20 % of k_conv_wgrad3 as a test + compare + select per tap), the
// address adds of the per-tap weight gradient's operand reads and the price of operand staging were measured. 0 in the product.
#ifndef OSI_ABLATE
#define OSI_ABLATE 0
#endif

// ---- MFMA over one LDS stage -------------------------------------------------------------------------
// A: R image (rows = GEMM rows), B: R image (rows = GEMM cols)
// [J0, J1) of the 4 sub-steps of 8 k each: a kernel may split the block to place other work between MFMAs of the same wave
template <int WM, int WN, int J0 = 0, int J1 = 4>
__device__ __forceinline__ void mma_RR(const float* sA, const float* sB, int arow, int brow, int lane,
                                       f32x16 (&acc)[WM][WN]) {
    const int h4 = (lane >> 5) * 4, l31 = lane & 31;
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        f32x4 a[WM], b[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) a[i] = *reinterpret_cast<const f32x4*>(sA + (arow + i * 32 + l31) * LDR + 8 * j + h4);
#pragma unroll
        for (int i = 0; i < WN; ++i) b[i] = *reinterpret_cast<const f32x4*>(sB + (brow + i * 32 + l31) * LDR + 8 * j + h4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int n = 0; n < WN; ++n)
                    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[n][e], acc[i][n], 0, 0, 0);
    }
}
// A: R image, B: C image [32][LDC]
template <int WM, int WN, int LDC>
__device__ __forceinline__ void mma_RC(const float* sA, const float* sB, int arow, int bcol, int lane,
                                       f32x16 (&acc)[WM][WN]) {
    const int h4 = (lane >> 5) * 4, l31 = lane & 31;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 a[WM];
#pragma unroll
        for (int i = 0; i < WM; ++i) a[i] = *reinterpret_cast<const f32x4*>(sA + (arow + i * 32 + l31) * LDR + 8 * j + h4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float b[WN];
#pragma unroll
            for (int n = 0; n < WN; ++n) b[n] = sB[(8 * j + h4 + e) * LDC + bcol + n * 32 + l31];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int n = 0; n < WN; ++n)
                    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[n], acc[i][n], 0, 0, 0);
        }
    }
}
// A: C image [32][LDA], B: C image [32][LDB]
// An operand that spans two 32-wide MFMA blocks (WM / WN = 2) is read INTERLEAVED: lane l takes columns 2 l and 2 l + 1 of the wave's 64
// (block 0 = the even columns, block 1 = the odd ones) with ONE ds_read_b64 whose K-step offset is an immediate. The plain mapping
// (block i = columns 32 i .. 32 i + 31) made the compiler pair the two 4-byte reads into a ds_read2_b32, whose 8-bit offsets cannot reach
// the next K step (two rows = 1056 B): every K step paid two address v_adds feeding two reads feeding a wait. The kernel's epilogue
// un-permutes (cc_index below).
template <int WM, int WN, int LDA, int LDB, int K0 = 0, int K1 = BK / 2>
__device__ __forceinline__ void mma_CC(const float* sA, const float* sB, int acol, int bcol, int lane,
                                       f32x16 (&acc)[WM][WN]) {
    static_assert(WM <= 2 && WN <= 2 && LDA % 2 == 0 && LDB % 2 == 0, "interleaved pairs");
    const int h = lane >> 5, l31 = lane & 31;
    const float* pa = sA + h * LDA + acol + (WM == 2 ? 2 * l31 : l31);
    const float* pb = sB + h * LDB + bcol + (WN == 2 ? 2 * l31 : l31);
#pragma unroll
    for (int ks = K0; ks < K1; ++ks) {
        float a[WM], b[WN];
        if constexpr (WM == 2) { const f32x2 v = *reinterpret_cast<const f32x2*>(pa + 2 * ks * LDA); a[0] = v[0]; a[1] = v[1]; }
        else a[0] = pa[2 * ks * LDA];
        if constexpr (WN == 2) { const f32x2 v = *reinterpret_cast<const f32x2*>(pb + 2 * ks * LDB); b[0] = v[0]; b[1] = v[1]; }
        else b[0] = pb[2 * ks * LDB];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int n = 0; n < WN; ++n)
                acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[n], acc[i][n], 0, 0, 0);
    }
}

// ======================================================================================================
// Forward
// ======================================================================================================
// NST = LDS stages: 2 = double buffered (one barrier per K tile), 1 = single buffered (two barriers, half the LDS, twice the
// resident workgroups per CU)
// XF = 1: the A operand is relu(x * in_scale[c] + in_shift[c]) (fused BatchNorm-apply + ReLU of the producer layer)
// XF = 2 (1x1 stride-1 only): the A operand is relu(x * in_scale[c] + in_shift[c] + res) — a whole block output (bn3 + identity
//         shortcut + ReLU) recomputed in the loader, so conv1 of the NEXT bottleneck need not wait for the block-output pass
// (the fused forms pin the register budget of their plain twins — 8 / 5 waves per SIMD — so that the extra loader work cannot
// cost occupancy, which is what these kernels live on)
// KS: the launch carries a K-split tail (tile_of_block_split); its own instantiation so that the plain kernels keep their scalar
// register count (<= 80 SGPRs = eight resident 256-thread workgroups per CU)
// W3 (3x3, stride 1, pad 1, 64x64 tile): ROW WINDOWS. For a fixed tap row r the taps s = 0, 1, 2 read the same activation rows shifted by
// one pixel, so ONE window per (r, 32-channel slice) is staged — in column-padded coordinates: slot of pixel m = m + m div W, i.e. one
// zero slot behind every image row, which is the left / right neighbour a border pixel must see: no mask on the MFMA side; the
// vertical validity stays the loader's select — and the three K tiles run from it, the A fragment of a lane read at its own base row + s.
// K-tile order (r, slice, s) instead of (r, s, slice). Activation rows loaded / transformed / stored per slice: 9 x 64 -> 3 x <= 78.
constexpr int W3_WROWS = 78;   // 64 pixels + a pad slot per image row they cross (<= 64 / W + 1, W >= 7) + the two outer neighbours
// OE (inference, plain input only): output epilogue [relu](fma(acc, osc, osh) [+ ores]) instead of the raw accumulators, see ConvP
template <int WM, int WN, bool STEM, int NST, int XF = 0, bool KS = false, bool W3 = false, bool OE = false>
__global__ __launch_bounds__(256, XF ? (WM * WN == 1 ? 8 : 5) : (NST == 1 ? (WM * WN >= 4 ? 3 : 4) : 2)) void k_conv_fwd(ConvP p) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    static_assert(!OE || (XF == 0 && !STEM && NST == 1), "output epilogue: the executor's single-buffered plain-input forms");
    static_assert(!W3 || (WM == 1 && WN == 1 && NST == 1 && !STEM && XF != 2), "row windows: the single-buffered 64x64 tile");
    constexpr int AROWS = W3 ? W3_WROWS : BM;      // rows of the A image in LDS
    constexpr int AR = W3 ? 3 : BM / 32, BR = BN / 32;  // float4 loads per thread per stage
    constexpr int STAGE = (AROWS + BN) * LDR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(!(XF && STEM), "the stem reads the image, not an activation");
    OSI_STAMP(p, blockIdx.x, 0); OSI_STAMP_ID(p, blockIdx.x);
    float* s_sc = smem + NST * STAGE;          // XF: per-input-channel scale | shift tables, behind the operand stages
    float* s_sh = s_sc + p.Cin;
    if (XF) {
        for (int c = threadIdx.x * 4; c < p.Cin; c += 1024) {
            *reinterpret_cast<f32x4*>(s_sc + c) = ld4(p.in_scale + c);
            *reinterpret_cast<f32x4*>(s_sh + c) = ld4(p.in_shift + c);
        }
        // visible to every wave after the first __syncthreads() below (before any sstore of transformed data is read)
    }

    static_assert(!KS || (WM == 1 && WN == 1 && !STEM && NST == 1), "the K-split tail is built for the single-buffered 64x64 tile");
    int mt, nt, ks = -1;
    if (KS) { if (!tile_of_block_split(p, blockIdx.x, mt, nt, ks)) return; }
    else if (!tile_of_block(blockIdx.x, p.MT, p.NT, mt, nt)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = mt * BM, n0 = nt * BN;
    const int kq = tid & 7, lr = tid >> 3;  // float4 slot inside a 32-float row, row inside a 32-row pass

    // per-thread im2col rows. Generic path: a_base = offset of tap (0,0) of the row's window (may lie outside the tensor, only
    // dereferenced when the tap is valid) and a_taps = bit (r*S+s) set when that tap is inside the image, so a load costs one add
    // and one bit test instead of re-deriving (hi, wi) and four comparisons.
    int a_base[AR], a_h0[AR], a_w0[AR];   // a_base: BYTE offset (incl. this thread's float4 slot), may be negative for halo rows
    unsigned a_taps[AR];
    bool a_ok[AR];
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes), rw = make_rsrc(p.w, p.w_bytes);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(XF == 2 ? p.res : p.x, p.x_bytes);   // XF = 2: the shortcut tensor, same shape as x
    if constexpr (W3) {
        // window slot u = lr + 32 i <-> padded index q(m0) - 1 + u; a_base: byte offset of the slot's pixel in tap row r = 1 (OOB: pad slot,
        // past the tensor, past the window); a_taps: bit r set when the pixel's row h + r - 1 is inside the image
        const uint32_t Wp = (uint32_t)p.W + 1u, q0 = (uint32_t)m0 + fdiv((uint32_t)m0, p.dWo);
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int u = lr + 32 * i, Q = (int)q0 - 1 + u;
            bool ok = u < W3_WROWS && Q >= 0;
            const uint32_t Qq = ok ? (uint32_t)Q : 0u, irow = fdiv(Qq, p.cW[0]), col = Qq - irow * Wp, mm = irow * (uint32_t)p.W + col;
            ok = ok && col < (uint32_t)p.W && mm < (uint32_t)p.M;
            const uint32_t mq = ok ? mm : 0u, b = fdiv(mq, p.dHoWo), h = fdiv(mq - b * p.dHoWo.d, p.dWo);
            a_ok[i] = ok; a_h0[i] = 0; a_w0[i] = 0;
            // (the three validity bits ride in the low bits of the 16-byte-aligned offset: one register per window slot)
            a_base[i] = ok ? (int)(((mq * (uint32_t)p.Cin + kq * 4) * 4) | (h > 0 ? 1u : 0u) | 2u | ((int)h < p.H - 1 ? 4u : 0u)) : (int)OOB;
            a_taps[i] = 0;
        }
    } else
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        int m = m0 + lr + 32 * i;
        a_ok[i] = m < p.M;
        uint32_t mm = a_ok[i] ? (uint32_t)m : 0u;
        uint32_t b = fdiv(mm, p.dHoWo);
        uint32_t rem = mm - b * p.dHoWo.d;
        uint32_t ho = fdiv(rem, p.dWo);
        uint32_t wo = rem - ho * p.dWo.d;
        a_h0[i] = (int)ho * p.stride - p.pad;
        a_w0[i] = (int)wo * p.stride - p.pad;
        a_taps[i] = 0;
        if (STEM) {
            a_base[i] = (int)b * p.H * p.W * p.Cin * 4;
        } else if (p.unit) {
            a_base[i] = a_ok[i] ? ((int)mm * p.Cin + kq * 4) * 4 : (int)OOB;
            a_taps[i] = a_ok[i] ? 1u : 0u;
        } else {
            a_base[i] = (((int)b * p.H * p.W + a_h0[i] * p.W + a_w0[i]) * p.Cin + kq * 4) * 4;
            for (int rr = 0; rr < p.R; ++rr)
                for (int ss = 0; ss < p.S; ++ss)
                    if ((unsigned)(a_h0[i] + rr) < (unsigned)p.H && (unsigned)(a_w0[i] + ss) < (unsigned)p.W) a_taps[i] |= 1u << (rr * p.S + ss);
            if (!a_ok[i]) a_taps[i] = 0;
        }
    }
    const uint32_t w_voff = ((uint32_t)(n0 + lr) * p.Ktot + kq * 4) * 4;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;

    const int Tall = p.Ktot / BK;
    const int t0 = (!KS || ks < 0) ? 0 : ks * p.ks_T;            // this workgroup's K tiles [t0, T)
    const int T = (!KS || ks < 0) ? Tall : min(Tall, t0 + p.ks_T);
    int r = 0, s = 0, c0 = 0;  // current tap / channel offset (non-stem)
    if (KS && !STEM && ks > 0) {
        if constexpr (W3) {            // tile t = (r, slice, s)
            const int KC3 = 3 * (p.Cin / BK), rem = t0 % KC3;
            r = t0 / KC3; c0 = (rem / 3) * BK; s = rem % 3;
        } else {
            const int tap = (t0 * BK) / p.Cin;
            c0 = t0 * BK - tap * p.Cin; r = tap / p.S; s = tap - r * p.S;
        }
    }
    int cs = s;              // W3: tap column of the tile the MFMAs are working on (the loader state runs one tile ahead)
    bool ld_new = false;     // W3: the loaded tile starts a new window (its activation rows sit in ra)
    int prow = 0;            // W3: window row of this lane's pixel for s = 0 (= its padded index relative to the tile's first)
    if constexpr (W3) {
        const uint32_t m = (uint32_t)min(m0 + wm * 32 + (lane & 31), p.M - 1);
        prow = (int)(m - (uint32_t)m0 + fdiv(m, p.dWo) - fdiv((uint32_t)m0, p.dWo));
    }
    int ld_c0 = 0, ld_tap = 0; // XF: channel offset / tap index of the tile sitting in ra (set by gload, used by sstore)
    f32x4 ra[AR], rb[BR];
    f32x4 rr[XF == 2 ? AR : 1];   // XF = 2: the shortcut rows of the tile sitting in ra

    auto gload = [&](int t) {
        if (XF) { ld_c0 = c0; ld_tap = r * p.S + s; }
        if (XF == 2) {
#pragma unroll
            for (int i = 0; i < AR; ++i) rr[i] = bld4(rres, (uint32_t)a_base[i], (uint32_t)(t * BK * 4));
        }
        // The run-time form (1x1 stride-1 or generic) is chosen ONCE per tile, outside the unrolled loop: with a branch per load the
        // compiler waited for every outstanding load before the next one (the other form's destination registers count as pending
        // across the join), so the rows of a tile paid one memory round trip each.
        if constexpr (W3) {
            ld_new = s == 0 || t == t0;
            if (ld_new) {
                if (XF) ld_tap = r;      // (the validity bit of a window row is its tap ROW's)
#pragma unroll
                for (int i = 0; i < AR; ++i) {
                    const bool ok = ((uint32_t)a_base[i] >> r) & 1;
                    ra[i] = bld4(rx, ok ? (uint32_t)((a_base[i] & ~15) + ((r - 1) * p.W * p.Cin + c0) * 4) : OOB, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < BR; ++i) rb[i] = bld4(rw, w_voff + (uint32_t)(32 * i) * p.Ktot * 4, (uint32_t)(((r * 3 + s) * p.Cin + c0) * 4));
            return;
        }
        if (STEM) {
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                int tap = t * 8 + kq;  // Cin = 4: one tap per float4
                int rr = tap / 7, ss = tap - rr * 7;
                int hi = a_h0[i] + rr, wi = a_w0[i] + ss;
                bool ok = a_ok[i] && tap < 49 && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                ra[i] = bld4(rx, ok ? (uint32_t)(a_base[i] + (hi * p.W + wi) * 16) : OOB, 0);
            }
        } else if (p.unit) {   // 1x1 stride-1: row m of the input, K offset in the scalar operand; no VALU at all
#pragma unroll
            for (int i = 0; i < AR; ++i) ra[i] = bld4(rx, (uint32_t)a_base[i], (uint32_t)(t * BK * 4));
        } else if (!(OSI_ABLATE & 64) || (r | s) == 0) {     // (64: the activation rows only for the first tap — ceiling of a window kernel)
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const bool ok = (a_taps[i] >> (r * p.S + s)) & 1;
                ra[i] = bld4(rx, ok ? (uint32_t)(a_base[i] + ((r * p.W + s) * p.Cin + c0) * 4) : OOB, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) rb[i] = bld4(rw, w_voff + (uint32_t)(32 * i) * p.Ktot * 4, (uint32_t)(t * BK * 4));
    };
    auto advance = [&]() {
        if constexpr (W3) {
            if (++s == 3) { s = 0; c0 += BK; if (c0 == p.Cin) { c0 = 0; ++r; } }
        } else if (!STEM) {
            c0 += BK;
            if (c0 == p.Cin) { c0 = 0; if (++s == p.S) { s = 0; ++r; } }
        }
    };
    // XF: the fused activation, applied in registers to the tile sitting in ra right before it is stored to LDS (after the MFMA
    // block, when the operand fragments are dead: no extra register pressure. Tried and rejected: running it between the wave's own
    // MFMAs to shorten the load -> LDS window — fragments + transform temporaries live together spill 48-128 B per lane).
    auto xform = [&]() {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(s_sc + ld_c0 + kq * 4), sh = *reinterpret_cast<const f32x4*>(s_sh + ld_c0 + kq * 4);
        // 1x1 stride-1 (`unit`): there is no padding, the only invalid rows are those past M in the last row tile — their accumulator rows
        // are never stored and the statistics mask them by row index (the K-split fix-up likewise), so the select after the activation
        // is skipped: every vector instruction of an fp32-MFMA kernel is paid in matrix-pipe time (profiles/r06_mfma_valu_coexec.txt), and
        // the bit test + compare + four selects per row were 14 of this loader's 30 per K tile. A uniform branch around arithmetic only.
        const bool select = W3 || !p.unit;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = __builtin_fmaf(ra[i][e], sc[e], sh[e]);
                if (XF == 2) t += rr[i][e];      // same two roundings as k_bn_apply<RES = 1>: fma, then add
                v[e] = fmaxf(t, 0.f);
            }
            ra[i] = v;
        }
        if (select) {
            // padding taps must read as the zero the reference pads the ACTIVATION with
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const bool ok = (((W3 ? (uint32_t)a_base[i] : a_taps[i]) >> ld_tap) & 1) != 0;
                ra[i] = ok ? ra[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto sstore = [&](int buf) {
        float* sA = smem + buf * STAGE;
        float* sB = sA + AROWS * LDR;
        if constexpr (W3) {
            if (ld_new) {
                if (XF) xform();
#pragma unroll
                for (int i = 0; i < AR; ++i)
                    if (lr + 32 * i < W3_WROWS) *reinterpret_cast<f32x4*>(sA + (lr + 32 * i) * LDR + kq * 4) = ra[i];
            }
        } else {
            if (XF && (!(OSI_ABLATE & 64) || p.unit || ld_tap == 0)) xform();
#pragma unroll
            for (int i = 0; i < AR; ++i) *reinterpret_cast<f32x4*>(sA + (lr + 32 * i) * LDR + kq * 4) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) *reinterpret_cast<f32x4*>(sB + (lr + 32 * i) * LDR + kq * 4) = rb[i];
    };

    gload(t0); advance();
    if (XF) __syncthreads();   // the scale / shift tables are complete
    sstore(0);
    __syncthreads();
    OSI_STAMP(p, blockIdx.x, 1);
    for (int t = t0; t < T; ++t) {
        const int buf = NST == 2 ? ((t - t0) & 1) : 0;
        if (!(OSI_ABLATE & 1)) if (t + 1 < T) { gload(t + 1); advance(); }
        const float* sA = smem + buf * STAGE;
        if (OSI_ABLATE & 16) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int n = 0; n < WN; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[i % AR][j & 3], rb[n % BR][j & 3], acc[i][n], 0, 0, 0);
        } else if constexpr (W3) {
            const float* pa = sA + (prow + cs) * LDR + (lane >> 5) * 4;
            const float* pb = sA + AROWS * LDR + (wn * 32 + (lane & 31)) * LDR + (lane >> 5) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(pa + 8 * j), b = *reinterpret_cast<const f32x4*>(pb + 8 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc[0][0], 0, 0, 0);
            }
            if (++cs == 3) cs = 0;
        } else
        mma_RR<WM, WN>(sA, sA + BM * LDR, wm * 32 * WM, wn * 32 * WN, lane, acc);
        if (!(OSI_ABLATE & 4)) if (NST == 1) __syncthreads();  // every wave is done reading the only stage
        if (!(OSI_ABLATE & 2)) if (t + 1 < T) sstore(NST == 2 ? (buf ^ 1) : 0);
        if (!(OSI_ABLATE & 4)) __syncthreads();
    }
    OSI_STAMP(p, blockIdx.x, 2);
    if (OSI_ABLATE & 32) { if (acc[0][0][0] == 123.456f) p.y[tid] = acc[0][0][1]; return; }

    bool stored = false;
    if constexpr (WM == 1 && WN == 1) {
        if (KS && ks >= 0) {
            // K split of a remainder tile: the raw accumulators go to the slab as a dense 64x64 tile (16-byte stores through the
            // same LDS transpose as below); output, statistics and row bounds are the fix-up pass's business
            constexpr int LDT = BN + 4;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) smem[(wm * 32 + acc_row(rr, lane)) * LDT + wn * 32 + (lane & 31)] = acc[0][0][rr];
            __syncthreads();
            float* dst = p.ks_slab + ((size_t)((mt - p.MT1) * p.NT + nt) * p.ks_S + ks) * (BM * BN);
            const int c4 = tid & 15, rg = tid >> 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rl = rg + 16 * k;
                *reinterpret_cast<f32x4*>(dst + rl * BN + c4 * 4) = *reinterpret_cast<const f32x4*>(smem + rl * LDT + c4 * 4);
            }
            OSI_STAMP(p, blockIdx.x, 3);
            return;
        }
        if (!p.accumulate) {
            // 64x64 tile: transpose through LDS so each lane stores 16 bytes (4 consecutive channels of a pixel), see k_conv_dgrad
            constexpr int LDT = BN + 4;
            static_assert(BM * LDT <= STAGE, "transposed tile must fit the operand stage");
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) smem[(wm * 32 + acc_row(rr, lane)) * LDT + wn * 32 + (lane & 31)] = acc[0][0][rr];
            __syncthreads();
            const int c4 = tid & 15, rg = tid >> 4;
            if constexpr (OE) {
                // the four rows' shortcut values first (all in flight together; no shortcut = zero records = zeros), then one fma, one
                // add, one max per element: the roundings of k_bn_apply<RES = 1> (fma, add) and of the fused loaders (fma, max)
                const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.ores ? p.ores : p.y, p.ores ? p.M * p.Cout * 4 : 0);
                const f32x4 sc = ld4(p.osc + n0 + c4 * 4), sh = ld4(p.osh + n0 + c4 * 4);
                f32x4 rv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int m = m0 + rg + 16 * k;
                    rv[k] = bld4(ro, m < p.M ? (uint32_t)((m * p.Cout + n0 + c4 * 4) * 4) : OOB, 0);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int rl = rg + 16 * k, m = m0 + rl;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(smem + rl * LDT + c4 * 4);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float t = __builtin_fmaf(a[e], sc[e], sh[e]) + rv[k][e]; v[e] = p.orelu ? fmaxf(t, 0.f) : t; }
                    if (m < p.M) *reinterpret_cast<f32x4*>(p.y + (size_t)m * p.Cout + n0 + c4 * 4) = v;
                }
            } else
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rl = rg + 16 * k, m = m0 + rl;
                if (m < p.M)
                    *reinterpret_cast<f32x4*>(p.y + (size_t)m * p.Cout + n0 + c4 * 4) = *reinterpret_cast<const f32x4*>(smem + rl * LDT + c4 * 4);
            }
            __syncthreads();   // the statistics below reuse the LDS
            stored = true;
        }
    }
    // epilogue: 32 lanes of a half-wave write 128 contiguous bytes of one output row
    if (!stored)
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int col = n0 + wn * 32 * WN + n * 32 + (lane & 31);
            if constexpr (OE) {
                const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.ores ? p.ores : p.y, p.ores ? p.M * p.Cout * 4 : 0);
                const float sc = p.osc[col], sh = p.osh[col];
                float rv[16];
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const int m = m0 + wm * 32 * WM + i * 32 + acc_row(rr, lane);
                    rv[rr] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ro, m < p.M ? (uint32_t)((m * p.Cout + col) * 4) : OOB, 0, 0));
                }
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const int m = m0 + wm * 32 * WM + i * 32 + acc_row(rr, lane);
                    const float t = __builtin_fmaf(acc[i][n][rr], sc, sh) + rv[rr];
                    if (m < p.M) p.y[(size_t)m * p.Cout + col] = p.orelu ? fmaxf(t, 0.f) : t;
                }
            } else
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                int m = m0 + wm * 32 * WM + i * 32 + acc_row(rr, lane);
                if (m < p.M) {
                    float* dst = p.y + (size_t)m * p.Cout + col;
                    *dst = p.accumulate ? *dst + acc[i][n][rr] : acc[i][n][rr];
                }
            }
        }

    // Optional BatchNorm batch statistics of this workgroup's rows, straight from the accumulators (saves a full re-read of
    // the conv output): per column a (count, mean, M2) triple per 32x32 tile -> Chan-merged over the wave's tiles -> over the
    // two wave rows through LDS -> one (mean, M2) pair per column in pmean/pm2[mt][Cout]. Fixed order, no atomics.
    if (p.pmean) {
        // Two code paths behind ONE uniform branch: a row tile that lies wholly inside the tensor (every tile of every network shape at
        // the usual batches: M is a multiple of 64) needs no row masks — 16 compares and 32 selects per 32 x 32 accumulator block that are
        // paid in matrix-pipe time (fp32 MFMAs hide no vector work, profiles/r06_mfma_valu_coexec.txt); same sums in the same order.
        auto stats = [&](auto FULLC) {
            constexpr bool full = decltype(FULLC)::value;
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                float cn = 0.f, cm = 0.f, cs = 0.f;
#pragma unroll
                for (int i = 0; i < WM; ++i) {
                    const int row0 = m0 + wm * 32 * WM + i * 32;
                    const float cnt = full ? 32.f : (float)min(32, max(0, p.M - row0));
                    float s = 0.f;
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) s += (full || row0 + acc_row(rr, lane) < p.M) ? acc[i][n][rr] : 0.f;
                    s += __shfl_xor(s, 32, 64);
                    const float mu = cnt > 0.f ? s / cnt : 0.f;
                    float q = 0.f;
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) {
                        const float dlt = acc[i][n][rr] - mu;
                        // (explicit fma in the unmasked path: the compiler would contract here and not there — the row walker and this
                        // kernel must agree bit for bit)
                        if constexpr (full) q = __builtin_fmaf(dlt, dlt, q);
                        else q += (row0 + acc_row(rr, lane) < p.M) ? dlt * dlt : 0.f;
                    }
                    q += __shfl_xor(q, 32, 64);
                    chan_merge(cn, cm, cs, cnt, mu, q);
                }
                if (lane < 32) {
                    float* dst = smem + (wm * BN + wn * 32 * WN + n * 32 + lane) * 3;  // the K loop ended with a barrier: LDS is free
                    dst[0] = cn; dst[1] = cm; dst[2] = cs;
                }
            }
        };
        if (m0 + BM <= p.M) stats(std::true_type{}); else stats(std::false_type{});
        __syncthreads();
        if (tid < BN) {
            const float* a = smem + tid * 3;
            const float* b = smem + (BN + tid) * 3;
            float cn = a[0], cm = a[1], cs = a[2];
            chan_merge(cn, cm, cs, b[0], b[1], b[2]);
            p.pmean[(size_t)mt * p.Cout + n0 + tid] = cm;
            p.pm2[(size_t)mt * p.Cout + n0 + tid] = cs;
        }
    }
    OSI_STAMP(p, blockIdx.x, 3);
}

// Epilogue of one 64x64 input-gradient tile, every lane owning 4 consecutive channels of a pixel (16-byte accesses): skip-connection
// addend, ReLU gate (stored bitmask or recomputed from the producer's pre-BN tensor), output, and the per-row-tile BatchNorm-backward
// partial sums of the consumer layer(s). `rd(row, c4)` returns the accumulated float4 of tile row `row`, channel quad `c4`: the
// convolution kernel reads its LDS-transposed accumulators, the fix-up pass of a K-split tail sums the tile's slab entries. One
// code path for both keeps their results identical in form (same gates, same reduction order inside the tile).
// FUSED: 0 = addend only; 1 = every fusion the ABI allows (tests, pool mode); 2 = "in-block" (the gate recomputed from y0 and / or
// the sums over y0; no addend, bitmask or second consumer: all four rows' loads in flight); 3 = "block input" (addend, bitmask,
// sums over y0 and y1; no recomputed gate). 2 and 3 are what the executor issues; they exist so that each fits 64 registers.
template <int FUSED, bool POOL, typename RD>
__device__ __forceinline__ void dgrad_epilogue64(const ConvP& p, float* smem, int cls, int mt, int m0, int n0, int Mc, int st, int ph,
                                                 int pw, const FastDiv& dHW, const FastDiv& dW, RD rd) {
    constexpr int BN = 64;
    constexpr bool USE_ADD = FUSED != 2, USE_BITS = FUSED == 1 || FUSED == 3, USE_Y1 = FUSED == 1 || FUSED == 3;
    constexpr bool USE_GATE = FUSED == 1 || FUSED == 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c4 = tid & 15, rg = tid >> 4;
    const int col = n0 + c4 * 4;
    // Straight-line code: every optional tensor sits behind a buffer descriptor with ZERO records when it is absent (the load
    // returns zeros without touching memory), rows past the end of the tensor use the out-of-range offset for loads AND the store,
    // and the run-time switches become selects. With a branch per optional tensor the compiler drained all outstanding loads at
    // every join: the four rows of a lane paid three to four dependent memory round trips EACH, and an input-gradient tile with
    // a short K loop spent longer in this epilogue than in its matrix work.
    constexpr int FULL = (int)0x80000000u;    // every tensor here is < 2 GiB (desc_ok); OOB = 0x80000000 is past any of them
    const bool has_sum = FUSED && p.esum != nullptr, has_gate = USE_GATE && p.escale0 != nullptr, has_bits = USE_BITS && p.ebits != nullptr;
    const bool has_y1 = USE_Y1 && has_sum && p.ey1 != nullptr;
    const __amdgpu_buffer_rsrc_t r_out = make_rsrc(p.y, FULL);
    f32x4 sg = {0, 0, 0, 0}, s0 = sg, s1 = sg, mu0 = sg, is0 = sg, gsc = sg, gsh = sg;
    if (FUSED) {
        const uint32_t cb = (uint32_t)col * 4u;
        mu0 = bld4(make_rsrc(p.emean0, has_sum ? FULL : 0), cb, 0); is0 = bld4(make_rsrc(p.einv0, has_sum ? FULL : 0), cb, 0);
        if (USE_GATE) { gsc = bld4(make_rsrc(p.escale0, has_gate ? FULL : 0), cb, 0); gsh = bld4(make_rsrc(p.eshift0, has_gate ? FULL : 0), cb, 0); }
    }
    uint32_t pixv[POOL ? 4 : 1], offb[4];    // pixel index and BYTE offset of this lane's float4 in row k; OOB past the tensor
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int m = m0 + rg + 16 * k;
        const bool valid = m < Mc;
        uint32_t pix = valid ? (uint32_t)m : 0u;
        if (st != 1) {
            const uint32_t b = fdiv(pix, dHW);
            const uint32_t rem = pix - b * dHW.d;
            const uint32_t h2 = fdiv(rem, dW), w2 = rem - h2 * dW.d;
            pix = (b * p.H + (ph + st * h2)) * p.W + (pw + st * w2);
        }
        if (POOL) pixv[k] = pix;
        offb[k] = valid ? (pix * p.Cin + col) * 4u : OOB;
    }
    const __amdgpu_buffer_rsrc_t r_y0 = make_rsrc(p.ey0, (has_sum || has_gate) ? FULL : 0);
    if constexpr (FUSED == 2) {
        // in-block: one optional tensor, so all four rows go out together
        f32x4 y0v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) y0v[k] = bld4(r_y0, offb[k], 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f32x4 v = rd(rg + 16 * k, c4);      // rows past Mc are exact zeros (their operand rows were range-checked loads)
#pragma unroll
            for (int e = 0; e < 4; ++e) {       // the producer's activation was never stored: its ReLU gate is recomputed from the
                const bool open = (__builtin_fmaf(y0v[k][e], gsc[e], gsh[e]) > 0.f) | !has_gate;   // pre-BN tensor, one fma as in the forward loader
                v[e] = open ? v[e] : 0.f;
            }
            sg += v;
            s0 += (v * (y0v[k] - mu0)) * is0;
            bst4(r_out, v, offb[k], 0);
        }
    } else {
        const __amdgpu_buffer_rsrc_t r_add = make_rsrc(p.addend, (USE_ADD && p.addend) ? FULL : 0);
        const __amdgpu_buffer_rsrc_t r_bits = make_rsrc(reinterpret_cast<const float*>(p.ebits), has_bits ? FULL : 0);
        const __amdgpu_buffer_rsrc_t r_y1 = make_rsrc(p.ey1, has_y1 ? FULL : 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int rl = rg + 16 * k;
            f32x4 v = rd(rl, c4);               // rows past Mc are exact zeros (their operand rows were range-checked loads)
            // sparse addend (a stride-2 1x1 branch wrote only the even-even pixels): elsewhere nothing is read. Computed without a
            // branch (st == 1 in that mode: pix = m = b*H*W + h*W + w), so that the rows stay one basic block
            // (round 6: the decomposition behind a uniform branch on eadd_even — arithmetic only, the load outside — saves ~48 vector
            // instructions per tile and costs four spilled registers: the join splits the block the rows' loads are scheduled in. Not kept.)
            const uint32_t apix = (uint32_t)(m0 + rl), ab = fdiv(apix, dHW), arem = apix - ab * dHW.d, ah = fdiv(arem, dW), aw = arem - ah * dW.d;
            const uint32_t aoff = (((ah | aw) & (uint32_t)p.eadd_even) != 0u) ? OOB : offb[k];   // eadd_even is 0 or 1: a mask, not a branch
            v += bld4(r_add, aoff, 0);          // may be the output buffer itself: read and written by the same lane
            if constexpr (FUSED && POOL) {
                // dx is the gradient w.r.t. a max-pooled activation: the BatchNorm-backward reductions of the layer BEFORE the pool
                // see this value at the window's arg-max pixel, gated by the window's ReLU (bit 7): sum g, sum g * xhat(arg-max)
                if (offb[k] != OOB) {
                    const uint32_t pix = pixv[k], off = offb[k] >> 2;
                    const uint32_t bb = fdiv(pix, dHW), rem = pix - bb * dHW.d, ho = fdiv(rem, dW), wo = rem - ho * dW.d;
                    const uint32_t idw = p.epool[off >> 2];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const uint32_t byte = (idw >> (8 * e)) & 0xffu, t = byte & 0x7fu, r = t / 3u, sx = t - 3u * r;
                        const uint32_t apix = (bb * p.epH + 2u * ho - 1u + r) * p.epW + 2u * wo - 1u + sx;
                        const float gv = (byte & 0x80u) ? v[e] : 0.f;
                        sg[e] += gv;
                        s0[e] += (gv * (p.ey0[(size_t)apix * p.Cin + col + e] - mu0[e])) * is0[e];
                    }
                }
            } else if constexpr (FUSED != 0) {
                // 1 bit / element: words (i4 >> 6) * 4 + component, bit i4 & 63 (bn.hip): 32 contiguous bytes per float4
                const uint32_t i4 = offb[k] >> 4, bit = i4 & 63u;
                // of each component's 64-bit word only the half holding this element's bit is fetched (4 registers, not 8)
                const uint32_t boff = offb[k] != OOB ? (i4 >> 6) * 32u + (bit >> 5) * 4u : OOB;
                uint32_t wb[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) wb[e] = __builtin_amdgcn_raw_buffer_load_b32(r_bits, boff, 8 * e, 0);
                const f32x4 y0v = bld4(r_y0, offb[k], 0), y1v = bld4(r_y1, offb[k], 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool keep = (((wb[e] >> (bit & 31u)) & 1u) != 0) | !has_bits;
                    const bool open = !USE_GATE || (__builtin_fmaf(y0v[e], gsc[e], gsh[e]) > 0.f) | !has_gate;
                    v[e] = (keep & open) ? v[e] : 0.f;
                }
                // BatchNorm-backward partial sums of the consumer(s): sum g, sum g*xhat0 [, sum g*xhat1] (written only when asked for)
                sg += v;
                s0 += (v * (y0v - mu0)) * is0;
                s1 += v * y1v;       // second consumer: raw sum, centred and scaled after the loop (its mean / invstd are not kept live)
            }
            bst4(r_out, v, offb[k], 0);
            // one row's loads in flight at a time: hoisting the next row's five loads over this row's arithmetic does not fit the
            // 64 registers that keep eight waves per SIMD (the scheduler would spill instead of giving up the overlap)
            if (FUSED == 1) __builtin_amdgcn_sched_barrier(0);
        }
        if (USE_Y1) {   // sum g * xhat1 = invstd1 * (sum g * y1 - mean1 * sum g), per lane (everything after this is linear)
            const uint32_t cb = (uint32_t)col * 4u;
            const f32x4 mu1 = bld4(make_rsrc(p.emean1, has_y1 ? FULL : 0), cb, 0), is1 = bld4(make_rsrc(p.einv1, has_y1 ? FULL : 0), cb, 0);
            s1 = (s1 - mu1 * sg) * is1;
        }
    }
    if (FUSED && p.esum) {
        // rows of one channel quad: 4 lanes of the wave (lane bits 4, 5), then the 4 waves through LDS, in wave order
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sg[e] += __shfl_xor(sg[e], 16, 64); s0[e] += __shfl_xor(s0[e], 16, 64); s1[e] += __shfl_xor(s1[e], 16, 64);
            sg[e] += __shfl_xor(sg[e], 32, 64); s0[e] += __shfl_xor(s0[e], 32, 64); s1[e] += __shfl_xor(s1[e], 32, 64);
        }
        __syncthreads();   // every lane is done reading the transposed tile
        if (lane < 16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float* dst = smem + (wave * BN + c4 * 4 + e) * 3;
                dst[0] = sg[e]; dst[1] = s0[e]; dst[2] = s1[e];
            }
        }
        __syncthreads();
        if (tid < BN) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) { const float* s = smem + (wv * BN + tid) * 3; a0 += s[0]; a1 += s[1]; a2 += s[2]; }
            const size_t prow = (size_t)(cls * p.MT + mt) * p.Cin + n0 + tid;
            const size_t pstride = (size_t)p.eP * p.Cin;
            p.esum[prow] = a0;
            p.esum[pstride + prow] = a1;
            if (p.ey1) p.esum[2 * pstride + prow] = a2;
        }
    }
}

// ======================================================================================================
// Input gradient. Stride-s convolutions are decomposed into s*s parity classes of input pixels; each
// class only visits the filter taps that can reach it, so no MFMA work is spent on structural zeros.
// blockIdx.y = class. GEMM N = Cin, K = (taps of the class) x Cout.
// ======================================================================================================
// KS: the launch carries a K-split tail (stride 1 only; see tile_of_block_split), its own instantiation like k_conv_fwd's
// POOL: pool-mode reductions in the epilogue (osi_dgrad_fusion.pool_idx) — one launch per step, its own instantiation so that the
// workhorse keeps its 8 waves per SIMD
// W3 (3x3, stride 1, pad 1, 64x64 tile): row windows as in k_conv_fwd — for a fixed tap row jr the taps js = 0, 1, 2 read the same dY rows
// shifted by one pixel (dY pixel of (m, jr, js) = m + (1 - jr) W + (1 - js)): one window per (jr, 32-cout slice) in column-padded
// coordinates, the A fragment of a lane read at its base row + 2 - js. K-tile order (jr, slice, js).
template <int WM, int WN, int NST, int FUSED, bool KS = false, bool POOL = false, bool W3 = false>
__global__ __launch_bounds__(256, NST == 1 ? (WM * WN >= 4 ? 3 : (WM * WN == 1 ? 8 : 4)) : 2) void k_conv_dgrad(ConvP p, int Hc0, int Wc0) {
    static_assert(!KS || (WM == 1 && WN == 1 && NST == 1), "the K-split tail is built for the single-buffered 64x64 tile");
    static_assert(!W3 || (WM == 1 && WN == 1 && NST == 1 && !POOL), "row windows: the single-buffered 64x64 tile");
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int AROWS = W3 ? W3_WROWS : BM;
    constexpr int AR = W3 ? 3 : BM / 32;
    constexpr int LDC = BN + 4;
    constexpr int BV = BN / 4;            // float4 per k-row of the weight tile
    constexpr int BRP = 256 / BV;         // k-rows covered per pass
    constexpr int BRN = BK / BRP;         // passes
    constexpr int STAGE = AROWS * LDR + BK * LDC;
    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef OSI_STAMPS
    const size_t sidx = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
#endif
    OSI_STAMP(p, sidx, 0); OSI_STAMP_ID(p, sidx);

    int mt, nt, ks = -1;
    if (KS) { if (!tile_of_block_split(p, blockIdx.x, mt, nt, ks)) return; }
    else if (!tile_of_block(blockIdx.x, p.MT, p.NT, mt, nt)) return;
    const int st = p.stride;
    const int cls = blockIdx.y, ph = cls / st, pw = cls - ph * st;
    // class grid: pixels h = ph + st*h2 < H
    const int Hc = (p.H - ph + st - 1) / st, Wc = (p.W - pw + st - 1) / st;
    const int Mc = p.B * Hc * Wc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = mt * BM, n0 = nt * BN;
    if (m0 >= Mc) {  // a row tile beyond this (smaller) parity class: contributes nothing, but its partial row must read as zero
        if (p.esum && tid < BN) {
            const size_t prow = (size_t)(cls * p.MT + mt) * p.Cin + n0 + tid, pstride = (size_t)p.eP * p.Cin;
            p.esum[prow] = 0.f; p.esum[pstride + prow] = 0.f;
            if (p.ey1) p.esum[2 * pstride + prow] = 0.f;
        }
        return;
    }
    const int kq = tid & 7, lr = tid >> 3;

    // taps reaching this class: r = rb + st*jr, (h + pad - r) / st = hb - jr
    const int rb = (ph + p.pad) % st, sb = (pw + p.pad) % st;
    const int nR = rb < p.R ? (p.R - rb + st - 1) / st : 0;
    const int nS = sb < p.S ? (p.S - sb + st - 1) / st : 0;
    if (p.skip_empty && nR * nS == 0) return;   // sparse form: no tap reaches this parity class, its pixels are not written at all

    const FastDiv dHW = p.cHW[cls], dW = p.cW[cls];
    // a_base = offset of the dY pixel reached through the class's first tap (jr = js = 0); a_taps = bit (jr*nS+js) set when
    // tap (jr, js) lands inside dY: one add and one bit test per load (see k_conv_fwd)
    int a_base[AR];
    unsigned a_taps[AR];
    bool a_ok[AR];
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes), rw = make_rsrc(p.w, p.w_bytes);
    if constexpr (W3) {
        // window slot u = lr + 32 i <-> padded index q(m0) - 1 + u (see k_conv_fwd): a_base = byte offset of the slot's pixel in dY for
        // jr = 1, with bit jr set in its low bits when row h + 1 - jr is inside the image; OOB: pad slot / past the tensor / past the window
        const uint32_t Wp = (uint32_t)p.W + 1u, q0 = (uint32_t)m0 + fdiv((uint32_t)m0, dW);
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int u = lr + 32 * i, Q = (int)q0 - 1 + u;
            bool ok = u < W3_WROWS && Q >= 0;
            const uint32_t Qq = ok ? (uint32_t)Q : 0u, irow = fdiv(Qq, p.cW[1]), col = Qq - irow * Wp, mm = irow * (uint32_t)p.W + col;
            ok = ok && col < (uint32_t)p.W && mm < (uint32_t)Mc;
            const uint32_t mq = ok ? mm : 0u, b = fdiv(mq, dHW), h = fdiv(mq - b * dHW.d, dW);
            a_ok[i] = ok; a_taps[i] = 0;
            a_base[i] = ok ? (int)(((mq * (uint32_t)p.Cout + kq * 4) * 4) | ((int)h < p.H - 1 ? 1u : 0u) | 2u | (h > 0 ? 4u : 0u)) : (int)OOB;
        }
    } else
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        int m = m0 + lr + 32 * i;
        a_ok[i] = m < Mc;
        uint32_t mm = a_ok[i] ? (uint32_t)m : 0u;
        if (p.unit) {  // dY row = pixel index, never out of bounds (invalid rows of the last tile: OOB sentinel -> zeros)
            a_base[i] = a_ok[i] ? ((int)mm * p.Cout + kq * 4) * 4 : (int)OOB; a_taps[i] = a_ok[i] ? 1u : 0u;
        } else {
            uint32_t b = fdiv(mm, dHW);
            uint32_t rem = mm - b * dHW.d;
            uint32_t h2 = fdiv(rem, dW), w2 = rem - h2 * dW.d;
            const int hb = (ph + st * (int)h2 + p.pad - rb) / st, wb = (pw + st * (int)w2 + p.pad - sb) / st;
            a_base[i] = ((((int)b * p.Ho + hb) * p.Wo + wb) * p.Cout + kq * 4) * 4;   // BYTE offset
            a_taps[i] = 0;
            for (int j = 0; j < nR; ++j)
                for (int k = 0; k < nS; ++k)
                    if ((unsigned)(hb - j) < (unsigned)p.Ho && (unsigned)(wb - k) < (unsigned)p.Wo) a_taps[i] |= 1u << (j * nS + k);
            if (!a_ok[i]) a_taps[i] = 0;
        }
    }
    (void)Hc0; (void)Wc0;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;

    const int KC = p.Cout / BK;          // K tiles per tap
    const int Tall = nR * nS * KC;
    const int t0 = (!KS || ks < 0) ? 0 : ks * p.ks_T;            // this workgroup's K tiles [t0, T)
    const int T = (!KS || ks < 0) ? Tall : min(Tall, t0 + p.ks_T);
    int jr = 0, js = 0, c0 = 0;
    if (KS && ks > 0) {
        if constexpr (W3) {            // tile t = (jr, slice, js)
            const int rem = t0 % (3 * KC);
            jr = t0 / (3 * KC); c0 = (rem / 3) * BK; js = rem % 3;
        } else {
            const int tap = t0 / KC;
            c0 = (t0 - tap * KC) * BK; jr = tap / nS; js = tap - jr * nS;
        }
    }
    int cjs = js;            // W3: tap column of the tile the MFMAs are working on (the loader state runs one tile ahead)
    bool ld_new = false, first = true;     // W3: the loaded tile starts a new window (its dY rows sit in ra)
    int prow = 0;            // W3: window row of this lane's pixel for js = 2 (its padded index relative to the tile's first)
    if constexpr (W3) {
        const uint32_t m = (uint32_t)min(m0 + wm * 32 + (lane & 31), Mc - 1);
        prow = (int)(m - (uint32_t)m0 + fdiv(m, dW) - fdiv((uint32_t)m0, dW));
    }
    f32x4 ra[AR], rbv[BRN];
    const int bk_row = tid / BV, bk_col = (tid % BV) * 4;

    auto gload = [&]() {
        const int r = rb + st * jr, s = sb + st * js;
        if constexpr (W3) {
            ld_new = js == 0 || first;
            first = false;
            if (ld_new) {
#pragma unroll
                for (int i = 0; i < AR; ++i) {
                    const bool ok = ((uint32_t)a_base[i] >> jr) & 1;
                    ra[i] = bld4(rx, ok ? (uint32_t)((a_base[i] & ~15) + ((1 - jr) * p.W * p.Cout + c0) * 4) : OOB, 0);
                }
            }
        } else
        if (p.unit) {   // the form is chosen once per tile, outside the unrolled loop (see k_conv_fwd's gload)
#pragma unroll
            for (int i = 0; i < AR; ++i) ra[i] = bld4(rx, (uint32_t)a_base[i], (uint32_t)(c0 * 4));
        } else {
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const bool ok = (a_taps[i] >> (jr * nS + js)) & 1;
                ra[i] = bld4(rx, ok ? (uint32_t)(a_base[i] + (c0 - (jr * p.Wo + js) * p.Cout) * 4) : OOB, 0);
            }
        }
        // weight tile: rows k = cout c0..c0+31, cols = cin n0..n0+BN-1 at tap (r,s); the tap / cout part is wave-uniform
        const uint32_t wsoff = (uint32_t)((c0 * p.Ktot + (r * p.S + s) * p.Cin) * 4);
#pragma unroll
        for (int i = 0; i < BRN; ++i)
            rbv[i] = bld4(rw, (uint32_t)(((bk_row + BRP * i) * p.Ktot + n0 + bk_col) * 4), wsoff);
    };
    auto advance = [&]() {
        if constexpr (W3) {
            if (++js == 3) { js = 0; c0 += BK; if (c0 == p.Cout) { c0 = 0; ++jr; } }
        } else {
            c0 += BK;
            if (c0 == p.Cout) { c0 = 0; if (++js == nS) { js = 0; ++jr; } }
        }
    };
    auto sstore = [&](int buf) {
        float* sA = smem + buf * STAGE;
        float* sB = sA + AROWS * LDR;
        if constexpr (W3) {
            if (ld_new) {
#pragma unroll
                for (int i = 0; i < AR; ++i)
                    if (lr + 32 * i < W3_WROWS) *reinterpret_cast<f32x4*>(sA + (lr + 32 * i) * LDR + kq * 4) = ra[i];
            }
        } else
#pragma unroll
        for (int i = 0; i < AR; ++i) *reinterpret_cast<f32x4*>(sA + (lr + 32 * i) * LDR + kq * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < BRN; ++i) *reinterpret_cast<f32x4*>(sB + (bk_row + BRP * i) * LDC + bk_col) = rbv[i];
    };

    if (T > t0) {
        gload(); advance();
        sstore(0);
        __syncthreads();
        OSI_STAMP(p, sidx, 1);
            for (int t = t0; t < T; ++t) {
            const int buf = NST == 2 ? ((t - t0) & 1) : 0;
                if (!(OSI_ABLATE & 1)) if (t + 1 < T) { gload(); advance(); }
            const float* sA = smem + buf * STAGE;
            if (OSI_ABLATE & 16) {
#pragma unroll
                for (int j = 0; j < 16; ++j)
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int n = 0; n < WN; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[i % AR][j & 3], rbv[n % BRN][j & 3], acc[i][n], 0, 0, 0);
            } else if constexpr (W3) {
                const int h4 = (lane >> 5) * 4;
                const float* pa = sA + (prow + 2 - cjs) * LDR + h4;
                const float* pb = sA + AROWS * LDR + h4 * LDC + wn * 32 + (lane & 31);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(pa + 8 * j);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], pb[(8 * j + e) * LDC], acc[0][0], 0, 0, 0);
                }
                if (++cjs == 3) cjs = 0;
            } else
            mma_RC<WM, WN, LDC>(sA, sA + BM * LDR, wm * 32 * WM, wn * 32 * WN, lane, acc);
            if (!(OSI_ABLATE & 4)) if (NST == 1) __syncthreads();
            if (!(OSI_ABLATE & 2)) if (t + 1 < T) sstore(NST == 2 ? (buf ^ 1) : 0);
            if (!(OSI_ABLATE & 4)) __syncthreads();
        }
        OSI_STAMP(p, sidx, 2);
    }
    if (OSI_ABLATE & 32) { if (acc[0][0][0] == 123.456f) p.y[tid] = acc[0][0][1]; return; }

    if constexpr (WM == 1 && WN == 1) {
        // Vectorised epilogue of the 64x64 tile (the form the executor uses): the accumulators are transposed through LDS so
        // that every lane owns 4 consecutive channels of a pixel — addend, activation(s) and output move as 16-byte accesses
        // (a quarter of the memory instructions of the accumulator layout), the ReLU mask words of a float4 are 32 contiguous
        // bytes, and the BatchNorm column sums start as float4 per lane.
        constexpr int LDT = BN + 4;
        static_assert(BM * LDT <= STAGE, "transposed tile must fit the operand stage");
        float* tile = smem;   // the K loop ended with a barrier: LDS is free
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            // rows past the end of the tensor must be exact zeros for the epilogue's column sums: the per-tap loader gets that from its
            // range-checked loads, the row-window form reads SOME window row for such a lane and clears the result here
            const bool live = !W3 || m0 + wm * 32 + acc_row(rr, lane) < Mc;
            tile[(wm * 32 + acc_row(rr, lane)) * LDT + wn * 32 + (lane & 31)] = live ? acc[0][0][rr] : 0.f;
        }
        __syncthreads();
        if (KS && ks >= 0) {   // K split of a remainder tile: raw accumulators to the slab, the epilogue runs in k_conv_dgrad_tail_fixup
            float* dst = p.ks_slab + ((size_t)((mt - p.MT1) * p.NT + nt) * p.ks_S + ks) * (BM * BN);
            const int c4 = tid & 15, rg = tid >> 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rl = rg + 16 * k;
                *reinterpret_cast<f32x4*>(dst + rl * BN + c4 * 4) = *reinterpret_cast<const f32x4*>(tile + rl * LDT + c4 * 4);
            }
            OSI_STAMP(p, sidx, 3);
            return;
        }
        dgrad_epilogue64<FUSED, POOL>(p, smem, cls, mt, m0, n0, Mc, st, ph, pw, dHW, dW,
                                [&](int rl, int c4) { return *reinterpret_cast<const f32x4*>(tile + rl * LDT + c4 * 4); });
        OSI_STAMP(p, sidx, 3);
        return;
    }

    // epilogue. Element offsets fit 31 bits (desc_ok). Rows are handled four at a time so that the fused form (addend + mask +
    // BatchNorm reductions) keeps few values live: the register footprint of the epilogue sets the occupancy of the K loop.
    float esg[WN], es0[WN], es1[WN];
#pragma unroll
    for (int n = 0; n < WN; ++n) { esg[n] = 0.f; es0[n] = 0.f; es1[n] = 0.f; }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // accumulator registers 4q..4q+3 = rows 8q + 4*(lane>>5) + 0..3
            uint32_t roff[4];
            bool rok[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int m = m0 + wm * 32 * WM + i * 32 + acc_row(4 * q + e, lane);
                rok[e] = m < Mc;
                uint32_t mm = rok[e] ? (uint32_t)m : 0u;
                uint32_t pix;
                if (st == 1) pix = mm;
                else {
                    uint32_t b = fdiv(mm, dHW);
                    uint32_t rem = mm - b * dHW.d;
                    uint32_t h2 = fdiv(rem, dW), w2 = rem - h2 * dW.d;
                    pix = (b * p.H + (ph + st * h2)) * p.W + (pw + st * w2);
                }
                roff[e] = pix * p.Cin;
            }
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                const int col = n0 + wn * 32 * WN + n * 32 + (lane & 31);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][n][4 * q + e];
                if (p.addend) {  // skip-connection sum (may be the output buffer itself: read-modify-write by the same lane)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rok[e] ? p.addend[roff[e] + col] : 0.f;
                }
                if (FUSED) {
                    if (p.ebits) {  // ReLU mask of the activation this gradient belongs to (1 bit / element, see bn.hip)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t i4 = (roff[e] + col) >> 2;
                            const unsigned long long w = rok[e] ? p.ebits[(size_t)(i4 >> 6) * 4 + (col & 3)] : 0ull;
                            v[e] = (w >> (i4 & 63)) & 1 ? v[e] : 0.f;
                        }
                    }
                    if (p.escale0) {  // ReLU gate recomputed from the producer's pre-BN tensor (see the 64x64 form above)
                        const float gsc = p.escale0[col], gsh = p.eshift0[col];
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            v[e] = (rok[e] && __builtin_fmaf(p.ey0[roff[e] + col], gsc, gsh) > 0.f) ? v[e] : 0.f;
                    }
                    if (p.esum) {   // BatchNorm-backward partial sums of the consumer(s): sum g, sum g*xhat0 [, sum g*xhat1]
                        const float mu0 = p.emean0[col], is0 = p.einv0[col];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float gv = rok[e] ? v[e] : 0.f;
                            esg[n] += gv;
                            es0[n] += gv * ((rok[e] ? p.ey0[roff[e] + col] : mu0) - mu0) * is0;
                        }
                        if (p.ey1) {
                            const float mu1 = p.emean1[col], is1 = p.einv1[col];
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                es1[n] += (rok[e] ? v[e] : 0.f) * ((rok[e] ? p.ey1[roff[e] + col] : mu1) - mu1) * is1;
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (rok[e]) p.y[roff[e] + col] = v[e];
            }
        }
    }
    if (FUSED && p.esum) {  // lane halves, then the two wave rows through LDS: one row of partials per workgroup
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            esg[n] += __shfl_xor(esg[n], 32, 64); es0[n] += __shfl_xor(es0[n], 32, 64); es1[n] += __shfl_xor(es1[n], 32, 64);
        }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < WN; ++n)
            if (lane < 32) {
                float* dst = smem + (wm * BN + wn * 32 * WN + n * 32 + lane) * 3;
                dst[0] = esg[n]; dst[1] = es0[n]; dst[2] = es1[n];
            }
        __syncthreads();
        if (tid < BN) {
            const float* a = smem + tid * 3;
            const float* b = smem + (BN + tid) * 3;
            const size_t prow = (size_t)(cls * p.MT + mt) * p.Cin + n0 + tid;
            const size_t pstride = (size_t)p.eP * p.Cin;
            p.esum[prow] = a[0] + b[0];
            p.esum[pstride + prow] = a[1] + b[1];
            if (p.ey1) p.esum[2 * pstride + prow] = a[2] + b[2];
        }
    }
}

// ======================================================================================================
// Weight gradient. GEMM rows = cout, cols = (tap, cin range), K = pixels. blockIdx.y = K split.
// STEM: Cin = 4 (padded RGB), 56 padded taps -> 224 columns, a column tile spans 16 taps.
// ======================================================================================================
// XF: the X operand is relu(x * in_scale[c] + in_shift[c]) (the producer layer's BatchNorm + ReLU applied in the loader)
template <int WM, int WN, bool STEM, int NST, bool XF = false>
// Register budget = what the split-K plan can keep resident anyway (plan_wgrad sizes a launch for two 128x128, four 128x64 or eight
// 64x64 workgroups per CU): 3 / 4 / 7-8 waves per SIMD. Asking for more only forces spills into the K loop.
__global__ __launch_bounds__(256, NST == 1 ? (WM * WN == 4 ? 4 : (WM * WN == 2 ? 5 : 8)) : 2) void k_conv_wgrad(ConvP p) {
    static_assert(!(XF && STEM), "the stem reads the image, not an activation");
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int AV = BM / 4, ARP = 256 / AV, ARN = BK / ARP;
    constexpr int BV = BN / 4, BRP = 256 / BV, BRN = BK / BRP;
    constexpr int STAGE = BK * (LDA + LDB);
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
#ifdef OSI_STAMPS
    const size_t sidx = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
#endif
    OSI_STAMP(p, sidx, 0); OSI_STAMP_ID(p, sidx);
    // tiles: rows over cout (MT), cols over (tap, cin tile) (NT), K splits
    int mt, ntile, split;
    if (p.wgroup) {
        // XCD-aware mapping on a 1-D grid. Workgroups b and b+8 share an XCD (and its L2); the workgroups that read the same bytes
        // take CONSECUTIVE dispatch slots of ONE XCD, so the first pulls a chunk over the fabric and the others hit that L2:
        //   gkind 0 (splits >= 8)  key = K split:          all (cout tile, cin tile, tap) workgroups of a split share its dY and X chunks
        //   gkind 1                key = (split, cout tile): its (cin tile, tap) workgroups share the dY tile
        //   gkind 2                key = (split, cin tile):  its (cout tile, tap) workgroups share the X slice
        //   gkind 3                key = (split, cout tile, cin tile): only the R*S taps of a cell share (dY tile, X windows)
        const int taps = STEM ? 1 : p.R * p.S;
        const int ctiles = p.NT / taps;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int u = slot % p.ginner;
        const int key = (slot / p.ginner) * 8 + xcd;
        if (key >= p.gkeys) return;
        const int tp = u % taps, v = u / taps;
        int ct;
        if (p.gkind == 0) { split = key; mt = v % p.MT; ct = v / p.MT; }
        else if (p.gkind == 1) { split = key / p.MT; mt = key - split * p.MT; ct = v; }
        else if (p.gkind == 2) { split = key / ctiles; ct = key - split * ctiles; mt = v; }
        else { mt = key % p.MT; const int rest = key / p.MT; ct = rest % ctiles; split = rest / ctiles; }
        ntile = tp * ctiles + ct;
    } else {
        mt = blockIdx.x % p.MT; ntile = blockIdx.x / p.MT; split = blockIdx.y;
    }
    const int n0 = mt * BM;  // cout offset
    int tap = 0, c0 = 0, r = 0, s = 0;
    if (!STEM) {
        const int ctiles = p.Cin / BN;
        tap = ntile / ctiles; c0 = (ntile - tap * ctiles) * BN;
        r = tap / p.S; s = tap - r * p.S;
    }
    const int kbeg = split * p.kchunk;
    const int kend = min(p.M, kbeg + p.kchunk);
    const int T = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) acc[i][n][rr] = 0.f;

    const int a_row = tid / AV, a_col = (tid % AV) * 4;
    const int b_row = tid / BV, b_col = (tid % BV) * 4;
    f32x4 ra[ARN], rbv[BRN];

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes), rdy = make_rsrc(p.w, p.w_bytes);
    // This workgroup works on ONE tap, so the input pixel of every output pixel of its split is known up front: resolve the
    // (image, row, col) divisions and the padding test for a window of TW K-tiles into LDS (4 KiB, rebuilt every TW tiles);
    // the loader then costs one ds_read + one add per row.
    constexpr int TW = 32;
    uint32_t* tbl = reinterpret_cast<uint32_t*>(smem + NST * STAGE);
    const bool use_tbl = !STEM && p.wtbl;
    float* s_sc = smem + NST * STAGE + TW * BK;   // XF: scale | shift of this workgroup's BN input channels (behind the offset table)
    float* s_sh = s_sc + BN;
    if (XF) {
        if (tid < BN / 4) {
            *reinterpret_cast<f32x4*>(s_sc + tid * 4) = ld4(p.in_scale + c0 + tid * 4);
            *reinterpret_cast<f32x4*>(s_sh + tid * 4) = ld4(p.in_shift + c0 + tid * 4);
        }
        __syncthreads();
    }
    int ld_t = 0;   // XF: K tile sitting in rbv
    auto build_tbl = [&](int t0) {
        for (int i = tid; i < TW * BK; i += 256) {
            const int m = kbeg + t0 * BK + i;
            uint32_t off = OOB;
            if (m < kend) {
                uint32_t b = fdiv((uint32_t)m, p.dHoWo);
                uint32_t rem = (uint32_t)m - b * p.dHoWo.d;
                uint32_t ho = fdiv(rem, p.dWo);
                uint32_t wo = rem - ho * p.dWo.d;
                int hi = (int)ho * p.stride - p.pad + r, wi = (int)wo * p.stride - p.pad + s;
                if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                    off = (uint32_t)((((int)b * p.H + hi) * p.W + wi) * p.Cin * 4);
            }
            tbl[i] = off;
        }
        __syncthreads();
    };
    auto gload = [&](int t) {
        const int kb = kbeg + t * BK;
        ld_t = t;
#pragma unroll
        for (int i = 0; i < ARN; ++i) {   // dY rows: pixel part in the scalar offset, zero rows past the split's end via OOB
            const int m = kb + a_row + ARP * i;
            ra[i] = bld4(rdy, m < kend ? (uint32_t)(((a_row + ARP * i) * p.Cout + n0 + a_col) * 4) : OOB, (uint32_t)kb * p.Cout * 4);
        }
        // ONE straight-line path for both run-time forms (a branch per load, or per form, made the compiler wait for every
        // outstanding load before the next one: the destination registers of the other form's loads count as pending across the
        // join): 1x1 stride-1 takes the pixel offset from m itself, everything else from the rolling LDS table; the table read is
        // issued either way (zeros / stale entries in the 1x1 stride-1 form, selected away)
        if (!STEM) {
            uint32_t toff[BRN];
#pragma unroll
            for (int i = 0; i < BRN; ++i) toff[i] = tbl[(t % TW) * BK + b_row + BRP * i];
#pragma unroll
            for (int i = 0; i < BRN; ++i) {
                const int m = kb + b_row + BRP * i;
                const uint32_t uoff = m < kend ? (uint32_t)m * (uint32_t)(p.Cin * 4) : OOB;   // pixel m of the output = pixel m of the input
                rbv[i] = bld4(rx, (p.unit ? uoff : toff[i]) + (uint32_t)((c0 + b_col) * 4), 0);   // OOB + a column offset is still past the end
            }
        } else {
#pragma unroll
            for (int i = 0; i < BRN; ++i) {
                const int m = kb + b_row + BRP * i;
                bool ok = m < kend;
                uint32_t mm = ok ? (uint32_t)m : 0u;
                uint32_t b = fdiv(mm, p.dHoWo);
                uint32_t rem = mm - b * p.dHoWo.d;
                uint32_t ho = fdiv(rem, p.dWo);
                uint32_t wo = rem - ho * p.dWo.d;
                int rr = r, ss = s, coff = c0 + b_col;
                if (STEM) {
                    int tp = ntile * (BN / 4) + (b_col >> 2);
                    rr = tp / 7; ss = tp - rr * 7; coff = 0;
                    ok = ok && tp < 49;
                }
                int hi = (int)ho * p.stride - p.pad + rr, wi = (int)wo * p.stride - p.pad + ss;
                ok = ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                rbv[i] = bld4(rx, ok ? (uint32_t)(((((int)b * p.H + hi) * p.W + wi) * p.Cin + coff) * 4) : OOB, 0);
            }
        }
    };
    // XF: fused activation on the X tile sitting in rbv, in registers, right before the LDS store (see k_conv_fwd)
    auto xform = [&]() {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(s_sc + b_col), sh = *reinterpret_cast<const f32x4*>(s_sh + b_col);
#pragma unroll
        for (int i = 0; i < BRN; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) rbv[i][e] = fmaxf(__builtin_fmaf(rbv[i][e], sc[e], sh[e]), 0.f);
        }
        // A padding tap must read as zero AFTER the activation: select. Pixels past the split's end need nothing: their dY row
        // is zero (range-checked load), so whatever finite value relu(shift) leaves in the X row is multiplied away — the 1x1
        // stride-1 form has no padding and therefore no select at all (a uniform branch around arithmetic only: vector instructions
        // are paid in matrix-pipe time, profiles/r06_mfma_valu_coexec.txt).
        if (!p.unit) {
#pragma unroll
            for (int i = 0; i < BRN; ++i) {
                const bool ok = tbl[(ld_t % TW) * BK + b_row + BRP * i] != OOB;
                rbv[i] = ok ? rbv[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    auto sstore = [&](int buf) {
        float* sA = smem + buf * STAGE;
        float* sB = sA + BK * LDA;
        if (XF) xform();
#pragma unroll
        for (int i = 0; i < ARN; ++i) *reinterpret_cast<f32x4*>(sA + (a_row + ARP * i) * LDA + a_col) = ra[i];
#pragma unroll
        for (int i = 0; i < BRN; ++i) *reinterpret_cast<f32x4*>(sB + (b_row + BRP * i) * LDB + b_col) = rbv[i];
    };
    if (T > 0) {
        if (use_tbl) build_tbl(0);
        gload(0);
        sstore(0);
        __syncthreads();
        OSI_STAMP(p, sidx, 1);
        for (int t = 0; t < T; ++t) {
            const int buf = NST == 2 ? (t & 1) : 0;
            // every wave passed the barrier that ended tile t-1, i.e. finished reading the window that ends with tile t
            if (use_tbl && t + 1 < T && (t + 1) % TW == 0) build_tbl(t + 1);
            if (!(OSI_ABLATE & 1)) if (t + 1 < T) gload(t + 1);
            const float* sA = smem + buf * STAGE;
            if (OSI_ABLATE & 16) {
#pragma unroll
                for (int ks = 0; ks < BK / 2; ++ks)
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int n = 0; n < WN; ++n) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[0][i & 3], rbv[0][n & 3], acc[i][n], 0, 0, 0);
            } else
            mma_CC<WM, WN, LDA, LDB>(sA, sA + BK * LDA, wm * 32 * WM, wn * 32 * WN, lane, acc);
            if (!(OSI_ABLATE & 4)) if (NST == 1) __syncthreads();
            if (!(OSI_ABLATE & 2)) if (t + 1 < T) sstore(NST == 2 ? (buf ^ 1) : 0);
            if (!(OSI_ABLATE & 4)) __syncthreads();
        }
        OSI_STAMP(p, sidx, 2);
    }

    // mma_CC's interleaved operand mapping: element e of MFMA block i sits at 2 e + i of the wave's 64 when the wave spans two blocks
    float* out = p.y + (size_t)split * p.slab_stride;
    const int colbase = STEM ? ntile * BN : tap * p.Cin + c0;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int row = n0 + wm * 32 * WM + (WM == 2 ? 2 * acc_row(rr, lane) + i : acc_row(rr, lane));
            if constexpr (WN == 2) {      // the lane's two columns are neighbours: one 8-byte store (Ktot and every column base are even)
                const int col = colbase + wn * 64 + 2 * (lane & 31);
                if (col < p.Ktot) *reinterpret_cast<f32x2*>(out + (size_t)row * p.Ktot + col) = f32x2{acc[i][0][rr], acc[i][1][rr]};
            } else {
                const int col = colbase + wn * 32 + (lane & 31);
                if (col < p.Ktot) out[(size_t)row * p.Ktot + col] = acc[i][0][rr];
            }
        }
    OSI_STAMP(p, sidx, 3);
}

// ======================================================================================================
// Weight gradient of a 3x3 / stride 1 / pad 1 convolution with ALL NINE TAPS in one workgroup.
// The per-tap kernel above streams the dY tile once per tap (nine workgroups per cell). Here one workgroup owns a (64 cout x 32 cin)
// cell of a K split and walks the split's pixels in runs of 32: per run it stages the dY rows ONCE and ONE window of X rows
// (32 + 2W + 2 flattened pixels: for stride 1 / pad 1 the input pixel of (output pixel q, tap r,s) is q + (r-1) W + (s-1) in the SAME
// flattened index, so a tap is a constant row shift inside the window) and feeds nine accumulator sets from them. The image-border
// zero padding is a 9-bit validity mask per pixel (built per run, one table read per K step) applied to the X operand.
// MFMA: v_mfma_f32_16x16x4_f32 (same exact-fp32 rate as 32x32x2) so that a wave's 32 cout x 16 cin tile x 9 taps fits 72 accumulator
// registers; the 18 MFMAs of a K step are independent (no accumulator latency exposed). dY traffic / 9, X traffic / (9 / window
// amplification), LDS reads 12 per 18 MFMAs.
// ======================================================================================================
constexpr int W3_BKP = 32;                 // pixels per K tile
constexpr int W3_LDA = 64 + 16;            // dY image row stride (floats): 16 consecutive cout x 4 pixels per ds_read_b32 -> two pixel rows
constexpr int W3_LDB = 32 + 16;            //   land on disjoint bank halves when the stride is 16 mod 32
// S2: stride 2 (pad 1, even H and W). The input splits into four parity sub-grids X_pq[b][i][j] = X[b][2 i + p][2 j + q], each with the
// OUTPUT's geometry; tap (r, s) reads sub-grid (p, q) = (r != 1, s != 1) at (ho + di, wo + dj) with di = -1 for r = 0 and dj = -1 for
// s = 0, else 0 — again a constant shift of the flattened (sub-grid) pixel index. The window is therefore four runs laid end to end
// in LDS: ee (32 rows, tap (1,1)), eo (33: taps (1,0), (1,2)), oe (32 + Wo: (0,1), (2,1)), oo (33 + Wo: the four corner taps) =
// 130 + 2 Wo rows of which each is ONE 128-byte piece of the tensor (its sub-grid pixel decomposed per row and run); the nine tap
// offsets become nine row offsets into that layout and the validity mask only has to clear r = 0 at ho = 0 and s = 0 at wo = 0.
template <int NWIN, bool XF, bool S2 = false>   // NWIN: window passes of 32 rows (2: W <= 15, 3: W <= 31, 5: W <= 63; S2: 5 or 6)
__global__ __launch_bounds__(256, 3) void k_conv_wgrad3(ConvP p) {   // 72 accumulator + staging registers: 3 waves per SIMD, no spills
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn2 = wave >> 1, wc = wave & 1;
    OSI_STAMP(p, blockIdx.x, 0); OSI_STAMP_ID(p, blockIdx.x);
    const int W = p.W, WIN = S2 ? 130 + 2 * p.Wo : W3_BKP + 2 * W + 2;
    float* sA = smem;                                   // [32][W3_LDA]  dY rows of the run
    float* sB = sA + W3_BKP * W3_LDA;                   // [WIN][W3_LDB] X window
    uint32_t* sM = reinterpret_cast<uint32_t*>(sB + NWIN * 32 * W3_LDB);   // [32] tap-validity bits per pixel of the run
    // block -> (cout tile, cin tile, split): whole K splits per XCD (see k_conv_wgrad)
    const int ctiles = p.Cin / 32;
    int mt, ct, split;
    {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int u = slot % p.ginner, key = (slot / p.ginner) * 8 + xcd;
        if (key >= p.gkeys) return;
        if (p.gkind == 0) { split = key; mt = u % p.MT; ct = u / p.MT; }
        else { mt = key % p.MT; const int rest = key / p.MT; ct = rest % ctiles; split = rest / ctiles; }
    }
    const int n0 = mt * 64, c0 = ct * 32;
    const int kbeg = split * p.kchunk, kend = min(p.M, kbeg + p.kchunk);
    const int T = kend > kbeg ? (kend - kbeg + W3_BKP - 1) / W3_BKP : 0;

    f32x4acc acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[t][i] = f32x4acc{0.f, 0.f, 0.f, 0.f};

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes), rdy = make_rsrc(p.w, p.w_bytes);
    const int a_row = tid >> 4, a_c4 = tid & 15;        // dY: 16 float4 per pixel row, 16 rows per pass, 2 passes
    const int b_row = tid >> 3, b_c4 = tid & 7;         // X window: 8 float4 per row, 32 rows per pass, NWIN passes
    f32x4 ra[2], rb[NWIN];
    f32x4 xsc = {0, 0, 0, 0}, xsh = xsc;
    if (XF) { xsc = ld4(p.in_scale + c0 + b_c4 * 4); xsh = ld4(p.in_shift + c0 + b_c4 * 4); }
    // S2: which sub-grid run this thread's window rows belong to and their offset from the run's first output pixel (fixed per thread)
    int s2_off[NWIN], s2_pq[NWIN], s2_c[NWIN];
    if (S2) {
        const int Wo = p.Wo;
#pragma unroll
        for (int i = 0; i < NWIN; ++i) {
            const int j = b_row + 32 * i;
            if (j < 32) { s2_pq[i] = 0; s2_off[i] = j; }
            else if (j < 65) { s2_pq[i] = 1; s2_off[i] = j - 32 - 1; }
            else if (j < 97 + Wo) { s2_pq[i] = 2; s2_off[i] = j - 65 - Wo; }
            else { s2_pq[i] = 3; s2_off[i] = j - (97 + Wo) - Wo - 1; }
            if (j >= WIN) s2_pq[i] = -1;
            s2_c[i] = (s2_pq[i] >> 1) * p.W + (s2_pq[i] & 1);
        }
    }

    auto gload = [&](int t) {
        const int q0 = kbeg + t * W3_BKP;
#pragma unroll
        for (int i = 0; i < 2; ++i) {      // rows past the split's end read as zero (sentinel offset): their products vanish
            const int q = q0 + a_row + 16 * i;
            ra[i] = bld4(rdy, q < kend ? (uint32_t)(((a_row + 16 * i) * p.Cout + n0 + a_c4 * 4) * 4) : OOB, (uint32_t)q0 * p.Cout * 4);
        }
        if constexpr (S2) {
#pragma unroll
            for (int i = 0; i < NWIN; ++i) {
                const int sq = q0 + s2_off[i];            // flattened pixel of the (p, q) sub-grid = an output-grid index
                const bool in = s2_pq[i] >= 0 && sq >= 0 && sq < p.M;
                // sub-grid pixel sq = (b Ho + hi) Wo + wi sits at input pixel (b H + 2 hi + p) W + 2 wi + q = 4 sq - 2 wi + (p W + q) for
                // H = 2 Ho, W = 2 Wo: ONE division (by Wo) per window row
                const uint32_t sqq = in ? (uint32_t)sq : 0u;
                const uint32_t wi = sqq - fdiv(sqq, p.dWo) * p.dWo.d;
                const uint32_t pix = 4 * sqq - 2 * wi + (uint32_t)s2_c[i];
                rb[i] = bld4(rx, in ? (pix * p.Cin + c0 + b_c4 * 4) * 4 : OOB, 0);
            }
        } else {
        const long wq0 = (long)q0 - W - 1;  // flattened input pixel of window row 0 (negative / past the end: range check -> zeros, masked anyway)
#pragma unroll
        for (int i = 0; i < NWIN; ++i) {
            const int j = b_row + 32 * i;
            const long wq = wq0 + j;
            const bool in = j < WIN && wq >= 0 && wq < (long)p.M;
            rb[i] = bld4(rx, in ? (uint32_t)((wq * p.Cin + c0 + b_c4 * 4) * 4) : OOB, 0);
        }
        }
    };
    uint32_t mbits = 0;
    auto sstore = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(sA + (a_row + 16 * i) * W3_LDA + a_c4 * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < NWIN; ++i) *reinterpret_cast<f32x4*>(sB + (b_row + 32 * i) * W3_LDB + b_c4 * 4) = rb[i];
        if (tid < W3_BKP) sM[tid] = mbits;
    };
    // register-side half of the staging, run BEFORE the barrier that frees the LDS images (the loads were issued a whole run earlier):
    // the fused input activation in place and the tap-validity bits of the next run; between the two barriers only LDS stores are left
    auto xform = [&](int t) {
        if (XF) {      // rows outside the tensor become relu(shift) garbage, which the tap mask removes
#pragma unroll
            for (int i = 0; i < NWIN; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[i][e] = fmaxf(__builtin_fmaf(rb[i][e], xsc[e], xsh[e]), 0.f);
        }
        if (tid < W3_BKP) {                 // tap validity of pixel q: bit (3 r + s) set when (h + r - 1, w + s - 1) is inside the image
            const int q = kbeg + t * W3_BKP + tid;
            uint32_t bits = 0;
            if (q < kend) {
                const uint32_t b = fdiv((uint32_t)q, p.dHoWo);
                const uint32_t rem = (uint32_t)q - b * p.dHoWo.d;
                const int h = (int)fdiv(rem, p.dWo), w = (int)(rem - fdiv(rem, p.dWo) * p.dWo.d);
                // S2 (h, w = output row / column): only the top / left taps can leave the image (2 ho + 1 <= H - 1 for even H)
                const uint32_t rowm = (h > 0 ? 1u : 0u) | 2u | ((S2 || h < p.H - 1) ? 4u : 0u);     // r = 0, 1, 2
                const uint32_t colm = (w > 0 ? 1u : 0u) | 2u | ((S2 || w < p.W - 1) ? 4u : 0u);     // s = 0, 1, 2
#pragma unroll
                for (int r = 0; r < 3; ++r) if ((rowm >> r) & 1) bits |= colm << (3 * r);
            }
            mbits = bits;
        }
    };

    const int l15 = lane & 15, lk = lane >> 4;
    // Row of tap (r, s) inside the window = j * Wx + imm with a compile-time imm: stride 1: r * W + s; stride 2 (the four sub-grid runs
    // laid end to end, see above): ee at 0, eo at 32, oe at 65, oo at 97 + Wo, plus Wo for r = 2 and 1 for s = 2. Three base pointers
    // (j = 0, 1, 2) per lane and every LDS read of a run is base + immediate: no address arithmetic inside the K loop.
    const int Wx = S2 ? p.Wo : W;
    const float* const bp0 = sB + lk * W3_LDB + 16 * wc + l15;
    const float* const bj[3] = {bp0, bp0 + Wx * W3_LDB, bp0 + 2 * Wx * W3_LDB};
    const float* const ap0 = sA + lk * W3_LDA + 32 * wn2 + l15;
    if (T > 0) {
        gload(0);
        xform(0);
        sstore(0);
        __syncthreads();
        OSI_STAMP(p, blockIdx.x, 1);
        for (int t = 0; t < T; ++t) {
            if (!(OSI_ABLATE & 1)) if (t + 1 < T) gload(t + 1);
            // K steps of 4 pixels, software-pipelined by hand: the 12 LDS reads of step k+1 are issued before the 18 MFMAs of step k
            // (two register sets, ping-pong). Left to itself the compiler either waits for each read right before its MFMA (rolled
            // loop) or hoists every read of the run and spills the accumulators (unrolled loop).
            struct Ops { float a0, a1, b[9]; uint32_t m; };
            auto lds_ops = [&](int ks, Ops& o) {
                if (OSI_ABLATE & 16) {     // operands stay in registers: a cheap VALU touch keeps the loop from collapsing
                    o.a0 = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, o.a0) ^ (uint32_t)ks); return;
                }
                o.a0 = ap0[4 * ks * W3_LDA]; o.a1 = ap0[4 * ks * W3_LDA + 16];     // this lane's pixel of the K step: 4 ks + lk
                if (!(OSI_ABLATE & 8)) o.m = sM[4 * ks + lk];
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int r = tp / 3, sx = tp % 3;
                    const int jb = !S2 ? r : (r == 2 ? 1 : 0) + (r != 1 && sx != 1 ? 1 : 0);
                    const int imm = !S2 ? sx : (r != 1 ? (sx != 1 ? 97 : 65) : (sx != 1 ? 32 : 0)) + (sx == 2 ? 1 : 0);
                    o.b[tp] = bj[jb][(4 * ks + imm) * W3_LDB];
                }
            };
            // The validity of (pixel, tap (r, s)) is rowm(pixel, r) & colm(pixel, s): the row part goes onto the dY operand (two masked
            // copies of a0 / a1), the column part onto the six X operands of s = 0 and s = 2 — four sign-extended bit fields and ten
            // v_and per K step instead of a bit test, a compare and a select per tap (27; tools/probes/wgrad3_ablate.sh priced the
            // tap mask at 17 - 20 % of the kernel). Consequence, accepted: a masked-out tap of the centre column (s = 1) multiplies 0 (dY side) by
            // the X value instead of dY by 0 (X side), so a non-finite activation (Inf / NaN) at a pixel whose tap row is out of the image
            // puts NaN into that weight gradient where the select form gave 0 — as the reference's own conv2d_weight does for a non-finite
            // input (0 * Inf): a network whose activations are finite (every tested one) sees no difference.
            auto mma_ops = [&](const Ops& o) {
                const uint32_t mr0 = (uint32_t)((int)(o.m << 30) >> 31), mr2 = (uint32_t)((int)(o.m << 24) >> 31);   // bits 1, 7: taps (0,1), (2,1)
                const uint32_t ms0 = (uint32_t)((int)(o.m << 28) >> 31), ms2 = (uint32_t)((int)(o.m << 26) >> 31);   // bits 3, 5: taps (1,0), (1,2)
                auto band = [](float v, uint32_t m) { return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) & m); };
                float a0r[3] = {band(o.a0, mr0), o.a0, band(o.a0, mr2)}, a1r[3] = {band(o.a1, mr0), o.a1, band(o.a1, mr2)};
                if (OSI_ABLATE & 8) { a0r[0] = a0r[2] = o.a0; a1r[0] = a1r[2] = o.a1; }
                float bm[9];
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) bm[tp] = (OSI_ABLATE & 8) || tp % 3 == 1 ? o.b[tp] : band(o.b[tp], tp % 3 == 0 ? ms0 : ms2);
                // all fourteen mask instructions first, then eighteen MFMAs back to back: an MFMA that reads the result of the VALU
                // instruction in front of it waits for it (the scheduler otherwise pairs each v_and with its consumer)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    acc[tp][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0r[tp / 3], bm[tp], acc[tp][0], 0, 0, 0);
                    acc[tp][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1r[tp / 3], bm[tp], acc[tp][1], 0, 0, 0);
                }
            };
            if constexpr (!S2) {
                Ops o0, o1;
                if (OSI_ABLATE & 16) {
                    o0.a0 = ra[0][0]; o0.a1 = ra[1][1]; o0.m = 0x1ff;
#pragma unroll
                    for (int tp = 0; tp < 9; ++tp) o0.b[tp] = rb[tp % NWIN][tp & 3];
                    o1 = o0;
                }
                lds_ops(0, o0);
#pragma unroll
                for (int ks = 0; ks < W3_BKP / 4; ks += 2) {      // unrolled: every LDS address is a base register + an immediate
                    lds_ops(ks + 1, o1);
                    mma_ops(o0);
                    __builtin_amdgcn_sched_barrier(0);           // (keeps the reads of later steps from being hoisted: they would spill)
                    if (ks + 2 < W3_BKP / 4) lds_ops(ks + 2, o0);
                    mma_ops(o1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {     // stride 2 stages 20 - 24 registers of X rows and their sub-grid bookkeeping: no room for a second operand set
#pragma unroll NWIN == 6 ? 1 : W3_BKP / 4      // (the six-pass window spills when unrolled)
                for (int ks = 0; ks < W3_BKP / 4; ++ks) {
                    Ops o;
                    if (OSI_ABLATE & 16) {
                        o.a0 = ra[0][0]; o.a1 = ra[1][1]; o.m = 0x1ff;
#pragma unroll
                        for (int tp = 0; tp < 9; ++tp) o.b[tp] = rb[tp % NWIN][tp & 3];
                    }
                    lds_ops(ks, o);
                    mma_ops(o);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!(OSI_ABLATE & 2)) if (t + 1 < T) xform(t + 1);
            if (!(OSI_ABLATE & 4)) __syncthreads();                 // every wave is done reading the run
            if (!(OSI_ABLATE & 2)) if (t + 1 < T) sstore(t + 1);
            if (!(OSI_ABLATE & 4)) __syncthreads();
        }
        OSI_STAMP(p, blockIdx.x, 2);
    }
    // C/D of 16x16x4: column (cin) = lane & 15, row (cout) = (lane >> 4) * 4 + reg
    float* out = p.y + (size_t)split * p.slab_stride;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = n0 + 32 * wn2 + 16 * i + lk * 4 + e;
                out[(size_t)row * p.Ktot + t * p.Cin + c0 + 16 * wc + l15] = acc[t][i][e];
            }
    OSI_STAMP(p, blockIdx.x, 3);
}

// ======================================================================================================
// Forward, 1x1 stride-1 convolutions with a SHORT K (Cin = 64 or 128): persistent row walker.
// With K = 64 a 64x64 tile is two K tiles = 32 MFMAs per wave between a prologue (row addresses, first operand round trip) and an
// epilogue (LDS transpose, 16 KB of output, BatchNorm partials): tools/wg_timeline.py shows a k_conv_fwd workgroup of 64->256 @56x56
// spending 27 % of its life in its K loop (prologue 4.4 + loop 3.3 + epilogue 4.2 us, 98 workgroups per CU): the layer is bound by
// those two latencies and by 514 MB of HBM traffic, not by the matrix pipe. Here a workgroup owns ONE column tile, keeps its weight tile
// (all of K) in LDS for its whole life, and walks row tiles mt = walker, walker + walkers, ...: the next tile's activation rows are
// fetched into registers BEFORE the MFMAs and the epilogue of the current one, so the load round trip hides behind them and no
// per-tile prologue is left. The four (Cout / 64) workgroups of a walker take consecutive dispatch slots of one XCD: they stream the
// same rows at about the same time and three of four reads hit that L2. Epilogue as in k_conv_fwd (LDS transpose -> 16-byte
// stores, BatchNorm (mean, M2) partials per row tile straight from the accumulators).
// ======================================================================================================
template <int KT, int XF, bool OE = false>   // KT = Cin / 32 K tiles (2 or 4); XF = 1: the A operand is relu(x * in_scale[c] + in_shift[c]); OE: output epilogue (ConvP)
__global__ __launch_bounds__(256, 4) void k_conv1x1_rows(ConvP p, int walkers) {
    static_assert(!OE || XF == 0, "output epilogue: plain input");
    constexpr int BN = 64, IMG = 64 * LDR, LDT = BN + 4;
    static_assert(64 * LDT <= KT * IMG, "the transposed output tile reuses the activation images");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sB = smem;                    // [KT][64][LDR] the column tile's weights, resident
    float* sA = smem + KT * IMG;         // [KT][64][LDR] the current row tile; reused as the transposed output tile
    float* s_sc = sA + KT * IMG;         // XF: [Cin] scale | [Cin] shift
    float* s_sh = s_sc + p.Cin;
    float* s_st = s_sh + p.Cin;          // [2][64][3] statistics of the two wave rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int kq = tid & 7, lr = tid >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int nt = slot % p.NT, walker = (slot / p.NT) * 8 + xcd;
    if (walker >= walkers) return;
    const int n0 = nt * BN;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes), rw = make_rsrc(p.w, p.w_bytes);
    {   // the weight tile: rows n0 + lr (+ 32), all of K
        f32x4 rb[KT][2];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int i = 0; i < 2; ++i) rb[kt][i] = bld4(rw, ((uint32_t)(n0 + lr + 32 * i) * p.Ktot + kq * 4) * 4, (uint32_t)(kt * BK * 4));
        if (XF) {
            for (int c = tid * 4; c < p.Cin; c += 1024) {
                *reinterpret_cast<f32x4*>(s_sc + c) = ld4(p.in_scale + c);
                *reinterpret_cast<f32x4*>(s_sh + c) = ld4(p.in_shift + c);
            }
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(sB + kt * IMG + (lr + 32 * i) * LDR + kq * 4) = rb[kt][i];
    }
    f32x4 ra[KT][2];
    bool rok[2] = {false, false};
    auto gloadA = [&](int mt) {          // row tile mt -> registers; rows past M read as zeros (range check)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mt * 64 + lr + 32 * i;
            rok[i] = m < p.M;
            const uint32_t off = rok[i] ? ((uint32_t)m * p.Cin + kq * 4) * 4 : OOB;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) ra[kt][i] = bld4(rx, off, (uint32_t)(kt * BK * 4));
        }
    };
    auto sstoreA = [&]() {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            f32x4 sc = {0, 0, 0, 0}, sh = sc;
            if (XF) { sc = *reinterpret_cast<const f32x4*>(s_sc + kt * BK + kq * 4); sh = *reinterpret_cast<const f32x4*>(s_sh + kt * BK + kq * 4); }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 v = ra[kt][i];
                if (XF) {     // rows past M (last tile only) carry relu(shift): never stored, masked out of the statistics by row index
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(__builtin_fmaf(v[e], sc[e], sh[e]), 0.f);
                }
                *reinterpret_cast<f32x4*>(sA + kt * IMG + (lr + 32 * i) * LDR + kq * 4) = v;
            }
        }
    };
    int mt = walker;
    gloadA(mt);
    if (XF) __syncthreads();             // the scale / shift tables are complete
    sstoreA();
    while (mt < p.MT) {
        __syncthreads();                 // the row tile (and, the first time, the weight tile) is visible
        const int mt_next = mt + walkers;
        if (mt_next < p.MT) gloadA(mt_next);      // in flight behind the MFMAs and the epilogue below
        f32x16 acc[1][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) mma_RR<1, 1>(sA + kt * IMG, sB + kt * IMG, wm * 32, wn * 32, lane, acc);
        __syncthreads();                 // every wave is done reading the row tile: its LDS becomes the transposed output tile
        const int m0 = mt * 64;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) sA[(wm * 32 + acc_row(rr, lane)) * LDT + wn * 32 + (lane & 31)] = acc[0][0][rr];
        __syncthreads();
        {   // 16-byte stores of 4 consecutive channels per lane (measured against 4-byte stores straight from the accumulators, which
            // save two barriers per tile: 86 vs 82 TFLOP/s on 64->256 @56x56)
            const int c4 = tid & 15, rg = tid >> 4;
            if constexpr (OE) {          // [relu](fma(acc, osc, osh) [+ ores]) as in k_conv_fwd
                const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.ores ? p.ores : p.y, p.ores ? p.M * p.Cout * 4 : 0);
                const f32x4 sc = ld4(p.osc + n0 + c4 * 4), sh = ld4(p.osh + n0 + c4 * 4);
                f32x4 rv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int m = m0 + rg + 16 * k;
                    rv[k] = bld4(ro, m < p.M ? (uint32_t)((m * p.Cout + n0 + c4 * 4) * 4) : OOB, 0);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int rl = rg + 16 * k, m = m0 + rl;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(sA + rl * LDT + c4 * 4);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float t = __builtin_fmaf(a[e], sc[e], sh[e]) + rv[k][e]; v[e] = p.orelu ? fmaxf(t, 0.f) : t; }
                    if (m < p.M) *reinterpret_cast<f32x4*>(p.y + (size_t)m * p.Cout + n0 + c4 * 4) = v;
                }
            } else
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rl = rg + 16 * k, m = m0 + rl;
                if (m < p.M)
                    *reinterpret_cast<f32x4*>(p.y + (size_t)m * p.Cout + n0 + c4 * 4) = *reinterpret_cast<const f32x4*>(sA + rl * LDT + c4 * 4);
            }
        }
        if (p.pmean) {                   // BatchNorm (mean, M2) partials of this row tile's 64 columns (see k_conv_fwd: full tiles unmasked)
            auto stats = [&](auto FULLC) {
                constexpr bool full = decltype(FULLC)::value;
                const int row0 = m0 + wm * 32;
                const float cnt = full ? 32.f : (float)min(32, max(0, p.M - row0));
                float sm = 0.f;
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) sm += (full || row0 + acc_row(rr, lane) < p.M) ? acc[0][0][rr] : 0.f;
                sm += __shfl_xor(sm, 32, 64);
                const float mu = cnt > 0.f ? sm / cnt : 0.f;
                float q = 0.f;
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const float dlt = acc[0][0][rr] - mu;
                    if constexpr (full) q = __builtin_fmaf(dlt, dlt, q);
                    else q += (row0 + acc_row(rr, lane) < p.M) ? dlt * dlt : 0.f;
                }
                q += __shfl_xor(q, 32, 64);
                if (lane < 32) {
                    float* dst = s_st + (wm * BN + wn * 32 + lane) * 3;
                    dst[0] = cnt; dst[1] = mu; dst[2] = q;
                }
            };
            if (m0 + 64 <= p.M) stats(std::true_type{}); else stats(std::false_type{});
        }
        __syncthreads();                 // transposed tile read, statistics of both wave rows written
        if (p.pmean && tid < BN) {
            const float* a = s_st + tid * 3;
            const float* b = s_st + (BN + tid) * 3;
            float cn = a[0], cm = a[1], cs = a[2];
            chan_merge(cn, cm, cs, b[0], b[1], b[2]);
            p.pmean[(size_t)mt * p.Cout + n0 + tid] = cm;
            p.pm2[(size_t)mt * p.Cout + n0 + tid] = cs;
        }
        if (mt_next < p.MT) sstoreA();   // (s_st is rewritten only behind the next tile's barriers, which its readers reach after reading)
        mt = mt_next;
    }
}

// Fix-up pass of a forward launch with a K-split tail (tile_of_block_split): one workgroup per remainder tile adds the tile's
// splits in split order (fixed: bitwise reproducible), writes the output rows and — like the convolution's own epilogue — the
// BatchNorm partial (mean, M2) of the tile's 64 columns over its valid rows (two passes over the register-resident tile).
__global__ __launch_bounds__(256) void k_conv_fwd_tail_fixup(ConvP p) {
    __shared__ float red[16][64];
    __shared__ float smean[64];
    const int tile = blockIdx.x;
    const int mt = p.MT1 + tile / p.NT, nt = tile - (tile / p.NT) * p.NT;
    const int m0 = mt * 64, n0 = nt * 64;
    const int tid = threadIdx.x, c4 = tid & 15, rg = tid >> 4;
    // the tile's S slabs behind one buffer descriptor: a split index past S reads as zero through the range check, so the loads of
    // four splits x four rows go out together (the split ORDER of every sum stays 0, 1, 2, ...: bitwise reproducible)
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.ks_slab + (size_t)tile * p.ks_S * 4096, p.ks_S * 16384);
    const uint32_t off = (uint32_t)(rg * 64 + c4 * 4) * 4u;
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = bld4(rs, off + (uint32_t)k * 4096u, 0);
    for (int s0 = 1; s0 < p.ks_S; s0 += 4) {
        f32x4 t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) t[j][k] = bld4(rs, off + (uint32_t)k * 4096u + (uint32_t)(s0 + j) * 16384u, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += t[j][k];
    }
    if (p.osc) {      // inference launch: the output epilogue [relu](fma(acc, osc, osh) [+ ores]) on the summed tile (as k_conv_fwd<OE>); no statistics
        const __amdgpu_buffer_rsrc_t ro = make_rsrc(p.ores ? p.ores : p.y, p.ores ? p.M * p.Cout * 4 : 0);
        const f32x4 sc = ld4(p.osc + n0 + c4 * 4), sh = ld4(p.osh + n0 + c4 * 4);
        f32x4 rv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = m0 + rg + 16 * k;
            rv[k] = bld4(ro, m < p.M ? (uint32_t)((m * p.Cout + n0 + c4 * 4) * 4) : OOB, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float t = __builtin_fmaf(v[k][e], sc[e], sh[e]) + rv[k][e]; v[k][e] = p.orelu ? fmaxf(t, 0.f) : t; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int rl = rg + 16 * k;
        if (m0 + rl < p.M) *reinterpret_cast<f32x4*>(p.y + (size_t)(m0 + rl) * p.Cout + n0 + c4 * 4) = v[k];
    }
    if (!p.pmean) return;
    const float cnt = (float)min(64, p.M - m0);
    f32x4 sm = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) if (m0 + rg + 16 * k < p.M) sm += v[k];
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rg][c4 * 4 + e] = sm[e];
    __syncthreads();
    if (tid < 64) {
        float a = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) a += red[g][tid];
        smean[tid] = a / cnt;
    }
    __syncthreads();
    const f32x4 mu = *reinterpret_cast<const f32x4*>(smean + c4 * 4);
    f32x4 q = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) if (m0 + rg + 16 * k < p.M) { const f32x4 d = v[k] - mu; q += d * d; }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rg][c4 * 4 + e] = q[e];     // every thread read smean before this barrier pair's first barrier
    __syncthreads();
    if (tid < 64) {
        float a = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) a += red[g][tid];
        p.pmean[(size_t)mt * p.Cout + n0 + tid] = smean[tid];
        p.pm2[(size_t)mt * p.Cout + n0 + tid] = a;
    }
}

// Fix-up pass of an input-gradient launch with a K-split tail (stride 1): per remainder tile, the splits are added in split order
// and the tile goes through the very epilogue of the convolution kernel (dgrad_epilogue64).
template <int FUSED>
__global__ __launch_bounds__(256) void k_conv_dgrad_tail_fixup(ConvP p) {
    __shared__ float red[4 * 64 * 3];
    const int tile = blockIdx.x;
    const int mt = p.MT1 + tile / p.NT, nt = tile - (tile / p.NT) * p.NT;
    const int nsp = p.ks_S;
    // one descriptor over the tile's slabs: splits past nsp read as zero, so eight loads go out together; sums stay in split order
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.ks_slab + (size_t)tile * nsp * 4096, nsp * 16384);
    dgrad_epilogue64<FUSED, false>(p, red, 0, mt, mt * 64, nt * 64, p.B * p.H * p.W, 1, 0, 0, p.cHW[0], p.cW[0], [&](int rl, int c4) {
        const uint32_t off = (uint32_t)(rl * 64 + c4 * 4) * 4u;
        f32x4 a = bld4(rs, off, 0);
        for (int s0 = 1; s0 < nsp; s0 += 8) {
            f32x4 t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = bld4(rs, off + (uint32_t)(s0 + j) * 16384u, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) a += t[j];
        }
        return a;
    });
}

// out[i] = sum_s slab[s][i]  (fixed order: bitwise reproducible)
// One workgroup = 16 consecutive float4 outputs x 16 split lanes: lane j sums splits j, j+16, ... (4 independent loads in
// flight), then the 16 lane partials are added in lane order. Many small dependent-latency chains instead of one long one.
__global__ __launch_bounds__(256) void k_slab_reduce(const float* slab, float* out, size_t n4, size_t stride4, int splits) {
    __shared__ f32x4 red[16][16];
    const int o = threadIdx.x & 15, l = threadIdx.x >> 4;
    const size_t i = (size_t)blockIdx.x * 16 + o;
    const f32x4* s = reinterpret_cast<const f32x4*>(slab);
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    if (i < n4) {
        int k = l;
        for (; k + 48 < splits; k += 64) {
            a0 += s[i + (size_t)k * stride4]; a1 += s[i + (size_t)(k + 16) * stride4];
            a2 += s[i + (size_t)(k + 32) * stride4]; a3 += s[i + (size_t)(k + 48) * stride4];
        }
        for (; k < splits; k += 16) a0 += s[i + (size_t)k * stride4];
    }
    red[l][o] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (l == 0 && i < n4) {
        f32x4 a = red[0][o];
#pragma unroll
        for (int k = 1; k < 16; ++k) a += red[k][o];
        reinterpret_cast<f32x4*>(out)[i] = a;
    }
}

// stem weight [64][7][7][3] (KRSC of the OIHW parameter) <-> packed [64][56][4] (zero padded taps/channel)
__global__ void k_stem_pack(const float* w, float* wp, int Cout) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cout * 224) return;
    int n = i / 224, k = i - n * 224, tap = k >> 2, c = k & 3;
    wp[i] = (tap < 49 && c < 3) ? w[(n * 49 + tap) * 3 + c] : 0.f;
}
__global__ void k_stem_unpack(const float* gp, float* g, int Cout) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cout * 147) return;
    int n = i / 147, k = i - n * 147, tap = k / 3, c = k - tap * 3;
    g[i] = gp[n * 224 + tap * 4 + c];
}

template <typename K>
static int set_smem(K kern, size_t bytes) {
    if (bytes > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
            return OSI_ERR_LAUNCH;
    }
    return OSI_OK;
}

static bool desc_ok(const osi_conv_desc* d) {
    if (!d) return false;
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0) return false;
    if (d->R <= 0 || d->S <= 0 || d->stride <= 0 || d->pad < 0) return false;
    if (d->Ho != (d->H + 2 * d->pad - d->R) / d->stride + 1) return false;
    if (d->Wo != (d->W + 2 * d->pad - d->S) / d->stride + 1) return false;
    if (d->Ho <= 0 || d->Wo <= 0) return false;
    // tensors below 2 GiB: 31-bit byte offsets for the buffer descriptors (and the OOB sentinel 0x80000000 stays out of range)
    if ((long)d->B * d->H * d->W * d->Cin >= (1l << 29)) return false;
    if ((long)d->B * d->Ho * d->Wo * d->Cout >= (1l << 29)) return false;
    if ((long)d->Cout * d->R * d->S * (d->Cin < 4 ? 4 : d->Cin) >= (1l << 29)) return false;
    return true;
}
static bool is_stem(const osi_conv_desc* d) { return d->Cin == 4 && d->R == 7 && d->S == 7; }

static ConvP make_p(const osi_conv_desc* d) {
    ConvP p{};
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
    p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad;
    p.M = d->B * d->Ho * d->Wo;
    p.Ktot = is_stem(d) ? 224 : d->R * d->S * d->Cin;
    p.dHoWo = make_fastdiv((uint32_t)(d->Ho * d->Wo));
    p.dWo = make_fastdiv((uint32_t)d->Wo);
#ifdef OSI_STAMPS
    p.stamps = g_osi_stamps;
#endif
    return p;
}

// ---- balanced remainder: plan ------------------------------------------------------------------------------------------------
struct TailPlan { int MT1, S, ksT, tiles; };   // tiles = remainder tiles ((MT - MT1) * NT); S <= 1: no split
// MT x NT tiles of 64x64, T K-tiles each. Splits the ragged last round when the model says it buys at least 8 % of the launch and
// every split keeps at least 2 K tiles. Measured (tools/bench_tail.py, B = 128): 3 rounds + 16 tiles (7x7 layers) +15 %, 6 rounds +
// 32 tiles (14x14, 256 channels) +7.5 %, 12 rounds + 64 tiles (a 5.8 % model gain) -1.5 ... -3.6 %: the fix-up pass and its launch
// boundary cost what the balance returns, so those launches stay single-pass. Few, long splits beat many short ones on the 1x1 layers
// (at most 8 splits of at least 16 K tiles: 2048->512 @7x7 108 -> 119 TFLOP/s, 1024->256 @14x14 119 -> 126 against 32 / 2).
static TailPlan plan_tail_split(long MT, int NT, int T) {
    TailPlan t{(int)MT, 1, T, 0};
    if (!g_osi_tuning.tail_split || T < 4 || g_osi_tuning.tail_mint < 1) return t;
    const long ncu = chip_cus(), V = MT * NT;
    const long q = V / ncu, r = V - q * ncu;
    if (r == 0 || q > g_osi_tuning.tail_qmax) return t;
    if ((double)(ncu - r) / (double)ncu / (double)(q + 1) * 100.0 < (double)g_osi_tuning.tail_gain) return t;
    long MT1 = q * ncu / NT;                       // full rounds, in whole row tiles
    long rem = (MT - MT1) * NT;
    long S = ncu / rem;
    if (S > T / g_osi_tuning.tail_mint) S = T / g_osi_tuning.tail_mint;
    if (S > g_osi_tuning.tail_smax) S = g_osi_tuning.tail_smax;
    if (q <= 2) {
        // Few rounds (small batches: B = 64 halves every M of the B = 128 plans; the 7x7 layers run 1.5 rounds, the 14x14 ones 3): the
        // remainder may be larger than half the chip, where one round of floor(N / rem) pieces per tile is no split at all. Pieces may
        // then take several rounds of their own: S pieces per tile cost ceil(rem S / N) / S of a tile time instead of 1 — pick the S
        // with the least modelled time (392 tiles on 256 CUs: S = 3 -> 1 + 2/3 instead of 2 rounds).
        const long smax = std::min<long>(g_osi_tuning.tail_smax, T / g_osi_tuning.tail_mint);
        double best = S >= 2 ? 1.0 / (double)S : 1.0;
        for (long c = 2; c <= smax; ++c) {
            const double cost = (double)((rem * c + ncu - 1) / ncu) / (double)c;
            if (cost < best - 1e-9) { best = cost; S = c; }
        }
        if ((1.0 - best) / (double)(q + 1) * 100.0 < (double)g_osi_tuning.tail_gain) return t;
    }
    if (S <= 1) return t;
    const int ksT = (int)((T + S - 1) / S);
    S = (T + ksT - 1) / ksT;
    if (S <= 1) return t;
    t.MT1 = (int)MT1; t.S = (int)S; t.ksT = ksT; t.tiles = (int)rem;
    return t;
}
static size_t tail_slab_floats(const TailPlan& t) { return t.S > 1 ? (size_t)t.tiles * t.S * 4096 : 0; }

template <int XF, bool W3 = false, bool OE = false>
static int launch_fwd_split(ConvP p, const TailPlan& tp, float* slab, hipStream_t st) {
    p.MT = osi_cdiv(p.M, 64); p.NT = p.Cout / 64;
    p.MT1 = tp.MT1; p.ks_S = tp.S; p.ks_T = tp.ksT; p.ks_slab = slab;
    p.g1 = osi_cdiv(p.MT1, 8) * 8 * p.NT;
    const int keys = (p.MT - p.MT1) * tp.S;
    const int grid = p.g1 + osi_cdiv(keys, 8) * 8 * p.NT;
    size_t smem = (size_t)((W3 ? W3_WROWS : 64) + 64) * LDR * sizeof(float);
    if (XF) smem += (size_t)2 * p.Cin * sizeof(float);
    if (W3) p.cW[0] = make_fastdiv((uint32_t)p.W + 1);     // row windows: division by the padded row length
    if (int e = set_smem(k_conv_fwd<1, 1, false, 1, XF, true, W3, OE>, smem)) return e;
    hipLaunchKernelGGL((k_conv_fwd<1, 1, false, 1, XF, true, W3, OE>), dim3(grid), dim3(256), smem, st, p);
    OSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_conv_fwd_tail_fixup, dim3(tp.tiles), dim3(256), 0, st, p);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

template <int WM, int WN, bool STEM, int NST = 2, int XF = 0, bool W3 = false, bool OE = false>
static int launch_fwd(ConvP p, hipStream_t st) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    p.MT = osi_cdiv(p.M, BM); p.NT = p.Cout / BN;
    size_t smem = NST * (size_t)((W3 ? W3_WROWS : BM) + BN) * LDR * sizeof(float);
    if (XF) smem += (size_t)2 * p.Cin * sizeof(float);
    if (W3) p.cW[0] = make_fastdiv((uint32_t)p.W + 1);     // row windows: division by the padded row length
    if (int e = set_smem(k_conv_fwd<WM, WN, STEM, NST, XF, false, W3, OE>, smem)) return e;
    int grid = osi_cdiv(p.MT, 8) * 8 * p.NT;
    hipLaunchKernelGGL((k_conv_fwd<WM, WN, STEM, NST, XF, false, W3, OE>), dim3(grid), dim3(256), smem, st, p);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}
// 1x1 stride-1 forward with Cin = 64 / 128 on the persistent row walker (k_conv1x1_rows): four resident workgroups per CU
template <int KT, int XF, bool OE = false>
static int launch_fwd_rows(ConvP p, hipStream_t st) {
    p.MT = osi_cdiv(p.M, 64); p.NT = p.Cout / 64;
    int walkers = (4 * hw_cus() / p.NT + 7) / 8 * 8;
    if (walkers < 8) walkers = 8;
    if (walkers > p.MT) walkers = (p.MT + 7) / 8 * 8;
    const size_t smem = ((size_t)2 * KT * 64 * LDR + 2 * p.Cin + 2 * 64 * 3) * sizeof(float);
    if (int e = set_smem(k_conv1x1_rows<KT, XF, OE>, smem)) return e;
    hipLaunchKernelGGL((k_conv1x1_rows<KT, XF, OE>), dim3(walkers * p.NT), dim3(256), smem, st, p, walkers);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}
// "fwd_rows": 0 off; 1 (default) Cin = 64 and at least eight row tiles per CU (the 56 x 56 layers of layer1 at batch >= 42: 64->256
// 78 -> 86 TFLOP/s, 64->64 76.5 -> 79.9; Cin = 128 @28x28 measured 102.5 -> 100.5: stays on k_conv_fwd); 2 every shape the kernel takes (tests)
static bool rows_rule(const osi_conv_desc* d, bool unit, const float* res) {
    const int m = g_osi_tuning.fwd_rows;
    if (!m || !unit || res || d->Cout % 64 || (d->Cin != 64 && d->Cin != 128)) return false;
    return m == 2 || (d->Cin == 64 && (long)d->B * d->Ho * d->Wo >= 64L * 8 * hw_cus());
}

template <int WM, int WN, int NST, int FUSED, bool POOL = false, bool W3 = false>
static int launch_dgrad_impl(ConvP p, hipStream_t st) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    const int s = p.stride;
    for (int ph = 0; ph < s; ++ph)
        for (int pw = 0; pw < s; ++pw) {
            const int hc = (p.H - ph + s - 1) / s, wc = (p.W - pw + s - 1) / s;
            p.cHW[ph * s + pw] = make_fastdiv((uint32_t)(hc * wc > 0 ? hc * wc : 1));
            p.cW[ph * s + pw] = make_fastdiv((uint32_t)(wc > 0 ? wc : 1));
        }
    p.unit = (p.R == 1 && p.S == 1 && s == 1 && p.pad == 0) ? 1 : 0;
    const int Hc = osi_cdiv(p.H, s), Wc = osi_cdiv(p.W, s);  // largest class
    p.MT = osi_cdiv((long)p.B * Hc * Wc, BM); p.NT = p.Cin / BN;
    size_t smem = 2 * (size_t)((W3 ? W3_WROWS : BM) * LDR + BK * (BN + 4)) * sizeof(float);
    smem = smem / 2 * NST;
    if (W3) p.cW[1] = make_fastdiv((uint32_t)p.W + 1);     // row windows: division by the padded row length
    if (int e = set_smem(k_conv_dgrad<WM, WN, NST, FUSED, false, POOL, W3>, smem)) return e;
    int grid = osi_cdiv(p.MT, 8) * 8 * p.NT;
    hipLaunchKernelGGL((k_conv_dgrad<WM, WN, NST, FUSED, false, POOL, W3>), dim3(grid, s * s), dim3(256), smem, st, p, Hc, Wc);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}
// 64x64 single-buffered input gradient with a K-split tail (stride 1): convolution launch + fix-up pass
template <int FUSED, bool W3 = false>
static int launch_dgrad_split(ConvP p, const TailPlan& tp, float* slab, hipStream_t st) {
    p.cHW[0] = make_fastdiv((uint32_t)(p.H * p.W)); p.cW[0] = make_fastdiv((uint32_t)p.W);
    p.unit = (p.R == 1 && p.S == 1 && p.pad == 0) ? 1 : 0;
    p.MT = osi_cdiv((long)p.B * p.H * p.W, 64); p.NT = p.Cin / 64;
    p.MT1 = tp.MT1; p.ks_S = tp.S; p.ks_T = tp.ksT; p.ks_slab = slab;
    p.g1 = osi_cdiv(p.MT1, 8) * 8 * p.NT;
    const int keys = (p.MT - p.MT1) * tp.S;
    const int grid = p.g1 + osi_cdiv(keys, 8) * 8 * p.NT;
    const size_t smem = (size_t)((W3 ? W3_WROWS : 64) * LDR + BK * (64 + 4)) * sizeof(float);
    if (W3) p.cW[1] = make_fastdiv((uint32_t)p.W + 1);
    hipLaunchKernelGGL((k_conv_dgrad<1, 1, 1, FUSED, true, false, W3>), dim3(grid, 1), dim3(256), smem, st, p, p.H, p.W);
    OSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_conv_dgrad_tail_fixup<FUSED>, dim3(tp.tiles), dim3(256), 0, st, p);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

static int dgrad_flavour(const ConvP& p) {
    if (!p.addend && !p.ebits && !p.ey1) return 2;     // in-block: gate recomputed from y0 and / or sums over y0
    if (!p.escale0) return 3;                          // block input: addend, bitmask, sums over y0 (and y1)
    return 1;
}
template <int WM, int WN, int NST = 2>
static int launch_dgrad(ConvP p, hipStream_t st) {
    // the fused epilogue (mask / BatchNorm reductions) is its own instantiation so that plain launches keep the small one
    if (p.ebits || p.esum || p.escale0) {
        if (NST != 1 || WM != 1) return OSI_ERR_ARG;   // fusion is built for the single-buffered 64-row tiles the executor uses
        if (p.epool) return WN == 1 ? launch_dgrad_impl<1, 1, 1, 1, true>(p, st) : OSI_ERR_ARG;
        if (WN == 1) {   // the two fusion flavours the executor issues have their own instantiations (see dgrad_epilogue64)
            const int fl = dgrad_flavour(p);
            if (fl == 2) return launch_dgrad_impl<1, 1, 1, 2>(p, st);
            if (fl == 3) return launch_dgrad_impl<1, 1, 1, 3>(p, st);
        }
        return launch_dgrad_impl<1, WN, 1, 1>(p, st);
    }
    return launch_dgrad_impl<WM, WN, NST, false>(p, st);
}
template <int WM, int WN, bool STEM, int NST = 2, bool XF = false>
static int launch_wgrad(ConvP p, int splits, hipStream_t st) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    p.MT = p.Cout / BM;
    p.NT = STEM ? osi_cdiv(p.Ktot, BN) : p.R * p.S * (p.Cin / BN);
    size_t smem = 2 * (size_t)BK * (BM + 4 + BN + 4) * sizeof(float);
    smem = smem / 2 * NST;
    p.wtbl = (!STEM && !p.unit) ? 1 : 0;
    smem += 32 * BK * sizeof(uint32_t);   // TW * BK offsets (read, and ignored, by the 1x1 stride-1 form too)
    if (XF) smem += 2 * BN * sizeof(float);                 // scale | shift of the column tile's channels
    if (int e = set_smem(k_conv_wgrad<WM, WN, STEM, NST, XF>, smem)) return e;
    p.splits = splits;
    p.wgroup = g_osi_tuning.wgrad_group;   // 0: 2-D grid, 1: only the taps of a cell share an XCD, 2 (default): whole K splits do
    if (p.wgroup) {
        const int taps = STEM ? 1 : p.R * p.S;
        const int ctiles = p.NT / taps;
        if (p.wgroup == 1) { p.gkind = 3; p.gkeys = p.MT * ctiles * splits; p.ginner = taps; }
        else if (splits >= 8) { p.gkind = 0; p.gkeys = splits; p.ginner = p.MT * ctiles * taps; }
        else {
            // few splits (many output tiles): spread (split, tile-row) or (split, tile-column) keys over the XCDs, whichever
            // re-reads fewer bytes across XCDs: rows share the dY tile and re-read X once per cout tile, columns the other way round
            const double dy_bytes = (double)p.Cout, x_bytes = (double)p.Cin;   // bytes per pixel of the two operands, up to a common factor
            const double cost_rows = x_bytes * (p.MT < 8 ? p.MT : 8) + dy_bytes, cost_cols = dy_bytes * (ctiles < 8 ? ctiles : 8) + x_bytes;
            if (cost_rows <= cost_cols) { p.gkind = 1; p.gkeys = splits * p.MT; p.ginner = ctiles * taps; }
            else { p.gkind = 2; p.gkeys = splits * ctiles; p.ginner = p.MT * taps; }
        }
        const long grid = ((long)p.gkeys + 7) / 8 * 8 * p.ginner;
        hipLaunchKernelGGL((k_conv_wgrad<WM, WN, STEM, NST, XF>), dim3((unsigned)grid), dim3(256), smem, st, p);
    } else {
        hipLaunchKernelGGL((k_conv_wgrad<WM, WN, STEM, NST, XF>), dim3(p.MT * p.NT, splits), dim3(256), smem, st, p);
    }
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

// all-taps 3x3 weight gradient: eligibility, split plan, launch
static bool wgrad3_ok(const osi_conv_desc* d) {
    if (!g_osi_tuning.wgrad3 || d->R != 3 || d->S != 3 || d->pad != 1 || d->Cin % 32 || d->Cout % 64) return false;
    if (d->stride == 1) return d->W <= 63;
    // stride 2: four parity sub-grids of the input (even H and W), window of 130 + 2 Wo rows in at most six passes
    return d->stride == 2 && g_osi_tuning.wgrad3 >= 2 && d->H % 2 == 0 && d->W % 2 == 0 && 130 + 2 * d->Wo <= 192;
}
static void plan_wgrad3(const osi_conv_desc* d, int& splits, int& kchunk) {
    const long cells = (long)(d->Cout / 64) * (d->Cin / 32);
    const long M = (long)d->B * d->Ho * d->Wo;
    // four 256-thread workgroups per CU (register budget of the nine accumulator sets) fill the machine; as for the per-tap kernel
    // the side stream gets half of that by default
    long s = (g_osi_tuning.wgrad3_blocks + cells - 1) / cells;
    const long maxs = (M + 8 * W3_BKP - 1) / (8 * W3_BKP);
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    if (s >= 8) { long r8 = (s + 4) / 8 * 8; if (r8 > maxs) r8 = maxs / 8 * 8; if (r8 >= 8) s = r8; }
    long chunk = ((M + s - 1) / s + W3_BKP - 1) / W3_BKP * W3_BKP;
    splits = (int)((M + chunk - 1) / chunk); kchunk = (int)chunk;
}
template <int NWIN, bool XF, bool S2 = false>
static int launch_wgrad3_n(ConvP p, int splits, hipStream_t st) {
    p.MT = p.Cout / 64;
    const int ctiles = p.Cin / 32;
    p.splits = splits;
    if (splits >= 8) { p.gkind = 0; p.gkeys = splits; p.ginner = p.MT * ctiles; }
    else { p.gkind = 3; p.gkeys = p.MT * ctiles * splits; p.ginner = 1; }
    const size_t smem = ((size_t)W3_BKP * W3_LDA + (size_t)NWIN * 32 * W3_LDB + 32) * sizeof(float);
    if (int e = set_smem(k_conv_wgrad3<NWIN, XF, S2>, smem)) return e;
    const long grid = ((long)p.gkeys + 7) / 8 * 8 * p.ginner;
    hipLaunchKernelGGL((k_conv_wgrad3<NWIN, XF, S2>), dim3((unsigned)grid), dim3(256), smem, st, p);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}
template <bool XF>
static int launch_wgrad3(const ConvP& p, int splits, hipStream_t st) {
    if (p.stride == 2) return 130 + 2 * p.Wo <= 160 ? launch_wgrad3_n<5, XF, true>(p, splits, st) : launch_wgrad3_n<6, XF, true>(p, splits, st);
    const int win = W3_BKP + 2 * p.W + 2;
    if (win <= 64) return launch_wgrad3_n<2, XF>(p, splits, st);
    if (win <= 96) return launch_wgrad3_n<3, XF>(p, splits, st);
    return launch_wgrad3_n<5, XF>(p, splits, st);
}

// wgrad geometry shared by the workspace query and the launcher
struct WgradPlan { int wm, wn, splits, kchunk; };
static WgradPlan plan_wgrad(const osi_conv_desc* d) {
    WgradPlan w;
    const bool stem = is_stem(d);
    // 128-wide tiles wherever the channel counts allow (osi_set_tuning("wgrad_tile", 64) forces 64x64 for A/B runs). Measured with the footprint
    // budget below: 64x64 everywhere is ~8 % faster ALONE (wgrad class 11.1 -> 10.2 ms/step) but 0.4 ms slower inside the
    // overlapped step (36.7-37.0 vs 36.3 ms): six short-lived small workgroups per CU disturb the dgrad chain more than two big ones.
    const int big = g_osi_tuning.wgrad_tile != 64;
    w.wm = (big && d->Cout % 128 == 0) ? 2 : 1;
    w.wn = (big && !stem && d->Cin % 128 == 0) ? 2 : 1;
    const int BMg = 64 * w.wm, BNg = 64 * w.wn;
    const int Ktot = stem ? 224 : d->R * d->S * d->Cin;
    const long tiles = (long)(d->Cout / BMg) * (stem ? osi_cdiv(Ktot, BNg) : d->R * d->S * (d->Cin / BNg));
    const long M = (long)d->B * d->Ho * d->Wo;
    // Footprint budget per launch in units of 64x64 workgroups (a 128x128 workgroup counts as four): 2048 = two 128x128 or eight
    // 64x64 workgroups per CU. Alone, twice that is ~10 % faster, but the weight gradients run on the executor's side stream next
    // to the data-gradient chain and must leave half of each CU's LDS, registers and wave slots to the critical path: the whole
    // step is 0.7 ms shorter this way (sweep in DESIGN.md §3). osi_set_tuning("wgrad_blocks", n) overrides (development).
    const int target64 = g_osi_tuning.wgrad_blocks;
    const int target = target64 / (w.wm * w.wn);
    long splits = (target + tiles - 1) / tiles;
    long maxs = (M + 8 * BK - 1) / (8 * BK);                // at least 8 K tiles per split (amortises the 64 KiB slab tile)
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
    // whole splits are dealt to the 8 XCDs (k_conv_wgrad's mapping): a multiple of 8 gives every XCD the same number of them
    if (g_osi_tuning.wgrad_group == 2 && splits >= 8) {
        long r8 = (splits + 4) / 8 * 8;
        if (r8 > maxs) r8 = maxs / 8 * 8;
        if (r8 >= 8) splits = r8;
    }
    long chunk = ((M + splits - 1) / splits + BK - 1) / BK * BK;
    splits = (M + chunk - 1) / chunk;
    w.splits = (int)splits; w.kchunk = (int)chunk;
    return w;
}

}  // namespace

namespace osi_conv {
int hw_cus() {
    static int n = 0;
    if (n == 0) {
        int v = 0, dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;      // MI355X
    }
    return n;
}
int chip_cus() {
    if (g_osi_tuning.tail_cus > 0) return g_osi_tuning.tail_cus;
    const int n = hw_cus() - g_osi_tuning.dp_reserved_cus;
    return n < 8 ? 8 : n;
}
bool conv_desc_ok(const osi_conv_desc* d) { return desc_ok(d); }
bool conv_is_stem(const osi_conv_desc* d) { return d && is_stem(d); }
int launch_slab_reduce(const float* slab, float* out, size_t n4, size_t stride4, int splits, hipStream_t st) {
    hipLaunchKernelGGL(k_slab_reduce, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, st, slab, out, n4, stride4, splits);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}
}  // namespace osi_conv

extern "C" {

#ifdef OSI_STAMPS
// diagnostic build only: where the conv kernels of every later launch store their per-workgroup stamps (8 words per workgroup of the
// launch grid; the caller sizes the buffer for the largest grid it launches). NULL switches stamping off.
void osi_debug_set_stamps(unsigned long long* buf) { g_osi_stamps = buf; }
#endif

static int fwd_tile_rows(int tile) {
    return (tile == OSI_TILE_128x128 || tile == OSI_TILE_128x64 || tile == OSI_TILE_128x128_S1 || tile == OSI_TILE_128x64_S1) ? 128 : 64;
}
static int conv_fwd_impl(const osi_conv_desc* d, const float* x, const float* w, float* y, int tile, float* pstats,
                         size_t pstats_bytes, int* P, int* rows_per_block, osi_stream_t stream, const float* in_scale = nullptr,
                         const float* in_shift = nullptr, const float* res = nullptr);

int osi_conv_fwd(const osi_conv_desc* d, const float* x, const float* w, float* y, int tile, osi_stream_t stream) {
    return conv_fwd_impl(d, x, w, y, tile, nullptr, 0, nullptr, nullptr, stream);
}

int osi_conv_fwd_act(const osi_conv_desc* d, const float* x, const float* in_scale, const float* in_shift, const float* w, float* y,
                     int tile, float* pstats, size_t pstats_bytes, int* P, int* rows_per_block, osi_stream_t stream) {
    OSI_REQUIRE(in_scale && in_shift);
    OSI_REQUIRE(!pstats || (P && rows_per_block));
    return conv_fwd_impl(d, x, w, y, tile, pstats, pstats_bytes, P, rows_per_block, stream, in_scale, in_shift);
}

int osi_conv_fwd_act2(const osi_conv_desc* d, const float* x, const float* in_scale, const float* in_shift, const float* res,
                      const float* w, float* y, int tile, float* pstats, size_t pstats_bytes, int* P, int* rows_per_block,
                      osi_stream_t stream) {
    OSI_REQUIRE(in_scale && in_shift && res);
    OSI_REQUIRE(!pstats || (P && rows_per_block));
    OSI_REQUIRE(d && d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0);   // the input is a block output: conv1 of a bottleneck
    return conv_fwd_impl(d, x, w, y, tile, pstats, pstats_bytes, P, rows_per_block, stream, in_scale, in_shift, res);
}

// floats of the statistics part of the forward workspace (row-tile partials + finalize scratch), rounded to 256 B: the slab of a
// K-split tail starts behind it
static size_t fwd_stats_floats(const osi_conv_desc* d) {
    const size_t n = (size_t)2 * osi_cdiv((long)d->B * d->Ho * d->Wo, 64) * d->Cout + (size_t)2 * 32 * d->Cout;
    return (n + 63) / 64 * 64;
}
static TailPlan fwd_tail_plan(const osi_conv_desc* d) {
    const long M = (long)d->B * d->Ho * d->Wo;
    if (is_stem(d) || d->Cout % 64 || d->Cin % BK) return TailPlan{(int)osi_cdiv(M, 64), 1, 0, 0};
    return plan_tail_split(osi_cdiv(M, 64), d->Cout / 64, d->R * d->S * d->Cin / BK);
}

size_t osi_conv_fwd_bnstats_workspace(const osi_conv_desc* d) {
    if (!desc_ok(d)) return 0;
    // row-tile partials + the 32 group pairs of osi_bn_finalize_stats' first level [+ the slab of a K-split tail]
    return (fwd_stats_floats(d) + tail_slab_floats(fwd_tail_plan(d))) * sizeof(float);
}

int osi_conv_fwd_bnstats(const osi_conv_desc* d, const float* x, const float* w, float* y, int tile, float* pstats,
                         size_t pstats_bytes, int* P, int* rows_per_block, osi_stream_t stream) {
    OSI_REQUIRE(pstats && P && rows_per_block);
    return conv_fwd_impl(d, x, w, y, tile, pstats, pstats_bytes, P, rows_per_block, stream);
}

static int conv_fwd_impl(const osi_conv_desc* d, const float* x, const float* w, float* y, int tile, float* pstats,
                         size_t pstats_bytes, int* P, int* rows_per_block, osi_stream_t stream, const float* in_scale,
                         const float* in_shift, const float* res) {
    OSI_REQUIRE(desc_ok(d) && x && w && y);
    hipStream_t st = (hipStream_t)stream;
    ConvP p = make_p(d);
    p.x = x; p.w = w; p.y = y; p.accumulate = 0;
    p.in_scale = in_scale; p.in_shift = in_shift; p.res = res;
    OSI_REQUIRE(!in_scale || (!is_stem(d) && d->Cin <= 4096 && d->R * d->S <= 32));
    p.unit = (d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0) ? 1 : 0;
    p.x_bytes = (int)((size_t)d->B * d->H * d->W * d->Cin * 4);
    p.w_bytes = (int)((size_t)d->Cout * p.Ktot * 4);
    auto with_stats = [&](int bm) -> int {
        if (!pstats) return OSI_OK;
        const int mt = osi_cdiv(p.M, bm);
        if (pstats_bytes < (size_t)2 * mt * d->Cout * sizeof(float)) return OSI_ERR_ARG;
        p.pmean = pstats; p.pm2 = pstats + (size_t)mt * d->Cout;
        *P = mt; *rows_per_block = bm;
        return OSI_OK;
    };
    if (is_stem(d)) {
        OSI_REQUIRE(d->Cout % 64 == 0 && d->stride >= 1);
        if (tile == OSI_TILE_AUTO && stem_direct_geometry(d)) {
            // direct form (stem_direct.hip): full 8 x 16 output tiles only, one BatchNorm partial per tile of 128 pixels
            const int ntiles = d->B * d->Ho * d->Wo / STEM_TILE_PIXELS;
            float *pm = nullptr, *pq = nullptr;
            if (pstats) {
                if (pstats_bytes < (size_t)2 * ntiles * 64 * sizeof(float)) return OSI_ERR_ARG;
                pm = pstats; pq = pstats + (size_t)ntiles * 64;
                *P = ntiles; *rows_per_block = STEM_TILE_PIXELS;
            }
            return launch_stem_fwd_direct(d, x, w, y, pm, pq, ntiles, st);
        }
        if (int e = with_stats(128)) return e;
        return launch_fwd<2, 1, true>(p, st);
    }
    OSI_REQUIRE(d->Cin % BK == 0 && d->Cout % 64 == 0);
    if (tile == OSI_TILE_AUTO && rows_rule(d, p.unit != 0, res)) {      // short-K 1x1 layers: persistent row walker
        if (int e = with_stats(64)) return e;
        if (d->Cin == 64) return in_scale ? launch_fwd_rows<2, 1>(p, st) : launch_fwd_rows<2, 0>(p, st);
        return in_scale ? launch_fwd_rows<4, 1>(p, st) : launch_fwd_rows<4, 0>(p, st);
    }
    // 3x3 / stride 1 / pad 1 on the row-window form (one activation window per tap ROW and channel slice, see k_conv_fwd W3)
    const bool w3 = tile == OSI_TILE_AUTO && g_osi_tuning.fwd_w3 && !res && d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1 && d->W >= 7 &&
                    d->H == d->Ho && d->W == d->Wo;
    if (tile == OSI_TILE_AUTO && pstats) {
        // ragged last round split along K (plan_tail_split) when the caller's workspace has room for the slab behind the statistics
        const TailPlan tp = fwd_tail_plan(d);
        if (tp.S > 1 && pstats_bytes >= (fwd_stats_floats(d) + tail_slab_floats(tp)) * sizeof(float)) {
            if (int e = with_stats(64)) return e;
            float* slab = pstats + fwd_stats_floats(d);
            if (w3) return in_scale ? launch_fwd_split<1, true>(p, tp, slab, st) : launch_fwd_split<0, true>(p, tp, slab, st);
            if (in_scale) return res ? launch_fwd_split<2>(p, tp, slab, st) : launch_fwd_split<1>(p, tp, slab, st);
            return launch_fwd_split<0>(p, tp, slab, st);
        }
    }
    if (w3) {
        if (int e = with_stats(64)) return e;
        return in_scale ? launch_fwd<1, 1, false, 1, 1, true>(p, st) : launch_fwd<1, 1, false, 1, 0, true>(p, st);
    }
    if (tile == OSI_TILE_AUTO) {
        // Measured on MI355X over the 22 ResNet-50 shapes at B=128 (tools/bench_conv.py, profiles/conv_layers_r01.txt): many
        // small workgroups (4 resident per CU) beat large tiles almost everywhere because the ragged last round of the launch is
        // shorter; only the 7x7-spatial layers with few column tiles prefer the wider 64x128 tile.
        // Single-buffered LDS (two barriers per K tile, but 7-8 resident workgroups per CU) beats the double-buffered form on
        // every shape: occupancy, not staging depth, is what hides the barrier and load latency of a 16-MFMA K step.
        // 64x64 everywhere except the 7x7-spatial layers with few tiles, where the wider column tile halves the A re-reads
        const long tiles64 = ((long)p.M + 63) / 64 * (d->Cout / 64);
        tile = (tiles64 < 1024 && d->Cout % 128 == 0) ? OSI_TILE_64x128_S1 : OSI_TILE_64x64_S1;
        if (g_osi_tuning.fwd_wide && d->Cout % 128 == 0) tile = OSI_TILE_64x128_S1;   // A/B: wide tiles wherever the channel count allows
    }
    if (int e = with_stats(fwd_tile_rows(tile))) return e;
    if (in_scale) {   // fused input activation: built for the single-buffered 64-row tiles the executor uses
        if (tile == OSI_TILE_64x64_S1) return res ? launch_fwd<1, 1, false, 1, 2>(p, st) : launch_fwd<1, 1, false, 1, 1>(p, st);
        if (tile == OSI_TILE_64x128_S1) {
            OSI_REQUIRE(d->Cout % 128 == 0);
            return res ? launch_fwd<1, 2, false, 1, 2>(p, st) : launch_fwd<1, 2, false, 1, 1>(p, st);
        }
        return OSI_ERR_ARG;
    }
    switch (tile) {
        case OSI_TILE_128x128: OSI_REQUIRE(d->Cout % 128 == 0); return launch_fwd<2, 2, false>(p, st);
        case OSI_TILE_128x64: return launch_fwd<2, 1, false>(p, st);
        case OSI_TILE_64x128: OSI_REQUIRE(d->Cout % 128 == 0); return launch_fwd<1, 2, false>(p, st);
        case OSI_TILE_64x64: return launch_fwd<1, 1, false>(p, st);
        case OSI_TILE_64x64_S1: return launch_fwd<1, 1, false, 1>(p, st);
        case OSI_TILE_64x128_S1: OSI_REQUIRE(d->Cout % 128 == 0); return launch_fwd<1, 2, false, 1>(p, st);
        case OSI_TILE_128x128_S1: OSI_REQUIRE(d->Cout % 128 == 0); return launch_fwd<2, 2, false, 1>(p, st);
        case OSI_TILE_128x64_S1: return launch_fwd<2, 1, false, 1>(p, st);
        default: return OSI_ERR_ARG;
    }
}

// ---- inference form: the convolution's own BatchNorm (eval coefficients), shortcut and ReLU in the epilogue --------------------------
size_t osi_conv_fwd_epilogue_workspace(const osi_conv_desc* d) {
    if (!desc_ok(d) || is_stem(d)) return 0;
    return tail_slab_floats(fwd_tail_plan(d)) * sizeof(float);     // slab of a K-split tail; 0: the launch is single-pass
}

int osi_conv_fwd_epilogue(const osi_conv_desc* d, const float* x, const float* w, float* out, const osi_conv_epilogue* e, void* ws,
                          size_t ws_bytes, osi_stream_t stream) {
    OSI_REQUIRE(desc_ok(d) && !is_stem(d) && x && w && out && e && e->scale && e->shift);
    OSI_REQUIRE(d->Cin % BK == 0 && d->Cout % 64 == 0);
    OSI_REQUIRE(e->residual != out);
    hipStream_t st = (hipStream_t)stream;
    ConvP p = make_p(d);
    p.x = x; p.w = w; p.y = out; p.accumulate = 0;
    p.osc = e->scale; p.osh = e->shift; p.ores = e->residual; p.orelu = e->relu ? 1 : 0;
    p.unit = (d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0) ? 1 : 0;
    p.x_bytes = (int)((size_t)d->B * d->H * d->W * d->Cin * 4);
    p.w_bytes = (int)((size_t)d->Cout * p.Ktot * 4);
    // the same launch plans as the training forward (conv_fwd_impl with OSI_TILE_AUTO): row walker, row windows, K-split tail, tile rule
    if (rows_rule(d, p.unit != 0, nullptr)) return d->Cin == 64 ? launch_fwd_rows<2, 0, true>(p, st) : launch_fwd_rows<4, 0, true>(p, st);
    const bool w3 = g_osi_tuning.fwd_w3 && d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1 && d->W >= 7 && d->H == d->Ho && d->W == d->Wo;
    const TailPlan tp = fwd_tail_plan(d);
    if (tp.S > 1 && ws && ws_bytes >= tail_slab_floats(tp) * sizeof(float))
        return w3 ? launch_fwd_split<0, true, true>(p, tp, (float*)ws, st) : launch_fwd_split<0, false, true>(p, tp, (float*)ws, st);
    if (w3) return launch_fwd<1, 1, false, 1, 0, true, true>(p, st);
    const long tiles64 = ((long)p.M + 63) / 64 * (d->Cout / 64);
    if (d->Cout % 128 == 0 && (tiles64 < 1024 || g_osi_tuning.fwd_wide)) return launch_fwd<1, 2, false, 1, 0, false, true>(p, st);
    return launch_fwd<1, 1, false, 1, 0, false, true>(p, st);
}

static bool dgrad_w3(const osi_conv_desc* d) {
    return g_osi_tuning.dgrad_w3 && d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1 && d->W >= 7 && d->H == d->Ho && d->W == d->Wo &&
           d->Cout % BK == 0 && d->Cin % 64 == 0;
}
static int dgrad_rows(int tile) {
    return (tile == OSI_TILE_128x128 || tile == OSI_TILE_128x64 || tile == OSI_TILE_128x128_S1 || tile == OSI_TILE_128x64_S1) ? 128 : 64;
}
static int conv_dgrad_impl(const osi_conv_desc* d, const float* dy, const float* w, float* dx, const float* addend,
                           const osi_dgrad_fusion* f, int tile, int* P, osi_stream_t stream, bool sparse = false);

int osi_conv_dgrad(const osi_conv_desc* d, const float* dy, const float* w, float* dx, int accumulate, int tile,
                   osi_stream_t stream) {
    OSI_REQUIRE(accumulate >= 0 && accumulate <= 2);
    return conv_dgrad_impl(d, dy, w, dx, accumulate == 1 ? dx : nullptr, nullptr, tile, nullptr, stream, accumulate == 2);
}

// floats of the partial-sum part of the fused input-gradient workspace, rounded to 256 B: the slab of a K-split tail starts behind it
static size_t dgrad_partial_floats(const osi_conv_desc* d) {
    const int s = d->stride;
    const long mt = osi_cdiv((long)d->B * osi_cdiv(d->H, s) * osi_cdiv(d->W, s), 64);
    return ((size_t)3 * s * s * mt * d->Cin + 63) / 64 * 64;
}
static TailPlan dgrad_tail_plan(const osi_conv_desc* d) {
    const long M = (long)d->B * d->H * d->W;
    if (d->stride != 1 || is_stem(d) || d->Cin % 64 || d->Cout % BK) return TailPlan{(int)osi_cdiv(M, 64), 1, 0, 0};
    return plan_tail_split(osi_cdiv(M, 64), d->Cin / 64, d->R * d->S * d->Cout / BK);
}

size_t osi_conv_dgrad_fused_workspace(const osi_conv_desc* d) {
    if (!desc_ok(d)) return 0;
    return (dgrad_partial_floats(d) + tail_slab_floats(dgrad_tail_plan(d))) * sizeof(float);
}

int osi_conv_dgrad_fused(const osi_conv_desc* d, const float* dy, const float* w, float* dx, const float* addend,
                         const osi_dgrad_fusion* f, int tile, int* P, osi_stream_t stream) {
    OSI_REQUIRE(f && P);
    return conv_dgrad_impl(d, dy, w, dx, addend, f, tile, P, stream);
}

static int conv_dgrad_impl(const osi_conv_desc* d, const float* dy, const float* w, float* dx, const float* addend,
                           const osi_dgrad_fusion* f, int tile, int* P, osi_stream_t stream, bool sparse) {
    OSI_REQUIRE(desc_ok(d) && dy && w && dx);
    OSI_REQUIRE(!is_stem(d));  // the image needs no gradient (train.py:128-139: input is a leaf without grad)
    OSI_REQUIRE(d->Cout % BK == 0 && d->Cin % 64 == 0 && d->stride <= 2);
    hipStream_t st = (hipStream_t)stream;
    ConvP p = make_p(d);
    p.x = dy; p.w = w; p.y = dx; p.addend = addend;
    p.skip_empty = sparse ? 1 : 0;
    p.x_bytes = (int)((size_t)d->B * d->Ho * d->Wo * d->Cout * 4);
    p.w_bytes = (int)((size_t)d->Cout * p.Ktot * 4);
    // Measured (profiles/conv_layers_r01.txt and the full step): with buffer loads the 64x64 single-buffered tile wins or ties
    // on every ResNet-50 shape (the dgrad class went 12.1 -> 11.0 ms per step against the 64x128 rule used before).
    const bool even_addend = f && f->addend_stride == 2;     // built for the 64x64 epilogue only: AUTO never picks the wide tile for it
    if (tile == OSI_TILE_AUTO) tile = (g_osi_tuning.dgrad_wide && d->Cin % 128 == 0 && !even_addend) ? OSI_TILE_64x128_S1 : OSI_TILE_64x64_S1;
    if (f) {
        OSI_REQUIRE(f->relu_mask || f->scale0 || !f->partials || f->pool_idx);
        if (f->pool_idx) {   // pool mode: stride-1 conv behind the stem's max-pool, reductions only (no gate on dx), one consumer
            OSI_REQUIRE(!f->relu_mask && !f->scale0 && f->partials && !f->y1 && d->stride == 1 && f->pool_H > 0 && f->pool_W > 0);
            OSI_REQUIRE((f->pool_H + 2 - 3) / 2 + 1 == d->H && (f->pool_W + 2 - 3) / 2 + 1 == d->W);
            OSI_REQUIRE(tile == OSI_TILE_AUTO || tile == OSI_TILE_64x64_S1);
            p.epool = (const uint32_t*)f->pool_idx; p.epH = f->pool_H; p.epW = f->pool_W;
        }
        OSI_REQUIRE(!f->scale0 || (!f->relu_mask && f->shift0 && f->y0));   // one gate source: the bitmask, or y0 * scale0 + shift0 > 0
        OSI_REQUIRE(f->addend_stride >= 0 && f->addend_stride <= 2);
        if (f->addend_stride == 2) {   // sparse addend: stride-1 convolution, 64x64 tiles, an addend, no pool mode
            OSI_REQUIRE(addend && d->stride == 1 && !f->pool_idx && tile == OSI_TILE_64x64_S1);
            p.eadd_even = 1;
        }
        p.ebits = (const unsigned long long*)f->relu_mask;
        p.escale0 = f->scale0; p.eshift0 = f->shift0;
        if (f->scale0) p.ey0 = f->y0;
        if (f->partials) {
            OSI_REQUIRE(f->y0 && f->mean0 && f->invstd0 && (!f->y1 || (f->mean1 && f->invstd1)));
            p.emean0 = f->mean0; p.einv0 = f->invstd0; p.emean1 = f->mean1; p.einv1 = f->invstd1;
            const int s = d->stride;
            const int mt = osi_cdiv((long)d->B * osi_cdiv(d->H, s) * osi_cdiv(d->W, s), dgrad_rows(tile));
            p.eP = s * s * mt;
            OSI_REQUIRE(f->partials_bytes >= (size_t)3 * p.eP * d->Cin * sizeof(float));
            p.ey0 = f->y0; p.ey1 = f->y1; p.esum = f->partials;
            *P = p.eP;
            // ragged last round split along K when the caller's workspace has room for the slab behind the partial sums
            const TailPlan tp = dgrad_tail_plan(d);
            if (!f->pool_idx && tile == OSI_TILE_64x64_S1 && tp.S > 1 && f->partials_bytes >= (dgrad_partial_floats(d) + tail_slab_floats(tp)) * sizeof(float))
            {
                float* slab = f->partials + dgrad_partial_floats(d);
                const int fl = dgrad_flavour(p);
                if (fl == 2 && dgrad_w3(d)) return launch_dgrad_split<2, true>(p, tp, slab, st);
                return fl == 2 ? launch_dgrad_split<2>(p, tp, slab, st) : fl == 3 ? launch_dgrad_split<3>(p, tp, slab, st) : launch_dgrad_split<1>(p, tp, slab, st);
            }
        }
    }
    // the executor's in-block 3x3 stride-1 input gradients on the row-window form
    if (tile == OSI_TILE_64x64_S1 && dgrad_w3(d) && (p.ebits || p.esum || p.escale0) && !p.epool && dgrad_flavour(p) == 2)
        return launch_dgrad_impl<1, 1, 1, 2, false, true>(p, st);
    switch (tile) {
        case OSI_TILE_128x128: OSI_REQUIRE(d->Cin % 128 == 0); return launch_dgrad<2, 2>(p, st);
        case OSI_TILE_128x64: return launch_dgrad<2, 1>(p, st);
        case OSI_TILE_64x128: OSI_REQUIRE(d->Cin % 128 == 0); return launch_dgrad<1, 2>(p, st);
        case OSI_TILE_64x64: return launch_dgrad<1, 1>(p, st);
        case OSI_TILE_64x64_S1: return launch_dgrad<1, 1, 1>(p, st);
        case OSI_TILE_64x128_S1: OSI_REQUIRE(d->Cin % 128 == 0); return launch_dgrad<1, 2, 1>(p, st);
        case OSI_TILE_128x128_S1: OSI_REQUIRE(d->Cin % 128 == 0); return launch_dgrad<2, 2, 1>(p, st);
        case OSI_TILE_128x64_S1: return launch_dgrad<2, 1, 1>(p, st);
        default: return OSI_ERR_ARG;
    }
}

// shapes the weight-gradient kernels take: Cout in 64s; Cin in 64s (per-tap kernel), in 32s for the all-taps 3x3 form, 4 for the stem
static bool wgrad_shape_ok(const osi_conv_desc* d) {
    return d->Cout % 64 == 0 && (is_stem(d) || d->Cin % 64 == 0 || wgrad3_ok(d));
}

size_t osi_conv_wgrad_workspace(const osi_conv_desc* d) {
    if (!desc_ok(d) || !wgrad_shape_ok(d)) return 0;
    WgradPlan w{1, 1, 1, 0};
    if (wgrad3_ok(d)) plan_wgrad3(d, w.splits, w.kchunk);
    else w = plan_wgrad(d);
    const size_t n = (size_t)d->Cout * (is_stem(d) ? 224 : d->R * d->S * d->Cin);
    return w.splits > 1 ? (size_t)w.splits * n * sizeof(float) : 0;
}

static int conv_wgrad_impl(const osi_conv_desc* d, const float* dy, const float* x, float* dw, void* ws, size_t ws_bytes,
                           osi_stream_t stream, const float* in_scale, const float* in_shift);

int osi_conv_wgrad(const osi_conv_desc* d, const float* dy, const float* x, float* dw, void* ws, size_t ws_bytes,
                   osi_stream_t stream) {
    return conv_wgrad_impl(d, dy, x, dw, ws, ws_bytes, stream, nullptr, nullptr);
}

int osi_conv_wgrad_act(const osi_conv_desc* d, const float* dy, const float* x, const float* in_scale, const float* in_shift, float* dw,
                       void* ws, size_t ws_bytes, osi_stream_t stream) {
    OSI_REQUIRE(in_scale && in_shift);
    return conv_wgrad_impl(d, dy, x, dw, ws, ws_bytes, stream, in_scale, in_shift);
}

static int conv_wgrad_impl(const osi_conv_desc* d, const float* dy, const float* x, float* dw, void* ws, size_t ws_bytes,
                           osi_stream_t stream, const float* in_scale, const float* in_shift) {
    OSI_REQUIRE(desc_ok(d) && dy && x && dw);
    OSI_REQUIRE(!in_scale || !is_stem(d));
    const bool stem = is_stem(d);
    OSI_REQUIRE(wgrad_shape_ok(d));
    hipStream_t st = (hipStream_t)stream;
    const bool all_taps = !stem && wgrad3_ok(d);
    WgradPlan w{1, 1, 1, 0};
    if (all_taps) plan_wgrad3(d, w.splits, w.kchunk);
    else w = plan_wgrad(d);
    ConvP p = make_p(d);
    const size_t n = (size_t)d->Cout * p.Ktot;
    OSI_REQUIRE(n % 4 == 0);
    if (w.splits > 1) OSI_REQUIRE(ws && ws_bytes >= (size_t)w.splits * n * sizeof(float));
    p.x = x; p.w = dy; p.y = w.splits > 1 ? (float*)ws : dw;
    p.unit = (!stem && d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0) ? 1 : 0;
    p.x_bytes = (int)((size_t)d->B * d->H * d->W * d->Cin * 4);
    p.w_bytes = (int)((size_t)d->B * d->Ho * d->Wo * d->Cout * 4);
    p.kchunk = w.kchunk; p.slab_stride = n;
    p.in_scale = in_scale; p.in_shift = in_shift;
    int e;
    // Single-buffered LDS by default (as in fwd/dgrad: twice the resident workgroups beat staging depth): 11.4 -> 10.3 ms per step
    // for the wgrad class and -0.45 ms on the overlapped step, measured with the side-stream schedule of the executor.
    const int nst = g_osi_tuning.wgrad_nst;
    if (all_taps) e = in_scale ? launch_wgrad3<true>(p, w.splits, st) : launch_wgrad3<false>(p, w.splits, st);
    else if (in_scale) {   // fused input activation: single-buffered forms only (the ones the executor uses)
        if (w.wm == 2 && w.wn == 2) e = launch_wgrad<2, 2, false, 1, true>(p, w.splits, st);
        else if (w.wm == 2) e = launch_wgrad<2, 1, false, 1, true>(p, w.splits, st);
        else if (w.wn == 2) e = launch_wgrad<1, 2, false, 1, true>(p, w.splits, st);
        else e = launch_wgrad<1, 1, false, 1, true>(p, w.splits, st);
    }
    else if (stem) e = launch_wgrad<1, 1, true>(p, w.splits, st);
    else if (w.wm == 2 && w.wn == 2) e = nst == 1 ? launch_wgrad<2, 2, false, 1>(p, w.splits, st) : launch_wgrad<2, 2, false>(p, w.splits, st);
    else if (w.wm == 2) e = nst == 1 ? launch_wgrad<2, 1, false, 1>(p, w.splits, st) : launch_wgrad<2, 1, false>(p, w.splits, st);
    else if (w.wn == 2) e = nst == 1 ? launch_wgrad<1, 2, false, 1>(p, w.splits, st) : launch_wgrad<1, 2, false>(p, w.splits, st);
    else e = nst == 1 ? launch_wgrad<1, 1, false, 1>(p, w.splits, st) : launch_wgrad<1, 1, false>(p, w.splits, st);
    if (e) return e;
    if (w.splits > 1) {
        size_t n4 = n / 4;
        hipLaunchKernelGGL(k_slab_reduce, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, st, (const float*)ws, dw, n4, n4, w.splits);
        OSI_LAUNCH_CHECK();
    }
    return OSI_OK;
}

int osi_stem_weight_pack(const float* w_krsc3, float* w_packed, int Cout, osi_stream_t stream) {
    OSI_REQUIRE(w_krsc3 && w_packed && Cout > 0);
    hipLaunchKernelGGL(k_stem_pack, dim3(osi_cdiv(Cout * 224, 256)), dim3(256), 0, (hipStream_t)stream, w_krsc3, w_packed, Cout);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}
int osi_stem_grad_unpack(const float* g_packed, float* g_krsc3, int Cout, osi_stream_t stream) {
    OSI_REQUIRE(g_packed && g_krsc3 && Cout > 0);
    hipLaunchKernelGGL(k_stem_unpack, dim3(osi_cdiv(Cout * 147, 256)), dim3(256), 0, (hipStream_t)stream, g_packed, g_krsc3, Cout);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

}  // extern "C"
