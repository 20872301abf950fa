// Winograd F(2x2, 3x3) on the exact-fp32 matrix cores of gfx950: the thirteen 3x3 / stride 1 / pad 1 convolutions of ResNet-50
// (conv2 of every bottleneck without a stride: torchvision's resnet50 under /root/reference/openset_imagenet/model.py:17,37) and their
// input gradients (autograd of train.py:138) with 2.25x fewer multiplies than the implicit GEMM in conv_igemm.hip.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, 4x4 input patch d, 3x3 filter g            (Lavin & Gray 2015)
//   -> 16 independent GEMMs   M_p[n][tile] = sum_k U_p[n][k] * V_p[k][tile],   p = (xi, nu) in 4 x 4,  k = input channel, n = output channel
//
// Arithmetic: every product and sum is fp32 (v_mfma_f32_32x32x2_f32 is an exact fp32 FMA chain; the transforms are fp32 adds and the two
// 0.5 factors of G). Against an fp64 convolution the error is BELOW the direct kernel's on the network's shapes (16 / 64 / ... products
// per accumulator chain instead of 9 times as many; tools/probes/winograd_f2x2.hip prints both) — the per-kernel parity bound of
// tests/test_production_shapes_gpu.py is unchanged.
//
// Kernel (k_wino): one PERSISTENT workgroup per CU = 4 waves (2 x 2); a unit = 64 tiles x 64 output channels; every wave owns 32 tiles x
// 32 channels for ALL 16 positions = 16 accumulator sets of 16 registers = 256 AGPRs (one wave per SIMD), so the output transform is
// register arithmetic. K loop over 16-channel slices:
//   V (B operand): every thread owns one (tile, 4-channel quad): 16 buffer loads of 16 B (its 4x4 patch; out of the image = the range
//     check's zeros), the optional fused input activation relu(fma(x, scale, shift)) with the padding selected to zero AFTER it (exactly
//     osi_conv_fwd_act's operand), B^T d B in registers, 16 ds_write_b128 into V[buf][p][tile][16 ch] — 16-byte chunks XOR-swizzled by
//     the tile index: conflict-free b128 writes and reads without padding, 2 x 64 KiB double buffered, one barrier per slice;
//   U (A operand): transformed once per launch by k_wino_weights INTO THE PER-LANE FRAGMENT ORDER and read straight from L2 into
//     registers (1 KiB contiguous per wave-load), software pipelined PF positions ahead on the in-order vmcnt queue — no LDS.
// The pipeline runs across units: the last slice of a unit stages the first slice of the workgroup's next unit, so only the first unit
// pays an exposed prologue. Program order is pinned per position with sched_barrier: the compiler otherwise sinks the loads next to
// their uses. fp32 MFMA runs at the vector rate — VALU work beside it is NOT hidden, so the patch transform (one clump per position,
// positions 8..15) is priced in full: ~25 % of the forward form, ~10 % of the input-gradient form (no activation).
//
// Epilogues (per lane: one tile = 2 x 2 pixels, 16 channels in 4 quads):
//   EPI 0 forward: y + optional BatchNorm (mean, M2) partials per 16 tiles (= one DPP row of lanes = 64 pixels, the row-tile size of the
//     direct kernels: P and rows_per_block feed osi_bn_finalize_stats unchanged), reduced with 4 DPP adds per value — no LDS, no barrier;
//   EPI 1 input gradient, "in-block" fusion of osi_conv_dgrad_fused: dx = gate . acc with the gate recomputed from the producer's pre-BN
//     tensor (fma(y0, scale0, shift0) > 0), and the per-64-pixel sums of g and g * xhat0 for the BatchNorm backward of that producer.
#include "conv_common.h"
#include <algorithm>
#include <type_traits>
#include <utility>

using namespace osi_conv;

namespace {

// diagnostic builds of the weight-gradient kernel only (tools): 1 = no operand loads inside the K loop, 2 = no transform / LDS stores (wrong results)
#ifndef OSI_WABL
#define OSI_WABL 0
#endif
// the same for k_wino (forward / input gradient): 1 = U fragments from one cache-resident address, 2 = no patch loads inside the K loop,
// 4 = no patch transform / LDS stores, 8 = no LDS operand reads, 16 = no epilogue arithmetic / stores
#ifndef OSI_FABL
#define OSI_FABL 0
#endif
// patch loads of the next slice: 0 = all sixteen at the top of a slice, n = spread over the first n positions (a burst blocks at the texture
// addresser's queue: 8 is 5 - 8 % faster than 0 on the 28^2 / 14^2 / 7^2 layers, profiles/NOTES_r05.md)
// output tile staged through LDS (1) or stored by its owner lanes (0)
#ifndef OSI_STAGE_OUT
#define OSI_STAGE_OUT 1
#endif
#ifndef OSI_XSPREAD
#define OSI_XSPREAD 8
#endif
constexpr int KC = 16;      // channels per K slice
constexpr int PF = 6;       // positions the U fragments are loaded ahead

struct WinoP {
    const float* x;      // operand tensor [B][H][W][Kc] (forward: the conv input, pre-activation when `sc`; input gradient: dY)
    const float* u;      // transformed weights in fragment order (k_wino_weights)
    float* y;            // result [B][H][W][Nc]
    const float* sc;     // fused input activation (forward), or NULL
    const float* sh;
    int H, W, Kc, Nc, TH, TW, T, KS, CB, MT, NT;
    int x_bytes, y_bytes, u_bytes;
    // forward statistics: [P][Nc] means then [P][Nc] M2, P = ceil(T / 16); cnt = valid pixels of a 16-tile group
    float* pmean;
    float* pm2;
    int P;
    float cnt, rcnt;
    // input gradient, in-block epilogue
    const float* ey0;    // producer's pre-BN tensor, same shape as y
    const float *escale0, *eshift0, *emean0, *einv0;
    float* esum;         // [3][P][Nc]: sum g, sum g * xhat0 (third plane unused)
    // forward, inference epilogue (EPI 2): y = [relu](fma(acc, osc[n], osh[n])) — the conv's own BatchNorm (eval coefficients) + ReLU
    const float *osc, *osh;
    int orelu;
    // work decomposition: q full rounds of whole units, r remainder units cut along K into G pieces (slab: 2 slots of 64 KiB per workgroup)
    int q, r, nfull;
    float* slab;
    int slab_bytes;
};

// Two fp32 adds / subtracts per lane in one instruction. Written as asm: left to itself the compiler scalarises <2 x float> arithmetic
// inside the pinned slice loop (v_add_f32_e64 per element), and the statement order is part of the kernel's schedule anyway.
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 r;
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 r;
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_neg(f32x2 a) {          // 0 - a (the sign of a zero aside, which no consumer here can tell)
    f32x2 r;
    asm volatile("v_pk_add_f32 %0, 0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 r;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ f32x4 pk_add(f32x4 a, f32x4 b) {
    const f32x2 lo = pk_add(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
    const f32x2 hi = pk_add(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
__device__ __forceinline__ f32x4 pk_sub(f32x4 a, f32x4 b) {
    const f32x2 lo = pk_sub(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1));
    const f32x2 hi = pk_sub(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

template <class F, int... I>
__device__ __forceinline__ void for_each_const(F& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

// sum over the 16 lanes of a DPP row (= 16 tiles = one statistics group), result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

// U_p = (G g G^T)_p in the order the MFMA A fragments are consumed:
//   index = ((((p * KS + ks) * CB + cb) * 2 + j) * 64 + lane) * 4 + e   <->   k = 16 ks + 8 j + 4 (lane / 32) + e,  n = 32 cb + lane % 32
// (lane half h of MFMA (j, e) of a slice consumes channel 8 j + 4 h + e: the order in which V's channels sit in LDS).
// w: KRSC [Cout][3][3][Cin].  FLIP = 0 (forward): k = cin, n = cout, g = w[n][.][.][k].  FLIP = 1 (input gradient): k = cout, n = cin,
// g[r][s] = w[k][2 - r][2 - s][n] (taps rotated by 180 degrees).
template <int FLIP>
__global__ __launch_bounds__(256) void k_wino_weights(const float* __restrict__ w, float* __restrict__ u, int Kc, int Nc, int KS, int CB) {
    // one thread per (4 consecutive k = one 16-byte fragment element group, n): 16 stores of 16 B, consecutive n = consecutive lanes = 512 B runs
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int K4 = Kc / 4;
    if (idx >= K4 * Nc) return;
    const int n = idx % Nc, k0 = (idx / Nc) * 4;
    f32x4 g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (FLIP) {
#pragma unroll
                for (int e = 0; e < 4; ++e) g[r][s][e] = w[((size_t)((k0 + e) * 3 + (2 - r)) * 3 + (2 - s)) * Nc + n];
            } else g[r][s] = ld4(w + ((size_t)(n * 3 + r) * 3 + s) * Kc + k0);
        }
    f32x4 t[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        t[0][s] = g[0][s];
        t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
        t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
        t[3][s] = g[2][s];
    }
    const int ks = k0 / KC, j = (k0 % KC) / 8, h = (k0 % 8) / 4, cb = n / 32, lane = h * 32 + n % 32;
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
        f32x4 v[4];
        v[0] = t[xi][0];
        v[1] = 0.5f * (t[xi][0] + t[xi][1] + t[xi][2]);
        v[2] = 0.5f * (t[xi][0] - t[xi][1] + t[xi][2]);
        v[3] = t[xi][2];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            const int p = xi * 4 + nu;
            *reinterpret_cast<f32x4*>(u + ((((size_t)(p * KS + ks) * CB + cb) * 2 + j) * 64 + lane) * 4) = v[nu];
        }
    }
}

// ---- work decomposition ------------------------------------------------------------------------------------------------------------
// V = MT x NT units in an XCD-friendly linear order (the NT column units of a tile block consecutive on one XCD: workgroup w and its units
// w + k G share blockIdx % 8). One workgroup per CU (G of them) would leave the last of ceil(V / G) rounds ragged — at 1.53 / 3.06 / 6.125
// units per CU (the 14x14 / 28x28 / 56x56 layers at batch 128) that is 23 / 23 / 12 % of the launch. So only the q = V / G FULL rounds run
// as whole units (phase 1: unit v = w + k G, ordinary epilogue); the r = V - q G remaining units are cut STREAM-K style: their r x KS
// K slices form one line that is dealt to the G workgroups in equal contiguous pieces (phase 2: workgroup w takes slices
// [w r KS / G, (w + 1) r KS / G), at most two units touched), each piece's output-transformed partial tile goes to a slab slot
// (w, first / second unit) in the per-lane register order, and k_wino_fixup adds a unit's pieces in workgroup order — fixed, so the result
// is bitwise reproducible — and runs the same epilogue code. The software pipeline crosses unit, phase and piece boundaries.
struct Seg { int v, ks0, ks1, slot; bool valid; };     // unit, K-slice range [ks0, ks1), slab slot (-1: a whole unit, ordinary epilogue)

__device__ __forceinline__ void unit_of(int v, int MT, int NT, int& mt, int& nt) {
    const int full = (MT / 8) * 8 * NT;                // units of the complete groups of 8 tile blocks
    if (v < full) {
        const int xcd = v & 7, slot = v >> 3;
        nt = slot % NT;
        mt = (slot / NT) * 8 + xcd;
    } else {
        const int rem = MT & 7, vv = v - full;         // the last (MT % 8) tile blocks: tile block fastest
        mt = (MT / 8) * 8 + vv % rem;
        nt = vv / rem;
    }
}
__device__ __forceinline__ int piece_begin(int w, int r, int KS, int G) { return (int)(((long)w * r * KS) / G); }

// first segment of workgroup w / the one after `c` (valid = false: nothing left)
__device__ __forceinline__ Seg seg_phase2(int w, int q, int r, int KS, int G, bool second, int first_u) {
    Seg s{0, 0, 0, -1, false};
    if (r == 0) return s;
    const int b0 = piece_begin(w, r, KS, G), b1 = piece_begin(w + 1, r, KS, G);
    if (b1 <= b0) return s;
    const int u0 = b0 / KS;
    if (!second) {
        s.v = q * G + u0; s.ks0 = b0 - u0 * KS; s.ks1 = min(KS, s.ks0 + (b1 - b0)); s.slot = 0; s.valid = true;
        return s;
    }
    if (b1 > (u0 + 1) * KS) { s.v = q * G + u0 + 1; s.ks0 = 0; s.ks1 = b1 - (u0 + 1) * KS; s.slot = 1; s.valid = true; }
    return s;
}
// nfull = units that run whole (q G; or all V of them when the remainder is not cut: knob wino_streamk = 0, then r = 0)
__device__ __forceinline__ Seg seg_first(int w, int nfull, int q, int r, int KS, int G) {
    if (w < nfull) return Seg{w, 0, KS, -1, true};
    return seg_phase2(w, q, r, KS, G, false, 0);
}
__device__ __forceinline__ Seg seg_next(const Seg& c, int w, int nfull, int q, int r, int KS, int G) {
    if (c.slot < 0) {                                   // phase 1
        if (c.v + G < nfull) return Seg{c.v + G, 0, KS, -1, true};
        return seg_phase2(w, q, r, KS, G, false, 0);
    }
    if (c.slot == 0) return seg_phase2(w, q, r, KS, G, true, 0);
    return Seg{0, 0, 0, -1, false};
}

// ---- epilogue of one channel quad (4 pixels x 4 consecutive channels of this lane's tile), shared by k_wino and k_wino_fixup -------------
// Forward: y + optional (mean, M2) statistics per 16 tiles.
// `put(k, g, v)` stores the 16 bytes of pixel k, channel quad g: straight to global memory (fix-up pass) or into the workgroup's LDS staging
// tile, from where whole pixel rows leave in 16-byte lanes (k_wino: a lane's own stores go to 64 different 128-byte lines per instruction)
template <bool ODD, class PUT>
__device__ __forceinline__ void epi0_quad(const WinoP& p, PUT put, int g, f32x4 (&o)[4], const uint32_t (&po)[4], int ch0,
                                          int part, int lane) {
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) put(k, g, o[k]);
        if (p.pmean) {
            // (mean, M2) of the group's valid pixels per channel: sum -> mean -> sum of squared deviations, every reduction a fixed-order
            // DPP row sum. Slots outside the image (ODD) are zeroed and their (0 - mean)^2 taken back out.
            f32x4 s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if constexpr (ODD) { if (po[k] == OOB) o[k] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                s1 += o[k];
            }
            f32x4 mean, m2;
#pragma unroll
            for (int e = 0; e < 4; ++e) mean[e] = row16_sum(s1[e]) * p.rcnt;
            f32x4 s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 dlt = o[k] - mean;
                s2 += dlt * dlt;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                m2[e] = row16_sum(s2[e]);
                if constexpr (ODD) m2[e] -= (64.f - p.cnt) * mean[e] * mean[e];
            }
            if ((lane & 15) == 0 && part < p.P) {
                *reinterpret_cast<f32x4*>(p.pmean + (size_t)part * p.Nc + ch0 + 8 * g) = mean;
                *reinterpret_cast<f32x4*>(p.pm2 + (size_t)part * p.Nc + ch0 + 8 * g) = m2;
            }
        }
    }
}

// Inference epilogue of one channel quad: the convolution's own BatchNorm (eval-mode scale / shift) and ReLU on the way out — one fma and
// one max per element, the expression the fused loaders evaluate on the consumer side in training (osi_conv_fwd_act)
template <class PUT>
__device__ __forceinline__ void epi2_quad(const WinoP& p, PUT put, int g, const f32x4 (&o)[4], const f32x4& sc, const f32x4& sh) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float t = __builtin_fmaf(o[k][e], sc[e], sh[e]); v[e] = p.orelu ? fmaxf(t, 0.f) : t; }
        put(k, g, v);
    }
}

// Operands of the input-gradient epilogue of one channel quad: the producer's pre-BN values at this lane's four pixels and the per-channel
// gate / xhat coefficients. Loaded one quad AHEAD of their use (k_wino) — a lone wave per SIMD has nothing else to cover the round trip.
struct Epi1Ops { f32x4 y0[4], esc, esh, emu, einv; };
__device__ __forceinline__ Epi1Ops epi1_load(const WinoP& p, __amdgpu_buffer_rsrc_t r0y, int g, const uint32_t (&po)[4], int ch0) {
    Epi1Ops L;
#pragma unroll
    for (int k = 0; k < 4; ++k) L.y0[k] = bld4(r0y, po[k] == OOB ? OOB : po[k] + 32u * g, 0);
    L.esc = ld4(p.escale0 + ch0 + 8 * g); L.esh = ld4(p.eshift0 + ch0 + 8 * g);
    L.emu = ld4(p.emean0 + ch0 + 8 * g); L.einv = ld4(p.einv0 + ch0 + 8 * g);
    return L;
}
// g = gate . acc, gate = fma(y0, scale0, shift0) > 0 (the forward loader's own expression); sums of g and g * xhat0 per 64 pixels
template <bool ODD, class PUT>
__device__ __forceinline__ void epi1_apply(const WinoP& p, PUT put, int g, const f32x4 (&o)[4], const uint32_t (&po)[4],
                                           int ch0, int part, int lane, const Epi1Ops& L) {
    f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sgx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f32x4 gv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bool on = __builtin_fmaf(L.y0[k][e], L.esc[e], L.esh[e]) > 0.f;
            if constexpr (ODD) on = on && po[k] != OOB;
            gv[e] = on ? o[k][e] : 0.f;
            sg[e] += gv[e];
            sgx[e] += gv[e] * ((L.y0[k][e] - L.emu[e]) * L.einv[e]);
        }
        put(k, g, gv);
    }
    if (p.esum) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { sg[e] = row16_sum(sg[e]); sgx[e] = row16_sum(sgx[e]); }
        if ((lane & 15) == 0 && part < p.P) {
            *reinterpret_cast<f32x4*>(p.esum + (size_t)part * p.Nc + ch0 + 8 * g) = sg;
            *reinterpret_cast<f32x4*>(p.esum + ((size_t)p.P + part) * p.Nc + ch0 + 8 * g) = sgx;
        }
    }
}

// byte offsets of this lane's four output pixels (tile `ntile` of tile block mt, channel base of column unit nt)
template <bool WIDE>
__device__ __forceinline__ void out_offsets(const WinoP& p, int mt, int nt, int ntile, int wm, int hh, bool live, uint32_t (&po)[4]) {
    const int THW = p.TH * p.TW;
    const int t = mt * (WIDE ? 32 : 64) + ntile;
    const int b = t / THW, rem = t - b * THW, th = rem / p.TW, tw = rem - th * p.TW;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int yy = 2 * th + i, xx = 2 * tw + j;
            const bool ok = live && t < p.T && yy < p.H && xx < p.W;
            po[i * 2 + j] = ok ? (uint32_t)((((b * p.H + yy) * p.W + xx) * p.Nc + (WIDE ? 128 : 64) * nt + 32 * wm + 4 * hh) * 4) : OOB;
        }
}

// XF: fused input activation. EPI: 0 forward (+ statistics when p.pmean), 1 input gradient with the in-block fused epilogue, 2 forward with
// the inference epilogue (own BatchNorm + ReLU applied to the output).
// ODD: some tile slots hold pixels outside the image (odd H / W) or past the last tile — their outputs are masked out of the sums.
// WIDE: the unit is 32 tiles x 128 channels instead of 64 x 64 — the four waves own the four 32-channel blocks of the SAME 32 tiles, so a
// tile's patch transform (the VALU work fp32 MFMAs cannot hide) serves twice as many channels: every thread transforms a (tile, channel
// PAIR) instead of a (tile, channel quad). For output-channel counts in 128s.
template <bool XF, int EPI, bool ODD, bool WIDE>
__global__ __launch_bounds__(256, 1) void k_wino(WinoP p) {
    constexpr int TB = WIDE ? 32 : 64, CW = WIDE ? 128 : 64, NV = WIDE ? 2 : 4;      // tiles / channels of a unit, channels per loader thread
    // V double buffer (2 x 64 KiB; 2 x 32 KiB WIDE) [+ WIDE: a 64 KiB output staging tile; the 64 x 64 form stages in the free V buffer]
    __shared__ __attribute__((aligned(16))) float sV[2 * 16 * TB * KC + (WIDE ? 16384 : 0)];
    __shared__ uint32_t sPo[256];        // byte offset of every pixel row of the unit's output tile (channel 0 of the unit), or OOB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = WIDE ? wave : wave >> 1, wn = WIDE ? 0 : wave & 1;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y, p.y_bytes);
    const __amdgpu_buffer_rsrc_t ru = make_rsrc(p.u, p.u_bytes);
    const int G = gridDim.x, wg = blockIdx.x;
    const int ltile = WIDE ? tid >> 3 : tid >> 2, q = WIDE ? tid & 7 : tid & 3;      // loader: tile, channel quad (pair when WIDE)
    const int ntile = 32 * wn + (lane & 31), hh = lane >> 5;
    const int THW = p.TH * p.TW;

    Seg seg = seg_first(wg, p.nfull, p.q, p.r, p.KS, G);
    if (!seg.valid) return;
    int mt, nt;
    unit_of(seg.v, p.MT, p.NT, mt, nt);

    uint32_t off[16];
    unsigned long long okm[16];      // lane masks of the valid patch pixels (v_cndmask's scalar operand; fused activation only)
    uint32_t po[4], po_next[4];      // byte offsets of this lane's four output pixels (+ channel base), current / next unit
    uint32_t ua, ua_next;            // byte offset of this lane's U fragments inside a (position, slice) block: (channel block, lane)
    // everything that depends on the unit: loader offsets + masks, epilogue offsets, U fragment base
    auto setup = [&](int mt_, int nt_, bool live, uint32_t (&po_)[4], uint32_t& ua_) {
        {
            const int t = mt_ * TB + ltile;
            const int b = t / THW, rem = t - b * THW, th = rem / p.TW, tw = rem - th * p.TW;
            const uint32_t base = (uint32_t)((((b * p.H + 2 * th - 1) * p.W + 2 * tw - 1) * p.Kc + q * NV) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int yy = 2 * th - 1 + i, xx = 2 * tw - 1 + j;
                    const bool ok = live && t < p.T && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
                    off[i * 4 + j] = ok ? base + (uint32_t)((i * p.W + j) * p.Kc * 4) : OOB;
                    if constexpr (XF) okm[i * 4 + j] = __builtin_amdgcn_ballot_w64(ok);
                }
        }
        out_offsets<WIDE>(p, mt_, nt_, ntile, wm, hh, live, po_);
        ua_ = (uint32_t)((((CW / 32) * nt_ + wm) * 2 * 64 + lane) * 16);       // + ((pos * KS + ks) * CB) * 2048 + j * 1024 bytes
    };
    setup(mt, nt, true, po, ua);
    // the thread's channels inside a tile's 16-channel row: 16-byte chunk (XOR-swizzled by the tile index) + offset inside the chunk
    float* const wbase = sV + ltile * KC + (WIDE ? 4 * ((q >> 1) ^ ((ltile >> 2) & 3)) + 2 * (q & 1) : 4 * (q ^ ((ltile >> 2) & 3)));
    const float* const rb0 = sV + ntile * KC + 4 * ((0 + hh) ^ ((ntile >> 2) & 3));   // chunk of MFMAs j = 0
    const float* const rb1 = sV + ntile * KC + 4 * ((2 + hh) ^ ((ntile >> 2) & 3));   // chunk of MFMAs j = 1
    const uint32_t ustep = (uint32_t)p.CB * 2048;      // bytes per (position, slice) block of U

    f32x16 acc[16];
    // the thread's 4 x 4 patch, NV channels per pixel. The transforms below are written on whole XT vectors: fp32 MFMAs hide no vector work
    // (profiles/r06_mfma_valu_coexec.txt), so every instruction counts, and two adds in one v_pk_add_f32 cost a lone wave 5.6 cycles
    // against 2 x 4.7 for the scalar pair
    using XT = std::conditional_t<WIDE, f32x2, f32x4>;
    XT xr[16];
    auto load_x1 = [&](int ks, int k) {
        if constexpr (WIDE) xr[k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rx, off[k], (uint32_t)ks * (KC * 4), 0));
        else xr[k] = bld4(rx, off[k], (uint32_t)ks * (KC * 4));
    };
    auto load_x = [&](int ks) {
#pragma unroll
        for (int k = 0; k < 16; ++k) load_x1(ks, k);
    };
    // activation (+ padding select) and column transform of patch column j. asm: IR passes otherwise regroup these scalar ops (SLP packs
    // them into v_pk_*, an anti-lever beside MFMAs, and sinks the selects to their users) whatever the machine scheduler is told.
    auto act_col = [&](int j, const f32x4& sc4, const f32x4& sh4) {
        if constexpr (XF) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < NV; ++e)
                    asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_max_f32 %0, 0, %0\n\tv_cndmask_b32 %0, 0, %0, %3"
                                 : "+v"(xr[i * 4 + j][e]) : "v"(sc4[e]), "v"(sh4[e]), "s"(okm[i * 4 + j]));
        }
        {
            const XT d0 = xr[j], d1 = xr[4 + j], d2 = xr[8 + j], d3 = xr[12 + j];
            xr[j] = pk_sub(d0, d2); xr[4 + j] = pk_add(d1, d2); xr[8 + j] = pk_sub(d2, d1); xr[12 + j] = pk_sub(d1, d3);
        }
    };
    // row transform of patch row i and its four LDS stores
    auto row_store = [&](int i, int buf) {
        float* w = wbase + buf * (16 * TB * KC) + (i * 4) * (TB * KC);
        const XT t0 = xr[i * 4], t1 = xr[i * 4 + 1], t2 = xr[i * 4 + 2], t3 = xr[i * 4 + 3];
        const XT o[4] = {pk_sub(t0, t2), pk_add(t1, t2), pk_sub(t2, t1), pk_sub(t1, t3)};
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) *reinterpret_cast<XT*>(w + nu * (TB * KC)) = o[nu];
    };
    auto ld_scale = [&](int ks, f32x4& sc4, f32x4& sh4) {
        if constexpr (XF) {
            if constexpr (WIDE) {
                const f32x2 a = *reinterpret_cast<const f32x2*>(p.sc + ks * KC + 2 * q), b = *reinterpret_cast<const f32x2*>(p.sh + ks * KC + 2 * q);
                sc4[0] = a[0]; sc4[1] = a[1]; sh4[0] = b[0]; sh4[1] = b[1];
            } else {
                sc4 = ld4(p.sc + ks * KC + 4 * q);
                sh4 = ld4(p.sh + ks * KC + 4 * q);
            }
        }
    };

    // prologue (first segment of this workgroup only): its first slice into buffer 0
    f32x4 sc4 = {1, 1, 1, 1}, sh4 = {0, 0, 0, 0};
    load_x(seg.ks0);
    ld_scale(seg.ks0, sc4, sh4);
#pragma unroll
    for (int j = 0; j < 4; ++j) act_col(j, sc4, sh4);
#pragma unroll
    for (int i = 0; i < 4; ++i) row_store(i, 0);

    // U fragment ring: slot (pos % 16), loaded PF positions ahead across slice AND segment boundaries
    f32x4 a[16][2];
    auto load_u = [&](int pos, uint32_t base, int ks) {
        const uint32_t soff = (OSI_FABL & 1) ? 0u : (uint32_t)(pos * p.KS + ks) * ustep;      // uniform: the scalar offset of the buffer load
        a[pos][0] = bld4(ru, base, soff);
        a[pos][1] = bld4(ru, base + 1024u, soff);
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) load_u(i, ua, seg.ks0);
    __syncthreads();

    int buf = 0;
    for (;;) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        Seg nxt{0, 0, 0, -1, false};
        const int cur_mt = mt, cur_nt = nt;
        for (int ks = seg.ks0; ks < seg.ks1; ++ks) {
            const bool last = ks + 1 == seg.ks1;
            int ksn = ks + 1;
            uint32_t uan = ua;
            if (last) {      // uniform: the next segment of this workgroup (or a dead one: every offset out of range) — no loads in here
                nxt = seg_next(seg, wg, p.nfull, p.q, p.r, p.KS, G);
                if (nxt.valid) unit_of(nxt.v, p.MT, p.NT, mt, nt);
                setup(mt, nt, nxt.valid, po_next, ua_next);
                ksn = nxt.valid ? nxt.ks0 : 0;
                uan = nxt.valid ? ua_next : ua;
            }
#if !(OSI_FABL & 2) && !OSI_XSPREAD
            load_x(ksn);
#endif
            ld_scale(ksn, sc4, sh4);
            const float* r0 = rb0 + buf * (16 * TB * KC);
            const float* r1 = rb1 + buf * (16 * TB * KC);
            f32x4 b0 = *reinterpret_cast<const f32x4*>(r0), b1 = *reinterpret_cast<const f32x4*>(r1);
            __builtin_amdgcn_sched_barrier(0);
            auto position = [&](auto POSC) {
                constexpr int pos = decltype(POSC)::value;
                if (pos + PF < 16) load_u(pos + PF, ua, ks);
                else load_u(pos + PF - 16, uan, ksn);
#if OSI_XSPREAD && !(OSI_FABL & 2)
                if constexpr (pos < OSI_XSPREAD) {      // the next slice's 16 patch loads, a few behind each of the first positions
                    constexpr int n0 = pos * 16 / OSI_XSPREAD, n1 = (pos + 1) * 16 / OSI_XSPREAD;
#pragma unroll
                    for (int k = n0; k < n1; ++k) load_x1(ksn, k);
                }
#endif
                f32x4 nb0 = b0, nb1 = b1;
                if (pos < 15 && !(OSI_FABL & 8)) {
                    nb0 = *reinterpret_cast<const f32x4*>(r0 + (pos + 1) * (TB * KC));
                    nb1 = *reinterpret_cast<const f32x4*>(r1 + (pos + 1) * (TB * KC));
                }
                const f32x4 a0 = a[pos][0], a1 = a[pos][1];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[pos] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[pos], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[pos] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[pos], 0, 0, 0);
#if !(OSI_FABL & 4)
                if constexpr (pos >= 8 && pos < 12) act_col(pos - 8, sc4, sh4);
                if constexpr (pos >= 12) row_store(pos - 12, buf ^ 1);
#endif
                b0 = nb0; b1 = nb1;
                __builtin_amdgcn_sched_barrier(0);
            };
            for_each_const(position, std::make_integer_sequence<int, 16>{});
            __syncthreads();
            buf ^= 1;
        }

        // ---- epilogue: Y = A^T M A per (tile, channel); per lane 4 pixels x 4 quads of 4 consecutive channels ------------------------
        const int ch0 = CW * cur_nt + 32 * wm + 4 * hh;                                  // + 8 g: first channel of quad g
        const int part = cur_mt * (TB / 16) + 2 * wn + ((lane >> 4) & 1);                // this lane's 16-tile statistics group
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.slab, p.slab_bytes);
        const __amdgpu_buffer_rsrc_t r0y = make_rsrc(EPI == 1 ? p.ey0 : p.y, p.y_bytes);
        Epi1Ops L1{};
        if constexpr (EPI == 1) { if (seg.slot < 0) L1 = epi1_load(p, r0y, 0, po, ch0); }
        float* const stage = WIDE ? sV + 2 * 16 * TB * KC : sV + (buf ^ 1) * (16 * TB * KC);      // free: the K loop's last barrier is behind us
        if (seg.slot < 0 && wm == 0 && hh == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) sPo[ntile * 4 + k] = po[k];       // (wm = 0, hh = 0: po is the offset of the unit's channel 0)
        }
#pragma unroll
        for (int g = (OSI_FABL & 16) ? 3 : 0; g < 4; ++g) {
            f32x4 o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                float s[2][4];
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) {
                    // the accumulators live in AGPRs; read HERE, one register at a time ("a" operand): left to itself the compiler copies all
                    // sixteen tuples into VGPRs where the epilogue's control flow begins (256 moves + spills of the pipeline's live state)
                    float m0, m1, m2, m3;
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m0) : "a"(acc[nu][r]));
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m1) : "a"(acc[4 + nu][r]));
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m2) : "a"(acc[8 + nu][r]));
                    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m3) : "a"(acc[12 + nu][r]));
                    s[0][nu] = m0 + m1 + m2;
                    s[1][nu] = m1 - m2 - m3;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    o[i * 2][e] = s[i][0] + s[i][1] + s[i][2];
                    o[i * 2 + 1][e] = s[i][1] - s[i][2] - s[i][3];
                }
            }
            if (seg.slot < 0) {
                // into the staging tile [pixel row = 4 tile + k][CW channels], 16-byte chunks XOR-swizzled by the tile index
                auto put = [&](int k, int gq, const f32x4& v) {
#if OSI_STAGE_OUT
                    const int chunk = (32 * wm + 8 * gq + 4 * hh) >> 2;
                    *reinterpret_cast<f32x4*>(stage + (ntile * 4 + k) * CW + 4 * (chunk ^ (ntile & 15))) = v;
#else
                    bst4(ry, v, po[k] == OOB ? OOB : po[k] + 32u * gq, 0);
#endif
                };
                if constexpr (EPI == 0) epi0_quad<ODD>(p, put, g, o, po, ch0, part, lane);
                else if constexpr (EPI == 2) epi2_quad(p, put, g, o, ld4(p.osc + ch0 + 8 * g), ld4(p.osh + ch0 + 8 * g));
                else {
                    const Epi1Ops Ln = epi1_load(p, r0y, g < 3 ? g + 1 : 3, po, ch0);      // the next quad's operands behind this quad's arithmetic
                    epi1_apply<ODD>(p, put, g, o, po, ch0, part, lane, L1);
                    L1 = Ln;
                }
            } else {           // a piece of a remainder unit: the partial tile in register order (the output transform is linear)
                const uint32_t so = (uint32_t)((wg * 2 + seg.slot) * 65536 + ((wave * 4 + g) * 4) * 1024 + lane * 16);
#pragma unroll
                for (int k = 0; k < 4; ++k) bst4(rs, o[k], so + (uint32_t)k * 1024u, 0);
            }
        }
        if (OSI_STAGE_OUT && seg.slot < 0) {
            // whole pixel rows out of the staging tile: consecutive lanes = consecutive 16-byte chunks of a row (256 / 512 contiguous bytes)
            __syncthreads();
            constexpr int CPR = CW / 4, NCH = TB * 4 * CPR;      // chunks per row, chunks of the tile
#pragma unroll 4
            for (int it = 0; it < NCH / 256; ++it) {
                const int idx = it * 256 + tid, row = idx / CPR, chunk = idx % CPR;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * CW + 4 * (chunk ^ ((row >> 2) & 15)));
                const uint32_t base = sPo[row];
                bst4(ry, v, base == OOB ? OOB : base + (uint32_t)chunk * 16u, 0);
            }
            __syncthreads();       // the next unit's K loop writes V into this region again
        }
        if (!nxt.valid) break;
        seg = nxt;
#pragma unroll
        for (int k = 0; k < 4; ++k) po[k] = po_next[k];
        ua = ua_next;
    }
}

// One WAVE per (remainder unit, wave role, channel quad) — 16 single-wave workgroups per unit, so that a small remainder still fills the
// chip: adds the unit's pieces (slab slots of the workgroups whose slice ranges meet it, in workgroup order) and runs the ordinary
// epilogue of that quad with k_wino's own lane roles (the statistics / sums are per DPP row of 16 lanes: a wave is self-contained).
template <int EPI, bool ODD, bool WIDE>
__global__ __launch_bounds__(64) void k_wino_fixup(WinoP p, int G) {
    const int lane = threadIdx.x, u = blockIdx.x >> 4, wave = (blockIdx.x >> 2) & 3, g = blockIdx.x & 3;
    const int wm = WIDE ? wave : wave >> 1, wn = WIDE ? 0 : wave & 1;
    const int ntile = 32 * wn + (lane & 31), hh = lane >> 5;
    int mt, nt;
    unit_of(p.q * G + u, p.MT, p.NT, mt, nt);
    uint32_t po[4];
    out_offsets<WIDE>(p, mt, nt, ntile, wm, hh, true, po);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y, p.y_bytes);
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.slab, p.slab_bytes);
    const int s0 = u * p.KS, s1 = s0 + p.KS;
    int w = (int)(((long)s0 * G) / ((long)p.r * p.KS));         // a workgroup at or before the first one that meets the unit
    w = w > 0 ? w - 1 : 0;
    while (piece_begin(w + 1, p.r, p.KS, G) <= s0) ++w;
    f32x4 o[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    int b0 = piece_begin(w, p.r, p.KS, G);
    for (int ww = w; ww < G && b0 < s1; ++ww) {
        const int b1 = piece_begin(ww + 1, p.r, p.KS, G);
        if (b1 > b0) {
            const int slot = (b0 / p.KS == u) ? 0 : 1;
            const uint32_t so = (uint32_t)((ww * 2 + slot) * 65536 + ((wave * 4 + g) * 4) * 1024 + lane * 16);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] += bld4(rs, so + (uint32_t)k * 1024u, 0);
        }
        b0 = b1;
    }
    const int ch0 = (WIDE ? 128 : 64) * nt + 32 * wm + 4 * hh, part = mt * (WIDE ? 2 : 4) + 2 * wn + ((lane >> 4) & 1);
    auto put = [&](int k, int gq, const f32x4& v) { bst4(ry, v, po[k] == OOB ? OOB : po[k] + 32u * gq, 0); };
    if constexpr (EPI == 0) epi0_quad<ODD>(p, put, g, o, po, ch0, part, lane);
    else if constexpr (EPI == 2) epi2_quad(p, put, g, o, ld4(p.osc + ch0 + 8 * g), ld4(p.osh + ch0 + 8 * g));
    else {
        const Epi1Ops L = epi1_load(p, make_rsrc(p.ey0, p.y_bytes), g, po, ch0);
        epi1_apply<ODD>(p, put, g, o, po, ch0, part, lane, L);
    }
}

// =====================================================================================================================================
// Weight gradient, Winograd F(3x3, 2x2): the transpose of the forward algorithm.  dW = sum over tiles of  G^T [ (A dY A^T) (.) (B^T d B) ] G
// (d = the 4x4 input patch of a tile — the SAME transform as the forward's V —, dY = the tile's 2x2 output gradients, A = the forward's
// output-transform matrix transposed). The sum over tiles is taken in the transformed domain:
//   S_p[cout][cin] = sum_tiles  Yt_p[cout][tile] * V_p[cin][tile],   p in 4 x 4   -> 16 GEMMs whose K axis is the TILE axis,
// then dW[cout][.][.][cin] = G^T S G (16 -> 9 values per (cout, cin), register arithmetic in the epilogue): 16 multiplies per tile and
// (cout, cin) pair instead of 36.
// One workgroup = one (64 couts x 64 cins) block x one contiguous run of tiles (split-K over the tile axis: blocks x splits = the CUs);
// every wave owns 32 x 32 of the block for all 16 positions (256 AGPRs, one wave per SIMD). K step = 8 tiles, both operands staged through
// LDS (images [p][k-half h][channel][4 tiles]: a lane's 16-byte read = its operand of the four MFMAs of a position; 2 x 32 KiB per stage,
// double buffered). Loader roles: wave w owns tiles {4 (w >> 1) + (w & 1), + 2} of the step for EVERY channel (lane = channel): tile
// coordinates, image-border validity and all load offsets are SCALAR (one buffer load per pixel with the offset in an SGPR), the
// transforms are per-lane scalar math on 2 tiles x 1 channel, 16-byte... 8-byte LDS stores.
struct WgradP {
    const float* x;      // conv input [B][H][W][Cin] (pre-activation when `sc`)
    const float* dy;     // [B][H][W][Cout]
    float* slab;         // [item][9][64][64] partial gradients, item = (kb * CBn + cb) * S + split
    const float* sc;     // fused input activation or NULL
    const float* sh;
    int H, W, Cin, Cout, TH, TW, T;
    int CBn, S, steps;   // cin blocks, splits per block, K steps (of 8 tiles) per split
    int x_bytes, dy_bytes;
    FastDiv dTHW, dTW;
};

template <bool XF>
__global__ __launch_bounds__(256, 1) void k_wino_wgrad(WgradP p) {
    __shared__ __attribute__((aligned(16))) float sL[2 * 2 * 16 * 2 * 64 * 4];   // [buf][operand][p][h][channel][4 tiles]: 128 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int item = blockIdx.x;
    const int split = item % p.S, blk = item / p.S, cb = blk % p.CBn, kb = blk / p.CBn;
    // loader role of this WAVE: k-half h and pair of k-pairs
    const int lh = __builtin_amdgcn_readfirstlane(wave & 1), kkp = __builtin_amdgcn_readfirstlane(wave >> 1);
    const uint32_t xlane = (uint32_t)((cb * 64 + lane) * 4), dlane = (uint32_t)((kb * 64 + lane) * 4);
    const int t_first = split * p.steps * 8;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(p.dy, p.dy_bytes);
    f32x2 scsh = {1.f, 0.f};
    if constexpr (XF) { scsh[0] = p.sc[cb * 64 + lane]; scsh[1] = p.sh[cb * 64 + lane]; }
    const int cin4 = p.Cin * 4, cout4 = p.Cout * 4;

    // Loads of a K step. Every offset is SCALAR (lane = channel): a pixel outside the image (or a tile past the end) is read at clamped
    // coordinates — always in bounds — and selected to zero afterwards by wave-uniform bits. Offsets are separable: byte offset of patch
    // pixel (i, j) = row part [i] (image base included) + column part [j]. prep_step computes the scalars of a step, issue_load<n> issues its
    // n-th load (0..39: tile n / 20, patch pixel n % 20 or dy pixel n % 20 - 16): a burst of 40 one-dword loads per wave blocks at the
    // texture addresser's queue (4 waves x 40 x 256 B: ~0.9 us per step, measured), spread over the positions of a step it is free.
    // raw operands of this wave, two register sets — 0: the step being transformed, 1: the step in flight —: 2 tiles x (16 patch pixels of x,
    // 4 pixels of dy) for the lane's channel + wave-uniform validity bits. Indexed by compile-time constants only.
    // (one f32x2 per pixel: component = the wave's tile i2 — the two tiles of a wave go through the same arithmetic, so every add / fma of
    // the transforms is ONE packed instruction for both: fp32 MFMAs hide no vector work, profiles/r06_mfma_valu_coexec.txt)
    f32x2 rx_[2][16], rd_[2][4];
    unsigned okx_[2][2], okd_[2][2];
    uint32_t srb[2][4], scb[2][4], sdr[2][2], sdc[2][2];
    auto prep_step = [&](int st, auto SETC) {
        constexpr int set = decltype(SETC)::value;
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            const int tl = 4 * kkp + 2 * i2 + lh;
            const int t_raw = __builtin_amdgcn_readfirstlane(t_first + st * 8 + tl);
            const bool live = t_raw < p.T;
            const uint32_t t = (uint32_t)(live ? t_raw : p.T - 1);
            const uint32_t b = fdiv(t, p.dTHW), rem = t - b * p.dTHW.d, th = fdiv(rem, p.dTW), tw = rem - th * p.dTW.d;
            const int y0 = 2 * (int)th - 1, x0 = 2 * (int)tw - 1;
            unsigned rok = 0, cok = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int yy = y0 + i, xx = x0 + i;
                rok |= ((unsigned)yy < (unsigned)p.H) ? (1u << i) : 0u;
                cok |= ((unsigned)xx < (unsigned)p.W) ? (1u << i) : 0u;
                srb[i2][i] = (uint32_t)((((int)b * p.H + min(max(yy, 0), p.H - 1)) * p.W) * cin4);
                scb[i2][i] = (uint32_t)(min(max(xx, 0), p.W - 1) * cin4);
            }
            if (!live) rok = 0;
            unsigned ok = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) ok |= ((rok >> i) & 1u) ? (cok << (4 * i)) : 0u;
            okx_[set][i2] = ok;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                sdr[i2][i] = (uint32_t)(((int)b * p.H + min(y0 + 1 + i, p.H - 1)) * p.W) * (uint32_t)cout4;
                sdc[i2][i] = (uint32_t)min(x0 + 1 + i, p.W - 1) * (uint32_t)cout4;
            }
            okd_[set][i2] = (((rok >> 1) & 1u) ? ((cok >> 1) & 3u) : 0u) | (((rok >> 2) & 1u) ? (((cok >> 1) & 3u) << 2) : 0u);
        }
    };
    auto issue_load = [&](auto NC, auto SETC) {
        constexpr int n = decltype(NC)::value, set = decltype(SETC)::value, i2 = n / 20, k = n % 20;
        if constexpr (k < 16)
            rx_[set][k][i2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, xlane, srb[i2][k / 4] + scb[i2][k % 4], 0));
        else
            rd_[set][k - 16][i2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, dlane, sdr[i2][(k - 16) / 2] + sdc[i2][(k - 16) % 2], 0));
    };
    auto load_step = [&](int st, auto SETC) {       // the whole step at once (prologue)
        prep_step(st, SETC);
        auto one = [&](auto NC) { issue_load(NC, SETC); };
        for_each_const(one, std::make_integer_sequence<int, 40>{});
    };
    // LDS images. operand 0 = Yt (rows = couts), 1 = V (rows = cins); float index ((((buf * 2 + op) * 16 + pos) * 2 + h) * 64 + ch) * 4 + kk
    auto img = [&](int buf, int op, int pos, int h, int ch) { return sL + ((((buf * 2 + op) * 16 + pos) * 2 + h) * 64 + ch) * 4; };
    // activation of a tile's patch (padding / dead tiles selected to zero after it) and zeroing of dy pixels outside the image
    auto act = [&](auto SETC) {                // both tiles: one packed fma per pixel, max and the (wave-uniform) padding select per tile
        constexpr int set = decltype(SETC)::value;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            f32x2 v = rx_[set][k];
            if constexpr (XF) {
                // v = v * scale + shift for both tiles: scale = the LOW half of the pair scsh for both lanes, shift = its HIGH half
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "+v"(v) : "v"(scsh));
                // (asm: fmaxf on a value the compiler did not produce itself gets a canonicalising v_max in front of the real one)
                asm volatile("v_max_f32 %0, 0, %0" : "+v"(v[0]));
                asm volatile("v_max_f32 %0, 0, %0" : "+v"(v[1]));
            }
            v[0] = ((okx_[set][0] >> k) & 1u) ? v[0] : 0.f;
            v[1] = ((okx_[set][1] >> k) & 1u) ? v[1] : 0.f;
            rx_[set][k] = v;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            rd_[set][k][0] = ((okd_[set][0] >> k) & 1u) ? rd_[set][k][0] : 0.f;
            rd_[set][k][1] = ((okd_[set][1] >> k) & 1u) ? rd_[set][k][1] : 0.f;
        }
    };
    auto col_x = [&](auto SETC) {      // B^T d: columns, both tiles
        constexpr int set = decltype(SETC)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x2 d0 = rx_[set][j], d1 = rx_[set][4 + j], d2 = rx_[set][8 + j], d3 = rx_[set][12 + j];
            rx_[set][j] = pk_sub(d0, d2); rx_[set][4 + j] = pk_add(d1, d2); rx_[set][8 + j] = pk_sub(d2, d1); rx_[set][12 + j] = pk_sub(d1, d3);
        }
    };
    // (B^T d B) row i of both tiles -> V image; (A dY A^T) row i of both tiles -> Yt image. One 8-byte store per position and operand.
    auto row_store = [&](auto SETC, int i, int buf) {
        constexpr int set = decltype(SETC)::value;
        const f32x2 t0 = rx_[set][i * 4], t1 = rx_[set][i * 4 + 1], t2 = rx_[set][i * 4 + 2], t3 = rx_[set][i * 4 + 3];
        const f32x2 v[4] = {pk_sub(t0, t2), pk_add(t1, t2), pk_sub(t2, t1), pk_sub(t1, t3)};
        // A = [[1,0],[1,1],[1,-1],[0,-1]]: rows of A dY:  r0 = dy0., r1 = dy0. + dy1., r2 = dy0. - dy1., r3 = -dy1.
        const f32x2 a = rd_[set][0], b = rd_[set][1], c = rd_[set][2], d = rd_[set][3];
        const f32x2 e0 = i == 0 ? a : i == 1 ? pk_add(a, c) : i == 2 ? pk_sub(a, c) : pk_neg(c);
        const f32x2 e1 = i == 0 ? b : i == 1 ? pk_add(b, d) : i == 2 ? pk_sub(b, d) : pk_neg(d);
        const f32x2 y[4] = {e0, pk_add(e0, e1), pk_sub(e0, e1), pk_neg(e1)};
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            *reinterpret_cast<f32x2*>(img(buf, 1, i * 4 + nu, lh, lane) + 2 * kkp) = v[nu];
            *reinterpret_cast<f32x2*>(img(buf, 0, i * 4 + nu, lh, lane) + 2 * kkp) = y[nu];
        }
    };

    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // prologue: step 0 transformed into buffer 0, step 1 in flight. The loads run TWO steps ahead of the MFMAs (a step is 16 x 4 MFMAs =
    // 1.7 us: one step of lead does not cover an HBM round trip when a lone wave per SIMD has nothing else to run), in two register sets.
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    load_step(0, S0{});
    load_step(p.steps > 1 ? 1 : 0, S1{});
    act(S0{}); col_x(S0{});
#pragma unroll
    for (int i = 0; i < 4; ++i) row_store(S0{}, i, 0);
    __syncthreads();

    const int hh = lane >> 5, l31 = lane & 31;
    // One K step st: MFMAs on LDS buffer st & 1 while the operands of step st + 1 are transformed into the other buffer (behind positions
    // TR0..15) and the loads of step st + 2 are in flight. The loads need TWO steps of lead: a step is 16 x 4 MFMAs = 1.7 us, every byte of
    // the 64-channel layers comes from HBM exactly once (~2 us under load), and a lone wave per SIMD has nothing else to run — with one
    // step of lead the kernel waited 0.9 us per step (ablation: 190 -> 146 us without the loads). Two register sets that swap roles would
    // need the loop unrolled by two (the register allocator then shuffles the 256 accumulators between the copies: 170 spills), so set 1
    // (in flight, a full step old) is COPIED into set 0 at the top of a step — 44 moves — and re-loaded.
    constexpr int TR0 = 10;
    for (int st = 0; st < p.steps; ++st) {
        const int buf = st & 1;
        {   // set 1 holds step st + 1 (loaded a full step ago): into set 0
#pragma unroll
            for (int k = 0; k < 16; ++k) rx_[0][k] = rx_[1][k];
#pragma unroll
            for (int k = 0; k < 4; ++k) rd_[0][k] = rd_[1][k];
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) { okx_[0][i2] = okx_[1][i2]; okd_[0][i2] = okd_[1][i2]; }
        }
#if !(OSI_WABL & 1)
        prep_step(st + 2 < p.steps ? st + 2 : p.steps - 1, S1{});   // past the end: a valid step re-loaded (never used): no branch around loads
#endif
        const float* ra = img(buf, 0, 0, hh, 32 * wm + l31);
        const float* rb = img(buf, 1, 0, hh, 32 * wn + l31);
        f32x4 a0 = *reinterpret_cast<const f32x4*>(ra), b0 = *reinterpret_cast<const f32x4*>(rb);
        __builtin_amdgcn_sched_barrier(0);
        auto position = [&](auto POSC) {
            constexpr int pos = decltype(POSC)::value;
            f32x4 na = a0, nb = b0;
            if (pos < 15) {
                na = *reinterpret_cast<const f32x4*>(ra + (pos + 1) * (2 * 64 * 4));
                nb = *reinterpret_cast<const f32x4*>(rb + (pos + 1) * (2 * 64 * 4));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[pos] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[pos], 0, 0, 0);
#if !(OSI_WABL & 1)
            {   // this position's share of the step-after-next's 40 loads
                constexpr int n0 = pos * 40 / 16, n1 = (pos + 1) * 40 / 16;
                if constexpr (n1 > n0) issue_load(std::integral_constant<int, n0>{}, S1{});
                if constexpr (n1 > n0 + 1) issue_load(std::integral_constant<int, n0 + 1>{}, S1{});
                if constexpr (n1 > n0 + 2) issue_load(std::integral_constant<int, n0 + 2>{}, S1{});
            }
#endif
#if !(OSI_WABL & 2)
            if constexpr (pos == TR0) act(S0{});
            if constexpr (pos == TR0 + 1) col_x(S0{});
            if constexpr (pos >= 12) row_store(S0{}, pos - 12, buf ^ 1);
#endif
            a0 = na; b0 = nb;
            __builtin_amdgcn_sched_barrier(0);
        };
        for_each_const(position, std::make_integer_sequence<int, 16>{});
        __syncthreads();
    }

    // ---- epilogue: dW = G^T S G per (cout, cin), written as a partial in [9][64 cout][64 cin] order (lane = cin: 128-byte runs) ---------
    // The lane's coordinates are taken afresh here (lane id from mbcnt, wave role from the scalar copies): kept from the top of the kernel
    // they would have to live through the K loop, whose 256 + 256 registers are all taken (two spills otherwise).
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int hh_e = lane_e >> 5, l31_e = lane_e & 31;      // (kkp = wave >> 1 = wm, lh = wave & 1 = wn: wave-uniform scalars)
    float* out = p.slab + (size_t)item * (9 * 4096) + (32 * lh + l31_e);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = 32 * kkp + (r & 3) + 8 * (r >> 2) + 4 * hh_e;       // cout row of accumulator register r
        float t[3][4];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            const float s0 = acc[nu][r], s1 = acc[4 + nu][r], s2 = acc[8 + nu][r], s3 = acc[12 + nu][r];
            t[0][nu] = s0 + 0.5f * (s1 + s2);
            t[1][nu] = 0.5f * (s1 - s2);
            t[2][nu] = 0.5f * (s1 + s2) + s3;
        }
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            out[(size_t)((rr * 3 + 0) * 64 + m) * 64] = t[rr][0] + 0.5f * (t[rr][1] + t[rr][2]);
            out[(size_t)((rr * 3 + 1) * 64 + m) * 64] = 0.5f * (t[rr][1] - t[rr][2]);
            out[(size_t)((rr * 3 + 2) * 64 + m) * 64] = 0.5f * (t[rr][1] + t[rr][2]) + t[rr][3];
        }
    }
}

// dW[cout][r][s][cin] = sum over the S partials of a block, in a FIXED order (bitwise reproducible): SG split groups per output (a
// power of two <= 16: 256 partials of the 64-channel layers would otherwise be one thread's serial chain on 36 workgroups), group g sums
// the partials s = g, g + SG, ..., the groups are added in group order through LDS. One thread per (4 consecutive cins, group).
__global__ __launch_bounds__(256) void k_wino_wgrad_reduce(const float* __restrict__ slab, float* __restrict__ dw, int Cin, int Cout, int CBn, int S,
                                                           int SG) {
    __shared__ f32x4 part[256];
    const int per = 256 / SG;                                  // outputs (of 4 floats) per workgroup
    const int o = threadIdx.x % per, g = threadIdx.x / per;
    const int idx = blockIdx.x * per + o;                      // (cout, tap, cin / 4)
    const int C4 = Cin / 4, n = Cout * 9 * C4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    int k = 0, tap = 0, c4 = 0;
    if (idx < n) {
        c4 = idx % C4; tap = (idx / C4) % 9; k = idx / (C4 * 9);
        const int cb = (c4 * 4) / 64, kb = k / 64;
        const float* src = slab + (size_t)((kb * CBn + cb) * S) * (9 * 4096) + (size_t)(tap * 64 + (k & 63)) * 64 + ((c4 * 4) & 63);
        for (int s = g; s < S; s += SG) a += ld4(src + (size_t)s * (9 * 4096));
    }
    part[threadIdx.x] = a;
    __syncthreads();
    if (g == 0 && idx < n) {
        for (int j = 1; j < SG; ++j) a += part[j * per + o];
        *reinterpret_cast<f32x4*>(dw + ((size_t)k * 9 + tap) * Cin + c4 * 4) = a;
    }
}

int backward_exclusive_cus();
struct WgradPlan { int S, steps; size_t slab_floats; };
WgradPlan plan_wgrad_wino(const osi_conv_desc* d) {
    const int T = d->B * ((d->H + 1) / 2) * ((d->W + 1) / 2);
    const int blocks = (d->Cout / 64) * (d->Cin / 64);
    int G = backward_exclusive_cus();      // blocks x splits = the CUs this backward-pass kernel may count on (see there)
    int S = G / blocks;
    if (S < 1) S = 1;
    const int max_s = (T + 63) / 64;                 // at least 8 K steps per split
    if (S > max_s) S = max_s;
    const int per = (T + S - 1) / S;
    WgradPlan w;
    w.S = S; w.steps = (per + 7) / 8;
    w.slab_floats = (size_t)blocks * S * 9 * 4096;
    return w;
}

bool wino_shape(const osi_conv_desc* d) {
    return conv_desc_ok(d) && d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1 && d->H == d->Ho && d->W == d->Wo && d->H >= 2 && d->W >= 2;
}

struct Geo { int TH, TW, T, MT, P; bool odd; float cnt; bool uniform; };
Geo geo_of(const osi_conv_desc* d) {
    Geo g{};
    g.TH = (d->H + 1) / 2; g.TW = (d->W + 1) / 2; g.T = d->B * g.TH * g.TW; g.MT = (g.T + 63) / 64; g.P = (g.T + 15) / 16;
    const bool even = d->H % 2 == 0 && d->W % 2 == 0;
    g.odd = !even || g.T % 64 != 0;
    // every 16-tile statistics group holds the same number of valid pixels: all groups full and either all tiles inside the image or a
    // group = whole images; or there is only one group (then it holds every pixel)
    const int thw = g.TH * g.TW;
    if (g.P == 1) { g.uniform = true; g.cnt = (float)((long)d->B * d->H * d->W); }
    else if (g.T % 16 == 0 && even) { g.uniform = true; g.cnt = 64.f; }
    else if (g.T % 16 == 0 && 16 % thw == 0) { g.uniform = true; g.cnt = (float)(16 / thw * d->H * d->W); }
    else { g.uniform = false; g.cnt = 64.f; }
    return g;
}

template <int FLIP>
int launch_weights(const float* w, float* u, int Kc, int Nc, hipStream_t st) {
    hipLaunchKernelGGL(k_wino_weights<FLIP>, dim3((unsigned)((Kc / 4 * Nc + 255) / 256)), dim3(256), 0, st, w, u, Kc, Nc, Kc / KC, Nc / 32);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

// the 32-tile x 128-channel unit wherever the output-channel count allows (knob wino_wide)
bool wide_units(int Nc) { return g_osi_tuning.wino_wide && Nc % 128 == 0; }

WinoP make_wp(const osi_conv_desc* d, const Geo& g, int Kc, int Nc) {
    WinoP p{};
    const bool wide = wide_units(Nc);
    p.H = d->H; p.W = d->W; p.Kc = Kc; p.Nc = Nc; p.TH = g.TH; p.TW = g.TW; p.T = g.T; p.KS = Kc / KC; p.CB = Nc / 32;
    p.MT = wide ? (g.T + 31) / 32 : g.MT; p.NT = Nc / (wide ? 128 : 64);
    p.x_bytes = (int)((size_t)d->B * d->H * d->W * Kc * 4);
    p.y_bytes = (int)((size_t)d->B * d->H * d->W * Nc * 4);
    p.u_bytes = (int)((size_t)16 * Kc * Nc * 4);
    p.P = g.P; p.cnt = g.cnt; p.rcnt = 1.f / g.cnt;
    return p;
}

// CUs the Winograd kernels of the BACKWARD pass may count on while data-parallel collectives are in flight. These kernels are one
// 512-register workgroup per CU: a CU that holds ONE resident RCCL channel workgroup is lost to them entirely (the direct kernels, 8
// waves per SIMD, lose an eighth of it — what dp_reserved_cus = ceil(channels / 8) was sized for), and a workgroup that finds no CU runs in
// a second round. So they leave the CHANNEL count free (8 x dp_reserved_cus, at most a quarter of the chip). The forward pass has no
// collective beside it and keeps the fwd / dgrad plans' figure. Measured at world size 1 (profiles/r05_ab_dp_reserved.txt): 32 CUs less
// cost the overlapped step 0.0 - 0.25 ms.
int backward_exclusive_cus() {
    if (g_osi_tuning.tail_cus > 0) return chip_cus();
    int res = 8 * g_osi_tuning.dp_reserved_cus;
    if (res > hw_cus() / 4) res = hw_cus() / 4;
    const int n = hw_cus() - res;
    return n < 8 ? 8 : n;
}
// persistent grid: one workgroup per CU the launch may use, a multiple of 8 (a workgroup's units keep its blockIdx % 8 = its XCD)
// Never more than the hardware's CUs: the knob tail_cus may name a larger chip (it is a plan parameter of the direct kernels' tail split,
// range-checked to 0 .. 4096 only), but a 512-register workgroup beyond the CU count would only queue behind the others — and the
// stream-K slab (slab_bytes_of) is sized for the hardware's CUs: workgroup w writes slots 2 w and 2 w + 1.
int wino_grid(bool backward) {
    int cus = backward ? backward_exclusive_cus() : chip_cus();
    if (cus > hw_cus()) cus = hw_cus();
    cus = cus / 8 * 8;
    return cus < 8 ? 8 : cus;
}
constexpr size_t SLOT_BYTES = 65536;      // one partial tile: 256 pixels x 64 channels, fp32
size_t u_bytes_of(const osi_conv_desc* d) { return ((size_t)16 * d->Cin * d->Cout * sizeof(float) + 255) / 256 * 256; }
size_t slab_bytes_of() { return (size_t)(hw_cus() / 8 * 8 < 8 ? 8 : hw_cus() / 8 * 8) * 2 * SLOT_BYTES; }

// q full rounds + r remainder units over G workgroups; slab behind the transformed weights
void plan_units(WinoP& p, void* slab, size_t slab_bytes, const osi_conv_desc* d, int G, bool input_gradient) {
    const int V = p.MT * p.NT;
    p.q = V / G; p.r = V - p.q * G; p.nfull = p.q * G;
    // short units (KS <= 4: the 64-channel layers) keep their ragged last round: a piece of one or two slices plus the fix-up pass costs
    // what the balance returns (measured: 205 vs 205 us forward, 232 vs 228 us input gradient at 56 x 56)
    // — unless the ragged round is a large part of a short launch (small batches: B = 64 leaves 16 units for a 4th round of 3.06)
    const bool heavy_tail = p.r > 0 && (double)(G - p.r) / (double)G / (double)(p.q + 1) >= 0.15;
    // knob wino_streamk: 0 = never, 1 = both directions, 2 = forward only, 3 = input gradient only (in the backward pass the weight
    // gradients of the side stream fill the CUs a ragged round leaves idle; the forward has nothing beside it)
    const int sk = g_osi_tuning.wino_streamk;
    const bool cut = sk == 1 || (sk == 2 && !input_gradient) || (sk == 3 && input_gradient);
    if (!cut || (p.KS <= 4 && !heavy_tail)) { p.nfull = V; p.r = 0; }
    p.slab = (float*)slab;
    // the buffer range of the slab descriptor is what the caller really passed (callers require slab_bytes >= G * 2 * SLOT_BYTES): a store
    // past it is dropped by the range check instead of landing in the neighbouring workspace
    p.slab_bytes = (int)std::min<size_t>(slab_bytes, (size_t)hw_cus() * 2 * SLOT_BYTES);
}

}  // namespace

extern "C" {

/* 1 when the Winograd form takes this convolution (3x3 / stride 1 / pad 1, Cin % 16 == 0 and Cout % 64 == 0 forward — the roles swap for
 * the input gradient —, every BatchNorm statistics group of 16 tiles holding the same number of pixels). */
int osi_conv_wino_eligible(const osi_conv_desc* d, int input_gradient) {
    if (!wino_shape(d)) return 0;
    const int Kc = input_gradient ? d->Cout : d->Cin, Nc = input_gradient ? d->Cin : d->Cout;
    if (Kc % KC || Nc % 64) return 0;
    const Geo g = geo_of(d);
    return (input_gradient || g.uniform) ? 1 : 0;
}

/* bytes both forms need: the transformed weights (16 positions x Cin x Cout floats) + the slab of the stream-K remainder (two 64 KiB
 * partial tiles per workgroup); 0 = not eligible */
size_t osi_conv_wino_workspace(const osi_conv_desc* d) {
    if (!wino_shape(d)) return 0;
    return u_bytes_of(d) + slab_bytes_of();
}

/* Winograd twin of osi_conv_fwd_act / osi_conv_fwd_bnstats: in_scale / in_shift may be NULL (plain input). P = ceil(tiles / 16),
 * rows_per_block = valid pixels of a 16-tile group (64 for even H, W). pstats needs 2 * P * Cout floats. */
// w != NULL: transformed into `u` first; w == NULL: `u` already holds this convolution's transformed weights (osi_conv_wino_transform_weights)
static int fwd_wino_impl(const osi_conv_desc* d, const float* x, const float* in_scale, const float* in_shift, const float* w, float* u,
                         float* y, void* slab, size_t slab_bytes, float* pstats, size_t pstats_bytes, int* P, int* rows_per_block,
                         osi_stream_t stream, const osi_conv_epilogue* epi = nullptr) {
    OSI_REQUIRE(x && u && y && slab && osi_conv_wino_eligible(d, 0));
    OSI_REQUIRE(!epi || (epi->scale && epi->shift && !epi->residual && !in_scale && !pstats));
    OSI_REQUIRE((in_scale == nullptr) == (in_shift == nullptr));
    OSI_REQUIRE(slab_bytes >= slab_bytes_of());
    OSI_REQUIRE(!pstats || (P && rows_per_block));
    hipStream_t st = (hipStream_t)stream;
    const Geo g = geo_of(d);
    WinoP p = make_wp(d, g, d->Cin, d->Cout);
    p.x = x; p.u = u; p.y = y; p.sc = in_scale; p.sh = in_shift;
    if (epi) { p.osc = epi->scale; p.osh = epi->shift; p.orelu = epi->relu ? 1 : 0; }
    if (pstats) {
        OSI_REQUIRE(pstats_bytes >= (size_t)2 * g.P * d->Cout * sizeof(float));
        p.pmean = pstats; p.pm2 = pstats + (size_t)g.P * d->Cout;
        *P = g.P; *rows_per_block = (int)g.cnt;
    }
    if (w) { if (int e = launch_weights<0>(w, u, d->Cin, d->Cout, st)) return e; }
    const int G = wino_grid(false);
    OSI_REQUIRE(slab_bytes >= (size_t)G * 2 * SLOT_BYTES);
    plan_units(p, slab, slab_bytes, d, G, false);
    const dim3 grid((unsigned)G), blk(256);
    const bool wide = wide_units(d->Cout), odd = g.odd || (wide && g.T % 32 != 0);
    auto launch = [&](auto XFC, auto ODDC, auto WIDEC) {
        constexpr bool xf = decltype(XFC)::value, od = decltype(ODDC)::value, wd = decltype(WIDEC)::value;
        hipLaunchKernelGGL((k_wino<xf, 0, od, wd>), grid, blk, 0, st, p);
        if (hipGetLastError() != hipSuccess) return OSI_ERR_LAUNCH;
        if (p.r > 0) {
            hipLaunchKernelGGL((k_wino_fixup<0, od, wd>), dim3((unsigned)p.r * 16), dim3(64), 0, st, p, G);
            if (hipGetLastError() != hipSuccess) return OSI_ERR_LAUNCH;
        }
        return OSI_OK;
    };
    auto launch_epi = [&](auto ODDC, auto WIDEC) {       // inference epilogue: plain input, own BatchNorm + ReLU on the output
        constexpr bool od = decltype(ODDC)::value, wd = decltype(WIDEC)::value;
        hipLaunchKernelGGL((k_wino<false, 2, od, wd>), grid, blk, 0, st, p);
        if (hipGetLastError() != hipSuccess) return OSI_ERR_LAUNCH;
        if (p.r > 0) {
            hipLaunchKernelGGL((k_wino_fixup<2, od, wd>), dim3((unsigned)p.r * 16), dim3(64), 0, st, p, G);
            if (hipGetLastError() != hipSuccess) return OSI_ERR_LAUNCH;
        }
        return OSI_OK;
    };
    using T_ = std::true_type; using F_ = std::false_type;
    if (epi) {
        if (odd) return wide ? launch_epi(T_{}, T_{}) : launch_epi(T_{}, F_{});
        return wide ? launch_epi(F_{}, T_{}) : launch_epi(F_{}, F_{});
    }
    const int sel = (in_scale ? 4 : 0) | (odd ? 2 : 0) | (wide ? 1 : 0);
    switch (sel) {
        case 0: return launch(F_{}, F_{}, F_{});
        case 1: return launch(F_{}, F_{}, T_{});
        case 2: return launch(F_{}, T_{}, F_{});
        case 3: return launch(F_{}, T_{}, T_{});
        case 4: return launch(T_{}, F_{}, F_{});
        case 5: return launch(T_{}, F_{}, T_{});
        case 6: return launch(T_{}, T_{}, F_{});
        default: return launch(T_{}, T_{}, T_{});
    }
}

int osi_conv_fwd_wino(const osi_conv_desc* d, const float* x, const float* in_scale, const float* in_shift, const float* w, float* y,
                      void* ws, size_t ws_bytes, float* pstats, size_t pstats_bytes, int* P, int* rows_per_block, osi_stream_t stream) {
    OSI_REQUIRE(w && ws && wino_shape(d) && ws_bytes >= osi_conv_wino_workspace(d));
    return fwd_wino_impl(d, x, in_scale, in_shift, w, (float*)ws, y, (char*)ws + u_bytes_of(d), ws_bytes - u_bytes_of(d), pstats, pstats_bytes, P,
                         rows_per_block, stream);
}

/* The weights of one convolution transformed ahead of its launches (they are the same for the forward and the backward pass of a step,
 * and for every forward until the optimizer runs): u = osi_conv_wino_weights_bytes(d) bytes, input_gradient selects the flipped /
 * transposed form osi_conv_dgrad_fused_wino_pre reads. The `_pre` calls take such a buffer instead of the raw weights, plus the shared
 * stream-K slab (osi_conv_wino_slab_bytes() bytes, any convolution). The executor transforms all its 3x3 layers on the side stream at the
 * start of a forward pass, off the critical path. */
size_t osi_conv_wino_weights_bytes(const osi_conv_desc* d) { return wino_shape(d) ? u_bytes_of(d) : 0; }
size_t osi_conv_wino_slab_bytes(void) { return slab_bytes_of(); }
int osi_conv_wino_transform_weights(const osi_conv_desc* d, const float* w, int input_gradient, float* u, size_t u_bytes, osi_stream_t stream) {
    OSI_REQUIRE(w && u && osi_conv_wino_eligible(d, input_gradient ? 1 : 0) && u_bytes >= u_bytes_of(d));
    return input_gradient ? launch_weights<1>(w, u, d->Cout, d->Cin, (hipStream_t)stream) : launch_weights<0>(w, u, d->Cin, d->Cout, (hipStream_t)stream);
}
int osi_conv_fwd_wino_pre(const osi_conv_desc* d, const float* x, const float* in_scale, const float* in_shift, const float* u, float* y,
                          void* slab, size_t slab_bytes, float* pstats, size_t pstats_bytes, int* P, int* rows_per_block, osi_stream_t stream) {
    return fwd_wino_impl(d, x, in_scale, in_shift, nullptr, const_cast<float*>(u), y, slab, slab_bytes, pstats, pstats_bytes, P, rows_per_block, stream);
}

/* Winograd twin of osi_conv_fwd_epilogue (no shortcut: conv2 of a bottleneck has none): out = [relu](conv2d(x, w) * scale[n] + shift[n]),
 * weights already transformed (osi_conv_wino_transform_weights). Eligible shapes as the forward form (osi_conv_wino_eligible(d, 0)). */
int osi_conv_fwd_wino_epilogue_pre(const osi_conv_desc* d, const float* x, const float* u, float* out, const osi_conv_epilogue* e, void* slab,
                                   size_t slab_bytes, osi_stream_t stream) {
    OSI_REQUIRE(e);
    return fwd_wino_impl(d, x, nullptr, nullptr, nullptr, const_cast<float*>(u), out, slab, slab_bytes, nullptr, 0, nullptr, nullptr, stream, e);
}

/* Winograd twin of osi_conv_dgrad_fused for the executor's "in-block" fusion only: no addend, no stored bitmask, one consumer, no pool
 * mode; gate recomputed from f->y0 (scale0 / shift0 required), partial sums optional. P = ceil(tiles / 16) groups of <= 64 pixels. */
static int dgrad_wino_impl(const osi_conv_desc* d, const float* dy, const float* w, float* u, float* dx, const osi_dgrad_fusion* f, void* slab,
                           size_t slab_bytes, int* P, osi_stream_t stream) {
    OSI_REQUIRE(dy && u && dx && slab && f && P && osi_conv_wino_eligible(d, 1));
    OSI_REQUIRE(!f->relu_mask && !f->y1 && !f->pool_idx && f->addend_stride != 2);
    OSI_REQUIRE(f->y0 && f->scale0 && f->shift0);
    OSI_REQUIRE(!f->partials || (f->mean0 && f->invstd0));
    OSI_REQUIRE(slab_bytes >= slab_bytes_of());
    hipStream_t st = (hipStream_t)stream;
    const Geo g = geo_of(d);
    WinoP p = make_wp(d, g, d->Cout, d->Cin);
    p.x = dy; p.u = u; p.y = dx;
    p.ey0 = f->y0; p.escale0 = f->scale0; p.eshift0 = f->shift0;
    // without partial sums the epilogue still evaluates xhat: any readable per-channel vectors do
    p.emean0 = f->mean0 ? f->mean0 : f->scale0; p.einv0 = f->invstd0 ? f->invstd0 : f->scale0;
    if (f->partials) {
        OSI_REQUIRE(f->partials_bytes >= (size_t)3 * g.P * d->Cin * sizeof(float));
        p.esum = f->partials;
    }
    *P = g.P;
    if (w) { if (int e = launch_weights<1>(w, u, d->Cout, d->Cin, st)) return e; }
    const int G = wino_grid(true);
    OSI_REQUIRE(slab_bytes >= (size_t)G * 2 * SLOT_BYTES);
    plan_units(p, slab, slab_bytes, d, G, true);
    const dim3 grid((unsigned)G), blk(256);
    const bool wide = wide_units(d->Cin), odd = g.odd || (wide && g.T % 32 != 0);
    auto launch = [&](auto ODDC, auto WIDEC) {
        constexpr bool od = decltype(ODDC)::value, wd = decltype(WIDEC)::value;
        hipLaunchKernelGGL((k_wino<false, 1, od, wd>), grid, blk, 0, st, p);
        if (hipGetLastError() != hipSuccess) return OSI_ERR_LAUNCH;
        if (p.r > 0) {
            hipLaunchKernelGGL((k_wino_fixup<1, od, wd>), dim3((unsigned)p.r * 16), dim3(64), 0, st, p, G);
            if (hipGetLastError() != hipSuccess) return OSI_ERR_LAUNCH;
        }
        return OSI_OK;
    };
    using T_ = std::true_type; using F_ = std::false_type;
    if (odd) return wide ? launch(T_{}, T_{}) : launch(T_{}, F_{});
    return wide ? launch(F_{}, T_{}) : launch(F_{}, F_{});
}

int osi_conv_dgrad_fused_wino(const osi_conv_desc* d, const float* dy, const float* w, float* dx, const osi_dgrad_fusion* f, void* ws,
                              size_t ws_bytes, int* P, osi_stream_t stream) {
    OSI_REQUIRE(w && ws && wino_shape(d) && ws_bytes >= osi_conv_wino_workspace(d));
    return dgrad_wino_impl(d, dy, w, (float*)ws, dx, f, (char*)ws + u_bytes_of(d), ws_bytes - u_bytes_of(d), P, stream);
}
int osi_conv_dgrad_fused_wino_pre(const osi_conv_desc* d, const float* dy, const float* u, float* dx, const osi_dgrad_fusion* f, void* slab,
                                  size_t slab_bytes, int* P, osi_stream_t stream) {
    return dgrad_wino_impl(d, dy, nullptr, const_cast<float*>(u), dx, f, slab, slab_bytes, P, stream);
}

/* Winograd F(3x3, 2x2) twin of osi_conv_wgrad / osi_conv_wgrad_act (in_scale / in_shift NULL: plain input) for 3x3 / stride 1 / pad 1
 * convolutions with Cin % 64 == 0 and Cout % 64 == 0: 2.25x fewer multiplies, deterministic (split-K over the tile axis into partial
 * slabs, fixed-order reduce). ws: osi_conv_wgrad_wino_workspace(d) bytes (0 = shape not taken). */
size_t osi_conv_wgrad_wino_workspace(const osi_conv_desc* d) {
    if (!wino_shape(d) || d->Cin % 64 || d->Cout % 64) return 0;
    return plan_wgrad_wino(d).slab_floats * sizeof(float);
}

int osi_conv_wgrad_wino(const osi_conv_desc* d, const float* dy, const float* x, const float* in_scale, const float* in_shift, float* dw,
                        void* ws, size_t ws_bytes, osi_stream_t stream) {
    OSI_REQUIRE(dy && x && dw && ws && wino_shape(d) && d->Cin % 64 == 0 && d->Cout % 64 == 0);
    OSI_REQUIRE((in_scale == nullptr) == (in_shift == nullptr));
    const WgradPlan w = plan_wgrad_wino(d);
    OSI_REQUIRE(ws_bytes >= w.slab_floats * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    WgradP p{};
    p.x = x; p.dy = dy; p.slab = (float*)ws; p.sc = in_scale; p.sh = in_shift;
    p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.TH = (d->H + 1) / 2; p.TW = (d->W + 1) / 2; p.T = d->B * p.TH * p.TW;
    p.CBn = d->Cin / 64; p.S = w.S; p.steps = w.steps;
    p.dTHW = make_fastdiv((uint32_t)(p.TH * p.TW)); p.dTW = make_fastdiv((uint32_t)p.TW);
    p.x_bytes = (int)((size_t)d->B * d->H * d->W * d->Cin * 4);
    p.dy_bytes = (int)((size_t)d->B * d->H * d->W * d->Cout * 4);
    const dim3 grid((unsigned)((d->Cout / 64) * p.CBn * p.S)), blk(256);
    if (in_scale) hipLaunchKernelGGL((k_wino_wgrad<true>), grid, blk, 0, st, p);
    else hipLaunchKernelGGL((k_wino_wgrad<false>), grid, blk, 0, st, p);
    OSI_LAUNCH_CHECK();
    const int n = d->Cout * 9 * (d->Cin / 4);
    int SG = 1;
    while (SG * 2 <= p.S && SG < 16) SG *= 2;
    const int per = 256 / SG;
    hipLaunchKernelGGL(k_wino_wgrad_reduce, dim3((unsigned)((n + per - 1) / per)), blk, 0, st, (const float*)ws, dw, d->Cin, d->Cout, p.CBn, p.S, SG);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

}  // extern "C"
