// Shared device/host helpers for the gfx950 kernels of the open-set ImageNet hot path.
// Everything here is CDNA4-only (wave64, MFMA f32 32x32x2); there is no other backend.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/osi.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define OSI_LAUNCH_CHECK()                                     \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) return OSI_ERR_LAUNCH;          \
    } while (0)

#define OSI_REQUIRE(cond)                                      \
    do {                                                       \
        if (!(cond)) return OSI_ERR_ARG;                       \
    } while (0)

// Process-wide tuning knobs (development A/B switches). Written only by osi_set_tuning() (abi.hip); launch functions read them
// and never touch the environment, so they stay stateless and re-entrant as include/osi.h promises. Defaults = measured optimum.
struct OsiTuning {
    int wgrad_tile;     // 64 forces 64x64 weight-gradient tiles, 0 = 128-wide tiles wherever the channel counts allow
    int wgrad_blocks;   // split-K footprint budget of one weight-gradient launch, in 64x64-workgroup units
    int wgrad_nst;      // LDS stages of the weight-gradient kernel (1 or 2)
    int bn_grid;        // grid cap of the BatchNorm stream kernels
    int bn_single_p;    // BatchNorm statistics: row-tile partials merged by ONE launch up to this count, two-level above it
    int wgrad3;         // 0 = per-tap kernel everywhere, 1 = 3x3 stride-1 weight gradients use the all-taps kernel (k_conv_wgrad3), 2 = stride 2 too
    int wgrad3_blocks;  // workgroups per launch the all-taps kernel's split-K plan aims for
    int fwd_wide, dgrad_wide;  // A/B: 64x128 forward / input-gradient tiles wherever the channel count allows (default 0: measured rule)
    int wgrad_group;    // weight-gradient block -> XCD mapping: 0 plain 2-D grid, 1 the R*S taps of a cell share an XCD, 2 whole K splits do
    int tail_split;     // 1 = forward / input-gradient launches split the tiles of their ragged last round along K (plan_tail_split)
    int tail_cus;       // CU count the tail plan balances for; 0 = ask the device (256 on MI355X)
    int tail_smax;      // most K splits a remainder tile is cut into
    int tail_mint;      // fewest K tiles (of 32) a split keeps
    int stem_direct;    // 1 = the stem convolution runs its direct form (k_stem_fwd_direct) where the geometry allows, 0 = implicit GEMM
    int tail_gain;      // balanced remainder: least modelled gain of a launch, in percent, for its ragged round to be split
    int tail_qmax;      // ... and most full rounds a launch may have
    int bn_grid_bwd;    // grid cap of the BatchNorm BACKWARD apply kernels (they run beside the weight gradients)
    int bn_wide_p;      // BatchNorm finalisation (forward statistics and backward sums): ONE 1024-thread launch up to this many partials
    int fwd_rows;         // fwd: 1x1 stride-1 convolutions with Cin = 64 / 128 on the persistent row walker (k_conv1x1_rows): 0 off, 1 Cin = 64 at >= 8 row tiles per CU, 2 every eligible shape (tests)
    int fwd_w3;           // fwd: 3x3 stride-1 convolutions stage one activation window per tap row (k_conv_fwd W3): 0 off, 1 on
    int dgrad_w3;         // dgrad: the same for the in-block fused 3x3 stride-1 input gradients (k_conv_dgrad W3): 0 off, 1 on
    int fwd_wino;         // fwd: the executor runs its 3x3 stride-1 convolutions in the Winograd F(2x2,3x3) form (conv_wino.hip): 0 off, 1 on
    int dgrad_wino;       // dgrad: the same for the in-block fused 3x3 stride-1 input gradients
    int wgrad_wino;       // wgrad: the executor's 3x3 stride-1 weight gradients (conv2, fused input activation) in the Winograd F(3x3,2x2) form
    int wino_wide;        // Winograd fwd / dgrad: units of 32 tiles x 128 channels where the channel count allows (the patch transform serves 2x the channels)
    int wino_streamk;     // Winograd forms: the units of the ragged last round are cut along K over all workgroups (stream-K): 0 = never, 1 = forward and input gradient, 2 = forward only (default), 3 = input gradient only
    int dp_reserved_cus;  // CUs' worth of wave slots the launch plans leave to co-resident communication kernels (data parallel); 0 = none
};
extern OsiTuning g_osi_tuning;

static inline int osi_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Unsigned division by a runtime-constant divisor: q = (n * mul) >> (32 + sh), exact for n < 2^31.
struct FastDiv {
    uint32_t mul, sh, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    if (d == 1) { f.mul = 0; f.sh = 0; return f; }
    uint32_t l = 0;
    while ((1ull << l) < d) ++l;           // l = ceil(log2 d)
    uint64_t m = ((1ull << (32 + l)) + d - 1) / d;  // ceil(2^(32+l)/d), fits 33 bits
    f.mul = (uint32_t)(m - (1ull << 32));  // low 32 bits; the implicit 2^32 is added back in the divide
    f.sh = l;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
    uint32_t t = __umulhi(n, f.mul);
    // q = (t + n) >> sh without overflow: ((n - t) >> 1) + t, then >> (sh-1). d == 1 (mul = sh = 0) is a select, not an early
    // return: a branch here, uniform as it is, splits every caller's basic block and with it the compiler's load scheduling
    const uint32_t q = (((n - t) >> 1) + t) >> ((f.sh - 1) & 31);
    return f.d == 1 ? n : q;
}

// Chan/Welford merge of two (count, mean, M2) triples: a <- a (+) b. Exact in the counts, no E[x^2]-E[x]^2 cancellation.
__device__ __forceinline__ void chan_merge(float& na, float& ma, float& sa, float nb, float mb, float sb) {
    if (nb == 0.f) return;
    if (na == 0.f) { na = nb; ma = mb; sa = sb; return; }
    const float n = na + nb, d = mb - ma;
    ma += d * (nb / n);
    sa += sb + d * d * (na * nb / n);
    na = n;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
