// Probe (not product code): exact-fp32 MFMA Winograd F(2x2, 3x3) forward convolution on gfx950, for the thirteen stride-1 3x3 layers of
// ResNet-50 (the convolutions under /root/reference/openset_imagenet/model.py:37 — cuDNN takes this route for fp32 3x3).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      per 2x2 output tile, 4x4 input patch d, 3x3 filter g  (Lavin & Gray)
//   -> 16 independent GEMMs  M_p[cout][tile] = sum_cin U_p[cout][cin] * V_p[cin][tile],  p = (xi, nu) in 4x4
//   4 multiplies per output and (cin, cout) pair instead of 9: the matrix pipe does 2.25x less work.
//
// One workgroup = 4 waves (2x2) owns 64 tiles x 64 output channels; each wave owns 32 tiles x 32 channels for ALL 16 positions:
// 16 accumulator sets of v_mfma_f32_32x32x2_f32 = 256 accumulator registers per lane (one wave per SIMD), so the output transform is
// register arithmetic (no cross-wave exchange). K loop over 16-channel slices:
//   V (B operand): every thread owns one (tile, 4-channel quad): 16 buffer loads of 16 B (the 4x4 patch; out-of-image = the range
//     check's zeros), fused input activation relu(fma(x, scale, shift)) (the BatchNorm + ReLU of the producer layer, padding selected to
//     zero AFTER it), B^T d B in registers, 16 ds_write_b128 into V[buf][p][tile][16 ch] (16-byte chunks XOR-swizzled: conflict-free
//     b128 reads and writes without padding — 2 x 64 KB, double buffered, one barrier per slice);
//   U (A operand): transformed ONCE per launch by k_wino_weights into the exact per-lane fragment order, read straight from global / L2
//     into registers (1 KiB contiguous per wave-load), software-pipelined PF positions ahead on the in-order vmcnt queue — no LDS.
// The loads of slice k+1's patches are issued at the top of slice k and their transform is interleaved with the MFMAs of positions 8..15.
//
// Prints per shape: time, effective TFLOP/s (DIRECT-convolution FLOPs / time), the MFMA-side rate, and max |err| against an fp64
// reference on sampled pixels next to the error of a plain fp32 FMA chain (= what the direct MFMA kernel computes) on the same pixels.
//
// build (here) + run (GPU box):
//   hipcc -O3 --offload-arch=gfx950 tools/probes/winograd_f2x2.hip -o tools/probes/bin/winograd_f2x2 && tools/probes/bin/winograd_f2x2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <utility>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e__ = (x);                                                                     \
        if (e__ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); exit(1); } \
    } while (0)

constexpr uint32_t OOB = 0x80000000u;
// -DABL=<bits>: timing ablations (WRONG results): 1 = U fragments always from one cache-resident address, 2 = no patch loads inside the
// K loop, 4 = no patch transform / LDS stores inside the K loop, 8 = no LDS operand reads inside the K loop, 16 = no epilogue
#ifndef ABL
#define ABL 0
#endif
constexpr int KC = 16;      // channels per K slice
constexpr int PF = 6;       // positions the U fragments are loaded ahead (the first fragment issued BEHIND a slice's patch loads is consumed
                            // when the patches are — vmcnt is one in-order queue)

struct WinoP {
    const float* x;      // [B][H][W][Cin]  (pre-activation when XF)
    const float* u;      // transformed weights in fragment order, see k_wino_weights
    float* y;            // [B][H][W][Cout]
    const float* sc;     // [Cin] fused input activation
    const float* sh;
    int B, H, W, Cin, Cout, TH, TW, T, KS, CB, MT, NT;
    int x_bytes, y_bytes, u_bytes;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bld4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void bst4(__amdgpu_buffer_rsrc_t r, f32x4 v, uint32_t voff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, 0, 0);
}

// U_p = (G g G^T)_p, written in the order the MFMA A fragments are consumed:
//   index = ((((p * KS + ks) * CB + cb) * 2 + j) * 64 + lane) * 4 + e   <->   cin = 16 ks + 8 j + 4 (lane / 32) + e,  cout = 32 cb + lane % 32
// (lane half h of MFMA (j, e) of a slice consumes channel 8 j + 4 h + e: the order in which V's channels sit in LDS)
// w: KRSC [Cout][3][3][Cin].  flip = 1: the input-gradient form (taps rotated by 180 degrees, the roles of cin / cout exchanged by the caller)
__global__ void k_wino_weights(const float* __restrict__ w, float* __restrict__ u, int Cin, int Cout, int KS, int CB) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int c = idx % Cin, k = idx / Cin;
    float g[3][3];
    for (int r = 0; r < 3; ++r)
        for (int s = 0; s < 3; ++s) g[r][s] = w[((size_t)(k * 3 + r) * 3 + s) * Cin + c];
    float t[4][3];
    for (int s = 0; s < 3; ++s) {
        t[0][s] = g[0][s];
        t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
        t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
        t[3][s] = g[2][s];
    }
    const int ks = c / KC, j = (c % KC) / 8, h = (c % 8) / 4, e = c % 4, cb = k / 32, lane = h * 32 + k % 32;
    for (int xi = 0; xi < 4; ++xi) {
        float v[4];
        v[0] = t[xi][0];
        v[1] = 0.5f * (t[xi][0] + t[xi][1] + t[xi][2]);
        v[2] = 0.5f * (t[xi][0] - t[xi][1] + t[xi][2]);
        v[3] = t[xi][2];
        for (int nu = 0; nu < 4; ++nu) {
            const int p = xi * 4 + nu;
            u[((((size_t)(p * KS + ks) * CB + cb) * 2 + j) * 64 + lane) * 4 + e] = v[nu];
        }
    }
}

__device__ __forceinline__ bool tile_of_block(int bid, int MT, int NT, int& mt, int& nt) {
    const int xcd = bid & 7, slot = bid >> 3;
    nt = slot % NT;
    mt = (slot / NT) * 8 + xcd;
    return mt < MT;
}

template <class F, int... I>
__device__ __forceinline__ void for_each_position(F& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}

// Persistent form: one workgroup per CU walks units bid, bid + G, ... (unit = 64 tiles x 64 couts; tile_of_block keeps the NT column
// units of a tile block on one XCD). The software pipeline runs ACROSS units: the last slice of a unit stages the first slice of the
// next one (patch loads, transform into the other LDS buffer, U fragments), so only the first unit of a workgroup pays an exposed
// prologue; the epilogue's stores drain behind the next unit's MFMAs.
template <bool XF>
__global__ __launch_bounds__(256, 1) void k_wino_fwd(WinoP p) {
    __shared__ __attribute__((aligned(16))) float sV[2 * 16 * 64 * KC];   // 128 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y, p.y_bytes);
    const int NB = ((p.MT + 7) / 8) * 8 * p.NT, G = gridDim.x;
    const int ltile = tid >> 2, q = tid & 3;
    const int ntile = 32 * wn + (lane & 31), hh = lane >> 5;
    const int THW = p.TH * p.TW;

    int bid = blockIdx.x, mt = 0, nt = 0;
    while (bid < NB && !tile_of_block(bid, p.MT, p.NT, mt, nt)) bid += G;
    if (bid >= NB) return;

    uint32_t off[16];
    unsigned long long okm[16];      // lane masks of the valid patch pixels (v_cndmask's scalar operand)
    uint32_t po[4], po_next[4];      // output byte offsets of this lane's tile (epilogue), current / next unit
    uint32_t ua, ua_next;            // byte offset of this lane's U fragments inside a (position, slice) block: (cout block, lane)
    const __amdgpu_buffer_rsrc_t ru = make_rsrc(p.u, p.u_bytes);
    // everything that depends on the unit: loader offsets + masks, epilogue offsets, U fragment base
    auto setup = [&](int mt_, int nt_, bool live, uint32_t (&po_)[4], uint32_t& ua_) {
        {
            const int t = mt_ * 64 + ltile;
            const int b = t / THW, rem = t - b * THW, th = rem / p.TW, tw = rem - th * p.TW;
            const uint32_t base = (uint32_t)((((b * p.H + 2 * th - 1) * p.W + 2 * tw - 1) * p.Cin + q * 4) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int yy = 2 * th - 1 + i, xx = 2 * tw - 1 + j;
                    const bool ok = live && t < p.T && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
                    off[i * 4 + j] = ok ? base + (uint32_t)((i * p.W + j) * p.Cin * 4) : OOB;
                    okm[i * 4 + j] = __builtin_amdgcn_ballot_w64(ok);
                }
        }
        {
            const int t = mt_ * 64 + ntile;
            const int b = t / THW, rem = t - b * THW, th = rem / p.TW, tw = rem - th * p.TW;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int yy = 2 * th + i, xx = 2 * tw + j;
                    const bool ok = live && t < p.T && yy < p.H && xx < p.W;
                    po_[i * 2 + j] = ok ? (uint32_t)((((b * p.H + yy) * p.W + xx) * p.Cout + 64 * nt_ + 32 * wm + 4 * hh) * 4) : OOB;
                }
        }
        ua_ = (uint32_t)(((2 * nt_ + wm) * 2 * 64 + lane) * 16);                       // + ((pos * KS + ks) * CB) * 2048 + j * 1024 bytes
    };
    setup(mt, nt, true, po, ua);
    float* const wbase = sV + ltile * KC + 4 * (q ^ ((ltile >> 2) & 3));      // + buf * 16384 + pos * 1024
    const float* const rb0 = sV + ntile * KC + 4 * ((0 + hh) ^ ((ntile >> 2) & 3));   // j = 0 chunk
    const float* const rb1 = sV + ntile * KC + 4 * ((2 + hh) ^ ((ntile >> 2) & 3));   // j = 1 chunk
    const uint32_t ustep = (uint32_t)p.CB * 2048;      // bytes per (position, slice) block of U

    f32x16 acc[16];
    f32x4 xr[16];
    auto load_x = [&](int ks) {
#pragma unroll
        for (int k = 0; k < 16; ++k) xr[k] = bld4(rx, off[k], (uint32_t)ks * (KC * 4));
    };
    // activation (+ padding select) and column transform of patch column j. asm: IR passes otherwise regroup these scalar ops (SLP
    // packs them into v_pk_*, sinks the selects to their users) whatever the machine scheduler is told.
    auto act_col = [&](int j, const f32x4& sc4, const f32x4& sh4) {
        if constexpr (XF) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_max_f32 %0, 0, %0\n\tv_cndmask_b32 %0, 0, %0, %3"
                                 : "+v"(xr[i * 4 + j][e]) : "v"(sc4[e]), "v"(sh4[e]), "s"(okm[i * 4 + j]));
        }
        const f32x4 d0 = xr[j], d1 = xr[4 + j], d2 = xr[8 + j], d3 = xr[12 + j];
        xr[j] = d0 - d2; xr[4 + j] = d1 + d2; xr[8 + j] = d2 - d1; xr[12 + j] = d1 - d3;
    };
    // row transform of patch row i and its four LDS stores
    auto row_store = [&](int i, int buf) {
        const f32x4 t0 = xr[i * 4], t1 = xr[i * 4 + 1], t2 = xr[i * 4 + 2], t3 = xr[i * 4 + 3];
        float* w = wbase + buf * (16 * 64 * KC) + (i * 4) * (64 * KC);
        *reinterpret_cast<f32x4*>(w) = t0 - t2;
        *reinterpret_cast<f32x4*>(w + 64 * KC) = t1 + t2;
        *reinterpret_cast<f32x4*>(w + 2 * 64 * KC) = t2 - t1;
        *reinterpret_cast<f32x4*>(w + 3 * 64 * KC) = t1 - t3;
    };
    auto ld_scale = [&](int ks, f32x4& sc4, f32x4& sh4) {
        if constexpr (XF) {
            sc4 = *reinterpret_cast<const f32x4*>(p.sc + ks * KC + 4 * q);
            sh4 = *reinterpret_cast<const f32x4*>(p.sh + ks * KC + 4 * q);
        }
    };

    // prologue (first unit of this workgroup only): slice 0 into buffer 0
    f32x4 sc4 = {1, 1, 1, 1}, sh4 = {0, 0, 0, 0};
    load_x(0);
    ld_scale(0, sc4, sh4);
#pragma unroll
    for (int j = 0; j < 4; ++j) act_col(j, sc4, sh4);
#pragma unroll
    for (int i = 0; i < 4; ++i) row_store(i, 0);

    // U fragment ring: slot (pos % 16), loaded PF positions ahead across slice AND unit boundaries
    f32x4 a[16][2];
    auto load_u = [&](int pos, uint32_t base, int ks) {
        const uint32_t soff = (ABL & 1) ? 0u : (uint32_t)(pos * p.KS + ks) * ustep;      // uniform: the scalar offset of the buffer load
        a[pos][0] = bld4(ru, base, soff);
        a[pos][1] = bld4(ru, base + 1024u, soff);
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) load_u(i, ua, 0);
    __syncthreads();

    int buf = 0;
    for (;;) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        bool more = false;
        for (int ks = 0; ks < p.KS; ++ks) {
            const bool last = ks + 1 == p.KS;
            int ksn = ks + 1;
            uint32_t uan = ua;
            if (last) {      // uniform: the next unit of this workgroup (or a dead one: every offset out of range) — no loads in here
                int nb = bid + G, mt2 = 0, nt2 = 0;
                while (nb < NB && !tile_of_block(nb, p.MT, p.NT, mt2, nt2)) nb += G;
                more = nb < NB;
                bid = nb;
                setup(mt2, nt2, more, po_next, ua_next);
                ksn = 0;
                uan = more ? ua_next : ua;
            }
            if (!(ABL & 2)) load_x(ksn);
            ld_scale(ksn, sc4, sh4);
            const float* r0 = rb0 + buf * (16 * 64 * KC);
            const float* r1 = rb1 + buf * (16 * 64 * KC);
            f32x4 b0 = *reinterpret_cast<const f32x4*>(r0), b1 = *reinterpret_cast<const f32x4*>(r1);
            __builtin_amdgcn_sched_barrier(0);
            auto position = [&](auto POSC) {
                constexpr int pos = decltype(POSC)::value;
                // program order is pinned per position (sched_barrier): fragment loads PF ahead, the next position's B reads, 8 MFMAs, then
                // a CLUMP of the next slice's patch transform (fp32 MFMA runs at the vector rate: VALU work beside it is not hidden, and
                // every MFMA <-> VALU switch costs a few cycles — one clump per position, not one instruction group per MFMA)
                if (pos + PF < 16) load_u(pos + PF, ua, ks);
                else load_u(pos + PF - 16, uan, ksn);
                f32x4 nb0 = b0, nb1 = b1;
                if (pos < 15 && !(ABL & 8)) {
                    nb0 = *reinterpret_cast<const f32x4*>(r0 + (pos + 1) * (64 * KC));
                    nb1 = *reinterpret_cast<const f32x4*>(r1 + (pos + 1) * (64 * KC));
                }
                const f32x4 a0 = a[pos][0], a1 = a[pos][1];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[pos] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], b0[e], acc[pos], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[pos] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], b1[e], acc[pos], 0, 0, 0);
                if constexpr (!(ABL & 4)) {
                    if constexpr (pos >= 8 && pos < 12) act_col(pos - 8, sc4, sh4);
                    if constexpr (pos >= 12) row_store(pos - 12, buf ^ 1);
                }
                b0 = nb0; b1 = nb1;
                __builtin_amdgcn_sched_barrier(0);
            };
            for_each_position(position, std::make_integer_sequence<int, 16>{});
            __syncthreads();
            buf ^= 1;
        }

        // ---- epilogue: Y = A^T M A per (tile, cout), 4 pixels x 4 consecutive couts per 16-byte store ------------------------------
#pragma unroll
        for (int g = (ABL & 16) ? 3 : 0; g < 4; ++g) {
            f32x4 o[2][2];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                float s[2][4];
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) {
                    const float m0 = acc[nu][r], m1 = acc[4 + nu][r], m2 = acc[8 + nu][r], m3 = acc[12 + nu][r];
                    s[0][nu] = m0 + m1 + m2;
                    s[1][nu] = m1 - m2 - m3;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    o[i][0][e] = s[i][0] + s[i][1] + s[i][2];
                    o[i][1][e] = s[i][1] - s[i][2] - s[i][3];
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) bst4(ry, o[i][j], po[i * 2 + j] == OOB ? OOB : po[i * 2 + j] + 32u * g);
        }
        if (!more) break;
#pragma unroll
        for (int k = 0; k < 4; ++k) po[k] = po_next[k];
        ua = ua_next;
    }
}

struct Shape { int Cin, Cout, H; double direct_tflops; };

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 128;
    // direct kernel's forward rate with the fused input activation at B = 128 (profiles/r04_conv_layers_steady.txt), for the ratio
    const Shape shapes[] = {{256, 256, 14, 115.6}, {64, 64, 56, 125.4}, {128, 128, 28, 124.9}, {512, 512, 7, 100.1}};
    int ncu = 256;
    { hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0)); ncu = pr.multiProcessorCount; }
    printf("Winograd F(2x2,3x3) exact-fp32 MFMA forward probe, B = %d, fused input activation, ABL = %d\n", B, ABL);
    for (const Shape& s : shapes) {
        const int H = s.H, W = s.H, Cin = s.Cin, Cout = s.Cout;
        WinoP p{};
        p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
        p.TH = (H + 1) / 2; p.TW = (W + 1) / 2; p.T = B * p.TH * p.TW;
        p.KS = Cin / KC; p.CB = Cout / 32; p.MT = (p.T + 63) / 64; p.NT = Cout / 64;
        const size_t nx = (size_t)B * H * W * Cin, ny = (size_t)B * H * W * Cout, nw = (size_t)Cout * 9 * Cin, nu = (size_t)16 * Cin * Cout;
        p.x_bytes = (int)(nx * 4); p.y_bytes = (int)(ny * 4); p.u_bytes = (int)(nu * 4);
        std::vector<float> hx(nx), hw(nw), hsc(Cin), hsh(Cin), hy(ny);
        std::mt19937 rng(1234 + Cin + H);
        std::normal_distribution<float> nd(0.f, 1.f);
        std::uniform_real_distribution<float> ud(0.5f, 1.5f);
        for (auto& v : hx) v = nd(rng);
        for (auto& v : hw) v = nd(rng) * 0.05f;
        for (auto& v : hsc) v = ud(rng);
        for (auto& v : hsh) v = nd(rng) * 0.5f;
        float *dx, *dw, *du, *dy, *dsc, *dsh;
        CHECK(hipMalloc(&dx, nx * 4)); CHECK(hipMalloc(&dw, nw * 4)); CHECK(hipMalloc(&du, nu * 4)); CHECK(hipMalloc(&dy, ny * 4));
        CHECK(hipMalloc(&dsc, Cin * 4)); CHECK(hipMalloc(&dsh, Cin * 4));
        CHECK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dsc, hsc.data(), Cin * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dsh, hsh.data(), Cin * 4, hipMemcpyHostToDevice));
        CHECK(hipMemset(dy, 0xff, ny * 4));
        p.x = dx; p.u = du; p.y = dy; p.sc = dsc; p.sh = dsh;
        const int NBh = ((p.MT + 7) / 8) * 8 * p.NT;
        const int grid = std::min(NBh, ncu * (argc > 2 ? atoi(argv[2]) : 1) / 1);
        auto run_w = [&]() { hipLaunchKernelGGL(k_wino_weights, dim3((Cin * Cout + 255) / 256), dim3(256), 0, 0, dw, du, Cin, Cout, p.KS, p.CB); };
        auto run = [&]() { hipLaunchKernelGGL(k_wino_fwd<true>, dim3(grid), dim3(256), 0, 0, p); };
        run_w(); run();
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(hy.data(), dy, ny * 4, hipMemcpyDeviceToHost));
        // ---- accuracy on sampled pixels: fp64 reference, and a plain fp32 FMA chain in (r, s, c) order (the direct kernel's arithmetic)
        double err_w = 0, err_d = 0, ref_max = 0;
        std::mt19937 prng(7);
        const int NS = 96;
        for (int sidx = 0; sidx < NS; ++sidx) {
            int b = prng() % B, yy = prng() % H, xx = prng() % W;
            if (sidx < 8) { yy = (sidx & 1) ? H - 1 : 0; xx = (sidx & 2) ? W - 1 : 0; b = (sidx & 4) ? B - 1 : 0; }   // the corners
            for (int k = 0; k < Cout; ++k) {
                double ref = 0; float chain = 0.f;
                for (int r = 0; r < 3; ++r)
                    for (int q = 0; q < 3; ++q) {
                        const int iy = yy + r - 1, ix = xx + q - 1;
                        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                        const float* xp = &hx[((size_t)(b * H + iy) * W + ix) * Cin];
                        const float* wp = &hw[((size_t)(k * 3 + r) * 3 + q) * Cin];
                        for (int c = 0; c < Cin; ++c) {
                            const float a = std::max(std::fmaf(xp[c], hsc[c], hsh[c]), 0.f);
                            ref += (double)a * (double)wp[c];
                            chain = std::fmaf(a, wp[c], chain);
                        }
                    }
                const float got = hy[((size_t)(b * H + yy) * W + xx) * Cout + k];
                err_w = std::max(err_w, std::fabs((double)got - ref));
                err_d = std::max(err_d, std::fabs((double)chain - ref));
                ref_max = std::max(ref_max, std::fabs(ref));
            }
        }
        // ---- timing -----------------------------------------------------------------------------------------------------------------
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int w = 0; w < 3; ++w) { for (int i = 0; i < 30; ++i) run(); CHECK(hipDeviceSynchronize()); }
        std::vector<float> ms;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) run();
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float t; CHECK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t / 20);
        }
        std::sort(ms.begin(), ms.end());
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) run_w();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float tw; CHECK(hipEventElapsedTime(&tw, e0, e1)); tw /= 20;
        const double direct_flop = 2.0 * B * H * W * (double)Cout * Cin * 9, mfma_flop = 2.0 * 16 * (double)p.MT * 64 * Cin * Cout;
        const double eff = direct_flop / (ms[2] * 1e-3) / 1e12, eff_w = direct_flop / ((ms[2] + tw) * 1e-3) / 1e12;
        printf("%4d->%4d 3x3 @%2d^2: %7.1f us (+ weights %5.1f us)  effective %6.1f TF/s (%6.1f with the weight transform) = %.2fx / %.2fx of the direct "
               "kernel's %.1f; MFMA-side %.1f TF/s; grid %d; max|err| wino %.3e direct-chain %.3e (ratio %.1f) at max|ref| %.2f\n",
               Cin, Cout, H, ms[2] * 1e3, tw * 1e3, eff, eff_w, eff / s.direct_tflops, eff_w / s.direct_tflops, s.direct_tflops,
               mfma_flop / (ms[2] * 1e-3) / 1e12, grid, err_w, err_d, err_w / std::max(err_d, 1e-30), ref_max);
        fflush(stdout);
        CHECK(hipFree(dx)); CHECK(hipFree(dw)); CHECK(hipFree(du)); CHECK(hipFree(dy)); CHECK(hipFree(dsc)); CHECK(hipFree(dsh));
    }
    return 0;
}
