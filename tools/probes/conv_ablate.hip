// Dev probe (GPU box): where does the 64x64 single-buffered fp32-MFMA tile loop lose its time?
// A GEMM-shaped copy of k_conv_fwd's 1x1 path (Y[M][N] = X[M][K] * W[N][K]^T, 4 waves of 32x32, BK = 32, buffer loads ->
// registers -> LDS R images -> ds_read_b128 -> v_mfma_f32_32x32x2_f32) with the loop's ingredients removed one at a time:
//   V0 full loop        V1 no global loads in the loop     V2 loads issued but never stored to LDS
//   V3 no barriers      V4 MFMA + LDS reads only           V5 MFMA only (operands in registers: the bare matrix-pipe rate)
//   V6 full loop, loads two tiles ahead (register double buffer)
//   V7 MFMA only with v_mfma_f32_16x16x4_f32 (4 independent 16x16 accumulators per wave, same FLOPs per cycle on paper)
//   V8 full loop on 16x16x4 (K order permuted inside each 16-k chunk so that one ds_read_b128 still feeds four MFMAs)
// Results are wrong for V1..V5 by construction; only the timing matters. Every variant keeps its inputs alive through asm volatile.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/conv_ablate.hip -o /tmp/conv_ablate && /tmp/conv_ablate
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int BK = 32, LDR = BK + 4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bld4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

template <int V>
__global__ __launch_bounds__(256, V == 6 ? 6 : 8) void k_gemm(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Y, int M,
                                                int N, int K, int MT, int NT, unsigned long long* stamps = nullptr) {
    // diagnostic clock read (MI355X_MICROARCH.md 'DVFS give-back' item 6): shader cycles / 100 MHz real-time ticks around the whole
    // workgroup; only when a stamp buffer is passed (a separate launch, never a timed one)
    unsigned long long c0 = 0, r0 = 0;
    if (stamps) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int nt = slot % NT, mt = (slot / NT) * 8 + xcd;
    if (mt >= MT) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = mt * 64, n0 = nt * 64, kq = tid & 7, lr = tid >> 3;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(X, M * K * 4), rw = make_rsrc(W, N * K * 4);
    uint32_t a_off[2], b_off[2];
    for (int i = 0; i < 2; ++i) {
        int m = m0 + lr + 32 * i;
        a_off[i] = m < M ? (uint32_t)((m * K + kq * 4) * 4) : 0x80000000u;
        b_off[i] = (uint32_t)(((n0 + lr + 32 * i) * K + kq * 4) * 4);
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4v acc4[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc4[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const int T = K / BK;
    f32x4 ra[2], rb[2], ra2[2], rb2[2];
    auto gload = [&](int t, f32x4 (&a)[2], f32x4 (&b)[2]) {
        for (int i = 0; i < 2; ++i) { a[i] = bld4(rx, a_off[i], (uint32_t)(t * BK * 4)); b[i] = bld4(rw, b_off[i], (uint32_t)(t * BK * 4)); }
    };
    auto sstore = [&](f32x4 (&a)[2], f32x4 (&b)[2]) {
        float* sA = smem; float* sB = sA + 64 * LDR;
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4*>(sA + (lr + 32 * i) * LDR + kq * 4) = a[i];
            *reinterpret_cast<f32x4*>(sB + (lr + 32 * i) * LDR + kq * 4) = b[i];
        }
    };
    auto mma = [&]() {
        const float* sA = smem; const float* sB = sA + 64 * LDR;
        const int h4 = (lane >> 5) * 4, l31 = lane & 31;
        if (V == 7 || V == 8) {
            // 16x16x4: lane (r = lane & 15, g = lane >> 4) supplies k = 4g + e to MFMA e of a 16-k chunk (same permutation for A and B)
            const int r15 = lane & 15, g4 = (lane >> 4) * 4;
#pragma unroll
            for (int c = 0; c < 2; ++c) {            // two 16-k chunks per 32-k tile
                f32x4 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (V == 7) { a[i] = ra[i]; b[i] = rb[i]; }
                    else {
                        a[i] = *reinterpret_cast<const f32x4*>(sA + (wm * 32 + 16 * i + r15) * LDR + 16 * c + g4);
                        b[i] = *reinterpret_cast<const f32x4*>(sB + (wn * 32 + 16 * i + r15) * LDR + 16 * c + g4);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[j][e], acc4[i][j], 0, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 a, b;
            if (V == 5) { a = ra[0]; b = rb[0]; }
            else {
                a = *reinterpret_cast<const f32x4*>(sA + (wm * 32 + l31) * LDR + 8 * j + h4);
                b = *reinterpret_cast<const f32x4*>(sB + (wn * 32 + l31) * LDR + 8 * j + h4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
        }
    };
    gload(0, ra, rb);
    sstore(ra, rb);
    __syncthreads();
    if (V == 6) {
        if (T > 1) gload(1, ra, rb);
        for (int t = 0; t < T; t += 2) {       // two tiles per trip: ra/rb hold t+1, ra2/rb2 receive t+2
            if (t + 2 < T) gload(t + 2, ra2, rb2);
            mma();
            __syncthreads();
            if (t + 1 < T) sstore(ra, rb);
            __syncthreads();
            if (t + 1 >= T) break;
            if (t + 3 < T) gload(t + 3, ra, rb);
            mma();
            __syncthreads();
            if (t + 2 < T) sstore(ra2, rb2);
            __syncthreads();
        }
    } else {
        for (int t = 0; t < T; ++t) {
            if (t + 1 < T && (V == 0 || V == 2 || V == 3 || V == 8)) gload(t + 1, ra, rb);
            mma();
            if (V != 3 && V != 4 && V != 5 && V != 7) __syncthreads();
            if (t + 1 < T) {
                if (V == 0 || V == 1 || V == 3 || V == 8) sstore(ra, rb);
                if (V == 2) asm volatile("" ::"v"(ra[0]), "v"(ra[1]), "v"(rb[0]), "v"(rb[1]));   // loads stay live (and waited for)
            }
            if (V != 3 && V != 4 && V != 5 && V != 7) __syncthreads();
        }
    }
    if (V == 7 || V == 8) {     // fold the four 16x16 accumulators into the store pattern below (layout irrelevant for timing)
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 4; ++r) acc[(i * 2 + j) * 4 + r] = acc4[i][j][r];
    }
    if (stamps && tid == 0) {
        stamps[2 * (size_t)blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
        stamps[2 * (size_t)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    // epilogue as in the product kernel's generic path (enough to keep the accumulators alive)
    const int col = n0 + wn * 32 + (lane & 31);
    for (int r = 0; r < 16; ++r) {
        int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M) Y[(size_t)m * N + col] = acc[r];
    }
}

// V10: the full loop with LDS-DMA staging (global_load_lds_dwordx4: global -> LDS without a VGPR round trip, no ds_write), two LDS
// buffers, ONE barrier per K tile. The LDS image is unpadded [rows][32 floats]; bank conflicts of the ds_read_b128 fragment reads are
// avoided by XOR-swizzling the 16-byte slot with (row >> 1) & 7 — on the SOURCE address of the DMA and on the read, never on the
// DMA's destination (which is wave-uniform base + lane * 16).
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
__global__ __launch_bounds__(256, 5) void k_gemm_dma(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Y, int M,
                                                    int N, int K, int MT, int NT) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int nt = slot % NT, mt = (slot / NT) * 8 + xcd;
    if (mt >= MT) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = mt * 64, n0 = nt * 64;
    constexpr int TILE = 64 * 32;             // floats per operand tile
    constexpr int STG = 2 * TILE;             // A | B
    // this lane's two rows per operand (pass i: rows 32 i + 8 wave + lane / 8), physical slot lane & 7 -> logical (source) slot
    size_t a_src[2], b_src[2];
    for (int i = 0; i < 2; ++i) {
        const int r = 32 * i + 8 * wave + (lane >> 3);
        const int sl = (lane & 7) ^ ((r >> 1) & 7);
        int m = m0 + r; if (m >= M) m = M - 1;          // rows past the end: any valid address, their results are never stored
        a_src[i] = (size_t)m * K + sl * 4;
        b_src[i] = (size_t)(n0 + r) * K + sl * 4;
    }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int T = K / BK;
    auto dma = [&](int t, int buf) {
        float* sA = smem + buf * STG; float* sB = sA + TILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(X + a_src[i] + (size_t)t * BK), (lds_ptr_t)(sA + (32 * i + 8 * wave) * 32), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(W + b_src[i] + (size_t)t * BK), (lds_ptr_t)(sB + (32 * i + 8 * wave) * 32), 16, 0, 0);
        }
    };
    const int h = lane >> 5, l31 = lane & 31;
    const int ra_ = wm * 32 + l31, rb_ = wn * 32 + l31;
    const int fa = (ra_ >> 1) & 7, fb = (rb_ >> 1) & 7;
    dma(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        if (t + 1 < T) dma(t + 1, buf ^ 1);
        const float* sA = smem + buf * STG; const float* sB = sA + TILE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(sA + ra_ * 32 + (((2 * j + h) ^ fa) << 2));
            const f32x4 b = *reinterpret_cast<const f32x4*>(sB + rb_ * 32 + (((2 * j + h) ^ fb) << 2));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
        }
        __syncthreads();      // hipcc drains vmcnt(0) here: tile t+1 has landed, and every wave is done reading tile t
    }
    const int col = n0 + wn * 32 + (lane & 31);
    for (int r = 0; r < 16; ++r) {
        int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M) Y[(size_t)m * N + col] = acc[r];
    }
}
static float run10(const float* X, const float* W, float* Y, int M, int N, int K, int reps) {
    const int MT = (M + 63) / 64, NT = N / 64;
    const int grid = (MT + 7) / 8 * 8 * NT;
    const size_t smem = (size_t)2 * 2 * 64 * 32 * sizeof(float);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_gemm_dma, dim3(grid), dim3(256), smem, 0, X, W, Y, M, N, K, MT, NT);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_gemm_dma, dim3(grid), dim3(256), smem, 0, X, W, Y, M, N, K, MT, NT);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

// V9: the full loop on a 32 (M) x 64 (N) tile with TWO waves (128 threads): twice the workgroups of the 64x64 form for the layers
// that have only 3-6 of those per CU (14x14 and 7x7 spatial sizes), at 1.5x the operand traffic per FLOP.
__global__ __launch_bounds__(128, 8) void k_gemm_32x64(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Y, int M,
                                                      int N, int K, int MT, int NT) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int nt = slot % NT, mt = (slot / NT) * 8 + xcd;
    if (mt >= MT) return;
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int m0 = mt * 32, n0 = nt * 64, kq = tid & 7, lr = tid >> 3;      // 16 rows per pass
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(X, M * K * 4), rw = make_rsrc(W, N * K * 4);
    uint32_t a_off[2], b_off[4];
    for (int i = 0; i < 2; ++i) { int m = m0 + lr + 16 * i; a_off[i] = m < M ? (uint32_t)((m * K + kq * 4) * 4) : 0x80000000u; }
    for (int i = 0; i < 4; ++i) b_off[i] = (uint32_t)(((n0 + lr + 16 * i) * K + kq * 4) * 4);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int T = K / BK;
    f32x4 ra[2], rb[4];
    float* sA = smem; float* sB = sA + 32 * LDR;
    auto gload = [&](int t) {
        for (int i = 0; i < 2; ++i) ra[i] = bld4(rx, a_off[i], (uint32_t)(t * BK * 4));
        for (int i = 0; i < 4; ++i) rb[i] = bld4(rw, b_off[i], (uint32_t)(t * BK * 4));
    };
    auto sstore = [&]() {
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(sA + (lr + 16 * i) * LDR + kq * 4) = ra[i];
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(sB + (lr + 16 * i) * LDR + kq * 4) = rb[i];
    };
    gload(0); sstore(); __syncthreads();
    const int h4 = (lane >> 5) * 4, l31 = lane & 31;
    for (int t = 0; t < T; ++t) {
        if (t + 1 < T) gload(t + 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(sA + l31 * LDR + 8 * j + h4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(sB + (wn * 32 + l31) * LDR + 8 * j + h4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
        }
        __syncthreads();
        if (t + 1 < T) sstore();
        __syncthreads();
    }
    const int col = n0 + wn * 32 + (lane & 31);
    for (int r = 0; r < 16; ++r) {
        int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M) Y[(size_t)m * N + col] = acc[r];
    }
}
static float run9(const float* X, const float* W, float* Y, int M, int N, int K, int reps) {
    const int MT = (M + 31) / 32, NT = N / 64;
    const int grid = (MT + 7) / 8 * 8 * NT;
    const size_t smem = (size_t)96 * LDR * sizeof(float);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_gemm_32x64, dim3(grid), dim3(128), smem, 0, X, W, Y, M, N, K, MT, NT);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_gemm_32x64, dim3(grid), dim3(128), smem, 0, X, W, Y, M, N, K, MT, NT);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}


// V11 (round 4): PERSISTENT workgroups with cross-tile prefetch. grid = min(tiles, 8 per CU); a workgroup walks tiles wg, wg + G, ...
// (same XCD-aware tile order as V0) and treats their K tiles as ONE stream: behind the last K tile of a tile it fetches the first K
// tile of the NEXT one, before the MFMAs and the epilogue, so no tile but the first pays a prologue round trip, no workgroup is
// launched per tile, and a new tile's first instructions are not the youngest wave's (tools/wg_timeline.py: 7-35 us prologues).
__global__ __launch_bounds__(256, 8) void k_gemm_persist(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Y, int M,
                                                        int N, int K, int MT, int NT, int ntiles_padded) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int kq = tid & 7, lr = tid >> 3;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(X, M * K * 4), rw = make_rsrc(W, N * K * 4);
    const int T = K / BK;
    const int G = gridDim.x;
    float* sA = smem; float* sB = sA + 64 * LDR;
    const int h4 = (lane >> 5) * 4, l31 = lane & 31;
    auto tile_of = [&](int b, int& mt, int& nt) {
        const int xcd = b & 7, slot = b >> 3;
        nt = slot % NT; mt = (slot / NT) * 8 + xcd;
        return mt < MT;
    };
    uint32_t a_off[2], b_off[2];
    auto offsets = [&](int mt, int nt) {
        for (int i = 0; i < 2; ++i) {
            const int m = mt * 64 + lr + 32 * i;
            a_off[i] = m < M ? (uint32_t)((m * K + kq * 4) * 4) : 0x80000000u;
            b_off[i] = (uint32_t)(((nt * 64 + lr + 32 * i) * K + kq * 4) * 4);
        }
    };
    f32x4 ra[2], rb[2];
    auto gload = [&](int t) {
        for (int i = 0; i < 2; ++i) { ra[i] = bld4(rx, a_off[i], (uint32_t)(t * BK * 4)); rb[i] = bld4(rw, b_off[i], (uint32_t)(t * BK * 4)); }
    };
    auto sstore = [&]() {
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4*>(sA + (lr + 32 * i) * LDR + kq * 4) = ra[i];
            *reinterpret_cast<f32x4*>(sB + (lr + 32 * i) * LDR + kq * 4) = rb[i];
        }
    };
    int b = blockIdx.x, mt, nt;
    while (b < ntiles_padded && !tile_of(b, mt, nt)) b += G;
    if (b >= ntiles_padded) return;
    offsets(mt, nt);
    gload(0); sstore();
    __syncthreads();
    while (true) {
        int bn = b + G, mtn = 0, ntn = 0;
        while (bn < ntiles_padded && !tile_of(bn, mtn, ntn)) bn += G;
        const bool has_next = bn < ntiles_padded;
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int t = 0; t < T; ++t) {
            if (t + 1 < T) gload(t + 1);
            else if (has_next) { offsets(mtn, ntn); gload(0); }     // the next tile's first K tile, behind this tile's last
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(sA + (wm * 32 + l31) * LDR + 8 * j + h4);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(sB + (wn * 32 + l31) * LDR + 8 * j + h4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], bb[e], acc, 0, 0, 0);
            }
            __syncthreads();
            if (t + 1 < T) { sstore(); __syncthreads(); }
        }
        // epilogue of tile (mt, nt) (the generic store pattern of V0), then the prefetched K tile goes to LDS
        const int col = nt * 64 + wn * 32 + (lane & 31);
        for (int r = 0; r < 16; ++r) {
            const int m = mt * 64 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < M) Y[(size_t)m * N + col] = acc[r];
        }
        if (!has_next) break;
        sstore();
        __syncthreads();
        b = bn; mt = mtn; nt = ntn;
    }
}
static float run11(const float* X, const float* W, float* Y, int M, int N, int K, int reps) {
    const int MT = (M + 63) / 64, NT = N / 64;
    const int padded = (MT + 7) / 8 * 8 * NT;
    const int grid = padded < 8 * 256 ? padded : 8 * 256;
    const size_t smem = (size_t)128 * LDR * sizeof(float);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_gemm_persist, dim3(grid), dim3(256), smem, 0, X, W, Y, M, N, K, MT, NT, padded);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_gemm_persist, dim3(grid), dim3(256), smem, 0, X, W, Y, M, N, K, MT, NT, padded);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

static int g_warm_launches = 30;   // back-to-back launches in front of the stamped one (`steady` mode: enough for ~0.4 s of load)
template <int V>
static double clock_ghz(const float* X, const float* W, float* Y, int M, int N, int K) {
    const int MT = (M + 63) / 64, NT = N / 64;
    const int grid = (MT + 7) / 8 * 8 * NT;
    const size_t smem = (size_t)128 * LDR * sizeof(float);
    unsigned long long* d;
    hipMalloc(&d, (size_t)grid * 16); hipMemset(d, 0, (size_t)grid * 16);
    for (int i = 0; i < g_warm_launches; ++i) hipLaunchKernelGGL((k_gemm<V>), dim3(grid), dim3(256), smem, 0, X, W, Y, M, N, K, MT, NT, (unsigned long long*)nullptr);
    hipLaunchKernelGGL((k_gemm<V>), dim3(grid), dim3(256), smem, 0, X, W, Y, M, N, K, MT, NT, d);      // hot: stamped launch after the back-to-back ones
    std::vector<unsigned long long> h((size_t)grid * 2);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    hipFree(d);
    std::vector<double> g;
    for (int b = 0; b < grid; ++b) if (h[2 * b + 1] > 50) g.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);   // cycles per 10 ns tick -> GHz
    if (g.empty()) return 0;
    std::sort(g.begin(), g.end());
    return g[g.size() / 2];
}

template <int V>
static float run(const float* X, const float* W, float* Y, int M, int N, int K, int reps) {
    const int MT = (M + 63) / 64, NT = N / 64;
    const int grid = (MT + 7) / 8 * 8 * NT;
    const size_t smem = (size_t)128 * LDR * sizeof(float);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_gemm<V>), dim3(grid), dim3(256), smem, 0, X, W, Y, M, N, K, MT, NT);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_gemm<V>), dim3(grid), dim3(256), smem, 0, X, W, Y, M, N, K, MT, NT);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

// `conv_ablate long`: ONE shape sized for dispatches of ~8 ms (M = 401408, N = K = 1024; 842 GFLOP), full loop and mfma-only, each
// launched 6 times back to back. Run under `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE`: per MI355X_MICROARCH.md ('DVFS give-back')
// GRBM_GUI_ACTIVE / 8 / dispatch wall time is the effective shader clock on dispatches this long — an out-of-kernel cross-check of
// the s_memtime / s_memrealtime clock printed here for the same launches.
static int long_mode() {
    const int M = 401408, N = 1024, K = 1024;
    float *X, *W, *Y;
    if (hipMalloc(&X, (size_t)M * K * 4) != hipSuccess || hipMalloc(&W, (size_t)N * K * 4) != hipSuccess ||
        hipMalloc(&Y, (size_t)M * N * 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    {
        std::vector<float> h((size_t)M * K);
        unsigned s = 12345u;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) / 16777216.0f - 0.5f; }
        hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    }
    const double gflop = 2.0 * M * N * K / 1e9;
    const float t0 = run<0>(X, W, Y, M, N, K, 6), t5 = run<5>(X, W, Y, M, N, K, 6);
    printf("long shape M=%d N=%d K=%d (%.0f GFLOP, %d workgroups): full loop %.3f ms = %.1f TFLOP/s | mfma-only %.3f ms = %.1f TFLOP/s\n", M, N, K,
           gflop, (M + 63) / 64 * (N / 64), t0, gflop / t0, t5, gflop / t5);
    printf("  in-kernel clock (s_memtime / s_memrealtime, median over workgroups): full loop %.3f GHz | mfma-only %.3f GHz\n",
           clock_ghz<0>(X, W, Y, M, N, K), clock_ghz<5>(X, W, Y, M, N, K));
    hipFree(X); hipFree(W); hipFree(Y);
    return 0;
}

// `conv_ablate steady`: the ResNet shapes again, but every clock stamp and every timing comes after ~0.4 s of back-to-back launches of
// the same kernel (the shader clock needs tens of milliseconds of load to settle after an idle gap — see `long` mode's first dispatches).
static int steady_mode() {
    struct Shape { int M, N, K; const char* what; };
    const Shape shapes[] = {{100352, 128, 512, "512->128 1x1 @28 (B=128)"}, {25088, 256, 2304, "256->256 3x3 @14 as GEMM"},
                            {401408, 256, 64, "64->256 1x1 @56"}, {25088, 1024, 256, "256->1024 1x1 @14"},
                            {6272, 512, 4608, "512->512 3x3 @7 as GEMM"}, {25088, 256, 1024, "1024->256 1x1 @14"},
                            {6272, 512, 2048, "2048->512 1x1 @7"}};
    for (const Shape& s : shapes) {
        float *X, *W, *Y;
        hipMalloc(&X, (size_t)s.M * s.K * 4); hipMalloc(&W, (size_t)s.N * s.K * 4); hipMalloc(&Y, (size_t)s.M * s.N * 4);
        std::vector<float> h((size_t)s.M * s.K);
        for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
        hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(W, h.data(), (size_t)s.N * s.K * 4, hipMemcpyHostToDevice);
        const double gflop = 2.0 * s.M * s.N * s.K / 1e9;
        const float t_cold = run<0>(X, W, Y, s.M, s.N, s.K, 10);
        const int n = (int)(400.0f / t_cold) + 30;
        g_warm_launches = n;
        const double c0 = clock_ghz<0>(X, W, Y, s.M, s.N, s.K);
        const float t0 = run<0>(X, W, Y, s.M, s.N, s.K, n);          // n launches timed as one block: steady state
        const double c5 = clock_ghz<5>(X, W, Y, s.M, s.N, s.K);
        const float t5 = run<5>(X, W, Y, s.M, s.N, s.K, n);
        printf("%-28s %6.2f WG/CU | after idle: full %.1f TFLOP/s | steady (%d launches): full %.1f TFLOP/s at %.2f GHz = %.0f %% of the pipe rate at that clock"
               " | mfma-only %.1f TFLOP/s at %.2f GHz = %.0f %%\n", s.what, (s.M + 63) / 64 * (s.N / 64) / 256.0, gflop / t_cold, n, gflop / t0, c0,
               100.0 * (gflop / t0) / (65.54 * c0), gflop / t5, c5, 100.0 * (gflop / t5) / (65.54 * c5));
        const float t[9] = {t0, run<1>(X, W, Y, s.M, s.N, s.K, n), run<2>(X, W, Y, s.M, s.N, s.K, n), run<3>(X, W, Y, s.M, s.N, s.K, n),
                            run<4>(X, W, Y, s.M, s.N, s.K, n), t5, run<6>(X, W, Y, s.M, s.N, s.K, n), run<7>(X, W, Y, s.M, s.N, s.K, n),
                            run<8>(X, W, Y, s.M, s.N, s.K, n)};
        printf("    steady TFLOP/s: full %.1f | no-gload %.1f | no-lds-store %.1f | no-barrier %.1f | mfma+ldsread %.1f | mfma-only %.1f | 2-ahead %.1f | "
               "16x16x4 mfma-only %.1f | 16x16x4 full %.1f | 32x64 tile, 2 waves %.1f | LDS-DMA double buffer %.1f\n", gflop / t[0], gflop / t[1], gflop / t[2],
               gflop / t[3], gflop / t[4], gflop / t[5], gflop / t[6], gflop / t[7], gflop / t[8], gflop / run9(X, W, Y, s.M, s.N, s.K, n),
               gflop / run10(X, W, Y, s.M, s.N, s.K, n));
        hipFree(X); hipFree(W); hipFree(Y);
    }
    return 0;
}


// `conv_ablate persist`: V0 (one workgroup per tile) against V11 (persistent workgroups, cross-tile prefetch) in steady state, with a
// bitwise comparison of the outputs (same K order per tile).
static int persist_mode() {
    struct Shape { int M, N, K; const char* what; };
    const Shape shapes[] = {{401408, 256, 64, "64->256 1x1 @56"}, {401408, 64, 256, "256->64 1x1 @56"}, {100352, 512, 128, "128->512 1x1 @28"},
                            {100352, 128, 512, "512->128 1x1 @28"}, {25088, 1024, 256, "256->1024 1x1 @14"}, {25088, 256, 1024, "1024->256 1x1 @14"},
                            {25088, 256, 2304, "256->256 3x3 @14 as GEMM"}, {6272, 2048, 512, "512->2048 1x1 @7"}, {6272, 512, 2048, "2048->512 1x1 @7"}};
    for (const Shape& s : shapes) {
        float *X, *W, *Y;
        hipMalloc(&X, (size_t)s.M * s.K * 4); hipMalloc(&W, (size_t)s.N * s.K * 4); hipMalloc(&Y, (size_t)s.M * s.N * 4);
        std::vector<float> h((size_t)s.M * s.K);
        for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
        hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(W, h.data(), (size_t)s.N * s.K * 4, hipMemcpyHostToDevice);
        const double gflop = 2.0 * s.M * s.N * s.K / 1e9;
        std::vector<float> y0((size_t)s.M * s.N), y1((size_t)s.M * s.N);
        run<0>(X, W, Y, s.M, s.N, s.K, 1); hipMemcpy(y0.data(), Y, y0.size() * 4, hipMemcpyDeviceToHost);
        hipMemset(Y, 0, y0.size() * 4);
        run11(X, W, Y, s.M, s.N, s.K, 1); hipMemcpy(y1.data(), Y, y1.size() * 4, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 0; i < y0.size(); ++i) bad += y0[i] != y1[i];
        const float t_cold = run<0>(X, W, Y, s.M, s.N, s.K, 10);
        const int n = (int)(300.0f / t_cold) + 30;
        float a[3], b[3];
        for (int r = 0; r < 3; ++r) { a[r] = run<0>(X, W, Y, s.M, s.N, s.K, n); b[r] = run11(X, W, Y, s.M, s.N, s.K, n); }
        std::sort(a, a + 3); std::sort(b, b + 3);
        printf("%-26s %6.2f tiles/CU T=%3d | one workgroup per tile %.1f TFLOP/s | persistent + cross-tile prefetch %.1f TFLOP/s (%+.1f %%) | %zu outputs differ\n",
               s.what, (s.M + 63) / 64 * (s.N / 64) / 256.0, s.K / 32, gflop / a[1], gflop / b[1], 100.0 * (a[1] / b[1] - 1.0), bad);
        hipFree(X); hipFree(W); hipFree(Y);
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "long")) return long_mode();
    if (argc > 1 && !strcmp(argv[1], "steady")) return steady_mode();
    if (argc > 1 && !strcmp(argv[1], "persist")) return persist_mode();
    struct Shape { int M, N, K; const char* what; };
    const Shape shapes[] = {{100352, 128, 512, "512->128 1x1 @28 (B=128)"}, {25088, 256, 2304, "256->256 3x3 @14 as GEMM"},
                            {401408, 256, 64, "64->256 1x1 @56"}, {25088, 1024, 256, "256->1024 1x1 @14"},
                            {6272, 512, 4608, "512->512 3x3 @7 as GEMM"}, {25088, 256, 1024, "1024->256 1x1 @14"},
                            {6272, 512, 2048, "2048->512 1x1 @7"}};
    for (const Shape& s : shapes) {
        float *X, *W, *Y;
        hipMalloc(&X, (size_t)s.M * s.K * 4); hipMalloc(&W, (size_t)s.N * s.K * 4); hipMalloc(&Y, (size_t)s.M * s.N * 4);
        std::vector<float> h((size_t)s.M * s.K);
        for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
        hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(W, h.data(), (size_t)s.N * s.K * 4, hipMemcpyHostToDevice);
        const double gflop = 2.0 * s.M * s.N * s.K / 1e9;
        {   // V10 must compute the same GEMM as V0 (same K order inside a tile: bitwise)
            std::vector<float> y0((size_t)s.M * s.N), y1((size_t)s.M * s.N);
            run<0>(X, W, Y, s.M, s.N, s.K, 1); hipMemcpy(y0.data(), Y, y0.size() * 4, hipMemcpyDeviceToHost);
            hipMemset(Y, 0, y0.size() * 4);
            run10(X, W, Y, s.M, s.N, s.K, 1); hipMemcpy(y1.data(), Y, y1.size() * 4, hipMemcpyDeviceToHost);
            size_t bad = 0; for (size_t i = 0; i < y0.size(); ++i) bad += y0[i] != y1[i];
            printf("  LDS-DMA variant vs full loop: %zu of %zu outputs differ\n", bad, y0.size());
        }
        printf("%-28s M=%d N=%d K=%d  (%d workgroups, %.2f per CU)\n", s.what, s.M, s.N, s.K, (s.M + 63) / 64 * (s.N / 64),
               (s.M + 63) / 64 * (s.N / 64) / 256.0);
        printf("  in-kernel clock (median over workgroups, after 30 back-to-back launches): full loop %.2f GHz | mfma-only %.2f GHz | no-gload %.2f GHz\n",
               clock_ghz<0>(X, W, Y, s.M, s.N, s.K), clock_ghz<5>(X, W, Y, s.M, s.N, s.K), clock_ghz<1>(X, W, Y, s.M, s.N, s.K));
        for (int round = 0; round < 2; ++round) {
            float t7 = run<7>(X, W, Y, s.M, s.N, s.K, 10), t8 = run<8>(X, W, Y, s.M, s.N, s.K, 10), t9 = run9(X, W, Y, s.M, s.N, s.K, 10), t10 = run10(X, W, Y, s.M, s.N, s.K, 10);
            float t[7] = {run<0>(X, W, Y, s.M, s.N, s.K, 10), run<1>(X, W, Y, s.M, s.N, s.K, 10), run<2>(X, W, Y, s.M, s.N, s.K, 10),
                          run<3>(X, W, Y, s.M, s.N, s.K, 10), run<4>(X, W, Y, s.M, s.N, s.K, 10), run<5>(X, W, Y, s.M, s.N, s.K, 10),
                          run<6>(X, W, Y, s.M, s.N, s.K, 10)};
            printf("  round %d TFLOP/s: full %.1f | no-gload %.1f | no-lds-store %.1f | no-barrier %.1f | mfma+ldsread %.1f | mfma-only %.1f | 2-ahead %.1f | 16x16x4 mfma-only %.1f | 16x16x4 full %.1f | 32x64 tile, 2 waves %.1f | LDS-DMA double buffer %.1f\n",
                   round, gflop / t[0], gflop / t[1], gflop / t[2], gflop / t[3], gflop / t[4], gflop / t[5], gflop / t[6], gflop / t7, gflop / t8, gflop / t9, gflop / t10);
        }
        hipFree(X); hipFree(W); hipFree(Y);
    }
    return 0;
}
