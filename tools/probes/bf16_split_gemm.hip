// Probe (not product code): fp32-equivalent GEMM from split-bf16 MFMA on gfx950.
//   C[M][N] = A[M][K] . B[N][K]^T, fp32 in / fp32 out. Each fp32 operand is split into three bf16 planes a = a0 + a1 + a2
//   (a0 = top 8 significant bits, a1 the next 8, a2 the rest) while it is staged into LDS; the products a_i * b_j with
//   i + j <= TERMS-class are issued as v_mfma_f32_32x32x16_bf16 into ONE fp32 accumulator:
//     TERMS = 3: a0b0 + a0b1 + a1b0            (~2^-16 relative per product)
//     TERMS = 6: + a0b2 + a1b1 + a2b0          (~2^-24: the order of fp32's own rounding)
//     TERMS = 9: all nine                      (~2^-32)
//   Prints the achieved fp32-equivalent TFLOP/s and the error against an fp64 reference on sampled rows, next to the error of a
//   plain fp32 FMA chain on the same rows. Question it answers for DESIGN.md: is bf16x6 a credible replacement for the native
//   fp32 MFMA (157 TF/s peak) in the conv kernels, and what would it buy?
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/probes/bf16_split_gemm.hip -o /tmp/bf16_split && /tmp/bf16_split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 32, LDS_LD = BK + 8;   // bf16 elements per LDS row (80 B: 16-byte aligned rows)

// truncate-split: the three pieces sum to `v` exactly up to the bits below 2^-24 |v|
__device__ __forceinline__ void split3(float v, unsigned& h0, unsigned& h1, unsigned& h2) {
    unsigned b0 = __float_as_uint(v) & 0xffff0000u;
    float r1 = v - __uint_as_float(b0);
    unsigned b1 = __float_as_uint(r1) & 0xffff0000u;
    float r2 = r1 - __uint_as_float(b1);
    unsigned b2 = __float_as_uint(r2) & 0xffff0000u;
    h0 = b0 >> 16; h1 = b1 >> 16; h2 = b2 >> 16;
}

// global -> registers (issued before the MFMAs of the current K tile), registers -> three bf16 LDS planes (after them)
__device__ __forceinline__ void gload(const float* __restrict__ g, int ld, int rows_valid, int tid, f32x4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i, row = f >> 3, q = f & 7;
        v[i] = f32x4{0, 0, 0, 0};
        if (row < rows_valid) v[i] = *reinterpret_cast<const f32x4*>(g + (size_t)row * ld + q * 4);
    }
}
__device__ __forceinline__ void sstore(const f32x4 (&v)[4], unsigned short* s, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i, row = f >> 3, q = f & 7;
        unsigned h[3][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split3(v[i][e], h[0][e], h[1][e], h[2][e]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            u32x2 w = {h[pl][0] | (h[pl][1] << 16), h[pl][2] | (h[pl][3] << 16)};
            *reinterpret_cast<u32x2*>(s + (size_t)pl * BM * LDS_LD + row * LDS_LD + q * 4) = w;
        }
    }
}

// B (the weights) split ONCE into three bf16 planes in global memory, [3][N][K]: the GEMM then stages it without VALU work
__global__ void k_presplit(const float* __restrict__ B, unsigned short* __restrict__ P, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned h0, h1, h2;
    split3(B[i], h0, h1, h2);
    P[i] = (unsigned short)h0; P[n + i] = (unsigned short)h1; P[2 * n + i] = (unsigned short)h2;
}
// planes -> registers: per thread 2 x 16-byte loads per plane (128 rows x 32 bf16 = 512 x 16 B, 2 per thread)
__device__ __forceinline__ void gload_planes(const unsigned short* __restrict__ P, size_t plane_stride, int ld, int rows_valid, int tid,
                                             uint4 (&v)[3][2]) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i, row = f >> 2, q = f & 3;
            v[pl][i] = uint4{0, 0, 0, 0};
            if (row < rows_valid) v[pl][i] = *reinterpret_cast<const uint4*>(P + pl * plane_stride + (size_t)row * ld + q * 8);
        }
}
__device__ __forceinline__ void sstore_planes(const uint4 (&v)[3][2], unsigned short* s, int tid) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i, row = f >> 2, q = f & 3;
            *reinterpret_cast<uint4*>(s + (size_t)pl * BN * LDS_LD + row * LDS_LD + q * 8) = v[pl][i];
        }
}

template <int TERMS, bool PRESPLIT>
__global__ __launch_bounds__(256, 2) void k_gemm(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M,
                                              int N, int K, const unsigned short* __restrict__ BP) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
    unsigned short* sA = smem;                       // [3][BM][LDS_LD]
    unsigned short* sB = smem + 3 * BM * LDS_LD;     // [3][BN][LDS_LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int r31 = lane & 31, h = lane >> 5;
    f32x4 pa[4], pb[4];
    uint4 pp[3][2];
    const size_t pstride = (size_t)N * K;
    const int mv = min(BM, M - m0), nv = min(BN, N - n0);
    gload(A + (size_t)m0 * K, K, mv, tid, pa);
    if (PRESPLIT) gload_planes(BP + (size_t)n0 * K, pstride, K, nv, tid, pp); else gload(B + (size_t)n0 * K, K, nv, tid, pb);
    sstore(pa, sA, tid);
    if (PRESPLIT) sstore_planes(pp, sB, tid); else sstore(pb, sB, tid);
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += BK) {
        const bool more = k0 + BK < K;
        if (more) {
            gload(A + (size_t)m0 * K + k0 + BK, K, mv, tid, pa);
            if (PRESPLIT) gload_planes(BP + (size_t)n0 * K + k0 + BK, pstride, K, nv, tid, pp); else gload(B + (size_t)n0 * K + k0 + BK, K, nv, tid, pb);
        }
#pragma unroll
        for (int ks = 0; ks < BK; ks += 16) {
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    a[i][pl] = *reinterpret_cast<const bf16x8*>(sA + (size_t)pl * BM * LDS_LD + (wm * 64 + i * 32 + r31) * LDS_LD + ks + 8 * h);
                    b[i][pl] = *reinterpret_cast<const bf16x8*>(sB + (size_t)pl * BN * LDS_LD + (wn * 64 + i * 32 + r31) * LDS_LD + ks + 8 * h);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // small terms first
                    if (TERMS >= 9) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][2], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][2], acc[i][j], 0, 0, 0);
                    }
                    if (TERMS >= 6) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
        if (more) { sstore(pa, sA, tid); if (PRESPLIT) sstore_planes(pp, sB, tid); else sstore(pb, sB, tid); }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int col = n0 + wn * 64 + j * 32 + r31;
                if (row < M && col < N) C[(size_t)row * N + col] = acc[i][j][r];
            }
}

template <int TERMS, bool PRESPLIT = false>
static void run(const float* dA, const float* dB, float* dC, int M, int N, int K, const std::vector<float>& A, const std::vector<float>& B,
                const std::vector<int>& rows, const std::vector<double>& ref, double ref_scale, double fp32_err,
                const unsigned short* dBP = nullptr) {
    const size_t smem = (size_t)3 * (BM + BN) * LDS_LD * sizeof(unsigned short);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm<TERMS, PRESPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    dim3 grid((M + BM - 1) / BM, (N + BN - 1) / BN);
    hipLaunchKernelGGL((k_gemm<TERMS, PRESPLIT>), grid, dim3(256), smem, 0, dA, dB, dC, M, N, K, dBP);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_gemm<TERMS, PRESPLIT>), grid, dim3(256), smem, 0, dA, dB, dC, M, N, K, dBP);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<float> Crow(N);
    double err = 0;
    for (size_t ri = 0; ri < rows.size(); ++ri) {
        hipMemcpy(Crow.data(), dC + (size_t)rows[ri] * N, N * sizeof(float), hipMemcpyDeviceToHost);
        for (int n = 0; n < N; ++n) err = fmax(err, fabs((double)Crow[n] - ref[ri * N + n]));
    }
    printf("  bf16x%d%s: %7.3f ms  %7.1f TFLOP/s (fp32-equivalent)   max|err| / max|C| = %.3e   (plain fp32 FMA chain: %.3e)\n", TERMS, PRESPLIT ? " (B pre-split)" : "", ms,
           2.0 * M * N * K / (ms * 1e-3) / 1e12, err / ref_scale, fp32_err / ref_scale);
}

int main(int argc, char** argv) {
    const int shapes[][3] = {{100352, 512, 128}, {25088, 256, 1024}, {25088, 1024, 256}, {401408, 64, 256}, {8192, 8192, 8192}};
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        std::vector<float> A((size_t)M * K), B((size_t)N * K);
        unsigned st = 12345u + M;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
        for (auto& v : A) v = rnd() * (1.0f + 3.0f * fabsf(rnd()));   // activations-like spread
        for (auto& v : B) v = rnd() * 0.1f;
        float *dA, *dB, *dC;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        std::vector<int> rows;
        for (int i = 0; i < 24; ++i) rows.push_back((int)(((long long)i * 2654435761u) % M));
        std::vector<double> ref(rows.size() * (size_t)N);
        double scale = 0, fp32_err = 0;
        for (size_t ri = 0; ri < rows.size(); ++ri)
            for (int n = 0; n < N; ++n) {
                double acc = 0; float f = 0.f;
                const float* a = &A[(size_t)rows[ri] * K]; const float* b = &B[(size_t)n * K];
                for (int k = 0; k < K; ++k) { acc += (double)a[k] * b[k]; f = fmaf(a[k], b[k], f); }
                ref[ri * N + n] = acc; scale = fmax(scale, fabs(acc)); fp32_err = fmax(fp32_err, fabs((double)f - acc));
            }
        printf("M=%d N=%d K=%d\n", M, N, K);
        run<3>(dA, dB, dC, M, N, K, A, B, rows, ref, scale, fp32_err);
        run<6>(dA, dB, dC, M, N, K, A, B, rows, ref, scale, fp32_err);
        run<9>(dA, dB, dC, M, N, K, A, B, rows, ref, scale, fp32_err);
        unsigned short* dBP;
        hipMalloc(&dBP, B.size() * 6);
        hipLaunchKernelGGL(k_presplit, dim3((unsigned)((B.size() + 255) / 256)), dim3(256), 0, 0, dB, dBP, B.size());
        run<6, true>(dA, dB, dC, M, N, K, A, B, rows, ref, scale, fp32_err, dBP);
        hipFree(dBP);
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    return 0;
}
