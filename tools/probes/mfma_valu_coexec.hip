// Probe (dev, GPU box): do ordinary vector instructions of ONE wave execute beside its own matrix instructions on gfx950?
// The Winograd kernels (csrc/conv_wino.hip) run one wave per SIMD and pay their transforms as VALU work between MFMAs; DESIGN.md section 3
// prices that work in full ("fp32 MFMA and VALU do not overlap within a lone wave"). This probe measures the claim directly and
// gives the SQ counters a known case:
//   mode f32   v_mfma_f32_32x32x2_f32   (64 pipe cycles, the conv kernels' instruction)
//   mode bf16  v_mfma_f32_32x32x16_bf16 (the guide's bf16 loops, where interleaved VALU work is reported hidden)
// Loop body: ONE matrix instruction followed by K independent v_fma_f32 (K = 0, 2, 4, 8, 12, 16, 24), 4 rotating accumulators,
// one wave per SIMD (256-thread workgroups, one per CU) or, with `waves` = 2, two waves per SIMD (512-thread workgroups).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_valu_coexec.hip -o tools/probes/bin/mfma_valu_coexec
//   tools/probes/bin/mfma_valu_coexec            -> table: cycles per loop iteration against K
//   rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -- tools/probes/bin/mfma_valu_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
// the same loop with K PACKED fp32 instructions (v_pk_add_f32: two results per lane and instruction) per matrix instruction: what would the
// Winograd transforms cost if their adds were paired?
template <int K>
__global__ __launch_bounds__(512) void k_probe_pk(float* out, int iters, float seed) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = seed * (float)(j + r);
    f32x2 v[24];
    for (int i = 0; i < 24; ++i) v[i] = f32x2{seed + (float)(threadIdx.x + i), seed - (float)i};
    const float a = seed + 1.f, b = seed + 2.f;
    const f32x2 c = {seed, seed + 3.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < K; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    for (int i = 0; i < 24; ++i) s += v[i][0] + v[i][1];
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int K>
static double run_pk(float* d, int iters, int waves, int cus) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_probe_pk<K>), dim3(cus), dim3(256 * waves), 0, 0, d, iters, 0.001f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_probe_pk<K>), dim3(cus), dim3(256 * waves), 0, 0, d, iters, 0.001f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int K, bool BF16>
__global__ __launch_bounds__(512) void k_probe(float* out, int iters, float seed) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = seed * (float)(j + r);
    float v[24];
    for (int i = 0; i < 24; ++i) v[i] = seed + (float)(threadIdx.x + i);
    const float a = seed + 1.f, b = seed + 2.f;
    bf16x8 ab, bb;
    for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(seed + (float)i); bb[i] = (__bf16)(seed - (float)i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (BF16) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[j], 0, 0, 0);
            else acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < K; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    for (int i = 0; i < 24; ++i) s += v[i];
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, bool BF16>
static double run(float* d, int iters, int waves, int cus) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_probe<K, BF16>), dim3(cus), dim3(256 * waves), 0, 0, d, iters, 0.001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_probe<K, BF16>), dim3(cus), dim3(256 * waves), 0, 0, d, iters, 0.001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main(int argc, char** argv) {
    int dev = 0, cus = 256, khz = 0;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, dev);
    float* d = nullptr;
    hipMalloc(&d, (size_t)cus * 512 * sizeof(float));
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    printf("one matrix instruction + K independent v_fma_f32 per loop step, %d steps x 4, %d workgroups; ns per step (and cycles at the nominal %.2f GHz)\n",
           iters, cus, khz / 1e6);
    printf("%-28s %8s %8s %8s %8s %8s %8s %8s\n", "K =", "0", "2", "4", "8", "12", "16", "24");
    for (int waves = 1; waves <= 2; ++waves) {
        double f[7] = {run<0, false>(d, iters, waves, cus), run<2, false>(d, iters, waves, cus), run<4, false>(d, iters, waves, cus), run<8, false>(d, iters, waves, cus),
                       run<12, false>(d, iters, waves, cus), run<16, false>(d, iters, waves, cus), run<24, false>(d, iters, waves, cus)};
        double h[7] = {run<0, true>(d, iters, waves, cus), run<2, true>(d, iters, waves, cus), run<4, true>(d, iters, waves, cus), run<8, true>(d, iters, waves, cus),
                       run<12, true>(d, iters, waves, cus), run<16, true>(d, iters, waves, cus), run<24, true>(d, iters, waves, cus)};
        const double steps = (double)iters * 4 * waves;      // matrix instructions per SIMD
        printf("f32 32x32x2, %d wave/SIMD  ns ", waves);
        for (double t : f) printf(" %8.2f", t * 1e6 / steps);
        printf("\n%-28s   ", "   cycles per step");
        for (double t : f) printf(" %8.1f", t * 1e6 / steps * khz / 1e6);
        double q[7] = {run_pk<0>(d, iters, waves, cus), run_pk<2>(d, iters, waves, cus), run_pk<4>(d, iters, waves, cus), run_pk<8>(d, iters, waves, cus),
                       run_pk<12>(d, iters, waves, cus), run_pk<16>(d, iters, waves, cus), run_pk<24>(d, iters, waves, cus)};
        printf("\nf32 + K v_pk_add_f32, %d w/S ns", waves);
        for (double t : q) printf(" %8.2f", t * 1e6 / steps);
        printf("\n%-28s   ", "   cycles per step");
        for (double t : q) printf(" %8.1f", t * 1e6 / steps * khz / 1e6);
        printf("\nbf16 32x32x16, %d wave/SIMD ns", waves);
        for (double t : h) printf(" %8.2f", t * 1e6 / steps);
        printf("\n%-28s   ", "   cycles per step");
        for (double t : h) printf(" %8.1f", t * 1e6 / steps * khz / 1e6);
        printf("\n");
    }
    hipFree(d);
    return 0;
}
