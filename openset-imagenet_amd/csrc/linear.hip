// The two small dense layers of the reference head: resnet_base.fc = Linear(2048, C) with bias
// (openset_imagenet/model.py:19-20) and logits = Linear(C, C, bias=False) (model.py:23-26).
// C is 30..152 (not a multiple of anything), FLOPs are 1e-5 of the step: plain fp32 FMA kernels, no MFMA.
//   y[b, o]  = sum_k x[b, k] * w[o, k] + bias[o]
//   dx[b, k] = sum_o dy[b, o] * w[o, k]
//   dw[o, k] = sum_b dy[b, o] * x[b, k]      db[o] = sum_b dy[b, o]
// Summation orders are fixed (no atomics): results are bitwise reproducible.
#include "osi_common.h"

namespace {

// one wave per output element, lanes stride over K
__global__ __launch_bounds__(256) void k_linear_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                   float* __restrict__ y, int B, int K, int O) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (o >= O) return;
    const float* xr = x + (size_t)b * K;
    const float* wr = w + (size_t)o * K;
    float acc = 0.f;
    if ((K & 3) == 0) {
        for (int k = lane * 4; k < K; k += 256) {
            f32x4 a = *reinterpret_cast<const f32x4*>(xr + k), c = *reinterpret_cast<const f32x4*>(wr + k);
            acc += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
        }
    } else {
        for (int k = lane; k < K; k += 64) acc += xr[k] * wr[k];
    }
    acc = wave_sum(acc);
    if (lane == 0) y[(size_t)b * O + o] = acc + (bias ? bias[o] : 0.f);
}

__global__ __launch_bounds__(256) void k_linear_dx(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int B,
                                                  int K, int O, int accumulate) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (k >= K) return;
    float acc = 0.f;
    int o = 0;
    for (; o + 8 <= O; o += 8) {    // eight loads in flight; the sum keeps the order o = 0, 1, 2, ...
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[j] = w[(size_t)(o + j) * K + k];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += dy[(size_t)b * O + o + j] * wv[j];
    }
    for (; o < O; ++o) acc += dy[(size_t)b * O + o] * w[(size_t)o * K + k];
    float* d = dx + (size_t)b * K + k;
    *d = accumulate ? *d + acc : acc;
}

__global__ __launch_bounds__(256) void k_linear_dw(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw, int B,
                                                  int K, int O) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int o = blockIdx.y;
    if (k >= K) return;
    float acc = 0.f;
    int b = 0;
    for (; b + 8 <= B; b += 8) {    // eight loads in flight; the sum keeps the order b = 0, 1, 2, ...
        float xv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xv[j] = x[(size_t)(b + j) * K + k];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += dy[(size_t)(b + j) * O + o] * xv[j];
    }
    for (; b < B; ++b) acc += dy[(size_t)b * O + o] * x[(size_t)b * K + k];
    dw[(size_t)o * K + k] = acc;
}

__global__ void k_linear_db(const float* __restrict__ dy, float* __restrict__ db, int B, int O) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= O) return;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += dy[(size_t)b * O + o];
    db[o] = acc;
}

}  // namespace

extern "C" {

int osi_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int O, osi_stream_t stream) {
    OSI_REQUIRE(x && w && y && B > 0 && K > 0 && O > 0 && B < 65536);
    hipLaunchKernelGGL(k_linear_fwd, dim3(osi_cdiv(O, 4), B), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, B, K, O);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

int osi_linear_bwd(const float* dy, const float* x, const float* w, float* dx, int dx_accumulate, float* dw, float* db, int B,
                   int K, int O, osi_stream_t stream) {
    OSI_REQUIRE(dy && x && w && B > 0 && K > 0 && O > 0 && B < 65536 && O < 65536);
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        hipLaunchKernelGGL(k_linear_dx, dim3(osi_cdiv(K, 256), B), dim3(256), 0, st, dy, w, dx, B, K, O, dx_accumulate);
        OSI_LAUNCH_CHECK();
    }
    if (dw) {
        hipLaunchKernelGGL(k_linear_dw, dim3(osi_cdiv(K, 256), O), dim3(256), 0, st, dy, x, dw, B, K, O);
        OSI_LAUNCH_CHECK();
    }
    if (db) {
        hipLaunchKernelGGL(k_linear_db, dim3(osi_cdiv(O, 64)), dim3(64), 0, st, dy, db, B, O);
        OSI_LAUNCH_CHECK();
    }
    return OSI_OK;
}

}  // extern "C"
