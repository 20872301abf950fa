// Fused optimizer steps over the flat fp32 parameter arena (one launch for all 162 tensors).
// Reference: torch.optim.Adam(lr) / torch.optim.SGD(lr, momentum=0.9) built at openset_imagenet/train.py:356-359 and
// stepped at train.py:139. Arithmetic follows torch's single-tensor rules:
//   Adam: m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
//   SGD : buf = g (first step) | mu*buf + g ; p -= lr*buf
// Also: arena utilities used by the loop (zero fill, int64 counter bump for num_batches_tracked, scale for DP averaging).
#include "osi_common.h"

namespace {

__global__ __launch_bounds__(256) void k_adam(f32x4* __restrict__ p, const f32x4* __restrict__ g, f32x4* __restrict__ m, f32x4* __restrict__ v,
                                             size_t n4, float step_size, float b2, float omb1, float omb2, float eps,
                                             float sqrt_bc2, float gscale) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    for (; i < n4; i += step) {
        f32x4 gg = g[i] * gscale, mm = m[i], vv = v[i], pp = p[i];
        mm = mm + (gg - mm) * omb1;              // torch: exp_avg.lerp_(grad, 1-beta1)
        vv = vv * b2 + gg * gg * omb2;         // torch: exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
        f32x4 den;
        den.x = sqrtf(vv.x) / sqrt_bc2 + eps; den.y = sqrtf(vv.y) / sqrt_bc2 + eps;
        den.z = sqrtf(vv.z) / sqrt_bc2 + eps; den.w = sqrtf(vv.w) / sqrt_bc2 + eps;
        pp = pp - (mm / den) * step_size;       // torch: param.addcdiv_(exp_avg, denom, value=-step_size)
        m[i] = mm; v[i] = vv; p[i] = pp;
    }
}

__global__ __launch_bounds__(256) void k_sgd(f32x4* __restrict__ p, const f32x4* __restrict__ g, f32x4* __restrict__ buf, size_t n4, float lr,
                                            float mu, int first, float gscale) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    for (; i < n4; i += step) {
        f32x4 gg = g[i] * gscale;
        f32x4 b = first ? gg : buf[i] * mu + gg;
        buf[i] = b;
        p[i] = p[i] - b * lr;
    }
}

__global__ __launch_bounds__(256) void k_fill(f32x4* __restrict__ p, size_t n4, float val) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    for (; i < n4; i += step) p[i] = f32x4{val, val, val, val};
}
__global__ __launch_bounds__(256) void k_scale(f32x4* __restrict__ p, size_t n4, float s) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    for (; i < n4; i += step) p[i] = p[i] * s;
}
__global__ void k_i64_add(long long* p, int n, long long inc) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += inc;
}

static int sgrid(size_t n4) {
    size_t g = (n4 + 255) / 256;
    return (int)(g > 2048 ? 2048 : (g ? g : 1));
}

}  // namespace

extern "C" {

int osi_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1,
                  double beta2, double eps, long long step, float grad_scale, osi_stream_t stream) {
    OSI_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && n % 4 == 0 && step >= 1);
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(k_adam, dim3(sgrid(n / 4)), dim3(256), 0, (hipStream_t)stream, (f32x4*)param, (const f32x4*)grad,
                       (f32x4*)exp_avg, (f32x4*)exp_avg_sq, n / 4, (float)(lr / bc1), (float)beta2, (float)(1.0 - beta1),
                       (float)(1.0 - beta2), (float)eps, (float)sqrt(bc2), grad_scale);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

int osi_sgd_step(float* param, const float* grad, float* momentum_buf, size_t n, float lr, float momentum, int first_step,
                 float grad_scale, osi_stream_t stream) {
    OSI_REQUIRE(param && grad && momentum_buf && n > 0 && n % 4 == 0);
    hipLaunchKernelGGL(k_sgd, dim3(sgrid(n / 4)), dim3(256), 0, (hipStream_t)stream, (f32x4*)param, (const f32x4*)grad,
                       (f32x4*)momentum_buf, n / 4, lr, momentum, first_step, grad_scale);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

int osi_fill_f32(float* p, size_t n, float value, osi_stream_t stream) {
    OSI_REQUIRE(p && n > 0 && n % 4 == 0);
    hipLaunchKernelGGL(k_fill, dim3(sgrid(n / 4)), dim3(256), 0, (hipStream_t)stream, (f32x4*)p, n / 4, value);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}
int osi_scale_f32(float* p, size_t n, float s, osi_stream_t stream) {
    OSI_REQUIRE(p && n > 0 && n % 4 == 0);
    hipLaunchKernelGGL(k_scale, dim3(sgrid(n / 4)), dim3(256), 0, (hipStream_t)stream, (f32x4*)p, n / 4, s);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}
int osi_i64_add(long long* p, int n, long long inc, osi_stream_t stream) {
    OSI_REQUIRE(p && n > 0);
    hipLaunchKernelGGL(k_i64_add, dim3(osi_cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, p, n, inc);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

}  // extern "C"
