// Stream kernels around the conv stack: NCHW image -> NHWC4 staging, MaxPool(3,2,1) fwd/bwd, global average
// pool fwd/bwd. They replace torchvision ResNet.maxpool / ResNet.avgpool / torch.flatten under
// openset_imagenet/model.py:37 (reference) and the implicit layout of the input batch (train.py:128).
#include "osi_common.h"

namespace {

// [B][3][H][W] -> [B][H][W][4], 4th channel zero (the stem conv treats the image as 4-channel, K = 49*4).
__global__ __launch_bounds__(256) void k_nchw3_to_nhwc4(const float* __restrict__ x, f32x4* __restrict__ y, int B, int HW) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)B * HW, step = (size_t)gridDim.x * 256;
    for (; i < n; i += step) {
        size_t b = i / HW, p = i - b * HW;
        const float* s = x + b * 3 * HW + p;
        y[i] = f32x4{s[0], s[HW], s[2 * (size_t)HW], 0.f};
    }
}

// MaxPool 3x3 stride 2 pad 1, NHWC. idx = position (0..8) of the first maximum in row-major window order
// (torch's tie rule: strictly-greater replaces), stored as one byte per output element.
__global__ __launch_bounds__(256) void k_maxpool_fwd(const f32x4* __restrict__ x, f32x4* __restrict__ y, uint32_t* __restrict__ idx,
                                                    int B, int H, int W, int C4, int Ho, int Wo) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)B * Ho * Wo * C4;
    if (i >= n) return;
    int c4 = (int)(i % C4);
    size_t t = i / C4;
    int wo = (int)(t % Wo); t /= Wo;
    int ho = (int)(t % Ho);
    int b = (int)(t / Ho);
    const float NEG = -__builtin_inff();
    f32x4 best = {NEG, NEG, NEG, NEG};
    uint32_t bi[4] = {0, 0, 0, 0};
    bool first[4] = {true, true, true, true};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            int h = ho * 2 - 1 + r, w = wo * 2 - 1 + s;
            if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
                f32x4 v = x[((size_t)(b * H + h) * W + w) * C4 + c4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (first[k] || v[k] > best[k]) { best[k] = v[k]; bi[k] = r * 3 + s; first[k] = false; }
            }
        }
    y[i] = best;
    idx[i] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
}

// Stem tail in one pass: pooled = maxpool3x3s2(relu(y * scale + shift)). The post-ReLU activation (the largest tensor of the
// network, 411 MB at B = 128) is never written; bit 7 of each index byte records "maximum > 0", the only ReLU-gate information the
// backward needs (gradient reaches a pixel only through a window whose maximum it is).
__global__ __launch_bounds__(256) void k_bn_relu_maxpool_fwd(const f32x4* __restrict__ y, const f32x4* __restrict__ scale,
                                                            const f32x4* __restrict__ shift, f32x4* __restrict__ pooled,
                                                            uint32_t* __restrict__ idx, int B, int H, int W, int C4, int Ho, int Wo) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)B * Ho * Wo * C4;
    if (i >= n) return;
    int c4 = (int)(i % C4);
    size_t t = i / C4;
    int wo = (int)(t % Wo); t /= Wo;
    int ho = (int)(t % Ho);
    int b = (int)(t / Ho);
    const f32x4 sc = scale[c4], sh = shift[c4];
    const float NEG = -__builtin_inff();
    f32x4 best = {NEG, NEG, NEG, NEG};
    uint32_t bi[4] = {0, 0, 0, 0};
    bool first[4] = {true, true, true, true};
    // the nine window loads go out together (coordinates clamped into the image, validity applied afterwards): with the load inside
    // the bounds test every tap was its own basic block and its own memory round trip
    f32x4 win[9];
    bool ok[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int h = ho * 2 - 1 + r, w = wo * 2 - 1 + s;
            ok[r * 3 + s] = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
            const int hc = min(max(h, 0), H - 1), wc = min(max(w, 0), W - 1);
            win[r * 3 + s] = y[((size_t)(b * H + hc) * W + wc) * C4 + c4];
        }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        f32x4 v = win[j] * sc + sh;      // same expression as k_bn_apply
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (ok[j] && (first[k] || v[k] > best[k])) { best[k] = v[k]; bi[k] = j; first[k] = false; }
    }
    pooled[i] = best;
#pragma unroll
    for (int k = 0; k < 4; ++k) bi[k] |= best[k] > 0.f ? 0x80u : 0u;
    idx[i] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
}

__global__ __launch_bounds__(256) void k_maxpool_bwd(const f32x4* __restrict__ dy, const uint32_t* __restrict__ idx, f32x4* __restrict__ dx,
                                                    int B, int H, int W, int C4, int Ho, int Wo) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)B * H * W * C4;
    if (i >= n) return;
    int c4 = (int)(i % C4);
    size_t t = i / C4;
    int w = (int)(t % W); t /= W;
    int h = (int)(t % H);
    int b = (int)(t / H);
    f32x4 acc = {0, 0, 0, 0};
    // windows containing (h, w): ho*2-1 <= h <= ho*2+1
    int ho_lo = h >> 1, ho_hi = (h + 1) >> 1;  // floor(h/2) .. floor((h+1)/2)
    int wo_lo = w >> 1, wo_hi = (w + 1) >> 1;
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
        if (ho >= Ho) continue;
        int r = h - (ho * 2 - 1);
        if (r < 0 || r > 2) continue;
        for (int wo = wo_lo; wo <= wo_hi; ++wo) {
            if (wo >= Wo) continue;
            int s = w - (wo * 2 - 1);
            if (s < 0 || s > 2) continue;
            size_t o = ((size_t)(b * Ho + ho) * Wo + wo) * C4 + c4;
            uint32_t id = idx[o];
            f32x4 g = dy[o];
            uint32_t me = r * 3 + s;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (((id >> (8 * k)) & 0xff) == me) acc[k] += g[k];
        }
    }
    dx[i] = acc;
}

// global average pool: [B][HW][C] -> [B][C]
__global__ __launch_bounds__(256) void k_avgpool_fwd(const f32x4* __restrict__ x, f32x4* __restrict__ y, int B, int HW, int C4) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * C4) return;
    int b = i / C4, c4 = i - b * C4;
    f32x4 s = {0, 0, 0, 0};
    const f32x4* xb = x + (size_t)b * HW * C4 + c4;
    int p = 0;
    for (; p + 7 <= HW; p += 7) {   // seven loads in flight (HW = 49 at 224x224); the sum keeps the pixel order
        f32x4 v[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) v[j] = xb[(size_t)(p + j) * C4];
#pragma unroll
        for (int j = 0; j < 7; ++j) s += v[j];
    }
    for (; p < HW; ++p) s += xb[(size_t)p * C4];
    y[i] = s / (float)HW;
}
__global__ __launch_bounds__(256) void k_avgpool_bwd(const f32x4* __restrict__ dy, f32x4* __restrict__ dx, int B, int HW, int C4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)B * HW * C4;
    if (i >= n) return;
    int c4 = (int)(i % C4);
    int b = (int)(i / ((size_t)HW * C4));
    dx[i] = dy[(size_t)b * C4 + c4] / (float)HW;
}

// uint8 [B][H][W][3] (a decoded, cropped RGB batch) -> fp32 [B][H][W][4]: ToTensor()'s value / 255 (a true division, like
// torch's .div(255)), optional per-image horizontal flip, 4th channel zero. One pass replaces ToTensor + RandomHorizontalFlip of
// the reference's transform (train.py:259-263) and the NCHW -> NHWC4 staging; the host link carries 1 byte per value, not 4.
__global__ __launch_bounds__(256) void k_u8hwc3_to_nhwc4(const unsigned char* __restrict__ x, const unsigned char* __restrict__ flip,
                                                        f32x4* __restrict__ y, int B, int H, int W) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)B * H * W, step = (size_t)gridDim.x * 256;
    for (; i < n; i += step) {
        const size_t row = i / W;                       // b * H + h
        const int w = (int)(i - row * W);
        const int b = (int)(row / H);
        const int ws = (flip && flip[b]) ? W - 1 - w : w;
        const unsigned char* s = x + (row * W + ws) * 3;
        y[i] = f32x4{(float)s[0] / 255.0f, (float)s[1] / 255.0f, (float)s[2] / 255.0f, 0.f};
    }
}

// uint8 canvas [B][Hc][Wc][3] (the decoded, Resize(256)'d image or a window of it, NOT yet cropped) -> fp32 [B][H][W][4]:
// RandomCrop / CenterCrop (per-image top-left corner crop[b] = (x0, y0) inside the canvas), RandomHorizontalFlip (of the CROPPED
// image, as transforms.Compose orders them at train.py:259-263) and ToTensor's value / 255 in one pass; 4th channel zero.
__global__ __launch_bounds__(256) void k_u8_crop_flip_to_nhwc4(const unsigned char* __restrict__ x, const int* __restrict__ crop,
                                                              const unsigned char* __restrict__ flip, f32x4* __restrict__ y,
                                                              int B, int Hc, int Wc, int H, int W) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)B * H * W, step = (size_t)gridDim.x * 256;
    for (; i < n; i += step) {
        const size_t row = i / W;                       // b * H + h
        const int w = (int)(i - row * W);
        const int b = (int)(row / H);
        const int h = (int)(row - (size_t)b * H);
        // corners arrive from the host loader: clamped HERE, in registers, into [0, Wc - W] x [0, Hc - H] so that a bad offset can
        // never read outside the canvas; the caller's tensor is read-only
        const int x0 = crop ? min(max(crop[2 * b], 0), Wc - W) : 0, y0 = crop ? min(max(crop[2 * b + 1], 0), Hc - H) : 0;
        const int ws = x0 + ((flip && flip[b]) ? W - 1 - w : w);
        const unsigned char* s = x + (((size_t)b * Hc + (y0 + h)) * Wc + ws) * 3;
        y[i] = f32x4{(float)s[0] / 255.0f, (float)s[1] / 255.0f, (float)s[2] / 255.0f, 0.f};
    }
}
}  // namespace

extern "C" {

int osi_u8_crop_flip_to_nhwc4(const unsigned char* canvas, const int* crop_xy, const unsigned char* flip, float* y, int B, int Hc, int Wc,
                              int H, int W, osi_stream_t stream) {
    OSI_REQUIRE(canvas && y && B > 0 && H > 0 && W > 0 && Hc >= H && Wc >= W);
    hipStream_t st = (hipStream_t)stream;
    size_t n = (size_t)B * H * W;
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_u8_crop_flip_to_nhwc4, dim3(grid), dim3(256), 0, st, canvas, crop_xy, flip, (f32x4*)y, B, Hc, Wc, H, W);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

int osi_u8hwc3_to_nhwc4(const unsigned char* x, const unsigned char* flip, float* y, int B, int H, int W, osi_stream_t stream) {
    OSI_REQUIRE(x && y && B > 0 && H > 0 && W > 0);
    size_t n = (size_t)B * H * W;
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_u8hwc3_to_nhwc4, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, flip, (f32x4*)y, B, H, W);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

int osi_nchw3_to_nhwc4(const float* x, float* y, int B, int H, int W, osi_stream_t stream) {
    OSI_REQUIRE(x && y && B > 0 && H > 0 && W > 0);
    size_t n = (size_t)B * H * W;
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_nchw3_to_nhwc4, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (f32x4*)y, B, H * W);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

int osi_maxpool3x3s2_fwd(const float* x, float* y, void* idx, int B, int H, int W, int C, osi_stream_t stream) {
    OSI_REQUIRE(x && y && idx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
    int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    size_t n = (size_t)B * Ho * Wo * (C / 4);
    OSI_REQUIRE(n < (1ul << 31));
    hipLaunchKernelGGL(k_maxpool_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)x, (f32x4*)y,
                       (uint32_t*)idx, B, H, W, C / 4, Ho, Wo);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}
int osi_bn_relu_maxpool_fwd(const float* y, const float* scale, const float* shift, float* pooled, void* idx, int B, int H, int W,
                            int C, osi_stream_t stream) {
    OSI_REQUIRE(y && scale && shift && pooled && idx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
    int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    size_t n = (size_t)B * Ho * Wo * (C / 4);
    OSI_REQUIRE(n < (1ul << 31));
    hipLaunchKernelGGL(k_bn_relu_maxpool_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)y,
                       (const f32x4*)scale, (const f32x4*)shift, (f32x4*)pooled, (uint32_t*)idx, B, H, W, C / 4, Ho, Wo);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}
int osi_maxpool3x3s2_bwd(const float* dy, const void* idx, float* dx, int B, int H, int W, int C, osi_stream_t stream) {
    OSI_REQUIRE(dy && dx && idx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
    int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    size_t n = (size_t)B * H * W * (C / 4);
    OSI_REQUIRE(n < (1ul << 31));
    hipLaunchKernelGGL(k_maxpool_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)dy,
                       (const uint32_t*)idx, (f32x4*)dx, B, H, W, C / 4, Ho, Wo);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}
int osi_avgpool_fwd(const float* x, float* y, int B, int HW, int C, osi_stream_t stream) {
    OSI_REQUIRE(x && y && B > 0 && HW > 0 && C > 0 && C % 4 == 0);
    hipLaunchKernelGGL(k_avgpool_fwd, dim3(osi_cdiv((long)B * C / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)x,
                       (f32x4*)y, B, HW, C / 4);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}
int osi_avgpool_bwd(const float* dy, float* dx, int B, int HW, int C, osi_stream_t stream) {
    OSI_REQUIRE(dy && dx && B > 0 && HW > 0 && C > 0 && C % 4 == 0);
    size_t n = (size_t)B * HW * (C / 4);
    hipLaunchKernelGGL(k_avgpool_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)dy,
                       (f32x4*)dx, B, HW, C / 4);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

}  // extern "C"
