// Fused forward+backward of the open-set training losses on [B, C] logits (fp32) and int64 labels.
//
// Reference semantics (file:line in /root/reference):
//   OSI_LOSS_ENTROPIC  openset_imagenet/losses.py:16-29  EntropicOpensetLoss: soft targets (one-hot for y >= 0, w/C for
//                      every negative label), CrossEntropyLoss() mean over ALL B rows.
//   OSI_LOSS_SOFTMAX   openset_imagenet/train.py:343     CrossEntropyLoss(ignore_index=-1): mean over non-ignored rows;
//                      all rows ignored -> 0/0 = NaN (kept).
//   OSI_LOSS_GARBAGE   openset_imagenet/train.py:344-347 CrossEntropyLoss(weight=class_weights): weighted mean.
//   objectosphere term (absent from the snapshot; SURVEY.md §8 a9, build-defined): + alpha/B * sum_i r_i^2,
//                      r_i = max(xi - |f_i|, 0) for y >= 0, |f_i| otherwise.
// With p = softmax(z):  dJ/dz_i = coef_i * p - t_i  (closed forms of SURVEY.md Appendix B).
//
// One workgroup of 16 waves handles the whole batch: a wave per row (lanes stride over C), two sweeps
// separated by a workgroup barrier because the normaliser (count / sum of class weights) is a batch quantity.
// Fixed summation order -> bitwise reproducible loss; no host synchronisation, no [B,C] target matrix.
#include "osi_common.h"

namespace {

constexpr int LOSS_THREADS = 1024;
constexpr int LOSS_WAVES = LOSS_THREADS / 64;

struct LossP {
    const float* z; const long long* y; const float* cw; const float* feat;
    float* loss; float* dz; float* dfeat;
    int B, C, F, mode;
    float unk_w; long long ignore_index; float xi, alpha;
};

__global__ __launch_bounds__(LOSS_THREADS) void k_loss(LossP p) {
    extern __shared__ float lse_s[];  // [B]
    __shared__ float red_num[LOSS_WAVES], red_den[LOSS_WAVES], red_obj[LOSS_WAVES];
    __shared__ float s_den;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C = p.C;
    float num = 0.f, den = 0.f, obj = 0.f;

    for (int i = wave; i < p.B; i += LOSS_WAVES) {
        const float* zr = p.z + (size_t)i * C;
        const long long yi = p.y[i];
        float mx = -__builtin_inff();
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, zr[c]);
        mx = wave_max(mx);
        float se = 0.f, sz = 0.f;
        for (int c = lane; c < C; c += 64) { float v = zr[c]; se += expf(v - mx); sz += v; }
        se = wave_sum(se); sz = wave_sum(sz);
        const float lse = mx + logf(se);
        if (lane == 0) lse_s[i] = lse;
        const bool lab_ok = yi >= 0 && yi < C;
        const float zy = lab_ok ? zr[yi] : 0.f;
        if (p.mode == OSI_LOSS_ENTROPIC) {
            if (yi >= 0) num += lab_ok ? lse - zy : 0.f;
            else num += p.unk_w * lse - (p.unk_w / (float)C) * sz;
        } else if (p.mode == OSI_LOSS_SOFTMAX) {
            if (lab_ok && yi != p.ignore_index) { num += lse - zy; den += 1.f; }
        } else {
            if (lab_ok) { float w = p.cw[yi]; num += w * (lse - zy); den += w; }
        }
        if (p.feat) {
            const float* fr = p.feat + (size_t)i * p.F;
            float ss = 0.f;
            for (int c = lane; c < p.F; c += 64) ss += fr[c] * fr[c];
            ss = wave_sum(ss);
            const float nrm = sqrtf(ss);
            float r, g;  // residual and d(r^2)/d|f|
            if (yi >= 0) { r = fmaxf(p.xi - nrm, 0.f); g = -2.f * r; }
            else { r = nrm; g = 2.f * r; }
            obj += r * r;
            const float k = nrm > 0.f ? p.alpha / (float)p.B * g / nrm : 0.f;
            for (int c = lane; c < p.F; c += 64) p.dfeat[(size_t)i * p.F + c] = k * fr[c];
        }
    }
    if (lane == 0) { red_num[wave] = num; red_den[wave] = den; red_obj[wave] = obj; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float n = 0.f, d = 0.f, o = 0.f;
        for (int k = 0; k < LOSS_WAVES; ++k) { n += red_num[k]; d += red_den[k]; o += red_obj[k]; }
        if (p.mode == OSI_LOSS_ENTROPIC) d = (float)p.B;
        s_den = d;
        float J = n / d;
        if (p.feat) J += p.alpha * o / (float)p.B;
        *p.loss = J;
    }
    __syncthreads();
    if (!p.dz) return;
    const float inv = 1.0f / s_den;

    for (int i = wave; i < p.B; i += LOSS_WAVES) {
        const float* zr = p.z + (size_t)i * C;
        float* dr = p.dz + (size_t)i * C;
        const long long yi = p.y[i];
        const float lse = lse_s[i];
        const bool lab_ok = yi >= 0 && yi < C;
        float coef, tval = 0.f, tuni = 0.f;  // grad = coef*p - (c==y ? tval : 0) - tuni
        if (p.mode == OSI_LOSS_ENTROPIC) {
            if (yi >= 0) { coef = inv; tval = inv; }
            else { coef = p.unk_w * inv; tuni = (p.unk_w / (float)C) * inv; }
        } else if (p.mode == OSI_LOSS_SOFTMAX) {
            const bool use = lab_ok && yi != p.ignore_index;
            coef = use ? inv : 0.f; tval = coef;
        } else {
            const float w = lab_ok ? p.cw[yi] : 0.f;
            coef = w * inv; tval = coef;
        }
        for (int c = lane; c < C; c += 64) {
            float g = coef * expf(zr[c] - lse) - tuni;
            if (lab_ok && c == (int)yi) g -= tval;
            dr[c] = g;
        }
    }
}

// softmax over rows (validation path, openset_imagenet/train.py:177)
__global__ __launch_bounds__(256) void k_softmax(const float* __restrict__ z, float* __restrict__ out, int B, int C) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= B) return;
    const float* zr = z + (size_t)i * C;
    float mx = -__builtin_inff();
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, zr[c]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += expf(zr[c] - mx);
    se = wave_sum(se);
    for (int c = lane; c < C; c += 64) out[(size_t)i * C + c] = expf(zr[c] - mx) / se;
}

// Validation confidences (reference openset_imagenet/metrics.py:8-42, called from train.py:187-192), accumulated on the device
// over the batches of an epoch instead of materialising the [N_val, C] score matrix: acc[0] += sum of softmax(z)[y] over known
// rows (y >= 0, y != unknown_class), acc[1] += their count, acc[2] += sum over rows with y == unknown_class of
// (1 + offset - max_{c < n_valid} softmax(z)[c]), acc[3] += their count. One workgroup, fixed order, double accumulators.
// SCORES: z already holds softmax scores (metrics.confidence's own argument) instead of logits.
template <bool SCORES>
__global__ __launch_bounds__(LOSS_THREADS) void k_confidence(const float* __restrict__ z, const long long* __restrict__ y, int B, int C,
                                                            float offset, long long unknown_class, int n_valid, double* acc) {
    __shared__ double red[4][LOSS_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double ks = 0, kc = 0, ns = 0, nc = 0;
    for (int i = wave; i < B; i += LOSS_WAVES) {
        const float* zr = z + (size_t)i * C;
        const long long yi = y[i];
        float mx = -__builtin_inff(), mv = -__builtin_inff();
        for (int c = lane; c < C; c += 64) { float v = zr[c]; mx = fmaxf(mx, v); if (c < n_valid) mv = fmaxf(mv, v); }
        mx = wave_max(mx); mv = wave_max(mv);
        if (SCORES) {
            if (yi == unknown_class) { ns += (double)(1.0f + offset - mv); nc += 1; }
            else if (yi >= 0 && yi < C) { ks += (double)zr[yi]; kc += 1; }
            continue;
        }
        float se = 0.f;
        for (int c = lane; c < C; c += 64) se += expf(zr[c] - mx);
        se = wave_sum(se);
        if (yi == unknown_class) { ns += (double)(1.0f + offset - expf(mv - mx) / se); nc += 1; }
        else if (yi >= 0 && yi < C) { ks += (double)(expf(zr[yi] - mx) / se); kc += 1; }
    }
    if (lane == 0) { red[0][wave] = ks; red[1][wave] = kc; red[2][wave] = ns; red[3][wave] = nc; }
    __syncthreads();
    if (threadIdx.x < 4) {
        double t = 0;
        for (int k = 0; k < LOSS_WAVES; ++k) t += red[threadIdx.x][k];
        acc[threadIdx.x] += t;
    }
}

}  // namespace

extern "C" {

int osi_confidence_accumulate(const float* logits, const long long* target, int B, int C, float offset, long long unknown_class,
                              int last_valid_class, double* acc4, osi_stream_t stream) {
    OSI_REQUIRE(logits && target && acc4 && B > 0 && C > 0);
    // python slicing scores[:, :last_valid_class]: None (encoded as 0 here) = all C columns, negative = C + last_valid_class
    const int n_valid = last_valid_class == 0 ? C : (last_valid_class < 0 ? C + last_valid_class : last_valid_class);
    OSI_REQUIRE(n_valid > 0 && n_valid <= C);
    hipLaunchKernelGGL(k_confidence<false>, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream, logits, target, B, C, offset,
                       unknown_class, n_valid, acc4);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

int osi_confidence_from_scores(const float* scores, const long long* target, int B, int C, float offset, long long unknown_class,
                               int last_valid_class, double* acc4, osi_stream_t stream) {
    OSI_REQUIRE(scores && target && acc4 && B > 0 && C > 0);
    const int n_valid = last_valid_class == 0 ? C : (last_valid_class < 0 ? C + last_valid_class : last_valid_class);
    OSI_REQUIRE(n_valid > 0 && n_valid <= C);
    hipLaunchKernelGGL(k_confidence<true>, dim3(1), dim3(LOSS_THREADS), 0, (hipStream_t)stream, scores, target, B, C, offset,
                       unknown_class, n_valid, acc4);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

int osi_loss_fwd_bwd(int mode, const float* logits, const long long* target, int B, int C, float unk_weight,
                     long long ignore_index, const float* class_weights, const float* features, int F, float xi, float alpha,
                     float* loss, float* dlogits, float* dfeatures, osi_stream_t stream) {
    OSI_REQUIRE(logits && target && loss && B > 0 && C > 0);
    OSI_REQUIRE(mode == OSI_LOSS_ENTROPIC || mode == OSI_LOSS_SOFTMAX || mode == OSI_LOSS_GARBAGE);
    OSI_REQUIRE(mode != OSI_LOSS_GARBAGE || class_weights);
    OSI_REQUIRE(!features || (dfeatures && F > 0));
    OSI_REQUIRE(B <= 15 * 1024);  // lse[B] lives in LDS
    LossP p;
    p.z = logits; p.y = target; p.cw = class_weights; p.feat = features;
    p.loss = loss; p.dz = dlogits; p.dfeat = dfeatures;
    p.B = B; p.C = C; p.F = F; p.mode = mode;
    p.unk_w = unk_weight; p.ignore_index = ignore_index; p.xi = xi; p.alpha = alpha;
    hipLaunchKernelGGL(k_loss, dim3(1), dim3(LOSS_THREADS), (size_t)B * sizeof(float), (hipStream_t)stream, p);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

int osi_softmax(const float* logits, float* out, int B, int C, osi_stream_t stream) {
    OSI_REQUIRE(logits && out && B > 0 && C > 0);
    hipLaunchKernelGGL(k_softmax, dim3(osi_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, logits, out, B, C);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

}  // extern "C"
