// Open-Set Classification Rate curve on the GPU (next-row f4 of SURVEY.md §8).
//
// Replaces util.calculate_oscr (reference openset_imagenet/util.py:90-122): for every threshold tau in the sorted unique
// target-class scores of the known samples except the largest,
//     ccr(tau) = #{known, argmax == label, score[label] > tau} / #known
//     fpr(tau) = #{label == unk_label, max score > tau}        / #{label == unk_label}
// The reference is a Python loop over tau with an O(N) numpy reduction inside (O(N_unique * N) host work on a host-resident score
// matrix). Here the score matrix stays on the device (validate()/get_arrays() produce it there) and everything is integer
// counting, so the result is bit-identical to the reference: the kernels emit the thresholds and two int64 counts per threshold,
// the two float64 divisions by the totals are left to the host mirror (numpy semantics for 0/0 included).
//
// All-pairs counting instead of a sort: N is a validation / test set (1e3 .. 1e5), N^2 comparisons are microseconds to
// milliseconds of VALU work, need no workspace beyond 3 arrays, and are order-free (deterministic) by construction.
#include "osi_common.h"

namespace {

constexpr int NT = 256;
enum : int { F_KNOWN = 1, F_UNK = 2, F_CORRECT = 4, F_FIRST = 8 };

// per sample: target score (known rows), max score, flags. argmax keeps the FIRST maximum (np.argmax, util.py:110).
template <typename T>
__global__ __launch_bounds__(NT) void k_oscr_rows(const T* __restrict__ scores, const long long* __restrict__ gt, int N, int C,
                                                 long long unk_label, T* __restrict__ tgt, T* __restrict__ mx, int* __restrict__ flags,
                                                 long long* totals) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= N) return;
    const T* row = scores + (size_t)i * C;
    T best = row[0];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
        const T v = row[c];
        if (v > best) { best = v; arg = c; }
    }
    const long long y = gt[i];
    int f = 0;
    T t = 0;
    if (y >= 0 && y < C) {
        f |= F_KNOWN;
        t = row[y];
        if (arg == (int)y) f |= F_CORRECT;
    }
    if (y == unk_label) f |= F_UNK;
    tgt[i] = t; mx[i] = best; flags[i] = f;
    if (f & F_KNOWN) atomicAdd((unsigned long long*)&totals[1], 1ull);   // integer atomics: order-free
    if (f & F_UNK) atomicAdd((unsigned long long*)&totals[2], 1ull);
}

// first occurrence of each distinct target score among the known samples (np.unique, util.py:114)
template <typename T>
__global__ __launch_bounds__(NT) void k_oscr_first(const T* __restrict__ tgt, int* flags, int N, long long* totals) {
    __shared__ T st[NT];
    __shared__ int sf[NT];
    const int i = blockIdx.x * NT + threadIdx.x;
    const bool act = i < N && (flags[i] & F_KNOWN);
    const T ti = i < N ? tgt[i] : T(0);
    bool dup = false;
    // only j < i matters: tiles up to this workgroup's own
    for (int j0 = 0; j0 <= (int)blockIdx.x * NT; j0 += NT) {
        const int j = j0 + threadIdx.x;
        st[threadIdx.x] = j < N ? tgt[j] : T(0);
        sf[threadIdx.x] = j < N ? (flags[j] & F_KNOWN) : 0;
        __syncthreads();
        if (act) {
            const int lim = min(NT, i - j0);      // j < i
            for (int k = 0; k < lim; ++k) dup |= sf[k] && st[k] == ti;
        }
        __syncthreads();
    }
    if (act && !dup) {
        // flags of other samples are only read for F_KNOWN (set by the previous kernel), so this in-place update does not race
        atomicOr(&flags[i], F_FIRST);
        atomicAdd((unsigned long long*)&totals[0], 1ull);
    }
}

// rank of each distinct score = number of distinct scores below it -> thresholds in ascending order
template <typename T>
__global__ __launch_bounds__(NT) void k_oscr_rank(const T* __restrict__ tgt, const int* __restrict__ flags, int N, T* __restrict__ taus) {
    __shared__ T st[NT];
    __shared__ int sf[NT];
    const int i = blockIdx.x * NT + threadIdx.x;
    const bool act = i < N && (flags[i] & F_FIRST);
    const T ti = i < N ? tgt[i] : T(0);
    int rank = 0;
    for (int j0 = 0; j0 < N; j0 += NT) {
        const int j = j0 + threadIdx.x;
        st[threadIdx.x] = j < N ? tgt[j] : T(0);
        sf[threadIdx.x] = j < N ? (flags[j] & F_FIRST) : 0;
        __syncthreads();
        if (act)
            for (int k = 0; k < NT; ++k) rank += (sf[k] && st[k] < ti) ? 1 : 0;
        __syncthreads();
    }
    if (act) taus[rank] = ti;
}

// counts per threshold; thresholds [0, n_unique - 1) are the curve (the largest distinct score is dropped, util.py:114)
template <typename T>
__global__ __launch_bounds__(NT) void k_oscr_counts(const T* __restrict__ tgt, const T* __restrict__ mx, const int* __restrict__ flags,
                                                   int N, const T* __restrict__ taus, const long long* __restrict__ totals,
                                                   long long* __restrict__ ccr_count, long long* __restrict__ fpr_count) {
    __shared__ T st[NT], sm[NT];
    __shared__ int sf[NT];
    const int u = blockIdx.x * NT + threadIdx.x;
    const long long npts = totals[0] - 1;
    const bool act = u < npts;
    const T tau = act ? taus[u] : T(0);
    long long c = 0, f = 0;
    if ((long long)blockIdx.x * NT >= npts) return;      // whole workgroup beyond the curve
    for (int j0 = 0; j0 < N; j0 += NT) {
        const int j = j0 + threadIdx.x;
        st[threadIdx.x] = j < N ? tgt[j] : T(0);
        sm[threadIdx.x] = j < N ? mx[j] : T(0);
        sf[threadIdx.x] = j < N ? flags[j] : 0;
        __syncthreads();
        if (act)
            for (int k = 0; k < NT; ++k) {
                const int fl = sf[k];
                c += ((fl & (F_KNOWN | F_CORRECT)) == (F_KNOWN | F_CORRECT) && st[k] > tau) ? 1 : 0;
                f += ((fl & F_UNK) && sm[k] > tau) ? 1 : 0;
            }
        __syncthreads();
    }
    if (act) { ccr_count[u] = c; fpr_count[u] = f; }
}

template <typename T>
int oscr_impl(const T* scores, const long long* gt, int N, int C, long long unk_label, void* ws, size_t ws_bytes, T* taus,
              long long* ccr_count, long long* fpr_count, long long* totals, osi_stream_t stream) {
    OSI_REQUIRE(scores && gt && ws && taus && ccr_count && fpr_count && totals && N > 0 && C > 0);
    OSI_REQUIRE(ws_bytes >= osi_oscr_workspace(N));
    hipStream_t st = (hipStream_t)stream;
    T* tgt = (T*)ws;                                  // sized for double
    T* mx = (T*)((char*)ws + (size_t)N * 8);
    int* flags = (int*)((char*)ws + (size_t)N * 16);
    if (hipMemsetAsync(totals, 0, 3 * sizeof(long long), st) != hipSuccess) return OSI_ERR_LAUNCH;
    const int grid = osi_cdiv(N, NT);
    hipLaunchKernelGGL(k_oscr_rows<T>, dim3(grid), dim3(NT), 0, st, scores, gt, N, C, unk_label, tgt, mx, flags, totals);
    OSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_oscr_first<T>, dim3(grid), dim3(NT), 0, st, (const T*)tgt, flags, N, totals);
    OSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_oscr_rank<T>, dim3(grid), dim3(NT), 0, st, (const T*)tgt, (const int*)flags, N, taus);
    OSI_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_oscr_counts<T>, dim3(grid), dim3(NT), 0, st, (const T*)tgt, (const T*)mx, (const int*)flags, N, (const T*)taus,
                       (const long long*)totals, ccr_count, fpr_count);
    OSI_LAUNCH_CHECK();
    return OSI_OK;
}

}  // namespace

extern "C" {

size_t osi_oscr_workspace(int N) { return N > 0 ? (size_t)N * 20 + 64 : 0; }

int osi_oscr_f32(const float* scores, const long long* gt, int N, int C, long long unk_label, void* ws, size_t ws_bytes,
                 float* taus, long long* ccr_count, long long* fpr_count, long long* totals, osi_stream_t stream) {
    return oscr_impl<float>(scores, gt, N, C, unk_label, ws, ws_bytes, taus, ccr_count, fpr_count, totals, stream);
}
int osi_oscr_f64(const double* scores, const long long* gt, int N, int C, long long unk_label, void* ws, size_t ws_bytes,
                 double* taus, long long* ccr_count, long long* fpr_count, long long* totals, osi_stream_t stream) {
    return oscr_impl<double>(scores, gt, N, C, unk_label, ws, ws_bytes, taus, ccr_count, fpr_count, totals, stream);
}

}  // extern "C"
