#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_atom(unsigned long long* slots, int NT, int G, int C) {
    int mt = blockIdx.x / NT, nt = blockIdx.x % NT;
    int tid = threadIdx.x;
    if (tid < 64) {
        int g = mt % G, col = nt * 64 + tid;
        atomicAdd(&slots[(size_t)(g * C + col) * 2], (unsigned long long)(tid + mt));
        atomicAdd(&slots[(size_t)(g * C + col) * 2 + 1], (unsigned long long)(tid * 3 + mt));
    }
}
__global__ void k_plain(unsigned long long* out, int NT, int C) {
    int mt = blockIdx.x / NT, nt = blockIdx.x % NT;
    int tid = threadIdx.x;
    if (tid < 64) { size_t o = ((size_t)mt * C + nt * 64 + tid) * 2; out[o] = tid + mt; out[o + 1] = tid * 3 + mt; }
}
int main() {
    const int MT = 6272;
    for (int NT : {1, 4}) for (int G : {32, 8}) {
        int C = NT * 64;
        unsigned long long *slots, *out;
        hipMalloc(&slots, (size_t)G * C * 16); hipMemset(slots, 0, (size_t)G * C * 16);
        hipMalloc(&out, (size_t)MT * C * 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0); for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_atom, dim3(MT * NT), dim3(256), 0, 0, slots, NT, G, C);
            hipEventRecord(e1); hipEventSynchronize(e1); float ta; hipEventElapsedTime(&ta, e0, e1);
            hipEventRecord(e0); for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_plain, dim3(MT * NT), dim3(256), 0, 0, out, NT, C);
            hipEventRecord(e1); hipEventSynchronize(e1); float tp; hipEventElapsedTime(&tp, e0, e1);
            if (rep) printf("NT=%d G=%d: atomics %.1f us/launch, plain partial stores %.1f us/launch\n", NT, G, ta / 20 * 1e3, tp / 20 * 1e3);
        }
        hipFree(slots); hipFree(out);
    }
    return 0;
}
