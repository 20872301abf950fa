// Device / host helpers shared by the convolution kernels (conv_igemm.hip) and the direct stem kernels (stem_direct.hip).
#pragma once
#include "osi_common.h"

namespace osi_conv {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// Buffer (SRSRC) loads: 32-bit byte offsets, and the hardware range check returns zeros for an offset >= num_records — the
// im2col zero padding costs one select on the OFFSET (sentinel OOB) instead of four on the data plus validity bookkeeping.
constexpr uint32_t OOB = 0x80000000u;   // every tensor here is < 2 GiB (desc_ok)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bld4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 bld4u(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// 16-byte store through the range check: an offset >= num_records (the OOB sentinel) is dropped
__device__ __forceinline__ void bst4(__amdgpu_buffer_rsrc_t r, f32x4 v, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x4acc __attribute__((ext_vector_type(4)));   // C/D of v_mfma_f32_16x16x4_f32
__device__ __forceinline__ f32x3 bld3(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(r, voff, soff, 0));
}

// Accumulator element (reg r of lane) -> row inside the 32x32 tile. Column = lane & 31.
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }


// internal host helpers implemented in conv_igemm.hip
int hw_cus();                        // CUs of the current device (256 when no device answers); no knob touches it
int chip_cus();                      // CUs the fwd / dgrad launch plans balance for: "tail_cus" if set, else hw_cus() - "dp_reserved_cus"
bool conv_desc_ok(const osi_conv_desc* d);
bool conv_is_stem(const osi_conv_desc* d);
// out[i] = sum over s of slab[s * stride4 + i] (float4 units), fixed order: bitwise reproducible
int launch_slab_reduce(const float* slab, float* out, size_t n4, size_t stride4, int splits, hipStream_t st);
// direct stem forward (stem_direct.hip): geometry test + launch; pmean / pm2 may be NULL
bool stem_direct_geometry(const osi_conv_desc* d);
int launch_stem_fwd_direct(const osi_conv_desc* d, const float* x4, const float* wpacked, float* y, float* pmean, float* pm2, int ntiles,
                           hipStream_t st);
constexpr int STEM_TILE_PIXELS = 128;   // 8 x 16 output pixels per tile

}  // namespace osi_conv
