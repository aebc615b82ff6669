// Shared by the weight-pack kernels (gemm_kernels.hip: pack_multi_kernel, gemm_v4.hip: pack4_kernel).
#pragma once
#include <hip/hip_runtime.h>

// The 15 taps of one dense (row, channel) pair are 60 contiguous bytes, 4-byte aligned: four 16-byte loads (the last one
// overlaps the third by a float, so nothing beyond the pair is touched) instead of 15 scalar ones -- every lane of a wave reads
// another cache line, so a pack is bound by the number of load instructions, not by bytes.
typedef float hmvae_f32x4u __attribute__((ext_vector_type(4), aligned(4)));
#ifdef P4_NARROW                               // timing A/B: the scalar loads of the first version
constexpr bool HMVAE_PACK_WIDE = false;
#else
constexpr bool HMVAE_PACK_WIDE = true;
#endif

__device__ __forceinline__ void hmvae_add_taps15(const float* __restrict__ src, float (&t)[16]) {
    const hmvae_f32x4u a = *(const hmvae_f32x4u*)src, b = *(const hmvae_f32x4u*)(src + 4), c = *(const hmvae_f32x4u*)(src + 8),
                       d = *(const hmvae_f32x4u*)(src + 11);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        t[k] += a[k];
        t[4 + k] += b[k];
        t[8 + k] += c[k];
    }
    t[12] += d[1];
    t[13] += d[2];
    t[14] += d[3];
}
