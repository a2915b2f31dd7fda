// MFMA fragment helpers shared by the kernels that feed v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_bf16 from
// "16 contiguous k per lane" fragments (attention.hip, gcn_fused.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fira {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// One 32-deep product of two fragments (16 values per lane, k = (lane>>5)*16 + s on both sides) accumulated into a
// 32x32 tile.  fp32: the exact 16-step v_mfma_f32_32x32x2_f32 chain.  BF (bf16 mode of the engine: torch.autocast runs
// the attention matmuls on bf16 operands with fp32 accumulation): both fragments are rounded to bf16 (RNE) and the
// product is TWO v_mfma_f32_32x32x16_bf16 -- slot e of lane group g holds k = g*16 + 8 j + e in step j on both sides, so
// every k is paired once; the dependent chain shrinks from 16 x 64 to 2 x 32 cycles.
typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 abf16x2 __attribute__((ext_vector_type(2)));
typedef float af32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t apk2(float a, float b) {
    const af32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, abf16x2));      // v_cvt_pk_bf16_f32
}
template <bool BF>
__device__ __forceinline__ f32x16 chain16(const float (&a)[16], const float (&b)[16], f32x16 acc) {
    if constexpr (BF) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uint4 ua, ub;
            ua.x = apk2(a[8 * j], a[8 * j + 1]); ua.y = apk2(a[8 * j + 2], a[8 * j + 3]);
            ua.z = apk2(a[8 * j + 4], a[8 * j + 5]); ua.w = apk2(a[8 * j + 6], a[8 * j + 7]);
            ub.x = apk2(b[8 * j], b[8 * j + 1]); ub.y = apk2(b[8 * j + 2], b[8 * j + 3]);
            ub.z = apk2(b[8 * j + 4], b[8 * j + 5]); ub.w = apk2(b[8 * j + 6], b[8 * j + 7]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(abf16x8, ua), __builtin_bit_cast(abf16x8, ub), acc,
                                                          0, 0, 0);
        }
    } else {
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = MFMA32(a[s], b[s], acc);
    }
    return acc;
}

__device__ __forceinline__ int acc_row(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

}  // namespace fira
