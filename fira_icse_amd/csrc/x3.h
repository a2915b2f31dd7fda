// Three-term bf16 split of fp32 operands for the fused encoder kernels (round 6; gcn_fused.hip, comb_fused.hip).
//   x = hi + mid + lo, each the bf16 (RNE) rounding of what the previous terms left; a product keeps six of the nine term products
//   (lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi -- smallest first, fp32 accumulation in the MFMA): the dropped ones are below 2^-24
//   of the product, so the result is as accurate as an fp32 chain at 96 instead of 256 matrix-core cycles per 16 x 16 x 32 block.
// LDS panels hold the A operand as three bf16 PLANES of [rows][256 k] (512-byte rows, the 16-byte columns XOR-swizzled by row & 15:
// the 16 lanes of a ds_read_b128 group read 16 distinct bank quads); weights arrive pre-split in fragment order (gcn_split_planes).
#pragma once
#include "mfma_frag.h"

namespace fira {

constexpr size_t GX_WPLANE = (size_t)FIRA_D * FIRA_D * 2;          // bytes of one bf16 plane of a [256, 256] weight
// byte offset of 16-byte column c16 (0..31: eight bf16 k values) of plane row `row`
__device__ __forceinline__ int gx_off(int row, int c16) { return row * 512 + ((c16 ^ (row & 15)) << 4); }
__device__ __forceinline__ uint32_t gx_pack(float a, float b) {
    const af32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, abf16x2));      // v_cvt_pk_bf16_f32 (RNE)
}
// what is left of the low / high element of a pair after its bf16 rounding in `pk` (exact in fp32)
__device__ __forceinline__ float gx_rest_lo(float a, uint32_t pk) { return a - __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float gx_rest_hi(float b, uint32_t pk) { return b - __builtin_bit_cast(float, pk & 0xffff0000u); }
// four consecutive k values of a row (a lane's float4 of a whole-row access: 16-byte column lane >> 1, half lane & 1) -> the planes
// (NP = 1: the bf16 mode of the engine -- one plane, the operand rounded once)
template <int NP = 3>
__device__ __forceinline__ void gx_store_row4(char* planes, size_t plane_bytes, int row, int lane, f32x4v u) {
    const int o = gx_off(row, lane >> 1) + (lane & 1) * 8;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
        const uint32_t p0 = gx_pack(u.x, u.y), p1 = gx_pack(u.z, u.w);
        *reinterpret_cast<uint2*>(planes + pl * plane_bytes + o) = uint2{p0, p1};
        if (pl + 1 < NP) { u.x = gx_rest_lo(u.x, p0); u.y = gx_rest_hi(u.y, p0); u.z = gx_rest_lo(u.z, p1); u.w = gx_rest_hi(u.w, p1); }
    }
}
// one element (row, k) -> the planes (the accumulator layout's scalar stores)
template <int NP = 3>
__device__ __forceinline__ void gx_store_elem(char* planes, size_t plane_bytes, int row, int k, float u) {
    const int o = gx_off(row, k >> 3) + (k & 7) * 2;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
        const uint32_t p0 = gx_pack(u, 0.f);
        *reinterpret_cast<uint16_t*>(planes + pl * plane_bytes + o) = (uint16_t)p0;
        if (pl + 1 < NP) u = gx_rest_lo(u, p0);
    }
}
// this lane's fragment base of a plane: row l15 of a tile, 16-byte column 4 s + kq of k step s  ->  byte (a_q ^ (s << 6)) + tile * 8192
__device__ __forceinline__ int gx_frag_base(int l15, int kq) { return l15 * 512 + (((kq ^ (l15 & 3)) << 4) | ((l15 >> 2) << 6)); }
// this lane's byte offset into a weight's planes: unit ((16-column block w) * 8 + k step s) * 64 + lane, 16 bytes each
__device__ __forceinline__ unsigned gx_wlane(int wave, int lane) { return (unsigned)(wave * 8 * 64 + lane) * 16u; }
// the term products of one k step of a tile: fragments from NP planes of the panel at `pa` (plane stride `pb` bytes), NP weight units b[]
template <int NP, typename ACC>
__device__ __forceinline__ void gx_terms(ACC& acc, const char* pa, size_t pb, const uint4 (&b)[3]) {
    const abf16x8 ah = *reinterpret_cast<const abf16x8*>(pa);
    const abf16x8 bh = __builtin_bit_cast(abf16x8, b[0]);
    if constexpr (NP == 3) {
        const abf16x8 am = *reinterpret_cast<const abf16x8*>(pa + pb);
        const abf16x8 al = *reinterpret_cast<const abf16x8*>(pa + 2 * pb);
        const abf16x8 bm = __builtin_bit_cast(abf16x8, b[1]), bl = __builtin_bit_cast(abf16x8, b[2]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
}
// six term products of one k step
#define FIRA_X3_MFMA(acc, ah, am, al, bh, bm, bl)                              \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);       \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);

}  // namespace fira
