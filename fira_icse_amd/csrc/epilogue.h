// Output step shared by the GEMM kernels: NV values of ONE output column (the 16 rows a lane holds of a 32x32 MFMA tile,
// or the 4 a thread combines in the latency kernels) -> C, with the optional bias / accumulate / ReLU / ReLU-mask /
// row-map / atomic (split-K) semantics of fira_gemm.
//
// Written so that NO memory instruction sits behind a per-element branch.  hipcc inserts s_waitcnt from one in-order
// counter (vmcnt) for loads AND stores; a load inside `if (accum) v += *p;` or `c_rows ? c_rows[row] : row` makes every
// join wait for vmcnt(0), i.e. each of the 16 stores of a lane waited for the previous store's acknowledgement --
// 16 store round trips (~10 us) at the end of every tile.  Here all optional reads are batched under wave-uniform
// branches ahead of the stores, rows / columns outside the matrix are handled by buffer-descriptor range checking
// (their byte offset gets bit 31: reads return 0, stores and atomics are dropped), and the stores stream back to back.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

namespace fira {

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32v4_t __attribute__((__vector_size__(16)));     // operand type of the 128-bit buffer builtins
constexpr int FIRA_BUF_FLAGS = 0x00020000;                  // gfx9 raw buffer descriptor, 32-bit data format
constexpr unsigned FIRA_OOB = 0x80000000u;                  // byte offset of a lane that must not touch memory
__device__ __forceinline__ rsrc_t buf_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, FIRA_BUF_FLAGS);
}
__device__ __forceinline__ float buf_load_f32(rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

// byte offsets are 31-bit: a plain output must stay below 2 GiB (host-side check of every launcher)
inline bool epilogue_fits(long rows, long ldc) { return rows * ldc * 4 < (1L << 31); }

// Optional closing step of a post-LN residual block's product (gnn_transformer.py:161,174): the stored value is the PRE-NORM
// sum  s = dropout(acc + bias) + res  -- what add_layernorm_fwd would form from the plain product -- so that the LayerNorm
// itself can run in the prologue of the kernel that consumes it (gemm_tile32_kernel<.., LN_A>).  Dropout element index =
// row * 256 + col, as in the row kernel (the backward kernels re-derive the same mask).
struct EpiRes {
    const float* res = nullptr;      // [M, ldr] residual rows (nullptr: feature off)
    int ldr = 0;
    float p = 0.f, inv_keep = 1.f;
    uint64_t seed = 0;
    uint32_t site = 0;
    uint32_t idx0 = 0;               // dropout element index of C[0,0]: the rows of a launch may be a slice of the site's rows
};

template <int NV>
__device__ __forceinline__ void epilogue_col(const float (&acc)[NV], const int (&row)[NV], int col, int M, int N,
                                             float* __restrict__ C, int ldc, const float* __restrict__ bias, bool relu,
                                             bool accum, bool atomic, const int32_t* __restrict__ c_rows,
                                             const float* __restrict__ relu_mask, const EpiRes er = EpiRes()) {
    const rsrc_t rC = buf_rsrc(C, 0x7fffffffu);
    const bool colok = col < N;
    const unsigned cb = (unsigned)col * 4u;
    const float bv = buf_load_f32(buf_rsrc(bias ? (const void*)bias : (const void*)C, bias ? (unsigned)N * 4u : 0u), cb);
    unsigned off[NV], moff[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const bool ok = colok && row[i] < M;
        moff[i] = ok ? (unsigned)row[i] * (unsigned)ldc * 4u + cb : FIRA_OOB;
        off[i] = moff[i];
    }
    if (c_rows) {                                           // wave-uniform
        const rsrc_t rR = buf_rsrc(c_rows, (unsigned)M * 4u);
        unsigned mapped[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) mapped[i] = __builtin_amdgcn_raw_buffer_load_b32(rR, (unsigned)row[i] * 4u, 0, 0);
#pragma unroll
        for (int i = 0; i < NV; ++i) off[i] = (moff[i] & FIRA_OOB) ? FIRA_OOB : mapped[i] * (unsigned)ldc * 4u + cb;
    }
    if (atomic) {                                           // split-K partial sums
#pragma unroll
        for (int i = 0; i < NV; ++i) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[i] + bv, rC, off[i], 0, 0);
        return;
    }
    float old[NV], keep[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { old[i] = 0.f; keep[i] = 1.f; }
    if (accum) {
#pragma unroll
        for (int i = 0; i < NV; ++i) old[i] = buf_load_f32(rC, off[i]);
    }
    if (relu_mask) {                                        // ReLU backward fused into the dgrad: keep where the saved activation > 0
        const rsrc_t rM = buf_rsrc(relu_mask, 0x7fffffffu);
#pragma unroll
        for (int i = 0; i < NV; ++i) keep[i] = buf_load_f32(rM, moff[i]) > 0.f ? 1.f : 0.f;
    }
    if (er.res) {                                           // wave-uniform: pre-norm sum of a residual block
        const rsrc_t rR = buf_rsrc(er.res, 0x7fffffffu);
        float rv[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i)
            rv[i] = buf_load_f32(rR, (moff[i] & FIRA_OOB) ? FIRA_OOB : (unsigned)row[i] * (unsigned)er.ldr * 4u + cb);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float v = acc[i] + bv;
            if (er.p > 0.f) v *= dropout_scale(er.seed, er.site, er.idx0 + (uint32_t)row[i] * FIRA_D + (uint32_t)col, er.p, er.inv_keep);
            v += rv[i];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rC, off[i], 0, 0);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float v = (acc[i] + bv) + old[i];
        v = relu ? fmaxf(v, 0.f) : v;
        v = keep[i] != 0.f ? v : 0.f;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rC, off[i], 0, 0);
    }
}

}  // namespace fira
