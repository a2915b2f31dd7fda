// Shared device/host helpers for libfira_hip (gfx950 only; wavefront = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/fira_hip.h"

#define FIRA_D 256          // model width every row-kernel is specialised for
#define FIRA_DH 32          // head width

namespace fira {

// thread-local last-error message (fira_last_error)
char* err_buf();
int set_err(const char* fmt, ...);

#define FIRA_CHECK_LAUNCH(what)                                                   \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) return fira::set_err("%s: %s", what, hipGetErrorString(e__)); \
    } while (0)

#define FIRA_REQUIRE(cond, ...)                                                   \
    do {                                                                          \
        if (!(cond)) return fira::set_err(__VA_ARGS__);                           \
    } while (0)

// ---------------------------------------------------------------- per-kernel-class event profiling
// Off by default (one relaxed bool test per launch).  When enabled (fira_prof_enable), every launcher brackets
// its kernel with two hipEvents on the launch stream; fira_prof_report synchronises and aggregates the elapsed
// time and the algorithmic work (FLOP for GEMM/attention, bytes for the memory-bound classes) per class.
// PROF_GEMM_DEC: the GEMM launches issued while a ProfDecoderTag is alive on the calling thread (the decoder's M = B*30 row
// products, forward and data gradients) -- the quantity north_star sets its MFMA target on; reported beside PROF_GEMM
// PROF_GCN: the fused GCN-layer launches (gcn_fused.hip): work = FLOP of the [rows,256]x[256,256] product, bytes = the
// algorithmic bytes of the whole layer (gather in, rows out)
// PROF_COMB: the fused Combination-block launches (comb_fused.hip): work = FLOP of its three [rows,256]x[256,256] products
enum ProfClass { PROF_GEMM = 0, PROF_SPMM, PROF_ATTN, PROF_ROWOPS, PROF_COPY, PROF_HEAD, PROF_ADAM, PROF_GEMM_DEC, PROF_GCN, PROF_COMB, PROF_DEC_REGION, PROF_NCLASS };
bool prof_on();
void prof_decoder_tag(int delta);      // +1 / -1 (nesting counter, thread-local)
struct ProfDecoderTag {
    ProfDecoderTag() { prof_decoder_tag(+1); }
    ~ProfDecoderTag() { prof_decoder_tag(-1); }
};
int prof_begin(hipStream_t s, int cls, double work, double bytes);
void prof_end(hipStream_t s, int idx);
struct ProfScope {
    hipStream_t s;
    bool on;
    int idx = -1;
    ProfScope(hipStream_t s_, int cls, double work, double bytes = 0.0) : s(s_), on(prof_on()) {
        if (on) idx = prof_begin(s, cls, work, bytes);
    }
    ~ProfScope() { if (on) prof_end(s, idx); }
};

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- counter-based dropout mask
// keep-mask bit for element `idx` of dropout site `site` under `seed`; the same function is
// evaluated in forward and backward, so no mask tensor is ever stored.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint32_t site, uint32_t idx, float p, float inv_keep) {
    uint32_t h = mix32(idx ^ mix32(site * 0x9E3779B9U + (uint32_t)seed) ^ (uint32_t)(seed >> 32));
    // uniform in [0,1): keep when u >= p
    float u = (float)(h >> 8) * (1.0f / 16777216.0f);
    return u >= p ? inv_keep : 0.0f;
}

// ---------------------------------------------------------------- wave-level reductions (64 lanes)
// Data-parallel-primitive (DPP) butterflies: a step is ONE VALU instruction whose operand comes from another lane of the same
// 16-lane row (quad permutes for lane^1 / lane^2, row_half_mirror / row_mirror for the other quad / the other half -- after the
// quad steps all lanes of a quad hold the same value, so a mirror is as good as an xor), then row_bcast:15 / row_bcast:31 carry
// the row sums into lane 63 and a readlane broadcasts it.  As __shfl_xor every step was a ds_bpermute_b32: an LDS-crossbar
// round trip (~100 cycles with its s_waitcnt) that the next step depends on -- 6 per reduction, 48 per wave in the row phase of
// the fused GCN kernel, ~190 per wave in the copy-score forward.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                                 ROW_MASK, 0xf, false));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_BCAST15 = 0x142, DPP_BCAST31 = 0x143,
              DPP_ROR8 = 0x128;          // row_ror:8 = lane ^ 8 inside a 16-lane row
// sum over the 8 lanes lane&~7 .. lane|7, in every one of them
__device__ __forceinline__ float sum8(float v) {
    v += dpp_take<DPP_XOR1, 0xf>(v, v);
    v += dpp_take<DPP_XOR2, 0xf>(v, v);
    v += dpp_take<DPP_HALF_MIRROR, 0xf>(v, v);
    return v;
}
// sum over the 16 lanes of a row, in every one of them
__device__ __forceinline__ float sum16(float v) {
    v = sum8(v);
    v += dpp_take<DPP_MIRROR, 0xf>(v, v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = sum16(v);
    v += dpp_take<DPP_BCAST15, 0xa>(0.f, v);           // rows 1, 3 += lane 15 of the row before
    v += dpp_take<DPP_BCAST31, 0xc>(0.f, v);           // rows 2, 3 += lane 31
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_take<DPP_XOR1, 0xf>(v, v));
    v = fmaxf(v, dpp_take<DPP_XOR2, 0xf>(v, v));
    v = fmaxf(v, dpp_take<DPP_HALF_MIRROR, 0xf>(v, v));
    v = fmaxf(v, dpp_take<DPP_MIRROR, 0xf>(v, v));
    v = fmaxf(v, dpp_take<DPP_BCAST15, 0xa>(v, v));    // (rows the mask leaves out keep their own value)
    v = fmaxf(v, dpp_take<DPP_BCAST31, 0xc>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ------------------------------------------------------------------------------------------------
// CombinationLayer (reference combination_layer.py:7-17): per element
//   a = q*k/sqrt(32), b = q*v/sqrt(32), (g0,g1) = softmax(a,b), c = g0*k + g1*v, then dropout(c).
// The head split/transposes of gnn_transformer.py:197-202 cancel.
// The two-way soft-max with the maximum subtracted has one exponential equal to exp(0) = 1: with e = exp(-|a - b|) the larger
// gate is 1 / (1 + e) and the smaller e / (1 + e) -- one v_exp_f32 and one v_rcp_f32 per element instead of two expf and an
// IEEE division (these kernels are bound by their VALU instruction stream, not by memory).
__device__ __forceinline__ void gate_elem(float q, float k, float v, float& g0, float& g1) {
    const float is = 1.0f / 5.656854249492381f;  // 1 / sqrt(32): reciprocal multiplies instead of divisions (<= 1 ulp)
    const float a = q * k * is, b = q * v * is;
    const float e = __builtin_amdgcn_exp2f(-fabsf(a - b) * 1.4426950408889634f);
    const float big = __builtin_amdgcn_rcpf(1.0f + e), small = e * big;
    g0 = a >= b ? big : small;
    g1 = a >= b ? small : big;
}

// Workgroup barrier for LDS traffic only.  __syncthreads() is s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier: it also waits for
// every outstanding GLOBAL load of the wave, i.e. for the weight fragments / next rows a kernel requests ahead of a barrier
// precisely so that they arrive under the phase behind it.  Only valid where no thread reads global memory that another
// thread of the workgroup wrote before the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace fira
