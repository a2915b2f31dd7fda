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
enum ProfClass { PROF_GEMM = 0, PROF_SPMM, PROF_ATTN, PROF_ROWOPS, PROF_COPY, PROF_HEAD, PROF_ADAM, PROF_GEMM_DEC, PROF_GCN, PROF_NCLASS };
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
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace fira
