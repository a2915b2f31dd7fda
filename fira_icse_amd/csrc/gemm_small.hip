// Latency-oriented fp32 MFMA GEMM for the skinny shapes of the decoder (M = B*30 rows in training, M = B*beam rows in
// the decode step; K = 256 / 768 / 1024): C[M,N] = A[M,K] . op(B) (+bias)(relu)(+C).
//
// The LDS-tiled kernel walks K serially behind one barrier per 32-wide tile, so a [960,256]x[256,256] product costs
// ~12-15 us although it is 0.13 GFLOP: it is bound by the dependent chain (8 tiles x (load latency + 16 MFMAs)), not
// by throughput.  Here one workgroup owns one 32x32 output tile and its 4 wavefronts split K among themselves
// (chunks of 32 round-robin); every wave fetches its operand fragments straight into registers in MFMA layout (all
// loads of a wave are in flight together: one memory latency), runs 16 MFMAs per chunk, and the four partial tiles
// are combined through LDS.  No split-K atomics, no pre-zeroed output, 4x shorter dependent chain.
//
// Fragment layout (as attention.hip): inside a 32-wide K chunk, MFMA step s uses k = (lane>>5)*16 + s, so a
// K-contiguous operand row gives each lane 16 contiguous floats (4 x 16-byte loads).
//   B_KCONTIG = true   B is an nn.Linear weight [N,K]  (forward)
//   B_KCONTIG = false  B is [K,N]                      (dgrad): 16 row-segment loads of 128 bytes per half-wave
#include "engine.h"
#include "epilogue.h"
#include <stdlib.h>

namespace fira {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool B_KCONTIG>
__global__ __launch_bounds__(256) void gemm_small_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                         const float* __restrict__ B, int ldb, float* __restrict__ C,
                                                         int ldc, const float* __restrict__ bias, int flags,
                                                         const int32_t* __restrict__ c_rows,
                                                         const float* __restrict__ relu_mask) {
    __shared__ float red[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const float* pa = A + (size_t)min(m0 + l31, M - 1) * lda + kh * 16;      // clamped rows: masked at the store
    const float* pb = B_KCONTIG ? B + (size_t)min(n0 + l31, N - 1) * ldb + kh * 16
                                : B + (size_t)(kh * 16) * ldb + min(n0 + l31, N - 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nchunk = K / 32;
    for (int c = wave; c < nchunk; c += 8) {
        // two chunks per trip: 16 loads per operand in flight before the first MFMA
        float a[2][16], b[2][16];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int cc = c + 4 * u;
            if (cc < nchunk) {
                const float4* qa = reinterpret_cast<const float4*>(pa + cc * 32);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = qa[q];
                    a[u][4 * q] = v.x; a[u][4 * q + 1] = v.y; a[u][4 * q + 2] = v.z; a[u][4 * q + 3] = v.w;
                }
                if (B_KCONTIG) {
                    const float4* qb = reinterpret_cast<const float4*>(pb + cc * 32);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = qb[q];
                        b[u][4 * q] = v.x; b[u][4 * q + 1] = v.y; b[u][4 * q + 2] = v.z; b[u][4 * q + 3] = v.w;
                    }
                } else {
                    const float* qb = pb + (size_t)cc * 32 * ldb;
#pragma unroll
                    for (int s = 0; s < 16; ++s) b[u][s] = qb[(size_t)s * ldb];
                }
            } else {
#pragma unroll
                for (int s = 0; s < 16; ++s) { a[u][s] = 0.f; b[u][s] = 0.f; }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][s], b[u][s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r * 64 + lane] = acc[r];
    __syncthreads();
    // thread t combines elements idx = t + 256 i of the tile: one column (lane & 31), four rows (epilogue.h)
    {
        const int ln = threadIdx.x & 63, rq = threadIdx.x >> 6;
        float vals[4];
        int rows[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = threadIdx.x + 256 * i, r = rq + 4 * i;
            vals[i] = (red[0][idx] + red[1][idx]) + (red[2][idx] + red[3][idx]);
            rows[i] = m0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        }
        epilogue_col<4>(vals, rows, n0 + (ln & 31), M, N, C, ldc, bias, flags & FIRA_GEMM_RELU, flags & FIRA_GEMM_ACCUM, false,
                        c_rows, relu_mask);
    }
}

// true if this kernel took the call
bool gemm_small_try(hipStream_t s, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                    float* C, int ldc, const float* bias, int flags, int* rc, const int32_t* c_rows,
                    const float* relu_mask) {
    *rc = 0;
    static const int mode = [] { const char* e = getenv("FIRA_SMALL_GEMM"); return e ? atoi(e) : 1; }();   // A/B switch
    if (mode == 0 && M > 64) return false;          // 0: only the decode-sized products (M <= 64) stay here
    // beyond ~4 rounds of 32x32 tiles the LDS-tiled kernel's operand reuse wins; M <= 64 (decode) always lands here
    const long tiles = (long)cdiv(M, 32) * cdiv(N, 32);
    if (tA || (M > 64 && tiles > 1024) || K % 32 != 0 || K < 64 || lda % 4 != 0 || ((uintptr_t)A % 16) != 0) return false;
    if (tB && (ldb % 4 != 0 || ((uintptr_t)B % 16) != 0)) return false;
    dim3 grid(cdiv(N, 32), cdiv(M, 32));
    if (tB) hipLaunchKernelGGL(gemm_small_kernel<true>, grid, dim3(256), 0, s, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, c_rows, relu_mask);
    else hipLaunchKernelGGL(gemm_small_kernel<false>, grid, dim3(256), 0, s, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, c_rows, relu_mask);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) *rc = set_err("gemm_small: %s", hipGetErrorString(e));
    return true;
}

}  // namespace fira
