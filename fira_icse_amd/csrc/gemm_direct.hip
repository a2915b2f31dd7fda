// LDS-free fp32 MFMA GEMM for the short-K shapes of this model (K = 256 / 768 / 1024 / 3072).
//
// Why: v_mfma_f32_32x32x2_f32 runs at the fp32 vector rate (64 cycles per instruction), so an output tile needs only
// ~16 FLOP per operand byte to keep the matrix pipe busy -- little enough that every wavefront can fetch its own
// operand fragments straight from global memory / L2 into registers in MFMA layout.  That removes the LDS staging,
// the transposing stores and, above all, the workgroup barrier per K tile that dominated the LDS-tiled kernel on
// K = 256 (PMC: 46 % of wave cycles parked in s_waitcnt/s_barrier, MFMA pipe 30 % busy).  Wavefronts are fully
// independent; the hardware overlaps the loads of one with the MFMAs of another.
//
// Fragment trick (same as attention.hip): the order of the reduction index inside an MFMA chain is free, so for a
// 32-wide K chunk step s of the chain uses k = (lane>>5)*16 + s.  A K-contiguous operand row then feeds a lane with
// 16 CONTIGUOUS floats (4 x 16-byte loads), and the two half-waves read the two halves of the same 128-byte line.
//   layout NT:  C[M,N] = A[M,K] . W[N,K]^T   (forward)      both operands K-contiguous
//   layout NN:  C[M,N] = A[M,K] . B[K,N]     (dgrad)        B rows are fetched as 128-byte row segments per k
// Each wave owns a (32*TM) x (32*TN) output tile; a workgroup is 4 waves stacked along M (they share the B
// fragments through L1).  Operand fragments of chunk c+1 are loaded while the MFMAs of chunk c run.
#include "engine.h"

namespace fira {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int TM, int TN, bool B_KCONTIG>
__global__ __launch_bounds__(256, 2) void gemm_direct_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                          const float* __restrict__ B, int ldb, float* __restrict__ C,
                                                          int ldc, const float* __restrict__ bias, int flags) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int m0 = (blockIdx.y * 4 + wave) * 32 * TM;
    const int n0 = blockIdx.x * 32 * TN;
    if (m0 >= M) return;

    // row pointers (clamped: out-of-range rows read a valid row and are masked at the store)
    const float* pa[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) pa[i] = A + (size_t)min(m0 + i * 32 + l31, M - 1) * lda + kh * 16;
    const float* pb[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        if (B_KCONTIG) pb[j] = B + (size_t)min(n0 + j * 32 + l31, N - 1) * ldb + kh * 16;
        else pb[j] = B + (size_t)(kh * 16) * ldb + min(n0 + j * 32 + l31, N - 1);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float a[2][TM][16], b[2][TN][16];
    auto fetch = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float4* p = reinterpret_cast<const float4*>(pa[i] + k0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = p[q];
                a[buf][i][4 * q + 0] = v.x; a[buf][i][4 * q + 1] = v.y; a[buf][i][4 * q + 2] = v.z; a[buf][i][4 * q + 3] = v.w;
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (B_KCONTIG) {
                const float4* p = reinterpret_cast<const float4*>(pb[j] + k0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = p[q];
                    b[buf][j][4 * q + 0] = v.x; b[buf][j][4 * q + 1] = v.y; b[buf][j][4 * q + 2] = v.z; b[buf][j][4 * q + 3] = v.w;
                }
            } else {
                const float* p = pb[j] + (size_t)k0 * ldb;
#pragma unroll
                for (int s = 0; s < 16; ++s) b[buf][j][s] = p[(size_t)s * ldb];
            }
        }
    };

    const int nchunk = K / 32;
    fetch(0, 0);
#pragma unroll 1
    for (int c = 0; c < nchunk; c += 2) {
        if (c + 1 < nchunk) fetch(1, (c + 1) * 32);
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][i][s], b[0][j][s], acc[i][j], 0, 0, 0);
        if (c + 1 < nchunk) {
            if (c + 2 < nchunk) fetch(0, (c + 2) * 32);
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][i][s], b[1][j][s], acc[i][j], 0, 0, 0);
        }
    }

    const bool relu = flags & FIRA_GEMM_RELU, accum = flags & FIRA_GEMM_ACCUM;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + j * 32 + l31;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row >= M) continue;
                float v = acc[i][j][r] + bv;
                float* p = C + (size_t)row * ldc + col;
                if (accum) v += *p;
                if (relu) v = fmaxf(v, 0.f);
                *p = v;
            }
    }
}

// true if the direct kernel handled the call
bool gemm_direct_try(hipStream_t s, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                     float* C, int ldc, const float* bias, int flags, int* rc) {
    *rc = 0;
    if (tA || K % 32 != 0 || K > 4096 || lda % 4 != 0 || ((uintptr_t)A % 16) != 0) return false;
    if (tB && (ldb % 4 != 0 || ((uintptr_t)B % 16) != 0)) return false;
    // wave tile: 64x64 when that still gives >= ~2 waves per SIMD (2048 waves), else smaller tiles for parallelism
    const long w22 = (long)cdiv(M, 64) * cdiv(N, 64), w12 = (long)cdiv(M, 32) * cdiv(N, 64);
#define GO(TM, TN)                                                                                                  \
    do {                                                                                                            \
        dim3 grid(cdiv(N, 32 * TN), cdiv(cdiv(M, 32 * TM), 4));                                                     \
        if (tB) hipLaunchKernelGGL((gemm_direct_kernel<TM, TN, true>), grid, dim3(256), 0, s, M, N, K, A, lda, B, ldb, C, \
                                   ldc, bias, flags);                                                               \
        else hipLaunchKernelGGL((gemm_direct_kernel<TM, TN, false>), grid, dim3(256), 0, s, M, N, K, A, lda, B, ldb, C,   \
                                ldc, bias, flags);                                                                  \
    } while (0)
    if (tB) {
        if (w22 >= 1536) GO(2, 2);
        else if (w12 >= 1024) GO(1, 2);
        else GO(1, 1);
    } else {
        // row-segment fetches of the [K,N] operand cost 16 addresses per fragment: only the small tile stays in
        // registers; large dgrad shapes keep the LDS-tiled kernel
        if (w12 >= 1024) return false;
        GO(1, 1);
    }
#undef GO
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) *rc = set_err("gemm_direct: %s", hipGetErrorString(e));
    return true;
}

}  // namespace fira
