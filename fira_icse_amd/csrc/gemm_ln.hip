// y = LayerNorm(dropout(A W^T + b) + residual): the closing step of every block of the reference
// (`self.layernorm(self.dropout(self.fc_o(x)) + residual)`, gnn_transformer.py:84-86, 159-161, 172-174, 203-205)
// as ONE kernel for the model width N = 256.
//
// The unfused sequence costs a GEMM launch, a kernel boundary (~5 us on an 8-XCD part: L2 write-back + dispatch) and
// a row kernel that re-reads the GEMM output from HBM.  Here a workgroup owns 32 complete rows: 8 wavefronts x (32
// rows x 32 columns) of v_mfma_f32_32x32x2_f32, operand fragments straight from global/L2 to registers in MFMA layout
// (as gemm_small.hip; the 8 waves read the same A rows, which the CU's vector L1 serves), the partial tiles meet in
// LDS (a second group of 8 waves takes the other half of K when K >= 1024), and the row phase of add_layernorm_fwd
// runs on the LDS tile: bias, dropout (same counter-based mask as the unfused kernel), residual, mean / variance,
// gamma / beta; it writes the pre-norm sum and the row statistics the backward pass needs, and y (optionally
// through a row map, the encoder's code-row scatter).
#include "engine.h"

namespace fira {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GL_PITCH = 260;      // floats per LDS tile row: 16-byte aligned rows, 4-bank skew between rows

__device__ __forceinline__ void gl_load(float (&a)[16], float (&b)[16], const float* pa, const float* pb) {
    const float4* qa = reinterpret_cast<const float4*>(pa);
    const float4* qb = reinterpret_cast<const float4*>(pb);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 u = qa[q], v = qb[q];
        a[4 * q] = u.x; a[4 * q + 1] = u.y; a[4 * q + 2] = u.z; a[4 * q + 3] = u.w;
        b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
    }
}

template <int KS>
__global__ __launch_bounds__(512 * KS) void gemm_ln_kernel(int M, int K, const float* __restrict__ A, int lda,
                                                           const float* __restrict__ W, int ldw,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ res,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           float* __restrict__ sum_out, float* __restrict__ y,
                                                           const int32_t* __restrict__ y_rows,
                                                           float* __restrict__ stats, float p, float inv_keep,
                                                           uint64_t seed, uint32_t site) {
    __shared__ __attribute__((aligned(16))) float tile[32 * GL_PITCH];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cg = wave & 7, kg = wave >> 3;                 // column group (32 columns), K group
    const int l31 = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.x * 32;
    const int kper = K / KS, nchunk = kper / 32;
    // inside a 32-wide K chunk MFMA step s uses k = kh*16 + s: 16 contiguous floats per lane and operand
    const float* pa = A + (size_t)min(m0 + l31, M - 1) * lda + kg * kper + kh * 16;   // clamped rows: masked below
    const float* pb = W + (size_t)(cg * 32 + l31) * ldw + kg * kper + kh * 16;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a0[16], b0[16], a1[16], b1[16];
    gl_load(a0, b0, pa, pb);
    for (int c = 0; c < nchunk; c += 2) {                    // chunk c+1 / c+2 in flight under the MFMAs of c / c+1
        if (c + 1 < nchunk) gl_load(a1, b1, pa + (c + 1) * 32, pb + (c + 1) * 32);
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc, 0, 0, 0);
        if (c + 2 < nchunk) gl_load(a0, b0, pa + (c + 2) * 32, pb + (c + 2) * 32);
        if (c + 1 < nchunk) {
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc, 0, 0, 0);
        }
    }
    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int g = 0; g < KS; ++g) {
        if (kg == g) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* q = &tile[((r & 3) + 8 * (r >> 2) + 4 * kh) * GL_PITCH + cg * 32 + l31];
                *q = (g == 0) ? acc[r] : *q + acc[r];
            }
        }
        __syncthreads();
    }
    // row phase (== add_layernorm_fwd_kernel on the tile)
    const float4 g4 = *reinterpret_cast<const float4*>(gamma + lane * 4);
    const float4 be4 = *reinterpret_cast<const float4*>(beta + lane * 4);
    const float4 bi4 = bias ? *reinterpret_cast<const float4*>(bias + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int rr = wave; rr < 32; rr += 8 * KS) {
        const int r = m0 + rr;
        if (r >= M) break;
        float4 a = *reinterpret_cast<const float4*>(&tile[rr * GL_PITCH + lane * 4]);
        a.x += bi4.x; a.y += bi4.y; a.z += bi4.z; a.w += bi4.w;
        const size_t o = (size_t)r * FIRA_D + lane * 4;
        if (p > 0.f) {
            const uint32_t e0 = (uint32_t)r * FIRA_D + lane * 4;
            a.x *= dropout_scale(seed, site, e0 + 0, p, inv_keep);
            a.y *= dropout_scale(seed, site, e0 + 1, p, inv_keep);
            a.z *= dropout_scale(seed, site, e0 + 2, p, inv_keep);
            a.w *= dropout_scale(seed, site, e0 + 3, p, inv_keep);
        }
        if (res) {
            const float4 b = *reinterpret_cast<const float4*>(res + o);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        const float mean = wave_sum(a.x + a.y + a.z + a.w) * (1.0f / FIRA_D);
        const float dx = a.x - mean, dy = a.y - mean, dz = a.z - mean, dw = a.w - mean;
        const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / FIRA_D);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        if (sum_out) *reinterpret_cast<float4*>(sum_out + o) = a;
        const size_t oy = y_rows ? (size_t)y_rows[r] * FIRA_D + lane * 4 : o;
        *reinterpret_cast<float4*>(y + oy) = make_float4(dx * rstd * g4.x + be4.x, dy * rstd * g4.y + be4.y,
                                                         dz * rstd * g4.z + be4.z, dw * rstd * g4.w + be4.w);
        if (stats && lane == 0) {
            stats[2 * r] = mean;
            stats[2 * r + 1] = rstd;
        }
    }
}

// true if the fused kernel took the call (otherwise the caller runs linear + add_layernorm_fwd)
bool linear_layernorm_try(hipStream_t s, int M, int K, const float* A, int lda, const float* W, const float* bias,
                          const float* res, const float* gamma, const float* beta, float* sum_out, float* y,
                          const int32_t* y_rows, float* stats, float dropout, uint64_t seed, uint32_t site, int* rc) {
    *rc = 0;
    if (M <= 0) return true;
    const int ks = K >= 1024 ? 2 : 1;
    if (K % (32 * ks) != 0 || lda % 4 != 0 || ((uintptr_t)A % 16) != 0 || ((uintptr_t)W % 16) != 0) return false;
    ProfScope prof(s, PROF_GEMM, 2.0 * M * (double)FIRA_D * K);
    const float inv_keep = dropout > 0.f ? 1.0f / (1.0f - dropout) : 1.0f;
    if (ks == 1)
        hipLaunchKernelGGL(gemm_ln_kernel<1>, dim3(cdiv(M, 32)), dim3(512), 0, s, M, K, A, lda, W, K, bias, res, gamma, beta,
                           sum_out, y, y_rows, stats, dropout, inv_keep, seed, site);
    else
        hipLaunchKernelGGL(gemm_ln_kernel<2>, dim3(cdiv(M, 32)), dim3(1024), 0, s, M, K, A, lda, W, K, bias, res, gamma,
                           beta, sum_out, y, y_rows, stats, dropout, inv_keep, seed, site);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) *rc = set_err("gemm_ln: %s", hipGetErrorString(e));
    return true;
}

}  // namespace fira

extern "C" int fira_linear_layernorm_fwd(void* stream, int M, int K, const float* A, int lda, const float* W,
                                         const float* bias, const float* res, const float* gamma, const float* beta,
                                         float* sum_out, float* y, float* stats, float dropout, uint64_t seed,
                                         uint32_t stream_id) {
    FIRA_REQUIRE(M >= 0 && K > 0 && A && W && gamma && beta && y, "fira_linear_layernorm_fwd: bad arguments");
    int rc;
    if (fira::linear_layernorm_try((hipStream_t)stream, M, K, A, lda, W, bias, res, gamma, beta, sum_out, y, nullptr, stats,
                                   dropout, seed, stream_id, &rc))
        return rc;
    FIRA_REQUIRE(sum_out != nullptr, "fira_linear_layernorm_fwd: this shape needs sum_out as scratch for the unfused path");
    rc = fira::gemm_f32_ex((hipStream_t)stream, 0, 1, M, FIRA_D, K, A, lda, W, K, sum_out, FIRA_D, bias, 0, 0, nullptr);
    if (rc) return rc;
    return fira::add_layernorm_fwd((hipStream_t)stream, M, sum_out, res, gamma, beta, y, stats, dropout, seed, stream_id,
                                   nullptr);
}
