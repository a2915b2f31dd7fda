// bf16 product for the model's dominant shape class, K = 256 (= d_model): C[M,N] (+)= A[M,256] . Bb[N,256]^T
// (A fp32 activations rounded to bf16 while staged, Bb a k-contiguous bf16 weight shadow, fp32 accumulate and store).
// Every nn.Linear forward with a 256-wide input (gnn_transformer.py:76,82,142-144,159,172,199; Model.py:16-19,54) and
// every data gradient with a 256-wide output gradient has this shape.
//
// Why not the tiled kernel of gemm_bf16.hip: with K = 256 a 64x64 tile is four K steps -- a pipeline that is all fill
// and drain -- and every tile re-reads its 64 KiB fp32 A panel; [24000,256]x[256,256] ran at 2.2 TB/s of operand
// traffic (22 us against an 8 us HBM floor), [17000,256]x[3072,256]^T at 1.3 TB/s.  Here the A panel is the stationary
// operand:
//   * a workgroup (4 waves) owns BM = 64 or 128 rows and a range of 64-column tiles.  The panel is read from HBM ONCE
//     (coalesced 1 KiB rows), rounded, and parked in LDS just long enough for each wave to pull the 16 MFMA A-fragments of
//     its 32-row slice into 64 VGPRs, where they stay for the whole sweep;
//   * the weight tiles (64 columns x 256 k, 32 KiB of bf16, L2-resident) stream through two LDS buffers that reuse the
//     panel's staging space: tile j+1 is in flight in registers while the 16 (or 32) MFMAs of tile j run; one barrier per
//     column tile, no barrier inside a tile, the K loop is fully unrolled;
//   * LDS rows are padded to 528 B (132 dwords: row r starts at bank 4r mod 64), which makes both the 16-byte staging
//     stores (16 lanes = 256 contiguous bytes of one row) and the ds_read_b128 fragment fetches (16 lanes = 16 rows at one
//     k chunk) conflict-free without a swizzle;
//   * work item = (row panel, column range), panel-major, mapped to workgroups with the same contiguous-range-per-XCD
//     rule as the other GEMMs, so the column ranges of one panel (and the weight tiles all of them stream) share an L2.
// Algorithmic traffic per launch: 4*M*256 (A, once) + 2*N*256 (weights, once from HBM) + 4*M*N (C).
#include "engine.h"
#include <algorithm>
#include <stdlib.h>

namespace fira {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int PK = 256;              // the K this kernel is specialised for
constexpr int PPITCH = 528;          // LDS row pitch in bytes (256 bf16 + 16 B pad)
constexpr int PBN = 64;              // columns per streamed weight tile

__device__ __forceinline__ uint32_t pack2(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));      // v_cvt_pk_bf16_f32 (RNE)
}

// RT = 32-row slices per workgroup: 2 (BM 64: waves = 2 row slices x 2 column halves) or 4 (BM 128: wave = row slice,
// both 32-column halves of the tile).
template <int RT, bool ACCUM>
__global__ __launch_bounds__(256) void gemm_bf16_k256_kernel(int M, int N, const float* __restrict__ A, int lda,
                                                             const uint16_t* __restrict__ Bb, int ldb,
                                                             float* __restrict__ C, int ldc,
                                                             const float* __restrict__ bias, int relu, int n_chunks,
                                                             int tiles_per_chunk, int n_items, int chunk,
                                                             const int32_t* __restrict__ c_rows,
                                                             const float* __restrict__ relu_mask) {
    constexpr int BM = 32 * RT, CG = 4 / RT, SN = 2 / CG;
    constexpr int LDS_BYTES = (BM > 2 * PBN ? BM : 2 * PBN) * PPITCH;
    __shared__ __attribute__((aligned(16))) char sm[LDS_BYTES];

    const int item = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (item >= n_items) return;
    const int panel = item / n_chunks, ch = item - panel * n_chunks;
    const int m0 = panel * BM;
    const int n_tiles = (N + PBN - 1) / PBN;
    const int jt0 = ch * tiles_per_chunk, nt = min(tiles_per_chunk, n_tiles - jt0);

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int kc = t & 31, rs = t >> 5;                    // staging role: 16-byte chunk kc of rows rs + 8*i

    // ---- weight tile 0 on its way (registers) before anything else
    u32x4 breg[8];
    auto fetch_b = [&](int jt) __attribute__((always_inline)) {
        const int n0 = jt * PBN;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int col = min(n0 + rs + 8 * i, N - 1);   // clamped: columns past N are never stored
            breg[i] = *reinterpret_cast<const u32x4*>(Bb + (size_t)col * ldb + kc * 8);
        }
    };
    auto put_b = [&](int buf) __attribute__((always_inline)) {
        char* sb = sm + buf * PBN * PPITCH;
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(sb + (rs + 8 * i) * PPITCH + kc * 16) = breg[i];
    };
    fetch_b(jt0);

    // ---- A panel: HBM -> bf16 -> LDS, 64 rows per pass
#pragma unroll
    for (int pass = 0; pass < BM / 64; ++pass) {
        f32x4 v[8][2];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = min(m0 + pass * 64 + rs + 8 * i, M - 1);      // clamped: rows past M are never stored
            const float* q = A + (size_t)row * lda + kc * 8;
            v[i][0] = *reinterpret_cast<const f32x4*>(q);
            v[i][1] = *reinterpret_cast<const f32x4*>(q + 4);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            u32x4 w = {pack2(v[i][0].x, v[i][0].y), pack2(v[i][0].z, v[i][0].w), pack2(v[i][1].x, v[i][1].y),
                       pack2(v[i][1].z, v[i][1].w)};
            *reinterpret_cast<u32x4*>(sm + (pass * 64 + rs + 8 * i) * PPITCH + kc * 16) = w;
        }
    }
    __syncthreads();

    // ---- this wave's A fragments: 32 rows x 256 k in 64 VGPRs (MFMA step s takes k = 16 s + 8 (lane >> 5) .. + 7)
    const int rt = RT == 4 ? wave : (wave & 1), cg = RT == 4 ? 0 : (wave >> 1);
    bf16x8 a[16];
    {
        const char* sa = sm + (rt * 32 + l31) * PPITCH + kh * 16;
#pragma unroll
        for (int s = 0; s < 16; ++s) a[s] = *reinterpret_cast<const bf16x8*>(sa + s * 32);
    }
    __syncthreads();                                        // the staging space now belongs to the weight tiles
    put_b(0);
    __syncthreads();

    const int boff = (cg * 32 * SN + l31) * PPITCH + kh * 16;
    for (int j = 0; j < nt; ++j) {
        const int n0 = (jt0 + j) * PBN;
        // previous C values (accumulate mode) and the next weight tile are requested before the MFMAs start
        f32x16 acc[SN];
        const int row_base = m0 + rt * 32 + 4 * kh;
#pragma unroll
        for (int sn = 0; sn < SN; ++sn) {
            const int col = n0 + (cg * SN + sn) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = 0.f;
                if (ACCUM) {
                    const int row = min(row_base + (r & 3) + 8 * (r >> 2), M - 1);
                    v = C[(size_t)(c_rows ? c_rows[row] : row) * ldc + min(col, N - 1)];
                }
                acc[sn][r] = v;
            }
        }
        fetch_b(min(jt0 + j + 1, n_tiles - 1));             // unconditional: no load behind a branch (see gemm_bf16.hip)
        const char* sb = sm + (j & 1) * PBN * PPITCH + boff;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int sn = 0; sn < SN; ++sn) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(sb + sn * 32 * PPITCH + s * 32);
                acc[sn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b, acc[sn], 0, 0, 0);
            }
        }
        put_b((j + 1) & 1);                                 // buffer (j+1)&1 was last read in iteration j-1 (barrier since)
        // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int sn = 0; sn < SN; ++sn) {
            const int col = n0 + (cg * SN + sn) * 32 + l31;
            if (col < N) {
                const float bv = bias ? bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_base + (r & 3) + 8 * (r >> 2);
                    if (row < M) {
                        float v = acc[sn][r] + bv;
                        if (relu) v = fmaxf(v, 0.f);
                        if (relu_mask && !(relu_mask[(size_t)row * ldc + col] > 0.f)) v = 0.f;
                        C[(size_t)(c_rows ? c_rows[row] : row) * ldc + col] = v;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// true if this kernel took the call
bool gemm_bf16_k256_try(hipStream_t s, int M, int N, int K, const float* A, int lda, const uint16_t* Bb, int ldb, float* C,
                        int ldc, const float* bias, int flags, const int32_t* c_rows, const float* relu_mask, int* rc) {
    *rc = 0;
    static const int mode = [] { const char* e = getenv("FIRA_PANEL_GEMM"); return e ? atoi(e) : 1; }();   // A/B switch
    if (!mode || K != PK || M < 64 || N < 32 || lda % 4 || ldb % 8 || ((uintptr_t)A % 16) || ((uintptr_t)Bb % 16)) return false;
    // BM 128 halves the weight-tile traffic per output; it needs enough rows to still fill the chip
    const bool big = mode == 2 ? false : mode == 3 ? true : (N >= 512 && (long)M * N >= (4L << 20));
    const int bm = big ? 128 : 64;
    const int panels = cdiv(M, bm), n_tiles = cdiv(N, PBN);
    const int slots = 512;                                   // 2 resident workgroups per CU
    int n_chunks = std::max(1, std::min(n_tiles, slots / std::max(1, panels)));
    const int tpc = cdiv(n_tiles, n_chunks);
    n_chunks = cdiv(n_tiles, tpc);
    const int n_items = panels * n_chunks, chunk = cdiv(n_items, 8);
    const int relu = flags & FIRA_GEMM_RELU;
    const bool accum = flags & FIRA_GEMM_ACCUM;
#define FIRA_LAUNCH(RT, AC)                                                                                              \
    hipLaunchKernelGGL((gemm_bf16_k256_kernel<RT, AC>), dim3(8 * chunk), dim3(256), 0, s, M, N, A, lda, Bb, ldb, C, ldc, bias, \
                       relu, n_chunks, tpc, n_items, chunk, c_rows, relu_mask)
    if (big) { if (accum) FIRA_LAUNCH(4, true); else FIRA_LAUNCH(4, false); }
    else { if (accum) FIRA_LAUNCH(2, true); else FIRA_LAUNCH(2, false); }
#undef FIRA_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) *rc = set_err("gemm_bf16_k256: %s", hipGetErrorString(e));
    return true;
}

}  // namespace fira
