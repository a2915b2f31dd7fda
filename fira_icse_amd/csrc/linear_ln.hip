// Fused  y = LayerNorm(dropout(X W^T + b [+ r c^T]) + res) * gamma + beta   for the 256-wide post-LN blocks
// (reference gnn_transformer.py:86, 159-161, 172-174, 203-205): the closing step of every Combination, GCN, attention and
// feed-forward block -- 30 sites per training step, 18 per decode step.
//
// STATUS: correct (op tests), exported (fira_linear_layernorm_fwd), NOT used by the engine by default -- measured slower than
// product + row kernel in the step and in the decode loop (numbers at linear_ln_fwd_try below); kept as the documented
// negative result for "fuse the post-LN block in fp32".
//
// Why a fused kernel was tried: the decoder side of the step is a chain of ~150 dependent launches of a few microseconds each
// (M = B*30 rows), bound by the ~5 us dispatch floor + ~2.5 us gap per launch, not by work; GEMM + LayerNorm as two launches
// costs 9.7 + 5.5 + 2.5 us for [960,256]x[256,256].  One workgroup here owns 32 COMPLETE output rows (all 256 columns), so
// the row statistics need no second pass over memory: 8 waves (512 threads), wave w computes the 32x32 tile of columns
// [32w, 32w+32) on the fp32 MFMA exactly like gemm_f32.hip (k-major LDS staging, two register stages, no load behind a
// branch), the biased tile goes to LDS, and each wave then normalises 4 rows the way add_layernorm_fwd_kernel does (same
// dropout indices: element r*256 + c of the site, so masks -- and the backward kernels -- are unchanged).
// MFMA rate per CU is that of the 64x64 GEMM tile (2 workgroups of 8 waves = 4 waves per SIMD, 16 MFMAs per wave and K tile).
#include "engine.h"
#include "epilogue.h"
#include <stdlib.h>

namespace fira {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int LBK = 32;                 // K tile
constexpr int LA_LD = 33;               // [k][row] pitch of the 32-row A tile
constexpr int LB_LD = 257;              // [k][col] pitch of the 256-row W tile
constexpr int LT_LD = 260;              // [row][col] pitch of the finished 32x256 tile

__global__ __launch_bounds__(512) void linear_ln_fwd_kernel(int M, int K, const float* __restrict__ X, int ldx,
                                                            const float* __restrict__ W, const float* __restrict__ bias,
                                                            const float* __restrict__ res, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ sum,
                                                            float* __restrict__ y, float* __restrict__ stats, float p,
                                                            float inv_keep, uint64_t seed, uint32_t site,
                                                            const int32_t* __restrict__ y_rows,
                                                            const float* __restrict__ r1_row,
                                                            const float* __restrict__ r1_col) {
    // staging: 2 x (A 32x33 + B 32x257) floats = 74 240 B; the finished tile (32 x 260 floats = 33 280 B) reuses it
    __shared__ __attribute__((aligned(16))) float sm[2 * LBK * (LA_LD + LB_LD)];
    float* const smA0 = sm;
    float* const smB0 = sm + 2 * LBK * LA_LD;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.x * 32;
    const int ntile = K / LBK;

    // A tile: threads 0..255 carry one 16-byte chunk (row t>>3, k (t&7)*4..+3); W tile: every thread 4 chunks
    const int a_row = min(m0 + (t >> 3), M - 1);               // clamped: rows past the end are never stored
    const float* pa = X + (size_t)a_row * ldx + (t & 7) * 4;
    const float* pb = W + (size_t)(t >> 3) * K + (t & 7) * 4;  // rows (t>>3) + 64*i of W
    const bool has_a = t < 256;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    struct Stage { f32x4 a; f32x4 b[4]; };
    auto fetch = [&](int k0, Stage& st) __attribute__((always_inline)) {
        if (has_a) st.a = *reinterpret_cast<const f32x4*>(pa + k0);
#pragma unroll
        for (int i = 0; i < 4; ++i) st.b[i] = *reinterpret_cast<const f32x4*>(pb + (size_t)64 * i * K + k0);
    };
    auto put = [&](const Stage& st, int buf) __attribute__((always_inline)) {
        float* sa = smA0 + buf * LBK * LA_LD;
        float* sb = smB0 + buf * LBK * LB_LD;
        const int kq = (t & 7) * 4, r = t >> 3;
        if (has_a) {
            sa[(kq + 0) * LA_LD + r] = st.a.x; sa[(kq + 1) * LA_LD + r] = st.a.y;
            sa[(kq + 2) * LA_LD + r] = st.a.z; sa[(kq + 3) * LA_LD + r] = st.a.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = r + 64 * i;
            sb[(kq + 0) * LB_LD + c] = st.b[i].x; sb[(kq + 1) * LB_LD + c] = st.b[i].y;
            sb[(kq + 2) * LB_LD + c] = st.b[i].z; sb[(kq + 3) * LB_LD + c] = st.b[i].w;
        }
    };
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const float* sa = smA0 + buf * LBK * LA_LD + kh * LA_LD + l31;
        const float* sb = smB0 + buf * LBK * LB_LD + kh * LB_LD + wave * 32 + l31;
        float a[2], b[2];
        a[0] = sa[0];
        b[0] = sb[0];
#pragma unroll
        for (int kk = 0; kk < LBK / 2; ++kk) {
            const int c = kk & 1, n = c ^ 1;
            if (kk + 1 < LBK / 2) {
                a[n] = sa[2 * (kk + 1) * LA_LD];
                b[n] = sb[2 * (kk + 1) * LB_LD];
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c], b[c], acc, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
    };

    // K pipeline: same shape as gemm_f32.hip (tile t -> stage / LDS buffer t & 1; unconditional steady state)
    Stage s0, s1;
    fetch(0, s0);
    put(s0, 0);
    if (ntile == 1) {
        __syncthreads();
        compute(0);
    } else {
        fetch(LBK, s1);
        __syncthreads();
        int it = 0;
        for (; it + 3 < ntile; it += 2) {
            fetch((it + 2) * LBK, s0);
            compute(0);
            put(s1, 1);
            __syncthreads();
            fetch((it + 3) * LBK, s1);
            compute(1);
            put(s0, 0);
            __syncthreads();
        }
        if (ntile - it == 3) {
            fetch((it + 2) * LBK, s0);
            compute(0);
            put(s1, 1);
            __syncthreads();
            compute(1);
            put(s0, 0);
            __syncthreads();
            compute(0);
        } else {
            compute(0);
            put(s1, 1);
            __syncthreads();
            compute(1);
        }
    }
    __syncthreads();                                           // every wave is done reading the staging buffers

    // finished tile -> LDS (C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5))
    float* const tile = sm;
    {
        const int col = wave * 32 + l31;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * kh) * LT_LD + col] = acc[r] + bv;
    }
    __syncthreads();

    // rows 4*wave .. 4*wave+3: rank-1 term, dropout, residual, LayerNorm (identical arithmetic to add_layernorm_fwd_kernel)
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + lane * 4);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + lane * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int lr = wave * 4 + q, r = m0 + lr;
        if (r >= M) break;                                     // wave-uniform
        f32x4 a = *reinterpret_cast<const f32x4*>(&tile[lr * LT_LD + lane * 4]);
        if (r1_row) {
            const float w = r1_row[r];
            const f32x4 c4 = *reinterpret_cast<const f32x4*>(r1_col + lane * 4);
            a.x = fmaf(w, c4.x, a.x); a.y = fmaf(w, c4.y, a.y); a.z = fmaf(w, c4.z, a.z); a.w = fmaf(w, c4.w, a.w);
        }
        if (p > 0.f) {
            const uint32_t e0 = (uint32_t)r * FIRA_D + lane * 4;
            a.x *= dropout_scale(seed, site, e0 + 0, p, inv_keep);
            a.y *= dropout_scale(seed, site, e0 + 1, p, inv_keep);
            a.z *= dropout_scale(seed, site, e0 + 2, p, inv_keep);
            a.w *= dropout_scale(seed, site, e0 + 3, p, inv_keep);
        }
        const size_t o = (size_t)r * FIRA_D + lane * 4;
        if (res) a += *reinterpret_cast<const f32x4*>(res + o);
        const float mean = wave_sum(a.x + a.y + a.z + a.w) * (1.0f / FIRA_D);
        const f32x4 d = a - mean;
        const float var = wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * (1.0f / FIRA_D);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        if (sum) *reinterpret_cast<f32x4*>(sum + o) = a;
        const size_t oy = y_rows ? (size_t)y_rows[r] * FIRA_D + lane * 4 : o;
        *reinterpret_cast<f32x4*>(y + oy) = d * rstd * g + bt;
        if (stats && lane == 0) {
            stats[2 * r] = mean;
            stats[2 * r + 1] = rstd;
        }
    }
}

// ---- K = 256: 16 complete rows per workgroup, no LDS staging, no barrier before the epilogue -------------------------------
// Halving the rows per workgroup halves the matrix-pipe time a CU spends on its block (3.4 us) and doubles the CUs in use
// (M = 960: 60).  Each of the 8 waves owns 32 output columns (two 16x16 tiles of v_mfma_f32_16x16x4_f32) and pulls its
// operands straight into registers in MFMA layout, like gemm_small.hip: with the K permutation "load q of lane group
// g = lane >> 4 holds k = 16 q + 4 g .. + 3" the four lanes of a row read 64 contiguous bytes per instruction and two
// consecutive instructions complete a 128-byte line (with "g owns 64 contiguous k" a line was touched by 8 instructions
// of 48 KB-per-wave footprints and the 32 KB L1 re-fetched it: 19 us instead of ~10) -- all 48 loads of a wave are in
// flight together, one memory round trip, and the second tile's loads are still arriving while the first tile's MFMAs run.  The 16x256 block then meets in LDS and
// each wave normalises two rows exactly like add_layernorm_fwd_kernel (same dropout indices).  Buffer descriptors make
// rows past M read zeros and drop their stores; there is no branch around a memory instruction (epilogue.h).
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int L16_LD = 260;              // [row][col] pitch of the 16x256 block in LDS

__global__ __launch_bounds__(512) void linear_ln16_fwd_kernel(int M, const float* __restrict__ X, int ldx,
                                                              const float* __restrict__ W, const float* __restrict__ bias,
                                                              const float* __restrict__ res, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ sum,
                                                              float* __restrict__ y, float* __restrict__ stats, float p,
                                                              float inv_keep, uint64_t seed, uint32_t site,
                                                              const int32_t* __restrict__ y_rows,
                                                              const float* __restrict__ r1_row,
                                                              const float* __restrict__ r1_col) {
    __shared__ __attribute__((aligned(16))) float tile[16 * L16_LD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * 16;

    const rsrc_t rX = buf_rsrc(X, ((unsigned)(M - 1) * (unsigned)ldx + 256u) * 4u);
    const rsrc_t rW = buf_rsrc(W, 256u * 256u * 4u);
    f32x4v a[16], b0[16], b1[16];
    {
        const unsigned xo = ((unsigned)(m0 + l15) * (unsigned)ldx + (unsigned)g * 4u) * 4u;       // rows past M: zeros
        const unsigned wo = ((unsigned)(wave * 32 + l15) * 256u + (unsigned)g * 4u) * 4u;
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rX, xo + 64u * q, 0, 0));
#pragma unroll
        for (int q = 0; q < 16; ++q) b0[q] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rW, wo + 64u * q, 0, 0));
#pragma unroll
        for (int q = 0; q < 16; ++q)
            b1[q] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rW, wo + 16u * 256u * 4u + 64u * q, 0, 0));
    }
    __builtin_amdgcn_sched_barrier(0);           // all 48 requests leave before the first MFMA (the scheduler would sink them)
    f32x4v c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].x, b0[q].x, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].y, b0[q].y, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].z, b0[q].z, c0, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].w, b0[q].w, c0, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].x, b1[q].x, c1, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].y, b1[q].y, c1, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].z, b1[q].z, c1, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].w, b1[q].w, c1, 0, 0, 0);
    }
    // C/D layout of the 16x16 MFMA: col = lane & 15, rows 4 (lane >> 4) + i
    {
        const int col = wave * 32 + l15;
        const rsrc_t rB = buf_rsrc(bias ? (const void*)bias : (const void*)W, bias ? 1024u : 0u);     // no bias: zeros
        const float bv0 = buf_load_f32(rB, (unsigned)col * 4u), bv1 = buf_load_f32(rB, (unsigned)(col + 16) * 4u);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            tile[(4 * g + i) * L16_LD + col] = c0[i] + bv0;
            tile[(4 * g + i) * L16_LD + col + 16] = c1[i] + bv1;
        }
    }
    __syncthreads();

    // rows 2*wave, 2*wave+1: rank-1 term, dropout, residual, LayerNorm (identical arithmetic to add_layernorm_fwd_kernel)
    const f32x4v gm = *reinterpret_cast<const f32x4v*>(gamma + lane * 4);
    const f32x4v bt = *reinterpret_cast<const f32x4v*>(beta + lane * 4);
    const rsrc_t rRes = buf_rsrc(res ? (const void*)res : (const void*)W, res ? (unsigned)M * 1024u : 0u);
    const rsrc_t rR1 = buf_rsrc(r1_row ? (const void*)r1_row : (const void*)W, r1_row ? (unsigned)M * 4u : 0u);
    const rsrc_t rC1 = buf_rsrc(r1_row ? (const void*)r1_col : (const void*)W, r1_row ? 1024u : 0u);
    const rsrc_t rMap = buf_rsrc(y_rows ? (const void*)y_rows : (const void*)W, y_rows ? (unsigned)M * 4u : 0u);
    const rsrc_t rSum = buf_rsrc(sum ? (const void*)sum : (const void*)W, sum ? (unsigned)M * 1024u : 0u);   // absent: stores dropped
    const rsrc_t rStats = buf_rsrc(stats ? (const void*)stats : (const void*)W, stats ? (unsigned)M * 8u : 0u);
    const rsrc_t rY = buf_rsrc(y, 0x7fffffffu);
    const unsigned lo = (unsigned)lane * 16u;
    f32x4v v[2], rs[2];
    float w1[2];
    unsigned oy[2];
    const f32x4v c4 = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rC1, lo, 0, 0));
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int lr = wave * 2 + q, r = m0 + lr;
        v[q] = *reinterpret_cast<const f32x4v*>(&tile[lr * L16_LD + lane * 4]);
        rs[q] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rRes, (unsigned)r * 1024u + lo, 0, 0));
        w1[q] = buf_load_f32(rR1, (unsigned)r * 4u);
        const unsigned mapped = __builtin_amdgcn_raw_buffer_load_b32(rMap, (unsigned)r * 4u, 0, 0);
        oy[q] = r < M ? (y_rows ? mapped : (unsigned)r) * 1024u + lo : FIRA_OOB;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = m0 + wave * 2 + q;
        f32x4v x = v[q];
        x.x = fmaf(w1[q], c4.x, x.x); x.y = fmaf(w1[q], c4.y, x.y); x.z = fmaf(w1[q], c4.z, x.z); x.w = fmaf(w1[q], c4.w, x.w);
        if (p > 0.f) {                                   // wave-uniform, no memory instruction inside
            const uint32_t e0 = (uint32_t)r * FIRA_D + lane * 4;
            x.x *= dropout_scale(seed, site, e0 + 0, p, inv_keep);
            x.y *= dropout_scale(seed, site, e0 + 1, p, inv_keep);
            x.z *= dropout_scale(seed, site, e0 + 2, p, inv_keep);
            x.w *= dropout_scale(seed, site, e0 + 3, p, inv_keep);
        }
        x += rs[q];
        const float mean = wave_sum(x.x + x.y + x.z + x.w) * (1.0f / FIRA_D);
        const f32x4v d = x - mean;
        const float var = wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * (1.0f / FIRA_D);
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
        const unsigned o = r < M ? (unsigned)r * 1024u + lo : FIRA_OOB;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32v4_t, x), rSum, o, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32v4_t, d * rstd * gm + bt), rY, oy[q], 0, 0);
        const unsigned so = (r < M && lane == 0) ? (unsigned)r * 8u : FIRA_OOB;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mean), rStats, so, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rstd), rStats, so == FIRA_OOB ? so : so + 4u, 0, 0);
    }
}

// true if the fused kernel took the call (N must be 256; K a multiple of 32; 16-byte aligned operands).
// Measured (profiles/r2_probes.md): a workgroup that owns complete rows is bound by ONE CU's fp32 MFMA rate -- 32 rows x 256
// columns x K=256 is 6.8 us of matrix-pipe time, and M = 960 decoder rows occupy 30 CUs -- so in the training step the
// fused launch (16 us, 45 us at K = 1024) loses to the 240-workgroup product + row kernel (9.4 + 5.5 us): 7 300 vs 7 690
// commits/s.  The 16-row kernel above (K = 256: 60 CUs, operands straight to registers, one round trip) narrows the gap but
// still loses: decoder blocks only 8 000-8 030 vs 8 130-8 230 commits/s, every K = 256 block 7 760; the decode step 0.69 vs
// 0.60 ms.  A product that 240 workgroups finish in one short round trip beats 60 workgroups that each run a 3.4 us MFMA
// chain plus the row step, launch overhead included.  The engine therefore calls the fused kernels only on request
// (FIRA_FUSED_LN=1|2|3); the C entry point always runs them.
bool linear_ln_fwd_try(hipStream_t s, int M, int K, const float* X, int ldx, const float* W, const float* bias, const float* res,
                       const float* gamma, const float* beta, float* sum, float* y, float* stats, float dropout,
                       uint64_t seed, uint32_t site, const int32_t* y_rows, const float* r1_row, const float* r1_col,
                       int* rc, bool force) {
    *rc = 0;
    // FIRA_FUSED_LN: 0 never | 1 K = 256 and M <= 4096 (decoder blocks, decode step) | 2 every K = 256 block | 3 everything
    static const int mode = [] { const char* e = getenv("FIRA_FUSED_LN"); return e ? atoi(e) : 0; }();
    if (!force && (mode == 0 || (K != 256 && mode < 3) || (mode == 1 && M > 4096))) return false;
    if (M <= 0 || K < 32 || K % 32 || ldx % 4 || ((uintptr_t)X % 16) || ((uintptr_t)W % 16)) return false;
    ProfScope prof(s, PROF_GEMM, 2.0 * M * 256.0 * K, 4.0 * ((double)M * K + 256.0 * K + 3.0 * M * 256.0));
    const float inv_keep = dropout > 0.f ? 1.0f / (1.0f - dropout) : 1.0f;
    static const int rows16 = [] { const char* e = getenv("FIRA_FUSED_LN16"); return e ? atoi(e) : 1; }();
    if (K == 256 && rows16 && M < (1 << 21))
        hipLaunchKernelGGL(linear_ln16_fwd_kernel, dim3(cdiv(M, 16)), dim3(512), 0, s, M, X, ldx, W, bias, res, gamma, beta, sum,
                           y, stats, dropout, inv_keep, seed, site, y_rows, r1_row, r1_col);
    else
        hipLaunchKernelGGL(linear_ln_fwd_kernel, dim3(cdiv(M, 32)), dim3(512), 0, s, M, K, X, ldx, W, bias, res, gamma, beta, sum,
                           y, stats, dropout, inv_keep, seed, site, y_rows, r1_row, r1_col);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) *rc = set_err("linear_ln_fwd: %s", hipGetErrorString(e));
    return true;
}

}  // namespace fira

// y = LayerNorm(dropout(X W^T + bias) + res) * gamma + beta; `sum` (optional) receives the pre-norm rows, `stats` {mean, rstd}
extern "C" int fira_linear_layernorm_fwd(void* stream, int M, int K, const float* X, int ldx, const float* W,
                                         const float* bias, const float* res, const float* gamma, const float* beta,
                                         float* sum, float* y, float* stats, float dropout, uint64_t seed,
                                         uint32_t stream_id) {
    int rc;
    FIRA_REQUIRE(X && W && gamma && beta && y, "fira_linear_layernorm_fwd: null argument");
    if (!fira::linear_ln_fwd_try((hipStream_t)stream, M, K, X, ldx, W, bias, res, gamma, beta, sum, y, stats, dropout, seed,
                                 stream_id, nullptr, nullptr, nullptr, &rc, true))
        return fira::set_err("fira_linear_layernorm_fwd: unsupported shape / alignment (M=%d K=%d ldx=%d)", M, K, ldx);
    return rc;
}
