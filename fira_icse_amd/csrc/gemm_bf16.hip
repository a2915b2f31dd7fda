// bf16 GEMM with fp32 accumulation on the gfx950 matrix cores (v_mfma_f32_32x32x16_bf16), fp32 storage on both sides.
//
// BASELINE configs[2] trains in bf16: every nn.Linear forward / dgrad / wgrad of the reference (the addmm/mm rows of
// SURVEY.md §2.3; gnn_transformer.py:76,82,142-144,159,172-173,199,203, Model.py:16-19,54) rounds its two operands to
// bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) on their way into LDS, multiplies on the bf16 MFMA and accumulates,
// adds the bias and writes the result in fp32 -- the arithmetic of torch.autocast(bfloat16) around F.linear.  Same
// interface as gemm_f32_ex (same four storage layouts, bias / ReLU / accumulate / split-K / row map / ReLU mask /
// fused bias-gradient column sums), so the engine switches dtype by switching one function.
//
// At the model's shapes (K = 256 ... 1024, d = 256) the bf16 matrix pipe needs ~1/16 of the fp32 pipe's time and the
// products become HBM/L2-bound: 2*M*N*K / (4*(M*K + N*K + M*N)) = 64 FLOP/B for [M,256]x[256,256] against a machine
// balance of 2500/6.3 = 400 FLOP/B.  The kernel is therefore organised around the operand stream, not the MFMA loop:
//   * 256 threads = 4 waves (2x2), tile BMxBN (128x128 or 64x64), BK = 64: one LDS row is 64 bf16 = 128 B = 8 chunks of
//     16 B; chunk c of row r lives at r*128 + ((c ^ f(r)) << 4) with f(r) = ((r>>1) ^ (r>>4)) & 7, which makes all three
//     access patterns conflict-free: the k-contiguous loader's 16-byte stores (8 lanes = 8 chunks of one row), the
//     k-strided (transposing) loader's stores (8 lanes = rows 4 apart at one chunk) and the MFMA operand fetch
//     (ds_read_b128: 16-lane groups = 16 rows distinct mod 16 at one chunk);
//   * k-contiguous operands (activations [M,K], nn.Linear weights [N,K]) are read with 2 x 16-byte loads per chunk;
//     k-strided operands (dgrad's weight [N,K] reduced over N, wgrad's dY and X reduced over the rows) are read as
//     8(k) x 4(r) register blocks -- 8 row-contiguous 16-byte loads -- and transposed for free when the four packed
//     8-element columns are stored; the bias-gradient column sums are taken from those fp32 registers;
//   * two register stages of look-ahead (tile t+2 in flight while t is consumed), one barrier per K tile;
//   * a 1-D grid with an XCD-aware tile order: workgroup b runs on XCD b % 8, and consecutive workgroups of one XCD
//     walk the tiles that share the LARGER operand's panel, so that panel is fetched from HBM once per XCD and the
//     re-reads hit its private 4 MiB L2.
#include "engine.h"
#include "epilogue.h"
#include <algorithm>
#include <stdlib.h>
#include <type_traits>

namespace fira {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector (HIP's float4 is a union-based struct that
                                                           // keeps conditionally-written register blocks in scratch)
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HK = 64;          // K tile (bf16 elements): one 128-byte LDS row

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));      // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ int lds_swz(int r) { return ((r >> 1) ^ (r >> 4)) & 7; }

// ---- k-contiguous operand: source element (r, k) at src[r*ld + k] ------------------------------------------------
template <int ROWS>
struct StageK {
    static constexpr int NV = ROWS / 32;                         // 8-element chunks per thread per tile
    f32x4 v[NV][2];
    // wave-uniform base pointer + 32-bit per-lane offset: lets the loads use the scalar-base addressing form, so the
    // K loop computes no per-lane 64-bit addresses (and needs no fresh address registers while loads are in flight)
    __device__ __forceinline__ void load_fast(const float* __restrict__ src, int ld, int r0, int k0, int t) {
        const float* base = src + (size_t)r0 * ld + k0;
        const unsigned o = (unsigned)(t >> 3) * (unsigned)ld + (unsigned)(t & 7) * 8u;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i][0] = *reinterpret_cast<const f32x4*>(base + (o + 32u * i * (unsigned)ld));
            v[i][1] = *reinterpret_cast<const f32x4*>(base + (o + 32u * i * (unsigned)ld + 4u));
        }
    }
    // edge tiles (branch-free): rows past the end are clamped to the last row -- their products land in output rows /
    // columns the epilogue never stores -- and elements past k_end read a clamped address and are replaced by zero
    __device__ __forceinline__ void load(const float* __restrict__ src, int ld, int r0, int r_end, int k0, int k_end, int t) {
        const int k = k0 + (t & 7) * 8;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const float* q = src + (size_t)min(r0 + (t >> 3) + 32 * i, r_end - 1) * ld;
            auto el = [&](int j) {
                const float x = q[min(k + j, k_end - 1)];
                return k + j < k_end ? x : 0.f;
            };
            v[i][0] = f32x4{el(0), el(1), el(2), el(3)};
            v[i][1] = f32x4{el(4), el(5), el(6), el(7)};
        }
    }
    __device__ __forceinline__ void store(char* __restrict__ lds, int t) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int r = (t >> 3) + 32 * i;
            uint4 q;
            q.x = pack_bf16(v[i][0].x, v[i][0].y); q.y = pack_bf16(v[i][0].z, v[i][0].w);
            q.z = pack_bf16(v[i][1].x, v[i][1].y); q.w = pack_bf16(v[i][1].z, v[i][1].w);
            *reinterpret_cast<uint4*>(lds + r * 128 + (((t & 7) ^ lds_swz(r)) << 4)) = q;
        }
    }
    __device__ __forceinline__ void colsum_add(float (&)[4]) const {}
};

// ---- k-strided operand: source element (r, k) at src[k*ld + r]; one 8(k) x 4(r) block per active thread ------------
// A 64-row tile has 128 blocks: the A operand takes threads 0..127, the B operand threads 128..255 (SLOT).
template <int ROWS, int SLOT>
struct StageR {
    static constexpr int TP = ROWS * 2 < 256 ? ROWS * 2 : 256;  // threads that carry a block
    f32x4 v[8];
    __device__ __forceinline__ static bool active(int t) { return TP == 256 || (t / TP) == SLOT; }
    __device__ __forceinline__ void load_fast(const float* __restrict__ src, int ld, int r0, int k0, int t) {
        if (!active(t)) return;
        const int idx = t % TP, rb = idx % (ROWS / 4), kb = idx / (ROWS / 4);
        const float* base = src + (size_t)k0 * ld + r0;
        const unsigned o = (unsigned)(kb * 8) * (unsigned)ld + (unsigned)rb * 4u;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(base + (o + (unsigned)j * (unsigned)ld));
    }
    __device__ __forceinline__ void load(const float* __restrict__ src, int ld, int r0, int r_end, int k0, int k_end, int t) {
        if (!active(t)) return;
        const int idx = t % TP, rb = idx % (ROWS / 4), kb = idx / (ROWS / 4);
        const int r = r0 + rb * 4;
        const int c0 = min(r, r_end - 1), c1 = min(r + 1, r_end - 1), c2 = min(r + 2, r_end - 1), c3 = min(r + 3, r_end - 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + kb * 8 + j;
            const float* q = src + (size_t)min(k, k_end - 1) * ld;
            const float x0 = q[c0], x1 = q[c1], x2 = q[c2], x3 = q[c3];
            const float keep = k < k_end ? 1.f : 0.f;          // (inf/nan past the end cannot occur: the address is clamped to real data)
            v[j] = f32x4{x0, x1, x2, x3} * keep;
        }
    }
    __device__ __forceinline__ void store(char* __restrict__ lds, int t) const {
        if (!active(t)) return;
        const int idx = t % TP, rb = idx % (ROWS / 4), kb = idx / (ROWS / 4);
#define FIRA_COL(q, m)                                                                              \
        {                                                                                           \
            const int r = rb * 4 + q;                                                               \
            uint4 w;                                                                                \
            w.x = pack_bf16(v[0].m, v[1].m); w.y = pack_bf16(v[2].m, v[3].m);                       \
            w.z = pack_bf16(v[4].m, v[5].m); w.w = pack_bf16(v[6].m, v[7].m);                       \
            *reinterpret_cast<uint4*>(lds + r * 128 + ((kb ^ lds_swz(r)) << 4)) = w;                \
        }
        FIRA_COL(0, x) FIRA_COL(1, y) FIRA_COL(2, z) FIRA_COL(3, w)
#undef FIRA_COL
    }
    // fused bias gradient: sums over k of this thread's 4 columns (rows r0 + rb*4 .. +3), fp32, before the rounding
    __device__ __forceinline__ void colsum_add(float (&cs)[4]) const {
#pragma unroll
        for (int j = 0; j < 8; ++j) { cs[0] += v[j].x; cs[1] += v[j].y; cs[2] += v[j].z; cs[3] += v[j].w; }
    }
};

// ---- k-contiguous operand already stored in bf16 (the weight shadows): source element (r, k) at src[r*ld + k] ---------
// 16 bytes = one 8-element chunk per load, no conversion, half the bytes and half the staging registers.
template <int ROWS>
struct StageKb {
    static constexpr int NV = ROWS / 32;
    uint4 v[NV];
    __device__ __forceinline__ void load_fast(const uint16_t* __restrict__ src, int ld, int r0, int k0, int t) {
        const uint16_t* base = src + (size_t)r0 * ld + k0;
        const unsigned o = (unsigned)(t >> 3) * (unsigned)ld + (unsigned)(t & 7) * 8u;
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const uint4*>(base + (o + 32u * i * (unsigned)ld));
    }
    // edge tiles: rows clamped (their products are never stored); chunks past k_end are zero; a chunk that straddles
    // k_end is read whole (the shadow buffer continues behind every tensor) and its tail half-words are cleared
    __device__ __forceinline__ void load(const uint16_t* __restrict__ src, int ld, int r0, int r_end, int k0, int k_end, int t) {
        const int k = k0 + (t & 7) * 8;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const uint16_t* q = src + (size_t)min(r0 + (t >> 3) + 32 * i, r_end - 1) * ld;
            uint4 x = *reinterpret_cast<const uint4*>(q + (k < k_end ? k : 0));
            const int nv = k_end - k;                             // valid elements of this chunk (<= 0: none)
            uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t lo = (2 * j < nv) ? 0xffffu : 0u, hi = (2 * j + 1 < nv) ? 0xffff0000u : 0u;
                w[j] &= (lo | hi);
            }
            v[i] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    __device__ __forceinline__ void store(char* __restrict__ lds, int t) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int r = (t >> 3) + 32 * i;
            *reinterpret_cast<uint4*>(lds + r * 128 + (((t & 7) ^ lds_swz(r)) << 4)) = v[i];
        }
    }
    __device__ __forceinline__ void colsum_add(float (&)[4]) const {}
};

template <int ROWS, bool CONTIG_K, int SLOT>
struct StageSel { using type = StageR<ROWS, SLOT>; };
template <int ROWS, int SLOT>
struct StageSel<ROWS, true, SLOT> { using type = StageK<ROWS>; };
// B operand type: float (rounded while staged) or uint16_t (bf16 weight shadow, always k-contiguous)
template <int ROWS, bool CONTIG_K, typename BT>
struct StageSelB { using type = typename StageSel<ROWS, CONTIG_K, 1>::type; };
template <int ROWS>
struct StageSelB<ROWS, true, uint16_t> { using type = StageKb<ROWS>; };

// One BM x BN output tile (tm, tn) over the K range of split z, by the 256 threads of a workgroup.
template <int BM, int BN, bool TA, bool TB, typename BT = float>
__device__ __forceinline__ void gemm_bf16_tile(int M, int N, int K, const float* __restrict__ A, int lda,
                                               const BT* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                                               const float* __restrict__ bias, int flags, int splitk, int k_chunk,
                                               float* __restrict__ colsum, const int32_t* __restrict__ c_rows,
                                               const float* __restrict__ relu_mask, int tm, int tn, int z) {
    using SA = typename StageSel<BM, !TA, 0>::type;       // A stored [M,K] (k contiguous) unless TA
    using SB = typename StageSelB<BN, TB, BT>::type;      // B stored [N,K] (k contiguous) when TB
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    __shared__ __attribute__((aligned(16))) char sm[2 * (BM + BN) * 128];

    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = z * k_chunk, kend = min(K, kbeg + k_chunk);
    const bool atomic = splitk > 1, first = z == 0;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntile = (kend - kbeg + HK - 1) / HK;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool do_cs = TA && colsum != nullptr && tn == 0;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};

    char* const smA = sm;
    char* const smB = sm + 2 * BM * 128;
    // per-lane operand rows of the MFMA fetch: row offset and swizzle key
    int offA[TM], keyA[TM], offB[TN], keyB[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int r = wm * WM + i * 32 + l31; offA[i] = r * 128; keyA[i] = lds_swz(r); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int r = wn * WN + j * 32 + l31; offB[j] = r * 128; keyB[j] = lds_swz(r); }

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const char* sa = smA + buf * BM * 128;
        const char* sb = smB + buf * BN * 128;
#pragma unroll
        for (int s = 0; s < HK / 16; ++s) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const bf16x8*>(sa + offA[i] + (((2 * s + kh) ^ keyA[i]) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const bf16x8*>(sb + offB[j] + (((2 * s + kh) ^ keyB[j]) << 4));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    // every fetched tile is stored exactly once; the column sums are taken there, when the registers are needed anyway
    auto put = [&](const SA& ra, const SB& rb, int buf) __attribute__((always_inline)) {
        if (do_cs) ra.colsum_add(cs);
        ra.store(smA + buf * BM * 128, t);
        rb.store(smB + buf * BN * 128, t);
    };

    // Interior workgroups (every row, column and K tile in range -- block-uniform, decided once) run a software
    // pipeline of straight 16-byte loads with two register stages.  Tile t is fetched into stage t & 1 and stored to
    // LDS buffer t & 1 one step before it is consumed:
    //     step(t):  fetch(t + 2) -> stage t & 1 | MFMAs on LDS[t & 1] | store stage (t + 1) & 1 -> LDS[(t + 1) & 1]
    // Both sub-steps of the steady-state loop are unconditional and the last <= 3 tiles are straight-line code per
    // remaining count, so that no load sits behind a branch: the compiler's s_waitcnt insertion merges the pending-load
    // state of both sides of every join, and a conditional prefetch (or a per-tile fast/edge choice) made it drain the
    // tiles in flight at the head of every iteration -- the loop then ran at one memory latency per K tile.
    if (m0 + BM <= M && n0 + BN <= N && (kend - kbeg) % HK == 0 && ntile > 0) {
        SA ra0, ra1;
        SB rb0, rb1;
        auto fetch = [&](int k0, SA& ra, SB& rb) __attribute__((always_inline)) {
            ra.load_fast(A, lda, m0, k0, t);
            rb.load_fast(B, ldb, n0, k0, t);
        };
        fetch(kbeg, ra0, rb0);
        put(ra0, rb0, 0);
        if (ntile == 1) {
            __syncthreads();
            compute(0);
        } else {
            fetch(kbeg + HK, ra1, rb1);
            __syncthreads();
            int it = 0;
            for (; it + 3 < ntile; it += 2) {                 // `it` stays even: stage / buffer roles are static
                fetch(kbeg + (it + 2) * HK, ra0, rb0);
                compute(0);
                put(ra1, rb1, 1);
                __syncthreads();
                fetch(kbeg + (it + 3) * HK, ra1, rb1);
                compute(1);
                put(ra0, rb0, 0);
                __syncthreads();
            }
            if (ntile - it == 3) {                            // tile it in LDS[0], tile it + 1 in flight (stage 1)
                fetch(kbeg + (it + 2) * HK, ra0, rb0);
                compute(0);
                put(ra1, rb1, 1);
                __syncthreads();
                compute(1);
                put(ra0, rb0, 0);
                __syncthreads();
                compute(0);
            } else {
                compute(0);
                put(ra1, rb1, 1);
                __syncthreads();
                compute(1);
            }
        }
    } else {
        // edge workgroups (last row / column of tiles, the split that holds a partial K tile): clamped branch-free loads,
        // one register stage, no overlap -- a few per cent of the workgroups at most
        SA ra;
        SB rb;
        for (int it = 0; it < ntile; ++it) {
            ra.load(A, lda, m0, M, kbeg + it * HK, kend, t);
            rb.load(B, ldb, n0, N, kbeg + it * HK, kend, t);
            put(ra, rb, 0);
            __syncthreads();
            compute(0);
            __syncthreads();
        }
    }
    __syncthreads();

    if constexpr (TA) {
        if (do_cs) {                               // block-level combine of the column sums, one atomic per column
            float* red = reinterpret_cast<float*>(sm);  // all tile reads are done (barrier at the end of the last step)
            for (int i = t; i < BM; i += 256) red[i] = 0.f;
            __syncthreads();
            if (SA::active(t)) {
                const int c = ((t % SA::TP) % (BM / 4)) * 4;
                atomicAdd(&red[c + 0], cs[0]); atomicAdd(&red[c + 1], cs[1]);
                atomicAdd(&red[c + 2], cs[2]); atomicAdd(&red[c + 3], cs[3]);
            }
            __syncthreads();
            for (int i = t; i < BM; i += 256)
                if (m0 + i < M) unsafeAtomicAdd(&colsum[m0 + i], red[i]);
        }
    }

    // epilogue (epilogue.h): C/D layout of the 32x32 MFMA (dtype-independent): col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool relu = flags & FIRA_GEMM_RELU;
    const bool accum = flags & FIRA_GEMM_ACCUM;
    const float* bias_p = (bias != nullptr && first) ? bias : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WN + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float vals[16];
            int rows[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                vals[r] = acc[i][j][r];
                rows[r] = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            }
            epilogue_col<16>(vals, rows, col, M, N, C, ldc, bias_p, relu, accum, atomic, c_rows, relu_mask);
        }
    }
}

// Workgroup -> tile order.  Hardware places workgroup b on XCD b % 8 (each XCD has a private 4 MiB L2), so logical work
// item i = (b % 8) * chunk + b / 8 gives every XCD one CONTIGUOUS range of items; the items are ordered so that
// neighbours share operand panels:  z (K split) slowest -- with splitk >= 8 an XCD owns whole K slabs of both operands
// (weight gradients: every tile of a slab re-reads it from that XCD's L2, not from HBM) -- then the tiles of the
// operand with the larger footprint, then the other dimension fastest.
__device__ __forceinline__ bool tile_of_block(int tiles_m, int tiles_n, int splitk, int spread_n, int chunk, int& tm, int& tn,
                                              int& z) {
    const int i = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (i >= tiles_m * tiles_n * splitk) return false;
    const int per = tiles_m * tiles_n;
    z = i / per;
    const int r = i - z * per;
    if (!spread_n) { tm = r / tiles_n; tn = r - tm * tiles_n; }
    else { tn = r / tiles_m; tm = r - tn * tiles_m; }
    return true;
}

template <int BM, int BN, bool TA, bool TB, typename BT = float>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                        const BT* __restrict__ B, int ldb, float* __restrict__ C,
                                                        int ldc, const float* __restrict__ bias, int flags,
                                                        int tiles_m, int tiles_n, int splitk, int k_chunk, int spread_n,
                                                        int chunk, float* __restrict__ colsum,
                                                        const int32_t* __restrict__ c_rows,
                                                        const float* __restrict__ relu_mask) {
    int tm, tn, z;
    if (!tile_of_block(tiles_m, tiles_n, splitk, spread_n, chunk, tm, tn, z)) return;
    gemm_bf16_tile<BM, BN, TA, TB, BT>(M, N, K, A, lda, B, ldb, C, ldc, bias, flags, splitk, k_chunk, colsum, c_rows,
                                       relu_mask, tm, tn, z);
}

// Grouped weight gradients: dW_i += dY_i^T X_i (+ db_i) for up to GROUP_MAX independent problems in ONE launch -- the
// decoder's and the head's small reductions over the B*30 target rows, which launched one by one are a fill and a
// drain each.  The table travels in the kernel arguments; workgroups are ordered problem-major with the same
// contiguous-range-per-XCD rule, so one XCD works on a few whole problems.
struct GroupProblem16 {
    const float* A;      // dY  [K, M] (transA layout)
    const float* B;      // X   [K, N]
    float* C;            // dW  [M, N], accumulated
    float* colsum;       // db  [M] or nullptr
    int M, N, K, lda, ldb, ldc, tiles_m, tiles_n, splitk, k_chunk;
};
constexpr int GROUP16_MAX = 40;
struct GroupTable16 {
    int n, chunk;
    int wg_start[GROUP16_MAX + 1];
    GroupProblem16 p[GROUP16_MAX];
};
__global__ __launch_bounds__(256) void gemm_bf16_grouped_wgrad_kernel(GroupTable16 g) {
    const int w = (blockIdx.x & 7) * g.chunk + (blockIdx.x >> 3);
    if (w >= g.wg_start[g.n]) return;
    int i = 0;
    while (i + 1 < g.n && w >= g.wg_start[i + 1]) ++i;               // uniform scan of <= 40 entries
    const GroupProblem16& q = g.p[i];
    const int r = w - g.wg_start[i];
    const int per = q.tiles_m * q.tiles_n;
    const int z = r / per, tile = r - z * per;
    const int tm = tile / q.tiles_n, tn = tile - tm * q.tiles_n;
    gemm_bf16_tile<64, 64, true, false>(q.M, q.N, q.K, q.A, q.lda, q.B, q.ldb, q.C, q.ldc, nullptr, FIRA_GEMM_ACCUM,
                                        q.splitk, q.k_chunk, q.colsum, nullptr, nullptr, tm, tn, z);
}

template <int BM, int BN, typename BT>
static int launch_bf16(hipStream_t s, int tA, int tB, int M, int N, int K, const float* A, int lda, const BT* B,
                       int ldb, float* C, int ldc, const float* bias, int flags, int splitk, float* colsum,
                       const int32_t* c_rows, const float* relu_mask) {
    const int tiles_m = cdiv(M, BM), tiles_n = cdiv(N, BN);
    const int k_chunk = cdiv(cdiv(K, splitk), HK) * HK;
    // walk the tiles of the operand with the larger footprint slowest (its panels are then read from HBM once)
    const int spread_n = (long)N > (long)M ? 1 : 0;
    const int chunk = cdiv(tiles_m * tiles_n * splitk, 8);
    dim3 grid(8 * chunk);
#define FIRA_GO(TA, TB)                                                                                               \
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, TA, TB, BT>), grid, dim3(256), 0, s, M, N, K, A, lda, B, ldb, C, ldc, \
                       bias, flags, tiles_m, tiles_n, splitk, k_chunk, spread_n, chunk, colsum, c_rows, relu_mask)
    if constexpr (std::is_same<BT, uint16_t>::value) {
        FIRA_GO(false, true);                         // weight shadows: always activations [M,K] x shadow [N,K]
    } else {
        if (!tA && tB) FIRA_GO(false, true);
        else if (!tA && !tB) FIRA_GO(false, false);
        else if (tA && !tB) FIRA_GO(true, false);
        else FIRA_GO(true, true);
    }
#undef FIRA_GO
    FIRA_CHECK_LAUNCH("gemm_bf16");
    return 0;
}

// Latency kernel for the skinny products of the decoder (M = B*30 target rows; a [1920,256]x[256,256] product is
// 0.25 GFLOP -- 0.1 us of bf16 MFMA time -- so the tiled kernel's four dependent K steps ARE its run time).  As in
// gemm_small.hip: one 32x32 output tile per workgroup, its 4 waves split K in chunks of 64, every wave fetches its operand
// fragments straight into registers (all loads of a wave in flight together: one memory latency), converts them to bf16,
// runs 4 MFMAs per chunk, and the four partial tiles are combined through LDS.  The reduction order of an MFMA chain is
// free, so inside a chunk lane-half kh takes the CONTIGUOUS k range kh*32 .. kh*32+31 (8 k per MFMA step): a k-contiguous
// operand row is then eight 16-byte loads per lane.
template <bool B_KCONTIG, typename BT = float>
__global__ __launch_bounds__(256) void gemm_bf16_small_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                              const BT* __restrict__ B, int ldb, float* __restrict__ C,
                                                              int ldc, const float* __restrict__ bias, int flags,
                                                              const int32_t* __restrict__ c_rows,
                                                              const float* __restrict__ relu_mask) {
    __shared__ float red[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const float* pa = A + (size_t)min(m0 + l31, M - 1) * lda + kh * 32;      // clamped rows: masked at the store
    const BT* pb = B_KCONTIG ? B + (size_t)min(n0 + l31, N - 1) * ldb + kh * 32
                             : B + (size_t)(kh * 32) * ldb + min(n0 + l31, N - 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nchunk = K / 64;
    for (int c = wave; c < nchunk; c += 4) {
        f32x4 a[8], b[8];
        uint4 bb[4];                                        // bf16 shadow: 32 k of this lane = 4 x 16 bytes, used as is
        const f32x4* qa = reinterpret_cast<const f32x4*>(pa + c * 64);
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = qa[q];
        if constexpr (std::is_same<BT, uint16_t>::value) {
            const uint4* qb = reinterpret_cast<const uint4*>(pb + c * 64);
#pragma unroll
            for (int q = 0; q < 4; ++q) bb[q] = qb[q];
        } else if (B_KCONTIG) {
            const f32x4* qb = reinterpret_cast<const f32x4*>(pb + c * 64);
#pragma unroll
            for (int q = 0; q < 8; ++q) b[q] = qb[q];
        } else {
            const float* qb = pb + (size_t)c * 64 * ldb;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                b[q] = f32x4{qb[(size_t)(4 * q) * ldb], qb[(size_t)(4 * q + 1) * ldb], qb[(size_t)(4 * q + 2) * ldb],
                             qb[(size_t)(4 * q + 3) * ldb]};
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {                       // MFMA step s: k = kh*32 + s*8 .. +7 of this chunk
            uint4 ua, ub;
            ua.x = pack_bf16(a[2 * s].x, a[2 * s].y); ua.y = pack_bf16(a[2 * s].z, a[2 * s].w);
            ua.z = pack_bf16(a[2 * s + 1].x, a[2 * s + 1].y); ua.w = pack_bf16(a[2 * s + 1].z, a[2 * s + 1].w);
            if constexpr (std::is_same<BT, uint16_t>::value) {
                ub = bb[s];
            } else {
                ub.x = pack_bf16(b[2 * s].x, b[2 * s].y); ub.y = pack_bf16(b[2 * s].z, b[2 * s].w);
                ub.z = pack_bf16(b[2 * s + 1].x, b[2 * s + 1].y); ub.w = pack_bf16(b[2 * s + 1].z, b[2 * s + 1].w);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub),
                                                          acc, 0, 0, 0);
        }
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

// Round 3: the latency kernel with COALESCED operand traffic (the fp32 twin and the rationale are in gemm_small.hip:
// gemm_tile32_kernel).  One 32x32 output tile per workgroup, its four wavefronts split K in chunks of 64; a chunk is read
// with wave-wide 16-byte loads of whole row segments (A: 4 rows x 256 bytes of fp32 per instruction, rounded to bf16 on its
// way into LDS; Bb: 8 rows x 128 bytes of the bf16 weight shadow per instruction), parked in a wave-private 8 KB LDS slot
// as 128-byte bf16 rows (16-byte slot g of row r stored at slot g ^ ((r>>1)&7): conflict-free for the stores and for the
// ds_read_b128 fragment fetch) and re-read in MFMA layout: step s of a chunk uses k = (lane>>5)*32 + 8 s .. + 7.
// NCH = chunks per wave (K = 256 NCH): the loop is unrolled, two chunks of loads in flight, no barrier in the K loop.
struct ChunkRegs16 { f32x4 a0, a1, a2, a3, a4, a5, a6, a7; uint4 b0, b1, b2, b3; };

__device__ __forceinline__ void tile32b_fetch(ChunkRegs16& R, const float* A, const unsigned (&oa)[8], const uint16_t* pb0,
                                              const uint16_t* pb1, const uint16_t* pb2, const uint16_t* pb3, int kc) {
    // A: instruction j covers rows 4j + (lane>>4); oa[j] = element offset of this lane's (clamped / mapped) row and column
    R.a0 = *reinterpret_cast<const f32x4*>(A + oa[0] + kc);
    R.a1 = *reinterpret_cast<const f32x4*>(A + oa[1] + kc);
    R.a2 = *reinterpret_cast<const f32x4*>(A + oa[2] + kc);
    R.a3 = *reinterpret_cast<const f32x4*>(A + oa[3] + kc);
    R.a4 = *reinterpret_cast<const f32x4*>(A + oa[4] + kc);
    R.a5 = *reinterpret_cast<const f32x4*>(A + oa[5] + kc);
    R.a6 = *reinterpret_cast<const f32x4*>(A + oa[6] + kc);
    R.a7 = *reinterpret_cast<const f32x4*>(A + oa[7] + kc);
    R.b0 = *reinterpret_cast<const uint4*>(pb0 + kc);
    R.b1 = *reinterpret_cast<const uint4*>(pb1 + kc);
    R.b2 = *reinterpret_cast<const uint4*>(pb2 + kc);
    R.b3 = *reinterpret_cast<const uint4*>(pb3 + kc);
}
__device__ __forceinline__ void tile32b_stage(const ChunkRegs16& R, char* slot, const int (&wa)[8], int wb) {
    __builtin_amdgcn_wave_barrier();                  // the previous chunk's fragment reads precede these stores
#define FIRA_ST_A(j, v) *reinterpret_cast<uint2*>(slot + wa[j]) = uint2{pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)};
    FIRA_ST_A(0, R.a0) FIRA_ST_A(1, R.a1) FIRA_ST_A(2, R.a2) FIRA_ST_A(3, R.a3)
    FIRA_ST_A(4, R.a4) FIRA_ST_A(5, R.a5) FIRA_ST_A(6, R.a6) FIRA_ST_A(7, R.a7)
#undef FIRA_ST_A
    *reinterpret_cast<uint4*>(slot + 4096 + wb) = R.b0;
    *reinterpret_cast<uint4*>(slot + 4096 + wb + 1024) = R.b1;
    *reinterpret_cast<uint4*>(slot + 4096 + wb + 2048) = R.b2;
    *reinterpret_cast<uint4*>(slot + 4096 + wb + 3072) = R.b3;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ f32x16 tile32b_compute(const char* slot, int f0, int f1, int f2, int f3, f32x16 acc) {
#define FIRA_STEP(f)                                                                                          \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(slot + f),                \
                                                  *reinterpret_cast<const bf16x8*>(slot + 4096 + f), acc, 0, 0, 0);
    FIRA_STEP(f0) FIRA_STEP(f1) FIRA_STEP(f2) FIRA_STEP(f3)
#undef FIRA_STEP
    return acc;
}

template <int NCH>
__global__ __launch_bounds__(256) void gemm_bf16_tile32_kernel(int M, int N, const float* __restrict__ A, int lda,
                                                               const uint16_t* __restrict__ Bb, int ldb,
                                                               float* __restrict__ C, int ldc,
                                                               const float* __restrict__ bias, int flags,
                                                               const int32_t* __restrict__ c_rows,
                                                               const float* __restrict__ relu_mask,
                                                               const int32_t* __restrict__ a_rows, int tiles_n) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 8192];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int t;                                            // XCD-chunked tile order (bijective for any grid size)
    {
        const int b = blockIdx.x, nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = b & 7;
        t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    const int m0 = (t / tiles_n) * 32, n0 = (t % tiles_n) * 32;
    char* slot = smem + wave * 8192;
    // A loads: row 4j + ar of the tile, 16-byte column ac (4 floats = half a 16-byte bf16 slot) of the 64-float chunk
    const int ar = lane >> 4, ac = lane & 15;
    unsigned oa[8];
    int wa[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = 4 * j + ar;
        int ra = min(m0 + r, M - 1);                               // clamped rows: masked at the store
        if (a_rows) ra = a_rows[ra];
        oa[j] = (unsigned)ra * (unsigned)lda + ac * 4;
        wa[j] = r * 128 + (((ac >> 1) ^ ((r >> 1) & 7)) << 4) + (ac & 1) * 8;
    }
    // Bb loads: row 8j + br, 16-byte slot bs (8 bf16) of the 64-element chunk
    const int br = lane >> 3, bs = lane & 7;
    const uint16_t* pb0 = Bb + (size_t)min(n0 + br, N - 1) * ldb + ((bs ^ ((br >> 1) & 7)) << 3);
    const uint16_t* pb1 = Bb + (size_t)min(n0 + 8 + br, N - 1) * ldb + ((bs ^ (4 + (br >> 1))) << 3);
    const uint16_t* pb2 = Bb + (size_t)min(n0 + 16 + br, N - 1) * ldb + ((bs ^ ((br >> 1) & 7)) << 3);
    const uint16_t* pb3 = Bb + (size_t)min(n0 + 24 + br, N - 1) * ldb + ((bs ^ (4 + (br >> 1))) << 3);
    const int wb = br * 128 + bs * 16;
    const int i31 = lane & 31, kh = lane >> 5, sw_i = (i31 >> 1) & 7;
    const int f0 = i31 * 128 + (((kh * 4 + 0) ^ sw_i) << 4), f1 = i31 * 128 + (((kh * 4 + 1) ^ sw_i) << 4);
    const int f2 = i31 * 128 + (((kh * 4 + 2) ^ sw_i) << 4), f3 = i31 * 128 + (((kh * 4 + 3) ^ sw_i) << 4);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    ChunkRegs16 R0, R1;
    tile32b_fetch(R0, A, oa, pb0, pb1, pb2, pb3, wave << 6);
    if (NCH > 1) tile32b_fetch(R1, A, oa, pb0, pb1, pb2, pb3, (wave + 4) << 6);
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
        if ((ci & 1) == 0) {
            tile32b_stage(R0, slot, wa, wb);
            if (ci + 2 < NCH) tile32b_fetch(R0, A, oa, pb0, pb1, pb2, pb3, (wave + 4 * (ci + 2)) << 6);
        } else {
            tile32b_stage(R1, slot, wa, wb);
            if (ci + 2 < NCH) tile32b_fetch(R1, A, oa, pb0, pb1, pb2, pb3, (wave + 4 * (ci + 2)) << 6);
        }
        acc = tile32b_compute(slot, f0, f1, f2, f3, acc);
    }
    __builtin_amdgcn_wave_barrier();
    float* red = reinterpret_cast<float*>(slot);                       // this wave's partial tile over its own slot
#pragma unroll
    for (int r = 0; r < 16; ++r) red[r * 64 + lane] = acc[r];
    __syncthreads();
    {
        const int ln = threadIdx.x & 63, rq = threadIdx.x >> 6;
        float vals[4];
        int rows[4];
        const float* p0 = reinterpret_cast<const float*>(smem);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = threadIdx.x + 256 * i, r = rq + 4 * i;
            vals[i] = (p0[idx] + p0[2048 + idx]) + (p0[4096 + idx] + p0[6144 + idx]);
            rows[i] = m0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        }
        epilogue_col<4>(vals, rows, n0 + (ln & 31), M, N, C, ldc, bias, flags & FIRA_GEMM_RELU, flags & FIRA_GEMM_ACCUM, false,
                        c_rows, relu_mask);
    }
}

// true if the coalesced bf16 tile kernel took the call
static bool gemm_bf16_tile32_try(hipStream_t s, int M, int N, int K, const float* A, int lda, const uint16_t* Bb, int ldb,
                                 float* C, int ldc, const float* bias, int flags, const int32_t* c_rows,
                                 const float* relu_mask, const int32_t* a_rows, int* rc) {
    if (K % 256 != 0 || K > 1024) return false;
    const int tiles_n = cdiv(N, 32);
    const dim3 grid(cdiv(M, 32) * tiles_n);
#define FIRA_T32B(NCH)                                                                                                 \
    case NCH:                                                                                                          \
        hipLaunchKernelGGL((gemm_bf16_tile32_kernel<NCH>), grid, dim3(256), 0, s, M, N, A, lda, Bb, ldb, C, ldc, bias, flags, \
                           c_rows, relu_mask, a_rows, tiles_n);                                                        \
        break;
    switch (K / 256) { FIRA_T32B(1) FIRA_T32B(2) FIRA_T32B(3) FIRA_T32B(4) }
#undef FIRA_T32B
    hipError_t e = hipGetLastError();
    *rc = e != hipSuccess ? set_err("gemm_bf16_tile32: %s", hipGetErrorString(e)) : 0;
    return true;
}

// Products this kernel does not take (handled by the fp32 kernels, i.e. computed more precisely, never less):
// unaligned operands, and the tiny ones (4-row mark table, 2-column gate) where a 64-wide tile is mostly padding.
// latency kernel (32x32 tiles, K split over the 4 waves) vs the 64x64 tiled kernel: up to ~4 rounds of 32x32 tiles; with a
// long reduction (K >= 768) only while the tiled kernel would leave most CUs idle (measured at M = 1920, N = 256, K = 1024:
// 14.0 us vs 9.7 us; at M = 960: 10.6 vs 9.9)
static bool small_kernel_wins(int M, int N, int K) {
    const long t32 = (long)cdiv(M, 32) * cdiv(N, 32);
    return t32 <= 1024 && !(K >= 768 && t32 > 256);
}
bool gemm_bf16_takes(int M, int N, int K) { return M >= 32 && N >= 32 && K >= 32; }
static bool bf16_shape_ok(int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb) {
    if (M < 32 || N < 32 || K < 32) return false;
    if (((uintptr_t)A % 16) || ((uintptr_t)B % 16) || (lda % 4) || (ldb % 4)) return false;
    return true;
}

int gemm_bf16_ex(hipStream_t s, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                 float* C, int ldc, const float* bias, int flags, int splitk, float* colsum, const int32_t* c_rows,
                 const float* relu_mask) {
    if (M <= 0 || N <= 0) return 0;
    if (!bf16_shape_ok(tA, tB, M, N, K, A, lda, B, ldb))
        return gemm_f32_ex(s, tA, tB, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, splitk, colsum, c_rows, relu_mask);
    FIRA_REQUIRE(!(relu_mask && (splitk > 1 || c_rows)), "gemm_bf16: the fused ReLU mask needs a plain (unsplit, unmapped) output");
    FIRA_REQUIRE(K > 0 && splitk >= 0, "gemm_bf16: bad K=%d splitk=%d", K, splitk);
    FIRA_REQUIRE(c_rows || epilogue_fits(M, ldc), "gemm_bf16: output of %d x %d floats exceeds the 2 GiB the kernels address", M, ldc);
    FIRA_REQUIRE(!(colsum && !tA), "gemm_bf16: fused column sums need the transA layout");
    ProfScope prof(s, PROF_GEMM, 2.0 * M * N * (double)K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    const bool can_split = (flags & FIRA_GEMM_ACCUM) && !(flags & FIRA_GEMM_RELU) && !relu_mask;
    int tile = ((flags >> FIRA_GEMM_TILE_SHIFT) & 3) - 1;       // -1: automatic; 0: 128x128; 1, 2: 64x64
    // skinny forward / dgrad shapes (the decoder): up to ~4 rounds of 32x32 tiles the latency kernel wins
    static const int small_mode = [] { const char* e = getenv("FIRA_SMALL_GEMM"); return e ? atoi(e) : 1; }();
    if (tile < 0 && splitk <= 1 && !colsum && !tA && small_mode && K % 64 == 0 && small_kernel_wins(M, N, K)) {
        dim3 grid(cdiv(N, 32), cdiv(M, 32));
        if (tB) hipLaunchKernelGGL(gemm_bf16_small_kernel<true>, grid, dim3(256), 0, s, M, N, K, A, lda, B, ldb, C, ldc, bias, flags & 3, c_rows, relu_mask);
        else hipLaunchKernelGGL(gemm_bf16_small_kernel<false>, grid, dim3(256), 0, s, M, N, K, A, lda, B, ldb, C, ldc, bias, flags & 3, c_rows, relu_mask);
        FIRA_CHECK_LAUNCH("gemm_bf16_small");
        return 0;
    }
    const long t128 = (long)cdiv(M, 128) * cdiv(N, 128), t64 = (long)cdiv(M, 64) * cdiv(N, 64);
    // 128x128 (one workgroup per CU, 2x fewer operand re-reads) only pays on long reductions with plenty of tiles
    // (measured: K = 3072 dgrad / wgrad of the cross K|V projection 1.4x faster, every K = 256 shape 1.5-1.9x slower)
    // ... or with few tiles but a reduction long enough that the split-K fan-out fills the chip anyway (the K|V
    // projection's weight gradient [3072,256] over 17 000 rows: 48 tiles x 16 splits, 139 us against 372 us with 64x64)
    if (tile < 0) {
        const long split128 = can_split ? std::min((768 + t128 - 1) / t128, (long)K / 256) : 1;
        tile = (K >= 2048 && (t128 >= 256 || t128 * split128 >= 512)) ? 0 : 2;
    }
    if (splitk == 0) {
        // memory-bound: aim at ~2 workgroups per CU; every split re-reads nothing (disjoint K ranges) but adds one
        // atomic per output element, so split only reductions that are long compared with the tile
        splitk = 1;
        const long tiles = tile == 0 ? t128 : t64;
        if (can_split && tiles < 512 && K >= 512) {
            // ~3 workgroups per CU; at least 4 K tiles per split; multiples of 8 so that every XCD owns whole K slabs
            long want = std::min((768 + tiles - 1) / tiles, (long)K / 256);
            if (want >= 8) want = want / 8 * 8;
            splitk = (int)std::max(1L, want);
        }
    }
    FIRA_REQUIRE(!(splitk > 1 && !can_split), "gemm_bf16: split-K needs accumulate semantics and no relu");
    flags &= 3;
    if (tile == 0) return launch_bf16<128, 128>(s, tA, tB, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, splitk, colsum, c_rows, relu_mask);
    return launch_bf16<64, 64>(s, tA, tB, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, splitk, colsum, c_rows, relu_mask);
}

// ---- grouped weight gradients (host side): problems are collected, then launched together
struct GroupBuilder16 {
    GroupTable16 t;
    GroupBuilder16() { t.n = 0; t.wg_start[0] = 0; }
};
static GroupBuilder16& group16() { static thread_local GroupBuilder16 g; return g; }

void gemm_bf16_group_reset() { group16().t.n = 0; }
bool gemm_bf16_group_full() { return group16().t.n == GROUP16_MAX; }
int gemm_bf16_group_flush(hipStream_t s) {
    GroupTable16& t = group16().t;
    if (t.n == 0) return 0;
    double flop = 0, bytes = 0;
    for (int i = 0; i < t.n; ++i) {
        const double M = t.p[i].M, N = t.p[i].N, K = t.p[i].K;
        flop += 2.0 * M * N * K;
        bytes += 4.0 * (M * K + N * K + M * N);
    }
    ProfScope prof(s, PROF_GEMM, flop, bytes);
    t.chunk = cdiv(t.wg_start[t.n], 8);
    hipLaunchKernelGGL(gemm_bf16_grouped_wgrad_kernel, dim3(8 * t.chunk), dim3(256), 0, s, t);
    t.n = 0;
    FIRA_CHECK_LAUNCH("gemm_bf16_grouped_wgrad");
    return 0;
}
// dW[M,N] += A^T B with A = dY [K,M], B = X [K,N]; db[M] += column sums of dY.  Queued; runs at the next flush on `s`.
int gemm_bf16_group_add_wgrad(hipStream_t s, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                              float* C, int ldc, float* colsum, int max_split) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    FIRA_REQUIRE(bf16_shape_ok(1, 0, M, N, K, A, lda, B, ldb), "gemm_bf16_group_add_wgrad: unsupported shape %dx%dx%d", M, N, K);
    GroupTable16& t = group16().t;
    if (t.n == GROUP16_MAX) {
        int rc = gemm_bf16_group_flush(s);
        if (rc) return rc;
    }
    GroupProblem16& q = t.p[t.n];
    q.A = A; q.B = B; q.C = C; q.colsum = colsum;
    q.M = M; q.N = N; q.K = K; q.lda = lda; q.ldb = ldb; q.ldc = ldc;
    q.tiles_m = cdiv(M, 64); q.tiles_n = cdiv(N, 64);
    // the group shares the chip (a few thousand workgroups in all): short K chains (<= 8 tiles) matter more than the
    // extra atomics of a split
    q.splitk = std::max(1, std::min(max_split > 0 ? max_split : 8, K / 512));
    q.k_chunk = cdiv(cdiv(K, q.splitk), HK) * HK;
    t.wg_start[t.n + 1] = t.wg_start[t.n] + q.tiles_m * q.tiles_n * q.splitk;
    ++t.n;
    return 0;
}

// C[M,N] (+)= A[M,K] (fp32, rounded while staged) . Bb[N,K]^T with Bb a bf16 weight shadow (k contiguous): the forward
// (Bb = shadow of W) and the data gradient (Bb = shadow of W^T) of every nn.Linear in bf16 mode.
int gemm_bf16_wb_ex(hipStream_t s, int M, int N, int K, const float* A, int lda, const uint16_t* Bb, int ldb, float* C,
                    int ldc, const float* bias, int flags, int splitk, const int32_t* c_rows, const float* relu_mask) {
    if (M <= 0 || N <= 0) return 0;
    FIRA_REQUIRE(M >= 32 && N >= 32 && K >= 32 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)Bb % 16) == 0 && lda % 4 == 0 &&
                 ldb % 8 == 0, "gemm_bf16_wb: unsupported shape / alignment %dx%dx%d", M, N, K);
    FIRA_REQUIRE(!(relu_mask && (splitk > 1 || c_rows)), "gemm_bf16_wb: the fused ReLU mask needs a plain output");
    FIRA_REQUIRE(c_rows || epilogue_fits(M, ldc), "gemm_bf16_wb: output of %d x %d floats exceeds the 2 GiB the kernels address", M, ldc);
    ProfScope prof(s, PROF_GEMM, 2.0 * M * N * (double)K, 4.0 * ((double)M * K + (double)M * N) + 2.0 * (double)N * K);
    const bool can_split = (flags & FIRA_GEMM_ACCUM) && !(flags & FIRA_GEMM_RELU) && !relu_mask;
    static const int small_mode = [] { const char* e = getenv("FIRA_SMALL_GEMM"); return e ? atoi(e) : 2; }();
    // the coalesced tile kernel stays ahead of the tiled / panel kernels for more rounds of tiles and for long reductions
    // (FIRA_SMALL_TILES16: A/B switch of the threshold)
    static const long max_t32 = [] { const char* e = getenv("FIRA_SMALL_TILES16"); return e ? atol(e) : 1024L; }();
    if (splitk <= 1 && small_mode >= 2 && K % 256 == 0 && K <= 1024 && (long)cdiv(M, 32) * cdiv(N, 32) <= max_t32 &&
        !small_kernel_wins(M, N, K)) {
        int rc;
        if (gemm_bf16_tile32_try(s, M, N, K, A, lda, Bb, ldb, C, ldc, bias, flags & 3, c_rows, relu_mask, nullptr, &rc)) return rc;
    }
    if (splitk <= 1 && small_mode && K % 64 == 0 && small_kernel_wins(M, N, K)) {
        int rc;
        if (small_mode != 1 && gemm_bf16_tile32_try(s, M, N, K, A, lda, Bb, ldb, C, ldc, bias, flags & 3, c_rows, relu_mask,
                                                    nullptr, &rc))
            return rc;
        dim3 grid(cdiv(N, 32), cdiv(M, 32));
        hipLaunchKernelGGL((gemm_bf16_small_kernel<true, uint16_t>), grid, dim3(256), 0, s, M, N, K, A, lda, Bb, ldb, C, ldc,
                           bias, flags & 3, c_rows, relu_mask);
        FIRA_CHECK_LAUNCH("gemm_bf16_small");
        return 0;
    }
    if (splitk <= 1) {                                  // K = 256: the A-stationary panel kernel (gemm_bf16_panel.hip)
        int rc;
        if (gemm_bf16_k256_try(s, M, N, K, A, lda, Bb, ldb, C, ldc, bias, flags, c_rows, relu_mask, &rc)) return rc;
        // K = 512 / 768 / 1024 as a chain of K = 256 slices on the same stream, the later ones accumulating (an atomic add
        // per element, one owner): the encoder's q|k data gradient [10 000, 256] x K 512 takes 2 x 12 us instead of the
        // tiled kernel's 43 us.  Not with a ReLU / ReLU mask, which must see the complete sum, and not for the decoder-sized
        // products with K = 768 / 1024, where one tiled launch (9.7 us) beats three or four chained ones.
        static const int slice_mode = [] { const char* e = getenv("FIRA_PANEL_SLICES"); return e ? atoi(e) : 1; }();
        if (slice_mode && K % 256 == 0 && K <= 1024 && (K == 512 || M >= 8192) && !(flags & FIRA_GEMM_RELU) && !relu_mask) {
            bool taken = true;
            for (int k0 = 0; k0 < K && taken; k0 += 256) {
                const int fl = k0 == 0 ? (flags & FIRA_GEMM_ACCUM) : FIRA_GEMM_ACCUM;
                taken = gemm_bf16_k256_try(s, M, N, 256, A + k0, lda, Bb + k0, ldb, C, ldc, k0 == 0 ? bias : nullptr, fl, c_rows,
                                           nullptr, &rc);
                if (rc) return rc;
                FIRA_REQUIRE(taken || k0 == 0, "gemm_bf16_wb: K slice %d refused after the first was issued", k0);
            }
            if (taken) return 0;
        }
    }
    const long t128 = (long)cdiv(M, 128) * cdiv(N, 128), t64 = (long)cdiv(M, 64) * cdiv(N, 64);
    const int tile = (t128 >= 256 && K >= 2048) ? 0 : 2;
    if (splitk == 0) {
        splitk = 1;
        const long tiles = tile == 0 ? t128 : t64;
        if (can_split && tiles < 512 && K >= 512) {
            long want = std::min((768 + tiles - 1) / tiles, (long)K / 256);
            if (want >= 8) want = want / 8 * 8;
            splitk = (int)std::max(1L, want);
        }
    }
    FIRA_REQUIRE(!(splitk > 1 && !can_split), "gemm_bf16_wb: split-K needs accumulate semantics and no relu");
    flags &= 3;
    if (tile == 0) return launch_bf16<128, 128, uint16_t>(s, 0, 1, M, N, K, A, lda, Bb, ldb, C, ldc, bias, flags, splitk, nullptr, c_rows, relu_mask);
    return launch_bf16<64, 64, uint16_t>(s, 0, 1, M, N, K, A, lda, Bb, ldb, C, ldc, bias, flags, splitk, nullptr, c_rows, relu_mask);
}

// ---- weight shadows -------------------------------------------------------------------------------------------------
// One launch per step: every 2-D weight the bf16 GEMMs read is copied to bf16 twice -- as stored ([out, in]: forward)
// and transposed ([in, out]: data gradient) -- at the same offset as in the fp32 parameter buffer.  64x64 tiles through
// LDS: both the read and the two writes are row-contiguous.
__global__ __launch_bounds__(256) void weight_shadow_kernel(ShadowTable tab, const float* __restrict__ P,
                                                            uint16_t* __restrict__ Wb, uint16_t* __restrict__ WbT) {
    __shared__ float tile[64][65];
    int e = 0;
    while (e + 1 < tab.n && (int)blockIdx.x >= tab.tile_start[e + 1]) ++e;
    const ShadowEntry& q = tab.e[e];
    const int tl = blockIdx.x - tab.tile_start[e];
    const int tiles_c = (q.cols + 63) / 64;
    const int r0 = (tl / tiles_c) * 64, c0 = (tl % tiles_c) * 64;
    const int t = threadIdx.x, tx = t & 63, ty = t >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        float v = 0.f;
        if (r < q.rows && c < q.cols) {
            v = P[q.offset + (size_t)r * q.cols + c];
            Wb[q.offset + (size_t)r * q.cols + c] = (uint16_t)(pack_bf16(v, 0.f) & 0xffffu);
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {                 // row c0 + i of the transposed matrix [cols, rows]
        const int c = c0 + i, r = r0 + tx;
        if (c < q.cols && r < q.rows) WbT[q.offset_t + (size_t)c * q.pitch_t + r] = (uint16_t)(pack_bf16(tile[tx][i], 0.f) & 0xffffu);
    }
}
int weight_shadows(hipStream_t s, const ShadowTable& tab, const float* P, uint16_t* Wb, uint16_t* WbT) {
    if (tab.n == 0) return 0;
    ProfScope prof(s, PROF_ROWOPS, 0.0);
    hipLaunchKernelGGL(weight_shadow_kernel, dim3(tab.tile_start[tab.n]), dim3(256), 0, s, tab, P, Wb, WbT);
    FIRA_CHECK_LAUNCH("weight_shadows");
    return 0;
}

}  // namespace fira

extern "C" int fira_weight_shadow(void* stream, int rows, int cols, const float* W, uint16_t* Wb, uint16_t* WbT) {
    FIRA_REQUIRE(rows > 0 && cols > 0 && W && Wb && WbT, "fira_weight_shadow: bad argument");
    fira::ShadowTable tab;
    tab.n = 1;
    tab.e[0] = fira::ShadowEntry{0, rows, cols, rows, 0};
    tab.tile_start[1] = fira::cdiv(rows, 64) * fira::cdiv(cols, 64);
    return fira::weight_shadows((hipStream_t)stream, tab, W, Wb, WbT);
}
extern "C" int fira_gemm_bf16_wb(void* stream, int M, int N, int K, const float* A, int lda, const uint16_t* Bb, int ldb,
                                 float* C, int ldc, const float* bias, int flags, int splitk) {
    FIRA_REQUIRE(splitk >= 0, "fira_gemm_bf16_wb: splitk must be >= 0 (0 = automatic)");
    return fira::gemm_bf16_wb_ex((hipStream_t)stream, M, N, K, A, lda, Bb, ldb, C, ldc, bias, flags, splitk, nullptr, nullptr);
}
extern "C" int fira_gemm_bf16(void* stream, int transA, int transB, int M, int N, int K, const float* A, int lda,
                              const float* B, int ldb, float* C, int ldc, const float* bias, int flags, int splitk) {
    FIRA_REQUIRE(splitk >= 0, "fira_gemm_bf16: splitk must be >= 0 (0 = automatic)");
    return fira::gemm_bf16_ex((hipStream_t)stream, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, splitk,
                              nullptr, nullptr, nullptr);
}
