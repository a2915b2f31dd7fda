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
#include <algorithm>

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
    __device__ __forceinline__ void load_fast(const float* __restrict__ src, int ld, int r0, int k0, int t) {
        const float* p = src + (size_t)(r0 + (t >> 3)) * ld + k0 + (t & 7) * 8;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i][0] = *reinterpret_cast<const f32x4*>(p + (size_t)32 * i * ld);
            v[i][1] = *reinterpret_cast<const f32x4*>(p + (size_t)32 * i * ld + 4);
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
        const float* p = src + (size_t)(k0 + kb * 8) * ld + r0 + rb * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4*>(p + (size_t)j * ld);
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

template <int ROWS, bool CONTIG_K, int SLOT>
struct StageSel { using type = StageR<ROWS, SLOT>; };
template <int ROWS, int SLOT>
struct StageSel<ROWS, true, SLOT> { using type = StageK<ROWS>; };

template <int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                        const float* __restrict__ B, int ldb, float* __restrict__ C,
                                                        int ldc, const float* __restrict__ bias, int flags,
                                                        int tiles_m, int tiles_n, int splitk, int k_chunk, int spread_n,
                                                        float* __restrict__ colsum, const int32_t* __restrict__ c_rows,
                                                        const float* __restrict__ relu_mask) {
    using SA = typename StageSel<BM, !TA, 0>::type;       // A stored [M,K] (k contiguous) unless TA
    using SB = typename StageSel<BN, TB, 1>::type;        // B stored [N,K] (k contiguous) when TB
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    __shared__ __attribute__((aligned(16))) char sm[2 * (BM + BN) * 128];

    // ---- tile of this workgroup (XCD-aware order, see the header) ----
    int tm, tn, z;
    {
        const int xcd = blockIdx.x & 7;
        int j = blockIdx.x >> 3;
        if (!spread_n) {                 // A panels are spread over the XCDs; one XCD walks the N tiles of its panel
            tn = j % tiles_n; j /= tiles_n;
            z = j % splitk; j /= splitk;
            tm = j * 8 + xcd;
        } else {                         // B panels (weights / wide outputs) are spread; one XCD walks the M tiles
            tm = j % tiles_m; j /= tiles_m;
            z = j % splitk; j /= splitk;
            tn = j * 8 + xcd;
        }
        if (tm >= tiles_m || tn >= tiles_n) return;
    }
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

    const bool a_in = m0 + BM <= M, b_in = n0 + BN <= N;                // block-uniform
    SA ra0, ra1;
    SB rb0, rb1;
    auto fetch = [&](int k0, SA& ra, SB& rb) __attribute__((always_inline)) {
        const bool k_in = k0 + HK <= kend;
        if (a_in && k_in) ra.load_fast(A, lda, m0, k0, t);
        else ra.load(A, lda, m0, M, k0, kend, t);
        if (b_in && k_in) rb.load_fast(B, ldb, n0, k0, t);
        else rb.load(B, ldb, n0, N, k0, kend, t);
    };
    char* const smA = sm;
    char* const smB = sm + 2 * BM * 128;
    // every fetched tile is stored exactly once; the column sums are taken there, when the registers are needed anyway
    auto put = [&](const SA& ra, const SB& rb, int buf) __attribute__((always_inline)) {
        if (do_cs) ra.colsum_add(cs);
        ra.store(smA + buf * BM * 128, t);
        rb.store(smB + buf * BN * 128, t);
    };
    if (ntile > 0) {
        fetch(kbeg, ra0, rb0);
        put(ra0, rb0, 0);
        if (ntile > 1) fetch(kbeg + HK, ra0, rb0);
    }
    __syncthreads();

    // per-lane operand rows of the MFMA fetch: row offset and swizzle key
    int offA[TM], keyA[TM], offB[TN], keyB[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int r = wm * WM + i * 32 + l31; offA[i] = r * 128; keyA[i] = lds_swz(r); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int r = wn * WN + j * 32 + l31; offB[j] = r * 128; keyB[j] = lds_swz(r); }

    int cur = 0;
    auto step = [&](int it, SA& pa, SB& pb, SA& qa, SB& qb) __attribute__((always_inline)) {
        if (it + 2 < ntile) fetch(kbeg + (it + 2) * HK, qa, qb);
        const char* sa = smA + cur * BM * 128;
        const char* sb = smB + cur * BN * 128;
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
        if (it + 1 < ntile) put(pa, pb, cur ^ 1);
        __syncthreads();
        cur ^= 1;
    };
    for (int it = 0; it < ntile; it += 2) {
        step(it, ra0, rb0, ra1, rb1);
        if (it + 1 < ntile) step(it + 1, ra1, rb1, ra0, rb0);
    }

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

    // epilogue: C/D layout of the 32x32 MFMA (dtype-independent): col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool relu = flags & FIRA_GEMM_RELU;
    const bool accum = flags & FIRA_GEMM_ACCUM;
    const bool add_bias = bias != nullptr && first;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WN + j * 32 + l31;
        if (col >= N) continue;
        const float bv = add_bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row >= M) continue;
                float v = acc[i][j][r] + bv;
                float* p = C + (size_t)(c_rows ? c_rows[row] : row) * ldc + col;
                if (atomic) {
                    unsafeAtomicAdd(p, v);
                } else {
                    if (accum) v += *p;
                    if (relu) v = fmaxf(v, 0.f);
                    if (relu_mask && !(relu_mask[(size_t)row * ldc + col] > 0.f)) v = 0.f;
                    *p = v;
                }
            }
        }
    }
}

template <int BM, int BN>
static int launch_bf16(hipStream_t s, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B,
                       int ldb, float* C, int ldc, const float* bias, int flags, int splitk, float* colsum,
                       const int32_t* c_rows, const float* relu_mask) {
    const int tiles_m = cdiv(M, BM), tiles_n = cdiv(N, BN);
    const int k_chunk = cdiv(cdiv(K, splitk), HK) * HK;
    // spread the operand with the larger footprint over the XCDs (its panels are then read from HBM once)
    const int spread_n = (!colsum && (long)N * K > (long)M * K) ? 1 : 0;
    const int groups = spread_n ? cdiv(tiles_n, 8) * tiles_m : cdiv(tiles_m, 8) * tiles_n;
    dim3 grid(8 * groups * splitk);
#define FIRA_GO(TA, TB)                                                                                           \
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, TA, TB>), grid, dim3(256), 0, s, M, N, K, A, lda, B, ldb, C, ldc, \
                       bias, flags, tiles_m, tiles_n, splitk, k_chunk, spread_n, colsum, c_rows, relu_mask)
    if (!tA && tB) FIRA_GO(false, true);
    else if (!tA && !tB) FIRA_GO(false, false);
    else if (tA && !tB) FIRA_GO(true, false);
    else FIRA_GO(true, true);
#undef FIRA_GO
    FIRA_CHECK_LAUNCH("gemm_bf16");
    return 0;
}

// Products this kernel does not take (handled by the fp32 kernels, i.e. computed more precisely, never less):
// unaligned operands, and the tiny ones (4-row mark table, 2-column gate) where a 64-wide tile is mostly padding.
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
    FIRA_REQUIRE(!(colsum && !tA), "gemm_bf16: fused column sums need the transA layout");
    ProfScope prof(s, PROF_GEMM, 2.0 * M * N * (double)K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    const bool can_split = (flags & FIRA_GEMM_ACCUM) && !(flags & FIRA_GEMM_RELU) && !relu_mask;
    int tile = ((flags >> FIRA_GEMM_TILE_SHIFT) & 3) - 1;       // -1: automatic; 0: 128x128; 1, 2: 64x64
    const long t128 = (long)cdiv(M, 128) * cdiv(N, 128), t64 = (long)cdiv(M, 64) * cdiv(N, 64);
    if (tile < 0) tile = (t128 >= 512 && !tA) ? 0 : 2;          // big tiles only when they still give every CU two
    if (splitk == 0) {
        // memory-bound: aim at ~2 workgroups per CU; every split re-reads nothing (disjoint K ranges) but adds one
        // atomic per output element, so split only reductions that are long compared with the tile
        splitk = 1;
        const long tiles = tile == 0 ? t128 : t64;
        if (can_split && tiles < 512 && K >= 1024) {
            const long want = (512 + tiles - 1) / tiles;
            splitk = (int)std::max(1L, std::min(want, (long)K / 512));
        }
    }
    FIRA_REQUIRE(!(splitk > 1 && !can_split), "gemm_bf16: split-K needs accumulate semantics and no relu");
    flags &= 3;
    if (tile == 0) return launch_bf16<128, 128>(s, tA, tB, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, splitk, colsum, c_rows, relu_mask);
    return launch_bf16<64, 64>(s, tA, tB, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, splitk, colsum, c_rows, relu_mask);
}

}  // namespace fira

extern "C" int fira_gemm_bf16(void* stream, int transA, int transB, int M, int N, int K, const float* A, int lda,
                              const float* B, int ldb, float* C, int ldc, const float* bias, int flags, int splitk) {
    FIRA_REQUIRE(splitk >= 0, "fira_gemm_bf16: splitk must be >= 0 (0 = automatic)");
    return fira::gemm_bf16_ex((hipStream_t)stream, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, splitk,
                              nullptr, nullptr, nullptr);
}
