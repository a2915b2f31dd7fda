// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fmaf chain).
//
// Replaces every nn.Linear forward/backward of the reference (addmm/mm rows of SURVEY.md §2.3):
//   forward   Y[M,N]  = X[M,K] · W[N,K]^T + b      transA=0, transB=1
//   dgrad     dX[M,K] = dY[M,N] · W[N,K]            transA=0, transB=0
//   wgrad     dW[N,K] += dY[M,N]^T · X[M,K]         transA=1, transB=0  (split-K + fp32 atomics)
//
// Structure: 256 threads = 4 waves in 2x2; block tile BMxBN (128x128 or 64x64), BK = 32;
// operands are staged global -> registers -> LDS in k-major order ([k][m], [k][n]) so that the MFMA
// operand fetch (lane l needs X[m0 + (l&31)][k + (l>>5)]) is one conflict-free ds_read_b32 for both
// storage orders; the register stage of tile t+1 is issued before the MFMAs of tile t (one barrier per
// K tile, two LDS buffers).  LDS row pitch 129 makes the transposing scalar writes of the k-contiguous
// path conflict-free (4*129 = 4 mod 32); the m-contiguous path writes 16-byte rows at pitch 132.
#include "engine.h"
#include "epilogue.h"
#include <cmath>
#include <stdlib.h>
#include <algorithm>

namespace fira {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// native 4-vector for the register stages (HIP's f32x4 is a union-based struct: conditionally written arrays of it are
// kept in scratch memory instead of registers)
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;

template <int ROWS, bool CONTIG_K>
struct TileLoader {
    // ROWS = BM or BN (extent along the non-reduction axis)
    static constexpr int NV = ROWS / 32;            // f32x4 per thread per tile
    static constexpr int LD = CONTIG_K ? ROWS + 1 : ROWS + 4;

    // source element (r, k): CONTIG_K ? src[r*ld + k] : src[k*ld + r]
    __device__ __forceinline__ static void load(f32x4 (&v)[NV], const float* __restrict__ src, int ld,
                                                int r0, int r_end, int k0, int k_end, bool vec_ok, int t) {
        if constexpr (CONTIG_K) {
            const int kq = (t & 7) * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int r = r0 + (t >> 3) + 32 * i;
                const int k = k0 + kq;
                f32x4 x = f32x4{0.f, 0.f, 0.f, 0.f};
                if (r < r_end) {
                    const float* p = src + (size_t)r * ld + k;
                    if (vec_ok && k + 3 < k_end) {
                        x = *reinterpret_cast<const f32x4*>(p);
                    } else {
                        if (k + 0 < k_end) x.x = p[0];
                        if (k + 1 < k_end) x.y = p[1];
                        if (k + 2 < k_end) x.z = p[2];
                        if (k + 3 < k_end) x.w = p[3];
                    }
                }
                v[i] = x;
            }
        } else {
            constexpr int CPR = ROWS / 4;           // f32x4 columns per k-row (32 or 16)
            constexpr int KSTEP = 256 / CPR;        // k-rows covered per pass (8 or 16)
            const int c = (t % CPR) * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int k = k0 + (t / CPR) + KSTEP * i;
                const int r = r0 + c;
                f32x4 x = f32x4{0.f, 0.f, 0.f, 0.f};
                if (k < k_end) {
                    const float* p = src + (size_t)k * ld + r;
                    if (vec_ok && r + 3 < r_end) {
                        x = *reinterpret_cast<const f32x4*>(p);
                    } else {
                        if (r + 0 < r_end) x.x = p[0];
                        if (r + 1 < r_end) x.y = p[1];
                        if (r + 2 < r_end) x.z = p[2];
                        if (r + 3 < r_end) x.w = p[3];
                    }
                }
                v[i] = x;
            }
        }
    }

    // interior tile (every row and k in range, 16-byte aligned rows): straight-line loads, no exec-mask branches, so
    // all NV 16-byte loads of a thread are in flight together
    __device__ __forceinline__ static void load_fast(f32x4 (&v)[NV], const float* __restrict__ src, int ld, int r0,
                                                     int k0, int t) {
        if constexpr (CONTIG_K) {
            const float* p = src + (size_t)(r0 + (t >> 3)) * ld + k0 + (t & 7) * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const f32x4*>(p + (size_t)32 * i * ld);
        } else {
            constexpr int CPR = ROWS / 4;
            constexpr int KSTEP = 256 / CPR;
            const float* p = src + (size_t)(k0 + t / CPR) * ld + r0 + (t % CPR) * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const f32x4*>(p + (size_t)KSTEP * i * ld);
        }
    }

    __device__ __forceinline__ static void store(const f32x4 (&v)[NV], float* __restrict__ lds, int t) {
        if constexpr (CONTIG_K) {
            const int kq = (t & 7) * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int r = (t >> 3) + 32 * i;
                lds[(kq + 0) * LD + r] = v[i].x;
                lds[(kq + 1) * LD + r] = v[i].y;
                lds[(kq + 2) * LD + r] = v[i].z;
                lds[(kq + 3) * LD + r] = v[i].w;
            }
        } else {
            constexpr int CPR = ROWS / 4;
            constexpr int KSTEP = 256 / CPR;
            const int c = (t % CPR) * 4;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int k = (t / CPR) + KSTEP * i;
                *reinterpret_cast<f32x4*>(&lds[k * LD + c]) = v[i];
            }
        }
    }
};

// One BM x BN output tile over the K range [kbeg, kend) by the 256 threads of a workgroup.  `atomic`: the tile is one of
// several K splits (partials combined with fp32 atomics); `first`: this split adds the bias; `cs_tile`: this workgroup
// owns the fused column sums of its A tile.
template <int BM, int BN, bool TA, bool TB>
__device__ __forceinline__ void gemm_tile(int M, int N, int K, const float* __restrict__ A, int lda,
                                          const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
                                          const float* __restrict__ bias, int flags, int vecA, int vecB,
                                          float* __restrict__ colsum, const int32_t* __restrict__ c_rows,
                                          const float* __restrict__ relu_mask, int m0, int n0, int kbeg, int kend,
                                          bool atomic, bool first, bool cs_tile) {
    using LA = TileLoader<BM, !TA>;       // A stored [M,K] (k contiguous) unless TA
    using LB = TileLoader<BN, TB>;        // B stored [N,K] (k contiguous) when TB
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    __shared__ __attribute__((aligned(16))) float smA[2][BK * LA::LD];
    __shared__ __attribute__((aligned(16))) float smB[2][BK * LB::LD];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntile = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fused bias gradient (wgrad layout only): colsum[m] += sum_k A[k][m] over this block's K range, taken from the
    // A-tile registers on their way to LDS (thread t always carries the same 4 columns of the tile)
    const bool do_cs = TA && colsum != nullptr && cs_tile;
    f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};
    auto cs_add = [&](const f32x4 (&v)[LA::NV]) {
#pragma unroll
        for (int i = 0; i < LA::NV; ++i) { cs.x += v[i].x; cs.y += v[i].y; cs.z += v[i].z; cs.w += v[i].w; }
    };

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const float* sa = smA[buf] + kh * LA::LD + wm * WM + l31;
        const float* sb = smB[buf] + kh * LB::LD + wn * WN + l31;
        // operand fragments of k-pair kk+1 are read from LDS before the MFMAs of k-pair kk are issued
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[0][i] = sa[i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[0][j] = sb[j * 32];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int c = kk & 1, n = c ^ 1;
            if (kk + 1 < BK / 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[n][i] = sa[2 * (kk + 1) * LA::LD + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[n][j] = sb[2 * (kk + 1) * LB::LD + j * 32];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
            // pin the order "LDS reads of the next k-pair, then this k-pair's MFMAs": the reads then complete under
            // the 64-cycle MFMAs instead of stalling the next group (the compiler otherwise sinks them below)
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
    };
    auto put = [&](const f32x4 (&ra)[LA::NV], const f32x4 (&rb)[LB::NV], int buf) __attribute__((always_inline)) {
        if (do_cs) cs_add(ra);
        LA::store(ra, smA[buf], t);
        LB::store(rb, smB[buf], t);
    };
    // Interior workgroups (every row, column and K tile in range: block-uniform, decided once) run a two-stage
    // register pipeline of straight 16-byte loads.  Tile t is fetched into stage t & 1 and stored to LDS buffer t & 1 one
    // step before it is consumed:
    //     step(t):  fetch(t + 2) -> stage t & 1 | MFMAs on LDS[t & 1] | store stage (t + 1) & 1 -> LDS[(t + 1) & 1]
    // With K = 256 a 64x64 workgroup runs only 8 steps of 16 MFMAs (~0.4 us) each, far less than a global-load latency.
    // The steady-state loop is unconditional and the last <= 3 tiles are straight-line code: a load behind a branch
    // (a conditional prefetch, or the per-tile fast/edge choice this kernel used to make) forces the compiler's
    // s_waitcnt insertion to merge both sides of the join, and it then drained the tiles in flight (vmcnt(0)) at the
    // head of every iteration -- 46 % of the wave cycles were parked there (profiles/r1e_instruction_mix.md).
    const bool interior = vecA && vecB && m0 + BM <= M && n0 + BN <= N && (kend - kbeg) % BK == 0 && ntile > 0;
    if (interior) {
        f32x4 ra0[LA::NV], rb0[LB::NV], ra1[LA::NV], rb1[LB::NV];
        auto fetch = [&](int k0, f32x4 (&ra)[LA::NV], f32x4 (&rb)[LB::NV]) __attribute__((always_inline)) {
            LA::load_fast(ra, A, lda, m0, k0, t);
            LB::load_fast(rb, B, ldb, n0, k0, t);
        };
        fetch(kbeg, ra0, rb0);
        put(ra0, rb0, 0);
        if (ntile == 1) {
            __syncthreads();
            compute(0);
        } else {
            fetch(kbeg + BK, ra1, rb1);
            __syncthreads();
            int it = 0;
            for (; it + 3 < ntile; it += 2) {                 // `it` stays even: stage / buffer roles are static
                fetch(kbeg + (it + 2) * BK, ra0, rb0);
                compute(0);
                put(ra1, rb1, 1);
                __syncthreads();
                fetch(kbeg + (it + 3) * BK, ra1, rb1);
                compute(1);
                put(ra0, rb0, 0);
                __syncthreads();
            }
            if (ntile - it == 3) {                            // tile it in LDS[0], tile it + 1 in flight (stage 1)
                fetch(kbeg + (it + 2) * BK, ra0, rb0);
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
        // edge workgroups (last row / column of tiles, unaligned operands, the split holding a partial K tile): guarded
        // loads, one register stage, no overlap
        f32x4 ra[LA::NV], rb[LB::NV];
        for (int it = 0; it < ntile; ++it) {
            LA::load(ra, A, lda, m0, M, kbeg + it * BK, kend, vecA != 0, t);
            LB::load(rb, B, ldb, n0, N, kbeg + it * BK, kend, vecB != 0, t);
            put(ra, rb, 0);
            __syncthreads();
            compute(0);
            __syncthreads();
        }
    }
    __syncthreads();

    if (do_cs) {                                   // block-level combine of the column sums, then one atomic per column
        float* red = smA[0];                       // all tile reads are done (barrier at the end of the last iteration)
        for (int i = t; i < BM; i += 256) red[i] = 0.f;
        __syncthreads();
        const int c = (t % (BM / 4)) * 4;
        atomicAdd(&red[c + 0], cs.x); atomicAdd(&red[c + 1], cs.y); atomicAdd(&red[c + 2], cs.z); atomicAdd(&red[c + 3], cs.w);
        __syncthreads();
        for (int i = t; i < BM; i += 256)
            if (m0 + i < M) unsafeAtomicAdd(&colsum[m0 + i], red[i]);
    }

    // epilogue (epilogue.h): C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
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

template <int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                       const float* __restrict__ B, int ldb, float* __restrict__ C,
                                                       int ldc, const float* __restrict__ bias, int flags,
                                                       int k_chunk, int vecA, int vecB, float* __restrict__ colsum,
                                                       const int32_t* __restrict__ c_rows,
                                                       const float* __restrict__ relu_mask, int tiles_m, int tiles_n,
                                                       int splitk, int spread_n, int chunk) {
    // Workgroup -> tile order (same rule as gemm_bf16.hip): workgroup b runs on XCD b % 8; logical item
    // i = (b % 8) * chunk + b / 8 gives every XCD a contiguous range of items ordered K split (slowest), tiles of the
    // larger operand, other dimension (fastest): weight-gradient K slabs and activation row panels are fetched from HBM
    // once per XCD and re-read from its private L2 by the neighbouring tiles.
    // (a capped grid -- FIRA_WGRAD_WGS, weight gradients only -- walks the items in strides of the grid: gridDim.x stays a
    // multiple of 8, so an item keeps its XCD)
    for (int b = blockIdx.x; b < 8 * chunk; b += gridDim.x) {
        const int i = (b & 7) * chunk + (b >> 3);
        if (i >= tiles_m * tiles_n * splitk) continue;
        const int per = tiles_m * tiles_n;
        const int z = i / per, r = i - z * per;
        int tm, tn;
        if (!spread_n) { tm = r / tiles_n; tn = r - tm * tiles_n; }
        else { tn = r / tiles_m; tm = r - tn * tiles_m; }
        const int kbeg = z * k_chunk;
        gemm_tile<BM, BN, TA, TB>(M, N, K, A, lda, B, ldb, C, ldc, bias, flags, vecA, vecB, colsum, c_rows, relu_mask,
                                  tm * BM, tn * BN, kbeg, min(K, kbeg + k_chunk), splitk > 1, z == 0, tn == 0);
        __syncthreads();                             // the next item restages the LDS tiles
    }
}

// Grouped weight gradients: dW_i (+)= dY_i^T X_i for up to GROUP_MAX independent problems in ONE launch.  The decoder's
// 36 weight gradients reduce over only B*30 target rows each (0.1-0.5 GFLOP): launched one by one they cost ~15 us
// apiece, mostly fill and drain; as one grid of ~2000 workgroups they run at the throughput of a single large GEMM.
// The problem table travels in the kernel arguments.
struct GroupProblem {
    const float* A;      // dY  [K, M] (transA layout)
    const float* B;      // X   [K, N]
    float* C;            // dW  [M, N], accumulated
    float* colsum;       // db  [M] or nullptr
    int M, N, K, lda, ldb, ldc, tiles_n, splitk, k_chunk;
};
constexpr int GROUP_MAX = 40;
struct GroupTable {
    int n;
    int wg_start[GROUP_MAX + 1];
    GroupProblem p[GROUP_MAX];
};
__global__ __launch_bounds__(256) void gemm_grouped_wgrad_kernel(GroupTable g) {
    const int total = g.wg_start[g.n];
    for (int b = blockIdx.x; b < total; b += gridDim.x) {                          // (one item per workgroup unless the grid is capped)
        int i = 0;
        while (i + 1 < g.n && b >= g.wg_start[i + 1]) ++i;                          // uniform scan of <= 40 entries
        const GroupProblem& q = g.p[i];
        const int w = b - g.wg_start[i];
        const int tile = w / q.splitk, z = w - tile * q.splitk;
        const int tm = tile / q.tiles_n, tn = tile - tm * q.tiles_n;
        const int kbeg = z * q.k_chunk;
        const int vecA = ((uintptr_t)q.A % 16 == 0) && (q.lda % 4 == 0), vecB = ((uintptr_t)q.B % 16 == 0) && (q.ldb % 4 == 0);
        gemm_tile<64, 64, true, false>(q.M, q.N, q.K, q.A, q.lda, q.B, q.ldb, q.C, q.ldc, nullptr, FIRA_GEMM_ACCUM, vecA, vecB,
                                       q.colsum, nullptr, nullptr, tm * 64, tn * 64, kbeg, min(q.K, kbeg + q.k_chunk),
                                       q.splitk > 1, z == 0, tn == 0);
        __syncthreads();
    }
}
// FIRA_WGRAD_WGS = n (experiment): weight-gradient launches use at most n workgroups (persistent over their tiles), so that
// they leave wave / LDS slots and memory bandwidth to the latency-bound kernels of the dependent chain they run beside
// Launches on the low-priority weight-gradient stream ask for FIRA_SIDE_LDS_PAD bytes of (unused) dynamic LDS on top of
// their tiles: four 34 KB workgroups fill a CU's 160 KB, and a workgroup of the dependent chain on the caller's stream
// (33 KB for the 32x32 tile kernel) then waits for one of them to END before it can be placed -- stream priority orders
// the dispatch of new workgroups, it frees nothing.  With the pad (default 7 KB) only three fit and the chain always finds
// room: +0.3 % step on one box (11 409 -> 11 446 commits/s over three pairs); 20 KB (two per CU) costs 0.8 %.
static hipStream_t g_pad_stream = nullptr;
void gemm_set_pad_stream(hipStream_t s) { g_pad_stream = s; }
static unsigned side_lds_pad(hipStream_t s) {
    static const int pad = [] { const char* e = getenv("FIRA_SIDE_LDS_PAD"); return e ? atoi(e) : 7168; }();
    return (pad > 0 && s && s == g_pad_stream) ? (unsigned)pad : 0u;
}

template <int BM, int BN>
static int launch(hipStream_t s, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                  float* C, int ldc, const float* bias, int flags, int splitk, float* colsum, const int32_t* c_rows,
                  const float* relu_mask) {
    const int tiles_m = cdiv(M, BM), tiles_n = cdiv(N, BN);
    const int spread_n = (long)N > (long)M ? 1 : 0;
    const int chunk = cdiv(tiles_m * tiles_n * splitk, 8);
    dim3 grid(8 * chunk);
    int k_chunk = cdiv(cdiv(K, splitk), BK) * BK;
    const int vecA = ((uintptr_t)A % 16 == 0) && (lda % 4 == 0);
    const int vecB = ((uintptr_t)B % 16 == 0) && (ldb % 4 == 0);
#define FIRA_GEMM_GO(TA, TB)                                                                                     \
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, TA, TB>), grid, dim3(256), side_lds_pad(s), s, M, N, K, A, lda, B, ldb, C, ldc, \
                       bias, flags, k_chunk, vecA, vecB, colsum, c_rows, relu_mask, tiles_m, tiles_n, splitk, spread_n, \
                       chunk)
    if (!tA && tB) FIRA_GEMM_GO(false, true);
    else if (!tA && !tB) FIRA_GEMM_GO(false, false);
    else if (tA && !tB) FIRA_GEMM_GO(true, false);
    else FIRA_GEMM_GO(true, true);
#undef FIRA_GEMM_GO
    FIRA_CHECK_LAUNCH("gemm_f32");
    return 0;
}

// Tile / split choice by a cost model fitted to the timings of scripts/gemm_step_shapes.py (profiles/r1e_gemm_shapes.md).
// A launch costs ceil(workgroups / 256 CUs) rounds of one workgroup's time: MFMA time at the tile's steady-state
// fraction of the CU's fp32 peak (614 GFLOP/s) plus a fill/drain cost that the co-resident workgroups of the small
// tile hide (4 of 64x64 fit a CU, 2 of 128x128) -- so with K = 256, where a workgroup runs only 8 k-steps, 64x64 wins
// on every shape of the training step, and 128x128 only pays off from K of a few thousand.  A split-K epilogue costs
// its BM*BN atomics (all 256 CUs drain into the same memory channels at once); a grid that does not fill the chip
// loses the overlap between co-resident workgroups.
struct TileChoice { int tile; int splitk; double cost; };
static TileChoice choose(int M, int N, int K, bool can_split) {
    static const int BMs[3] = {128, 64, 64}, BNs[3] = {128, 128, 64};
    static const double eff[3] = {0.72, 0.66, 0.60};
    static const double fill[3] = {6.0e-6, 2.0e-6, 0.0};
    TileChoice best{0, 1, 1e30};
    for (int t = 0; t < 3; ++t) {
        const long tiles = (long)cdiv(M, BMs[t]) * cdiv(N, BNs[t]);
        const int max_split = can_split ? std::max(1, K / 128) : 1;
        for (int sk = 1; sk <= max_split; sk = (sk < 4 ? sk + 1 : sk * 2)) {
            const double kk = (double)cdiv(cdiv(K, sk), BK) * BK;
            double t_wg = 2.0 * BMs[t] * BNs[t] * kk / (614e9 * eff[t]) + fill[t];
            if (sk > 1) t_wg += BMs[t] * BNs[t] * 0.83e-9;
            if (sk == 1 && tiles <= 256) t_wg *= 1.3;
            const double cost = std::ceil((double)(tiles * sk) / 256.0) * t_wg;
            if (cost < best.cost * 0.97) best = TileChoice{t, sk, cost};
        }
    }
    return best;
}

int gemm_f32_ex(hipStream_t s, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                float* C, int ldc, const float* bias, int flags, int splitk, float* colsum, const int32_t* c_rows,
                const float* relu_mask) {
    if (M <= 0 || N <= 0) return 0;
    FIRA_REQUIRE(!(relu_mask && (splitk > 1 || c_rows)), "gemm_f32: the fused ReLU mask needs a plain (unsplit, unmapped) output");
    FIRA_REQUIRE(K > 0 && splitk >= 0, "gemm_f32: bad K=%d splitk=%d", K, splitk);
    FIRA_REQUIRE(c_rows || epilogue_fits(M, ldc), "gemm_f32: output of %d x %d floats exceeds the 2 GiB the kernels address", M, ldc);
    FIRA_REQUIRE(!(colsum && !tA), "gemm_f32: fused column sums need the transA layout");
    ProfScope prof(s, PROF_GEMM, 2.0 * M * N * (double)K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    const bool can_split = (flags & FIRA_GEMM_ACCUM) && !(flags & FIRA_GEMM_RELU) && !relu_mask;
    int tile = ((flags >> FIRA_GEMM_TILE_SHIFT) & 3) - 1;       // -1: automatic
    if (tile < 0 && splitk <= 1 && !colsum) {                   // skinny forward / dgrad shapes: latency kernel
        int rc;
        if (gemm_small_try(s, tA, tB, M, N, K, A, lda, B, ldb, C, ldc, bias, flags & 3, &rc, c_rows, relu_mask)) return rc;
    }
    if (splitk == 0 || tile < 0) {
        const TileChoice c = choose(M, N, K, can_split && splitk == 0);
        if (tile < 0) tile = c.tile;
        if (splitk == 0) splitk = c.splitk;
    }
    FIRA_REQUIRE(!(splitk > 1 && !can_split), "gemm_f32: split-K needs accumulate semantics and no relu");
    flags &= 3;
    if (tile == 0) return launch<128, 128>(s, tA, tB, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, splitk, colsum, c_rows, relu_mask);
    if (tile == 1) return launch<64, 128>(s, tA, tB, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, splitk, colsum, c_rows, relu_mask);
    return launch<64, 64>(s, tA, tB, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, splitk, colsum, c_rows, relu_mask);
}

// ---- grouped weight gradients (host side): problems are collected, then launched together
struct GroupBuilder {
    GroupTable t;
    GroupBuilder() { t.n = 0; t.wg_start[0] = 0; }
};
static GroupBuilder& group() { static thread_local GroupBuilder g; return g; }

void gemm_group_reset() { group().t.n = 0; }
bool gemm_group_full() { return group().t.n == GROUP_MAX; }
int gemm_group_flush(hipStream_t s) {
    GroupTable& t = group().t;
    if (t.n == 0) return 0;
    double flop = 0, bytes = 0;
    for (int i = 0; i < t.n; ++i) {
        const double M = t.p[i].M, N = t.p[i].N, K = t.p[i].K;
        flop += 2.0 * M * N * K;
        bytes += 4.0 * (M * K + N * K + M * N);
    }
    ProfScope prof(s, PROF_GEMM, flop, bytes);
    hipLaunchKernelGGL(gemm_grouped_wgrad_kernel, dim3(t.wg_start[t.n]), dim3(256), side_lds_pad(s), s, t);
    t.n = 0;
    FIRA_CHECK_LAUNCH("gemm_grouped_wgrad");
    return 0;
}
// dW[M,N] += A^T B with A = dY [K,M], B = X [K,N]; db[M] += column sums of dY.  Queued; runs at the next flush on `s`.
int gemm_group_add_wgrad(hipStream_t s, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                         int ldc, float* colsum, int max_split) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    GroupTable& t = group().t;
    if (t.n == GROUP_MAX) {
        int rc = gemm_group_flush(s);
        if (rc) return rc;
    }
    GroupProblem& q = t.p[t.n];
    q.A = A; q.B = B; q.C = C; q.colsum = colsum;
    q.M = M; q.N = N; q.K = K; q.lda = lda; q.ldb = ldb; q.ldc = ldc;
    q.tiles_n = cdiv(N, 64);
    // the whole group shares the chip: two K splits per tile keep the tail short without drowning the result in atomics
    q.splitk = K >= 512 ? 2 : 1;
    if (max_split > 0) q.splitk = std::max(1, std::min(max_split, K / 512));
    q.k_chunk = cdiv(cdiv(K, q.splitk), BK) * BK;
    t.wg_start[t.n + 1] = t.wg_start[t.n] + cdiv(M, 64) * q.tiles_n * q.splitk;
    ++t.n;
    return 0;
}

int gemm_f32(hipStream_t s, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
             float* C, int ldc, const float* bias, int flags, int splitk) {
    return gemm_f32_ex(s, tA, tB, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, splitk, nullptr, nullptr, nullptr);
}

}  // namespace fira

extern "C" int fira_gemm_f32(void* stream, int transA, int transB, int M, int N, int K, const float* A, int lda,
                             const float* B, int ldb, float* C, int ldc, const float* bias, int flags, int splitk) {
    FIRA_REQUIRE(splitk >= 0, "fira_gemm_f32: splitk must be >= 0 (0 = automatic)");
    return fira::gemm_f32((hipStream_t)stream, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, splitk);
}
