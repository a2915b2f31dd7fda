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
typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// ---------------------------------------------------------------------------------------------------------------------
// Round 3: the same decomposition (one 32x32 output tile per workgroup, the four wavefronts split K in chunks of 32) with
// COALESCED operand traffic.  The kernel above fetches MFMA fragments straight from global memory: lane (i, kh) reads
// 64 bytes of row i, so every load instruction touches 32 cache lines and uses 32 bytes of each (8.6 % of the fp32 MFMA
// peak on [960,256,256]).  Here a chunk is 32 rows x 128 bytes = 32 whole cache lines per operand, read by
// 4 + 4 wave-wide 16-byte loads (lane -> row 8j + lane/8, 16-byte slot lane%8: 8 full lines per instruction), parked in
// a wave-private 8 KB LDS slot and re-read in MFMA fragment layout:
//   A / k-contiguous B:  row r at r*128 bytes, global 16-byte slot g stored at slot g ^ ((r>>1)&7): the 8-lane groups of
//                        the ds_write_b128 and the 16-lane groups of the ds_read_b128 fragment fetch are conflict-free
//   B stored [K,N]:      k-row at k*128 bytes (the tile's 32 columns), fragment fetch by ds_read_b32 (bank = column)
// MFMA step t of a chunk uses k = (lane>>5)*16 + t for both operands.  A wave touches only its own slot and its own
// loads (up to two chunks in flight, the next one requested before the current one's MFMAs): no barrier in the K loop;
// the four partial tiles are combined through the same LDS.  Tile order is XCD-aware (block b runs on XCD b % 8): every
// XCD gets a contiguous range of row blocks with all their column tiles, so an activation panel is fetched by one L2.
// a_rows (optional): A row r is read from row a_rows[r] (a gather folded into the product).
__device__ __forceinline__ int swz8(int row) { return (row >> 1) & 7; }

struct ChunkRegs { f32x4 a0, a1, a2, a3, b0, b1, b2, b3; };

template <bool B_KCONTIG>
__device__ __forceinline__ void tile32_fetch(ChunkRegs& R, const float* pa0, const float* pa1, const float* pa2,
                                             const float* pa3, const float* pb0, const float* pb1, const float* pb2,
                                             const float* pb3, int kc, int ldb) {
    const size_t ob = B_KCONTIG ? (size_t)kc : (size_t)kc * ldb;
    R.a0 = *reinterpret_cast<const f32x4*>(pa0 + kc);
    R.a1 = *reinterpret_cast<const f32x4*>(pa1 + kc);
    R.a2 = *reinterpret_cast<const f32x4*>(pa2 + kc);
    R.a3 = *reinterpret_cast<const f32x4*>(pa3 + kc);
    R.b0 = *reinterpret_cast<const f32x4*>(pb0 + ob);
    R.b1 = *reinterpret_cast<const f32x4*>(pb1 + ob);
    R.b2 = *reinterpret_cast<const f32x4*>(pb2 + ob);
    R.b3 = *reinterpret_cast<const f32x4*>(pb3 + ob);
}

#define FIRA_MFMA4(va, vb)                                                    \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va.x, vb.x, acc, 0, 0, 0);     \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va.y, vb.y, acc, 0, 0, 0);     \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va.z, vb.z, acc, 0, 0, 0);     \
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va.w, vb.w, acc, 0, 0, 0);

// park one chunk in the wave's slot ...
__device__ __forceinline__ void tile32_stage(const ChunkRegs& R, char* slot, int wr_off) {
    __builtin_amdgcn_wave_barrier();                  // the previous chunk's fragment reads precede these stores
    *reinterpret_cast<f32x4*>(slot + wr_off) = R.a0;
    *reinterpret_cast<f32x4*>(slot + wr_off + 1024) = R.a1;
    *reinterpret_cast<f32x4*>(slot + wr_off + 2048) = R.a2;
    *reinterpret_cast<f32x4*>(slot + wr_off + 3072) = R.a3;
    *reinterpret_cast<f32x4*>(slot + 4096 + wr_off) = R.b0;
    *reinterpret_cast<f32x4*>(slot + 4096 + wr_off + 1024) = R.b1;
    *reinterpret_cast<f32x4*>(slot + 4096 + wr_off + 2048) = R.b2;
    *reinterpret_cast<f32x4*>(slot + 4096 + wr_off + 3072) = R.b3;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// ... re-read it in fragment layout, 16 MFMA steps
template <bool B_KCONTIG>
__device__ __forceinline__ f32x16 tile32_compute(const char* slot, int fa0, int fa1, int fa2, int fa3, int fb, f32x16 acc) {
    const f32x4 va0 = *reinterpret_cast<const f32x4*>(slot + fa0);
    const f32x4 va1 = *reinterpret_cast<const f32x4*>(slot + fa1);
    const f32x4 va2 = *reinterpret_cast<const f32x4*>(slot + fa2);
    const f32x4 va3 = *reinterpret_cast<const f32x4*>(slot + fa3);
    if (B_KCONTIG) {
        const f32x4 vb0 = *reinterpret_cast<const f32x4*>(slot + 4096 + fa0);
        const f32x4 vb1 = *reinterpret_cast<const f32x4*>(slot + 4096 + fa1);
        const f32x4 vb2 = *reinterpret_cast<const f32x4*>(slot + 4096 + fa2);
        const f32x4 vb3 = *reinterpret_cast<const f32x4*>(slot + 4096 + fa3);
        FIRA_MFMA4(va0, vb0) FIRA_MFMA4(va1, vb1) FIRA_MFMA4(va2, vb2) FIRA_MFMA4(va3, vb3)
    } else {
        const char* pb = slot + 4096 + fb;            // k-row (lane>>5)*16, column lane&31; one k-row = 128 bytes
        f32x4 vb0, vb1, vb2, vb3;
        vb0.x = *reinterpret_cast<const float*>(pb);        vb0.y = *reinterpret_cast<const float*>(pb + 128);
        vb0.z = *reinterpret_cast<const float*>(pb + 256);  vb0.w = *reinterpret_cast<const float*>(pb + 384);
        vb1.x = *reinterpret_cast<const float*>(pb + 512);  vb1.y = *reinterpret_cast<const float*>(pb + 640);
        vb1.z = *reinterpret_cast<const float*>(pb + 768);  vb1.w = *reinterpret_cast<const float*>(pb + 896);
        vb2.x = *reinterpret_cast<const float*>(pb + 1024); vb2.y = *reinterpret_cast<const float*>(pb + 1152);
        vb2.z = *reinterpret_cast<const float*>(pb + 1280); vb2.w = *reinterpret_cast<const float*>(pb + 1408);
        vb3.x = *reinterpret_cast<const float*>(pb + 1536); vb3.y = *reinterpret_cast<const float*>(pb + 1664);
        vb3.z = *reinterpret_cast<const float*>(pb + 1792); vb3.w = *reinterpret_cast<const float*>(pb + 1920);
        FIRA_MFMA4(va0, vb0) FIRA_MFMA4(va1, vb1) FIRA_MFMA4(va2, vb2) FIRA_MFMA4(va3, vb3)
    }
    return acc;
}

// NCH = chunks per wave (K = 128 * NCH): the chunk loop is unrolled at compile time, so the two register sets of the
// loads in flight are named registers and the accumulator never leaves the AGPRs
// LN_A (K = 256 only): the A operand is the PRE-NORM sum s of a residual block (EpiRes of the producing product); the kernel
// normalises its 32 complete rows itself before the MFMAs -- a = (s - mean) * rstd * gamma + beta, two-pass statistics like
// add_layernorm_fwd (wave partials over the wave's 64 columns, combined through LDS: two barriers) -- and the workgroups of
// column tile 0 store the normalised rows (x_out: the block's output, read by later kernels as residual / saved activation)
// and the row statistics (stats_out: mean, rstd for the backward pass).  This replaces the LayerNorm launch between two
// products of the dependent chain (gnn_transformer.py:161 -> :147, :174 -> :170).
struct LnA {
    const float* gamma = nullptr;
    const float* beta = nullptr;
    float* x_out = nullptr;          // [M, 256]
    float* stats_out = nullptr;      // [M, 2] or nullptr
};

// LN_B (K = 256 only, round 4): the LayerNorm BACKWARD of a residual block in the prologue of the data-gradient product that
// consumes it (gnn_transformer.py:161,174 backward: the decoder's 18 LayerNorm-backward launches per step sat between two
// products of the dependent chain).  A = dy, the gradient w.r.t. the block's output rows; with the saved pre-norm rows s
// and their statistics the kernel forms, for its 32 complete rows,
//     xh = (s - mean) rstd,  h = dy gamma,  ds = (h - mean_k(h) - xh mean_k(h xh)) rstd,  dx = ds * dropout mask
// (one pass: both row means from the same registers, combined over the four waves through LDS) and multiplies dx into the
// product.  The workgroups of column tile 0 also publish ds (the residual-branch gradient), dx (the weight gradient's
// operand) and the row block's partial column sums {sum dy xh | sum dy} for dgamma / dbeta (deferred reduction).  ds must
// not alias dy: the other column tiles of the row block read dy while tile 0 writes ds.
struct LnB {
    const float* sum = nullptr;      // [M, 256] saved pre-norm rows
    const float* stats = nullptr;    // [M, 2] mean, rstd
    const float* gamma = nullptr;
    float* ds = nullptr;             // [M, 256]
    float* dx = nullptr;             // [M, 256]
    float* part = nullptr;           // [ceil(M / 32), 512] partial {dgamma | dbeta}
    float p = 0.f, inv_keep = 1.f;
    uint64_t seed = 0;
    uint32_t site = 0;
    uint32_t idx0 = 0;               // dropout element index of row 0, column 0 (a row slice of the site: engine.hip lanes)
};

// NW (round 5): wavefronts per workgroup.  The K chunks go round-robin over the waves, so a long reduction on a SMALL grid
// (the FFN's K = 1024 products at M = 64 .. 600 rows: 16 .. 136 workgroups on 256 CUs) can be cut into 2 chunks per wave by
// 8 / 12 / 16 waves instead of 4 / 6 / 8 serial rounds of (stage, 16 MFMAs) on 4: the workgroup's dependent chain is what such
// a launch costs.  The wave-private staging slots then take NW x 8 KB of DYNAMIC LDS (128 KB at 16 waves: one workgroup per CU,
// so only grids of at most 256 workgroups take that shape; tile32_waves()).
template <bool B_KCONTIG, int NCH, bool LN_A = false, bool LN_B = false, int NW = 4>
__global__ __launch_bounds__(NW * 64) void gemm_tile32_kernel(int M, int N, const float* __restrict__ A, int lda,
                                                          const float* __restrict__ B, int ldb, float* __restrict__ C,
                                                          int ldc, const float* __restrict__ bias, int flags,
                                                          const int32_t* __restrict__ c_rows,
                                                          const float* __restrict__ relu_mask,
                                                          const int32_t* __restrict__ a_rows, int tiles_n, const EpiRes er,
                                                          const LnA ln, const LnB lb) {
    static_assert(NW == 4 || (!LN_A && !LN_B), "the LayerNorm prologues are written for four waves");
    extern __shared__ __attribute__((aligned(16))) char tile32_dyn[];
    __shared__ __attribute__((aligned(16))) char smem4[NW == 4 ? 4 * 8192 : 16];
    char* const smem = NW == 4 ? smem4 : tile32_dyn;
    __shared__ float ln_red[2][4][32];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-chunked tile order (bijective for any grid size)
    int t;
    {
        const int b = blockIdx.x, nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = b & 7;
        t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    const int m0 = (t / tiles_n) * 32, n0 = (t % tiles_n) * 32;
    char* slot = smem + wave * 8192;
    const int lr = lane >> 3, ls = lane & 7;
    // row 8j + lr of the tile, j = 0..3: swz8(8j + lr) = (4j + (lr>>1)) & 7
    const int sw_even = lr >> 1, sw_odd = 4 + (lr >> 1);
    int ra0 = min(m0 + lr, M - 1), ra1 = min(m0 + 8 + lr, M - 1), ra2 = min(m0 + 16 + lr, M - 1), ra3 = min(m0 + 24 + lr, M - 1);
    if (a_rows) { ra0 = a_rows[ra0]; ra1 = a_rows[ra1]; ra2 = a_rows[ra2]; ra3 = a_rows[ra3]; }   // clamped rows: masked at the store
    const float* pa0 = A + (size_t)ra0 * lda + ((ls ^ sw_even) << 2);
    const float* pa1 = A + (size_t)ra1 * lda + ((ls ^ sw_odd) << 2);
    const float* pa2 = A + (size_t)ra2 * lda + ((ls ^ sw_even) << 2);
    const float* pa3 = A + (size_t)ra3 * lda + ((ls ^ sw_odd) << 2);
    const float *pb0, *pb1, *pb2, *pb3;
    if (B_KCONTIG) {
        pb0 = B + (size_t)min(n0 + lr, N - 1) * ldb + ((ls ^ sw_even) << 2);
        pb1 = B + (size_t)min(n0 + 8 + lr, N - 1) * ldb + ((ls ^ sw_odd) << 2);
        pb2 = B + (size_t)min(n0 + 16 + lr, N - 1) * ldb + ((ls ^ sw_even) << 2);
        pb3 = B + (size_t)min(n0 + 24 + lr, N - 1) * ldb + ((ls ^ sw_odd) << 2);
    } else {
        pb0 = B + (size_t)lr * ldb + n0 + (ls << 2);
        pb1 = pb0 + (size_t)8 * ldb;
        pb2 = pb0 + (size_t)16 * ldb;
        pb3 = pb0 + (size_t)24 * ldb;
    }
    const int i31 = lane & 31, kh = lane >> 5;
    const int wr_off = lr * 128 + ls * 16;
    const int sw_i = swz8(i31);
    const int fa0 = i31 * 128 + (((kh * 4 + 0) ^ sw_i) << 4), fa1 = i31 * 128 + (((kh * 4 + 1) ^ sw_i) << 4);
    const int fa2 = i31 * 128 + (((kh * 4 + 2) ^ sw_i) << 4), fa3 = i31 * 128 + (((kh * 4 + 3) ^ sw_i) << 4);
    const int fb = kh * 16 * 128 + i31 * 4;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    ChunkRegs R0, R1;
    tile32_fetch<B_KCONTIG>(R0, pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3, wave << 5, ldb);
    if (NCH > 1) tile32_fetch<B_KCONTIG>(R1, pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3, (wave + NW) << 5, ldb);
    if (LN_A) {
        static_assert(!LN_A || NCH == 2, "the LayerNorm prologue needs the whole row in the two chunks of the four waves");
        // this lane holds, of rows 8j + lr (j = 0..3), the four columns kc + ((ls ^ sw_j) << 2) .. + 3 of each chunk
        auto hsum = [](const f32x4 v) { return (v.x + v.y) + (v.z + v.w); };
        // columns of this lane: chunk c (k base kc = (wave + 4c) * 32), rows j even -> ls ^ sw_even, j odd -> ls ^ sw_odd.
        // gamma / beta are requested HERE, with the operand tiles still in flight (round 4: they used to be loaded after the two
        // statistics barriers -- one more dependent memory round trip in every LayerNorm-prologue product of the chain)
        const int ce = (ls ^ sw_even) << 2, co = (ls ^ sw_odd) << 2;
        const int k0 = wave << 5, k1 = (wave + 4) << 5;
        const f32x4 g0e = *reinterpret_cast<const f32x4*>(ln.gamma + k0 + ce), g0o = *reinterpret_cast<const f32x4*>(ln.gamma + k0 + co);
        const f32x4 g1e = *reinterpret_cast<const f32x4*>(ln.gamma + k1 + ce), g1o = *reinterpret_cast<const f32x4*>(ln.gamma + k1 + co);
        const f32x4 b0e = *reinterpret_cast<const f32x4*>(ln.beta + k0 + ce), b0o = *reinterpret_cast<const f32x4*>(ln.beta + k0 + co);
        const f32x4 b1e = *reinterpret_cast<const f32x4*>(ln.beta + k1 + ce), b1o = *reinterpret_cast<const f32x4*>(ln.beta + k1 + co);
        asm volatile("" ::: "memory");             // (keep the requests above the first reduction: the scheduler sinks loads)
        float mean[4], rstd[4];
        {
            const float p0 = sum8(hsum(R0.a0) + hsum(R1.a0)), p1 = sum8(hsum(R0.a1) + hsum(R1.a1));
            const float p2 = sum8(hsum(R0.a2) + hsum(R1.a2)), p3 = sum8(hsum(R0.a3) + hsum(R1.a3));
            if (ls == 0) { ln_red[0][wave][lr] = p0; ln_red[0][wave][8 + lr] = p1; ln_red[0][wave][16 + lr] = p2; ln_red[0][wave][24 + lr] = p3; }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j)
                mean[j] = ((ln_red[0][0][8 * j + lr] + ln_red[0][1][8 * j + lr]) + (ln_red[0][2][8 * j + lr] + ln_red[0][3][8 * j + lr])) *
                          (1.0f / FIRA_D);
        }
        {
            auto sq = [](const f32x4 v, float m) { const f32x4 d = v - m; return (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w); };
            const float p0 = sum8(sq(R0.a0, mean[0]) + sq(R1.a0, mean[0])), p1 = sum8(sq(R0.a1, mean[1]) + sq(R1.a1, mean[1]));
            const float p2 = sum8(sq(R0.a2, mean[2]) + sq(R1.a2, mean[2])), p3 = sum8(sq(R0.a3, mean[3]) + sq(R1.a3, mean[3]));
            if (ls == 0) { ln_red[1][wave][lr] = p0; ln_red[1][wave][8 + lr] = p1; ln_red[1][wave][16 + lr] = p2; ln_red[1][wave][24 + lr] = p3; }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float var = ((ln_red[1][0][8 * j + lr] + ln_red[1][1][8 * j + lr]) + (ln_red[1][2][8 * j + lr] + ln_red[1][3][8 * j + lr])) *
                                  (1.0f / FIRA_D);
                rstd[j] = 1.0f / sqrtf(var + 1e-5f);
            }
        }
        R0.a0 = (R0.a0 - mean[0]) * rstd[0] * g0e + b0e; R1.a0 = (R1.a0 - mean[0]) * rstd[0] * g1e + b1e;
        R0.a1 = (R0.a1 - mean[1]) * rstd[1] * g0o + b0o; R1.a1 = (R1.a1 - mean[1]) * rstd[1] * g1o + b1o;
        R0.a2 = (R0.a2 - mean[2]) * rstd[2] * g0e + b0e; R1.a2 = (R1.a2 - mean[2]) * rstd[2] * g1e + b1e;
        R0.a3 = (R0.a3 - mean[3]) * rstd[3] * g0o + b0o; R1.a3 = (R1.a3 - mean[3]) * rstd[3] * g1o + b1o;
        if (n0 == 0) {                              // column tile 0 publishes the block's output rows and statistics
            const int r0 = m0 + lr, r1 = m0 + 8 + lr, r2 = m0 + 16 + lr, r3 = m0 + 24 + lr;
            float* xo = ln.x_out;
            if (r0 < M) { *reinterpret_cast<f32x4*>(xo + (size_t)r0 * FIRA_D + k0 + ce) = R0.a0; *reinterpret_cast<f32x4*>(xo + (size_t)r0 * FIRA_D + k1 + ce) = R1.a0; }
            if (r1 < M) { *reinterpret_cast<f32x4*>(xo + (size_t)r1 * FIRA_D + k0 + co) = R0.a1; *reinterpret_cast<f32x4*>(xo + (size_t)r1 * FIRA_D + k1 + co) = R1.a1; }
            if (r2 < M) { *reinterpret_cast<f32x4*>(xo + (size_t)r2 * FIRA_D + k0 + ce) = R0.a2; *reinterpret_cast<f32x4*>(xo + (size_t)r2 * FIRA_D + k1 + ce) = R1.a2; }
            if (r3 < M) { *reinterpret_cast<f32x4*>(xo + (size_t)r3 * FIRA_D + k0 + co) = R0.a3; *reinterpret_cast<f32x4*>(xo + (size_t)r3 * FIRA_D + k1 + co) = R1.a3; }
            if (ln.stats_out && wave == 0 && ls == 0) {
                if (r0 < M) { ln.stats_out[2 * r0] = mean[0]; ln.stats_out[2 * r0 + 1] = rstd[0]; }
                if (r1 < M) { ln.stats_out[2 * r1] = mean[1]; ln.stats_out[2 * r1 + 1] = rstd[1]; }
                if (r2 < M) { ln.stats_out[2 * r2] = mean[2]; ln.stats_out[2 * r2 + 1] = rstd[2]; }
                if (r3 < M) { ln.stats_out[2 * r3] = mean[3]; ln.stats_out[2 * r3 + 1] = rstd[3]; }
            }
        }
    }
    if constexpr (LN_B) {
        static_assert(!LN_B || (NCH == 2 && !LN_A), "the LayerNorm-backward prologue needs the whole row in the two chunks of the four waves");
        auto hsum = [](const f32x4 v) { return (v.x + v.y) + (v.z + v.w); };
        const int ce = (ls ^ sw_even) << 2, co = (ls ^ sw_odd) << 2;
        const int k0 = wave << 5, k1 = (wave + 4) << 5;
        // the saved pre-norm rows at the positions of the A registers, the statistics of the lane's four rows
        const float* ps0 = lb.sum + (size_t)ra0 * FIRA_D; const float* ps1 = lb.sum + (size_t)ra1 * FIRA_D;
        const float* ps2 = lb.sum + (size_t)ra2 * FIRA_D; const float* ps3 = lb.sum + (size_t)ra3 * FIRA_D;
        f32x4 s00 = *reinterpret_cast<const f32x4*>(ps0 + k0 + ce), s10 = *reinterpret_cast<const f32x4*>(ps0 + k1 + ce);
        f32x4 s01 = *reinterpret_cast<const f32x4*>(ps1 + k0 + co), s11 = *reinterpret_cast<const f32x4*>(ps1 + k1 + co);
        f32x4 s02 = *reinterpret_cast<const f32x4*>(ps2 + k0 + ce), s12 = *reinterpret_cast<const f32x4*>(ps2 + k1 + ce);
        f32x4 s03 = *reinterpret_cast<const f32x4*>(ps3 + k0 + co), s13 = *reinterpret_cast<const f32x4*>(ps3 + k1 + co);
        const float mean0 = lb.stats[2 * ra0], rstd0 = lb.stats[2 * ra0 + 1], mean1 = lb.stats[2 * ra1], rstd1 = lb.stats[2 * ra1 + 1];
        const float mean2 = lb.stats[2 * ra2], rstd2 = lb.stats[2 * ra2 + 1], mean3 = lb.stats[2 * ra3], rstd3 = lb.stats[2 * ra3 + 1];
        const f32x4 g0e = *reinterpret_cast<const f32x4*>(lb.gamma + k0 + ce), g0o = *reinterpret_cast<const f32x4*>(lb.gamma + k0 + co);
        const f32x4 g1e = *reinterpret_cast<const f32x4*>(lb.gamma + k1 + ce), g1o = *reinterpret_cast<const f32x4*>(lb.gamma + k1 + co);
        // xh (over s), h (kept), then the two row sums
        s00 = (s00 - mean0) * rstd0; s10 = (s10 - mean0) * rstd0; s01 = (s01 - mean1) * rstd1; s11 = (s11 - mean1) * rstd1;
        s02 = (s02 - mean2) * rstd2; s12 = (s12 - mean2) * rstd2; s03 = (s03 - mean3) * rstd3; s13 = (s13 - mean3) * rstd3;
        const f32x4 h00 = R0.a0 * g0e, h10 = R1.a0 * g1e, h01 = R0.a1 * g0o, h11 = R1.a1 * g1o;
        const f32x4 h02 = R0.a2 * g0e, h12 = R1.a2 * g1e, h03 = R0.a3 * g0o, h13 = R1.a3 * g1o;
        {
            const float p0 = sum8(hsum(h00) + hsum(h10)), p1 = sum8(hsum(h01) + hsum(h11));
            const float p2 = sum8(hsum(h02) + hsum(h12)), p3 = sum8(hsum(h03) + hsum(h13));
            const float q0 = sum8(hsum(h00 * s00) + hsum(h10 * s10)), q1 = sum8(hsum(h01 * s01) + hsum(h11 * s11));
            const float q2 = sum8(hsum(h02 * s02) + hsum(h12 * s12)), q3 = sum8(hsum(h03 * s03) + hsum(h13 * s13));
            if (ls == 0) {
                ln_red[0][wave][lr] = p0; ln_red[0][wave][8 + lr] = p1; ln_red[0][wave][16 + lr] = p2; ln_red[0][wave][24 + lr] = p3;
                ln_red[1][wave][lr] = q0; ln_red[1][wave][8 + lr] = q1; ln_red[1][wave][16 + lr] = q2; ln_red[1][wave][24 + lr] = q3;
            }
        }
        if (lb.part && n0 == 0) {
            // partial dgamma / dbeta of this row block: the wave's 32 rows x 64 columns of dy xh (then dy) go through its private
            // LDS slot ([row][64 columns]), lane L sums column (wave + 4 (L >> 5)) * 32 + (L & 31) over the rows.  Rows past M
            // were clamped to row M - 1 and must not count: their dy is zeroed here.
            const bool v0 = m0 + lr < M, v1 = m0 + 8 + lr < M, v2 = m0 + 16 + lr < M, v3 = m0 + 24 + lr < M;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            const f32x4 d00 = v0 ? R0.a0 : z, d10 = v0 ? R1.a0 : z, d01 = v1 ? R0.a1 : z, d11 = v1 ? R1.a1 : z;
            const f32x4 d02 = v2 ? R0.a2 : z, d12 = v2 ? R1.a2 : z, d03 = v3 ? R0.a3 : z, d13 = v3 ? R1.a3 : z;
            float* sc = reinterpret_cast<float*>(slot);
            float colsum[2];
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                __builtin_amdgcn_wave_barrier();
                *reinterpret_cast<f32x4*>(sc + (lr) * 64 + ce) = pass ? d00 : d00 * s00;
                *reinterpret_cast<f32x4*>(sc + (lr) * 64 + 32 + ce) = pass ? d10 : d10 * s10;
                *reinterpret_cast<f32x4*>(sc + (8 + lr) * 64 + co) = pass ? d01 : d01 * s01;
                *reinterpret_cast<f32x4*>(sc + (8 + lr) * 64 + 32 + co) = pass ? d11 : d11 * s11;
                *reinterpret_cast<f32x4*>(sc + (16 + lr) * 64 + ce) = pass ? d02 : d02 * s02;
                *reinterpret_cast<f32x4*>(sc + (16 + lr) * 64 + 32 + ce) = pass ? d12 : d12 * s12;
                *reinterpret_cast<f32x4*>(sc + (24 + lr) * 64 + co) = pass ? d03 : d03 * s03;
                *reinterpret_cast<f32x4*>(sc + (24 + lr) * 64 + 32 + co) = pass ? d13 : d13 * s13;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int r = 0; r < 32; r += 2) { a0 += sc[r * 64 + lane]; a1 += sc[(r + 1) * 64 + lane]; }
                colsum[pass] = a0 + a1;
            }
            const int col = ((wave + 4 * (lane >> 5)) << 5) + (lane & 31);
            float* pp = lb.part + (size_t)(m0 >> 5) * (2 * FIRA_D);
            pp[col] = colsum[0];
            pp[FIRA_D + col] = colsum[1];
        }
        __syncthreads();
        float m1[4], m2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            m1[j] = ((ln_red[0][0][8 * j + lr] + ln_red[0][1][8 * j + lr]) + (ln_red[0][2][8 * j + lr] + ln_red[0][3][8 * j + lr])) * (1.0f / FIRA_D);
            m2[j] = ((ln_red[1][0][8 * j + lr] + ln_red[1][1][8 * j + lr]) + (ln_red[1][2][8 * j + lr] + ln_red[1][3][8 * j + lr])) * (1.0f / FIRA_D);
        }
        // ds over the h registers, dx = ds * mask over the A registers
        const f32x4 e00 = (h00 - m1[0] - s00 * m2[0]) * rstd0, e10 = (h10 - m1[0] - s10 * m2[0]) * rstd0;
        const f32x4 e01 = (h01 - m1[1] - s01 * m2[1]) * rstd1, e11 = (h11 - m1[1] - s11 * m2[1]) * rstd1;
        const f32x4 e02 = (h02 - m1[2] - s02 * m2[2]) * rstd2, e12 = (h12 - m1[2] - s12 * m2[2]) * rstd2;
        const f32x4 e03 = (h03 - m1[3] - s03 * m2[3]) * rstd3, e13 = (h13 - m1[3] - s13 * m2[3]) * rstd3;
        auto drop = [&](const f32x4 v, int row, int col) {
            if (lb.p <= 0.f) return v;
            const uint32_t e0 = lb.idx0 + (uint32_t)row * FIRA_D + (uint32_t)col;
            return f32x4{v.x * dropout_scale(lb.seed, lb.site, e0, lb.p, lb.inv_keep), v.y * dropout_scale(lb.seed, lb.site, e0 + 1, lb.p, lb.inv_keep),
                         v.z * dropout_scale(lb.seed, lb.site, e0 + 2, lb.p, lb.inv_keep), v.w * dropout_scale(lb.seed, lb.site, e0 + 3, lb.p, lb.inv_keep)};
        };
        const int r0 = m0 + lr, r1 = m0 + 8 + lr, r2 = m0 + 16 + lr, r3 = m0 + 24 + lr;
        R0.a0 = drop(e00, r0, k0 + ce); R1.a0 = drop(e10, r0, k1 + ce); R0.a1 = drop(e01, r1, k0 + co); R1.a1 = drop(e11, r1, k1 + co);
        R0.a2 = drop(e02, r2, k0 + ce); R1.a2 = drop(e12, r2, k1 + ce); R0.a3 = drop(e03, r3, k0 + co); R1.a3 = drop(e13, r3, k1 + co);
        if (n0 == 0) {                              // column tile 0 publishes ds and dx
            float* o1 = lb.ds;
            float* o2 = lb.dx;
            if (r0 < M) { *reinterpret_cast<f32x4*>(o1 + (size_t)r0 * FIRA_D + k0 + ce) = e00; *reinterpret_cast<f32x4*>(o1 + (size_t)r0 * FIRA_D + k1 + ce) = e10;
                          *reinterpret_cast<f32x4*>(o2 + (size_t)r0 * FIRA_D + k0 + ce) = R0.a0; *reinterpret_cast<f32x4*>(o2 + (size_t)r0 * FIRA_D + k1 + ce) = R1.a0; }
            if (r1 < M) { *reinterpret_cast<f32x4*>(o1 + (size_t)r1 * FIRA_D + k0 + co) = e01; *reinterpret_cast<f32x4*>(o1 + (size_t)r1 * FIRA_D + k1 + co) = e11;
                          *reinterpret_cast<f32x4*>(o2 + (size_t)r1 * FIRA_D + k0 + co) = R0.a1; *reinterpret_cast<f32x4*>(o2 + (size_t)r1 * FIRA_D + k1 + co) = R1.a1; }
            if (r2 < M) { *reinterpret_cast<f32x4*>(o1 + (size_t)r2 * FIRA_D + k0 + ce) = e02; *reinterpret_cast<f32x4*>(o1 + (size_t)r2 * FIRA_D + k1 + ce) = e12;
                          *reinterpret_cast<f32x4*>(o2 + (size_t)r2 * FIRA_D + k0 + ce) = R0.a2; *reinterpret_cast<f32x4*>(o2 + (size_t)r2 * FIRA_D + k1 + ce) = R1.a2; }
            if (r3 < M) { *reinterpret_cast<f32x4*>(o1 + (size_t)r3 * FIRA_D + k0 + co) = e03; *reinterpret_cast<f32x4*>(o1 + (size_t)r3 * FIRA_D + k1 + co) = e13;
                          *reinterpret_cast<f32x4*>(o2 + (size_t)r3 * FIRA_D + k0 + co) = R0.a3; *reinterpret_cast<f32x4*>(o2 + (size_t)r3 * FIRA_D + k1 + co) = R1.a3; }
        }
    }
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
        if ((ci & 1) == 0) {
            tile32_stage(R0, slot, wr_off);
            if (ci + 2 < NCH) tile32_fetch<B_KCONTIG>(R0, pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3, (wave + NW * (ci + 2)) << 5, ldb);
        } else {
            tile32_stage(R1, slot, wr_off);
            if (ci + 2 < NCH) tile32_fetch<B_KCONTIG>(R1, pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3, (wave + NW * (ci + 2)) << 5, ldb);
        }
        acc = tile32_compute<B_KCONTIG>(slot, fa0, fa1, fa2, fa3, fb, acc);
    }
    __builtin_amdgcn_wave_barrier();
    float* red = reinterpret_cast<float*>(slot);                       // this wave's partial tile over its own slot
#pragma unroll
    for (int r = 0; r < 16; ++r) red[r * 64 + lane] = acc[r];
    __syncthreads();
    {
        // thread t combines elements idx = t + 64 NW i (< 1024) of the tile (accumulator register idx / 64 of lane idx % 64): one
        // column, up to ceil(16 / NW) rows
        constexpr int NV = (16 + NW - 1) / NW;
        const int lnn = threadIdx.x & 63;
        float vals[NV];
        int rows[NV];
        const float* p0 = reinterpret_cast<const float*>(smem);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + NW * 64 * i, r = idx >> 6;
            const bool in = (16 % NW == 0) || idx < 1024;        // (12 waves: the second round covers registers 12..15 only)
            const int ix = in ? idx : threadIdx.x;
            if constexpr (NW == 4) {
                vals[i] = (p0[ix] + p0[2048 + ix]) + (p0[4096 + ix] + p0[6144 + ix]);
            } else {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < NW; w += 4)
                    v += (p0[w * 2048 + ix] + p0[(w + 1) * 2048 + ix]) + (p0[(w + 2) * 2048 + ix] + p0[(w + 3) * 2048 + ix]);
                vals[i] = v;
            }
            rows[i] = in ? m0 + (r & 3) + 8 * (r >> 2) + 4 * (lnn >> 5) : M;      // row M: outside, dropped by the descriptor
        }
        epilogue_col<NV>(vals, rows, n0 + (lnn & 31), M, N, C, ldc, bias, flags & FIRA_GEMM_RELU, flags & FIRA_GEMM_ACCUM, false,
                         c_rows, relu_mask, er);
    }
}

template <bool B_KCONTIG, int NCH>
static void tile32_launch(hipStream_t s, dim3 grid, int M, int N, const float* A, int lda, const float* B, int ldb, float* C,
                          int ldc, const float* bias, int flags, const int32_t* c_rows, const float* relu_mask,
                          const int32_t* a_rows, int tiles_n, const EpiRes& er) {
    hipLaunchKernelGGL((gemm_tile32_kernel<B_KCONTIG, NCH, false, false>), grid, dim3(256), 0, s, M, N, A, lda, B, ldb, C, ldc, bias,
                       flags, c_rows, relu_mask, a_rows, tiles_n, er, LnA(), LnB());
}

// FIRA_TILE32_WAVES=0: always four waves (A/B switch).  8 waves (64 KB of slots: two workgroups per CU) up to 512 workgroups,
// 12 / 16 waves (96 / 128 KB: one per CU) up to 256.
static int tile32_waves(int K, int n_wg) {
    static const bool off = [] { const char* e = getenv("FIRA_TILE32_WAVES"); return e && e[0] == '0'; }();
    if (off) return 4;
    const int chunks = K / 32;
    if (chunks == 32 && n_wg <= 256) return 16;
    if (chunks == 24 && n_wg <= 256) return 12;
    if ((chunks == 16 || chunks == 32) && n_wg <= 512) return 8;
    return 4;
}
template <bool B_KCONTIG, int NCH, int NW>
static int tile32_launch_wide_t(hipStream_t s, dim3 grid, int M, int N, const float* A, int lda, const float* B, int ldb, float* C,
                                int ldc, const float* bias, int flags, const int32_t* c_rows, const float* relu_mask,
                                const int32_t* a_rows, int tiles_n, const EpiRes& er) {
    auto kern = gemm_tile32_kernel<B_KCONTIG, NCH, false, false, NW>;
    static const hipError_t attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, NW * 8192);
    if (attr != hipSuccess) return set_err("gemm_tile32: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), NW * 8192, s, M, N, A, lda, B, ldb, C, ldc, bias, flags, c_rows, relu_mask, a_rows,
                       tiles_n, er, LnA(), LnB());
    hipError_t e = hipGetLastError();
    return e != hipSuccess ? set_err("gemm_tile32 (%d waves): %s", NW, hipGetErrorString(e)) : 0;
}
static int tile32_launch_wide(hipStream_t s, int nw, int K, int tB, dim3 grid, int M, int N, const float* A, int lda, const float* B,
                              int ldb, float* C, int ldc, const float* bias, int flags, const int32_t* c_rows,
                              const float* relu_mask, const int32_t* a_rows, int tiles_n, const EpiRes& er) {
#define FIRA_T32W(NCH, NW)                                                                                             \
    return tB ? tile32_launch_wide_t<true, NCH, NW>(s, grid, M, N, A, lda, B, ldb, C, ldc, bias, flags, c_rows, relu_mask, a_rows, tiles_n, er) \
              : tile32_launch_wide_t<false, NCH, NW>(s, grid, M, N, A, lda, B, ldb, C, ldc, bias, flags, c_rows, relu_mask, a_rows, tiles_n, er);
    const int per_wave = K / 32 / nw;              // chunks per wave
    if (nw == 16 && per_wave == 2) { FIRA_T32W(2, 16) }
    if (nw == 12 && per_wave == 2) { FIRA_T32W(2, 12) }
    if (nw == 8 && per_wave == 2) { FIRA_T32W(2, 8) }
    if (nw == 8 && per_wave == 4) { FIRA_T32W(4, 8) }
    return set_err("gemm_tile32: unsupported shape K = %d on %d waves", K, nw);
#undef FIRA_T32W
}

// shapes the coalesced tile kernel takes (the engine asks before it plans a fused LayerNorm prologue / residual epilogue)
bool gemm_tile32_takes(int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb) {
    static const int mode = [] { const char* e = getenv("FIRA_SMALL_GEMM"); return e ? atoi(e) : 2; }();
    if (mode == 1 || M <= 0 || N <= 0) return false;
    if (K % 128 != 0 || lda % 4 != 0 || ((uintptr_t)A % 16) != 0 || ldb % 4 != 0 || ((uintptr_t)B % 16) != 0) return false;
    const int nch = K / 128;
    if (nch != 1 && nch != 2 && nch != 3 && nch != 4 && nch != 6 && nch != 8) return false;
    if (!tB && N % 32 != 0) return false;                               // [K,N] tiles are read as whole 128-byte row segments
    return true;
}

// true if the coalesced tile kernel took the call (FIRA_SMALL_GEMM=1 keeps the round-2 fragment-load kernel: A/B switch)
bool gemm_tile32_try(hipStream_t s, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                     float* C, int ldc, const float* bias, int flags, int* rc, const int32_t* c_rows,
                     const float* relu_mask, const int32_t* a_rows, const EpiRes* er_in) {
    if (!gemm_tile32_takes(tB, M, N, K, A, lda, B, ldb)) return false;
    const EpiRes er = er_in ? *er_in : EpiRes();
    const int nch = K / 128;
    const int tiles_n = cdiv(N, 32);
    const dim3 grid(cdiv(M, 32) * tiles_n);
    {   // long reduction on a small grid: more waves per workgroup, two chunks each (see the kernel's NW)
        const int nw = tile32_waves(K, (int)grid.x);
        if (nw > 4) {
            *rc = tile32_launch_wide(s, nw, K, tB, grid, M, N, A, lda, B, ldb, C, ldc, bias, flags, c_rows, relu_mask, a_rows, tiles_n, er);
            return true;
        }
    }
#define FIRA_T32(NCH)                                                                                                  \
    case NCH:                                                                                                          \
        if (tB) tile32_launch<true, NCH>(s, grid, M, N, A, lda, B, ldb, C, ldc, bias, flags, c_rows, relu_mask, a_rows, tiles_n, er); \
        else tile32_launch<false, NCH>(s, grid, M, N, A, lda, B, ldb, C, ldc, bias, flags, c_rows, relu_mask, a_rows, tiles_n, er); \
        break;
    switch (nch) { FIRA_T32(1) FIRA_T32(2) FIRA_T32(3) FIRA_T32(4) FIRA_T32(6) FIRA_T32(8) }
#undef FIRA_T32
    hipError_t e = hipGetLastError();
    *rc = e != hipSuccess ? set_err("gemm_tile32: %s", hipGetErrorString(e)) : 0;
    return true;
}

// Y = LN(S) W^T + b (+ relu) in one launch, S the pre-norm sums [M,256] of a residual block (see LnA): W is an nn.Linear
// weight [N,256].  Also stores x = LN(S) [M,256] and the row statistics.  false: shape not taken (the caller runs the row
// kernel and a plain product instead).
bool gemm_tile32_ln_try(hipStream_t s, int M, int N, const float* S, int lds, const float* W, const float* bias, float* Y,
                        int ldy, int flags, const float* gamma, const float* beta, float* x_out, float* stats_out, int* rc) {
    if (!gemm_tile32_takes(1, M, N, FIRA_D, S, lds, W, FIRA_D)) return false;
    static const bool off = [] { const char* e = getenv("FIRA_LN_PROLOGUE"); return e && e[0] == '0'; }();     // A/B switch
    if (off || ((uintptr_t)gamma % 16) || ((uintptr_t)beta % 16) || ((uintptr_t)x_out % 16)) return false;
    ProfScope prof(s, PROF_GEMM, 2.0 * M * N * (double)FIRA_D, 4.0 * ((double)M * FIRA_D + (double)N * FIRA_D + (double)M * N));
    const int tiles_n = cdiv(N, 32);
    LnA ln;
    ln.gamma = gamma; ln.beta = beta; ln.x_out = x_out; ln.stats_out = stats_out;
    hipLaunchKernelGGL((gemm_tile32_kernel<true, 2, true, false>), dim3(cdiv(M, 32) * tiles_n), dim3(256), 0, s, M, N, S, lds, W, FIRA_D,
                       Y, ldy, bias, flags & 3, (const int32_t*)nullptr, (const float*)nullptr, (const int32_t*)nullptr, tiles_n,
                       EpiRes(), ln, LnB());
    hipError_t e = hipGetLastError();
    *rc = e != hipSuccess ? set_err("gemm_tile32_ln: %s", hipGetErrorString(e)) : 0;
    return true;
}

// dX[M,N] = LNbwd(dy)[M,256] . W[256,N] (W stored [256, N] row-major: the data gradient through a closing nn.Linear with
// weight [256 out, N in]) in one launch, plus ds / dx_drop / the partial dgamma | dbeta rows (see LnB).  relu_mask (optional,
// [M,N]): output zeroed where mask <= 0 (the FFN's ReLU backward).  false: shape not taken.
int gemm_tile32_lnb_blocks(int M) { return cdiv(M, 32); }
bool gemm_tile32_lnb_try(hipStream_t s, int M, int N, const float* dy, const float* W, int ldw, float* dX, int lddx,
                         const float* relu_mask, const float* sum, const float* stats, const float* gamma, float* ds,
                         float* dx_drop, float* part, float dropout, uint64_t seed, uint32_t site, int* rc, uint32_t idx0) {
    static const bool off = [] { const char* e = getenv("FIRA_LN_BWD_PROLOGUE"); return e && e[0] == '0'; }();     // A/B switch
    if (off || !gemm_tile32_takes(0, M, N, FIRA_D, dy, FIRA_D, W, ldw)) return false;
    if (((uintptr_t)sum % 16) || ((uintptr_t)gamma % 16) || ((uintptr_t)ds % 16) || ((uintptr_t)dx_drop % 16) || ds == dy) return false;
    ProfScope prof(s, PROF_GEMM, 2.0 * M * N * (double)FIRA_D, 4.0 * ((double)M * FIRA_D + (double)N * FIRA_D + (double)M * N));
    const int tiles_n = cdiv(N, 32);
    LnB lb;
    lb.sum = sum; lb.stats = stats; lb.gamma = gamma; lb.ds = ds; lb.dx = dx_drop; lb.part = part;
    lb.p = dropout; lb.inv_keep = dropout > 0.f ? 1.0f / (1.0f - dropout) : 1.0f; lb.seed = seed; lb.site = site; lb.idx0 = idx0;
    hipLaunchKernelGGL((gemm_tile32_kernel<false, 2, false, true>), dim3(cdiv(M, 32) * tiles_n), dim3(256), 0, s, M, N, dy, FIRA_D, W, ldw,
                       dX, lddx, (const float*)nullptr, 0, (const int32_t*)nullptr, relu_mask, (const int32_t*)nullptr, tiles_n,
                       EpiRes(), LnA(), lb);
    hipError_t e = hipGetLastError();
    *rc = e != hipSuccess ? set_err("gemm_tile32_lnb: %s", hipGetErrorString(e)) : 0;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the folded GCN weight's gradient dW21 goes back to the reference's parameters through two 256^3 products per layer,
//     dW2 += dW21 W1^T        dW1 += W2^T dW21         (engine.hip: W21 = W2 W1)
// which were two launches per layer on the weight-gradient stream (7 + 5 us each, launch-bound: 72 us of that stream's serial
// chain per step, and the end of the step waits for it).  Here ALL of them are one launch after the last layer's weight
// gradient: one 32x32 output tile per workgroup, the four waves split K = 256 (gemm_small_kernel's decomposition: every wave
// fetches its fragments straight into registers, 2 x 16 MFMAs, partial tiles combined through LDS), operands addressed by
// (row stride, k stride) so that both products read the matrices as stored.
// mode 0: C[i,j] += sum_k A[i*a_rs + k*a_ks] B[j*b_rs + k*b_ks];  1: C[i,j] = the same sum;
// mode 2: C[i] = sum_k A[i*a_rs + k] B[k]  (a 256-row matrix-vector product: the first 8 workgroups of the entry, 32 rows each)
struct UnfoldGemm { const float *A, *B; float* C; int a_rs, a_ks, b_rs, b_ks, mode; };
struct UnfoldGemmTable { int n = 0; UnfoldGemm e[32]; };
__global__ __launch_bounds__(256) void unfold_gemm_kernel(const UnfoldGemmTable tab) {
    __shared__ float red[4][1024];
    const UnfoldGemm& q = tab.e[blockIdx.y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    if (q.mode == 2) {                                  // (workgroup-uniform) matrix-vector product: a wave owns 8 rows, a lane 4 k
        if (blockIdx.x >= 8) return;
        const f32x4 x = *reinterpret_cast<const f32x4*>(q.B + lane * 4);
        f32x4 a[8];
#pragma unroll
        for (int r = 0; r < 8; ++r)
            a[r] = *reinterpret_cast<const f32x4*>(q.A + (size_t)(blockIdx.x * 32 + wave * 8 + r) * q.a_rs + lane * 4);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float d = wave_sum((a[r].x * x.x + a[r].y * x.y) + (a[r].z * x.z + a[r].w * x.w));
            if (lane == 0) q.C[blockIdx.x * 32 + wave * 8 + r] = d;
        }
        return;
    }
    const int m0 = (blockIdx.x >> 3) * 32, n0 = (blockIdx.x & 7) * 32;
    const float* pa = q.A + (size_t)(m0 + l31) * q.a_rs;
    const float* pb = q.B + (size_t)(n0 + l31) * q.b_rs;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a[2][16], b[2][16];
#pragma unroll
    for (int u = 0; u < 2; ++u) {                       // chunks wave, wave + 4 of the eight 32-wide k chunks: all loads in flight at once
        const int k0 = (wave + 4 * u) * 32 + kh * 16;
        if (q.a_ks == 1) {                              // (workgroup-uniform)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(pa + k0 + 4 * v);
                a[u][4 * v] = x.x; a[u][4 * v + 1] = x.y; a[u][4 * v + 2] = x.z; a[u][4 * v + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) a[u][t] = pa[(size_t)(k0 + t) * q.a_ks];
        }
        if (q.b_ks == 1) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(pb + k0 + 4 * v);
                b[u][4 * v] = x.x; b[u][4 * v + 1] = x.y; b[u][4 * v + 2] = x.z; b[u][4 * v + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) b[u][t] = pb[(size_t)(k0 + t) * q.b_ks];
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t], b[u][t], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r * 64 + lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = threadIdx.x + 256 * i;          // accumulator register r = idx / 64 of lane idx % 64
        const int r = idx >> 6, ln = idx & 63;
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5), col = n0 + (ln & 31);
        float* pc = q.C + (size_t)row * FIRA_D + col;
        const float v = (red[0][idx] + red[1][idx]) + (red[2][idx] + red[3][idx]);
        if (q.mode == 0) *pc += v;
        else *pc = v;
    }
}
int gcn_unfold_products(hipStream_t s, int n_layers, const float* const* dW21, const float* const* W1, const float* const* W2,
                        float* const* dW1, float* const* dW2) {
    if (n_layers <= 0) return 0;
    if (n_layers > 16) return set_err("gcn_unfold_products: %d layers", n_layers);
    UnfoldGemmTable tab;
    for (int l = 0; l < n_layers; ++l) {
        tab.e[tab.n++] = UnfoldGemm{dW21[l], W1[l], dW2[l], FIRA_D, 1, FIRA_D, 1, 0};     // dW2[i,j] += sum_k dW21[i,k] W1[j,k]
        tab.e[tab.n++] = UnfoldGemm{W2[l], dW21[l], dW1[l], 1, FIRA_D, 1, FIRA_D, 0};     // dW1[i,j] += sum_k W2[k,i] dW21[k,j]
    }
    ProfScope prof(s, PROF_GEMM, 2.0 * tab.n * FIRA_D * (double)FIRA_D * FIRA_D, 4.0 * tab.n * 3.0 * FIRA_D * FIRA_D);
    hipLaunchKernelGGL(unfold_gemm_kernel, dim3(64, tab.n), dim3(256), 0, s, tab);
    hipError_t e = hipGetLastError();
    return e != hipSuccess ? set_err("gcn_unfold_products: %s", hipGetErrorString(e)) : 0;
}

// Round 5: the FOLD itself (engine.hip: encoder_forward) through the same kernel.  Per layer
//     W21 = W2 W1        W21t = W1^T W2^T (the k-major copy the fused GCN forward streams)        c21 = W2 b1
// were two product launches + a transpose launch per layer at the head of the auxiliary stream: 14 launch-bound kernels
// (~100 us of that stream) that the caller's stream waited for at the first / second GCN layer -- FIRA_WAIT_PROBE: ~28 us per
// step at batch 32, ~48 at batch 64, ~75 in bf16 mode (one mark behind all of them).  One launch now.  W21t is computed as a
// product of its own (the operands swap roles: same products, same k order, so W21t[j][i] == W21[i][j] bit for bit) rather
// than stored transposed: all stores stay coalesced.
int gcn_fold_weights(hipStream_t s, int n_layers, const float* const* W2, const float* const* W1, const float* const* b1,
                     float* W21, float* W21t, float* c21) {
    if (n_layers <= 0) return 0;
    if (n_layers > 10) return set_err("gcn_fold_weights: %d layers", n_layers);
    UnfoldGemmTable tab;
    for (int l = 0; l < n_layers; ++l) {
        float* w = W21 + (size_t)l * FIRA_D * FIRA_D;
        tab.e[tab.n++] = UnfoldGemm{W2[l], W1[l], w, FIRA_D, 1, 1, FIRA_D, 1};            // W21[i,j] = sum_k W2[i,k] W1[k,j]
        if (W21t)                                                                          // W21t[j,i] = sum_k W1[k,j] W2[i,k]
            tab.e[tab.n++] = UnfoldGemm{W1[l], W2[l], W21t + (size_t)l * FIRA_D * FIRA_D, 1, FIRA_D, FIRA_D, 1, 1};
        tab.e[tab.n++] = UnfoldGemm{W2[l], b1[l], c21 + (size_t)l * FIRA_D, FIRA_D, 1, 0, 0, 2};   // c21[i] = sum_k W2[i,k] b1[k]
    }
    ProfScope prof(s, PROF_GEMM, 2.0 * n_layers * (W21t ? 2 : 1) * FIRA_D * (double)FIRA_D * FIRA_D, 4.0 * tab.n * 3.0 * FIRA_D * FIRA_D);
    hipLaunchKernelGGL(unfold_gemm_kernel, dim3(64, tab.n), dim3(256), 0, s, tab);
    hipError_t e = hipGetLastError();
    return e != hipSuccess ? set_err("gcn_fold_weights: %s", hipGetErrorString(e)) : 0;
}

// true if this kernel took the call
bool gemm_small_try(hipStream_t s, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                    float* C, int ldc, const float* bias, int flags, int* rc, const int32_t* c_rows,
                    const float* relu_mask) {
    *rc = 0;
    // A/B switch: 0 = decode-sized products only, 1 = the round-2 fragment-load kernel, 2 (default) = coalesced tile kernel
    static const int mode = [] { const char* e = getenv("FIRA_SMALL_GEMM"); return e ? atoi(e) : 2; }();
    if (mode == 0 && M > 64) return false;          // 0: only the decode-sized products (M <= 64) stay here
    // beyond ~4 rounds of 32x32 tiles the LDS-tiled kernel's operand reuse wins; M <= 64 (decode) always lands here
    const long tiles = (long)cdiv(M, 32) * cdiv(N, 32);
    // the coalesced tile kernel keeps winning for more rounds of tiles than the fragment-load one did (FIRA_SMALL_TILES: A/B)
    static const long max_tiles = [] { const char* e = getenv("FIRA_SMALL_TILES"); return e ? atol(e) : 1024L; }();
    if (!tA && mode != 1 && M > 64 && tiles > 1024 && tiles <= max_tiles) {
        if (gemm_tile32_try(s, tB, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, rc, c_rows, relu_mask, nullptr, nullptr)) return true;
    }
    if (tA || (M > 64 && tiles > 1024) || K % 32 != 0 || K < 64 || lda % 4 != 0 || ((uintptr_t)A % 16) != 0) return false;
    if (tB && (ldb % 4 != 0 || ((uintptr_t)B % 16) != 0)) return false;
    if (mode != 1 && gemm_tile32_try(s, tB, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, rc, c_rows, relu_mask, nullptr, nullptr))
        return true;
    dim3 grid(cdiv(N, 32), cdiv(M, 32));
    if (tB) hipLaunchKernelGGL(gemm_small_kernel<true>, grid, dim3(256), 0, s, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, c_rows, relu_mask);
    else hipLaunchKernelGGL(gemm_small_kernel<false>, grid, dim3(256), 0, s, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, c_rows, relu_mask);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) *rc = set_err("gemm_small: %s", hipGetErrorString(e));
    return true;
}

}  // namespace fira

extern "C" {
int fira_linear_presum_f32(void* stream, int M, int K, const float* X, int ldx, const float* W, const float* bias,
                           const float* res, float* sum, float dropout, uint64_t seed, uint32_t stream_id) {
    FIRA_REQUIRE(X && W && res && sum && M > 0, "fira_linear_presum_f32: bad argument");
    FIRA_REQUIRE(dropout >= 0.f && dropout < 1.f, "fira_linear_presum_f32: dropout must be in [0,1)");
    fira::EpiRes er;
    er.res = res; er.ldr = FIRA_D; er.p = dropout; er.inv_keep = dropout > 0.f ? 1.0f / (1.0f - dropout) : 1.0f;
    er.seed = seed; er.site = stream_id;
    int rc = 0;
    if (!fira::gemm_tile32_try((hipStream_t)stream, 1, M, FIRA_D, K, X, ldx, W, K, sum, FIRA_D, bias, 0, &rc, nullptr, nullptr,
                               nullptr, &er))
        return fira::set_err("fira_linear_presum_f32: shape %d x 256 x %d / alignment not taken by the tile kernel", M, K);
    return rc;
}
int fira_ln_bwd_linear_f32(void* stream, int M, int N, const float* dy, const float* Wt, float* dX, const float* relu_mask,
                           const float* sum, const float* stats, const float* gamma, float* ds, float* dx_drop, float* dgamma,
                           float* dbeta, float* part, float dropout, uint64_t seed, uint32_t site) {
    FIRA_REQUIRE(dy && Wt && dX && sum && stats && gamma && ds && dx_drop && dgamma && dbeta && part && M > 0 && N > 0,
                 "fira_ln_bwd_linear_f32: bad argument");
    FIRA_REQUIRE(ds != dy, "fira_ln_bwd_linear_f32: ds must not alias dy");
    FIRA_REQUIRE(dropout >= 0.f && dropout < 1.f, "fira_ln_bwd_linear_f32: dropout must be in [0,1)");
    int rc = 0;
    if (!fira::gemm_tile32_lnb_try((hipStream_t)stream, M, N, dy, Wt, N, dX, N, relu_mask, sum, stats, gamma, ds, dx_drop, part,
                                   dropout, seed, site, &rc))
        return fira::set_err("fira_ln_bwd_linear_f32: shape %d x %d / alignment not taken by the tile kernel", M, N);
    if (rc) return rc;
    fira::RedTable tab;
    const int nb = fira::gemm_tile32_lnb_blocks(M);
    tab.e[0] = fira::RedEntry{dgamma, part, FIRA_D, nb, 2 * FIRA_D};
    tab.e[1] = fira::RedEntry{dbeta, part + FIRA_D, FIRA_D, nb, 2 * FIRA_D};
    tab.n = 2;
    return fira::deferred_reduce((hipStream_t)stream, tab);
}
int fira_ln_linear_f32(void* stream, int M, int N, const float* S, int lds, const float* W, const float* bias, float* Y,
                       int ldy, int relu, const float* gamma, const float* beta, float* x_out, float* stats_out) {
    FIRA_REQUIRE(S && W && Y && gamma && beta && x_out && M > 0 && N > 0, "fira_ln_linear_f32: bad argument");
    int rc = 0;
    if (!fira::gemm_tile32_ln_try((hipStream_t)stream, M, N, S, lds, W, bias, Y, ldy, relu ? FIRA_GEMM_RELU : 0, gamma, beta, x_out,
                                  stats_out, &rc))
        return fira::set_err("fira_ln_linear_f32: shape %d x %d / alignment not taken by the tile kernel", M, N);
    return rc;
}
}

