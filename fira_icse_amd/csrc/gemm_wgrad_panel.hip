// Weight gradients as PANEL products on the bf16 matrix cores (round 6):   C[M,N] += A^T B,   A [K, lda], B [K, ldb]
// (dW += dY^T X of every nn.Linear backward: gnn_transformer.py:76,82,142-144,159,172-173,199,203, Model.py:16-19,54 -- the
// reduction runs over the ROWS of both operands, K = node / target / memory rows of the batch; db += column sums of dY).
//
// Why another kernel.  The tiled kernels (gemm_f32.hip / gemm_bf16.hip: 64x64 output tiles, K split into slabs) read every
// operand element once per 64-column slice of the OTHER operand -- a [K,256]^T [K,256] product streams both operands four times
// (from L2 at best), in 256-byte row segments, and adds 4 096 atomics per workgroup to the result: the encoder's 18 weight
// gradients are 220 us of the weight-gradient stream at batch 32 in fp32 (0.48 of the fp32 MFMA peak) and 264 us per launch in
// bf16 at batch 64 (0.05 of the bf16 peak, 1.7 TB/s), the long pole of the step's tail.  Here ONE workgroup owns a 256 x 256
// output tile for a slab of K:
//   * both operands arrive as whole 1 KiB rows (8-byte loads, 512 contiguous bytes per wave and row), each element ONCE per
//     output tile -- for the 256-wide products of the encoder that is once, period;
//   * 8 waves (2 x 4), 128 x 64 outputs each = 8 accumulator tiles of v_mfma_f32_32x32x16_bf16; per 16-row step a wave fetches
//     4 + 2 operand fragments per part from LDS (ds_read_b128, conflict-free: see pn_unit) for 8 MFMAs per part pair;
//   * NP = 1 (bf16 mode): operands rounded to bf16 (RNE) while staged, as every bf16 product of the engine;
//     NP = 3 (fp32 mode): every fp32 operand is split into three bf16 terms  x = hi + mid + lo  (each the RNE rounding of what
//     is left: 24 mantissa bits in all) and the product is formed from the six term pairs whose weight is >= 2^-16,
//         a b ~ hi hi + (hi mid + mid hi) + (hi lo + lo hi + mid mid),
//     each an EXACT bf16 x bf16 product accumulated in fp32 -- the dropped pairs are below 2^-24 of |a b|, i.e. the result is
//     an fp32-accurate product (measured against fp64: tests/test_ops_gpu.py) at 16 / 6 = 2.7x the fp32 MFMA rate;
//   * NO atomics: a problem whose K fits one slab adds its tile to C itself (the workgroup owns the tile: plain read-add-store
//     of whole 128-byte segments); the slabs of a split problem leave their 256 x 256 partial tiles in a caller-provided scratch
//     buffer (plain coalesced stores, Infinity-Cache-resident) and ONE closing launch adds them to C.  (First version: one float
//     atomic per element and slab -- 65 536 per workgroup; measured ~250 G atomics/s for the whole chip, i.e. 60 us for the
//     encoder's group at ten slabs per tile, more than its MFMA time.)
// Per 16-row step a workgroup moves 32 KB from HBM for 48 (NP = 3) or 8 (NP = 1) MFMAs per wave: in fp32 mode the matrix
// pipe and HBM are balanced (16.6 GFLOP x 6 / 2.5 PFLOP/s = 40 us against 228 MB / 6 TB/s = 38 us for the encoder's group at
// batch 32), in bf16 mode the launch is HBM-bound.
#include "engine.h"
#include "epilogue.h"
#include <algorithm>
#include <stdlib.h>

namespace fira {

typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pbf16x2 __attribute__((ext_vector_type(2)));
typedef float pf32x2 __attribute__((ext_vector_type(2)));
typedef float pf32x16 __attribute__((ext_vector_type(16)));

constexpr int PN_T = 256;            // output tile (rows of C = columns of A) x (columns of C = columns of B)
constexpr int PN_KB = 16;            // reduction rows per step = the k depth of one MFMA
constexpr int PN_PART = PN_T * 32;   // bytes of one operand part in LDS: 256 rows x 16 bf16
constexpr int PANEL_MAX = 48;

struct PanelProblem {
    const float *A, *B;
    float *C, *colsum;
    float* part;                     // slabs > 1: [tiles_m * tiles_n][slabs][256 x 256] partial tiles (scratch)
    int M, N, K, lda, ldb, ldc;
    int tiles_m, tiles_n, slabs, k_slab;
};
struct PanelTable {
    int n;
    int wg_start[PANEL_MAX + 1];
    PanelProblem p[PANEL_MAX];
};

// 16-byte unit of (row, k half) inside an operand part: TWO PLANES of 256 units, one per k half (8 bf16 of a row each).
//   fetch  ds_read_b128, lanes of one 16-lane group ({0-3,12-15,20-27} / {4-11,16-19,28-31} of a half-wave) share the k half and
//          read rows that are distinct mod 16: 16 distinct bank quads, conflict-free;
//   store  ds_write_b128 is served 8 consecutive lanes at a time over 32 banks = 8 units: a lane owns the row pair (2c, 2c + 1);
//          lanes 0-3 of every eight store their even row first, lanes 4-7 their odd row -- rows {0,2,4,6,9,11,13,15} then
//          {1,3,5,7,8,10,12,14}: all eight units distinct mod 8 in both instructions (rows 32 bytes apart in one plane of
//          [row][k half] order -- the first version -- made every such store a 4-way conflict, ~1 us per 16-row step).
__device__ __forceinline__ int pn_unit(int row, int kh) { return kh * PN_T + row; }

__device__ __forceinline__ uint32_t pn_pack(float a, float b) {
    pf32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, pbf16x2));      // v_cvt_pk_bf16_f32 (RNE)
}
// what is left of (a, b) after their bf16 roundings `pk` (exact in fp32: the rounding error of an 8-bit significand cut
// has at most 16 significant bits)
__device__ __forceinline__ void pn_rest(float& a, float& b, uint32_t pk) {
    a -= __builtin_bit_cast(float, pk << 16);
    b -= __builtin_bit_cast(float, pk & 0xffff0000u);
}

// Workgroup barrier of the K loop: LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL load
// (s_waitcnt vmcnt(0)) -- here the rows of the next two steps are in flight across every barrier by design, and with it each
// step paid a full memory round trip (4.4 us per step measured against 1.3 us of MFMAs).  The compiler still places the vmcnt
// waits the staging registers need.
__device__ __forceinline__ void pn_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NP>
__global__ __launch_bounds__(512) void wgrad_panel_kernel(const PanelTable tab) {
    extern __shared__ __attribute__((aligned(16))) char pn_lds[];      // [2 buffers][A | B][NP parts][256 rows x 32 B]
    __shared__ float pn_cs[PN_T];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // ---- which problem, tile and K slab (block-uniform)
    int pi = 0;
    while (pi + 1 < tab.n && (int)blockIdx.x >= tab.wg_start[pi + 1]) ++pi;
    const PanelProblem& q = tab.p[pi];
    int w = blockIdx.x - tab.wg_start[pi];
    const int z = w % q.slabs;
    w /= q.slabs;
    const int tn = w % q.tiles_n, tm = w / q.tiles_n;
    const int m0 = tm * PN_T, n0 = tn * PN_T;
    const int k_beg = z * q.k_slab, k_end = min(q.K, k_beg + q.k_slab);
    const int nstep = (k_end - k_beg + PN_KB - 1) / PN_KB;
    // ---- staging role of this thread: operand (waves 0-3: A, 4-7: B), k half (8 rows) and a pair of columns
    const int opnd = t >> 8, idx = t & 255, kb = idx >> 7, cb = idx & 127;
    const float* src = opnd ? q.B : q.A;
    const int ld = opnd ? q.ldb : q.lda;
    const int c0 = (opnd ? n0 : m0) + 2 * cb;                 // first of this thread's two columns
    const int c_lim = opnd ? q.N : q.M;                       // columns past it: products never stored, reads stay inside the rows
    // Rows >= k_end read zero: the descriptor ends with the slab, and the whole byte offset rides in the per-lane operand (the
    // scalar offset of a buffer load is not range-checked) -- so do the prefetches past the last step.  Columns past the
    // operand's width (the last tile of a [V = 24 650]-row result) read the next row's data: finite, and their products are
    // never stored.
    (void)c_lim;
    const rsrc_t rs = buf_rsrc(src, (unsigned)min((long)k_end * ld * 4, 0x7fffffffL));
    // two register stages: the rows of step s + 2 are requested at the head of step s (one 16-row step is 1.3 us of MFMAs in
    // fp32 mode, less than a loaded HBM round trip: with one stage the workgroup -- the only one on its CU -- stood waiting for
    // its operands in every step, 4.3 us per step measured)
    pf32x2 v0[8], v1[8];
    auto fetch = [&](int step, pf32x2 (&v)[8]) __attribute__((always_inline)) {
        const unsigned off0 = ((unsigned)(k_beg + step * PN_KB + kb * 8) * (unsigned)ld + (unsigned)c0) * 4u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint64_t raw = __builtin_bit_cast(uint64_t, __builtin_amdgcn_raw_buffer_load_b64(rs, off0 + (unsigned)j * (unsigned)ld * 4u, 0, 0));
            v[j] = __builtin_bit_cast(pf32x2, raw);
        }
    };
    float cs0 = 0.f, cs1 = 0.f;
    const bool do_cs = q.colsum != nullptr && tn == 0 && opnd == 0;
    char* const my_lds = pn_lds + opnd * (NP * PN_PART);
    const bool odd_first = (cb >> 2) & 1;                                      // (see pn_unit: store order of the row pair)
    const int u_first = pn_unit(2 * cb + (odd_first ? 1 : 0), kb) << 4, u_second = pn_unit(2 * cb + (odd_first ? 0 : 1), kb) << 4;
    auto put = [&](int buf, const pf32x2 (&v)[8]) __attribute__((always_inline)) {
        char* base = my_lds + buf * (2 * NP * PN_PART);
        float a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = v[j].x; b[j] = v[j].y; }
        if (do_cs) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { cs0 += a[j]; cs1 += b[j]; }
        }
#pragma unroll
        for (int part = 0; part < NP; ++part) {
            uint4 wa, wb;
            wa.x = pn_pack(a[0], a[1]); wa.y = pn_pack(a[2], a[3]); wa.z = pn_pack(a[4], a[5]); wa.w = pn_pack(a[6], a[7]);
            wb.x = pn_pack(b[0], b[1]); wb.y = pn_pack(b[2], b[3]); wb.z = pn_pack(b[4], b[5]); wb.w = pn_pack(b[6], b[7]);
            const uint4 w_first = odd_first ? wb : wa, w_second = odd_first ? wa : wb;
            *reinterpret_cast<uint4*>(base + part * PN_PART + u_first) = w_first;
            *reinterpret_cast<uint4*>(base + part * PN_PART + u_second) = w_second;
            if (part + 1 < NP) {
                pn_rest(a[0], a[1], wa.x); pn_rest(a[2], a[3], wa.y); pn_rest(a[4], a[5], wa.z); pn_rest(a[6], a[7], wa.w);
                pn_rest(b[0], b[1], wb.x); pn_rest(b[2], b[3], wb.y); pn_rest(b[4], b[5], wb.z); pn_rest(b[6], b[7], wb.w);
            }
        }
    };
    // ---- MFMA role: wave (wm, wn) owns rows wm*128 .. +127 and columns wn*64 .. +63 of the tile
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, kh = lane >> 5;
    int offA[4], offB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) offA[i] = pn_unit(wm * 128 + i * 32 + l31, kh) << 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) offB[j] = NP * PN_PART + (pn_unit(wn * 64 + j * 32 + l31, kh) << 4);
    pf32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    pbf16x8 a[NP][4], b[NP][2];
    auto frags = [&](int buf) __attribute__((always_inline)) {
        const char* base = pn_lds + buf * (2 * NP * PN_PART);
#pragma unroll
        for (int part = 0; part < NP; ++part) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[part][i] = *reinterpret_cast<const pbf16x8*>(base + part * PN_PART + offA[i]);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[part][j] = *reinterpret_cast<const pbf16x8*>(base + part * PN_PART + offB[j]);
        }
    };
    auto compute = [&]() __attribute__((always_inline)) {
        // term pairs, smallest first: (lo, hi) (hi, lo) (mid, mid) (mid, hi) (hi, mid) (hi, hi)
#pragma unroll
        for (int pr = (NP == 3 ? 0 : 5); pr < 6; ++pr) {
            const int pa = NP == 3 ? ((0x201100 >> (4 * (5 - pr))) & 3) : 0;     // 2 0 1 1 0 0
            const int pb = NP == 3 ? ((0x021010 >> (4 * (5 - pr))) & 3) : 0;     // 0 2 1 0 1 0
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa][i], b[pb][j], acc[i][j], 0, 0, 0);
        }
    };
    // ---- one barrier per step: tile s + 1 is fetched before the MFMAs of tile s and stored into the other buffer behind them
    if (nstep > 0) {
        fetch(0, v0);
        fetch(1, v1);                           // (past the slab: the descriptor returns zeros; such a tile multiplies to nothing)
        put(0, v0);
        fetch(2, v0);
        pn_barrier();
        // per step: the rows of step s + 1 (requested a whole step ago) are split and stored into the OTHER buffer first --
        // the LDS stores then drain under the MFMAs of step s; only then are their registers re-used for step s + 3
        // (fetching the fragments of step s AHEAD of the staging stores, so that the scheduler may issue the split's VALU work
        //  between the MFMAs, was tried: 72 fragment registers live across the split spill 129 registers at 256 per lane)
        for (int s = 0; s < nstep; s += 2) {    // stage / buffer roles are static: steps in pairs
            put(1, v1);
            fetch(s + 2 + 1, v1);
            frags(0);
            compute();
            pn_barrier();
            put(0, v0);                         // (v0 was re-fetched for step s + 2 one step ago)
            fetch(s + 2 + 2, v0);
            if (s + 1 < nstep) { frags(1); compute(); }      // (block-uniform)
            pn_barrier();
        }
    }
    // ---- bias gradient: column sums of this slab's A rows (tile column 0 only)
    if (q.colsum != nullptr && tn == 0) {
        for (int i = t; i < PN_T; i += 512) pn_cs[i] = 0.f;
        __syncthreads();
        if (opnd == 0) { atomicAdd(&pn_cs[2 * cb], cs0); atomicAdd(&pn_cs[2 * cb + 1], cs1); }
        __syncthreads();
        for (int i = t; i < PN_T; i += 512)
            if (m0 + i < q.M) unsafeAtomicAdd(&q.colsum[m0 + i], pn_cs[i]);
    }
    // ---- the tile: accumulator register r of tile (i, j) = row (r & 3) + 8 (r >> 2) + 4 kh, column l31
    if (q.slabs > 1) {
        // a slab of a split problem: the partial tile as it is (rows / columns past M / N hold products of neighbouring data:
        // the closing launch never reads them)
        float* pt = q.part + ((size_t)(tm * q.tiles_n + tn) * q.slabs + z) * (PN_T * PN_T);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    pt[(wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * PN_T + wn * 64 + j * 32 + l31] = acc[i][j][r];
        return;
    }
    // the only slab: this workgroup owns the tile -- C += acc without atomics (all reads of a wave ahead of its stores)
    const rsrc_t rC = buf_rsrc(q.C, 0x7fffffffu);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned off[16];
            float old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                off[r] = (row < q.M && col < q.N) ? ((unsigned)row * (unsigned)q.ldc + (unsigned)col) * 4u : FIRA_OOB;
                old[r] = buf_load_f32(rC, off[r]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, old[r] + acc[i][j][r]), rC, off[r], 0, 0);
        }
    }
}

// closing launch of the split problems: C[tile] += sum over the slabs' partial tiles.  A thread owns one float4 of one tile
// (64 workgroups per tile); the slabs' float4s are requested four at a time.
__global__ __launch_bounds__(256) void wgrad_panel_reduce_kernel(const PanelTable tab) {
    int b = blockIdx.x, pi = 0;
    for (pi = 0; pi < tab.n; ++pi) {                    // block -> (split problem, tile, 1/64 of the tile): block-uniform
        const PanelProblem& q = tab.p[pi];
        if (q.slabs <= 1) continue;
        const int nb = q.tiles_m * q.tiles_n * 64;
        if (b < nb) break;
        b -= nb;
    }
    if (pi >= tab.n) return;
    const PanelProblem& q = tab.p[pi];
    const int tile = b >> 6, idx = (b & 63) * 256 + threadIdx.x;      // float4 index inside the tile: row = idx / 64
    const int tm = tile / q.tiles_n, tn = tile % q.tiles_n;
    const int row = tm * PN_T + (idx >> 6), col = tn * PN_T + (idx & 63) * 4;
    const float4* pt = reinterpret_cast<const float4*>(q.part + (size_t)tile * q.slabs * (PN_T * PN_T)) + idx;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    int z = 0;
    for (; z + 4 <= q.slabs; z += 4) {
        const float4 x0 = pt[(size_t)z * (PN_T * PN_T / 4)], x1 = pt[(size_t)(z + 1) * (PN_T * PN_T / 4)];
        const float4 x2 = pt[(size_t)(z + 2) * (PN_T * PN_T / 4)], x3 = pt[(size_t)(z + 3) * (PN_T * PN_T / 4)];
        a.x += (x0.x + x1.x) + (x2.x + x3.x); a.y += (x0.y + x1.y) + (x2.y + x3.y);
        a.z += (x0.z + x1.z) + (x2.z + x3.z); a.w += (x0.w + x1.w) + (x2.w + x3.w);
    }
    for (; z < q.slabs; ++z) {
        const float4 x = pt[(size_t)z * (PN_T * PN_T / 4)];
        a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
    }
    if (row >= q.M || col >= q.N) return;
    float4* c = reinterpret_cast<float4*>(q.C + (size_t)row * q.ldc + col);
    if ((((uintptr_t)c) & 15) == 0) {
        float4 o = *c;
        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        *c = o;
    } else {
        float* cf = reinterpret_cast<float*>(c);
        cf[0] += a.x; cf[1] += a.y; cf[2] += a.z; cf[3] += a.w;
    }
}

// ---- host side: a queue of problems, one launch -----------------------------------------------------------------------
struct PanelBuilder {
    PanelTable t;
    PanelBuilder() { t.n = 0; t.wg_start[0] = 0; }
};
static PanelBuilder& panel() { static thread_local PanelBuilder b; return b; }

bool gemm_wgrad_panel_on() {
    static const bool off = [] { const char* e = getenv("FIRA_WGRAD_PANEL"); return e && e[0] == '0'; }();   // A/B switch
    return !off;
}
// shapes the kernel takes: at least one full 32-row MFMA tile of output rows, whole 256-column tiles of B, 8-byte aligned
// column pairs, a result the 31-bit byte offsets of the epilogue reach
bool gemm_wgrad_panel_takes(int M, int N, int K, const float* A, int lda, const float* B, int ldb, int ldc) {
    return gemm_wgrad_panel_on() && M >= 32 && N >= PN_T && N % PN_T == 0 && K >= 1 && lda % 2 == 0 && ldb % 2 == 0 &&
           ((uintptr_t)A % 8) == 0 && ((uintptr_t)B % 8) == 0 && (long)K * lda * 4 < 0x7fffffffL && (long)K * ldb * 4 < 0x7fffffffL &&
           epilogue_fits(M, ldc);
}
void gemm_wgrad_panel_reset() { panel().t.n = 0; }
bool gemm_wgrad_panel_full() { return panel().t.n == PANEL_MAX; }
int gemm_wgrad_panel_pending() { return panel().t.n; }

int gemm_wgrad_panel_add(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc, float* colsum) {
    PanelTable& t = panel().t;
    FIRA_REQUIRE(t.n < PANEL_MAX, "gemm_wgrad_panel_add: queue full (flush first)");
    PanelProblem& q = t.p[t.n++];
    q.A = A; q.B = B; q.C = C; q.colsum = colsum;
    q.M = M; q.N = N; q.K = K; q.lda = lda; q.ldb = ldb; q.ldc = ldc;
    q.tiles_m = cdiv(M, PN_T); q.tiles_n = N / PN_T;
    q.slabs = 1; q.k_slab = K; q.part = nullptr;
    return 0;
}
// scratch for the partial tiles of split problems (256 KB per workgroup of a split problem); without one nothing is split
static thread_local float* g_panel_scratch = nullptr;
static thread_local size_t g_panel_scratch_floats = 0;
void gemm_wgrad_panel_scratch(float* buf, size_t n_floats) { g_panel_scratch = buf; g_panel_scratch_floats = n_floats; }
// np: 1 = operands rounded to bf16 once (bf16 mode), 3 = three-term split (fp32-accurate)
int gemm_wgrad_panel_flush(hipStream_t s, int np) {
    PanelTable& t = panel().t;
    if (t.n == 0) return 0;
    // Slab length: the launch's reduction rows spread over about one workgroup per CU (a slab is a multiple of 16 rows and at
    // least 256 of them: every slab of a split problem costs a 256 KB partial tile written and read once more); a launch
    // with many tiles (the vocabulary projection: 97) is not split at all.
    static const int target_wgs = [] { const char* e = getenv("FIRA_WGRAD_PANEL_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
    double work = 0, flop = 0, bytes = 0;
    for (int i = 0; i < t.n; ++i) {
        const PanelProblem& q = t.p[i];
        work += (double)q.tiles_m * q.tiles_n * q.K;
        flop += 2.0 * q.M * q.N * (double)q.K;
        bytes += 4.0 * ((double)q.M * q.K + (double)q.N * q.K + (double)q.M * q.N);
    }
    int tiles_total = 0;
    for (int i = 0; i < t.n; ++i) tiles_total += t.p[i].tiles_m * t.p[i].tiles_n;
    int per = std::max(256, (int)((work / target_wgs + PN_KB - 1) / PN_KB) * PN_KB);
    if (3 * tiles_total >= target_wgs) per = 1 << 30;       // enough tiles to occupy a good part of the chip: no partial tiles
    size_t need = 0;
    for (int pass = 0; pass < 8; ++pass) {              // longer slabs until the partial tiles fit the scratch buffer
        need = 0;
        for (int i = 0; i < t.n; ++i) {
            PanelProblem& q = t.p[i];
            q.slabs = g_panel_scratch ? std::max(1, cdiv(q.K, per)) : 1;
            q.k_slab = cdiv(cdiv(q.K, q.slabs), PN_KB) * PN_KB;
            q.slabs = cdiv(q.K, q.k_slab);
            if (q.slabs > 1) need += (size_t)q.tiles_m * q.tiles_n * q.slabs * (PN_T * PN_T);
        }
        if (need <= g_panel_scratch_floats) break;
        per *= 2;
    }
    if (need > g_panel_scratch_floats) {                 // (cannot happen with per doubling up to 256x: unsplit needs nothing)
        for (int i = 0; i < t.n; ++i) { t.p[i].slabs = 1; t.p[i].k_slab = cdiv(t.p[i].K, PN_KB) * PN_KB; }
        need = 0;
    }
    size_t off = 0;
    int bands = 0;
    for (int i = 0; i < t.n; ++i) {
        PanelProblem& q = t.p[i];
        q.part = nullptr;
        if (q.slabs > 1) {
            q.part = g_panel_scratch + off;
            off += (size_t)q.tiles_m * q.tiles_n * q.slabs * (PN_T * PN_T);
            bands += q.tiles_m * q.tiles_n * 64;
        }
        t.wg_start[i + 1] = t.wg_start[i] + q.tiles_m * q.tiles_n * q.slabs;
    }
    ProfScope prof(s, PROF_GEMM, flop, bytes);
    const size_t lds = (size_t)2 * 2 * np * PN_PART;
    static bool attr_set[2] = {false, false};
    if (np == 3) {
        if (!attr_set[1]) { (void)hipFuncSetAttribute((const void*)wgrad_panel_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set[1] = true; }
        hipLaunchKernelGGL(wgrad_panel_kernel<3>, dim3(t.wg_start[t.n]), dim3(512), lds, s, t);
    } else {
        if (!attr_set[0]) { (void)hipFuncSetAttribute((const void*)wgrad_panel_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set[0] = true; }
        hipLaunchKernelGGL(wgrad_panel_kernel<1>, dim3(t.wg_start[t.n]), dim3(512), lds, s, t);
    }
    if (bands > 0) hipLaunchKernelGGL(wgrad_panel_reduce_kernel, dim3(bands), dim3(256), 0, s, t);
    t.n = 0;
    FIRA_CHECK_LAUNCH("wgrad_panel");
    return 0;
}
// one product, launched at once
int gemm_wgrad_panel(hipStream_t s, int np, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                     int ldc, float* colsum) {
    FIRA_REQUIRE(panel().t.n == 0, "gemm_wgrad_panel: problems are queued (flush first)");
    if (int rc = gemm_wgrad_panel_add(M, N, K, A, lda, B, ldb, C, ldc, colsum)) return rc;
    return gemm_wgrad_panel_flush(s, np);
}

}  // namespace fira

extern "C" int fira_gemm_wgrad_panel(void* stream, int dtype, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                                     float* C, int ldc, float* colsum, float* scratch, size_t scratch_floats) {
    FIRA_REQUIRE(A && B && C && (dtype == 0 || dtype == 1), "fira_gemm_wgrad_panel: bad argument");
    FIRA_REQUIRE(fira::gemm_wgrad_panel_takes(M, N, K, A, lda, B, ldb, ldc),
                 "fira_gemm_wgrad_panel: shape %d x %d x %d / alignment not taken (M >= 32, N %% 256 == 0, even pitches)", M, N, K);
    fira::gemm_wgrad_panel_scratch(scratch, scratch ? scratch_floats : 0);
    const int rc = fira::gemm_wgrad_panel((hipStream_t)stream, dtype == 1 ? 1 : 3, M, N, K, A, lda, B, ldb, C, ldc, colsum);
    fira::gemm_wgrad_panel_scratch(nullptr, 0);
    return rc;
}
