// The Combination block of one encoder layer as ONE launch per direction (reference gnn_transformer.py:176-205 with
// combination_layer.py:7-17; code rows only):
//
//   forward    q | k = Xc [Wq | Wk]^T + b          (two [n,256]x[256,256] products)
//              c     = dropout( g0 k + g1 v[mark] ),  (g0, g1) = softmax(q k / sqrt 32, q v / sqrt 32)   per element
//              s     = dropout( c Wo^T + bo ) + Xc ;  X[code_rows] = LayerNorm(s)
//
// replacing the four launches  [n,512,256] product -> gate row kernel -> [n,256,256] product -> add + LayerNorm row kernel
// (15 + 4 + 8 + 4.3 us of kernel time at batch 32 and three boundaries of the forward chain, where nothing else runs beside
// the caller's stream) and the round trips of q|k and c through HBM between them (they are still WRITTEN once: the backward
// pass reads q|k for the gate's derivative and c as the output projection's weight-gradient operand).
//
// Geometry: that of gcn_fused.hip -- ONE workgroup (16 waves) per CU over a contiguous range of 16-row tiles, up to CF_TMAX
// tiles per pass in a swizzled fp32 LDS panel, every wave owning 16 output columns of all of them, v_mfma_f32_16x16x4_f32 with
// B fragments straight from the L2-resident K-MAJOR weights (transposed once per step on the auxiliary stream) through buffer
// loads requested a 64-wide k chunk ahead.  The gate runs in the ACCUMULATOR layout (lane (l15, kq), register r = row 4 kq + r,
// column 16 wave + l15 of a tile: q and k of one element sit in the same lane), so q|k never go back through LDS; c does (it is
// the next product's A operand), and the closing rows leave through the panel as whole 1 KiB rows.
// At batch 32 (3 700 code rows = 230 tiles) every CU owns one tile: 3 x 16x256x256 products = 10 us of its fp32 MFMA pipe.
#include "engine.h"
#include "mfma_frag.h"
#include "epilogue.h"
#include "x3.h"

namespace fira {

constexpr int CF_TILE = 16;               // rows per MFMA tile
constexpr int CF_TMAX = 2;                // tiles per pass (32 panel rows)
constexpr int CF_ROWS = CF_TILE * CF_TMAX;
constexpr int CF_WAVES = 16;
constexpr int CF_RPW = CF_ROWS / CF_WAVES;   // rows per wave in the load / row phases
constexpr int CF_GRID = 256;              // one workgroup per CU
constexpr size_t CF_LDS = 84 * 1024;      // > half of a CU's 160 KB: never two of these workgroups on one CU (see gcn_fused.hip)

typedef float cf_acc __attribute__((ext_vector_type(4)));

struct CombFusedArgs {
    int n_rows;
    const float* Xc;                      // [n, 256] code rows: the block's input and residual
    const float *WqT, *WkT, *WoT;         // k-major [256 k][256 n]
    const float *bqk, *bo;                // [512], [256]
    const float* vtab;                    // value rows of the 4 marks: vtab[m * ldv + col]
    int ldv;
    const int32_t* mark;                  // [n]
    float *qk, *c;                        // saved for the backward pass: [n, 512], [n, 256]
    const float *gamma, *beta;
    float *sum, *y, *stats;               // pre-norm rows [n,256] (compact), output rows (at y_rows[r]), (mean, rstd) [n,2]
    const int32_t* y_rows;
    float p, inv_keep;
    uint64_t seed;
    uint32_t site_gate, site_out;
};

// float offset of 16-byte column `quad` (0..63) of panel row `row` (gcn_fused.hip's swizzle)
__device__ __forceinline__ int cf_off(int row, int quad) { return row * FIRA_D + ((quad ^ (row & 11)) << 2); }

// The first 64-wide k chunk of a weight's B fragments (k = 16 kq + s of this wave's 16 columns): requested EARLY -- at the top
// of the kernel for Wq / Wk (under the load of the code rows), before the gate for Wo -- so that no product starts with a bare
// L2 round trip (one tile per CU at batch 32: a product is 0.9 us of MFMA issue per wave, a round trip costs more)
__device__ __forceinline__ void cf_first_chunk(const float* __restrict__ W, unsigned wlane, float (&b0)[16]) {
    const rsrc_t rW = buf_rsrc(W, FIRA_D * FIRA_D * 4u);
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2)
        b0[s2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rW, wlane, s2 * FIRA_D * 4, 0));
}

// acc[tt] += panel tile tt [16, 256] x Wk[256 k][16 columns of this wave]   (tt < nt).  On entry bx = cf_first_chunk(W); on
// exit bx = the first chunk of Wnext (if given), requested under the MFMAs of W's last chunk into the fragment buffer that
// chunk leaves free -- a chain of products never opens with a bare round trip and holds no extra registers for it.
// BF (bf16 mode of the engine): both fragments rounded to bf16 (RNE, as the staged operands of gemm_bf16*.hip), two
// v_mfma_f32_16x16x32_bf16 per 64-wide chunk and tile, fp32 accumulation
template <bool BF, int TM>
__device__ __forceinline__ void cf_product(const float* __restrict__ sm_u, const int (&a_off)[4], const float* __restrict__ W,
                                           unsigned wlane, int nt, float (&bx)[16], cf_acc (&acc)[TM],
                                           const float* __restrict__ Wnext = nullptr) {
    constexpr int NC = FIRA_D / 64;
    static_assert(NC % 2 == 0, "the buffer the last chunk leaves free is b[0]");
    const rsrc_t rW = buf_rsrc(W, FIRA_D * FIRA_D * 4u);
    const rsrc_t rN = buf_rsrc(Wnext ? Wnext : W, FIRA_D * FIRA_D * 4u);
    float b[2][16];
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) b[0][s2] = bx[s2];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c + 1 < NC) {
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2)
                b[(c + 1) & 1][s2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                   rW, wlane, ((c + 1) * 64 + s2) * FIRA_D * 4, 0));
        } else if (Wnext) {                              // (block-uniform)
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2)
                b[0][s2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rN, wlane, s2 * FIRA_D * 4, 0));
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int tt = 0; tt < TM; ++tt) {
            if (tt < nt) {                               // block-uniform
                float af[16];
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const f32x4v q = *reinterpret_cast<const f32x4v*>(&sm_u[a_off[ii] + tt * (CF_TILE * FIRA_D) + c * 64]);
                    af[4 * ii] = q.x; af[4 * ii + 1] = q.y; af[4 * ii + 2] = q.z; af[4 * ii + 3] = q.w;
                }
                if constexpr (BF) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        uint4 ua, ub;
                        ua.x = apk2(af[8 * j], af[8 * j + 1]); ua.y = apk2(af[8 * j + 2], af[8 * j + 3]);
                        ua.z = apk2(af[8 * j + 4], af[8 * j + 5]); ua.w = apk2(af[8 * j + 6], af[8 * j + 7]);
                        ub.x = apk2(b[c & 1][8 * j], b[c & 1][8 * j + 1]); ub.y = apk2(b[c & 1][8 * j + 2], b[c & 1][8 * j + 3]);
                        ub.z = apk2(b[c & 1][8 * j + 4], b[c & 1][8 * j + 5]); ub.w = apk2(b[c & 1][8 * j + 6], b[c & 1][8 * j + 7]);
                        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(abf16x8, ua),
                                                                          __builtin_bit_cast(abf16x8, ub), acc[tt], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int s2 = 0; s2 < 16; ++s2)
                        acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s2], b[c & 1][s2], acc[tt], 0, 0, 0);
                }
            }
        }
        asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) bx[s2] = b[0][s2];
}

// X3 (round 6, fp32 mode; x3.h): the same chain on the bf16 matrix cores at fp32 accuracy.  The panel holds the A operand as
// three bf16 planes, the weights arrive as pre-split planes in fragment order (gcn_split_planes of B[n][k] with out = U B^T:
// Wq / Wk / Wo as stored for the forward block, their transposes for the backward one); a k step of 32 is six
// v_mfma_f32_16x16x32_bf16 per tile.  bx = the three planes' units of k step 0, handed from product to product as above.
// NP = 3: fp32 mode (three terms); NP = 1: the engine's bf16 mode on the same machinery (one plane: operands rounded once)
template <int NP>
__device__ __forceinline__ void cx_first(const uint16_t* __restrict__ Wx, unsigned xlane, uint4 (&b0)[3]) {
    const rsrc_t rW = buf_rsrc(Wx, 3u * (unsigned)GX_WPLANE);
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
        b0[pl] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rW, xlane, pl * (int)GX_WPLANE, 0));
}
template <int TM, int NP>
__device__ __forceinline__ void cx_product(const char* __restrict__ sm_p, size_t plane_bytes, int a_q, const uint16_t* __restrict__ Wx,
                                           unsigned xlane, int nt, uint4 (&bx)[3], cf_acc (&acc)[TM],
                                           const uint16_t* __restrict__ Wnext = nullptr) {
    const rsrc_t rW = buf_rsrc(Wx, 3u * (unsigned)GX_WPLANE);
    const rsrc_t rN = buf_rsrc(Wnext ? Wnext : Wx, 3u * (unsigned)GX_WPLANE);
    uint4 b[2][3];
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) b[0][pl] = bx[pl];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        if (ks + 1 < 8) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                b[(ks + 1) & 1][pl] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                    rW, xlane, pl * (int)GX_WPLANE + (ks + 1) * 1024, 0));
        } else if (Wnext) {                              // (block-uniform; k step 7 reads b[1]: b[0] is free)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                b[0][pl] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rN, xlane, pl * (int)GX_WPLANE, 0));
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int tt = 0; tt < TM; ++tt) {
            if (tt < nt)                                 // block-uniform
                gx_terms<NP>(acc[tt], sm_p + ((a_q ^ (ks << 6)) + tt * (CF_TILE * 512)), plane_bytes, b[ks & 1]);
        }
        asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) bx[pl] = b[0][pl];
}
constexpr size_t CX_PLANE = (size_t)CF_ROWS * 512;        // forward: bytes of one bf16 plane of the 32-row panel

template <bool BF, int NP = 0>
__global__ __launch_bounds__(CF_WAVES * 64) void comb_fused_fwd_kernel(const CombFusedArgs a) {
    constexpr bool X3 = NP > 0;
    constexpr int NPX = X3 ? NP : 3;
    extern __shared__ __attribute__((aligned(16))) float cf_lds[];
    float* const sm_u = cf_lds;                                            // [32][256] swizzled (X3: the closing rows only)
    char* const sm_p = reinterpret_cast<char*>(cf_lds);                    // X3: three bf16 planes [32][256] (48 KB)
    int* const sm_mark = reinterpret_cast<int*>(cf_lds + (X3 ? 3 * CX_PLANE / 4 : (size_t)CF_ROWS * FIRA_D));   // [32]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int wg = (blockIdx.x & 7) * (CF_GRID / 8) + (blockIdx.x >> 3);      // XCD b % 8 owns a contiguous eighth of the tiles
    const int n_tiles = (a.n_rows + CF_TILE - 1) / CF_TILE;
    const int tq = n_tiles / CF_GRID, tr = n_tiles % CF_GRID;
    const int t_beg = wg * tq + min(wg, tr), t_cnt = tq + (wg < tr ? 1 : 0);
    if (t_cnt == 0) return;
    int a_off[4], d_off[4], r_off[CF_RPW];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) a_off[ii] = cf_off(l15, kq * 4 + ii);
#pragma unroll
    for (int r = 0; r < 4; ++r) d_off[r] = cf_off(4 * kq + r, (wave * 16 + l15) >> 2) + (l15 & 3);
#pragma unroll
    for (int i = 0; i < CF_RPW; ++i) r_off[i] = cf_off(wave * CF_RPW + i, lane);
    // this lane's output column in every product, and what the gate needs of it
    const int col = wave * 16 + l15;
    const unsigned wlane = (unsigned)((kq * 16) * FIRA_D + col) * 4u;
    const float bq = a.bqk[col], bk = a.bqk[FIRA_D + col];
    const float vt0 = a.vtab[col], vt1 = a.vtab[(size_t)a.ldv + col], vt2 = a.vtab[(size_t)2 * a.ldv + col],
                vt3 = a.vtab[(size_t)3 * a.ldv + col];
    const rsrc_t rQK = buf_rsrc(a.qk, (unsigned)((size_t)a.n_rows * 2 * FIRA_D * 4));
    const rsrc_t rC = buf_rsrc(a.c, (unsigned)((size_t)a.n_rows * FIRA_D * 4));
    const unsigned xlane = gx_wlane(wave, lane);
    const int a_q = gx_frag_base(l15, kq);
    const uint16_t* const Wx = reinterpret_cast<const uint16_t*>(a.WqT);      // X3: planes of Wq | Wk | Wo (three matrices)
    constexpr size_t WX = 3 * (size_t)FIRA_D * FIRA_D;                        // bf16 elements of one matrix's planes

    for (int pass = 0; pass < t_cnt; pass += CF_TMAX) {
        const int nt = min(CF_TMAX, t_cnt - pass);
        const int row0 = (t_beg + pass) * CF_TILE;
        const int row_end = min(a.n_rows, row0 + nt * CF_TILE);
        // ------------------------------------------------------------ 1. the tile's code rows -> panel
        float bx[16];
        uint4 bx3[3];
        if constexpr (X3) cx_first<NPX>(Wx, xlane, bx3);
        else cf_first_chunk(a.WqT, wlane, bx);
        asm volatile("" ::: "memory");
        {
            f32x4v x[CF_RPW];
#pragma unroll
            for (int i = 0; i < CF_RPW; ++i) {
                const int row = row0 + wave * CF_RPW + i;
                x[i] = *reinterpret_cast<const f32x4v*>(a.Xc + (size_t)min(row, a.n_rows - 1) * FIRA_D + lane * 4);
            }
            if (t < CF_ROWS) sm_mark[t] = a.mark[min(row0 + t, a.n_rows - 1)];
#pragma unroll
            for (int i = 0; i < CF_RPW; ++i) {
                const int row = row0 + wave * CF_RPW + i;
                const f32x4v xv = row < row_end ? x[i] : f32x4v{0.f, 0.f, 0.f, 0.f};
                if constexpr (X3) gx_store_row4<NPX>(sm_p, CX_PLANE, wave * CF_RPW + i, lane, xv);
                else *reinterpret_cast<f32x4v*>(&sm_u[r_off[i]]) = xv;
            }
        }
        lds_barrier();   
        // ------------------------------------------------------------ 2. q and k: two products over the same panel
        cf_acc aq[CF_TMAX], ak[CF_TMAX];
#pragma unroll
        for (int tt = 0; tt < CF_TMAX; ++tt) { aq[tt] = cf_acc{0.f, 0.f, 0.f, 0.f}; ak[tt] = cf_acc{0.f, 0.f, 0.f, 0.f}; }
        if constexpr (X3) {
            cx_product<CF_TMAX, NPX>(sm_p, CX_PLANE, a_q, Wx, xlane, nt, bx3, aq, Wx + WX);
            cx_product<CF_TMAX, NPX>(sm_p, CX_PLANE, a_q, Wx + WX, xlane, nt, bx3, ak, Wx + 2 * WX);
        } else {
        cf_product<BF, CF_TMAX>(sm_u, a_off, a.WqT, wlane, nt, bx, aq, a.WkT);
        cf_product<BF, CF_TMAX>(sm_u, a_off, a.WkT, wlane, nt, bx, ak, a.WoT);       // (Wo's first chunk: in flight under the gate)
        }
        lds_barrier();                                    // every wave has read its last A fragment: the panel can be overwritten
        // ------------------------------------------------------------ 3. the gate, element by element in the accumulator layout
#pragma unroll
        for (int tt = 0; tt < CF_TMAX; ++tt) {
            if (tt < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = tt * CF_TILE + 4 * kq + r, row = row0 + rl;
                    const bool live = row < row_end;
                    const float q = aq[tt][r] + bq, k = ak[tt][r] + bk;
                    const int m = sm_mark[rl];
                    const float v = m == 0 ? vt0 : m == 1 ? vt1 : m == 2 ? vt2 : vt3;
                    float g0, g1;
                    gate_elem(q, k, v, g0, g1);
                    float c = g0 * k + g1 * v;
                    if (a.p > 0.f) c *= dropout_scale(a.seed, a.site_gate, (uint32_t)row * FIRA_D + col, a.p, a.inv_keep);
                    const unsigned o = live ? ((unsigned)row * FIRA_D + col) * 4u : FIRA_OOB;       // rows past the end: dropped
                    const unsigned o2 = live ? ((unsigned)row * 2 * FIRA_D + col) * 4u : FIRA_OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, q), rQK, o2, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, k), rQK, o2, FIRA_D * 4, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c), rC, o, 0, 0);
                    if constexpr (X3) gx_store_elem<NPX>(sm_p, CX_PLANE, rl, col, live ? c : 0.f);
                    else sm_u[d_off[r] + tt * (CF_TILE * FIRA_D)] = live ? c : 0.f;
                }
            }
        }
        lds_barrier();   
        // ------------------------------------------------------------ 4. the output projection
        cf_acc ao[CF_TMAX];
#pragma unroll
        for (int tt = 0; tt < CF_TMAX; ++tt) ao[tt] = cf_acc{0.f, 0.f, 0.f, 0.f};
        if constexpr (X3) cx_product<CF_TMAX, NPX>(sm_p, CX_PLANE, a_q, Wx + 2 * WX, xlane, nt, bx3, ao);
        else cf_product<BF, CF_TMAX>(sm_u, a_off, a.WoT, wlane, nt, bx, ao);
        // what the closing rows need from memory, requested before the accumulators go back through the panel
        const int rbase = row0 + wave * CF_RPW;
        const f32x4v bias4 = *reinterpret_cast<const f32x4v*>(a.bo + lane * 4);
        const f32x4v g4 = *reinterpret_cast<const f32x4v*>(a.gamma + lane * 4);
        const f32x4v be4 = *reinterpret_cast<const f32x4v*>(a.beta + lane * 4);
        f32x4v res[CF_RPW];
        int yr[CF_RPW];
#pragma unroll
        for (int i = 0; i < CF_RPW; ++i) {
            const int row = min(rbase + i, a.n_rows - 1);
            res[i] = *reinterpret_cast<const f32x4v*>(a.Xc + (size_t)row * FIRA_D + lane * 4);
            yr[i] = a.y_rows ? a.y_rows[row] : row;
        }
        asm volatile("" ::: "memory");
        lds_barrier();   
#pragma unroll
        for (int tt = 0; tt < CF_TMAX; ++tt) {
            if (tt < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sm_u[d_off[r] + tt * (CF_TILE * FIRA_D)] = ao[tt][r];
            }
        }
        lds_barrier();   
        // ------------------------------------------------------------ 5. whole rows: bias, dropout, residual, LayerNorm
        if (rbase < row_end) {
#pragma unroll
            for (int i = 0; i < CF_RPW; ++i) {
                const int row = rbase + i;
                if (row >= row_end) continue;                        // wave-uniform
                f32x4v x = *reinterpret_cast<const f32x4v*>(&sm_u[r_off[i]]);
                x = x + bias4;
                if (a.p > 0.f) {
                    const uint32_t e0 = (uint32_t)row * FIRA_D + lane * 4;
                    x.x *= dropout_scale(a.seed, a.site_out, e0 + 0, a.p, a.inv_keep);
                    x.y *= dropout_scale(a.seed, a.site_out, e0 + 1, a.p, a.inv_keep);
                    x.z *= dropout_scale(a.seed, a.site_out, e0 + 2, a.p, a.inv_keep);
                    x.w *= dropout_scale(a.seed, a.site_out, e0 + 3, a.p, a.inv_keep);
                }
                x = x + res[i];
                const float mean = wave_sum(x.x + x.y + x.z + x.w) * (1.0f / FIRA_D);
                const f32x4v d = x - mean;
                const float var = wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * (1.0f / FIRA_D);
                const float rstd = 1.0f / sqrtf(var + 1e-5f);
                __builtin_nontemporal_store(x, reinterpret_cast<f32x4v*>(a.sum + (size_t)row * FIRA_D + lane * 4));   // (backward only)
                const f32x4v out = {d.x * rstd * g4.x + be4.x, d.y * rstd * g4.y + be4.y, d.z * rstd * g4.z + be4.z,
                                    d.w * rstd * g4.w + be4.w};
                *reinterpret_cast<f32x4v*>(a.y + (size_t)yr[i] * FIRA_D + lane * 4) = out;
                if (lane == 0) {
                    a.stats[2 * row] = mean;
                    a.stats[2 * row + 1] = rstd;
                }
            }
        }
        if (pass + CF_TMAX < t_cnt) lds_barrier();         // the next pass overwrites the panel
    }
}

// ---- (round 6) out = X W^T + bias for N = nb x 256 output columns on the same machinery --------------------------------------
// The cross-attention K|V projection of all layers (gnn_transformer.py:139-141 on the encoder's memory rows: [n_mem, 256] x
// [nl * 512, 256]^T, 6.8 GFLOP at batch 32) ran on the auxiliary stream as one fp32 MFMA launch per layer (19 us each at 0.34 of
// the fp32 peak), beside the decoder's forward chain.  Here a workgroup stages its 16-row tiles of X ONCE as bf16
// planes in LDS and multiplies them with every 256-column block of the weight in turn (cx_product: pre-split weight planes in
// fragment order streamed from L2, six term products per k step; NP = 1: the bf16 mode's one plane).  Bound by the weight stream:
// every workgroup reads nb x 384 KB of planes from L2 (one 16-row tile per CU at batch 32).
struct LinearX3Args {
    int n_rows;
    const float* X;          // [n, ldx] fp32 rows
    int ldx;
    const uint16_t* Wx;      // nb matrices' planes (3 * 65536 bf16 each, gcn_split_planes of W[256 j .. 256 j + 255][256])
    int nb;
    const float* bias;       // [nb * 256] or nullptr
    float* out;              // [n, ldo]
    int ldo;
};
template <int NP>
__global__ __launch_bounds__(CF_WAVES * 64) void linear_x3_kernel(const LinearX3Args a) {
    extern __shared__ __attribute__((aligned(16))) float cf_lds[];
    char* const sm_p = reinterpret_cast<char*>(cf_lds);                    // three bf16 planes [32][256] (48 KB)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int grid = gridDim.x;                          // a multiple of 8 (the launch: >= two tiles per workgroup where there are enough)
    const int wg = (blockIdx.x & 7) * (grid / 8) + (blockIdx.x >> 3);
    const int n_tiles = (a.n_rows + CF_TILE - 1) / CF_TILE;
    const int tq = n_tiles / grid, tr = n_tiles % grid;
    const int t_beg = wg * tq + min(wg, tr), t_cnt = tq + (wg < tr ? 1 : 0);
    if (t_cnt == 0) return;
    const int col = wave * 16 + l15;
    const unsigned xlane = gx_wlane(wave, lane);
    const int a_q = gx_frag_base(l15, kq);
    constexpr size_t WX = 3 * (size_t)FIRA_D * FIRA_D;
    const rsrc_t rO = buf_rsrc(a.out, (unsigned)((size_t)a.n_rows * a.ldo * 4));
    for (int pass = 0; pass < t_cnt; pass += CF_TMAX) {
        const int nt = min(CF_TMAX, t_cnt - pass);
        const int row0 = (t_beg + pass) * CF_TILE;
        const int row_end = min(a.n_rows, row0 + nt * CF_TILE);
        uint4 bx3[3];
        cx_first<NP>(a.Wx, xlane, bx3);                  // (block 0's first k step: in flight under the row loads)
        asm volatile("" ::: "memory");
        {
            f32x4v x[CF_RPW];
#pragma unroll
            for (int i = 0; i < CF_RPW; ++i) {
                const int row = row0 + wave * CF_RPW + i;
                x[i] = *reinterpret_cast<const f32x4v*>(a.X + (size_t)min(row, a.n_rows - 1) * a.ldx + lane * 4);
            }
#pragma unroll
            for (int i = 0; i < CF_RPW; ++i) {
                const int row = row0 + wave * CF_RPW + i;
                gx_store_row4<NP>(sm_p, CX_PLANE, wave * CF_RPW + i, lane, row < row_end ? x[i] : f32x4v{0.f, 0.f, 0.f, 0.f});
            }
        }
        lds_barrier();   
        for (int blk = 0; blk < a.nb; ++blk) {
            cf_acc acc[CF_TMAX];
#pragma unroll
            for (int tt = 0; tt < CF_TMAX; ++tt) acc[tt] = cf_acc{0.f, 0.f, 0.f, 0.f};
            const float bias = a.bias ? a.bias[blk * FIRA_D + col] : 0.f;
            cx_product<CF_TMAX, NP>(sm_p, CX_PLANE, a_q, a.Wx + blk * WX, xlane, nt, bx3, acc,
                                    blk + 1 < a.nb ? a.Wx + (blk + 1) * WX : nullptr);
#pragma unroll
            for (int tt = 0; tt < CF_TMAX; ++tt) {
                if (tt < nt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = row0 + tt * CF_TILE + 4 * kq + r;
                        const unsigned o = row < row_end ? ((unsigned)row * (unsigned)a.ldo + (unsigned)(blk * FIRA_D + col)) * 4u : FIRA_OOB;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[tt][r] + bias), rO, o, 0, 0);
                    }
                }
            }
        }
        if (pass + CF_TMAX < t_cnt) lds_barrier();         // the next pass overwrites the panel
    }
}
// Workgroups of a linear_x3 launch: every workgroup streams ALL of the weight's planes from L2 (384 KB per 256 x 256 block and
// pass), so a pass should carry its two tiles and the launch should not be wider than it must: with one 16-row tile per CU
// (batch 32: 271 tiles on 256 workgroups) a d-memory pair moved 256 x 1.5 MB through L2 and was no faster than the fp32 GEMM it
// replaced; on 136 workgroups of two tiles it is (+1.6 % of the step), and at batch 64 (507 tiles) 128 workgroups of four tiles
// beat 256 of two (+1.1 % against +-0; profiles/r6_probes.md).  The launches run beside the decoder's chain: the CUs they
// leave alone are not idle.  FIRA_LINX3_MAX_WGS (default 136).
static int linear_x3_grid(int M) {
    static const int cap = [] { const char* e = getenv("FIRA_LINX3_MAX_WGS"); const int v = e ? atoi(e) : 136; return std::max(8, std::min(CF_GRID, v / 8 * 8)); }();
    const int n_tiles = cdiv(M, CF_TILE);
    const int g = (cdiv(n_tiles, CF_TMAX) + 7) / 8 * 8;
    return std::max(8, std::min(cap, g));
}
// The data-gradient shape of the same: out [M, 256] (+)= A [M, nkb * 256] B, B given as the planes of its nkb TRANSPOSED
// [256, 256] row blocks (Bt_kb[n][k] = B[256 kb + k][n]: for a weight stored [N, 256] with out = dY W, the k-major copy of its
// kb-th row block).  The d-memory products of the decoder's backward pass (d memory += dK|dV Wkv, K = 1024 per layer pair; three
// fp32 MFMA launches of 40-70 us beside the decoder's backward chain: skipping them -- a timing probe -- was worth +3.0 % of the
// step at batch 32, +2.8 % at batch 64, profiles/r6_probes.md).  One accumulator per tile over all K blocks; the panel is
// re-staged per block with the next block's rows requested under the product.
struct LinearX3KArgs {
    int n_rows;
    const float* A;          // [n, lda] fp32 rows, nkb * 256 columns used
    int lda;
    const uint16_t* Wx;      // planes of the nkb transposed blocks
    int nkb;
    float* out;              // [n, ldo], 256 columns
    int ldo;
    int accum;
};
template <int NP>
__global__ __launch_bounds__(CF_WAVES * 64) void linear_x3_kacc_kernel(const LinearX3KArgs a) {
    extern __shared__ __attribute__((aligned(16))) float cf_lds[];
    char* const sm_p = reinterpret_cast<char*>(cf_lds);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int grid = gridDim.x;                          // a multiple of 8 (the launch: >= two tiles per workgroup where there are enough)
    const int wg = (blockIdx.x & 7) * (grid / 8) + (blockIdx.x >> 3);
    const int n_tiles = (a.n_rows + CF_TILE - 1) / CF_TILE;
    const int tq = n_tiles / grid, tr = n_tiles % grid;
    const int t_beg = wg * tq + min(wg, tr), t_cnt = tq + (wg < tr ? 1 : 0);
    if (t_cnt == 0) return;
    const int col = wave * 16 + l15;
    const unsigned xlane = gx_wlane(wave, lane);
    const int a_q = gx_frag_base(l15, kq);
    constexpr size_t WX = 3 * (size_t)FIRA_D * FIRA_D;
    const rsrc_t rO = buf_rsrc(a.out, (unsigned)((size_t)a.n_rows * a.ldo * 4));
    for (int pass = 0; pass < t_cnt; pass += CF_TMAX) {
        const int nt = min(CF_TMAX, t_cnt - pass);
        const int row0 = (t_beg + pass) * CF_TILE;
        const int row_end = min(a.n_rows, row0 + nt * CF_TILE);
        uint4 bx3[3];
        cx_first<NP>(a.Wx, xlane, bx3);
        asm volatile("" ::: "memory");
        const float* arow[CF_RPW];
        f32x4v x[CF_RPW];
#pragma unroll
        for (int i = 0; i < CF_RPW; ++i) {
            arow[i] = a.A + (size_t)min(row0 + wave * CF_RPW + i, a.n_rows - 1) * a.lda + lane * 4;
            x[i] = *reinterpret_cast<const f32x4v*>(arow[i]);
        }
        float old[CF_TMAX][4];                            // the rows the product is added to: requested early
#pragma unroll
        for (int tt = 0; tt < CF_TMAX; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + tt * CF_TILE + 4 * kq + r;
                const unsigned o = (a.accum && tt < nt && row < row_end) ? ((unsigned)row * (unsigned)a.ldo + (unsigned)col) * 4u : FIRA_OOB;
                old[tt][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rO, o, 0, 0));      // (out of range: 0)
            }
        cf_acc acc[CF_TMAX];
#pragma unroll
        for (int tt = 0; tt < CF_TMAX; ++tt) acc[tt] = cf_acc{0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < a.nkb; ++kb) {
#pragma unroll
            for (int i = 0; i < CF_RPW; ++i) {
                const int row = row0 + wave * CF_RPW + i;
                gx_store_row4<NP>(sm_p, CX_PLANE, wave * CF_RPW + i, lane, row < row_end ? x[i] : f32x4v{0.f, 0.f, 0.f, 0.f});
            }
            lds_barrier();   
            if (kb + 1 < a.nkb) {                        // the next block's rows: in flight under this block's product
#pragma unroll
                for (int i = 0; i < CF_RPW; ++i) x[i] = *reinterpret_cast<const f32x4v*>(arow[i] + (kb + 1) * FIRA_D);
            }
            cx_product<CF_TMAX, NP>(sm_p, CX_PLANE, a_q, a.Wx + kb * WX, xlane, nt, bx3, acc,
                                    kb + 1 < a.nkb ? a.Wx + (kb + 1) * WX : nullptr);
            lds_barrier();                                // every wave has read its fragments: the panel can be overwritten
        }
#pragma unroll
        for (int tt = 0; tt < CF_TMAX; ++tt) {
            if (tt < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + tt * CF_TILE + 4 * kq + r;
                    const unsigned o = row < row_end ? ((unsigned)row * (unsigned)a.ldo + (unsigned)col) * 4u : FIRA_OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[tt][r] + old[tt][r]), rO, o, 0, 0);
                }
            }
        }
    }
}
int linear_x3_kacc(hipStream_t s, int M, const float* A, int lda, const uint16_t* Wx, int nkb, float* out, int ldo, bool accum,
                   bool one_plane) {
    if (M <= 0 || nkb <= 0) return 0;
    FIRA_REQUIRE(A && Wx && out && lda % 4 == 0 && (uintptr_t)A % 16 == 0 && lda >= nkb * FIRA_D, "linear_x3_kacc: bad argument");
    FIRA_REQUIRE((size_t)M * ldo * 4 < (1ull << 31), "linear_x3_kacc: %d x %d floats exceed the 2 GiB the kernel addresses", M, ldo);
    ProfScope prof(s, PROF_GEMM, 2.0 * M * (double)nkb * FIRA_D * FIRA_D, 4.0 * ((double)M * nkb * FIRA_D + 2.0 * M * FIRA_D) + 6.0 * nkb * FIRA_D * FIRA_D);
    LinearX3KArgs a{M, A, lda, Wx, nkb, out, ldo, accum ? 1 : 0};
    const size_t lds = 3 * CX_PLANE + 256;
    const int grid = linear_x3_grid(M);
    if (one_plane) hipLaunchKernelGGL(linear_x3_kacc_kernel<1>, dim3(grid), dim3(CF_WAVES * 64), lds, s, a);
    else hipLaunchKernelGGL(linear_x3_kacc_kernel<3>, dim3(grid), dim3(CF_WAVES * 64), lds, s, a);
    FIRA_CHECK_LAUNCH("linear_x3_kacc");
    return 0;
}

// ---- the generator projection's data gradient: d dec [R, 256] += dlogits [R, V] Wout [V, 256] -----------------------------
// (Model.py:54 backward.)  The head's largest product after the weight gradient: 6.7 GFLOP at batch 32 as one fp32 MFMA launch
// of 58 us on the auxiliary stream beside the copy head's backward kernel (skipping it -- a timing probe -- was worth +1.6 % of
// the step).  The reduction runs over the VOCABULARY, which is the contiguous dimension of dlogits and the row dimension of
// Wout: split_planes_t writes the planes of Wout's transposed 256-row blocks (Bt_kb[n][k] = Wout[256 kb + k][n], rows past V zero)
// once per step, and the product is the K-accumulating kernel above with the K blocks split over workgroups: workgroup = (pair of
// 16-row tiles, chunk of K blocks), partial sums added with float atomics (13 chunks at batch 32).
__global__ __launch_bounds__(256) void split_planes_t_kernel(const float* __restrict__ W, int V, uint16_t* __restrict__ planes, int np) {
    const int kb = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;        // j < 8192: n = j % 256 (coalesced reads), unit c = j / 256
    const int n = j & 255, c = j >> 8;
    float u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int v = kb * FIRA_D + c * 8 + i;
        u[i] = v < V ? W[(size_t)v * FIRA_D + n] : 0.f;
    }
    const int unit = ((n >> 4) * 8 + (c >> 2)) * 64 + (c & 3) * 16 + (n & 15);
    uint16_t* dst = planes + (size_t)kb * 3 * FIRA_D * FIRA_D + (size_t)unit * 8;
    for (int pl = 0; pl < np; ++pl) {
        const uint32_t p0 = gx_pack(u[0], u[1]), p1 = gx_pack(u[2], u[3]), p2 = gx_pack(u[4], u[5]), p3 = gx_pack(u[6], u[7]);
        *reinterpret_cast<uint4*>(dst + (size_t)pl * FIRA_D * FIRA_D) = uint4{p0, p1, p2, p3};
        if (pl + 1 < np) {
            u[0] = gx_rest_lo(u[0], p0); u[1] = gx_rest_hi(u[1], p0); u[2] = gx_rest_lo(u[2], p1); u[3] = gx_rest_hi(u[3], p1);
            u[4] = gx_rest_lo(u[4], p2); u[5] = gx_rest_hi(u[5], p2); u[6] = gx_rest_lo(u[6], p3); u[7] = gx_rest_hi(u[7], p3);
        }
    }
}
// planes [ceil(V / 256)][3][65536] bf16 of the transposed row blocks of W [V, 256]; one_plane: only plane 0 is written
int split_planes_t(hipStream_t s, const float* W, int V, uint16_t* planes, bool one_plane) {
    if (V <= 0) return 0;
    hipLaunchKernelGGL(split_planes_t_kernel, dim3(32, cdiv(V, FIRA_D)), dim3(256), 0, s, W, V, planes, one_plane ? 1 : 3);
    FIRA_CHECK_LAUNCH("split_planes_t");
    return 0;
}
struct DgradSplitKArgs {
    int n_rows;
    const float* A;          // [n, lda] fp32 rows, K valid columns
    int lda, K;
    const uint16_t* Wx;      // planes of the ceil(K / 256) transposed blocks
    int nkb, kb_chunk, n_pairs;
    float* out;              // [n, ldo], 256 columns: += by float atomics
    int ldo;
};
template <int NP>
__global__ __launch_bounds__(CF_WAVES * 64) void dgrad_x3_splitk_kernel(const DgradSplitKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float cf_lds[];
    char* const sm_p = reinterpret_cast<char*>(cf_lds);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int pi = blockIdx.x % a.n_pairs, ci = blockIdx.x / a.n_pairs;
    const int row0 = pi * CF_ROWS;
    const int row_end = min(a.n_rows, row0 + CF_ROWS);
    const int nt = (row_end - row0 + CF_TILE - 1) / CF_TILE;
    const int kb0 = ci * a.kb_chunk, kb1 = min(a.nkb, kb0 + a.kb_chunk);
    if (kb0 >= kb1) return;
    const int col = wave * 16 + l15;
    const unsigned xlane = gx_wlane(wave, lane);
    const int a_q = gx_frag_base(l15, kq);
    constexpr size_t WX = 3 * (size_t)FIRA_D * FIRA_D;
    uint4 bx3[3];
    cx_first<NP>(a.Wx + kb0 * WX, xlane, bx3);
    asm volatile("" ::: "memory");
    const float* arow[CF_RPW];
    bool live[CF_RPW];
    f32x4v x[CF_RPW];
    auto fetch = [&](int kb) {
#pragma unroll
        for (int i = 0; i < CF_RPW; ++i) {
            const int k = kb * FIRA_D + lane * 4;
            // (the last block is ragged: columns past K are padding of the row pitch, never multiplied)
            if (k + 3 < a.K) x[i] = *reinterpret_cast<const f32x4v*>(arow[i] + k);
            else x[i] = f32x4v{k < a.K ? arow[i][k] : 0.f, k + 1 < a.K ? arow[i][k + 1] : 0.f, k + 2 < a.K ? arow[i][k + 2] : 0.f, 0.f};
            if (!live[i]) x[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
        }
    };
#pragma unroll
    for (int i = 0; i < CF_RPW; ++i) {
        const int row = row0 + wave * CF_RPW + i;
        live[i] = row < row_end;
        arow[i] = a.A + (size_t)min(row, a.n_rows - 1) * a.lda;
    }
    fetch(kb0);
    cf_acc acc[CF_TMAX];
#pragma unroll
    for (int tt = 0; tt < CF_TMAX; ++tt) acc[tt] = cf_acc{0.f, 0.f, 0.f, 0.f};
    for (int kb = kb0; kb < kb1; ++kb) {
#pragma unroll
        for (int i = 0; i < CF_RPW; ++i) gx_store_row4<NP>(sm_p, CX_PLANE, wave * CF_RPW + i, lane, x[i]);
        lds_barrier();   
        if (kb + 1 < kb1) fetch(kb + 1);                 // in flight under this block's product
        cx_product<CF_TMAX, NP>(sm_p, CX_PLANE, a_q, a.Wx + kb * WX, xlane, nt, bx3, acc, kb + 1 < kb1 ? a.Wx + (kb + 1) * WX : nullptr);
        lds_barrier();   
    }
#pragma unroll
    for (int tt = 0; tt < CF_TMAX; ++tt) {
        if (tt < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + tt * CF_TILE + 4 * kq + r;
                if (row < row_end) unsafeAtomicAdd(a.out + (size_t)row * a.ldo + col, acc[tt][r]);
            }
        }
    }
}
// out [M, ldo >= 256] += A [M, K] W for W [K, 256] given as split_planes_t(W, K); K any size (A's row pitch lda >= K, lda % 4 == 0)
int dgrad_x3_splitk(hipStream_t s, int M, int K, const float* A, int lda, const uint16_t* Wx, float* out, int ldo, bool one_plane) {
    if (M <= 0 || K <= 0) return 0;
    FIRA_REQUIRE(A && Wx && out && lda % 4 == 0 && (uintptr_t)A % 16 == 0 && lda >= K && ldo >= FIRA_D, "dgrad_x3_splitk: bad argument");
    ProfScope prof(s, PROF_GEMM, 2.0 * M * (double)K * FIRA_D, 4.0 * ((double)M * K + 2.0 * M * FIRA_D) + 6.0 * (double)K * FIRA_D);
    DgradSplitKArgs a{};
    a.n_rows = M; a.A = A; a.lda = lda; a.K = K; a.Wx = Wx; a.nkb = cdiv(K, FIRA_D); a.out = out; a.ldo = ldo;
    a.n_pairs = cdiv(M, CF_ROWS);
    // K blocks per workgroup: about 256 workgroups in all, at least four blocks each (a workgroup streams 384 KB of planes per block)
    const int chunks = std::max(1, std::min(cdiv(a.nkb, 4), cdiv(CF_GRID, a.n_pairs)));
    a.kb_chunk = cdiv(a.nkb, chunks);
    const int grid = a.n_pairs * cdiv(a.nkb, a.kb_chunk);
    const size_t lds = 3 * CX_PLANE + 256;
    if (one_plane) hipLaunchKernelGGL(dgrad_x3_splitk_kernel<1>, dim3(grid), dim3(CF_WAVES * 64), lds, s, a);
    else hipLaunchKernelGGL(dgrad_x3_splitk_kernel<3>, dim3(grid), dim3(CF_WAVES * 64), lds, s, a);
    FIRA_CHECK_LAUNCH("dgrad_x3_splitk");
    return 0;
}

// one_plane: the engine's bf16 mode (operands rounded to bf16 once); else three terms per operand (fp32-accurate)
int linear_x3(hipStream_t s, int M, const float* X, int ldx, const uint16_t* Wx, int nb, const float* bias, float* out, int ldo,
              bool one_plane) {
    if (M <= 0 || nb <= 0) return 0;
    FIRA_REQUIRE(X && Wx && out && ldx % 4 == 0 && (uintptr_t)X % 16 == 0, "linear_x3: bad argument");
    FIRA_REQUIRE((size_t)M * ldo * 4 < (1ull << 31), "linear_x3: %d x %d floats exceed the 2 GiB the kernel addresses", M, ldo);
    ProfScope prof(s, PROF_GEMM, 2.0 * M * (double)nb * FIRA_D * FIRA_D, 4.0 * ((double)M * FIRA_D + (double)M * nb * FIRA_D) + 6.0 * nb * FIRA_D * FIRA_D);
    LinearX3Args a{M, X, ldx, Wx, nb, bias, out, ldo};
    const size_t lds = 3 * CX_PLANE + 256;
    const int grid = linear_x3_grid(M);
    if (one_plane) hipLaunchKernelGGL(linear_x3_kernel<1>, dim3(grid), dim3(CF_WAVES * 64), lds, s, a);
    else hipLaunchKernelGGL(linear_x3_kernel<3>, dim3(grid), dim3(CF_WAVES * 64), lds, s, a);
    FIRA_CHECK_LAUNCH("linear_x3");
    return 0;
}

int comb_fused_fwd(hipStream_t s, int n_rows, const float* Xc, const float* WqT, const float* WkT, const float* WoT,
                   const float* bqk, const float* bo, const float* vtab, int ldv, const int32_t* mark, float* qk, float* c,
                   const float* gamma, const float* beta, float* sum, float* y, const int32_t* y_rows, float* stats,
                   float dropout, uint64_t seed, uint32_t site_gate, uint32_t site_out, int bf16, const uint16_t* Wx) {
    if (n_rows <= 0) return 0;
    FIRA_REQUIRE(Xc && WqT && WkT && WoT && bqk && bo && vtab && mark && qk && c && gamma && beta && sum && y && stats,
                 "comb_fused_fwd: null pointer argument");
    FIRA_REQUIRE((uintptr_t)Xc % 16 == 0 && (uintptr_t)sum % 16 == 0 && (uintptr_t)y % 16 == 0 && (uintptr_t)WqT % 16 == 0,
                 "comb_fused_fwd: rows must be 16-byte aligned");
    FIRA_REQUIRE((size_t)n_rows * 2 * FIRA_D * 4 < (1ull << 31), "comb_fused_fwd: %d rows exceed the 2 GiB the kernel addresses", n_rows);
    // three [n,256]x[256,256] products; bytes: Xc in (twice: operand and residual, the second read an L2 hit), q|k, c, sum, y out
    ProfScope prof(s, PROF_COMB, 3.0 * 2.0 * n_rows * FIRA_D * FIRA_D, 4.0 * n_rows * FIRA_D * (1.0 + 2.0 + 1.0 + 1.0 + 1.0));
    CombFusedArgs a{};
    a.n_rows = n_rows; a.Xc = Xc; a.WqT = WqT; a.WkT = WkT; a.WoT = WoT; a.bqk = bqk; a.bo = bo; a.vtab = vtab; a.ldv = ldv;
    a.mark = mark; a.qk = qk; a.c = c; a.gamma = gamma; a.beta = beta; a.sum = sum; a.y = y; a.stats = stats; a.y_rows = y_rows;
    a.p = dropout; a.inv_keep = dropout > 0.f ? 1.0f / (1.0f - dropout) : 1.0f; a.seed = seed; a.site_gate = site_gate;
    a.site_out = site_out;
    static const int attr = [] {
        hipError_t e = hipFuncSetAttribute((const void*)comb_fused_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CF_LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)comb_fused_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CF_LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)comb_fused_fwd_kernel<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CF_LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)comb_fused_fwd_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CF_LDS);
        return e == hipSuccess ? 0 : set_err("comb_fused: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    }();
    if (attr) return attr;
    if (Wx) {                            // bf16 planes (Wx: planes of Wq | Wk | Wo as stored): three terms in fp32 mode, one in bf16 mode
        a.WqT = reinterpret_cast<const float*>(Wx);
        if (bf16) hipLaunchKernelGGL((comb_fused_fwd_kernel<false, 1>), dim3(CF_GRID), dim3(CF_WAVES * 64), CF_LDS, s, a);
        else hipLaunchKernelGGL((comb_fused_fwd_kernel<false, 3>), dim3(CF_GRID), dim3(CF_WAVES * 64), CF_LDS, s, a);
    } else
    if (bf16) hipLaunchKernelGGL(comb_fused_fwd_kernel<true>, dim3(CF_GRID), dim3(CF_WAVES * 64), CF_LDS, s, a);
    else hipLaunchKernelGGL(comb_fused_fwd_kernel<false>, dim3(CF_GRID), dim3(CF_WAVES * 64), CF_LDS, s, a);
    FIRA_CHECK_LAUNCH("comb_fused_fwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the block, one launch (replaces LayerNorm backward -> [n,256,256] data gradient -> gate backward ->
// [n,256,512] data gradient with accumulate: 5 + 8 + 6 + 14.5 us and three boundaries at batch 32):
//   rows      dy = dG[rows[r]],  xh = (s - mean) rstd,  h = dy gamma,  ds = (h - mean(h) - xh mean(h xh)) rstd,
//             dYc = ds * mask_out  (-> HBM: the output projection's weight-gradient operand, and the panel);
//             the wave keeps ds and its share of sum dy xh | sum dy (dgamma | dbeta) in registers
//   product   dc = dYc Wo                           (Wo as stored [out][in] IS k-major for this product)
//   gate      per element in the accumulator layout: (dq, dk, dv) from (q, k, v[mark], dc * mask_gate)  (rowops.hip:
//             combination_bwd_kernel's formulas); dq | dk -> HBM (the q|k weight gradient's operand) and two panels; dv summed per
//             mark value and column in registers
//   product   dX = dq Wq + dk Wk                    (Wq | Wk as stored)
//   rows      dG[rows[r]] = ds + dX
// and at the end the workgroup's partial {dgamma | dbeta} [512] and dvtab [4][256] rows for the deferred reduction.
// The backward kernel walks ONE tile per pass (a lane's registers: the LayerNorm row it keeps, its share of four column sums
// and the saved q | k of its elements on top of the product's fragments -- two tiles spilled 35-56 registers per lane)
constexpr int CB_TMAX = 1;
constexpr int CB_ROWS = CF_TILE * CB_TMAX;
constexpr int CB_RPW = CB_ROWS / CF_WAVES;

struct CombFusedBwdArgs {
    int n_rows;
    float* dG;                            // node-row gradients (read at rows[r], overwritten there)
    const int32_t* rows;
    const float *sum, *stats, *gamma;     // saved pre-norm rows [n,256], (mean, rstd) [n,2], LayerNorm weight
    const float *Wo, *Wqk;                // nn.Linear weights as stored: [256,256], [512,256]
    const float* qk;                      // saved [n,512]
    const float* vtab;
    int ldv;
    const int32_t* mark;
    float *dYc, *dqk;                     // [n,256], [n,512]
    float *part_ln, *part_v;              // [CF_GRID][512], [CF_GRID][1024]
    float p, inv_keep;
    uint64_t seed;
    uint32_t site_gate, site_out;
};

constexpr size_t CBX_PLANE = (size_t)CB_ROWS * 512;       // backward: bytes of one bf16 plane of a 16-row panel
constexpr size_t CB_LDS_X3 = 100 * 1024;                  // two panels of three planes (48 KB) + marks + the column sums (48 KB)
constexpr size_t CB_LDS_X3_2 = 2 * 3 * (size_t)2 * CF_TILE * 512 + 256 + 48 * 1024 + 1024;   // the same at two tiles per pass (145 KB)
// TM (round 6, late): tiles per pass.  Every pass streams the three products' weight planes (1.15 MB in fp32 mode) through the
// CU, so a workgroup with two tiles (batch 64: 429 tiles of code rows on 256 workgroups) pays for them twice at one tile per pass
// and once at two: +0.7 % fp32 / +1.3 % bf16 at batch 64.  At batch 32 (229 tiles: nobody has two) the two-tile form LOSES 0.8 %
// (the three-term form spills 18 registers, 145 KB of LDS): TM = 2 only when some workgroup has more than one tile, plane forms
// only (the fp32-panel forms spill 35-56 registers at two tiles).
template <bool BF, int NP = 0, int TM = 1>
__global__ __launch_bounds__(CF_WAVES * 64) void comb_fused_bwd_kernel(const CombFusedBwdArgs a) {
    constexpr int CB_TMAX = TM, CB_ROWS = CF_TILE * TM, CB_RPW = CB_ROWS / CF_WAVES;      // (shadow the one-tile constants above)
    constexpr size_t CBX_PLANE = (size_t)CB_ROWS * 512;
    constexpr bool X3 = NP > 0;
    constexpr int NPX = X3 ? NP : 3;
    extern __shared__ __attribute__((aligned(16))) float cf_lds[];
    constexpr size_t PANEL = X3 ? 3 * CBX_PLANE / 4 : (size_t)CB_ROWS * FIRA_D;      // floats of one panel
    float* const sm_u = cf_lds;                                            // panel 1 [16][256]: dYc, then dq, then dX (X3: fp32 only for dX)
    float* const sm_w = cf_lds + PANEL;                                    // panel 2 [16][256]: dk
    char* const sm_p1 = reinterpret_cast<char*>(sm_u);                     // X3: the panels as three bf16 planes each
    char* const sm_p2 = reinterpret_cast<char*>(sm_w);
    int* const sm_mark = reinterpret_cast<int*>(cf_lds + 2 * PANEL);
    // column sums of the whole workgroup, kept in LDS between the passes (in registers they cost 12 per lane next to the
    // product's fragments): every lane owns its slots -- plain read-modify-write, no atomics
    float* const red = cf_lds + 2 * PANEL + 64;                         // [16 waves][dgamma 256 | dbeta 256]
    float* const redv = red + CF_WAVES * 2 * FIRA_D;                    // [4 kq][4 marks][256 columns]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    const int wg = (blockIdx.x & 7) * (CF_GRID / 8) + (blockIdx.x >> 3);
    const int n_tiles = (a.n_rows + CF_TILE - 1) / CF_TILE;
    const int tq = n_tiles / CF_GRID, tr = n_tiles % CF_GRID;
    const int t_beg = wg * tq + min(wg, tr), t_cnt = tq + (wg < tr ? 1 : 0);
    int a_off[4], d_off[4], r_off[CB_RPW];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) a_off[ii] = cf_off(l15, kq * 4 + ii);
#pragma unroll
    for (int r = 0; r < 4; ++r) d_off[r] = cf_off(4 * kq + r, (wave * 16 + l15) >> 2) + (l15 & 3);
#pragma unroll
    for (int i = 0; i < CB_RPW; ++i) r_off[i] = cf_off(wave * CB_RPW + i, lane);
    const int col = wave * 16 + l15;
    const unsigned wlane = (unsigned)((kq * 16) * FIRA_D + col) * 4u;
    const rsrc_t rQK = buf_rsrc(a.qk, (unsigned)((size_t)a.n_rows * 2 * FIRA_D * 4));
    const rsrc_t rDQK = buf_rsrc(a.dqk, (unsigned)((size_t)a.n_rows * 2 * FIRA_D * 4));
    const float is = 1.0f / 5.656854249492381f;
    const unsigned xlane = gx_wlane(wave, lane);
    const int a_q = gx_frag_base(l15, kq);
    const uint16_t* const WTx = reinterpret_cast<const uint16_t*>(a.Wo);      // X3: planes of Wq^T | Wk^T | Wo^T (three matrices)
    constexpr size_t WX = 3 * (size_t)FIRA_D * FIRA_D;
    *reinterpret_cast<f32x4v*>(&red[wave * 2 * FIRA_D + lane * 4]) = f32x4v{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4v*>(&red[wave * 2 * FIRA_D + FIRA_D + lane * 4]) = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; ++m) redv[(kq * 4 + m) * FIRA_D + col] = 0.f;

    for (int pass = 0; pass < t_cnt; pass += CB_TMAX) {
        const int nt = min(CB_TMAX, t_cnt - pass);
        const int row0 = (t_beg + pass) * CF_TILE;
        const int row_end = min(a.n_rows, row0 + nt * CF_TILE);
        float bx[16];
        uint4 bx3[3];
        if constexpr (X3) cx_first<NPX>(WTx + 2 * WX, xlane, bx3);
        else cf_first_chunk(a.Wo, wlane, bx);
        asm volatile("" ::: "memory");
        // ------------------------------------------------------------ 1. LayerNorm backward of this wave's rows -> dYc
        const int rbase = row0 + wave * CB_RPW;
        int nr[CB_RPW];
        f32x4v ds[CB_RPW];
        {
            f32x4v d[CB_RPW], sv[CB_RPW];
            float mean[CB_RPW], rstd[CB_RPW];
            const f32x4v gam = *reinterpret_cast<const f32x4v*>(a.gamma + lane * 4);      // (per pass: an L2 hit, four registers less)
#pragma unroll
            for (int i = 0; i < CB_RPW; ++i) nr[i] = a.rows[min(rbase + i, a.n_rows - 1)];
            if (t < CB_ROWS) sm_mark[t] = a.mark[min(row0 + t, a.n_rows - 1)];
#pragma unroll
            for (int i = 0; i < CB_RPW; ++i) {
                const int row = min(rbase + i, a.n_rows - 1);
                d[i] = *reinterpret_cast<const f32x4v*>(a.dG + (size_t)nr[i] * FIRA_D + lane * 4);
                sv[i] = *reinterpret_cast<const f32x4v*>(a.sum + (size_t)row * FIRA_D + lane * 4);
                mean[i] = a.stats[2 * row];
                rstd[i] = a.stats[2 * row + 1];
            }
#pragma unroll
            for (int i = 0; i < CB_RPW; ++i) {
                const bool live = rbase + i < row_end;              // wave-uniform
                if (!live) { d[i] = f32x4v{0.f, 0.f, 0.f, 0.f}; rstd[i] = 0.f; }
                const f32x4v xh = (sv[i] - mean[i]) * rstd[i];
                *reinterpret_cast<f32x4v*>(&red[wave * 2 * FIRA_D + lane * 4]) += d[i] * xh;
                *reinterpret_cast<f32x4v*>(&red[wave * 2 * FIRA_D + FIRA_D + lane * 4]) += d[i];
                const f32x4v h = d[i] * gam;
                const f32x4v hx = h * xh;
                const float m1 = wave_sum(h.x + h.y + h.z + h.w) * (1.0f / FIRA_D);
                const float m2 = wave_sum(hx.x + hx.y + hx.z + hx.w) * (1.0f / FIRA_D);
                ds[i] = (h - m1 - xh * m2) * rstd[i];
                f32x4v o4 = ds[i];
                if (a.p > 0.f) {
                    const uint32_t e0 = (uint32_t)(rbase + i) * FIRA_D + lane * 4;
                    o4.x *= dropout_scale(a.seed, a.site_out, e0 + 0, a.p, a.inv_keep);
                    o4.y *= dropout_scale(a.seed, a.site_out, e0 + 1, a.p, a.inv_keep);
                    o4.z *= dropout_scale(a.seed, a.site_out, e0 + 2, a.p, a.inv_keep);
                    o4.w *= dropout_scale(a.seed, a.site_out, e0 + 3, a.p, a.inv_keep);
                }
                if constexpr (X3) gx_store_row4<NPX>(sm_p1, CBX_PLANE, wave * CB_RPW + i, lane, o4);
                else *reinterpret_cast<f32x4v*>(&sm_u[r_off[i]]) = o4;                 // (rows past the end: zeros)
                if (live) *reinterpret_cast<f32x4v*>(a.dYc + (size_t)(rbase + i) * FIRA_D + lane * 4) = o4;
            }
        }
        lds_barrier();   
        // ------------------------------------------------------------ 2. dc = dYc Wo; the saved q | k of this lane's elements on the way
        float qs[CB_TMAX][4], ks[CB_TMAX][4];
        auto load_qk = [&]() {
#pragma unroll
            for (int tt = 0; tt < CB_TMAX; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + tt * CF_TILE + 4 * kq + r;
                    const unsigned o2 = row < row_end ? ((unsigned)row * 2 * FIRA_D + col) * 4u : FIRA_OOB;   // past the end: reads 0
                    qs[tt][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rQK, o2, 0, 0));
                    ks[tt][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rQK, o2, FIRA_D * 4, 0));
                }
            asm volatile("" ::: "memory");
        };
        // (fp32: requested ahead of the product's 10 us of MFMA issue; bf16: the product is short and its packed fragments need the
        //  registers -- the request follows it)
        if constexpr (!BF) load_qk();
        cf_acc ac[CB_TMAX];
#pragma unroll
        for (int tt = 0; tt < CB_TMAX; ++tt) ac[tt] = cf_acc{0.f, 0.f, 0.f, 0.f};
        if constexpr (X3) cx_product<CB_TMAX, NPX>(sm_p1, CBX_PLANE, a_q, WTx + 2 * WX, xlane, nt, bx3, ac, WTx);
        else
        cf_product<BF, CB_TMAX>(sm_u, a_off, a.Wo, wlane, nt, bx, ac, a.Wqk);    // (Wq's first chunk: in flight under the gate's arithmetic)
        if constexpr (BF) load_qk();
        lds_barrier();                                                   // the panel's A fragments are consumed
        // ------------------------------------------------------------ 3. gate backward in the accumulator layout
#pragma unroll
        for (int tt = 0; tt < CB_TMAX; ++tt) {
            if (tt < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rl = tt * CF_TILE + 4 * kq + r, row = row0 + rl;
                    const bool live = row < row_end;
                    float dc = ac[tt][r];
                    if (a.p > 0.f) dc *= dropout_scale(a.seed, a.site_gate, (uint32_t)row * FIRA_D + col, a.p, a.inv_keep);
                    const int m = sm_mark[rl];
                    const float v = a.vtab[(size_t)m * a.ldv + col];            // (four rows of 1 KiB: L1 / L2 hits)
                    const float q = qs[tt][r], k = ks[tt][r];
                    float g0, g1;
                    gate_elem(q, k, v, g0, g1);
                    const float dg0 = dc * k, dg1 = dc * v;
                    const float dot = g0 * dg0 + g1 * dg1;
                    const float da = g0 * (dg0 - dot), db = g1 * (dg1 - dot);
                    const float dq = live ? (da * k + db * v) * is : 0.f;
                    const float dk = live ? dc * g0 + da * q * is : 0.f;
                    const float dv = live ? dc * g1 + db * q * is : 0.f;
                    const unsigned o2 = live ? ((unsigned)row * 2 * FIRA_D + col) * 4u : FIRA_OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dq), rDQK, o2, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dk), rDQK, o2, FIRA_D * 4, 0);
                    redv[(kq * 4 + m) * FIRA_D + col] += dv;             // (dv = 0 past the end; this lane's own slot)
                    if constexpr (X3) {
                        gx_store_elem<NPX>(sm_p1, CBX_PLANE, rl, col, dq);
                        gx_store_elem<NPX>(sm_p2, CBX_PLANE, rl, col, dk);
                    } else {
                    sm_u[d_off[r] + tt * (CF_TILE * FIRA_D)] = dq;
                    sm_w[d_off[r] + tt * (CF_TILE * FIRA_D)] = dk;
                    }
                }
            }
        }
        lds_barrier();   
        // ------------------------------------------------------------ 4. dX = dq Wq + dk Wk
        cf_acc ax[CB_TMAX];
#pragma unroll
        for (int tt = 0; tt < CB_TMAX; ++tt) ax[tt] = cf_acc{0.f, 0.f, 0.f, 0.f};
        if constexpr (X3) {
            cx_product<CB_TMAX, NPX>(sm_p1, CBX_PLANE, a_q, WTx, xlane, nt, bx3, ax, WTx + WX);
            cx_product<CB_TMAX, NPX>(sm_p2, CBX_PLANE, a_q, WTx + WX, xlane, nt, bx3, ax);
        } else {
        cf_product<BF, CB_TMAX>(sm_u, a_off, a.Wqk, wlane, nt, bx, ax, a.Wqk + (size_t)FIRA_D * FIRA_D);
        cf_product<BF, CB_TMAX>(sm_w, a_off, a.Wqk + (size_t)FIRA_D * FIRA_D, wlane, nt, bx, ax);
        }
        lds_barrier();   
#pragma unroll
        for (int tt = 0; tt < CB_TMAX; ++tt) {
            if (tt < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sm_u[d_off[r] + tt * (CF_TILE * FIRA_D)] = ax[tt][r];
            }
        }
        lds_barrier();   
        // ------------------------------------------------------------ 5. rows: dG = ds + dX
#pragma unroll
        for (int i = 0; i < CB_RPW; ++i) {
            if (rbase + i >= row_end) continue;                         // wave-uniform
            const f32x4v x = *reinterpret_cast<const f32x4v*>(&sm_u[r_off[i]]);
            *reinterpret_cast<f32x4v*>(a.dG + (size_t)nr[i] * FIRA_D + lane * 4) = ds[i] + x;
        }
        lds_barrier();                                                   // the next pass (or the reduction below) reuses the panels
    }
    // ---------------------------------------------------------------- the workgroup's partial rows (every workgroup writes: the
    // reducer sums CF_GRID of them)
    lds_barrier();   
    if (t < 2 * FIRA_D) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < CF_WAVES; ++w) acc += red[w * 2 * FIRA_D + t];
        a.part_ln[(size_t)blockIdx.x * 2 * FIRA_D + t] = acc;
    }
    {
        const int m = t >> 8, cc = t & 255;                             // 1024 threads = 4 marks x 256 columns
        a.part_v[(size_t)blockIdx.x * 4 * FIRA_D + t] = (redv[(0 * 4 + m) * FIRA_D + cc] + redv[(1 * 4 + m) * FIRA_D + cc]) +
                                                        (redv[(2 * 4 + m) * FIRA_D + cc] + redv[(3 * 4 + m) * FIRA_D + cc]);
    }
}

int comb_fused_bwd_parts() { return CF_GRID; }
int comb_fused_bwd(hipStream_t s, int n_rows, float* dG, const int32_t* rows, const float* sum, const float* stats,
                   const float* gamma, const float* Wo, const float* Wqk, const float* qk, const float* vtab, int ldv,
                   const int32_t* mark, float* dYc, float* dqk, float* part_ln, float* part_v, float dropout, uint64_t seed,
                   uint32_t site_gate, uint32_t site_out, int bf16, const uint16_t* WTx) {
    if (n_rows <= 0) return 0;
    FIRA_REQUIRE(dG && rows && sum && stats && gamma && Wo && Wqk && qk && vtab && mark && dYc && dqk && part_ln && part_v,
                 "comb_fused_bwd: null pointer argument");
    FIRA_REQUIRE((uintptr_t)dG % 16 == 0 && (uintptr_t)sum % 16 == 0 && (uintptr_t)dYc % 16 == 0 && (uintptr_t)Wo % 16 == 0,
                 "comb_fused_bwd: rows must be 16-byte aligned");
    FIRA_REQUIRE((size_t)n_rows * 2 * FIRA_D * 4 < (1ull << 31), "comb_fused_bwd: %d rows exceed the 2 GiB the kernel addresses", n_rows);
    // three [n,256]x[256,256] products; bytes: dy, s in; q|k in; dYc, dq|dk out; the node rows out
    ProfScope prof(s, PROF_COMB, 3.0 * 2.0 * n_rows * FIRA_D * FIRA_D, 4.0 * n_rows * FIRA_D * (2.0 + 2.0 + 1.0 + 2.0 + 1.0));
    CombFusedBwdArgs a{};
    a.n_rows = n_rows; a.dG = dG; a.rows = rows; a.sum = sum; a.stats = stats; a.gamma = gamma; a.Wo = Wo; a.Wqk = Wqk; a.qk = qk;
    a.vtab = vtab; a.ldv = ldv; a.mark = mark; a.dYc = dYc; a.dqk = dqk; a.part_ln = part_ln; a.part_v = part_v;
    a.p = dropout; a.inv_keep = dropout > 0.f ? 1.0f / (1.0f - dropout) : 1.0f; a.seed = seed; a.site_gate = site_gate;
    a.site_out = site_out;
    static const int attr = [] {
        hipError_t e = hipFuncSetAttribute((const void*)comb_fused_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CF_LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)comb_fused_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CF_LDS);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)comb_fused_bwd_kernel<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CB_LDS_X3);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)comb_fused_bwd_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CB_LDS_X3);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)comb_fused_bwd_kernel<false, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CB_LDS_X3_2);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)comb_fused_bwd_kernel<false, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CB_LDS_X3_2);
        return e == hipSuccess ? 0 : set_err("comb_fused: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    }();
    if (attr) return attr;
    if (WTx) {                           // bf16 planes (WTx: planes of Wq^T | Wk^T | Wo^T): three terms in fp32 mode, one in bf16 mode
        a.Wo = reinterpret_cast<const float*>(WTx);
        // FIRA_COMB_BWD_TILES=1|2 forces the tiles per pass (A/B switch); default: two once some workgroup has more than one tile
        static const int tm_env = [] { const char* e = getenv("FIRA_COMB_BWD_TILES"); return e ? atoi(e) : 0; }();
        const int n_tiles = (n_rows + CF_TILE - 1) / CF_TILE;
        const bool two = tm_env == 2 || (tm_env != 1 && n_tiles > CF_GRID);
        if (two) {
            if (bf16) hipLaunchKernelGGL((comb_fused_bwd_kernel<false, 1, 2>), dim3(CF_GRID), dim3(CF_WAVES * 64), CB_LDS_X3_2, s, a);
            else hipLaunchKernelGGL((comb_fused_bwd_kernel<false, 3, 2>), dim3(CF_GRID), dim3(CF_WAVES * 64), CB_LDS_X3_2, s, a);
        } else
        if (bf16) hipLaunchKernelGGL((comb_fused_bwd_kernel<false, 1>), dim3(CF_GRID), dim3(CF_WAVES * 64), CB_LDS_X3, s, a);
        else hipLaunchKernelGGL((comb_fused_bwd_kernel<false, 3>), dim3(CF_GRID), dim3(CF_WAVES * 64), CB_LDS_X3, s, a);
    } else
    if (bf16) hipLaunchKernelGGL(comb_fused_bwd_kernel<true>, dim3(CF_GRID), dim3(CF_WAVES * 64), CF_LDS, s, a);
    else hipLaunchKernelGGL(comb_fused_bwd_kernel<false>, dim3(CF_GRID), dim3(CF_WAVES * 64), CF_LDS, s, a);
    FIRA_CHECK_LAUNCH("comb_fused_bwd");
    return 0;
}

// Wt[i] = W[i]^T for a table of [256,256] matrices anywhere in memory (the Combination weights live in the flat parameter buffer)
__global__ __launch_bounds__(256) void transpose256_table_kernel(const TransposeTable tab) {
    __shared__ float tile[64][65];
    const float* src = tab.src[blockIdx.z];
    float* dst = tab.dst[blockIdx.z];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) tile[r][tx] = src[(size_t)(r0 + r) * FIRA_D + c0 + tx];
    __syncthreads();
    for (int r = ty; r < 64; r += 4) dst[(size_t)(c0 + r) * FIRA_D + r0 + tx] = tile[tx][r];
}
int transpose256_table(hipStream_t s, const TransposeTable& tab) {
    if (tab.n <= 0) return 0;
    hipLaunchKernelGGL(transpose256_table_kernel, dim3(FIRA_D / 64, FIRA_D / 64, tab.n), dim3(256), 0, s, tab);
    FIRA_CHECK_LAUNCH("transpose256_table");
    return 0;
}

}  // namespace fira

extern "C" {
int fira_linear_dgrad_x3(void* stream, int M, int K, const float* dy, int lddy, const uint16_t* wt_planes, float* dx, int lddx,
                         int accumulate, int dtype) {
    FIRA_REQUIRE(dy && wt_planes && dx && M > 0 && K > 0 && K % FIRA_D == 0 && lddx >= FIRA_D, "fira_linear_dgrad_x3: bad argument");
    FIRA_REQUIRE(dtype == FIRA_F32X3 || dtype == FIRA_BF16X1, "fira_linear_dgrad_x3: dtype must be FIRA_F32X3 or FIRA_BF16X1");
    return fira::linear_x3_kacc((hipStream_t)stream, M, dy, lddy, wt_planes, K / FIRA_D, dx, lddx, accumulate != 0,
                                dtype == FIRA_BF16X1);
}
size_t fira_dgrad_x3_splitk_planes_bytes(int K) { return (size_t)((K + FIRA_D - 1) / FIRA_D) * 3 * FIRA_D * FIRA_D * sizeof(uint16_t); }
int fira_dgrad_x3_splitk(void* stream, int M, int K, const float* dy, int lddy, const float* W, uint16_t* planes_ws, float* dx,
                         int lddx, int dtype) {
    FIRA_REQUIRE(dy && W && planes_ws && dx && M > 0 && K > 0, "fira_dgrad_x3_splitk: bad argument");
    FIRA_REQUIRE(dtype == FIRA_F32X3 || dtype == FIRA_BF16X1, "fira_dgrad_x3_splitk: dtype must be FIRA_F32X3 or FIRA_BF16X1");
    const bool one = dtype == FIRA_BF16X1;
    const int rc = fira::split_planes_t((hipStream_t)stream, W, K, planes_ws, one);
    return rc ? rc : fira::dgrad_x3_splitk((hipStream_t)stream, M, K, dy, lddy, planes_ws, dx, lddx, one);
}
int fira_linear_x3(void* stream, int M, int N, const float* x, int ldx, const uint16_t* w_planes, const float* bias, float* out,
                   int ldo, int dtype) {
    FIRA_REQUIRE(x && w_planes && out && M > 0 && N > 0 && N % FIRA_D == 0 && ldo >= N, "fira_linear_x3: bad argument");
    FIRA_REQUIRE(dtype == FIRA_F32X3 || dtype == FIRA_BF16X1, "fira_linear_x3: dtype must be FIRA_F32X3 or FIRA_BF16X1");
    return fira::linear_x3((hipStream_t)stream, M, x, ldx, w_planes, N / FIRA_D, bias, out, ldo, dtype == FIRA_BF16X1);
}
int fira_combination_block_bwd(void* stream, int n_rows, float* dG, const int32_t* rows, const float* sum, const float* stats,
                               const float* gamma, const float* Wo, const float* Wqk, const float* qk, const float* vtab, int ldv,
                               const int32_t* mark, float* dYc, float* dqk, float* dgamma, float* dbeta, float* dvtab, int lddv,
                               float* part, float dropout, uint64_t seed, uint32_t site_gate, uint32_t site_out, int dtype) {
    FIRA_REQUIRE(dropout >= 0.f && dropout < 1.f, "fira_combination_block_bwd: dropout must be in [0,1)");
    FIRA_REQUIRE(dtype >= FIRA_F32 && dtype <= FIRA_BF16X1, "fira_combination_block_bwd: dtype must be FIRA_F32, FIRA_BF16, FIRA_F32X3 or FIRA_BF16X1");
    FIRA_REQUIRE(dgamma && dbeta && dvtab && part, "fira_combination_block_bwd: null pointer argument");
    const int nb = fira::comb_fused_bwd_parts();
    float* part_ln = part;
    float* part_v = part + (size_t)nb * 2 * FIRA_D;
    if (int rc = fira::comb_fused_bwd((hipStream_t)stream, n_rows, dG, rows, sum, stats, gamma, Wo, Wqk, qk, vtab, ldv, mark, dYc, dqk,
                                      part_ln, part_v, dropout, seed, site_gate, site_out, dtype == FIRA_BF16 || dtype == FIRA_BF16X1,
                                      dtype >= FIRA_F32X3 ? reinterpret_cast<const uint16_t*>(Wo) : nullptr))
        return rc;
    if (n_rows <= 0) return 0;
    fira::RedTable tab;
    tab.e[0] = fira::RedEntry{dgamma, part_ln, FIRA_D, nb, 2 * FIRA_D};
    tab.e[1] = fira::RedEntry{dbeta, part_ln + FIRA_D, FIRA_D, nb, 2 * FIRA_D};
    for (int k = 0; k < 4; ++k) tab.e[2 + k] = fira::RedEntry{dvtab + (size_t)k * lddv, part_v + k * FIRA_D, FIRA_D, nb, 4 * FIRA_D};
    tab.n = 6;
    return fira::deferred_reduce((hipStream_t)stream, tab);
}
int fira_combination_block_bwd_part_floats(void) { return fira::comb_fused_bwd_parts() * 6 * FIRA_D; }
int fira_combination_block_fwd(void* stream, int n_rows, const float* Xc, const float* WqT, const float* WkT, const float* WoT,
                               const float* bqk, const float* bo, const float* vtab, int ldv, const int32_t* mark, float* qk,
                               float* c, const float* gamma, const float* beta, float* sum, float* y, const int32_t* y_rows,
                               float* stats, float dropout, uint64_t seed, uint32_t site_gate, uint32_t site_out, int dtype) {
    FIRA_REQUIRE(dropout >= 0.f && dropout < 1.f, "fira_combination_block_fwd: dropout must be in [0,1)");
    FIRA_REQUIRE(dtype >= FIRA_F32 && dtype <= FIRA_BF16X1, "fira_combination_block_fwd: dtype must be FIRA_F32, FIRA_BF16, FIRA_F32X3 or FIRA_BF16X1");
    return fira::comb_fused_fwd((hipStream_t)stream, n_rows, Xc, WqT, WkT, WoT, bqk, bo, vtab, ldv, mark, qk, c, gamma, beta, sum,
                                y, y_rows, stats, dropout, seed, site_gate, site_out, dtype == FIRA_BF16 || dtype == FIRA_BF16X1,
                                dtype >= FIRA_F32X3 ? reinterpret_cast<const uint16_t*>(WqT) : nullptr);
}
}
