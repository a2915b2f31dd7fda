// One GCN layer of the encoder as ONE launch (reference gnn_transformer.py:74-86), in the folded form of engine.hip:
//
//   forward    Y = LayerNorm( dropout( (A_hat X) W21^T + b2 + (A_hat 1) c21^T ) + X )
//   backward   V = A_hat dY ;   dX += V W21        (A_hat (dY W21) = (A_hat dY) W21: the aggregation moves in FRONT of
//                                                   the product, so both directions are "gather, then multiply")
//
// replacing the three launches  CSR SpMM -> [Nc,256]x[256,256] GEMM -> add+LayerNorm  (and  GEMM -> SpMM  in the backward
// pass) and the round trip of the aggregated rows Z = A_hat X through HBM (2 x 9.6 MB per layer at batch 32).  The weight
// gradient uses the same identity: dW21 = dY^T (A_hat X) = (A_hat dY)^T X = V^T X, so Z is never stored; the backward
// launch leaves V for it.
//
// Geometry.  The product is MFMA-bound in fp32 (1.3 GFLOP per launch at batch 32 = 8.4 us of the whole chip), so what
// matters is how evenly the rows fall on the 256 CUs and how often the 256 KB weight is streamed:
//   * ONE workgroup (16 waves) per CU, each owning a CONTIGUOUS range of 16-row tiles: ceil or floor of n_tiles / 256
//     (10 000 rows = 627 tiles: 2 or 3 tiles per CU; 32-row workgroups would put 64 rows on some CUs and 32 on others:
//     version 1 of this file, 36 us per launch against 33 us for the three separate kernels);
//   * up to four tiles (64 rows) per pass: the gathered rows sit in a 64 KB fp32 LDS panel, every wave owns 16 output
//     columns of ALL of them, so a pass streams the weight once (64 FLOP per byte of L2 traffic at four tiles);
//   * v_mfma_f32_16x16x4_f32 (same rate as 32x32x2; its 16-row tile is what makes the split even); one accumulator per
//     tile: up to four independent chains per wave cover the 40-cycle dependent-issue latency.
// Phases of a pass:
//   1. gather   wave w aggregates rows 4w .. 4w+3 (64 lanes x float4 = one 1 KiB neighbour row per load; row offsets and
//               the first 16 (col, val) pairs of all four rows in two batched round trips, four neighbour rows of each
//               of the four rows per further round trip) into the panel.  Panel rows are 256 floats with the 16-byte
//               columns XOR-swizzled by (row & 11): the fragment fetch below (ds_read_b128, 16 rows x two k quarters per
//               lane group) is then conflict-free, and so are the whole-row accesses;
//   2. product  A fragments (16 contiguous k per lane) from the panel, B fragments straight from the L2-resident k-major
//               weight (whole 64-byte segments per k row, requested a 64-wide k chunk ahead; the empty asm statements pin
//               that order -- the scheduler otherwise sinks every load to its first use);
//   3. rows     the results go back through the panel, and wave w finishes rows 4w .. 4w+3 with whole-row (1 KiB,
//               coalesced) accesses: bias, rank-1 term, dropout, residual, LayerNorm, second compact copy (forward) /
//               accumulate into the gradient rows (backward).
#include "engine.h"
#include "mfma_frag.h"
#include "epilogue.h"
#include "x3.h"

namespace fira {

constexpr int GF_TILE = 16;               // rows per MFMA tile
constexpr int GF_TMAX = 4;                // tiles per pass (64 panel rows)
constexpr int GF_ROWS = GF_TILE * GF_TMAX;
constexpr int GF_WAVES = 16;
constexpr int GF_RPW = GF_ROWS / GF_WAVES;   // rows per wave in the gather / row phases
constexpr int GF_GRID = 256;              // one workgroup per CU
// dynamic LDS: the panel + the row sums; asking for more than half of a CU's 160 KB keeps the dispatcher from placing two
// of these workgroups on one CU while another CU stays empty
constexpr size_t GF_LDS = 84 * 1024;
// X3 (round 6): the product on the bf16 matrix cores at fp32 accuracy.  The gathered rows are split once, by the wave that
// gathered them, into three bf16 terms  u = hi + mid + lo  (each the RNE rounding of what the previous ones left) and stored as
// three bf16 PLANES of the panel ([64 rows][256 k] x 2 B each: 96 KB); the weight arrives pre-split in fragment order
// (gcn_split_planes: once per step and layer, beside the fold).  A k step of 32 is then six v_mfma_f32_16x16x32_bf16 per tile
// -- lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi, smallest first, fp32 accumulation; the three dropped terms are below 2^-24
// of the product -- instead of eight v_mfma_f32_16x16x4_f32: 96 against 256 matrix-core cycles.  Per pass of three tiles the
// product phase is 3.8 us of matrix-core time against 10.2 us, 1.15 MB of fragment reads from LDS (256 B / clk: 1.9 us) and
// 384 KB of weight planes from L2 (2.6 us at 64 B / clk and CU).  Plane rows are 512 B with the 16-byte columns XOR-swizzled by
// (row & 15): the 16 lanes of a ds_read_b128 group read 16 distinct bank quads.
constexpr size_t GX_PLANE = (size_t)GF_ROWS * 512;                 // bytes of one bf16 plane of the panel
constexpr size_t GF_LDS_X3 = 3 * GX_PLANE + 1024;                  // 97 KB: the planes + the row sums (one workgroup per CU)

typedef float f32x4acc __attribute__((ext_vector_type(4)));

struct GcnFusedArgs {
    int n_rows;
    const int32_t *rowptr, *col;
    const float* val;
    const float* X;          // gather source rows [n_rows, 256]
    const float* W;          // K-MAJOR weight Wk [256 k][256 n]: out[m][n] = sum_k U[m][k] Wk[k][n]
    // forward epilogue
    const float *bias, *r1_col, *res, *gamma, *beta;
    float *sum, *y, *stats, *rowsum_out;
    const int32_t* slot2;
    float* y2;
    float p, inv_keep;
    uint64_t seed;
    uint32_t site;
    // backward epilogue
    float* u_out;            // V = A_hat dY rows (the weight gradient's operand)
    float* acc_out;          // rows the product is added to
};

// float offset of 16-byte column `quad` (0..63) of panel row `row`
__device__ __forceinline__ int gf_off(int row, int quad) { return row * FIRA_D + ((quad ^ (row & 11)) << 2); }

// NP = 0: the fp32 panel (fp32 MFMA chains, or BF = its fragments rounded to bf16 by every wave); NP = 3: the X3 form above;
// NP = 1: the bf16 mode of the engine on the same machinery -- one plane, both operands rounded to bf16 once (RNE), one MFMA per
// k step and tile (the values the BF form multiplies, a third of the X3 form's LDS and L2 traffic)
template <bool BF, bool BWD, int NP = 0>
__global__ __launch_bounds__(GF_WAVES * 64) void gcn_fused_kernel(const GcnFusedArgs a) {
    constexpr bool X3 = NP > 0;
    extern __shared__ __attribute__((aligned(16))) float gf_lds[];
    float* const sm_u = gf_lds;                          // [64][256] swizzled (X3: the RESULT rows only, over the dead planes)
    char* const sm_p = reinterpret_cast<char*>(gf_lds);  // X3: three bf16 planes [64][256]
    float* const sm_rs = X3 ? gf_lds + 3 * GX_PLANE / 4 : gf_lds + GF_ROWS * FIRA_D;      // [64] row sums of A_hat
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l15 = lane & 15, kq = lane >> 4;
    // workgroup b runs on XCD b % 8: logical id (b % 8) * 32 + b / 8 gives every XCD a contiguous eighth of the tiles (whole
    // graphs: the gather's re-reads of neighbour rows stay in that XCD's L2)
    const int wg = (blockIdx.x & 7) * (GF_GRID / 8) + (blockIdx.x >> 3);
    const int n_tiles = (a.n_rows + GF_TILE - 1) / GF_TILE;
    const int tq = n_tiles / GF_GRID, tr = n_tiles % GF_GRID;
    const int t_beg = wg * tq + min(wg, tr), t_cnt = tq + (wg < tr ? 1 : 0);
    // Panel addresses as (per-lane base) + (compile-time offset): the swizzle term (row & 11) depends on the lane only --
    //   fragment fetch   row 16 tt + l15, column quad 16 c + (4 kq + ii):  base a_off[ii], + tt * 4096 + c * 64 floats
    //   result store     row 16 tt + 4 kq + r, column 16 wave + l15:       base d_off[r],  + tt * 4096
    //   whole rows       row 4 wave + i, column quad = lane:               base r_off[i]
    int a_off[4], d_off[4], r_off[GF_RPW];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) a_off[ii] = gf_off(l15, kq * 4 + ii);
#pragma unroll
    for (int r = 0; r < 4; ++r) d_off[r] = gf_off(4 * kq + r, (wave * 16 + l15) >> 2) + (l15 & 3);
#pragma unroll
    for (int i = 0; i < GF_RPW; ++i) r_off[i] = gf_off(wave * GF_RPW + i, lane);

    for (int pass = 0; pass < t_cnt; pass += GF_TMAX) {
        const int nt = min(GF_TMAX, t_cnt - pass);
        const int row0 = (t_beg + pass) * GF_TILE;
        const int row_end = min(a.n_rows, row0 + nt * GF_TILE);          // rows [row0, row_end) are this pass's
        // ------------------------------------------------------------ 1. gather: U = A_hat X rows -> panel
        if (wave * GF_RPW < nt * GF_TILE) {
            const int rbase = row0 + wave * GF_RPW;
            int rp = 0;
            if (lane <= GF_RPW) rp = a.rowptr[min(rbase + lane, a.n_rows)];
            int beg[GF_RPW], cnt[GF_RPW];
#pragma unroll
            for (int i = 0; i < GF_RPW; ++i) {
                beg[i] = __builtin_amdgcn_readlane(rp, i);
                cnt[i] = rbase + i < row_end ? __builtin_amdgcn_readlane(rp, i + 1) - beg[i] : 0;
            }
            // the first 16 (col, val) pairs of row i live in lanes 16 i .. 16 i + 15
            const int gi = lane >> 4, ge = lane & 15;
            const int my_beg = __shfl(rp, gi, 64);
            const int my_cnt = rbase + gi < row_end ? __shfl(rp, gi + 1, 64) - my_beg : 0;
            int c = 0;
            float v = 0.f;
            if (ge < my_cnt) {
                c = a.col[my_beg + ge];
                v = a.val[my_beg + ge];
            }
            int cmax = 0;
#pragma unroll
            for (int i = 0; i < GF_RPW; ++i) cmax = max(cmax, min(cnt[i], 16));
            f32x4v acc[GF_RPW];
#pragma unroll
            for (int i = 0; i < GF_RPW; ++i) acc[i] = f32x4v{0.f, 0.f, 0.f, 0.f};
            // four neighbours of each of the four rows per round trip; lanes past a row's end carry (col 0, val 0): an
            // unconditional load of a valid row times zero
            for (int jb = 0; jb < cmax; jb += 4) {
                f32x4v x[GF_RPW][4];
                float w[GF_RPW][4];
#pragma unroll
                for (int i = 0; i < GF_RPW; ++i)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int src = i * 16 + jb + u;
                        // (a wave-uniform source lane: v_readlane, the neighbour row's base address is scalar -- as __shfl each of these was a
                        //  ds_bpermute round trip ahead of the load it addresses)
                        const int cj = __builtin_amdgcn_readlane(c, src);
                        w[i][u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
                        x[i][u] = *reinterpret_cast<const f32x4v*>(a.X + (size_t)cj * FIRA_D + lane * 4);
                    }
#pragma unroll
                for (int i = 0; i < GF_RPW; ++i)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc[i].x = fmaf(w[i][u], x[i][u].x, acc[i].x); acc[i].y = fmaf(w[i][u], x[i][u].y, acc[i].y);
                        acc[i].z = fmaf(w[i][u], x[i][u].z, acc[i].z); acc[i].w = fmaf(w[i][u], x[i][u].w, acc[i].w);
                    }
            }
            // row sums of A_hat (the rank-1 term's row factor): over the 16-lane group, then the rare tail
            float vs = v;
            vs = sum16(vs);
            float vsum[GF_RPW];
#pragma unroll
            for (int i = 0; i < GF_RPW; ++i) vsum[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vs), i * 16));
            // rows with more than 16 entries (hub nodes): the remaining entries 64 at a time, as spmm_rowwave_kernel does
#pragma unroll
            for (int i = 0; i < GF_RPW; ++i) {
                if (cnt[i] <= 16) continue;                      // wave-uniform
                float extra = 0.f;
                for (int base = beg[i] + 16; base < beg[i] + cnt[i]; base += 64) {
                    const int n = min(64, beg[i] + cnt[i] - base);
                    int c2 = 0;
                    float v2 = 0.f;
                    if (lane < n) {
                        c2 = a.col[base + lane];
                        v2 = a.val[base + lane];
                    }
                    extra += v2;
                    for (int j = 0; j < n; j += 4) {             // (lanes past n hold col 0 / val 0)
                        f32x4v x4[4];
                        float w4[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int sl = min(j + u, 63);
                            const int cj = j + u < 64 ? __builtin_amdgcn_readlane(c2, sl) : 0;
                            w4[u] = j + u < 64 ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v2), sl)) : 0.f;
                            x4[u] = *reinterpret_cast<const f32x4v*>(a.X + (size_t)cj * FIRA_D + lane * 4);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            acc[i].x = fmaf(w4[u], x4[u].x, acc[i].x); acc[i].y = fmaf(w4[u], x4[u].y, acc[i].y);
                            acc[i].z = fmaf(w4[u], x4[u].z, acc[i].z); acc[i].w = fmaf(w4[u], x4[u].w, acc[i].w);
                        }
                    }
                }
                extra = wave_sum(extra);
                vsum[i] += extra;
            }
#pragma unroll
            for (int i = 0; i < GF_RPW; ++i) {
                const int lr = wave * GF_RPW + i;
                if constexpr (X3) {
                    gx_store_row4<X3 ? NP : 3>(sm_p, GX_PLANE, lr, lane, acc[i]);
                } else
                *reinterpret_cast<f32x4v*>(&sm_u[r_off[i]]) = acc[i];                  // (rows past the end: zeros)
                if (lane == 0) sm_rs[lr] = vsum[i];
                if (BWD && rbase + i < row_end && a.u_out)
                    *reinterpret_cast<f32x4v*>(a.u_out + (size_t)(rbase + i) * FIRA_D + lane * 4) = acc[i];
                if (!BWD && a.rowsum_out && lane == 0 && rbase + i < row_end) a.rowsum_out[rbase + i] = vsum[i];
            }
        }
        // (round 5) the first 64-wide k chunk of this wave's B fragments is requested BEFORE the barrier that closes the gather: a wave
        // that is done early has its weights on the way while the slowest wave still collects neighbours (with the gather's 64
        // row registers live the request cannot move further up: 128 VGPRs per lane at 1024 threads)
        const rsrc_t rW = buf_rsrc(a.W, X3 ? 3u * (unsigned)GX_WPLANE : FIRA_D * FIRA_D * 4u);
        const unsigned wlane = (unsigned)((kq * 16) * FIRA_D + wave * 16 + l15) * 4u;
        float b[2][16];
        // X3: the weight planes are stored in fragment order -- unit ((16-column block w) * 8 + k step s) * 64 + lane, 16 bytes
        // each: one contiguous 1 KiB per wave, plane and k step
        const unsigned xlane = gx_wlane(wave, lane);
        uint4 bx[2][3];
        if constexpr (X3) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                bx[0][pl] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rW, xlane, pl * (int)GX_WPLANE, 0));
        } else {
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2)
            b[0][s2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rW, wlane, s2 * FIRA_D * 4, 0));
        }
        asm volatile("" ::: "memory");
        lds_barrier();   

        // ------------------------------------------------------------ 2. product: panel [16 nt, 256] x Wk, 16 columns per wave
        f32x4acc acc[GF_TMAX];
#pragma unroll
        for (int tt = 0; tt < GF_TMAX; ++tt) acc[tt] = f32x4acc{0.f, 0.f, 0.f, 0.f};
        if constexpr (X3) {
            // fragment of tile tt, k step s: row 16 tt + l15, 16-byte column 4 s + kq  ->  byte (a_q ^ (s << 6)) + tt * 8192 of a plane
            const int a_q = gx_frag_base(l15, kq);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 1 < 8) {
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        bx[(ks + 1) & 1][pl] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                              rW, xlane, pl * (int)GX_WPLANE + (ks + 1) * 1024, 0));
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int tt = 0; tt < GF_TMAX; ++tt) {
                    if (tt < nt)                             // block-uniform
                        gx_terms<X3 ? NP : 3>(acc[tt], sm_p + ((a_q ^ (ks << 6)) + tt * (GF_TILE * 512)), GX_PLANE, bx[ks & 1]);
                }
                asm volatile("" ::: "memory");
            }
        } else {
            constexpr int NC = FIRA_D / 64;              // k chunks: lane (row l15, quarter kq) holds k = 64 c + 16 kq + s
            // (buffer loads: ONE per-lane byte offset in a VGPR, the k row as the scalar offset -- with flat pointers the
            //  compiler kept a 64-bit address pair per 4 KB window of the weight, hoisted all 64 of them and spilled)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (c + 1 < NC) {
#pragma unroll
                    for (int s2 = 0; s2 < 16; ++s2)
                        b[(c + 1) & 1][s2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                           rW, wlane, ((c + 1) * 64 + s2) * FIRA_D * 4, 0));
                }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int tt = 0; tt < GF_TMAX; ++tt) {
                    if (tt < nt) {                           // block-uniform
                        float af[16];
#pragma unroll
                        for (int ii = 0; ii < 4; ++ii) {
                            const f32x4v q = *reinterpret_cast<const f32x4v*>(&sm_u[a_off[ii] + tt * (GF_TILE * FIRA_D) + c * 64]);
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
        }
        lds_barrier();                                    // every wave has read its last A fragment: the panel can be overwritten
        // accumulator register r of tile tt: row 16 tt + 4 kq + r, column 16 wave + l15
#pragma unroll
        for (int tt = 0; tt < GF_TMAX; ++tt) {
            if (tt < nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sm_u[d_off[r] + tt * (GF_TILE * FIRA_D)] = acc[tt][r];
            }
        }
        lds_barrier();   

        // ------------------------------------------------------------ 3. whole rows
        const int rbase = row0 + wave * GF_RPW;
        if (rbase < row_end) {
            if constexpr (BWD) {
                f32x4v old[GF_RPW];
#pragma unroll
                for (int i = 0; i < GF_RPW; ++i) {
                    const int row = min(rbase + i, row_end - 1);
                    old[i] = *reinterpret_cast<const f32x4v*>(a.acc_out + (size_t)row * FIRA_D + lane * 4);
                }
#pragma unroll
                for (int i = 0; i < GF_RPW; ++i) {
                    if (rbase + i >= row_end) continue;
                    const f32x4v d = *reinterpret_cast<const f32x4v*>(&sm_u[r_off[i]]);
                    *reinterpret_cast<f32x4v*>(a.acc_out + (size_t)(rbase + i) * FIRA_D + lane * 4) = old[i] + d;
                }
            } else {
                const f32x4v bias4 = *reinterpret_cast<const f32x4v*>(a.bias + lane * 4);
                const f32x4v c4 = *reinterpret_cast<const f32x4v*>(a.r1_col + lane * 4);
                const f32x4v g4 = *reinterpret_cast<const f32x4v*>(a.gamma + lane * 4);
                const f32x4v be4 = *reinterpret_cast<const f32x4v*>(a.beta + lane * 4);
                f32x4v res[GF_RPW];
                int s2[GF_RPW];
#pragma unroll
                for (int i = 0; i < GF_RPW; ++i) {
                    const int row = min(rbase + i, row_end - 1);
                    res[i] = *reinterpret_cast<const f32x4v*>(a.res + (size_t)row * FIRA_D + lane * 4);
                    s2[i] = a.slot2 ? a.slot2[row] : -1;
                }
#pragma unroll
                for (int i = 0; i < GF_RPW; ++i) {
                    const int row = rbase + i;
                    if (row >= row_end) continue;                    // wave-uniform
                    const int lr = wave * GF_RPW + i;
                    f32x4v x = *reinterpret_cast<const f32x4v*>(&sm_u[r_off[i]]);
                    x = x + bias4;                                   // the product's bias, then the rank-1 term (add_layernorm_fwd's order)
                    const float w = sm_rs[lr];
                    x.x = fmaf(w, c4.x, x.x); x.y = fmaf(w, c4.y, x.y); x.z = fmaf(w, c4.z, x.z); x.w = fmaf(w, c4.w, x.w);
                    if (a.p > 0.f) {
                        const uint32_t e0 = (uint32_t)row * FIRA_D + lane * 4;
                        x.x *= dropout_scale(a.seed, a.site, e0 + 0, a.p, a.inv_keep);
                        x.y *= dropout_scale(a.seed, a.site, e0 + 1, a.p, a.inv_keep);
                        x.z *= dropout_scale(a.seed, a.site, e0 + 2, a.p, a.inv_keep);
                        x.w *= dropout_scale(a.seed, a.site, e0 + 3, a.p, a.inv_keep);
                    }
                    x = x + res[i];
                    const float mean = wave_sum(x.x + x.y + x.z + x.w) * (1.0f / FIRA_D);
                    const f32x4v d = x - mean;
                    const float var = wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w) * (1.0f / FIRA_D);
                    const float rstd = 1.0f / sqrtf(var + 1e-5f);
                    const size_t o = (size_t)row * FIRA_D + lane * 4;
                    __builtin_nontemporal_store(x, reinterpret_cast<f32x4v*>(a.sum + o));     // (read by the backward pass only)
                    const f32x4v out = {d.x * rstd * g4.x + be4.x, d.y * rstd * g4.y + be4.y, d.z * rstd * g4.z + be4.z,
                                        d.w * rstd * g4.w + be4.w};
                    *reinterpret_cast<f32x4v*>(a.y + o) = out;
                    if (s2[i] >= 0) *reinterpret_cast<f32x4v*>(a.y2 + (size_t)s2[i] * FIRA_D + lane * 4) = out;
                    if (lane == 0) {
                        a.stats[2 * row] = mean;
                        a.stats[2 * row + 1] = rstd;
                    }
                }
            }
        }
        if (pass + GF_TMAX < t_cnt) lds_barrier();         // the next pass overwrites the panel
    }
}

// Algorithmic bytes of one launch (bench.py adds the batch's 8 * nnz for col / val itself, as for csr_spmm):
//   forward   rowptr + gathered rows in + pre-norm rows out + normalised rows out      (the residual row is a gathered row)
//   backward  rowptr + gradient rows in + V rows out + accumulated rows in and out
static double gcn_fused_bytes(int n_rows, bool bwd) {
    return 4.0 * (n_rows + 1) + (bwd ? 4.0 : 3.0) * n_rows * FIRA_D * 4.0;
}
template <typename K>
static int gcn_fused_lds(K kernel, size_t bytes = GF_LDS) {            // once per kernel: allow the > 64 KB dynamic LDS request
    const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    return e == hipSuccess ? 0 : set_err("gcn_fused: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
}

int gcn_fused_fwd(hipStream_t s, int n_rows, const int32_t* rowptr, const int32_t* col, const float* val, const float* X,
                  const float* Wk, const float* bias, const float* r1_col, const float* gamma, const float* beta, float* sum,
                  float* y, float* stats, float* rowsum_out, const int32_t* slot2, float* y2, float dropout, uint64_t seed,
                  uint32_t site, int bf16, const uint16_t* Wx) {
    if (n_rows <= 0) return 0;
    FIRA_REQUIRE(rowptr && col && val && X && Wk && bias && r1_col && gamma && beta && sum && y && stats,
                 "gcn_fused_fwd: null pointer argument");
    FIRA_REQUIRE((uintptr_t)X % 16 == 0 && (uintptr_t)Wk % 16 == 0 && (uintptr_t)sum % 16 == 0 && (uintptr_t)y % 16 == 0,
                 "gcn_fused_fwd: rows must be 16-byte aligned");
    ProfScope prof(s, PROF_GCN, 2.0 * n_rows * FIRA_D * FIRA_D, gcn_fused_bytes(n_rows, false));
    GcnFusedArgs a{};
    a.n_rows = n_rows; a.rowptr = rowptr; a.col = col; a.val = val; a.X = X; a.W = Wk;
    a.bias = bias; a.r1_col = r1_col; a.res = X; a.gamma = gamma; a.beta = beta;
    a.sum = sum; a.y = y; a.stats = stats; a.rowsum_out = rowsum_out; a.slot2 = y2 ? slot2 : nullptr; a.y2 = y2;
    a.p = dropout; a.inv_keep = dropout > 0.f ? 1.0f / (1.0f - dropout) : 1.0f; a.seed = seed; a.site = site;
    static const int attr = gcn_fused_lds(gcn_fused_kernel<true, false>) | gcn_fused_lds(gcn_fused_kernel<false, false>) |
                            gcn_fused_lds(gcn_fused_kernel<false, false, 3>, GF_LDS_X3) |
                            gcn_fused_lds(gcn_fused_kernel<false, false, 1>, GF_LDS_X3);
    if (attr) return attr;
    if (Wx) {                            // the product on bf16 planes (Wx: the weight's planes, gcn_split_planes): three terms in
        a.W = reinterpret_cast<const float*>(Wx);                                    // fp32 mode, one in bf16 mode
        if (bf16) hipLaunchKernelGGL((gcn_fused_kernel<false, false, 1>), dim3(GF_GRID), dim3(GF_WAVES * 64), GF_LDS_X3, s, a);
        else hipLaunchKernelGGL((gcn_fused_kernel<false, false, 3>), dim3(GF_GRID), dim3(GF_WAVES * 64), GF_LDS_X3, s, a);
    } else
    if (bf16) hipLaunchKernelGGL((gcn_fused_kernel<true, false>), dim3(GF_GRID), dim3(GF_WAVES * 64), GF_LDS, s, a);
    else hipLaunchKernelGGL((gcn_fused_kernel<false, false>), dim3(GF_GRID), dim3(GF_WAVES * 64), GF_LDS, s, a);
    FIRA_CHECK_LAUNCH("gcn_fused_fwd");
    return 0;
}

int gcn_fused_bwd(hipStream_t s, int n_rows, const int32_t* rowptr, const int32_t* col, const float* val, const float* dY,
                  const float* Wk, float* u_out, float* acc_out, int bf16, const uint16_t* Wx) {
    if (n_rows <= 0) return 0;
    FIRA_REQUIRE(rowptr && col && val && dY && Wk && acc_out, "gcn_fused_bwd: null pointer argument");
    FIRA_REQUIRE((uintptr_t)dY % 16 == 0 && (uintptr_t)Wk % 16 == 0 && (uintptr_t)acc_out % 16 == 0 && (uintptr_t)u_out % 16 == 0,
                 "gcn_fused_bwd: rows must be 16-byte aligned");
    ProfScope prof(s, PROF_GCN, 2.0 * n_rows * FIRA_D * FIRA_D, gcn_fused_bytes(n_rows, true));
    GcnFusedArgs a{};
    a.n_rows = n_rows; a.rowptr = rowptr; a.col = col; a.val = val; a.X = dY; a.W = Wk;
    a.u_out = u_out; a.acc_out = acc_out;
    static const int attr = gcn_fused_lds(gcn_fused_kernel<true, true>) | gcn_fused_lds(gcn_fused_kernel<false, true>) |
                            gcn_fused_lds(gcn_fused_kernel<false, true, 3>, GF_LDS_X3) |
                            gcn_fused_lds(gcn_fused_kernel<false, true, 1>, GF_LDS_X3);
    if (attr) return attr;
    if (Wx) {
        a.W = reinterpret_cast<const float*>(Wx);
        if (bf16) hipLaunchKernelGGL((gcn_fused_kernel<false, true, 1>), dim3(GF_GRID), dim3(GF_WAVES * 64), GF_LDS_X3, s, a);
        else hipLaunchKernelGGL((gcn_fused_kernel<false, true, 3>), dim3(GF_GRID), dim3(GF_WAVES * 64), GF_LDS_X3, s, a);
    } else
    if (bf16) hipLaunchKernelGGL((gcn_fused_kernel<true, true>), dim3(GF_GRID), dim3(GF_WAVES * 64), GF_LDS, s, a);
    else hipLaunchKernelGGL((gcn_fused_kernel<false, true>), dim3(GF_GRID), dim3(GF_WAVES * 64), GF_LDS, s, a);
    FIRA_CHECK_LAUNCH("gcn_fused_bwd");
    return 0;
}

// The three bf16 planes of n [256 n][256 k] fp32 matrices in the fragment order the X3 product streams (see gcn_fused_kernel):
// plane pl of matrix m at dst[m] + pl * 65536 elements; unit ((n / 16) * 8 + k / 32) * 64 + ((k / 8) % 4) * 16 + n % 16 holds the
// eight values k .. k + 7 (k a multiple of 8) of row n.  A thread owns one unit: two float4 reads (a row's units are consecutive
// threads), three 16-byte stores.
struct SplitTable { int n = 0; const float* src[48]; uint16_t* dst[48]; };
__global__ __launch_bounds__(256) void gcn_split_planes_kernel(const SplitTable tab) {
    const int m = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;        // j < 8192: row n = j / 32, unit c = j % 32 of the row
    const int n = j >> 5, c = j & 31;
    const float* src = tab.src[m] + (size_t)n * FIRA_D + c * 8;
    f32x4v u0 = *reinterpret_cast<const f32x4v*>(src), u1 = *reinterpret_cast<const f32x4v*>(src + 4);
    const int unit = ((n >> 4) * 8 + (c >> 2)) * 64 + (c & 3) * 16 + (n & 15);
    uint16_t* dst = tab.dst[m] + (size_t)unit * 8;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const uint32_t p0 = gx_pack(u0.x, u0.y), p1 = gx_pack(u0.z, u0.w), p2 = gx_pack(u1.x, u1.y), p3 = gx_pack(u1.z, u1.w);
        *reinterpret_cast<uint4*>(dst + (size_t)pl * FIRA_D * FIRA_D) = uint4{p0, p1, p2, p3};
        if (pl < 2) {
            u0.x = gx_rest_lo(u0.x, p0); u0.y = gx_rest_hi(u0.y, p0); u0.z = gx_rest_lo(u0.z, p1); u0.w = gx_rest_hi(u0.w, p1);
            u1.x = gx_rest_lo(u1.x, p2); u1.y = gx_rest_hi(u1.y, p2); u1.z = gx_rest_lo(u1.z, p3); u1.w = gx_rest_hi(u1.w, p3);
        }
    }
}
// n matrices [256 n][256 k] (fp32, row-major) -> their planes (3 * 65536 bf16 each)
int gcn_split_planes(hipStream_t s, int n, const float* const* src, uint16_t* const* dst) {
    if (n <= 0) return 0;
    FIRA_REQUIRE(n <= 48, "gcn_split_planes: %d matrices", n);
    SplitTable tab;
    tab.n = n;
    for (int i = 0; i < n; ++i) { tab.src[i] = src[i]; tab.dst[i] = dst[i]; }
    hipLaunchKernelGGL(gcn_split_planes_kernel, dim3(FIRA_D * FIRA_D / 8 / 256, n), dim3(256), 0, s, tab);
    FIRA_CHECK_LAUNCH("gcn_split_planes");
    return 0;
}

// Wt[l] = W[l]^T for nl stacked [256,256] matrices (the forward launch computes U W21^T and wants the weight k-major, i.e.
// W21^T row-major; the backward launch computes V W21 and reads W21 itself); 64 x 64 tiles through LDS
__global__ __launch_bounds__(256) void transpose256_kernel(const float* __restrict__ W, float* __restrict__ Wt) {
    __shared__ float tile[64][65];
    const int l = blockIdx.z, r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const float* src = W + (size_t)l * FIRA_D * FIRA_D;
    float* dst = Wt + (size_t)l * FIRA_D * FIRA_D;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) tile[r][tx] = src[(size_t)(r0 + r) * FIRA_D + c0 + tx];
    __syncthreads();
    for (int r = ty; r < 64; r += 4) dst[(size_t)(c0 + r) * FIRA_D + r0 + tx] = tile[tx][r];
}
int transpose256(hipStream_t s, int nl, const float* W, float* Wt) {
    if (nl <= 0) return 0;
    hipLaunchKernelGGL(transpose256_kernel, dim3(FIRA_D / 64, FIRA_D / 64, nl), dim3(256), 0, s, W, Wt);
    FIRA_CHECK_LAUNCH("transpose256");
    return 0;
}

}  // namespace fira

extern "C" {
int fira_gcn_layer_fwd(void* stream, int n_rows, const int32_t* rowptr, const int32_t* col, const float* val, const float* X,
                       const float* W21t, const float* bias, const float* c21, const float* gamma, const float* beta,
                       float* sum, float* y, float* stats, float* rowsum_out, float dropout, uint64_t seed, uint32_t site,
                       int dtype) {
    FIRA_REQUIRE(dtype >= FIRA_F32 && dtype <= FIRA_BF16X1, "fira_gcn_layer_fwd: dtype must be FIRA_F32, FIRA_BF16, FIRA_F32X3 or FIRA_BF16X1");
    return fira::gcn_fused_fwd((hipStream_t)stream, n_rows, rowptr, col, val, X, W21t, bias, c21, gamma, beta, sum, y, stats,
                               rowsum_out, nullptr, nullptr, dropout, seed, site, dtype == FIRA_BF16 || dtype == FIRA_BF16X1,
                               dtype >= FIRA_F32X3 ? reinterpret_cast<const uint16_t*>(W21t) : nullptr);
}
int fira_gcn_layer_bwd(void* stream, int n_rows, const int32_t* rowptr, const int32_t* col, const float* val, const float* dY,
                       const float* W21, float* V, float* dX, int dtype) {
    FIRA_REQUIRE(dtype >= FIRA_F32 && dtype <= FIRA_BF16X1, "fira_gcn_layer_bwd: dtype must be FIRA_F32, FIRA_BF16, FIRA_F32X3 or FIRA_BF16X1");
    FIRA_REQUIRE(V != nullptr, "fira_gcn_layer_bwd: V (the weight gradient's operand) must be given");
    return fira::gcn_fused_bwd((hipStream_t)stream, n_rows, rowptr, col, val, dY, W21, V, dX, dtype == FIRA_BF16 || dtype == FIRA_BF16X1,
                               dtype >= FIRA_F32X3 ? reinterpret_cast<const uint16_t*>(W21) : nullptr);
}
int fira_gcn_weight_planes(void* stream, int n_mats, const float* B, uint16_t* planes) {
    FIRA_REQUIRE(B && planes && n_mats > 0 && n_mats <= 24, "fira_gcn_weight_planes: bad argument");
    FIRA_REQUIRE((uintptr_t)B % 16 == 0 && (uintptr_t)planes % 16 == 0, "fira_gcn_weight_planes: 16-byte aligned pointers");
    const float* src[24];
    uint16_t* dst[24];
    for (int i = 0; i < n_mats; ++i) { src[i] = B + (size_t)i * FIRA_D * FIRA_D; dst[i] = planes + (size_t)i * 3 * FIRA_D * FIRA_D; }
    return fira::gcn_split_planes((hipStream_t)stream, n_mats, src, dst);
}
}
