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
// A workgroup (8 waves) owns 32 COMPLETE node rows:
//   1. gather   wave w aggregates rows 4w .. 4w+3 (64 lanes x float4 = one 1 KiB neighbour row per load; row offsets and
//               the first 16 (col, val) pairs of all four rows are fetched in two batched round trips, the first four
//               neighbour rows of all four rows in a third) into a [32][256] fp32 panel in LDS (pitch 260: the
//               ds_read_b128 fragment fetch of phase 2 is conflict-free);
//   2. product  wave w owns output columns 32w .. 32w+31: A fragments (16 contiguous k per lane) from the panel, B
//               fragments straight from the L2-resident 256 KB k-major weight (whole 128-byte lines per instruction,
//               requested 2-4 chunks ahead), 8 chunks x 16 v_mfma_f32_32x32x2_f32 -- or 8 x 2 v_mfma_f32_32x32x16_bf16;
//   3. rows     the 32x256 result goes back through the panel, and wave w finishes rows 4w .. 4w+3 with whole-row
//               (1 KiB, coalesced) accesses: bias, rank-1 term, dropout, residual, LayerNorm, second compact copy
//               (forward) / accumulate into the gradient rows (backward).
// fp32: 32 x 256 x 256 is 6.8 us of one CU's MFMA time per workgroup (294 workgroups at batch 32: the launch is
// MFMA-bound, ~1.2 GFLOP); bf16: gather-bound.
#include "engine.h"
#include "mfma_frag.h"

namespace fira {

constexpr int GF_ROWS = 32;               // node rows per workgroup
constexpr int GF_PITCH = FIRA_D + 4;      // LDS row pitch in floats (rows 16 bytes apart in bank space)
constexpr int GF_WAVES = 8;
constexpr int GF_RPW = GF_ROWS / GF_WAVES;   // rows per wave in the gather / row phases

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

template <bool BF, bool BWD>
__global__ __launch_bounds__(GF_WAVES * 64) void gcn_fused_kernel(const GcnFusedArgs a) {
    __shared__ __attribute__((aligned(16))) float sm_u[GF_ROWS * GF_PITCH];
    __shared__ float sm_rs[GF_ROWS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    // XCD-aware row-block order (spmm.hip): workgroup b runs on XCD b % 8; every XCD owns a contiguous eighth of the rows
    const int per_xcd = gridDim.x >> 3;
    const int r0 = ((blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3)) * GF_ROWS;
    if (r0 >= a.n_rows) return;

    // ---------------------------------------------------------------- 1. gather: U = A_hat X rows -> LDS
    {
        const int rbase = r0 + wave * GF_RPW;
        int rp = 0;
        if (lane <= GF_RPW) rp = a.rowptr[min(rbase + lane, a.n_rows)];
        int beg[GF_RPW], cnt[GF_RPW];
#pragma unroll
        for (int i = 0; i < GF_RPW; ++i) {
            beg[i] = __shfl(rp, i, 64);
            cnt[i] = rbase + i < a.n_rows ? __shfl(rp, i + 1, 64) - beg[i] : 0;
        }
        // the first 16 (col, val) pairs of row i live in lanes 16 i .. 16 i + 15
        const int gi = lane >> 4, ge = lane & 15;
        const int my_beg = __shfl(rp, gi, 64);
        const int my_cnt = rbase + gi < a.n_rows ? __shfl(rp, gi + 1, 64) - my_beg : 0;
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
                    const int cj = __shfl(c, src, 64);
                    w[i][u] = __shfl(v, src, 64);
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
        vs += __shfl_xor(vs, 1, 64); vs += __shfl_xor(vs, 2, 64); vs += __shfl_xor(vs, 4, 64); vs += __shfl_xor(vs, 8, 64);
        float vsum[GF_RPW];
#pragma unroll
        for (int i = 0; i < GF_RPW; ++i) vsum[i] = __shfl(vs, i * 16, 64);
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
                        const int cj = j + u < 64 ? __shfl(c2, sl, 64) : 0;
                        w4[u] = j + u < 64 ? __shfl(v2, sl, 64) : 0.f;
                        x4[u] = *reinterpret_cast<const f32x4v*>(a.X + (size_t)cj * FIRA_D + lane * 4);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc[i].x = fmaf(w4[u], x4[u].x, acc[i].x); acc[i].y = fmaf(w4[u], x4[u].y, acc[i].y);
                        acc[i].z = fmaf(w4[u], x4[u].z, acc[i].z); acc[i].w = fmaf(w4[u], x4[u].w, acc[i].w);
                    }
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) extra += __shfl_xor(extra, o, 64);
            vsum[i] += extra;
        }
#pragma unroll
        for (int i = 0; i < GF_RPW; ++i) {
            const int lr = wave * GF_RPW + i;
            *reinterpret_cast<f32x4v*>(&sm_u[lr * GF_PITCH + lane * 4]) = acc[i];
            if (lane == 0) sm_rs[lr] = vsum[i];
            if (BWD && rbase + i < a.n_rows && a.u_out)
                *reinterpret_cast<f32x4v*>(a.u_out + (size_t)(rbase + i) * FIRA_D + lane * 4) = acc[i];
            if (!BWD && a.rowsum_out && lane == 0 && rbase + i < a.n_rows) a.rowsum_out[rbase + i] = vsum[i];
        }
    }
    __syncthreads();

    // ---------------------------------------------------------------- 2. product: panel [32,256] x Wk -> 32 x 32 per wave
    // B fragments come from the K-MAJOR weight Wk[k][n]: lane (l31, kh) holds Wk[c*32 + kh*16 + s][n0 + l31], s = 0..15 --
    // every load instruction reads two whole 128-byte lines (one per kh), each line exactly once per workgroup.  (A first
    // version read the n-major weight as 4 x 16 bytes per lane: 32 lines per instruction, each line touched by four
    // instructions -- with eight waves the 32 KB L1 thrashed and the launch took 38 us instead of 15.)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
        constexpr int NC = FIRA_D / 32;              // k chunks
        constexpr int PF = BF ? 4 : 3;               // B chunks in registers: PF - 1 requested ahead (bf16: the MFMAs hide nothing)
        const float* wp = a.W + (size_t)(kh * 16) * FIRA_D + wave * 32 + l31;
        const float* up = sm_u + l31 * GF_PITCH + kh * 16;
        float b[PF][16];
        // chunk c + PF - 1 is requested BEFORE the MFMA chain of chunk c; the empty asm statements pin that order (left alone,
        // the scheduler sinks every load to just above its first use to shorten live ranges, and each group of four MFMAs
        // then waits a full L2 round trip: 39 us per launch instead of 15)
#pragma unroll
        for (int c = 0; c < PF - 1; ++c)
#pragma unroll
            for (int s2 = 0; s2 < 16; ++s2) b[c][s2] = wp[(size_t)(c * 32 + s2) * FIRA_D];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c + PF - 1 < NC) {
#pragma unroll
                for (int s2 = 0; s2 < 16; ++s2) b[(c + PF - 1) % PF][s2] = wp[(size_t)((c + PF - 1) * 32 + s2) * FIRA_D];
            }
            asm volatile("" ::: "memory");
            float af[16];
            load_frag(af, up + c * 32, true);
            acc = chain16<BF>(af, b[c % PF], acc);
            asm volatile("" ::: "memory");
        }
    }
    __syncthreads();                                 // every wave has read its last A fragment: the panel can be overwritten
#pragma unroll
    for (int r = 0; r < 16; ++r) sm_u[acc_row(r, kh) * GF_PITCH + wave * 32 + l31] = acc[r];
    __syncthreads();

    // ---------------------------------------------------------------- 3. whole rows
    const int rbase = r0 + wave * GF_RPW;
    if constexpr (BWD) {
        f32x4v old[GF_RPW];
#pragma unroll
        for (int i = 0; i < GF_RPW; ++i) {
            const int row = min(rbase + i, a.n_rows - 1);
            old[i] = *reinterpret_cast<const f32x4v*>(a.acc_out + (size_t)row * FIRA_D + lane * 4);
        }
#pragma unroll
        for (int i = 0; i < GF_RPW; ++i) {
            if (rbase + i >= a.n_rows) continue;
            const f32x4v d = *reinterpret_cast<const f32x4v*>(&sm_u[(wave * GF_RPW + i) * GF_PITCH + lane * 4]);
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
            const int row = min(rbase + i, a.n_rows - 1);
            res[i] = *reinterpret_cast<const f32x4v*>(a.res + (size_t)row * FIRA_D + lane * 4);
            s2[i] = a.slot2 ? a.slot2[row] : -1;
        }
#pragma unroll
        for (int i = 0; i < GF_RPW; ++i) {
            const int row = rbase + i;
            if (row >= a.n_rows) continue;                   // wave-uniform
            const int lr = wave * GF_RPW + i;
            f32x4v x = *reinterpret_cast<const f32x4v*>(&sm_u[lr * GF_PITCH + lane * 4]);
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
            *reinterpret_cast<f32x4v*>(a.sum + o) = x;
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

// Algorithmic bytes of one launch (bench.py adds the batch's 8 * nnz for col / val itself, as for csr_spmm):
//   forward   rowptr + gathered rows in + pre-norm rows out + normalised rows out      (the residual row is a gathered row)
//   backward  rowptr + gradient rows in + V rows out + accumulated rows in and out
static double gcn_fused_bytes(int n_rows, bool bwd) {
    return 4.0 * (n_rows + 1) + (bwd ? 4.0 : 3.0) * n_rows * FIRA_D * 4.0;
}
static int gcn_fused_grid(int n_rows) { return cdiv(cdiv(n_rows, GF_ROWS), 8) * 8; }

int gcn_fused_fwd(hipStream_t s, int n_rows, const int32_t* rowptr, const int32_t* col, const float* val, const float* X,
                  const float* Wk, const float* bias, const float* r1_col, const float* gamma, const float* beta, float* sum,
                  float* y, float* stats, float* rowsum_out, const int32_t* slot2, float* y2, float dropout, uint64_t seed,
                  uint32_t site, int bf16) {
    if (n_rows <= 0) return 0;
    FIRA_REQUIRE(rowptr && col && val && X && Wk && bias && r1_col && gamma && beta && sum && y && stats,
                 "gcn_fused_fwd: null pointer argument");
    FIRA_REQUIRE((uintptr_t)X % 16 == 0 && (uintptr_t)Wk % 16 == 0 && (uintptr_t)sum % 16 == 0 && (uintptr_t)y % 16 == 0,
                 "gcn_fused_fwd: rows must be 16-byte aligned");
    ProfScope prof(s, PROF_SPMM, gcn_fused_bytes(n_rows, false));
    GcnFusedArgs a{};
    a.n_rows = n_rows; a.rowptr = rowptr; a.col = col; a.val = val; a.X = X; a.W = Wk;
    a.bias = bias; a.r1_col = r1_col; a.res = X; a.gamma = gamma; a.beta = beta;
    a.sum = sum; a.y = y; a.stats = stats; a.rowsum_out = rowsum_out; a.slot2 = y2 ? slot2 : nullptr; a.y2 = y2;
    a.p = dropout; a.inv_keep = dropout > 0.f ? 1.0f / (1.0f - dropout) : 1.0f; a.seed = seed; a.site = site;
    if (bf16) hipLaunchKernelGGL((gcn_fused_kernel<true, false>), dim3(gcn_fused_grid(n_rows)), dim3(GF_WAVES * 64), 0, s, a);
    else hipLaunchKernelGGL((gcn_fused_kernel<false, false>), dim3(gcn_fused_grid(n_rows)), dim3(GF_WAVES * 64), 0, s, a);
    FIRA_CHECK_LAUNCH("gcn_fused_fwd");
    return 0;
}

int gcn_fused_bwd(hipStream_t s, int n_rows, const int32_t* rowptr, const int32_t* col, const float* val, const float* dY,
                  const float* Wk, float* u_out, float* acc_out, int bf16) {
    if (n_rows <= 0) return 0;
    FIRA_REQUIRE(rowptr && col && val && dY && Wk && acc_out, "gcn_fused_bwd: null pointer argument");
    FIRA_REQUIRE((uintptr_t)dY % 16 == 0 && (uintptr_t)Wk % 16 == 0 && (uintptr_t)acc_out % 16 == 0 && (uintptr_t)u_out % 16 == 0,
                 "gcn_fused_bwd: rows must be 16-byte aligned");
    ProfScope prof(s, PROF_SPMM, gcn_fused_bytes(n_rows, true));
    GcnFusedArgs a{};
    a.n_rows = n_rows; a.rowptr = rowptr; a.col = col; a.val = val; a.X = dY; a.W = Wk;
    a.u_out = u_out; a.acc_out = acc_out;
    if (bf16) hipLaunchKernelGGL((gcn_fused_kernel<true, true>), dim3(gcn_fused_grid(n_rows)), dim3(GF_WAVES * 64), 0, s, a);
    else hipLaunchKernelGGL((gcn_fused_kernel<false, true>), dim3(gcn_fused_grid(n_rows)), dim3(GF_WAVES * 64), 0, s, a);
    FIRA_CHECK_LAUNCH("gcn_fused_bwd");
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
    FIRA_REQUIRE(dtype == FIRA_F32 || dtype == FIRA_BF16, "fira_gcn_layer_fwd: dtype must be FIRA_F32 or FIRA_BF16");
    return fira::gcn_fused_fwd((hipStream_t)stream, n_rows, rowptr, col, val, X, W21t, bias, c21, gamma, beta, sum, y, stats,
                               rowsum_out, nullptr, nullptr, dropout, seed, site, dtype == FIRA_BF16);
}
int fira_gcn_layer_bwd(void* stream, int n_rows, const int32_t* rowptr, const int32_t* col, const float* val, const float* dY,
                       const float* W21, float* V, float* dX, int dtype) {
    FIRA_REQUIRE(dtype == FIRA_F32 || dtype == FIRA_BF16, "fira_gcn_layer_bwd: dtype must be FIRA_F32 or FIRA_BF16");
    FIRA_REQUIRE(V != nullptr, "fira_gcn_layer_bwd: V (the weight gradient's operand) must be given");
    return fira::gcn_fused_bwd((hipStream_t)stream, n_rows, rowptr, col, val, dY, W21, V, dX, dtype == FIRA_BF16);
}
}
