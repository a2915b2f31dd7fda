// Row-wise (256-wide) memory-bound kernels: embedding gather/scatter, the element-wise
// Combination gate, fused dropout + residual + LayerNorm, bias-gradient column sums.
// One wavefront owns one 256-float row as float4 per lane (a 1 KiB coalesced access);
// row statistics are wave reductions (no LDS, no atomics on the forward path).
#include "engine.h"
#include "adam_rows.h"
#include "epilogue.h"
#include <algorithm>
#include <stdlib.h>

namespace fira {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// Embedding gathers of Encoder.forward / Decoder.forward (reference gnn_transformer.py:46-52,110-113),
// written straight into the [B, N, 256] node buffer (replaces the torch.cat of gnn_transformer.py:58).
__global__ __launch_bounds__(256) void embed_gather_fwd_kernel(int rows, int L, const int32_t* __restrict__ idx,
                                                               const float* __restrict__ table,
                                                               const float* __restrict__ pos, float* __restrict__ out,
                                                               int out_bstride, int out_off) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int b = r / L, i = r - b * L;
    const int id = idx[r];
    float4 v = *reinterpret_cast<const float4*>(table + (size_t)id * FIRA_D + lane * 4);
    if (pos) {
        const float4 p = *reinterpret_cast<const float4*>(pos + (size_t)i * FIRA_D + lane * 4);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    *reinterpret_cast<float4*>(out + ((size_t)b * out_bstride + out_off + i) * FIRA_D + lane * 4) = v;
}

// backward of the gather: dtable[idx] += dout[row] (embedding_dense_backward of SURVEY.md §2.3);
// rows whose id is the padding index get no gradient (nn.Embedding(padding_idx=0)).
__global__ __launch_bounds__(256) void embed_gather_bwd_kernel(int rows, int L, const int32_t* __restrict__ idx,
                                                               float* __restrict__ dtable,
                                                               const float* __restrict__ dout, int out_bstride,
                                                               int out_off, int padding_idx) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int b = r / L, i = r - b * L;
    const int id = idx[r];
    if (id == padding_idx) return;
    const float4 g = *reinterpret_cast<const float4*>(dout + ((size_t)b * out_bstride + out_off + i) * FIRA_D + lane * 4);
    float* p = dtable + (size_t)id * FIRA_D + lane * 4;
    unsafeAtomicAdd(p + 0, g.x);
    unsafeAtomicAdd(p + 1, g.y);
    unsafeAtomicAdd(p + 2, g.z);
    unsafeAtomicAdd(p + 3, g.w);
}

// The decoder's token embedding on the computed target rows (fira_batch.dec_off): compact row r is flat position
// row_bt[r] = b*T + t;  out[r] = table[idx[b*T+t]] + pos[t]  and its backward (padding_idx semantics as above).
__global__ __launch_bounds__(256) void embed_rows_fwd_kernel(int R, int T, const int32_t* __restrict__ row_bt,
                                                             const int32_t* __restrict__ idx,
                                                             const float* __restrict__ table,
                                                             const float* __restrict__ pos, float* __restrict__ out,
                                                             AdamRowsView vw) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int d = row_bt[r], t = d % T;
    float4 v = adam_rows_load(table, idx[d], lane, vw);
    const float4 p = *reinterpret_cast<const float4*>(pos + (size_t)t * FIRA_D + lane * 4);
    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    *reinterpret_cast<float4*>(out + (size_t)r * FIRA_D + lane * 4) = v;
}
__global__ __launch_bounds__(256) void embed_rows_bwd_kernel(int R, const int32_t* __restrict__ row_bt,
                                                             const int32_t* __restrict__ idx, float* __restrict__ dtable,
                                                             const float* __restrict__ dout, int padding_idx) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int id = idx[row_bt[r]];
    if (id == padding_idx) return;
    const float4 g = *reinterpret_cast<const float4*>(dout + (size_t)r * FIRA_D + lane * 4);
    float* p = dtable + (size_t)id * FIRA_D + lane * 4;
    unsafeAtomicAdd(p + 0, g.x);
    unsafeAtomicAdd(p + 1, g.y);
    unsafeAtomicAdd(p + 2, g.z);
    unsafeAtomicAdd(p + 3, g.w);
}

// Grouped form: the positions were grouped by table row on the host (fira_batch.emb_*).  One wave per item sums its
// <= 32 gradient rows in registers and adds the result to the table row once: a token that occurs 1000 times in the
// batch costs 32 same-address atomics per element instead of 1000 serialised ones.
__global__ __launch_bounds__(256) void embed_grouped_bwd_kernel(int n_items, const int32_t* __restrict__ item_tok,
                                                                const int32_t* __restrict__ item_ptr,
                                                                const int32_t* __restrict__ rows,
                                                                float* __restrict__ dtable,
                                                                const float* __restrict__ dnode) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int beg = item_ptr[item], cnt = min(item_ptr[item + 1] - beg, 64);
    const int mine = lane < cnt ? rows[beg + lane] : 0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < cnt; k += 4) {
        float4 g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = __shfl(mine, min(k + u, cnt - 1), 64);
            g[u] = *reinterpret_cast<const float4*>(dnode + (size_t)r * FIRA_D + lane * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k + u < cnt) { acc.x += g[u].x; acc.y += g[u].y; acc.z += g[u].z; acc.w += g[u].w; }
    }
    // items are grouped by word id (model.embedding_items / fira_host_node_lists: a word with more than 32 rows is cut into
    // ADJACENT items): a word with a single item is the only writer of its gradient row in this launch -- one 16-byte
    // read-modify-write instead of four float atomics per lane (768 k atomics per launch at batch 32 otherwise)
    const int tok = item_tok[item];
    const bool shared = (item > 0 && item_tok[item - 1] == tok) || (item + 1 < n_items && item_tok[item + 1] == tok);
    float* p = dtable + (size_t)tok * FIRA_D + lane * 4;
    if (!shared) {
        float4 cur = *reinterpret_cast<float4*>(p);
        cur.x += acc.x; cur.y += acc.y; cur.z += acc.z; cur.w += acc.w;
        *reinterpret_cast<float4*>(p) = cur;
        return;
    }
    unsafeAtomicAdd(p + 0, acc.x);
    unsafeAtomicAdd(p + 1, acc.y);
    unsafeAtomicAdd(p + 2, acc.z);
    unsafeAtomicAdd(p + 3, acc.w);
}

// Same for a SMALL table (the 71-row AST/edit-operation vocabulary): thousands of rows hit the same few table rows, so
// global atomics serialise; each workgroup first reduces its slice of rows into an LDS copy of the table.
__global__ __launch_bounds__(256) void embed_gather_bwd_small_kernel(int rows, int L, const int32_t* __restrict__ idx,
                                                                     float* __restrict__ dtable,
                                                                     const float* __restrict__ dout, int out_bstride,
                                                                     int out_off, int padding_idx, int table_rows,
                                                                     int rows_per_block) {
    extern __shared__ float tab[];                       // [table_rows][256]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < table_rows * FIRA_D; i += 256) tab[i] = 0.f;
    __syncthreads();
    const int r_beg = blockIdx.x * rows_per_block, r_end = min(rows, r_beg + rows_per_block);
    for (int r = r_beg + wave; r < r_end; r += 4) {
        const int id = idx[r];
        if (id == padding_idx) continue;
        const int b = r / L, i = r - b * L;
        const float4 g = *reinterpret_cast<const float4*>(dout + ((size_t)b * out_bstride + out_off + i) * FIRA_D + lane * 4);
        float* p = tab + (size_t)id * FIRA_D + lane * 4;
        atomicAdd(p + 0, g.x); atomicAdd(p + 1, g.y); atomicAdd(p + 2, g.z); atomicAdd(p + 3, g.w);
    }
    __syncthreads();
    for (int i = t; i < table_rows * FIRA_D; i += 256) {
        const float v = tab[i];
        if (v != 0.f) unsafeAtomicAdd(&dtable[i], v);
    }
}

// List form of the small-table scatter: (compact node row, table id) pairs instead of the dense [B, L] id array.
// Round 5: one workgroup per (table row j, slice of ELS_SLICE list items).  Its four waves scan the slice's ids with one
// coalesced load each, and gather ONLY the rows whose id is j (up to eight row loads in flight per wave) into register sums:
// one atomic per column and workgroup, none when the slice holds no such row.  (The round-1 kernel kept an LDS copy of the
// whole table per workgroup: 72 KB zeroed, filled through LDS atomics and flushed with up to 18 000 global atomics per
// workgroup -- 29 us on the tail of every step for ~5 MB of gradient rows.)
constexpr int ELS_SLICE = 256;
__global__ __launch_bounds__(256) void embed_list_bwd_small_kernel(int n, const int32_t* __restrict__ rows,
                                                                   const int32_t* __restrict__ ids,
                                                                   float* __restrict__ dtable,
                                                                   const float* __restrict__ dnode) {
    __shared__ __attribute__((aligned(16))) float red[4][FIRA_D];
    __shared__ int any;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int j = blockIdx.y;
    const int k = blockIdx.x * ELS_SLICE + t;
    if (t == 0) any = 0;
    int id = -1, row = 0;
    if (k < n) { id = ids[k]; row = rows[k]; }
    unsigned long long m = __ballot(id == j);
    __syncthreads();
    if (m != 0ull && lane == 0) any = 1;                   // (benign race: every writer stores 1)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    while (m) {                                            // wave-uniform: the matching lanes, eight at a time
        f32x4 g[8];
        bool ok[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            ok[u] = m != 0ull;
            const int src = ok[u] ? (int)__builtin_ctzll(m) : 0;
            if (ok[u]) m &= m - 1;
            const int r = __builtin_amdgcn_readlane(row, src);
            g[u] = *reinterpret_cast<const f32x4*>(dnode + (size_t)r * FIRA_D + lane * 4);    // (not ok: some listed row, unused)
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (ok[u]) acc += g[u];
    }
    *reinterpret_cast<f32x4*>(&red[wave][lane * 4]) = acc;
    __syncthreads();
    if (!any) return;
    const float v = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
    if (v != 0.f) unsafeAtomicAdd(&dtable[(size_t)j * FIRA_D + t], v);
}


__global__ __launch_bounds__(256) void combination_fwd_kernel(int M, const float* __restrict__ qk,
                                                              const float* __restrict__ vtab, int ldv,
                                                              const int32_t* __restrict__ mark,
                                                              float* __restrict__ out, float p, float inv_keep,
                                                              uint64_t seed, uint32_t site) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    const float4 q4 = *reinterpret_cast<const float4*>(qk + (size_t)r * 2 * FIRA_D + lane * 4);
    const float4 k4 = *reinterpret_cast<const float4*>(qk + (size_t)r * 2 * FIRA_D + FIRA_D + lane * 4);
    const float4 v4 = *reinterpret_cast<const float4*>(vtab + (size_t)mark[r] * ldv + lane * 4);
    const float q[4] = {q4.x, q4.y, q4.z, q4.w}, k[4] = {k4.x, k4.y, k4.z, k4.w}, v[4] = {v4.x, v4.y, v4.z, v4.w};
    float c[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float g0, g1;
        gate_elem(q[e], k[e], v[e], g0, g1);
        c[e] = g0 * k[e] + g1 * v[e];
        if (p > 0.f) c[e] *= dropout_scale(seed, site, (uint32_t)r * FIRA_D + lane * 4 + e, p, inv_keep);
    }
    *reinterpret_cast<float4*>(out + (size_t)r * FIRA_D + lane * 4) = make_float4(c[0], c[1], c[2], c[3]);
}

// backward: dq, dk per element; dv is reduced over all rows sharing a mark value into dvtab[4,256].
// A wave owns COMB_RW consecutive rows and requests every operand of all of them at once; the 4-row value table sits in
// registers (no load waits for a mark), so a workgroup is ONE memory round trip long (the row-at-a-time loop with the
// mark -> value-row dependency was eight: 13-18 us per launch for 3 500 rows).  The four waves park their column sums side
// by side in LDS (plain stores) and the workgroup adds them up: no zero fill, no LDS atomics.
constexpr int COMB_RW = 4;
__global__ __launch_bounds__(256) void combination_bwd_kernel(int M, const float* __restrict__ qk,
                                                              const float* __restrict__ vtab, int ldv,
                                                              const int32_t* __restrict__ mark,
                                                              const float* __restrict__ dout,
                                                              float* __restrict__ dqk, float* __restrict__ dvtab,
                                                              int lddv, float p, float inv_keep, uint64_t seed,
                                                              uint32_t site, int rows_per_block,
                                                              float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float red[4][4 * FIRA_D];          // [wave][mark value][column]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float vt[4][4];                    // value table: [mark value][element]
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float4 v4 = *reinterpret_cast<const float4*>(vtab + (size_t)a * ldv + lane * 4);
        vt[a][0] = v4.x; vt[a][1] = v4.y; vt[a][2] = v4.z; vt[a][3] = v4.w;
    }
    float dv_acc[4][4];                // [mark value][element]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) dv_acc[a][e] = 0.f;
    const float is = 1.0f / 5.656854249492381f;
    const int r_beg = blockIdx.x * rows_per_block, r_end = min(M, r_beg + rows_per_block);
    for (int r0 = r_beg + wave * COMB_RW; r0 < r_end; r0 += 4 * COMB_RW) {
        float4 q4[COMB_RW], k4[COMB_RW], d4[COMB_RW];
        int mk[COMB_RW];
#pragma unroll
        for (int u = 0; u < COMB_RW; ++u) {                  // rows past the end: a real row is read, nothing is stored
            const int r = min(r0 + u, r_end - 1);
            q4[u] = *reinterpret_cast<const float4*>(qk + (size_t)r * 2 * FIRA_D + lane * 4);
            k4[u] = *reinterpret_cast<const float4*>(qk + (size_t)r * 2 * FIRA_D + FIRA_D + lane * 4);
            d4[u] = *reinterpret_cast<const float4*>(dout + (size_t)r * FIRA_D + lane * 4);
            mk[u] = mark[r];
        }
        // one row per wave: the mark is wave-uniform.  Its value row and its dv accumulator are picked ARITHMETICALLY with
        // four scalar 0/1 weights (exact: one weight is 1, the others 0) -- as `mk == a ? .. : ..` chains on a vector register
        // the compiler built a tree of 170 branches per four rows.
        float w[COMB_RW][4];
#pragma unroll
        for (int u = 0; u < COMB_RW; ++u) {
            const int m = __builtin_amdgcn_readfirstlane(mk[u]);
#pragma unroll
            for (int a = 0; a < 4; ++a) w[u][a] = m == a ? 1.0f : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < COMB_RW; ++u) {
            const int r = r0 + u;
            if (r >= r_end) break;                           // wave-uniform
            const float q[4] = {q4[u].x, q4[u].y, q4[u].z, q4[u].w}, k[4] = {k4[u].x, k4[u].y, k4[u].z, k4[u].w};
            float dc[4] = {d4[u].x, d4[u].y, d4[u].z, d4[u].w};
            float dq[4], dk[4], dv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = fmaf(w[u][0], vt[0][e], fmaf(w[u][1], vt[1][e], fmaf(w[u][2], vt[2][e], w[u][3] * vt[3][e])));
                if (p > 0.f) dc[e] *= dropout_scale(seed, site, (uint32_t)r * FIRA_D + lane * 4 + e, p, inv_keep);
                float g0, g1;
                gate_elem(q[e], k[e], v, g0, g1);
                const float dg0 = dc[e] * k[e], dg1 = dc[e] * v;
                const float dot = g0 * dg0 + g1 * dg1;
                const float da = g0 * (dg0 - dot), db = g1 * (dg1 - dot);
                dq[e] = (da * k[e] + db * v) * is;
                dk[e] = dc[e] * g0 + da * q[e] * is;
                dv[e] = dc[e] * g1 + db * q[e] * is;
            }
            *reinterpret_cast<float4*>(dqk + (size_t)r * 2 * FIRA_D + lane * 4) = make_float4(dq[0], dq[1], dq[2], dq[3]);
            *reinterpret_cast<float4*>(dqk + (size_t)r * 2 * FIRA_D + FIRA_D + lane * 4) = make_float4(dk[0], dk[1], dk[2], dk[3]);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int e = 0; e < 4; ++e) dv_acc[a][e] = fmaf(w[u][a], dv[e], dv_acc[a][e]);
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
        *reinterpret_cast<float4*>(&red[wave][a * FIRA_D + lane * 4]) =
            make_float4(dv_acc[a][0], dv_acc[a][1], dv_acc[a][2], dv_acc[a][3]);
    __syncthreads();
    for (int i = t; i < 4 * FIRA_D; i += 256) {
        const float x = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        if (part)                                 // deferred reduction: see add_layernorm_bwd_kernel
            part[(size_t)blockIdx.x * 4 * FIRA_D + i] = x;
        else if (x != 0.f)
            unsafeAtomicAdd(&dvtab[(size_t)(i / FIRA_D) * lddv + (i % FIRA_D)], x);
    }
}

// ------------------------------------------------------------------------------------------------
// y = LayerNorm(dropout(x) + res) (eps = 1e-5, biased variance): the post-LN residual of every block
// (reference gnn_transformer.py:86,161,174,205).  x is overwritten with the pre-norm sum.
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(int M, float* __restrict__ x,
                                                                const float* __restrict__ res,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                float* __restrict__ y, float* __restrict__ stats,
                                                                float p, float inv_keep, uint64_t seed,
                                                                uint32_t site, const int32_t* __restrict__ y_rows,
                                                                const float* __restrict__ r1_row,
                                                                const float* __restrict__ r1_col,
                                                                const int32_t* __restrict__ slot2,
                                                                float* __restrict__ y2, uint32_t idx0) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    const int s2 = slot2 ? slot2[r] : -1;              // rows with a slot are stored a second time, compactly, at y2[slot]
    const size_t o = (size_t)r * FIRA_D + lane * 4;
    const size_t oy = y_rows ? (size_t)y_rows[r] * FIRA_D + lane * 4 : o;      // optional scatter of the output rows
    float4 a = *reinterpret_cast<const float4*>(x + o);
    if (r1_row) {                                    // rank-1 part of the branch output: x += r1_row[r] * r1_col[:]
        const float w = r1_row[r];
        const float4 c4 = *reinterpret_cast<const float4*>(r1_col + lane * 4);
        a.x = fmaf(w, c4.x, a.x); a.y = fmaf(w, c4.y, a.y); a.z = fmaf(w, c4.z, a.z); a.w = fmaf(w, c4.w, a.w);
    }
    if (p > 0.f) {
        const uint32_t e0 = idx0 + (uint32_t)r * FIRA_D + lane * 4;
        a.x *= dropout_scale(seed, site, e0 + 0, p, inv_keep);
        a.y *= dropout_scale(seed, site, e0 + 1, p, inv_keep);
        a.z *= dropout_scale(seed, site, e0 + 2, p, inv_keep);
        a.w *= dropout_scale(seed, site, e0 + 3, p, inv_keep);
    }
    if (res) {
        const float4 b = *reinterpret_cast<const float4*>(res + o);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const float mean = wave_sum(a.x + a.y + a.z + a.w) * (1.0f / FIRA_D);
    const float dx = a.x - mean, dy = a.y - mean, dz = a.z - mean, dw = a.w - mean;
    const float var = wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / FIRA_D);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float4 g = *reinterpret_cast<const float4*>(gamma + lane * 4);
    const float4 bt = *reinterpret_cast<const float4*>(beta + lane * 4);
    *reinterpret_cast<float4*>(x + o) = a;
    const float4 out = make_float4(dx * rstd * g.x + bt.x, dy * rstd * g.y + bt.y, dz * rstd * g.z + bt.z, dw * rstd * g.w + bt.w);
    *reinterpret_cast<float4*>(y + oy) = out;
    if (s2 >= 0) *reinterpret_cast<float4*>(y2 + (size_t)s2 * FIRA_D + lane * 4) = out;
    if (stats && lane == 0) {
        stats[2 * r] = mean;
        stats[2 * r + 1] = rstd;
    }
}

// backward of the block above.  ds = gradient wrt the pre-norm sum (also the residual-branch gradient);
// dx_drop (optional) = ds * keep-mask / (1-p): the gradient wrt the un-dropped GEMM output.
// A wave takes 4 consecutive rows per step and requests everything it reads for them (dy, the saved pre-norm rows, the
// statistics) before the first reduction, then streams the stores: one memory round trip per step.  All accesses go
// through buffer descriptors (rows past M read zeros and drop their stores), so there is no branch around a memory
// instruction -- with the per-row `if`s and the optional row map read inside the loop, every store was followed by an
// s_waitcnt vmcnt(0) and each 2-row iteration cost two dependent round trips (see epilogue.h).
// EXTRA (engine, GCN blocks): two more column sums of the un-dropped gradient rows dx -- sum_r dx[r,:] (the folded
// product's bias gradient) and sum_r row_w[r] * dx[r,:] (the gradient of c = W2 b1, which reaches the output through the
// rank-1 term (A_hat 1) c^T) -- so that no separate column-sum launch reads dx again.
constexpr int LNB_ROWS = 16;                    // rows per workgroup and step (4 waves x 4 rows)
template <bool EXTRA>
__global__ __launch_bounds__(256) void add_layernorm_bwd_kernel(int M, const float* dy,   // may alias ds (row-mapped, in place)
                                                                const float* __restrict__ sum,
                                                                const float* __restrict__ stats,
                                                                const float* __restrict__ gamma,
                                                                float* ds, float* __restrict__ dx_drop,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                float p, float inv_keep, uint64_t seed, uint32_t site,
                                                                int steps, const int32_t* __restrict__ rows,
                                                                float* __restrict__ part,
                                                                const float* __restrict__ row_w, uint32_t idx0) {
    constexpr int NV = EXTRA ? 4 : 2;               // column-sum vectors per workgroup: dgamma | dbeta (| sum dx | sum w dx)
    __shared__ __attribute__((aligned(16))) float red[4][NV * FIRA_D];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + lane * 4);
    f32x4 dg = {0.f, 0.f, 0.f, 0.f}, db = {0.f, 0.f, 0.f, 0.f}, dsum = {0.f, 0.f, 0.f, 0.f}, dwsum = {0.f, 0.f, 0.f, 0.f};
    const rsrc_t rW = buf_rsrc(EXTRA ? (const void*)row_w : (const void*)sum, EXTRA ? (unsigned)M * 4u : 0u);
    const rsrc_t rDy = buf_rsrc(dy, 0x7fffffffu), rDs = buf_rsrc(ds, 0x7fffffffu);
    const rsrc_t rSum = buf_rsrc(sum, (unsigned)M * FIRA_D * 4u), rStats = buf_rsrc(stats, (unsigned)M * 8u);
    const rsrc_t rMap = buf_rsrc(rows ? (const void*)rows : (const void*)sum, rows ? (unsigned)M * 4u : 0u);
    const rsrc_t rDx = buf_rsrc(dx_drop ? (const void*)dx_drop : (const void*)sum, dx_drop ? (unsigned)M * FIRA_D * 4u : 0u);
    const unsigned lo = (unsigned)lane * 16u;
    for (int st = 0; st < steps; ++st) {
        const int r0 = (blockIdx.x * steps + st) * LNB_ROWS + wave * 4;
        unsigned om[4];
        f32x4 d[4], sv[4];
        float mean[4], rstd[4], rw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                       // dy / ds live in the mapped rows
            const unsigned mapped = __builtin_amdgcn_raw_buffer_load_b32(rMap, (unsigned)(r0 + u) * 4u, 0, 0);
            om[u] = rows ? mapped : (unsigned)(r0 + u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u;
            const unsigned o = (unsigned)r * (FIRA_D * 4u) + lo;
            om[u] = r < M ? om[u] * (FIRA_D * 4u) + lo : FIRA_OOB;
            d[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rDy, om[u], 0, 0));
            sv[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rSum, o, 0, 0));
            mean[u] = buf_load_f32(rStats, (unsigned)r * 8u);
            rstd[u] = buf_load_f32(rStats, (unsigned)r * 8u + 4u);
            rw[u] = buf_load_f32(rW, (unsigned)r * 4u);     // (no EXTRA: zero records, reads 0)
        }
        f32x4 xh[4], h[4];
        float m1[4], m2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                       // rows past M: d = 0, rstd = 0 -> no contribution anywhere
            xh[u] = (sv[u] - mean[u]) * rstd[u];
            dg += d[u] * xh[u];
            db += d[u];
            h[u] = d[u] * g;
            m1[u] = h[u].x + h[u].y + h[u].z + h[u].w;
            const f32x4 hx = h[u] * xh[u];
            m2[u] = hx.x + hx.y + hx.z + hx.w;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            m1[u] = wave_sum(m1[u]) * (1.0f / FIRA_D);
            m2[u] = wave_sum(m2[u]) * (1.0f / FIRA_D);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u;
            f32x4 o4 = (h[u] - m1[u] - xh[u] * m2[u]) * rstd[u];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32v4_t, o4), rDs, om[u], 0, 0);
            if (p > 0.f) {                                  // wave-uniform; no memory instruction inside
                const uint32_t e0 = idx0 + (uint32_t)r * FIRA_D + lane * 4;
                o4.x *= dropout_scale(seed, site, e0 + 0, p, inv_keep);
                o4.y *= dropout_scale(seed, site, e0 + 1, p, inv_keep);
                o4.z *= dropout_scale(seed, site, e0 + 2, p, inv_keep);
                o4.w *= dropout_scale(seed, site, e0 + 3, p, inv_keep);
            }
            // no dx_drop: the descriptor has zero records and the store is dropped
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32v4_t, o4), rDx, (unsigned)r * (FIRA_D * 4u) + lo, 0, 0);
            if (EXTRA) {                                    // rows past M: o4 = 0
                dsum += o4;
                dwsum += o4 * rw[u];
            }
        }
    }
    // the four waves park their column sums side by side (plain 16-byte LDS stores: no zero fill, no LDS atomics) and the
    // workgroup adds them up
    *reinterpret_cast<f32x4*>(&red[wave][lane * 4]) = dg;
    *reinterpret_cast<f32x4*>(&red[wave][FIRA_D + lane * 4]) = db;
    if (EXTRA) {
        *reinterpret_cast<f32x4*>(&red[wave][2 * FIRA_D + lane * 4]) = dsum;
        *reinterpret_cast<f32x4*>(&red[wave][3 * FIRA_D + lane * 4]) = dwsum;
    }
    __syncthreads();
    for (int i = t; i < NV * FIRA_D; i += 256) {
        const float v = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        if (part) {
            // deferred reduction (engine): every workgroup stores its NV x 256 partial sums; one reducer launch adds them
            // up.  (n workgroups adding to the SAME 512 addresses serialise in L2: ~30 ns per same-address atomic, i.e.
            // ~15 us of tail for 500 workgroups -- more than the kernel's streaming time for the decoder-sized launches)
            part[(size_t)blockIdx.x * NV * FIRA_D + i] = v;
        } else if (i < FIRA_D) {
            unsafeAtomicAdd(&dgamma[i], v);
        } else if (i < 2 * FIRA_D) {
            unsafeAtomicAdd(&dbeta[i - FIRA_D], v);
        }
    }
}

// Bias part of the folded GCN's parameter gradients (engine.hip): with c = W2 b1,
//   dW2[i,k] += dc[i] * b1[k]          db1[k] += sum_i W2[i,k] * dc[i]
// One workgroup per output row i of dW2 / per column block of db1; 256 x 256 problem, latency only.
__global__ __launch_bounds__(256) void gcn_bias_unfold_kernel(const float* __restrict__ W2, const float* __restrict__ b1,
                                                              const float* __restrict__ dc, float* __restrict__ dW2,
                                                              float* __restrict__ db1) {
    const int i = blockIdx.x, k = threadIdx.x;
    const float dci = dc[i];
    dW2[(size_t)i * FIRA_D + k] += dci * b1[k];
    unsafeAtomicAdd(&db1[k], W2[(size_t)i * FIRA_D + k] * dci);
}
// the same for every GCN layer in one launch (grid.y = layer): the dc vectors come out of the deferred reduction at
// the end of the backward pass
// Round 5: a workgroup owns UNFOLD_ROWS rows i of one layer and thread k keeps  sum_i W2[i,k] dc[i]  in a register: 8 atomics
// per address instead of 256 (one workgroup per row i made 256 workgroups add to the same 256 floats of db1: ~13 us of
// serialised same-address atomics on the step's tail, right in front of Adam).
constexpr int UNFOLD_ROWS = 32;
__global__ __launch_bounds__(256) void gcn_bias_unfold_all_kernel(UnfoldTable tab) {
    const UnfoldEntry& q = tab.e[blockIdx.y];
    const int i0 = blockIdx.x * UNFOLD_ROWS, k = threadIdx.x;
    const float b1k = q.b1[k];
    float w2[UNFOLD_ROWS], g[UNFOLD_ROWS];
#pragma unroll
    for (int r = 0; r < UNFOLD_ROWS; ++r) {                 // every operand requested before the first use
        w2[r] = q.W2[(size_t)(i0 + r) * FIRA_D + k];
        g[r] = q.dW2[(size_t)(i0 + r) * FIRA_D + k];
    }
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < UNFOLD_ROWS; ++r) {
        const float dci = q.dc[i0 + r];                     // (wave-uniform: a scalar load)
        q.dW2[(size_t)(i0 + r) * FIRA_D + k] = g[r] + dci * b1k;
        acc = fmaf(w2[r], dci, acc);
    }
    unsafeAtomicAdd(&q.db1[k], acc);
}

// ------------------------------------------------------------------------------------------------
// out[n] += sum_m X[m,n]: bias gradients.  Block = 256 threads over 64 columns x 4 row-phases.
__global__ __launch_bounds__(256) void colsum_kernel(int M, int N, const float* __restrict__ X, int ldx,
                                                     float* __restrict__ out, int rows_per_block,
                                                     const float* __restrict__ w) {   // optional row weights
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ph = threadIdx.x >> 6;
    const int r_beg = blockIdx.y * rows_per_block, r_end = min(M, r_beg + rows_per_block);
    float acc = 0.f;
    if (c < N)
        for (int r = r_beg + ph; r < r_end; r += 4) acc += (w ? w[r] : 1.0f) * X[(size_t)r * ldx + c];
    red[ph][threadIdx.x & 63] = acc;
    __syncthreads();
    if (ph == 0 && c < N) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        unsafeAtomicAdd(&out[c], v);
    }
}


// ------------------------------------------------------------------------------------------------
// Row movers through index lists (W floats per row, W a multiple of 256):
//   MODE 0  out[r]          = in[src[r]]          (gather;  src == nullptr: identity)
//   MODE 1  out[dst[r]]     = in[r]               (scatter, overwrite)
//   MODE 2  out[dst[r]]    += in[r]               (scatter-add; dst has no duplicates: plain read-modify-write)
//   MODE 3  out[dst[r]]     = in[src[r]]          (gather + scatter)
template <int MODE>
__global__ __launch_bounds__(256) void rows_idx_kernel(int R, int W, float* __restrict__ out, int ld_out,
                                                       const float* __restrict__ in, int ld_in,
                                                       const int32_t* __restrict__ src, const int32_t* __restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const size_t ri = (MODE == 0 || MODE == 3) ? (size_t)(src ? src[r] : r) : (size_t)r;
    const size_t ro = (MODE == 0) ? (size_t)r : (size_t)dst[r];
    for (int c = lane * 4; c < W; c += 256) {
        float4 a = *reinterpret_cast<const float4*>(in + ri * ld_in + c);
        if (MODE == 2) {
            const float4 b = *reinterpret_cast<const float4*>(out + ro * ld_out + c);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        *reinterpret_cast<float4*>(out + ro * ld_out + c) = a;
    }
}
// sinusoidal tables (gnn_transformer.py:10-19): evaluated in fp64 like the reference's python floats, stored fp32
__global__ void pos_table_kernel(int L, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= L * FIRA_D) return;
    const int pos = i / FIRA_D, c = i % FIRA_D, j = c >> 1;
    const double a = (double)pos / pow(10000.0, (double)(2 * j) / (double)FIRA_D);
    out[i] = (float)((c & 1) ? cos(a) : sin(a));
}
// Everything a step derives from the ids alone, in one launch: both sinusoidal tables, the key masks, and the inverse
// of the head-row list (compact_row[flat target row] = position in the ascending `rows` list, or -1; rows == nullptr:
// every row, identity) together with the identity list itself.
__global__ void prep_kernel(int B, int L, int S, int T, const int32_t* __restrict__ sou, const int32_t* __restrict__ sub,
                            const int32_t* __restrict__ tar, int32_t* __restrict__ mem_valid,
                            int32_t* __restrict__ tar_valid, float* __restrict__ pos_code, float* __restrict__ pos_tar,
                            int R, const int32_t* __restrict__ rows, int32_t* __restrict__ compact_row,
                            int32_t* __restrict__ iota, float* __restrict__ loss_sum, int32_t* __restrict__ n_tok,
                            int Nc, const int32_t* __restrict__ code_rows, int Cc, int32_t* __restrict__ code_slot,
                            const int32_t* __restrict__ mem_rows, int Mc, int32_t* __restrict__ mem_slot,
                            // computed target rows (fira_batch.dec_off; nullptr = every row): row_bt[compact row] = its flat
                            // b*T + t index, rows_c[k] = compact row of head row k; compact_row is then indexed by compact row
                            const int32_t* __restrict__ dec_off, int32_t* __restrict__ row_bt,
                            int32_t* __restrict__ rows_c,
                            // computed MEMORY rows as ragged attention keys: mem_off[b] .. mem_off[b+1] = commit b's range in
                            // the (ascending) mem_dst list, mem_valid_c[k] = key mask of compact memory row k
                            const int32_t* __restrict__ mem_dst, int32_t* __restrict__ mem_off,
                            int32_t* __restrict__ mem_valid_c) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int W = L + S;
    if (mem_off && i <= B) {                     // first list entry whose dense slot is >= i*W
        int lo = 0, hi = Mc;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (mem_dst[mid] < i * W) lo = mid + 1; else hi = mid; }
        mem_off[i] = lo;
    }
    if (mem_valid_c && i < Mc) {
        const int d = mem_dst[i], b = d / W, j = d - b * W;
        mem_valid_c[i] = (j < L ? sou[b * L + j] : sub[b * S + j - L]) != 0;
    }
    if (code_slot && i < Nc) {
        // inverse of the (ascending) code-row and memory-row lists: slot of compact node i in each list, or -1.  Kernels that
        // produce node rows use them to store the listed rows a second time, compactly (the gather launches they replace
        // sat on the dependent chain)
        int lo = 0, hi = Cc;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (code_rows[mid] < i) lo = mid + 1; else hi = mid; }
        code_slot[i] = (lo < Cc && code_rows[lo] == i) ? lo : -1;
        lo = 0; hi = Mc;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (mem_rows[mid] < i) lo = mid + 1; else hi = mid; }
        mem_slot[i] = (lo < Mc && mem_rows[lo] == i) ? lo : -1;
    }
    if (i == 0) {                                // the step's loss accumulators start at zero (no separate fills)
        if (loss_sum) *loss_sum = 0.f;
        if (n_tok) *n_tok = 0;
    }
    if (i < B * W) {
        const int b = i / W, j = i - b * W;
        mem_valid[i] = (j < L ? sou[b * L + j] : sub[b * S + j - L]) != 0;
    }
    if (tar && i < B * T) {
        tar_valid[i] = tar[i] != 0;
        if (compact_row) {
            int v = i;
            if (rows) {                          // binary search in the ascending list
                int lo = 0, hi = R;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (rows[mid] < i) lo = mid + 1; else hi = mid;
                }
                v = (lo < R && rows[lo] == i) ? lo : -1;
            }
            int ci = i;                          // position of flat row i in the computed-row layout (-1: left out)
            if (dec_off) {
                const int b = i / T, t = i - b * T, o = dec_off[b];
                ci = t < dec_off[b + 1] - o ? o + t : -1;
                if (ci >= 0) row_bt[ci] = i;
                if (rows && v >= 0) rows_c[v] = ci >= 0 ? ci : 0;      // (a head row is always a computed row)
                if (!rows) v = ci;                                     // no head-row list: every computed row, in order
            }
            if (ci >= 0) compact_row[ci] = v;
        }
        if (iota) iota[i] = i;
    }
    if (i < (L + T) * FIRA_D) {
        const bool code = i < L * FIRA_D;
        const int k = code ? i : i - L * FIRA_D;
        const int pos = k / FIRA_D, c = k % FIRA_D, j = c >> 1;
        const double a = (double)pos / pow(10000.0, (double)(2 * j) / (double)FIRA_D);
        (code ? pos_code : pos_tar)[k] = (float)((c & 1) ? cos(a) : sin(a));
    }
}

// Node features of the COMPUTED nodes straight into the compact layout (gnn_transformer.py:46-52,58): one wave per
// node; code tokens add their position row.
__global__ __launch_bounds__(256) void node_features_kernel(int Nc, const int32_t* __restrict__ node_rows, int N, int L,
                                                            int S, const int32_t* __restrict__ sou,
                                                            const int32_t* __restrict__ sub,
                                                            const int32_t* __restrict__ ast,
                                                            const float* __restrict__ emb,
                                                            const float* __restrict__ ast_emb,
                                                            const float* __restrict__ pos_code,
                                                            float* __restrict__ X, const int32_t* __restrict__ slot2,
                                                            float* __restrict__ X2, AdamRowsView vw) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= Nc) return;
    const int g = node_rows[r], b = g / N, loc = g - b * N, A = N - L - S;
    const int s2 = slot2 ? slot2[r] : -1;              // second, compact copy of the listed rows (the first layer's code rows)
    float4 v;
    if (loc < L) {
        v = adam_rows_load(emb, sou[b * L + loc], lane, vw);         // (vw: the word table under a row-sparse Adam)
        const float4 p = *reinterpret_cast<const float4*>(pos_code + (size_t)loc * FIRA_D + lane * 4);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    } else if (loc < L + S) {
        v = adam_rows_load(emb, sub[b * S + loc - L], lane, vw);
    } else {
        v = *reinterpret_cast<const float4*>(ast_emb + (size_t)ast[b * A + loc - L - S] * FIRA_D + lane * 4);
    }
    *reinterpret_cast<float4*>(X + (size_t)r * FIRA_D + lane * 4) = v;
    if (s2 >= 0) *reinterpret_cast<float4*>(X2 + (size_t)s2 * FIRA_D + lane * 4) = v;
}

// ---------------------------------------------------------------- host launchers
// W floats of each row are moved; rows are ld_out / ld_in floats apart (a column block of a wider matrix)
int rows_move_ld(hipStream_t s, int mode, int R, int W, float* out, int ld_out, const float* in, int ld_in,
                 const int32_t* src, const int32_t* dst) {
    if (R <= 0) return 0;
    FIRA_REQUIRE(W % 256 == 0 && ld_out % 4 == 0 && ld_in % 4 == 0 && mode >= 0 && mode <= 3,
                 "rows_move: bad width %d / mode %d", W, mode);
    ProfScope prof(s, PROF_ROWOPS, 0.0);
    dim3 g(cdiv(R, 4)), b(256);
    if (mode == 0) hipLaunchKernelGGL(rows_idx_kernel<0>, g, b, 0, s, R, W, out, ld_out, in, ld_in, src, dst);
    else if (mode == 1) hipLaunchKernelGGL(rows_idx_kernel<1>, g, b, 0, s, R, W, out, ld_out, in, ld_in, src, dst);
    else if (mode == 2) hipLaunchKernelGGL(rows_idx_kernel<2>, g, b, 0, s, R, W, out, ld_out, in, ld_in, src, dst);
    else hipLaunchKernelGGL(rows_idx_kernel<3>, g, b, 0, s, R, W, out, ld_out, in, ld_in, src, dst);
    FIRA_CHECK_LAUNCH("rows_move");
    return 0;
}
int rows_move(hipStream_t s, int mode, int R, int W, float* out, const float* in, const int32_t* src, const int32_t* dst) {
    return rows_move_ld(s, mode, R, W, out, W, in, W, src, dst);
}
int rows_gather_idx(hipStream_t s, int R, float* compact, const float* src, const int32_t* rows) {
    return rows_move(s, 0, R, FIRA_D, compact, src, rows, nullptr);
}
int rows_scatter_add_idx(hipStream_t s, int R, const float* compact, float* dst, const int32_t* rows) {
    return rows_move(s, 2, R, FIRA_D, dst, compact, nullptr, rows);
}
__global__ void tar_mask_kernel(int n, const int32_t* __restrict__ tar, int32_t* __restrict__ valid) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) valid[i] = tar[i] != 0;
}
int tar_mask(hipStream_t s, int n, const int32_t* tar, int32_t* valid) {
    hipLaunchKernelGGL(tar_mask_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, n, tar, valid);
    FIRA_CHECK_LAUNCH("tar_mask");
    return 0;
}
int prep(hipStream_t s, int B, int L, int S, int T, const int32_t* sou, const int32_t* sub, const int32_t* tar,
         int32_t* mem_valid, int32_t* tar_valid, float* pos_code, float* pos_tar, int R, const int32_t* rows,
         int32_t* compact_row, int32_t* iota, float* loss_sum, int32_t* n_tok, int Nc, const int32_t* code_rows, int Cc,
         int32_t* code_slot, const int32_t* mem_rows, int Mc, int32_t* mem_slot, const int32_t* dec_off, int32_t* row_bt,
         int32_t* rows_c, const int32_t* mem_dst, int32_t* mem_off, int32_t* mem_valid_c) {
    const int n = std::max(std::max(std::max(B * (L + S), B * T), (L + T) * FIRA_D), code_slot ? Nc : 0);
    hipLaunchKernelGGL(prep_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, B, L, S, T, sou, sub, tar, mem_valid, tar_valid,
                       pos_code, pos_tar, R, rows, compact_row, iota, loss_sum, n_tok, Nc, code_rows, Cc, code_slot, mem_rows,
                       Mc, mem_slot, dec_off, row_bt, rows_c, mem_dst, mem_dst ? mem_off : nullptr, mem_dst ? mem_valid_c : nullptr);
    FIRA_CHECK_LAUNCH("prep");
    return 0;
}
int node_features(hipStream_t s, int Nc, const int32_t* node_rows, int N, int L, int S, const int32_t* sou,
                  const int32_t* sub, const int32_t* ast, const float* emb, const float* ast_emb, const float* pos_code,
                  float* X, const int32_t* slot2, float* X2, const AdamRowsView* vw) {
    ProfScope prof(s, PROF_ROWOPS, 0.0);
    if (Nc <= 0) return 0;
    hipLaunchKernelGGL(node_features_kernel, dim3(cdiv(Nc, 4)), dim3(256), 0, s, Nc, node_rows, N, L, S, sou, sub, ast, emb,
                       ast_emb, pos_code, X, slot2, X2, vw ? *vw : AdamRowsView());
    FIRA_CHECK_LAUNCH("node_features");
    return 0;
}
int fill_pos_tables(hipStream_t s, int L, float* pos_code, int T, float* pos_tar) {
    hipLaunchKernelGGL(pos_table_kernel, dim3(cdiv(L * FIRA_D, 256)), dim3(256), 0, s, L, pos_code);
    hipLaunchKernelGGL(pos_table_kernel, dim3(cdiv(T * FIRA_D, 256)), dim3(256), 0, s, T, pos_tar);
    FIRA_CHECK_LAUNCH("fill_pos_tables");
    return 0;
}
int embed_gather_fwd(hipStream_t s, int B, int L, const int32_t* idx, const float* table, const float* pos, float* out,
                     int out_bstride, int out_off) {
    ProfScope prof(s, PROF_ROWOPS, 0.0);
    const int rows = B * L;
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(embed_gather_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, rows, L, idx, table, pos, out,
                       out_bstride, out_off);
    FIRA_CHECK_LAUNCH("embed_gather_fwd");
    return 0;
}
int embed_rows_fwd(hipStream_t s, int R, int T, const int32_t* row_bt, const int32_t* idx, const float* table,
                   const float* pos, float* out, const AdamRowsView* vw) {
    ProfScope prof(s, PROF_ROWOPS, 0.0);
    if (R <= 0) return 0;
    hipLaunchKernelGGL(embed_rows_fwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, s, R, T, row_bt, idx, table, pos, out,
                       vw ? *vw : AdamRowsView());
    FIRA_CHECK_LAUNCH("embed_rows_fwd");
    return 0;
}
int embed_rows_bwd(hipStream_t s, int R, const int32_t* row_bt, const int32_t* idx, float* dtable, const float* dout,
                   int padding_idx) {
    ProfScope prof(s, PROF_ROWOPS, 0.0);
    if (R <= 0) return 0;
    hipLaunchKernelGGL(embed_rows_bwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, s, R, row_bt, idx, dtable, dout, padding_idx);
    FIRA_CHECK_LAUNCH("embed_rows_bwd");
    return 0;
}
int embed_gather_bwd_small(hipStream_t s, int B, int L, const int32_t* idx, float* dtable, const float* dout,
                           int out_bstride, int out_off, int padding_idx, int table_rows) {
    const int rows = B * L;
    if (rows <= 0) return 0;
    FIRA_REQUIRE(table_rows > 0 && table_rows * FIRA_D * 4 <= 150 * 1024, "embed_gather_bwd_small: table of %d rows does not fit LDS", table_rows);
    ProfScope prof(s, PROF_ROWOPS, 0.0);
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t ae = hipFuncSetAttribute((const void*)embed_gather_bwd_small_kernel,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        FIRA_REQUIRE(ae == hipSuccess, "embed_gather_bwd_small: cannot raise the dynamic LDS limit: %s", hipGetErrorString(ae));
        attr_set = true;
    }
    const int rpb = std::max(64, cdiv(rows, 96));
    hipLaunchKernelGGL(embed_gather_bwd_small_kernel, dim3(cdiv(rows, rpb)), dim3(256),
                       (size_t)table_rows * FIRA_D * sizeof(float), s, rows, L, idx, dtable, dout, out_bstride, out_off,
                       padding_idx, table_rows, rpb);
    FIRA_CHECK_LAUNCH("embed_gather_bwd_small");
    return 0;
}
int embed_grouped_bwd(hipStream_t s, int n_items, const int32_t* item_tok, const int32_t* item_ptr, const int32_t* rows,
                      float* dtable, const float* dnode) {
    ProfScope prof(s, PROF_ROWOPS, 0.0);
    if (n_items <= 0) return 0;
    hipLaunchKernelGGL(embed_grouped_bwd_kernel, dim3(cdiv(n_items, 4)), dim3(256), 0, s, n_items, item_tok, item_ptr,
                       rows, dtable, dnode);
    FIRA_CHECK_LAUNCH("embed_grouped_bwd");
    return 0;
}
int embed_list_bwd_small(hipStream_t s, int n, const int32_t* rows, const int32_t* ids, float* dtable, const float* dnode,
                         int table_rows) {
    if (n <= 0) return 0;
    FIRA_REQUIRE(table_rows > 0 && table_rows <= 65535, "embed_list_bwd_small: table of %d rows", table_rows);
    ProfScope prof(s, PROF_ROWOPS, 0.0);
    hipLaunchKernelGGL(embed_list_bwd_small_kernel, dim3(cdiv(n, ELS_SLICE), table_rows), dim3(256), 0, s, n, rows, ids, dtable,
                       dnode);
    FIRA_CHECK_LAUNCH("embed_list_bwd_small");
    return 0;
}
int embed_gather_bwd(hipStream_t s, int B, int L, const int32_t* idx, float* dtable, const float* dout,
                     int out_bstride, int out_off, int padding_idx) {
    ProfScope prof(s, PROF_ROWOPS, 0.0);
    const int rows = B * L;
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(embed_gather_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, rows, L, idx, dtable, dout,
                       out_bstride, out_off, padding_idx);
    FIRA_CHECK_LAUNCH("embed_gather_bwd");
    return 0;
}
int combination_fwd(hipStream_t s, int M, const float* qk, const float* vtab, int ldv, const int32_t* mark, float* out,
                    float dropout, uint64_t seed, uint32_t site) {
    ProfScope prof(s, PROF_ROWOPS, 0.0);
    if (M <= 0) return 0;
    const float inv_keep = dropout > 0.f ? 1.0f / (1.0f - dropout) : 1.0f;
    hipLaunchKernelGGL(combination_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, M, qk, vtab, ldv, mark, out, dropout,
                       inv_keep, seed, site);
    FIRA_CHECK_LAUNCH("combination_fwd");
    return 0;
}
int combination_bwd_blocks(int M) { return M > 0 ? cdiv(M, 16) : 0; }
int combination_bwd(hipStream_t s, int M, const float* qk, const float* vtab, int ldv, const int32_t* mark,
                    const float* dout, float* dqk, float* dvtab, int lddv, float dropout, uint64_t seed, uint32_t site,
                    float* part) {
    ProfScope prof(s, PROF_ROWOPS, 0.0);
    if (M <= 0) return 0;
    const float inv_keep = dropout > 0.f ? 1.0f / (1.0f - dropout) : 1.0f;
    const int rpb = 16;
    hipLaunchKernelGGL(combination_bwd_kernel, dim3(cdiv(M, rpb)), dim3(256), 0, s, M, qk, vtab, ldv, mark, dout, dqk,
                       dvtab, lddv, dropout, inv_keep, seed, site, rpb, part);
    FIRA_CHECK_LAUNCH("combination_bwd");
    return 0;
}
int add_layernorm_fwd(hipStream_t s, int M, float* x, const float* res, const float* gamma, const float* beta,
                      float* y, float* stats, float dropout, uint64_t seed, uint32_t site, const int32_t* y_rows,
                      const float* r1_row, const float* r1_col, const int32_t* slot2, float* y2, uint32_t idx0) {
    ProfScope prof(s, PROF_ROWOPS, 0.0);
    if (M <= 0) return 0;
    const float inv_keep = dropout > 0.f ? 1.0f / (1.0f - dropout) : 1.0f;
    hipLaunchKernelGGL(add_layernorm_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, M, x, res, gamma, beta, y, stats,
                       dropout, inv_keep, seed, site, y_rows, r1_row, r1_col, y2 ? slot2 : nullptr, y2, idx0);
    FIRA_CHECK_LAUNCH("add_layernorm_fwd");
    return 0;
}
// rows per workgroup / workgroups of one launch.  With atomics: ~500 workgroups (enough to fill the chip, few enough that
// the same-address atomics do not dominate); with deferred partials: up to 1024 workgroups of >= 8 rows.
// steps of 16 rows per workgroup: 1 when the partial sums are deferred (one round trip per workgroup), otherwise enough
// to keep the same-address atomics of the final reduction at <= 512 workgroups
static int ln_bwd_steps(int M, bool deferred) { return deferred ? 1 : std::max(1, cdiv(cdiv(M, LNB_ROWS), 512)); }
int add_layernorm_bwd_blocks(int M) { return M > 0 ? cdiv(M, LNB_ROWS) : 0; }
int add_layernorm_bwd(hipStream_t s, int M, const float* dy, const float* sum, const float* stats, const float* gamma,
                      float* ds, float* dx_drop, float* dgamma, float* dbeta, float dropout, uint64_t seed,
                      uint32_t site, const int32_t* rows, float* part, const float* row_w, uint32_t idx0) {
    ProfScope prof(s, PROF_ROWOPS, 0.0);
    if (M <= 0) return 0;
    FIRA_REQUIRE(M < (1 << 21), "add_layernorm_bwd: %d rows exceed the 2 GiB the kernel addresses", M);
    FIRA_REQUIRE(!(row_w && !part), "add_layernorm_bwd: the extra column sums need the partial-row output");
    const float inv_keep = dropout > 0.f ? 1.0f / (1.0f - dropout) : 1.0f;
    const int steps = ln_bwd_steps(M, part != nullptr);
    if (row_w)
        hipLaunchKernelGGL(add_layernorm_bwd_kernel<true>, dim3(cdiv(M, LNB_ROWS * steps)), dim3(256), 0, s, M, dy, sum, stats,
                           gamma, ds, dx_drop, dgamma, dbeta, dropout, inv_keep, seed, site, steps, rows, part, row_w, idx0);
    else
        hipLaunchKernelGGL(add_layernorm_bwd_kernel<false>, dim3(cdiv(M, LNB_ROWS * steps)), dim3(256), 0, s, M, dy, sum, stats,
                           gamma, ds, dx_drop, dgamma, dbeta, dropout, inv_keep, seed, site, steps, rows, part, row_w, idx0);
    FIRA_CHECK_LAUNCH("add_layernorm_bwd");
    return 0;
}
int gcn_bias_unfold_all(hipStream_t s, const UnfoldTable& tab) {
    if (tab.n <= 0) return 0;
    hipLaunchKernelGGL(gcn_bias_unfold_all_kernel, dim3(FIRA_D / UNFOLD_ROWS, tab.n), dim3(FIRA_D), 0, s, tab);
    FIRA_CHECK_LAUNCH("gcn_bias_unfold_all");
    return 0;
}
int gcn_bias_unfold(hipStream_t s, const float* W2, const float* b1, const float* dc, float* dW2, float* db1) {
    hipLaunchKernelGGL(gcn_bias_unfold_kernel, dim3(FIRA_D), dim3(FIRA_D), 0, s, W2, b1, dc, dW2, db1);
    FIRA_CHECK_LAUNCH("gcn_bias_unfold");
    return 0;
}
// out[r, :] = g[r,0] * w[0, :] + g[r,1] * w[1, :]   (the data gradient of the 2-way copy gate, Model.py:19: a rank-2 product;
// the tiled GEMM kernel spent 23 us of the dependent chain on it, queued behind the vocabulary projection's gradients)
__global__ __launch_bounds__(256) void rank2_rows_kernel(int M, const float* __restrict__ g, const float* __restrict__ w,
                                                         float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= M) return;
    const float g0 = g[2 * r], g1 = g[2 * r + 1];
    const float4 a = *reinterpret_cast<const float4*>(w + lane * 4), b = *reinterpret_cast<const float4*>(w + FIRA_D + lane * 4);
    *reinterpret_cast<float4*>(out + (size_t)r * FIRA_D + lane * 4) =
        make_float4(fmaf(g0, a.x, g1 * b.x), fmaf(g0, a.y, g1 * b.y), fmaf(g0, a.z, g1 * b.z), fmaf(g0, a.w, g1 * b.w));
}
int rank2_rows(hipStream_t s, int M, const float* g, const float* w, float* out) {
    ProfScope prof(s, PROF_ROWOPS, 0.0);
    if (M <= 0) return 0;
    hipLaunchKernelGGL(rank2_rows_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, M, g, w, out);
    FIRA_CHECK_LAUNCH("rank2_rows");
    return 0;
}
int colsum(hipStream_t s, int M, int N, const float* X, int ldx, float* out, const float* row_weight) {
    ProfScope prof(s, PROF_ROWOPS, 0.0);
    if (M <= 0 || N <= 0) return 0;
    const int rpb = 256;
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 64), cdiv(M, rpb)), dim3(256), 0, s, M, N, X, ldx, out, rpb, row_weight);
    FIRA_CHECK_LAUNCH("colsum");
    return 0;
}

// grid (len, BR, 2*nl): one wave-row copy per block of 64 threads
__global__ __launch_bounds__(64) void permute_cache_kernel(int BR, int T, const int32_t* __restrict__ parent,
                                                           const float* __restrict__ ksrc, const float* __restrict__ vsrc,
                                                           float* __restrict__ kdst, float* __restrict__ vdst, int nl) {
    const int pos = blockIdx.x, r = blockIdx.y, which = blockIdx.z, lane = threadIdx.x;
    const int l = which >> 1;
    const float* src = (which & 1) ? vsrc : ksrc;
    float* dst = (which & 1) ? vdst : kdst;
    const int pr = parent ? parent[r] : r;
    const size_t lay = (size_t)l * BR * T * FIRA_D;
    *reinterpret_cast<float4*>(dst + lay + ((size_t)r * T + pos) * FIRA_D + lane * 4) =
        *reinterpret_cast<const float4*>(src + lay + ((size_t)pr * T + pos) * FIRA_D + lane * 4);
}
__global__ void permute_hist_kernel(int BR, int T, int len, const int32_t* __restrict__ parent,
                                    const int32_t* __restrict__ src, int32_t* __restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= BR * len) return;
    const int r = i / len, pos = i - r * len;
    dst[r * T + pos] = src[(parent ? parent[r] : r) * T + pos];
}
int permute_cache(hipStream_t s, int nl, int BR, int T, int len, const int32_t* parent, const float* ksrc,
                  const float* vsrc, float* kdst, float* vdst, const int32_t* hist_src, int32_t* hist_dst) {
    if (len <= 0 || BR <= 0) return 0;
    hipLaunchKernelGGL(permute_cache_kernel, dim3(len, BR, 2 * nl), dim3(64), 0, s, BR, T, parent, ksrc, vsrc, kdst,
                       vdst, nl);
    hipLaunchKernelGGL(permute_hist_kernel, dim3(cdiv(BR * len, 256)), dim3(256), 0, s, BR, T, len, parent, hist_src,
                       hist_dst);
    FIRA_CHECK_LAUNCH("permute_cache");
    return 0;
}
__global__ void mark_history_kernel(int BR, int T, int step, const int32_t* __restrict__ tokens, int32_t* __restrict__ hist) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < BR) hist[r * T + step] = tokens[r] != 0;
}
// One decode step's first launch: hist[r, step] = tokens[r] != 0 (key-valid history of the self-attention cache) and
// x[r] = table[tokens[r]] + pos_row (gnn_transformer.py:110-113 at position `step`), one wave per hypothesis row.
__global__ __launch_bounds__(256) void decode_embed_kernel(int BR, int T, int step, const int32_t* __restrict__ tokens,
                                                           const float* __restrict__ table,
                                                           const float* __restrict__ pos_row, float* __restrict__ x,
                                                           int32_t* __restrict__ hist) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= BR) return;
    const int id = tokens[r];
    float4 v = *reinterpret_cast<const float4*>(table + (size_t)id * FIRA_D + lane * 4);
    const float4 p = *reinterpret_cast<const float4*>(pos_row + lane * 4);
    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    *reinterpret_cast<float4*>(x + (size_t)r * FIRA_D + lane * 4) = v;
    if (lane == 0) hist[r * T + step] = id != 0;
}
int decode_embed(hipStream_t s, int BR, int T, int step, const int32_t* tokens, const float* table, const float* pos_row,
                 float* x, int32_t* hist) {
    hipLaunchKernelGGL(decode_embed_kernel, dim3(cdiv(BR, 4)), dim3(256), 0, s, BR, T, step, tokens, table, pos_row, x, hist);
    FIRA_CHECK_LAUNCH("decode_embed");
    return 0;
}
// fp32 -> raw bf16 (round to nearest even), 4 elements per thread: the decode loop's optional bf16 copy of the cross K|V
typedef __bf16 rbf16x2 __attribute__((ext_vector_type(2)));
typedef float rf32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void rows_to_bf16_kernel(int64_t n4, const float4* __restrict__ in, uint2* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = in[i];
        const rf32x2 a = {v.x, v.y}, b = {v.z, v.w};
        out[i] = uint2{__builtin_bit_cast(uint32_t, __builtin_convertvector(a, rbf16x2)),
                       __builtin_bit_cast(uint32_t, __builtin_convertvector(b, rbf16x2))};
    }
}
int rows_to_bf16(hipStream_t s, int64_t n, const float* in, uint16_t* out) {
    if (n <= 0) return 0;
    FIRA_REQUIRE(n % 4 == 0 && (uintptr_t)in % 16 == 0 && (uintptr_t)out % 8 == 0, "rows_to_bf16: bad size / alignment");
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(rows_to_bf16_kernel, dim3((unsigned)std::min<int64_t>((n4 + 255) / 256, 256 * 32)), dim3(256), 0, s, n4,
                       reinterpret_cast<const float4*>(in), reinterpret_cast<uint2*>(out));
    FIRA_CHECK_LAUNCH("rows_to_bf16");
    return 0;
}
// out[i] = float(in[i]) (exact): the way back from the bf16 gradient wire format (parallel.GradReducer)
__global__ __launch_bounds__(256) void rows_from_bf16_kernel(int64_t n4, const uint2* __restrict__ in, float4* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const uint2 q = in[i];
        float4 o;
        o.x = __builtin_bit_cast(float, q.x << 16); o.y = __builtin_bit_cast(float, q.x & 0xffff0000u);
        o.z = __builtin_bit_cast(float, q.y << 16); o.w = __builtin_bit_cast(float, q.y & 0xffff0000u);
        out[i] = o;
    }
}
int rows_from_bf16(hipStream_t s, int64_t n, const uint16_t* in, float* out) {
    if (n <= 0) return 0;
    FIRA_REQUIRE(n % 4 == 0 && (uintptr_t)in % 8 == 0 && (uintptr_t)out % 16 == 0, "rows_from_bf16: bad size / alignment");
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(rows_from_bf16_kernel, dim3((unsigned)std::min<int64_t>((n4 + 255) / 256, 256 * 32)), dim3(256), 0, s, n4,
                       reinterpret_cast<const uint2*>(in), reinterpret_cast<float4*>(out));
    FIRA_CHECK_LAUNCH("rows_from_bf16");
    return 0;
}
int mark_history(hipStream_t s, int BR, int T, int step, const int32_t* tokens, int32_t* hist) {
    hipLaunchKernelGGL(mark_history_kernel, dim3(cdiv(BR, 256)), dim3(256), 0, s, BR, T, step, tokens, hist);
    FIRA_CHECK_LAUNCH("mark_history");
    return 0;
}

// Deferred column reductions: dst[c] += sum_p src[p * stride + c] for a table of (dst, src, width, n_part, stride).
// The backward kernels that reduce over rows into a few hundred addresses (LayerNorm gamma/beta, the Combination's
// 4-row value table, the copy head's w / bias) store one partial row per workgroup; this kernel, launched once for the
// decoder-side entries and once for the encoder-side ones, adds them up: a workgroup sums RED_ROWS partial rows of 64
// columns (4 row phases) and adds the result to the destination.
constexpr int RED_ROWS = 64;
__global__ __launch_bounds__(256) void deferred_reduce_kernel(RedTable tab) {
    __shared__ float sm[4][64];
    int e = 0;
    while (e + 1 < tab.n && (int)blockIdx.x >= tab.wg_start[e + 1]) ++e;
    const RedEntry& q = tab.e[e];
    const int w = blockIdx.x - tab.wg_start[e];
    const int ncb = (q.width + 63) / 64;                 // column blocks of this entry
    const int cb = w % ncb, chunk = w / ncb;             // this workgroup: 64 columns x RED_ROWS partial rows
    const int c = cb * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    const int p_end = min(q.n_part, (chunk + 1) * RED_ROWS);
    float acc = 0.f;
    if (c < q.width)
        for (int p = chunk * RED_ROWS + ph; p < p_end; p += 4) acc += q.src[(size_t)p * q.stride + c];
    sm[ph][threadIdx.x & 63] = acc;
    __syncthreads();
    if (ph == 0 && c < q.width)                          // <= n_part / RED_ROWS workgroups per address: a handful
        unsafeAtomicAdd(&q.dst[c], (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]));
}
int deferred_reduce(hipStream_t s, RedTable& tab) {
    if (tab.n == 0) return 0;
    ProfScope prof(s, PROF_ROWOPS, 0.0);
    tab.wg_start[0] = 0;
    for (int i = 0; i < tab.n; ++i)
        tab.wg_start[i + 1] = tab.wg_start[i] + cdiv(tab.e[i].width, 64) * cdiv(tab.e[i].n_part, RED_ROWS);
    hipLaunchKernelGGL(deferred_reduce_kernel, dim3(tab.wg_start[tab.n]), dim3(256), 0, s, tab);
    tab.n = 0;
    FIRA_CHECK_LAUNCH("deferred_reduce");
    return 0;
}

// keep/scale factors of one dropout site, as every kernel derives them (test surface: lets the CPU oracle apply the
// SAME masks, so training-mode loss and gradients are compared, not just their statistics)
__global__ __launch_bounds__(256) void dropout_mask_kernel(int64_t n, float* __restrict__ out, float p, float inv_keep,
                                                           uint64_t seed, uint32_t site) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = p > 0.f ? dropout_scale(seed, site, (uint32_t)i, p, inv_keep) : 1.f;
}

}  // namespace fira

extern "C" {
int fira_f32_to_bf16(void* stream, int64_t n, const float* in, uint16_t* out) {
    FIRA_REQUIRE(in && out && n >= 0, "fira_f32_to_bf16: bad argument");
    return fira::rows_to_bf16((hipStream_t)stream, n, in, out);
}
int fira_bf16_to_f32(void* stream, int64_t n, const uint16_t* in, float* out) {
    FIRA_REQUIRE(in && out && n >= 0, "fira_bf16_to_f32: bad argument");
    return fira::rows_from_bf16((hipStream_t)stream, n, in, out);
}
int fira_dropout_mask(void* stream, uint64_t seed, uint32_t site, int64_t n, float p, float* out) {
    FIRA_REQUIRE(out && n >= 0 && p >= 0.f && p < 1.f, "fira_dropout_mask: bad argument");
    if (n == 0) return 0;
    hipLaunchKernelGGL(fira::dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n,
                       out, p, 1.0f / (1.0f - p), seed, site);
    FIRA_CHECK_LAUNCH("dropout_mask");
    return 0;
}
int fira_embed_gather_fwd(void* stream, int B, int L, const int32_t* idx, const float* table, const float* pos,
                          float* out, int out_bstride, int out_off) {
    return fira::embed_gather_fwd((hipStream_t)stream, B, L, idx, table, pos, out, out_bstride, out_off);
}
int fira_embed_gather_bwd(void* stream, int B, int L, const int32_t* idx, float* dtable, const float* dout,
                          int out_bstride, int out_off, int padding_idx) {
    return fira::embed_gather_bwd((hipStream_t)stream, B, L, idx, dtable, dout, out_bstride, out_off, padding_idx);
}
int fira_combination_fwd(void* stream, int M, const float* qk, const float* vtab, const int32_t* mark, float* out,
                         float dropout, uint64_t seed, uint32_t stream_id) {
    return fira::combination_fwd((hipStream_t)stream, M, qk, vtab, FIRA_D, mark, out, dropout, seed, stream_id);
}
int fira_combination_bwd(void* stream, int M, const float* qk, const float* vtab, const int32_t* mark,
                         const float* dout, float* dqk, float* dvtab, float dropout, uint64_t seed,
                         uint32_t stream_id) {
    return fira::combination_bwd((hipStream_t)stream, M, qk, vtab, FIRA_D, mark, dout, dqk, dvtab, FIRA_D, dropout, seed,
                                 stream_id);
}
int fira_add_layernorm_fwd(void* stream, int M, float* x, const float* res, const float* gamma, const float* beta,
                           float* y, float* stats, float dropout, uint64_t seed, uint32_t stream_id) {
    return fira::add_layernorm_fwd((hipStream_t)stream, M, x, res, gamma, beta, y, stats, dropout, seed, stream_id,
                                   nullptr);
}
int fira_add_layernorm_bwd(void* stream, int M, const float* dy, const float* sum, const float* stats,
                           const float* gamma, float* ds, float* dx_drop, float* dgamma, float* dbeta, float dropout,
                           uint64_t seed, uint32_t stream_id) {
    return fira::add_layernorm_bwd((hipStream_t)stream, M, dy, sum, stats, gamma, ds, dx_drop, dgamma, dbeta, dropout,
                                   seed, stream_id);
}
int fira_colsum_f32(void* stream, int M, int N, const float* X, int ldx, float* out) {
    return fira::colsum((hipStream_t)stream, M, N, X, ldx, out);
}
}
