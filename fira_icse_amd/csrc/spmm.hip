// CSR SpMM for the GCN aggregation  Z = A_hat · H  (reference gnn_transformer.py:80 does a dense
// torch.bmm over a [B,650,650] float64->float32 adjacency that is ~0.5 % dense).
//
// Data layout: one block-diagonal CSR over all B graphs of the batch (rowptr int32[B*N+1], col int32
// global node id, val fp32); features row-major [B*N, 256] fp32.  A_hat is symmetric, so the backward
// dH = A_hat^T · dZ is the same kernel.
//
// variant 1 (default, realistic graphs, ~4 nnz/row): one wavefront per output row; the 64 lanes hold the
//   256-wide row as float4 (one coalesced 1 KiB load per neighbour row); col/val of the row are fetched
//   64 at a time with one coalesced load and broadcast lane-to-lane (readlane), neighbour loads are
//   issued 4 deep before the FMAs.  No atomics: the reduction over neighbours is a per-lane register sum.
//   Neighbour re-reads (each H row is used ~4x) hit L2: a graph's H is 665 KB and the workgroup->row mapping is
//   XCD-aware (each XCD's private L2 serves whole graphs).
// variant 2 (dense stress graphs, SURVEY.md config 5, ~120 nnz/row): one workgroup per (graph, 64-column
//   slab); the slab of H (rows x 64 floats) is staged once into LDS with coalesced float4 loads and every
//   neighbour gather is an LDS read (ds_read_b128: a 16-lane group covers one 64-float row and owns one
//   output row, so a wave aggregates 4 rows at once with register sums only).
#include "common.h"
#include <stdlib.h>
#include "epilogue.h"

namespace fira {

typedef float f32x4s __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void spmm_rowwave_kernel(int n_rows, const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col,
                                                           const float* __restrict__ val,
                                                           const float* __restrict__ X, int ldx,
                                                           float* __restrict__ Y, int ldy, int accum,
                                                           float* __restrict__ rowsum) {
    const int lane = threadIdx.x & 63;
    // XCD-aware row mapping: workgroup b runs on XCD b % 8 and each XCD has its own L2, so workgroup b is given the
    // row group (b % 8) * ceil(nwg / 8) + b / 8: every XCD then owns one contiguous eighth of the rows (whole graphs of
    // the block-diagonal batch), and the ~4x re-reads of a graph's H rows hit that XCD's L2 instead of each XCD
    // fetching them again (PMC: 2.6x the compulsory read traffic with the round-robin mapping).
    const int per_xcd = (gridDim.x + 7) >> 3;
    const int grp = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int row = grp * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float vsum = 0.f;
    for (int base = beg; base < end; base += 64) {
        const int n = min(64, end - base);
        int c = 0;
        float v = 0.f;
        if (lane < n) {
            c = col[base + lane];
            v = val[base + lane];
        }
        vsum += v;
        int j = 0;
        for (; j + 4 <= n; j += 4) {
            const int c0 = __shfl(c, j, 64), c1 = __shfl(c, j + 1, 64), c2 = __shfl(c, j + 2, 64),
                      c3 = __shfl(c, j + 3, 64);
            const float4 x0 = *reinterpret_cast<const float4*>(X + (size_t)c0 * ldx + lane * 4);
            const float4 x1 = *reinterpret_cast<const float4*>(X + (size_t)c1 * ldx + lane * 4);
            const float4 x2 = *reinterpret_cast<const float4*>(X + (size_t)c2 * ldx + lane * 4);
            const float4 x3 = *reinterpret_cast<const float4*>(X + (size_t)c3 * ldx + lane * 4);
            const float v0 = __shfl(v, j, 64), v1 = __shfl(v, j + 1, 64), v2 = __shfl(v, j + 2, 64),
                        v3 = __shfl(v, j + 3, 64);
            acc.x = fmaf(v0, x0.x, acc.x); acc.y = fmaf(v0, x0.y, acc.y); acc.z = fmaf(v0, x0.z, acc.z); acc.w = fmaf(v0, x0.w, acc.w);
            acc.x = fmaf(v1, x1.x, acc.x); acc.y = fmaf(v1, x1.y, acc.y); acc.z = fmaf(v1, x1.z, acc.z); acc.w = fmaf(v1, x1.w, acc.w);
            acc.x = fmaf(v2, x2.x, acc.x); acc.y = fmaf(v2, x2.y, acc.y); acc.z = fmaf(v2, x2.z, acc.z); acc.w = fmaf(v2, x2.w, acc.w);
            acc.x = fmaf(v3, x3.x, acc.x); acc.y = fmaf(v3, x3.y, acc.y); acc.z = fmaf(v3, x3.z, acc.z); acc.w = fmaf(v3, x3.w, acc.w);
        }
        for (; j < n; ++j) {
            const int c0 = __shfl(c, j, 64);
            const float v0 = __shfl(v, j, 64);
            const float4 x0 = *reinterpret_cast<const float4*>(X + (size_t)c0 * ldx + lane * 4);
            acc.x = fmaf(v0, x0.x, acc.x); acc.y = fmaf(v0, x0.y, acc.y); acc.z = fmaf(v0, x0.z, acc.z); acc.w = fmaf(v0, x0.w, acc.w);
        }
    }
    float4* yp = reinterpret_cast<float4*>(Y + (size_t)row * ldy + lane * 4);
    if (accum) {                                     // Y += A X (the aggregation's backward lands on a residual gradient)
        const float4 y0 = *yp;
        acc.x += y0.x; acc.y += y0.y; acc.z += y0.z; acc.w += y0.w;
    }
    *yp = acc;
    if (rowsum) {                                    // r = A 1, needed where a bias travels through the aggregation
        vsum = wave_sum(vsum);
        if (lane == 0) rowsum[row] = vsum;
    }
}

// Round 5 -- the row-per-wave kernel above is bound by its chain of three dependent round trips (row offsets -> (col, val) ->
// neighbour rows) for ONE KiB of output per wave: 41 600 waves in five generations of ~5 us.  Here a wave owns FOUR consecutive
// rows and batches every trip over them (the gather of gcn_fused.hip): one load for the five row offsets, one for the first
// 16 (col, val) pairs of each row (lanes 16 i .. 16 i + 15 = row i), then the listed neighbour rows of all four rows EIGHT
// per round trip, whichever row they belong to (the set bits of one ballot word, in lane order; a lane group's index is the
// accumulator).  Entries beyond the last one of a trip carry the out-of-range offset of the buffer descriptor: no request.
// Rows with more than 16 entries finish as the kernel above does.  4x fewer waves, 4-5 trips per four rows instead of 12.
constexpr int SB_U = 8;          // neighbour rows per round trip (16: 14.4 against 13.4 us on batch 64's compact rows)
template <bool ACCUM>
__global__ __launch_bounds__(256) void spmm_rowbatch_kernel(int n_rows, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ col,
                                                            const float* __restrict__ val, const float* __restrict__ X,
                                                            int ldx, float* __restrict__ Y, int ldy,
                                                            float* __restrict__ rowsum) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_xcd = (gridDim.x + 7) >> 3;
    const int grp = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);          // XCD-aware: see spmm_rowwave_kernel
    const int rbase = grp * 16 + wave * 4;
    if (rbase >= n_rows) return;
    int rp = 0;
    if (lane <= 4) rp = rowptr[min(rbase + lane, n_rows)];
    int beg[4], cnt[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        beg[i] = __builtin_amdgcn_readlane(rp, i);
        cnt[i] = rbase + i < n_rows ? __builtin_amdgcn_readlane(rp, i + 1) - beg[i] : 0;
    }
    const int gi = lane >> 4, ge = lane & 15;
    const int my_beg = gi == 0 ? beg[0] : gi == 1 ? beg[1] : gi == 2 ? beg[2] : beg[3];
    const int my_cnt = gi == 0 ? cnt[0] : gi == 1 ? cnt[1] : gi == 2 ? cnt[2] : cnt[3];
    int c = 0;
    float v = 0.f;
    if (ge < my_cnt) {
        c = col[my_beg + ge];
        v = val[my_beg + ge];
    }
    f32x4s old[ACCUM ? 4 : 1];
    if constexpr (ACCUM) {                                // (requested with the index lists still in flight)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            old[i] = *reinterpret_cast<const f32x4s*>(Y + (size_t)min(rbase + i, n_rows - 1) * ldy + lane * 4);
    }
    const rsrc_t rX = buf_rsrc(X, (unsigned)((size_t)n_rows * ldx * 4));
    f32x4s acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4s{0.f, 0.f, 0.f, 0.f};
    unsigned long long m = __ballot(ge < my_cnt);
    while (m) {                                           // wave-uniform: the listed entries of the four rows, SB_U per trip
        f32x4s g[SB_U];
        float w[SB_U];
        int own[SB_U];
#pragma unroll
        for (int u = 0; u < SB_U; ++u) {
            const bool ok = m != 0ull;
            const int src = ok ? (int)__builtin_ctzll(m) : 0;
            if (ok) m &= m - 1;
            own[u] = ok ? src >> 4 : -1;
            const int cj = __builtin_amdgcn_readlane(c, src);
            w[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
            g[u] = __builtin_bit_cast(f32x4s, __builtin_amdgcn_raw_buffer_load_b128(rX, ok ? (unsigned)lane * 16u : FIRA_OOB,
                                                                                    (unsigned)cj * (unsigned)ldx * 4u, 0));
        }
#pragma unroll
        for (int u = 0; u < SB_U; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (own[u] == i) {                        // wave-uniform
                    acc[i].x = fmaf(w[u], g[u].x, acc[i].x); acc[i].y = fmaf(w[u], g[u].y, acc[i].y);
                    acc[i].z = fmaf(w[u], g[u].z, acc[i].z); acc[i].w = fmaf(w[u], g[u].w, acc[i].w);
                }
        }
    }
    float vs = sum16(v);
    float vsum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) vsum[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vs), i * 16));
#pragma unroll
    for (int i = 0; i < 4; ++i) {                         // hub rows: entries 16 .. as the row-per-wave kernel walks them
        if (cnt[i] <= 16) continue;                       // wave-uniform
        float extra = 0.f;
        for (int base = beg[i] + 16; base < beg[i] + cnt[i]; base += 64) {
            const int n = min(64, beg[i] + cnt[i] - base);
            int c2 = 0;
            float v2 = 0.f;
            if (lane < n) {
                c2 = col[base + lane];
                v2 = val[base + lane];
            }
            extra += v2;
            for (int j = 0; j < n; j += 4) {              // (lanes past n hold col 0 / val 0: a valid row times zero)
                f32x4s x4[4];
                float w4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int sl = min(j + u, 63);
                    const int cj = __builtin_amdgcn_readlane(c2, sl);
                    w4[u] = j + u < 64 ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v2), sl)) : 0.f;
                    x4[u] = *reinterpret_cast<const f32x4s*>(X + (size_t)cj * ldx + lane * 4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc[i].x = fmaf(w4[u], x4[u].x, acc[i].x); acc[i].y = fmaf(w4[u], x4[u].y, acc[i].y);
                    acc[i].z = fmaf(w4[u], x4[u].z, acc[i].z); acc[i].w = fmaf(w4[u], x4[u].w, acc[i].w);
                }
            }
        }
        vsum[i] += wave_sum(extra);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (rbase + i >= n_rows) continue;                // wave-uniform
        f32x4s y = acc[i];
        if constexpr (ACCUM) y += old[i];
        // (streamed: nobody gathers from Y in this launch, the lines should not push X rows out of the L2)
        __builtin_nontemporal_store(y, reinterpret_cast<f32x4s*>(Y + (size_t)(rbase + i) * ldy + lane * 4));
        if (rowsum && lane == 0) rowsum[rbase + i] = vsum[i];
    }
}

// LDS-staged variant: grid (4 column slabs, n_graphs); block 1024 threads = 16 waves.
// LDS: the graph's 64-column slab of H (graph_rows x 64 floats, <= 128 KiB) + a per-wave (col, val) staging area.
// Each 16-lane group of a wave owns ONE output row (16 lanes x float4 = the 64 slab columns), so a wave aggregates
// 4 rows at once with register sums only.  A row's (col, val) list is first copied into LDS with coalesced loads by
// the group's own 16 lanes, so the inner loop touches LDS only: one broadcast ds_read_b64 for (col, val) and one
// ds_read_b128 for the neighbour's float4, 8 neighbours in flight per lane.
// crossover densities (fira_csr_spmm; profiles/r3_spmm_crossover.md): CSR row-wave below 2 %; fp32: the LDS-slab CSR
// kernel up to ~27 %, the block-dense fp32 MFMA kernel above (it is MFMA-bound: 17 GFLOP per launch whatever the
// density); bf16 operands allowed: the block-dense bf16 kernel from 2.5 %
constexpr double SPMM_LDS_AT = 0.02, SPMM_DENSE_AT_F32 = 0.27, SPMM_DENSE_AT_BF16 = 0.025;
constexpr int SLAB = 64;
constexpr int STAGE = 64;             // (col,val) entries staged per group per pass
constexpr int LDS_WAVES = 16;         // 1024 threads: 4 waves per SIMD keep the LDS pipe busy
__global__ __launch_bounds__(1024) void spmm_lds_kernel(int graph_rows, const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col,
                                                       const float* __restrict__ val,
                                                       const float* __restrict__ X, int ldx,
                                                       float* __restrict__ Y, int ldy) {
    extern __shared__ __attribute__((aligned(16))) float slab[];   // [graph_rows][64] then int2 stage[8][4][STAGE]
    const int g = blockIdx.y, c0 = blockIdx.x * SLAB;
    const int row0 = g * graph_rows;
    const int t = threadIdx.x;
    for (int r = t >> 4; r < graph_rows; r += LDS_WAVES * 4) {
        const float4 x = *reinterpret_cast<const float4*>(X + (size_t)(row0 + r) * ldx + c0 + (t & 15) * 4);
        *reinterpret_cast<float4*>(&slab[r * SLAB + (t & 15) * 4]) = x;
    }
    __syncthreads();
    const int lane = t & 63, wave = t >> 6;
    const int sub = lane >> 4, l16 = lane & 15;
    const int q = l16 * 4;
    int2* stage = reinterpret_cast<int2*>(slab + (size_t)graph_rows * SLAB) + (wave * 4 + sub) * STAGE;

    auto gather = [&](float4& acc, int n) {          // acc += sum over stage[0..n) of val * slab[col]
        int j = 0;
        for (; j + 8 <= n; j += 8) {
            int2 e[8];
#pragma unroll
            for (int u = 0; u < 8; u += 2) {           // two (col,val) pairs per broadcast ds_read_b128
                const int4 ee = *reinterpret_cast<const int4*>(&stage[j + u]);
                e[u] = make_int2(ee.x, ee.y);
                e[u + 1] = make_int2(ee.z, ee.w);
            }
            float4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const float4*>(&slab[e[u].x * SLAB + q]);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float v = __int_as_float(e[u].y);
                acc.x = fmaf(v, x[u].x, acc.x); acc.y = fmaf(v, x[u].y, acc.y);
                acc.z = fmaf(v, x[u].z, acc.z); acc.w = fmaf(v, x[u].w, acc.w);
            }
        }
        for (; j < n; ++j) {
            const int2 e = stage[j];
            const float4 x = *reinterpret_cast<const float4*>(&slab[e.x * SLAB + q]);
            const float v = __int_as_float(e.y);
            acc.x = fmaf(v, x.x, acc.x); acc.y = fmaf(v, x.y, acc.y);
            acc.z = fmaf(v, x.z, acc.z); acc.w = fmaf(v, x.w, acc.w);
        }
    };
    // (col, val) of a row's first STAGE entries, one register pair per lane and pass slot
    int2 pre[STAGE / 16];
    int beg = 0, end = 0;
    auto prefetch = [&](int r) {
        const bool live = r < graph_rows;
        beg = live ? rowptr[row0 + r] : 0;
        end = live ? rowptr[row0 + r + 1] : 0;
#pragma unroll
        for (int u = 0; u < STAGE / 16; ++u) {
            const int i = beg + u * 16 + l16;
            pre[u] = i < end ? make_int2(col[i] - row0, __float_as_int(val[i])) : make_int2(0, 0);
        }
    };
    prefetch(wave * 4 + sub);
    for (int r4 = wave * 4; r4 < graph_rows; r4 += LDS_WAVES * 4) {
        const int r = r4 + sub;
        const int cbeg = beg, cend = end;
#pragma unroll
        for (int u = 0; u < STAGE / 16; ++u) stage[u * 16 + l16] = pre[u];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        prefetch(r + LDS_WAVES * 4);                   // next row's index list streams in under this row's gathers
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        gather(acc, min(STAGE, cend - cbeg));
        for (int base = cbeg + STAGE; base < cend; base += STAGE) {      // rows longer than one staging pass
            __builtin_amdgcn_wave_barrier();
            const int n = min(STAGE, cend - base);
#pragma unroll
            for (int u = 0; u < STAGE / 16; ++u) {
                const int i = u * 16 + l16;
                if (i < n) stage[i] = make_int2(col[base + i] - row0, __float_as_int(val[base + i]));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            gather(acc, n);
        }
        __builtin_amdgcn_wave_barrier();
        if (r < graph_rows) *reinterpret_cast<float4*>(Y + (size_t)(row0 + r) * ldy + c0 + q) = acc;
    }
}

int csr_spmm_ex(hipStream_t s, int n_rows, const int32_t* rowptr, const int32_t* col, const float* val, const float* X,
                int ldx, float* Y, int ldy, int graph_rows, int variant, int accum, float* rowsum);
int csr_spmm_dense(hipStream_t s, int n_rows, const int32_t* rowptr, const int32_t* col, const float* val, const float* X,
                   int ldx, float* Y, int ldy, int graph_rows, int bf16);          // spmm_dense.hip
int csr_spmm(hipStream_t s, int n_rows, const int32_t* rowptr, const int32_t* col, const float* val, const float* X,
             int ldx, float* Y, int ldy, int graph_rows, int variant) {
    return csr_spmm_ex(s, n_rows, rowptr, col, val, X, ldx, Y, ldy, graph_rows, variant, 0, nullptr);
}
int csr_spmm_ex(hipStream_t s, int n_rows, const int32_t* rowptr, const int32_t* col, const float* val, const float* X,
                int ldx, float* Y, int ldy, int graph_rows, int variant, int accum, float* rowsum) {
    if (n_rows <= 0) return 0;
    FIRA_REQUIRE(!((accum || rowsum) && variant >= 2 && variant != 5), "csr_spmm: accumulate / row sums need a row-per-wave variant (1, 5)");
    FIRA_REQUIRE(variant >= 0 && variant <= 5, "csr_spmm: variant must be 0..5");
    if (variant == 3 || variant == 4) return csr_spmm_dense(s, n_rows, rowptr, col, val, X, ldx, Y, ldy, graph_rows, variant == 4);
    FIRA_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ((uintptr_t)X % 16 == 0) && ((uintptr_t)Y % 16 == 0),
                 "csr_spmm: feature rows must be 16-byte aligned");
    if (variant == 0) variant = 1;
    // algorithmic bytes: rowptr + (col,val) + features in + features out.  nnz is not known on the
    // host here: the caller's nnz is folded in by fira_prof users through the count (bench.py adds 8*nnz itself).
    ProfScope prof(s, PROF_SPMM, 4.0 * (n_rows + 1) + 2.0 * n_rows * FIRA_D * 4.0);
    if (variant == 2) {
        const size_t lds = (size_t)graph_rows * SLAB * sizeof(float) + (size_t)LDS_WAVES * 4 * STAGE * sizeof(int2);
        FIRA_REQUIRE(graph_rows > 0 && n_rows % graph_rows == 0 && lds <= 160 * 1024,
                     "csr_spmm: LDS variant needs rows-per-graph (%d) dividing n_rows and <= 512", graph_rows);
        static bool attr_set = false;
        if (!attr_set) {
            const hipError_t ae = hipFuncSetAttribute((const void*)spmm_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                      160 * 1024);
            FIRA_REQUIRE(ae == hipSuccess, "csr_spmm: cannot raise the dynamic LDS limit: %s", hipGetErrorString(ae));
            attr_set = true;
        }
        hipLaunchKernelGGL(spmm_lds_kernel, dim3(FIRA_D / SLAB, n_rows / graph_rows), dim3(LDS_WAVES * 64), lds, s, graph_rows,
                           rowptr, col, val, X, ldx, Y, ldy);
    } else {
        // grid rounded up to a multiple of 8 so that the XCD remap above is a bijection onto the row groups
        // FIRA_SPMM_ROWBATCH=0: the round-1 kernel, one row per wave (A/B switch); also for feature matrices of 2 GiB and more
        // (the batched kernel addresses X through a buffer descriptor)
        static const bool batch_off = [] { const char* e = getenv("FIRA_SPMM_ROWBATCH"); return e && e[0] == '0'; }();
        if (variant != 5 && !batch_off && (size_t)n_rows * ldx * 4 < (1ull << 31))
        {
            const dim3 grid(cdiv(cdiv(n_rows, 16), 8) * 8);
            if (accum) hipLaunchKernelGGL(spmm_rowbatch_kernel<true>, grid, dim3(256), 0, s, n_rows, rowptr, col, val, X, ldx, Y, ldy, rowsum);
            else hipLaunchKernelGGL(spmm_rowbatch_kernel<false>, grid, dim3(256), 0, s, n_rows, rowptr, col, val, X, ldx, Y, ldy, rowsum);
        }
        else
        hipLaunchKernelGGL(spmm_rowwave_kernel, dim3(cdiv(cdiv(n_rows, 4), 8) * 8), dim3(256), 0, s, n_rows, rowptr, col,
                           val, X, ldx, Y, ldy, accum, rowsum);
    }
    FIRA_CHECK_LAUNCH("csr_spmm");
    return 0;
}

}  // namespace fira

// Variant choice from the measured crossover (profiles/r3_spmm_crossover.md; 128 graphs x 512 nodes): the row-per-wave CSR
// kernel wins while a row gathers a handful of neighbours, then the LDS-slab CSR kernel, then the block-dense MFMA kernels
// (thresholds above: fraction of the N x N block that is non-zero); bf16 operands only where the caller's dtype says so.
extern "C" int fira_csr_spmm(void* stream, int n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col,
                             const float* val, const float* X, int ldx, float* Y, int ldy, int graph_rows, int variant,
                             int dtype) {
    FIRA_REQUIRE(dtype == FIRA_F32 || dtype == FIRA_BF16, "fira_csr_spmm: dtype must be FIRA_F32 or FIRA_BF16");
    if (variant == 0) {
        variant = 1;
        if (graph_rows > 0 && graph_rows <= 512 && n_rows > 0 && n_rows % graph_rows == 0) {
            const double density = (double)nnz / ((double)n_rows * graph_rows);
            if (dtype == FIRA_BF16 && density >= fira::SPMM_DENSE_AT_BF16) variant = 4;
            else if (density >= fira::SPMM_DENSE_AT_F32) variant = 3;
            else if (density >= fira::SPMM_LDS_AT) variant = 2;
        }
    } else if (variant == 3 && dtype == FIRA_BF16) {
        variant = 4;
    }
    return fira::csr_spmm((hipStream_t)stream, n_rows, rowptr, col, val, X, ldx, Y, ldy, graph_rows, variant);
}

extern "C" int fira_csr_spmm_f32(void* stream, int n_rows, const int32_t* rowptr, const int32_t* col,
                                 const float* val, const float* X, int ldx, float* Y, int ldy, int graph_rows,
                                 int variant) {
    return fira::csr_spmm((hipStream_t)stream, n_rows, rowptr, col, val, X, ldx, Y, ldy, graph_rows, variant);
}
