// Dual-copy output head of the reference (Model.py:15-20, 54-86):
//   copy score   s[b,t,j] = w . tanh(Ws mem[b,j] + Wt dec[b,t]) + c      (additive attention)
//   gate         g[b,t]   = softmax(Wp dec[b,t] + bp)                    (2-way)
//   p            = [ g0 * softmax(out_fc(dec)) ; g1 * softmax(mask(s, -1e9)) ]
//   loss         = -log(clamp(p[label], 1e-10, 1)),  label = tar_label shifted left, 0 = ignore
// The reference materialises a [B,30,370,256] tanh tensor (11.4 MB per commit, twice for backward; 44 % of
// its CPU step) and the [B,30,25020] probability tensor five times.  Here the tanh tile lives
// in registers (forward and re-computed in backward) and the loss kernel turns the logits into their
// gradient in place, one pass after the soft-max statistics.
#include "engine.h"
#include "adam_rows.h"
#include <stdlib.h>
#include "epilogue.h"

namespace fira {

constexpr int T_MAX = 32;

// tanh(x) = 1 - 2 / (1 + e^{2x}) on the fast exp / reciprocal units: |abs error| < 2e-7 over the whole range
// (saturates correctly: e^{2x} -> inf gives 1, -> 0 gives -1).  The copy score sums 256 such terms times w ~ 0.06.
__device__ __forceinline__ float tanh_fast(float x) {
    const float e = __expf(2.0f * x);
    return 1.0f - __fdividef(2.0f, 1.0f + e);
}

// ------------------------------------------------------------------------------------------------
// forward: grid (S chunks of 32 rows, B); 4 waves; one wave per memory row j, lanes hold 4 of the 256 dims.
__global__ __launch_bounds__(256) void copy_score_fwd_kernel(int T, int S, const float* __restrict__ src,
                                                             const float* __restrict__ tgt,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias,
                                                             float* __restrict__ score, int qpk,
                                                             const int32_t* __restrict__ mem_valid, int slots,
                                                             const int32_t* __restrict__ tar_label, int V,
                                                             const int32_t* __restrict__ t_off) {
    // t_off (optional, [B+1]): ragged target rows -- commit b's rows are t_off[b] .. t_off[b+1] of tgt / score (the
    // decoder's computed target rows, position t = row - t_off[b]); labels stay dense [B, T]
    __shared__ __attribute__((aligned(16))) float sm_tgt[T_MAX * FIRA_D];
    __shared__ unsigned sm_mask;
    const int b = blockIdx.y, t0 = threadIdx.x, lane = t0 & 63, wave = t0 >> 6;
    // training: only rows whose label is a copied token read their copy scores (the NLL takes one entry of
    // [gen ; copy], Model.py:80-86); the others are left unwritten and head_loss neither reads nor propagates them
    if (t0 == 0) sm_mask = tar_label ? 0u : 0xffffffffu;
    __syncthreads();
    const int tb = t_off ? t_off[b] : b * T;
    const int Tb = t_off ? t_off[b + 1] - tb : T;
    if (tar_label && t0 < Tb && t0 < T - 1 && tar_label[b * T + t0 + 1] >= V) atomicOr(&sm_mask, 1u << t0);
    src += (size_t)(b / qpk) * S * FIRA_D - (size_t)b * S * FIRA_D;     // qpk target batches share one memory
    const int32_t* mv = mem_valid ? mem_valid + (size_t)(b / qpk) * S : nullptr;
    for (int i = t0; i < Tb * (FIRA_D / 4); i += 256)
        reinterpret_cast<float4*>(sm_tgt)[i] = reinterpret_cast<const float4*>(tgt + (size_t)tb * FIRA_D)[i];
    __syncthreads();
    const unsigned mask = sm_mask;
    if (mask == 0u) return;
    const float4 w4 = *reinterpret_cast<const float4*>(w + lane * 4);
    const float c = bias[0];
    const int j_end = min(S, (int)(blockIdx.x + 1) * slots);
    for (int j = blockIdx.x * slots + wave; j < j_end; j += 4) {
        if (mv && mv[j] == 0) {                       // masked slot: its score is replaced by -1e9 downstream
            if (lane < Tb) score[((size_t)tb + lane) * S + j] = 0.f;
            continue;
        }
        const float4 s4 = *reinterpret_cast<const float4*>(src + ((size_t)b * S + j) * FIRA_D + lane * 4);
        for (int t = 0; t < Tb; ++t) {
            if (!((mask >> t) & 1u)) continue;
            const float4 x = *reinterpret_cast<const float4*>(&sm_tgt[t * FIRA_D + lane * 4]);
            float a = w4.x * tanh_fast(s4.x + x.x);
            a = fmaf(w4.y, tanh_fast(s4.y + x.y), a);
            a = fmaf(w4.z, tanh_fast(s4.z + x.z), a);
            a = fmaf(w4.w, tanh_fast(s4.w + x.w), a);
            a = wave_sum(a);
            if (lane == 0) score[((size_t)tb + t) * S + j] = a + c;
        }
    }
}

// backward (tanh re-computed): dsrc[b,j,:] = sum_t g*w*(1-th^2)   (owned by one wave: plain store)
//                              dtgt[b,t,:] += sum_j (same)        (registers -> LDS -> global atomics)
//                              dw += sum g*th ; dbias += sum g
// Only target rows whose label is a copied token carry a non-zero dscore row (Model.py:80-86: the NLL picks one entry
// of the [gen ; copy] distribution), typically a handful of the 30 positions: the workgroup stages its [T, slots]
// tile of dscore in LDS, derives the set of rows with any non-zero entry and spends tanh work only on those.
// Round 4: the active rows are processed in chunks of COPY_KA, so only COPY_KA target rows (and as many partial dtgt
// rows) live in LDS: 19 KB per workgroup instead of 68 KB.  The launch sits on the dependent chain right behind the fork
// of the vocabulary projection's weight gradient, whose workgroups (4 x 34 KB per CU) leave 21 KB of LDS free: at 68 KB a
// workgroup of this kernel could only be placed once TWO of them had left the same CU -- and the freed slot was refilled
// from the weight-gradient queue first (123 us in the step for ~20 us of work).
constexpr int COPY_KA = 8;
__global__ __launch_bounds__(256) void copy_score_bwd_kernel(int T, int S, const float* __restrict__ src,
                                                             const float* __restrict__ tgt,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ dscore,
                                                             float* __restrict__ dsrc, float* __restrict__ dtgt,
                                                             float* __restrict__ dw, float* __restrict__ dbias,
                                                             const int32_t* __restrict__ mem_valid, int slots,
                                                             float* __restrict__ part,
                                                             const int32_t* __restrict__ t_off) {
    constexpr int SLOTS_MAX = 16, SPW = SLOTS_MAX / 4;            // slots per wave
    __shared__ __attribute__((aligned(16))) float sm_tgt[COPY_KA * FIRA_D];
    __shared__ float sm_dt[COPY_KA * FIRA_D];
    __shared__ float sm_dw[FIRA_D + 1];
    __shared__ float sm_g[T_MAX * SLOTS_MAX];
    __shared__ unsigned sm_mask;
    const int b = blockIdx.y, t0 = threadIdx.x, lane = t0 & 63, wave = t0 >> 6;
    const int j0 = blockIdx.x * slots, j_end = min(S, j0 + slots);
    const int tb = t_off ? t_off[b] : b * T;                       // ragged target rows: see copy_score_fwd_kernel
    const int Tb = t_off ? t_off[b + 1] - tb : T;
    if (t0 == 0) sm_mask = 0u;
    __syncthreads();
    for (int i = t0; i < Tb * slots; i += 256) {
        const int t = i / slots, j = j0 + i % slots;
        float g = 0.f;
        if (j < j_end && !(mem_valid && mem_valid[(size_t)b * S + j] == 0)) g = dscore[((size_t)tb + t) * S + j];
        sm_g[t * SLOTS_MAX + i % slots] = g;
        if (g != 0.f) atomicOr(&sm_mask, 1u << t);
    }
    for (int i = t0; i < FIRA_D + 1; i += 256) sm_dw[i] = 0.f;
    __syncthreads();
    const unsigned mask = sm_mask;                                 // uniform: rows of this tile with any gradient
    float* const my_part = part ? part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * COPY_PART_STRIDE : nullptr;
    if (mask == 0u) {                                              // nothing flows into this tile's memory slots
        for (int j = j0 + wave; j < j_end; j += 4)
            *reinterpret_cast<float4*>(dsrc + ((size_t)b * S + j) * FIRA_D + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (my_part)
            for (int i = t0; i < FIRA_D + 1; i += 256) my_part[i] = 0.f;
        return;
    }
    const float4 w4 = *reinterpret_cast<const float4*>(w + lane * 4);
    const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
    // this wave's memory slots j0 + wave + 4 q: their source rows and gradient accumulators stay in registers for all chunks
    float4 s4[SPW];
    float ds[SPW][4];
    bool live[SPW];
#pragma unroll
    for (int q = 0; q < SPW; ++q) {
        const int j = j0 + wave + 4 * q;
        live[q] = j < j_end && !(mem_valid && mem_valid[(size_t)b * S + min(j, S - 1)] == 0);   // masked slot: no gradient through masked_fill
        s4[q] = live[q] ? *reinterpret_cast<const float4*>(src + ((size_t)b * S + j) * FIRA_D + lane * 4)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int e = 0; e < 4; ++e) ds[q][e] = 0.f;
    }
    float dwa[4] = {0.f, 0.f, 0.f, 0.f};
    float dba = 0.f;
    unsigned rem = mask;
    while (rem) {                                                  // block-uniform: chunks of up to COPY_KA active rows
        int tl[COPY_KA], nk = 0;
#pragma unroll
        for (int k = 0; k < COPY_KA; ++k) {
            tl[k] = 0;
            if (rem) { tl[k] = __ffs(rem) - 1; rem &= rem - 1; nk = k + 1; }
        }
#pragma unroll
        for (int k = 0; k < COPY_KA; ++k)
            if (k < nk && t0 < FIRA_D / 4)
                reinterpret_cast<float4*>(sm_tgt)[k * (FIRA_D / 4) + t0] =
                    reinterpret_cast<const float4*>(tgt + ((size_t)tb + tl[k]) * FIRA_D)[t0];
        for (int i = t0; i < COPY_KA * FIRA_D; i += 256) sm_dt[i] = 0.f;
        __syncthreads();
        float dt[COPY_KA][4];
#pragma unroll
        for (int k = 0; k < COPY_KA; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) dt[k][e] = 0.f;
#pragma unroll
        for (int q = 0; q < SPW; ++q) {
            if (!live[q]) continue;                                // wave-uniform
            const int jj = wave + 4 * q;
#pragma unroll
            for (int k = 0; k < COPY_KA; ++k) {
                if (k < nk) {
                    const float g = sm_g[tl[k] * SLOTS_MAX + jj];
                    const float4 x = *reinterpret_cast<const float4*>(&sm_tgt[k * FIRA_D + lane * 4]);
                    const float th[4] = {tanh_fast(s4[q].x + x.x), tanh_fast(s4[q].y + x.y), tanh_fast(s4[q].z + x.z),
                                         tanh_fast(s4[q].w + x.w)};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = g * wv[e] * (1.f - th[e] * th[e]);
                        ds[q][e] += u;
                        dt[k][e] += u;
                        dwa[e] = fmaf(g, th[e], dwa[e]);
                    }
                    dba += g;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < COPY_KA; ++k)
            if (k < nk) {
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(&sm_dt[k * FIRA_D + lane * 4 + e], dt[k][e]);
            }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < COPY_KA; ++k)
            if (k < nk) unsafeAtomicAdd(&dtgt[((size_t)tb + tl[k]) * FIRA_D + t0], sm_dt[k * FIRA_D + t0]);
        __syncthreads();                                           // sm_tgt / sm_dt are rewritten by the next chunk
    }
#pragma unroll
    for (int q = 0; q < SPW; ++q) {
        const int j = j0 + wave + 4 * q;
        if (j < j_end)
            *reinterpret_cast<float4*>(dsrc + ((size_t)b * S + j) * FIRA_D + lane * 4) = make_float4(ds[q][0], ds[q][1], ds[q][2], ds[q][3]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(&sm_dw[lane * 4 + e], dwa[e]);
    if (lane == 0) atomicAdd(&sm_dw[FIRA_D], dba);
    __syncthreads();
    if (my_part) {
        // deferred reduction (rowops.hip): B * S/16 workgroups adding to the same 257 addresses serialise in L2 (~50 ns per
        // same-address atomic: 1536 workgroups = ~75 us, most of this kernel's former run time)
        for (int i = t0; i < FIRA_D + 1; i += 256) my_part[i] = sm_dw[i];
        return;
    }
    for (int i = t0; i < FIRA_D; i += 256) unsafeAtomicAdd(&dw[i], sm_dw[i]);
    if (t0 == 0) unsafeAtomicAdd(dbias, sm_dw[FIRA_D]);
}

// ------------------------------------------------------------------------------------------------
// block-wide reductions for the loss kernel (256 threads)
__device__ __forceinline__ float block_max(float v, float* sm) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}
__device__ __forceinline__ float block_sum(float v, float* sm) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
// arg-max with first-occurrence tie-break (torch.argmax semantics)
__device__ __forceinline__ void block_argmax(float& v, int& idx, float* smv, int* smi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { smv[threadIdx.x >> 6] = v; smi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    v = smv[0]; idx = smi[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (smv[k] > v || (smv[k] == v && smi[k] < idx)) { v = smv[k]; idx = smi[k]; }
}

// one workgroup per (b,t) row.  REG: the row's V logits stay in registers (HL_PAIRS float2 per thread) between the
// soft-max statistics and the in-place gradient, so the [R,V] matrix is read once and written once.
constexpr int HL_PAIRS = 49;                     // 2 * 256 * 49 = 25088 >= the reference vocabulary (24650)
template <bool REG>
__global__ __launch_bounds__(256) void head_loss_kernel(int T, int V, int S, const int32_t* __restrict__ compact_row,
                                                        float* __restrict__ logits, int ldl,
                                                        float* __restrict__ score,
                                                        const int32_t* __restrict__ mem_valid,
                                                        float* __restrict__ gate_logits,
                                                        const int32_t* __restrict__ tar_label,
                                                        float* __restrict__ loss_sum, int32_t* __restrict__ n_tok,
                                                        int32_t* __restrict__ argmax_out, int want_grad,
                                                        const int32_t* __restrict__ row_bt) {
    __shared__ float smf[4];
    __shared__ int smi[4];
    // bt: row of score / gate_logits / compact_row (a computed target row when row_bt lists them: row_bt[bt] = its flat
    // b*T + t position; otherwise bt is that position itself)
    const int bt = blockIdx.x, tid = threadIdx.x;
    const int flat = row_bt ? row_bt[bt] : bt;
    const int b = flat / T, t = flat - b * T;
    const int crow = compact_row ? compact_row[bt] : bt;
    const int y = (t + 1 < T) ? tar_label[b * T + t + 1] : 0;        // label = cat(tar_label, 0)[:, 1:]
    float* lrow = crow >= 0 ? logits + (size_t)crow * ldl : nullptr;
    float* srow = score + (size_t)bt * S;
    const int32_t* mv = mem_valid + (size_t)b * S;

    // gate = softmax(z0, z1)
    const float z0 = gate_logits[2 * bt], z1 = gate_logits[2 * bt + 1];
    const float zm = fmaxf(z0, z1);
    const float e0 = expf(z0 - zm), e1 = expf(z1 - zm);
    const float g0 = e0 / (e0 + e1), g1 = e1 / (e0 + e1);

    // copy soft-max statistics over the S memory slots (-1e9 where masked); rows whose label is a generated token
    // only need them for the arg-max output (their copy scores may not even have been computed, see copy_score_fwd)
    float cmax = -INFINITY, csum = 1.f;
    int cidx = 0x7fffffff;
    if (argmax_out || (y >= V && y - V < S)) {
        for (int j = tid; j < S; j += 256) {
            const float x = mv[j] ? srow[j] : -1e9f;
            if (x > cmax) { cmax = x; cidx = j; }
        }
        block_argmax(cmax, cidx, smf, smi);
        csum = 0.f;
        for (int j = tid; j < S; j += 256) csum += expf((mv[j] ? srow[j] : -1e9f) - cmax);
        csum = block_sum(csum, smf);
    }

    // generator soft-max statistics over the V logits
    float gmax = -INFINITY, gsum = 1.f;
    int gidx = 0x7fffffff;
    float2 reg[REG ? HL_PAIRS : 1];
    const int n2 = V >> 1;
    if (lrow) {
        if (REG) {
            const float2* l2 = reinterpret_cast<const float2*>(lrow);
#pragma unroll
            for (int i = 0; i < HL_PAIRS; ++i) {
                const int j2 = tid + 256 * i;
                reg[i] = j2 < n2 ? l2[j2] : make_float2(-INFINITY, -INFINITY);
            }
#pragma unroll
            for (int i = 0; i < HL_PAIRS; ++i) {             // ascending index within the thread: first maximum wins
                const int j = 2 * (tid + 256 * i);
                if (reg[i].x > gmax) { gmax = reg[i].x; gidx = j; }
                if (reg[i].y > gmax) { gmax = reg[i].y; gidx = j + 1; }
            }
        } else {
            for (int j = tid; j < V; j += 256) {
                const float x = lrow[j];
                if (x > gmax) { gmax = x; gidx = j; }
            }
        }
        block_argmax(gmax, gidx, smf, smi);
        gsum = 0.f;
        if (REG) {
#pragma unroll
            for (int i = 0; i < HL_PAIRS; ++i) gsum += expf(reg[i].x - gmax) + expf(reg[i].y - gmax);
        } else {
            for (int j = tid; j < V; j += 256) gsum += expf(lrow[j] - gmax);
        }
        gsum = block_sum(gsum, smf);
    }

    if (argmax_out) {
        // teacher-forced argmax over [g0*p_gen ; g1*p_copy] (Model.py:85-86); first index wins ties
        const float pg = lrow ? g0 * (1.0f / gsum) : -1.f;
        const float pc = g1 * (1.0f / csum);
        if (tid == 0) argmax_out[bt] = (pg >= pc) ? gidx : V + cidx;
    }

    // loss of this row
    bool live = false, is_copy = false;
    float p = 1.f;
    if (y != 0) {
        if (y < V) {
            if (lrow) { p = g0 * (expf(lrow[y] - gmax) / gsum); live = true; }
        } else if (y - V < S) {
            is_copy = true;
            p = g1 * (expf((mv[y - V] ? srow[y - V] : -1e9f) - cmax) / csum);
            live = true;
        }
    }
    const bool pass = live && p >= 1e-10f && p <= 1.0f;     // clamp(min=1e-10,max=1) blocks the gradient outside
    if (live && tid == 0 && loss_sum) {
        const float pc = fminf(fmaxf(p, 1e-10f), 1.0f);
        unsafeAtomicAdd(loss_sum, -logf(pc));
        atomicAdd(n_tok, 1);
    }
    if (!want_grad) return;
    __syncthreads();      // every thread has read lrow[y] / srow[y-V] before they are overwritten

    // d loss / d gate logits:  g_k - [k == chosen branch]
    if (tid == 0) {
        gate_logits[2 * bt] = pass ? g0 - (is_copy ? 0.f : 1.f) : 0.f;
        gate_logits[2 * bt + 1] = pass ? g1 - (is_copy ? 1.f : 0.f) : 0.f;
    }
    // d loss / d copy scores (masked slots receive none: masked_fill)
    const bool copy_grad = pass && is_copy;
    const float inv_csum = 1.0f / csum;
    for (int j = tid; j < S; j += 256) {
        float d = 0.f;
        if (copy_grad && mv[j]) d = expf(srow[j] - cmax) * inv_csum - (j == y - V ? 1.f : 0.f);
        srow[j] = d;
    }
    // d loss / d logits, in place
    if (lrow) {
        const bool gen_grad = pass && !is_copy;
        const float inv_gsum = 1.0f / gsum;          // one reciprocal per row instead of a division per logit
        if (REG) {
            float2* l2 = reinterpret_cast<float2*>(lrow);
#pragma unroll
            for (int i = 0; i < HL_PAIRS; ++i) {
                const int j2 = tid + 256 * i;
                if (j2 < n2) {
                    float2 d = make_float2(0.f, 0.f);
                    if (gen_grad) {
                        d.x = expf(reg[i].x - gmax) * inv_gsum - (2 * j2 == y ? 1.f : 0.f);
                        d.y = expf(reg[i].y - gmax) * inv_gsum - (2 * j2 + 1 == y ? 1.f : 0.f);
                    }
                    l2[j2] = d;
                }
            }
        } else {
            for (int j = tid; j < V; j += 256) {
                float d = 0.f;
                if (gen_grad) d = expf(lrow[j] - gmax) * inv_gsum - (j == y ? 1.f : 0.f);
                lrow[j] = d;
            }
        }
    }
}

// Training-time variant of head_loss_kernel<true> (no arg-max output).  The full kernel spends ~35 VALU instructions per
// logit (two libm expf, first-maximum bookkeeping) and is instruction-bound, not memory-bound: 71 us for 530 rows, 76 us
// for 1 020 (four resident workgroups per CU either way).  Here the generator soft-max costs ONE exponential per logit:
// e = exp(x - max) replaces x in its register, is summed, and is scaled into the gradient.
// exp(x - m) = exp2(fma(x, log2 e, -m log2 e)): the rounding of m log2 e is a common factor of every term of the row and
// cancels in e / sum, the fma rounds x log2 e once (relative error <= 1.3e-6 at x - m = -40, < 1e-7 near the maximum).
__global__ __launch_bounds__(256) void head_loss_train_kernel(int T, int V, int S, const int32_t* __restrict__ compact_row,
                                                              float* __restrict__ logits, int ldl,
                                                              float* __restrict__ score,
                                                              const int32_t* __restrict__ mem_valid,
                                                              float* __restrict__ gate_logits,
                                                              const int32_t* __restrict__ tar_label,
                                                              float* __restrict__ loss_sum, int32_t* __restrict__ n_tok,
                                                              int want_grad, const int32_t* __restrict__ row_bt) {
    __shared__ float smf[4];
    constexpr float L2E = 1.4426950408889634f;
    const int bt = blockIdx.x, tid = threadIdx.x;
    const int flat = row_bt ? row_bt[bt] : bt;
    const int b = flat / T, t = flat - b * T;
    const int crow = compact_row ? compact_row[bt] : bt;
    const int y = (t + 1 < T) ? tar_label[b * T + t + 1] : 0;        // label = cat(tar_label, 0)[:, 1:]
    float* lrow = crow >= 0 ? logits + (size_t)crow * ldl : nullptr;
    float* srow = score + (size_t)bt * S;
    const int32_t* mv = mem_valid + (size_t)b * S;

    const float z0 = gate_logits[2 * bt], z1 = gate_logits[2 * bt + 1];
    const float zm = fmaxf(z0, z1);
    const float e0 = expf(z0 - zm), e1 = expf(z1 - zm);
    const float g0 = e0 / (e0 + e1), g1 = e1 / (e0 + e1);

    // copy soft-max statistics: only rows whose label is a copy slot need them
    float cmax = -INFINITY, csum = 1.f;
    const bool copy_row = y >= V && y - V < S;
    if (copy_row) {
        for (int j = tid; j < S; j += 256) cmax = fmaxf(cmax, mv[j] ? srow[j] : -1e9f);
        cmax = block_max(cmax, smf);
        csum = 0.f;
        for (int j = tid; j < S; j += 256) csum += expf((mv[j] ? srow[j] : -1e9f) - cmax);
        csum = block_sum(csum, smf);
    }

    // generator soft-max: the row's logits live in registers from here to the gradient store
    float2 reg[HL_PAIRS];
    const int n2 = V >> 1;
    float gsum = 1.f, c = 0.f, ly = 0.f;
    if (lrow) {
        const float2* l2 = reinterpret_cast<const float2*>(lrow);
#pragma unroll
        for (int i = 0; i < HL_PAIRS; ++i) {
            const int j2 = tid + 256 * i;
            reg[i] = j2 < n2 ? l2[j2] : make_float2(-INFINITY, -INFINITY);
        }
        if (y != 0 && y < V) ly = lrow[y];
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < HL_PAIRS; ++i) m = fmaxf(m, fmaxf(reg[i].x, reg[i].y));
        c = -block_max(m, smf) * L2E;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i < HL_PAIRS; ++i) {                 // exp2(-inf) = 0 for the slots past V
            reg[i].x = __builtin_amdgcn_exp2f(fmaf(reg[i].x, L2E, c));
            reg[i].y = __builtin_amdgcn_exp2f(fmaf(reg[i].y, L2E, c));
            s0 += reg[i].x;
            s1 += reg[i].y;
        }
        gsum = block_sum(s0 + s1, smf);
    }
    const float ey = __builtin_amdgcn_exp2f(fmaf(ly, L2E, c));          // the label's own term, as its register holds it

    bool live = false, is_copy = false;
    float p = 1.f;
    if (y != 0) {
        if (y < V) {
            if (lrow) { p = g0 * (ey / gsum); live = true; }
        } else if (copy_row) {
            is_copy = true;
            p = g1 * (expf((mv[y - V] ? srow[y - V] : -1e9f) - cmax) / csum);
            live = true;
        }
    }
    const bool pass = live && p >= 1e-10f && p <= 1.0f;     // clamp(min=1e-10,max=1) blocks the gradient outside
    if (live && tid == 0 && loss_sum) {
        const float pc = fminf(fmaxf(p, 1e-10f), 1.0f);
        unsafeAtomicAdd(loss_sum, -logf(pc));
        atomicAdd(n_tok, 1);
    }
    if (!want_grad) return;
    __syncthreads();      // every thread has read srow[y-V] before the scores are overwritten

    if (tid == 0) {
        gate_logits[2 * bt] = pass ? g0 - (is_copy ? 0.f : 1.f) : 0.f;
        gate_logits[2 * bt + 1] = pass ? g1 - (is_copy ? 1.f : 0.f) : 0.f;
    }
    const bool copy_grad = pass && is_copy;
    const float inv_csum = 1.0f / csum;
    for (int j = tid; j < S; j += 256) {
        float d = 0.f;
        if (copy_grad && mv[j]) d = expf(srow[j] - cmax) * inv_csum - (j == y - V ? 1.f : 0.f);
        srow[j] = d;
    }
    if (lrow) {
        const bool gen_grad = pass && !is_copy;
        const float inv_gsum = 1.0f / gsum;
        const float scale = gen_grad ? inv_gsum : 0.f;       // e is finite: rows without a generator gradient store zeros
        float2* l2 = reinterpret_cast<float2*>(lrow);
#pragma unroll
        for (int i = 0; i < HL_PAIRS; ++i) {
            const int j2 = tid + 256 * i;
            if (j2 < n2) l2[j2] = make_float2(reg[i].x * scale, reg[i].y * scale);
        }
        // "- [j == y]": the thread that stored the label's pair rewrites that one element (same thread, same address:
        // program order holds)
        if (gen_grad && tid == ((y >> 1) & 255)) lrow[y] = ey * inv_gsum - 1.f;
    }
}

// ------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam defaults, run_model.py:396), one fused pass over the flat parameter buffer.
__global__ __launch_bounds__(256) void adam_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, float lr,
                                                   float beta1, float beta2, float eps, float bc1, float bc2_sqrt,
                                                   const float* __restrict__ scale_ptr, int scale_is_count) {
    // scale_is_count: *scale_ptr is a (global, all-reduced) token count; the normaliser 1 / max(count, 1) is formed here
    const float sv = scale_ptr ? *scale_ptr : 1.0f;
    const float scale = scale_is_count ? 1.0f / fmaxf(sv, 1.0f) : sv;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        adam_elem(p[i], m[i], v[i], g[i] * scale, beta1, beta2, eps, step_size, bc2_sqrt);
}

// The same step for micro-batched training: the gradient is g0 (+ g1), each the loss_SUM gradient of one micro-batch
// computed concurrently on its own stream, and the normaliser 1 / max(n_tok0 (+ n_tok1), 1) of run_model.py:105 is formed
// here from the device counters (no separate launch, no host sync).
__global__ __launch_bounds__(256) void adam_mb_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g0,
                                                      const float* __restrict__ g1, float* __restrict__ m,
                                                      float* __restrict__ v, float lr, float beta1, float beta2, float eps,
                                                      float bc1, float bc2_sqrt, const int32_t* __restrict__ n0,
                                                      const int32_t* __restrict__ n1) {
    const int nt = *n0 + (n1 ? *n1 : 0);
    const float scale = 1.0f / (float)(nt > 0 ? nt : 1);
    const int64_t stride = (int64_t)gridDim.x * 256;
    const float step_size = lr / bc1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        adam_elem(p[i], m[i], v[i], (g1 ? g0[i] + g1[i] : g0[i]) * scale, beta1, beta2, eps, step_size, bc2_sqrt);
}

// ------------------------------------------------------------------------------------------------
// Row-sparse form of the same update for the two vocabulary-sized embedding tables (round 6).  A row of an embedding table
// whose gradient row is all zero still moves under Adam (its moments decay, the parameter follows the decayed first moment),
// but that update is a function of (p, m, v, step) alone: it can be applied LATER, in registers, bit for bit, when the row is
// next needed -- by a forward pass that gathers it (adam_rows_catchup_kernel, over the batch's token ids) or by a step whose
// gradient row is not zero (adam_rows_kernel).  last[r] = the step up to which row r's (p, m, v) in memory are current.  A
// training step then moves 28 bytes per parameter only for the rows its batch touched (12 % / 1.5 % of the two tables at
// batch 32) instead of for all 2 x 6.3 M of them: -330 MB of the 780 MB the dense update moves.  Every ADAM_ROWS_K-th step
// updates every row (force), so a catch-up never spans more than ADAM_ROWS_K - 1 steps and the bias corrections of the
// skipped steps fit the kernel's argument block (AdamRowsHist: host-computed exactly as adam_step_mb computes its own).
// lr / beta / eps must not change inside a window without a sync (fira_adam_rows_sync).  One wave per 256-float row.
// gz: a zero the compiler cannot see through (the skipped updates must run the instruction sequence of the dense kernel)
__global__ __launch_bounds__(256) void adam_rows_kernel(AdamRowsTables tb, const float* __restrict__ gbase, float lr,
                                                        float beta1, float beta2, float eps, int step, AdamRowsHist h,
                                                        const int32_t* __restrict__ n0, const float* __restrict__ count,
                                                        int force, float gz, int it_lo, int it_hi) {
    float scale;
    if (count) scale = 1.0f / fmaxf(*count, 1.0f);
    else { const int nt = *n0; scale = 1.0f / (float)(nt > 0 ? nt : 1); }
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * 4;
    const int total = it_hi;                              // rows it_lo .. it_hi of the two tables' combined row space
    const float ss = lr / h.bc1[step % ADAM_ROWS_K], b2s = h.bc2s[step % ADAM_ROWS_K];
    // four gradient rows requested per trip (the launch is a stream of 1 KB reads with a wave-uniform skip: one row per trip
    // left it latency-bound at 2.5 TB/s)
    for (int it0 = it_lo + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4; it0 < total; it0 += nw * 4) {
        float4 gq[4];
        size_t oq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int it = min(it0 + u, total - 1);
            const int t = it >= tb.rows[0] ? 1 : 0;
            oq[u] = (size_t)tb.off[t] + (size_t)(it - (t ? tb.rows[0] : 0)) * 256 + lane * 4;
            gq[u] = *reinterpret_cast<const float4*>(gbase + oq[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int it = it0 + u;
            if (it >= total) break;
            const float4 gv = gq[u];
            const size_t o = oq[u];
            const bool nz = gv.x != 0.f || gv.y != 0.f || gv.z != 0.f || gv.w != 0.f;
            if (!force && !__any(nz)) continue;           // wave-uniform: the row waits for its next reader
            const int l = max(tb.last[it], step - ADAM_ROWS_K);
            float4 pv = *reinterpret_cast<float4*>(tb.p + o), mv = *reinterpret_cast<float4*>(tb.m + o),
                   vv = *reinterpret_cast<float4*>(tb.v + o);
            adam_row_zero_steps(pv, mv, vv, l + 1, step - 1, gz, lr, beta1, beta2, eps, h);
            adam_elem(pv.x, mv.x, vv.x, gv.x * scale, beta1, beta2, eps, ss, b2s);
            adam_elem(pv.y, mv.y, vv.y, gv.y * scale, beta1, beta2, eps, ss, b2s);
            adam_elem(pv.z, mv.z, vv.z, gv.z * scale, beta1, beta2, eps, ss, b2s);
            adam_elem(pv.w, mv.w, vv.w, gv.w * scale, beta1, beta2, eps, ss, b2s);
            *reinterpret_cast<float4*>(tb.p + o) = pv;
            *reinterpret_cast<float4*>(tb.m + o) = mv;
            *reinterpret_cast<float4*>(tb.v + o) = vv;
            if (lane == 0) tb.last[it] = step;
        }
    }
}
// Rows brought up to step `to` (zero-gradient updates only): the rows listed in the id arrays of `ls` (duplicates and
// out-of-range ids allowed: a row is claimed through atomicMax on last[]), or, with no list, every row of both tables.
__global__ __launch_bounds__(256) void adam_rows_catchup_kernel(AdamRowsTables tb, AdamRowsLists ls, float lr, float beta1,
                                                                float beta2, float eps, int to, AdamRowsHist h, float gz) {
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * 4;
    const int total = ls.n_lists ? ls.end[ls.n_lists - 1] : tb.rows[0] + tb.rows[1];
    for (int it = blockIdx.x * 4 + (threadIdx.x >> 6); it < total; it += nw) {
        int t, r, l;
        if (ls.n_lists) {
            int k = 0;
            while (it >= ls.end[k]) ++k;
            t = ls.table[k];
            r = ls.ids[k][it - (k ? ls.end[k - 1] : 0)];
            if (r < 0 || r >= tb.rows[t]) continue;
            int32_t* last = tb.last + (t ? tb.rows[0] : 0);
            l = 0;
            if (lane == 0) l = atomicMax(&last[r], to);
            l = __builtin_amdgcn_readfirstlane(l);
        } else {
            t = it >= tb.rows[0] ? 1 : 0;
            r = it - (t ? tb.rows[0] : 0);
            l = tb.last[it];
        }
        if (l >= to) continue;
        l = max(l, to - (ADAM_ROWS_K - 1));
        const size_t o = (size_t)tb.off[t] + (size_t)r * 256 + lane * 4;
        float4 pv = *reinterpret_cast<float4*>(tb.p + o), mv = *reinterpret_cast<float4*>(tb.m + o),
               vv = *reinterpret_cast<float4*>(tb.v + o);
        adam_row_zero_steps(pv, mv, vv, l + 1, to, gz, lr, beta1, beta2, eps, h);
        *reinterpret_cast<float4*>(tb.p + o) = pv;
        *reinterpret_cast<float4*>(tb.m + o) = mv;
        *reinterpret_cast<float4*>(tb.v + o) = vv;
        if (!ls.n_lists && lane == 0) tb.last[it] = to;
    }
}

// out2 = {loss_sum, (float) n_tok}: the two scalars a data-parallel step all-reduces, as one fp32 pair (exact below 2^24 tokens)
__global__ void pack_stats_kernel(const float* __restrict__ loss_sum, const int32_t* __restrict__ n_tok, float* __restrict__ out2) {
    out2[0] = *loss_sum;
    out2[1] = (float)*n_tok;
}
// inv_scale[0] = 1 / max(n_tok, 1): the token-count normaliser of run_model.py:105 without a host sync
__global__ void inv_count_kernel(const int32_t* __restrict__ n_tok, float* __restrict__ out) {
    const int n = *n_tok;
    *out = 1.0f / (float)(n > 0 ? n : 1);
}


// ------------------------------------------------------------------------------------------------
// Decode-time output distribution (run_model.py:257-265): dist[r,:] = [g0*softmax(logits) ; g1*softmax(mask(score))]
// for one position per row, plus the arg-max entry (greedy fast path).  One workgroup per row.
__global__ __launch_bounds__(256) void decode_dist_kernel(int V, int S, const float* __restrict__ logits, int ldl,
                                                          const float* __restrict__ score,
                                                          const int32_t* __restrict__ mem_valid, int qpk,
                                                          const float* __restrict__ gate_logits,
                                                          float* __restrict__ dist, int32_t* __restrict__ best_id,
                                                          float* __restrict__ best_p) {
    __shared__ float smf[4];
    __shared__ int smi[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    const float* lrow = logits + (size_t)r * ldl;
    const float* srow = score + (size_t)r * S;
    const int32_t* mv = mem_valid + (size_t)(r / qpk) * S;
    const float z0 = gate_logits[2 * r], z1 = gate_logits[2 * r + 1];
    const float zm = fmaxf(z0, z1);
    const float e0 = expf(z0 - zm), e1 = expf(z1 - zm);
    const float g0 = e0 / (e0 + e1), g1 = e1 / (e0 + e1);
    float cmax = -INFINITY, gmax = -INFINITY;
    int cidx = 0x7fffffff, gidx = 0x7fffffff;
    for (int j = tid; j < S; j += 256) {
        const float x = mv[j] ? srow[j] : -1e9f;
        if (x > cmax) { cmax = x; cidx = j; }
    }
    block_argmax(cmax, cidx, smf, smi);
    float csum = 0.f;
    for (int j = tid; j < S; j += 256) csum += expf((mv[j] ? srow[j] : -1e9f) - cmax);
    csum = block_sum(csum, smf);
    for (int j = tid; j < V; j += 256) {
        const float x = lrow[j];
        if (x > gmax) { gmax = x; gidx = j; }
    }
    block_argmax(gmax, gidx, smf, smi);
    float gsum = 0.f;
    for (int j = tid; j < V; j += 256) gsum += expf(lrow[j] - gmax);
    gsum = block_sum(gsum, smf);
    if (dist) {
        float* drow = dist + (size_t)r * (V + S);
        const float sg = g0 * (1.0f / gsum), sc = g1 * (1.0f / csum);
        for (int j = tid; j < V; j += 256) drow[j] = sg * expf(lrow[j] - gmax);
        for (int j = tid; j < S; j += 256) drow[V + j] = sc * expf((mv[j] ? srow[j] : -1e9f) - cmax);
    }
    if (tid == 0 && best_id) {
        const float pg = g0 * (1.0f / gsum), pc = g1 * (1.0f / csum);
        best_id[r] = pg >= pc ? gidx : V + cidx;
        if (best_p) best_p[r] = pg >= pc ? pg : pc;
    }
}
// Same arithmetic, same per-thread element order (so every sum and every arg-max is bit-identical to the kernel above), but
// the row's V logits are requested ONCE, all together, and stay in registers: the loop version walks the row three times
// with one dependent 4-byte load per iteration (97 iterations x 3 passes = 51 us per step at V = 24 650; this one: one
// round trip).  DD_NPT * 256 >= V.
constexpr int DD_NPT = 100;
__global__ __launch_bounds__(256) void decode_dist_reg_kernel(int V, int S, const float* __restrict__ logits, int ldl,
                                                              const float* __restrict__ score,
                                                              const int32_t* __restrict__ mem_valid, int qpk,
                                                              const float* __restrict__ gate_logits,
                                                              float* __restrict__ dist, int32_t* __restrict__ best_id,
                                                              float* __restrict__ best_p) {
    __shared__ float smf[4];
    __shared__ int smi[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    const rsrc_t rL = buf_rsrc(logits + (size_t)r * ldl, (unsigned)V * 4u);
    float x[DD_NPT];
#pragma unroll
    for (int i = 0; i < DD_NPT; ++i) x[i] = buf_load_f32(rL, (unsigned)(tid + 256 * i) * 4u);       // past V: 0, replaced below
    const float* srow = score + (size_t)r * S;
    const int32_t* mv = mem_valid + (size_t)(r / qpk) * S;
    const float z0 = gate_logits[2 * r], z1 = gate_logits[2 * r + 1];
    const float zm = fmaxf(z0, z1);
    const float e0 = expf(z0 - zm), e1 = expf(z1 - zm);
    const float g0 = e0 / (e0 + e1), g1 = e1 / (e0 + e1);
    float cmax = -INFINITY, gmax = -INFINITY;
    int cidx = 0x7fffffff, gidx = 0x7fffffff;
    for (int j = tid; j < S; j += 256) {
        const float v = mv[j] ? srow[j] : -1e9f;
        if (v > cmax) { cmax = v; cidx = j; }
    }
    block_argmax(cmax, cidx, smf, smi);
    float csum = 0.f;
    for (int j = tid; j < S; j += 256) csum += expf((mv[j] ? srow[j] : -1e9f) - cmax);
    csum = block_sum(csum, smf);
#pragma unroll
    for (int i = 0; i < DD_NPT; ++i) {                       // ascending index within the thread: first maximum wins
        const int j = tid + 256 * i;
        x[i] = j < V ? x[i] : -INFINITY;
        if (x[i] > gmax) { gmax = x[i]; gidx = j; }
    }
    block_argmax(gmax, gidx, smf, smi);
    float gsum = 0.f;
#pragma unroll
    for (int i = 0; i < DD_NPT; ++i) {
        x[i] = expf(x[i] - gmax);                            // exp(-inf) = 0 past V
        gsum += (tid + 256 * i < V) ? x[i] : 0.f;
    }
    gsum = block_sum(gsum, smf);
    if (dist) {
        float* drow = dist + (size_t)r * (V + S);
        const float sg = g0 * (1.0f / gsum), sc = g1 * (1.0f / csum);
        const rsrc_t rD = buf_rsrc(drow, (unsigned)V * 4u);
#pragma unroll
        for (int i = 0; i < DD_NPT; ++i)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sg * x[i]), rD, (unsigned)(tid + 256 * i) * 4u, 0, 0);
        for (int j = tid; j < S; j += 256) drow[V + j] = sc * expf((mv[j] ? srow[j] : -1e9f) - cmax);
    }
    if (tid == 0 && best_id) {
        const float pg = g0 * (1.0f / gsum), pc = g1 * (1.0f / csum);
        best_id[r] = pg >= pc ? gidx : V + cidx;
        if (best_p) best_p[r] = pg >= pc ? pg : pc;
    }
}
// The decode loop's form: 1024 threads per hypothesis row (16 waves: the 64 rows of a greedy batch are 64 workgroups, and
// with 256 threads each a CU ran one wave per SIMD behind 100 dependent-latency loads), the row's logits requested once
// and kept in registers (25 per thread), and the 2-way gate LinearProb(x) formed here from the decoder row instead of
// by its own [64 x 2 x 256] product launch.  First-occurrence arg-max as torch.argmax.
constexpr int DDW_NT = 1024, DDW_NPT = 25;
__device__ __forceinline__ float block16_sum(float v, float* sm) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < DDW_NT / 64; ++k) t += sm[k];
    return t;
}
__device__ __forceinline__ void block16_argmax(float& v, int& idx, float* smv, int* smi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { smv[threadIdx.x >> 6] = v; smi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    v = smv[0]; idx = smi[0];
#pragma unroll
    for (int k = 1; k < DDW_NT / 64; ++k)
        if (smv[k] > v || (smv[k] == v && smi[k] < idx)) { v = smv[k]; idx = smi[k]; }
}
__global__ __launch_bounds__(DDW_NT) void decode_dist_wide_kernel(int V, int S, const float* __restrict__ logits, int ldl,
                                                                  const float* __restrict__ score,
                                                                  const int32_t* __restrict__ mem_valid, int qpk,
                                                                  const float* __restrict__ gate_logits,
                                                                  const float* __restrict__ xrow,
                                                                  const float* __restrict__ wp,
                                                                  const float* __restrict__ bp, float* __restrict__ dist,
                                                                  int32_t* __restrict__ best_id, float* __restrict__ best_p) {
    __shared__ float smf[DDW_NT / 64];
    __shared__ int smi[DDW_NT / 64];
    const int r = blockIdx.x, tid = threadIdx.x;
    const rsrc_t rL = buf_rsrc(logits + (size_t)r * ldl, (unsigned)V * 4u);
    float x[DDW_NPT];
#pragma unroll
    for (int i = 0; i < DDW_NPT; ++i) x[i] = buf_load_f32(rL, (unsigned)(tid + DDW_NT * i) * 4u);   // past V: 0, replaced below
    const float* srow = score + (size_t)r * S;
    const int32_t* mv = mem_valid + (size_t)(r / qpk) * S;
    float z0, z1;
    if (gate_logits) {
        z0 = gate_logits[2 * r]; z1 = gate_logits[2 * r + 1];
    } else {                                                   // gate = x wp^T + bp: two 256-long dot products
        const float xv = tid < FIRA_D ? xrow[(size_t)r * FIRA_D + tid] : 0.f;
        const float a0 = tid < FIRA_D ? xv * wp[tid] : 0.f, a1 = tid < FIRA_D ? xv * wp[FIRA_D + tid] : 0.f;
        z0 = block16_sum(a0, smf) + bp[0];
        z1 = block16_sum(a1, smf) + bp[1];
    }
    const float zm = fmaxf(z0, z1);
    const float e0 = expf(z0 - zm), e1 = expf(z1 - zm);
    const float g0 = e0 / (e0 + e1), g1 = e1 / (e0 + e1);
    float cmax = -INFINITY, gmax = -INFINITY;
    int cidx = 0x7fffffff, gidx = 0x7fffffff;
    const float sv = tid < S ? (mv[tid] ? srow[tid] : -1e9f) : -INFINITY;       // S <= 1024: one slot per thread
    if (tid < S) { cmax = sv; cidx = tid; }
    block16_argmax(cmax, cidx, smf, smi);
    const float ce = tid < S ? expf(sv - cmax) : 0.f;
    const float csum = block16_sum(ce, smf);
#pragma unroll
    for (int i = 0; i < DDW_NPT; ++i) {                      // ascending index within the thread: first maximum wins
        const int j = tid + DDW_NT * i;
        x[i] = j < V ? x[i] : -INFINITY;
        if (x[i] > gmax) { gmax = x[i]; gidx = j; }
    }
    block16_argmax(gmax, gidx, smf, smi);
    float gsum = 0.f;
#pragma unroll
    for (int i = 0; i < DDW_NPT; ++i) {
        x[i] = expf(x[i] - gmax);                            // exp(-inf) = 0 past V
        gsum += x[i];
    }
    gsum = block16_sum(gsum, smf);
    if (dist) {
        float* drow = dist + (size_t)r * (V + S);
        const float sg = g0 * (1.0f / gsum), sc = g1 * (1.0f / csum);
        const rsrc_t rD = buf_rsrc(drow, (unsigned)V * 4u);
#pragma unroll
        for (int i = 0; i < DDW_NPT; ++i)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sg * x[i]), rD, (unsigned)(tid + DDW_NT * i) * 4u, 0, 0);
        if (tid < S) drow[V + tid] = sc * ce;
    }
    if (tid == 0 && best_id) {
        const float pg = g0 * (1.0f / gsum), pc = g1 * (1.0f / csum);
        best_id[r] = pg >= pc ? gidx : V + cidx;
        if (best_p) best_p[r] = pg >= pc ? pg : pc;
    }
}
int decode_dist(hipStream_t s, int R, int V, int S, const float* logits, int ldl, const float* score,
                const int32_t* mem_valid, int qpk, const float* gate_logits, float* dist, int32_t* best_id,
                float* best_p, const float* x, const float* wp, const float* bp) {
    ProfScope prof(s, PROF_HEAD, 0.0);
    if (R <= 0) return 0;
    FIRA_REQUIRE(gate_logits || (x && wp && bp), "decode_dist: needs the gate logits or the rows / weights to form them");
    if (V <= DDW_NPT * DDW_NT && S <= DDW_NT) {      // the 1024-thread kernel wherever the row fits its registers
        hipLaunchKernelGGL(decode_dist_wide_kernel, dim3(R), dim3(DDW_NT), 0, s, V, S, logits, ldl, score, mem_valid, qpk,
                           gate_logits, x, wp, bp, dist, best_id, best_p);
        FIRA_CHECK_LAUNCH("decode_dist");
        return 0;
    }
    FIRA_REQUIRE(gate_logits, "decode_dist: vocabulary %d / %d memory slots need precomputed gate logits", V, S);
    if (V <= DD_NPT * 256)
        hipLaunchKernelGGL(decode_dist_reg_kernel, dim3(R), dim3(256), 0, s, V, S, logits, ldl, score, mem_valid, qpk,
                           gate_logits, dist, best_id, best_p);
    else
    hipLaunchKernelGGL(decode_dist_kernel, dim3(R), dim3(256), 0, s, V, S, logits, ldl, score, mem_valid, qpk,
                       gate_logits, dist, best_id, best_p);
    FIRA_CHECK_LAUNCH("decode_dist");
    return 0;
}

int copy_score_fwd_ex(hipStream_t s, int B, int T, int S, const float* src, const float* tgt, const float* w,
                      const float* bias, float* score, int qpk, const int32_t* mem_valid, const int32_t* tar_label,
                      int V, const int32_t* t_off) {
    ProfScope prof(s, PROF_COPY, 0.0);
    if (B <= 0) return 0;
    FIRA_REQUIRE(T <= T_MAX && qpk >= 1, "copy_score_fwd: T=%d > %d", T, T_MAX);
    const int slots = 8;          // memory slots per workgroup: small chunks so that masked stretches cost nothing
    hipLaunchKernelGGL(copy_score_fwd_kernel, dim3(cdiv(S, slots), B), dim3(256), 0, s, T, S, src, tgt, w, bias, score,
                       qpk, mem_valid, slots, tar_label, V, t_off);
    FIRA_CHECK_LAUNCH("copy_score_fwd");
    return 0;
}
int copy_score_fwd(hipStream_t s, int B, int T, int S, const float* src, const float* tgt, const float* w,
                   const float* bias, float* score) {
    return copy_score_fwd_ex(s, B, T, S, src, tgt, w, bias, score, 1, nullptr, nullptr, 0, nullptr);
}
int copy_score_bwd_blocks(int B, int S) { return B > 0 ? cdiv(S, 16) * B : 0; }
int copy_score_bwd_ex(hipStream_t s, int B, int T, int S, const float* src, const float* tgt, const float* w,
                      const float* dscore, float* dsrc, float* dtgt, float* dw, float* dbias, const int32_t* mem_valid,
                      float* part, const int32_t* t_off) {
    ProfScope prof(s, PROF_COPY, 0.0);
    if (B <= 0) return 0;
    FIRA_REQUIRE(T <= T_MAX, "copy_score_bwd: T=%d > %d", T, T_MAX);
    const int slots = 16;                                   // == SLOTS_MAX of the kernel
    hipLaunchKernelGGL(copy_score_bwd_kernel, dim3(cdiv(S, slots), B), dim3(256), 0, s, T, S, src, tgt, w, dscore, dsrc,
                       dtgt, dw, dbias, mem_valid, slots, part, t_off);
    FIRA_CHECK_LAUNCH("copy_score_bwd");
    return 0;
}
int copy_score_bwd(hipStream_t s, int B, int T, int S, const float* src, const float* tgt, const float* w,
                   const float* dscore, float* dsrc, float* dtgt, float* dw, float* dbias) {
    return copy_score_bwd_ex(s, B, T, S, src, tgt, w, dscore, dsrc, dtgt, dw, dbias, nullptr, nullptr, nullptr);
}
int head_loss(hipStream_t s, int BT, int T, int V, int S, const int32_t* compact_row, float* logits, int ldl,
              float* score, const int32_t* mem_valid, float* gate_logits, const int32_t* tar_label, float* loss_sum,
              int32_t* n_tok, int32_t* argmax_out, int want_grad, const int32_t* row_bt) {
    ProfScope prof(s, PROF_HEAD, 0.0);
    if (BT <= 0) return 0;
    const bool reg = (V % 2 == 0) && (V / 2 <= 256 * HL_PAIRS) && (ldl % 2 == 0) && ((uintptr_t)logits % 8 == 0);
    if (reg && !argmax_out)
        hipLaunchKernelGGL(head_loss_train_kernel, dim3(BT), dim3(256), 0, s, T, V, S, compact_row, logits, ldl, score,
                           mem_valid, gate_logits, tar_label, loss_sum, n_tok, want_grad, row_bt);
    else if (reg)
        hipLaunchKernelGGL(head_loss_kernel<true>, dim3(BT), dim3(256), 0, s, T, V, S, compact_row, logits, ldl, score,
                           mem_valid, gate_logits, tar_label, loss_sum, n_tok, argmax_out, want_grad, row_bt);
    else
        hipLaunchKernelGGL(head_loss_kernel<false>, dim3(BT), dim3(256), 0, s, T, V, S, compact_row, logits, ldl, score,
                           mem_valid, gate_logits, tar_label, loss_sum, n_tok, argmax_out, want_grad, row_bt);
    FIRA_CHECK_LAUNCH("head_loss");
    return 0;
}
int adam_step(hipStream_t s, int64_t n, float* p, const float* g, float* m, float* v, float lr, float beta1,
              float beta2, float eps, int step, const float* scale_ptr, int scale_is_count) {
    ProfScope prof(s, PROF_ADAM, 0.0);
    if (n <= 0) return 0;
    FIRA_REQUIRE(step >= 1, "adam_step: step must start at 1");
    const double bc1 = 1.0 - pow((double)beta1, step);
    const double bc2 = 1.0 - pow((double)beta2, step);
    const int grid = (int)std::min<int64_t>(cdiv64(n, 256), 256 * 16);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, s, n, p, g, m, v, lr, beta1, beta2, eps, (float)bc1,
                       (float)sqrt(bc2), scale_ptr, scale_is_count);
    FIRA_CHECK_LAUNCH("adam_step");
    return 0;
}
int adam_step_mb(hipStream_t s, int64_t n, float* p, const float* g0, const float* g1, float* m, float* v, float lr,
                 float beta1, float beta2, float eps, int step, const int32_t* n0, const int32_t* n1) {
    ProfScope prof(s, PROF_ADAM, 0.0);
    if (n <= 0) return 0;
    const double bc1 = 1.0 - pow((double)beta1, step);
    const double bc2 = 1.0 - pow((double)beta2, step);
    const int grid = (int)std::min<int64_t>(cdiv64(n, 256), 256 * 16);
    hipLaunchKernelGGL(adam_mb_kernel, dim3(grid), dim3(256), 0, s, n, p, g0, g1, m, v, lr, beta1, beta2, eps, (float)bc1,
                       (float)sqrt(bc2), n0, n1);
    FIRA_CHECK_LAUNCH("adam_step_mb");
    return 0;
}
static AdamRowsHist adam_rows_hist(float beta1, float beta2, int end) {
    AdamRowsHist h;
    for (int k = 0; k < ADAM_ROWS_K; ++k) h.bc1[k] = h.bc2s[k] = 1.f;
    for (int j = std::max(1, end - ADAM_ROWS_K + 1); j <= end; ++j) {       // as adam_step_mb forms them for step j
        h.bc1[j % ADAM_ROWS_K] = (float)(1.0 - pow((double)beta1, j));
        h.bc2s[j % ADAM_ROWS_K] = (float)sqrt(1.0 - pow((double)beta2, j));
    }
    return h;
}
AdamRowsView adam_rows_view(const AdamRowsTables& tb, int table, float lr, float beta1, float beta2, float eps, int to) {
    AdamRowsView vw;
    if (to < 1) return vw;                                   // nothing owed ahead of step 1
    vw.m = tb.m + tb.off[table]; vw.v = tb.v + tb.off[table];
    vw.last = tb.last + (table ? tb.rows[0] : 0);
    vw.to = to; vw.lr = lr; vw.beta1 = beta1; vw.beta2 = beta2; vw.eps = eps; vw.gz = 0.f;
    vw.h = adam_rows_hist(beta1, beta2, to);
    return vw;
}
int adam_rows_step(hipStream_t s, const AdamRowsTables& tb, const float* g, float lr, float beta1, float beta2, float eps,
                   int step, const int32_t* n0, const float* count, int tables) {
    ProfScope prof(s, PROF_ADAM, 0.0);
    FIRA_REQUIRE(step >= 1 && tb.last && (n0 || count), "adam_rows_step: bad argument");
    const int it_lo = (tables & 1) ? 0 : tb.rows[0], it_hi = (tables & 2) ? tb.rows[0] + tb.rows[1] : tb.rows[0];
    const int total = it_hi - it_lo;
    if (total <= 0) return 0;
    const int grid = std::min(cdiv(total, 16), 256 * 8);
    hipLaunchKernelGGL(adam_rows_kernel, dim3(grid), dim3(256), 0, s, tb, g, lr, beta1, beta2, eps, step,
                       adam_rows_hist(beta1, beta2, step), n0, count, step % ADAM_ROWS_K == 0 ? 1 : 0, 0.0f, it_lo, it_hi);
    FIRA_CHECK_LAUNCH("adam_rows_step");
    return 0;
}
int adam_rows_catchup(hipStream_t s, const AdamRowsTables& tb, const AdamRowsLists* ls, float lr, float beta1, float beta2,
                      float eps, int to) {
    ProfScope prof(s, PROF_ADAM, 0.0);
    FIRA_REQUIRE(tb.last, "adam_rows_catchup: bad argument");
    if (to < 1) return 0;
    AdamRowsLists none{};
    const AdamRowsLists& l = ls ? *ls : none;
    const int total = l.n_lists ? l.end[l.n_lists - 1] : tb.rows[0] + tb.rows[1];
    if (total <= 0) return 0;
    const int grid = std::min(cdiv(total, 4), 256 * 8);
    hipLaunchKernelGGL(adam_rows_catchup_kernel, dim3(grid), dim3(256), 0, s, tb, l, lr, beta1, beta2, eps, to,
                       adam_rows_hist(beta1, beta2, to), 0.0f);
    FIRA_CHECK_LAUNCH("adam_rows_catchup");
    return 0;
}
int inv_count(hipStream_t s, const int32_t* n_tok, float* out) {
    hipLaunchKernelGGL(inv_count_kernel, dim3(1), dim3(1), 0, s, n_tok, out);
    FIRA_CHECK_LAUNCH("inv_count");
    return 0;
}

}  // namespace fira

extern "C" {
int fira_copy_score_fwd(void* stream, int B, int T, int S, const float* src, const float* tgt, const float* w,
                        const float* bias, float* score) {
    return fira::copy_score_fwd((hipStream_t)stream, B, T, S, src, tgt, w, bias, score);
}
int fira_copy_score_bwd(void* stream, int B, int T, int S, const float* src, const float* tgt, const float* w,
                        const float* dscore, float* dsrc, float* dtgt, float* dw, float* dbias) {
    return fira::copy_score_bwd((hipStream_t)stream, B, T, S, src, tgt, w, dscore, dsrc, dtgt, dw, dbias);
}
int fira_head_loss(void* stream, int BT, int T, int V, int S, const int32_t* compact_row, float* logits, int ldl,
                   float* score, const int32_t* mem_valid, float* gate_logits, const int32_t* tar_label,
                   float* loss_sum, int32_t* n_tok, int32_t* argmax_out, int want_grad) {
    return fira::head_loss((hipStream_t)stream, BT, T, V, S, compact_row, logits, ldl, score, mem_valid, gate_logits,
                           tar_label, loss_sum, n_tok, argmax_out, want_grad);
}
int fira_adam_step_mb(void* stream, int64_t n, float* p, const float* g0, const float* g1, float* m, float* v, float lr,
                      float beta1, float beta2, float eps, int step, const int32_t* n_tok0, const int32_t* n_tok1) {
    FIRA_REQUIRE(p && g0 && m && v && n_tok0 && step >= 1, "fira_adam_step_mb: bad argument");
    return fira::adam_step_mb((hipStream_t)stream, n, p, g0, g1, m, v, lr, beta1, beta2, eps, step, n_tok0, n_tok1);
}
int fira_inv_count(void* stream, const int32_t* n_tok, float* out) {
    return fira::inv_count((hipStream_t)stream, n_tok, out);
}
int fira_adam_step(void* stream, int64_t n, float* p, const float* g, float* m, float* v, float lr, float beta1,
                   float beta2, float eps, int step, const float* inv_scale_ntok) {
    return fira::adam_step((hipStream_t)stream, n, p, g, m, v, lr, beta1, beta2, eps, step, inv_scale_ntok, 0);
}
int fira_adam_step_count(void* stream, int64_t n, float* p, const float* g, float* m, float* v, float lr, float beta1,
                         float beta2, float eps, int step, const float* count) {
    FIRA_REQUIRE(p && g && m && v && count && step >= 1, "fira_adam_step_count: bad argument");
    return fira::adam_step((hipStream_t)stream, n, p, g, m, v, lr, beta1, beta2, eps, step, count, 1);
}
int fira_pack_stats(void* stream, const float* loss_sum, const int32_t* n_tok, float* out2) {
    FIRA_REQUIRE(loss_sum && n_tok && out2, "fira_pack_stats: null pointer argument");
    hipLaunchKernelGGL(fira::pack_stats_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, loss_sum, n_tok, out2);
    FIRA_CHECK_LAUNCH("pack_stats");
    return 0;
}
}
