// Hypothesis bookkeeping of the reference's test-time search (run_model.py:225-340), one launch pair per step.
//
// The reference keeps the hypotheses in python lists: per step and beam slot it re-pads them, rebuilds tensors, runs
// the decoder, multiplies the distribution by the hypothesis probability, overwrites finished rows with -1,
// concatenates the slots that still run plus the probabilities of the finished hypotheses (padded with -1), sorts
// all ~75 000 candidates of every commit and rebuilds the lists from the first `beam` entries.  Here the state stays in
// HBM (double-buffered [B, beam, T] ids / lengths / probabilities) and two kernels do that work:
//   beam_prepare  which hypotheses are finished, which slots still run, the decoder's input token of this step
//   beam_select   per commit: the `beam` best candidates by (probability descending, index ascending) in one pass
//                 over the step's distribution rows -- no sort --, copy-id resolution, new hypotheses, cache parents
// Nothing returns to the host; `done` latches when no slot runs any more (run_model.py:276-279) and turns the
// remaining steps into state copies, so a driver may poll it every few steps.
#include <limits.h>
#include <stdlib.h>
#include "engine.h"

namespace fira {

constexpr int BEAM_MAX = 8;

__global__ __launch_bounds__(256) void beam_prepare_kernel(int rows, int beam, int T, int step,
                                                           const int32_t* __restrict__ gen,
                                                           const int32_t* __restrict__ length,
                                                           int32_t* __restrict__ tok, int32_t* __restrict__ fin,
                                                           int32_t* __restrict__ active, int32_t* __restrict__ done) {
    __shared__ int act[BEAM_MAX];
    const int t = threadIdx.x;
    if (t < BEAM_MAX) act[t] = 0;
    __syncthreads();
    for (int r = t; r < rows; r += 256) {
        const int len = length[r];
        const int f = gen[(size_t)r * T + len - 1] == 1;            // last token is <eos> (run_model.py:235)
        fin[r] = f;
        if (!f) atomicOr(&act[r % beam], 1);                         // the slot runs iff some commit is unfinished in it
        tok[r] = len > step ? gen[(size_t)r * T + step] : 0;         // the <pad>-extended hypothesis at this position
    }
    __syncthreads();
    if (t == 0) {
        int n = 0;
        for (int j = 0; j < beam; ++j) { active[j] = act[j]; n += act[j]; }
        active[BEAM_MAX] = n;
        if (n == 0) *done = 1;
    }
}

struct Cand { float v; int i; };
__device__ __forceinline__ bool better(float v, int i, float w, int j) { return v > w || (v == w && i < j); }

// NB = length of the per-thread candidate list (>= beam: only the `beam` best of a thread can be among the `beam` best of
// the commit), NT threads per commit.  The result is the exact top-`beam` under the total order (probability descending,
// flattened index ascending), so it does not depend on NB / NT: round 2 went from 256 threads x 8-deep lists (166 us per
// step at beam 3: 293 candidates per thread, every one pushed through an 8-deep insertion) to 1024 x NB.
template <int NB, int NT>
__global__ __launch_bounds__(NT) void beam_select_kernel(int beam, int T, int W, int V, int L, int S,
                                                          const float* __restrict__ dist,
                                                          const int32_t* __restrict__ fin,
                                                          const int32_t* __restrict__ active,
                                                          const int32_t* __restrict__ done,
                                                          const int32_t* __restrict__ sou,
                                                          const int32_t* __restrict__ sub,
                                                          const int32_t* __restrict__ gen_in,
                                                          const int32_t* __restrict__ len_in,
                                                          const float* __restrict__ prob_in,
                                                          int32_t* __restrict__ gen_out, int32_t* __restrict__ len_out,
                                                          float* __restrict__ prob_out, int32_t* __restrict__ parent) {
    __shared__ float smv[NT / 64];
    __shared__ int smi[NT / 64];
    __shared__ float sel_v[BEAM_MAX];
    __shared__ int sel_i[BEAM_MAX], act_slot[BEAM_MAX], order[BEAM_MAX], src_of[BEAM_MAX], tok_of[BEAM_MAX],
        carry_of[BEAM_MAX];
    const int b = blockIdx.x, t = threadIdx.x, r0 = b * beam;
    if (*done) {                                                     // search over: hand the state on unchanged
        for (int x = t; x < beam * T; x += NT) gen_out[(size_t)r0 * T + x] = gen_in[(size_t)r0 * T + x];
        if (t < beam) { len_out[r0 + t] = len_in[r0 + t]; prob_out[r0 + t] = prob_in[r0 + t]; parent[r0 + t] = r0 + t; }
        return;
    }
    const int n_act = active[BEAM_MAX];
    // thread-local best `beam` candidates, kept sorted
    float lv[NB];
    int li[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) { lv[q] = -INFINITY; li[q] = INT_MAX; }
    auto offer = [&](float v, int i) {
        if (!better(v, i, lv[NB - 1], li[NB - 1])) return;
        lv[NB - 1] = v; li[NB - 1] = i;
#pragma unroll
        for (int q = NB - 1; q > 0; --q)
            if (better(lv[q], li[q], lv[q - 1], li[q - 1])) {
                const float tv = lv[q]; lv[q] = lv[q - 1]; lv[q - 1] = tv;
                const int ti = li[q]; li[q] = li[q - 1]; li[q - 1] = ti;
            }
    };
    int k = 0;
    for (int j = 0; j < beam; ++j) {
        if (!active[j]) continue;
        if (t == 0) act_slot[k] = j;
        const bool f = fin[r0 + j] != 0;
        const float pj = prob_in[r0 + j];
        const float* row = dist + (size_t)(r0 + j) * W;
        // run_model.py:268-272; four loads in flight per trip (the list update is a dependent chain)
        for (int w0 = t; w0 < W; w0 += 4 * NT) {
            float x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = row[min(w0 + u * NT, W - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (w0 + u * NT < W) offer(f ? -1.0f : x[u] * pj, k * W + w0 + u * NT);
        }
        ++k;
    }
    if (t == 0) {                                                    // finished hypotheses, slot order, -1 padding (:283-296)
        int c = 0;
        for (int j = 0; j < beam; ++j)
            if (fin[r0 + j]) { order[c] = j; offer(prob_in[r0 + j], n_act * W + c); ++c; }
        for (int q = c; q < beam; ++q) { order[q] = 0; offer(-1.0f, n_act * W + q); }
    }
    // `beam` rounds of a block-wide arg-max over the list heads; the owner of the winner pops it
    for (int round = 0; round < beam; ++round) {
        float v = lv[0];
        int i = li[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(v, o, 64);
            const int oi = __shfl_xor(i, o, 64);
            if (better(ov, oi, v, i)) { v = ov; i = oi; }
        }
        __syncthreads();
        if ((t & 63) == 0) { smv[t >> 6] = v; smi[t >> 6] = i; }
        __syncthreads();
        v = smv[0]; i = smi[0];
#pragma unroll
        for (int q = 1; q < NT / 64; ++q)
            if (better(smv[q], smi[q], v, i)) { v = smv[q]; i = smi[q]; }
        if (t == 0) { sel_v[round] = v; sel_i[round] = i; }
        if (li[0] == i) {
#pragma unroll
            for (int q = 0; q < NB - 1; ++q) { lv[q] = lv[q + 1]; li[q] = li[q + 1]; }
            lv[NB - 1] = -INFINITY; li[NB - 1] = INT_MAX;
        }
    }
    __syncthreads();
    if (t < beam) {                                                  // run_model.py:305-340
        const int idx = sel_i[t];
        const int which = idx / W, w = idx - which * W;
        const int carry = which >= n_act;
        const int src = carry ? order[min(w, beam - 1)] : act_slot[which];
        int nt = w;
        if (w >= V + L) nt = sub[(size_t)b * S + min(w - V - L, S - 1)];
        else if (w >= V) nt = sou[(size_t)b * L + (w - V)];
        src_of[t] = src; tok_of[t] = nt; carry_of[t] = carry;
        const int sl = len_in[r0 + src];
        len_out[r0 + t] = carry ? sl : sl + 1;
        prob_out[r0 + t] = sel_v[t];
        parent[r0 + t] = r0 + src;
    }
    __syncthreads();
    for (int x = t; x < beam * T; x += NT) {
        const int c = x / T, p = x - c * T;
        const int src = src_of[c];
        int g = gen_in[(size_t)(r0 + src) * T + p];
        if (!carry_of[c] && p == min(len_in[r0 + src], T - 1)) g = tok_of[c];
        gen_out[(size_t)(r0 + c) * T + p] = g;
    }
}

// Greedy (beam 1) bookkeeping of one step, run_model.py:305-340 with one hypothesis per commit: resolve the chosen
// output index to a vocabulary id (copy slots read the commit's code / sub-token ids), append it, multiply the running
// probability, and stop the hypothesis at <eos>.  n_alive[step] (zeroed by the caller before the first step) receives
// the number of hypotheses still running after this step, so the host can stop early with ONE read-back per chunk.
__global__ __launch_bounds__(256) void greedy_advance_kernel(int B, int T, int V, int L, int S, int step,
                                                             const int32_t* __restrict__ best_id,
                                                             const float* __restrict__ best_p,
                                                             const int32_t* __restrict__ sou,
                                                             const int32_t* __restrict__ sub, int32_t* __restrict__ out,
                                                             int32_t* __restrict__ length, float* __restrict__ prob,
                                                             int32_t* __restrict__ alive, int32_t* __restrict__ tok,
                                                             int32_t* __restrict__ n_alive) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    int still = 0;
    if (b < B) {
        if (alive[b]) {
            const int w = best_id[b];
            int nt = w;
            if (w >= V + L) nt = sub[(size_t)b * S + min(w - V - L, S - 1)];
            else if (w >= V) nt = sou[(size_t)b * L + (w - V)];
            out[(size_t)b * T + step + 1] = nt;
            prob[b] *= best_p[b];
            length[b] += 1;
            still = nt != 1;                       // <eos> = 1 (config.EOS; the CLI checks the vocabulary agrees)
            alive[b] = still;
            tok[b] = still ? nt : 0;
        } else {
            tok[b] = 0;
        }
    }
    const unsigned long long m = __ballot(still);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&n_alive[step], __popcll(m));
}

// Row-wise top-k of a logits matrix (k <= BEAM_MAX): one workgroup per row, k passes -- pass j takes the best entry that comes
// after pass j - 1's pick in the total order (value descending, index ascending; NaN never wins).  The row (98 KB at the
// reference's vocabulary) stays in L2 between the passes.
__global__ __launch_bounds__(256) void row_topk_kernel(int V, int k, const float* __restrict__ logits, int ldl,
                                                       int32_t* __restrict__ ids, float* __restrict__ vals) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ float pv;
    __shared__ int pi;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* row = logits + (size_t)blockIdx.x * ldl;
    float prev_v = INFINITY;
    int prev_i = -1;
    for (int j = 0; j < k; ++j) {
        float bv = -INFINITY;
        int bi = INT_MAX;
        for (int i = t; i < V; i += 256) {
            const float v = row[i];
            const bool after = v < prev_v || (v == prev_v && i > prev_i);       // not picked by an earlier pass
            if (after && better(v, i, bv, bi)) { bv = v; bi = i; }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
        __syncthreads();
        if (t == 0) {
            float v = sv[0];
            int i = si[0];
            for (int w = 1; w < 4; ++w)
                if (better(sv[w], si[w], v, i)) { v = sv[w]; i = si[w]; }
            pv = v; pi = i;
            ids[(size_t)blockIdx.x * k + j] = i == INT_MAX ? -1 : i;
            vals[(size_t)blockIdx.x * k + j] = v;
        }
        __syncthreads();
        prev_v = pv; prev_i = pi;
        __syncthreads();
    }
}
int row_topk(hipStream_t s, int R, int V, int k, const float* logits, int ldl, int32_t* ids, float* vals) {
    if (R <= 0) return 0;
    FIRA_REQUIRE(k >= 1 && k <= BEAM_MAX && V >= k, "row_topk: k=%d must be in [1, %d] and <= V", k, BEAM_MAX);
    hipLaunchKernelGGL(row_topk_kernel, dim3(R), dim3(256), 0, s, V, k, logits, ldl, ids, vals);
    FIRA_CHECK_LAUNCH("row_topk");
    return 0;
}

}  // namespace fira

extern "C" {
int fira_greedy_advance(void* stream, const fira_dims* d, int B, int step, const int32_t* best_id, const float* best_p,
                        const int32_t* sou, const int32_t* sub_token, int32_t* out, int32_t* length, float* prob,
                        int32_t* alive, int32_t* tokens, int32_t* n_alive) {
    FIRA_REQUIRE(d && B > 0 && step >= 0 && step + 1 < d->tar_len, "fira_greedy_advance: bad step %d", step);
    hipLaunchKernelGGL(fira::greedy_advance_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B,
                       d->tar_len, d->vocab, d->sou_len, d->sub_len, step, best_id, best_p, sou, sub_token, out, length,
                       prob, alive, tokens, n_alive);
    FIRA_CHECK_LAUNCH("greedy_advance");
    return 0;
}
int fira_beam_prepare(void* stream, int B, int n_beam, int T, int step, const int32_t* gen, const int32_t* length,
                      int32_t* tokens, int32_t* finished, int32_t* active, int32_t* done) {
    FIRA_REQUIRE(B > 0 && n_beam >= 1 && n_beam <= fira::BEAM_MAX, "fira_beam_prepare: beam %d outside 1..%d", n_beam,
                 fira::BEAM_MAX);
    hipLaunchKernelGGL(fira::beam_prepare_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, B * n_beam, n_beam, T, step,
                       gen, length, tokens, finished, active, done);
    FIRA_CHECK_LAUNCH("beam_prepare");
    return 0;
}
int fira_beam_select(void* stream, const fira_dims* d, int B, int n_beam, const float* dist, const int32_t* finished,
                     const int32_t* active, const int32_t* done, const int32_t* sou, const int32_t* sub_token,
                     const int32_t* gen_in, const int32_t* len_in, const float* prob_in, int32_t* gen_out,
                     int32_t* len_out, float* prob_out, int32_t* parent) {
    FIRA_REQUIRE(d && B > 0 && n_beam >= 1 && n_beam <= fira::BEAM_MAX, "fira_beam_select: beam %d outside 1..%d", n_beam,
                 fira::BEAM_MAX);
    const int S = d->sub_len, L = d->sou_len, W = d->vocab + L + S;
    if (n_beam <= 4)
        hipLaunchKernelGGL((fira::beam_select_kernel<4, 1024>), dim3(B), dim3(1024), 0, (hipStream_t)stream, n_beam, d->tar_len, W,
                           d->vocab, L, S, dist, finished, active, done, sou, sub_token, gen_in, len_in, prob_in, gen_out,
                           len_out, prob_out, parent);
    else
        hipLaunchKernelGGL((fira::beam_select_kernel<fira::BEAM_MAX, 1024>), dim3(B), dim3(1024), 0, (hipStream_t)stream, n_beam,
                           d->tar_len, W, d->vocab, L, S, dist, finished, active, done, sou, sub_token, gen_in, len_in, prob_in,
                           gen_out, len_out, prob_out, parent);
    FIRA_CHECK_LAUNCH("beam_select");
    return 0;
}
}
