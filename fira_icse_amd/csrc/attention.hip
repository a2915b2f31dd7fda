// Multi-head attention core for the decoder (reference gnn_transformer.py:137-158): per (commit, head)
//   O = softmax( masked_fill(Q K^T / sqrt(32), mask == 0, -1e9) ) V          head width 32, Tq <= 32.
// Sequence geometry is tiny and fixed (Tq = 30; Tk = 30 self / 370 cross), so one workgroup handles one
// (b, h): NW wavefronts split the key tiles (32 keys each; cross attention: 12 tiles on 12 waves, 3 per SIMD, which
// hides the operand-load latency that a 4-wave version exposed), everything stays in registers and the only
// LDS traffic is the cross-wave combine of the row statistics and of the 32x32 output tile.
//
// The matrix products run on v_mfma_f32_32x32x2_f32 (exact fp32).  Two operand tricks keep it shuffle-free:
//  * the reduction index order of an MFMA chain is free, so operand fragments are fetched as 16 contiguous
//    floats per lane:   X[row = lane&31][ (lane>>5)*16 + s ],  s = MFMA step 0..15   (4 x 16-byte loads);
//  * scores are produced TRANSPOSED (S^T = K Q^T: rows = keys in the accumulator registers, column = query
//    in the lane), so the soft-max row reduction is in-lane, and accumulator register s of S^T is exactly
//    the A-operand fragment of step s for P·V when the V rows are fetched in the accumulator's row order
//    key(s, lane>>5) = (s&3) + 8*(s>>2) + 4*(lane>>5).
// The backward also needs the non-transposed tile (for dK = dS^T Q and dV = P^T dO); it is recomputed by
// swapping the two operand fragments of the same MFMA chain, never by a transpose.
#include "engine.h"
#include "mfma_frag.h"

namespace fira {

constexpr float SQRT_DH = 5.656854249492381f;
// the kernels multiply by reciprocals (IEEE division is a ~10-instruction VALU sequence and these kernels are bound by
// their non-MFMA VALU work: profiles/r1e_probes.md); the result differs from a true division by at most 1 ulp
constexpr float INV_SQRT_DH = 1.0f / SQRT_DH;
constexpr int MAX_TK = 384;

// Rows of masked keys are never read (their K/V fragments are zero-filled): their score is -1e9 whatever K holds and
// their soft-max weight is exactly 0, so callers may leave those rows uninitialised.
// masked, scaled score.  beyond Tk: -inf (not part of the soft-max at all); masked key: -1e9 as the reference.
__device__ __forceinline__ float mask_score(float raw, int key, int query, int Tk, const int* kv, int causal, int q_pos0,
                                            bool& masked) {
    if (key >= Tk) { masked = true; return -INFINITY; }
    masked = (kv[key] == 0) || (causal && key > query + q_pos0);
    return masked ? -1e9f : raw * INV_SQRT_DH;
}

// The row ranges of a (batch entry, K/V entry) pair -- q_off[b], q_off[b+1], k_off[bk], k_off[bk+1] -- fetched by lanes 0..3
// with ONE vector load and broadcast: as four scalar loads behind null-pointer branches they were four DEPENDENT round
// trips at the head of every attention launch (one cold workgroup per CU: the scalar cache never helps).
struct AttnRanges { int qb, tq, k0, tk; };
__device__ __forceinline__ AttnRanges attn_ranges(const int32_t* __restrict__ q_off, const int32_t* __restrict__ k_off, int b,
                                                  int bk, int Tq, int Tk, bool selfk, int lane) {
    int v = 0;
    const int32_t* src = nullptr;
    if (lane < 2 && q_off) src = q_off + b + lane;
    if (lane >= 2 && lane < 4 && k_off && !selfk) src = k_off + bk + (lane - 2);
    if (src) v = *src;
    AttnRanges r;
    r.qb = q_off ? __builtin_amdgcn_readlane(v, 0) : b * Tq;
    r.tq = q_off ? __builtin_amdgcn_readlane(v, 1) - r.qb : Tq;
    r.k0 = (k_off && !selfk) ? __builtin_amdgcn_readlane(v, 2) : 0;
    r.tk = selfk ? r.tq : (k_off ? min(__builtin_amdgcn_readlane(v, 3) - r.k0, Tk) : Tk);
    return r;
}

template <int NW, int TPW, bool BF>
__global__ __launch_bounds__(NW * 64) void attention_fwd_kernel(int H, int Tq, int Tk, const float* __restrict__ Q,
                                                                int ldq, const float* __restrict__ K, int ldk,
                                                                const float* __restrict__ V, int ldv,
                                                                const int32_t* __restrict__ key_valid, int causal,
                                                                int q_pos0, float* __restrict__ O, int ldo, int kb,
                                                                int kvb, int qpk, const int32_t* __restrict__ q_off,
                                                                int self_kv, const int32_t* __restrict__ k_off) {
    // q_off (optional): the queries of batch entry b are rows q_off[b] .. q_off[b+1] of Q / O (a ragged, compact row
    // layout: the decoder's computed target rows); with self_kv the keys / values are the same rows of K / V.
    // k_off (optional, cross attention): RAGGED key rows -- the keys / values of K/V batch entry bk are rows
    // k_off[bk] .. k_off[bk+1] of K / V (at most Tk of them: the commit's computed memory rows, no padding rows in between),
    // and key_valid is indexed by the same compact rows.  Otherwise key_valid is dense (kvb entries per batch entry).
    // kb: K/V rows per batch entry, kvb: key_valid entries per batch entry, qpk: consecutive query batches
    // that share one K/V batch entry (beam rows of one commit share the encoder memory)
    __shared__ int sm_kv[MAX_TK];
    __shared__ float sm_red[NW][32];
    __shared__ float sm_o[NW > 1 ? NW * 1024 : 1];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int bk = b / qpk;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const bool selfk = q_off && self_kv;
    const AttnRanges rg = attn_ranges(q_off, k_off, b, bk, Tq, Tk, selfk, lane);
    const int qb = rg.qb, tq = rg.tq, k0 = rg.k0, tk = rg.tk;
    const int NT = (tk + 31) / 32;
    if (tk <= 0) {                                   // no key at all (never in the engine: a commit has its <start> token)
        for (int idx = t; idx < tq * FIRA_DH; idx += NW * 64)
            O[((size_t)qb + idx / FIRA_DH) * ldo + h * FIRA_DH + idx % FIRA_DH] = 0.f;
        return;
    }
    const int32_t* kvp = key_valid + (k_off && !selfk ? (size_t)k0 : (size_t)bk * kvb);
    const size_t krow0 = selfk ? (size_t)qb : (k_off ? (size_t)k0 : (size_t)bk * kb);
    K += krow0 * ldk;                                // this batch entry's key/value rows
    V += krow0 * ldv;

    // every operand of the wave is requested here, before the key mask has reached LDS: the fragments of masked keys are
    // zeroed afterwards instead of not being loaded (one memory round trip instead of three; see the backward kernel)
    float bq[16];
    load_frag(bq, Q + ((size_t)qb + min(l31, tq - 1)) * ldq + h * FIRA_DH + kh * 16, true);
    float ak[TPW][16], vv[TPW][16];                  // K fragment of key tile*32 + l31; V rows of key acc_row(s,kh)
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int kt = wave + i * NW;
        if (kt < NT) {                               // (wave-uniform: a wave without a key tile requests nothing)
            load_frag(ak[i], K + (size_t)min(kt * 32 + l31, tk - 1) * ldk + h * FIRA_DH + kh * 16, true);
#pragma unroll
            for (int s = 0; s < 16; ++s) vv[i][s] = V[(size_t)min(kt * 32 + acc_row(s, kh), tk - 1) * ldv + h * FIRA_DH + l31];
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) { ak[i][s] = 0.f; vv[i][s] = 0.f; }
        }
    }
    // the key mask travels in the SAME round trip, behind the operand requests (round 4: it used to be fetched and parked in
    // LDS before the first operand load was issued -- one more dependent round trip)
    asm volatile("" ::: "memory");
    for (int i = t; i < tk; i += NW * 64) sm_kv[i] = kvp[i];
    __syncthreads();
    if (l31 >= tq) {
#pragma unroll
        for (int s = 0; s < 16; ++s) bq[s] = 0.f;
    }
    // live[i]: the wave's key tile holds at least one unmasked key.  A tile without one contributes nothing: its scores are
    // the constant -1e9, its soft-max weights exp(-1e9 - max) are exactly 0 next to any real score (and its V fragment is
    // zero-filled anyway), so both MFMA chains are skipped -- on FIRA-shaped batches about half of the twelve memory tiles
    // (padding behind the code tokens and behind the sub-tokens), which halves the load of the SIMD's MFMA pipe.
    bool live[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int kt = wave + i * NW;
        const int key = kt * 32 + l31;
        const bool ok = kt < NT && key < tk && sm_kv[key < tk ? key : 0] != 0;
        live[i] = __ballot(ok) != 0;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int kr = kt * 32 + acc_row(s, kh);
            if (!ok) ak[i][s] = 0.f;
            if (!(kt < NT && kr < tk && sm_kv[kr < tk ? kr : 0] != 0)) vv[i][s] = 0.f;
        }
    }

    f32x16 st[TPW];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int kt = wave + i * NW;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[i][r] = -INFINITY;
        if (kt < NT && live[i]) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            acc = chain16<BF>(ak[i], bq, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                bool masked;
                const float x = mask_score(acc[r], kt * 32 + acc_row(r, kh), l31, tk, sm_kv, causal, q_pos0, masked);
                st[i][r] = x;
                mx = fmaxf(mx, x);
            }
        } else if (kt < NT) {                            // every key masked: the scores mask_score would return
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x = kt * 32 + acc_row(r, kh) < tk ? -1e9f : -INFINITY;
                st[i][r] = x;
                mx = fmaxf(mx, x);
            }
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (NW > 1) {
        if (kh == 0) sm_red[wave][l31] = mx;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NW; ++w) mx = fmaxf(mx, sm_red[w][l31]);
        __syncthreads();
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = expf(st[i][r] - mx);       // exp(-inf) = 0 for keys beyond Tk / tiles not owned
            st[i][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 32, 64);
    if (NW > 1) {
        if (kh == 0) sm_red[wave][l31] = sum;
        __syncthreads();
        sum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += sm_red[w][l31];
    }
    const float inv_sum = 1.0f / sum;
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int kt = wave + i * NW;
        if (kt < NT && live[i]) {
            float pn[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) pn[s] = st[i][s] * inv_sum;
            o = chain16<BF>(pn, vv[i], o);
        }
    }
    // o[r]: query = acc_row(r, kh), d = l31
    if (NW > 1) {
        const int nwa = min(NW, NT);                 // waves that own a key tile (the others hold zeros: not exchanged)
        if (wave < nwa) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sm_o[wave * 1024 + r * 64 + lane] = o[r];
        }
        __syncthreads();
        for (int idx = t; idx < 1024; idx += NW * 64) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) if (w < nwa) v += sm_o[w * 1024 + idx];
            const int r = idx >> 6, ln = idx & 63;
            const int q = acc_row(r, ln >> 5);
            if (q < tq) O[((size_t)qb + q) * ldo + h * FIRA_DH + (ln & 31)] = v;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = acc_row(r, kh);
            if (q < tq) O[((size_t)qb + q) * ldo + h * FIRA_DH + l31] = o[r];
        }
    }
}

// Backward.  Every operand a wave needs -- its K / V fragments, the K rows of its key tile, the query-side fragments and
// rows -- is requested at the top of the kernel, before the key mask has even reached LDS: the fragments of masked keys
// are zeroed afterwards instead of not being loaded (their rows lie inside the caller's buffers), so the kernel waits for
// ONE memory round trip instead of one per phase (round 2 loaded K and V twice, phase by phase, behind the mask).
template <int NW, int TPW, bool BF>
__global__ __launch_bounds__(NW * 64) void attention_bwd_kernel(
    int H, int Tq, int Tk, const float* __restrict__ Q, int ldq, const float* __restrict__ K, int ldk,
    const float* __restrict__ V, int ldv, const int32_t* __restrict__ key_valid, int causal, int q_pos0,
    const float* __restrict__ O, int ldo, const float* __restrict__ dO, int lddo, float* __restrict__ dQ, int lddq,
    float* __restrict__ dK, int lddk, float* __restrict__ dV, int lddv, const int32_t* __restrict__ q_off, int self_kv,
    const int32_t* __restrict__ k_off) {             // k_off: ragged key rows, see attention_fwd_kernel
    __shared__ int sm_kv[MAX_TK];
    __shared__ float sm_red[NW][32];
    __shared__ float sm_m[32], sm_sum[32], sm_delta[32];
    __shared__ float sm_o[NW > 1 ? NW * 1024 : 1];
    __shared__ float sm_q[32 * 33], sm_do[32 * 33];             // Q and dO tiles [query][d] (pitch 33) for phase N's row operands
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const bool selfk = q_off && self_kv;
    const AttnRanges rg = attn_ranges(q_off, k_off, b, b, Tq, Tk, selfk, lane);       // ragged rows: see attention_fwd_kernel
    const int qb = rg.qb, tq = rg.tq, k0 = rg.k0, tk = rg.tk;
    const size_t kbase = selfk ? (size_t)qb : (k_off ? (size_t)k0 : (size_t)b * Tk);
    const int NT = (tk + 31) / 32;
    if (tk <= 0) {                                   // no key: nothing flows back to the queries (and there is no dK / dV row)
        for (int idx = t; idx < tq * FIRA_DH; idx += NW * 64)
            dQ[((size_t)qb + idx / FIRA_DH) * lddq + h * FIRA_DH + idx % FIRA_DH] = 0.f;
        return;
    }
    const int32_t* kvp = key_valid + (k_off && !selfk ? (size_t)k0 : (size_t)b * Tk);

    // ---- all operands, one round trip (rows past the end are clamped to a real row and zeroed below) ----------
    const int ql = min(l31, tq - 1);
    float bq[16], bdo[16], bo[16];                   // fragments X[query = lane&31][kh*16 + s]
    load_frag(bq, Q + ((size_t)qb + ql) * ldq + h * FIRA_DH + kh * 16, true);
    load_frag(bdo, dO + ((size_t)qb + ql) * lddo + h * FIRA_DH + kh * 16, true);
    load_frag(bo, O + ((size_t)qb + ql) * ldo + h * FIRA_DH + kh * 16, true);
    float ak[TPW][16], av[TPW][16], kvv[TPW][16];    // K / V fragments of key = tile*32 + l31; K rows of key acc_row(s,kh)
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int kt = wave + i * NW;
        if (kt < NT) {                               // (wave-uniform: a wave without a key tile requests nothing)
            const int key = min(kt * 32 + l31, tk - 1);
            load_frag(ak[i], K + (kbase + key) * ldk + h * FIRA_DH + kh * 16, true);
            load_frag(av[i], V + (kbase + key) * ldv + h * FIRA_DH + kh * 16, true);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int kr = min(kt * 32 + acc_row(s, kh), tk - 1);
                kvv[i][s] = K[(kbase + kr) * ldk + h * FIRA_DH + l31];
            }
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s) { ak[i][s] = 0.f; av[i][s] = 0.f; kvv[i][s] = 0.f; }
        }
    }
    asm volatile("" ::: "memory");                   // the key mask rides behind the operand requests (see the forward kernel)
    for (int i = t; i < tk; i += NW * 64) sm_kv[i] = kvp[i];
    __syncthreads();
    if (l31 >= tq) {
#pragma unroll
        for (int s = 0; s < 16; ++s) { bq[s] = 0.f; bdo[s] = 0.f; bo[s] = 0.f; }
    }
    if (wave == 0) {                                 // the same tiles, transposed access in phase N: through LDS
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            sm_q[l31 * 33 + kh * 16 + s] = bq[s];
            sm_do[l31 * 33 + kh * 16 + s] = bdo[s];
        }
    }
    bool live[TPW];                                  // see attention_fwd_kernel: tiles without an unmasked key skip their MFMA chains
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int kt = wave + i * NW;
        const int key = kt * 32 + l31;
        const bool ok = kt < NT && key < tk && sm_kv[key < tk ? key : 0] != 0;
        live[i] = __ballot(ok) != 0;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int kr = kt * 32 + acc_row(s, kh);
            if (!ok) { ak[i][s] = 0.f; av[i][s] = 0.f; }
            if (!(kt < NT && kr < tk && sm_kv[kr < tk ? kr : 0] != 0)) kvv[i][s] = 0.f;
        }
    }
    float delta;
    {
        float d = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) d = fmaf(bdo[s], bo[s], d);
        delta = d + __shfl_xor(d, 32, 64);           // rowsum(dO * O) = rowsum(P * dP)
    }

    // ---- statistics (identical to the forward) -------------------------------------------------
    f32x16 st[TPW];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int kt = wave + i * NW;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[i][r] = -INFINITY;
        if (kt < NT && live[i]) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            acc = chain16<BF>(ak[i], bq, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                bool masked;
                const float x = mask_score(acc[r], kt * 32 + acc_row(r, kh), l31, tk, sm_kv, causal, q_pos0, masked);
                st[i][r] = x;
                mx = fmaxf(mx, x);
            }
        } else if (kt < NT) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x = kt * 32 + acc_row(r, kh) < tk ? -1e9f : -INFINITY;
                st[i][r] = x;
                mx = fmaxf(mx, x);
            }
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (NW > 1) {
        if (kh == 0) sm_red[wave][l31] = mx;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NW; ++w) mx = fmaxf(mx, sm_red[w][l31]);
        __syncthreads();
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = expf(st[i][r] - mx);
            st[i][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 32, 64);
    if (NW > 1) {
        if (kh == 0) sm_red[wave][l31] = sum;
        __syncthreads();
        sum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += sm_red[w][l31];
    }
    const float inv_sum = 1.0f / sum;
    if (wave == 0 && kh == 0) {
        sm_m[l31] = mx;
        sm_sum[l31] = inv_sum;                       // reciprocal of the soft-max denominator
        sm_delta[l31] = delta;
    }
    __syncthreads();

    // ---- phase T (transposed tiles: register row = key, lane column = query): dQ -------------------
    f32x16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int kt = wave + i * NW;
        if (kt < NT && live[i]) {                      // a tile of masked keys: dS = 0, nothing flows into dQ
            f32x16 dpt;
#pragma unroll
            for (int r = 0; r < 16; ++r) dpt[r] = 0.f;
            dpt = chain16<BF>(av[i], bdo, dpt);                                     // dP^T = V dO^T
            float dsv[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int kr = kt * 32 + acc_row(s, kh);
                const bool dead = kr >= tk || sm_kv[kr < tk ? kr : 0] == 0 || (causal && kr > l31 + q_pos0);
                const float p = st[i][s] * inv_sum;
                dsv[s] = dead ? 0.f : p * (dpt[s] - delta) * INV_SQRT_DH;
            }
            dq = chain16<BF>(dsv, kvv[i], dq);                                     // dQ += dS K
        }
    }
    if (NW > 1) {
        const int nwa = min(NW, NT);                 // waves that own a key tile
        if (wave < nwa) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sm_o[wave * 1024 + r * 64 + lane] = dq[r];
        }
        __syncthreads();
        for (int idx = t; idx < 1024; idx += NW * 64) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) if (w < nwa) v += sm_o[w * 1024 + idx];
            const int r = idx >> 6, ln = idx & 63;
            const int q = acc_row(r, ln >> 5);
            if (q < tq) dQ[((size_t)qb + q) * lddq + h * FIRA_DH + (ln & 31)] = v;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = acc_row(r, kh);
            if (q < tq) dQ[((size_t)qb + q) * lddq + h * FIRA_DH + l31] = dq[r];
        }
    }

    // ---- phase N (register row = query, lane column = key): dK, dV ---------------------------------
    float qrow[16], dorow[16];              // B operands: X[query = acc_row(s,kh)][d = l31] (rows >= tq are zero)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        qrow[s] = sm_q[acc_row(s, kh) * 33 + l31];
        dorow[s] = sm_do[acc_row(s, kh) * 33 + l31];
    }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int kt = wave + i * NW;
        if (kt < NT) {
            const int key = kt * 32 + l31;
            f32x16 sN, dpN;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sN[r] = 0.f; dpN[r] = 0.f; }
            f32x16 dk, dv;
#pragma unroll
            for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
            if (live[i]) {
                sN = chain16<BF>(bq, ak[i], sN);             // S   = Q K^T
                dpN = chain16<BF>(bdo, av[i], dpN);          // dP  = dO V^T
                float pv[16], dsn[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const int q = acc_row(s, kh);
                    bool masked;
                    const float x = mask_score(sN[s], key, q, tk, sm_kv, causal, q_pos0, masked);
                    const float p = expf(x - sm_m[q]) * sm_sum[q];
                    pv[s] = p;
                    dsn[s] = masked ? 0.f : p * (dpN[s] - sm_delta[q]) * INV_SQRT_DH;
                }
                dk = chain16<BF>(dsn, qrow, dk);          // dK += dS^T Q
                dv = chain16<BF>(pv, dorow, dv);          // dV += P^T dO
            }                                             // (a tile of masked keys: its dK / dV rows are zero)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kr = kt * 32 + acc_row(r, kh);
                if (kr < tk) {
                    dK[(kbase + kr) * lddk + h * FIRA_DH + l31] = dk[r];
                    dV[(kbase + kr) * lddv + h * FIRA_DH + l31] = dv[r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Decode-step attention (one query per hypothesis row; run_model.py:256 through the K/V-cached step of engine.hip).
// The MFMA kernel above spends a 32-query tile on one query and fetches K/V in fragment layout (a load instruction
// touches 32 lines and uses 32 bytes of each); at Tq = 1 the step is a pure stream over the VALID keys' K and V rows
// (128 bytes per key, head and operand), so this kernel is organised around that stream:
//   * one workgroup (4 waves) per (commit, head); the qpk beam rows of a commit share its memory K/V, which is read ONCE
//     and kept in registers for all of them;
//   * the valid keys are compacted first (ballot + popcount), masked keys are never touched (their soft-max weight is
//     exactly 0 in the reference: exp(-1e9 - max));
//   * 8 lanes share a key row (16 bytes each: every load instruction covers 8 whole 128-byte rows); q.k = 4 fmas per
//     lane + 3 xor-shuffles, p.V accumulates 4 floats per lane, reduced over the 8 key-lanes of a wave and the 4 waves
//     at the end.
// Self-attention with merged q|k|v projections: the newest key / value of row b (index Tk-1) is read from Knew / Vnew
// (the projection's output row) instead of the cache, and this kernel appends it to the cache for the later steps.
// KV16: K / V rows are raw bf16 (the optional bf16 cross-K|V cache of the decode loop: 64 bytes per key, head and operand
// instead of 128; 8 lanes x 8 bytes); the arithmetic stays fp32 on the widened values.  ldk / ldv count ELEMENTS.
__device__ __forceinline__ f32x4v widen4(uint2 u) {
    return f32x4v{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                  __uint_as_float(u.y & 0xffff0000u)};
}
template <int NSLOT, bool KV16>
__global__ __launch_bounds__(256) void decode_attention_kernel(int H, int Tk, const float* __restrict__ Q, int ldq,
                                                               const void* __restrict__ Kv, int ldk,
                                                               const void* __restrict__ Vv, int ldv,
                                                               const int32_t* __restrict__ key_valid, float* __restrict__ O,
                                                               int ldo, int kb, int kvb, int qpk,
                                                               const float* __restrict__ Knew,
                                                               const float* __restrict__ Vnew, int ldn,
                                                               float* __restrict__ Kc_out, float* __restrict__ Vc_out,
                                                               const int32_t* __restrict__ k_off) {
    // k_off (optional): ragged key rows -- commit bk's keys / values are rows k_off[bk] .. k_off[bk+1] of K / V and
    // key_valid is indexed by the same compact rows (the engine's cross K|V of the computed memory rows)
    __shared__ int sm_list[MAX_TK];
    __shared__ int sm_cnt[8];
    __shared__ float sm_f[4][36];
    const int bk = blockIdx.x / H, h = blockIdx.x % H;          // K/V batch entry (commit), head
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int r = lane >> 3, c = lane & 7;                       // key sub-row of the wave, 16-byte column of the head
    // ---- ordered compaction of the valid keys (thread t looks at keys t and t + 256)
    int nv;
    const int kbeg = k_off ? k_off[bk] : 0;
    if (k_off) Tk = min(Tk, k_off[bk + 1] - kbeg);
    const size_t krow0 = k_off ? (size_t)kbeg : (size_t)bk * kb;
    {
        const int32_t* kvp = key_valid + (k_off ? (size_t)kbeg : (size_t)bk * kvb);
        const int k0 = t, k1 = t + 256;
        const bool v0 = k0 < Tk && kvp[k0] != 0;
        const bool v1 = k1 < Tk && kvp[k1] != 0;
        const unsigned long long m0 = __ballot(v0), m1 = __ballot(v1);
        if (lane == 0) { sm_cnt[wave] = __popcll(m0); sm_cnt[4 + wave] = __popcll(m1); }
        __syncthreads();
        int base0 = 0, base1 = 0, tot0 = 0, tot1 = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int c0 = sm_cnt[w], c1 = sm_cnt[4 + w];
            if (w < wave) { base0 += c0; base1 += c1; }
            tot0 += c0; tot1 += c1;
        }
        nv = tot0 + tot1;
        const unsigned long long below = (1ull << lane) - 1ull;
        if (v0) sm_list[base0 + __popcll(m0 & below)] = k0;
        if (v1) sm_list[tot0 + base1 + __popcll(m1 & below)] = k1;
        __syncthreads();
    }
    const size_t hoff = (size_t)h * FIRA_DH + c * 4;
    const float* K = static_cast<const float*>(Kv);
    const float* V = static_cast<const float*>(Vv);
    const float* Kb = K + krow0 * ldk + hoff;
    const float* Vb = V + krow0 * ldv + hoff;
    const uint16_t* Kh = static_cast<const uint16_t*>(Kv) + krow0 * ldk + hoff;
    const uint16_t* Vh = static_cast<const uint16_t*>(Vv) + krow0 * ldv + hoff;
    // ---- this lane's key slots: list index j*32 + wave*8 + r
    f32x4v kf[NSLOT], vf[NSLOT];
    bool have[NSLOT];
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        const int idx = j * 32 + wave * 8 + r;
        have[j] = idx < nv;
        const int key = have[j] ? sm_list[idx] : 0;
        const float* pk = Kb + (size_t)key * ldk;
        const float* pv = Vb + (size_t)key * ldv;
        if (Knew && key == Tk - 1) {                             // the newest key: still in the projection's output row
            pk = Knew + (size_t)bk * ldn + hoff;
            pv = Vnew + (size_t)bk * ldn + hoff;
        }
        if (KV16) {
            kf[j] = have[j] ? widen4(*reinterpret_cast<const uint2*>(Kh + (size_t)key * ldk)) : f32x4v{0.f, 0.f, 0.f, 0.f};
            vf[j] = have[j] ? widen4(*reinterpret_cast<const uint2*>(Vh + (size_t)key * ldv)) : f32x4v{0.f, 0.f, 0.f, 0.f};
        } else {
            kf[j] = have[j] ? *reinterpret_cast<const f32x4v*>(pk) : f32x4v{0.f, 0.f, 0.f, 0.f};
            vf[j] = have[j] ? *reinterpret_cast<const f32x4v*>(pv) : f32x4v{0.f, 0.f, 0.f, 0.f};
        }
    }
    if (Knew && Kc_out && t < 16) {                              // append the new key / value of this head to the cache
        const int cc = t & 7;
        const size_t src = (size_t)bk * ldn + (size_t)h * FIRA_DH + cc * 4;
        const size_t dst = ((size_t)bk * kb + (Tk - 1)) * ldk + (size_t)h * FIRA_DH + cc * 4;
        if (t < 8) *reinterpret_cast<f32x4v*>(Kc_out + dst) = *reinterpret_cast<const f32x4v*>(Knew + src);
        else *reinterpret_cast<f32x4v*>(Vc_out + ((size_t)bk * kb + (Tk - 1)) * ldv + (size_t)h * FIRA_DH + cc * 4) =
                 *reinterpret_cast<const f32x4v*>(Vnew + src);
    }
    for (int qi = 0; qi < qpk; ++qi) {
        const int b = bk * qpk + qi;
        const f32x4v q4 = *reinterpret_cast<const f32x4v*>(Q + (size_t)b * ldq + hoff);
        float sc[NSLOT];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            float d = q4.x * kf[j].x;
            d = fmaf(q4.y, kf[j].y, d); d = fmaf(q4.z, kf[j].z, d); d = fmaf(q4.w, kf[j].w, d);
            d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
            sc[j] = have[j] ? d * INV_SQRT_DH : -INFINITY;
            mx = fmaxf(mx, sc[j]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 8, 64)); mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (lane == 0) sm_f[wave][33] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(sm_f[0][33], sm_f[1][33]), fmaxf(sm_f[2][33], sm_f[3][33]));
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            const float p = have[j] ? __expf(sc[j] - mx) : 0.f;
            sum += p;
            acc += vf[j] * p;
        }
        // over the 8 key sub-rows of the wave (lanes with equal c), then over the 4 waves
#pragma unroll
        for (int o = 8; o < 64; o <<= 1) {
            sum += __shfl_xor(sum, o, 64);
            acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64);
            acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
        }
        if (lane < 8) {
            *reinterpret_cast<f32x4v*>(&sm_f[wave][lane * 4]) = acc;
            if (lane == 0) sm_f[wave][34] = sum;
        }
        __syncthreads();
        if (t < 32) {
            const float tot = (sm_f[0][34] + sm_f[1][34]) + (sm_f[2][34] + sm_f[3][34]);
            const float o = (sm_f[0][t] + sm_f[1][t]) + (sm_f[2][t] + sm_f[3][t]);
            O[(size_t)b * ldo + (size_t)h * FIRA_DH + t] = nv > 0 ? o / tot : 0.f;
        }
        __syncthreads();
    }
}

// bf16 K / V rows (raw bf16, ldk / ldv in elements, rows 8-byte aligned): the decode loop's optional bf16 cross-K|V cache
int decode_attention_kv16(hipStream_t s, int BR, int H, int Tk, const float* Q, int ldq, const uint16_t* K, int ldk,
                          const uint16_t* V, int ldv, const int32_t* key_valid, float* O, int ldo, int kb, int kvb, int qpk,
                          const int32_t* k_off) {
    ProfScope prof(s, PROF_ATTN, 0.0);
    if (BR <= 0) return 0;
    FIRA_REQUIRE(Tk >= 1 && Tk <= MAX_TK && qpk >= 1 && BR % qpk == 0 && kb >= Tk && kvb >= Tk, "decode_attention: bad geometry");
    FIRA_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && (uintptr_t)Q % 16 == 0 && (uintptr_t)K % 8 == 0 &&
                     (uintptr_t)V % 8 == 0, "decode_attention: rows must be 8-byte (bf16) / 16-byte (fp32) aligned");
    const dim3 grid((BR / qpk) * H);
    if (Tk <= 32)
        hipLaunchKernelGGL((decode_attention_kernel<1, true>), grid, dim3(256), 0, s, H, Tk, Q, ldq, K, ldk, V, ldv, key_valid, O,
                           ldo, kb, kvb, qpk, nullptr, nullptr, 0, nullptr, nullptr, k_off);
    else
        hipLaunchKernelGGL((decode_attention_kernel<12, true>), grid, dim3(256), 0, s, H, Tk, Q, ldq, K, ldk, V, ldv, key_valid, O,
                           ldo, kb, kvb, qpk, nullptr, nullptr, 0, nullptr, nullptr, k_off);
    FIRA_CHECK_LAUNCH("decode_attention_kv16");
    return 0;
}
int decode_attention(hipStream_t s, int BR, int H, int Tk, const float* Q, int ldq, const float* K, int ldk,
                     const float* V, int ldv, const int32_t* key_valid, float* O, int ldo, int kb, int kvb, int qpk,
                     const float* Knew, const float* Vnew, int ldn, float* Kc_out, float* Vc_out, const int32_t* k_off) {
    ProfScope prof(s, PROF_ATTN, 0.0);
    if (BR <= 0) return 0;
    FIRA_REQUIRE(Tk >= 1 && Tk <= MAX_TK && qpk >= 1 && BR % qpk == 0 && kb >= Tk && kvb >= Tk, "decode_attention: bad geometry");
    FIRA_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && (uintptr_t)Q % 16 == 0 && (uintptr_t)K % 16 == 0 &&
                     (uintptr_t)V % 16 == 0 && (!Knew || (ldn % 4 == 0 && (uintptr_t)Knew % 16 == 0 && (uintptr_t)Vnew % 16 == 0)),
                 "decode_attention: rows must be 16-byte aligned");
    FIRA_REQUIRE(!Knew || (qpk == 1 && !k_off), "decode_attention: merged new keys need one query per K/V entry and dense key rows");
    const dim3 grid((BR / qpk) * H);
    if (Tk <= 32)
        hipLaunchKernelGGL((decode_attention_kernel<1, false>), grid, dim3(256), 0, s, H, Tk, Q, ldq, K, ldk, V, ldv, key_valid, O,
                           ldo, kb, kvb, qpk, Knew, Vnew, ldn, Kc_out, Vc_out, k_off);
    else
        hipLaunchKernelGGL((decode_attention_kernel<12, false>), grid, dim3(256), 0, s, H, Tk, Q, ldq, K, ldk, V, ldv, key_valid, O,
                           ldo, kb, kvb, qpk, Knew, Vnew, ldn, Kc_out, Vc_out, k_off);
    FIRA_CHECK_LAUNCH("decode_attention");
    return 0;
}

static int check_geometry(const char* who, int Tq, int Tk, int ldq, int ldk, int ldv, const void* Q, const void* K,
                          const void* V) {
    FIRA_REQUIRE(Tq >= 1 && Tq <= 32, "%s: Tq=%d must be in 1..32", who, Tq);
    FIRA_REQUIRE(Tk >= 1 && Tk <= MAX_TK, "%s: Tk=%d must be in 1..%d", who, Tk, MAX_TK);
    FIRA_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && (uintptr_t)Q % 16 == 0 && (uintptr_t)K % 16 == 0 &&
                     (uintptr_t)V % 16 == 0,
                 "%s: Q/K/V rows must be 16-byte aligned", who);
    return 0;
}

int attention_fwd_ex(hipStream_t s, int B, int H, int Tq, int Tk, const float* Q, int ldq, const float* K, int ldk,
                     const float* V, int ldv, const int32_t* key_valid, int causal, int q_pos0, float* O, int ldo,
                     int kb, int kvb, int qpk, const int32_t* q_off, int self_kv, int bf16, const int32_t* k_off) {
    ProfScope prof(s, PROF_ATTN, 0.0);
    if (B <= 0) return 0;
    if (int e = check_geometry("attention_fwd", Tq, Tk, ldq, ldk, ldv, Q, K, V)) return e;
    FIRA_REQUIRE(kb >= Tk && kvb >= Tk && qpk >= 1, "attention_fwd: bad batch strides");
#define FIRA_ATT_FWD(NW_, BF_)                                                                                      \
    hipLaunchKernelGGL((attention_fwd_kernel<NW_, 1, BF_>), dim3(B * H), dim3(NW_ * 64), 0, s, H, Tq, Tk, Q, ldq, K, ldk, V, \
                       ldv, key_valid, causal, q_pos0, O, ldo, kb, kvb, qpk, q_off, self_kv, k_off)
    if (Tk <= 32) { if (bf16) FIRA_ATT_FWD(1, true); else FIRA_ATT_FWD(1, false); }
    else { if (bf16) FIRA_ATT_FWD(12, true); else FIRA_ATT_FWD(12, false); }
#undef FIRA_ATT_FWD
    FIRA_CHECK_LAUNCH("attention_fwd");
    return 0;
}
int attention_fwd(hipStream_t s, int B, int H, int Tq, int Tk, const float* Q, int ldq, const float* K, int ldk,
                  const float* V, int ldv, const int32_t* key_valid, int causal, int q_pos0, float* O, int ldo,
                  const int32_t* q_off, int self_kv, int bf16, const int32_t* k_off) {
    return attention_fwd_ex(s, B, H, Tq, Tk, Q, ldq, K, ldk, V, ldv, key_valid, causal, q_pos0, O, ldo, Tk, Tk, 1, q_off, self_kv,
                            bf16, k_off);
}

int attention_bwd(hipStream_t s, int B, int H, int Tq, int Tk, const float* Q, int ldq, const float* K, int ldk,
                  const float* V, int ldv, const int32_t* key_valid, int causal, int q_pos0, const float* O, int ldo,
                  const float* dO, int lddo, float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv,
                  const int32_t* q_off, int self_kv, int bf16, const int32_t* k_off) {
    ProfScope prof(s, PROF_ATTN, 0.0);
    if (B <= 0) return 0;
    if (int e = check_geometry("attention_bwd", Tq, Tk, ldq, ldk, ldv, Q, K, V)) return e;
    FIRA_REQUIRE(ldo % 4 == 0 && lddo % 4 == 0 && (uintptr_t)O % 16 == 0 && (uintptr_t)dO % 16 == 0,
                 "attention_bwd: O/dO rows must be 16-byte aligned");
#define FIRA_ATT_BWD(NW_, BF_)                                                                                      \
    hipLaunchKernelGGL((attention_bwd_kernel<NW_, 1, BF_>), dim3(B * H), dim3(NW_ * 64), 0, s, H, Tq, Tk, Q, ldq, K, ldk, V, \
                       ldv, key_valid, causal, q_pos0, O, ldo, dO, lddo, dQ, lddq, dK, lddk, dV, lddv, q_off, self_kv, k_off)
    if (Tk <= 32) { if (bf16) FIRA_ATT_BWD(1, true); else FIRA_ATT_BWD(1, false); }
    else { if (bf16) FIRA_ATT_BWD(12, true); else FIRA_ATT_BWD(12, false); }
#undef FIRA_ATT_BWD
    FIRA_CHECK_LAUNCH("attention_bwd");
    return 0;
}

}  // namespace fira

extern "C" {
int fira_attention_fwd(void* stream, int B, int H, int Tq, int Tk, const float* Q, int ldq, const float* K, int ldk,
                       const float* V, int ldv, const int32_t* key_valid, int causal, int q_pos0, float* O, int ldo) {
    return fira::attention_fwd((hipStream_t)stream, B, H, Tq, Tk, Q, ldq, K, ldk, V, ldv, key_valid, causal, q_pos0, O,
                               ldo);
}
int fira_decode_attention(void* stream, int BR, int H, int Tk, const float* Q, int ldq, const float* K, int ldk,
                          const float* V, int ldv, const int32_t* key_valid, float* O, int ldo, int kb, int kvb, int qpk,
                          const float* Knew, const float* Vnew, int ldn, float* Kc_out, float* Vc_out,
                          const int32_t* k_off) {
    FIRA_REQUIRE(Q && K && V && key_valid && O, "fira_decode_attention: null pointer argument");
    return fira::decode_attention((hipStream_t)stream, BR, H, Tk, Q, ldq, K, ldk, V, ldv, key_valid, O, ldo, kb, kvb, qpk, Knew,
                                  Vnew, ldn, Kc_out, Vc_out, k_off);
}
int fira_attention_bwd(void* stream, int B, int H, int Tq, int Tk, const float* Q, int ldq, const float* K, int ldk,
                       const float* V, int ldv, const int32_t* key_valid, int causal, int q_pos0, const float* O,
                       int ldo, const float* dO, int lddo, float* dQ, int lddq, float* dK, int lddk, float* dV,
                       int lddv) {
    return fira::attention_bwd((hipStream_t)stream, B, H, Tq, Tk, Q, ldq, K, ldk, V, ldv, key_valid, causal, q_pos0, O,
                               ldo, dO, lddo, dQ, lddq, dK, lddk, dV, lddv);
}
int fira_attention_fwd_ex(void* stream, int B, int H, int Tq, int Tk, const float* Q, int ldq, const float* K, int ldk,
                          const float* V, int ldv, const int32_t* key_valid, int causal, int q_pos0, float* O, int ldo,
                          const int32_t* q_off, int self_kv, int dtype, const int32_t* k_off) {
    FIRA_REQUIRE(dtype == FIRA_F32 || dtype == FIRA_BF16, "fira_attention_fwd_ex: dtype must be FIRA_F32 or FIRA_BF16");
    FIRA_REQUIRE(!(k_off && self_kv), "fira_attention_fwd_ex: k_off (ragged memory rows) and self_kv exclude each other");
    return fira::attention_fwd((hipStream_t)stream, B, H, Tq, Tk, Q, ldq, K, ldk, V, ldv, key_valid, causal, q_pos0, O,
                               ldo, q_off, self_kv, dtype == FIRA_BF16, k_off);
}
int fira_attention_bwd_ex(void* stream, int B, int H, int Tq, int Tk, const float* Q, int ldq, const float* K, int ldk,
                          const float* V, int ldv, const int32_t* key_valid, int causal, int q_pos0, const float* O,
                          int ldo, const float* dO, int lddo, float* dQ, int lddq, float* dK, int lddk, float* dV,
                          int lddv, const int32_t* q_off, int self_kv, int dtype, const int32_t* k_off) {
    FIRA_REQUIRE(dtype == FIRA_F32 || dtype == FIRA_BF16, "fira_attention_bwd_ex: dtype must be FIRA_F32 or FIRA_BF16");
    FIRA_REQUIRE(!(k_off && self_kv), "fira_attention_bwd_ex: k_off (ragged memory rows) and self_kv exclude each other");
    return fira::attention_bwd((hipStream_t)stream, B, H, Tq, Tk, Q, ldq, K, ldk, V, ldv, key_valid, causal, q_pos0, O,
                               ldo, dO, lddo, dQ, lddq, dK, lddk, dV, lddv, q_off, self_kv, dtype == FIRA_BF16, k_off);
}
}
