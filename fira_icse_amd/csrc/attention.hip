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
#include "epilogue.h"
#include "mfma_frag.h"

namespace fira {

constexpr float SQRT_DH = 5.656854249492381f;
// the kernels multiply by reciprocals (IEEE division is a ~10-instruction VALU sequence and these kernels are bound by
// their non-MFMA VALU work: profiles/r1e_probes.md); the result differs from a true division by at most 1 ulp
constexpr float INV_SQRT_DH = 1.0f / SQRT_DH;
constexpr int MAX_TK = 384;

// ---- masks as bit vectors --------------------------------------------------------------------------------------------
// A 32x32 score tile lives in 16 accumulator registers per lane; register s of a lane with kh = lane>>5 holds row
// acc_row(s, kh) = c(s) + 4 kh, c(s) = (s&3) + 8 (s>>2), and column lane&31.  Every mask the kernels need is a function of
// (row, column), and per lane the 16 rows differ only in the compile-time constant c(s): each mask is therefore ONE 32-bit
// word per lane and tile whose bit c(s) answers register s (bit test = v_bfe, select = v_bfi: 2 VALU per element).  Round 3
// answered the same questions with an LDS look-up, compares and a divergent branch per element (3 700 instructions in the
// single-wave backward, 1 150 of them scalar exec-mask bookkeeping); the kernels are bound by exactly that issue stream.
// Scores are kept in the base-2 domain (raw * log2(e)/sqrt(32); the soft-max is invariant), so the exponential is one
// v_exp_f32; a masked key's score is the reference's -1e9 (times log2 e), a key beyond Tk is -inf.
constexpr float LOG2E = 1.4426950408889634f;
constexpr float SCORE2 = INV_SQRT_DH * LOG2E;
constexpr float MASKED2 = -1e9f * LOG2E;
__device__ __forceinline__ constexpr int acc_c(int s) { return (s & 3) + 8 * (s >> 2); }
// bits [0, n) set (n may be <= 0 or >= 32)
__device__ __forceinline__ uint32_t low_bits(int n) { return n <= 0 ? 0u : (n >= 32 ? 0xffffffffu : (1u << n) - 1u); }
// sel ? a : b on the bit patterns, sel = all ones / all zeros
__device__ __forceinline__ float bit_select(int sel, float a, float b) {
    return __int_as_float((sel & __float_as_int(a)) | (~sel & __float_as_int(b)));
}
__device__ __forceinline__ int bit_of(uint32_t word, int c) { return __builtin_amdgcn_sbfe((int)word, c, 1); }   // 0 / -1

// 16 contiguous floats of a row through a buffer descriptor (rows outside the descriptor read 0)
__device__ __forceinline__ void buf_frag(float (&f)[16], rsrc_t r, unsigned off) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4v v = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(r, off + 16u * i, 0, 0));
        f[4 * i + 0] = v.x; f[4 * i + 1] = v.y; f[4 * i + 2] = v.z; f[4 * i + 3] = v.w;
    }
}
// the 16 accumulator-order rows of column lane&31: X[row0 + acc_row(s, kh)][lane&31], off0 = byte offset of row0 + 4 kh
__device__ __forceinline__ void buf_rows(float (&f)[16], rsrc_t r, unsigned off0, unsigned ld4) {
#pragma unroll
    for (int s = 0; s < 16; ++s) f[s] = buf_load_f32(r, off0 + (unsigned)acc_c(s) * ld4);
}
// descriptor over rows [0, n) of a [*, ld] matrix slice that is 32 floats wide (n <= 0: nothing is addressable)
__device__ __forceinline__ rsrc_t rows_rsrc(const float* base, int n, int ld) {
    return buf_rsrc(base, n > 0 ? (unsigned)(n - 1) * (unsigned)ld * 4u + 128u : 0u);
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

// Key-tile state of a wave: which of the tile's 32 keys exist and are unmasked, as lane predicates and as row bit words.
struct TileMask {
    bool ok;            // this lane's key (tile*32 + lane&31) is inside Tk and unmasked
    uint32_t valid;     // bit j: key row 4 kh + j of the tile is inside Tk and unmasked   (accumulator-row order)
    uint32_t outside;   // bit j: key row 4 kh + j lies beyond Tk
    bool live;          // the tile holds at least one unmasked key
};
__device__ __forceinline__ TileMask tile_mask(int kvld, int n_in, int kh) {
    TileMask m;
    m.ok = kvld != 0;
    const uint32_t vbits = (uint32_t)__ballot(m.ok);          // both halves of the wave hold the same 32 keys
    m.live = vbits != 0;
    m.valid = vbits >> (4 * kh);
    m.outside = ~low_bits(n_in) >> (4 * kh);
    return m;
}

template <int NW, bool BF>
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
    // this (batch entry, head)'s slices behind buffer descriptors: the base is wave-uniform (scalar address arithmetic), a lane
    // adds a 32-bit byte offset, and rows outside [0, tq) / [0, tk) read zeros and drop their stores -- no clamps, no
    // zero-fills of the edge rows and no branch around any memory instruction
    const size_t krow0 = selfk ? (size_t)qb : (k_off ? (size_t)k0 : (size_t)bk * kb);
    const rsrc_t rQ = rows_rsrc(Q + (size_t)qb * ldq + h * FIRA_DH, tq, ldq);
    const rsrc_t rK = rows_rsrc(K + krow0 * ldk + h * FIRA_DH, tk, ldk);
    const rsrc_t rV = rows_rsrc(V + krow0 * ldv + h * FIRA_DH, tk, ldv);
    const rsrc_t rO = rows_rsrc(O + (size_t)qb * ldo + h * FIRA_DH, tq, ldo);
    const rsrc_t rM = buf_rsrc(key_valid + (k_off && !selfk ? (size_t)k0 : (size_t)bk * kvb), (unsigned)tk * 4u);
    const unsigned ldk4 = (unsigned)ldk * 4u, ldv4 = (unsigned)ldv * 4u, ldo4 = (unsigned)ldo * 4u;
    const int kt = wave;                             // the wave's key tile (one per wave: Tk <= NW * 32)
    const bool mine = kt < NT;                       // (wave-uniform: a wave without a key tile requests nothing)

    // every operand of the wave and its slice of the key mask, one round trip
    float bq[16], ak[16], vv[16];                    // Q / K fragments [row = lane&31][kh*16 + s]; V rows of key acc_row(s,kh)
    int kvld = 0;
    buf_frag(bq, rQ, (unsigned)l31 * (unsigned)ldq * 4u + (unsigned)kh * 64u);
    if (mine) {
        buf_frag(ak, rK, (unsigned)(kt * 32 + l31) * ldk4 + (unsigned)kh * 64u);
        buf_rows(vv, rV, (unsigned)(kt * 32 + 4 * kh) * ldv4 + (unsigned)l31 * 4u, ldv4);
        kvld = __builtin_amdgcn_raw_buffer_load_b32(rM, (unsigned)(kt * 32 + l31) * 4u, 0, 0);
    }
    // A tile without an unmasked key contributes nothing: its scores are the constant -1e9, its soft-max weights
    // exp(-1e9 - max) are exactly 0 next to any real score, so both MFMA chains are skipped -- on FIRA-shaped dense batches
    // about half of the twelve memory tiles (with ragged key rows: none but the commit's last).
    const TileMask tm = tile_mask(kvld, tk - kt * 32, kh);
    const bool live = mine && tm.live;
    // masked(row c) = key masked, or (causal) key row kt*32 + 4 kh + c  >  query l31 + q_pos0
    const uint32_t maskedT = ~tm.valid | (causal ? ~low_bits(l31 + q_pos0 - kt * 32 - 4 * kh + 1) : 0u);

    f32x16 st;
    float mx = -INFINITY;
    if (live) {
        // rows of masked keys are zero-filled: callers may leave them uninitialised (their weight is exactly 0)
#pragma unroll
        for (int s = 0; s < 16; ++s) vv[s] = bit_select(bit_of(tm.valid, acc_c(s)), vv[s], 0.f);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = chain16<BF>(ak, bq, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float x = bit_select(bit_of(maskedT, acc_c(r)), MASKED2, acc[r] * SCORE2);
            x = bit_select(bit_of(tm.outside, acc_c(r)), -INFINITY, x);
            st[r] = x;
            mx = fmaxf(mx, x);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {               // every key masked (or no tile): the scores the branch above would give
            const float x = mine ? bit_select(bit_of(tm.outside, acc_c(r)), -INFINITY, MASKED2) : -INFINITY;
            st[r] = x;
            mx = fmaxf(mx, x);
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
    for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(st[r] - mx);      // exp2(-inf) = 0 for keys beyond Tk / waves without a tile
        st[r] = p;
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
    if (live) {
        float pn[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) pn[s] = st[s] * inv_sum;
        o = chain16<BF>(pn, vv, o);
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
            for (int w = 0; w < nwa; ++w) v += sm_o[w * 1024 + idx];
            const int r = idx >> 6, ln = idx & 63;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), rO,
                                                  (unsigned)acc_row(r, ln >> 5) * ldo4 + (unsigned)(ln & 31) * 4u, 0, 0);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(o[r]), rO,
                                                  (unsigned)(4 * kh + acc_c(r)) * ldo4 + (unsigned)l31 * 4u, 0, 0);
    }
}

// Backward.  Every operand a wave needs -- its K / V fragments, the K rows of its key tile, the query-side fragments and
// rows, its slice of the key mask -- is requested at the top of the kernel in ONE memory round trip (round 2 loaded K and V
// twice, phase by phase, behind the mask; round 3 staged the mask and the transposed query tiles through LDS behind a
// barrier).  Masked keys' fragments are zeroed afterwards instead of not being loaded (their rows lie inside the caller's
// buffers).
template <int NW, bool BF>
__global__ __launch_bounds__(NW * 64) void attention_bwd_kernel(
    int H, int Tq, int Tk, const float* __restrict__ Q, int ldq, const float* __restrict__ K, int ldk,
    const float* __restrict__ V, int ldv, const int32_t* __restrict__ key_valid, int causal, int q_pos0,
    const float* __restrict__ O, int ldo, const float* __restrict__ dO, int lddo, float* __restrict__ dQ, int lddq,
    float* __restrict__ dK, int lddk, float* __restrict__ dV, int lddv, const int32_t* __restrict__ q_off, int self_kv,
    const int32_t* __restrict__ k_off) {             // k_off: ragged key rows, see attention_fwd_kernel
    __shared__ float sm_red[NW][32];
    __shared__ __attribute__((aligned(16))) float sm_m[32], sm_sum[32], sm_delta[32];
    __shared__ float sm_o[NW > 1 ? NW * 1024 : 1];
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
    const rsrc_t rQ = rows_rsrc(Q + (size_t)qb * ldq + h * FIRA_DH, tq, ldq);
    const rsrc_t rO = rows_rsrc(O + (size_t)qb * ldo + h * FIRA_DH, tq, ldo);
    const rsrc_t rdO = rows_rsrc(dO + (size_t)qb * lddo + h * FIRA_DH, tq, lddo);
    const rsrc_t rdQ = rows_rsrc(dQ + (size_t)qb * lddq + h * FIRA_DH, tq, lddq);
    const rsrc_t rK = rows_rsrc(K + kbase * ldk + h * FIRA_DH, tk, ldk);
    const rsrc_t rV = rows_rsrc(V + kbase * ldv + h * FIRA_DH, tk, ldv);
    const rsrc_t rdK = rows_rsrc(dK + kbase * lddk + h * FIRA_DH, tk, lddk);
    const rsrc_t rdV = rows_rsrc(dV + kbase * lddv + h * FIRA_DH, tk, lddv);
    const rsrc_t rM = buf_rsrc(key_valid + (k_off && !selfk ? (size_t)k0 : (size_t)b * Tk), (unsigned)tk * 4u);
    const unsigned ldq4 = (unsigned)ldq * 4u, ldk4 = (unsigned)ldk * 4u, ldv4 = (unsigned)ldv * 4u;
    const unsigned lddo4 = (unsigned)lddo * 4u, lddq4 = (unsigned)lddq * 4u, lddk4 = (unsigned)lddk * 4u, lddv4 = (unsigned)lddv * 4u;
    const int kt = wave;
    const bool mine = kt < NT;

    // ---- all operands, one round trip (rows past the end read zeros) ---------------------------------------------------
    float bq[16], bdo[16], bo[16];                   // fragments X[query = lane&31][kh*16 + s]
    const unsigned fq = (unsigned)kh * 64u;
    buf_frag(bq, rQ, (unsigned)l31 * ldq4 + fq);
    buf_frag(bdo, rdO, (unsigned)l31 * lddo4 + fq);
    buf_frag(bo, rO, (unsigned)l31 * (unsigned)ldo * 4u + fq);
    float ak[16], av[16], kvv[16];                   // K / V fragments of key = tile*32 + l31; K rows of key acc_row(s,kh)
    float qrow[16], dorow[16];                       // Q / dO rows of query acc_row(s,kh), column d = l31 (phase N's B operands)
    int kvld = 0;
    if (mine) {
        const unsigned krow = (unsigned)(kt * 32 + l31);
        buf_frag(ak, rK, krow * ldk4 + fq);
        buf_frag(av, rV, krow * ldv4 + fq);
        buf_rows(kvv, rK, (unsigned)(kt * 32 + 4 * kh) * ldk4 + (unsigned)l31 * 4u, ldk4);
        kvld = __builtin_amdgcn_raw_buffer_load_b32(rM, krow * 4u, 0, 0);
        buf_rows(qrow, rQ, (unsigned)(4 * kh) * ldq4 + (unsigned)l31 * 4u, ldq4);
        buf_rows(dorow, rdO, (unsigned)(4 * kh) * lddo4 + (unsigned)l31 * 4u, lddo4);
    }
    const TileMask tm = tile_mask(kvld, tk - kt * 32, kh);
    const bool live = mine && tm.live;               // see attention_fwd_kernel: tiles without an unmasked key skip their chains
    const uint32_t maskedT = ~tm.valid | (causal ? ~low_bits(l31 + q_pos0 - kt * 32 - 4 * kh + 1) : 0u);
    float delta;
    {
        float d = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) d = fmaf(bdo[s], bo[s], d);
        delta = d + __shfl_xor(d, 32, 64);           // rowsum(dO * O) = rowsum(P * dP)
    }

    // ---- statistics (identical to the forward) -------------------------------------------------
    f32x16 st;
    float mx = -INFINITY;
    if (live) {
#pragma unroll
        for (int s = 0; s < 16; ++s) kvv[s] = bit_select(bit_of(tm.valid, acc_c(s)), kvv[s], 0.f);
        if (!tm.ok) {                                // (lane predicate: a masked key's K / V rows may be uninitialised)
#pragma unroll
            for (int s = 0; s < 16; ++s) { ak[s] = 0.f; av[s] = 0.f; }
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = chain16<BF>(ak, bq, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float x = bit_select(bit_of(maskedT, acc_c(r)), MASKED2, acc[r] * SCORE2);
            x = bit_select(bit_of(tm.outside, acc_c(r)), -INFINITY, x);
            st[r] = x;
            mx = fmaxf(mx, x);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float x = mine ? bit_select(bit_of(tm.outside, acc_c(r)), -INFINITY, MASKED2) : -INFINITY;
            st[r] = x;
            mx = fmaxf(mx, x);
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
    for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(st[r] - mx);
        st[r] = p;
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
    if (wave == 0 && kh == 0) {                      // per-query statistics, for phase N's register-row = query layout
        sm_m[l31] = mx;
        sm_sum[l31] = inv_sum;                       // reciprocal of the soft-max denominator
        sm_delta[l31] = delta;
    }
    __syncthreads();

    // ---- phase T (transposed tiles: register row = key, lane column = query): dQ -------------------
    f32x16 dq;
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[r] = 0.f;
    if (live) {                                      // a tile of masked keys: dS = 0, nothing flows into dQ
        f32x16 dpt;
#pragma unroll
        for (int r = 0; r < 16; ++r) dpt[r] = 0.f;
        dpt = chain16<BF>(av, bdo, dpt);                                     // dP^T = V dO^T
        float dsv[16];
        const float sc = inv_sum * INV_SQRT_DH;
#pragma unroll
        for (int s = 0; s < 16; ++s)
            dsv[s] = bit_select(bit_of(maskedT, acc_c(s)), 0.f, st[s] * sc * (dpt[s] - delta));
        dq = chain16<BF>(dsv, kvv, dq);                                      // dQ += dS K
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
            for (int w = 0; w < nwa; ++w) v += sm_o[w * 1024 + idx];
            const int r = idx >> 6, ln = idx & 63;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), rdQ,
                                                  (unsigned)acc_row(r, ln >> 5) * lddq4 + (unsigned)(ln & 31) * 4u, 0, 0);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(dq[r]), rdQ,
                                                  (unsigned)(4 * kh + acc_c(r)) * lddq4 + (unsigned)l31 * 4u, 0, 0);
    }

    // ---- phase N (register row = query, lane column = key): dK, dV ---------------------------------
    if (mine) {
        f32x16 dk, dv;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[r] = 0.f; dv[r] = 0.f; }
        if (live) {
            // statistics of query 4 kh + c(s): four 16-byte LDS reads per vector (c(s) runs over 4 groups of 4 consecutive rows)
            float qm[16], qs[16], qd[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4v m4 = *reinterpret_cast<const f32x4v*>(&sm_m[4 * kh + 8 * g]);
                const f32x4v s4 = *reinterpret_cast<const f32x4v*>(&sm_sum[4 * kh + 8 * g]);
                const f32x4v d4 = *reinterpret_cast<const f32x4v*>(&sm_delta[4 * kh + 8 * g]);
#pragma unroll
                for (int j = 0; j < 4; ++j) { qm[4 * g + j] = m4[j]; qs[4 * g + j] = s4[j]; qd[4 * g + j] = d4[j]; }
            }
            f32x16 sN, dpN;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sN[r] = 0.f; dpN[r] = 0.f; }
            sN = chain16<BF>(bq, ak, sN);                // S   = Q K^T
            dpN = chain16<BF>(bdo, av, dpN);             // dP  = dO V^T
            // masked(query row c) = key masked, or (causal) key > query 4 kh + c + q_pos0
            const int key = kt * 32 + l31;
            const uint32_t maskedN = tm.ok ? (causal ? low_bits(key - q_pos0 - 4 * kh) : 0u) : 0xffffffffu;
            const float gone = key < tk ? MASKED2 : -INFINITY;
            float pv[16], dsn[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int sel = bit_of(maskedN, acc_c(s));
                const float x = bit_select(sel, gone, sN[s] * SCORE2);
                const float p = __builtin_amdgcn_exp2f(x - qm[s]) * qs[s];
                pv[s] = p;
                dsn[s] = bit_select(sel, 0.f, p * (dpN[s] - qd[s]) * INV_SQRT_DH);
            }
            dk = chain16<BF>(dsn, qrow, dk);             // dK += dS^T Q
            dv = chain16<BF>(pv, dorow, dv);             // dV += P^T dO
        }                                                // (a tile of masked keys: its dK / dV rows are zero)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned row = (unsigned)(kt * 32 + 4 * kh + acc_c(r));
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(dk[r]), rdK, row * lddk4 + (unsigned)l31 * 4u, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(dv[r]), rdV, row * lddv4 + (unsigned)l31 * 4u, 0, 0);
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
            d = sum8(d);                             // (DPP: common.h)
            sc[j] = have[j] ? d * INV_SQRT_DH : -INFINITY;
            mx = fmaxf(mx, sc[j]);
        }
        mx = wave_max(mx);                           // (the 8 lanes of a key hold the same score)
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
        sum += dpp_take<DPP_ROR8, 0xf>(sum, sum);      // lane ^ 8 inside the 16-lane row: one DPP instruction
        acc.x += dpp_take<DPP_ROR8, 0xf>(acc.x, acc.x); acc.y += dpp_take<DPP_ROR8, 0xf>(acc.y, acc.y);
        acc.z += dpp_take<DPP_ROR8, 0xf>(acc.z, acc.z); acc.w += dpp_take<DPP_ROR8, 0xf>(acc.w, acc.w);
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
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

// ---- (round 6) the decode step's whole self-attention block of one row as ONE launch -------------------------------------
// run_model.py:250-267 runs, per decoder layer and step, gnn_transformer.py:137-161 on the newest position: the one-query self
// attention over <= 30 cached keys, the output projection fc_o, the residual, the post-LayerNorm -- and then the cross
// attention's query projection fc_q (gnn_transformer.py:142) reads the result.  At M = B * beam <= 192 rows these were three
// dependent launches of the step's 53 (attention 4.8 us + two 32x32-tile products 5.8 us each, each ~1 us of work behind
// a launch and a memory round trip).  Here one workgroup owns one ROW: scores of the 8 heads x <= 32 keys (a thread per
// (head, key), a 128-byte key slice each), soft-max over the 32 lanes of a head, o = P V (a thread per output element,
// coalesced 1 KiB value rows), then the two 256 x 256 products as MATRIX-VECTOR products on the VALU against k-major copies
// of the weights (coalesced 1 KiB rows, L2-resident: 256 KB per product and workgroup -- at <= 192 rows the re-reads are
// cheaper than a launch boundary), the LayerNorm in between.  The newest key / value come from the merged q|k|v projection's
// output row and are appended to the cache, as decode_attention does.
__global__ __launch_bounds__(256) void decode_self_block_kernel(int Tk, int T, const float* __restrict__ qkv,
                                                                float* __restrict__ Kc, float* __restrict__ Vc,
                                                                const int32_t* __restrict__ hist,
                                                                const float* __restrict__ WoT, const float* __restrict__ bo,
                                                                const float* __restrict__ xres,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ WqT, const float* __restrict__ bq,
                                                                float* __restrict__ xa, float* __restrict__ qc) {
    __shared__ __attribute__((aligned(16))) float s_q[FIRA_D], s_o[FIRA_D], s_x[FIRA_D];
    __shared__ float s_p[8][32];
    __shared__ float s_red[2][4];
    __shared__ __attribute__((aligned(16))) float s_part[4][FIRA_D];
    const int row = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* my = qkv + (size_t)row * 3 * FIRA_D;
    const float knew = my[FIRA_D + t], vnew = my[2 * FIRA_D + t];
    s_q[t] = my[t];
    float* kc_row = Kc + (size_t)row * T * FIRA_D;
    float* vc_row = Vc + (size_t)row * T * FIRA_D;
    kc_row[(size_t)(Tk - 1) * FIRA_D + t] = knew;                 // append this step's key / value to the cache
    vc_row[(size_t)(Tk - 1) * FIRA_D + t] = vnew;
    __syncthreads();
    // ---- scores: thread = (head h, key j); the newest key is read from the projection's row (the cache store above is not
    // ordered against this workgroup's own loads)
    {
        const int h = t >> 5, j = t & 31;
        const bool valid = j < Tk && hist[(size_t)row * T + j] != 0;
        const float* kp = (j == Tk - 1) ? my + FIRA_D + h * FIRA_DH : kc_row + (size_t)min(j, Tk - 1) * FIRA_D + h * FIRA_DH;
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < FIRA_DH / 4; ++c) {
            const f32x4v k4 = *reinterpret_cast<const f32x4v*>(kp + c * 4);
            const f32x4v q4 = *reinterpret_cast<const f32x4v*>(&s_q[h * FIRA_DH + c * 4]);
            d = fmaf(q4.x, k4.x, d); d = fmaf(q4.y, k4.y, d); d = fmaf(q4.z, k4.z, d); d = fmaf(q4.w, k4.w, d);
        }
        const float sc = valid ? d * INV_SQRT_DH : -INFINITY;
        float mx = sc;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));   // (xor < 32 stays inside the head's half-wave)
        const float pr = valid ? __expf(sc - mx) : 0.f;
        float sum = pr;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) sum += __shfl_xor(sum, o, 64);
        s_p[h][j] = sum > 0.f ? pr / sum : 0.f;                    // (no valid key: zeros, as decode_attention writes)
    }
    __syncthreads();
    // ---- o = P V: thread = output element (head t >> 5)
    {
        const int h = t >> 5;
        float o = 0.f;
        int j = 0;
        for (; j + 4 <= Tk - 1; j += 4) {                          // cached values, four rows in flight
            const float v0 = vc_row[(size_t)j * FIRA_D + t], v1 = vc_row[(size_t)(j + 1) * FIRA_D + t];
            const float v2 = vc_row[(size_t)(j + 2) * FIRA_D + t], v3 = vc_row[(size_t)(j + 3) * FIRA_D + t];
            o = fmaf(s_p[h][j], v0, o); o = fmaf(s_p[h][j + 1], v1, o); o = fmaf(s_p[h][j + 2], v2, o); o = fmaf(s_p[h][j + 3], v3, o);
        }
        for (; j < Tk - 1; ++j) o = fmaf(s_p[h][j], vc_row[(size_t)j * FIRA_D + t], o);
        o = fmaf(s_p[h][Tk - 1], vnew, o);
        s_o[t] = o;
    }
    __syncthreads();
    // ---- s = o Wo^T + bo + x ;  xa = LayerNorm(s)       (thread = output column; WoT[k][n] coalesced over n)
    // y[n] = bias[n] + sum_k v[k] WT[k][n]: wave w takes k in [64 w, 64 w + 64), a lane four columns (16-byte loads, one
    // whole 1 KiB weight row per wave and instruction, 16 rows in flight); the four partial vectors meet in LDS.  (A thread
    // per column walking all 256 k with 8 loads in flight -- the first version -- was a chain of 32 dependent round trips per
    // product: the kernel was slower than the three launches it replaces.)
    auto matvec = [&](const float* __restrict__ WT, const float* v, float bias_n) -> float {
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};
        const float* wp = WT + (size_t)(wave * 64) * FIRA_D + lane * 4;
#pragma unroll
        for (int k0 = 0; k0 < 64; k0 += 16) {
            f32x4v w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) w[u] = *reinterpret_cast<const f32x4v*>(wp + (size_t)(k0 + u) * FIRA_D);
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += w[u] * v[wave * 64 + k0 + u];
        }
        __syncthreads();                                   // (s_part may still be read from the previous product)
        *reinterpret_cast<f32x4v*>(&s_part[wave][lane * 4]) = acc;
        __syncthreads();
        return bias_n + ((s_part[0][t] + s_part[1][t]) + (s_part[2][t] + s_part[3][t]));
    };
    const float sres = matvec(WoT, s_o, bo[t]) + xres[(size_t)row * FIRA_D + t];
    float part = wave_sum(sres);
    if (lane == 0) s_red[0][wave] = part;
    __syncthreads();
    const float mean = ((s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3])) * (1.0f / FIRA_D);
    const float dv = sres - mean;
    part = wave_sum(dv * dv);
    if (lane == 0) s_red[1][wave] = part;
    __syncthreads();
    const float var = ((s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3])) * (1.0f / FIRA_D);
    const float y = dv * (1.0f / sqrtf(var + 1e-5f)) * gamma[t] + beta[t];
    xa[(size_t)row * FIRA_D + t] = y;
    s_x[t] = y;
    __syncthreads();
    // ---- the cross attention's query: qc = xa Wq^T + bq
    qc[(size_t)row * FIRA_D + t] = matvec(WqT, s_x, bq[t]);
}
int decode_self_block(hipStream_t s, int BR, int Tk, int T, const float* qkv, float* Kc, float* Vc, const int32_t* hist,
                      const float* WoT, const float* bo, const float* xres, const float* gamma, const float* beta,
                      const float* WqT, const float* bq, float* xa, float* qc) {
    ProfScope prof(s, PROF_ATTN, 0.0);
    if (BR <= 0) return 0;
    FIRA_REQUIRE(Tk >= 1 && Tk <= 32 && Tk <= T, "decode_self_block: %d keys (at most 32)", Tk);
    hipLaunchKernelGGL(decode_self_block_kernel, dim3(BR), dim3(256), 0, s, Tk, T, qkv, Kc, Vc, hist, WoT, bo, xres, gamma, beta,
                       WqT, bq, xa, qc);
    FIRA_CHECK_LAUNCH("decode_self_block");
    return 0;
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
    hipLaunchKernelGGL((attention_fwd_kernel<NW_, BF_>), dim3(B * H), dim3(NW_ * 64), 0, s, H, Tq, Tk, Q, ldq, K, ldk, V, \
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
    hipLaunchKernelGGL((attention_bwd_kernel<NW_, BF_>), dim3(B * H), dim3(NW_ * 64), 0, s, H, Tq, Tk, Q, ldq, K, ldk, V, \
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
