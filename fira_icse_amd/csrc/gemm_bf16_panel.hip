// bf16 product for the model's dominant shape class, K = 256 (= d_model): C[M,N] (+)= A[M,256] . Bb[N,256]^T
// (A fp32 activations rounded to bf16 while staged, Bb a k-contiguous bf16 weight shadow, fp32 accumulate and store).
// Every nn.Linear forward with a 256-wide input (gnn_transformer.py:76,82,142-144,159,172,199; Model.py:16-19,54) and
// every data gradient with a 256-wide output gradient has this shape.
//
// Why not the tiled kernel of gemm_bf16.hip: with K = 256 a 64x64 tile is four K steps -- a pipeline that is all fill
// and drain -- and every tile re-reads its 64 KiB fp32 A panel; [24000,256]x[256,256] ran at 2.2 TB/s of operand
// traffic (22 us against an 8 us HBM floor), [17000,256]x[3072,256]^T at 1.3 TB/s.  Here the A panel is the stationary
// operand:
//   * a workgroup (4 waves) owns BM = 64 or 128 rows and a range of 64-column tiles.  The panel is read from HBM ONCE
//     (coalesced 1 KiB rows), rounded, and parked in LDS just long enough for each wave to pull the 16 MFMA A-fragments of
//     its 32-row slice into 64 VGPRs, where they stay for the whole sweep;
//   * the weight tiles (64 columns x 256 k, 32 KiB of bf16, L2-resident) stream through two LDS buffers that reuse the
//     panel's staging space: tile j+1 is in flight in registers while the 16 (or 32) MFMAs of tile j run; one barrier per
//     column tile, no barrier inside a tile, the K loop is fully unrolled; a tile's registers are requested two tiles
//     ahead, so waiting for them never waits for the C stores issued in between (vmcnt is one in-order counter for
//     loads and stores: with one stage every tile stalled on the previous tile's store acknowledgements);
//   * LDS rows are padded to 528 B (132 dwords: row r starts at bank 4r mod 64), which makes both the 16-byte staging
//     stores (16 lanes = 256 contiguous bytes of one row) and the ds_read_b128 fragment fetches (16 lanes = 16 rows at one
//     k chunk) conflict-free without a swizzle;
//   * work item = (row panel, column range), panel-major, mapped to workgroups with the same contiguous-range-per-XCD
//     rule as the other GEMMs, so the column ranges of one panel (and the weight tiles all of them stream) share an L2.
// Algorithmic traffic per launch: 4*M*256 (A, once) + 2*N*256 (weights, once from HBM) + 4*M*N (C).
#include "engine.h"
#include <algorithm>
#include <stdlib.h>

namespace fira {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int PK = 256;              // the K this kernel is specialised for
constexpr int PPITCH = 528;          // LDS row pitch in bytes (256 bf16 + 16 B pad)
constexpr int PBN = 64;              // columns per streamed weight tile

__device__ __forceinline__ uint32_t pack2(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));      // v_cvt_pk_bf16_f32 (RNE)
}

// All global traffic goes through buffer descriptors: out-of-range rows / columns (ragged last panel, ragged last tile)
// read zeros and drop their stores in hardware, so the kernel has no clamps and -- more important -- no branch around any
// memory instruction: hipcc's s_waitcnt insertion counts loads and stores in one in-order counter (vmcnt) and falls back to
// "wait for everything" as soon as an instruction may or may not have been issued.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32v4 __attribute__((__vector_size__(16)));
constexpr int BUF_FLAGS = 0x00020000;                     // gfx9 raw buffer, 32-bit data format
constexpr unsigned OOB = 0x80000000u;                     // added to the byte offset of a lane that must not touch memory
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, BUF_FLAGS);
}

// RT = 32-row slices per workgroup: 2 (BM 64: waves = 2 row slices x 2 column halves) or 4 (BM 128: wave = row slice,
// both 32-column halves of the tile).  EXTRA = the epilogue honours an output row map and / or a ReLU mask.
template <int RT, bool ACCUM, bool EXTRA>
__global__ __launch_bounds__(256) void gemm_bf16_k256_kernel(int M, int N, const float* __restrict__ A, int lda,
                                                             const uint16_t* __restrict__ Bb, int ldb,
                                                             float* __restrict__ C, int ldc,
                                                             const float* __restrict__ bias, int relu, int n_chunks,
                                                             int tiles_per_chunk, int n_items, int chunk,
                                                             const int32_t* __restrict__ c_rows,
                                                             const float* __restrict__ relu_mask) {
    constexpr int BM = 32 * RT, CG = 4 / RT, SN = 2 / CG;
    constexpr int LDS_BYTES = (BM > 2 * PBN ? BM : 2 * PBN) * PPITCH;
    __shared__ __attribute__((aligned(16))) char sm[LDS_BYTES];

    const int item = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (item >= n_items) return;
    const int panel = item / n_chunks, ch = item - panel * n_chunks;
    const int m0 = panel * BM;
    const int n_tiles = (N + PBN - 1) / PBN;
    const int jt0 = ch * tiles_per_chunk, nt = min(tiles_per_chunk, n_tiles - jt0);

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int kc = t & 31, rs = t >> 5;                    // staging role: 16-byte chunk kc of rows rs + 8*i
    const int rt = RT == 4 ? wave : (wave & 1), cg = RT == 4 ? 0 : (wave >> 1);      // compute role: row slice, column group

    const rsrc_t rA = make_rsrc(A, ((unsigned)(M - 1) * (unsigned)lda + PK) * 4u);
    const rsrc_t rB = make_rsrc(Bb, ((unsigned)(N - 1) * (unsigned)ldb + PK) * 2u);
    const rsrc_t rC = make_rsrc(C, 0x7fffffffu);
    const rsrc_t rBias = make_rsrc(bias ? (const void*)bias : (const void*)C, bias ? (unsigned)N * 4u : 0u);   // no bias: zeros
    const rsrc_t rRows = make_rsrc(c_rows ? (const void*)c_rows : (const void*)C, c_rows ? (unsigned)M * 4u : 0u);
    const rsrc_t rMask = make_rsrc(relu_mask ? (const void*)relu_mask : (const void*)C, relu_mask ? 0x7fffffffu : 0u);

    // ---- weight tiles travel global -> registers -> LDS; two register stages (tiles j+1 and j+2 in flight)
    const unsigned bvo = ((unsigned)rs * (unsigned)ldb + (unsigned)kc * 8u) * 2u;     // lane part of a tile's byte offsets
    auto fetch_b = [&](int jt, u32v4 (&br)[8]) __attribute__((always_inline)) {
        const unsigned base = (unsigned)(jt * PBN) * (unsigned)ldb * 2u + bvo;       // past the last tile: zeros
#pragma unroll
        for (int i = 0; i < 8; ++i) br[i] = __builtin_amdgcn_raw_buffer_load_b128(rB, base + (unsigned)(8 * i) * (unsigned)ldb * 2u, 0, 0);
    };
    auto put_b = [&](const u32v4 (&br)[8], int buf) __attribute__((always_inline)) {
        char* sb = sm + buf * PBN * PPITCH;
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<u32v4*>(sb + (rs + 8 * i) * PPITCH + kc * 16) = br[i];
    };
    // bias values travel one tile ahead as well (a load waited for right after the previous tile's stores would wait for
    // those stores: one in-order counter)
    const int colw = (cg * SN) * 32 + l31;                  // this lane's column inside a tile (+ 32 sn)
    auto fetch_bias = [&](int jt, float (&bv)[SN]) __attribute__((always_inline)) {
#pragma unroll
        for (int sn = 0; sn < SN; ++sn)
            bv[sn] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rBias, (unsigned)(jt * PBN + colw + 32 * sn) * 4u, 0, 0));
    };
    // Issue order of the prologue matters to the compiler's wait counts inside the loop: at the loop head it merges the
    // entry state with the back-edge state and keeps the SMALLER number of younger operations per pending load, so the
    // stage-1 request is issued before the panel's loads (which stand in for the steady state's C stores in that count).
    u32v4 br0[8], br1[8];
    float bv0[SN], bv1[SN];
    fetch_b(jt0, br0);
    fetch_b(jt0 + 1, br1);
    fetch_bias(jt0, bv0);

    // ---- A panel: HBM -> bf16 -> LDS, 64 rows per pass
#pragma unroll
    for (int pass = 0; pass < BM / 64; ++pass) {
        u32v4 v[8][2];
        const unsigned base = ((unsigned)(m0 + pass * 64 + rs) * (unsigned)lda + (unsigned)kc * 8u) * 4u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned o = base + (unsigned)(8 * i) * (unsigned)lda * 4u;
            v[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rA, o, 0, 0);
            v[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rA, o + 16u, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 lo = __builtin_bit_cast(f32x4, v[i][0]), hi = __builtin_bit_cast(f32x4, v[i][1]);
            u32v4 w = {pack2(lo.x, lo.y), pack2(lo.z, lo.w), pack2(hi.x, hi.y), pack2(hi.z, hi.w)};
            *reinterpret_cast<u32v4*>(sm + (pass * 64 + rs + 8 * i) * PPITCH + kc * 16) = w;
        }
    }
    __syncthreads();

    // ---- this wave's A fragments: 32 rows x 256 k in 64 VGPRs (MFMA step s takes k = 16 s + 8 (lane >> 5) .. + 7)
    bf16x8 a[16];
    {
        const char* sa = sm + (rt * 32 + l31) * PPITCH + kh * 16;
#pragma unroll
        for (int s = 0; s < 16; ++s) a[s] = *reinterpret_cast<const bf16x8*>(sa + s * 32);
    }
    __syncthreads();                                        // the staging space now belongs to the weight tiles
    put_b(br0, 0);
    __syncthreads();

    const int boff = (cg * 32 * SN + l31) * PPITCH + kh * 16;
    const int row_base = m0 + rt * 32 + 4 * kh;
    // output rows of this lane (fixed for the whole sweep): byte offset of the row start, OOB for rows past M
    unsigned crow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row_base + (r & 3) + 8 * (r >> 2);
        unsigned orow = (unsigned)row;
        if (EXTRA) {
            const unsigned mapped = __builtin_amdgcn_raw_buffer_load_b32(rRows, (unsigned)row * 4u, 0, 0);   // 0 without a map
            orow = c_rows ? mapped : (unsigned)row;
        }
        crow[r] = orow * (unsigned)ldc * 4u + (row < M ? 0u : OOB);
    }
    // One column tile: the tile after next requested into the stage freed one step ago | MFMAs on LDS[buf] | the next
    // tile's stage -> LDS[buf ^ 1] | C stores | barrier.  The stage written to LDS was requested BEFORE the previous
    // tile's stores, so its wait leaves those stores and the newest loads in flight: the sweep never stalls on a store
    // acknowledgement.
    auto tile_step = [&](int j, int buf, u32v4 (&br_next)[8], u32v4 (&br_free)[8], const float (&bv)[SN],
                         float (&bv_next)[SN]) __attribute__((always_inline)) {
        const int n0 = (jt0 + j) * PBN;
        fetch_b(jt0 + j + 2, br_free);                      // br_free went to LDS one step ago
        fetch_bias(jt0 + j + 1, bv_next);
        __builtin_amdgcn_sched_barrier(0);                  // the requests stay ahead of the MFMAs (the scheduler would sink
                                                            // them below the LDS refill to save registers)
        unsigned ccol[SN];
#pragma unroll
        for (int sn = 0; sn < SN; ++sn) {
            const int col = n0 + colw + 32 * sn;
            ccol[sn] = (unsigned)col * 4u + (col < N ? 0u : OOB);
        }
        float keep[SN][16];
        if (EXTRA) {
#pragma unroll
            for (int sn = 0; sn < SN; ++sn)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_base + (r & 3) + 8 * (r >> 2);
                    const unsigned mo = (row < M && !(ccol[sn] & OOB)) ? (unsigned)row * (unsigned)ldc * 4u + ccol[sn] : OOB;
                    const float mk = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rMask, mo, 0, 0));
                    keep[sn][r] = (relu_mask && !(mk > 0.f)) ? 0.f : 1.f;
                }
        }
        f32x16 acc[SN];
#pragma unroll
        for (int sn = 0; sn < SN; ++sn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[sn][r] = 0.f;
        const char* sb = sm + buf * PBN * PPITCH + boff;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int sn = 0; sn < SN; ++sn) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(sb + sn * 32 * PPITCH + s * 32);
                acc[sn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b, acc[sn], 0, 0, 0);
            }
        }
        put_b(br_next, buf ^ 1);                            // LDS[buf ^ 1] was last read one step ago (barrier since)
        // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int sn = 0; sn < SN; ++sn) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[sn][r] + bv[sn];
                const unsigned off = crow[r] + ccol[sn];    // either part may carry the OOB bit (both: wraps to in-range!)
                const unsigned o = ((crow[r] | ccol[sn]) & OOB) ? OOB : off;
                if (ACCUM) {
                    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, rC, o, 0, 0);    // one owner per element: C += v
                } else {
                    v = relu ? fmaxf(v, 0.f) : v;
                    if (EXTRA) v *= keep[sn][r];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rC, o, 0, 0);
                }
            }
        }
        __syncthreads();
    };
    for (int j = 0; j < nt; j += 2) {                       // stage / buffer roles are static inside the pair
        tile_step(j, 0, br1, br0, bv0, bv1);
        if (j + 1 >= nt) break;
        tile_step(j + 1, 1, br0, br1, bv1, bv0);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// y = LayerNorm(dropout(X Wb^T + b [+ r c^T]) + res) * gamma + beta in ONE launch (bf16 mode; N = 256, K = 256): the panel
// kernel above with BM = 64 already owns 64 COMPLETE output rows, and with the bf16 matrix pipe the 64x256x256 block is
// 0.85 us of MFMA time -- the reason the fp32 twin (linear_ln.hip) loses does not apply.  The four column tiles accumulate
// into 4 x 16 registers per lane; the row statistics are reduced over the 32 lanes of a half-wave with xor shuffles and
// over the two column-half waves of a row slice through LDS (two passes: mean, then the centred sum of squares, like
// add_layernorm_fwd_kernel); dropout uses the same element indices (row * 256 + col), so the backward kernels and masks
// are unchanged.  No store is issued before the last weight tile has been consumed.
__global__ __launch_bounds__(256, 2) void linear_ln_bf16_kernel(int M, const float* __restrict__ A, int lda,
                                                             const uint16_t* __restrict__ Bb, int ldb,
                                                             const float* __restrict__ bias, const float* __restrict__ res,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ sum, float* __restrict__ y,
                                                             float* __restrict__ stats, float p, float inv_keep,
                                                             uint64_t seed, uint32_t site,
                                                             const int32_t* __restrict__ y_rows,
                                                             const float* __restrict__ r1_row,
                                                             const float* __restrict__ r1_col, int n_items, int chunk) {
    constexpr int BM = 64;
    __shared__ __attribute__((aligned(16))) char sm[2 * PBN * PPITCH];
    __shared__ float red[2][2][2][32];                    // [pass][row slice][column half][row in slice]

    const int item = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (item >= n_items) return;
    const int m0 = item * BM;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const int kc = t & 31, rs = t >> 5;
    const int rt = wave & 1, cg = wave >> 1;

    const rsrc_t rA = make_rsrc(A, ((unsigned)(M - 1) * (unsigned)lda + PK) * 4u);
    const rsrc_t rB = make_rsrc(Bb, (255u * (unsigned)ldb + PK) * 2u);
    const unsigned bvo = ((unsigned)rs * (unsigned)ldb + (unsigned)kc * 8u) * 2u;
    auto fetch_b = [&](int jt, u32v4 (&br)[8]) __attribute__((always_inline)) {
        const unsigned base = (unsigned)(jt * PBN) * (unsigned)ldb * 2u + bvo;       // past the last tile: zeros
#pragma unroll
        for (int i = 0; i < 8; ++i) br[i] = __builtin_amdgcn_raw_buffer_load_b128(rB, base + (unsigned)(8 * i) * (unsigned)ldb * 2u, 0, 0);
    };
    auto put_b = [&](const u32v4 (&br)[8], int buf) __attribute__((always_inline)) {
        char* sb = sm + buf * PBN * PPITCH;
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<u32v4*>(sb + (rs + 8 * i) * PPITCH + kc * 16) = br[i];
    };
    u32v4 br0[8], br1[8];
    fetch_b(0, br0);
    fetch_b(1, br1);
    {
        u32v4 v[8][2];
        const unsigned base = ((unsigned)(m0 + rs) * (unsigned)lda + (unsigned)kc * 8u) * 4u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned o = base + (unsigned)(8 * i) * (unsigned)lda * 4u;
            v[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rA, o, 0, 0);
            v[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rA, o + 16u, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 lo = __builtin_bit_cast(f32x4, v[i][0]), hi = __builtin_bit_cast(f32x4, v[i][1]);
            u32v4 w = {pack2(lo.x, lo.y), pack2(lo.z, lo.w), pack2(hi.x, hi.y), pack2(hi.z, hi.w)};
            *reinterpret_cast<u32v4*>(sm + (rs + 8 * i) * PPITCH + kc * 16) = w;
        }
    }
    __syncthreads();
    bf16x8 a[16];
    {
        const char* sa = sm + (rt * 32 + l31) * PPITCH + kh * 16;
#pragma unroll
        for (int s = 0; s < 16; ++s) a[s] = *reinterpret_cast<const bf16x8*>(sa + s * 32);
    }
    __syncthreads();
    put_b(br0, 0);
    __syncthreads();

    const int boff = (cg * 32 + l31) * PPITCH + kh * 16;
    f32x16 acc[4];
    auto tile_step = [&](int j, int buf, u32v4 (&br_next)[8], u32v4 (&br_free)[8]) __attribute__((always_inline)) {
        fetch_b(j + 2, br_free);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const char* sb = sm + buf * PBN * PPITCH + boff;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(sb + s * 32);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b, acc[j], 0, 0, 0);
        }
        put_b(br_next, buf ^ 1);
        __syncthreads();
    };
    tile_step(0, 0, br1, br0);
    tile_step(1, 1, br0, br1);
    tile_step(2, 0, br1, br0);
    tile_step(3, 1, br0, br1);

    // ---- epilogue: x = dropout(acc + b + r c^T) + res; two-pass row statistics; sum / y / stats ----------------------
    const int row_base = m0 + rt * 32 + 4 * kh;
    const rsrc_t rBias = make_rsrc(bias ? (const void*)bias : (const void*)A, bias ? 1024u : 0u);
    const rsrc_t rRes = make_rsrc(res ? (const void*)res : (const void*)A, res ? (unsigned)M * 1024u : 0u);
    const rsrc_t rR1 = make_rsrc(r1_row ? (const void*)r1_row : (const void*)A, r1_row ? (unsigned)M * 4u : 0u);
    const rsrc_t rC1 = make_rsrc(r1_row ? (const void*)r1_col : (const void*)A, r1_row ? 1024u : 0u);
    const rsrc_t rMap = make_rsrc(y_rows ? (const void*)y_rows : (const void*)A, y_rows ? (unsigned)M * 4u : 0u);
    const rsrc_t rSum = make_rsrc(sum ? (const void*)sum : (const void*)A, sum ? (unsigned)M * 1024u : 0u);
    const rsrc_t rStats = make_rsrc(stats ? (const void*)stats : (const void*)A, stats ? (unsigned)M * 8u : 0u);
    const rsrc_t rY = make_rsrc(y, 0x7fffffffu);
    float bv[4], c1[4], gm[4], bt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned co = (unsigned)(j * 64 + cg * 32 + l31) * 4u;
        bv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rBias, co, 0, 0));
        c1[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rC1, co, 0, 0));
        gm[j] = gamma[j * 64 + cg * 32 + l31];
        bt[j] = beta[j * 64 + cg * 32 + l31];
    }
    float w1[16];
    unsigned orow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row_base + (r & 3) + 8 * (r >> 2);
        w1[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rR1, (unsigned)row * 4u, 0, 0));
        const unsigned mapped = __builtin_amdgcn_raw_buffer_load_b32(rMap, (unsigned)row * 4u, 0, 0);
        orow[r] = row < M ? (y_rows ? mapped : (unsigned)row) * 1024u : OOB;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = j * 64 + cg * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row_base + (r & 3) + 8 * (r >> 2);
            const float rv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rRes, (unsigned)row * 1024u + (unsigned)col * 4u, 0, 0));
            float x = fmaf(w1[r], c1[j], acc[j][r] + bv[j]);
            if (p > 0.f) x *= dropout_scale(seed, site, (uint32_t)row * FIRA_D + col, p, inv_keep);     // wave-uniform branch
            acc[j][r] = x + rv;
        }
    }
    // pass 1: row means
    float mean[16], rstd[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
        v = sum16(v);                                     // DPP (common.h): the 16-lane row sums, then rows 1 / 3 take
        v += dpp_take<DPP_BCAST15, 0xa>(0.f, v);          // lane 15 of rows 0 / 2: lanes 16..31 and 48..63 hold the 32-lane sums
        mean[r] = v;
    }
    if (l31 == 16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[0][rt][cg][(r & 3) + 8 * (r >> 2) + 4 * kh] = mean[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lr = (r & 3) + 8 * (r >> 2) + 4 * kh;
        mean[r] = (red[0][rt][0][lr] + red[0][rt][1][lr]) * (1.0f / FIRA_D);
    }
    // pass 2: centred sums of squares
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float d = acc[j][r] - mean[r]; v = fmaf(d, d, v); }
        v = sum16(v);                                     // DPP (common.h): the 16-lane row sums, then rows 1 / 3 take
        v += dpp_take<DPP_BCAST15, 0xa>(0.f, v);          // lane 15 of rows 0 / 2: lanes 16..31 and 48..63 hold the 32-lane sums
        rstd[r] = v;
    }
    if (l31 == 16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[1][rt][cg][(r & 3) + 8 * (r >> 2) + 4 * kh] = rstd[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lr = (r & 3) + 8 * (r >> 2) + 4 * kh;
        const float var = (red[1][rt][0][lr] + red[1][rt][1][lr]) * (1.0f / FIRA_D);
        rstd[r] = 1.0f / sqrtf(var + 1e-5f);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned co = (unsigned)(j * 64 + cg * 32 + l31) * 4u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row_base + (r & 3) + 8 * (r >> 2);
            const unsigned so = row < M ? (unsigned)row * 1024u + co : OOB;
            const float x = acc[j][r];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), rSum, so, 0, 0);
            const float yv = (x - mean[r]) * rstd[r] * gm[j] + bt[j];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yv), rY, orow[r] == OOB ? OOB : orow[r] + co, 0, 0);
        }
    }
    if (l31 == 0 && cg == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row_base + (r & 3) + 8 * (r >> 2);
            const unsigned so = row < M ? (unsigned)row * 8u : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mean[r]), rStats, so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rstd[r]), rStats, so == OOB ? OOB : so + 4u, 0, 0);
        }
    }
}

// true if the fused kernel took the call (bf16 mode, K = 256, a 256-row k-contiguous bf16 weight shadow)
bool linear_ln_bf16_try(hipStream_t s, int M, int K, const float* X, int ldx, const uint16_t* Wb, int ldb, const float* bias,
                        const float* res, const float* gamma, const float* beta, float* sum, float* y, float* stats,
                        float dropout, uint64_t seed, uint32_t site, const int32_t* y_rows, const float* r1_row,
                        const float* r1_col, int* rc, bool force) {
    *rc = 0;
    // FIRA_FUSED_LN_BF16: 0 never | 1 every K = 256 block | 2 only the encoder-sized ones (M >= 4096) | 3 only the decoder-sized
    // Measured on one box (profiles/r2_probes.md): 2 is +0.5 % on the batch-64 step, 3 is -1.8 % (M = 1 920 rows are 30
    // workgroups here against 480 in the latency GEMM), 1 is -2.3 %.
    static const int mode = [] { const char* e = getenv("FIRA_FUSED_LN_BF16"); return e ? atoi(e) : 2; }();
    if (!force && (mode == 0 || (mode == 2 && M < 4096) || (mode == 3 && M >= 4096))) return false;
    if (K != PK || M < 64 || M >= (1 << 21) || ldx % 4 || ldb % 8 || ((uintptr_t)X % 16) || ((uintptr_t)Wb % 16) ||
        (long)M * ldx >= (1L << 29))
        return false;
    ProfScope prof(s, PROF_GEMM, 2.0 * M * 256.0 * K, 4.0 * ((double)M * K + 3.0 * M * 256.0) + 2.0 * 256.0 * K);
    const float inv_keep = dropout > 0.f ? 1.0f / (1.0f - dropout) : 1.0f;
    const int n_items = cdiv(M, 64), chunk = cdiv(n_items, 8);
    hipLaunchKernelGGL(linear_ln_bf16_kernel, dim3(8 * chunk), dim3(256), 0, s, M, X, ldx, Wb, ldb, bias, res, gamma, beta, sum, y,
                       stats, dropout, inv_keep, seed, site, y_rows, r1_row, r1_col, n_items, chunk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) *rc = set_err("linear_ln_bf16: %s", hipGetErrorString(e));
    return true;
}

// true if this kernel took the call
bool gemm_bf16_k256_try(hipStream_t s, int M, int N, int K, const float* A, int lda, const uint16_t* Bb, int ldb, float* C,
                        int ldc, const float* bias, int flags, const int32_t* c_rows, const float* relu_mask, int* rc) {
    *rc = 0;
    static const int mode = [] { const char* e = getenv("FIRA_PANEL_GEMM"); return e ? atoi(e) : 1; }();   // A/B switch
    if (!mode || K != PK || M < 64 || N < 32 || lda % 4 || ldb % 8 || ((uintptr_t)A % 16) || ((uintptr_t)Bb % 16)) return false;
    // BM 128 halves the weight-tile traffic per output; it needs enough rows to still fill the chip
    const bool big = mode == 2 ? false : mode == 3 ? true : (N >= 512 && (long)M * N >= (4L << 20));
    const int bm = big ? 128 : 64;
    const int panels = cdiv(M, bm), n_tiles = cdiv(N, PBN);
    const int slots = 512;                                   // 2 resident workgroups per CU
    int n_chunks = std::max(1, std::min(n_tiles, slots / std::max(1, panels)));
    const int tpc = cdiv(n_tiles, n_chunks);
    n_chunks = cdiv(n_tiles, tpc);
    const int n_items = panels * n_chunks, chunk = cdiv(n_items, 8);
    const int relu = flags & FIRA_GEMM_RELU;
    const bool accum = flags & FIRA_GEMM_ACCUM;
    // 31-bit byte offsets inside (bit 31 marks lanes that must not touch memory); with a row map the caller vouches for C
    if ((long)M * ldc >= (1L << 29) || (long)N * ldb >= (1L << 30) || (long)M * lda >= (1L << 29)) return false;
    if (accum && (relu || relu_mask)) return false;         // C += v is an atomic add here; relu(C + v) is not expressible
    const bool extra = c_rows != nullptr || relu_mask != nullptr;
#define FIRA_LAUNCH(RT, AC, EX)                                                                                          \
    hipLaunchKernelGGL((gemm_bf16_k256_kernel<RT, AC, EX>), dim3(8 * chunk), dim3(256), 0, s, M, N, A, lda, Bb, ldb, C, ldc, \
                       bias, relu, n_chunks, tpc, n_items, chunk, c_rows, relu_mask)
#define FIRA_PICK(RT)                                                                                                    \
    do {                                                                                                                 \
        if (accum) { if (extra) FIRA_LAUNCH(RT, true, true); else FIRA_LAUNCH(RT, true, false); }                        \
        else { if (extra) FIRA_LAUNCH(RT, false, true); else FIRA_LAUNCH(RT, false, false); }                            \
    } while (0)
    if (big) FIRA_PICK(4); else FIRA_PICK(2);
#undef FIRA_PICK
#undef FIRA_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) *rc = set_err("gemm_bf16_k256: %s", hipGetErrorString(e));
    return true;
}

}  // namespace fira

// y = LayerNorm(dropout(X Wb^T + bias) + res) * gamma + beta with Wb a [256, 256] bf16 weight shadow (row pitch ldb)
extern "C" int fira_linear_layernorm_bf16_fwd(void* stream, int M, const float* X, int ldx, const uint16_t* Wb, int ldb,
                                              const float* bias, const float* res, const float* gamma, const float* beta,
                                              float* sum, float* y, float* stats, float dropout, uint64_t seed,
                                              uint32_t stream_id) {
    int rc;
    FIRA_REQUIRE(X && Wb && gamma && beta && y, "fira_linear_layernorm_bf16_fwd: null argument");
    if (!fira::linear_ln_bf16_try((hipStream_t)stream, M, 256, X, ldx, Wb, ldb, bias, res, gamma, beta, sum, y, stats, dropout,
                                  seed, stream_id, nullptr, nullptr, nullptr, &rc, true))
        return fira::set_err("fira_linear_layernorm_bf16_fwd: unsupported shape / alignment (M=%d ldx=%d ldb=%d; needs M >= 64)", M,
                             ldx, ldb);
    return rc;
}
