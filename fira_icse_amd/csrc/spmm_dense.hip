// Block-dense aggregation  Z = A_hat · H  on the matrix cores, for graph batches that are dense enough that a CSR gather
// loses to the plain product the reference actually runs (gnn_transformer.py:80: torch.bmm(edge.float(), x) over a
// dense [B,N,N] adjacency).  BASELINE config 5 (128 graphs x 512 nodes, ~22 % dense) is that case: 7.6 M gathered rows of
// 1 KiB per launch keep the CSR kernels on the L2 / LDS gather rate (spmm.hip: 180-300 us), while the same product is
// 17 GFLOP -- 7 us of bf16 MFMA time, 110 us of fp32 MFMA time -- over 195 MB of compulsory traffic (24 us of HBM).
//
// The input stays the engine's block-diagonal CSR (no dense adjacency in HBM): a workgroup owns R consecutive rows of
// ONE graph and all 256 feature columns, and
//   1. densifies its R x N slice of A_hat into LDS (zero fill, then one LDS store per CSR entry; a wave owns consecutive rows
//      and requests all their entries in one round trip -- densify_rows; k-contiguous rows = MFMA A-operand layout);
//   2. walks the graph's N feature rows in chunks: every wavefront owns 64 output columns, so each H element is fetched
//      ONCE per workgroup, straight into registers in MFMA B-fragment layout (lane = column, registers = k: dword loads
//      whose 32-lane halves cover whole 128-byte row segments), one chunk ahead of the MFMAs; A fragments come from LDS
//      (row pitch = 16 bytes mod 256: the 16-lane groups of a ds_read_b128 cover all 64 banks);
//   3. stores the fp32 tile.
// The other row blocks of a graph re-read its H from the XCD's L2: the workgroup -> (graph, row block) order is
// XCD-aware (block b runs on XCD b % 8; every XCD gets a contiguous range of graphs).
//
//   spmm_dense_f32_kernel   R = 32, v_mfma_f32_32x32x2_f32: the reference's fp32 arithmetic (sum order differs only)
//   spmm_dense_bf16_kernel  R = 64, v_mfma_f32_32x32x16_bf16: A_hat and H rounded to bf16 (RNE), fp32 accumulate --
//                           torch.autocast's bmm, i.e. the aggregation of BASELINE configs[2]'s dtype
//
// Precondition (as data.py / graphs.py build their CSR): inside a row the column ids are sorted; equal neighbours are
// summed.  Columns outside the row's own graph block are ignored (the adjacency is block-diagonal).
#include "common.h"
#include "epilogue.h"
#include <stdlib.h>

namespace fira {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_bf16_2(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));      // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ void store_elem(char* p, float v, float) { *reinterpret_cast<float*>(p) = v; }
__device__ __forceinline__ void store_elem(char* p, float v, uint16_t) {
    *reinterpret_cast<uint16_t*>(p) = (uint16_t)(pack_bf16_2(v, 0.f) & 0xffffu);
}

// R x KP slice of the adjacency (rows r0.. of the graph whose first node is row0) -> LDS tile, element type T.
// A wave owns R / (NT / 64) consecutive rows of the slice and requests ALL their entries before it uses any: DN_SLOTS slots of
// 64 entries per row, every lane of every slot a buffer load (lanes without an entry carry the descriptor's out-of-range
// offset: no request, no branch), one round trip for the slice.  A slot is then one row's entries in lane order, so the row is
// known (no search) and the neighbour's column is the adjacent lane's (wave_shl / wave_shr DPP).  Fast path of a slot -- no
// entry has an equal successor (one compare + ballot) -- every entry stores its value; otherwise the LAST entry of a run of
// equal columns (sorted rows) stores the run's sum, the earlier ones walked back in global memory.  Rows longer than the
// slots finish in a tail loop.  (Round 5's form -- threads striding over the whole slice, eight entries in flight, the row
// found by walking the LDS copy of the row offsets, the predecessor's column by a dependent load -- cost 35 us of the bf16
// launch's 82 on config 5; a per-entry search over the wave's 16 row boundaries, 16.5 us per 128-row slice: this form is
// bound by its ~1 000 instructions per wave.)  Entry offsets travel as 31-bit byte offsets of a buffer descriptor: fewer than
// 2^29 entries per launch (a block-diagonal batch of <= 512-node graphs holds at most 512 per row: a million rows).
constexpr int DN_SLOTS = 3;
template <typename T, int R, int NT>
__device__ __forceinline__ void densify_rows(char* tile, int* sm_rp, int pitch, int graph_rows, int row0, int r0,
                                             const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                             const float* __restrict__ val) {
    constexpr int RW = R / (NT / 64);                                // rows of a wave
    static_assert(R % (NT / 64) == 0 && NT > R, "densify_rows: rows per wave");
    const int nr = min(R, graph_rows - r0);                          // rows of this slice
    if ((int)threadIdx.x <= nr) sm_rp[threadIdx.x] = rowptr[row0 + r0 + threadIdx.x];
    const int n16 = R * pitch / 16;
    for (int i = threadIdx.x; i < n16; i += NT) reinterpret_cast<uint4*>(tile)[i] = uint4{0u, 0u, 0u, 0u};
    __syncthreads();
    const int lane = threadIdx.x & 63, w0 = (threadIdx.x >> 6) * RW;
    const rsrc_t rC = buf_rsrc(col, 0x7fffffffu), rV = buf_rsrc(val, 0x7fffffffu);
    int pc[RW][DN_SLOTS];
    float pv[RW][DN_SLOTS];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {                                // (rows past the slice: rb == re, no lane has an entry)
        const int rb = sm_rp[min(w0 + rr, nr)], re = sm_rp[min(w0 + rr + 1, nr)];
#pragma unroll
        for (int sl = 0; sl < DN_SLOTS; ++sl) {
            const int e = rb + sl * 64 + lane;
            const unsigned off = e < re ? (unsigned)e * 4u : FIRA_OOB;
            pc[rr][sl] = (int)__builtin_amdgcn_raw_buffer_load_b32(rC, off, 0, 0);     // (no arithmetic on it here: that would wait)
            pv[rr][sl] = buf_load_f32(rV, off);
        }
    }
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        if (w0 + rr >= nr) break;                                    // (wave-uniform)
        const int rb = sm_rp[w0 + rr], re = sm_rp[w0 + rr + 1];
        char* rowp = tile + (size_t)(w0 + rr) * pitch;
#pragma unroll
        for (int sl = 0; sl < DN_SLOTS; ++sl) {
            if (rb + sl * 64 >= re) break;                           // (wave-uniform)
            const int c = pc[rr][sl] - row0;
            const int nx0 = sl + 1 < DN_SLOTS ? __builtin_amdgcn_readlane(pc[rr][sl + 1 < DN_SLOTS ? sl + 1 : sl], 0) - row0 : -2;
            int cn = __builtin_amdgcn_update_dpp(nx0, c, 0x130, 0xf, 0xf, false);          // wave_shl:1: lane i <- lane i + 1
            const int e = rb + sl * 64 + lane;
            const bool more = sl + 1 == DN_SLOTS && rb + DN_SLOTS * 64 < re;               // (wave-uniform)
            if (__ballot(e + 1 < re && cn == c) == 0ull && !more) {
                if (e < re && (unsigned)c < (unsigned)graph_rows) store_elem(rowp + (size_t)c * sizeof(T), pv[rr][sl], T());
                continue;
            }
            const int pv0 = sl > 0 ? __builtin_amdgcn_readlane(pc[rr][sl > 0 ? sl - 1 : 0], 63) - row0 : -2;
            const int cp = __builtin_amdgcn_update_dpp(pv0, c, 0x138, 0xf, 0xf, false);    // wave_shr:1: lane i <- lane i - 1
            if (more && lane == 63) cn = col[e + 1] - row0;
            if (e >= re || (e + 1 < re && cn == c) || (unsigned)c >= (unsigned)graph_rows) continue;
            float sum = pv[rr][sl];
            if (e > rb && cp == c)
                for (int k = e - 1; k >= rb && col[k] - row0 == c; --k) sum += val[k];
            store_elem(rowp + (size_t)c * sizeof(T), sum, T());
        }
        for (int base = rb + DN_SLOTS * 64; base < re; base += 64) {  // rows longer than the slots (rare)
            const int e = base + lane;
            const int c = e < re ? col[e] - row0 : -1;
            const int cn = e + 1 < re ? col[e + 1] - row0 : -2;
            if (e >= re || cn == c || (unsigned)c >= (unsigned)graph_rows) continue;
            float sum = val[e];
            for (int k = e - 1; k >= rb && col[k] - row0 == c; --k) sum += val[k];
            store_elem(rowp + (size_t)c * sizeof(T), sum, T());
        }
    }
    __syncthreads();
}

__device__ __forceinline__ int acc_row32(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

// ---------------------------------------------------------------------------------------------------------------------
// bf16: RT row tiles (32 RT rows) per workgroup of NW waves; a wave owns CT = 8 / NW column tiles of all RT row tiles.
// The first chunk of H is requested before the adjacency slice is densified (it does not depend on it).
constexpr int DB_KC = 64;
template <int RT, int NW>
__global__ __launch_bounds__(NW * 64) void spmm_dense_bf16_kernel(int graph_rows, int KP, const int32_t* __restrict__ rowptr,
                                                                  const int32_t* __restrict__ col,
                                                                  const float* __restrict__ val,
                                                                  const float* __restrict__ X, int ldx,
                                                                  float* __restrict__ Y, int ldy, int nrb, int n_items,
                                                                  int chunk) {
    constexpr int CT = 8 / NW, R = 32 * RT;
    extern __shared__ __attribute__((aligned(16))) char tile[];
    __shared__ int sm_rp[R + 1];
    const int item = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (item >= n_items) return;
    const int g = item / nrb, rb = item - g * nrb;
    const int row0 = g * graph_rows, r0 = rb * R;
    const int pitch = KP * 2 + 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kg = lane >> 5;
    const rsrc_t rX = buf_rsrc(X + (size_t)row0 * ldx, (unsigned)graph_rows * (unsigned)ldx * 4u);   // rows >= graph_rows read 0
    const unsigned ldxb = (unsigned)ldx * 4u;
    const unsigned nb = (unsigned)(wave * (CT * 32) + l31) * 4u;
    // B fragments of one 64-row chunk of H: [column tile][k step] x 8 k values (k = chunk*64 + step*16 + kg*8 + i), in a
    // ring of NB register buffers: NB - 1 chunks are in flight while one is multiplied
    constexpr int NB = 2;             // (3 buffers: no gain measured, and 96 + 64 accumulator registers spill at two waves per SIMD)
    float raw[NB][CT][4][8];
#define FIRA_DB_FETCH(kc, buf)                                                                                     \
    _Pragma("unroll") for (int ct = 0; ct < CT; ++ct)                                                              \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                               \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                  \
        raw[buf][ct][ks][i] = buf_load_f32(rX, (unsigned)((kc) * DB_KC + ks * 16 + kg * 8 + i) * ldxb + nb + ct * 128u)
#pragma unroll
    for (int j = 0; j < NB - 1; ++j) FIRA_DB_FETCH(j, j);
    densify_rows<uint16_t, R, NW * 64>(tile, sm_rp, pitch, graph_rows, row0, r0, rowptr, col, val);

    f32x16 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const char* a_base = tile + (size_t)l31 * pitch + kg * 16;
    const int nkc = KP / DB_KC;
    for (int kc0 = 0; kc0 < nkc; kc0 += NB) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {                                // buffer j holds chunk kc0 + j
            const int kc = kc0 + j;
            if (kc >= nkc) break;                                     // (uniform)
            bf16x8 b[CT][4];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    uint4 u;
                    u.x = pack_bf16_2(raw[j][ct][ks][0], raw[j][ct][ks][1]); u.y = pack_bf16_2(raw[j][ct][ks][2], raw[j][ct][ks][3]);
                    u.z = pack_bf16_2(raw[j][ct][ks][4], raw[j][ct][ks][5]); u.w = pack_bf16_2(raw[j][ct][ks][6], raw[j][ct][ks][7]);
                    b[ct][ks] = __builtin_bit_cast(bf16x8, u);
                }
            FIRA_DB_FETCH(kc + NB - 1, (j + NB - 1) % NB);            // past the last chunk: out of range, reads zeros
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(a_base + (size_t)rt * 32 * pitch + (kc * DB_KC + ks * 16) * 2);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[ct][ks], acc[rt][ct], 0, 0, 0);
                }
            }
        }
    }
#undef FIRA_DB_FETCH
    const rsrc_t rY = buf_rsrc(Y + (size_t)row0 * ldy, (unsigned)graph_rows * (unsigned)ldy * 4u);   // rows >= graph_rows: dropped
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned row = (unsigned)(r0 + rt * 32 + acc_row32(r, kg));
                const float v = acc[rt][ct][r];          // (bit_cast of a vector ELEMENT lvalue reads element 0: copy first)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rY,
                                                      row * (unsigned)ldy * 4u + nb + ct * 128u, 0, 0);
            }
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32: the same decomposition on v_mfma_f32_32x32x2_f32.  Inside a 32-wide k chunk MFMA step s uses k = (lane>>5)*16 + s
// for both operands (the order of an MFMA chain's reduction index is free): the A fragment is 16 contiguous floats of
// the LDS row (4 x ds_read_b128), the B fragment 16 dword loads of 128-byte row segments.
constexpr int DF_KC = 32;
template <int RT, int NW>
__global__ __launch_bounds__(NW * 64) void spmm_dense_f32_kernel(int graph_rows, int KP, const int32_t* __restrict__ rowptr,
                                                                 const int32_t* __restrict__ col,
                                                                 const float* __restrict__ val,
                                                                 const float* __restrict__ X, int ldx,
                                                                 float* __restrict__ Y, int ldy, int nrb, int n_items,
                                                                 int chunk) {
    constexpr int CT = 8 / NW, R = 32 * RT;
    extern __shared__ __attribute__((aligned(16))) char tile[];
    __shared__ int sm_rp[R + 1];
    const int item = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (item >= n_items) return;
    const int g = item / nrb, rb = item - g * nrb;
    const int row0 = g * graph_rows, r0 = rb * R;
    const int pitch = KP * 4 + 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kh = lane >> 5;
    const rsrc_t rX = buf_rsrc(X + (size_t)row0 * ldx, (unsigned)graph_rows * (unsigned)ldx * 4u);
    const unsigned ldxb = (unsigned)ldx * 4u;
    const unsigned nb = (unsigned)(wave * (CT * 32) + l31) * 4u;
    constexpr int NB = 2;                                            // ring of chunk buffers: see the bf16 kernel
    float raw[NB][CT][16];
#define FIRA_DF_FETCH(kc, buf)                                                                                     \
    _Pragma("unroll") for (int ct = 0; ct < CT; ++ct)                                                              \
    _Pragma("unroll") for (int s = 0; s < 16; ++s)                                                                 \
        raw[buf][ct][s] = buf_load_f32(rX, (unsigned)((kc) * DF_KC + kh * 16 + s) * ldxb + nb + ct * 128u)
#pragma unroll
    for (int j = 0; j < NB - 1; ++j) FIRA_DF_FETCH(j, j);
    densify_rows<float, R, NW * 64>(tile, sm_rp, pitch, graph_rows, row0, r0, rowptr, col, val);

    f32x16 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const char* a_base = tile + (size_t)l31 * pitch + kh * 64;
    const int nkc = KP / DF_KC;
    for (int kc0 = 0; kc0 < nkc; kc0 += NB) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int kc = kc0 + j;
            if (kc >= nkc) break;                                     // (uniform)
            float b[CT][16];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int s = 0; s < 16; ++s) b[ct][s] = raw[j][ct][s];
            FIRA_DF_FETCH(kc + NB - 1, (j + NB - 1) % NB);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float a[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(a_base + (size_t)rt * 32 * pitch + (size_t)kc * DF_KC * 4 + q * 16);
                    a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
                }
#pragma unroll
                for (int s = 0; s < 16; ++s)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[ct][s], acc[rt][ct], 0, 0, 0);
            }
        }
    }
#undef FIRA_DF_FETCH
    const rsrc_t rY = buf_rsrc(Y + (size_t)row0 * ldy, (unsigned)graph_rows * (unsigned)ldy * 4u);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned row = (unsigned)(r0 + rt * 32 + acc_row32(r, kh));
                const float v = acc[rt][ct][r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rY,
                                                      row * (unsigned)ldy * 4u + nb + ct * 128u, 0, 0);
            }
}


template <typename K>
static int raise_lds(K kernel, bool* done) {
    if (*done) return 0;
    const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    FIRA_REQUIRE(e == hipSuccess, "csr_spmm: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    *done = true;
    return 0;
}

// bf16 != 0: bf16 operands (fp32 accumulate); otherwise fp32 MFMA.  Shape of the workgroup (FIRA_SPMM_DENSE_SHAPE, A/B):
//   0 (default) = 128-row (bf16) / 64-row (fp32) slices, 8 waves, one workgroup per CU (H re-read from L2 4x / 8x per
//   graph); 1 = 64 / 32 rows, 4 waves, two workgroups per CU (8x / 16x; graphs of <= 64 / 32 rows always take this form).
// Measured (profiles/r3_spmm_crossover.md): shape 0 is 5-10 % ahead.  Tried and dropped: deeper rings of H chunks (3-4
// in flight: no gain, the bf16 form spills), and a 16-byte-access form (lane j of column tile t = column 4j + t, so one
// float4 load feeds four column tiles and the stores are 16 bytes wide: 4 waves at one per SIMD -- 20-40 % SLOWER).
int csr_spmm_dense(hipStream_t s, int n_rows, const int32_t* rowptr, const int32_t* col, const float* val, const float* X,
                   int ldx, float* Y, int ldy, int graph_rows, int bf16) {
    if (n_rows <= 0) return 0;
    FIRA_REQUIRE(graph_rows > 0 && graph_rows <= 512 && n_rows % graph_rows == 0,
                 "csr_spmm: the block-dense variants need rows-per-graph (%d) <= 512 dividing n_rows", graph_rows);
    FIRA_REQUIRE(ldx >= FIRA_D && ldy >= FIRA_D && (long)graph_rows * ldx * 4 < (1L << 31) && (long)graph_rows * ldy * 4 < (1L << 31),
                 "csr_spmm: bad leading dimensions");
    static const int shape = [] { const char* e = getenv("FIRA_SPMM_DENSE_SHAPE"); return e ? atoi(e) : 0; }();
    const int KP = cdiv(graph_rows, 64) * 64;
    const bool big = shape == 0 && graph_rows > (bf16 ? 64 : 32);
    const int R = bf16 ? (big ? 128 : 64) : (big ? 64 : 32);
    const int nrb = cdiv(graph_rows, R);
    const int n_items = (n_rows / graph_rows) * nrb;
    const int chunk = cdiv(n_items, 8);
    const size_t lds = (size_t)R * (bf16 ? KP * 2 + 16 : KP * 4 + 16);
    static bool a0 = false, a1 = false, a2 = false, a3 = false;
    ProfScope prof(s, PROF_SPMM, 4.0 * (n_rows + 1) + 2.0 * n_rows * FIRA_D * 4.0);
#define FIRA_DENSE_GO(KERNEL, NW, FLAG)                                                                              \
    do {                                                                                                           \
        if (int rc = raise_lds(KERNEL, &FLAG)) return rc;                                                          \
        hipLaunchKernelGGL(KERNEL, dim3(8 * chunk), dim3(NW * 64), lds, s, graph_rows, KP, rowptr, col, val, X, ldx, Y, ldy, \
                           nrb, n_items, chunk);                                                                   \
    } while (0)
    if (bf16 && big) FIRA_DENSE_GO((spmm_dense_bf16_kernel<4, 8>), 8, a0);
    else if (bf16) FIRA_DENSE_GO((spmm_dense_bf16_kernel<2, 4>), 4, a1);
    else if (big) FIRA_DENSE_GO((spmm_dense_f32_kernel<2, 8>), 8, a2);
    else FIRA_DENSE_GO((spmm_dense_f32_kernel<1, 4>), 4, a3);
#undef FIRA_DENSE_GO
    FIRA_CHECK_LAUNCH("csr_spmm_dense");
    return 0;
}

}  // namespace fira
