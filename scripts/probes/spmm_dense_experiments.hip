// Block-dense aggregation  Z = A_hat · H  on the matrix cores, for graph batches that are dense enough that a CSR gather
// loses to the plain product the reference actually runs (gnn_transformer.py:80: torch.bmm(edge.float(), x) over a
// dense [B,N,N] adjacency).  BASELINE config 5 (128 graphs x 512 nodes, ~22 % dense) is that case: 7.6 M gathered rows of
// 1 KiB per launch keep the CSR kernels on the L2 / LDS gather rate (spmm.hip: 180-300 us), while the same product is
// 17 GFLOP -- 7 us of bf16 MFMA time, 110 us of fp32 MFMA time -- over 195 MB of compulsory traffic (24 us of HBM).
//
// The input stays the engine's block-diagonal CSR (no dense adjacency in HBM): a workgroup owns R consecutive rows of
// ONE graph and all 256 feature columns, and
//   1. densifies its R x N slice of A_hat into LDS (zero fill, then one LDS store per CSR entry; the slice's entries are
//      one contiguous range of col/val, read with coalesced loads; k-contiguous rows = MFMA A-operand layout);
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
// A wave owns R / (NT / 64) consecutive rows of the slice, i.e. ONE contiguous range of col / val: its lanes read the range 64
// entries per load, DN_IT loads of each array in flight per pass (a row-by-row walk is a chain of three dependent global
// round trips per row; the round-5 form -- threads striding over the whole slice, eight entries in flight, the row found by
// walking the LDS copy of the row offsets, the predecessor's column by a dependent load -- cost 35 us of the launch's 82 on
// config 5).  The row of an entry = the number of the wave's row boundaries at or below it (the boundaries are wave-uniform);
// neighbours' columns (for the runs of equal columns) come from the adjacent lane.
__device__ unsigned long long g_dbg[1024 * 8];
#define FIRA_STAMP(i) do { if (threadIdx.x == 0) g_dbg[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
constexpr int DN_IT = 16;
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
    FIRA_STAMP(4);
    const int lane = threadIdx.x & 63, w0 = (threadIdx.x >> 6) * RW;
    if (w0 < nr) {                                                   // (wave-uniform)
        int b[RW + 1];                                               // the wave's row boundaries (uniform values)
#pragma unroll
        for (int k = 0; k <= RW; ++k) b[k] = sm_rp[min(w0 + k, nr)];
        const int beg = b[0], end = b[RW];
        for (int base = beg; base < end; base += 64 * DN_IT) {
            int c[DN_IT];
            float v[DN_IT];
#pragma unroll
            for (int u = 0; u < DN_IT; ++u) {                        // everything requested before anything is used
                const int e = base + u * 64 + lane;
                c[u] = e < end ? col[e] : -1;
                v[u] = e < end ? val[e] : 0.f;
            }
            // the columns just outside the pass, for its first / last lane
            const int c_before = (lane == 0 && base > beg) ? col[base - 1] : -1;
            const int c_after = (lane == 63 && base + 64 * DN_IT < end) ? col[base + 64 * DN_IT] : -1;
            if (base == beg && v[DN_IT - 1] != 12345.f) FIRA_STAMP(5);
            if (base != beg && v[DN_IT - 1] != 12345.f) FIRA_STAMP(6);
#pragma unroll
            for (int u = 0; u < DN_IT; ++u) {
                if (base + u * 64 >= end) break;                     // (wave-uniform)
                const int e = base + u * 64 + lane;
                int lo = 0;
#pragma unroll
                for (int k = 1; k < RW; ++k) lo += b[k] <= e ? 1 : 0;
                const int rb = sm_rp[min(w0 + lo, nr)], re = sm_rp[min(w0 + lo + 1, nr)];
                int cn = __shfl_down(c[u], 1, 64), cp = __shfl_up(c[u], 1, 64);
                if (lane == 63) cn = u + 1 < DN_IT ? __builtin_amdgcn_readlane(c[u + 1 < DN_IT ? u + 1 : u], 0) : c_after;
                if (lane == 0) cp = u > 0 ? __builtin_amdgcn_readlane(c[u > 0 ? u - 1 : 0], 63) : c_before;
                const int cl = c[u] - row0;
                // the LAST entry of a run of equal columns (sorted rows) stores the run's sum
                if (e >= end || (e + 1 < re && cn == c[u]) || (unsigned)cl >= (unsigned)graph_rows) continue;
                float sum = v[u];
                if (e > rb && cp == c[u])                            // rare: the run's earlier entries, walked back
                    for (int k = e - 1; k >= rb && col[k] == c[u]; --k) sum += val[k];
                store_elem(tile + (size_t)(w0 + lo) * pitch + (size_t)cl * sizeof(T), sum, T());
            }
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
                                                                  int chunk, int probe) {
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
    if (!(probe & 1)) densify_rows<uint16_t, R, NW * 64>(tile, sm_rp, pitch, graph_rows, row0, r0, rowptr, col, val);

    f32x16 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const char* a_base = tile + (size_t)l31 * pitch + kg * 16;
    const int nkc = (probe & 2) ? 1 : KP / DB_KC;
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
    if ((probe & 4) && acc[0][0][0] != 12345.f) return;
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
// bf16, round 6: the kernel above is bound by the NUMBER of its memory instructions -- a lane-per-column B fragment is eight
// dword loads, every wave-load moves 256 bytes at the address unit's 16 cycles per wave-instruction (16 B / clk / CU measured:
// 13.6 us per 128-row slice for the 512 KB of H), and the fp32 tile leaves through as many dword stores.  Here H goes through
// LDS: a wave fetches one (16-row chunk, 8-row half) unit with eight 16-byte loads (lane l = columns 4l .. 4l+3, all 256
// columns: 1 KiB per wave-load), rounds it, and writes the four B fragments a lane then holds (columns 4l + t, eight k each)
// as 16-byte LDS stores into a ring of three 8 KiB chunk stages; column tile T = (t, half) of the product is the columns
// 4 (32 half + n) + t, so a fragment read is 64 consecutive 16-byte slots.  Wave w loads the units of chunks w/2, w/2 + 4, ...
// four steps ahead of their use; one barrier per 16-wide k step.  The finished tile is staged in LDS (over the adjacency
// slice) and leaves as whole 1 KiB rows, 16 bytes per lane.
constexpr int D2_NS = 3, D2_STAGE = 8192, D2_SP = 1024 + 16;
template <int RT>
__global__ __launch_bounds__(512) void spmm_dense_bf16_lds_kernel(int graph_rows, int KP, const int32_t* __restrict__ rowptr,
                                                                  const int32_t* __restrict__ col,
                                                                  const float* __restrict__ val,
                                                                  const float* __restrict__ X, int ldx,
                                                                  float* __restrict__ Y, int ldy, int nrb, int n_items,
                                                                  int chunk, int tile_bytes, int probe) {
    constexpr int R = 32 * RT, NW = 8;
    extern __shared__ __attribute__((aligned(16))) char tile[];
    __shared__ int sm_rp[R + 1];
    const int item = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (item >= n_items) return;
    const int g = item / nrb, rb = item - g * nrb;
    const int row0 = g * graph_rows, r0 = rb * R;
    const int pitch = KP * 2 + 16;
    char* xs = tile + tile_bytes;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kg = lane >> 5;
    const rsrc_t rX = buf_rsrc(X + (size_t)row0 * ldx, (unsigned)graph_rows * (unsigned)ldx * 4u);   // rows >= graph_rows read 0
    const unsigned ldxb = (unsigned)ldx * 4u;
    const int nk = (probe & 2) ? 4 : KP / 16;
    // (probe) the row blocks of a graph start their walk over k a quarter apart
    const int rot = (probe & 16) ? ((rb * nk / nrb) & ~3) : 0;
    // Every wave moves rows 2 wave, 2 wave + 1 of EVERY chunk (lane l = columns 4l .. 4l+3): one dword -- the pair of k values
    // -- of the four fragments' slots; eight chunks in flight per wave in a static ring of registers (the loop below runs
    // eight steps per trip), so every step every wave does the same small piece of loader work between barrier and MFMAs.
    // (A wave per unit of 8 or 16 rows made the loading wave the critical path of its step: 1 100 cycles per step.)
    f32x4 raw[8][2];
    const unsigned xoff = (unsigned)(2 * wave) * ldxb + (unsigned)lane * 16u;
#define FIRA_D2_FETCH(u, q)                                                                                        \
    do {                                                                                                           \
        const unsigned ro = (unsigned)(((u) + rot) % nk) * 16u * ldxb + xoff;                                      \
        raw[q][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rX, ro, 0, 0));                \
        raw[q][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rX, ro + ldxb, 0, 0));         \
    } while (0)
#define FIRA_D2_PUT(u, q)                                                                                          \
    do {                                                                                                           \
        char* st = xs + ((u) % D2_NS) * D2_STAGE + ((wave >> 2) * 4 * 64 + lane) * 16 + (wave & 3) * 4;            \
        _Pragma("unroll") for (int tt = 0; tt < 4; ++tt)                                                           \
            *reinterpret_cast<uint32_t*>(st + tt * 1024) = pack_bf16_2(raw[q][0][tt], raw[q][1][tt]);              \
        if ((u) + 8 < nk) FIRA_D2_FETCH((u) + 8, q);                                                               \
    } while (0)
    FIRA_STAMP(0);
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (q < nk) FIRA_D2_FETCH(q, q);                             // (do not depend on the adjacency slice)
    if (!(probe & 1)) densify_rows<uint16_t, R, NW * 64>(tile, sm_rp, pitch, graph_rows, row0, r0, rowptr, col, val);

    FIRA_STAMP(1);
    f32x16 acc[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int t = wave & 3, half = wave >> 2;                        // this wave's column tile: columns 4 (32 half + n) + t
    const char* a_base = tile + (size_t)l31 * pitch + kg * 16;
    const char* b_base = xs + ((kg * 4 + t) * 64 + half * 32 + l31) * 16;
    for (int s8 = -2; s8 < nk; s8 += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int s = s8 + k;
            if (s >= nk) break;                                      // (uniform)
            // (LDS-only barrier: __syncthreads() also waits for the units in flight from global memory)
            if (s >= 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // stage (s + 2) % 3 was read in step s - 1
            if (s + 2 < nk && !(probe & 32)) FIRA_D2_PUT(s + 2, k);                   // this wave's two rows of the chunk of step s + 2
            if (s < 0 || (probe & 64)) continue;
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(b_base + (s % D2_NS) * D2_STAGE);
            const int ka = (s + rot) % nk;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(a_base + (size_t)rt * 32 * pitch + ka * 32);
                acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[rt], 0, 0, 0);
            }
        }
    }
#undef FIRA_D2_PUT
#undef FIRA_D2_FETCH
    __syncthreads();                                                 // the slice and the stages are free: the tile is staged
    FIRA_STAMP(2);
    if ((probe & 4) && acc[0][0] != 12345.f) return;
    if (probe & 8) {
        const rsrc_t rY = buf_rsrc(Y + (size_t)row0 * ldy, (unsigned)graph_rows * (unsigned)ldy * 4u);
        for (int rt = 0; rt < RT; ++rt) for (int r = 0; r < 16; ++r) { const float v = acc[rt][r];
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rY, (unsigned)(r0 + rt * 32 + acc_row32(r, kg)) * (unsigned)ldy * 4u + (unsigned)(4 * (half * 32 + l31) + t) * 4u, 0, 0); }
        return;
    }
    // [row][t][j]: the value of column 4 j + t at (row, t, j), so that writes (consecutive j) and reads are conflict-free
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            *reinterpret_cast<float*>(tile + (size_t)(rt * 32 + acc_row32(r, kg)) * D2_SP + (t * 64 + half * 32 + l31) * 4) = acc[rt][r];
    __syncthreads();
    const int nr = min(R, graph_rows - r0);
    for (int row = wave; row < nr; row += NW) {
        const char* src = tile + (size_t)row * D2_SP + lane * 4;
        f32x4 y;
        y.x = *reinterpret_cast<const float*>(src); y.y = *reinterpret_cast<const float*>(src + 256);
        y.z = *reinterpret_cast<const float*>(src + 512); y.w = *reinterpret_cast<const float*>(src + 768);
        *reinterpret_cast<f32x4*>(Y + (size_t)(row0 + r0 + row) * ldy + lane * 4) = y;
    }
    FIRA_STAMP(3);
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16, streaming form (round 6).  The row-block kernels above read a graph's H once per row block -- four times per graph at
// 128 rows per block, and the four blocks run side by side in lock-step, so all of it comes through the fabric: 268 MB per
// launch on config 5, measured at the 8.4 TB/s the L2s deliver to 256 CUs (16 us per round whatever the loader's shape), in
// series with 16 us of densifying per block.  Here H is read ONCE: a workgroup owns (graph, 128-column half), each of its
// eight waves keeps its 16 columns x 512 k of H as bf16 B fragments in 64 registers for the whole launch, and the adjacency
// STREAMS past: tiles of 16 rows are densified into one of two 16.6 KB LDS buffers (a wave owns rows 2w, 2w + 1 of every
// tile: it zeroes them and stores its entries, three 64-entry slots per row requested two tiles ahead), one barrier, then every
// wave multiplies the tile by its fragments (v_mfma_f32_16x16x32_bf16, one A fragment read per 32 k) and stores its 16 x 16
// piece.  The two halves of a graph densify the same rows (the second reads them from L2); bytes through the fabric:
// 67 MB of H + 2 x 60 MB of (col, val) instead of 268 + 60.
constexpr int DS_SLOTS = 3;                                          // 64-entry slots of a row in flight; longer rows: a tail loop
template <int NKS>                                                   // k steps of 32 held in registers: KP == 32 NKS (compile-time: no k-step guards)
__global__ __launch_bounds__(512) void spmm_dense_bf16_stream_kernel(int graph_rows, const int32_t* __restrict__ rowptr,
                                                                     const int32_t* __restrict__ col,
                                                                     const float* __restrict__ val,
                                                                     const float* __restrict__ X, int ldx,
                                                                     float* __restrict__ Y, int ldy, int n_items, int chunk, int probe) {
    extern __shared__ __attribute__((aligned(16))) char lds[];       // int rp[graph_rows + 1 (+pad)] | tile buffers 2 x 16 x pitch
    const int item = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);   // XCD-aware: both halves of a graph on one XCD
    if (item >= n_items) return;
    const int g = item >> 1, h = item & 1;
    const int row0 = g * graph_rows;
    constexpr int KP = 32 * NKS, pitch = KP * 2 + 16;
    const int rp_bytes = ((graph_rows + 1) * 4 + 15) & ~15;
    int* sm_rp = reinterpret_cast<int*>(lds);
    char* tiles = lds + rp_bytes;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n16 = lane & 15, kq = lane >> 4;
    const int cbase = h * 128 + wave * 16;
    const rsrc_t rX = buf_rsrc(X + (size_t)row0 * ldx, (unsigned)graph_rows * (unsigned)ldx * 4u);   // rows >= graph_rows read 0
    const unsigned ldxb = (unsigned)ldx * 4u;
    FIRA_STAMP(0);
    for (int i = threadIdx.x; i <= graph_rows; i += 512) sm_rp[i] = rowptr[row0 + i];
    // this wave's 16 columns of H as B fragments: lane (n, kq) holds k = 32 ks + 8 kq .. + 7 of column cbase + n
    bf16x8 bfrag[NKS];
    const unsigned xo = (unsigned)(cbase + n16) * 4u + (unsigned)(kq * 8) * ldxb;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float raw[NKS / 2][8];
#pragma unroll
        for (int j = 0; j < NKS / 2; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                raw[j][i] = buf_load_f32(rX, xo + (unsigned)((half * (NKS / 2) + j) * 32 + i) * ldxb);   // (k >= graph_rows: 0)
#pragma unroll
        for (int j = 0; j < NKS / 2; ++j) {
            uint4 u;
            u.x = pack_bf16_2(raw[j][0], raw[j][1]); u.y = pack_bf16_2(raw[j][2], raw[j][3]);
            u.z = pack_bf16_2(raw[j][4], raw[j][5]); u.w = pack_bf16_2(raw[j][6], raw[j][7]);
            bfrag[half * (NKS / 2) + j] = __builtin_bit_cast(bf16x8, u);
        }
    }
    __syncthreads();                                                 // the row offsets
    FIRA_STAMP(1);
    long long t_sc = 0, t_fe = 0, t_ba = 0, t_mm = 0, t_wait = 0; (void)t_wait;

    const int n_tiles = (graph_rows + 15) >> 4;
    // (col, val) of this wave's two rows of a tile, DS_SLOTS x 64 entries each, in one of two register sets
    int pc[2][2][DS_SLOTS];
    float pv[2][2][DS_SLOTS];
    // (every load is issued by every lane -- lanes without an entry carry the descriptor's out-of-range offset -- and none sits
    // behind a branch: with loads the compiler cannot count, its wait before the first use of a set is vmcnt(0), i.e. also for
    // the tile stores issued a moment ago: 2 000 cycles per tile measured)
    const rsrc_t rC = buf_rsrc(col, 0x7fffffffu), rV = buf_rsrc(val, 0x7fffffffu);
#define FIRA_DS_FETCH(ti, set)                                                                                     \
    _Pragma("unroll") for (int rr = 0; rr < 2; ++rr) {                                                             \
        const int r = min((ti) * 16 + wave * 2 + rr, graph_rows);                                                  \
        const int rb = sm_rp[r], re = sm_rp[min(r + 1, graph_rows)];                                               \
        _Pragma("unroll") for (int sl = 0; sl < DS_SLOTS; ++sl) {                                                  \
            const int e = rb + sl * 64 + lane;                                                                     \
            const unsigned off = e < re ? (unsigned)e * 4u : FIRA_OOB;                                             \
            pc[set][rr][sl] = (int)__builtin_amdgcn_raw_buffer_load_b32(rC, off, 0, 0);                            \
            pv[set][rr][sl] = buf_load_f32(rV, off);                                                               \
        }                                                                                                          \
    }
    // zero this wave's two rows of the buffer, then store the entries.  Fast path of a 64-entry slot (no entry has an equal
    // successor: one DPP + compare + ballot): every entry stores its value.  Otherwise the LAST entry of a run of equal
    // columns (sorted rows) stores the run's sum, the earlier ones walked back in global memory.
#define FIRA_DS_SCATTER(ti, set)                                                                                   \
    do {                                                                                                           \
        char* buf = tiles + ((ti) & 1) * 16 * pitch + (wave * 2) * pitch;                                          \
        _Pragma("unroll") for (int o = 0; o < 2 * pitch; o += 1024)                                                \
            if (o + lane * 16 < 2 * pitch) *reinterpret_cast<uint4*>(buf + o + lane * 16) = uint4{0u, 0u, 0u, 0u};     \
        _Pragma("unroll") for (int rr = 0; rr < 2; ++rr) {                                                         \
            const int r = (ti) * 16 + wave * 2 + rr;                                                               \
            if (r >= graph_rows) break;                                                                            \
            const int rb = sm_rp[r], re = sm_rp[r + 1];                                                            \
            char* rowp = buf + rr * pitch;                                                                         \
            _Pragma("unroll") for (int sl = 0; sl < DS_SLOTS; ++sl) {                                              \
                if (rb + sl * 64 >= re) break;                                                                     \
                const int c = pc[set][rr][sl] - row0;                                                              \
                const int nx0 = sl + 1 < DS_SLOTS ? __builtin_amdgcn_readlane(pc[set][rr][sl + 1 < DS_SLOTS ? sl + 1 : sl], 0) - row0 : -2; \
                int cn = __builtin_amdgcn_update_dpp(nx0, c, 0x130, 0xf, 0xf, false);      /* wave_shl:1 */          \
                const int e = rb + sl * 64 + lane;                                                                 \
                const bool more = sl + 1 == DS_SLOTS && rb + DS_SLOTS * 64 < re;           /* (uniform) */           \
                if (__ballot(e + 1 < re && cn == c) == 0ull && !more) {                                            \
                    if (e < re && (unsigned)c < (unsigned)graph_rows)                                              \
                        *reinterpret_cast<uint16_t*>(rowp + c * 2) = (uint16_t)(pack_bf16_2(pv[set][rr][sl], 0.f) & 0xffffu); \
                    continue;                                                                                      \
                }                                                                                                  \
                const int pv0 = sl > 0 ? __builtin_amdgcn_readlane(pc[set][rr][sl > 0 ? sl - 1 : 0], 63) - row0 : -2; \
                const int cp = __builtin_amdgcn_update_dpp(pv0, c, 0x138, 0xf, 0xf, false);      /* wave_shr:1 */    \
                if (more && lane == 63) cn = col[e + 1] - row0;                                                    \
                if (e >= re || (e + 1 < re && cn == c) || (unsigned)c >= (unsigned)graph_rows) continue;           \
                float sum = pv[set][rr][sl];                                                                       \
                if (e > rb && cp == c)                                                                             \
                    for (int k = e - 1; k >= rb && col[k] - row0 == c; --k) sum += val[k];                         \
                *reinterpret_cast<uint16_t*>(rowp + c * 2) = (uint16_t)(pack_bf16_2(sum, 0.f) & 0xffffu);          \
            }                                                                                                      \
            for (int base = rb + DS_SLOTS * 64; base < re; base += 64) {     /* rows longer than the slots: rare */  \
                const int e = base + lane;                                                                         \
                const int c = e < re ? col[e] - row0 : -1;                                                         \
                const int cn = e + 1 < re ? col[e + 1] - row0 : -2;                                                \
                if (e >= re || cn == c || (unsigned)c >= (unsigned)graph_rows) continue;                           \
                float sum = val[e];                                                                                \
                for (int k = e - 1; k >= rb && col[k] - row0 == c; --k) sum += val[k];                             \
                *reinterpret_cast<uint16_t*>(rowp + c * 2) = (uint16_t)(pack_bf16_2(sum, 0.f) & 0xffffu);          \
            }                                                                                                      \
        }                                                                                                          \
    } while (0)
    const rsrc_t rY = buf_rsrc(Y + (size_t)row0 * ldy, (unsigned)graph_rows * (unsigned)ldy * 4u);   // rows >= graph_rows: dropped
    const unsigned yo = (unsigned)(cbase + n16) * 4u + (unsigned)(kq * 4) * (unsigned)ldy * 4u;
    float yprev[4] = {0.f, 0.f, 0.f, 0.f};
    int tprev = -1;
    FIRA_DS_FETCH(0, 0);
    FIRA_DS_FETCH(1, 1);
    FIRA_DS_SCATTER(0, 0);
    FIRA_DS_FETCH(2, 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // Step of tile ti, one barrier each: the tile's A fragments are REQUESTED, then this wave densifies its rows of tile ti + 1
    // into the other buffer (vector instructions, LDS stores) while they arrive, requests the entries of tile ti + 3, stores the
    // previous tile's piece, and only then multiplies -- the LDS reads (16 KB per wave and tile: the phase's bound), the
    // densifying and the matrix pipe overlap inside every wave instead of following each other between barriers.
    for (int t2 = 0; t2 < n_tiles; t2 += 2) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ti = t2 + q;
            if (ti >= n_tiles) break;                                // (uniform)
            const char* ap = tiles + (ti & 1) * 16 * pitch + n16 * pitch + kq * 16;
            bf16x8 af[NKS];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                af[ks] = *reinterpret_cast<const bf16x8*>(ap + ks * 64);
            if (ti + 1 < n_tiles && !(probe & 1)) {
                if (q == 0) FIRA_DS_SCATTER(ti + 1, 1);
                else FIRA_DS_SCATTER(ti + 1, 0);
            }
            if (q == 0) { FIRA_DS_FETCH(ti + 3, 1); }                // (past the last tile: no lane has an entry)
            else { FIRA_DS_FETCH(ti + 3, 0); }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yprev[i]), rY,
                                                      tprev < 0 ? FIRA_OOB : yo + (unsigned)(tprev * 16 + i) * (unsigned)ldy * 4u, 0, 0);
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            if (!(probe & 2))
#pragma unroll
            for (int ks = 0; ks < NKS; ks += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], bfrag[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks + 1], bfrag[ks + 1], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) yprev[i] = acc0[i] + acc1[i];
            tprev = ti;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // tile ti + 1 is complete, tile ti's buffer is free
        }
    }
#undef FIRA_DS_FETCH
#undef FIRA_DS_SCATTER
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yprev[i]), rY,
                                              tprev < 0 ? FIRA_OOB : yo + (unsigned)(tprev * 16 + i) * (unsigned)ldy * 4u, 0, 0);
    FIRA_STAMP(2);
    if (threadIdx.x == 0) { g_dbg[blockIdx.x * 8 + 3] = t_sc; g_dbg[blockIdx.x * 8 + 4] = t_fe; g_dbg[blockIdx.x * 8 + 5] = t_ba; g_dbg[blockIdx.x * 8 + 6] = t_mm; g_dbg[blockIdx.x * 8 + 7] = t_wait; }
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
                                                                 int chunk, int probe) {
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
                           nrb, n_items, chunk, probe);                                                                   \
    } while (0)
    static const int probe = [] { const char* e = getenv("FIRA_SPMM_DENSE_PROBE"); return e ? atoi(e) : 0; }();
    static const int stream_on = [] { const char* e = getenv("FIRA_SPMM_DENSE_STREAM"); return e ? atoi(e) : 1; }();
    if (bf16 && stream_on) {
        const int n_it = 2 * (n_rows / graph_rows), ch = cdiv(n_it, 8);
        int nks = 2;                                                 // k padded to a power of two: rows of H beyond the graph read as 0
        while (32 * nks < graph_rows) nks *= 2;
        const size_t ldsb = (((size_t)(graph_rows + 1) * 4 + 15) & ~(size_t)15) + 2 * 16 * (size_t)(32 * nks * 2 + 16);
#define FIRA_STREAM_GO(N)                                                                                          \
        hipLaunchKernelGGL(spmm_dense_bf16_stream_kernel<N>, dim3(8 * ch), dim3(512), ldsb, s, graph_rows, rowptr, col, val, X, ldx, \
                           Y, ldy, n_it, ch, probe)
        static const int probe = [] { const char* e = getenv("FIRA_SPMM_DENSE_PROBE"); return e ? atoi(e) : 0; }();
        if (nks == 2) FIRA_STREAM_GO(2);
        else if (nks == 4) FIRA_STREAM_GO(4);
        else if (nks == 8) FIRA_STREAM_GO(8);
        else FIRA_STREAM_GO(16);
#undef FIRA_STREAM_GO
        FIRA_CHECK_LAUNCH("csr_spmm_dense");
        return 0;
    }
    static const bool lds_off = [] { const char* e = getenv("FIRA_SPMM_DENSE_LDS"); return e && e[0] == '0'; }();
    if (bf16 && big && !lds_off && ldx % 4 == 0 && ldy % 4 == 0 && (uintptr_t)X % 16 == 0 && (uintptr_t)Y % 16 == 0) {
        static bool a4 = false;
        if (!a4) {
            const hipError_t e = hipFuncSetAttribute((const void*)spmm_dense_bf16_lds_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
            FIRA_REQUIRE(e == hipSuccess, "csr_spmm: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
            a4 = true;
        }
        const size_t tile_bytes = (size_t)128 * (KP * 2 + 16);
        const size_t lds2 = tile_bytes + D2_NS * D2_STAGE > (size_t)128 * D2_SP ? tile_bytes + D2_NS * D2_STAGE : (size_t)128 * D2_SP;
        hipLaunchKernelGGL(spmm_dense_bf16_lds_kernel<4>, dim3(8 * chunk), dim3(512), lds2, s, graph_rows, KP, rowptr, col, val, X, ldx,
                           Y, ldy, nrb, n_items, chunk, (int)tile_bytes, probe);
    } else
    if (bf16 && big) FIRA_DENSE_GO((spmm_dense_bf16_kernel<4, 8>), 8, a0);
    else if (bf16) FIRA_DENSE_GO((spmm_dense_bf16_kernel<2, 4>), 4, a1);
    else if (big) FIRA_DENSE_GO((spmm_dense_f32_kernel<2, 8>), 8, a2);
    else FIRA_DENSE_GO((spmm_dense_f32_kernel<1, 4>), 4, a3);
#undef FIRA_DENSE_GO
    FIRA_CHECK_LAUNCH("csr_spmm_dense");
    return 0;
}

}  // namespace fira
extern "C" int fira_dbg_read(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fira::g_dbg), (size_t)n * 8);
}
