// The generator projection logits = x Wout^T + bout (reference Model.py:54 / :85, out_fc) on the bf16 matrix cores at fp32
// accuracy (round 6; x3.h): [R, 24650] x K = 256 -- 6.3 GFLOP at batch 32, the one product of the caller's chain that ran at the
// fp32 MFMA roof (gemm_f32_kernel<64,64>: 53 us = 0.75 of the fp32 peak).
//
// Orientation: the WEIGHT rows are the LDS-resident operand.  A workgroup owns a contiguous range of 16-row vocabulary tiles
// (one workgroup per CU, 1 541 tiles over 256 workgroups = 6 or 7 each), reads its rows of Wout ONCE, as whole fp32 rows from HBM
// (25 MB per launch in all, no pre-split pass over the 6.3 M weights), splits them into three bf16 planes in LDS and multiplies
// them with the decoder's output rows, which arrive as pre-split planes in fragment order (rows_split_planes: R x 256 values,
// L2-resident).  The product is out^T: tile rows = vocabulary entries, tile columns = target rows; a lane of the accumulator
// holds four consecutive vocabulary entries of one target row = one 16-byte store into logits[row][n .. n + 3].
// Per pass of <= 4 tiles and chunk of 32 target-row blocks: wave w (of 8: 256 registers per lane) owns the row blocks w, w + 8,
// w + 16, w + 24 (16 accumulators), streams their planes one k step ahead (two register sets) and reads the weight fragments
// from LDS once per four blocks.  (The first form -- 16 waves, two blocks each -- spilled 43 registers at 128 per lane and ran a
// whole extra chunk for the two blocks beyond 32 at 530 rows: slower than the fp32 kernel.)
#include "engine.h"
#include "mfma_frag.h"
#include "epilogue.h"
#include "x3.h"

namespace fira {

constexpr int HX_TILE = 16, HX_TMAX = 4, HX_ROWS = HX_TILE * HX_TMAX, HX_WAVES = 8, HX_NB = 4, HX_GRID = 256;
constexpr size_t HX_PLANE = (size_t)HX_ROWS * 512;
constexpr size_t HX_LDS = 3 * HX_PLANE;                            // 96 KB: one workgroup per CU

typedef float hx_acc __attribute__((ext_vector_type(4)));

// planes of R fp32 rows [R][256] (row stride ld) as the STREAMED operand of an X3 product: unit ((row / 16) * 8 + k / 32) * 64 +
// ((k / 8) % 4) * 16 + row % 16 holds eight consecutive k of a row; plane pl at dst + pl * Rpad * 256 (Rpad = R rounded up to 16;
// rows past R are zero).  A thread owns one unit.
__global__ __launch_bounds__(256) void rows_split_planes_kernel(int R, int Rpad, const float* __restrict__ src, int ld,
                                                                uint16_t* __restrict__ dst) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int m = j >> 5, c = j & 31;
    if (m >= Rpad) return;
    f32x4v u0 = {0.f, 0.f, 0.f, 0.f}, u1 = {0.f, 0.f, 0.f, 0.f};
    if (m < R) {
        const float* p = src + (size_t)m * ld + c * 8;
        u0 = *reinterpret_cast<const f32x4v*>(p);
        u1 = *reinterpret_cast<const f32x4v*>(p + 4);
    }
    const int unit = ((m >> 4) * 8 + (c >> 2)) * 64 + (c & 3) * 16 + (m & 15);
    uint16_t* d = dst + (size_t)unit * 8;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const uint32_t p0 = gx_pack(u0.x, u0.y), p1 = gx_pack(u0.z, u0.w), p2 = gx_pack(u1.x, u1.y), p3 = gx_pack(u1.z, u1.w);
        *reinterpret_cast<uint4*>(d + (size_t)pl * Rpad * FIRA_D) = uint4{p0, p1, p2, p3};
        if (pl < 2) {
            u0.x = gx_rest_lo(u0.x, p0); u0.y = gx_rest_hi(u0.y, p0); u0.z = gx_rest_lo(u0.z, p1); u0.w = gx_rest_hi(u0.w, p1);
            u1.x = gx_rest_lo(u1.x, p2); u1.y = gx_rest_hi(u1.y, p2); u1.z = gx_rest_lo(u1.z, p3); u1.w = gx_rest_hi(u1.w, p3);
        }
    }
}

// one chunk of a pass for a wave with CNT (1..4) row blocks: compile-time block count -- the k loop is one basic block, the
// scheduler can run the fragment reads of a tile under the MFMAs of the previous one; the six terms go block by block, so that
// consecutive MFMAs never depend on each other
template <int CNT>
__device__ __forceinline__ void hx_chunk(const char* __restrict__ hx_lds, int a_q, const rsrc_t rX, unsigned plane_x, int mb_first,
                                         int lane, int l15, int kq, int nt, int n0, int R, int V, const float* __restrict__ bias,
                                         float* __restrict__ out, int ldo) {
    unsigned xo[CNT];
#pragma unroll
    for (int j = 0; j < CNT; ++j) xo[j] = (unsigned)((mb_first + HX_WAVES * j) * 8 * 64 + lane) * 16u;
    hx_acc acc[HX_TMAX][CNT];
#pragma unroll
    for (int tt = 0; tt < HX_TMAX; ++tt)
#pragma unroll
        for (int j = 0; j < CNT; ++j) acc[tt][j] = hx_acc{0.f, 0.f, 0.f, 0.f};
    uint4 b[2][CNT][3];
#pragma unroll
    for (int j = 0; j < CNT; ++j)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            b[0][j][pl] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rX, xo[j] + pl * plane_x, 0, 0));
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        if (ks + 1 < 8) {
#pragma unroll
            for (int j = 0; j < CNT; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    b[(ks + 1) & 1][j][pl] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                           rX, xo[j] + pl * plane_x, (ks + 1) * 1024, 0));
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int tt = 0; tt < HX_TMAX; ++tt) {
            if (tt < nt) {                               // block-uniform
                const char* pa = hx_lds + ((a_q ^ (ks << 6)) + tt * (HX_TILE * 512));
                const abf16x8 ah = *reinterpret_cast<const abf16x8*>(pa);
                const abf16x8 am = *reinterpret_cast<const abf16x8*>(pa + HX_PLANE);
                const abf16x8 al = *reinterpret_cast<const abf16x8*>(pa + 2 * HX_PLANE);
#define HX_TERM(A, P)                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < CNT; ++j)                                                                    \
        acc[tt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, __builtin_bit_cast(abf16x8, b[ks & 1][j][P]), acc[tt][j], 0, 0, 0);
                HX_TERM(al, 0) HX_TERM(ah, 2) HX_TERM(am, 1) HX_TERM(am, 0) HX_TERM(ah, 1) HX_TERM(ah, 0)
#undef HX_TERM
            }
        }
        asm volatile("" ::: "memory");
    }
    // accumulator register r of (tile tt, block j): vocabulary entry n0 + 16 tt + 4 kq + r, target row 16 mb + l15
#pragma unroll
    for (int tt = 0; tt < HX_TMAX; ++tt) {
        if (tt < nt) {
            const int n = n0 + tt * HX_TILE + 4 * kq;
            f32x4v bv = {0.f, 0.f, 0.f, 0.f};
            if (n + 3 < V) bv = *reinterpret_cast<const f32x4v*>(bias + n);
            else {
                if (n < V) bv.x = bias[n];
                if (n + 1 < V) bv.y = bias[n + 1];
                if (n + 2 < V) bv.z = bias[n + 2];
            }
#pragma unroll
            for (int j = 0; j < CNT; ++j) {
                const int m = (mb_first + HX_WAVES * j) * 16 + l15;
                if (m >= R) continue;
                const f32x4v v = {acc[tt][j][0] + bv.x, acc[tt][j][1] + bv.y, acc[tt][j][2] + bv.z, acc[tt][j][3] + bv.w};
                float* po = out + (size_t)m * ldo + n;
                if (n + 3 < V) *reinterpret_cast<f32x4v*>(po) = v;
                else {
                    if (n < V) po[0] = v.x;
                    if (n + 1 < V) po[1] = v.y;
                    if (n + 2 < V) po[2] = v.z;
                }
            }
        }
    }
}

__global__ __launch_bounds__(HX_WAVES * 64) void head_logits_x3_kernel(int R, int Rpad, int V, const float* __restrict__ W,
                                                                       const float* __restrict__ bias,
                                                                       const uint16_t* __restrict__ xp, float* __restrict__ out,
                                                                       int ldo) {
    extern __shared__ __attribute__((aligned(16))) char hx_lds[];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int n_tiles = (V + HX_TILE - 1) / HX_TILE;
    const int tq = n_tiles / HX_GRID, tr = n_tiles % HX_GRID;
    const int wg = blockIdx.x;
    const int t_beg = wg * tq + min(wg, tr), t_cnt = tq + (wg < tr ? 1 : 0);
    const int a_q = gx_frag_base(l15, kq);
    const int n_mb = Rpad >> 4;                                  // 16-row blocks of target rows
    const unsigned plane_x = (unsigned)Rpad * FIRA_D * 2u;       // bytes of one plane of x
    const rsrc_t rX = buf_rsrc(xp, 3u * plane_x);

    for (int pass = 0; pass < t_cnt; pass += HX_TMAX) {
        const int nt = min(HX_TMAX, t_cnt - pass);
        const int n0 = (t_beg + pass) * HX_TILE;
        // ------------------------------------------------------------ 1. the pass's weight rows -> three planes (8 rows per wave)
        {
            f32x4v w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int n = n0 + wave * 8 + i;
                w[i] = (wave * 8 + i < nt * HX_TILE && n < V) ? *reinterpret_cast<const f32x4v*>(W + (size_t)n * FIRA_D + lane * 4)
                                                              : f32x4v{0.f, 0.f, 0.f, 0.f};
            }
            if (wave * 8 < nt * HX_TILE) {
#pragma unroll
                for (int i = 0; i < 8; ++i) gx_store_row4(hx_lds, HX_PLANE, wave * 8 + i, lane, w[i]);
            }
        }
        lds_barrier();   
        // ------------------------------------------------------------ 2. chunks of 32 target-row blocks: up to four per wave
        for (int mb0 = 0; mb0 < n_mb; mb0 += HX_NB * HX_WAVES) {
            // this wave's blocks mb0 + wave + 8 j, j < cnt (wave-uniform)
            const int cnt = min(HX_NB, (n_mb - mb0 - wave + HX_WAVES - 1) / HX_WAVES);
            const int mbf = mb0 + wave;
            switch (cnt) {
                case 4: hx_chunk<4>(hx_lds, a_q, rX, plane_x, mbf, lane, l15, kq, nt, n0, R, V, bias, out, ldo); break;
                case 3: hx_chunk<3>(hx_lds, a_q, rX, plane_x, mbf, lane, l15, kq, nt, n0, R, V, bias, out, ldo); break;
                case 2: hx_chunk<2>(hx_lds, a_q, rX, plane_x, mbf, lane, l15, kq, nt, n0, R, V, bias, out, ldo); break;
                case 1: hx_chunk<1>(hx_lds, a_q, rX, plane_x, mbf, lane, l15, kq, nt, n0, R, V, bias, out, ldo); break;
                default: break;
            }
        }
        if (pass + HX_TMAX < t_cnt) lds_barrier();         // the next pass overwrites the planes
    }
}

size_t head_logits_x3_scratch_elems(int R) { return (size_t)((R + 15) / 16 * 16) * FIRA_D * 3; }

// logits[R, V] = x[R, 256] Wout[V, 256]^T + bias;  xplanes: head_logits_x3_scratch_elems(R) bf16 of scratch
int head_logits_x3(hipStream_t s, int R, int V, const float* x, int ldx, const float* W, const float* bias, float* out, int ldo,
                   uint16_t* xplanes) {
    if (R <= 0 || V <= 0) return 0;
    FIRA_REQUIRE(x && W && bias && out && xplanes, "head_logits_x3: null pointer argument");
    FIRA_REQUIRE((uintptr_t)x % 16 == 0 && ldx % 4 == 0 && (uintptr_t)W % 16 == 0 && (uintptr_t)out % 16 == 0 && ldo % 4 == 0 &&
                 (uintptr_t)bias % 16 == 0 && (uintptr_t)xplanes % 16 == 0, "head_logits_x3: 16-byte aligned rows");
    const int Rpad = (R + 15) / 16 * 16;
    FIRA_REQUIRE((size_t)Rpad * FIRA_D * 6 < (1ull << 31), "head_logits_x3: %d rows exceed the 2 GiB the kernel addresses", R);
    ProfScope prof(s, PROF_GEMM, 2.0 * R * (double)V * FIRA_D, 4.0 * ((double)R * FIRA_D + (double)V * FIRA_D + (double)R * V));
    hipLaunchKernelGGL(rows_split_planes_kernel, dim3((Rpad * 32 + 255) / 256), dim3(256), 0, s, R, Rpad, x, ldx, xplanes);
    static const hipError_t attr = hipFuncSetAttribute((const void*)head_logits_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)HX_LDS);
    if (attr != hipSuccess) return set_err("head_logits_x3: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr));
    hipLaunchKernelGGL(head_logits_x3_kernel, dim3(HX_GRID), dim3(HX_WAVES * 64), HX_LDS, s, R, Rpad, V, W, bias, xplanes, out, ldo);
    FIRA_CHECK_LAUNCH("head_logits_x3");
    return 0;
}

}  // namespace fira

extern "C" {
size_t fira_head_logits_x3_scratch_bytes(int R) { return fira::head_logits_x3_scratch_elems(R) * sizeof(uint16_t); }
int fira_head_logits_x3(void* stream, int R, int V, const float* x, int ldx, const float* W, const float* bias, float* logits,
                        int ldl, void* scratch) {
    return fira::head_logits_x3((hipStream_t)stream, R, V, x, ldx, W, bias, logits, ldl, reinterpret_cast<uint16_t*>(scratch));
}
}
