// Internal interfaces between the translation units of libfira_hip (not part of the C ABI).
#pragma once
#include "common.h"
#include <string>
#include <vector>

namespace fira {

// ---- op launchers (defined next to their kernels) ---------------------------------------------------
int gemm_f32(hipStream_t s, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
             float* C, int ldc, const float* bias, int flags, int splitk);
// splitk = 0: automatic tile / split selection; colsum (transA only): colsum[m] += sum_k A[k][m] (fused bias gradient)
int gemm_f32_ex(hipStream_t s, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                float* C, int ldc, const float* bias, int flags, int splitk, float* colsum,
                const int32_t* c_rows = nullptr, const float* relu_mask = nullptr);
// c_rows: output row r is written to C row c_rows[r];  relu_mask [M,N] (ld = ldc): output zeroed where mask <= 0
// the same product with both operands rounded to bf16 on their way into LDS (bf16 MFMA, fp32 accumulate, fp32 storage);
// shapes the bf16 kernel does not take (tiny / unaligned) are forwarded to gemm_f32_ex
int gemm_bf16_ex(hipStream_t s, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                 float* C, int ldc, const float* bias, int flags, int splitk, float* colsum,
                 const int32_t* c_rows = nullptr, const float* relu_mask = nullptr);
// bf16 weight shadows (gemm_bf16.hip): table of the 2-D weights the GEMMs read, one conversion launch per step
// offset: of the fp32 tensor in the parameter buffer = of its as-stored bf16 copy in Wb;  offset_t / pitch_t: start and row
// pitch (elements) of the transposed copy in WbT -- the transposed copies are packed on their own (padded rows do not
// fit the tensor's parameter slot: a [V,256] weight with V % 8 != 0 would spill into its neighbour's shadow)
struct ShadowEntry { int64_t offset; int rows, cols, pitch_t; int64_t offset_t; };
constexpr int SHADOW_MAX = 64;
struct ShadowTable {
    int n = 0;
    int64_t wbt_elems = 0;           // elements of WbT the table addresses
    int tile_start[SHADOW_MAX + 1] = {0};
    ShadowEntry e[SHADOW_MAX];
};
int weight_shadows(hipStream_t s, const ShadowTable& tab, const float* P, uint16_t* Wb, uint16_t* WbT);
int gemm_bf16_wb_ex(hipStream_t s, int M, int N, int K, const float* A, int lda, const uint16_t* Bb, int ldb, float* C,
                    int ldc, const float* bias, int flags, int splitk, const int32_t* c_rows = nullptr,
                    const float* relu_mask = nullptr);
// max_split (both grouped queues): 0 = the default split of the reduction (sized for ~40 problems sharing the chip);
// a small group of long reductions (one encoder layer's three weight gradients) asks for more slabs
int gemm_bf16_group_add_wgrad(hipStream_t s, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                              float* C, int ldc, float* colsum, int max_split = 0);
int gemm_bf16_group_flush(hipStream_t s);
bool gemm_bf16_takes(int M, int N, int K);     // false: the product is too small for the bf16 tiles (runs in fp32)
void gemm_bf16_group_reset();
bool gemm_bf16_group_full();     // the next add would launch the queue by itself: the caller forks + flushes first
// weight gradients as 256 x 256 panel products on the bf16 matrix cores (gemm_wgrad_panel.hip): C[M,N] += A^T B with A [K, lda],
// B [K, ldb]; np = 1 (operands rounded to bf16) or 3 (three-term split: fp32-accurate).  A queue like the grouped launches.
bool gemm_wgrad_panel_on();
bool gemm_wgrad_panel_takes(int M, int N, int K, const float* A, int lda, const float* B, int ldb, int ldc);
void gemm_wgrad_panel_reset();
bool gemm_wgrad_panel_full();
int gemm_wgrad_panel_pending();
int gemm_wgrad_panel_add(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc, float* colsum);
int gemm_wgrad_panel_flush(hipStream_t s, int np);
void gemm_wgrad_panel_scratch(float* buf, size_t n_floats);      // scratch for split problems (thread-local; nullptr: no splits)
int gemm_wgrad_panel(hipStream_t s, int np, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                     int ldc, float* colsum);
// grouped weight gradients (one launch for many small dW += dY^T X problems; see gemm_f32.hip)
int gemm_group_add_wgrad(hipStream_t s, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                         int ldc, float* colsum, int max_split = 0);
int gemm_group_flush(hipStream_t s);
void gemm_group_reset();
bool gemm_group_full();
// coalesced 32x32 tile kernel (gemm_small.hip): shape test, launch with an optional residual epilogue (epilogue.h: EpiRes),
// the stream whose gemm_f32 launches pad their LDS request (gemm_f32.hip: side_lds_pad)
void gemm_set_pad_stream(hipStream_t s);
// and the product with a LayerNorm prologue on its A operand (K = 256)
struct EpiRes;
bool gemm_tile32_takes(int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb);
bool gemm_tile32_try(hipStream_t s, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                     float* C, int ldc, const float* bias, int flags, int* rc, const int32_t* c_rows,
                     const float* relu_mask, const int32_t* a_rows, const EpiRes* er);
bool gemm_tile32_ln_try(hipStream_t s, int M, int N, const float* S, int lds, const float* W, const float* bias, float* Y,
                        int ldy, int flags, const float* gamma, const float* beta, float* x_out, float* stats_out, int* rc);
// LayerNorm backward in the prologue of the data-gradient product dX[M,N] = dx_drop . W (W row-major [256,N]); writes ds,
// dx_drop and [gemm_tile32_lnb_blocks(M), 512] partial {dgamma | dbeta} rows (gemm_small.hip: LnB)
int gemm_tile32_lnb_blocks(int M);
bool gemm_tile32_lnb_try(hipStream_t s, int M, int N, const float* dy, const float* W, int ldw, float* dX, int lddx,
                         const float* relu_mask, const float* sum, const float* stats, const float* gamma, float* ds,
                         float* dx_drop, float* part, float dropout, uint64_t seed, uint32_t site, int* rc,
                         uint32_t idx0 = 0);      // idx0: dropout element index of row 0 (the rows are a slice of the site's rows)
bool gemm_small_try(hipStream_t s, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                    float* C, int ldc, const float* bias, int flags, int* rc, const int32_t* c_rows = nullptr,
                    const float* relu_mask = nullptr);
int csr_spmm(hipStream_t s, int n_rows, const int32_t* rowptr, const int32_t* col, const float* val, const float* X,
             int ldx, float* Y, int ldy, int graph_rows, int variant);
int csr_spmm_ex(hipStream_t s, int n_rows, const int32_t* rowptr, const int32_t* col, const float* val, const float* X,
                int ldx, float* Y, int ldy, int graph_rows, int variant, int accum, float* rowsum);
// One GCN layer per launch (gcn_fused.hip): gather A_hat X (or A_hat dY) into LDS, multiply by the folded weight, finish
// whole rows.  Wk = the weight K-MAJOR ([256 k][256 n], out = U Wk).  forward (Wk = W21^T): sum = dropout(U Wk + bias +
// (A_hat 1) r1_col^T) + X, y = LN(sum) (+ second compact copy at y2[slot2[r]]), rowsum_out (optional) = A_hat 1.
// backward (Wk = W21): u_out = A_hat dY, acc_out += u_out Wk.
int gcn_fused_fwd(hipStream_t s, int n_rows, const int32_t* rowptr, const int32_t* col, const float* val, const float* X,
                  const float* Wk, const float* bias, const float* r1_col, const float* gamma, const float* beta, float* sum,
                  float* y, float* stats, float* rowsum_out, const int32_t* slot2, float* y2, float dropout, uint64_t seed,
                  uint32_t site, int bf16, const uint16_t* Wx = nullptr);
int gcn_fused_bwd(hipStream_t s, int n_rows, const int32_t* rowptr, const int32_t* col, const float* val, const float* dY,
                  const float* Wk, float* u_out, float* acc_out, int bf16, const uint16_t* Wx = nullptr);
// (round 6) Wx: the weight as three bf16 planes in fragment order (gcn_split_planes of the [256 n][256 k] matrix B with
// out = U B^T: W21 for the forward product, W21^T for the backward one) -- fp32 mode then runs the product as six bf16 MFMA
// terms per k step at fp32 accuracy (gcn_fused.hip: X3)
int gcn_split_planes(hipStream_t s, int n, const float* const* src, uint16_t* const* dst);
// (round 6, head_x3.hip) logits[R, V] = x[R, 256] W[V, 256]^T + bias on the bf16 matrix cores at fp32 accuracy (three bf16 terms
// per operand); xplanes: head_logits_x3_scratch_elems(R) bf16 of scratch (the planes of x)
size_t head_logits_x3_scratch_elems(int R);
int head_logits_x3(hipStream_t s, int R, int V, const float* x, int ldx, const float* W, const float* bias, float* out, int ldo,
                   uint16_t* xplanes);
int transpose256(hipStream_t s, int nl, const float* W, float* Wt);       // Wt[l] = W[l]^T, nl stacked [256,256] matrices
// One Combination block per launch (comb_fused.hip): q|k projections, the two-way gate, the output projection, dropout,
// residual and LayerNorm of gnn_transformer.py:176-205 on the code rows.  WqT / WkT / WoT: the three weights K-MAJOR
// ([256 in][256 out], i.e. transposed nn.Linear weights: transpose256_table).  Stores q|k [n,512] and c [n,256] (what the
// backward pass reads), the pre-norm rows `sum` [n,256], (mean, rstd) `stats` [n,2], and the output rows at y[y_rows[r]].
// out [M, ldo] = X [M, 256] W^T + bias for N = nb * 256 columns; Wx: the planes of W's nb [256, 256] row blocks (gcn_split_planes);
// three bf16 terms per operand (fp32-accurate) or, one_plane, the bf16 mode's single rounding (comb_fused.hip: linear_x3_kernel)
int linear_x3(hipStream_t s, int M, const float* X, int ldx, const uint16_t* Wx, int nb, const float* bias, float* out, int ldo,
              bool one_plane);
// out [M, 256] (+)= A [M, nkb * 256] B; Wx: the planes of B's nkb TRANSPOSED [256, 256] row blocks (transpose256_table + gcn_split_planes)
// the generator projection's data gradient: out [M, ldo] += A [M, K] W (float atomics over chunks of K), W [K, 256] as the planes
// split_planes_t writes (ceil(K / 256) transposed row blocks, 3 * 65536 bf16 each)
int split_planes_t(hipStream_t s, const float* W, int V, uint16_t* planes, bool one_plane);
int dgrad_x3_splitk(hipStream_t s, int M, int K, const float* A, int lda, const uint16_t* Wx, float* out, int ldo, bool one_plane);
int linear_x3_kacc(hipStream_t s, int M, const float* A, int lda, const uint16_t* Wx, int nkb, float* out, int ldo, bool accum,
                   bool one_plane);
int comb_fused_fwd(hipStream_t s, int n_rows, const float* Xc, const float* WqT, const float* WkT, const float* WoT,
                   const float* bqk, const float* bo, const float* vtab, int ldv, const int32_t* mark, float* qk, float* c,
                   const float* gamma, const float* beta, float* sum, float* y, const int32_t* y_rows, float* stats,
                   float dropout, uint64_t seed, uint32_t site_gate, uint32_t site_out, int bf16, const uint16_t* Wx = nullptr);
// (round 6) Wx / WTx: gcn_split_planes of Wq | Wk | Wo as stored (forward) / of their transposes (backward), three matrices of
// 3 x 65536 bf16 each -- fp32 mode then runs the three products as six bf16 MFMA terms per k step (comb_fused.hip: X3)
// Backward of the block in one launch: LayerNorm backward of the rows dG[rows[r]], data gradient through Wo, gate backward,
// data gradient through Wq | Wk; dG[rows[r]] = ds + dX.  Leaves dYc [n,256] and dq|dk [n,512] for the weight gradients and
// comb_fused_bwd_parts() partial rows: part_ln [.][512] = {dgamma | dbeta}, part_v [.][1024] = dvtab [4][256].
int comb_fused_bwd_parts();
int comb_fused_bwd(hipStream_t s, int n_rows, float* dG, const int32_t* rows, const float* sum, const float* stats,
                   const float* gamma, const float* Wo, const float* Wqk, const float* qk, const float* vtab, int ldv,
                   const int32_t* mark, float* dYc, float* dqk, float* part_ln, float* part_v, float dropout, uint64_t seed,
                   uint32_t site_gate, uint32_t site_out, int bf16, const uint16_t* WTx = nullptr);
struct TransposeTable { int n = 0; const float* src[24]; float* dst[24]; };
int transpose256_table(hipStream_t s, const TransposeTable& tab);         // dst[i] = src[i]^T, [256,256] each
int embed_gather_fwd(hipStream_t s, int B, int L, const int32_t* idx, const float* table, const float* pos, float* out,
                     int out_bstride, int out_off);
int embed_grouped_bwd(hipStream_t s, int n_items, const int32_t* item_tok, const int32_t* item_ptr, const int32_t* rows,
                      float* dtable, const float* dnode);
int embed_list_bwd_small(hipStream_t s, int n, const int32_t* rows, const int32_t* ids, float* dtable, const float* dnode,
                         int table_rows);
int embed_gather_bwd(hipStream_t s, int B, int L, const int32_t* idx, float* dtable, const float* dout,
                     int out_bstride, int out_off, int padding_idx);
int embed_gather_bwd_small(hipStream_t s, int B, int L, const int32_t* idx, float* dtable, const float* dout,
                           int out_bstride, int out_off, int padding_idx, int table_rows);
int combination_fwd(hipStream_t s, int M, const float* qk, const float* vtab, int ldv, const int32_t* mark, float* out,
                    float dropout, uint64_t seed, uint32_t site);
// part (optional): the workgroups store their partial dvtab [4*256] rows there (combination_bwd_blocks(M) rows) instead of
// adding to dvtab with atomics; the caller reduces them later (deferred_reduce)
int combination_bwd(hipStream_t s, int M, const float* qk, const float* vtab, int ldv, const int32_t* mark,
                    const float* dout, float* dqk, float* dvtab, int lddv, float dropout, uint64_t seed, uint32_t site,
                    float* part = nullptr);
int combination_bwd_blocks(int M);
// deferred column reductions (rowops.hip)
struct RedEntry { float* dst; const float* src; int width, n_part, stride; };
constexpr int RED_MAX = 96;
struct RedTable {
    int n = 0;
    int wg_start[RED_MAX + 1] = {0};
    RedEntry e[RED_MAX];
};
int deferred_reduce(hipStream_t s, RedTable& tab);       // dst[c] += sum_p src[p*stride + c]; empties the table
// y_rows (optional): output row r is stored at row y_rows[r]
int add_layernorm_fwd(hipStream_t s, int M, float* x, const float* res, const float* gamma, const float* beta,
                      float* y, float* stats, float dropout, uint64_t seed, uint32_t site, const int32_t* y_rows, const float* r1_row = nullptr,
                      const float* r1_col = nullptr,    // x += r1_row[r] * r1_col[:] before the dropout
                      const int32_t* slot2 = nullptr, float* y2 = nullptr,    // rows with slot2[r] >= 0 are also stored at y2[slot2[r]]
                      uint32_t idx0 = 0);       // dropout element index of row 0, column 0 (a row slice of the site: decoder lanes)
bool gemm_bf16_k256_try(hipStream_t s, int M, int N, int K, const float* A, int lda, const uint16_t* Bb, int ldb, float* C,
                        int ldc, const float* bias, int flags, const int32_t* c_rows, const float* relu_mask, int* rc);
bool linear_ln_bf16_try(hipStream_t s, int M, int K, const float* X, int ldx, const uint16_t* Wb, int ldb, const float* bias,
                        const float* res, const float* gamma, const float* beta, float* sum, float* y, float* stats,
                        float dropout, uint64_t seed, uint32_t site, const int32_t* y_rows, const float* r1_row,
                        const float* r1_col, int* rc, bool force = false);
int add_layernorm_bwd(hipStream_t s, int M, const float* dy, const float* sum, const float* stats, const float* gamma,
                      float* ds, float* dx_drop, float* dgamma, float* dbeta, float dropout, uint64_t seed,
                      uint32_t site, const int32_t* rows = nullptr,    // rows: dy and ds are row-mapped (dy[rows[r]], ds[rows[r]])
                      float* part = nullptr,    // part: [add_layernorm_bwd_blocks(M), 512] partial {dgamma | dbeta} rows instead of atomics
                      // row_w (with part): partial rows are 1024 wide, {dgamma | dbeta | sum_r dx[r,:] | sum_r row_w[r] dx[r,:]}
                      // with dx the un-dropped gradient rows (what dx_drop receives)
                      const float* row_w = nullptr, uint32_t idx0 = 0);
int add_layernorm_bwd_blocks(int M);
int gcn_bias_unfold(hipStream_t s, const float* W2, const float* b1, const float* dc, float* dW2, float* db1);
struct UnfoldEntry { const float *W2, *b1, *dc; float *dW2, *db1; };
struct UnfoldTable { int n = 0; UnfoldEntry e[16]; };
int gcn_bias_unfold_all(hipStream_t s, const UnfoldTable& tab);      // every GCN layer in one launch
// dW2[l] += dW21[l] W1[l]^T and dW1[l] += W2[l]^T dW21[l] for every layer in ONE launch (gemm_small.hip; [256,256] matrices)
int gcn_fold_weights(hipStream_t s, int n_layers, const float* const* W2, const float* const* W1, const float* const* b1,
                     float* W21, float* W21t, float* c21);       // W21 = W2 W1, W21t = its transpose, c21 = W2 b1: one launch
int gcn_unfold_products(hipStream_t s, int n_layers, const float* const* dW21, const float* const* W1, const float* const* W2,
                        float* const* dW1, float* const* dW2);
int colsum(hipStream_t s, int M, int N, const float* X, int ldx, float* out, const float* row_weight = nullptr);
int rank2_rows(hipStream_t s, int M, const float* g, const float* w, float* out);   // out[M,256] = g[M,2] w[2,256]
// index-list row movers (W floats per row): mode 0 out[r]=in[src[r]], 1 out[dst[r]]=in[r], 2 out[dst[r]]+=in[r],
// 3 out[dst[r]]=in[src[r]]
int prep(hipStream_t s, int B, int L, int S, int T, const int32_t* sou, const int32_t* sub, const int32_t* tar,
         int32_t* mem_valid, int32_t* tar_valid, float* pos_code, float* pos_tar, int R, const int32_t* rows,
         int32_t* compact_row, int32_t* iota, float* loss_sum = nullptr, int32_t* n_tok = nullptr,    // the two scalars are zeroed
         // optional: slot of every compact node in the ascending code-row / memory-row lists (-1: not listed)
         int Nc = 0, const int32_t* code_rows = nullptr, int Cc = 0, int32_t* code_slot = nullptr,
         const int32_t* mem_rows = nullptr, int Mc = 0, int32_t* mem_slot = nullptr,
         // optional: computed target rows (fira_batch.dec_off) -> row_bt[compact row] = flat b*T+t, rows_c[k] = compact row of
         // head row k; compact_row is then indexed by compact row
         const int32_t* dec_off = nullptr, int32_t* row_bt = nullptr, int32_t* rows_c = nullptr,
         // optional: the computed memory rows as ragged attention keys -- mem_dst [Mc] ascending dense slots (b*(L+S) + local)
         // -> mem_off [B+1] (commit b's range of the list) and mem_valid_c [Mc] (key mask per listed row)
         const int32_t* mem_dst = nullptr, int32_t* mem_off = nullptr, int32_t* mem_valid_c = nullptr);
// the decoder's token embedding on a list of target rows (row_bt[r] = flat b*T + t) and its backward
struct AdamRowsView;
int embed_rows_fwd(hipStream_t s, int R, int T, const int32_t* row_bt, const int32_t* idx, const float* table,
                   const float* pos, float* out, const AdamRowsView* vw = nullptr);   // vw: the table under a row-sparse Adam
int embed_rows_bwd(hipStream_t s, int R, const int32_t* row_bt, const int32_t* idx, float* dtable, const float* dout,
                   int padding_idx);
int node_features(hipStream_t s, int Nc, const int32_t* node_rows, int N, int L, int S, const int32_t* sou,
                  const int32_t* sub, const int32_t* ast, const float* emb, const float* ast_emb, const float* pos_code,
                  float* X, const int32_t* slot2 = nullptr, float* X2 = nullptr,     // rows with a slot also go to X2[slot]
                  const AdamRowsView* vw = nullptr);                                 // vw: `emb` under a row-sparse Adam
int rows_move(hipStream_t s, int mode, int R, int W, float* out, const float* in, const int32_t* src, const int32_t* dst);
int rows_move_ld(hipStream_t s, int mode, int R, int W, float* out, int ld_out, const float* in, int ld_in,
                 const int32_t* src, const int32_t* dst);
// compact[r,:] = src[rows[r],:] ;  dst[rows[r],:] += compact[r,:]
int rows_gather_idx(hipStream_t s, int R, float* compact, const float* src, const int32_t* rows);
int rows_scatter_add_idx(hipStream_t s, int R, const float* compact, float* dst, const int32_t* rows);
int fill_pos_tables(hipStream_t s, int L, float* pos_code, int T, float* pos_tar);
int tar_mask(hipStream_t s, int n, const int32_t* tar, int32_t* valid);
// beam re-ordering of the decoder self-attention cache: dst[r, 0:len) = src[parent[r], 0:len) for nl layers of K and V,
// plus the key-valid history; then hist_dst[r, step] = tokens[r] != 0
int permute_cache(hipStream_t s, int nl, int BR, int T, int len, const int32_t* parent, const float* ksrc,
                  const float* vsrc, float* kdst, float* vdst, const int32_t* hist_src, int32_t* hist_dst);
int mark_history(hipStream_t s, int BR, int T, int step, const int32_t* tokens, int32_t* hist);
// the two in one launch for the decode step: hist[r, step] = tokens[r] != 0;  x[r] = table[tokens[r]] + pos_row
int decode_embed(hipStream_t s, int BR, int T, int step, const int32_t* tokens, const float* table, const float* pos_row,
                 float* x, int32_t* hist);

// q_off (optional, [B+1]): ragged query rows -- batch entry b's queries are rows q_off[b] .. q_off[b+1] of Q / O / dO / dQ
// (at most Tq of them); self_kv: its keys / values are the same rows of K / V (/ dK / dV).  key_valid stays dense.
int attention_fwd(hipStream_t s, int B, int H, int Tq, int Tk, const float* Q, int ldq, const float* K, int ldk,
                  const float* V, int ldv, const int32_t* key_valid, int causal, int q_pos0, float* O, int ldo,
                  const int32_t* q_off = nullptr, int self_kv = 0,
                  int bf16 = 0,       // 1: operands of the four matmuls rounded to bf16, bf16 MFMA, fp32 accumulate / soft-max
                  // k_off (optional, [B+1], cross attention): ragged KEY rows -- batch entry b's keys / values are rows
                  // k_off[b] .. k_off[b+1] of K / V (at most Tk) and key_valid is indexed by the same compact rows
                  const int32_t* k_off = nullptr);
int attention_fwd_ex(hipStream_t s, int B, int H, int Tq, int Tk, const float* Q, int ldq, const float* K, int ldk,
                     const float* V, int ldv, const int32_t* key_valid, int causal, int q_pos0, float* O, int ldo,
                     int kb, int kvb, int qpk, const int32_t* q_off = nullptr, int self_kv = 0, int bf16 = 0,
                     const int32_t* k_off = nullptr);
// one query per row (decode step): K/V streamed once per (commit, head) over the valid keys only; optional merged new key
int decode_attention(hipStream_t s, int BR, int H, int Tk, const float* Q, int ldq, const float* K, int ldk,
                     const float* V, int ldv, const int32_t* key_valid, float* O, int ldo, int kb, int kvb, int qpk,
                     const float* Knew = nullptr, const float* Vnew = nullptr, int ldn = 0, float* Kc_out = nullptr,
                     float* Vc_out = nullptr, const int32_t* k_off = nullptr);
// the decode step's self-attention block of every row as one launch (attention.hip): one-query attention over the <= 32 cached
// keys (+ this step's, appended), xa = LN(o Wo^T + bo + xres), qc = xa Wq^T + bq; WoT / WqT are the k-major copies
int decode_self_block(hipStream_t s, int BR, int Tk, int T, const float* qkv, float* Kc, float* Vc, const int32_t* hist,
                      const float* WoT, const float* bo, const float* xres, const float* gamma, const float* beta,
                      const float* WqT, const float* bq, float* xa, float* qc);
int decode_attention_kv16(hipStream_t s, int BR, int H, int Tk, const float* Q, int ldq, const uint16_t* K, int ldk,
                          const uint16_t* V, int ldv, const int32_t* key_valid, float* O, int ldo, int kb, int kvb, int qpk,
                          const int32_t* k_off = nullptr);
int rows_to_bf16(hipStream_t s, int64_t n, const float* in, uint16_t* out);      // out[i] = bf16(in[i]) (RNE), n % 4 == 0
int rows_from_bf16(hipStream_t s, int64_t n, const uint16_t* in, float* out);    // out[i] = float(in[i]), n % 4 == 0
int attention_bwd(hipStream_t s, int B, int H, int Tq, int Tk, const float* Q, int ldq, const float* K, int ldk,
                  const float* V, int ldv, const int32_t* key_valid, int causal, int q_pos0, const float* O, int ldo,
                  const float* dO, int lddo, float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv,
                  const int32_t* q_off = nullptr, int self_kv = 0, int bf16 = 0, const int32_t* k_off = nullptr);
int copy_score_fwd(hipStream_t s, int B, int T, int S, const float* src, const float* tgt, const float* w,
                   const float* bias, float* score);
// mem_valid (optional, [B/qpk, S]): slots with 0 are skipped (score 0 / zero gradient): they are masked to -1e9 later
int copy_score_fwd_ex(hipStream_t s, int B, int T, int S, const float* src, const float* tgt, const float* w,
                      const float* bias, float* score, int qpk, const int32_t* mem_valid,
                      const int32_t* tar_label = nullptr, int V = 0,
                      const int32_t* t_off = nullptr);      // [B+1] ragged target rows (tgt / score rows of commit b)
// part (optional): [copy_score_bwd_blocks(B, S), COPY_PART_STRIDE] partial rows {dw[256] | dbias} instead of atomics
int copy_score_bwd_ex(hipStream_t s, int B, int T, int S, const float* src, const float* tgt, const float* w,
                      const float* dscore, float* dsrc, float* dtgt, float* dw, float* dbias, const int32_t* mem_valid,
                      float* part = nullptr, const int32_t* t_off = nullptr);
int copy_score_bwd_blocks(int B, int S);
constexpr int COPY_PART_STRIDE = 264;
// gate_logits == nullptr: the 2-way gate LinearProb(x) = x wp^T + bp is formed inside (x [R,256], wp [2,256], bp [2])
int decode_dist(hipStream_t s, int R, int V, int S, const float* logits, int ldl, const float* score,
                const int32_t* mem_valid, int qpk, const float* gate_logits, float* dist, int32_t* best_id,
                float* best_p, const float* x = nullptr, const float* wp = nullptr, const float* bp = nullptr);
int inv_count(hipStream_t s, const int32_t* n_tok, float* out);
// ids / vals [R, k]: the k largest entries of every logits row, value descending, ties by ascending index (beam.hip; k <= 8)
int row_topk(hipStream_t s, int R, int V, int k, const float* logits, int ldl, int32_t* ids, float* vals);
int copy_score_bwd(hipStream_t s, int B, int T, int S, const float* src, const float* tgt, const float* w,
                   const float* dscore, float* dsrc, float* dtgt, float* dw, float* dbias);
int head_loss(hipStream_t s, int BT, int T, int V, int S, const int32_t* compact_row, float* logits, int ldl,
              float* score, const int32_t* mem_valid, float* gate_logits, const int32_t* tar_label, float* loss_sum,
              int32_t* n_tok, int32_t* argmax_out, int want_grad,
              const int32_t* row_bt = nullptr);      // BT computed target rows, row_bt[r] = flat b*T + t (nullptr: r itself)
int adam_step(hipStream_t s, int64_t n, float* p, const float* g, float* m, float* v, float lr, float beta1,
              float beta2, float eps, int step, const float* scale_ptr, int scale_is_count = 0);
int adam_step_mb(hipStream_t s, int64_t n, float* p, const float* g0, const float* g1, float* m, float* v, float lr,
                 float beta1, float beta2, float eps, int step, const int32_t* n0, const int32_t* n1);
// Row-sparse Adam of the two vocabulary-sized embedding tables (copyhead.hip: adam_rows_kernel).  p / m / v: the flat buffers;
// off[t] / rows[t]: first float and row count of table t (0 = decoder.embedding, 1 = encoder.embedding; 256 floats per row);
// last [rows[0] + rows[1]]: the step up to which each row is current.
constexpr int ADAM_ROWS_K = 32;            // every K-th step updates every row: a catch-up spans at most K - 1 steps
struct AdamRowsHist { float bc1[ADAM_ROWS_K], bc2s[ADAM_ROWS_K]; };          // bias corrections of step j at [j % K]
struct AdamRowsTables { float *p, *m, *v; int64_t off[2]; int rows[2]; int32_t* last; };
// a lazily updated table seen by a forward gather (adam_rows.h: adam_rows_load): m / v / last of THAT table, rows current at
// step `to` once their owed zero-gradient updates are applied; last == nullptr: a plain table
struct AdamRowsView {
    const float *m = nullptr, *v = nullptr;
    const int32_t* last = nullptr;
    int to = 0;
    float lr = 0.f, beta1 = 0.f, beta2 = 0.f, eps = 0.f, gz = 0.f;
    AdamRowsHist h;
};
AdamRowsView adam_rows_view(const AdamRowsTables& tb, int table, float lr, float beta1, float beta2, float eps, int to);
struct AdamRowsLists { int n_lists; int end[4]; int table[4]; const int32_t* ids[4]; };   // end[k]: items of lists 0..k
// the step `step` on every row whose gradient row is not zero (every row when step % K == 0); g: the flat gradient buffer;
// normaliser 1 / max(*count, 1) if count else 1 / max(*n0, 1)
int adam_rows_step(hipStream_t s, const AdamRowsTables& tb, const float* g, float lr, float beta1, float beta2, float eps,
                   int step, const int32_t* n0, const float* count, int tables = 3);    // tables: bit t = table t takes part
// rows brought up to step `to` by zero-gradient updates: the listed ids (ls) or, ls == nullptr, every row
int adam_rows_catchup(hipStream_t s, const AdamRowsTables& tb, const AdamRowsLists* ls, float lr, float beta1, float beta2,
                      float eps, int to);

// ---- parameter layout ---------------------------------------------------------------------------------
struct ParamInfo {
    std::string name;
    int64_t offset = -1, numel = 0;
    int ndim = 0;
    int64_t shape[2] = {0, 0};
};

struct EncLayer {
    int64_t wqk, bqk, w2, b2, wo, bo, ln1g, ln1b, fc1w, fc1b, fc2w, fc2b, ln2g, ln2b;
};
struct DecLayer {
    int64_t wqkv, bqkv, wo_s, bo_s, lns_g, lns_b, wq_c, bq_c, wkv_c, bkv_c, wo_c, bo_c, lnc_g, lnc_b, w1, b1, w2, b2,
        lnf_g, lnf_b;
};
struct Layout {
    fira_dims d;
    std::vector<ParamInfo> infos;     // reference state_dict order
    int64_t total = 0;                // floats
    int64_t split = 0, live = 0;      // [0,split) decoder+head, [split,live) encoder, [live,total) dead tensors
    int64_t emb, ast_emb, mark_emb, w2_all, b2_all, dec_emb, wkv_all, bkv_all, wout, bout, ws, wt, wres, bres, wp, bp;
    std::vector<EncLayer> enc;
    std::vector<DecLayer> dec;
};
const Layout* get_layout(const fira_dims* d);     // cached per geometry; nullptr + error on bad dims

}  // namespace fira
