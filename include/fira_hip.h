/*
 * fira_hip.h — C ABI of libfira_hip.so: the MI355X (gfx950) compute path of the FIRA
 * commit-message model (GNN encoder + Transformer decoder with dual copy head).
 *
 * The reference (DJjjjhao/FIRA-ICSE) has no FFI: its hot path is a chain of stock PyTorch ops
 * (SURVEY.md §2.2).  Each entry point below replaces the reference lines cited next to it; a
 * binding (ctypes / cgo / JNI) only needs this header.  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller unless marked "host";
 *   - nothing allocates: scratch comes from the caller (fira_workspace_bytes);
 *   - work is enqueued on the hipStream_t passed as `void* stream` (NULL = default stream),
 *     nothing synchronises; distinct streams may be driven from distinct threads;
 *   - return 0 on success, non-zero on error (message: fira_last_error(), thread-local);
 *   - activations are row-major [rows, 256] fp32; nn.Linear weights are [out, in] row-major
 *     exactly as in the reference's state_dict; ids are int32.
 */
#ifndef FIRA_HIP_H
#define FIRA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FIRA_ABI_VERSION 10

/* ---- model geometry: reference run_model.py:30-46 (args) ---------------------------------- */
typedef struct fira_dims {
    int32_t sou_len;         /* 210 code-token nodes                                   */
    int32_t sub_len;         /* 160 sub-token nodes                                    */
    int32_t ast_len;         /* 280 AST + edit-operation nodes                         */
    int32_t tar_len;         /* 30 message positions                                   */
    int32_t d_model;         /* 256 (kernels are specialised for 256)                  */
    int32_t n_head;          /* 8   (head width 32)                                    */
    int32_t n_layer;         /* 6                                                      */
    int32_t vocab;           /* 24650                                                  */
    int32_t ast_vocab;       /* 71                                                     */
    int32_t d_ff;            /* 1024                                                   */
} fira_dims;

/* ---- one collated batch of commits (replaces the 8-tensor batch of Dataset.py:336-343) ---- */
typedef struct fira_batch {
    int32_t B;               /* commits in the batch                                   */
    int32_t nnz;             /* entries of the block-diagonal adjacency                */
    const int32_t* sou;      /* [B, sou_len]  code-token ids                           */
    const int32_t* tar;      /* [B, tar_len]  decoder input ids                        */
    const int32_t* mark;     /* [B, sou_len]  0 pad,1 deleted,2 context,3 added (dense; the kernels read code_mark) */
    const int32_t* ast_change; /* [B, ast_len]                                         */
    const int32_t* tar_label;/* [B, tar_len]  labels incl. copy ids (>= vocab)         */
    const int32_t* sub_token;/* [B, sub_len]                                           */
    /* ---- computed node list.  The encoder only runs on the nodes listed here; padded nodes (id 0 and no edge but
     * their self-loop) may be left out: they are masked wherever they could be read and get zero gradient
     * (SURVEY.md §8a N1).  Listing every node (identity lists) reproduces the reference's dense computation. ---- */
    int32_t n_nodes;         /* Nc: computed nodes of the batch                                              */
    const int32_t* node_rows;/* [Nc] ascending global node index b*N + local (N = sou+sub+ast)               */
    const int32_t* rowptr;   /* [Nc + 1] CSR row offsets over the computed nodes                             */
    const int32_t* col;      /* [nnz] COMPACT node ids (position in node_rows)                               */
    const float*   val;      /* [nnz] D^-1/2 (A+I) D^-1/2 entries (fp32 of Dataset.py:291)                   */
    int32_t n_code;          /* Cc: computed code-token nodes (local < sou_len)                              */
    const int32_t* code_rows;/* [Cc] their compact ids, ASCENDING (v4: kernels invert the list by binary search)    */
    const int32_t* code_mark;/* [Cc] their mark value (0 pad,1 deleted,2 context,3 added)                    */
    int32_t n_mem;           /* Mc: computed memory nodes (local < sou_len + sub_len)                        */
    const int32_t* mem_rows; /* [Mc] their compact ids, ASCENDING                                            */
    const int32_t* mem_dst;  /* [Mc] their dense memory row b*(sou_len+sub_len) + local                      */
    const int32_t* head_rows;/* [n_head_rows] optional: flat (b*tar_len+t) indices of the target rows whose shifted
                                label is a vocabulary id (0 < label < vocab), ascending, no duplicates; only these
                                rows need the vocabulary GEMM in training.  NULL = all rows.           */
    int32_t n_head_rows;
    /* ---- optional (emb_items NULL = scatter with atomics per position): the code / sub-token positions of the batch
     * grouped by word id for the embedding gradient (gnn_transformer.py:46-52 backward).  Frequent tokens ('(' ';' ...)
     * occur hundreds of times per batch; one wave sums up to 32 positions of one id before touching the table row. */
    int32_t n_emb_items;
    const int32_t* emb_item_tok; /* [n_emb_items] word id of the item (never 0 = padding_idx); the items of one word
                                  * are ADJACENT (a word with a single item is written without atomics)           */
    const int32_t* emb_item_ptr; /* [n_emb_items + 1] offsets into emb_rows; at most 32 rows per item              */
    const int32_t* emb_rows;     /* COMPACT node ids (position in node_rows) of code / sub-token nodes, grouped by word id */
    /* ---- optional, together with emb_*: the computed AST / edit-operation nodes with a non-zero id, so that the whole
     * embedding gradient is taken from the compact node rows (no dense [B,650,256] scatter in the backward pass) */
    int32_t n_ast_items;
    const int32_t* ast_rows;     /* [n_ast_items] compact node ids                                               */
    const int32_t* ast_ids;      /* [n_ast_items] their ast_change ids (never 0)                                  */
    /* ---- optional (v5): computed TARGET rows.  The decoder of a training step only has to run on the positions that are
     * read by someone: position t of commit b matters as an attention key while tar[b,t] != 0 and as a loss row while the
     * shifted label tar_label[b,t+1] != 0 -- the padded tail (about 45 % of the B*30 rows on FIRA's data; it is computed by
     * the reference too, gnn_transformer.py:108-122, but carries zero loss weight, Model.py:82, and is masked as a key)
     * can be left out.  dec_off[b] .. dec_off[b+1] = the range of commit b's rows in the compact target-row layout; the
     * rows kept are the PREFIX t < dec_off[b+1] - dec_off[b] of its tar_len positions (so compact position == position,
     * causal order is preserved) and the prefix must cover every position with a non-zero id or shifted label.
     * NULL (or fira_train_opts.compact_dec == 0) = all B*tar_len rows. */
    const int32_t* dec_off;      /* [B + 1] ascending, dec_off[0] = 0, 1 <= dec_off[b+1] - dec_off[b] <= tar_len          */
    int32_t n_dec_rows;          /* dec_off[B]                                                                         */
    /* ---- optional (v8): a HOST copy of dec_off ([B + 1], host memory, valid for the duration of the call).  With it the
     * library may split the decoder's chain at a commit boundary into two commit-lanes on two streams (it needs the row
     * offset of that boundary on the host to size the launches); NULL = one lane.  Results do not depend on it. */
    const int32_t* dec_off_host;
} fira_batch;

typedef struct fira_train_opts {
    float    dropout;        /* 0.1 in the reference (Attention/FFN/Combination); 0 = off */
    float    gcn_dropout;    /* 0.2 (gnn_transformer.py:43)                             */
    uint64_t seed;           /* dropout stream seed; masks are re-derived in backward   */
    int32_t  compact_head;   /* 1 = run the vocabulary head only on rows whose label != 0 (results identical) */
    int32_t  dtype;          /* FIRA_F32 (reference arithmetic) or FIRA_BF16: nn.Linear products on the bf16 MFMA with
                                fp32 accumulation (operands rounded to bf16, everything stored in fp32) -- BASELINE
                                configs[2]; LayerNorm / soft-max / loss / Adam are fp32 in both modes           */
    int32_t  compact_dec;    /* (v5) 1 = run the decoder / output head on the rows of fira_batch.dec_off only: loss and
                                gradients are those of the dense run (only rows nobody reads are skipped); the dropout
                                element index of a decoder site is then (compact row) * 256 + column            */
    int32_t  zero_grads;     /* (v5) 1 = the call clears grads[0, live) itself, beside the encoder's forward pass on a
                                library-owned stream, instead of the caller filling 111 MB ahead of the step    */
} fira_train_opts;
#define FIRA_F32 0
#define FIRA_BF16 1
#define FIRA_F32X3 2      /* (v9, fira_gcn_layer_* and fira_combination_block_* only) fp32 data and accuracy, the product on the bf16 matrix cores: every operand as
                           * three bf16 terms hi + mid + lo, six of the nine term products, fp32 accumulation (see below)           */
#define FIRA_BF16X1 3     /* (v9, same entries) the engine's bf16 mode on the same machinery: ONE bf16 plane -- both operands rounded to
                           * bf16 once (RNE), fp32 accumulation; weight argument = the planes as for FIRA_F32X3                       */

const char* fira_last_error(void);
int         fira_abi_version(void);

/* ---- parameter layout: one flat fp32 buffer holding the 338 state-dict tensors ------------
 * (SURVEY.md §8b "Checkpoint").  Index order == reference state_dict order.                  */
int    fira_param_count(const fira_dims* d);
/* name_buf >= 128 bytes; shape has up to 2 entries (ndim returned). offset/numel in floats.  */
int    fira_param_info(const fira_dims* d, int index, char* name_buf, int64_t* offset, int64_t* numel,
                       int32_t* ndim, int64_t shape[2]);
int64_t fira_param_total(const fira_dims* d);            /* floats in the flat buffer        */
/* Gradient readiness groups of the flat buffer: [0, split) = output head + decoder (complete when the `mid` event of
 * fira_train_fwd_bwd fires), [split, live) = encoder, [live, total) = tensors no kernel touches. */
int    fira_param_groups(const fira_dims* d, int64_t* split, int64_t* live);

/* bytes of caller-provided scratch for a batch of B commits. mode: 0 = forward only, 1 = training */
size_t fira_workspace_bytes(const fira_dims* d, int B, int mode);
/* scratch for fira_decode_begin / fira_decode_step with n_beam hypotheses per commit */
size_t fira_decode_workspace_bytes(const fira_dims* d, int B, int n_beam);
/* (v6) the same for the flags fira_decode_begin_ex / fira_decode_step_ex will be called with: FIRA_DECODE_KV_BF16 adds the
 * bf16 copy of the cross-attention K|V (148 MB at batch 64); flags 0 == fira_decode_workspace_bytes */
size_t fira_decode_workspace_bytes_ex(const fira_dims* d, int B, int n_beam, int flags);

/* Per-kernel-class HIP-event profiling (bench.py's roofline leg).  Classes: 0 GEMM (work = FLOP), 1 CSR SpMM
 * (work = algorithmic bytes), 2 attention, 3 row ops, 4 copy score, 5 head/loss, 6 Adam.  report() synchronises,
 * returns per class the summed event time [ms], summed work and launch count since the last report, and resets. */
#define FIRA_PROF_NCLASS 7
/* two more classes are reported when n_class asks for them: 7 = the decoder's M = B*30 products (a sub-set of class 0's
 * launches, split out of it), 8 = the fused GCN-layer launches (work = FLOP of the product, bytes = algorithmic bytes) */
void fira_prof_enable(int on);
/* bytes (may be NULL): for the GEMM class, the algorithmic operand + result bytes 4*(M*K + N*K + M*N) of its launches */
int  fira_prof_report(int n_class, double* ms, double* work, double* bytes, int64_t* count);

/* =========================== op level (one reference op each) ============================== */

/* C[M,N] (+)= op(A)·op(B) (+ bias[n]) (relu).  Replaces every nn.Linear / its backward
 * (addmm/mm calls listed in SURVEY.md §2.3).  transA=0: A stored [M,K] (lda); transA=1: A stored
 * [K,M].  transB=1: B stored [N,K] (the nn.Linear weight layout); transB=0: B stored [K,N].
 * flags: bit0 relu, bit1 accumulate into C (C += ...), splitk >= 1 partitions K over grid.z (0 = automatic tile / split choice, what the fused entry points use)
 * (partials combined with fp32 atomics; requires accumulate semantics, C pre-initialised).     */
#define FIRA_GEMM_RELU 1
#define FIRA_GEMM_ACCUM 2
/* bits 4-5 (testing / tuning): force a tile: 0 auto, 1 128x128, 2 64x128, 3 64x64 */
#define FIRA_GEMM_TILE_SHIFT 4
int fira_gemm_f32(void* stream, int transA, int transB, int M, int N, int K,
                  const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                  const float* bias, int flags, int splitk);
/* The same product on the bf16 MFMA (v_mfma_f32_32x32x16_bf16): A and B are fp32 in memory, rounded to bf16
 * (nearest-even) while they are staged, accumulated and stored in fp32 -- torch.autocast(bfloat16) around F.linear.
 * Tile bits: 0 auto, 1 128x128, 2/3 64x64.  Products it does not take (M, N or K < 32, unaligned) run in fp32.   */
int fira_gemm_bf16(void* stream, int transA, int transB, int M, int N, int K,
                   const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                   const float* bias, int flags, int splitk);

/* bf16 weight shadows.  In bf16 mode the model-level entry points keep, inside the workspace, a bf16 copy of every 2-D
 * weight as stored ([out, in]) and transposed ([in, out]), refreshed by one launch per call: the forward product reads
 * the first, the data gradient the second -- both as the k-contiguous operand of the same kernel, at half the weight
 * bytes and without transposing loads.  Op-level forms of the two steps:
 *   fira_weight_shadow : Wb[rows, cols] = bf16(W), WbT[cols, rows] = bf16(W^T)                  (uint16_t = raw bf16)
 *   fira_gemm_bf16_wb  : C[M,N] (+)= A[M,K] (fp32, rounded while staged) . Bb[N,K]^T (+bias)(relu); ldb % 8 == 0      */
int fira_weight_shadow(void* stream, int rows, int cols, const float* W, uint16_t* Wb, uint16_t* WbT);
int fira_gemm_bf16_wb(void* stream, int M, int N, int K, const float* A, int lda, const uint16_t* Bb, int ldb,
                      float* C, int ldc, const float* bias, int flags, int splitk);

/* Y[r,:] = sum_j val[j] * X[col[j],:]  over CSR row r (d = 256).  The GCN aggregation
 * torch.bmm(edge.float(), x) (gnn_transformer.py:80); its backward is the same call because
 * the normalised adjacency is symmetric.  variant: 0 auto, 1 row gather (round 5: a wave owns four
 * consecutive rows and batches their index and neighbour-row requests), 2 LDS-staged (needs graph_rows > 0:
 * rows per graph, cols local to the graph's row block), 5 the round-1 gather, one row per wave (kept for
 * feature matrices of 2 GiB and more and as the A/B partner: FIRA_SPMM_ROWBATCH=0 makes variant 1 run it). */
int fira_csr_spmm_f32(void* stream, int n_rows, const int32_t* rowptr, const int32_t* col, const float* val,
                      const float* X, int ldx, float* Y, int ldy, int graph_rows, int variant);
/* The same aggregation with the block-dense MFMA variants and the measured variant choice:
 *   3  block-dense, fp32 MFMA   (graph_rows <= 512; a workgroup densifies 32 rows of one graph's adjacency into LDS
 *                                and multiplies them with the graph's feature rows: the literal torch.bmm of
 *                                gnn_transformer.py:80, fp32 arithmetic)
 *   4  block-dense, bf16 MFMA   (A_hat and X rounded to bf16, fp32 accumulate: torch.autocast's bmm, FIRA_BF16 only)
 *   0  auto: by density nnz / (n_rows * graph_rows) -- row-per-wave CSR below the measured crossover, block-dense above
 *      (profiles/r3_spmm_crossover.md); dtype (FIRA_F32 | FIRA_BF16) says whether bf16 operands are allowed.
 * CSR rows must hold sorted column ids (equal neighbours are summed), as data.py builds them.                       */
int fira_csr_spmm(void* stream, int n_rows, int64_t nnz, const int32_t* rowptr, const int32_t* col, const float* val,
                  const float* X, int ldx, float* Y, int ldy, int graph_rows, int variant, int dtype);

/* One GCN layer per launch (v6; reference gnn_transformer.py:74-86: fc1 -> torch.bmm(edge, x) -> fc2 -> dropout -> +x ->
 * LayerNorm).  There is no non-linearity between fc1, the aggregation and fc2, so the layer is evaluated in the folded form
 *     sum = dropout( (A_hat X) W21^T + b2 + (A_hat 1) c21^T ) + X ,   y = LayerNorm(sum)
 * with W21 = fc2.weight . fc1.weight [256,256] and c21 = fc2.weight . fc1.bias [256] formed by the caller; the forward
 * launch takes it TRANSPOSED (W21t = W21^T row-major, i.e. [in, out]: the kernel streams the weight k-major).  A workgroup
 * gathers 32 rows of A_hat X from the CSR adjacency into LDS, multiplies them with W21 on the MFMA (fp32 chains, or bf16
 * operands with fp32 accumulation for FIRA_BF16) and finishes the rows: the aggregated rows never travel through HBM.
 * CSR as for fira_csr_spmm_f32 (global column ids, sorted).  sum / y [n_rows,256], stats [n_rows,2] = {mean, 1/std} for
 * the backward LayerNorm; rowsum_out (optional) [n_rows] = A_hat 1; dropout element index = row*256 + col of `site`.
 *
 * Backward of the data path (A_hat symmetric):  V = A_hat dY ;  dX += V W21 ;  the weight gradient is dW21 = V^T X.
 * dY = gradient w.r.t. the un-dropped branch output (what fira_add_layernorm_bwd returns as dx_drop); W21 row-major
 * [out, in] (k-major for this product); V [n_rows,256] is written, dX [n_rows,256] is accumulated into.                  */
int fira_gcn_layer_fwd(void* stream, int n_rows, const int32_t* rowptr, const int32_t* col, const float* val,
                       const float* X, const float* W21t, const float* bias, const float* c21, const float* gamma,
                       const float* beta, float* sum, float* y, float* stats, float* rowsum_out, float dropout,
                       uint64_t seed, uint32_t site, int dtype);
int fira_gcn_layer_bwd(void* stream, int n_rows, const int32_t* rowptr, const int32_t* col, const float* val,
                       const float* dY, const float* W21, float* V, float* dX, int dtype);
/* (v9) dtype FIRA_F32X3 -- what the engine's fp32 mode runs: the gathered rows are split into three bf16 terms by the wave
 * that gathered them, the weight arrives pre-split: the weight argument (W21t / W21 above) is then the output of
 * fira_gcn_weight_planes for the matrix B [256 n][256 k] with  out = U B^T , i.e. B = W21 for fira_gcn_layer_fwd and
 * B = W21^T for fira_gcn_layer_bwd.  planes: n_mats x 3 x 65536 bf16 (hi | mid | lo plane, each in MFMA fragment order).
 * The result differs from the FIRA_F32 launch by fp32 rounding noise only (the dropped lo.lo, lo.mid, mid.lo terms are
 * below 2^-24 of a product).                                                                                           */
int fira_gcn_weight_planes(void* stream, int n_mats, const float* B, uint16_t* planes);
/* (v10) out [M, ldo] = x [M, 256] W^T + bias for N = a multiple of 256 output columns, W [N, 256] row-major given as the planes of
 * its N / 256 row blocks (fira_gcn_weight_planes(N / 256, W, planes)): the nn.Linear of the cross-attention K|V projection of all
 * layers on the encoder's memory rows (gnn_transformer.py:139-141; what fira_train_step runs on its auxiliary stream).
 * dtype FIRA_F32X3: three bf16 terms per operand (fp32-accurate); FIRA_BF16X1: both operands rounded to bf16 once.            */
int fira_linear_x3(void* stream, int M, int N, const float* x, int ldx, const uint16_t* w_planes, const float* bias, float* out,
                   int ldo, int dtype);
/* ... and its data gradient: dx [M, lddx >= 256] (+)= dy [M, K] W for W [K, 256] row-major with K a multiple of 256 (the d-memory
 * products of the decoder's backward pass, K = 1024 per layer pair).  wt_planes: the planes of the TRANSPOSED [256, 256] row
 * blocks of W (block kb transposed, then fira_gcn_weight_planes), K / 256 of them.                                           */
int fira_linear_dgrad_x3(void* stream, int M, int K, const float* dy, int lddy, const uint16_t* wt_planes, float* dx, int lddx,
                         int accumulate, int dtype);
/* (v10) The generator projection's data gradient (Model.py:54 backward): dx [M, lddx >= 256] += dy [M, K] W for W [K, 256], K ANY
 * size (the vocabulary; dy's row pitch lddy >= K, a multiple of 4 floats; columns past K are never read as operands).  The
 * reduction is split over workgroups and summed with float atomics: dx must hold what the product is added to (zeros for the
 * plain product).  planes_ws: fira_dgrad_x3_splitk_planes_bytes(K) bytes of scratch (the planes of W's transposed row blocks,
 * written by the call).  dtype FIRA_F32X3 | FIRA_BF16X1.                                                                     */
size_t fira_dgrad_x3_splitk_planes_bytes(int K);
int fira_dgrad_x3_splitk(void* stream, int M, int K, const float* dy, int lddy, const float* W, uint16_t* planes_ws, float* dx,
                         int lddx, int dtype);
/* The same for fira_combination_block_{fwd,bwd} with dtype FIRA_F32X3: the forward launch takes, in its WqT argument, the planes
 * of the THREE stacked matrices Wq | Wk | Wo as nn.Linear stores them ([out][in]; WkT / WoT are then ignored, pass WqT again);
 * the backward launch takes, in its Wo argument, the planes of Wq^T | Wk^T | Wo^T (Wqk is then ignored).                      */

/* One Combination block (gnn_transformer.py:176-205 with combination_layer.py:7-17) on n_rows code rows as ONE launch
 * (round 5, csrc/comb_fused.hip; dtype FIRA_F32, or FIRA_BF16 = the operands of the three products rounded to bf16, fp32
 * accumulation and storage):
 *     q | k = Xc [Wq | Wk]^T + bqk ;   c = dropout_gate( g0 k + g1 vtab[mark] ),  (g0, g1) = softmax(q k / sqrt 32, q v / sqrt 32)
 *     sum   = dropout_out( c Wo^T + bo ) + Xc ;   y[y_rows[r]] = LayerNorm(sum[r]) ;   stats[r] = {mean, 1/std}
 * WqT / WkT / WoT are the three nn.Linear weights TRANSPOSED ([256 in][256 out], contiguous): the kernel streams them
 * k-major.  vtab: the value projection of the 4-row mark table, row m at vtab + m*ldv.  q|k [n_rows,512] and c [n_rows,256]
 * are stored for the backward pass (fira_combination_bwd reads q|k).  y_rows (optional): output row map (the code rows of
 * the node buffer); dropout element index = row*256 + col under site_gate / site_out (the masks fira_combination_fwd /
 * fira_add_layernorm_fwd draw).                                                                                          */
int fira_combination_block_fwd(void* stream, int n_rows, const float* Xc, const float* WqT, const float* WkT,
                               const float* WoT, const float* bqk, const float* bo, const float* vtab, int ldv,
                               const int32_t* mark, float* qk, float* c, const float* gamma, const float* beta, float* sum,
                               float* y, const int32_t* y_rows, float* stats, float dropout, uint64_t seed,
                               uint32_t site_gate, uint32_t site_out, int dtype);

/* Backward of the block, one launch + one reduction launch: with dy = dG[rows[r]] (the gradient w.r.t. the block's output rows)
 *     ds = LayerNorm'(dy; sum, stats, gamma) ;  dYc = ds * mask_out ;  dc = dYc Wo ;  (dq, dk, dv) = gate'(q, k, vtab[mark], dc * mask_gate)
 *     dG[rows[r]] = ds + dq Wq + dk Wk
 * Wo [256,256] and Wqk [512,256] as nn.Linear stores them.  Written: dYc [n_rows,256] and dqk [n_rows,512] (the operands of the
 * weight gradients dWo = dYc^T c, dWqk = dqk^T Xc); accumulated into: dgamma, dbeta [256], dvtab (row m at dvtab + m*lddv).
 * part: scratch of fira_combination_block_bwd_part_floats() floats.                                                       */
int fira_combination_block_bwd(void* stream, int n_rows, float* dG, const int32_t* rows, const float* sum, const float* stats,
                               const float* gamma, const float* Wo, const float* Wqk, const float* qk, const float* vtab,
                               int ldv, const int32_t* mark, float* dYc, float* dqk, float* dgamma, float* dbeta,
                               float* dvtab, int lddv, float* part, float dropout, uint64_t seed, uint32_t site_gate,
                               uint32_t site_out, int dtype);
int fira_combination_block_bwd_part_floats(void);

/* out[(b*out_bstride + out_off + i), :] = table[idx[b*L + i], :] (+ pos[i,:])  — the embedding
 * gathers of gnn_transformer.py:46-52,110-113 written straight into the node buffer.           */
int fira_embed_gather_fwd(void* stream, int B, int L, const int32_t* idx, const float* table,
                          const float* pos, float* out, int out_bstride, int out_off);
/* dtable[idx,:] += dout[row,:]  (rows with idx == padding_idx skipped; padding_idx < 0: none) */
int fira_embed_gather_bwd(void* stream, int B, int L, const int32_t* idx, float* dtable,
                          const float* dout, int out_bstride, int out_off, int padding_idx);

/* CombinationLayer (combination_layer.py:7-17): c = softmax2(q*k/s, q*v/s) . (k, v), s = sqrt(32),
 * v = vtab[mark[row]] (vtab = linear_layers[2] applied to the 4-row mark table).  qk: [M,512] = [q|k]. */
int fira_combination_fwd(void* stream, int M, const float* qk, const float* vtab, const int32_t* mark,
                         float* out, float dropout, uint64_t seed, uint32_t stream_id);
int fira_combination_bwd(void* stream, int M, const float* qk, const float* vtab, const int32_t* mark,
                         const float* dout, float* dqk, float* dvtab /* [4,256] += */,
                         float dropout, uint64_t seed, uint32_t stream_id);

/* y = LayerNorm(dropout(x) + res) * gamma + beta  (eps 1e-5), rows of 256.  x is overwritten with the
 * pre-norm sum (kept for backward); stats[row] = {mean, rstd}.  Post-LN residual blocks of
 * gnn_transformer.py:86,161,174,205.                                                            */
int fira_add_layernorm_fwd(void* stream, int M, float* x, const float* res, const float* gamma,
                           const float* beta, float* y, float* stats, float dropout, uint64_t seed,
                           uint32_t stream_id);
/* A post-LN residual block split at its LayerNorm (gnn_transformer.py:161,174): the closing product stores the PRE-NORM sum
 *     sum[M,256] = dropout(X W^T + bias) + res          X [M,K] (row pitch ldx), W [256,K], K in {128..1024 step 128}
 * (what fira_add_layernorm_fwd forms from the plain product; same dropout element indices), and the product that consumes the
 * block's output normalises its A rows itself:
 *     Y[M,N] = LN(S) W^T + bias (+relu)                 S [M,256] (row pitch lds), W [N,256];  x_out [M,256] = LN(S) and
 *                                                       stats_out [M,2] = {mean, rstd} are stored as well
 * -- one launch less per block on the dependent chain than product + row kernel + product.  fp32 MFMA, operands 16-byte
 * aligned; returns an error for shapes the coalesced tile kernel does not take (the engine then uses the three launches). */
int fira_linear_presum_f32(void* stream, int M, int K, const float* X, int ldx, const float* W, const float* bias,
                           const float* res, float* sum, float dropout, uint64_t seed, uint32_t stream_id);
int fira_ln_linear_f32(void* stream, int M, int N, const float* S, int lds, const float* W, const float* bias, float* Y,
                       int ldy, int relu, const float* gamma, const float* beta, float* x_out, float* stats_out);
/* (v6) The LayerNorm BACKWARD of such a block inside the data-gradient product that consumes it (the decoder's backward
 * chain: gnn_transformer.py:161,174 backward followed by the closing nn.Linear's data gradient): with dy [M,256] the
 * gradient w.r.t. the block's output, sum / stats the saved pre-norm rows and {mean, 1/std},
 *     ds = LayerNorm'(dy)  [M,256]   (gradient w.r.t. the pre-norm sum = the residual-branch gradient; must not alias dy)
 *     dx_drop = ds * dropout mask of `site`            (the weight gradient's operand)
 *     dX[M,N] = dx_drop . Wt   (Wt row-major [256, N] = the closing nn.Linear's weight [256 out, N in]; relu_mask [M,N]
 *                               optional: dX zeroed where mask <= 0 -- the FFN's ReLU backward)
 *     dgamma / dbeta [256] += column sums (through `part`, caller scratch of ceil(M/32) * 512 floats).                    */
int fira_ln_bwd_linear_f32(void* stream, int M, int N, const float* dy, const float* Wt, float* dX, const float* relu_mask,
                           const float* sum, const float* stats, const float* gamma, float* ds, float* dx_drop,
                           float* dgamma, float* dbeta, float* part, float dropout, uint64_t seed, uint32_t site);
/* bf16 mode twin (K = 256): Wb is the [256,256] bf16 shadow of the weight (fira_weight_shadow), row pitch ldb a multiple
 * of 8; X is rounded to bf16 while staged, fp32 accumulation, fp32 LayerNorm.  M >= 64.                                  */
int fira_linear_layernorm_bf16_fwd(void* stream, int M, const float* X, int ldx, const uint16_t* Wb, int ldb,
                                   const float* bias, const float* res, const float* gamma, const float* beta,
                                   float* sum, float* y, float* stats, float dropout, uint64_t seed,
                                   uint32_t stream_id);
/* ds = dLN/d(sum); dgamma/dbeta += ...;  if dx_drop != NULL: dx_drop = ds * mask/(1-p)           */
int fira_add_layernorm_bwd(void* stream, int M, const float* dy, const float* sum, const float* stats,
                           const float* gamma, float* ds, float* dx_drop, float* dgamma, float* dbeta,
                           float dropout, uint64_t seed, uint32_t stream_id);

/* Dropout is counter-based: the keep/scale factor of element `idx` of a dropout site is a pure function of
 * (seed, site, idx), re-derived in the backward kernels (no mask tensors).  Sites of the model-level entry points:
 * FIRA_SITE(layer, kind); idx = row * 256 + column with `row` the kernel's own row index: the position in
 * fira_batch.code_rows for the two Combination sites, the compact node id for the GCN site, b*tar_len + t for the
 * decoder sites.  fira_dropout_mask writes out[i] = 0 or 1/(1-p) for idx = 0..n-1 (all 1 when p == 0): the hook that
 * lets a CPU restatement of the model run with the SAME masks (tests/test_dropout_gpu.py).
 * Reference sites: combination_layer.py:15-16, gnn_transformer.py:205 (Combination), :83 (GCN), :161 (both
 * attentions), :174 (FeedForward).                                                                               */
#define FIRA_SITE_GATE 0      /* dropout on the gated mix inside the Combination      */
#define FIRA_SITE_COMB_OUT 1  /* on the Combination's output projection               */
#define FIRA_SITE_GCN 2       /* on the GCN's fc2 output (p = gcn_dropout)            */
#define FIRA_SITE_SELF 3      /* on the self-attention output projection              */
#define FIRA_SITE_CROSS 4     /* on the cross-attention output projection             */
#define FIRA_SITE_FFN 5       /* on the feed-forward output                           */
#define FIRA_SITE(layer, kind) ((uint32_t)((layer) * 8 + (kind) + 1))
int fira_dropout_mask(void* stream, uint64_t seed, uint32_t site, int64_t n, float p, float* out);

/* colsum[n] += sum_m X[m,n]  (bias gradients) */
int fira_colsum_f32(void* stream, int M, int N, const float* X, int ldx, float* out);

/* Multi-head attention core of gnn_transformer.py:137-158 (head width 32): per (b, head)
 * O = softmax(mask(Q K^T / sqrt(32), -1e9)) V.  Q rows [B*Tq, ldq], K/V rows [B*Tk, ldk/ldv]; head h at
 * columns h*32.. ; key_valid [B,Tk] (0 = masked); causal != 0 adds key <= query + q_pos0.        */
int fira_attention_fwd(void* stream, int B, int H, int Tq, int Tk, const float* Q, int ldq,
                       const float* K, int ldk, const float* V, int ldv, const int32_t* key_valid,
                       int causal, int q_pos0, float* O, int ldo);
int fira_attention_bwd(void* stream, int B, int H, int Tq, int Tk, const float* Q, int ldq,
                       const float* K, int ldk, const float* V, int ldv, const int32_t* key_valid,
                       int causal, int q_pos0, const float* O, int ldo, const float* dO, int lddo,
                       float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv);
/* Precondition of all attention entry points: every (batch entry, query) sees at least ONE unmasked key.  The reference's
 * masked_fill(-1e9) + softmax would then average V uniformly over ALL Tk keys, masked ones included; the kernels never read
 * rows of masked keys, so an all-masked row returns 0 instead (FIRA batches cannot produce one: position 0 of every
 * commit is its <start> token, Dataset.py:140, and a causal query always sees itself).
 * The forms the engine calls (v5, k_off: v6):
 *   q_off  (optional, [B+1]) ragged query rows: batch entry b's queries are rows q_off[b] .. q_off[b+1] of Q / O / dO /
 *          dQ (at most Tq of them: the decoder's computed target rows, fira_batch.dec_off); self_kv != 0: its keys and
 *          values are the same rows of K / V / dK / dV (self-attention).  key_valid stays dense [B, Tk].
 *   k_off  (optional, [B+1], not with self_kv) ragged KEY rows: batch entry b's keys / values are rows k_off[b] ..
 *          k_off[b+1] of K / V / dK / dV (at most Tk of them) and key_valid is indexed by the same compact rows
 *          (key_valid[k_off[b] + i]).  The engine keeps the cross-attention K|V of the COMPUTED memory rows only
 *          (fira_batch.mem_rows: ~47 % of the B*370 slots), in that layout -- the padded slots of the reference's
 *          [B,370] memory (Model.py:48,61) are masked keys whose soft-max weight is exactly 0.
 *   dtype  FIRA_BF16: the operands of the four matmuls (Q K^T, P V and their gradients) are rounded to bf16 and
 *          multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; scaling, mask, soft-max in fp32 -- what
 *          torch.autocast does to Attention.forward (gnn_transformer.py:149-156).  FIRA_F32: exact fp32 MFMA chains. */
int fira_attention_fwd_ex(void* stream, int B, int H, int Tq, int Tk, const float* Q, int ldq,
                          const float* K, int ldk, const float* V, int ldv, const int32_t* key_valid,
                          int causal, int q_pos0, float* O, int ldo, const int32_t* q_off, int self_kv, int dtype,
                          const int32_t* k_off);
int fira_attention_bwd_ex(void* stream, int B, int H, int Tq, int Tk, const float* Q, int ldq,
                          const float* K, int ldk, const float* V, int ldv, const int32_t* key_valid,
                          int causal, int q_pos0, const float* O, int ldo, const float* dO, int lddo,
                          float* dQ, int lddq, float* dK, int lddk, float* dV, int lddv,
                          const int32_t* q_off, int self_kv, int dtype, const int32_t* k_off);

/* Attention of ONE query per row (the K/V-cached decode step of run_model.py:256): O[b, h*32..] = softmax(q.K^T/sqrt(32)
 * over the valid keys) V.  Row b uses K/V batch entry b / qpk (kb rows of ldk floats per entry; key_valid [BR/qpk, kvb]):
 * the qpk beam rows of a commit share its memory.  Masked keys are never read.  Knew / Vnew (optional, qpk == 1): row b's
 * key / value number Tk-1 is taken from Knew[b*ldn..] / Vnew[b*ldn..] (the merged q|k|v projection's output) and is
 * appended to Kc_out / Vc_out (same geometry as K / V) for the later steps.  k_off (v6, optional, [BR/qpk + 1], not
 * with Knew): ragged key rows as in fira_attention_fwd_ex -- entry e's keys are rows k_off[e] .. k_off[e+1] of K / V and
 * key_valid is indexed by the same rows (kb / kvb are then ignored).                                                 */
int fira_decode_attention(void* stream, int BR, int H, int Tk, const float* Q, int ldq, const float* K, int ldk,
                          const float* V, int ldv, const int32_t* key_valid, float* O, int ldo, int kb, int kvb, int qpk,
                          const float* Knew, const float* Vnew, int ldn, float* Kc_out, float* Vc_out,
                          const int32_t* k_off);

/* CopyNet score (Model.py:15-18): score[b,t,s] = w . tanh(src[b,s,:] + tgt[b,t,:]) + bias, never
 * materialising the [B,T,S,256] tensor.  bwd re-computes tanh.                                   */
int fira_copy_score_fwd(void* stream, int B, int T, int S, const float* src, const float* tgt,
                        const float* w, const float* bias, float* score);
int fira_copy_score_bwd(void* stream, int B, int T, int S, const float* src, const float* tgt,
                        const float* w, const float* dscore, float* dsrc /* = */, float* dtgt /* += */,
                        float* dw /* += */, float* dbias /* += */);

/* Output head + loss (Model.py:54-86), one workgroup per target row bt = b*T + t (BT rows):
 * p = [g0*softmax(logits) ; g1*softmax(mask(score,-1e9))], loss = -log(clamp(p[label],1e-10,1)),
 * label = tar_label shifted left (Model.py:71-77).  compact_row[bt] = row of `logits` holding (b,t), or -1
 * when that row was not computed (NULL = identity).  loss_sum / n_tok are device scalars (+=).  With
 * want_grad the three inputs are overwritten in place by dlogits / dscore / dgate_logits.  With
 * argmax_out != NULL the teacher-forced argmax id of every row is written ('dev' stage, Model.py:85-86). */
int fira_head_loss(void* stream, int BT, int T, int V, int S, const int32_t* compact_row,
                   float* logits, int ldl, float* score /* [BT,S] */, const int32_t* mem_valid /* [B,S] */,
                   float* gate_logits /* [BT,2] */, const int32_t* tar_label /* [B,T] */,
                   float* loss_sum, int32_t* n_tok, int32_t* argmax_out, int want_grad);

/* Adam (run_model.py:396 torch.optim.Adam defaults) over a flat buffer.  grad_scale (device scalar,
 * may be NULL) multiplies g first: 1/n_tok of run_model.py:105 without a host sync.              */
int fira_adam_step(void* stream, int64_t n, float* p, const float* g, float* m, float* v,
                   float lr, float beta1, float beta2, float eps, int step, const float* inv_scale_ntok);
/* Adam for a step whose batch ran as one or two micro-batches: g = g0 (+ g1, may be NULL), each the gradient of the
 * micro-batch's loss SUM; the normaliser 1 / max(n_tok0 (+ n_tok1), 1) of run_model.py:105 is formed inside the kernel
 * from the device token counters.                                                                                  */
int fira_adam_step_mb(void* stream, int64_t n, float* p, const float* g0, const float* g1, float* m, float* v,
                      float lr, float beta1, float beta2, float eps, int step, const int32_t* n_tok0,
                      const int32_t* n_tok1);
/* Data-parallel form (run_model.py:105 over the GLOBAL batch): `count` is a device float holding the all-reduced token count;
 * the gradient is scaled by 1 / max(count, 1) inside the kernel -- no reciprocal / clamp launches between the collective and
 * the update.  fira_pack_stats writes the pair a rank contributes to that collective: out2 = {loss_sum, (float) n_tok}.  */
int fira_adam_step_count(void* stream, int64_t n, float* p, const float* g, float* m, float* v,
                         float lr, float beta1, float beta2, float eps, int step, const float* count);
int fira_pack_stats(void* stream, const float* loss_sum, const int32_t* n_tok, float* out2);
/* out[0] = 1 / max(n_tok[0], 1) on the device */
int fira_inv_count(void* stream, const int32_t* n_tok, float* out);

/* measurement aid: n dependent tiny kernels on `stream`, optionally forking the library's side stream after each (mode 1),
 * recording an event only (mode 2) or forking every 8th kernel (mode 3); scratch: >= 128 floats                       */
int fira_debug_chain(void* stream, int n, int mode, float* scratch);

/* =========================== model level (one call per step) =============================== */

/* forward + backward of TransModel.forward(..., 'train') (Model.py:38-84, run_model.py:104-108):
 * grads (flat, same layout as params) += d(loss_sum)/dparams; loss_sum / n_tok are device scalars
 * that are overwritten.  A fresh gradient: zero grads[0, live) first, or set opts->zero_grads and the
 * call clears it itself (beside the encoder's forward pass, on a library-owned stream).            */
int fira_train_fwd_bwd(void* stream, const fira_dims* d, const fira_batch* batch, const float* params,
                       float* grads, void* workspace, size_t workspace_bytes, const fira_train_opts* opts,
                       float* loss_sum, int32_t* n_tok, void* mid_event);
/* mid_event: optional hipEvent_t recorded on `stream` once the gradients of group [0, split) are final (after the
 * decoder backward, before the encoder backward), so that their RCCL all-reduce can overlap the rest.           */

/* (v8) Weight gradient as a panel product on the bf16 matrix cores (csrc/gemm_wgrad_panel.hip):
 *     C[M,N] += A^T B ;  colsum[m] += sum_k A[k][m]  (optional: the bias gradient)       A [K, lda], B [K, ldb] row-major
 * i.e. dW += dY^T X, db += column sums of dY for an nn.Linear with dY = A, X = B.  One workgroup owns a 256 x 256 tile of C
 * for a slab of K; every operand element is read once per tile.  dtype FIRA_BF16: operands rounded to bf16 (RNE), fp32
 * accumulation; FIRA_F32: every operand split into three bf16 terms, six exact term products per fp32 product -- fp32-accurate
 * (error of the order of 2^-24 of |a||b| per term, tests/test_ops_gpu.py) at 2.7x the fp32 MFMA rate.  Shapes: M >= 32,
 * N a multiple of 256, lda / ldb even, A / B 8-byte aligned, K * ld * 4 < 2 GiB.  C is accumulated into (no atomics: a
 * workgroup owns its tile).  scratch (optional, scratch_floats >= 65 536 per workgroup of a split product): lets K be split
 * into slabs over the chip -- the slabs' partial tiles go there and a closing launch adds them to C; NULL = one slab. */
int fira_gemm_wgrad_panel(void* stream, int dtype, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                          float* C, int ldc, float* colsum, float* scratch, size_t scratch_floats);

/* (v8) FeedForward block of a decoder layer (gnn_transformer.py:163-174) as one entry (two products + the row kernel, or the
 * LayerNorm-prologue forms, chosen as in the model-level calls):
 *     h = relu(x W1^T + b1) [M,F] ;  sum = dropout(h W2^T + b2) + x ;  y = LayerNorm(sum) ;  stats = {mean, 1/std}
 * w1 [F,256], w2 [256,F] as nn.Linear stores them; dropout element index = row*256 + col of `site`.  Backward: dy = gradient
 * w.r.t. y; dx [M,256] is written (must not alias dy); dw1 / db1 / dw2 / db2 / dgamma / dbeta are accumulated into; dyf_ws
 * [M,256] and dh_ws [M,F] are scratch.  dtype FIRA_F32 | FIRA_BF16 (operands of the products rounded, fp32 accumulate). */
int fira_ffn_fwd(void* stream, int M, int F, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                 const float* gamma, const float* beta, float* h, float* sum, float* y, float* stats, float dropout,
                 uint64_t seed, uint32_t site, int dtype);
int fira_ffn_bwd(void* stream, int M, int F, const float* dy, const float* x, const float* h, const float* sum,
                 const float* stats, const float* w1, const float* w2, const float* gamma, float* dx, float* dyf_ws,
                 float* dh_ws, float* dw1, float* db1, float* dw2, float* db2, float* dgamma, float* dbeta, float dropout,
                 uint64_t seed, uint32_t site, int dtype);
/* (v8) Output head + arg-top-k (Model.py:54,85; the candidate ranking of run_model.py:305 restricted to the generator):
 * logits_ws [R, ldl >= V] = x [R,256] Wout^T + bout; ids / vals [R, k] = the k <= 8 largest logits of every row, value
 * descending, ties by ascending id (the order fira_beam_select uses). */
int fira_head_topk(void* stream, int R, int V, int k, const float* x, const float* wout, const float* bout, float* logits_ws,
                   int ldl, int32_t* ids, float* vals, int dtype);

/* (v9) The generator projection alone, logits[R, V] = x[R, 256] W[V, 256]^T + bias (Model.py:54 / :85, out_fc), on the bf16
 * matrix cores at fp32 accuracy: both operands as three bf16 terms, six of the nine term products, fp32 accumulation
 * (csrc/head_x3.hip) -- what the model-level calls run in fp32 mode.  x rows at stride ldx, logits rows at stride ldl (both
 * multiples of 4 floats, 16-byte aligned); scratch: fira_head_logits_x3_scratch_bytes(R) bytes (the planes of x).            */
size_t fira_head_logits_x3_scratch_bytes(int R);
int fira_head_logits_x3(void* stream, int R, int V, const float* x, int ldx, const float* W, const float* bias, float* logits,
                        int ldl, void* scratch);


/* (v7) One optimisation step in one call: `loss.backward(); optimizer.step()` of run_model.py:104-111 for a single device --
 * fira_train_fwd_bwd followed by fira_adam_step_mb over [0, live) with the token normaliser of run_model.py:105, bit for bit
 * (Adam is element-wise).  What the single call adds is the ORDER: the update of the head + decoder slice [0, split) is
 * enqueued as soon as the caller's stream has finished the encoder's backward chain, beside the last weight gradients of the
 * library's other streams; [split, live) follows their join.  params is updated in place, m / v are the Adam moments
 * ([>= live] floats each), step counts from 1.  Data-parallel runs keep the two-call form (the all-reduce sits between). */
typedef struct fira_adam_opts {
    float   lr, beta1, beta2, eps;
    int32_t step;
    float*  m;
    float*  v;
} fira_adam_opts;
int fira_train_step(void* stream, const fira_dims* d, const fira_batch* batch, float* params, float* grads,
                    void* workspace, size_t workspace_bytes, const fira_train_opts* opts, float* loss_sum,
                    int32_t* n_tok, const fira_adam_opts* adam);

/* (v8) The same step for DATA-PARALLEL runs (replaces nn.DataParallel's scatter / replicate / gather of run_model.py:392-394
 * around run_model.py:104-111), as two calls with the caller's collectives in between:
 *   fira_train_step_begin  forward + backward of the output head and the decoder; records mid_event (as fira_train_fwd_bwd
 *                          does) when every gradient of [0, split) is final.  The step stays pending on the calling thread;
 *                          the batch's arrays, the parameter / gradient buffers and the workspace must stay untouched.
 *   -- caller: all-reduce (loss_sum, n_tok) and grads[0, split) on its own stream behind mid_event, record early_event there --
 *   fira_train_step_end    backward of the encoder; Adam of [0, split) on the caller's stream as soon as that stream has
 *                          passed the encoder's chain AND early_event (beside the library's last weight gradients), scaled by
 *                          1 / max(*count, 1) (count: device float = the all-reduced token count; NULL = this rank's n_tok);
 *                          then the join: grads[split, live) are final on `stream` when the call returns.  adam NULL = no update.
 *   -- caller: all-reduce grads[split, live), then fira_adam_step_count on that slice --
 * One process per GPU, one thread per step: the pending step is thread-local.  Same arithmetic as fira_train_fwd_bwd +
 * fira_adam_step_count (tests/test_dp_gpu.py: two ranks == one process). */
int fira_train_step_begin(void* stream, const fira_dims* d, const fira_batch* batch, const float* params, float* grads,
                          void* workspace, size_t workspace_bytes, const fira_train_opts* opts, float* loss_sum,
                          int32_t* n_tok, void* mid_event /* hipEvent_t, required */);
int fira_train_step_end(void* stream, float* params, const fira_adam_opts* adam, void* early_event /* hipEvent_t or NULL */,
                        const float* count);

/* (v10) fira_train_step with a ROW-SPARSE Adam of the two vocabulary-sized embedding tables (decoder.embedding.weight,
 * encoder.embedding.weight: 2 x 6.3 M of the 27.8 M parameters; run_model.py:396's torch.optim.Adam updates every row of both
 * every step).  An embedding row whose gradient row is all zero still moves under Adam -- its moments decay and the parameter
 * follows -- but that update is a function of the row's (p, m, v) and the step number alone, so it is applied LATER, in
 * registers, bit for bit: ahead of a forward pass for the rows that pass gathers (the batch's token ids), or together with the
 * next step whose gradient row is not zero.  row_step [2 * vocab] int32 (zero-initialised with the moments; decoder table
 * first) holds the step up to which each row of params / m / v is current.  Every 32nd step updates every row, so a row lags
 * at most 31 steps.  Results are those of fira_train_step bit for bit ONCE fira_adam_rows_sync has run: call it (same adam
 * values, step = the last completed step) before anything else reads the tables or the moments -- checkpoint, dev pass,
 * search, another optimizer -- and before changing lr / beta / eps.  adam->step must advance by one per call.  Needs
 * opts->zero_grads = 1.  (Data parallel: fira_train_step_begin_rows / _end_rows below.)                                      */
int fira_train_step_rows(void* stream, const fira_dims* d, const fira_batch* batch, float* params, float* grads,
                         void* workspace, size_t workspace_bytes, const fira_train_opts* opts, float* loss_sum,
                         int32_t* n_tok, const fira_adam_opts* adam, int32_t* row_step);
int fira_adam_rows_sync(void* stream, const fira_dims* d, float* params, const fira_adam_opts* adam, int32_t* row_step);
/* The two pieces fira_train_step_rows is made of, as op-level entries (params / grads / m / v: the flat buffers of geometry d):
 * fira_adam_rows_catchup  rows ids[0 .. n_ids) of table `table` (0 decoder.embedding, 1 encoder.embedding; duplicates and
 *                         out-of-range ids allowed) brought up to step adam->step by zero-gradient updates;
 * fira_adam_rows_step     step adam->step on every row of the selected tables whose gradient row is not all zero (every row
 *                         when step % 32 == 0), normaliser 1 / max(*n_tok, 1) as fira_adam_step_mb or 1 / max(*count, 1).   */
int fira_adam_rows_catchup(void* stream, const fira_dims* d, float* params, const fira_adam_opts* adam, int32_t* row_step,
                           int table, const int32_t* ids, int n_ids);
int fira_adam_rows_step(void* stream, const fira_dims* d, float* params, const float* grads, const fira_adam_opts* adam,
                        int32_t* row_step, const int32_t* n_tok, const float* count /* NULL, or the normaliser 1 / max(*count, 1)
                        of fira_adam_step_count */, int tables /* bit t: table t takes part; 3 = both */);
/* The data-parallel step (fira_train_step_begin / _end) with the same update.  The union of the ranks' touched rows is what
 * the ALL-REDUCED gradient shows as non-zero rows, the same on every rank, so the replicas (tables, moments, row_step) stay
 * identical.  _begin_rows takes the optimizer's values for the lazy reads of its forward pass (nothing is updated there);
 * _end_rows updates [0, split) as fira_train_step_end does, decoder.embedding by rows; encoder.embedding is the caller's:
 * fira_adam_rows_step(tables = 2, count) + fira_adam_step_count on the rest of [split, live) behind the late bucket.          */
int fira_train_step_begin_rows(void* stream, const fira_dims* d, const fira_batch* batch, const float* params, float* grads,
                               void* workspace, size_t workspace_bytes, const fira_train_opts* opts, float* loss_sum,
                               int32_t* n_tok, void* mid_event, const fira_adam_opts* adam, int32_t* row_step);
int fira_train_step_end_rows(void* stream, float* params, const fira_adam_opts* adam, void* early_event, const float* count,
                             int32_t* row_step);

/* (v8) bf16 wire format of a gradient bucket (BASELINE configs[2]: 55.6 MB instead of 111.2 MB per step on xGMI):
 * out[i] = bf16(in[i]) (round to nearest even) / out[i] = float(in[i]).  n % 4 == 0, 16-byte aligned buffers. */
int fira_f32_to_bf16(void* stream, int64_t n, const float* in, uint16_t* out);
int fira_bf16_to_f32(void* stream, int64_t n, const uint16_t* in, float* out);

/* TransModel.forward(..., 'dev') (Model.py:85-86): teacher-forced argmax ids [B, tar_len].         */
int fira_forward_dev(void* stream, const fira_dims* d, const fira_batch* batch, const float* params,
                     void* workspace, size_t workspace_bytes, int32_t* ids_out, float* loss_sum, int32_t* n_tok,
                     int dtype /* FIRA_F32 | FIRA_BF16 */);

/* Encoder once per batch (run_model.py:202-207) + everything of the decode loop that does not depend
 * on the generated prefix: memory [B,S,256], mem_valid [B,S], cross-attention K/V of all layers,
 * LinearSource(memory).  State lives in the caller's workspace.                                     */
int fira_decode_begin(void* stream, const fira_dims* d, const fira_batch* batch, const float* params,
                      void* workspace, size_t workspace_bytes, int n_beam);
/* One decode step for every (commit, beam) row with KV cache: consumes tokens[B*n_beam] (ids at position
 * `step`), produces dist [B*n_beam, vocab+S] = the reference's `output[:, step, :]` (run_model.py:256-267)
 * if dist != NULL, and/or greedy argmax ids + their probability.  parent[B*n_beam] (may be NULL) re-orders
 * the self-attention cache rows after a beam re-ranking.                                            */
int fira_decode_step(void* stream, const fira_dims* d, const float* params, void* workspace,
                     size_t workspace_bytes, int B, int n_beam, int step, const int32_t* tokens,
                     const int32_t* parent, float* dist, int32_t* best_id, float* best_p);
/* The same two calls with option flags (both calls of one search must carry the same flags):
 *   FIRA_DECODE_KV_BF16  keep a bf16 copy of the cross-attention K|V rows of all layers and stream THAT in every step
 *                        (the cross K|V are 315 of the 379 MB a step moves at batch 64); the arithmetic stays fp32 on the
 *                        widened values.  Not the default: with it the searched ids are no longer bit-identical to the
 *                        reference's fp32 run (bench.py reports the agreement). */
#define FIRA_DECODE_KV_BF16 1
int fira_decode_begin_ex(void* stream, const fira_dims* d, const fira_batch* batch, const float* params,
                         void* workspace, size_t workspace_bytes, int n_beam, int flags);
int fira_decode_step_ex(void* stream, const fira_dims* d, const float* params, void* workspace,
                        size_t workspace_bytes, int B, int n_beam, int step, const int32_t* tokens,
                        const int32_t* parent, float* dist, int32_t* best_id, float* best_p, int flags);

/* Hypothesis bookkeeping of the search loop on the device (run_model.py:225-246 and :268-340).  State per commit:
 * gen [B*n_beam, tar_len] ids starting with <start>, length [B*n_beam], prob [B*n_beam] (slot 0 starts at 1, the others
 * at 0), double-buffered by the caller.  `active` has 9 ints (flags of up to 8 slots + their count), `done` 1 int that
 * latches once no slot runs; after that fira_beam_select only copies the state through.
 *   fira_beam_prepare : finished[r] = last token is <eos>; active[j] = some commit unfinished in slot j;
 *                       tokens[r] = id at position `step` (0 once the hypothesis is shorter) -> fira_decode_step
 *   fira_beam_select  : the n_beam best of { dist[slot j] * prob[j] (-1 if finished) for running slots, in slot order }
 *                       ++ { prob of finished hypotheses, slot order, -1 padded }, ranked by (value desc, index asc);
 *                       copy ids resolved through sou / sub_token; writes the new state and parent[] for the KV cache */
int fira_beam_prepare(void* stream, int B, int n_beam, int tar_len, int step, const int32_t* gen, const int32_t* length,
                      int32_t* tokens, int32_t* finished, int32_t* active, int32_t* done);
int fira_beam_select(void* stream, const fira_dims* d, int B, int n_beam, const float* dist, const int32_t* finished,
                     const int32_t* active, const int32_t* done, const int32_t* sou, const int32_t* sub_token,
                     const int32_t* gen_in, const int32_t* len_in, const float* prob_in, int32_t* gen_out,
                     int32_t* len_out, float* prob_out, int32_t* parent);

/* Greedy search bookkeeping (the same loop at beam 1 without the [B, vocab+S] distribution): consumes the arg-max
 * index / probability fira_decode_step wrote, resolves copy indices through sou / sub_token, appends the id at
 * out[b, step+1], multiplies prob, bumps length, clears alive[b] at <eos>, writes the next step's input ids to `tokens`
 * (0 for ended hypotheses) and adds the number of still-running hypotheses to n_alive[step] (caller zeroes n_alive
 * [tar_len] before the first step).  run_model.py:305-340 with beam_size 1.                                       */
int fira_greedy_advance(void* stream, const fira_dims* d, int B, int step, const int32_t* best_id, const float* best_p,
                        const int32_t* sou, const int32_t* sub_token, int32_t* out, int32_t* length, float* prob,
                        int32_t* alive, int32_t* tokens, int32_t* n_alive);

/* Decoder.forward over all tar_len positions on caller-supplied memory [B, sou+sub, 256] / mem_valid [B, sou+sub]
 * (gnn_transformer.py:108-122; the call of run_model.py:256).  Workspace: fira_workspace_bytes(d, B, 0).          */
int fira_decoder_forward(void* stream, const fira_dims* d, const float* params, void* workspace,
                         size_t workspace_bytes, int B, const int32_t* tar, const float* memory,
                         const int32_t* mem_valid, float* out);

/* read-only views into a decode workspace (device pointers), for tests and the Python driver */
const float*   fira_decode_memory(const fira_dims* d, void* workspace, int B, int n_beam);
const int32_t* fira_decode_mem_valid(const fira_dims* d, void* workspace, int B, int n_beam);

/* ---- host-side helper (CPU only, no device work) -----------------------------------------------------------------
 * The index lists of fira_batch from one collated batch (int64 id arrays as the reference's Dataset.py produces them,
 * block-diagonal CSR with GLOBAL node ids b*N + local): computed nodes + compact CSR, code / memory rows, word-id
 * groups (items of at most `chunk` positions) and AST / edit (row, id) pairs.  Output arrays are caller-allocated at their
 * worst-case sizes: node_rows [B*N], rowptr_c [B*N+1], col_c / val_c [nnz], code_rows / code_mark [B*L],
 * mem_rows / mem_dst [B*(L+S)], item_tok [B*(L+S)], item_ptr [B*(L+S)+1], emb_rows [B*(L+S)], ast_rows / ast_ids
 * [B*(N-L-S)]; counts[7] = {n_nodes, nnz_c, n_code, n_mem, n_items, n_emb_rows, n_ast}.
 * Same result, bit for bit, as fira_icse_amd/model.py: computed_nodes + compact_embedding_lists (the specification).    */
int fira_host_node_lists(int B, int N, int L, int S, int skip_padding, const int64_t* sou, const int64_t* sub_token,
                         const int64_t* ast_change, const int64_t* mark, const int32_t* rowptr, const int32_t* col,
                         const float* val, int chunk, int32_t* node_rows, int32_t* rowptr_c, int32_t* col_c,
                         float* val_c, int32_t* code_rows, int32_t* code_mark, int32_t* mem_rows, int32_t* mem_dst,
                         int32_t* item_tok, int32_t* item_ptr, int32_t* emb_rows, int32_t* ast_rows, int32_t* ast_ids,
                         int32_t* counts);

/* Block-diagonal CSR of a batch from a store of per-commit CSR graphs (replaces Dataset.py:336-343's dense collate):
 * store_rowptr [n_commits, N+1], store_offset [n_commits+1] (prefix sums of the commits' nnz), store_col / store_val the
 * concatenated entries; rowptr [B*N+1], col (global node ids b*N + local) and val [sum nnz] are caller-allocated.   */
int fira_host_collate_csr(int B, int N, const int64_t* idx, int64_t n_commits, const int32_t* store_rowptr,
                          const int64_t* store_offset, const int32_t* store_col, const float* store_val,
                          int32_t* rowptr, int32_t* col, float* val);

#ifdef __cplusplus
}
#endif
#endif /* FIRA_HIP_H */
