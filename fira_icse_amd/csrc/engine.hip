// Model-level entry points: chain the hand-written kernels into TransModel.forward / backward
// (reference Model.py:38-86, gnn_transformer.py:45-62,108-122) and the decode step of run_model.py:225-274.
// One host call enqueues the whole step on the caller's stream; nothing here synchronises or allocates.
//
// HBM layout of a training workspace (all fp32 unless noted; NB = B*650 node rows, CB = B*210 code rows,
// MB = B*370 memory rows, TB = B*30 target rows):
//   node buffers  X[l] [NB,256], l = 0..6   layer inputs; X[l] code rows are updated in place by the Combination
//   per enc layer Xc [CB,256] (code rows before the update), qk [CB,512], c [CB,256], s1 [CB,256] + stats,
//                 Z [NB,256], s2 [NB,256] + stats
//   decoder       kv_all [Mc <= MB, 6*512 + pad] (cross-attention K|V of the COMPUTED memory rows, commit b's rows at
//                 mem_off[b] .. mem_off[b+1]), per layer qkv [TB,768], ao, s_a, x_a, qc, ao2, s_c, x_c, h [TB,1024], s_f, x_f
//   head          mem [MB,256], src [MB,256], tgt [TB,256], score [TB,370], gate [TB,2], dec_c [R,256], logits [R, ldl]
//   backward      gradient temporaries of the same shapes (ping-pong node buffers, dkv_all, ...)
#include "engine.h"
#include <hip/hip_ext.h>
#include "epilogue.h"
#include <stdlib.h>
#include <map>
#include <mutex>

namespace fira {

static int64_t shadow_wbt_elems(const Layout& L);      // elements of the packed transposed bf16 shadows (shadow_table)

struct Arena {
    char* base;
    size_t used = 0;
    explicit Arena(void* b) : base((char*)b) {}
    template <typename T>
    T* get(size_t n) {
        size_t bytes = (n * sizeof(T) + 255) / 256 * 256;
        T* p = base ? (T*)(base + used) : nullptr;
        used += bytes;
        return p;
    }
    float* f(size_t n) { return get<float>(n); }
};

struct EncSave {
    float *Xc, *qk, *c, *s1, *st1, *Z, *s2, *st2;
};
struct DecSave {
    float *qkv, *ao, *s_a, *st_a, *x_a, *qc, *ao2, *s_c, *st_c, *x_c, *h, *s_f, *st_f, *x_f;
};

struct EncGrad {
    float *dY2, *dYc, *dqk;
};
struct DecGrad {
    float *dYf, *dh, *dYc, *dq, *dYs, *dqkv;
};


// Row pitch of the cross-attention K|V rows (kv_all / dkv_all: [computed memory rows, 6 * 512] fp32).  The attention kernels read
// ONE 128-byte head slice per key row; at the natural pitch of 12 288 bytes = 48 x 256 the slices of a whole launch land
// on 8 of the 128 memory channels (gcd(48, 128) = 16).  Padding the row by 64 floats makes the pitch 49 x 256 bytes: every
// channel in turn.  FIRA_KV_PAD overrides the pad (floats, multiple of 4; 0 = the natural pitch) for A/B runs.
static int kv_row_pad() {
    static const int pad = [] { const char* e = getenv("FIRA_KV_PAD"); const int v = e ? atoi(e) : 64; return v < 0 ? 0 : (v + 3) / 4 * 4; }();
    return pad;
}

struct Plan {
    int B, NB, CB, MB, TB, N, L, S, A, T, V, ldl, nl, F;
    int kvp;                        // row pitch (floats) of kv_all / dkv_all: nl * 512 + kv_row_pad()
    float *pos_code, *pos_tar;
    int32_t *mem_valid, *tar_valid, *compact_row, *iota, *code_slot, *mem_slot, *row_bt, *rows_c;
    int32_t *mem_off, *mem_valid_c;      // ragged cross-attention keys: commit b's computed memory rows, their key mask
    std::vector<float*> X;          // nl + 1 node buffers
    std::vector<EncSave> enc;
    std::vector<DecSave> dec;
    std::vector<EncGrad> encg;      // per-layer weight-gradient operands: never reused inside a step, so the
    std::vector<DecGrad> decg;      // wgrad GEMMs can run on the side stream while the dgrad chain continues
    float *vtab_all, *H, *mem, *mem_c, *src_c, *x0, *kv_all, *src, *tgt, *score, *gate, *dec_c, *logits;
    float *W21, *c21, *rsum;        // GCN: per layer fc2.weight . fc1.weight [256,256] and fc2.weight . fc1.bias [256]; A_hat 1
    float *W21t;                    // the folded weights transposed = k-major for U W21^T: what the fused GCN forward streams
    uint16_t *W21x = nullptr, *W21tx = nullptr;   // (round 6) three bf16 planes of W21 / W21^T per layer, fragment order (gcn_fused.hip: X3)
    uint16_t *WcX = nullptr, *WcTX = nullptr;     // the same of the Combination weights Wq | Wk | Wo per layer, as stored / transposed
    uint16_t* WkvX = nullptr;                     // ... and of the stacked cross-attention K|V weight: nl * 2 row blocks of [256, 256]
    float* WkvT = nullptr;                        // k-major copies of those blocks and their planes (the d-memory products: linear_x3_kacc)
    uint16_t* WkvTX = nullptr;
    uint16_t* WoutTX = nullptr;                   // planes of the generator projection's transposed 256-row blocks (its data gradient)
    uint16_t *xh_planes = nullptr;                // planes of the head's input rows (head_x3.hip)
    float *WcT;                     // per layer Wq^T | Wk^T | Wo^T of the Combination block, k-major (comb_fused.hip)
    float *dW21, *dc21;             // their gradients (training), one contiguous block zeroed per step
    float *inv_ntok;
    // backward temporaries (training only)
    float *dXa, *dXb, *dNB2, *dCB_a, *dvtab_all;
    float *zero_beg = nullptr, *zero_end = nullptr;
    uint16_t *wb = nullptr, *wbt = nullptr;      // bf16 shadows of the 2-D weights, as stored / transposed (bf16 mode)
    uint16_t *w21b = nullptr, *w21bt = nullptr;  // the same for the folded GCN weights W21 (formed per step)
    float* red_buf = nullptr;                    // per-workgroup partial rows of the deferred column reductions
    size_t red_cap = 0;
    float* panel_scratch = nullptr;              // partial tiles of split panel weight gradients (training)
    size_t panel_floats = 0;
    float *dmem_c, *dsrc, *dsrc_c, *dtgt, *dkv_all, *ddec, *ddec_c, *dT_a, *dT_c;

    size_t build(void* ws, const fira_dims& d, int B_, bool training) {
        Arena a(ws);
        B = B_; L = d.sou_len; S = d.sub_len; A = d.ast_len; T = d.tar_len; V = d.vocab; nl = d.n_layer; F = d.d_ff;
        N = L + S + A;
        NB = B * N; CB = B * L; MB = B * (L + S); TB = B * T;
        ldl = (V + 63) / 64 * 64;
        const size_t D = FIRA_D;
        pos_code = a.f((size_t)L * D);
        pos_tar = a.f((size_t)T * D);
        mem_valid = a.get<int32_t>((size_t)MB);
        tar_valid = a.get<int32_t>((size_t)TB);
        compact_row = a.get<int32_t>((size_t)TB);
        iota = a.get<int32_t>((size_t)TB);
        row_bt = a.get<int32_t>((size_t)TB);          // computed target rows (fira_batch.dec_off): compact row -> flat b*T + t
        rows_c = a.get<int32_t>((size_t)TB);          // the head-row list in compact-row indexing
        code_slot = a.get<int32_t>((size_t)NB);
        mem_slot = a.get<int32_t>((size_t)NB);
        mem_off = a.get<int32_t>((size_t)B + 1);
        mem_valid_c = a.get<int32_t>((size_t)MB);
        inv_ntok = a.f(64);
        {
            const Layout* lay = get_layout(&d);
            const size_t tot = lay ? (size_t)lay->total : 0;
            wb = a.get<uint16_t>(tot);
            wbt = a.get<uint16_t>((lay ? (size_t)shadow_wbt_elems(*lay) : 0) + 64);   // + slack: an edge chunk may be read whole behind the last copy
            w21b = a.get<uint16_t>((size_t)nl * D * D);
            w21bt = a.get<uint16_t>((size_t)nl * D * D + 64);
        }
        X.resize(nl + 1);
        for (int l = 0; l <= nl; ++l) X[l] = a.f((size_t)NB * D);
        enc.resize(nl);
        for (int l = 0; l < nl; ++l) {
            EncSave& e = enc[l];
            e.Xc = a.f((size_t)CB * D); e.qk = a.f((size_t)CB * 2 * D); e.c = a.f((size_t)CB * D);
            e.s1 = a.f((size_t)CB * D); e.st1 = a.f((size_t)CB * 2);
            e.Z = a.f((size_t)NB * D); e.s2 = a.f((size_t)NB * D); e.st2 = a.f((size_t)NB * 2);
        }
        vtab_all = a.f((size_t)4 * nl * D);
        W21 = a.f((size_t)nl * D * D); c21 = a.f((size_t)nl * D); rsum = a.f((size_t)NB);
        W21t = a.f((size_t)nl * D * D);
        W21x = a.get<uint16_t>((size_t)nl * 3 * D * D);
        W21tx = a.get<uint16_t>((size_t)nl * 3 * D * D);
        WcX = a.get<uint16_t>((size_t)nl * 9 * D * D);
        WcTX = a.get<uint16_t>((size_t)nl * 9 * D * D);
        WkvX = a.get<uint16_t>((size_t)nl * 6 * D * D);
        WkvT = a.f((size_t)nl * 2 * D * D);
        WkvTX = a.get<uint16_t>((size_t)nl * 6 * D * D);
        WoutTX = a.get<uint16_t>((size_t)cdiv(V, D) * 3 * D * D);
        xh_planes = a.get<uint16_t>(head_logits_x3_scratch_elems(TB));
        WcT = a.f((size_t)nl * 3 * D * D);
        H = a.f((size_t)NB * D);
        mem = a.f((size_t)MB * D);
        mem_c = a.f((size_t)MB * D);
        src_c = a.f((size_t)MB * D);
        x0 = a.f((size_t)TB * D);
        kvp = nl * 2 * D + kv_row_pad();
        kv_all = a.f((size_t)MB * kvp);
        dec.resize(nl);
        for (int l = 0; l < nl; ++l) {
            DecSave& e = dec[l];
            e.qkv = a.f((size_t)TB * 3 * D); e.ao = a.f((size_t)TB * D); e.s_a = a.f((size_t)TB * D);
            e.st_a = a.f((size_t)TB * 2); e.x_a = a.f((size_t)TB * D); e.qc = a.f((size_t)TB * D);
            e.ao2 = a.f((size_t)TB * D); e.s_c = a.f((size_t)TB * D); e.st_c = a.f((size_t)TB * 2);
            e.x_c = a.f((size_t)TB * D); e.h = a.f((size_t)TB * F); e.s_f = a.f((size_t)TB * D);
            e.st_f = a.f((size_t)TB * 2); e.x_f = a.f((size_t)TB * D);
        }
        src = a.f((size_t)MB * D); tgt = a.f((size_t)TB * D);
        score = a.f((size_t)TB * (L + S)); gate = a.f((size_t)TB * 2);
        dec_c = a.f((size_t)TB * D); logits = a.f((size_t)TB * ldl);
        if (training) {
            dXa = a.f((size_t)NB * D); dXb = a.f((size_t)NB * D); dNB2 = a.f((size_t)NB * D);
            dCB_a = a.f((size_t)CB * D);
            red_cap = (size_t)nl * ((size_t)cdiv(NB, 16) * 4 * D + (size_t)cdiv(CB, 16) * 2 * D)                 // encoder LNs (GCN: + 2 sums)
                    + (size_t)3 * nl * cdiv(TB, 16) * 2 * D                                                      // decoder LNs
                    + (size_t)nl * std::max((size_t)cdiv(CB, 16) * 4 * D, (size_t)comb_fused_bwd_parts() * 6 * D + 64)  // Combination (fused: one {LN | dvtab} row pair per workgroup)
                    + (size_t)cdiv(L + S, 16) * B * COPY_PART_STRIDE + 4096;                                     // copy head
            red_buf = a.f(red_cap);
            // partial tiles of the panel weight-gradient launches (gemm_wgrad_panel.hip): one 256 x 256 tile per workgroup of a
            // split product, ~one workgroup per CU and launch; used by the launches of the weight-gradient stream, in order
            panel_floats = (size_t)320 * 256 * 256;
            panel_scratch = a.f(panel_floats);
            // buffers that must start a backward pass at zero, contiguous: ONE fill per step (zero_beg .. zero_end)
            zero_beg = (float*)a.get<char>(0);
            dvtab_all = a.f((size_t)4 * nl * D);
            dW21 = a.f((size_t)nl * D * D + (size_t)nl * D); dc21 = dW21 + (size_t)nl * D * D;
            dtgt = a.f((size_t)TB * D);
            ddec_c = a.f((size_t)TB * D);
            zero_end = (float*)a.get<char>(0);
            dmem_c = a.f((size_t)MB * D); dsrc = a.f((size_t)MB * D); dsrc_c = a.f((size_t)MB * D);
            dkv_all = a.f((size_t)MB * kvp);
            ddec = a.f((size_t)TB * D);
            dT_a = a.f((size_t)TB * D); dT_c = a.f((size_t)TB * D);
            encg.resize(nl);
            decg.resize(nl);
            for (int l = 0; l < nl; ++l) {
                EncGrad& g = encg[l];
                g.dY2 = a.f((size_t)NB * D); g.dYc = a.f((size_t)CB * D);
                g.dqk = a.f((size_t)CB * 2 * D);
                DecGrad& h = decg[l];
                h.dYf = a.f((size_t)TB * D); h.dh = a.f((size_t)TB * F); h.dYc = a.f((size_t)TB * D);
                h.dq = a.f((size_t)TB * D); h.dYs = a.f((size_t)TB * D); h.dqkv = a.f((size_t)TB * 3 * D);
            }
        }
        return a.used;
    }
};

#define TRY(x)                 \
    do {                       \
        if (int e__ = (x)) return e__; \
    } while (0)

static int zero(hipStream_t s, void* p, size_t bytes) {
    hipError_t e = hipMemsetAsync(p, 0, bytes, s);
    if (e != hipSuccess) return set_err("hipMemsetAsync: %s", hipGetErrorString(e));
    return 0;
}

// Compute dtype of the model-level entry points (fira_train_opts.dtype / the dtype argument of fira_forward_dev):
// 0 = fp32 MFMA (the reference's arithmetic), 1 = bf16 MFMA with fp32 accumulation and fp32 storage (BASELINE
// configs[2]).  Only the nn.Linear products switch; LayerNorm, soft-max, the gate, the loss and Adam stay fp32, and
// so do the products on parameters alone (the folded GCN weights) and the tiny ones gemm_bf16_ex forwards.
static thread_local int g_dtype = 0;
static thread_local int g_lanes = 1;            // commit-lanes of the running decoder pass (decoder_lanes)
struct DtypeScope {
    int prev;
    explicit DtypeScope(int d) : prev(g_dtype) { g_dtype = d; }
    ~DtypeScope() { g_dtype = prev; }
};
// bf16 weight shadows of the current call (workspace-resident, refreshed by one launch at the start of the call):
// [g_P, g_P + g_total) is the fp32 parameter buffer they mirror, at identical offsets
static thread_local const float* g_P = nullptr;
static thread_local int64_t g_total = 0;
static thread_local const uint16_t* g_Wb = nullptr;
static thread_local const uint16_t* g_WbT = nullptr;
static thread_local const ShadowTable* g_tab = nullptr;
static thread_local const float* g_W21 = nullptr;          // folded GCN weights [nl, 256, 256] (workspace) and their shadows
static thread_local int64_t g_W21n = 0;
static thread_local const uint16_t* g_W21b = nullptr;
static thread_local const uint16_t* g_W21bT = nullptr;

// every 2-D weight a GEMM of the training / dev path reads, in the shapes the engine multiplies them in
static const ShadowTable* shadow_table(const Layout& L) {
    static std::mutex mu;
    static std::map<const Layout*, ShadowTable*> cache;
    std::lock_guard<std::mutex> g(mu);
    auto it = cache.find(&L);
    if (it != cache.end()) return it->second;
    ShadowTable* t = new ShadowTable();
    const int D = FIRA_D;
    auto add = [&](int64_t off, int rows, int cols) {
        if (t->n == SHADOW_MAX) return;
        // transposed shadow [cols, rows]: rows padded to a multiple of 8 elements so that every row starts 16-byte
        // aligned (the vocabulary projection: 24650 -> 24656).  The padded copy is larger than the tensor, so the
        // transposed copies are packed at their own offsets (64-element aligned), whatever the vocabulary size is
        const int pitch = (rows + 7) / 8 * 8;
        t->e[t->n] = ShadowEntry{off, rows, cols, pitch, t->wbt_elems};
        t->wbt_elems += ((int64_t)cols * pitch + 63) / 64 * 64;
        t->tile_start[t->n + 1] = t->tile_start[t->n] + cdiv(rows, 64) * cdiv(cols, 64);
        ++t->n;
    };
    for (int l = 0; l < L.d.n_layer; ++l) {
        add(L.enc[l].wqk, 2 * D, D);
        add(L.enc[l].wo, D, D);
        add(L.dec[l].wqkv, 3 * D, D);
        add(L.dec[l].wo_s, D, D);
        add(L.dec[l].wq_c, D, D);
        add(L.dec[l].wo_c, D, D);
        add(L.dec[l].w1, L.d.d_ff, D);
        add(L.dec[l].w2, D, L.d.d_ff);
    }
    add(L.wkv_all, L.d.n_layer * 2 * D, D);
    add(L.wout, L.d.vocab, D);
    add(L.ws, D, D);
    add(L.wt, D, D);
    cache[&L] = t;
    return t;
}
static int64_t shadow_wbt_elems(const Layout& L) { return shadow_table(L)->wbt_elems; }
struct ShadowScope {            // publishes / withdraws the shadows of the running call
    ShadowScope(const float* P, int64_t total, const uint16_t* wb, const uint16_t* wbt, const ShadowTable* tab) {
        g_P = P; g_total = total; g_Wb = wb; g_WbT = wbt; g_tab = tab;
    }
    ~ShadowScope() {
        g_P = nullptr; g_total = 0; g_Wb = nullptr; g_WbT = nullptr; g_tab = nullptr;
        g_W21 = nullptr; g_W21n = 0; g_W21b = nullptr; g_W21bT = nullptr;
    }
};
// shadow of the weight (or contiguous row slice of a weight) starting at W: as stored, or transposed (+ its row pitch)
static bool shadow_of(const float* W, bool transposed, const uint16_t** out, int* ld) {
    static const bool off_switch = [] { const char* e = getenv("FIRA_NO_SHADOW"); return e && e[0] == '1'; }();   // A/B
    if (off_switch || !g_Wb || !g_tab) return false;
    if (g_W21b && W >= g_W21 && W < g_W21 + g_W21n) {           // a folded GCN weight: uniform [256,256] blocks
        const int64_t off = W - g_W21;
        if (off % (FIRA_D * FIRA_D)) return false;
        *out = (transposed ? g_W21bT : g_W21b) + off;
        *ld = FIRA_D;
        return true;
    }
    if (W < g_P || W >= g_P + g_total) return false;
    const int64_t off = W - g_P;
    for (int i = 0; i < g_tab->n; ++i) {
        const ShadowEntry& e = g_tab->e[i];
        if (off < e.offset || off >= e.offset + (int64_t)e.rows * e.cols) continue;
        const int64_t rel = off - e.offset;
        if (rel % e.cols) return false;
        if (!transposed) { *out = g_Wb + off; *ld = e.cols; }
        else { *out = g_WbT + e.offset_t + rel / e.cols; *ld = e.pitch_t; }
        return true;
    }
    return false;
}
static inline int gemm_any(hipStream_t s, int tA, int tB, int M, int N, int K, const float* A, int lda, const float* B,
                           int ldb, float* C, int ldc, const float* bias, int flags, int splitk, float* colsum,
                           const int32_t* c_rows = nullptr, const float* relu_mask = nullptr) {
    if (g_dtype == 1) {
        // forward (B = W [N,K]) and data gradient (B = W [K,N], reduced over its rows) read the bf16 shadows: the same
        // kernel with a k-contiguous bf16 B operand, half the weight bytes and no transposing loads
        const uint16_t* wb;
        int ldw;
        if (!tA && !colsum && gemm_bf16_takes(M, N, K) && lda % 4 == 0 && shadow_of(B, !tB, &wb, &ldw) &&
            (tB ? ldb == K : ldb == N) && ldw % 8 == 0 && ((uintptr_t)wb % 16) == 0)
            return gemm_bf16_wb_ex(s, M, N, K, A, lda, wb, ldw, C, ldc, bias, flags, splitk, c_rows, relu_mask);
        return gemm_bf16_ex(s, tA, tB, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, splitk, colsum, c_rows, relu_mask);
    }
    return gemm_f32_ex(s, tA, tB, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, splitk, colsum, c_rows, relu_mask);
}

// Y = X W^T + b
static inline int linear(hipStream_t s, int M, int N, int K, const float* X, int ldx, const float* W, const float* b,
                         float* Y, int ldy, int flags = 0) {
    return gemm_any(s, 0, 1, M, N, K, X, ldx, W, K, Y, ldy, b, flags, 0, nullptr);
}
// y = LN(dropout(X W^T + b [+ r c^T]) + res): the closing step of every block.  By default the product followed by the row
// kernel; one fused launch where that was measured to win -- bf16 mode, K = 256, encoder-sized blocks (the panel kernel's
// LayerNorm epilogue, gemm_bf16_panel.hip; FIRA_FUSED_LN_BF16).  (The fp32 "workgroup owns complete rows" kernels of round 2
// lost in the step and are gone; the decoder's blocks are split at their LayerNorm instead: linear_presum below.)
static inline int linear_ln(hipStream_t s, int M, int K, const float* X, int ldx, const float* W, const float* b,
                            const float* res, const float* gamma, const float* beta, float* sum, float* y, float* stats,
                            float p_drop, uint64_t seed, uint32_t st, const int32_t* y_rows = nullptr,
                            const float* r1_row = nullptr, const float* r1_col = nullptr,
                            // second, compact copy of the listed output rows: y2[k] = y[rows2[k]] for k < n2, slot2 = the
                            // inverse list (row -> k or -1); written by the row kernel itself, or by a gather after a fused launch
                            const int32_t* slot2 = nullptr, float* y2 = nullptr, int n2 = 0, const int32_t* rows2 = nullptr,
                            uint32_t idx0 = 0) {     // dropout element index of row 0 (the rows are a slice of the site's rows)
    bool fused = false;
    int rc = 0;
    if (g_dtype != 0 && idx0 == 0 && g_lanes == 1) {   // bf16: the panel kernel with a LayerNorm epilogue (gemm_bf16_panel.hip)
        const uint16_t* wb;
        int ldw;
        fused = shadow_of(W, false, &wb, &ldw) && ldw == K &&
                linear_ln_bf16_try(s, M, K, X, ldx, wb, ldw, b, res, gamma, beta, sum, y, stats, p_drop, seed, st, y_rows, r1_row,
                                   r1_col, &rc);
    }
    if (fused) {
        TRY(rc);
        if (y2 && n2 > 0) TRY(rows_move(s, 0, n2, FIRA_D, y2, y, rows2, nullptr));
        return 0;
    }
    TRY(linear(s, M, FIRA_D, K, X, ldx, W, b, sum, FIRA_D));
    return add_layernorm_fwd(s, M, sum, res, gamma, beta, y, stats, p_drop, seed, st, y_rows, r1_row, r1_col, slot2, y2, idx0);
}
// The same block with its LayerNorm moved into the CONSUMER (fp32, coalesced tile kernel): the closing product stores the
// pre-norm sum  sum = dropout(X W^T + b) + res  (EpiRes epilogue) and the next product normalises its A rows itself
// (gemm_tile32_ln_try) -- one launch less per block on the dependent chain.  Both return false when the shape is not
// taken; the caller then runs linear_ln / linear.
static bool presum_on() {
    static const bool off = [] { const char* e = getenv("FIRA_LN_PROLOGUE"); return e && e[0] == '0'; }();     // A/B switch
    return !off && g_dtype == 0;
}
static inline bool linear_presum(hipStream_t s, int M, int K, const float* X, int ldx, const float* W, const float* b,
                                 const float* res, float* sum, float p_drop, uint64_t seed, uint32_t st, int* rc,
                                 uint32_t idx0 = 0) {
    if (!presum_on() || !gemm_tile32_takes(1, M, FIRA_D, K, X, ldx, W, K)) return false;
    EpiRes er;
    er.res = res; er.ldr = FIRA_D; er.p = p_drop; er.inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    er.seed = seed; er.site = st; er.idx0 = idx0;
    ProfScope prof(s, PROF_GEMM, 2.0 * M * FIRA_D * (double)K, 4.0 * ((double)M * K + (double)FIRA_D * K + 2.0 * M * FIRA_D));
    return gemm_tile32_try(s, 1, M, FIRA_D, K, X, ldx, W, K, sum, FIRA_D, b, 0, rc, nullptr, nullptr, nullptr, &er);
}
// dX (+)= dY W          (W stored [N,K]; reduce over N)
static inline int linear_dgrad(hipStream_t s, int M, int N, int K, const float* dY, int lddy, const float* W, float* dX,
                               int lddx, bool accum) {
    return gemm_any(s, 0, 0, M, K, N, dY, lddy, W, K, dX, lddx, nullptr, accum ? FIRA_GEMM_ACCUM : 0, 0, nullptr);
}
// Weight gradients are off the critical path of the backward pass (nothing downstream reads them), so they are
// issued on a second HIP stream: each one waits for the event that marks its operands ready on the main stream and
// runs beside the dgrad chain, filling the CUs the small decoder-side kernels leave idle.  The main stream joins the
// side stream before the mid-event (data-parallel bucket hand-off) and at the end of the backward pass.
struct SideStream {
    hipStream_t stream = nullptr;
    hipStream_t aux = nullptr;   // high priority: the few off-chain products the main stream waits for later
    hipStream_t lane = nullptr;  // second commit-lane of the decoder's chain (round 6; created on first use)
    std::vector<hipEvent_t> events;
    size_t next = 0;
    bool enabled = true;
    bool wide = true;            // FIRA_SIDE_WGRAD_ONLY=1: only the weight gradients use the side stream (A/B switch)
    int init() {
        if (stream) return 0;
        const char* off = getenv("FIRA_NO_WGRAD_OVERLAP");
        enabled = !(off && off[0] == '1');
        const char* narrow = getenv("FIRA_SIDE_WGRAD_ONLY");
        wide = !(narrow && narrow[0] == '1');
        // weight gradients: LOWEST priority -- nothing waits for them before the end of the step (or the mid-event), and
        // the big ones (the vocabulary projection: 138 MB of dlogits at batch 64) otherwise take the CUs from the
        // dependent chain on the caller's stream exactly when it has only small kernels to offer
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;   // no priorities: plain streams
        hipError_t e = hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, least);
        if (e != hipSuccess) return set_err("hipStreamCreate: %s", hipGetErrorString(e));
        gemm_set_pad_stream(stream);
        e = hipStreamCreateWithPriority(&aux, hipStreamNonBlocking, greatest);
        if (e != hipSuccess) return set_err("hipStreamCreateWithPriority: %s", hipGetErrorString(e));
        events.resize(512);
        for (auto& ev : events) {
            e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e != hipSuccess) return set_err("hipEventCreate: %s", hipGetErrorString(e));
        }
        return 0;
    }
    hipEvent_t ev() { hipEvent_t e = events[next]; next = (next + 1) % events.size(); return e; }
};
static SideStream& side() { static thread_local SideStream s; return s; }

// fork: the side stream waits for everything enqueued on `main` so far
static int side_fork(hipStream_t main_s) {
    SideStream& sd = side();
    hipEvent_t e = sd.ev();
    if (hipEventRecord(e, main_s) != hipSuccess || hipStreamWaitEvent(sd.stream, e, 0) != hipSuccess)
        return set_err("side stream fork failed");
    return 0;
}
// FIRA_WAIT_PROBE=1 (measurement aid): every wait of the caller's stream for one of the library's streams is bracketed by two
// timed events on the caller's stream; every 20th training call synchronises and prints, per source line of the wait, the
// mean time the caller's stream stood there (stderr).  Off: no event is created.
struct WaitProbe {
    bool on = false, init = false;
    struct Rec { int line; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    int calls = 0;
    bool enabled() {
        if (!init) { const char* e = getenv("FIRA_WAIT_PROBE"); on = e && e[0] == '1'; init = true; }
        return on;
    }
    hipEvent_t ev() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    void report() {
        (void)hipDeviceSynchronize();
        std::vector<std::pair<int, std::pair<double, int>>> acc;
        for (auto& r : recs) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) ms = 0.f;
            bool found = false;
            for (auto& q : acc) if (q.first == r.line) { q.second.first += ms; q.second.second++; found = true; }
            if (!found) acc.push_back({r.line, {ms, 1}});
            pool.push_back(r.a); pool.push_back(r.b);
        }
        recs.clear();
        fprintf(stderr, "[fira wait probe] over %d calls, us per call the caller's stream spent at each wait (engine.hip line: us):", calls);
        double tot = 0;
        for (auto& q : acc) { fprintf(stderr, "  %d: %.1f", q.first, 1e3 * q.second.first / calls); tot += 1e3 * q.second.first / calls; }
        fprintf(stderr, "  | total %.1f\n", tot);
        calls = 0;
    }
};
static WaitProbe& wait_probe() { static thread_local WaitProbe w; return w; }
static int stream_wait_probed(hipStream_t main_s, hipEvent_t e, int line) {
    WaitProbe& w = wait_probe();
    if (!w.enabled()) return hipStreamWaitEvent(main_s, e, 0) == hipSuccess ? 0 : 1;
    WaitProbe::Rec r{line, w.ev(), w.ev()};
    (void)hipEventRecord(r.a, main_s);
    const bool ok = hipStreamWaitEvent(main_s, e, 0) == hipSuccess;
    (void)hipEventRecord(r.b, main_s);
    w.recs.push_back(r);
    return ok ? 0 : 1;
}

// join: `main` waits for everything enqueued on the side stream so far
static int side_join(hipStream_t main_s, int line = 0) {
    SideStream& sd = side();
    hipEvent_t e = sd.ev();
    if (hipEventRecord(e, sd.stream) != hipSuccess || stream_wait_probed(main_s, e, line))
        return set_err("side stream join failed");
    return 0;
}

static inline bool side_on() { SideStream& sd = side(); return sd.stream && sd.enabled && sd.wide; }
// the auxiliary stream waits for everything enqueued on `main` so far
static int aux_fork(hipStream_t main_s) {
    SideStream& sd = side();
    hipEvent_t e = sd.ev();
    if (hipEventRecord(e, main_s) != hipSuccess || hipStreamWaitEvent(sd.aux, e, 0) != hipSuccess)
        return set_err("aux stream fork failed");
    return 0;
}
// mark: an event at the current tail of the auxiliary stream;  wait: a stream waits for such a mark
static int side_mark(hipEvent_t* out) {
    SideStream& sd = side();
    hipEvent_t e = sd.ev();
    if (hipEventRecord(e, sd.aux) != hipSuccess) return set_err("aux stream mark failed");
    *out = e;
    return 0;
}
static int main_wait(hipStream_t main_s, hipEvent_t e, int line = 0) {
    if (stream_wait_probed(main_s, e, line)) return set_err("stream wait failed");
    return 0;
}

// dW += dY^T X ; db += colsum(dY)      (reduce over the M rows: split-K over rows keeps the chip busy)
// dY and X must stay untouched until the next side_join (per-layer slots of Plan::encg / decg, saved activations).
// (round 6) Weight gradients as panel products on the bf16 matrix cores (gemm_wgrad_panel.hip: 256 x 256 output tiles, every
// operand row read once, no atomics; fp32 mode = three-term bf16 split, fp32-accurate) where the shape allows: dW [N,K] with
// K (the layer's input width) a multiple of 256 and N >= 32.  FIRA_WGRAD_PANEL=0 restores the tiled kernels everywhere;
// FIRA_WGRAD_PANEL_MASK selects the classes (1 encoder groups, 2 decoder / head groups, 4 vocabulary, 8 stacked K|V).
enum { PANEL_ENC = 1, PANEL_DEC = 2, PANEL_VOCAB = 4, PANEL_KV = 8 };
// (measured, profiles/r6_probes.md: fp32 -- every class wins, mask 15; bf16 -- the decoder / head groups (K = the ~1 000 computed
//  target rows) lose to the tiled grouped kernel, whose single-rounding MFMAs are already short: mask 13)
static inline bool panel_class(int cls) {
    static const int forced = [] { const char* e = getenv("FIRA_WGRAD_PANEL_MASK"); return e ? atoi(e) : -1; }();
    const int mask = forced >= 0 ? forced : (g_dtype == 1 ? 13 : 15);
    return gemm_wgrad_panel_on() && (mask & cls);
}
static inline bool panel_takes(int cls, int M, int N, int K, const float* dY, int lddy, const float* X, int ldx) {
    return panel_class(cls) && gemm_wgrad_panel_takes(N, K, M, dY, lddy, X, ldx, K);
}
static inline int panel_np() { return g_dtype == 1 ? 1 : 3; }

static inline int linear_wgrad(hipStream_t s, int M, int N, int K, const float* dY, int lddy, const float* X, int ldx,
                               float* dW, float* db, int panel_cls = 0) {
    SideStream& sd = side();
    hipStream_t ws = s;
    if (sd.stream && sd.enabled) {
        TRY(side_fork(s));
        ws = sd.stream;
    }
    if (panel_cls && gemm_wgrad_panel_pending() == 0 && panel_takes(panel_cls, M, N, K, dY, lddy, X, ldx))
        return gemm_wgrad_panel(ws, panel_np(), N, K, M, dY, lddy, X, ldx, dW, K, db);
    return gemm_any(ws, 1, 0, N, K, M, dY, lddy, X, ldx, dW, K, nullptr, FIRA_GEMM_ACCUM, 0, db);
}

// The same for the small reductions over the B*30 target rows (decoder layers, gate / target projections of the head):
// queued and launched as ONE grouped kernel after the decoder's backward loop (gemm_f32.hip), instead of ~40 launches
// of ~15 us that are mostly fill and drain.  Operands must stay untouched until then (they are per-layer slots).
static inline int flush_grouped_wgrads(hipStream_t s);
// A full queue launches itself at the next add -- on the weight-gradient stream, WITHOUT a fork from the caller's stream, i.e.
// possibly ahead of the kernels that write its operands (deeper models: > 40 queued problems).  Fork + flush first.
static inline int group_make_room(hipStream_t s) {
    if ((g_dtype == 1 ? gemm_bf16_group_full() : gemm_group_full()) || gemm_wgrad_panel_full()) return flush_grouped_wgrads(s);
    return 0;
}
static inline int linear_wgrad_grouped(hipStream_t s, int M, int N, int K, const float* dY, int lddy, const float* X,
                                       int ldx, float* dW, float* db, int max_split = 0) {
    SideStream& sd = side();
    if (!(sd.stream && sd.enabled)) return linear_wgrad(s, M, N, K, dY, lddy, X, ldx, dW, db);
    TRY(group_make_room(s));
    if (panel_takes(PANEL_DEC, M, N, K, dY, lddy, X, ldx)) return gemm_wgrad_panel_add(N, K, M, dY, lddy, X, ldx, dW, K, db);
    if (g_dtype == 1) {
        if (!gemm_bf16_takes(N, K, M)) return linear_wgrad(s, M, N, K, dY, lddy, X, ldx, dW, db);   // e.g. the 2-column gate
        return gemm_bf16_group_add_wgrad(sd.stream, N, K, M, dY, lddy, X, ldx, dW, K, db, max_split);
    }
    return gemm_group_add_wgrad(sd.stream, N, K, M, dY, lddy, X, ldx, dW, K, db, max_split);
}
// One encoder layer's three weight gradients (folded GCN weight, Combination output and q|k projections): queued while
// the layer's data-gradient chain runs, then issued as ONE grouped launch behind ONE fork (enc_wgrads_flush) -- every fork
// is an event record on the caller's stream, i.e. a barrier packet in the dependent chain (scripts/event_cost.py: ~4-5 us
// each), and the layer used to pay four of them.  FIRA_ENC_WGRAD_GROUP=0: one fork + launch per gradient (A/B switch).
static inline bool enc_group_on() {
    static const bool off = [] { const char* e = getenv("FIRA_ENC_WGRAD_GROUP"); return e && e[0] == '0'; }();
    SideStream& sd = side();
    return !off && sd.stream && sd.enabled;
}
static inline int enc_wgrad(hipStream_t s, int M, int N, int K, const float* dY, int lddy, const float* X, int ldx, float* dW,
                            float* db) {
    if (!enc_group_on()) return linear_wgrad(s, M, N, K, dY, lddy, X, ldx, dW, db, PANEL_ENC);
    SideStream& sd = side();
    TRY(group_make_room(s));
    if (panel_takes(PANEL_ENC, M, N, K, dY, lddy, X, ldx)) return gemm_wgrad_panel_add(N, K, M, dY, lddy, X, ldx, dW, K, db);
    if (g_dtype == 1) {
        if (!gemm_bf16_takes(N, K, M)) return linear_wgrad(s, M, N, K, dY, lddy, X, ldx, dW, db);
        return gemm_bf16_group_add_wgrad(sd.stream, N, K, M, dY, lddy, X, ldx, dW, K, db, 32);
    }
    return gemm_group_add_wgrad(sd.stream, N, K, M, dY, lddy, X, ldx, dW, K, db, 32);
}
// dW [N,K] += dY^T X on stream ws, which already waits for its operands: panel product where the shape allows, tiled otherwise
static inline int wgrad_on(hipStream_t ws, int panel_cls, int M, int N, int K, const float* dY, int lddy, const float* X, int ldx,
                           float* dW, float* db) {
    if (gemm_wgrad_panel_pending() == 0 && panel_takes(panel_cls, M, N, K, dY, lddy, X, ldx))
        return gemm_wgrad_panel(ws, panel_np(), N, K, M, dY, lddy, X, ldx, dW, K, db);
    return gemm_any(ws, 1, 0, N, K, M, dY, lddy, X, ldx, dW, K, nullptr, FIRA_GEMM_ACCUM, 0, db);
}
static inline int flush_wgrad_queues(hipStream_t ws) {     // the queued launches on the weight-gradient stream: tiled group, then panels
    TRY(g_dtype == 1 ? gemm_bf16_group_flush(ws) : gemm_group_flush(ws));
    return gemm_wgrad_panel_flush(ws, panel_np());
}
static inline int flush_grouped_wgrads(hipStream_t s) {
    SideStream& sd = side();
    if (!(sd.stream && sd.enabled)) return 0;
    TRY(side_fork(s));
    return flush_wgrad_queues(sd.stream);
}

// Deferred column reductions of the backward pass (rowops.hip: deferred_reduce): kernels park one partial row per
// workgroup in Plan::red_buf; the table is flushed (ONE launch) before the mid-event for the decoder-side parameters
// and at the end of the backward pass for the encoder-side ones.
struct RedCollector {
    float* buf = nullptr;
    size_t cap = 0, used = 0;
    RedTable tab;
    void reset(float* b, size_t c) { buf = b; cap = c; used = 0; tab.n = 0; }
    // nullptr: no room (floats, or the `entries` table rows the caller is going to add()) -> the kernel falls back to atomics
    float* alloc(size_t n, int entries = 4) {
        if (!buf || used + n > cap || tab.n + entries > RED_MAX) return nullptr;
        float* p = buf + used;
        used += (n + 63) / 64 * 64;
        return p;
    }
    void add(float* dst, const float* src, int width, int n_part, int stride) {
        tab.e[tab.n++] = RedEntry{dst, src, width, n_part, stride};
    }
};
static RedCollector& red() { static thread_local RedCollector r; return r; }

// LayerNorm backward with its dgamma / dbeta reduction deferred
static int ln_bwd(hipStream_t s, int M, const float* dy, const float* sum, const float* stats, const float* gamma, float* ds,
                  float* dx_drop, float* dgamma, float* dbeta, float dropout, uint64_t seed, uint32_t st,
                  const int32_t* rows = nullptr,
                  // optional (GCN blocks): the kernel also leaves sum_r dx[r,:] -> dsum and sum_r row_w[r] dx[r,:] -> dwsum
                  // (deferred like dgamma / dbeta); *extras tells whether it did (false: no room for the partial rows)
                  const float* row_w = nullptr, float* dsum = nullptr, float* dwsum = nullptr, bool* extras = nullptr,
                  uint32_t idx0 = 0) {
    const int nb = add_layernorm_bwd_blocks(M);
    static const bool no_extra = [] { const char* e = getenv("FIRA_LN_BWD_EXTRA"); return e && e[0] == '0'; }();   // A/B switch
    const bool want = row_w && dsum && dwsum && !no_extra;
    float* part = want ? red().alloc((size_t)nb * 4 * FIRA_D) : nullptr;
    const bool ex = part != nullptr;
    if (!part) part = red().alloc((size_t)nb * 2 * FIRA_D);
    if (extras) *extras = ex;
    const int w = ex ? 4 * FIRA_D : 2 * FIRA_D;
    TRY(add_layernorm_bwd(s, M, dy, sum, stats, gamma, ds, dx_drop, dgamma, dbeta, dropout, seed, st, rows, part,
                          ex ? row_w : nullptr, idx0));
    if (part) {
        red().add(dgamma, part, FIRA_D, nb, w);
        red().add(dbeta, part + FIRA_D, FIRA_D, nb, w);
        if (ex) {
            red().add(dsum, part + 2 * FIRA_D, FIRA_D, nb, w);
            red().add(dwsum, part + 3 * FIRA_D, FIRA_D, nb, w);
        }
    }
    return 0;
}

// LayerNorm backward of a decoder block + the data-gradient product that consumes its output, as ONE launch where the
// coalesced tile kernel takes the shape (fp32, 256-wide rows: gemm_tile32_lnb_try) -- otherwise the row kernel followed by
// the product.  dX[M,N] = dx_drop[M,256] . W[256,N] (W row-major [256,N]) (masked by relu_mask > 0); ds must not alias dy.
static int ln_bwd_dgrad(hipStream_t s, int M, int N, const float* dy, const float* sum, const float* stats, const float* gamma,
                        float* ds, float* dx_drop, float* dgamma, float* dbeta, float dropout, uint64_t seed, uint32_t st,
                        const float* W, int ldw, float* dX, int lddx, const float* relu_mask, uint32_t idx0 = 0) {
    // Up to ~1 500 rows (batch 64): beyond, the prologue repeated in each of the N / 32 column tiles of a row block is no longer
    // hidden by idle CUs (batch 170, 2 700 rows: 20 177 -> 20 567 commits/s without it; batch 64 neutral, batch 32 +0.5 % with it)
    static const int lnb_max_rows = [] { const char* e = getenv("FIRA_LN_BWD_MAX_ROWS"); return e ? atoi(e) : 1536; }();
    if (g_dtype == 0 && M <= lnb_max_rows && ds != dy && gemm_tile32_takes(0, M, N, FIRA_D, dy, FIRA_D, W, ldw)) {
        const int nb = gemm_tile32_lnb_blocks(M);
        float* part = red().alloc((size_t)nb * 2 * FIRA_D);
        int rc = 0;
        if (part && gemm_tile32_lnb_try(s, M, N, dy, W, ldw, dX, lddx, relu_mask, sum, stats, gamma, ds, dx_drop, part, dropout, seed,
                                        st, &rc, idx0)) {
            TRY(rc);
            red().add(dgamma, part, FIRA_D, nb, 2 * FIRA_D);
            red().add(dbeta, part + FIRA_D, FIRA_D, nb, 2 * FIRA_D);
            return 0;
        }
    }
    TRY(ln_bwd(s, M, dy, sum, stats, gamma, ds, dx_drop, dgamma, dbeta, dropout, seed, st, nullptr, nullptr, nullptr, nullptr, nullptr, idx0));
    return gemm_any(s, 0, 0, M, N, FIRA_D, dx_drop, FIRA_D, W, ldw, dX, lddx, nullptr, 0, 0, nullptr, nullptr, relu_mask);
}

// bf16 mode: the attention matmuls run on bf16 operands too (torch.autocast semantics); FIRA_ATTN_BF16=0 keeps the fp32
// MFMA chains (A/B switch)
static inline int attn_bf16() {
    static const bool off = [] { const char* e = getenv("FIRA_ATTN_BF16"); return e && e[0] == '0'; }();
    return g_dtype == 1 && !off;
}

// GCN layer as one fused launch per direction (gcn_fused.hip) instead of SpMM + product + add-LayerNorm (forward) /
// product + SpMM (backward); FIRA_GCN_FUSED=0 restores the separate kernels (A/B switch)
// ... and by the batch's density (round 6): the fused kernels' gather is built for FIRA's graphs (3-4 entries per computed row,
// the first 16 of a row in one batched round trip); rows beyond 16 entries take a 64-at-a-time tail loop per row, and on BASELINE
// config 5's graphs (116 entries per row) the fused forward costs 518 us against 319 us for aggregation + product + row kernel
// (bench.py: gcn_cfg5).  Batches averaging more than FIRA_GCN_FUSED_MAX_DEG entries per computed row (default 48) run the separate
// kernels; set per call by check_batch.
static thread_local bool g_dense_graphs = false;
static inline bool gcn_fused_on() {
    static const bool off = [] { const char* e = getenv("FIRA_GCN_FUSED"); return e && e[0] == '0'; }();
    return !off && !g_dense_graphs;
}
static inline void note_graph_density(const fira_batch* b) {
    static const double max_deg = [] { const char* e = getenv("FIRA_GCN_FUSED_MAX_DEG"); return e ? atof(e) : 48.0; }();
    g_dense_graphs = b->n_nodes > 0 && (double)b->nnz > max_deg * (double)b->n_nodes;
}

// ... and of the backward pass only (FIRA_GCN_FUSED_BWD=0: V = A_hat dY by the CSR kernel, dX += V W21 by the product -- the same
// identity in two launches).  Measured in bf16 at batch 64, where the product is a few microseconds of MFMA time and the
// fused launch's phase latencies are exposed: 16 239 / 16 203 commits/s fused against 16 110 / 15 982 -- fused stays the
// default in both modes.
static inline bool gcn_fused_bwd_on() {
    static const bool off = [] { const char* e = getenv("FIRA_GCN_FUSED_BWD"); return e && e[0] == '0'; }();
    return !off;
}

// Combination block as one fused launch (comb_fused.hip) instead of product + gate kernel + product + add-LayerNorm, in fp32
// and in bf16 mode (operands rounded as the panel products round them).  FIRA_COMB_FUSED=0 restores the separate kernels,
// FIRA_COMB_FUSED_BF16=0 in bf16 mode only (A/B switches).
static inline bool comb_fused_on() {
    static const bool off = [] { const char* e = getenv("FIRA_COMB_FUSED"); return e && e[0] == '0'; }();
    static const bool off16 = [] { const char* e = getenv("FIRA_COMB_FUSED_BF16"); return e && e[0] == '0'; }();
    return !off && !(g_dtype == 1 && off16);
}

static inline bool comb_fused_bwd_on() {         // FIRA_COMB_FUSED_BWD=0: the backward pass keeps its four launches (A/B switch)
    static const bool off = [] { const char* e = getenv("FIRA_COMB_FUSED_BWD"); return e && e[0] == '0'; }();
    return !off && comb_fused_on();
}

enum Site { SITE_GATE = 0, SITE_COMB_OUT = 1, SITE_GCN = 2, SITE_SELF = 3, SITE_CROSS = 4, SITE_FFN = 5 };
static inline uint32_t site(int layer, int kind) { return (uint32_t)(layer * 8 + kind + 1); }

struct Ctx {
    hipStream_t s;
    const Layout* L;
    const fira_batch* bt;
    const float* P;
    float* G;
    Plan* pl;
    float p_drop, p_gcn;
    uint64_t seed;
    // memory-side projections deferred to the side stream by encoder_forward (see there): marks to wait for
    // target rows that need the vocabulary head (fira_batch.head_rows, or nullptr = every row -> Plan::iota)
    int R = 0;
    const int32_t* rows = nullptr;
    // computed target rows (fira_batch.dec_off + fira_train_opts.compact_dec): the decoder, the head and their backward run
    // on Td rows -- commit b's are dec_off[b] .. dec_off[b+1], flat position row_bt[r]; rows / compact_row then index
    // those rows (rows_dense keeps the caller's flat head-row list for the prep launch).  No list: Td = B*T, all nullptr.
    int Td = 0;
    const int32_t* dec_off = nullptr;
    const int32_t* row_bt = nullptr;
    const int32_t* rows_dense = nullptr;
    bool deferred = false;
    // cross-attention K|V rows: the computed memory rows only, ragged per commit (Plan::mem_off / mem_valid_c) -- what
    // encoder_forward leaves; false: a dense [B, 370] memory supplied by the caller (fira_decoder_forward)
    bool kv_ragged = false;
    // the search's encoder pass: everything on the caller's stream.  Several searches run side by side on their own streams
    // (decode.Searcher.greedy_many); the library's auxiliary stream is ONE per thread, and HIP multiplexes streams onto a few
    // hardware queues: a fork onto it from one search queued behind another search's 290-kernel graph on the same queue
    // (measured: four batches in flight 54 ms instead of 18 once a trainer had created the streams)
    bool serial = false;
    float* loss_sum = nullptr;      // zeroed by the prep launch (head_loss accumulates into them)
    int32_t* n_tok = nullptr;
    hipEvent_t ev_kv[16] = {};
    int kv_waited = 0;
    hipEvent_t ev_src = nullptr;
    hipEvent_t ev_zero = nullptr;   // the backward pass's zero-initialised buffers were cleared on the auxiliary stream
    // fira_train_step (round 5): the optimizer update is part of the call.  Adam is element-wise, so the head + decoder slice
    // [0, split) is updated as soon as the caller's stream has finished the encoder's backward chain -- while the weight-gradient
    // and auxiliary streams still work on the encoder's last gradients -- and [split, live) after the join.  Pw = the parameters,
    // writable (nothing reads [0, split) after the decoder's backward pass; see backward()).
    const fira_adam_opts* adam = nullptr;
    float* Pw = nullptr;
    // (round 6) row-sparse Adam of the two vocabulary-sized embedding tables (fira_train_step_rows; copyhead.hip:
    // adam_rows_kernel): row_step [2 * vocab] = the step up to which each row of decoder.embedding / encoder.embedding is
    // current.  The rows this batch gathers are brought up to step - 1 ahead of the forward pass; the update itself runs on
    // the rows whose gradient row is not zero.
    bool dp_begin = false;           // fira_train_step_begin(_rows): the first half of a data-parallel step (see backward_decoder: dec_every)
    bool wout_planes = false;        // (round 6) WoutTX holds this step's planes: the vocabulary data gradient runs as dgrad_x3_splitk
    bool kv_planes = false;          // (round 6) WkvTX holds this step's planes: the d-memory products run as linear_x3_kacc
    int32_t* row_step = nullptr;
    fira_adam_opts rows_ad{};        // lr / beta / eps / step / moments the lazy reads use (a COPY: a begun data-parallel step outlives the call)
    // data-parallel form of the same (fira_train_step_begin / _end, round 6): the gradients of [0, split) are all-reduced by the
    // caller between the two calls -- Adam of [0, split) waits for ev_early (the caller's event behind that collective) instead of
    // the local weight-gradient mark, scales by the all-reduced token count `count` (a device float), and [split, live) is the
    // caller's (its bucket is reduced after the call)
    bool adam_a_only = false;
    hipEvent_t ev_early = nullptr;
    const float* count = nullptr;
    // bf16 mode (round 5): the refresh of the bf16 weight shadows (one 46 us launch) runs on the auxiliary stream behind the GCN
    // fold instead of at the head of the caller's stream -- with the fused Combination / GCN kernels (which round the fp32
    // weights themselves) the first reader of a shadow is the decoder (and the auxiliary stream's own K|V projections)
    const ShadowTable* shadow_tab = nullptr;
    hipEvent_t ev_shadow = nullptr;
    // (round 6) commit-lanes of the decoder's chain: see decoder_lanes()
    int n_lanes = 1;
    struct Lane { hipStream_t s; int b0, nb, r0, nr; } lanes[2] = {};
};
typedef Ctx::Lane Lane;

static AdamRowsTables adam_rows_tables(const Layout& L, float* params, const fira_adam_opts& ad, int32_t* row_step) {
    AdamRowsTables tb;
    tb.p = params; tb.m = ad.m; tb.v = ad.v;
    tb.off[0] = L.dec_emb; tb.off[1] = L.emb;
    tb.rows[0] = tb.rows[1] = L.d.vocab;
    tb.last = row_step;
    return tb;
}

// ------------------------------------------------------------------------------------------ commit-lanes (round 6)
// The decoder's layers -- forward and backward -- are a dependent chain of ~100 launches whose kernels each leave most of the
// chip idle (17 .. 136 workgroups of a 32x32 tile kernel, one workgroup per (commit, head) in the attention kernels) and cost
// their latency, not their work.  Commits are independent (every kernel of the chain is row-wise or per commit, and the
// weight gradients are sums over commits): the batch's computed target rows are cut at a commit boundary into TWO contiguous
// lanes, lane 0 on the caller's stream and lane 1 on a library-owned stream, each running the same launches on its rows of the
// SAME activation buffers; the grouped weight-gradient launches and the d-memory products on the other streams read the
// operands of both lanes and are forked from both.  Results are those of one lane (same dropout masks: the kernels take the
// element index of a slice's first row) up to the row-block boundaries of the deferred column sums.
// FIRA_DEC_LANES=1|2 (default: see decoder_lanes); needs the host copy of dec_off (fira_batch.dec_off_host).
static int decoder_lanes(Ctx& c) {
    // Measured (profiles/r6_probes.md, same-box triples): two lanes are -1.9 % at batch 32 (530 rows: every launch of the chain is
    // a latency-bound workgroup chain whose duration does not shrink with half the rows -- the two lanes' launches take as long
    // each as the one they replace, and the extra forks / joins are pure cost), +2.2 % at batch 64 (1 130 rows, fp32), +0.6 % in
    // bf16 at batch 64.  Default: two lanes from FIRA_DEC_LANES_MIN_ROWS computed target rows; FIRA_DEC_LANES=1|2 forces.
    static const int forced = [] { const char* e = getenv("FIRA_DEC_LANES"); return e ? (atoi(e) >= 2 ? 2 : 1) : 0; }();
    static const int min_rows = [] { const char* e = getenv("FIRA_DEC_LANES_MIN_ROWS"); return e ? atoi(e) : 768; }();
    const int want = forced ? forced : (c.Td >= min_rows ? 2 : 1);
    c.n_lanes = g_lanes = 1;
    c.lanes[0] = Lane{c.s, 0, c.pl->B, 0, c.Td};
    const fira_batch& bt = *c.bt;
    if (want < 2 || !c.dec_off || !bt.dec_off_host || bt.B < 2 || !side_on() || c.serial || !c.kv_ragged) return 0;
    SideStream& sd = side();
    if (!sd.lane) {
        // same (default) priority as a caller's stream: the two lanes are peers
        if (hipStreamCreateWithFlags(&sd.lane, hipStreamNonBlocking) != hipSuccess) return set_err("lane stream: create failed");
    }
    int bm = 1, best = 1 << 30;
    for (int b = 1; b < bt.B; ++b) {                        // the commit boundary closest to half of the rows
        const int d = abs(2 * bt.dec_off_host[b] - c.Td);
        if (d < best) { best = d; bm = b; }
    }
    const int rm = bt.dec_off_host[bm];
    if (rm <= 0 || rm >= c.Td || bt.dec_off_host[bt.B] != c.Td) return 0;
    c.n_lanes = g_lanes = 2;
    c.lanes[0] = Lane{c.s, 0, bm, 0, rm};
    c.lanes[1] = Lane{sd.lane, bm, bt.B - bm, rm, c.Td - rm};
    return 0;
}
// lane 1 starts behind everything the caller's stream holds so far / the caller's stream continues behind lane 1
static int lanes_fork(Ctx& c) {
    for (int k = 1; k < c.n_lanes; ++k) {
        hipEvent_t e = side().ev();
        if (hipEventRecord(e, c.s) != hipSuccess || hipStreamWaitEvent(c.lanes[k].s, e, 0) != hipSuccess) return set_err("lane fork failed");
    }
    return 0;
}
static int lanes_join(Ctx& c) {
    for (int k = 1; k < c.n_lanes; ++k) {
        hipEvent_t e = side().ev();
        if (hipEventRecord(e, c.lanes[k].s) != hipSuccess || hipStreamWaitEvent(c.s, e, 0) != hipSuccess) return set_err("lane join failed");
    }
    return 0;
}
// `target` (the weight-gradient or the auxiliary stream) waits for everything enqueued on EVERY lane so far
static int lanes_fork_to(Ctx& c, hipStream_t target) {
    for (int k = 0; k < c.n_lanes; ++k) {
        hipEvent_t e = side().ev();
        if (hipEventRecord(e, c.lanes[k].s) != hipSuccess || hipStreamWaitEvent(target, e, 0) != hipSuccess) return set_err("lane -> side stream fork failed");
    }
    return 0;
}

// ------------------------------------------------------------------------------------------ encoder forward
// The encoder runs on the batch's COMPUTED node list only (fira_batch.node_rows): padded nodes carry nothing but a
// self-loop, are masked as attention keys / copy slots and receive exactly zero gradient (SURVEY.md §8a note N1), so
// leaving them out changes no consumed value.  Nc = n_nodes, Cc = n_code, Mc = n_mem below.
// FIRA_FOLD_ONE=0: the folded GCN weights as two products + a transpose per layer (14 launches) instead of one launch (A/B switch)
// FIRA_GCN_X3=0: the fused GCN product of fp32 mode as fp32 MFMAs instead of three bf16 terms (A/B switch; gcn_fused.hip)
static inline bool fold_one_launch();
// FIRA_COMB_X3=0: the same switch for the fused Combination block's three products per direction (comb_fused.hip)
static inline bool comb_x3_on(int nl) {
    static const bool off = [] { const char* e = getenv("FIRA_COMB_X3"); return e && e[0] == '0'; }();
    static const bool off16 = [] { const char* e = getenv("FIRA_X1_BF16"); return e && e[0] == '0'; }();
    return !off && (g_dtype == 0 || !off16) && comb_fused_on() && nl <= 8;
}
// the training step's K|V projection of the memory rows as linear_x3 (three-term planes in fp32 mode, one plane in bf16 mode)
// instead of one fp32 / bf16 GEMM launch per layer; FIRA_KV_X3=0: those launches (A/B switch)
static inline bool kv_x3_on(int nl) {
    static const bool off = [] { const char* e = getenv("FIRA_KV_X3"); return e && e[0] == '0'; }();
    return !off && nl * 2 <= 24;
}
static inline bool gcn_x3_on(int nl) {
    static const bool off = [] { const char* e = getenv("FIRA_GCN_X3"); return e && e[0] == '0'; }();
    // (bf16 mode runs the one-plane form of the same kernels; FIRA_X1_BF16=0 = its round-5 kernels, A/B switch)
    static const bool off16 = [] { const char* e = getenv("FIRA_X1_BF16"); return e && e[0] == '0'; }();
    return !off && (g_dtype == 0 || !off16) && gcn_fused_on() && fold_one_launch() && nl <= 10;
}
static inline bool fold_one_launch() {
    static const bool off = [] { const char* e = getenv("FIRA_FOLD_ONE"); return e && e[0] == '0'; }();
    return !off;
}

// the target-word rows this batch's decoder gathers, brought up to step - 1 in memory (the dense-target-row path only: the
// compact paths read lazily, adam_rows_load)
static int adam_rows_prefetch(Ctx& c, hipStream_t st) {
    const fira_batch& bt = *c.bt;
    const Plan& p = *c.pl;
    const fira_adam_opts& ad = c.rows_ad;
    AdamRowsLists ls{};
    auto add = [&](const int32_t* ids, int n, int table) {
        if (!ids || n <= 0) return;
        const int k = ls.n_lists++;
        ls.ids[k] = ids; ls.table[k] = table;
        ls.end[k] = (k ? ls.end[k - 1] : 0) + n;
    };
    add(bt.tar, p.B * p.T, 0);
    return adam_rows_catchup(st, adam_rows_tables(*c.L, const_cast<float*>(c.P), ad, c.row_step), &ls, ad.lr, ad.beta1, ad.beta2, ad.eps,
                             ad.step - 1);
}

static int encoder_forward(Ctx& c, bool defer_memory_proj) {
    Plan& p = *c.pl;
    const Layout& L = *c.L;
    const fira_batch& bt = *c.bt;
    hipStream_t s = c.s;
    const int D = FIRA_D, Nc = bt.n_nodes, Cc = bt.n_code, Mc = bt.n_mem, KV = p.nl * 2 * D;
    // GCN (gnn_transformer.py:74-86) has no non-linearity between fc1, the aggregation and fc2, so
    //     fc2(A_hat fc1(X)) = (A_hat X) (W2 W1)^T + (A_hat 1) (W2 b1)^T + b2 :
    // one [Nc,256]x[256,256] product per layer instead of two (and one dgrad, one wgrad in the backward pass); the folded
    // 256x256 weights are formed here, per step, from the current parameters (same value up to fp32 re-association).
    // They only read parameters: the auxiliary stream forms them from the very start of the call, beside the embedding /
    // first Combination kernels.  Layer 0's pair is marked on its own (the caller's stream reaches its first GCN product
    // ~100 us into the call, 12 launches of the auxiliary stream would not be through by then); the rest is awaited at
    // layer 1.
    hipEvent_t ev_fold0 = nullptr, ev_fold = nullptr, ev_comb = nullptr;
    static const bool one_wait = [] { const char* e = getenv("FIRA_FOLD_ONE_WAIT"); return !(e && e[0] == '0'); }();   // A/B switch
    bool fold_waited = false;
    {
        const bool ax = side_on() && !c.serial;
        hipStream_t fs = ax ? side().aux : s;
        if (ax) TRY(aux_fork(s));
        if (comb_fused_on()) {
            // k-major copies of the Combination weights of every layer (Wq^T, Wk^T, Wo^T): one launch, the first thing on the
            // auxiliary stream -- layer 0's block is the third launch of the caller's stream
            TransposeTable tt;
            for (int l = 0; l < p.nl && tt.n + 3 <= 24; ++l) {
                const EncLayer& w = L.enc[l];
                float* dst = p.WcT + (size_t)l * 3 * D * D;
                tt.src[tt.n] = c.P + w.wqk; tt.dst[tt.n++] = dst;
                tt.src[tt.n] = c.P + w.wqk + (size_t)D * D; tt.dst[tt.n++] = dst + (size_t)D * D;
                tt.src[tt.n] = c.P + w.wo; tt.dst[tt.n++] = dst + (size_t)2 * D * D;
            }
            TRY(transpose256_table(fs, tt));
            if (comb_x3_on(p.nl)) {      // planes of Wq | Wk | Wo as stored (forward) and of the k-major copies (backward)
                const float* src[48];
                uint16_t* dst[48];
                int n = 0;
                for (int l = 0; l < p.nl; ++l) {
                    const EncLayer& w = L.enc[l];
                    const float* st[3] = {c.P + w.wqk, c.P + w.wqk + (size_t)D * D, c.P + w.wo};
                    for (int j = 0; j < 3; ++j) {
                        src[n] = st[j]; dst[n++] = p.WcX + ((size_t)l * 3 + j) * 3 * D * D;
                        if (c.G) { src[n] = p.WcT + ((size_t)l * 3 + j) * D * D; dst[n++] = p.WcTX + ((size_t)l * 3 + j) * 3 * D * D; }
                    }
                }
                TRY(gcn_split_planes(fs, n, src, dst));
            }
            // (with the one-launch fold right behind it, ONE mark serves the first Combination block and the first GCN layer: every
            //  wait is a barrier packet in the caller's chain, and the fold is through ~25 us into the call, before the first
            //  Combination block starts)
            if (ax && !(one_wait && fold_one_launch() && p.nl <= 10 && (!g_Wb || gcn_fused_on()))) TRY(side_mark(&ev_comb));
        }
        const bool fold_one = fold_one_launch() && p.nl <= 10;
        if (fold_one) {
            // (round 5) every layer's W21, its k-major copy and c21: ONE launch (gemm_small.hip: gcn_fold_weights).  The fused GCN
            // kernels read the fp32 matrices in both modes, so the mark behind this launch is all the first layer waits for
            const float *fW2[16], *fW1[16], *fb1[16];
            for (int l = 0; l < p.nl; ++l) { fW2[l] = c.P + L.enc[l].fc2w; fW1[l] = c.P + L.enc[l].fc1w; fb1[l] = c.P + L.enc[l].fc1b; }
            TRY(gcn_fold_weights(fs, p.nl, fW2, fW1, fb1, p.W21, gcn_fused_on() ? p.W21t : nullptr, p.c21));
            if (gcn_x3_on(p.nl)) {       // the planes of W21 (forward: out = U W21^T) and of W21^T (backward: out = V W21)
                const float* src[24];
                uint16_t* dst[24];
                int n = 0;
                for (int l = 0; l < p.nl; ++l) {
                    src[n] = p.W21 + (size_t)l * D * D; dst[n++] = p.W21x + (size_t)l * 3 * D * D;
                    if (c.G) { src[n] = p.W21t + (size_t)l * D * D; dst[n++] = p.W21tx + (size_t)l * 3 * D * D; }
                }
                TRY(gcn_split_planes(fs, n, src, dst));
            }
            if (ax && (!g_Wb || gcn_fused_on())) TRY(side_mark(&ev_fold0));
        } else
        for (int l = 0; l < p.nl; ++l) {
            const EncLayer& w = L.enc[l];
            TRY(gemm_f32_ex(fs, 0, 0, D, D, D, c.P + w.fc2w, D, c.P + w.fc1w, D, p.W21 + (size_t)l * D * D, D, nullptr, 0, 0, nullptr));
            TRY(gemm_f32_ex(fs, 0, 1, D, 1, D, c.P + w.fc2w, D, c.P + w.fc1b, D, p.c21 + (size_t)l * D, 1, nullptr, 0, 0, nullptr));
            if (l == 0 && gcn_fused_on()) TRY(transpose256(fs, 1, p.W21, p.W21t));       // (layer 0's first: see ev_fold0)
            if (ax && l == 0 && !g_Wb && p.nl > 1) TRY(side_mark(&ev_fold0));
        }
        if (g_Wb) {                                  // bf16 mode: shadows of the folded weights, on the same stream
            ShadowTable t21;
            for (int l = 0; l < p.nl && l < SHADOW_MAX; ++l) {
                t21.e[l] = ShadowEntry{(int64_t)l * D * D, D, D, D, (int64_t)l * D * D};
                t21.tile_start[l + 1] = t21.tile_start[l] + cdiv(D, 64) * cdiv(D, 64);
                t21.n = l + 1;
            }
            TRY(weight_shadows(fs, t21, p.W21, p.w21b, p.w21bt));
            g_W21 = p.W21; g_W21n = (int64_t)p.nl * D * D; g_W21b = p.w21b; g_W21bT = p.w21bt;
        }
        if (!fold_one && gcn_fused_on() && p.nl > 1) TRY(transpose256(fs, p.nl - 1, p.W21 + (size_t)D * D, p.W21t + (size_t)D * D));
        // (the closing mark only where something follows the fold that the encoder's own launches read: the unfused GCN path's
        //  bf16 shadows of the folded weights, or the layer-by-layer fold)
        if (ax && !(one_wait && ev_fold0 && fold_one && gcn_fused_on())) TRY(side_mark(&ev_fold));
        if (!ev_fold0) ev_fold0 = ev_fold;           // bf16 mode / one layer: a single mark
        if (c.shadow_tab) {                          // (see Ctx::shadow_tab)
            TRY(weight_shadows(fs, *c.shadow_tab, c.P, p.wb, p.wbt));
            if (ax) TRY(side_mark(&c.ev_shadow));
        }
    }
    // masks, position tables, inverse of the head-row list: one launch; node features straight into the compact layout
    const int32_t* head_list = c.dec_off ? c.rows_dense : c.rows;          // flat (b*T + t) head rows, as the caller lists them
    TRY(prep(s, p.B, p.L, p.S, p.T, bt.sou, bt.sub_token, bt.tar, p.mem_valid, bt.tar ? p.tar_valid : nullptr, p.pos_code,
             p.pos_tar, c.R, head_list, bt.tar ? p.compact_row : nullptr, head_list ? nullptr : p.iota, c.loss_sum, c.n_tok,
             Nc, bt.code_rows, Cc, p.code_slot, bt.mem_rows, Mc, p.mem_slot, c.dec_off, p.row_bt, p.rows_c, bt.mem_dst,
             p.mem_off, p.mem_valid_c));
    c.kv_ragged = true;
    // layer 0's code rows are also stored compactly (Xc of the first Combination): no gather launch on the chain
    // (row-sparse Adam: a word row the last steps did not touch is read with the zero-gradient updates it still owes applied in
    //  registers -- adam_rows.h; the view is empty otherwise)
    AdamRowsView vw_emb;
    if (c.row_step)
        vw_emb = adam_rows_view(adam_rows_tables(L, nullptr, c.rows_ad, c.row_step), 1, c.rows_ad.lr, c.rows_ad.beta1, c.rows_ad.beta2,
                                c.rows_ad.eps, c.rows_ad.step - 1);
    TRY(node_features(s, Nc, bt.node_rows, p.N, p.L, p.S, bt.sou, bt.sub_token, bt.ast_change, c.P + L.emb, c.P + L.ast_emb,
                      p.pos_code, p.X[0], p.code_slot, p.enc[0].Xc, &vw_emb));
    // value projection of the 4-row mark table for all layers at once: vtab_all [4, nl*256]
    TRY(linear(s, 4, p.nl * D, D, c.P + L.mark_emb, D, c.P + L.w2_all, c.P + L.b2_all, p.vtab_all, p.nl * D));
    for (int l = 0; l < p.nl; ++l) {
        const EncLayer& w = L.enc[l];
        EncSave& e = p.enc[l];
        float* X = p.X[l];
        // Combination (gnn_transformer.py:192-205): code rows only; the result overwrites them in place.  e.Xc (the code
        // rows before the update: residual, and the q|k weight gradient's operand) was stored by the kernel that produced X
        if (comb_fused_on() && l < 8) {
            if (l == 0 && ev_comb) TRY(main_wait(s, ev_comb, __LINE__));
            else if (l == 0 && ev_fold0) { TRY(main_wait(s, ev_fold0, __LINE__)); fold_waited = true; }
            const float* wt = p.WcT + (size_t)l * 3 * D * D;
            TRY(comb_fused_fwd(s, Cc, e.Xc, wt, wt + (size_t)D * D, wt + (size_t)2 * D * D, c.P + w.bqk, c.P + w.bo,
                               p.vtab_all + l * D, p.nl * D, bt.code_mark, e.qk, e.c, c.P + w.ln1g, c.P + w.ln1b, e.s1, X,
                               bt.code_rows, e.st1, c.p_drop, c.seed, site(l, SITE_GATE), site(l, SITE_COMB_OUT), g_dtype == 1,
                               comb_x3_on(p.nl) ? p.WcX + (size_t)l * 9 * D * D : nullptr));
        } else {
        TRY(linear(s, Cc, 2 * D, D, e.Xc, D, c.P + w.wqk, c.P + w.bqk, e.qk, 2 * D));
        TRY(combination_fwd(s, Cc, e.qk, p.vtab_all + l * D, p.nl * D, bt.code_mark, e.c, c.p_drop, c.seed, site(l, SITE_GATE)));
        TRY(linear_ln(s, Cc, D, e.c, D, c.P + w.wo, c.P + w.bo, e.Xc, c.P + w.ln1g, c.P + w.ln1b, e.s1, X, e.st1, c.p_drop,
                      c.seed, site(l, SITE_COMB_OUT), bt.code_rows));
        }
        // GCN in folded form: U = A_hat X -> U W21^T + b2 -> (+ r c^T) dropout, +X, LN
        const bool fused = gcn_fused_on();
        if (!fused) TRY(csr_spmm_ex(s, Nc, bt.rowptr, bt.col, bt.val, X, D, e.Z, D, 0, 1, 0, l == 0 ? p.rsum : nullptr));
        if (l == 0 && ev_fold0 && !fold_waited) TRY(main_wait(s, ev_fold0, __LINE__));   // the product below is the first reader of W21 / c21
        if (l == 1 && ev_fold && ev_fold != ev_fold0) TRY(main_wait(s, ev_fold, __LINE__));
        // second store of the output: the next layer's code rows (its Xc), or after the last layer the memory rows
        // (memory = [code ; sub-token] rows, Model.py:48: compact copy for the GEMMs.  The cross-attention K|V stay in that
        // compact row order -- the attention kernels take commit b's key range from mem_off; the dense [B,370,256] rows the
        // copy kernels read are scattered by LinearSource's projection below, rows of masked slots are never read there)
        const bool last = l + 1 == p.nl;
        if (fused)             // one launch: gather, product, bias + rank-1 term, dropout, residual, LayerNorm, both stores
            TRY(gcn_fused_fwd(s, Nc, bt.rowptr, bt.col, bt.val, X, p.W21t + (size_t)l * D * D, c.P + w.fc2b, p.c21 + (size_t)l * D,
                              c.P + w.ln2g, c.P + w.ln2b, e.s2, p.X[l + 1], e.st2, l == 0 ? p.rsum : nullptr,
                              last ? p.mem_slot : p.code_slot, last ? p.mem_c : p.enc[l + 1].Xc, c.p_gcn, c.seed,
                              site(l, SITE_GCN), g_dtype == 1, gcn_x3_on(p.nl) ? p.W21x + (size_t)l * 3 * D * D : nullptr));
        else
        TRY(linear_ln(s, Nc, D, e.Z, D, p.W21 + (size_t)l * D * D, c.P + w.fc2b, X, c.P + w.ln2g, c.P + w.ln2b, e.s2,
                      p.X[l + 1], e.st2, c.p_gcn, c.seed, site(l, SITE_GCN), nullptr, p.rsum, p.c21 + (size_t)l * D,
                      last ? p.mem_slot : p.code_slot, last ? p.mem_c : p.enc[l + 1].Xc, last ? Mc : Cc,
                      last ? bt.mem_rows : bt.code_rows));
    }
    c.deferred = false;
    if (defer_memory_proj && side_on() && p.nl <= 16) {
        // The decoder consumes layer l's K|V only at its l-th cross attention and LinearSource(memory) only in the
        // copy head, and the decoder's own chain is a string of small latency-bound kernels: the projections run per
        // layer on the side stream under it; decoder_forward / head_forward wait for the mark they need.
        TRY(aux_fork(s));
        hipStream_t ss = side().aux;
        const bool kvx = c.G && kv_x3_on(p.nl);         // (the training step only: the search keeps the fp32 chains of its reference lines)
        if (kvx) {
            // (round 6) two launches, one per mark: layers 0-1, then the rest; the planes of the stacked weight first
            const float* src[24];
            uint16_t* dst[24];
            for (int j = 0; j < p.nl * 2; ++j) { src[j] = c.P + L.wkv_all + (size_t)j * D * D; dst[j] = p.WkvX + (size_t)j * 3 * D * D; }
            TRY(gcn_split_planes(ss, p.nl * 2, src, dst));
        }
        for (int l = 0; l < p.nl; ++l) {
            const size_t o = (size_t)l * 2 * D;
            if (kvx) {
                const int l1 = l == 0 ? std::min(2, p.nl) : p.nl;            // layers l .. l1 in this launch
                if (l == 0 || l == std::min(2, p.nl))
                    TRY(linear_x3(ss, Mc, p.mem_c, D, p.WkvX + (size_t)l * 2 * 3 * D * D, (l1 - l) * 2, c.P + L.bkv_all + o, p.kv_all + o,
                                  p.kvp, g_dtype == 1));
            } else
            TRY(gemm_any(ss, 0, 1, Mc, 2 * D, D, p.mem_c, D, c.P + L.wkv_all + o * D, D, p.kv_all + o, p.kvp, c.P + L.bkv_all + o, 0,
                         0, nullptr));
            // two marks (behind layers 1 and the last one) instead of one per layer: every wait is a barrier packet in the
            // decoder's dependent chain, and the projections (~10 us each) are far ahead of the layers that read them
            // (layer 2's cross attention starts ~170 us after this fork; one mark per layer measured -0.5 % in round 4)
            if (l == std::min(1, p.nl - 1) || l == p.nl - 1) TRY(side_mark(&c.ev_kv[l]));
        }
        // (LinearSource: output rows straight to their dense [B,370] slots through the row map of the GEMM epilogue)
        TRY(gemm_any(ss, 0, 1, Mc, D, D, p.mem_c, D, c.P + L.ws, D, p.src, D, nullptr, 0, 0, nullptr, bt.mem_dst));
        TRY(side_mark(&c.ev_src));
        if (kvx && p.nl * 2 <= 24) {
            // the transposed blocks' planes for the backward pass's d-memory products: behind everything the forward pass waits for
            TransposeTable tt;
            const float* src[24];
            uint16_t* dst[24];
            for (int j = 0; j < p.nl * 2; ++j) {
                tt.src[tt.n] = c.P + L.wkv_all + (size_t)j * D * D; tt.dst[tt.n++] = p.WkvT + (size_t)j * D * D;
                src[j] = p.WkvT + (size_t)j * D * D; dst[j] = p.WkvTX + (size_t)j * 3 * D * D;
            }
            TRY(transpose256_table(ss, tt));
            TRY(gcn_split_planes(ss, p.nl * 2, src, dst));
            c.kv_planes = true;
        }
        // ... and, in bf16 mode, the plane of the generator projection's transposed row blocks (one 38 MB pass under the decoder's
        // forward chain) for its data gradient as dgrad_x3_splitk: +1.0 % at batch 64.  (fp32 mode keeps the fp32 MFMA launch: the
        // three-plane form streams 680 MB of planes through L2 per launch at batch 32 and LOST 0.9 % -- profiles/r6_probes.md;
        // FIRA_VOCAB_DGRAD_X3=1 forces it, =0 switches it off in both modes.)
        static const int vd_mode = [] { const char* e = getenv("FIRA_VOCAB_DGRAD_X3"); return e ? atoi(e) : -1; }();
        if (c.G && (vd_mode == 1 || (vd_mode == -1 && g_dtype == 1))) {
            TRY(split_planes_t(ss, c.P + L.wout, p.V, p.WoutTX, g_dtype == 1));
            c.wout_planes = true;
        }
        c.deferred = true;
        return 0;
    }
    TRY(gemm_any(s, 0, 1, Mc, KV, D, p.mem_c, D, c.P + L.wkv_all, D, p.kv_all, p.kvp, c.P + L.bkv_all, 0, 0, nullptr));
    TRY(gemm_any(s, 0, 1, Mc, D, D, p.mem_c, D, c.P + L.ws, D, p.src, D, nullptr, 0, 0, nullptr, bt.mem_dst));
    return 0;
}

// ------------------------------------------------------------------------------------------ decoder forward
static int decoder_forward(Ctx& c) {
    Plan& p = *c.pl;
    const Layout& L = *c.L;
    const int D = FIRA_D, H = L.d.n_head, Sm = p.L + p.S;
    if (c.row_bt) {
        AdamRowsView vw;                          // (see encoder_forward: the target-word table under a row-sparse Adam)
        if (c.row_step)
            vw = adam_rows_view(adam_rows_tables(L, nullptr, c.rows_ad, c.row_step), 0, c.rows_ad.lr, c.rows_ad.beta1, c.rows_ad.beta2,
                                c.rows_ad.eps, c.rows_ad.step - 1);
        TRY(embed_rows_fwd(c.s, c.Td, p.T, c.row_bt, c.bt->tar, c.P + L.dec_emb, p.pos_tar, p.x0, &vw));
    } else {
        if (c.row_step) TRY(adam_rows_prefetch(c, c.s));      // dense target rows: the batch's rows brought up to date first
        TRY(embed_gather_fwd(c.s, p.B, p.T, c.bt->tar, c.P + L.dec_emb, p.pos_tar, p.x0, p.T, 0));
    }
    ProfDecoderTag prof_tag;                   // the M = B*30 products below are reported as their own class
    TRY(decoder_lanes(c));
    // wall time of the (possibly two-lane) layer loop on the caller's stream: what the decoder's products cost the step
    ProfScope prof_region(c.s, PROF_DEC_REGION, 0.0);
    TRY(lanes_fork(c));
    // Per lane (Ctx::lanes; one lane = the whole batch on the caller's stream): x = the current rows; pend_*: a block whose
    // pre-norm sums are stored but whose LayerNorm is still owed (it runs in the prologue of the next product: see
    // linear_presum).  y = LN(pend_sum) goes to pend_y, the statistics to pend_st.  All pointers are buffer BASES: a lane
    // addresses its rows r0 .. r0 + nr of every buffer.
    struct LaneState {
        const float *x = nullptr, *pend_sum = nullptr, *pend_g = nullptr, *pend_b = nullptr;
        float *pend_y = nullptr, *pend_st = nullptr;
        int kv_waited = 0;
    } state[2];
    for (int k = 0; k < c.n_lanes; ++k) state[k].x = p.x0;
    for (int l = 0; l < p.nl; ++l) {
        const DecLayer& w = L.dec[l];
        DecSave& e = p.dec[l];
        const bool last = l + 1 == p.nl;
        for (int k = 0; k < c.n_lanes; ++k) {
            const Lane& ln = c.lanes[k];
            LaneState& st = state[k];
            hipStream_t s = ln.s;
            const size_t r0 = (size_t)ln.r0;
            const int nr = ln.nr;
            const uint32_t idx0 = (uint32_t)(r0 * D);          // dropout element index of the lane's first row
            // Y = x W^T + b where x is either materialised or owed (pend_*): the fused launch, or the row kernel + a plain product
            auto consume = [&](int N, const float* W, const float* b, float* Y, int flags) -> int {
                if (st.pend_sum) {
                    int rc = 0;
                    float* sum = const_cast<float*>(st.pend_sum) + r0 * D;
                    const bool fused = gemm_tile32_ln_try(s, nr, N, sum, D, W, b, Y + r0 * N, N, flags, st.pend_g, st.pend_b,
                                                          st.pend_y + r0 * D, st.pend_st + r0 * 2, &rc);
                    if (!fused)
                        rc = add_layernorm_fwd(s, nr, sum, nullptr, st.pend_g, st.pend_b, st.pend_y + r0 * D, st.pend_st + r0 * 2, 0.f, 0, 0,
                                               nullptr);
                    st.pend_sum = nullptr;
                    if (rc || fused) return rc;
                }
                return linear(s, nr, N, D, st.x + r0 * D, D, W, b, Y + r0 * N, N, flags);
            };
            // closing product of a block: y = LN(dropout(X W^T + b) + res); its LayerNorm is deferred to the next product when
            // the shapes allow it
            auto close_block = [&](int K, const float* X, const float* W, const float* b, const float* res, const float* g,
                                   const float* be, float* sum, float* y, float* stt, uint32_t site_id, bool may_defer,
                                   const int32_t* slot2, float* y2, int n2, const int32_t* rows2) -> int {
                int rc = 0;
                if (may_defer && linear_presum(s, nr, K, X + r0 * K, K, W, b, res + r0 * D, sum + r0 * D, c.p_drop, c.seed, site_id,
                                               &rc, idx0)) {
                    st.pend_sum = sum; st.pend_g = g; st.pend_b = be; st.pend_y = y; st.pend_st = stt;
                    return rc;
                }
                return linear_ln(s, nr, K, X + r0 * K, K, W, b, res + r0 * D, g, be, sum + r0 * D, y + r0 * D, stt + r0 * 2, c.p_drop,
                                 c.seed, site_id, nullptr, nullptr, nullptr, slot2 ? slot2 + r0 : nullptr, y2, n2, rows2, idx0);
            };
            // (a lane's commits: ranges of the ragged row lists start at its first commit; the dense key mask of the self
            //  attention is indexed by commit)
            const int32_t* q_off = c.dec_off ? c.dec_off + ln.b0 : nullptr;
            TRY(consume(3 * D, c.P + w.wqkv, c.P + w.bqkv, e.qkv, 0));
            TRY(attention_fwd(s, ln.nb, H, p.T, p.T, e.qkv, 3 * D, e.qkv + D, 3 * D, e.qkv + 2 * D, 3 * D, p.tar_valid + (size_t)ln.b0 * p.T,
                              1, 0, e.ao, D, q_off, 1, attn_bf16()));
            TRY(close_block(D, e.ao, c.P + w.wo_s, c.P + w.bo_s, st.x, c.P + w.lns_g, c.P + w.lns_b, e.s_a, e.x_a, e.st_a,
                            site(l, SITE_SELF), true, nullptr, nullptr, 0, nullptr));
            st.x = e.x_a;
            TRY(consume(D, c.P + w.wq_c, c.P + w.bq_c, e.qc, 0));
            // this layer's K|V rows (auxiliary stream): the first mark at or behind the layer, waited for once per lane
            if (c.deferred) {
                int m = l;
                while (m < p.nl - 1 && c.ev_kv[m] == nullptr) ++m;     // the first mark at or behind this layer
                if (c.ev_kv[m] != nullptr && st.kv_waited < m + 1) {
                    TRY(main_wait(s, c.ev_kv[m], __LINE__));
                    st.kv_waited = m + 1;                              // layers < kv_waited are covered
                }
            }
            TRY(attention_fwd(s, ln.nb, H, p.T, Sm, e.qc, D, p.kv_all + l * 2 * D, p.kvp, p.kv_all + l * 2 * D + D, p.kvp,
                              c.kv_ragged ? p.mem_valid_c : p.mem_valid, 0, 0, e.ao2, D, q_off, 0, attn_bf16(),
                              c.kv_ragged ? p.mem_off + ln.b0 : nullptr));
            TRY(close_block(D, e.ao2, c.P + w.wo_c, c.P + w.bo_c, e.x_a, c.P + w.lnc_g, c.P + w.lnc_b, e.s_c, e.x_c, e.st_c,
                            site(l, SITE_CROSS), true, nullptr, nullptr, 0, nullptr));
            st.x = e.x_c;
            TRY(consume(p.F, c.P + w.w1, c.P + w.b1, e.h, FIRA_GEMM_RELU));
            // the last layer's output rows that need the vocabulary head (Ctx::rows) are also stored compactly (dec_c); its
            // LayerNorm is not deferred (several consumers: vocabulary head, target projection, gate)
            const bool head_copy = last && c.rows != nullptr && c.R > 0;
            TRY(close_block(p.F, e.h, c.P + w.w2, c.P + w.b2, e.x_c, c.P + w.lnf_g, c.P + w.lnf_b, e.s_f, e.x_f, e.st_f,
                            site(l, SITE_FFN), !last, head_copy ? p.compact_row : nullptr, head_copy ? p.dec_c : nullptr,
                            head_copy ? c.R : 0, head_copy ? c.rows : nullptr));
            st.x = e.x_f;
        }
    }
    TRY(lanes_join(c));
    return 0;
}

// ------------------------------------------------------------------------------------------ head forward (+ loss)
// The vocabulary GEMM runs on the R rows listed in `rows` (bt indices); the copy / gate branch is dense.
static int head_forward(Ctx& c, int R, const int32_t* rows, float* loss_sum, int32_t* n_tok, int32_t* argmax_out,
                        int want_grad) {
    Plan& p = *c.pl;
    const Layout& L = *c.L;
    hipStream_t s = c.s;
    const int D = FIRA_D, Sm = p.L + p.S;
    const float* dec = p.dec[p.nl - 1].x_f;
    // rows == Ctx::rows (a proper sub-list): decoder_forward stored them compactly already; every row: use them in place
    const float* dec_rows = (c.rows != nullptr && rows == c.rows) ? p.dec_c : dec;
    if (dec_rows == dec && R != c.Td) {                         // a sub-list the decoder pass did not know about
        TRY(rows_gather_idx(s, R, p.dec_c, dec, rows));
        dec_rows = p.dec_c;
    }
    // fp32 mode: the generator projection as three bf16 terms per operand (head_x3.hip; FIRA_HEAD_X3=0 = the fp32 tiled kernel)
    static const bool head_x3_off = [] { const char* e = getenv("FIRA_HEAD_X3"); return e && e[0] == '0'; }();
    if (g_dtype == 0 && !head_x3_off && R >= 64 && R <= p.TB)
        TRY(head_logits_x3(s, R, p.V, dec_rows, D, c.P + L.wout, c.P + L.bout, p.logits, p.ldl, p.xh_planes));
    else
    TRY(linear(s, R, p.V, D, dec_rows, D, c.P + L.wout, c.P + L.bout, p.logits, p.ldl));
    TRY(gemm_any(s, 0, 1, c.Td, D, D, dec, D, c.P + L.wt, D, p.tgt, D, nullptr, 0, 0, nullptr));
    if (c.deferred) TRY(main_wait(s, c.ev_src, __LINE__));                   // LinearSource(memory) (side stream)
    // teacher-forced ids (dev) need every row's copy distribution; the training loss only the copy-labelled rows
    TRY(copy_score_fwd_ex(s, p.B, p.T, Sm, p.src, p.tgt, c.P + L.wres, c.P + L.bres, p.score, 1, p.mem_valid,
                          argmax_out ? nullptr : c.bt->tar_label, p.V, c.dec_off));
    TRY(linear(s, c.Td, 2, D, dec, D, c.P + L.wp, c.P + L.bp, p.gate, 2));
    TRY(head_loss(s, c.Td, p.T, p.V, Sm, p.compact_row, p.logits, p.ldl, p.score, p.mem_valid, p.gate,
                  c.bt->tar_label, loss_sum, n_tok, argmax_out, want_grad, c.row_bt));
    return 0;
}

// ------------------------------------------------------------------------------------------ backward
// How many encoder layers' weight gradients go into one grouped launch (their operands are per-layer slots).  Measured
// (profiles/r5_probes.md): the encoder's backward chain and its weight gradients are BOTH bound by the fp32 MFMA pipe, so a
// layer's grouped launch beside the next layer's chain only slowed that chain (fused GCN backward 21 -> 48 us in the step);
// ALL layers' gradients as one launch behind the chain -- beside the embedding gradients and the HBM-bound Adam of [0, split)
// -- is +2.0 % at batch 32, +0.5 % at batch 64, +0.8 % in bf16 at batch 64, and -0.4 % at batch 170, where that launch is
// ~1.5 ms long and nothing is left to hide it.  Default: all layers up to 32 768 computed node rows, per layer beyond;
// FIRA_ENC_WGRAD_EVERY=n overrides.
static inline int enc_wgrad_every(int n_rows, int n_layers) {
    static const int forced = [] { const char* e = getenv("FIRA_ENC_WGRAD_EVERY"); const int v = e ? atoi(e) : 0; return v >= 1 ? v : 0; }();
    if (forced) return forced;
    return n_rows <= 32768 ? std::max(1, n_layers) : 1;
}

static inline bool fewer_forks() {
    static const bool off = [] { const char* e = getenv("FIRA_FEWER_FORKS"); return e && e[0] == '0'; }();
    return !off;
}

// what the encoder half of the backward pass needs from the decoder half (the two halves are one call, or the two calls
// fira_train_step_begin / fira_train_step_end with the caller's collective in between)
struct BwdMid {
    hipEvent_t ev_dmem = nullptr;      // d memory is complete (auxiliary stream)
    hipEvent_t ev_groupA = nullptr;    // every gradient of [0, split) is final (weight-gradient stream)
};

static int backward_decoder(Ctx& c, int R, const int32_t* rows, hipEvent_t mid_event, BwdMid& mid) {
    Plan& p = *c.pl;
    const Layout& L = *c.L;
    hipStream_t s = c.s;
    const int D = FIRA_D, H = L.d.n_head, KV = p.nl * 2 * D, Sm = p.L + p.S;
    const fira_batch& bt = *c.bt;
    const int Nc = bt.n_nodes, Cc = bt.n_code, Mc = bt.n_mem;
    const float* dec = p.dec[p.nl - 1].x_f;
    float* G = c.G;

    gemm_group_reset();
    gemm_bf16_group_reset();
    gemm_wgrad_panel_reset();
    gemm_wgrad_panel_scratch(p.panel_scratch, p.panel_floats);
    red().reset(p.red_buf, p.red_cap);
    // ---- head: p.logits / p.score / p.gate now hold dlogits / dscore / dgate_logits -------------------------
    // The vocabulary dgrad ([R, V] x [V, 256], the head's largest product) and the copy branch are independent until
    // both land in ddec: the former runs on the side stream under the latter.
    const bool so = side_on();
    hipStream_t ss = so ? side().aux : s;
    hipEvent_t ev_dfc = nullptr;
    // dvtab_all, dW21|dc21, dtgt, ddec_c: accumulated into below, one fill for all of them -- issued on the auxiliary stream
    // at the start of the step (under the encoder's forward pass) when there is one, otherwise here
    if (c.ev_zero) TRY(main_wait(s, c.ev_zero, __LINE__));
    else TRY(zero(s, p.zero_beg, (size_t)((char*)p.zero_end - (char*)p.zero_beg)));
    if (R > 0) {
        if (so) TRY(aux_fork(s));
        // ddec_rows = dlogits W_out, split over the vocabulary axis
        if (c.wout_planes) TRY(dgrad_x3_splitk(ss, R, p.V, p.logits, p.ldl, p.WoutTX, p.ddec_c, D, g_dtype == 1));
        else
        TRY(gemm_any(ss, 0, 0, R, D, p.V, p.logits, p.ldl, c.P + L.wout, D, p.ddec_c, D, nullptr, FIRA_GEMM_ACCUM, 0,
                        nullptr));
        if (so) TRY(side_mark(&ev_dfc));
    }
    // (the vocabulary projection's weight gradient, the step's largest product, is forked HERE, under the copy branch: forked
    // behind the join of the head it ran under the decoder layers' small launches instead and the step lost 0.6 %, same box)
    if (R > 0)
        TRY(linear_wgrad(s, R, p.V, D, p.logits, p.ldl, (c.rows != nullptr && rows == c.rows) ? p.dec_c : dec, D, G + L.wout,
                         G + L.bout, PANEL_VOCAB));
    if (g_dtype == 0) TRY(rank2_rows(s, c.Td, p.gate, c.P + L.wp, p.ddec));        // ddec = dgate Wp: a rank-2 row kernel
    else TRY(linear_dgrad(s, c.Td, 2, D, p.gate, 2, c.P + L.wp, p.ddec, D, false));
    TRY(linear_wgrad_grouped(s, c.Td, 2, D, p.gate, 2, dec, D, G + L.wp, G + L.bp));
    {
        const int nb = copy_score_bwd_blocks(p.B, Sm);
        float* part = red().alloc((size_t)nb * COPY_PART_STRIDE);
        TRY(copy_score_bwd_ex(s, p.B, p.T, Sm, p.src, p.tgt, c.P + L.wres, p.score, p.dsrc, p.dtgt, G + L.wres, G + L.bres,
                              p.mem_valid, part, c.dec_off));
        if (part) {
            red().add(G + L.wres, part, D, nb, COPY_PART_STRIDE);
            red().add(G + L.bres, part + D, 1, nb, COPY_PART_STRIDE);
        }
    }
    TRY(linear_dgrad(s, c.Td, D, D, p.dtgt, D, c.P + L.wt, p.ddec, D, true));
    TRY(linear_wgrad_grouped(s, c.Td, D, D, p.dtgt, D, dec, D, G + L.wt, nullptr));
    TRY(rows_move(s, 0, Mc, D, p.dsrc_c, p.dsrc, bt.mem_dst, nullptr));
    // d memory (compact rows) = dsrc Ws + sum_l dKV_l Wkv_l: nothing reads it before the encoder's backward pass, so
    // the whole accumulation lives on the side stream (in order: this product initialises dmem_c, the per-layer
    // products of the decoder loop below add to it)
    if (so) TRY(aux_fork(s));
    TRY(linear_dgrad(ss, Mc, D, D, p.dsrc_c, D, c.P + L.ws, p.dmem_c, D, false));
    // (round 5: LinearSource's weight gradient rides in the decoder's grouped launch -- its own fork was an event record on the
    //  caller's stream right behind the auxiliary stream's; FIRA_FEWER_FORKS=0: its own fork and launch)
    if (fewer_forks()) TRY(linear_wgrad_grouped(s, Mc, D, D, p.dsrc_c, D, p.mem_c, D, G + L.ws, nullptr, 8));
    else TRY(linear_wgrad(s, Mc, D, D, p.dsrc_c, D, p.mem_c, D, G + L.ws, nullptr));
    if (R > 0) {
        if (ev_dfc) TRY(main_wait(s, ev_dfc, __LINE__));
        TRY(rows_scatter_add_idx(s, R, p.ddec_c, p.ddec, rows));
    }

    // ---- decoder layers, last to first ------------------------------------------------------------------
    // Three [Td,256] gradient buffers rotate through the layers (ddec, dT_a, dT_c): bx holds the gradient w.r.t. the layer's
    // output rows, by / bz are free.  Every LayerNorm backward writes its residual-branch gradient to a buffer OTHER than the
    // one it reads (it runs in the prologue of the product that follows it: the product's other column tiles still read dy).
    float *bx = p.ddec, *by = p.dT_a, *bz = p.dT_c;
    // The decoder's weight gradients (twelve small products per two layers + their row block of the stacked K|V weight) in launches
    // of dec_every layers beside the DECODER's chain (latency-bound, the MFMA pipe idle) instead of one launch behind it, beside the
    // first layers of the encoder's MFMA-bound chain.  Round 4 / early round 5 measured the early launches at -0.6 % (the
    // weight-gradient stream was busy with the encoder's per-layer launches afterwards anyway); with those behind the chain:
    // three layers per launch +0.9 % (fast class) / +-0 (slow class) at batch 32, +1.0 % at batch 64, two layers -1.3 %, bf16
    // neutral (profiles/r5_probes.md).  fp32, single device (data-parallel runs hand the bucket over at the mid event);
    // FIRA_DEC_WGRAD_EVERY=0: one launch behind the loop.
    static const int dec_every_env = [] { const char* e = getenv("FIRA_DEC_WGRAD_EVERY"); const int v = e ? atoi(e) : 3; return v > 0 ? v : 0; }();
    // Round 6 re-measured both exclusions.  (1) A mid event (data-parallel steps) used to force the single launch; with the panel
    // kernels that launch is the SLOW form (FIRA_DEC_WGRAD_EVERY=0 on one device: -11 % at batch 32, -17 % at batch 64), and the
    // two-call step on one device (scripts/probes/dp1_probe.py) goes 2.716 -> 2.444 ms at batch 32, 4.214 -> 3.546 ms at batch 64
    // with the launches of three layers.  Every gradient of [0, split) is still final when the mid event fires
    // (scripts/probes/mid_event_probe.py: a stream waiting for the event snapshots the slice -- no element changes afterwards).
    // Taken by every caller with a mid event: fira_train_step_begin / _begin_rows and fira_train_fwd_bwd (the ZeRO-1 and the
    // unfused trainers; FIRA_DEC_WGRAD_DP=0 = the single launch under a mid event, A/B switch).  (The ZeRO-1 test that kept the
    // second group on the single launch for a while was a ReLU tie of the golden batch, not this schedule: one hidden unit of
    // decoder layer 3 has a pre-activation of +-1e-9 in step 1 and two SINGLE-process runs disagree on its mask as often --
    // scripts/probes/nondet_probe.py, profiles/r6_probes.md.)
    // (2) bf16 mode: +0.8 % at batch 64 with the launches of three layers (23 513 -> 23 691, same-box triple); FIRA_DEC_WGRAD_BF16=0.
    static const int dec_dp = [] { const char* e = getenv("FIRA_DEC_WGRAD_DP"); return e ? atoi(e) : -1; }();
    static const bool dec_bf16_off = [] { const char* e = getenv("FIRA_DEC_WGRAD_BF16"); return e && e[0] == '0'; }();
    const bool mid_single = mid_event && dec_dp == 0;
    const int dec_every = (mid_single || (g_dtype != 0 && dec_bf16_off) || dec_every_env == 0 || p.nl % dec_every_env != 0) ? 0 : dec_every_env;
    {
    ProfScope prof_region(s, PROF_DEC_REGION, 0.0);      // wall time of the decoder's backward layers (see decoder_forward)
    TRY(lanes_fork(c));                        // (lane 1 starts behind the head's backward kernels)
    // the other streams' launches inside the loop read rows of EVERY lane: they fork from all of them
    auto aux_fork_all = [&]() -> int { return c.n_lanes > 1 ? lanes_fork_to(c, side().aux) : aux_fork(s); };
    auto flush_wgrads_all = [&]() -> int {
        if (c.n_lanes == 1) return flush_grouped_wgrads(s);
        SideStream& sd = side();
        if (!(sd.stream && sd.enabled)) return 0;
        TRY(lanes_fork_to(c, sd.stream));
        return flush_wgrad_queues(sd.stream);
    };
    for (int l = p.nl - 1; l >= 0; --l) {
        ProfDecoderTag prof_tag;               // data gradients of the M = B*30 products (the grouped wgrads flush later)
        const DecLayer& w = L.dec[l];
        DecSave& e = p.dec[l];
        DecGrad& g = p.decg[l];
        const float* x_in = l == 0 ? p.x0 : p.dec[l - 1].x_f;
        // dK|dV of this layer and of the one above it (adjacent column blocks of dkv_all / row blocks of the stacked K|V
        // weight) -> d memory, beside the chain.  TWO layers per fork: a fork costs the dependent chain ~15 us
        // (r3_event_cost.txt) and hides ~20 us of work per layer
        // (two lanes, round 6: the product forks from the point where each lane's cross-attention backward has written its rows of
        //  dkv_all -- ev_kv_done, recorded there -- not from the lanes' tails behind the whole layer: the last pair, which the
        //  encoder's backward pass waits for, then runs beside layer 0's self-attention backward instead of behind it.
        //  FIRA_DMEM_FORK_EARLY=0: from the tails, A/B switch)
        static const bool dmem_early_off = [] { const char* e = getenv("FIRA_DMEM_FORK_EARLY"); return e && e[0] == '0'; }();
        hipEvent_t ev_kv_done[2] = {nullptr, nullptr};
        auto dmem_pair = [&]() -> int {
            if (!(so && (l % 2 == 0 || l == 0))) return 0;
            const int nlay = std::min(2, p.nl - l);
            const size_t o = (size_t)l * 2 * D;
            if (c.n_lanes > 1 && ev_kv_done[0] && ev_kv_done[1]) {
                for (int k = 0; k < c.n_lanes; ++k)
                    if (hipStreamWaitEvent(ss, ev_kv_done[k], 0) != hipSuccess) return set_err("lane -> auxiliary stream fork failed");
            } else
            TRY(aux_fork_all());
            prof_decoder_tag(-1);               // a memory-row product: not one of the decoder's M = B*30 ones
            // (round 6) FIRA_DMEM_X3=0: the fp32 / bf16 GEMM launch instead of linear_x3_kacc (A/B switch)
            static const bool dmem_x3_off = [] { const char* e = getenv("FIRA_DMEM_X3"); return e && e[0] == '0'; }();
            const int rc_kv = (c.kv_planes && !dmem_x3_off)
                    ? linear_x3_kacc(ss, Mc, p.dkv_all + o, p.kvp, p.WkvTX + (size_t)l * 2 * 3 * D * D, nlay * 2, p.dmem_c, D, true, g_dtype == 1)
                    : linear_dgrad(ss, Mc, nlay * 2 * D, D, p.dkv_all + o, p.kvp, c.P + L.wkv_all + o * D, p.dmem_c, D, true);
            prof_decoder_tag(+1);
            return rc_kv;
        };
        for (int k = 0; k < c.n_lanes; ++k) {
            const Lane& ln = c.lanes[k];
            hipStream_t ls = ln.s;
            const size_t r0 = (size_t)ln.r0;
            const int nr = ln.nr;
            const uint32_t idx0 = (uint32_t)(r0 * D);
            const int32_t* q_off = c.dec_off ? c.dec_off + ln.b0 : nullptr;
            const bool q = k == 0;             // the weight gradients read the rows of all lanes: queued once
            // FeedForward (gnn_transformer.py:170-174): LayerNorm backward + d hidden = (dYf W2) masked by the saved activation > 0
            // (ReLU backward in the GEMM epilogue)
            TRY(ln_bwd_dgrad(ls, nr, p.F, bx + r0 * D, e.s_f + r0 * D, e.st_f + r0 * 2, c.P + w.lnf_g, by + r0 * D, g.dYf + r0 * D,
                             G + w.lnf_g, G + w.lnf_b, c.p_drop, c.seed, site(l, SITE_FFN), c.P + w.w2, p.F, g.dh + r0 * p.F, p.F,
                             e.h + r0 * p.F, idx0));
            if (q) TRY(linear_wgrad_grouped(s, c.Td, D, p.F, g.dYf, D, e.h, p.F, G + w.w2, G + w.b2));
            if (q) TRY(linear_wgrad_grouped(s, c.Td, p.F, D, g.dh, p.F, e.x_c, D, G + w.w1, G + w.b1));
            TRY(linear_dgrad(ls, nr, p.F, D, g.dh + r0 * p.F, p.F, c.P + w.w1, by + r0 * D, D, true));               // by = d x_c
            // cross attention: LayerNorm backward + d ao2 = dYc Wo_c
            TRY(ln_bwd_dgrad(ls, nr, D, by + r0 * D, e.s_c + r0 * D, e.st_c + r0 * 2, c.P + w.lnc_g, bz + r0 * D, g.dYc + r0 * D,
                             G + w.lnc_g, G + w.lnc_b, c.p_drop, c.seed, site(l, SITE_CROSS), c.P + w.wo_c, D, bx + r0 * D, D, nullptr,
                             idx0));                                                                              // bx = d ao2
            if (q) TRY(linear_wgrad_grouped(s, c.Td, D, D, g.dYc, D, e.ao2, D, G + w.wo_c, G + w.bo_c));
            // (ragged key rows: dkv_all holds the computed memory rows only, every one of them written by this launch)
            TRY(attention_bwd(ls, ln.nb, H, p.T, Sm, e.qc, D, p.kv_all + l * 2 * D, p.kvp, p.kv_all + l * 2 * D + D, p.kvp,
                              p.mem_valid_c, 0, 0, e.ao2, D, bx, D, g.dq, D, p.dkv_all + l * 2 * D, p.kvp,
                              p.dkv_all + l * 2 * D + D, p.kvp, q_off, 0, attn_bf16(), p.mem_off + ln.b0));
            if (c.n_lanes == 1) TRY(dmem_pair());
            else if (!dmem_early_off && so && l % 2 == 0 && k < 2) {
                ev_kv_done[k] = side().ev();
                if (hipEventRecord(ev_kv_done[k], ls) != hipSuccess) return set_err("lane event record failed");
            }
            if (q) TRY(linear_wgrad_grouped(s, c.Td, D, D, g.dq, D, e.x_a, D, G + w.wq_c, G + w.bq_c));
            TRY(linear_dgrad(ls, nr, D, D, g.dq + r0 * D, D, c.P + w.wq_c, bz + r0 * D, D, true));                 // bz = d x_a
            // self attention: LayerNorm backward + d ao = dYs Wo_s
            TRY(ln_bwd_dgrad(ls, nr, D, bz + r0 * D, e.s_a + r0 * D, e.st_a + r0 * 2, c.P + w.lns_g, by + r0 * D, g.dYs + r0 * D,
                             G + w.lns_g, G + w.lns_b, c.p_drop, c.seed, site(l, SITE_SELF), c.P + w.wo_s, D, bx + r0 * D, D, nullptr,
                             idx0));                                                                              // bx = d ao
            if (q) TRY(linear_wgrad_grouped(s, c.Td, D, D, g.dYs, D, e.ao, D, G + w.wo_s, G + w.bo_s));
            TRY(attention_bwd(ls, ln.nb, H, p.T, p.T, e.qkv, 3 * D, e.qkv + D, 3 * D, e.qkv + 2 * D, 3 * D,
                              p.tar_valid + (size_t)ln.b0 * p.T, 1, 0, e.ao, D, bx, D, g.dqkv, 3 * D, g.dqkv + D, 3 * D, g.dqkv + 2 * D,
                              3 * D, q_off, 1, attn_bf16()));
            if (q) TRY(linear_wgrad_grouped(s, c.Td, 3 * D, D, g.dqkv, 3 * D, x_in, D, G + w.wqkv, G + w.bqkv));
            TRY(linear_dgrad(ls, nr, 3 * D, D, g.dqkv + r0 * 3 * D, 3 * D, c.P + w.wqkv, by + r0 * D, D, true));   // by = d x_in
        }
        if (c.n_lanes > 1) TRY(dmem_pair());   // (behind the layer of every lane: dkv_all's column block is complete)
        if (dec_every > 0 && l % dec_every == 0 && side().stream && side().enabled) {
            // the weight gradients of the last dec_every layers: one grouped launch + their row block of the stacked K|V weight
            const int nlay = std::min(dec_every, p.nl - l);
            const size_t o = (size_t)l * 2 * D;
            prof_decoder_tag(-1);               // weight gradients: not among the decoder's forward / data-gradient products
            int rc_w = flush_wgrads_all();
            if (!rc_w)
                rc_w = wgrad_on(side().stream, PANEL_KV, Mc, nlay * 2 * D, D, p.dkv_all + o, p.kvp, p.mem_c, D, G + L.wkv_all + o * D,
                                G + L.bkv_all + o);
            prof_decoder_tag(+1);
            TRY(rc_w);
        }
        float* t = bx; bx = by; by = t;         // the next layer's output gradient is in (the old) by; bz stays free
    }
    TRY(lanes_join(c));
    }
    const float* dy = bx;
    TRY(flush_grouped_wgrads(s));                                  // the decoder's and the head's small weight gradients
    // decoder embedding.  The table has no padding_idx (gnn_transformer.py:92-93), but rows of padded target positions
    // carry an exactly-zero gradient (never attended as keys, zero loss weight): skipping id 0 only drops the
    // hundreds of serialised atomic additions of 0.0 onto table row 0.
    // (nothing on the chain reads it: behind the fork of the grouped launch above, on the weight-gradient stream; dy is a
    // decoder-only buffer, untouched until the next step)
    {
        hipStream_t es = (side().stream && side().enabled) ? side().stream : s;
        if (c.row_bt) TRY(embed_rows_bwd(es, c.Td, c.row_bt, c.bt->tar, G + L.dec_emb, dy, 0));
        else TRY(embed_gather_bwd(es, p.B, p.T, c.bt->tar, G + L.dec_emb, dy, p.T, 0, 0));
    }
    // cross-attention K|V projections of all layers (computed memory rows only)
    hipEvent_t& ev_dmem = mid.ev_dmem;
    if (so) {
        TRY(side_mark(&ev_dmem));                // dmem_c is complete at this point of the auxiliary stream
    } else {
        TRY(linear_dgrad(s, Mc, KV, D, p.dkv_all, p.kvp, c.P + L.wkv_all, p.dmem_c, D, true));
    }
    // (round 5: no fork of its own -- the weight-gradient stream waited for the caller's stream at the grouped launch above, and
    //  dkv_all was complete by then)
    if (dec_every > 0 && side().stream && side().enabled) {
        // (the K|V weight gradient went out in row blocks inside the loop)
    } else if (fewer_forks() && side().stream && side().enabled)
        TRY(wgrad_on(side().stream, PANEL_KV, Mc, KV, D, p.dkv_all, p.kvp, p.mem_c, D, G + L.wkv_all, G + L.bkv_all));
    else
    TRY(linear_wgrad(s, Mc, KV, D, p.dkv_all, p.kvp, p.mem_c, D, G + L.wkv_all, G + L.bkv_all, PANEL_KV));
    // decoder LayerNorms, copy head: their partial rows were written before the fork of the weight gradient above, and only
    // the end of the step (or the mid-event below, which waits for this stream) reads the sums: off the dependent chain
    TRY(deferred_reduce(side().stream && side().enabled ? side().stream : s, red().tab));
    // every gradient of [0, split) is final once the weight-gradient stream has passed this point (its launches above: the
    // vocabulary / copy / decoder weight gradients, the decoder embedding, the stacked K|V weight, the deferred column sums;
    // the auxiliary stream writes data gradients only)
    hipEvent_t& ev_groupA = mid.ev_groupA;
    if (c.adam && side().stream && side().enabled) {
        ev_groupA = side().ev();
        if (hipEventRecord(ev_groupA, side().stream) != hipSuccess) return set_err("group-A mark failed");
    }
    if (mid_event) {                         // gradients of [0, split) are final from here on
        // The event fires when BOTH the caller's stream and the weight-gradient stream have reached this point, without
        // holding up either of them: the auxiliary stream (idle for the rest of the backward pass) waits for the two and
        // records it.  (Joining the weight-gradient stream into the caller's stream here made the encoder's backward wait
        // for the vocabulary projection's weight gradient.)
        SideStream& sd = side();
        hipStream_t es = s;
        if (sd.stream && sd.enabled) {
            TRY(aux_fork(s));
            hipEvent_t e2 = sd.ev();
            if (hipEventRecord(e2, sd.stream) != hipSuccess || hipStreamWaitEvent(sd.aux, e2, 0) != hipSuccess)
                return set_err("mid-event: weight-gradient stream mark failed");
            es = sd.aux;
        }
        hipError_t e = hipEventRecord(mid_event, es);
        if (e != hipSuccess) return set_err("hipEventRecord: %s", hipGetErrorString(e));
    }
    return 0;
}

static int backward_encoder(Ctx& c, BwdMid& mid) {
    Plan& p = *c.pl;
    const Layout& L = *c.L;
    hipStream_t s = c.s;
    const int D = FIRA_D;
    const fira_batch& bt = *c.bt;
    const int Nc = bt.n_nodes, Cc = bt.n_code, Mc = bt.n_mem;
    float* G = c.G;
    hipEvent_t ev_dmem = mid.ev_dmem, ev_groupA = mid.ev_groupA;
    // ---- encoder layers, last to first (compact node rows) -------------------------------------------------
    float* dXn = p.dXa;
    float* other = p.dXb;
    UnfoldTable unfold_tab;                  // GCN layers whose dc comes out of the deferred reduction at the end
    // (round 5) FIRA_UNFOLD_LATE=0: the folded weight's two unfold products per layer as two launches behind that layer's grouped
    // weight gradients (A/B switch); default: all layers in one launch behind the last group
    static const bool unfold_late_off = [] { const char* e = getenv("FIRA_UNFOLD_LATE"); return e && e[0] == '0'; }();
    const bool unfold_late = !unfold_late_off && enc_group_on() && p.nl <= 16;
    // The per-layer unfold reads dW21 right behind the layer's grouped launch: without the late launch (the switch, or more
    // than 16 layers) every layer flushes its own group -- a group spanning several layers would leave dW21 unwritten
    // (still the zero fill) when the layer's unfold products run.
    const int wgrad_every = unfold_late ? enc_wgrad_every(Nc, p.nl) : 1;
    const float *uf_dW21[16], *uf_W1[16], *uf_W2[16];
    float *uf_dW1[16], *uf_dW2[16];
    int uf_n = 0;
    if (!c.ev_zero) TRY(zero(s, dXn, (size_t)Nc * D * sizeof(float)));   // AST/edit rows of the last layer feed nothing
    if (ev_dmem) TRY(main_wait(s, ev_dmem, __LINE__));
    TRY(rows_move(s, 1, Mc, D, dXn, p.dmem_c, nullptr, bt.mem_rows));
    for (int l = p.nl - 1; l >= 0; --l) {
        const EncLayer& w = L.enc[l];
        EncSave& e = p.enc[l];
        EncGrad& g = p.encg[l];
        // (the LayerNorm backward also leaves the column sums of dY the folded form needs: db2 and dc, deferred)
        float* dc21 = p.dc21 + (size_t)l * D;
        bool sums = false;
        const bool room = unfold_tab.n < 16;     // (asked for only when the closing unfold launch can take the layer)
        TRY(ln_bwd(s, Nc, dXn, e.s2, e.st2, c.P + w.ln2g, other, g.dY2, G + w.ln2g, G + w.ln2b, c.p_gcn,
                              c.seed, site(l, SITE_GCN), nullptr, room ? p.rsum : nullptr, G + w.fc2b, dc21, &sums));
        if (sums) unfold_tab.e[unfold_tab.n++] = UnfoldEntry{c.P + w.fc2w, c.P + w.fc1b, dc21, G + w.fc2w, G + w.fc1b};
        // GCN, folded form (see encoder_forward): Y = U W21^T + r c^T + b2 with U = A_hat X
        //   weight space: dW21 += dY^T U (+ db2 by the fused column sums), dc += dY^T r           (side stream)
        //   data space:   dU = dY W21, dX += A_hat dU                                              (main stream)
        //   and back to the reference's parameters: dW2 += dW21 W1^T + dc b1^T, dW1 += W2^T dW21, db1 += W2^T dc
        float* dW21 = p.dW21 + (size_t)l * D * D;
        const bool grouped = enc_group_on();
        // back to the reference's parameters (reads dW21: after its weight gradient on the same stream)
        bool flushed = false;                    // this layer's grouped launch went out (behind a fork from the caller's stream)
        auto unfold = [&]() -> int {
            hipStream_t ws = s;
            if (side().stream && side().enabled) {
                // (dY2 / dW21's operands are written on the caller's stream: whatever reads them over there needs a fork behind
                //  them -- the grouped flush of this layer was one)
                if (!grouped || (!flushed && (!sums || !unfold_late))) TRY(side_fork(s));
                ws = side().stream;
            }
            if (!sums) TRY(colsum(ws, Nc, D, g.dY2, D, dc21, p.rsum));
            if (unfold_late) {                       // the two products of every layer: ONE launch behind the last layer's group
                uf_dW21[uf_n] = dW21; uf_W1[uf_n] = c.P + w.fc1w; uf_W2[uf_n] = c.P + w.fc2w;
                uf_dW1[uf_n] = G + w.fc1w; uf_dW2[uf_n] = G + w.fc2w;
                ++uf_n;
            } else {
            TRY(gemm_f32_ex(ws, 0, 1, D, D, D, dW21, D, c.P + w.fc1w, D, G + w.fc2w, D, nullptr, FIRA_GEMM_ACCUM, 0, nullptr));
            TRY(gemm_f32_ex(ws, 1, 0, D, D, D, c.P + w.fc2w, D, dW21, D, G + w.fc1w, D, nullptr, FIRA_GEMM_ACCUM, 0, nullptr));
            }
            if (!sums) TRY(gcn_bias_unfold(ws, c.P + w.fc2w, c.P + w.fc1b, dc21, G + w.fc2w, G + w.fc1b));
            return 0;
        };
        if (gcn_fused_on()) {
            // one launch: V = A_hat dY (stored in e.Z, which the fused forward pass does not use), other = ds + V W21.
            // The weight gradient follows from the same V: dW21 = dY^T (A_hat X) = (A_hat dY)^T X = V^T X
            if (gcn_fused_bwd_on()) {
                TRY(gcn_fused_bwd(s, Nc, bt.rowptr, bt.col, bt.val, g.dY2, p.W21 + (size_t)l * D * D, e.Z, other, g_dtype == 1,
                                  gcn_x3_on(p.nl) ? p.W21tx + (size_t)l * 3 * D * D : nullptr));
            } else {
                TRY(csr_spmm_ex(s, Nc, bt.rowptr, bt.col, bt.val, g.dY2, D, e.Z, D, 0, 1, 0, nullptr));            // V
                TRY(linear_dgrad(s, Nc, D, D, e.Z, D, p.W21 + (size_t)l * D * D, other, D, true));                 // other += V W21
            }
            TRY(enc_wgrad(s, Nc, D, D, e.Z, D, p.X[l], D, dW21, nullptr));
            if (!sums) TRY(colsum(s, Nc, D, g.dY2, D, G + w.fc2b));     // (db2 = column sums of dY, not of V)
            if (!grouped) TRY(unfold());
        } else {
        TRY(enc_wgrad(s, Nc, D, D, g.dY2, D, e.Z, D, dW21, sums ? nullptr : G + w.fc2b));
        if (!grouped) TRY(unfold());
        TRY(linear_dgrad(s, Nc, D, D, g.dY2, D, p.W21 + (size_t)l * D * D, p.dNB2, D, false));  // dU
        TRY(csr_spmm_ex(s, Nc, bt.rowptr, bt.col, bt.val, p.dNB2, D, other, D, 0, 1, 1, nullptr));   // other = ds + A_hat dU
        }
        // Combination on the code rows, in place inside `other` through the code-row map: the LayerNorm backward reads
        // dG[code rows] and leaves the residual-branch gradient there; the q|k projection's dgrad adds to the same rows
        bool comb_done = false;
        if (comb_fused_bwd_on()) {
            // one launch (comb_fused.hip): LayerNorm backward, dgrad through Wo, gate backward, dgrad through Wq | Wk
            const int nb = comb_fused_bwd_parts();
            // (both blocks or neither: six table rows and nb * 6 * D floats asked for at once)
            float* part_ln = red().alloc((size_t)nb * 6 * D, 6);
            float* part_v = part_ln ? part_ln + (size_t)nb * 2 * D : nullptr;
            if (part_v) {
                TRY(comb_fused_bwd(s, Cc, other, bt.code_rows, e.s1, e.st1, c.P + w.ln1g, c.P + w.wo, c.P + w.wqk, e.qk,
                                   p.vtab_all + l * D, p.nl * D, bt.code_mark, g.dYc, g.dqk, part_ln, part_v, c.p_drop, c.seed,
                                   site(l, SITE_GATE), site(l, SITE_COMB_OUT), g_dtype == 1,
                                   comb_x3_on(p.nl) ? p.WcTX + (size_t)l * 9 * D * D : nullptr));
                red().add(G + w.ln1g, part_ln, D, nb, 2 * D);
                red().add(G + w.ln1b, part_ln + D, D, nb, 2 * D);
                for (int k = 0; k < 4; ++k)
                    red().add(p.dvtab_all + (size_t)k * p.nl * D + l * D, part_v + k * D, D, nb, 4 * D);
                TRY(enc_wgrad(s, Cc, D, D, g.dYc, D, e.c, D, G + w.wo, G + w.bo));
                comb_done = true;
            }
        }
        if (!comb_done) {
        TRY(ln_bwd(s, Cc, other, e.s1, e.st1, c.P + w.ln1g, other, g.dYc, G + w.ln1g, G + w.ln1b, c.p_drop,
                              c.seed, site(l, SITE_COMB_OUT), bt.code_rows));
        TRY(enc_wgrad(s, Cc, D, D, g.dYc, D, e.c, D, G + w.wo, G + w.bo));
        TRY(linear_dgrad(s, Cc, D, D, g.dYc, D, c.P + w.wo, p.dCB_a, D, false));               // d c
        {
            const int nb = combination_bwd_blocks(Cc);
            float* part = red().alloc((size_t)nb * 4 * D);
            TRY(combination_bwd(s, Cc, e.qk, p.vtab_all + l * D, p.nl * D, bt.code_mark, p.dCB_a, g.dqk,
                                p.dvtab_all + l * D, p.nl * D, c.p_drop, c.seed, site(l, SITE_GATE), part));
            if (part)
                for (int k = 0; k < 4; ++k)
                    red().add(p.dvtab_all + (size_t)k * p.nl * D + l * D, part + k * D, D, nb, 4 * D);
        }
        }
        TRY(enc_wgrad(s, Cc, 2 * D, D, g.dqk, 2 * D, e.Xc, D, G + w.wqk, G + w.bqk));
        if (grouped) {                           // the layer's three weight gradients: one fork, one launch, then the unfold
            // (one launch for every n layers' gradients: enc_wgrad_every)
            if ((p.nl - l) % wgrad_every == 0 || l == 0) { TRY(flush_grouped_wgrads(s)); flushed = true; }
            TRY(unfold());
        }
        if (!comb_done)
            TRY(gemm_any(s, 0, 0, Cc, D, 2 * D, g.dqk, 2 * D, c.P + w.wqk, D, other, D, nullptr, FIRA_GEMM_ACCUM, 0, nullptr,
                         bt.code_rows));                                                       // other = dX[l]
        float* tmp = dXn; dXn = other; other = tmp;
    }
    if (uf_n > 0) TRY(gcn_unfold_products(side().stream, uf_n, uf_dW21, uf_W1, uf_W2, uf_dW1, uf_dW2));   // (in order behind the last group)
    // encoder LayerNorms, dvtab_all and the two products that read it: beside the embedding kernels below, on the auxiliary
    // stream (idle since the decoder's backward pass)
    hipEvent_t ev_tail = nullptr;
    int64_t adam_b0 = L.split;               // fira_train_step: first parameter the closing Adam launch still has to update
    {
        const bool ax = side_on();
        hipStream_t rs = ax ? side().aux : s;
        if (ax) TRY(aux_fork(s));
        TRY(deferred_reduce(rs, red().tab));
        // value projection of the mark table: vtab_all = mark_emb W2_all^T + b2_all
        TRY(gemm_f32_ex(rs, 1, 0, p.nl * D, D, 4, p.dvtab_all, p.nl * D, c.P + L.mark_emb, D, G + L.w2_all, D, nullptr,
                        FIRA_GEMM_ACCUM, 1, G + L.b2_all));
        // rows 1..3 only: padding_idx row 0 never gets a gradient (its slot of the zeroed gradient buffer stays untouched)
        // (a 3-row product over K = nl * 256: split over K -- as one latency-kernel chain it was a 48 us launch, the longest of
        //  the auxiliary stream's tail)
        TRY(gemm_f32_ex(rs, 0, 0, 3, D, p.nl * D, p.dvtab_all + (size_t)p.nl * D, p.nl * D, c.P + L.w2_all, D, G + L.mark_emb + D, D,
                        nullptr, FIRA_GEMM_ACCUM, std::max(2, p.nl), nullptr));
        if (ax) TRY(side_mark(&ev_tail));
    }
    // embeddings (padding_idx = 0 on all three encoder tables: gnn_transformer.py:32-39)
    if (bt.emb_item_tok && bt.emb_item_ptr && bt.emb_rows && bt.ast_rows && bt.ast_ids && L.d.ast_vocab <= 128) {
        // the batch lists its id-carrying nodes by compact row: the gradient is read where the backward pass left it
        TRY(embed_grouped_bwd(s, bt.n_emb_items, bt.emb_item_tok, bt.emb_item_ptr, bt.emb_rows, G + L.emb, dXn));
        TRY(embed_list_bwd_small(s, bt.n_ast_items, bt.ast_rows, bt.ast_ids, G + L.ast_emb, dXn, L.d.ast_vocab));
    } else {                                                       // id arrays only: back to the dense [B,650,256] layout
        TRY(zero(s, p.H, (size_t)p.NB * D * sizeof(float)));
        TRY(rows_move(s, 1, Nc, D, p.H, dXn, nullptr, bt.node_rows));
        TRY(embed_gather_bwd(s, p.B, p.L, bt.sou, G + L.emb, p.H, p.N, 0, 0));
        TRY(embed_gather_bwd(s, p.B, p.S, bt.sub_token, G + L.emb, p.H, p.N, p.L, 0));
        if (L.d.ast_vocab <= 128)
            TRY(embed_gather_bwd_small(s, p.B, p.A, bt.ast_change, G + L.ast_emb, p.H, p.N, p.L + p.S, 0, L.d.ast_vocab));
        else
            TRY(embed_gather_bwd(s, p.B, p.A, bt.ast_change, G + L.ast_emb, p.H, p.N, p.L + p.S, 0));
    }
    if (c.adam) {
        // INVARIANT this early update rests on (tests/test_model_gpu.py::test_one_call_step_equals_backward_then_adam
        // compares with the two-call path in fp32 and bf16): behind the decoder's backward pass NOTHING, on any stream, reads a
        // parameter of [0, mark_emb) -- the encoder's backward kernels read encoder weights of [mark_emb, live) and workspace
        // copies (W21, WcT, the bf16 shadows), the weight-gradient stream reads activations only, the auxiliary stream's reads of
        // wkv_all / ws precede ev_dmem, which the caller's stream has waited for -- and the gradients of [0, split) are final at
        // ev_groupA, those of the two embedding tables when the two launches above have run.  A new kernel that breaks either
        // half must move this update behind the final join.
        // Adam on [0, split) (69 % of the live parameters) HERE: the caller's stream would otherwise stand waiting for the last
        // grouped weight gradient / the unfold products / the deferred reductions of the other two streams (timeline: 30-60 us)
        // and then run the whole update alone.  Nothing enqueued after the decoder's backward pass reads a parameter of
        // [0, split): the encoder's backward kernels read encoder weights, the weight-gradient stream reads activations.
        const fira_adam_opts& ad = *c.adam;
        if (c.adam_a_only) {
            // data parallel: the bucket's all-reduce (enqueued by the caller between the two calls) has passed ev_early; the
            // normaliser is the all-reduced token count
            if (c.ev_early) TRY(main_wait(s, c.ev_early, __LINE__));
            // (row-sparse tables, data parallel: decoder.embedding -- the head of this slice -- by the rows of the ALL-REDUCED
            //  gradient that are not zero: the union of the ranks' touched rows, the same on every rank)
            const int64_t a0 = c.row_step ? L.dec_emb + (int64_t)L.d.vocab * D : 0;
            if (c.count) TRY(adam_step(s, L.split - a0, c.Pw + a0, G + a0, ad.m + a0, ad.v + a0, ad.lr, ad.beta1, ad.beta2, ad.eps, ad.step, c.count, 1));
            else TRY(adam_step_mb(s, L.split - a0, c.Pw + a0, G + a0, nullptr, ad.m + a0, ad.v + a0, ad.lr, ad.beta1, ad.beta2, ad.eps, ad.step, c.n_tok, nullptr));
            if (c.row_step)
                TRY(adam_rows_step(s, adam_rows_tables(L, c.Pw, ad, c.row_step), G, ad.lr, ad.beta1, ad.beta2, ad.eps, ad.step,
                                   c.count ? nullptr : c.n_tok, c.count, 1));
        } else {
        if (ev_groupA) TRY(main_wait(s, ev_groupA, __LINE__));
        // (row-sparse tables: decoder.embedding is the head of [0, split) -- layout.cpp -- and is left to the rows launch below)
        const int64_t a0 = c.row_step ? L.dec_emb + (int64_t)L.d.vocab * D : 0;
        TRY(adam_step_mb(s, L.split - a0, c.Pw + a0, G + a0, nullptr, ad.m + a0, ad.v + a0, ad.lr, ad.beta1, ad.beta2, ad.eps,
                         ad.step, c.n_tok, nullptr));
        }
        // ... and the encoder's two embedding tables (the head of group B: layout.cpp), whose gradients the two launches above
        // on this stream have just completed -- also ahead of the join
        static const bool emb_early_off = [] { const char* e = getenv("FIRA_ADAM_EMB_EARLY"); return e && e[0] == '0'; }();   // A/B switch
        if (!emb_early_off && !c.adam_a_only) adam_b0 = L.mark_emb;
        if (c.row_step && !c.adam_a_only) {
            // both vocabulary-sized tables as ONE launch over the rows whose gradient row is not zero (the gradient rows are
            // inspected: 2 x 25 MB read instead of 2 x 177 MB moved); the 71-row table joins the closing launch
            TRY(adam_rows_step(s, adam_rows_tables(L, c.Pw, ad, c.row_step), G, ad.lr, ad.beta1, ad.beta2, ad.eps, ad.step,
                               c.n_tok, nullptr));
            adam_b0 = L.ast_emb;
        } else
        if (adam_b0 > L.split)
        TRY(adam_step_mb(s, adam_b0 - L.split, c.Pw + L.split, G + L.split, nullptr, ad.m + L.split, ad.v + L.split, ad.lr, ad.beta1,
                         ad.beta2, ad.eps, ad.step, c.n_tok, nullptr));
    }
    if (ev_tail) TRY(main_wait(s, ev_tail, __LINE__));
    if (side().stream && side().enabled) TRY(side_join(s, __LINE__));        // every weight gradient is complete past this point
    // dc of every GCN layer is final (deferred reduction above) and so are the side stream's additions to dW2: back to the
    // reference's fc2.weight / fc1.bias gradients, one launch for all layers
    TRY(gcn_bias_unfold_all(s, unfold_tab));
    if (c.adam && !c.adam_a_only) {
        const fira_adam_opts& ad = *c.adam;
        TRY(adam_step_mb(s, L.live - adam_b0, c.Pw + adam_b0, G + adam_b0, nullptr, ad.m + adam_b0, ad.v + adam_b0, ad.lr, ad.beta1,
                         ad.beta2, ad.eps, ad.step, c.n_tok, nullptr));
    }
    return 0;
}

static int backward(Ctx& c, int R, const int32_t* rows, hipEvent_t mid_event) {
    BwdMid mid;
    TRY(backward_decoder(c, R, rows, mid_event, mid));
    return backward_encoder(c, mid);
}

static int check_batch(const fira_batch* b) {
    FIRA_REQUIRE(b && b->B > 0, "empty batch");
    FIRA_REQUIRE(b->sou && b->mark && b->ast_change && b->sub_token && b->rowptr && b->col && b->val,
                 "batch has null encoder inputs");
    FIRA_REQUIRE(b->node_rows && b->code_rows && b->code_mark && b->mem_rows && b->mem_dst,
                 "batch has null node lists (node_rows / code_rows / code_mark / mem_rows / mem_dst)");
    FIRA_REQUIRE(b->n_nodes > 0 && b->n_code > 0 && b->n_mem > 0 && b->n_code <= b->n_nodes && b->n_mem <= b->n_nodes,
                 "inconsistent node counts %d / %d / %d", b->n_nodes, b->n_code, b->n_mem);
    note_graph_density(b);
    return 0;
}
static int check_counts(const fira_batch* b, const Plan& p) {
    FIRA_REQUIRE(b->n_nodes <= p.NB && b->n_code <= p.CB && b->n_mem <= p.MB, "node lists exceed the batch geometry");
    return 0;
}

// ------------------------------------------------------------------------------------------ decode state
struct DecodePlan {
    Plan enc;
    int BR;
    float *kc[2], *vc[2];
    int32_t* hist[2];
    float *x, *q, *qkv, *ao, *s, *xa, *qc, *xc, *h, *tgt, *score, *gate, *logits;
    float* wsT;                      // (round 6) k-major copies of every layer's self-attention fc_o and cross-attention fc_q weights
    uint16_t* kv16;                  // optional bf16 copy of the cross K|V rows of all layers (FIRA_DECODE_KV_BF16)
    // flags: FIRA_DECODE_KV_BF16 reserves the bf16 copy of the cross K|V (148 MB at batch 64); without it nothing is reserved
    size_t build(void* ws, const fira_dims& d, int B, int n_beam, int flags = 0) {
        size_t used = enc.build(ws, d, B, false);
        Arena a(ws ? (char*)ws + used : nullptr);
        BR = B * n_beam;
        const size_t D = FIRA_D, T = d.tar_len, cache = (size_t)d.n_layer * BR * T * D;
        for (int i = 0; i < 2; ++i) {
            kc[i] = a.f(cache); vc[i] = a.f(cache);
            hist[i] = a.get<int32_t>((size_t)BR * T);
        }
        x = a.f(BR * D); q = a.f(BR * D); qkv = a.f(BR * 3 * D); ao = a.f(BR * D); s = a.f(BR * D); xa = a.f(BR * D); qc = a.f(BR * D);
        xc = a.f(BR * D); h = a.f((size_t)BR * d.d_ff); tgt = a.f(BR * D);
        score = a.f((size_t)BR * (d.sou_len + d.sub_len)); gate = a.f((size_t)BR * 2);
        logits = a.f((size_t)BR * enc.ldl);
        wsT = a.f((size_t)d.n_layer * 2 * D * D);
        kv16 = (flags & FIRA_DECODE_KV_BF16) ? a.get<uint16_t>((size_t)enc.MB * enc.kvp) : nullptr;      // (last: the other offsets do not move)
        return used + a.used;
    }
};

}  // namespace fira

using namespace fira;

extern "C" {

size_t fira_workspace_bytes(const fira_dims* d, int B, int mode) {
    if (!get_layout(d) || B <= 0) return 0;
    Plan p;
    return p.build(nullptr, *d, B, mode == 1);
}
size_t fira_decode_workspace_bytes(const fira_dims* d, int B, int n_beam) {
    return fira_decode_workspace_bytes_ex(d, B, n_beam, 0);
}
size_t fira_decode_workspace_bytes_ex(const fira_dims* d, int B, int n_beam, int flags) {
    if (!get_layout(d) || B <= 0 || n_beam <= 0 || (flags & ~FIRA_DECODE_KV_BF16)) return 0;
    DecodePlan dp;
    return dp.build(nullptr, *d, B, n_beam, flags);
}

// fira_train_step_begin leaves its step here (per thread) for fira_train_step_end: the plan of the workspace, the call's
// context (a copy of the batch descriptor: its arrays must stay valid until the second call) and what the two halves of the
// backward pass share.  The dtype / shadow scopes of the first call are re-entered by the second.
struct PendingStep {
    bool active = false;
    Plan plan;
    fira_batch batch;
    Ctx ctx{};
    BwdMid mid;
    int dtype = 0;
    const float* params = nullptr;
    const ShadowTable* tab = nullptr;
    const float* W21 = nullptr;
    int64_t W21n = 0;
    const uint16_t *W21b = nullptr, *W21bT = nullptr;
};
static thread_local PendingStep g_pending;

static int train_call(void* stream, const fira_dims* d, const fira_batch* batch, const float* params, float* grads,
                      void* workspace, size_t workspace_bytes, const fira_train_opts* opts, float* loss_sum,
                      int32_t* n_tok, void* mid_event, float* params_w, const fira_adam_opts* adam, bool begin_only = false,
                      int32_t* row_step = nullptr) {
    const Layout* L = get_layout(d);
    if (!L) return 1;
    TRY(check_batch(batch));
    FIRA_REQUIRE(batch->tar && batch->tar_label, "training batch needs tar and tar_label");
    FIRA_REQUIRE(params && grads && workspace && loss_sum && n_tok, "null pointer argument");
    Plan p;
    const size_t need = p.build(workspace, *d, batch->B, true);
    FIRA_REQUIRE(need <= workspace_bytes, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    TRY(check_counts(batch, p));
    Ctx c{(hipStream_t)stream, L, batch, params, grads, &p, opts ? opts->dropout : 0.f, opts ? opts->gcn_dropout : 0.f,
          opts ? opts->seed : 0};
    FIRA_REQUIRE(c.p_drop >= 0.f && c.p_drop < 1.f && c.p_gcn >= 0.f && c.p_gcn < 1.f, "dropout must be in [0,1)");
    FIRA_REQUIRE(!opts || opts->dtype == 0 || opts->dtype == 1, "fira_train_opts.dtype must be 0 (fp32) or 1 (bf16)");
    DtypeScope dtype_scope(opts ? opts->dtype : 0);
    TRY(side().init());
    const bool bf16 = opts && opts->dtype == 1;
    const ShadowTable* tab = bf16 ? shadow_table(*L) : nullptr;
    ShadowScope shadow_scope(params, L->total, bf16 ? p.wb : nullptr, bf16 ? p.wbt : nullptr, tab);
    // FIRA_SHADOWS_LATE=0: the shadow refresh at the head of the caller's stream, as before round 5 (A/B switch)
    static const bool shadows_late_off = [] { const char* e = getenv("FIRA_SHADOWS_LATE"); return e && e[0] == '0'; }();
    const bool defer_sh = bf16 && !shadows_late_off && side_on() && gcn_fused_on() && comb_fused_on() && L->d.n_layer <= 8;
    if (bf16 && !defer_sh) TRY(weight_shadows(c.s, *tab, params, p.wb, p.wbt));
    if (defer_sh) c.shadow_tab = tab;
    // computed target rows: the decoder / head run on the prefix rows the batch lists (fira_batch.dec_off)
    c.Td = p.TB;
    if (opts && opts->compact_dec && batch->dec_off) {
        FIRA_REQUIRE(batch->n_dec_rows >= batch->B && batch->n_dec_rows <= p.TB, "bad n_dec_rows %d", batch->n_dec_rows);
        c.Td = batch->n_dec_rows;
        c.dec_off = batch->dec_off;
        c.row_bt = p.row_bt;
    }
    int R = c.Td;
    const int32_t* rows = p.iota;
    if (opts && opts->compact_head && batch->head_rows) {
        R = batch->n_head_rows;
        FIRA_REQUIRE(R >= 0 && R <= c.Td, "bad n_head_rows %d", R);
        c.rows_dense = batch->head_rows;
        rows = c.dec_off ? p.rows_c : batch->head_rows;        // in the indexing of the rows the decoder computes
        c.R = R;
        c.rows = rows;
    }
    c.loss_sum = loss_sum;
    c.n_tok = n_tok;
    c.adam = adam;
    c.Pw = params_w;
    const bool zero_g = opts && opts->zero_grads;
    if (!side_on() && zero_g) TRY(zero(c.s, grads, (size_t)L->live * sizeof(float)));
    if (adam && row_step && L->d.d_model == FIRA_D) {        // (the forward gathers read lagging rows lazily: adam_rows_load)
        c.row_step = row_step;
        c.rows_ad = *adam;
    }
    c.dp_begin = begin_only;
    if (begin_only) c.adam = nullptr;        // (fira_train_step_begin_rows: the optimizer's values serve the lazy reads only)
    TRY(encoder_forward(c, true));
    if (side_on()) {
        // the buffers the backward pass accumulates into are cleared on the auxiliary stream beside the forward pass (they
        // are backward-only) -- and with them, on request, the gradient buffer itself (111 MB the caller would otherwise
        // fill ahead of the step).  Enqueued AFTER the encoder's own auxiliary work: that stream is in order, and the folded
        // GCN weights at its head are what the caller's stream waits for first (with the fills ahead of them the first
        // aggregation launch stood 160 us idle under the profiler; same-box A/B 9 142 / 9 128 -> 9 206 / 9 322 commits/s); it forked from the caller's stream inside
        // encoder_forward, so the fills still follow the previous step's last reader.
        TRY(zero(side().aux, p.zero_beg, (size_t)((char*)p.zero_end - (char*)p.zero_beg)));
        TRY(zero(side().aux, p.dXa, (size_t)batch->n_nodes * FIRA_D * sizeof(float)));
        if (zero_g) TRY(zero(side().aux, grads, (size_t)L->live * sizeof(float)));
        TRY(side_mark(&c.ev_zero));
    }
    if (c.ev_shadow) TRY(main_wait(c.s, c.ev_shadow, __LINE__));      // the decoder's products read the bf16 shadows
    TRY(decoder_forward(c));
    TRY(head_forward(c, R, rows, loss_sum, n_tok, nullptr, 1));
    if (begin_only) {
        // first half of a data-parallel step: up to the point where the gradients of [0, split) are final (mid_event)
        PendingStep& ps = g_pending;
        ps.active = false;
        ps.mid = BwdMid();
        TRY(backward_decoder(c, R, rows, (hipEvent_t)mid_event, ps.mid));
        ps.plan = p;
        ps.batch = *batch;
        ps.ctx = c;
        ps.ctx.pl = &ps.plan;
        ps.ctx.bt = &ps.batch;
        ps.dtype = opts ? opts->dtype : 0;
        ps.params = params;
        ps.tab = tab;
        ps.W21 = g_W21; ps.W21n = g_W21n; ps.W21b = g_W21b; ps.W21bT = g_W21bT;
        ps.active = true;
        return 0;
    }
    TRY(backward(c, R, rows, (hipEvent_t)mid_event));
    if (wait_probe().enabled() && ++wait_probe().calls == 20) wait_probe().report();
    return 0;
}

int fira_train_fwd_bwd(void* stream, const fira_dims* d, const fira_batch* batch, const float* params, float* grads,
                       void* workspace, size_t workspace_bytes, const fira_train_opts* opts, float* loss_sum,
                       int32_t* n_tok, void* mid_event) {
    return train_call(stream, d, batch, params, grads, workspace, workspace_bytes, opts, loss_sum, n_tok, mid_event, nullptr,
                      nullptr);
}

// ---- op-level entries named by SURVEY.md 8(b) that the model-level calls compose from smaller launches ------------------
// FeedForward block (gnn_transformer.py:163-174): h = relu(x W1^T + b1); sum = dropout(h W2^T + b2) + x; y = LayerNorm(sum).
int fira_ffn_fwd(void* stream, int M, int F, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                 const float* gamma, const float* beta, float* h, float* sum, float* y, float* stats, float dropout,
                 uint64_t seed, uint32_t site_id, int dtype) {
    FIRA_REQUIRE(x && w1 && b1 && w2 && b2 && gamma && beta && h && sum && y && stats && M > 0 && F > 0, "fira_ffn_fwd: bad argument");
    FIRA_REQUIRE(dropout >= 0.f && dropout < 1.f && (dtype == 0 || dtype == 1), "fira_ffn_fwd: bad dropout / dtype");
    DtypeScope dtype_scope(dtype);
    hipStream_t s = (hipStream_t)stream;
    const int D = FIRA_D;
    TRY(linear(s, M, F, D, x, D, w1, b1, h, F, FIRA_GEMM_RELU));
    return linear_ln(s, M, F, h, F, w2, b2, x, gamma, beta, sum, y, stats, dropout, seed, site_id);
}
// Its backward: dy = gradient w.r.t. y.  dx [M,256] is WRITTEN (gradient w.r.t. x: residual branch + through the two
// products); dw1 [F,256], db1 [F], dw2 [256,F], db2 [256], dgamma, dbeta [256] are ACCUMULATED into.  dyf_ws [M,256] and
// dh_ws [M,F] are caller-provided scratch (they hold d(h W2^T + b2) and d(x W1^T + b1) on return).
int fira_ffn_bwd(void* stream, int M, int F, const float* dy, const float* x, const float* h, const float* sum,
                 const float* stats, const float* w1, const float* w2, const float* gamma, float* dx, float* dyf_ws,
                 float* dh_ws, float* dw1, float* db1, float* dw2, float* db2, float* dgamma, float* dbeta, float dropout,
                 uint64_t seed, uint32_t site_id, int dtype) {
    FIRA_REQUIRE(dy && x && h && sum && stats && w1 && w2 && gamma && dx && dyf_ws && dh_ws && dw1 && db1 && dw2 && db2 && dgamma &&
                 dbeta && M > 0 && F > 0, "fira_ffn_bwd: bad argument");
    FIRA_REQUIRE(dx != dy, "fira_ffn_bwd: dx must not alias dy");
    FIRA_REQUIRE(dropout >= 0.f && dropout < 1.f && (dtype == 0 || dtype == 1), "fira_ffn_bwd: bad dropout / dtype");
    DtypeScope dtype_scope(dtype);
    hipStream_t s = (hipStream_t)stream;
    const int D = FIRA_D;
    red().reset(nullptr, 0);                    // stand-alone: column sums by atomics, nothing deferred
    // LayerNorm backward (residual-branch gradient -> dx, un-dropped branch gradient -> dyf_ws) + d hidden = (dyf W2) where h > 0
    TRY(ln_bwd_dgrad(s, M, F, dy, sum, stats, gamma, dx, dyf_ws, dgamma, dbeta, dropout, seed, site_id, w2, F, dh_ws, F, h));
    TRY(gemm_any(s, 1, 0, D, F, M, dyf_ws, D, h, F, dw2, F, nullptr, FIRA_GEMM_ACCUM, 0, db2));     // dW2 += dyf^T h, db2 += colsum
    TRY(gemm_any(s, 1, 0, F, D, M, dh_ws, F, x, D, dw1, D, nullptr, FIRA_GEMM_ACCUM, 0, db1));      // dW1 += dh^T x, db1 += colsum
    return linear_dgrad(s, M, F, D, dh_ws, F, w1, dx, D, true);                                      // dx += dh W1
}
// Output head for teacher-forced / search decoding (Model.py:54 + the arg-max of Model.py:85, run_model.py:305): logits =
// x Wout^T + bout into logits_ws [R, ldl >= V] and, per row, the k largest logits (value descending, ties by ascending id).
int fira_head_topk(void* stream, int R, int V, int k, const float* x, const float* wout, const float* bout, float* logits_ws,
                   int ldl, int32_t* ids, float* vals, int dtype) {
    FIRA_REQUIRE(x && wout && logits_ws && ids && vals && R > 0 && V > 0 && ldl >= V, "fira_head_topk: bad argument");
    FIRA_REQUIRE(dtype == 0 || dtype == 1, "fira_head_topk: bad dtype");
    DtypeScope dtype_scope(dtype);
    hipStream_t s = (hipStream_t)stream;
    TRY(linear(s, R, V, FIRA_D, x, FIRA_D, wout, bout, logits_ws, ldl));
    return row_topk(s, R, V, k, logits_ws, ldl, ids, vals);
}

int fira_train_step_begin(void* stream, const fira_dims* d, const fira_batch* batch, const float* params, float* grads,
                          void* workspace, size_t workspace_bytes, const fira_train_opts* opts, float* loss_sum,
                          int32_t* n_tok, void* mid_event) {
    FIRA_REQUIRE(mid_event, "fira_train_step_begin: mid_event missing (the caller's collective waits for it)");
    return train_call(stream, d, batch, params, grads, workspace, workspace_bytes, opts, loss_sum, n_tok, mid_event, nullptr,
                      nullptr, true);
}

// (v10) the data-parallel step with the row-sparse update of the word tables: _begin_rows takes the optimizer's values for the
// lazy reads of its forward pass (nothing is updated there), _end_rows updates decoder.embedding by rows inside the library;
// encoder.embedding is the caller's (fira_adam_rows_step with tables = 2 behind the late bucket's all-reduce)
static thread_local int32_t* g_end_row_step = nullptr;
int fira_train_step_begin_rows(void* stream, const fira_dims* d, const fira_batch* batch, const float* params, float* grads,
                               void* workspace, size_t workspace_bytes, const fira_train_opts* opts, float* loss_sum,
                               int32_t* n_tok, void* mid_event, const fira_adam_opts* adam, int32_t* row_step) {
    FIRA_REQUIRE(mid_event, "fira_train_step_begin_rows: mid_event missing (the caller's collective waits for it)");
    FIRA_REQUIRE(adam && adam->m && adam->v && adam->step >= 1 && row_step, "fira_train_step_begin_rows: bad Adam arguments");
    FIRA_REQUIRE(opts && opts->zero_grads, "fira_train_step_begin_rows: needs opts.zero_grads");
    // (adam only parameterises the lazy reads; train_call(begin_only) returns before any update)
    return train_call(stream, d, batch, params, grads, workspace, workspace_bytes, opts, loss_sum, n_tok, mid_event, nullptr,
                      adam, true, row_step);
}
int fira_train_step_end_rows(void* stream, float* params, const fira_adam_opts* adam, void* early_event, const float* count,
                             int32_t* row_step) {
    FIRA_REQUIRE(adam && row_step, "fira_train_step_end_rows: bad argument");
    g_end_row_step = row_step;
    const int rc = fira_train_step_end(stream, params, adam, early_event, count);
    g_end_row_step = nullptr;
    return rc;
}
int fira_train_step_end(void* stream, float* params, const fira_adam_opts* adam, void* early_event, const float* count) {
    PendingStep& ps = g_pending;
    FIRA_REQUIRE(ps.active, "fira_train_step_end: no step begun on this thread (fira_train_step_begin)");
    ps.active = false;
    FIRA_REQUIRE((hipStream_t)stream == ps.ctx.s, "fira_train_step_end: not the stream the step was begun on");
    FIRA_REQUIRE(!adam || (adam->m && adam->v && adam->step >= 1 && params), "fira_train_step_end: bad Adam arguments");
    FIRA_REQUIRE(!params || params == ps.params, "fira_train_step_end: not the parameter buffer the step was begun with");
    DtypeScope dtype_scope(ps.dtype);
    const bool bf16 = ps.dtype == 1;
    ShadowScope shadow_scope(ps.params, ps.ctx.L->total, bf16 ? ps.plan.wb : nullptr, bf16 ? ps.plan.wbt : nullptr, ps.tab);
    g_W21 = ps.W21; g_W21n = ps.W21n; g_W21b = ps.W21b; g_W21bT = ps.W21bT;
    Ctx& c = ps.ctx;
    c.adam = adam;
    c.Pw = params;
    c.row_step = g_end_row_step;             // (fira_train_step_end_rows; nullptr: every row of decoder.embedding, as before)
    c.adam_a_only = true;
    c.ev_early = (hipEvent_t)early_event;
    c.count = count;
    TRY(backward_encoder(c, ps.mid));
    if (wait_probe().enabled() && ++wait_probe().calls == 20) wait_probe().report();
    return 0;
}

int fira_train_step(void* stream, const fira_dims* d, const fira_batch* batch, float* params, float* grads, void* workspace,
                    size_t workspace_bytes, const fira_train_opts* opts, float* loss_sum, int32_t* n_tok,
                    const fira_adam_opts* adam) {
    FIRA_REQUIRE(adam && adam->m && adam->v, "fira_train_step: Adam moments missing");
    FIRA_REQUIRE(adam->step >= 1, "fira_train_step: the Adam step counter starts at 1");
    return train_call(stream, d, batch, params, grads, workspace, workspace_bytes, opts, loss_sum, n_tok, nullptr, params, adam);
}

int fira_train_step_rows(void* stream, const fira_dims* d, const fira_batch* batch, float* params, float* grads, void* workspace,
                         size_t workspace_bytes, const fira_train_opts* opts, float* loss_sum, int32_t* n_tok,
                         const fira_adam_opts* adam, int32_t* row_step) {
    FIRA_REQUIRE(adam && adam->m && adam->v, "fira_train_step_rows: Adam moments missing");
    FIRA_REQUIRE(adam->step >= 1, "fira_train_step_rows: the Adam step counter starts at 1");
    FIRA_REQUIRE(row_step, "fira_train_step_rows: row_step missing");
    FIRA_REQUIRE(opts && opts->zero_grads, "fira_train_step_rows: needs opts.zero_grads (an all-zero gradient row = an untouched row)");
    return train_call(stream, d, batch, params, grads, workspace, workspace_bytes, opts, loss_sum, n_tok, nullptr, params, adam,
                      false, row_step);
}
// op-level pieces of the same (tests/test_adam_rows_gpu.py compares them with fira_adam_step_mb bit for bit)
int fira_adam_rows_step(void* stream, const fira_dims* d, float* params, const float* grads, const fira_adam_opts* adam,
                        int32_t* row_step, const int32_t* n_tok, const float* count, int tables) {
    const Layout* L = get_layout(d);
    if (!L) return 1;
    FIRA_REQUIRE(params && grads && adam && adam->m && adam->v && row_step && (n_tok || count) && adam->step >= 1 && tables >= 1 &&
                     tables <= 3, "fira_adam_rows_step: bad argument");
    FIRA_REQUIRE(L->d.d_model == FIRA_D, "fira_adam_rows_step: model width must be %d", FIRA_D);
    return adam_rows_step((hipStream_t)stream, adam_rows_tables(*L, params, *adam, row_step), grads, adam->lr, adam->beta1,
                          adam->beta2, adam->eps, adam->step, count ? nullptr : n_tok, count, tables);
}
int fira_adam_rows_catchup(void* stream, const fira_dims* d, float* params, const fira_adam_opts* adam, int32_t* row_step,
                           int table, const int32_t* ids, int n_ids) {
    const Layout* L = get_layout(d);
    if (!L) return 1;
    FIRA_REQUIRE(params && adam && adam->m && adam->v && row_step && ids && n_ids >= 0 && (table == 0 || table == 1),
                 "fira_adam_rows_catchup: bad argument");
    FIRA_REQUIRE(L->d.d_model == FIRA_D, "fira_adam_rows_catchup: model width must be %d", FIRA_D);
    AdamRowsLists ls{};
    ls.n_lists = 1; ls.ids[0] = ids; ls.table[0] = table; ls.end[0] = n_ids;
    return adam_rows_catchup((hipStream_t)stream, adam_rows_tables(*L, params, *adam, row_step), &ls, adam->lr, adam->beta1,
                             adam->beta2, adam->eps, adam->step);
}
int fira_adam_rows_sync(void* stream, const fira_dims* d, float* params, const fira_adam_opts* adam, int32_t* row_step) {
    const Layout* L = get_layout(d);
    if (!L) return 1;
    FIRA_REQUIRE(params && adam && adam->m && adam->v && row_step && adam->step >= 0, "fira_adam_rows_sync: bad argument");
    FIRA_REQUIRE(L->d.d_model == FIRA_D, "fira_adam_rows_sync: model width must be %d", FIRA_D);
    return adam_rows_catchup((hipStream_t)stream, adam_rows_tables(*L, params, *adam, row_step), nullptr, adam->lr, adam->beta1,
                             adam->beta2, adam->eps, adam->step);
}

int fira_forward_dev(void* stream, const fira_dims* d, const fira_batch* batch, const float* params, void* workspace,
                     size_t workspace_bytes, int32_t* ids_out, float* loss_sum, int32_t* n_tok, int dtype) {
    const Layout* L = get_layout(d);
    if (!L) return 1;
    FIRA_REQUIRE(dtype == 0 || dtype == 1, "fira_forward_dev: dtype must be 0 (fp32) or 1 (bf16)");
    DtypeScope dtype_scope(dtype);
    TRY(check_batch(batch));
    FIRA_REQUIRE(batch->tar && batch->tar_label && params && workspace && ids_out, "null pointer argument");
    Plan p;
    const size_t need = p.build(workspace, *d, batch->B, false);
    FIRA_REQUIRE(need <= workspace_bytes, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    TRY(check_counts(batch, p));
    Ctx c{(hipStream_t)stream, L, batch, params, nullptr, &p, 0.f, 0.f, 0};
    c.Td = p.TB;
    c.loss_sum = loss_sum;
    c.n_tok = n_tok;
    const ShadowTable* tab = dtype == 1 ? shadow_table(*L) : nullptr;
    ShadowScope shadow_scope(params, L->total, dtype == 1 ? p.wb : nullptr, dtype == 1 ? p.wbt : nullptr, tab);
    if (dtype == 1) TRY(weight_shadows(c.s, *tab, params, p.wb, p.wbt));
    TRY(encoder_forward(c, true));
    TRY(decoder_forward(c));
    TRY(head_forward(c, p.TB, p.iota, loss_sum, n_tok, ids_out, 0));
    return 0;
}

int fira_decode_begin(void* stream, const fira_dims* d, const fira_batch* batch, const float* params, void* workspace,
                      size_t workspace_bytes, int n_beam) {
    return fira_decode_begin_ex(stream, d, batch, params, workspace, workspace_bytes, n_beam, 0);
}
int fira_decode_begin_ex(void* stream, const fira_dims* d, const fira_batch* batch, const float* params, void* workspace,
                         size_t workspace_bytes, int n_beam, int flags) {
    const Layout* L = get_layout(d);
    if (!L) return 1;
    TRY(check_batch(batch));
    FIRA_REQUIRE(params && workspace && n_beam >= 1, "bad argument");
    FIRA_REQUIRE((flags & ~FIRA_DECODE_KV_BF16) == 0, "fira_decode_begin_ex: unknown flags %d", flags);
    DecodePlan dp;
    const size_t need = dp.build(workspace, *d, batch->B, n_beam, flags);
    FIRA_REQUIRE(need <= workspace_bytes, "workspace too small: need %zu bytes, got %zu (fira_decode_workspace_bytes_ex with the same flags)",
                 need, workspace_bytes);
    Plan& p = dp.enc;
    fira_batch b2 = *batch;
    b2.tar = nullptr;
    Ctx c{(hipStream_t)stream, L, &b2, params, nullptr, &p, 0.f, 0.f, 0};
    c.serial = true;
    TRY(check_counts(batch, p));
    TRY(encoder_forward(c, false));     // also leaves kv_all (cross K|V of all layers) and src = LinearSource(memory)
    TRY(rows_move(c.s, 1, b2.n_mem, FIRA_D, p.mem, p.mem_c, nullptr, b2.mem_dst));   // dense memory view for callers
    {   // k-major copies of fc_o (self attention) and fc_q (cross attention) of every layer: decode_self_block streams them
        TransposeTable tt;
        for (int l = 0; l < p.nl && tt.n + 2 <= 24; ++l) {
            tt.src[tt.n] = params + L->dec[l].wo_s; tt.dst[tt.n++] = dp.wsT + (size_t)(2 * l) * FIRA_D * FIRA_D;
            tt.src[tt.n] = params + L->dec[l].wq_c; tt.dst[tt.n++] = dp.wsT + (size_t)(2 * l + 1) * FIRA_D * FIRA_D;
        }
        TRY(transpose256_table(c.s, tt));
    }
    // optional: the cross K|V rows the step loop streams 30 times, once more in bf16 (half the bytes per step; rows of
    // masked slots are converted too -- they are never read)
    if (flags & FIRA_DECODE_KV_BF16) TRY(rows_to_bf16(c.s, (int64_t)b2.n_mem * p.kvp, p.kv_all, dp.kv16));
    return 0;
}

int fira_decode_step(void* stream, const fira_dims* d, const float* params, void* workspace, size_t workspace_bytes,
                     int B, int n_beam, int step, const int32_t* tokens, const int32_t* parent, float* dist,
                     int32_t* best_id, float* best_p) {
    return fira_decode_step_ex(stream, d, params, workspace, workspace_bytes, B, n_beam, step, tokens, parent, dist, best_id,
                               best_p, 0);
}
int fira_decode_step_ex(void* stream, const fira_dims* d, const float* params, void* workspace, size_t workspace_bytes,
                        int B, int n_beam, int step, const int32_t* tokens, const int32_t* parent, float* dist,
                        int32_t* best_id, float* best_p, int flags) {
    const Layout* Lp = get_layout(d);
    if (!Lp) return 1;
    const Layout& L = *Lp;
    FIRA_REQUIRE(params && workspace && tokens && B > 0 && n_beam >= 1, "bad argument");
    FIRA_REQUIRE(step >= 0 && step < d->tar_len, "step %d out of range", step);
    FIRA_REQUIRE((flags & ~FIRA_DECODE_KV_BF16) == 0, "fira_decode_step_ex: unknown flags %d", flags);
    DecodePlan dp;
    const size_t need = dp.build(workspace, *d, B, n_beam, flags);
    FIRA_REQUIRE(need <= workspace_bytes, "workspace too small: need %zu bytes, got %zu (fira_decode_workspace_bytes_ex with the same flags)",
                 need, workspace_bytes);
    Plan& p = dp.enc;
    hipStream_t s = (hipStream_t)stream;
    const int D = FIRA_D, H = d->n_head, T = p.T, KV = p.nl * 2 * D, Sm = p.L + p.S, BR = dp.BR;
    const int cur = n_beam > 1 ? (step & 1) : 0, prev = n_beam > 1 ? ((step + 1) & 1) : 0;
    if (n_beam > 1 && step > 0)
        TRY(permute_cache(s, p.nl, BR, T, step, parent, dp.kc[prev], dp.vc[prev], dp.kc[cur], dp.vc[cur], dp.hist[prev],
                          dp.hist[cur]));
    // key-valid history of this step + token embedding + position `step` (gnn_transformer.py:110-113): one launch
    TRY(decode_embed(s, BR, T, step, tokens, params + L.dec_emb, p.pos_tar + (size_t)step * D, dp.x, dp.hist[cur]));
    const size_t lay = (size_t)BR * T * D;
    // LayerNorms deferred into the next product's prologue (see decoder_forward): pend_* = the block still owed
    const float *pend_g = nullptr, *pend_b = nullptr;
    float* pend_y = nullptr;
    bool pending = false;
    DtypeScope dtype_scope(0);                  // the search runs the reference's fp32 arithmetic
    // FIRA_DECODE_SELF_BLOCK=0: the three launches (A/B switch); the fused kernel needs the model's 8 x 32 heads and <= 32 keys
    static const bool self_block_off = [] { const char* e = getenv("FIRA_DECODE_SELF_BLOCK"); return e && e[0] == '0'; }();
    const bool self_block = !self_block_off && H * FIRA_DH == FIRA_D && H == 8 && T <= 32 && p.nl <= 12;
    auto consume = [&](int N, const float* xin, const float* W, const float* b, float* Y, int flags) -> int {
        if (pending) {
            int rc = 0;
            const bool fused = gemm_tile32_ln_try(s, BR, N, dp.s, D, W, b, Y, N, flags, pend_g, pend_b, pend_y, nullptr, &rc);
            if (!fused) rc = add_layernorm_fwd(s, BR, dp.s, nullptr, pend_g, pend_b, pend_y, nullptr, 0.f, 0, 0, nullptr);
            pending = false;
            if (rc || fused) return rc;
        }
        return linear(s, BR, N, D, xin, D, W, b, Y, N, flags);
    };
    auto close_block = [&](int K, const float* X, const float* W, const float* b, const float* res, const float* g, const float* be,
                           float* y, bool may_defer) -> int {
        int rc = 0;
        if (may_defer && linear_presum(s, BR, K, X, K, W, b, res, dp.s, 0.f, 0, 0, &rc)) {
            pending = true; pend_g = g; pend_b = be; pend_y = y;
            return rc;
        }
        return linear_ln(s, BR, K, X, K, W, b, res, g, be, dp.s, y, nullptr, 0.f, 0, 0);
    };
    for (int l = 0; l < p.nl; ++l) {
        const DecLayer& w = L.dec[l];
        float* kc = dp.kc[cur] + l * lay;
        float* vc = dp.vc[cur] + l * lay;
        // q|k|v as ONE product; the attention kernel reads the new key / value from its output row and appends them to the
        // cache (decode_attention, attention.hip).  (The round-2 path -- three projections + the 32-query MFMA attention
        // kernel, FIRA_DECODE_ATTN=0 -- lost by 0.2 ms per step in round 3 and is gone.)
        TRY(consume(3 * D, dp.x, params + w.wqkv, params + w.bqkv, dp.qkv, 0));
        if (self_block) {
            // (round 6) attention over the cached keys + fc_o + residual + LayerNorm + the cross attention's fc_q: ONE launch per
            // layer instead of three of the dependent chain (attention.hip: decode_self_block)
            TRY(decode_self_block(s, BR, step + 1, T, dp.qkv, kc, vc, dp.hist[cur], dp.wsT + (size_t)(2 * l) * D * D, params + w.bo_s,
                                  dp.x, params + w.lns_g, params + w.lns_b, dp.wsT + (size_t)(2 * l + 1) * D * D, params + w.bq_c, dp.xa,
                                  dp.qc));
        } else {
        TRY(decode_attention(s, BR, H, step + 1, dp.qkv, 3 * D, kc, D, vc, D, dp.hist[cur], dp.ao, D, T, T, 1,
                             dp.qkv + D, dp.qkv + 2 * D, 3 * D, kc, vc));
        TRY(close_block(D, dp.ao, params + w.wo_s, params + w.bo_s, dp.x, params + w.lns_g, params + w.lns_b, dp.xa, true));
        TRY(consume(D, dp.xa, params + w.wq_c, params + w.bq_c, dp.qc, 0));
        }
        if (flags & FIRA_DECODE_KV_BF16)
            TRY(decode_attention_kv16(s, BR, H, Sm, dp.qc, D, dp.kv16 + l * 2 * D, p.kvp, dp.kv16 + l * 2 * D + D, p.kvp, p.mem_valid_c,
                                      dp.ao, D, Sm, Sm, n_beam, p.mem_off));
        else
            TRY(decode_attention(s, BR, H, Sm, dp.qc, D, p.kv_all + l * 2 * D, p.kvp, p.kv_all + l * 2 * D + D, p.kvp, p.mem_valid_c,
                                 dp.ao, D, Sm, Sm, n_beam, nullptr, nullptr, 0, nullptr, nullptr, p.mem_off));
        TRY(close_block(D, dp.ao, params + w.wo_c, params + w.bo_c, dp.xa, params + w.lnc_g, params + w.lnc_b, dp.xc, true));
        TRY(consume(p.F, dp.xc, params + w.w1, params + w.b1, dp.h, FIRA_GEMM_RELU));
        // (the last block's LayerNorm is owed too: the target projection of the copy head consumes it and leaves x behind)
        TRY(close_block(p.F, dp.h, params + w.w2, params + w.b2, dp.xc, params + w.lnf_g, params + w.lnf_b, dp.x, true));
        // (round 6, measured and removed: fc_o + LayerNorm + FeedForward as two ROW kernels -- dot products on the k-contiguous
        // weights, grid rows x 4 -- 0.351 against 0.333 ms per step: every row's workgroup pulls 512 KB of weights through its own
        // CU's L2 port, 128 MB per launch at 64 rows, where the 32-row tiles amortise them; profiles/r6_probes.md)
    }
    TRY(consume(D, dp.x, params + L.wt, nullptr, dp.tgt, 0));      // LinearTarget(LN(..)) [+ x materialised]
    TRY(linear(s, BR, p.V, D, dp.x, D, params + L.wout, params + L.bout, dp.logits, p.ldl));
    TRY(copy_score_fwd_ex(s, BR, 1, Sm, p.src, dp.tgt, params + L.wres, params + L.bres, dp.score, n_beam, p.mem_valid));
    // the 2-way gate LinearProb(x) is formed inside the distribution kernel
    TRY(decode_dist(s, BR, p.V, Sm, dp.logits, p.ldl, dp.score, p.mem_valid, n_beam, nullptr, dist, best_id, best_p, dp.x,
                    params + L.wp, params + L.bp));
    return 0;
}

// Decoder.forward on caller-supplied memory (the piecewise surface the reference's test loop uses:
// model.decoder(ids, memory, mem_mask, tar_pad_mask), run_model.py:256): full 30-position recompute.
int fira_decoder_forward(void* stream, const fira_dims* d, const float* params, void* workspace, size_t workspace_bytes,
                         int B, const int32_t* tar, const float* memory, const int32_t* mem_valid, float* out) {
    const Layout* L = get_layout(d);
    if (!L) return 1;
    FIRA_REQUIRE(params && workspace && tar && memory && mem_valid && out && B > 0, "bad argument");
    Plan p;
    const size_t need = p.build(workspace, *d, B, false);
    FIRA_REQUIRE(need <= workspace_bytes, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    const int D = FIRA_D, KV = p.nl * 2 * D;
    fira_batch b{};
    b.B = B;
    b.tar = tar;
    Ctx c{s, L, &b, params, nullptr, &p, 0.f, 0.f, 0};
    c.Td = p.TB;
    TRY(fill_pos_tables(s, p.L, p.pos_code, p.T, p.pos_tar));
    hipError_t e = hipMemcpyAsync(p.mem_valid, mem_valid, (size_t)p.MB * sizeof(int32_t), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return set_err("hipMemcpyAsync: %s", hipGetErrorString(e));
    TRY(tar_mask(s, p.TB, tar, p.tar_valid));
    TRY(linear(s, p.MB, KV, D, memory, D, params + L->wkv_all, params + L->bkv_all, p.kv_all, p.kvp));
    TRY(decoder_forward(c));
    e = hipMemcpyAsync(out, p.dec[p.nl - 1].x_f, (size_t)p.TB * D * sizeof(float), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return set_err("hipMemcpyAsync: %s", hipGetErrorString(e));
    return 0;
}

// Measurement aid (scripts/event_cost.py): n dependent tiny kernels on `stream`; mode 1 forks the library's weight-gradient
// stream after every kernel (event record on `stream` + wait + a tiny kernel there) as linear_wgrad does, mode 2 does the
// event record alone, mode 3 forks every 8th kernel.  Joins at the end.  Tells what a fork costs the dependent chain.
__global__ void debug_tiny_kernel(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }
int fira_debug_chain(void* stream, int n, int mode, float* scratch) {
    FIRA_REQUIRE(scratch && n > 0, "fira_debug_chain: bad argument");
    TRY(side().init());
    hipStream_t s = (hipStream_t)stream;
    SideStream& sd = side();
    for (int i = 0; i < n; ++i) {
        if (mode >= 4 && mode <= 6) {
            // the event rides on the kernel's own completion signal (hipExtLaunchKernelGGL's stopEvent): no barrier packet
            // of its own on `stream`.  4: attached to every kernel, nobody waits; 5: the other stream waits for it and runs
            // a kernel (a fork without hipEventRecord); 6: the same for every 8th kernel
            const bool ev = mode != 6 || (i & 7) == 7;
            hipEvent_t e = ev ? sd.ev() : nullptr;
            hipExtLaunchKernelGGL(debug_tiny_kernel, dim3(240), dim3(256), 0, s, nullptr, e, 0, scratch);
            if (ev && mode != 4) {
                if (hipStreamWaitEvent(sd.stream, e, 0) != hipSuccess) return set_err("stream wait failed");
                hipLaunchKernelGGL(debug_tiny_kernel, dim3(240), dim3(256), 0, sd.stream, scratch + 64);
            }
            continue;
        }
        hipLaunchKernelGGL(debug_tiny_kernel, dim3(240), dim3(256), 0, s, scratch);
        if (mode == 1 || (mode == 3 && (i & 7) == 7)) {
            TRY(side_fork(s));
            hipLaunchKernelGGL(debug_tiny_kernel, dim3(240), dim3(256), 0, sd.stream, scratch + 64);
        } else if (mode == 2) {
            hipEvent_t e = sd.ev();
            if (hipEventRecord(e, s) != hipSuccess) return set_err("event record failed");
        }
    }
    if (mode == 1 || mode == 3 || mode == 5 || mode == 6) TRY(side_join(s, __LINE__));
    FIRA_CHECK_LAUNCH("debug_chain");
    return 0;
}

const float* fira_decode_memory(const fira_dims* d, void* workspace, int B, int n_beam) {
    if (!get_layout(d) || !workspace) return nullptr;
    DecodePlan dp;
    dp.build(workspace, *d, B, n_beam);
    return dp.enc.mem;
}
const int32_t* fira_decode_mem_valid(const fira_dims* d, void* workspace, int B, int n_beam) {
    if (!get_layout(d) || !workspace) return nullptr;
    DecodePlan dp;
    dp.build(workspace, *d, B, n_beam);
    return dp.enc.mem_valid;
}

}  // extern "C"
