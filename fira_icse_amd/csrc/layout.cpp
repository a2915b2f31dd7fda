// Flat parameter layout: the 338 tensors of the reference's state_dict (SURVEY.md §8b "Checkpoint") in one
// fp32 buffer.  Index order is the reference's state_dict order; *placement* groups the weights that one GEMM
// consumes together (q|k of a Combination, q|k|v of a self-attention, k|v of all six cross-attentions, the six
// Combination value projections), so the fused GEMMs read one contiguous [N_total, 256] weight.  Every tensor
// starts on a 256-byte boundary.  One flat buffer makes Adam a single kernel and the data-parallel gradient
// all-reduce a single bucketed RCCL call.
#include "engine.h"
#include <stdarg.h>
#include <map>
#include <mutex>

namespace fira {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
int set_err(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return 1;
}

// ---------------------------------------------------------------- event profiler
namespace {
struct ProfRec { int cls; double work, bytes; hipEvent_t a, b; };
struct ProfState {
    std::mutex mu;
    bool on = false;
    std::vector<ProfRec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);                 // profiling aid only: a failed event shows up as a zero interval
        return e;
    }
};
ProfState& PS() { static ProfState s; return s; }
}  // namespace
bool prof_on() { return PS().on; }
static thread_local int tl_decoder_tag = 0;
void prof_decoder_tag(int delta) { tl_decoder_tag += delta; }
int prof_begin(hipStream_t s, int cls, double work, double bytes) {
    ProfState& p = PS();
    if (cls == PROF_GEMM && tl_decoder_tag > 0) cls = PROF_GEMM_DEC;
    std::lock_guard<std::mutex> g(p.mu);
    ProfRec r{cls, work, bytes, p.get(), p.get()};
    (void)hipEventRecord(r.a, s);
    p.recs.push_back(r);
    return (int)p.recs.size() - 1;        // the record this scope closes (two host threads may enqueue concurrently)
}
void prof_end(hipStream_t s, int idx) {
    ProfState& p = PS();
    std::lock_guard<std::mutex> g(p.mu);
    if (idx >= 0 && idx < (int)p.recs.size()) (void)hipEventRecord(p.recs[idx].b, s);
}

namespace {

struct Builder {
    Layout* L;
    std::map<std::string, int> index;
    void add(const std::string& name, int64_t r, int64_t c = -1) {
        ParamInfo p;
        p.name = name;
        p.ndim = c < 0 ? 1 : 2;
        p.shape[0] = r;
        p.shape[1] = c < 0 ? 0 : c;
        p.numel = c < 0 ? r : r * c;
        index[name] = (int)L->infos.size();
        L->infos.push_back(p);
    }
    void linear(const std::string& pre, int64_t out, int64_t in, bool bias = true) {
        add(pre + ".weight", out, in);
        if (bias) add(pre + ".bias", out);
    }
    int64_t place(const std::string& name) {
        ParamInfo& p = L->infos[index.at(name)];
        if (p.offset < 0) {
            p.offset = L->total;
            L->total += (p.numel + 63) / 64 * 64;
        }
        return p.offset;
    }
};

Layout* build(const fira_dims& d) {
    Layout* L = new Layout();
    L->d = d;
    Builder b{L};
    const int64_t D = d.d_model, V = d.vocab, F = d.d_ff;
    char buf[160];
    auto S = [&](const char* fmt, int i) { snprintf(buf, sizeof buf, fmt, i); return std::string(buf); };
    // ---- names in the reference's state_dict order (gnn_transformer.py:21-43,88-106; Model.py:24-36) ----
    b.add("encoder.embedding.weight", V, D);
    b.add("encoder.ast_change_embedding.weight", d.ast_vocab, D);
    b.add("encoder.mark_embedding.weight", 4, D);
    for (int l = 0; l < 3; ++l) {                        // dead LSTM: kept for checkpoint parity
        b.add(S("encoder.lstm.weight_ih_l%d", l), 4 * D, D);
        b.add(S("encoder.lstm.weight_hh_l%d", l), 4 * D, D);
        b.add(S("encoder.lstm.bias_ih_l%d", l), 4 * D);
        b.add(S("encoder.lstm.bias_hh_l%d", l), 4 * D);
    }
    for (int list = 1; list <= 2; ++list)                // combination_list1 is dead too
        for (int i = 0; i < d.n_layer; ++i) {
            snprintf(buf, sizeof buf, "encoder.combination_list%d.%d", list, i);
            std::string pre = buf;
            for (int j = 0; j < 3; ++j) b.linear(pre + S(".linear_layers.%d", j), D, D);
            b.linear(pre + ".output_linear", D, D);
            b.add(pre + ".layernorm.weight", D);
            b.add(pre + ".layernorm.bias", D);
        }
    for (int i = 0; i < d.n_layer; ++i) {
        std::string pre = S("encoder.gcn_list.%d", i);
        b.linear(pre + ".fc1", D, D);
        b.linear(pre + ".fc2", D, D);
        b.add(pre + ".layernorm.weight", D);
        b.add(pre + ".layernorm.bias", D);
    }
    b.add("decoder.embedding.weight", V, D);
    for (int kind = 0; kind < 2; ++kind)
        for (int i = 0; i < d.n_layer; ++i) {
            std::string pre = S(kind == 0 ? "decoder.attention_list.%d" : "decoder.cross_attention_list.%d", i);
            b.linear(pre + ".fc_q", D, D);
            b.linear(pre + ".fc_k", D, D);
            b.linear(pre + ".fc_v", D, D);
            b.linear(pre + ".fc_o", D, D);
            b.add(pre + ".layernorm.weight", D);
            b.add(pre + ".layernorm.bias", D);
        }
    for (int i = 0; i < d.n_layer; ++i) {
        std::string pre = S("decoder.feed_forward_list.%d", i);
        b.linear(pre + ".fc1", F, D);
        b.linear(pre + ".fc2", D, F);
        b.add(pre + ".layernorm.weight", D);
        b.add(pre + ".layernorm.bias", D);
    }
    b.linear("out_fc", V, D);
    b.linear("gate_fc", 1, D);                           // dead
    b.linear("copy_net.LinearSource", D, D, false);
    b.linear("copy_net.LinearTarget", D, D, false);
    b.linear("copy_net.LinearRes", 1, D);
    b.linear("copy_net.LinearProb", 2, D);

    // ---- placement (offsets) -------------------------------------------------------------------------
    // Group A first: everything whose gradient is complete once the decoder backward has finished (head, decoder
    // layers, decoder embedding, cross K|V projections) -- the data-parallel all-reduce of this slice overlaps the
    // encoder backward.  Group B (encoder) follows; `split` is the boundary.
    L->enc.resize(d.n_layer);
    L->dec.resize(d.n_layer);
    L->dec_emb = b.place("decoder.embedding.weight");
    L->wout = b.place("out_fc.weight");
    L->bout = b.place("out_fc.bias");
    // cross-attention K|V projections of all layers, contiguous: one [6*512, 256] GEMM over the memory rows
    for (int i = 0; i < d.n_layer; ++i) {
        std::string pre = S("decoder.cross_attention_list.%d", i);
        int64_t o = b.place(pre + ".fc_k.weight");
        b.place(pre + ".fc_v.weight");
        if (i == 0) L->wkv_all = o;
        L->dec[i].wkv_c = o;
    }
    for (int i = 0; i < d.n_layer; ++i) {
        std::string pre = S("decoder.cross_attention_list.%d", i);
        int64_t o = b.place(pre + ".fc_k.bias");
        b.place(pre + ".fc_v.bias");
        if (i == 0) L->bkv_all = o;
        L->dec[i].bkv_c = o;
    }
    for (int i = 0; i < d.n_layer; ++i) {
        DecLayer& e = L->dec[i];
        std::string pre = S("decoder.attention_list.%d", i);
        e.wqkv = b.place(pre + ".fc_q.weight");
        b.place(pre + ".fc_k.weight");
        b.place(pre + ".fc_v.weight");
        e.bqkv = b.place(pre + ".fc_q.bias");
        b.place(pre + ".fc_k.bias");
        b.place(pre + ".fc_v.bias");
        e.wo_s = b.place(pre + ".fc_o.weight");
        e.bo_s = b.place(pre + ".fc_o.bias");
        e.lns_g = b.place(pre + ".layernorm.weight");
        e.lns_b = b.place(pre + ".layernorm.bias");
        pre = S("decoder.cross_attention_list.%d", i);
        e.wq_c = b.place(pre + ".fc_q.weight");
        e.bq_c = b.place(pre + ".fc_q.bias");
        e.wo_c = b.place(pre + ".fc_o.weight");
        e.bo_c = b.place(pre + ".fc_o.bias");
        e.lnc_g = b.place(pre + ".layernorm.weight");
        e.lnc_b = b.place(pre + ".layernorm.bias");
        pre = S("decoder.feed_forward_list.%d", i);
        e.w1 = b.place(pre + ".fc1.weight");
        e.b1 = b.place(pre + ".fc1.bias");
        e.w2 = b.place(pre + ".fc2.weight");
        e.b2 = b.place(pre + ".fc2.bias");
        e.lnf_g = b.place(pre + ".layernorm.weight");
        e.lnf_b = b.place(pre + ".layernorm.bias");
    }
    L->ws = b.place("copy_net.LinearSource.weight");
    L->wt = b.place("copy_net.LinearTarget.weight");
    L->wres = b.place("copy_net.LinearRes.weight");
    L->bres = b.place("copy_net.LinearRes.bias");
    L->wp = b.place("copy_net.LinearProb.weight");
    L->bp = b.place("copy_net.LinearProb.bias");
    L->split = L->total;
    // ---- group B: encoder ----
    L->emb = b.place("encoder.embedding.weight");
    L->ast_emb = b.place("encoder.ast_change_embedding.weight");
    L->mark_emb = b.place("encoder.mark_embedding.weight");
    // value projections of the six live Combination layers, contiguous: one [6*256, 256] GEMM on the 4-row mark table
    for (int i = 0; i < d.n_layer; ++i) {
        int64_t o = b.place(S("encoder.combination_list2.%d.linear_layers.2.weight", i));
        if (i == 0) L->w2_all = o;
        L->enc[i].w2 = o;
    }
    for (int i = 0; i < d.n_layer; ++i) {
        int64_t o = b.place(S("encoder.combination_list2.%d.linear_layers.2.bias", i));
        if (i == 0) L->b2_all = o;
        L->enc[i].b2 = o;
    }
    for (int i = 0; i < d.n_layer; ++i) {
        EncLayer& e = L->enc[i];
        std::string pre = S("encoder.combination_list2.%d", i);
        e.wqk = b.place(pre + ".linear_layers.0.weight");
        b.place(pre + ".linear_layers.1.weight");
        e.bqk = b.place(pre + ".linear_layers.0.bias");
        b.place(pre + ".linear_layers.1.bias");
        e.wo = b.place(pre + ".output_linear.weight");
        e.bo = b.place(pre + ".output_linear.bias");
        e.ln1g = b.place(pre + ".layernorm.weight");
        e.ln1b = b.place(pre + ".layernorm.bias");
        pre = S("encoder.gcn_list.%d", i);
        e.fc1w = b.place(pre + ".fc1.weight");
        e.fc1b = b.place(pre + ".fc1.bias");
        e.fc2w = b.place(pre + ".fc2.weight");
        e.fc2b = b.place(pre + ".fc2.bias");
        e.ln2g = b.place(pre + ".layernorm.weight");
        e.ln2b = b.place(pre + ".layernorm.bias");
    }
    L->live = L->total;
    // dead tensors last (never touched by a kernel; their gradient stays zero, Adam leaves them unchanged)
    for (auto& p : L->infos)
        if (p.offset < 0) b.place(p.name);
    return L;
}

}  // namespace

const Layout* get_layout(const fira_dims* d) {
    if (!d) { set_err("null fira_dims"); return nullptr; }
    if (d->d_model != FIRA_D || d->n_head * FIRA_DH != FIRA_D) {
        set_err("kernels are specialised for d_model=256, head width 32 (got d_model=%d n_head=%d)", d->d_model, d->n_head);
        return nullptr;
    }
    if (d->tar_len < 2 || d->tar_len > 32 || d->sou_len + d->sub_len > 384 || d->sou_len < 1 || d->sub_len < 0 ||
        d->ast_len < 0 || d->n_layer < 1 || d->n_layer > 16 || d->vocab < 4 || d->d_ff % 64 != 0) {
        set_err("unsupported geometry (tar_len 2..32, sou_len+sub_len <= 384, d_ff %% 64 == 0)");
        return nullptr;
    }
    static std::mutex mu;
    static std::map<std::string, Layout*> cache;
    std::lock_guard<std::mutex> g(mu);
    std::string key((const char*)d, sizeof(fira_dims));
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    Layout* L = build(*d);
    cache[key] = L;
    return L;
}

}  // namespace fira

extern "C" {
const char* fira_last_error(void) { return fira::err_buf(); }
int fira_abi_version(void) { return FIRA_ABI_VERSION; }
int fira_param_count(const fira_dims* d) {
    const fira::Layout* L = fira::get_layout(d);
    return L ? (int)L->infos.size() : -1;
}
int fira_param_info(const fira_dims* d, int index, char* name_buf, int64_t* offset, int64_t* numel, int32_t* ndim,
                    int64_t shape[2]) {
    const fira::Layout* L = fira::get_layout(d);
    if (!L) return 1;
    if (index < 0 || index >= (int)L->infos.size()) return fira::set_err("param index %d out of range", index);
    const fira::ParamInfo& p = L->infos[index];
    if (name_buf) { strncpy(name_buf, p.name.c_str(), 127); name_buf[127] = 0; }
    if (offset) *offset = p.offset;
    if (numel) *numel = p.numel;
    if (ndim) *ndim = p.ndim;
    if (shape) { shape[0] = p.shape[0]; shape[1] = p.shape[1]; }
    return 0;
}
void fira_prof_enable(int on) {
    fira::ProfState& p = fira::PS();
    std::lock_guard<std::mutex> g(p.mu);
    p.on = on != 0;
}
int fira_prof_report(int n_class, double* ms, double* work, double* bytes, int64_t* count) {
    fira::ProfState& p = fira::PS();
    std::lock_guard<std::mutex> g(p.mu);
    for (int i = 0; i < n_class; ++i) { ms[i] = 0; work[i] = 0; count[i] = 0; if (bytes) bytes[i] = 0; }
    for (auto& r : p.recs) {
        (void)hipEventSynchronize(r.b);
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) t = 0.f;
        if (r.cls < n_class) {
            ms[r.cls] += t; work[r.cls] += r.work; count[r.cls] += 1;
            if (bytes) bytes[r.cls] += r.bytes;
        }
        p.pool.push_back(r.a);
        p.pool.push_back(r.b);
    }
    p.recs.clear();
    return 0;
}
int64_t fira_param_total(const fira_dims* d) {
    const fira::Layout* L = fira::get_layout(d);
    return L ? L->total : -1;
}
int fira_param_groups(const fira_dims* d, int64_t* split, int64_t* live) {
    const fira::Layout* L = fira::get_layout(d);
    if (!L) return 1;
    if (split) *split = L->split;
    if (live) *live = L->live;
    return 0;
}
}
