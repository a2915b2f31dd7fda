// Host-side (CPU) construction of fira_batch's index lists from one collated batch: the computed-node list with its compact
// CSR adjacency, the code / memory row lists, the word-id groups of the embedding gradient and the AST / edit (row, id)
// pairs -- model.py's computed_nodes + embedding_items + compact_embedding_lists as ONE pass in C++.
//
// Why: those three functions are ~45 numpy calls over 40 000-element arrays; they cost the prefetch thread 2.3 ms per
// batch of 64 commits and hold the GIL most of that time, which on a slow host CPU makes the loader, not the GPU, the bound
// of the fresh-batch-every-step rate (host_inclusive on the bench line: 7.9 ms per step against 4.8 ms resident).  This
// function does the same in ~0.15 ms with the GIL released (ctypes).  No device code, no HIP call.
// The numpy functions stay the specification: tests/test_host_lists.py requires bit-identical arrays on random batches.
#include "engine.h"
#include <algorithm>
#include <vector>

extern "C" int fira_host_node_lists(int B, int N, int L, int S, int skip_padding, const int64_t* sou,
                                    const int64_t* sub_token, const int64_t* ast_change, const int64_t* mark,
                                    const int32_t* rowptr, const int32_t* col, const float* val, int chunk,
                                    int32_t* node_rows, int32_t* rowptr_c, int32_t* col_c, float* val_c,
                                    int32_t* code_rows, int32_t* code_mark, int32_t* mem_rows, int32_t* mem_dst,
                                    int32_t* item_tok, int32_t* item_ptr, int32_t* emb_rows, int32_t* ast_rows,
                                    int32_t* ast_ids, int32_t* counts) {
    FIRA_REQUIRE(B > 0 && N > 0 && L >= 0 && S >= 0 && L + S <= N && chunk > 0, "fira_host_node_lists: bad dimensions");
    FIRA_REQUIRE(sou && sub_token && ast_change && mark && rowptr && col && val && counts, "fira_host_node_lists: null input");
    const int A = N - L - S;
    const int64_t total = (int64_t)B * N;
    std::vector<int32_t> cmap((size_t)total, -1);
    auto id_of = [&](int b, int local) -> int64_t {
        if (local < L) return sou[(size_t)b * L + local];
        if (local < L + S) return sub_token[(size_t)b * S + (local - L)];
        return ast_change[(size_t)b * A + (local - L - S)];
    };
    // ---- computed nodes: non-zero id, or an edge besides the self-loop (model.computed_nodes) ------------------------
    int n_nodes = 0, n_code = 0, n_mem = 0, n_ast = 0;
    int64_t nnz_c = 0;
    for (int b = 0; b < B; ++b)
        for (int local = 0; local < N; ++local) {
            const int64_t g = (int64_t)b * N + local;
            const int deg = rowptr[g + 1] - rowptr[g];
            const int64_t id = id_of(b, local);
            if (skip_padding && id == 0 && deg <= 1) continue;
            cmap[(size_t)g] = n_nodes;
            node_rows[n_nodes] = (int32_t)g;
            if (local < L) {
                code_rows[n_code] = n_nodes;
                code_mark[n_code] = (int32_t)mark[(size_t)b * L + local];
                ++n_code;
            }
            if (local < L + S) {
                mem_rows[n_mem] = n_nodes;
                mem_dst[n_mem] = b * (L + S) + local;
                ++n_mem;
            } else if (id != 0) {                                   // AST / edit node with an id (compact_embedding_lists)
                ast_rows[n_ast] = n_nodes;
                ast_ids[n_ast] = (int32_t)id;
                ++n_ast;
            }
            ++n_nodes;
        }
    // compact CSR (second pass: every column must already have its compact id)
    rowptr_c[0] = 0;
    for (int c = 0; c < n_nodes; ++c) {
        const int64_t g = node_rows[c];
        for (int e = rowptr[g]; e < rowptr[g + 1]; ++e) {
            const int32_t cc = cmap[(size_t)col[e]];
            FIRA_REQUIRE(cc >= 0, "fira_host_node_lists: a computed node has an edge to a skipped node");
            col_c[nnz_c] = cc;
            val_c[nnz_c] = val[e];
            ++nnz_c;
        }
        rowptr_c[c + 1] = (int32_t)nnz_c;
    }
    // ---- word-id groups of the code / sub-token positions (model.embedding_items), rows as compact ids -------------
    // stable order by token id, ties in (commit, position) order: a counting sort over the id range
    int64_t max_id = 0;
    int n_pos = 0;
    for (int b = 0; b < B; ++b)
        for (int local = 0; local < L + S; ++local) {
            const int64_t id = id_of(b, local);
            FIRA_REQUIRE(id >= 0, "fira_host_node_lists: negative token id");
            if (id != 0) { ++n_pos; max_id = std::max(max_id, id); }
        }
    int n_items = 0;
    if (n_pos > 0) {
        FIRA_REQUIRE(max_id < (1 << 26), "fira_host_node_lists: token id %lld too large for the counting sort", (long long)max_id);
        std::vector<int32_t> start((size_t)max_id + 2, 0);
        for (int b = 0; b < B; ++b)
            for (int local = 0; local < L + S; ++local) {
                const int64_t id = id_of(b, local);
                if (id != 0) ++start[(size_t)id + 1];
            }
        for (int64_t t = 1; t <= max_id + 1; ++t) start[(size_t)t] += start[(size_t)t - 1];      // start[id] = first slot of id
        std::vector<int32_t> fill(start.begin(), start.end() - 1);
        for (int b = 0; b < B; ++b)
            for (int local = 0; local < L + S; ++local) {
                const int64_t id = id_of(b, local);
                if (id == 0) continue;
                const int32_t cc = cmap[(size_t)b * N + local];
                FIRA_REQUIRE(cc >= 0, "fira_host_node_lists: a node with a non-zero id is not in the computed list");
                emb_rows[fill[(size_t)id]++] = cc;
            }
        for (int64_t id = 1; id <= max_id; ++id) {                 // groups in ascending id order, split into `chunk`-sized items
            const int32_t lo = start[(size_t)id], hi = start[(size_t)id + 1];
            for (int32_t s0 = lo; s0 < hi; s0 += chunk) {
                item_tok[n_items] = (int32_t)id;
                item_ptr[n_items] = s0;
                ++n_items;
            }
        }
        item_ptr[n_items] = n_pos;
    } else {
        item_ptr[0] = 0;
    }
    counts[0] = n_nodes; counts[1] = (int32_t)nnz_c; counts[2] = n_code; counts[3] = n_mem;
    counts[4] = n_items; counts[5] = n_pos; counts[6] = n_ast;
    return 0;
}

// Block-diagonal CSR of a batch from the store's per-commit CSR (data.GraphStore.batch): commit idx[b]'s rows move to
// rows b*N .., its columns into graph b's node block.  rowptr [B*N+1], col / val [sum of the commits' nnz].
extern "C" int fira_host_collate_csr(int B, int N, const int64_t* idx, int64_t n_commits, const int32_t* store_rowptr,
                                     const int64_t* store_offset, const int32_t* store_col, const float* store_val,
                                     int32_t* rowptr, int32_t* col, float* val) {
    FIRA_REQUIRE(B >= 0 && N > 0 && idx && store_rowptr && store_offset && store_col && store_val && rowptr && col && val,
                 "fira_host_collate_csr: bad argument");
    int64_t base = 0;
    for (int b = 0; b < B; ++b) {
        const int64_t r = idx[b];
        FIRA_REQUIRE(r >= 0 && r < n_commits, "fira_host_collate_csr: commit index %lld outside the store", (long long)r);
        const int32_t* rp = store_rowptr + (size_t)r * (N + 1);
        const int64_t off = store_offset[r], nnz = rp[N];
        FIRA_REQUIRE(base + nnz < (1LL << 31), "fira_host_collate_csr: more than 2^31 entries in one batch");
        for (int i = 0; i < N; ++i) rowptr[(size_t)b * N + i] = (int32_t)(rp[i] + base);
        const int32_t shift = b * N;
        for (int64_t e = 0; e < nnz; ++e) {
            col[base + e] = store_col[off + e] + shift;
            val[base + e] = store_val[off + e];
        }
        base += nnz;
    }
    rowptr[(size_t)B * N] = (int32_t)base;
    return 0;
}
