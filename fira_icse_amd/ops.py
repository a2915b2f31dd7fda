"""Thin tensor-level wrappers over the op-level C ABI (one reference op each; see include/fira_hip.h).

They take/return ``torch`` CUDA tensors, launch on the current stream and never fall back to PyTorch maths.
Used by the piecewise drop-in surface (``model.encoder`` / ``decoder`` / ``out_fc`` / ``copy_net``) and by the
per-kernel parity tests.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, cur_stream, ptr


def _f32(t):
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "expected a contiguous fp32 CUDA tensor"
    return t


def _i32(t):
    assert t.is_cuda and t.dtype == torch.int32 and t.is_contiguous(), "expected a contiguous int32 CUDA tensor"
    return t


def gemm(A, B, *, transA=False, transB=True, bias=None, relu=False, out=None, accumulate=False, splitk=1,
         tile=0, dtype="f32"):
    """C = op(A) op(B) (+bias)(relu).  transB=True: B is an nn.Linear weight [N,K].  dtype "bf16": operands rounded to
    bf16 on the way to the matrix cores, fp32 accumulation and storage (fira_gemm_bf16)."""
    for t in (A, B):
        assert t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1
    M, K = (A.shape[1], A.shape[0]) if transA else A.shape
    N = B.shape[0] if transB else B.shape[1]
    assert (B.shape[1] if transB else B.shape[0]) == K
    if out is None:
        out = torch.zeros((M, N), dtype=torch.float32, device=A.device) if accumulate else \
            torch.empty((M, N), dtype=torch.float32, device=A.device)
    flags = (1 if relu else 0) | (2 if accumulate else 0) | (tile << 4)     # tile: 0 auto, 1 128x128, 2 64x128, 3 64x64
    fn = _lib.lib().fira_gemm_bf16 if dtype == "bf16" else _lib.lib().fira_gemm_f32
    check(fn(cur_stream(), int(transA), int(transB), M, N, K, ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(out),
             out.stride(0), ptr(bias), flags, splitk), "fira_gemm_%s" % dtype)
    return out


def weight_shadow(W):
    """bf16 shadows of a 2-D fp32 weight: (as stored [rows, cols], transposed [cols, rows]), raw bf16 in int16 tensors."""
    rows, cols = W.shape
    wb = torch.empty((rows, cols), dtype=torch.int16, device=W.device)
    wbt = torch.empty((cols, rows), dtype=torch.int16, device=W.device)
    check(_lib.lib().fira_weight_shadow(cur_stream(), rows, cols, ptr(_f32(W)), ptr(wb), ptr(wbt)), "fira_weight_shadow")
    return wb, wbt


def gemm_wb(A, Bb, *, bias=None, relu=False, out=None, accumulate=False, splitk=1):
    """C = A . Bb^T with Bb a bf16 shadow [N, K] (int16 tensor, any row pitch that is a multiple of 8)."""
    M, K = A.shape
    N = Bb.shape[0]
    assert Bb.shape[1] == K and Bb.dtype == torch.int16 and Bb.stride(1) == 1
    if out is None:
        out = (torch.zeros if accumulate else torch.empty)((M, N), dtype=torch.float32, device=A.device)
    flags = (1 if relu else 0) | (2 if accumulate else 0)
    check(_lib.lib().fira_gemm_bf16_wb(cur_stream(), M, N, K, ptr(A), A.stride(0), ptr(Bb), Bb.stride(0), ptr(out),
                                       out.stride(0), ptr(bias), flags, splitk), "fira_gemm_bf16_wb")
    return out


def csr_spmm(rowptr, col, val, X, graph_rows=0, variant=0, out=None, dtype=0, auto=False):
    """Z = A_hat X over a block-diagonal CSR.  variant 1/2: CSR gather kernels, 3/4: block-dense fp32 / bf16 MFMA;
    ``auto`` (with variant 0): the library picks by density (fira_csr_spmm), ``dtype`` 1 allows bf16 operands."""
    for t in (X,) + ((out,) if out is not None else ()):      # rows may be strided (a column block of a wider buffer)
        assert t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1
    Y = torch.empty_like(X, memory_format=torch.contiguous_format) if out is None else out
    assert Y.shape == X.shape
    if auto or variant >= 3:
        check(_lib.lib().fira_csr_spmm(cur_stream(), X.shape[0], int(col.numel()), ptr(_i32(rowptr)), ptr(_i32(col)),
                                       ptr(_f32(val)), ptr(X), X.stride(0), ptr(Y), Y.stride(0), graph_rows, variant, dtype),
              "fira_csr_spmm")
        return Y
    check(_lib.lib().fira_csr_spmm_f32(cur_stream(), X.shape[0], ptr(_i32(rowptr)), ptr(_i32(col)), ptr(_f32(val)),
                                       ptr(X), X.stride(0), ptr(Y), Y.stride(0), graph_rows, variant),
          "fira_csr_spmm_f32")
    return Y


def embed_gather(idx, table, pos=None, out=None, out_bstride=None, out_off=0):
    B, L = idx.shape
    if out is None:
        out = torch.empty((B, L, 256), dtype=torch.float32, device=table.device)
        out_bstride = L
    check(_lib.lib().fira_embed_gather_fwd(cur_stream(), B, L, ptr(_i32(idx)), ptr(_f32(table)), ptr(pos), ptr(out),
                                           out_bstride, out_off), "fira_embed_gather_fwd")
    return out


def embed_scatter_add(idx, dtable, dout, out_bstride, out_off, padding_idx):
    B, L = idx.shape
    check(_lib.lib().fira_embed_gather_bwd(cur_stream(), B, L, ptr(_i32(idx)), ptr(_f32(dtable)), ptr(_f32(dout)),
                                           out_bstride, out_off, padding_idx), "fira_embed_gather_bwd")
    return dtable


def combination_fwd(qk, vtab, mark, dropout=0.0, seed=0, site=0):
    M = qk.shape[0]
    out = torch.empty((M, 256), dtype=torch.float32, device=qk.device)
    check(_lib.lib().fira_combination_fwd(cur_stream(), M, ptr(_f32(qk)), ptr(_f32(vtab)), ptr(_i32(mark)), ptr(out),
                                          dropout, seed, site), "fira_combination_fwd")
    return out


def combination_bwd(qk, vtab, mark, dout, dropout=0.0, seed=0, site=0):
    M = qk.shape[0]
    dqk = torch.empty_like(qk)
    dvtab = torch.zeros_like(vtab)
    check(_lib.lib().fira_combination_bwd(cur_stream(), M, ptr(_f32(qk)), ptr(_f32(vtab)), ptr(_i32(mark)),
                                          ptr(_f32(dout)), ptr(dqk), ptr(dvtab), dropout, seed, site),
          "fira_combination_bwd")
    return dqk, dvtab


def add_layernorm_fwd(x, res, gamma, beta, dropout=0.0, seed=0, site=0):
    """Returns (y, pre-norm sum, stats[M,2]); ``x`` is consumed (overwritten with the sum)."""
    M = x.shape[0]
    y = torch.empty_like(x)
    stats = torch.empty((M, 2), dtype=torch.float32, device=x.device)
    check(_lib.lib().fira_add_layernorm_fwd(cur_stream(), M, ptr(_f32(x)), ptr(res), ptr(_f32(gamma)), ptr(_f32(beta)),
                                            ptr(y), ptr(stats), dropout, seed, site), "fira_add_layernorm_fwd")
    return y, x, stats


def linear_presum(x, w, bias, res, dropout=0.0, seed=0, site=0):
    """Pre-norm sum of a residual block: dropout(x @ w.T + bias) + res  (w [256, K])."""
    M, K = x.shape
    assert w.shape == (256, K) and x.stride(1) == 1
    s = torch.empty((M, 256), dtype=torch.float32, device=x.device)
    check(_lib.lib().fira_linear_presum_f32(cur_stream(), M, K, ptr(x), x.stride(0), ptr(_f32(w)), ptr(bias), ptr(_f32(res)),
                                            ptr(s), dropout, seed, site), "fira_linear_presum_f32")
    return s


def ln_linear(s, w, bias, gamma, beta, relu=False):
    """(LN(s) @ w.T + bias [relu], LN(s), stats[M,2]) in one launch: the LayerNorm runs in the product's prologue."""
    M = s.shape[0]
    N = w.shape[0]
    assert w.shape[1] == 256 and s.shape[1] == 256 and s.stride(1) == 1
    y = torch.empty((M, N), dtype=torch.float32, device=s.device)
    x = torch.empty((M, 256), dtype=torch.float32, device=s.device)
    stats = torch.empty((M, 2), dtype=torch.float32, device=s.device)
    check(_lib.lib().fira_ln_linear_f32(cur_stream(), M, N, ptr(s), s.stride(0), ptr(_f32(w)), ptr(bias), ptr(y), N, int(relu),
                                        ptr(_f32(gamma)), ptr(_f32(beta)), ptr(x), ptr(stats)), "fira_ln_linear_f32")
    return y, x, stats


def linear_layernorm_bf16_fwd(x, wb, bias, res, gamma, beta, dropout=0.0, seed=0, site=0):
    """bf16 twin of linear_layernorm_fwd: ``wb`` is the [256, 256] bf16 shadow (int16 tensor) of the weight."""
    M, K = x.shape
    assert K == 256 and wb.shape == (256, 256) and wb.dtype == torch.int16 and x.stride(1) == 1
    y = torch.empty((M, 256), dtype=torch.float32, device=x.device)
    s = torch.empty_like(y)
    stats = torch.empty((M, 2), dtype=torch.float32, device=x.device)
    check(_lib.lib().fira_linear_layernorm_bf16_fwd(cur_stream(), M, ptr(x), x.stride(0), ptr(wb), wb.stride(0), ptr(bias),
                                                    ptr(res), ptr(_f32(gamma)), ptr(_f32(beta)), ptr(s), ptr(y), ptr(stats),
                                                    dropout, seed, site), "fira_linear_layernorm_bf16_fwd")
    return y, s, stats


def add_layernorm_bwd(dy, s, stats, gamma, dropout=0.0, seed=0, site=0, want_dx_drop=False):
    M = dy.shape[0]
    ds = torch.empty_like(dy)
    dxd = torch.empty_like(dy) if want_dx_drop else None
    dg = torch.zeros(256, dtype=torch.float32, device=dy.device)
    db = torch.zeros(256, dtype=torch.float32, device=dy.device)
    check(_lib.lib().fira_add_layernorm_bwd(cur_stream(), M, ptr(_f32(dy)), ptr(_f32(s)), ptr(_f32(stats)),
                                            ptr(_f32(gamma)), ptr(ds), ptr(dxd), ptr(dg), ptr(db), dropout, seed, site),
          "fira_add_layernorm_bwd")
    return ds, dxd, dg, db


def ln_bwd_linear(dy, Wt, summ, stats, gamma, relu_mask=None, dropout=0.0, seed=0, site=0):
    """fira_ln_bwd_linear_f32: (dX, ds, dx_drop, dgamma, dbeta) -- LayerNorm backward in the prologue of dX = dx_drop . Wt."""
    M, N = dy.shape[0], Wt.shape[1]
    dX = torch.empty((M, N), dtype=torch.float32, device=dy.device)
    ds, dxd = torch.empty_like(dy), torch.empty_like(dy)
    dg = torch.zeros(256, dtype=torch.float32, device=dy.device)
    db = torch.zeros(256, dtype=torch.float32, device=dy.device)
    part = torch.empty(((M + 31) // 32) * 512, dtype=torch.float32, device=dy.device)
    check(_lib.lib().fira_ln_bwd_linear_f32(cur_stream(), M, N, ptr(_f32(dy)), ptr(_f32(Wt)), ptr(dX),
                                            ptr(None if relu_mask is None else _f32(relu_mask)), ptr(_f32(summ)),
                                            ptr(_f32(stats)), ptr(_f32(gamma)), ptr(ds), ptr(dxd), ptr(dg), ptr(db), ptr(part),
                                            dropout, seed, site), "fira_ln_bwd_linear_f32")
    return dX, ds, dxd, dg, db


def dropout_mask(seed, site, n, p, device="cuda"):
    """Scale factors (0 or 1/(1-p)) of dropout site ``site`` for element indices 0..n-1 under ``seed``."""
    out = torch.empty(n, dtype=torch.float32, device=device)
    check(_lib.lib().fira_dropout_mask(cur_stream(), seed, site, n, p, ptr(out)), "fira_dropout_mask")
    return out


def colsum(X):
    out = torch.zeros(X.shape[1], dtype=torch.float32, device=X.device)
    check(_lib.lib().fira_colsum_f32(cur_stream(), X.shape[0], X.shape[1], ptr(_f32(X)), X.stride(0), ptr(out)),
          "fira_colsum_f32")
    return out


def gcn_weight_planes(B):
    """fira_gcn_weight_planes: the three bf16 planes (fragment order) of n stacked [256, 256] fp32 matrices B[n][k] for the
    FIRA_F32X3 form of fira_gcn_layer_{fwd,bwd} (out = U B^T)."""
    B = _f32(B).contiguous()
    n = B.numel() // (256 * 256)
    planes = torch.empty(n * 3 * 256 * 256, dtype=torch.int16, device=B.device)
    check(_lib.lib().fira_gcn_weight_planes(cur_stream(), n, ptr(B), ptr(planes)), "fira_gcn_weight_planes")
    return planes


def linear_x3(x, W, bias=None, dtype=2, ldo=None):
    """fira_linear_x3: out = x W^T + bias, x [M, 256], W [N, 256] with N a multiple of 256, on the bf16 matrix cores -- dtype 2:
    three bf16 terms per operand (fp32-accurate), 3: one plane (the bf16 mode's rounding).  Returns out [M, N] (row pitch ldo)."""
    x = _f32(x)
    M, N = x.shape[0], W.shape[0]
    planes = gcn_weight_planes(W)
    ldo = N if ldo is None else ldo
    out = torch.zeros((M, ldo), dtype=torch.float32, device=x.device)
    check(_lib.lib().fira_linear_x3(cur_stream(), M, N, ptr(x), x.stride(0), ptr(planes), ptr(None if bias is None else _f32(bias)),
                                    ptr(out), ldo, dtype), "fira_linear_x3")
    return out[:, :N]


def linear_dgrad_x3(dy, W, dx=None, dtype=2):
    """fira_linear_dgrad_x3: dx (+)= dy W for W [K, 256] (K a multiple of 256), dy [M, K]; dx None: a fresh result."""
    assert dy.is_cuda and dy.dtype == torch.float32 and dy.stride(1) == 1          # (a column slice of wider rows is fine)
    M, K = dy.shape
    Wt = _f32(W).view(K // 256, 256, 256).transpose(1, 2).contiguous()
    planes = gcn_weight_planes(Wt)
    acc = dx is not None
    if dx is None:
        dx = torch.empty((M, 256), dtype=torch.float32, device=dy.device)
    check(_lib.lib().fira_linear_dgrad_x3(cur_stream(), M, K, ptr(dy), dy.stride(0), ptr(planes), ptr(dx), dx.stride(0), 1 if acc else 0,
                                          dtype), "fira_linear_dgrad_x3")
    return dx


def dgrad_x3_splitk(dy, W, dx, dtype=2):
    """fira_dgrad_x3_splitk: dx += dy W, dy [M, K] (row pitch a multiple of 4 floats), W [K, 256], K any size."""
    assert dy.is_cuda and dy.dtype == torch.float32 and dy.stride(1) == 1
    M, K = dy.shape
    ws = torch.empty(_lib.lib().fira_dgrad_x3_splitk_planes_bytes(K), dtype=torch.uint8, device=dy.device)
    check(_lib.lib().fira_dgrad_x3_splitk(cur_stream(), M, K, ptr(dy), dy.stride(0), ptr(_f32(W)), ptr(ws), ptr(dx), dx.stride(0),
                                          dtype), "fira_dgrad_x3_splitk")
    return dx


def gcn_layer_fwd(rowptr, col, val, X, W21t, bias, c21, gamma, beta, dropout=0.0, seed=0, site=0, dtype=0, want_rowsum=True):
    """fira_gcn_layer_fwd: (sum, y, stats, rowsum) of one folded GCN layer on the CSR adjacency (global column ids);
    W21t = W21^T contiguous.  dtype 2 / 3 (FIRA_F32X3 / FIRA_BF16X1): the planes of W21 are formed here."""
    n = X.shape[0]
    if dtype >= 2:
        W21t = gcn_weight_planes(_f32(W21t).t().contiguous())
    else:
        W21t = _f32(W21t)
    summ, y = torch.empty_like(X), torch.empty_like(X)
    stats = torch.empty((n, 2), dtype=torch.float32, device=X.device)
    rs = torch.empty(n, dtype=torch.float32, device=X.device) if want_rowsum else None
    check(_lib.lib().fira_gcn_layer_fwd(cur_stream(), n, ptr(_i32(rowptr)), ptr(_i32(col)), ptr(_f32(val)), ptr(_f32(X)),
                                        ptr(W21t), ptr(_f32(bias)), ptr(_f32(c21)), ptr(_f32(gamma)), ptr(_f32(beta)),
                                        ptr(summ), ptr(y), ptr(stats), ptr(rs), dropout, seed, site, dtype),
          "fira_gcn_layer_fwd")
    return summ, y, stats, rs


def combination_block_fwd(Xc, Wqk, bqk, Wo, bo, vtab, mark, gamma, beta, dropout=0.0, seed=0, site_gate=0, site_out=0, y=None,
                          y_rows=None, dtype=0):
    """fira_combination_block_fwd: (qk, c, sum, y, stats) of one Combination block on the code rows Xc [n,256]; Wqk [512,256]
    and Wo [256,256] as nn.Linear stores them (transposed here: the kernel streams k-major copies); vtab [4, >=256] rows of the
    projected mark table (row stride = vtab.stride(0)).  y / y_rows: optional node buffer and row map for the output rows."""
    n = Xc.shape[0]
    dev = Xc.device
    assert vtab.is_cuda and vtab.dtype == torch.float32 and vtab.stride(1) == 1 and vtab.shape[0] == 4, "vtab: fp32 rows"
    if dtype >= 2:          # FIRA_F32X3 / FIRA_BF16X1: planes of Wq | Wk | Wo as stored
        WqT = WkT = WoT = gcn_weight_planes(torch.cat([_f32(Wqk), _f32(Wo)], 0))
    else:
        WqT, WkT, WoT = (_f32(Wqk[:256]).t().contiguous(), _f32(Wqk[256:]).t().contiguous(), _f32(Wo).t().contiguous())
    qk = torch.empty((n, 512), dtype=torch.float32, device=dev)
    c, summ = torch.empty_like(Xc), torch.empty_like(Xc)
    if y is None:
        y = torch.empty_like(Xc)
    stats = torch.empty((n, 2), dtype=torch.float32, device=dev)
    yr = None if y_rows is None else _i32(y_rows)
    check(_lib.lib().fira_combination_block_fwd(cur_stream(), n, ptr(_f32(Xc)), ptr(WqT), ptr(WkT), ptr(WoT), ptr(_f32(bqk)),
                                                ptr(_f32(bo)), ptr(vtab), vtab.stride(0), ptr(_i32(mark)), ptr(qk), ptr(c),
                                                ptr(_f32(gamma)), ptr(_f32(beta)), ptr(summ), ptr(y), ptr(yr), ptr(stats),
                                                dropout, seed, site_gate, site_out, dtype), "fira_combination_block_fwd")
    return qk, c, summ, y, stats


def combination_block_bwd(dG, rows, summ, stats, gamma, Wo, Wqk, qk, vtab, mark, dropout=0.0, seed=0, site_gate=0, site_out=0,
                          dtype=0):
    """fira_combination_block_bwd: dG [N,256] is updated in place at rows[r]; returns (dYc, dqk, dgamma, dbeta, dvtab)."""
    n = summ.shape[0]
    dev = summ.device
    assert vtab.is_cuda and vtab.dtype == torch.float32 and vtab.stride(1) == 1 and vtab.shape[0] == 4, "vtab: fp32 rows"
    dYc = torch.empty((n, 256), dtype=torch.float32, device=dev)
    dqk = torch.empty((n, 512), dtype=torch.float32, device=dev)
    dgamma, dbeta = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
    dvtab = torch.zeros((4, 256), device=dev)
    part = torch.empty(_lib.lib().fira_combination_block_bwd_part_floats(), dtype=torch.float32, device=dev)
    if dtype >= 2:          # FIRA_F32X3 / FIRA_BF16X1: planes of Wq^T | Wk^T | Wo^T
        Wo = gcn_weight_planes(torch.stack([_f32(Wqk[:256]).t(), _f32(Wqk[256:]).t(), _f32(Wo).t()]).contiguous())
    else:
        Wo = _f32(Wo)
    check(_lib.lib().fira_combination_block_bwd(cur_stream(), n, ptr(_f32(dG)), ptr(_i32(rows)), ptr(_f32(summ)), ptr(_f32(stats)),
                                                ptr(_f32(gamma)), ptr(Wo), ptr(_f32(Wqk)), ptr(_f32(qk)), ptr(vtab),
                                                vtab.stride(0), ptr(_i32(mark)), ptr(dYc), ptr(dqk), ptr(dgamma), ptr(dbeta),
                                                ptr(dvtab), 256, ptr(part), dropout, seed, site_gate, site_out, dtype),
          "fira_combination_block_bwd")
    return dYc, dqk, dgamma, dbeta, dvtab


def gcn_layer_bwd(rowptr, col, val, dY, W21, dX, dtype=0):
    """fira_gcn_layer_bwd: V = A_hat dY (returned), dX += V W21 in place."""
    V = torch.empty_like(dY)
    W = gcn_weight_planes(_f32(W21).t().contiguous()) if dtype >= 2 else _f32(W21)     # (FIRA_F32X3: out = V B^T with B = W21^T)
    check(_lib.lib().fira_gcn_layer_bwd(cur_stream(), dY.shape[0], ptr(_i32(rowptr)), ptr(_i32(col)), ptr(_f32(val)),
                                        ptr(_f32(dY)), ptr(W), ptr(V), ptr(_f32(dX)), dtype), "fira_gcn_layer_bwd")
    return V


def attention_ragged_fwd(q, k, v, key_valid, q_off, Tq, Tk, causal=False, self_kv=False, dtype=0, heads=8, k_off=None):
    """The engine's form (fira_attention_fwd_ex): q [R,256] compact query rows, commit b's are q_off[b] .. q_off[b+1]
    (<= Tq of them); k / v [B*Tk,256] dense, or the same compact rows with self_kv, or (k_off [B+1]) RAGGED key rows:
    commit b's keys are rows k_off[b] .. k_off[b+1] of k / v and key_valid is a flat mask over those rows;
    dtype 1 = bf16 MFMA."""
    B = q_off.numel() - 1
    o = torch.zeros_like(q)
    ko = None if k_off is None else _i32(k_off)
    check(_lib.lib().fira_attention_fwd_ex(cur_stream(), B, heads, Tq, Tk, ptr(q), q.stride(0), ptr(k), k.stride(0), ptr(v),
                                           v.stride(0), ptr(_i32(key_valid)), int(causal), 0, ptr(o), 256, ptr(_i32(q_off)),
                                           int(self_kv), dtype, ptr(ko)), "fira_attention_fwd_ex")
    return o


def attention_ragged_bwd(q, k, v, key_valid, o, do, q_off, Tq, Tk, causal=False, self_kv=False, dtype=0, heads=8, k_off=None,
                         fill=0.0):
    """fill: value the gradient buffers hold before the call (the kernel must overwrite every row it owns)."""
    B = q_off.numel() - 1
    dq, dk, dv = torch.full_like(q, fill), torch.full_like(k, fill), torch.full_like(v, fill)
    ko = None if k_off is None else _i32(k_off)
    check(_lib.lib().fira_attention_bwd_ex(cur_stream(), B, heads, Tq, Tk, ptr(q), q.stride(0), ptr(k), k.stride(0), ptr(v),
                                           v.stride(0), ptr(_i32(key_valid)), int(causal), 0, ptr(_f32(o)), 256,
                                           ptr(_f32(do)), 256, ptr(dq), dq.stride(0), ptr(dk), dk.stride(0), ptr(dv),
                                           dv.stride(0), ptr(_i32(q_off)), int(self_kv), dtype, ptr(ko)),
          "fira_attention_bwd_ex")
    return dq, dk, dv


def attention_fwd(q, k, v, key_valid, causal=False, q_pos0=0, heads=8):
    """q [B,Tq,256], k/v [B,Tk,256] (any row stride), key_valid int32 [B,Tk] -> o [B,Tq,256]."""
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    o = torch.empty((B, Tq, 256), dtype=torch.float32, device=q.device)
    check(_lib.lib().fira_attention_fwd(cur_stream(), B, heads, Tq, Tk, ptr(q), q.stride(1), ptr(k), k.stride(1),
                                        ptr(v), v.stride(1), ptr(_i32(key_valid)), int(causal), q_pos0, ptr(o), 256),
          "fira_attention_fwd")
    return o


def decode_attention(q, k, v, key_valid, tk=None, qpk=1, heads=8, knew=None, vnew=None, k_off=None):
    """One query per row: q [BR,256] (any row stride), k / v [BR/qpk, kb, 256] (any row stride), key_valid int32
    [BR/qpk, kvb] -> o [BR,256].  knew / vnew [BR,256] (row stride = their stride(0)): key tk-1 of every row comes from
    them and is appended to k / v in place.  k_off [BR/qpk + 1]: ragged key rows -- k / v are [rows, 256] (any row stride),
    entry e's keys are rows k_off[e] .. k_off[e+1], key_valid a flat mask over those rows (tk = the largest range)."""
    BR = q.shape[0]
    if k_off is not None:
        kb = kvb = tk
        ldk, ldv = k.stride(0), v.stride(0)
    else:
        kb, kvb = k.shape[1], key_valid.shape[1]
        ldk, ldv = k.stride(1), v.stride(1)
    tk = kb if tk is None else tk
    o = torch.empty((BR, 256), dtype=torch.float32, device=q.device)
    ko = None if k_off is None else _i32(k_off)
    check(_lib.lib().fira_decode_attention(cur_stream(), BR, heads, tk, ptr(q), q.stride(0), ptr(k), ldk, ptr(v),
                                           ldv, ptr(_i32(key_valid)), ptr(o), 256, kb, kvb, qpk, ptr(knew),
                                           ptr(vnew), knew.stride(0) if knew is not None else 0,
                                           ptr(k) if knew is not None else None, ptr(v) if knew is not None else None,
                                           ptr(ko)),
          "fira_decode_attention")
    return o


def attention_bwd(q, k, v, key_valid, o, do, causal=False, q_pos0=0, heads=8):
    B, Tq, _ = q.shape
    Tk = k.shape[1]
    dq = torch.empty((B, Tq, 256), dtype=torch.float32, device=q.device)
    dk = torch.empty((B, Tk, 256), dtype=torch.float32, device=q.device)
    dv = torch.empty((B, Tk, 256), dtype=torch.float32, device=q.device)
    check(_lib.lib().fira_attention_bwd(cur_stream(), B, heads, Tq, Tk, ptr(q), q.stride(1), ptr(k), k.stride(1),
                                        ptr(v), v.stride(1), ptr(_i32(key_valid)), int(causal), q_pos0, ptr(_f32(o)),
                                        256, ptr(_f32(do)), 256, ptr(dq), 256, ptr(dk), 256, ptr(dv), 256),
          "fira_attention_bwd")
    return dq, dk, dv


def copy_score_fwd(src, tgt, w, bias):
    B, S, _ = src.shape
    T = tgt.shape[1]
    score = torch.empty((B, T, S), dtype=torch.float32, device=src.device)
    check(_lib.lib().fira_copy_score_fwd(cur_stream(), B, T, S, ptr(_f32(src)), ptr(_f32(tgt)), ptr(_f32(w)),
                                         ptr(_f32(bias)), ptr(score)), "fira_copy_score_fwd")
    return score


def copy_score_bwd(src, tgt, w, dscore):
    B, S, _ = src.shape
    T = tgt.shape[1]
    dsrc = torch.empty_like(src)
    dtgt = torch.zeros_like(tgt)
    dw = torch.zeros(256, dtype=torch.float32, device=src.device)
    db = torch.zeros(1, dtype=torch.float32, device=src.device)
    check(_lib.lib().fira_copy_score_bwd(cur_stream(), B, T, S, ptr(_f32(src)), ptr(_f32(tgt)), ptr(_f32(w)),
                                         ptr(_f32(dscore)), ptr(dsrc), ptr(dtgt), ptr(dw), ptr(db)),
          "fira_copy_score_bwd")
    return dsrc, dtgt, dw, db


def head_loss(logits, score, mem_valid, gate_logits, tar_label, V, compact_row=None, want_grad=True, argmax=False):
    """In place: logits/score/gate_logits become their gradients when want_grad.  Returns (loss_sum, n_tok, ids)."""
    B, T = tar_label.shape
    S = score.shape[-1]
    loss = torch.zeros(1, dtype=torch.float32, device=logits.device)
    ntok = torch.zeros(1, dtype=torch.int32, device=logits.device)
    ids = torch.empty((B, T), dtype=torch.int32, device=logits.device) if argmax else None
    check(_lib.lib().fira_head_loss(cur_stream(), B * T, T, V, S, ptr(compact_row), ptr(_f32(logits)),
                                    logits.stride(0), ptr(_f32(score)), ptr(_i32(mem_valid)), ptr(_f32(gate_logits)),
                                    ptr(_i32(tar_label)), ptr(loss), ptr(ntok), ptr(ids), int(want_grad)),
          "fira_head_loss")
    return loss, ntok, ids


def adam_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, inv_scale=None):
    check(_lib.lib().fira_adam_step(cur_stream(), p.numel(), ptr(_f32(p)), ptr(_f32(g)), ptr(_f32(m)), ptr(_f32(v)),
                                    lr, beta1, beta2, eps, step, ptr(inv_scale)), "fira_adam_step")


def adam_step_count(p, g, m, v, lr, step, count, beta1=0.9, beta2=0.999, eps=1e-8):
    """Adam on g / max(count, 1): ``count`` is a device float (the all-reduced token count of a data-parallel step)."""
    check(_lib.lib().fira_adam_step_count(cur_stream(), p.numel(), ptr(_f32(p)), ptr(_f32(g)), ptr(_f32(m)), ptr(_f32(v)),
                                          lr, beta1, beta2, eps, step, ptr(_f32(count))), "fira_adam_step_count")


def pack_stats(loss_sum, n_tok, out2):
    """out2 = [loss_sum, float(n_tok)] on the device: what a rank contributes to the step's 2-scalar all-reduce."""
    check(_lib.lib().fira_pack_stats(cur_stream(), ptr(_f32(loss_sum)), ptr(_i32(n_tok)), ptr(_f32(out2))), "fira_pack_stats")
    return out2


def adam_step_mb(p, g0, g1, m, v, lr, step, n_tok0, n_tok1=None, beta1=0.9, beta2=0.999, eps=1e-8):
    """Adam on g0 (+ g1) / max(n_tok0 (+ n_tok1), 1): one launch, normaliser formed on the device."""
    check(_lib.lib().fira_adam_step_mb(cur_stream(), p.numel(), ptr(_f32(p)), ptr(_f32(g0)), ptr(g1), ptr(_f32(m)),
                                       ptr(_f32(v)), lr, beta1, beta2, eps, step, ptr(_i32(n_tok0)), ptr(n_tok1)),
          "fira_adam_step_mb")


def inv_count(n_tok, out):
    check(_lib.lib().fira_inv_count(cur_stream(), ptr(_i32(n_tok)), ptr(_f32(out))), "fira_inv_count")
    return out


def f32_to_bf16(src, dst):
    """dst (torch.bfloat16, same numel) = bf16(src) on the current stream: the gradient wire format (fira_f32_to_bf16)."""
    n = src.numel()
    assert dst.numel() == n and dst.dtype == torch.bfloat16 and src.dtype == torch.float32 and n % 4 == 0
    check(_lib.lib().fira_f32_to_bf16(cur_stream(), n, ptr(src), ptr(dst)), "fira_f32_to_bf16")
    return dst


def bf16_to_f32(src, dst):
    n = src.numel()
    assert dst.numel() == n and src.dtype == torch.bfloat16 and dst.dtype == torch.float32 and n % 4 == 0
    check(_lib.lib().fira_bf16_to_f32(cur_stream(), n, ptr(src), ptr(dst)), "fira_bf16_to_f32")
    return dst


def ffn_fwd(x, w1, b1, w2, b2, gamma, beta, dropout=0.0, seed=0, site=0, dtype=0):
    """fira_ffn_fwd: (h, sum, y, stats) of the FeedForward block LN(dropout(relu(x W1^T + b1) W2^T + b2) + x)."""
    M, F = x.shape[0], w1.shape[0]
    h = torch.empty((M, F), dtype=torch.float32, device=x.device)
    summ, y = torch.empty_like(x), torch.empty_like(x)
    stats = torch.empty((M, 2), dtype=torch.float32, device=x.device)
    check(_lib.lib().fira_ffn_fwd(cur_stream(), M, F, ptr(_f32(x)), ptr(_f32(w1)), ptr(_f32(b1)), ptr(_f32(w2)), ptr(_f32(b2)),
                                  ptr(_f32(gamma)), ptr(_f32(beta)), ptr(h), ptr(summ), ptr(y), ptr(stats), dropout, seed, site,
                                  dtype), "fira_ffn_fwd")
    return h, summ, y, stats


def ffn_bwd(dy, x, h, summ, stats, w1, w2, gamma, dropout=0.0, seed=0, site=0, dtype=0):
    """fira_ffn_bwd: (dx, dw1, db1, dw2, db2, dgamma, dbeta) -- the parameter gradients start from zero here."""
    M, F = x.shape[0], w1.shape[0]
    dev = x.device
    z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
    dx, dyf, dh = torch.empty_like(x), torch.empty_like(x), torch.empty((M, F), dtype=torch.float32, device=dev)
    dw1, db1, dw2, db2, dg, db = z(F, 256), z(F), z(256, F), z(256), z(256), z(256)
    check(_lib.lib().fira_ffn_bwd(cur_stream(), M, F, ptr(_f32(dy)), ptr(_f32(x)), ptr(_f32(h)), ptr(_f32(summ)), ptr(_f32(stats)),
                                  ptr(_f32(w1)), ptr(_f32(w2)), ptr(_f32(gamma)), ptr(dx), ptr(dyf), ptr(dh), ptr(dw1), ptr(db1),
                                  ptr(dw2), ptr(db2), ptr(dg), ptr(db), dropout, seed, site, dtype), "fira_ffn_bwd")
    return dx, dw1, db1, dw2, db2, dg, db


def head_topk(x, wout, bout, k, dtype=0):
    """fira_head_topk: (ids [R,k] int32, vals [R,k], logits [R,V]) of the generator head x Wout^T + bout."""
    R, V = x.shape[0], wout.shape[0]
    ldl = (V + 63) // 64 * 64
    logits = torch.empty((R, ldl), dtype=torch.float32, device=x.device)
    ids = torch.empty((R, k), dtype=torch.int32, device=x.device)
    vals = torch.empty((R, k), dtype=torch.float32, device=x.device)
    check(_lib.lib().fira_head_topk(cur_stream(), R, V, k, ptr(_f32(x)), ptr(_f32(wout)), ptr(_f32(bout)), ptr(logits), ldl,
                                    ptr(ids), ptr(vals), dtype), "fira_head_topk")
    return ids, vals, logits[:, :V]


def head_logits_x3(x, wout, bout):
    """fira_head_logits_x3: logits [R, V] = x Wout^T + bout on the bf16 matrix cores at fp32 accuracy (three bf16 terms)."""
    R, V = x.shape[0], wout.shape[0]
    ldl = (V + 63) // 64 * 64
    logits = torch.empty((R, ldl), dtype=torch.float32, device=x.device)
    scratch = torch.empty(_lib.lib().fira_head_logits_x3_scratch_bytes(R), dtype=torch.uint8, device=x.device)
    x = _f32(x)
    check(_lib.lib().fira_head_logits_x3(cur_stream(), R, V, ptr(x), x.stride(0), ptr(_f32(wout)), ptr(_f32(bout)), ptr(logits), ldl,
                                         ptr(scratch)), "fira_head_logits_x3")
    return logits[:, :V]


_panel_scratch = {}


def gemm_wgrad_panel(A, B, out, colsum=None, dtype=0, split=True):
    """fira_gemm_wgrad_panel: out[M,N] += A^T B (A [K, >=M], B [K, N] row-major, row pitches = their strides), optional
    colsum[M] += column sums of A.  dtype 0: fp32-accurate three-term bf16 split, 1: bf16-rounded operands.  ``split``: hand
    the library a 64 MB scratch buffer so that it may cut K into slabs over the chip."""
    K, M, N = A.shape[0], out.shape[0], out.shape[1]
    assert B.shape[0] == K and B.shape[1] == N and A.stride(1) == 1 and B.stride(1) == 1 and out.stride(1) == 1
    scratch = None
    if split:
        scratch = _panel_scratch.get(A.device)
        if scratch is None:
            scratch = _panel_scratch[A.device] = torch.empty(256 * 65536, dtype=torch.float32, device=A.device)
    check(_lib.lib().fira_gemm_wgrad_panel(cur_stream(), dtype, M, N, K, ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(out),
                                           out.stride(0), ptr(colsum), ptr(scratch), scratch.numel() if split else 0),
          "fira_gemm_wgrad_panel")
    return out
