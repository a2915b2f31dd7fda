// Device pieces of Adam shared by the dense kernels, the row-sparse kernels (copyhead.hip) and the forward gathers that read
// a lazily updated embedding table (rowops.hip).  Everything that must agree bit for bit lives here, once.
#pragma once
#include "engine.h"

namespace fira {

// One element's update -- shared by the dense and the row-sparse kernels so that both run the same instruction sequence.
__device__ __forceinline__ void adam_elem(float& p, float& m, float& v, float gi, float beta1, float beta2, float eps,
                                          float step_size, float bc2_sqrt) {
    // (the fused multiply-adds are spelled out and contraction is off: left to the compiler, the dense and the row-sparse
    //  kernel contracted `v * beta2 + t * gi` differently and their results differed in the last bit)
#pragma clang fp contract(off)
    const float mi = __builtin_fmaf(1.0f - beta1, gi - m, m);      // exp_avg.lerp_(grad, 1-beta1)
    const float g2 = (1.0f - beta2) * gi * gi;
    const float vi = __builtin_fmaf(v, beta2, g2);                 // mul_(beta2).addcmul_(g, g, 1-beta2)
    m = mi;
    v = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p = __builtin_fmaf(-step_size, mi / denom, p);
}

__device__ __forceinline__ void adam_row_zero_steps(float4& pv, float4& mv, float4& vv, int from, int to, float gz, float lr,
                                                    float beta1, float beta2, float eps, const AdamRowsHist& h) {
    for (int j = from; j <= to; ++j) {                    // the updates of steps from..to with a zero gradient row
        const float ss = lr / h.bc1[j % ADAM_ROWS_K], b2s = h.bc2s[j % ADAM_ROWS_K];
        adam_elem(pv.x, mv.x, vv.x, gz, beta1, beta2, eps, ss, b2s);
        adam_elem(pv.y, mv.y, vv.y, gz, beta1, beta2, eps, ss, b2s);
        adam_elem(pv.z, mv.z, vv.z, gz, beta1, beta2, eps, ss, b2s);
        adam_elem(pv.w, mv.w, vv.w, gz, beta1, beta2, eps, ss, b2s);
    }
}

// Row `row` of a lazily updated table as a forward pass must see it: the stored row if it is current at step vw.to, else
// the stored row with the zero-gradient updates it still owes applied in registers (nothing is written: the row's next
// adam_rows_kernel / sync does that).  lane * 4 = first of the lane's four columns.
__device__ __forceinline__ float4 adam_rows_load(const float* __restrict__ table, int row, int lane, const AdamRowsView& vw) {
    const size_t o = (size_t)row * FIRA_D + lane * 4;
    float4 pv = *reinterpret_cast<const float4*>(table + o);
    if (vw.last) {
        const int l = max(vw.last[row], vw.to - (ADAM_ROWS_K - 1));
        if (l < vw.to) {                                          // wave-uniform
            float4 mv = *reinterpret_cast<const float4*>(vw.m + o), vv = *reinterpret_cast<const float4*>(vw.v + o);
            adam_row_zero_steps(pv, mv, vv, l + 1, vw.to, vw.gz, vw.lr, vw.beta1, vw.beta2, vw.eps, vw.h);
        }
    }
    return pv;
}

}  // namespace fira
