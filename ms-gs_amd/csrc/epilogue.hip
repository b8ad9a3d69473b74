// Train-step epilogue: the elementwise passes the reference runs over all P Gaussians after
// loss.backward() — the Adam update of the leaf tensors (/root/reference/scene/gaussian_model.py:235-248,
// train.py:416-418: torch.optim.Adam(l, lr=0.0, eps=1e-15).step()) and the densification statistics
// (train.py:239-250, gaussian_model.py:663-704) — each as ONE HBM-streaming launch.
#include "msgs_internal.h"

#pragma clang fp contract(off)

namespace msgs {

// ------------------------------------------------------------------------------------------------------------
// Multi-tensor Adam.  One launch covers every tensor of the optimizer; a workgroup owns ADAM_CHUNK consecutive
// floats of one tensor.  Arithmetic follows torch.optim.Adam's single-tensor formulation term by term
// (torch/optim/adam.py, _single_tensor_adam; amsgrad = False, weight_decay = 0, maximize = False):
//     m <- fma(1 - beta1, g - m, m)                      (Tensor.lerp_, weight < 0.5 branch)
//     v <- fma((1 - beta2) * g, g, v * beta2)            (mul_ then addcmul_)
//     p <- p + ((-lr / (1 - beta1^t)) * m) / (sqrt(v) / sqrt(1 - beta2^t) + eps)      (addcdiv_)
// with the roundings of the ATen CPU kernels, so that the numpy oracle (pinned to torch.optim.Adam) is matched.
// 28 algorithmic bytes per float: read p, g, m, v; write p, m, v.
// ------------------------------------------------------------------------------------------------------------
constexpr int ADAM_THREADS = 256;
constexpr int ADAM_VEC = 4;
constexpr int ADAM_ITERS = 4;
constexpr int ADAM_CHUNK = ADAM_THREADS * ADAM_VEC * ADAM_ITERS;   // 4096 floats per workgroup

struct AdamSlot {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;
    int vec_ok;
};

struct AdamTable {
    AdamSlot t[MSGS_ADAM_MAX_TENSORS];
    uint32_t first_block[MSGS_ADAM_MAX_TENSORS + 1];
    float neg_step_size[MSGS_ADAM_MAX_TENSORS];     // -lr / bias_correction1, rounded from double like torch's scalar
    int n;
    AdamScalars s;                                  // (adam_update: msgs_internal.h — shared with the per-Gaussian backward)
};

__global__ __launch_bounds__(ADAM_THREADS) void adam_multi_kernel(const AdamTable a) {
    int ti = 0;
#pragma unroll
    for (int k = 1; k < MSGS_ADAM_MAX_TENSORS; ++k) ti += (k < a.n && blockIdx.x >= a.first_block[k]) ? 1 : 0;
    const AdamSlot T = a.t[ti];
    const float nss = a.neg_step_size[ti];
    const int64_t base = (int64_t)(blockIdx.x - a.first_block[ti]) * ADAM_CHUNK;
    float* __restrict__ P = T.param;
    const float* __restrict__ G = T.grad;
    float* __restrict__ M = T.exp_avg;
    float* __restrict__ V = T.exp_avg_sq;
    if (T.vec_ok && base + ADAM_CHUNK <= T.n) {       // full chunk, 16-B aligned: dwordx4 streams
        float4 p[ADAM_ITERS], g[ADAM_ITERS], m[ADAM_ITERS], v[ADAM_ITERS];
#pragma unroll
        for (int it = 0; it < ADAM_ITERS; ++it) {
            const int64_t i = base + (int64_t)(it * ADAM_THREADS + threadIdx.x) * ADAM_VEC;
            p[it] = *reinterpret_cast<const float4*>(P + i);
            g[it] = *reinterpret_cast<const float4*>(G + i);
            m[it] = *reinterpret_cast<const float4*>(M + i);
            v[it] = *reinterpret_cast<const float4*>(V + i);
        }
#pragma unroll
        for (int it = 0; it < ADAM_ITERS; ++it) {
            adam_update(p[it].x, g[it].x, m[it].x, v[it].x, nss, a.s);
            adam_update(p[it].y, g[it].y, m[it].y, v[it].y, nss, a.s);
            adam_update(p[it].z, g[it].z, m[it].z, v[it].z, nss, a.s);
            adam_update(p[it].w, g[it].w, m[it].w, v[it].w, nss, a.s);
            const int64_t i = base + (int64_t)(it * ADAM_THREADS + threadIdx.x) * ADAM_VEC;
            *reinterpret_cast<float4*>(P + i) = p[it];
            *reinterpret_cast<float4*>(M + i) = m[it];
            *reinterpret_cast<float4*>(V + i) = v[it];
        }
        return;
    }
    for (int k = threadIdx.x; k < ADAM_CHUNK; k += ADAM_THREADS) {     // ragged tail / unaligned tensor
        const int64_t i = base + k;
        if (i >= T.n) break;
        float p = P[i], m = M[i], v = V[i];
        adam_update(p, G[i], m, v, nss, a.s);
        P[i] = p; M[i] = m; V[i] = v;
    }
}

hipError_t launch_adam(const msgs_adam_tensor_t* tensors, int n, int64_t step, double beta1, double beta2, double eps,
                       hipStream_t s) {
    AdamTable a{};
    a.n = n;
    uint32_t blocks = 0;
    for (int k = 0; k < n; ++k) {
        a.t[k] = AdamSlot{tensors[k].param, tensors[k].grad, tensors[k].exp_avg, tensors[k].exp_avg_sq, tensors[k].n, 0};
        const uintptr_t bits = (uintptr_t)tensors[k].param | (uintptr_t)tensors[k].grad | (uintptr_t)tensors[k].exp_avg |
                               (uintptr_t)tensors[k].exp_avg_sq;
        a.t[k].vec_ok = (bits & 15) == 0;
        a.first_block[k] = blocks;
        blocks += (uint32_t)((tensors[k].n + ADAM_CHUNK - 1) / ADAM_CHUNK);
        a.neg_step_size[k] = adam_neg_step_size(tensors[k].lr, step, beta1);
    }
    for (int k = n; k <= MSGS_ADAM_MAX_TENSORS; ++k) a.first_block[k] = blocks;
    a.s = adam_scalars(step, beta1, beta2, eps);
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks), dim3(ADAM_THREADS), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Densification statistics: one thread per Gaussian, every update the reference performs between backward() and
// optimizer.step(), selected by `flags`.  visibility_filter = radii > 0 (gaussian_renderer/__init__.py:117).
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void densify_stats_kernel(const msgs_densify_stats_t d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.P) return;
    const int r = d.radii[i];
    if (r <= 0) return;                                   // every update below is masked by visibility_filter
    if (d.flags & MSGS_STATS_BASE_MASK) d.base_mask[i] = 1;                       // gaussian_model.py:702-704
    if (d.flags & MSGS_STATS_PIXEL_SIZES) {                                        // gaussian_model.py:663-687
        if (d.target_reso_lvl[i] == (int64_t)d.reso_lvl) {
            const float ps = d.pixel_sizes[i];
            if (d.reso_lvl > 0) {
                const float decayed = d.max_pixel_sizes[i] * 0.95f;
                d.max_pixel_sizes[i] = fmaxf(decayed, ps);                         // torch.max propagates NaN: none occur
            }
            if (d.reso_lvl < d.reso_lvls - 1) {
                const float grown = fmaxf(d.min_pixel_sizes[i] * 1.05f, -1.f);     // torch.clip(x, -1)
                float out = grown;
                if (ps > 0.f) out = (grown < 0.f) ? ps : fminf(grown, ps);
                d.min_pixel_sizes[i] = out;
            }
        }
    }
    if (d.flags & MSGS_STATS_DENSIFY) {                                            // train.py:247-250
        d.max_radii2D[i] = fmaxf(d.max_radii2D[i], (float)r);
        const float gx = d.means2D_grad[3 * i + 0], gy = d.means2D_grad[3 * i + 1];
        const int64_t k = (int64_t)i * d.reso_lvls + d.reso_lvl;                   // [P, reso_lvls, 1]
        d.xyz_gradient_accum[k] += sqrtf(gx * gx + gy * gy);                       // gaussian_model.py:698-700
        d.denom[k] += 1.f;
    }
}

hipError_t launch_densify_stats(const msgs_densify_stats_t& d, hipStream_t s) {
    if (d.P == 0) return hipSuccess;
    hipLaunchKernelGGL(densify_stats_kernel, dim3((d.P + 255) / 256), dim3(256), 0, s, d);
    return hipGetLastError();
}

}  // namespace msgs
