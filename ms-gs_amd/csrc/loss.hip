// Fused photometric loss of the MS-GS train step:
//     loss = (1 - lambda) * mean|img - gt| + lambda * (1 - SSIM(img, gt))          /root/reference/train.py:209-211
// with the reference's SSIM (/root/reference/utils/loss_utils.py:23-63: 11x11 Gaussian window, sigma 1.5, zero
// padding, per-channel, C1 = 0.01^2, C2 = 0.03^2, mean over all elements) and its gradient w.r.t. img, produced in the
// [C,H,W] layout the blend backward consumes.  The reference runs 5 grouped conv2d + ~15 elementwise kernels forward
// and as many again backward; here it is two launches:
//   ssim_stats_kernel   per 32x32 tile and channel: the five windowed moments (separable 11-tap passes through LDS),
//                       the SSIM map, |img-gt|, block partial sums, and the three per-pixel derivative maps
//                       A = dS/dmu1 (total), B = dS/dsigma1^2, Cm = dS/dsigma12
//   ssim_grad_kernel    dL/dimg(q) = gs * [conv(A) + 2 img conv(B) + gt conv(Cm)](q) + gl * sign(img - gt)(q);
//                       its first block also folds the partial sums in a fixed order (deterministic loss value)
// Algorithmic HBM bytes per pixel-channel: stats 8 read + 12 written, grad 20 read + 4 written.
#include "msgs_internal.h"

#pragma clang fp contract(off)

namespace msgs {

constexpr int LT = 32;              // tile edge (outputs)
constexpr int LR = 5;               // window radius
constexpr int LE = LT + 2 * LR;     // 42: tile + halo
constexpr int LIN_STRIDE = LE + 1;  // 43: odd row stride, conflict-free column walks
constexpr int LH_STRIDE = LT + 1;   // 33
constexpr int LOSS_THREADS = 256;

struct SsimWindow { float w[11]; };

void ssim_window_host(float w[11]) {
    // loss_utils.py:23-25: exp(-(x - 5)^2 / (2 sigma^2)) evaluated in double, stored as float32, divided by their
    // float32 sum (torch's sum of the 11 values is the correctly rounded exact sum: accumulate in double, round once)
    double sum = 0.0;
    for (int i = 0; i < 11; ++i) {
        w[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5));
        sum += (double)w[i];
    }
    for (int i = 0; i < 11; ++i) w[i] /= (float)sum;
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_allreduce_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// horizontal 11-tap pass over NM maps held as products of the staged inputs; one work item = (row, 4 adjacent columns)
template <int NM, typename F>
__device__ __forceinline__ void horizontal_pass(const SsimWindow& win, float* hm, F&& taps) {
    for (int item = threadIdx.x; item < LE * (LT / 4); item += LOSS_THREADS) {
        const int r = item / (LT / 4), c0 = (item % (LT / 4)) * 4;
        float acc[NM][4];
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[m][j] = 0.f;
#pragma unroll
        for (int t = 0; t < 14; ++t) {
            float v[NM];
            taps(r, c0 + t, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = t - j;                      // tap index for output column c0 + j
                if (k >= 0 && k < 11) {
#pragma unroll
                    for (int m = 0; m < NM; ++m) acc[m][j] = __fmaf_rn(win.w[k], v[m], acc[m][j]);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) hm[(m * LE + r) * LH_STRIDE + c0 + j] = acc[m][j];
    }
}

// vertical 11-tap pass: thread owns column c and 4 adjacent rows r0..r0+3 of the tile
template <int NM>
__device__ __forceinline__ void vertical_pass(const SsimWindow& win, const float* hm, int c, int r0, float out[NM][4]) {
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) out[m][j] = 0.f;
#pragma unroll
    for (int t = 0; t < 14; ++t) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const float v = hm[(m * LE + r0 + t) * LH_STRIDE + c];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = t - j;
                if (k >= 0 && k < 11) out[m][j] = __fmaf_rn(win.w[k], v, out[m][j]);
            }
        }
    }
}

__global__ __launch_bounds__(LOSS_THREADS) void ssim_stats_kernel(const float* __restrict__ img,
                                                                   const float* __restrict__ gt, int H, int W,
                                                                   SsimWindow win, float* __restrict__ maps,
                                                                   float* __restrict__ partials, int write_maps) {
    __shared__ float sx[LE * LIN_STRIDE], sy[LE * LIN_STRIDE];
    __shared__ float hm[5 * LE * LH_STRIDE];
    __shared__ float red[4];
    const int ch = blockIdx.z;
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT;
    const size_t plane = (size_t)H * W;
    const float* X = img + ch * plane;
    const float* Y = gt + ch * plane;
    float l1 = 0.f;
    for (int i = threadIdx.x; i < LE * LE; i += LOSS_THREADS) {
        const int r = i / LE, c = i % LE;
        const int y = y0 + r - LR, x = x0 + c - LR;
        float a = 0.f, b = 0.f;
        if (x >= 0 && x < W && y >= 0 && y < H) {
            a = X[(size_t)y * W + x];
            b = Y[(size_t)y * W + x];
            if (r >= LR && r < LR + LT && c >= LR && c < LR + LT) l1 += fabsf(a - b);
        }
        sx[r * LIN_STRIDE + c] = a;
        sy[r * LIN_STRIDE + c] = b;
    }
    __syncthreads();
    horizontal_pass<5>(win, hm, [&](int r, int c, float v[5]) {
        const float a = sx[r * LIN_STRIDE + c], b = sy[r * LIN_STRIDE + c];
        v[0] = a; v[1] = b; v[2] = a * a; v[3] = b * b; v[4] = a * b;
    });
    __syncthreads();
    const int c = threadIdx.x & 31, r0 = (threadIdx.x >> 5) * 4;
    float mo[5][4];
    vertical_pass<5>(win, hm, c, r0, mo);
    constexpr float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    float ssim_sum = 0.f;
    const int x = x0 + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = y0 + r0 + j;
        if (x < W && y < H) {
            const float m1 = mo[0][j], m2 = mo[1][j];
            const float m1s = m1 * m1, m2s = m2 * m2, m12 = m1 * m2;
            const float s1 = mo[2][j] - m1s, s2 = mo[3][j] - m2s, s12 = mo[4][j] - m12;
            const float a1 = 2.f * m12 + C1, a2 = 2.f * s12 + C2;
            const float b1 = m1s + m2s + C1, b2 = s1 + s2 + C2;
            const float inv = 1.f / (b1 * b2);
            const float S = (a1 * a2) * inv;
            ssim_sum += S;
            if (write_maps) {
                const float dS_ds12 = 2.f * a1 * inv;
                const float dS_ds1 = -S / b2;
                const float dS_dm1 = 2.f * m2 * a2 * inv - 2.f * m1 * S / b1;
                const size_t o = (size_t)y * W + x;
                maps[(0 * gridDim.z + ch) * plane + o] = dS_dm1 - 2.f * m1 * dS_ds1 - m2 * dS_ds12;
                maps[(1 * gridDim.z + ch) * plane + o] = dS_ds1;
                maps[(2 * gridDim.z + ch) * plane + o] = dS_ds12;
            }
        }
    }
    const float bs = block_sum_256(ssim_sum, red);
    const float bl = block_sum_256(l1, red);
    if (threadIdx.x == 0) {
        const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partials[2 * b] = bs;
        partials[2 * b + 1] = bl;
    }
}

// out[0] = loss, out[1] = L1 mean, out[2] = SSIM mean; fixed summation order -> run-to-run identical
__device__ void finalize_loss(const float* partials, int nblocks, float lambda, float inv_n, float* out, float* red) {
    double s = 0.0, l = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += LOSS_THREADS) {
        s += (double)partials[2 * i];
        l += (double)partials[2 * i + 1];
    }
    __shared__ double dred[2 * LOSS_THREADS];
    dred[threadIdx.x] = s;
    dred[LOSS_THREADS + threadIdx.x] = l;
    __syncthreads();
    for (int k = LOSS_THREADS / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            dred[threadIdx.x] += dred[threadIdx.x + k];
            dred[LOSS_THREADS + threadIdx.x] += dred[LOSS_THREADS + threadIdx.x + k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float ssim = (float)(dred[0] * (double)inv_n), l1m = (float)(dred[LOSS_THREADS] * (double)inv_n);
        out[0] = (1.f - lambda) * l1m + lambda * (1.f - ssim);
        out[1] = l1m;
        out[2] = ssim;
    }
}

__global__ __launch_bounds__(LOSS_THREADS) void loss_finalize_kernel(const float* partials, int nblocks, float lambda,
                                                                      float inv_n, float* out) {
    __shared__ float red[4];
    finalize_loss(partials, nblocks, lambda, inv_n, out, red);
}

// gs = -lambda / N * upstream, gl = (1 - lambda) / N * upstream; upstream read from device memory (nullable = 1)
__global__ __launch_bounds__(LOSS_THREADS) void ssim_grad_kernel(const float* __restrict__ img,
                                                                  const float* __restrict__ gt, int H, int W,
                                                                  SsimWindow win, const float* __restrict__ maps,
                                                                  float lambda, float inv_n,
                                                                  const float* __restrict__ upstream,
                                                                  float* __restrict__ dL_dimg) {
    __shared__ float sm[3 * LE * LIN_STRIDE];
    __shared__ float hm[3 * LE * LH_STRIDE];
    const int ch = blockIdx.z;
    const int x0 = blockIdx.x * LT, y0 = blockIdx.y * LT;
    const size_t plane = (size_t)H * W;
    for (int i = threadIdx.x; i < LE * LE; i += LOSS_THREADS) {
        const int r = i / LE, c = i % LE;
        const int y = y0 + r - LR, x = x0 + c - LR;
        const bool in = x >= 0 && x < W && y >= 0 && y < H;
        const size_t o = (size_t)y * W + x;
#pragma unroll
        for (int m = 0; m < 3; ++m) sm[(m * LE + r) * LIN_STRIDE + c] = in ? maps[(m * gridDim.z + ch) * plane + o] : 0.f;
    }
    __syncthreads();
    horizontal_pass<3>(win, hm, [&](int r, int c, float v[3]) {
#pragma unroll
        for (int m = 0; m < 3; ++m) v[m] = sm[(m * LE + r) * LIN_STRIDE + c];
    });
    __syncthreads();
    const int c = threadIdx.x & 31, r0 = (threadIdx.x >> 5) * 4;
    float g[3][4];
    vertical_pass<3>(win, hm, c, r0, g);
    const float up = upstream ? upstream[0] : 1.f;
    const float gs = -(lambda * inv_n) * up, gl = ((1.f - lambda) * inv_n) * up;
    const int x = x0 + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int y = y0 + r0 + j;
        if (x < W && y < H) {
            const size_t o = ch * plane + (size_t)y * W + x;
            const float a = img[o], b = gt[o];
            const float d = a - b;
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            dL_dimg[o] = gs * (g[0][j] + 2.f * a * g[1][j] + b * g[2][j]) + gl * sgn;
        }
    }
}

size_t loss_scratch_bytes(int C, int H, int W) {
    const size_t plane = (size_t)H * W;
    const size_t nblocks = (size_t)((W + LT - 1) / LT) * ((H + LT - 1) / LT) * C;
    return align256(3 * C * plane * sizeof(float)) + align256(2 * nblocks * sizeof(float));
}

hipError_t launch_loss_forward(const float* img, const float* gt, int C, int H, int W, float lambda, float* out3,
                               char* scratch, int write_maps, hipStream_t s) {
    SsimWindow win;
    ssim_window_host(win.w);
    const size_t plane = (size_t)H * W;
    float* maps = (float*)scratch;
    float* partials = (float*)(scratch + align256(3 * C * plane * sizeof(float)));
    const dim3 grid((W + LT - 1) / LT, (H + LT - 1) / LT, C);
    const int nblocks = (int)(grid.x * grid.y * grid.z);
    hipLaunchKernelGGL(ssim_stats_kernel, grid, dim3(LOSS_THREADS), 0, s, img, gt, H, W, win, maps, partials, write_maps);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(LOSS_THREADS), 0, s, partials, nblocks, lambda,
                       (float)(1.0 / ((double)C * (double)plane)), out3);
    return hipGetLastError();
}

hipError_t launch_loss_backward(const float* img, const float* gt, int C, int H, int W, float lambda,
                                const float* upstream, const char* scratch, float* dL_dimg, hipStream_t s) {
    SsimWindow win;
    ssim_window_host(win.w);
    const dim3 grid((W + LT - 1) / LT, (H + LT - 1) / LT, C);
    hipLaunchKernelGGL(ssim_grad_kernel, grid, dim3(LOSS_THREADS), 0, s, img, gt, H, W, win, (const float*)scratch, lambda,
                       (float)(1.0 / ((double)C * (double)H * (double)W)), upstream, dL_dimg);
    return hipGetLastError();
}

}  // namespace msgs
