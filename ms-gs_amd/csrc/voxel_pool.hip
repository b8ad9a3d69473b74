// voxel_pool.hip — GPU voxel-average pooling for the "insert large Gaussians" step of MS-GS
// (north-star "large-Gaussian insertion").
//
// Replaces the eleven CPU calls of open3d.ml.torch.layers.VoxelPooling(position_fn='center',
// feature_fn='average') in /root/reference/scene/gaussian_model.py:802-816 (each preceded by a .cpu() and followed
// by a .cuda()).  open3d is a third-party dependency that is not vendored in the reference (environment.yml only);
// its published behaviour is restated here: a point belongs to voxel floor(p / voxel_size) per axis, pooled
// features are the arithmetic mean over the voxel's points, pooled positions are the voxel centres
// ('center') or the mean position ('average').  The order of the output voxels is not specified by open3d; here
// voxels come out in ascending (z, y, x) voxel-index order and points inside a voxel are summed in ascending
// point-id order, so the result is deterministic (no atomics).
//
// The grouping is built ONCE per position set and reused for every feature tensor:
//   voxel key (3 x 21 bit) -> stable LSD radix sort on the low 32 bits, then on the high 31 bits (sort.hip)
//   -> segment heads -> exclusive scan -> segment starts.
// Every kernel is HBM-streaming; the whole build moves ~100 B per point.
#include "msgs_internal.h"

namespace msgs {

namespace {

constexpr int VP_BIAS = 1 << 20;     // voxel indices are clamped to [-2^20, 2^20)

__global__ void vp_keys_kernel(const float* __restrict__ pos, int64_t M, float inv_vs, uint32_t* __restrict__ lo,
                               uint32_t* __restrict__ hi) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    uint64_t key = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float v = floorf(pos[3 * i + a] * inv_vs);
        v = fminf(fmaxf(v, (float)-VP_BIAS), (float)(VP_BIAS - 1));
        key |= (uint64_t)((int)v + VP_BIAS) << (21 * a);            // x: bits 0-20, y: 21-41, z: 42-62
    }
    lo[i] = (uint32_t)key;
    hi[i] = (uint32_t)(key >> 32);
}

__global__ void vp_gather_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ idx, int64_t M,
                                 uint32_t* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) dst[i] = src[idx[i]];
}

// flags[j] = 1 where sorted element j starts a new voxel
__global__ void vp_heads_kernel(const uint32_t* __restrict__ hi_sorted, const uint32_t* __restrict__ lo,
                                const uint32_t* __restrict__ order, int64_t M, uint32_t* __restrict__ flags) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    bool head = j == 0;
    if (!head) head = hi_sorted[j] != hi_sorted[j - 1] || lo[order[j]] != lo[order[j - 1]];
    flags[j] = head ? 1u : 0u;
}

__global__ void vp_segments_kernel(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ excl,
                                   const uint32_t* __restrict__ hi_sorted, const uint32_t* __restrict__ lo,
                                   const uint32_t* __restrict__ order, int64_t M, uint32_t* __restrict__ seg_start,
                                   int32_t* __restrict__ voxel_index) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    if (flags[j]) {
        const uint32_t v = excl[j];
        seg_start[v] = (uint32_t)j;
        if (voxel_index) {
            const uint64_t key = ((uint64_t)hi_sorted[j] << 32) | lo[order[j]];
#pragma unroll
            for (int a = 0; a < 3; ++a) voxel_index[3 * (int64_t)v + a] = (int)((key >> (21 * a)) & 0x1FFFFFu) - VP_BIAS;
        }
    }
    if (j == M - 1) seg_start[excl[j] + flags[j]] = (uint32_t)M;     // sentinel end
}

// one thread per (voxel, feature): members are summed in ascending point-id order
__global__ void vp_mean_kernel(const float* __restrict__ feat, int F, const uint32_t* __restrict__ order,
                               const uint32_t* __restrict__ seg_start, int64_t Mv, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Mv * F) return;
    const int64_t v = t / F;
    const int f = (int)(t - v * F);
    const uint32_t a = seg_start[v], b = seg_start[v + 1];
    float s = 0.f;
    for (uint32_t m = a; m < b; ++m) s += feat[(int64_t)order[m] * F + f];
    out[t] = s / (float)(b - a);
}

struct VpScratch {
    size_t lo, hi, lo_s, hi_g, hi_s, idx1, flags, excl, sort, partials, total_out, total;
    explicit VpScratch(int64_t M) {
        const size_t n = (size_t)(M > 0 ? M : 1);
        size_t o = 0;
        lo = o;        o = align256(o + 4 * n);
        hi = o;        o = align256(o + 4 * n);
        lo_s = o;      o = align256(o + 4 * n);
        hi_g = o;      o = align256(o + 4 * n);
        hi_s = o;      o = align256(o + 4 * n);
        idx1 = o;      o = align256(o + 4 * n);
        flags = o;     o = align256(o + 4 * n);
        excl = o;      o = align256(o + 4 * n);
        sort = o;      o = align256(o + SortScratch((int64_t)n).total);
        partials = o;  o = align256(o + 8 * (size_t)(scan_blocks((int64_t)n) + 2));
        total_out = o; o = align256(o + 64);
        total = o;
    }
};

}  // namespace

size_t voxel_pool_scratch_bytes(int64_t M) { return VpScratch(M).total; }

hipError_t voxel_pool_build(const float* positions, int64_t M, float voxel_size, uint32_t* order, uint32_t* seg_start,
                            int32_t* voxel_index, char* scratch, int64_t* num_voxels_host, hipStream_t s) {
    *num_voxels_host = 0;
    if (M <= 0) return hipSuccess;
    const VpScratch L(M);
    uint32_t* lo = (uint32_t*)(scratch + L.lo);
    uint32_t* hi = (uint32_t*)(scratch + L.hi);
    uint32_t* lo_s = (uint32_t*)(scratch + L.lo_s);
    uint32_t* hi_g = (uint32_t*)(scratch + L.hi_g);
    uint32_t* hi_s = (uint32_t*)(scratch + L.hi_s);
    uint32_t* idx1 = (uint32_t*)(scratch + L.idx1);
    uint32_t* flags = (uint32_t*)(scratch + L.flags);
    uint32_t* excl = (uint32_t*)(scratch + L.excl);
    uint64_t* total_dev = (uint64_t*)(scratch + L.total_out);
    const unsigned nb = (unsigned)((M + 255) / 256);
    hipLaunchKernelGGL(vp_keys_kernel, dim3(nb), dim3(256), 0, s, positions, M, 1.0f / voxel_size, lo, hi);
    hipError_t e = radix_sort_pairs(lo, nullptr, lo_s, idx1, M, 0, 32, scratch + L.sort, s);   // by low 32 bits
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(vp_gather_kernel, dim3(nb), dim3(256), 0, s, hi, idx1, M, hi_g);
    e = radix_sort_pairs(hi_g, idx1, hi_s, order, M, 0, 31, scratch + L.sort, s);               // stable, by high 31 bits
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(vp_heads_kernel, dim3(nb), dim3(256), 0, s, hi_s, lo, order, M, flags);
    e = exclusive_scan_u32(flags, nullptr, excl, M, (uint64_t*)(scratch + L.partials), total_dev, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(vp_segments_kernel, dim3(nb), dim3(256), 0, s, flags, excl, hi_s, lo, order, M, seg_start,
                       voxel_index);
    uint64_t total = 0;
    e = hipMemcpyAsync(&total, total_dev, sizeof(uint64_t), hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    *num_voxels_host = (int64_t)total;
    return hipGetLastError();
}

hipError_t voxel_pool_average(const float* features, int F, const uint32_t* order, const uint32_t* seg_start,
                              int64_t Mv, float* out, hipStream_t s) {
    if (Mv <= 0 || F <= 0) return hipSuccess;
    const int64_t n = Mv * F;
    hipLaunchKernelGGL(vp_mean_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, features, F, order, seg_start,
                       Mv, out);
    return hipGetLastError();
}

}  // namespace msgs
