// distCUDA2: mean squared distance from every point to its 3 nearest neighbours — the initial Gaussian scale of
// GaussianModel.create_from_pcd (/root/reference/scene/gaussian_model.py:199).  The reference takes it from the
// un-vendored submodule simple-knn (/root/reference/.gitmodules); its published behaviour is restated: exact 3-NN
// over all other points (a point is excluded only by its own index, duplicates count with distance 0), squared
// Euclidean distance in float32, mean of the three.
//
// MI355X design: 30-bit Morton sort of the points (the library's radix sort), then one wave per 64 consecutive
// sorted points ("box"): the wave's 64 queries are spatially coherent, so the branch-and-bound over the two-level
// box hierarchy (superbox = 64 boxes) is nearly wave-uniform; superboxes are visited outwards from the wave's own,
// which tightens the bound at once.  Candidate points are read at wave-uniform addresses (one broadcast fetch for
// 64 lanes).  Exact for any distribution; the bound only prunes.
#include "msgs_internal.h"

#include <cfloat>

#pragma clang fp contract(off)

namespace msgs {
namespace {

constexpr int KB = 64;          // points per box = one wave
constexpr int KS = 64;          // boxes per superbox

struct KnnScratch {
    size_t bbox, keys, keys_s, ids, pts, box_lo, box_hi, sup_lo, sup_hi, sort, total;
    explicit KnnScratch(int64_t P) {
        const int64_t nb = (P + KB - 1) / KB, ns = (nb + KS - 1) / KS;
        size_t o = 0;
        bbox = o;   o = align256(o + 6 * sizeof(uint32_t));
        keys = o;   o = align256(o + 4 * (size_t)P);
        keys_s = o; o = align256(o + 4 * (size_t)P);
        ids = o;    o = align256(o + 4 * (size_t)P);
        pts = o;    o = align256(o + 16 * (size_t)P);
        box_lo = o; o = align256(o + 16 * (size_t)nb);
        box_hi = o; o = align256(o + 16 * (size_t)nb);
        sup_lo = o; o = align256(o + 16 * (size_t)ns);
        sup_hi = o; o = align256(o + 16 * (size_t)ns);
        sort = o;   o = align256(o + SortScratch(P).total);
        total = o;
    }
};

// order-preserving float <-> uint (for atomicMin / atomicMax on floats of either sign)
__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

__global__ void knn_bbox_init_kernel(uint32_t* bbox) {
    if (threadIdx.x < 3) bbox[threadIdx.x] = 0xFFFFFFFFu;
    else if (threadIdx.x < 6) bbox[threadIdx.x] = 0u;
}

__global__ __launch_bounds__(256) void knn_bbox_kernel(const float* __restrict__ p, int64_t P, uint32_t* bbox) {
    __shared__ float s_lo[4][3], s_hi[4][3];
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x)
        for (int k = 0; k < 3; ++k) {
            const float v = p[3 * i + k];
            lo[k] = fminf(lo[k], v);
            hi[k] = fmaxf(hi[k], v);
        }
    for (int k = 0; k < 3; ++k) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
        if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6][k] = lo[k]; s_hi[threadIdx.x >> 6][k] = hi[k]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {                                   // 6 atomics per workgroup on 6 addresses
        const int k = threadIdx.x;
        atomicMin(&bbox[k], f2ord(fminf(fminf(s_lo[0][k], s_lo[1][k]), fminf(s_lo[2][k], s_lo[3][k]))));
        atomicMax(&bbox[3 + k], f2ord(fmaxf(fmaxf(s_hi[0][k], s_hi[1][k]), fmaxf(s_hi[2][k], s_hi[3][k]))));
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t x) {     // 10 bits -> every third bit
    x = (x | (x << 16)) & 0x030000FFu;
    x = (x | (x << 8)) & 0x0300F00Fu;
    x = (x | (x << 4)) & 0x030C30C3u;
    x = (x | (x << 2)) & 0x09249249u;
    return x;
}

__global__ __launch_bounds__(256) void knn_morton_kernel(const float* __restrict__ p, int64_t P,
                                                         const uint32_t* __restrict__ bbox, uint32_t* keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    uint32_t code = 0;
    for (int k = 0; k < 3; ++k) {
        const float lo = ord2f(bbox[k]), hi = ord2f(bbox[3 + k]);
        const float ext = hi - lo;
        float t = ext > 0.f ? (p[3 * i + k] - lo) / ext : 0.f;
        t = fminf(fmaxf(t, 0.f), 1.f);                          // NaN -> 0
        const uint32_t q = min((uint32_t)(t * 1023.f), 1023u);
        code |= spread10(q) << (2 - k);
    }
    keys[i] = code;
}

// sorted copy of the points (float4: xyz + original index bits) and the per-box AABBs
__global__ __launch_bounds__(KB) void knn_boxes_kernel(const float* __restrict__ p, const uint32_t* __restrict__ ids,
                                                        int64_t P, float4* __restrict__ pts, float4* box_lo,
                                                        float4* box_hi) {
    const int64_t r = (int64_t)blockIdx.x * KB + threadIdx.x;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (r < P) {
        const uint32_t id = ids[r];
        const float x = p[3 * (size_t)id], y = p[3 * (size_t)id + 1], z = p[3 * (size_t)id + 2];
        pts[r] = make_float4(x, y, z, __uint_as_float(id));
        lo[0] = hi[0] = x; lo[1] = hi[1] = y; lo[2] = hi[2] = z;
    }
    for (int k = 0; k < 3; ++k)
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    if (threadIdx.x == 0) {
        box_lo[blockIdx.x] = make_float4(lo[0], lo[1], lo[2], 0.f);
        box_hi[blockIdx.x] = make_float4(hi[0], hi[1], hi[2], 0.f);
    }
}

__global__ __launch_bounds__(KS) void knn_superboxes_kernel(const float4* __restrict__ box_lo,
                                                             const float4* __restrict__ box_hi, int64_t nb,
                                                             float4* sup_lo, float4* sup_hi) {
    const int64_t b = (int64_t)blockIdx.x * KS + threadIdx.x;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (b < nb) {
        const float4 l = box_lo[b], h = box_hi[b];
        lo[0] = l.x; lo[1] = l.y; lo[2] = l.z; hi[0] = h.x; hi[1] = h.y; hi[2] = h.z;
    }
    for (int k = 0; k < 3; ++k)
        for (int off = 32; off > 0; off >>= 1) {
            lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
            hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
        }
    if (threadIdx.x == 0) {
        sup_lo[blockIdx.x] = make_float4(lo[0], lo[1], lo[2], 0.f);
        sup_hi[blockIdx.x] = make_float4(hi[0], hi[1], hi[2], 0.f);
    }
}

__device__ __forceinline__ float aabb_dist2(float x, float y, float z, float4 lo, float4 hi) {
    const float dx = fmaxf(fmaxf(lo.x - x, x - hi.x), 0.f);
    const float dy = fmaxf(fmaxf(lo.y - y, y - hi.y), 0.f);
    const float dz = fmaxf(fmaxf(lo.z - z, z - hi.z), 0.f);
    return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ void insert3(float d, float& b0, float& b1, float& b2) {
    if (d < b2) {
        if (d < b1) {
            b2 = b1;
            if (d < b0) { b1 = b0; b0 = d; } else b1 = d;
        } else b2 = d;
    }
}

__device__ __forceinline__ void scan_box(const float4* __restrict__ pts, int64_t first, int count, int64_t self,
                                         float x, float y, float z, float& b0, float& b1, float& b2) {
    for (int j = 0; j < count; ++j) {
        const float4 q = pts[first + j];                       // wave-uniform address
        const float dx = q.x - x, dy = q.y - y, dz = q.z - z;
        float d = dx * dx + dy * dy + dz * dz;
        if (first + j == self) d = FLT_MAX;                    // excluded by index only
        insert3(d, b0, b1, b2);
    }
}

__global__ __launch_bounds__(KB) void knn_query_kernel(const float4* __restrict__ pts, int64_t P,
                                                        const float4* __restrict__ box_lo,
                                                        const float4* __restrict__ box_hi, int64_t nb,
                                                        const float4* __restrict__ sup_lo,
                                                        const float4* __restrict__ sup_hi, int64_t ns,
                                                        float* __restrict__ mean_dist2) {
    const int64_t box = blockIdx.x;
    const int64_t r = box * KB + threadIdx.x;
    const bool live = r < P;
    const float4 me = pts[live ? r : P - 1];
    const float x = me.x, y = me.y, z = me.z;
    float b0 = FLT_MAX, b1 = FLT_MAX, b2 = FLT_MAX;
    const int64_t self = live ? r : -1;
    // own box first, then its two neighbours in Morton order: a tight bound before any pruning test
    for (int64_t b = box - 1; b <= box + 1; ++b)
        if (b >= 0 && b < nb) scan_box(pts, b * KB, (int)min((int64_t)KB, P - b * KB), self, x, y, z, b0, b1, b2);
    const int64_t own_sup = box / KS;
    for (int64_t step = 0; step < 2 * ns; ++step) {            // own superbox, then alternately outwards
        const int64_t k = (step + 1) / 2;
        const int64_t S = (step & 1) ? own_sup - k : own_sup + k;
        if (S < 0 || S >= ns) continue;
        const bool want_s = live && aabb_dist2(x, y, z, sup_lo[S], sup_hi[S]) < b2;
        if (!__any(want_s)) continue;
        const int64_t bend = min(nb, (S + 1) * KS);
        for (int64_t b = S * KS; b < bend; ++b) {
            if (b >= box - 1 && b <= box + 1) continue;        // already scanned
            const bool want = live && aabb_dist2(x, y, z, box_lo[b], box_hi[b]) < b2;
            if (!__any(want)) continue;
            scan_box(pts, b * KB, (int)min((int64_t)KB, P - b * KB), self, x, y, z, b0, b1, b2);
        }
    }
    if (live) mean_dist2[__float_as_uint(me.w)] = (b0 + b1 + b2) / 3.0f;
}

}  // namespace

size_t knn_scratch_bytes(int64_t P) { return KnnScratch(P > 0 ? P : 1).total; }

hipError_t knn_mean_dist2(const float* points, int64_t P, float* mean_dist2, char* scratch, hipStream_t s) {
    const KnnScratch L(P);
    const int64_t nb = (P + KB - 1) / KB, ns = (nb + KS - 1) / KS;
    uint32_t* bbox = (uint32_t*)(scratch + L.bbox);
    uint32_t* keys = (uint32_t*)(scratch + L.keys);
    uint32_t* keys_s = (uint32_t*)(scratch + L.keys_s);
    uint32_t* ids = (uint32_t*)(scratch + L.ids);
    float4* pts = (float4*)(scratch + L.pts);
    float4 *blo = (float4*)(scratch + L.box_lo), *bhi = (float4*)(scratch + L.box_hi);
    float4 *slo = (float4*)(scratch + L.sup_lo), *shi = (float4*)(scratch + L.sup_hi);
    const unsigned g256 = (unsigned)((P + 255) / 256);
    hipLaunchKernelGGL(knn_bbox_init_kernel, dim3(1), dim3(64), 0, s, bbox);
    hipLaunchKernelGGL(knn_bbox_kernel, dim3(g256 < 512u ? g256 : 512u), dim3(256), 0, s, points, P, bbox);
    hipLaunchKernelGGL(knn_morton_kernel, dim3(g256), dim3(256), 0, s, points, P, bbox, keys);
    hipError_t e = radix_sort_pairs(keys, nullptr, keys_s, ids, P, 0, 32, scratch + L.sort, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(knn_boxes_kernel, dim3((unsigned)nb), dim3(KB), 0, s, points, ids, P, pts, blo, bhi);
    hipLaunchKernelGGL(knn_superboxes_kernel, dim3((unsigned)ns), dim3(KS), 0, s, blo, bhi, nb, slo, shi);
    hipLaunchKernelGGL(knn_query_kernel, dim3((unsigned)nb), dim3(KB), 0, s, pts, P, blo, bhi, nb, slo, shi, ns, mean_dist2);
    return hipGetLastError();
}

}  // namespace msgs
