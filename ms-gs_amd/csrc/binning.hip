// binning.hip — instance emission and tile ranges (App. A.2).
//
// emit_kernel walks the Gaussians in DEPTH ORDER (order[] from the depth sort) and writes one
// (tile id, Gaussian id) pair per tile whose pixel centres the Gaussian's alpha >= 1/255 level set
// reaches, at the slot given by the exclusive scan of the per-Gaussian counts.  The subsequent
// stable sort by tile id (sort.hip) then produces, inside every tile, the reference order
// (depth bits ascending, ties by Gaussian index).
//
// ranges_kernel finds the [start, end) slice of every tile in the sorted key array.
// Both are HBM-streaming: 8 B written per instance (emit), 4 B read per instance (ranges).
#include "msgs_internal.h"

namespace msgs {

namespace {

__global__ __launch_bounds__(256) void emit_kernel(ViewParams vp, int P, const char* __restrict__ geom,
                                                   uint32_t* __restrict__ keys, uint32_t* __restrict__ ids,
                                                   int64_t D) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P) return;
    const GeomLayout L(P);
    const uint32_t* order = reinterpret_cast<const uint32_t*>(geom + L.order);
    const uint32_t* offs = reinterpret_cast<const uint32_t*>(geom + L.offs);
    const uint32_t* tiles = reinterpret_cast<const uint32_t*>(geom + L.tiles);
    const uint2* rect = reinterpret_cast<const uint2*>(geom + L.rect);
    const GaussRec* rec = reinterpret_cast<const GaussRec*>(geom + L.rec);
    const uint32_t gi = order[r];
    const uint32_t count = tiles[gi];
    if (count == 0) return;
    int64_t off = offs[r];
    const int64_t end = min((int64_t)off + count, D);
    const uint2 rc = rect[gi];
    const int minx = rc.x & 0xFFFF, miny = rc.x >> 16, maxx = rc.y & 0xFFFF, maxy = rc.y >> 16;
    const float4 r0 = rec[gi].r0;
    const float conC = rec[gi].r1.x;
    const float tau2 = rec[gi].r2.w;
    const bool test = tau2 > -1.0e38f;
    for (int ty = miny; ty < maxy && off < end; ++ty)
        for (int tx = minx; tx < maxx && off < end; ++tx) {
            const float x0 = (float)(tx * TILE), y0 = (float)(ty * TILE);
            if (!test || levelset_hits_rect(r0.x, r0.y, r0.z, r0.w, conC, tau2, x0, x0 + (TILE - 1), y0, y0 + (TILE - 1))) {
                keys[off] = (uint32_t)(ty * vp.gx + tx);
                ids[off] = gi;
                ++off;
            }
        }
    // Defensive: the count and this loop evaluate the same deterministic predicate, so the slots are
    // always filled exactly; should they ever not be, park the leftovers on a sentinel tile.
    for (; off < end; ++off) { keys[off] = (uint32_t)(vp.gx * vp.gy); ids[off] = gi; }
}

__global__ __launch_bounds__(256) void ranges_kernel(const uint32_t* __restrict__ keys, int64_t D,
                                                     uint2* __restrict__ ranges, int num_tiles) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    const uint32_t t = keys[i];
    if (t >= (uint32_t)num_tiles) return;
    if (i == 0 || keys[i - 1] != t) ranges[t].x = (uint32_t)i;
    if (i == D - 1 || keys[i + 1] != t) ranges[t].y = (uint32_t)(i + 1);
}

}  // namespace

hipError_t launch_emit(const ViewParams& vp, int P, const char* geom, uint32_t* keys, uint32_t* ids, int64_t D,
                       hipStream_t s) {
    if (P == 0 || D == 0) return hipSuccess;
    hipLaunchKernelGGL(emit_kernel, dim3((P + 255) / 256), dim3(256), 0, s, vp, P, geom, keys, ids, D);
    return hipGetLastError();
}

hipError_t launch_ranges(const uint32_t* keys, int64_t D, uint2* ranges, int num_tiles, hipStream_t s) {
    hipError_t e = hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)num_tiles, s);
    if (e != hipSuccess) return e;
    if (D == 0) return hipSuccess;
    hipLaunchKernelGGL(ranges_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, s, keys, D, ranges, num_tiles);
    return hipGetLastError();
}

}  // namespace msgs
