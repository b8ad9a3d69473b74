// cost of launching capacity-sized grids whose blocks exit at once (speculative stage 2 / slab B): us per launch by grid size
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(const unsigned* n, unsigned* out) {
    if (blockIdx.x * 4096u >= *n) return;
    out[blockIdx.x] = threadIdx.x;
}
int main() {
    unsigned *n, *out;
    hipMalloc(&n, 4); hipMalloc(&out, 4 << 20);
    unsigned one = 1; hipMemcpy(n, &one, 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (unsigned g : {1u, 256u, 2048u, 4096u, 8192u, 16384u, 32768u, 65536u}) {
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, n, out);
        hipDeviceSynchronize();
        hipEventRecord(a, 0);
        for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, n, out);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("grid %6u: %.2f us per launch (20 back to back)\n", g, 1000.f * ms / 20);
    }
    return 0;
}
