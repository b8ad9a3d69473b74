// launch_chain_calib.hip — what does a chain of N DEPENDENT small kernels cost on gfx950, launched (a) one by one on a stream,
// (b) as one captured hipGraph, (c) as a hipGraph whose kernel-node parameters are rewritten before every launch
// (hipGraphExecKernelNodeSetParams: what a library that receives fresh pointers per call would have to do)?
// The small configurations of the rasterizer (C1 / C2, pyramid levels k >= 3) are bound by a chain of ~32 dependent launches
// (DESIGN §5.4); this measures whether a graph would shorten it.  Every kernel is one wave of `blocks` workgroups that reads the
// word its predecessor wrote (a real dependency), does `work` dependent FMAs, and writes its own word.
// Build: hipcc --offload-arch=gfx950 -O3 tools/launch_chain_calib.hip -o tools/launch_chain_calib ; run: tools/launch_chain_calib
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void link(const float* in, float* out, int work) {
    float v = in[0] + threadIdx.x * 1e-6f;
    for (int i = 0; i < work; ++i) v = fmaf(v, 0.999f, 1e-3f);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = v;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float* buf;
    CHECK(hipMalloc(&buf, 4096 * sizeof(float)));
    CHECK(hipMemset(buf, 0, 4096 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int REPS = 200;
    printf("# chain of N dependent kernels (256 threads x blocks, `work` dependent FMAs each); per CHAIN, median of %d: GPU time between\n"
           "# events [us] / host time spent launching [us]\n", REPS);
    printf("%5s %7s %6s | %22s | %22s | %30s\n", "N", "blocks", "work", "stream launches", "graph", "graph + SetParams per node");
    for (int N : {8, 11, 16, 32}) for (int blocks : {1, 256, 2048}) for (int work : {0, 2000}) {
        auto launch_chain = [&](hipStream_t st) {
            for (int k = 0; k < N; ++k) link<<<blocks, 256, 0, st>>>(buf + 16 * k, buf + 16 * (k + 1), work);
        };
        // ---- capture
        hipGraph_t graph; hipGraphExec_t exec;
        CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        launch_chain(s);
        CHECK(hipStreamEndCapture(s, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        size_t nn = 0;
        CHECK(hipGraphGetNodes(graph, nullptr, &nn));
        std::vector<hipGraphNode_t> nodes(nn);
        CHECK(hipGraphGetNodes(graph, nodes.data(), &nn));
        std::vector<hipKernelNodeParams> params(nn);
        for (size_t i = 0; i < nn; ++i) CHECK(hipGraphKernelNodeGetParams(nodes[i], &params[i]));
        double res[3][2];
        for (int mode = 0; mode < 3; ++mode) {
            std::vector<double> gpu, host;
            for (int r = 0; r < REPS + 20; ++r) {
                CHECK(hipEventRecord(e0, s));
                const double h0 = now_us();
                if (mode == 0) launch_chain(s);
                else {
                    if (mode == 2) for (size_t i = 0; i < nn; ++i) CHECK(hipGraphExecKernelNodeSetParams(exec, nodes[i], &params[i]));
                    CHECK(hipGraphLaunch(exec, s));
                }
                const double h1 = now_us();
                CHECK(hipEventRecord(e1, s));
                CHECK(hipStreamSynchronize(s));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 20) { gpu.push_back(ms * 1e3); host.push_back(h1 - h0); }
            }
            std::sort(gpu.begin(), gpu.end()); std::sort(host.begin(), host.end());
            res[mode][0] = gpu[gpu.size() / 2]; res[mode][1] = host[host.size() / 2];
        }
        printf("%5d %7d %6d | %10.1f / %8.1f | %10.1f / %8.1f | %14.1f / %12.1f\n", N, blocks, work, res[0][0], res[0][1], res[1][0], res[1][1],
               res[2][0], res[2][1]);
        CHECK(hipGraphExecDestroy(exec)); CHECK(hipGraphDestroy(graph));
    }
    return 0;
}
