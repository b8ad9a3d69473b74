// r5_calib.hip — two hardware calibrations asked for by the round-4 review, on gfx950:
//
//  (b) "exec": does a wave64 VALU instruction whose EXEC mask has an all-zero 32-lane half issue in ONE pass of the SIMD-32
//      instead of two?  If yes, lane efficiency at 32-lane granularity would come for free in the blend kernels (their hit
//      masks are ballots already).  Streams of 64 independent v_fma_f32 between an EXEC write and its restore, every SIMD of
//      the chip running w = 4 / 8 waves; cycles per instruction per SIMD for EXEC = all lanes, the low half, the high half,
//      every other lane, the low 16 lanes, one lane.
//  (c) "atomics": the blend backward adds nine per-Gaussian sums with ONE global_atomic_add_f64 instruction from nine lanes
//      into an 80-byte record; PMC shows ~64 B of fabric write per atomic INSTRUCTION and 1.98x the algorithmic traffic.
//      Same access pattern (one wave per block, random record ids over 1 M records, 200 atomics per wave, 8160 waves), record
//      layouts: 80-B records (as shipped), 128-B padded records (the nine doubles inside one 128-B line), 64-B aligned 8-double
//      records + the ninth sum in a second array, float32 sums in 48-B records and in 64-B padded records.  Wall time per
//      atomic instruction here; HBM bytes per kernel from `rocprofv3 --pmc WRITE_SIZE` / `FETCH_SIZE` runs of this binary
//      (kernel names carry the layout).
// Build: hipcc --offload-arch=gfx950 -O3 tools/r5_calib.hip -o tools/r5_calib ; run: tools/r5_calib [exec|atomics|all]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// ---------------------------------------------------------------------------------------------------------- (b) EXEC halves
constexpr int ITER = 2000;
#define FMA8 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
__global__ __launch_bounds__(256) void k_exec(float* out, unsigned long long* ticks, float a, float b, unsigned long long mask) {
    float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f, v4 = v0 + 4.f, v5 = v0 + 5.f, v6 = v0 + 6.f,
          v7 = v0 + 7.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
        asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, %10\n"
                     FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA8 FMA8
                     "s_mov_b64 exec, s[20:21]\n"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)
                     : "v"(a), "v"(b), "s"(mask)
                     : "s20", "s21", "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

static void run_exec() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clock_ghz = prop.clockRate * 1e-6;
    printf("# (b) EXEC halves: device %s, %d CUs, clockRate %.3f GHz; 64 v_fma_f32 per EXEC write, ITER=%d\n", prop.gcnArchName, cus,
           clock_ghz, ITER);
    const int max_waves = cus * 4 * 8;
    float* out; unsigned long long* ticks;
    CHECK(hipMalloc(&out, sizeof(float) * 64 * (size_t)max_waves));
    CHECK(hipMalloc(&ticks, 8 * (size_t)max_waves));
    std::vector<unsigned long long> h(max_waves);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct M { const char* name; unsigned long long mask; };
    const M masks[] = {{"all 64 lanes", ~0ull}, {"low 32 lanes", 0x00000000FFFFFFFFull}, {"high 32 lanes", 0xFFFFFFFF00000000ull},
                       {"every other lane", 0x5555555555555555ull}, {"low 16 lanes", 0xFFFFull}, {"lanes 0-15 and 32-47", 0x0000FFFF0000FFFFull},
                       {"one lane", 1ull}};
    printf("%-24s %6s %9s %9s\n", "EXEC", "w/SIMD", "cyc_tick", "cyc_wall");
    for (const M& m : masks)
        for (int w : {1, 4, 8}) {
            const int blocks = cus * w;
            const double instr = (double)ITER * 64;
            k_exec<<<blocks, 256>>>(out, ticks, 1.0000001f, 1e-9f, m.mask);
            CHECK(hipDeviceSynchronize());
            double best = 1e30; unsigned long long worst = 0;
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipEventRecord(e0));
                k_exec<<<blocks, 256>>>(out, ticks, 1.0000001f, 1e-9f, m.mask);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) {
                    best = ms;
                    CHECK(hipMemcpy(h.data(), ticks, 8 * (size_t)blocks * 4, hipMemcpyDeviceToHost));
                    worst = *std::max_element(h.begin(), h.begin() + blocks * 4);
                }
            }
            printf("%-24s %6d %9.2f %9.2f\n", m.name, w, (double)worst / (w * instr), best * 1e-3 * clock_ghz * 1e9 / (w * instr));
            fflush(stdout);
        }
    CHECK(hipFree(out)); CHECK(hipFree(ticks));
}

// ------------------------------------------------------------------------------------------------------------- (c) atomics
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
constexpr int ATOMS_PER_WAVE = 200;
// the nine lanes of the shipped kernel's atomic (blend.hip: lanes 0, 1, 4, 5, 8, 9, 12, 13 and 2), component = order below
__device__ __forceinline__ int component_of_lane(int lane) {
    switch (lane) { case 0: return 0; case 1: return 1; case 4: return 2; case 5: return 3; case 8: return 4; case 9: return 5;
                    case 12: return 6; case 13: return 7; case 2: return 8; default: return -1; }
}
template <int REC_BYTES>
__global__ __launch_bounds__(64) void atomics_f64_records(char* rec, uint32_t n_rec, float v) {
    const int c = component_of_lane(threadIdx.x);
    for (int it = 0; it < ATOMS_PER_WAVE; ++it) {
        const uint32_t id = hash32(blockIdx.x * 7919u + it * 104729u) % n_rec;       // wave-uniform
        if (c >= 0) atomicAdd(reinterpret_cast<double*>(rec + (size_t)id * REC_BYTES) + c, (double)(v * (c + 1)));
    }
}
// eight doubles in a 64-byte aligned record + the ninth in an array of its own
__global__ __launch_bounds__(64) void atomics_f64_split_8_plus_1(char* rec, double* ninth, uint32_t n_rec, float v) {
    const int c = component_of_lane(threadIdx.x);
    for (int it = 0; it < ATOMS_PER_WAVE; ++it) {
        const uint32_t id = hash32(blockIdx.x * 7919u + it * 104729u) % n_rec;
        if (c >= 0 && c < 8) atomicAdd(reinterpret_cast<double*>(rec + (size_t)id * 64) + c, (double)(v * (c + 1)));
        else if (c == 8) atomicAdd(ninth + id, (double)(v * 9));
    }
}
template <int REC_BYTES>
__global__ __launch_bounds__(64) void atomics_f32_records(char* rec, uint32_t n_rec, float v) {
    const int c = component_of_lane(threadIdx.x);
    for (int it = 0; it < ATOMS_PER_WAVE; ++it) {
        const uint32_t id = hash32(blockIdx.x * 7919u + it * 104729u) % n_rec;
        if (c >= 0) atomicAdd(reinterpret_cast<float*>(rec + (size_t)id * REC_BYTES) + c, v * (c + 1));
    }
}
// plain (non-atomic) stores of the same nine values, 80-B records: what the pattern costs without the atomic unit
__global__ __launch_bounds__(64) void stores_f64_records_80(char* rec, uint32_t n_rec, float v) {
    const int c = component_of_lane(threadIdx.x);
    for (int it = 0; it < ATOMS_PER_WAVE; ++it) {
        const uint32_t id = hash32(blockIdx.x * 7919u + it * 104729u) % n_rec;
        if (c >= 0) reinterpret_cast<double*>(rec + (size_t)id * 80)[c] = (double)(v * (c + 1));
    }
}

static void run_atomics() {
    const uint32_t n_rec = 1000000;
    const int waves = 8160;
    char* rec; double* ninth;
    CHECK(hipMalloc(&rec, (size_t)n_rec * 128));
    CHECK(hipMalloc(&ninth, (size_t)n_rec * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("# (c) atomics: %d waves x %d atomic instructions (nine lanes each), random ids over %u records\n", waves, ATOMS_PER_WAVE, n_rec);
    printf("%-44s %10s %12s\n", "layout", "us/launch", "ns/atomic-instr (chip)");
    auto time_it = [&](const char* name, auto launch) {
        CHECK(hipMemset(rec, 0, (size_t)n_rec * 128));
        launch();
        CHECK(hipDeviceSynchronize());
        double best = 1e30;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipEventRecord(e0));
            launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, (double)ms);
        }
        printf("%-44s %10.1f %12.3f\n", name, best * 1e3, best * 1e6 / ((double)waves * ATOMS_PER_WAVE));
        fflush(stdout);
    };
    time_it("f64 x9, 80-B records (shipped)", [&] { atomics_f64_records<80><<<waves, 64>>>(rec, n_rec, 1e-3f); });
    time_it("f64 x9, 128-B padded records", [&] { atomics_f64_records<128><<<waves, 64>>>(rec, n_rec, 1e-3f); });
    time_it("f64 x8 in a 64-B record + ninth elsewhere", [&] { atomics_f64_split_8_plus_1<<<waves, 64>>>(rec, ninth, n_rec, 1e-3f); });
    time_it("f32 x9, 48-B records", [&] { atomics_f32_records<48><<<waves, 64>>>(rec, n_rec, 1e-3f); });
    time_it("f32 x9, 64-B padded records", [&] { atomics_f32_records<64><<<waves, 64>>>(rec, n_rec, 1e-3f); });
    time_it("plain f64 stores x9, 80-B records", [&] { stores_f64_records_80<<<waves, 64>>>(rec, n_rec, 1e-3f); });
    CHECK(hipFree(rec)); CHECK(hipFree(ninth));
}

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "all";
    setvbuf(stdout, nullptr, _IOLBF, 0);
    if (!strcmp(what, "exec") || !strcmp(what, "all")) run_exec();
    if (!strcmp(what, "atomics") || !strcmp(what, "all")) run_atomics();
    return 0;
}
