// icache_calib.hip - cost of executing code for the first time in a launch (instruction fetch) on gfx950:
// the same 4096 VALU instructions per wave as straight-line code (16 KB, every line fetched once) and as a 64-trip loop
// over 64 instructions (256 B, hot after the first trip).  One wave per SIMD (grid 256 x 256 threads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define EIGHT "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"

template <int REPT, int TRIPS>
__global__ __launch_bounds__(256) void k(unsigned* out, unsigned seed) {
    unsigned a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = threadIdx.x;
    for (int t = 0; t < TRIPS; ++t) {
        if constexpr (REPT == 512) asm volatile(".rept 512\n" EIGHT ".endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if constexpr (REPT == 128) asm volatile(".rept 128\n" EIGHT ".endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        if constexpr (REPT == 8) asm volatile(".rept 8\n" EIGHT ".endr" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    }
    const unsigned x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (x == 0x12345678u) out[blockIdx.x] = x;
}

template <class F> float time_us(F launch, int reps) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) launch();
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main() {
    unsigned* out; CHK(hipMalloc(&out, 1 << 20));
    const int reps = 300;
    for (int grid : {256, 768}) {
        printf("{\"grid\": %d, \"straight_4096_instr_16KB_us\": %.2f, ", grid, time_us([&] { hipLaunchKernelGGL((k<512, 1>), dim3(grid), dim3(256), 0, 0, out, 1u); }, reps));
        printf("\"loop_64x64_instr_us\": %.2f, ", time_us([&] { hipLaunchKernelGGL((k<8, 64>), dim3(grid), dim3(256), 0, 0, out, 1u); }, reps));
        printf("\"straight_1024_instr_4KB_us\": %.2f, ", time_us([&] { hipLaunchKernelGGL((k<128, 1>), dim3(grid), dim3(256), 0, 0, out, 1u); }, reps));
        printf("\"loop_16x64_instr_us\": %.2f, ", time_us([&] { hipLaunchKernelGGL((k<8, 16>), dim3(grid), dim3(256), 0, 0, out, 1u); }, reps));
        printf("\"loop_1x64_instr_us\": %.2f}\n", time_us([&] { hipLaunchKernelGGL((k<8, 1>), dim3(grid), dim3(256), 0, 0, out, 1u); }, reps));
    }
    return 0;
}
