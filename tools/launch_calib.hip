// launch_calib.hip - what does a short kernel pay for (a) a large kernel-argument block, (b) cold instruction fetch?
//   hipcc --offload-arch=gfx950 -O3 tools/launch_calib.hip -o tools/launch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int N> struct Args { unsigned v[N]; unsigned* out; };

// reads `touch` words spread over the whole argument block, then one store
template <int N>
__global__ __launch_bounds__(512) void k_args(Args<N> a, int touch) {
    unsigned s = 0;
    for (int i = 0; i < touch; ++i) s += a.v[(i * (N / (touch > 0 ? touch : 1))) % N];
    if (threadIdx.x == 0 && s == 0x12345678u) a.out[blockIdx.x] = s;
}

// long straight-line code: K dependent-free blocks of distinct instructions, executed once (cold fetch)
template <int K>
__global__ __launch_bounds__(512) void k_code(unsigned* out, unsigned seed) {
    unsigned x = seed + threadIdx.x;
#pragma unroll
    for (int i = 0; i < K; ++i) { x = x * 1664525u + (1013904223u + i); x ^= x >> ((i % 13) + 3); }
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
    const int grid = 768, reps = 400;
    { Args<8> a{}; a.out = out; printf("{\"kernel\": \"args 32 B, touch 1\", \"us_per_launch\": %.2f}\n", time_us([&] { hipLaunchKernelGGL(k_args<8>, dim3(grid), dim3(512), 0, 0, a, 1); }, reps)); }
    { Args<256> a{}; a.out = out; printf("{\"kernel\": \"args 1 KB, touch 1\", \"us_per_launch\": %.2f}\n", time_us([&] { hipLaunchKernelGGL(k_args<256>, dim3(grid), dim3(512), 0, 0, a, 1); }, reps)); }
    { Args<256> a{}; a.out = out; printf("{\"kernel\": \"args 1 KB, touch 16\", \"us_per_launch\": %.2f}\n", time_us([&] { hipLaunchKernelGGL(k_args<256>, dim3(grid), dim3(512), 0, 0, a, 16); }, reps)); }
    { Args<768> a{}; a.out = out; printf("{\"kernel\": \"args 3 KB, touch 1\", \"us_per_launch\": %.2f}\n", time_us([&] { hipLaunchKernelGGL(k_args<768>, dim3(grid), dim3(512), 0, 0, a, 1); }, reps)); }
    { Args<768> a{}; a.out = out; printf("{\"kernel\": \"args 3 KB, touch 48\", \"us_per_launch\": %.2f}\n", time_us([&] { hipLaunchKernelGGL(k_args<768>, dim3(grid), dim3(512), 0, 0, a, 48); }, reps)); }
    printf("{\"kernel\": \"code 64 steps\", \"us_per_launch\": %.2f}\n", time_us([&] { hipLaunchKernelGGL(k_code<64>, dim3(grid), dim3(512), 0, 0, out, 1u); }, reps));
    printf("{\"kernel\": \"code 1024 steps (~20 KB)\", \"us_per_launch\": %.2f}\n", time_us([&] { hipLaunchKernelGGL(k_code<1024>, dim3(grid), dim3(512), 0, 0, out, 1u); }, reps));
    printf("{\"kernel\": \"code 4096 steps (~80 KB)\", \"us_per_launch\": %.2f}\n", time_us([&] { hipLaunchKernelGGL(k_code<4096>, dim3(grid), dim3(512), 0, 0, out, 1u); }, reps));
    for (int g : {256, 768, 2048}) printf("{\"kernel\": \"args 32 B, grid %d\", \"us_per_launch\": %.2f}\n", g, time_us([&] { Args<8> a{}; a.out = out; hipLaunchKernelGGL(k_args<8>, dim3(g), dim3(512), 0, 0, a, 1); }, reps));
    return 0;
}
