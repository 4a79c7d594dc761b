// valu_calib.hip - how many cycles does one wave64 instruction occupy its SIMD on gfx950?
// (DESIGN.md assumed 4 cycles per wave64 VALU op, the MI355X guide says 2.)
//
//   hipcc --offload-arch=gfx950 -O3 tools/valu_calib.hip -o tools/valu_calib && tools/valu_calib
//
// Every wave runs K back-to-back instructions of one kind on 8 independent register chains (no RAW stall), with
// W waves per SIMD resident on every SIMD of the chip.  cycles per instruction per SIMD =
//   elapsed(s) * clock(Hz) / (instructions issued per SIMD).  The clock is taken from s_memtime/wall_clock ratios:
// the kernel reports its own s_memtime span (shader cycles, MI355X_MICROARCH.md "s_memtime tick = shader cycle").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIter = 4096;

template <int KIND>
__global__ __launch_bounds__(256) void k_chain(uint32_t* out, unsigned long long* cyc, uint32_t seed) {
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i;
    uint32_t b = seed ^ 0x9E3779B9u, c = seed * 7u + 1u;
    __shared__ uint64_t tab[1024];
    if (KIND >= 7) { for (int i = threadIdx.x; i < 1024; i += 256) tab[i] = i * 0x9E3779B97F4A7C15ull; __syncthreads(); }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIter; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (KIND == 1) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80" : "+v"(a[i]) : "v"(b), "v"(c));
            if (KIND == 2) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (KIND == 3) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (KIND == 4) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (KIND == 5) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (KIND == 6) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (KIND == 7) {                       // ds_read_b64, conflict-free-ish addresses, result folded into the chain
                uint64_t v;
                const uint32_t addr = ((a[i] >> 3) & 1023u) * 8u;
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
                a[i] ^= (uint32_t)v;
            }
        }
        if (KIND == 8) {                           // 8 independent ds_read_b64 in flight per wait
            uint64_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { const uint32_t addr = ((a[i] + it) & 1023u) * 8u; asm volatile("ds_read_b64 %0, %1" : "=v"(v[i]) : "v"(addr)); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] ^= (uint32_t)v[i];
        }
        if (KIND == 9) {                           // v_permlane32_swap pairs
#pragma unroll
            for (int i = 0; i < 8; i += 2) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[i + 1]));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_iter, int blocks_per_cu, int cus, uint32_t* out, unsigned long long* cyc) {
    const int grid = cus * blocks_per_cu;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_chain<KIND>, dim3(grid), dim3(256), 0, 0, out, cyc, 12345u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_chain<KIND>, dim3(grid), dim3(256), 0, 0, out, cyc, 12345u);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid);
    CHK(hipMemcpy(h.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= grid;
    // one 256-thread block = 4 waves = one wave per SIMD; blocks_per_cu waves per SIMD
    const double inst_per_simd = (double)kIter * per_iter * blocks_per_cu;
    // s_memtime counts at a fixed 100 MHz on this part (constant "wall clock"); report both views
    printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"kernel_ms\": %.4f, \"inst_per_simd\": %.0f, "
           "\"ns_per_inst_per_simd\": %.4f, \"cycles_at_2400MHz\": %.3f, \"readcyclecounter_span\": %.0f}\n",
           name, blocks_per_cu, ms, inst_per_simd, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4, mean);
}

int main() {
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    uint32_t* out; unsigned long long* cyc;
    CHK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    CHK(hipMalloc(&cyc, (size_t)cus * 8 * 8));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", p.gcnArchName, cus, p.clockRate);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_and_b32", 8, w, cus, out, cyc);
        run<1>("v_bitop3_b32", 8, w, cus, out, cyc);
    }
    run<2>("v_and_or_b32", 8, 4, cus, out, cyc);
    run<3>("v_bcnt_u32_b32", 8, 4, cus, out, cyc);
    run<4>("v_mov_b32_dpp", 8, 4, cus, out, cyc);
    run<5>("v_perm_b32", 8, 4, cus, out, cyc);
    run<6>("v_alignbit_b32", 8, 4, cus, out, cyc);
    run<7>("ds_read_b64+wait", 8, 4, cus, out, cyc);
    run<8>("ds_read_b64 x8 per wait", 8, 4, cus, out, cyc);
    run<8>("ds_read_b64 x8 per wait", 8, 8, cus, out, cyc);
    run<9>("v_permlane32_swap", 4, 4, cus, out, cyc);
    return 0;
}
