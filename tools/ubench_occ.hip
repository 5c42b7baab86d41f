// Absolute VALU issue rate per SIMD as a function of resident waves per SIMD (independent v_add_u32 /
// v_dot2c streams), timed with HIP events over a grid that fills every CU exactly once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP, int ILP>
__global__ void k(int *out, int iters, uint32_t seed) {
    uint32_t a[8], b = seed * 3 + threadIdx.x, c = seed ^ 0x12345;
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 7 + i + seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint32_t &d = a[i % ILP];
                if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(d) : "v"(b));
                if (OP == 1) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(d) : "v"(b), "v"(c));
            }
    }
    int s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 1023] = s;
}
template <int OP, int ILP>
void run(const char *name, int *out, int ncu) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int threads = 256 * (waves_per_simd > 4 ? 4 : waves_per_simd); const int blocks_per_cu = waves_per_simd > 4 ? waves_per_simd / 4 : 1;
        hipLaunchKernelGGL((k<OP, ILP>), dim3(ncu * blocks_per_cu), dim3(threads), 0, 0, out, 10, 1u);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k<OP, ILP>), dim3(ncu * blocks_per_cu), dim3(threads), 0, 0, out, iters, 1u);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = (double)iters * 32 * waves_per_simd;
        printf("%-10s ILP %d waves/SIMD %d: %.3f ms, %.2f ns per SIMD wave-instr (%.2f cycles @2.4 GHz)\n", name, ILP, waves_per_simd, ms,
               ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
}
int main() {
    int *out; (void)hipMalloc(&out, 1024 * 4);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    printf("CUs %d clock %d kHz\n", p.multiProcessorCount, p.clockRate);
    run<0, 8>("v_add_u32", out, p.multiProcessorCount);
    run<0, 1>("v_add_u32", out, p.multiProcessorCount);
    run<0, 2>("v_add_u32", out, p.multiProcessorCount);
    run<1, 8>("v_dot2c", out, p.multiProcessorCount);
    run<1, 1>("v_dot2c", out, p.multiProcessorCount);
    return 0;
}
