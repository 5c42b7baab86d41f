// Issue rate of the VALU instructions the scan kernel leans on: cycles per wave-instruction and SIMD,
// for independent streams and for one dependent chain, at 1 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short v2s __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int OP, bool DEP>
__global__ void k(int *out, long long *cyc, int iters, uint32_t seed) {
    uint32_t a[8], b = seed * 3 + threadIdx.x, c = seed ^ 0x12345;
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 7 + i + seed;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint32_t &d = DEP ? a[0] : a[i];
                if (OP == 0) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 1) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 2) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 3) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(d) : "v"(b));
                if (OP == 4) asm volatile("v_pk_sub_i16 %0, %0, %1 clamp" : "+v"(d) : "v"(b));
                if (OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(d) : "v"(b));
                if (OP == 6) asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 7) asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 8) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 9) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(d) : "v"(b));
                if (OP == 10) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 11) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(d));
            }
        }
    }
    const long long t1 = clock64();
    int s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP>
void run(const char *name, int *out, long long *cyc) {
    const int iters = 2000;
    for (int dep = 0; dep < 2; ++dep)
        for (int threads : {256, 1024}) {
            if (dep) hipLaunchKernelGGL((k<OP, true>), dim3(1), dim3(threads), 0, 0, out, cyc, iters, 1u);
            else hipLaunchKernelGGL((k<OP, false>), dim3(1), dim3(threads), 0, 0, out, cyc, iters, 1u);
            hipDeviceSynchronize();
            long long c;
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            // clock64 = s_memtime counts at 100 MHz on gfx9; report ticks per instruction of one wave and scaled by waves/SIMD
            printf("%-16s %s waves/SIMD %d : %.3f ticks per wave-instr, %.3f per SIMD-instr\n", name, dep ? "dependent  " : "independent", threads / 256,
                   (double)c / (iters * 32.0), (double)c / (iters * 32.0) / (threads / 256));
        }
}
int main() {
    int *out; long long *cyc;
    hipMalloc(&out, 1024 * 4); hipMalloc(&cyc, 8);
    run<5>("v_add_u32", out, cyc);
    run<0>("v_dot2c_i32_i16", out, cyc);
    run<1>("v_dot2_i32_i16", out, cyc);
    run<2>("v_mad_i32_i24", out, cyc);
    run<8>("v_mad_u32_u24", out, cyc);
    run<3>("v_alignbit_b32", out, cyc);
    run<4>("v_pk_sub_i16", out, cyc);
    run<6>("v_pk_mad_i16", out, cyc);
    run<7>("v_dot4c_i32_i8", out, cyc);
    run<9>("v_mul_lo_u32", out, cyc);
    run<10>("v_perm_b32", out, cyc);
    run<11>("v_bfe_u32", out, cyc);
    return 0;
}
