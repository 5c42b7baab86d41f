// VALU issue rate of the integer instructions the scan kernel is made of, at 1 and 4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short short2v __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int iters) {
    uint32_t a[8];
    for (int j = 0; j < 8; ++j) a[j] = threadIdx.x * 2654435761u + j;
    const uint32_t c = blockIdx.x + 3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
            if (OP == 1) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
            if (OP == 2) asm volatile("v_alignbyte_b32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
            if (OP == 3) asm volatile("v_dot2_i32_i16 %0, %1, %1, %0" : "+v"(a[j]) : "v"(c));
            if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
            if (OP == 5) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[j]) : "v"(c));
            if (OP == 6) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
            if (OP == 7) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a[j]));
            if (OP == 8) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[j]) : "v"(c));
            if (OP == 9) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[j]) : "v"(c));
            if (OP == 10) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(c) : "vcc");
        }
    }
    uint32_t r = 0;
    for (int j = 0; j < 8; ++j) r ^= a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP>
void run(const char *name, uint32_t *o, int per) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int threads : {256, 1024}) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            k<OP><<<256, threads>>>(o, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double inst_per_simd = (double)iters * 8 * per * (threads / 256);
        printf("%-16s %d wave(s)/SIMD: %.3f ms -> %.2f ns per wave-instruction per SIMD\n", name, threads / 256, ms, ms * 1e6 / inst_per_simd);
    }
}
int main() {
    uint32_t *o; (void)hipMalloc(&o, 256 * 1024 * 4);
    run<0>("v_add_u32", o, 1); run<1>("v_and_b32", o, 1); run<2>("v_alignbyte_b32", o, 1); run<3>("v_dot2_i32_i16", o, 1);
    run<4>("v_fma_f32", o, 1); run<5>("v_lshl_add_u32", o, 1); run<6>("v_and_or_b32", o, 1); run<7>("v_ashrrev_i32", o, 1);
    run<8>("v_pk_add_u16", o, 1); run<9>("v_mad_u32_u24", o, 1); run<10>("v_cmp+v_cndmask", o, 2);
    return 0;
}
