// Saturated issue cost of single VALU instructions on one SIMD: 8 resident wavefronts per SIMD, 8 independent
// streams per wavefront, timed with HIP events over a grid that fills every CU.  cycles = ns * clock from GRBM-free estimate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP>
__global__ void k(int *out, int iters, uint32_t seed) {
    uint32_t a[8], b = seed * 3 + threadIdx.x, c = seed ^ 0x12345;
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 7 + i + seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint32_t &d = a[i];
                if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(d) : "v"(b));
                if (OP == 1) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 2) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 3) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 4) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 5) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 6) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(d) : "v"(b));
                if (OP == 7) asm volatile("v_pk_sub_i16 %0, %0, %1 clamp" : "+v"(d) : "v"(b));
                if (OP == 8) asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 9) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 10) asm volatile("v_and_b32 %0, %0, %1" : "+v"(d) : "v"(b));
                if (OP == 11) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(d));
                if (OP == 12) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 13) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 14) asm volatile("v_lshl_or_b32 %0, %0, 5, %1" : "+v"(d) : "v"(b));
                if (OP == 15) asm volatile("v_pk_min_i16 %0, %0, %1" : "+v"(d) : "v"(b));
                if (OP == 16) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(d) : "v"(b));
                if (OP == 17) asm volatile("v_mov_b32_dpp %0, %1 wave_rol:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(b));
                if (OP == 18) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(d) : "v"(b));
                if (OP == 19) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(d), "v"(b) : "vcc");
                if (OP == 20) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 21) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(d) : "v"(b));
                if (OP == 22) asm volatile("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 23) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(d) : "v"(b));
                if (OP == 24) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(d));
                if (OP == 25) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_0" : "+v"(d) : "v"(b));
                if (OP == 26) asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(b));
                if (OP == 27) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(d) : "v"(b));
                if (OP == 28) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 29) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(d) : "v"(b));
                if (OP == 30) asm volatile("v_alignbyte_b32 %0, %0, %1, 2" : "+v"(d) : "v"(b));
                if (OP == 31) asm volatile("v_max_i16 %0, %0, %1" : "+v"(d) : "v"(b));
                if (OP == 32) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(d) : "v"(b));
                if (OP == 33) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 34) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(d) : "v"(b));
                if (OP == 35) asm volatile("v_not_b32 %0, %1" : "=v"(d) : "v"(b));
                if (OP == 36) asm volatile("v_pk_lshrrev_b16 %0, 1, %0" : "+v"(d));
                if (OP == 37) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(d) : "v"(b));
                if (OP == 38) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(uint64_t *)&a[i & 6]) : "v"(b), "v"(c) : "vcc");
                if (OP == 39) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(d) : "v"(b), "v"(c));
                if (OP == 40) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(d) : "v"(b), "v"(c));
            }
    }
    int s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 1023] = s;
}
template <int OP>
void run(const char *name, int *out, int ncu) {
    const int iters = 10000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP>), dim3(ncu * 2), dim3(1024), 0, 0, out, 10, 1u);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<OP>), dim3(ncu * 2), dim3(1024), 0, 0, out, iters, 1u);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 32 * 8;
    printf("%-22s %.3f ns per wave-instr and SIMD\n", name, ms * 1e6 / instr_per_simd);
}
#define R(n, s) run<n>(s, out, p.multiProcessorCount);
int main() {
    int *out; (void)hipMalloc(&out, 1024 * 4);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    R(0, "v_add_u32") R(34, "v_sub_u32") R(10, "v_and_b32") R(16, "v_xor_b32") R(35, "v_not_b32") R(26, "v_mov_b32") R(11, "v_lshrrev_b32") R(24, "v_bfe_u32")
    R(6, "v_alignbit_b32") R(30, "v_alignbyte_b32") R(9, "v_perm_b32") R(12, "v_and_or_b32") R(40, "v_or3_b32") R(13, "v_bfi_b32") R(14, "v_lshl_or_b32") R(21, "v_lshl_add_u32")
    R(20, "v_add3_u32") R(39, "v_xad_u32") R(25, "v_add_u32_sdwa") R(17, "v_mov_b32_dpp") R(18, "v_cndmask_b32") R(19, "v_cmp_lt_u32") R(29, "v_mbcnt_lo") R(33, "v_max3_i32")
    R(31, "v_max_i16") R(7, "v_pk_sub_i16 clamp") R(27, "v_pk_add_u16") R(15, "v_pk_min_i16") R(32, "v_pk_max_i16") R(36, "v_pk_lshrrev_b16") R(37, "v_pk_mul_lo_u16") R(8, "v_pk_mad_i16")
    R(1, "v_dot2c_i32_i16") R(2, "v_dot2_i32_i16") R(3, "v_dot4_u32_u8") R(4, "v_mad_i32_i24") R(5, "v_mad_u32_u24") R(22, "v_mad_i32_i16") R(23, "v_mul_i32_i24") R(28, "v_sad_u16") R(38, "v_mad_u64_u32")
    return 0;
}
