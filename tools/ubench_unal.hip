// throughput of ds_read_b64 at 8-byte aligned vs 2-byte aligned addresses, and ds_read2_b32 (4-byte aligned), 16 waves per CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(1024) void k(int *out, int iters, int stride) {
    __shared__ __attribute__((aligned(16))) uint16_t m[32768];
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) m[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t base = (uint32_t)(uintptr_t)m + wave * 4096 + lane * stride * 2;  // bytes
    if (MODE == 1) base += 2;  // 2-byte aligned only
    uint64_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint64_t v[8];
        const uint32_t a = base + (it & 7) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 2) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v[j]) : "v"(a), "n"(j * 16), "n"(j * 16 + 1));
            else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[j]) : "v"(a), "n"(j * 64));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    out[blockIdx.x * 1024 + threadIdx.x] = (int)acc;
}
int main() {
    int *o; (void)hipMalloc(&o, 256 * 1024 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int stride : {12, 4, 1}) for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (mode == 0) k<0><<<256, 1024>>>(o, iters, stride);
            if (mode == 1) k<1><<<256, 1024>>>(o, iters, stride);
            if (mode == 2) k<2><<<256, 1024>>>(o, iters, stride);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("lane stride %2d samples, %s: %.3f ms -> %.2f ns per wave-instruction per CU\n", stride,
               mode == 0 ? "ds_read_b64 aligned  " : mode == 1 ? "ds_read_b64 +2 bytes " : "ds_read2_b32         ", ms, ms * 1e6 / (iters * 8.0 * 16));
    }
    return 0;
}
