// ubench_lds.hip -- experiment: LDS read cost by access width / pattern on gfx950 (cycles per wave-instruction per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int iters, uint32_t seed) {
    __shared__ uint32_t s[16384];  // 64 KB
    for (int i = threadIdx.x; i < 16384; i += 1024) s[i] = i * 2654435761u;
    __syncthreads();
    const uint16_t *s16 = (const uint16_t *)s;
    uint32_t acc = 0;
    uint32_t idx = (MODE & 1) ? ((threadIdx.x * 2654435761u + seed) >> 8) : threadIdx.x;  // odd modes: scattered
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const uint32_t i = (idx + u * 97 + it * 13) & 16383;
            if (MODE < 2) acc += s16[i * 2 + (u & 1)];                 // ds_read_u16
            else if (MODE < 4) acc += s[i];                             // ds_read_b32
            else if (MODE < 6) { acc += s[i & ~1u] ^ s[(i & ~1u) + 1]; }  // two adjacent dwords (read2_b32 / b64)
            else { const uint4 v = ((const uint4 *)s)[i >> 2]; acc += v.x ^ v.y ^ v.z ^ v.w; }  // b128
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

int main() {
    uint32_t *d; CK(hipMalloc(&d, 256 * 1024 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"u16 consecutive", "u16 scattered", "b32 consecutive", "b32 scattered", "2xb32 consecutive", "2xb32 scattered", "b128 consecutive", "b128 scattered"};
    const int iters = 2000;
    for (int m = 0; m < 8; ++m) {
        float best = 1e9;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            switch (m) {
                case 0: k<0><<<256, 1024>>>(d, iters, 1); break; case 1: k<1><<<256, 1024>>>(d, iters, 1); break;
                case 2: k<2><<<256, 1024>>>(d, iters, 1); break; case 3: k<3><<<256, 1024>>>(d, iters, 1); break;
                case 4: k<4><<<256, 1024>>>(d, iters, 1); break; case 5: k<5><<<256, 1024>>>(d, iters, 1); break;
                case 6: k<6><<<256, 1024>>>(d, iters, 1); break; case 7: k<7><<<256, 1024>>>(d, iters, 1); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        // wave-level read statements per CU = 16 waves * iters * 16
        const double stm = 16.0 * iters * 16;
        printf("%-20s %7.3f ms  %.1f ns per wave-statement per CU (~%.1f cycles @2.4GHz)\n", names[m], best, best * 1e6 / stm, best * 1e6 / stm * 2.4);
    }
    return 0;
}
