#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int sdot2(uint32_t a, uint32_t b, int c) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, a), __builtin_bit_cast(v2s, b), c, false);
}
__global__ void k(const uint32_t *in, int *out, int *out2, int iters) {
    __shared__ __attribute__((aligned(16))) uint16_t m[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) m[i] = (uint16_t)(in[i] ^ 0x8000u);
    __syncthreads();
    int acc = 0;
    int e = (threadIdx.x * 7 + 1) & 4095;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        uint32_t addr = (uint32_t)(uintptr_t)(&m[(e + it * 13) & 4095]);
        uint64_t v;
        asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
        acc += sdot2((uint32_t)v, 0x0003fffbu /* (-5, 3) */, sdot2((uint32_t)(v >> 32), 0x00000002u, 0));
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) out2[blockIdx.x] = (int)(t1 - t0);
}
int main() {
    std::vector<uint32_t> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = (i * 2654435761u >> 16) & 0xffff;
    uint32_t *d; int *o, *o2;
    hipMalloc(&d, 8192 * 4); hipMalloc(&o, 256 * 4); hipMalloc(&o2, 4);
    hipMemcpy(d, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    const int iters = 1000;
    k<<<1, 256>>>(d, o, o2, iters);
    std::vector<int> r(256); int cyc;
    hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost); hipMemcpy(&cyc, o2, 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        int acc = 0; int e = (t * 7 + 1) & 4095;
        for (int it = 0; it < iters; ++it) {
            int b = (e + it * 13) & 4095;
            auto s = [&](int i) { return (int)(int16_t)(uint16_t)(h[i] ^ 0x8000u); };
            acc += -5 * s(b) + 3 * s(b + 1) + 2 * s(b + 2);
        }
        if (acc != r[t]) { if (bad < 5) printf("mismatch t=%d got %d want %d\n", t, r[t], acc); ++bad; }
    }
    printf("unaligned ds_read_b64 + sdot2: %s, %d clock64 ticks for %d iters\n", bad ? "WRONG" : "ok", cyc, iters);
    return bad != 0;
}
