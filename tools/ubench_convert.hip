// ubench_convert.hip -- experiment: cost of UC8 IQ -> magnitude conversion variants on gfx950.
// Each variant converts N samples and folds the magnitudes into a checksum (no magnitude writeback).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// A: global LUT gather (v0)
__global__ __launch_bounds__(256) void k_global(const uint4 *iq, size_t nvec, const uint16_t *lut, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        uint4 v = iq[i];
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc += lut[w[k] & 0xffff]; acc += lut[w[k] >> 16]; }
    }
    if (acc == 0x12345) out[0] = acc;
    atomicAdd(&out[1], acc);
}

// B: full 128 KB LUT in LDS, 1024 threads, 1 WG/CU, with XOR swizzle option
template <int SWZ>
__global__ __launch_bounds__(1024) void k_lds_full(const uint4 *iq, size_t nvec, const uint16_t *lut, uint32_t *out) {
    extern __shared__ uint16_t s[];
    for (int i = threadIdx.x; i < 65536; i += 1024) {
        int j = SWZ ? (i ^ ((i >> 8) & 0x3e)) : i;
        s[j] = lut[i];
    }
    __syncthreads();
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 1024) {
        uint4 v = iq[i];
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t a = w[k] & 0xffff, b = w[k] >> 16;
            if (SWZ) { a ^= (a >> 8) & 0x3e; b ^= (b >> 8) & 0x3e; }
            acc += s[a]; acc += s[b];
        }
    }
    if (acc == 0x12345) out[0] = acc;
    atomicAdd(&out[1], acc);
}

// C: folded 128x128 LUT (32 KB) in LDS, 512 threads, swizzled
template <int SWZ>
__global__ __launch_bounds__(512) void k_lds_fold(const uint4 *iq, size_t nvec, const uint16_t *lut, uint32_t *out) {
    __shared__ uint16_t s[16384];
    for (int i = threadIdx.x; i < 16384; i += 512) {
        int a = i & 127, b = i >> 7;
        int j = SWZ ? (i ^ ((b & 31) << 1)) : i;
        s[j] = lut[((128 + b) << 8) | (128 + a)];
    }
    __syncthreads();
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 512) {
        uint4 v = iq[i];
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t x = w[k];
            uint32_t t = (~x >> 7) & 0x01010101u;
            uint32_t f = (x ^ ((t << 7) - t)) & 0x7f7f7f7fu;
            uint32_t i0 = (f & 0x7f) | ((f >> 1) & 0x3f80);
            uint32_t i1 = ((f >> 16) & 0x7f) | ((f >> 17) & 0x3f80);
            if (SWZ) { i0 ^= (i0 >> 6) & 0x3e; i1 ^= (i1 >> 6) & 0x3e; }
            acc += s[i0]; acc += s[i1];
        }
    }
    if (acc == 0x12345) out[0] = acc;
    atomicAdd(&out[1], acc);
}

// D: no conversion at all (pure streaming read)
__global__ __launch_bounds__(256) void k_stream(const uint4 *iq, size_t nvec, const uint16_t *lut, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        uint4 v = iq[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345) out[0] = acc;
    atomicAdd(&out[1], acc);
}

int main(int argc, char **argv) {
    size_t n = (size_t)1 << 30;
    std::vector<uint16_t> lut(65536);
    for (int q = 0; q < 256; ++q) for (int i = 0; i < 256; ++i) {
        float fi = (i - 127.5) / 127.5, fq = (q - 127.5) / 127.5; float m = fi * fi + fq * fq; if (m > 1) m = 1;
        lut[(q << 8) | i] = (uint16_t)(sqrtf(m) * 65535.0f + 0.5f);
    }
    uint16_t *dlut; uint32_t *dout; uint8_t *diq;
    CK(hipMalloc(&dlut, 131072)); CK(hipMemcpy(dlut, lut.data(), 131072, hipMemcpyHostToDevice));
    CK(hipMalloc(&dout, 64)); CK(hipMalloc(&diq, 2 * n));
    std::vector<uint8_t> h(1 << 26);
    CK(hipFuncSetAttribute((const void *)k_lds_full<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    CK(hipFuncSetAttribute((const void *)k_lds_full<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        // 0: gaussian-ish noise sigma 3 around 127.5, 1: uniform random, 2: noise + bursts
        uint32_t st = 12345;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
        for (size_t i = 0; i < h.size(); ++i) {
            if (mode == 1) h[i] = rnd() & 255;
            else {
                int s = 0; for (int k = 0; k < 4; ++k) s += rnd() & 0xffff; double g = (s - 131070.0) / 65536.0 * 1.732 * 3.0;
                double amp = (mode == 2 && ((i >> 9) % 7 == 0) && ((i >> 1) % 3 == 0)) ? 60.0 : 0.0;
                int v = (int)floor(127.5 + g + amp + 0.5); h[i] = v < 0 ? 0 : (v > 255 ? 255 : v);
            }
        }
        for (size_t off = 0; off < 2 * n; off += h.size()) CK(hipMemcpy(diq + off, h.data(), h.size(), hipMemcpyHostToDevice));
        const char *names[] = {"stream", "global_lut", "lds_full", "lds_full_swz", "lds_fold", "lds_fold_swz"};
        for (int var = 0; var < 6; ++var) {
            float best = 1e9; uint32_t sum = 0;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipMemset(dout, 0, 64));
                CK(hipEventRecord(e0));
                size_t nvec = n / 8;
                switch (var) {
                    case 0: k_stream<<<2048, 256>>>((const uint4 *)diq, nvec, dlut, dout); break;
                    case 1: k_global<<<2048, 256>>>((const uint4 *)diq, nvec, dlut, dout); break;
                    case 2: k_lds_full<0><<<256, 1024, 131072>>>((const uint4 *)diq, nvec, dlut, dout); break;
                    case 3: k_lds_full<1><<<256, 1024, 131072>>>((const uint4 *)diq, nvec, dlut, dout); break;
                    case 4: k_lds_fold<0><<<1024, 512>>>((const uint4 *)diq, nvec, dlut, dout); break;
                    case 5: k_lds_fold<1><<<1024, 512>>>((const uint4 *)diq, nvec, dlut, dout); break;
                }
                CK(hipGetLastError());
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
                uint32_t o[2]; CK(hipMemcpy(o, dout, 8, hipMemcpyDeviceToHost)); sum = o[1];
            }
            printf("mode %d %-14s %8.3f ms  %7.1f GB/s  sum %08x\n", mode, names[var], best, 2.0 * n / best / 1e6, sum);
        }
    }
    return 0;
}
