// FETCH_SIZE calibration on a known byte count (MI355X_MICROARCH.md "HBM": calibrate in your own access pattern):
//   k_stream : grid-stride 16-B-per-lane reads of a 2 GiB buffer, every byte once
//   k_strips : the scan kernel's pattern -- each wavefront reads 294 consecutive 16-B vectors (4704 B) at a 4096-B stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void k_stream(const uint4 *p, size_t nvec, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(1024) void k_strips(const uint8_t *p, size_t nbytes, uint32_t *out) {
    const size_t nstrips = nbytes / 4096 - 1;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t acc = 0;
    for (size_t s = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6); s < nstrips; s += (size_t)gridDim.x * 16)
        for (int v = lane; v < 294; v += 64) { const uint4 x = *(const uint4 *)(p + s * 4096 + 16 * v); acc ^= x.x ^ x.y ^ x.z ^ x.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t bytes = 2ull << 30;
    uint8_t *d; uint32_t *o;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
    (void)hipMemset(d, 1, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        k_stream<<<2048, 256>>>((const uint4 *)d, bytes / 16, o);
        k_strips<<<256, 1024>>>(d, bytes, o);
    }
    (void)hipDeviceSynchronize();
    printf("k_stream reads %zu bytes; k_strips reads %zu bytes (%.4f x the buffer)\n", bytes, (bytes / 4096 - 1) * 4704, 4704.0 / 4096);
    return 0;
}
