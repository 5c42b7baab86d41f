// noalloc.hip -- rsb_convert_noalloc: the stand-alone converter (rsb_convert_device) for callers that cannot allocate.
//
// rsb_convert_device takes its work memory from the context (the UC8 table, the integer sums, and for --dcfilter the tile
// arrays of the chunked scan and the per-block means: hipMalloc'ed by rsb_create / on first use).  This variant runs the
// SAME kernels on a file-scope device array instead and calls no hipMalloc / hipFree / allocating routine: for code under
// HIP graph capture, tight serving loops and harnesses that forbid allocation.  No counterpart in the reference.
//
// The array exists once per GPU (every device that loads the library's code object holds its own copy at its own address,
// until the process ends).  Its address is looked up per device with hipGetSymbolAddress at set-up -- rsb_create does it for
// its device, otherwise the first rsb_convert_noalloc call on a device (that first call also uploads the UC8 table with a
// blocking copy: make it outside any stream capture).  Host code never passes the array's NAME to a kernel (in host code
// that is a host-side stand-in, not a device address); kernels, memsets and copies get the looked-up pointer.
//
// Every call on a GPU shares that GPU's array: what must start at zero is cleared on the call's stream, and two calls that
// overlap on one GPU (different streams, different threads) race.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <mutex>

#include "../../include/readsb_amd.h"
#include "kernels.hpp"
#include "modes_host.hpp"

namespace {

using namespace rsb;

constexpr uint64_t kMaxSamples = RSB_NOALLOC_MAX_SAMPLES;
constexpr size_t kMaxTiles = kMaxSamples / kDcTileSamples + 2;  // launch_convert_dc: one entry per 4096-sample tile, + 1
constexpr size_t kMaxBlocks = kMaxSamples / kBlock + 1;

struct Scratch {
    unsigned long long sums[32];          // [0], [1]: sum(mag), sum(mag^2) of the integer converter (convert.c:71-83); 256 bytes, cleared per call
    double2 tile_sums[kMaxTiles];         // --dcfilter: the chunked scan's per-tile contributions ...
    double2 tile_start[kMaxTiles];        // ... and starting states
    double block_means[2 * kMaxBlocks];   // --dcfilter: mean level / mean power of every 131072-sample block of the call
    uint16_t lut[65536];                  // convert.c:35-61, written once per device at set-up
};

// ~0.66 MB per GPU
static __device__ __align__(256) unsigned char scratch[sizeof(Scratch)];

constexpr int kMaxDevices = 64;
std::mutex g_mu;
Scratch *g_ptr[kMaxDevices];  // per device: where ITS copy of `scratch` lives (nullptr = not set up yet)

// once per device, with that device current
int setup_current_device(int dev, Scratch **out) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_ptr[dev]) {
        void *p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            std::fprintf(stderr, "rsb_convert_noalloc: hipGetSymbolAddress(scratch) on device %d: %s\n", dev, hipGetErrorString(e));
            return RSB_ENODEV;
        }
        Scratch *sp = (Scratch *)p;
        e = hipMemcpy(sp->lut, uc8_magnitude_lut(), sizeof(sp->lut), hipMemcpyHostToDevice);  // blocking; no allocation of ours
        if (e != hipSuccess) {
            std::fprintf(stderr, "rsb_convert_noalloc: UC8 table upload on device %d: %s\n", dev, hipGetErrorString(e));
            return RSB_ENODEV;
        }
        g_ptr[dev] = sp;
    }
    *out = g_ptr[dev];
    return RSB_OK;
}

// the means as rsb_convert_device forms them on the host, here on the device (IEEE double, no contraction: same bits)
__global__ void k_na_means_int(const unsigned long long *sums, uint64_t n, double *out) {  // convert.c:104-110
    out[0] = (double)sums[0] / 65536.0 / (double)(unsigned)n;
    out[1] = (double)sums[1] / 65535.0 / 65535.0 / (double)(unsigned)n;
}
__global__ void k_na_means_dc(const double *block_means, uint64_t n, double *out) {  // one call = one mean; blocks weighted by their length
    const uint64_t nblk = (n + kBlock - 1) / kBlock;
    double lvl = 0, pwr = 0;
    for (uint64_t b = 0; b < nblk; ++b) {
        const double len = (double)(b + 1 < nblk ? (uint64_t)kBlock : n - b * kBlock);
        lvl += block_means[2 * b] * len;
        pwr += block_means[2 * b + 1] * len;
    }
    out[0] = lvl / (double)n;
    out[1] = pwr / (double)n;
}

#define NA_TRY(call)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            std::fprintf(stderr, "rsb_convert_noalloc: %s: %s\n", #call, hipGetErrorString(e_)); \
            return RSB_ENODEV;                                                                  \
        }                                                                                       \
    } while (0)

}  // namespace

// rsb_create's share of the set-up (the library's init call): the device is current there
int rsb_internal_noalloc_setup(int device) {
    if (device < 0 || device >= kMaxDevices) return RSB_EINVAL;
    Scratch *sp;
    return setup_current_device(device, &sp);
}

extern "C" int rsb_convert_noalloc(int input_format, int filter_dc, const void *iq_dev, uint16_t *mag_dev, uint64_t nsamples, void *stream,
                                   double *dc_state_dev, double *means_dev) {
    if (input_format < RSB_UC8 || input_format > RSB_SC16Q11) {
        std::fprintf(stderr, "no suitable converter for format=%d dc=%d\n", input_format, filter_dc);  // convert.c:460-464
        return RSB_EINVAL;
    }
    if (!iq_dev || !mag_dev || (filter_dc && !dc_state_dev)) return RSB_EINVAL;
    if (nsamples > kMaxSamples) {
        std::fprintf(stderr, "rsb_convert_noalloc: %llu samples exceed the fixed work memory of this entry point (at most %llu per call; "
                             "use rsb_convert_device, or split the call)\n", (unsigned long long)nsamples, (unsigned long long)kMaxSamples);
        return RSB_EINVAL;
    }
    if (nsamples == 0) return RSB_OK;
    int dev = -1;
    NA_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return RSB_EINVAL;
    Scratch *sp = nullptr;
    const int rc = setup_current_device(dev, &sp);
    if (rc != RSB_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (filter_dc) {
        // convert.c:113-162 / :164-213 / :374-423 as the chunked scan of rsb_convert_device; the DC state is the caller's
        NA_TRY(launch_convert_dc(input_format, iq_dev, mag_dev, nsamples, sp->tile_sums, sp->tile_start, (double2 *)dc_state_dev,
                                 means_dev ? sp->block_means : nullptr, s));
        if (means_dev) {
            hipLaunchKernelGGL(k_na_means_dc, dim3(1), dim3(1), 0, s, sp->block_means, nsamples, means_dev);
            NA_TRY(hipGetLastError());
        }
        return RSB_OK;
    }
    const bool int_sums = means_dev && input_format == RSB_UC8;
    if (int_sums) NA_TRY(hipMemsetAsync(sp->sums, 0, sizeof(sp->sums), s));  // the last call's sums are still there
    NA_TRY(launch_convert(input_format, iq_dev, mag_dev, nsamples, sp->lut, int_sums ? (uint64_t *)sp->sums : nullptr, s));
    if (int_sums) {
        hipLaunchKernelGGL(k_na_means_int, dim3(1), dim3(1), 0, s, sp->sums, nsamples, means_dev);
        NA_TRY(hipGetLastError());
    } else if (means_dev) {
        // SC16 / SC16Q11: float32 running sums in sample order (convert.c:226-262), straight into the caller's two doubles
        NA_TRY(launch_block_means_float(input_format, iq_dev, nsamples, nsamples, means_dev, s));
    }
    return RSB_OK;
}
