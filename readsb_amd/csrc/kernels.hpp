// kernels.hpp -- launch interface between the C ABI (capi.cpp) and the HIP kernels (kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "modes_host.hpp"

namespace rsb {

// sample source formats a kernel can read; the first three are input_format_t (convert.h:29-32)
enum : int { kFmtUC8 = 0, kFmtSC16 = 1, kFmtSC16Q11 = 2, kFmtMag16 = 3 };

// device counters (one uint32 each)
enum : int {
    kCntKeys = 0,      // key records appended by the scan kernel
    kCntSurvivors,     // survivors appended by the prune kernel
    kCntAc,            // Mode A/C candidates
    kCntGated,         // offsets that passed the preamble gate
    kCntCount = 8
};

struct DeviceTables {      // read-only, built once per context
    const uint16_t *uc8_lut;       // 65536 x u16
    const uint32_t *crc_bytes;     // 256 x u32
    const uint32_t *syn_keys[2];   // [0] = 56-bit frames, [1] = 112-bit frames
    const uint32_t *syn_vals[2];
    uint32_t syn_mask[2];
    int nfix;
};

// One batch = the scan offsets [0, n) of stream S = carry(326) ++ iq(n); see DESIGN.md "stream model".
struct BatchArgs {
    const void *iq;        // n samples
    const void *carry;     // 326 samples of the same format (the tail of the previous batch)
    int carry_valid;       // 0: the 326 leading magnitudes are zero (sdr_ifile.c:200-204)
    int fmt;
    uint64_t n;
    DeviceTables tab;
    KeyRec *keys;
    uint32_t key_cap;
    uint32_t *surv_g;      // offsets of the candidates kept by the prune kernel
    Survivor *survivors;
    uint32_t surv_cap;
    AcCand *ac;
    uint32_t ac_cap;
    uint32_t *adder_bitmap;  // 2^24 bits: superset of every address the filter can hold
    uint32_t *counters;
    uint64_t *block_sums;    // per 131072-block: sum(mag), sum(mag^2) over its new samples
    int emit_dead;           // also keep gated offsets whose five phases are all dead (exact stats)
    int keep_all;            // prune keeps every key record (filter owned by the caller)
};

hipError_t launch_scan(const BatchArgs &a, hipStream_t s);
hipError_t launch_prune_finalize(const BatchArgs &a, hipStream_t s);
hipError_t launch_modeac(const BatchArgs &a, hipStream_t s);
hipError_t launch_convert(int fmt, const void *iq, uint16_t *mag, uint64_t n, const uint16_t *lut, uint64_t *sums, hipStream_t s);

}  // namespace rsb
