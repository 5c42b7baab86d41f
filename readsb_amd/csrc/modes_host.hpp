// modes_host.hpp -- host-side pieces of the receive path that are inherently sequential or
// one-time: CRC/syndrome tables (crc.c), the ICAO address filter (icao_filter.c), message
// scoring and the CRC-dependent part of decode (mode_s.c:311-409, 424-555, 717-726) and the
// exact replay of the reference's greedy scan over the candidates the GPU kept
// (demod_2400.c:289-374, 611-660).  Product code: must not depend on anything under oracle/.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/readsb_amd.h"

namespace rsb {

constexpr uint32_t kBlock = RSB_BLOCK_SAMPLES;
constexpr uint32_t kTrailing = RSB_TRAILING_SAMPLES;

// ---- records shared with the kernels (kernels.hip includes this header) -------------------

// per-phase classes in a key record (what the phase's score can become, over all filter states)
enum : uint32_t {
    kDead = 0,        // -2 whatever the filter holds
    kNeedCrc = 1,     // Address/Parity DFs: alive iff the 24-bit syndrome is a filter member
    kNeedAddr = 2,    // DF11 with IID != 0: alive iff the (repaired) AA is a filter member
    kAlive = 3        // DF17/18 correctable, DF11 IID == 0 correctable: score >= 0 regardless
};

struct KeyRec {       // 24 B, written by the scan kernel for every gated offset it keeps
    uint32_t g;       // batch-relative scan offset
    uint32_t key[5];  // class << 24 | 24-bit value, one per try_phase 4..8
};

struct Survivor {     // 96 B, written by the finalize kernel
    uint32_t g;
    uint8_t vbytes[5];    // bytes actually sliced per phase: 1, 7 or 14 (demod_2400.c:272-285)
    uint8_t pad;
    uint8_t msg[5][14];
    uint64_t pow134;      // sum of mag^2 over 134 / 268 samples from S[g+19] (demod_2400.c:345-351)
    uint64_t pow268;
};
static_assert(sizeof(KeyRec) == 24, "KeyRec layout");
static_assert(sizeof(Survivor) == 96, "Survivor layout");

struct AcCand {       // 12 B, Mode A/C candidate that passed every test of demod_2400.c:535-622
    uint32_t g;       // batch-relative f1_sample
    uint32_t f2_clock;// block-local 60 MHz clock of F2
    uint32_t modeac;
};

// device-side syndrome hash: open addressing, linear probe, 0 = empty
struct DevSyndromeTable {
    std::vector<uint32_t> keys;  // syndrome
    std::vector<uint32_t> vals;  // errors | bit0 << 8 | bit1 << 16
    uint32_t mask = 0;
};
#if defined(__HIPCC__)
#define RSB_HD __host__ __device__
#else
#define RSB_HD
#endif
RSB_HD inline uint32_t syndrome_hash(uint32_t s, uint32_t mask) { return ((s * 0x9E3779B1u) >> 12) & mask; }

// ---- CRC ------------------------------------------------------------------------------------

struct SyndromeEntry {
    uint32_t syndrome;
    int8_t errors;
    int8_t bit[2];
};

class CrcTables {
public:
    explicit CrcTables(int nfix);
    int nfix() const { return nfix_; }
    const SyndromeEntry *diagnose(uint32_t syndrome, int bits) const;  // crc.c:383-406
    const std::vector<SyndromeEntry> &table(int bits) const { return bits == 56 ? short_ : long_; }
    DevSyndromeTable device_table(int bits) const;

    static uint32_t checksum(const uint8_t *msg, int bits);           // crc.c:67-82
    static const uint32_t *byte_table();                              // crc.c:35
    static const uint32_t *single_bit_syndromes();                    // crc.c:40

private:
    void build(int bits, int max_correct, int max_detect, std::vector<SyndromeEntry> &out);
    int nfix_;
    std::vector<SyndromeEntry> short_, long_;
};

// ---- ICAO filter (icao_filter.c) ----------------------------------------------------------------

class IcaoFilter {
public:
    IcaoFilter() { init(); }
    void init();
    void add(uint32_t addr);
    bool test(uint32_t addr) const;
    void expire();

private:
    static constexpr uint32_t kSize = 4096;
    static uint32_t hash(uint32_t a);
    uint32_t a_[kSize], b_[kSize];
    uint32_t *active_;
};

// ---- scoring / decode -----------------------------------------------------------------------------

int score_message(const CrcTables &crc, const IcaoFilter &filter, const uint8_t *msg, int validbits);
// msg[14] is repaired in place; fills everything in m except timestamp / score / signal_level
int decode_crc_part(const CrcTables &crc, IcaoFilter &filter, rsb_message &m, uint8_t msg[14]);

// UC8 magnitude table, indexed by the little-endian 16-bit sample word (convert.c:35-61)
const uint16_t *uc8_magnitude_lut();

// ---- replay ---------------------------------------------------------------------------------------

class Replayer {
public:
    explicit Replayer(int nfix) : crc_(nfix) { reset(); }
    void reset();
    IcaoFilter &filter() { return filter_; }
    const CrcTables &crc() const { return crc_; }
    rsb_stats &stats() { return stats_; }
    // g0 = stream position of the batch (multiple of kBlock); records need not be sorted.
    void run_batch(uint64_t g0, std::vector<Survivor> &surv, std::vector<AcCand> &ac, std::vector<rsb_message> &out);

private:
    CrcTables crc_;
    IcaoFilter filter_;
    rsb_stats stats_;
};

}  // namespace rsb
