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

#if defined(__HIPCC__)
#define RSB_HD __host__ __device__
#else
#define RSB_HD
#endif

namespace rsb {

constexpr uint32_t kBlock = RSB_BLOCK_SAMPLES;
constexpr uint32_t kTrailing = RSB_TRAILING_SAMPLES;

// ---- records shared with the kernels (kernels.hip includes this header) -------------------

// One sliced trial phase of a gated scan offset that can still score >= 0 under some filter state.
// info = combo << 27 | phase << 24 | value; combo says which branch of scoreModesMessage /
// decodeModesMessage (mode_s.c:330-408, 445-555) the phase takes and how many bits the syndrome
// repair flips; value is the 24-bit number the ICAO filter is asked about (the CRC syndrome for
// Address/Parity and Comm-B, the repaired AA otherwise).
struct ItemRec {      // 8 B, written by the scan kernel
    uint32_t g;       // batch-relative scan offset
    uint32_t info;
};
enum : uint32_t {
    kComboApShort = 0,   // DF0/4/5
    kComboApLong = 1,    // DF16/24
    kComboCommB = 2,     // DF20/21
    kComboDf11Iid0 = 3,  // + errors (0..1)
    kComboDf11IidX = 5,  // + errors (0..1)
    kComboDf17 = 7,      // 7: clean, 8/9: 1 bit (AA same / changed), 10/11: 2 bits (same / changed)
    kComboDf18 = 12,     // same layout as DF17
    kComboMarker = 31    // "this offset passed the gate" (exact-statistics mode only)
};
RSB_HD inline uint32_t item_combo(uint32_t info) { return info >> 27; }
RSB_HD inline uint32_t item_phase(uint32_t info) { return (info >> 24) & 7u; }
RSB_HD inline uint32_t item_value(uint32_t info) { return info & 0xffffffu; }
// does this combo need a filter member to score >= 0?
RSB_HD inline bool combo_needs_member(uint32_t c) { return c <= kComboCommB || c == kComboDf11IidX || c == kComboDf11IidX + 1; }

struct Survivor {     // 96 B, written by the finalize kernel
    uint32_t g;
    uint8_t vbytes[5];    // bytes actually sliced per phase: 1, 7 or 14 (demod_2400.c:272-285)
    uint8_t pad;
    uint8_t msg[5][14];
    uint64_t pow134;      // sum of mag^2 over 134 / 268 samples from S[g+19] (demod_2400.c:345-351)
    uint64_t pow268;
};
static_assert(sizeof(ItemRec) == 8, "ItemRec layout");
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
    // for the device mirror: every address either generation holds, and how full the active table is
    void contents(std::vector<uint32_t> &out) const;
    uint32_t active_slots_used() const;
    static constexpr uint32_t kSlots = 4096;

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
// the same magnitudes indexed by P / 2, P = I (255 - I) + Q (255 - Q); 0 where byte pairs with one P disagree (48 pairs, two entries)
const uint16_t *uc8_radial_lut();       // kRadialTotal entries: radial table + patch entries
uint32_t uc8_radial_patch_p1();         // DeviceTables::uc8_patch_p1

// ---- replay ---------------------------------------------------------------------------------------

class Replayer {
public:
    explicit Replayer(int nfix) : crc_(nfix) { reset(); }
    // samples per demodulator call (one struct mag_buf): skip windows and Mode A/C runs start fresh at every multiple of it, counted
    // from the batch's first sample (a batch starts on a buffer boundary)
    void set_block_len(uint32_t n) { block_len_ = n ? n : kBlock; }
    void reset();
    IcaoFilter &filter() { return filter_; }
    const CrcTables &crc() const { return crc_; }
    rsb_stats &stats() { return stats_; }
    // g0 = stream position of the batch (the start of a block); records need not be sorted.
    // Exact-statistics mode (RSB_FLAG_EXACT_STATS): surv holds EVERY gated offset of the batch, and
    // blk_mean_power[b] = mag_buf.mean_power of block b of the nsamples-long batch; the demodulator's
    // counters and power sums are then updated as demod_2400.c:175,304-310,329-337,355-362,378-382 do.
    void run_batch(uint64_t g0, std::vector<Survivor> &surv, std::vector<AcCand> &ac, std::vector<rsb_message> &out,
                   const double *blk_mean_power = nullptr, uint64_t nsamples = 0);
    // After a device replay: msgs holds the batch's Mode S messages in scan order, blk_counts[b] of them per
    // 131072-sample block; interleave each block's Mode A/C messages behind them (readsb.c:883-886).
    void merge_modeac(uint64_t g0, const std::vector<rsb_message> &msgs, const uint32_t *blk_counts, size_t nblocks,
                      std::vector<AcCand> &ac, std::vector<rsb_message> &out);

    // The Mode A/C half of that merge alone: the candidates' messages after the greedy skip, in block order, and how
    // many each block has (counts[nblocks]); the caller interleaves them with the Mode S list wherever that lives.
    void modeac_messages(uint64_t g0, size_t nblocks, std::vector<AcCand> &ac, std::vector<rsb_message> &out, uint32_t *counts);

private:
    CrcTables crc_;
    IcaoFilter filter_;
    rsb_stats stats_;
    uint32_t block_len_ = kBlock;
};

}  // namespace rsb
