// modes_host.cpp -- see modes_host.hpp.  Product code; nothing here comes from oracle/.
#include "modes_host.hpp"
#include "kernels.hpp"

#include <algorithm>
#include <cmath>
#include <mutex>

namespace rsb {

// =============================================================================== CRC tables

namespace {

constexpr uint32_t kPoly = 0xfff409u;  // crc.c:31

struct CrcConst {
    uint32_t byte_table[256];
    uint32_t sbs[112];
    CrcConst() {
        // crc.c:46-57: remainder of every single-byte message
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i << 16;
            for (int j = 0; j < 8; ++j) c = (c & 0x800000u) ? ((c << 1) ^ kPoly) : (c << 1);
            byte_table[i] = c & 0xffffffu;
        }
        // crc.c:59-64: syndrome of a single set bit i of a 112-bit frame.  The CRC is linear, so
        // walk backwards from the parity field: bit 111-k of the frame is x^k for k < 24, and each
        // earlier data bit multiplies by x once more (mod the generator).
        for (int k = 0; k < 24; ++k) sbs[111 - k] = 1u << k;
        uint32_t r = 1u << 23;
        for (int i = 87; i >= 0; --i) {
            r = (r & 0x800000u) ? (((r << 1) ^ kPoly) & 0xffffffu) : (r << 1);
            sbs[i] = r;
        }
    }
};

const CrcConst &crc_const() {
    static const CrcConst c;
    return c;
}

}  // namespace

uint32_t CrcTables::checksum(const uint8_t *msg, int bits) {
    const uint32_t *t = crc_const().byte_table;
    const int n = bits / 8;
    uint32_t rem = 0;
    for (int i = 0; i < n - 3; ++i) rem = ((rem << 8) ^ t[msg[i] ^ (rem >> 16)]) & 0xffffffu;
    return rem ^ (uint32_t(msg[n - 3]) << 16) ^ (uint32_t(msg[n - 2]) << 8) ^ msg[n - 1];
}

const uint32_t *CrcTables::byte_table() { return crc_const().byte_table; }
const uint32_t *CrcTables::single_bit_syndromes() { return crc_const().sbs; }

CrcTables::CrcTables(int nfix) : nfix_(nfix) {
    // modesChecksumInit, crc.c:353-378
    if (nfix == 1) {
        build(56, 1, 1, short_);
        build(112, 1, 1, long_);
    } else if (nfix >= 2) {
        build(56, 2, 4, short_);
        build(112, 2, 4, long_);
    }
}

// prepareErrorTable, crc.c:180-350, restated as: every 1..max_correct-bit pattern over bits
// 5..bits-1 (crc.c:210-211); keep a syndrome only if exactly one such pattern produces it
// (crc.c:243-263) and no (max_correct+1 .. max_detect)-bit pattern does (crc.c:266-294).
void CrcTables::build(int bits, int max_correct, int max_detect, std::vector<SyndromeEntry> &out) {
    const uint32_t *sbs = crc_const().sbs + (112 - bits);
    std::vector<uint8_t> hits(1u << 24, 0);  // 0, 1, 2 (= "more than one"), 3 (= killed)
    auto bump = [&](uint32_t s) {
        if (hits[s] < 2) ++hits[s];
    };
    for (int i = 5; i < bits; ++i) {
        bump(sbs[i]);
        if (max_correct >= 2)
            for (int j = i + 1; j < bits; ++j) bump(sbs[i] ^ sbs[j]);
    }
    if (max_detect > max_correct) {
        // 3- and 4-bit patterns (the only case the reference uses is correct 2 / detect 4)
        for (int i = 5; i < bits; ++i)
            for (int j = i + 1; j < bits; ++j) {
                const uint32_t s2 = sbs[i] ^ sbs[j];
                for (int k = j + 1; k < bits; ++k) {
                    const uint32_t s3 = s2 ^ sbs[k];
                    if (max_correct < 3 && max_detect >= 3) hits[s3] = 3;
                    if (max_detect >= 4)
                        for (int l = k + 1; l < bits; ++l) hits[s3 ^ sbs[l]] = 3;
                }
            }
    }
    out.clear();
    for (int i = 5; i < bits; ++i) {
        if (hits[sbs[i]] == 1) out.push_back({sbs[i], 1, {int8_t(i), -1}});
        if (max_correct >= 2)
            for (int j = i + 1; j < bits; ++j) {
                const uint32_t s = sbs[i] ^ sbs[j];
                if (hits[s] == 1) out.push_back({s, 2, {int8_t(i), int8_t(j)}});
            }
    }
    std::sort(out.begin(), out.end(), [](const SyndromeEntry &x, const SyndromeEntry &y) { return x.syndrome < y.syndrome; });
}

const SyndromeEntry *CrcTables::diagnose(uint32_t syndrome, int bits) const {
    static const SyndromeEntry kNoErrors = {0, 0, {0, 0}};  // crc.c:28,389-390
    if (syndrome == 0) return &kNoErrors;
    const std::vector<SyndromeEntry> &t = (bits == 56) ? short_ : long_;
    auto it = std::lower_bound(t.begin(), t.end(), syndrome, [](const SyndromeEntry &e, uint32_t s) { return e.syndrome < s; });
    return (it != t.end() && it->syndrome == syndrome) ? &*it : nullptr;
}

DevSyndromeTable CrcTables::device_table(int bits) const {
    const std::vector<SyndromeEntry> &t = (bits == 56) ? short_ : long_;
    uint32_t cap = 64;
    while (cap < 2 * t.size() + 1) cap <<= 1;
    DevSyndromeTable d;
    d.keys.assign(cap, 0);
    d.vals.assign(cap, 0);
    d.mask = cap - 1;
    for (const SyndromeEntry &e : t) {
        uint32_t h = syndrome_hash(e.syndrome, d.mask);
        while (d.keys[h]) h = (h + 1) & d.mask;
        d.keys[h] = e.syndrome;
        d.vals[h] = uint32_t(e.errors) | (uint32_t(uint8_t(e.bit[0])) << 8) | (uint32_t(uint8_t(e.bit[1])) << 16);
    }
    return d;
}

// =============================================================================== ICAO filter

uint32_t IcaoFilter::hash(uint32_t a) {  // icao_filter.c:42-63, Jenkins one-at-a-time over 3 bytes
    uint32_t h = 0;
    for (int shift = 0; shift < 24; shift += 8) {
        h += (a >> shift) & 0xff;
        h += h << 10;
        h ^= h >> 6;
    }
    h += h << 3;
    h ^= h >> 11;
    h += h << 15;
    return h & (kSize - 1);
}

void IcaoFilter::init() {  // icao_filter.c:65-69
    std::memset(a_, 0, sizeof(a_));
    std::memset(b_, 0, sizeof(b_));
    active_ = a_;
}

void IcaoFilter::add(uint32_t addr) {  // icao_filter.c:71-95
    uint32_t h0 = hash(addr), h = h0;
    while (active_[h] && active_[h] != addr) {
        h = (h + 1) & (kSize - 1);
        if (h == h0) return;  // table full: the reference only prints
    }
    if (!active_[h]) active_[h] = addr;
    // second key with the top byte ignored, for DF20/21 Data/Parity (icao_filter.c:84-94)
    const uint32_t low = addr & 0x00ffffu;
    h0 = h = hash(low);
    while (active_[h] && (active_[h] & 0x00ffffu) != low) {
        h = (h + 1) & (kSize - 1);
        if (h == h0) return;
    }
    if (!active_[h]) active_[h] = addr;
}

bool IcaoFilter::test(uint32_t addr) const {  // icao_filter.c:97-119: both generations are consulted
    for (const uint32_t *t : {a_, b_}) {
        uint32_t h0 = hash(addr), h = h0;
        while (t[h] && t[h] != addr) {
            h = (h + 1) & (kSize - 1);
            if (h == h0) break;
        }
        if (t[h] == addr) return true;  // NB: addr 0 matches an empty slot, as in the reference
    }
    return false;
}

void IcaoFilter::expire() {  // icao_filter.c:151-158 (one flip)
    uint32_t *next = (active_ == a_) ? b_ : a_;
    std::memset(next, 0, sizeof(a_));
    active_ = next;
}

void IcaoFilter::contents(std::vector<uint32_t> &out) const {
    out.clear();
    for (const uint32_t *t : {a_, b_})
        for (uint32_t i = 0; i < kSize; ++i)
            if (t[i]) out.push_back(t[i]);
}

uint32_t IcaoFilter::active_slots_used() const {
    uint32_t n = 0;
    for (uint32_t i = 0; i < kSize; ++i) n += active_[i] != 0;
    return n;
}

// =============================================================================== score / decode

namespace {

inline uint32_t aa_field(const uint8_t *msg) {  // getbits(msg, 9, 32)
    return (uint32_t(msg[1]) << 16) | (uint32_t(msg[2]) << 8) | msg[3];
}

inline bool all_zero(const uint8_t *p, int n) {
    for (int i = 0; i < n; ++i)
        if (p[i]) return false;
    return true;
}

// mode_s.c:266-281
inline void repair_aa(uint32_t &addr, const SyndromeEntry *e) {
    for (int i = 0; i < e->errors; ++i)
        if (e->bit[i] >= 8 && e->bit[i] <= 31) addr ^= 1u << (31 - e->bit[i]);
}

inline void flip(uint8_t *msg, const SyndromeEntry *e) {  // modesChecksumFix, crc.c:410-418
    for (int i = 0; i < e->errors; ++i) msg[e->bit[i] >> 3] ^= uint8_t(1u << (7 - (e->bit[i] & 7)));
}

enum DfKind { kBadDf, kAddressParity, kAllCall, kSquitter, kCommB };

inline DfKind df_kind(int df) {
    switch (df) {
        case 0: case 4: case 5: case 16:
        case 24: case 25: case 26: case 27: case 28: case 29: case 30: case 31:
            return kAddressParity;
        case 11:
            return kAllCall;
        case 17: case 18:
            return kSquitter;
        case 20: case 21:
            return kCommB;
        default:
            return kBadDf;
    }
}

}  // namespace

// scoreModesMessage, mode_s.c:311-409
int score_message(const CrcTables &crc, const IcaoFilter &filter, const uint8_t *msg, int validbits) {
    if (validbits < 56) return -2;
    const int df = msg[0] >> 3;
    const int msgbits = (df & 0x10) ? 112 : 56;  // mode_s.c:81-83
    if (validbits < msgbits) return -2;
    if (all_zero(msg, msgbits / 8)) return -2;
    const uint32_t syn = CrcTables::checksum(msg, msgbits);
    switch (df_kind(df)) {
        case kAddressParity:
            return filter.test(syn) ? 1000 : -1;
        case kAllCall: {
            const SyndromeEntry *e = crc.diagnose(syn & 0xffff80u, msgbits);
            if (!e || e->errors > 1) return -2;  // mode_s.c:350-358
            uint32_t addr = aa_field(msg);
            repair_aa(addr, e);
            const bool known = filter.test(addr);
            if ((syn & 0x7f) == 0) return (known ? 1600 : 750) / (e->errors + 1);
            return known ? 1000 / (e->errors + 1) : -1;
        }
        case kSquitter: {
            const SyndromeEntry *e = crc.diagnose(syn, msgbits);
            if (!e) return -2;
            uint32_t addr = aa_field(msg);
            repair_aa(addr, e);
            return (filter.test(addr) ? 1800 : 1400) / (e->errors + 1);
        }
        case kCommB:
            return filter.test(syn) ? 1000 : -2;  // mode_s.c:391-403
        default:
            return -2;
    }
}

// decodeModesMessage up to the field decode, mode_s.c:424-555, plus the filter add of 717-726
int decode_crc_part(const CrcTables &crc, IcaoFilter &filter, rsb_message &m, uint8_t msg[14]) {
    if (all_zero(msg, 7)) return -2;
    const int df = msg[0] >> 3;
    m.msgtype = uint8_t(df);
    m.msgbits = (df & 0x10) ? 112 : 56;
    m.crc = CrcTables::checksum(msg, m.msgbits);
    m.correctedbits = 0;
    m.addr = 0;
    m.iid = 0;
    switch (df_kind(df)) {
        case kAddressParity:
        case kCommB:
            // mode_s.c:461-466 and :541-550 come to the same thing: exact filter match or -1
            if (!filter.test(m.crc)) return -1;
            m.source = 3;  // SOURCE_MODE_S
            m.addr = m.crc;
            break;
        case kAllCall:
            m.iid = m.crc & 0x7f;
            if (m.crc & 0xffff80u) {
                const SyndromeEntry *e = crc.diagnose(m.crc & 0xffff80u, m.msgbits);
                if (!e || e->errors > 1) return -2;
                m.correctedbits = e->errors;
                flip(msg, e);
                if (!filter.test(aa_field(msg))) return -1;  // mode_s.c:495-498
            }
            m.source = 4;  // SOURCE_MODE_S_CHECKED
            break;
        case kSquitter:
            if (m.crc != 0) {
                const SyndromeEntry *e = crc.diagnose(m.crc, m.msgbits);
                if (!e) return -2;
                const uint32_t before = aa_field(msg);
                m.correctedbits = e->errors;
                flip(msg, e);
                const uint32_t after = aa_field(msg);
                if (before != after && !filter.test(after)) return -1;  // mode_s.c:524-526
            }
            m.source = 7;  // SOURCE_ADSB
            break;
        default:
            return -2;
    }
    if (df == 11 || df == 17 || df == 18) m.addr = aa_field(msg);  // mode_s.c:560-562
    if (!m.correctedbits && (df == 17 || (df == 11 && m.iid == 0))) filter.add(m.addr);  // mode_s.c:717-726
    return 0;
}

// =============================================================================== UC8 LUT

static uint32_t radial_p1 = 0xffffffffu;  // P of patch class 1 (set with the table)

const uint16_t *uc8_radial_lut() {
    // convert.c:35-61 tabulates a function of (2 I - 255)^2 + (2 Q - 255)^2 = 130050 - 4 P, so P / 2 indexes the magnitudes --
    // except where the reference's float rounding gives byte pairs with one P two different values (magnitudes 60465 / 60466 and
    // 49681 / 49682): those entries hold 0, a magnitude no sample has (the smallest is 363), and the kernels take such samples from
    // the full table.  Found here by comparing, not listed by hand.
    static std::vector<uint16_t> rad;
    static std::once_flag once;
    std::call_once(once, [] {
        const uint16_t *full = uc8_magnitude_lut();
        rad.assign(rsb::kRadialEntries, 0);
        std::vector<uint8_t> state(rsb::kRadialEntries, 0);  // 0 unset, 1 set, 2 conflicting
        for (int q = 0; q < 256; ++q)
            for (int i = 0; i < 256; ++i) {
                const int k = (i * (255 - i) + q * (255 - q)) >> 1;
                const uint16_t v = full[(q << 8) | i];
                if (state[k] == 0) {
                    rad[k] = v;
                    state[k] = 1;
                } else if (state[k] == 1 && rad[k] != v) {
                    state[k] = 2;
                }
            }
        for (int k = 0; k < rsb::kRadialEntries; ++k)
            if (state[k] != 1) rad[k] = 0;
        // The byte pairs of a conflicting entry lie on one circle a^2 + b^2 = 130050 - 4 P, a = |2 I - 255|, b = |2 Q - 255|, and the
        // table value depends on the unordered pair {a, b} only (the squares are equal for +-a, the sum commutes): a alone picks
        // it.  Two such P exist; the kernels patch from these 2 x 128 entries in LDS instead of the full table in global memory.
        std::vector<int> classes;
        for (int k = 0; k < rsb::kRadialEntries; ++k)
            if (state[k] == 2) classes.push_back(k);
        if (classes.size() > 2) {
            std::fprintf(stderr, "readsb_amd: UC8 table: %zu radial entries with two magnitudes (2 provided for)\n", classes.size());
            std::abort();
        }
        rad.resize(rsb::kRadialTotal, 0);
        radial_p1 = classes.size() == 2 ? 2u * (uint32_t)classes[1] : 0xffffffffu;
        for (int q = 0; q < 256; ++q)
            for (int i = 0; i < 256; ++i) {
                const int k = (i * (255 - i) + q * (255 - q)) >> 1;
                if (state[k] != 2) continue;
                const int cls = (classes.size() == 2 && k == classes[1]) ? 1 : 0, aidx = i >= 128 ? i - 128 : 127 - i;
                uint16_t &e = rad[rsb::kRadialEntries + 128 * cls + aidx];
                const uint16_t v = full[(q << 8) | i];
                if (e != 0 && e != v) {
                    std::fprintf(stderr, "readsb_amd: UC8 table: |2 I - 255| does not determine the magnitude on circle %d\n", 2 * k);
                    std::abort();
                }
                e = v;
            }
    });
    return rad.data();
}

uint32_t uc8_radial_patch_p1() {
    (void)uc8_radial_lut();
    return radial_p1;
}

const uint16_t *uc8_magnitude_lut() {
    static std::vector<uint16_t> lut;
    static std::once_flag once;
    std::call_once(once, [] {
        lut.resize(65536);
        // convert.c:45-58.  The sample word is I | Q << 8 and the value is symmetric in (I, Q).
        // Built on the host with separately rounded float ops (this file is compiled with
        // -ffp-contract=off), exactly as the reference's x86-64 build does.
        float f[256];
        for (int v = 0; v < 256; ++v) f[v] = float((v - 127.5) / 127.5);
        for (int q = 0; q < 256; ++q)
            for (int i = 0; i < 256; ++i) {
                volatile float a = f[i] * f[i];
                volatile float b = f[q] * f[q];
                float magsq = a + b;
                if (magsq > 1) magsq = 1;
                volatile float scaled = std::sqrt(magsq) * 65535.0f;
                lut[(q << 8) | i] = uint16_t(scaled + 0.5f);
            }
    });
    return lut.data();
}

// =============================================================================== replay

void Replayer::reset() {
    filter_.init();
    std::memset(&stats_, 0, sizeof(stats_));
}

namespace {

template <typename T>
void sort_by_g(std::vector<T> &v) {
    // LSD radix sort on the 32-bit offset, 11 bits per pass; stable
    if (v.size() < 2) return;
    std::vector<T> tmp(v.size());
    T *src = v.data(), *dst = tmp.data();
    for (int shift = 0; shift < 33; shift += 11) {
        size_t count[2049] = {0};
        for (size_t i = 0; i < v.size(); ++i) ++count[((src[i].g >> shift) & 2047u) + 1];
        for (int b = 0; b < 2048; ++b) count[b + 1] += count[b];
        for (size_t i = 0; i < v.size(); ++i) dst[count[(src[i].g >> shift) & 2047u]++] = src[i];
        std::swap(src, dst);
    }
    if (src != v.data()) std::memcpy(v.data(), src, v.size() * sizeof(T));
}

}  // namespace

// demodulate2400AC's emission for one block: greedy skip (demod_2400.c:659) over its candidates
static void replay_modeac_block(uint64_t g0, uint64_t blk_start, uint64_t blk_end, const std::vector<AcCand> &ac, size_t &ia, rsb_stats &stats,
                                std::vector<rsb_message> &out) {
    uint64_t next_f = 0;
    for (; ia < ac.size() && g0 + ac[ia].g < blk_end; ++ia) {
        const AcCand &c = ac[ia];
        const uint64_t g = g0 + c.g;
        if (g < next_f) continue;
        rsb_message m;
        std::memset(&m, 0, sizeof(m));
        m.timestamp = 5 * blk_start + c.f2_clock / 5;  // demod_2400.c:649
        // decodeModeAMessage, mode_ac.c:165-199
        m.source = 1;  // SOURCE_MODE_AC
        m.msgtype = 32;
        m.msgbits = 16;
        m.msg[0] = uint8_t(c.modeac >> 8);
        m.msg[1] = uint8_t(c.modeac);
        m.addr = (c.modeac & 0x0000FF7Fu) | (1u << 24);
        next_f = g + 70;  // demod_2400.c:659 plus the loop's own ++
        ++stats.demod_modeac;
        out.push_back(m);
    }
}

void Replayer::run_batch(uint64_t g0, std::vector<Survivor> &surv, std::vector<AcCand> &ac, std::vector<rsb_message> &out,
                         const double *blk_mean_power, uint64_t nsamples) {
    sort_by_g(surv);
    sort_by_g(ac);
    const bool exact = blk_mean_power != nullptr;
    std::vector<uint64_t> blk_scaled;  // exact statistics: sum_scaled_signal_power of every block (demod_2400.c:357)
    const uint64_t L = block_len_;
    if (exact) blk_scaled.assign(size_t((nsamples + L - 1) / L), 0);
    size_t is = 0, ia = 0;
    while (is < surv.size() || ia < ac.size()) {
        // the next 131072-sample block that holds anything; each block is one demodulate2400 call
        // followed by one demodulate2400AC call (readsb.c:883-886), so skip windows start fresh
        uint64_t blk = UINT64_MAX;
        if (is < surv.size()) blk = surv[is].g / L;
        if (ia < ac.size()) blk = std::min(blk, uint64_t(ac[ia].g / L));
        const uint64_t blk_start = g0 + blk * L, blk_end = blk_start + L;

        uint64_t next_j = 0;  // first offset the greedy scan looks at again
        for (; is < surv.size() && g0 + surv[is].g < blk_end; ++is) {
            Survivor &s = surv[is];
            const uint64_t g = g0 + s.g;
            if (g < next_j) continue;  // inside the previous message (demod_2400.c:371)
            if (exact) ++stats_.demod_preambles;  // demod_2400.c:175
            int best = -2, best_phase = -1;
            for (int p = 0; p < 5; ++p) {  // demod_2400.c:179-301, strict '>' keeps the earliest phase
                const int sc = score_message(crc_, filter_, s.msg[p], s.vbytes[p] * 8);
                if (sc > best) {
                    best = sc;
                    best_phase = p;
                }
            }
            if (best < 0) {  // demod_2400.c:304-310
                if (exact) ++(best == -1 ? stats_.demod_rejected_unknown_icao : stats_.demod_rejected_bad);
                continue;
            }
            rsb_message m;
            std::memset(&m, 0, sizeof(m));
            m.timestamp = 5 * g + (8 + 56) * 12 + uint64_t(best_phase + 4);  // demod_2400.c:320
            m.score = best;
            uint8_t work[14];
            std::memcpy(work, s.msg[best_phase], 14);
            const int result = decode_crc_part(crc_, filter_, m, work);
            if (result < 0) {  // demod_2400.c:329-335: no skip
                if (exact) ++(result == -1 ? stats_.demod_rejected_unknown_icao : stats_.demod_rejected_bad);
                continue;
            }
            std::memcpy(m.msg, work, m.msgbits / 8);
            const int signal_len = m.msgbits * 12 / 5;  // demod_2400.c:345
            const uint64_t scaled = (m.msgbits == 112) ? s.pow268 : s.pow134;
            const double signal_power = scaled / 65535.0 / 65535.0;  // demod_2400.c:353-354
            m.signal_level = signal_power / signal_len;
            next_j = g + signal_len + 1;
            ++stats_.demod_accepted[m.correctedbits];
            if (exact) {  // demod_2400.c:355-362
                stats_.signal_power_sum += signal_power;
                stats_.signal_power_count += uint64_t(signal_len);
                blk_scaled[size_t((g - g0) / L)] += scaled;
                if (m.signal_level > stats_.peak_signal_power) stats_.peak_signal_power = m.signal_level;
                if (m.signal_level > 0.50119) ++stats_.strong_signal_count;
            }
            out.push_back(m);
        }

        replay_modeac_block(g0, blk_start, blk_end, ac, ia, stats_, out);
    }
    for (size_t b = 0; b < blk_scaled.size(); ++b) {  // demod_2400.c:378-382, once per demodulate2400 call
        const uint64_t len = std::min<uint64_t>(L, nsamples - b * L);
        const double sum_signal_power = blk_scaled[b] / 65535.0 / 65535.0;
        stats_.noise_power_sum += (blk_mean_power[b] * len - sum_signal_power);
        stats_.noise_power_count += len;
    }
}

void Replayer::merge_modeac(uint64_t g0, const std::vector<rsb_message> &msgs, const uint32_t *blk_counts, size_t nblocks,
                            std::vector<AcCand> &ac, std::vector<rsb_message> &out) {
    sort_by_g(ac);
    size_t im = 0, ia = 0;
    for (size_t b = 0; b < nblocks; ++b) {
        const uint64_t blk_start = g0 + b * block_len_;
        out.insert(out.end(), msgs.begin() + im, msgs.begin() + im + blk_counts[b]);
        im += blk_counts[b];
        replay_modeac_block(g0, blk_start, blk_start + block_len_, ac, ia, stats_, out);
    }
}

void Replayer::modeac_messages(uint64_t g0, size_t nblocks, std::vector<AcCand> &ac, std::vector<rsb_message> &out, uint32_t *counts) {
    sort_by_g(ac);
    size_t ia = 0;
    for (size_t b = 0; b < nblocks; ++b) {
        const uint64_t blk_start = g0 + b * block_len_;
        const size_t before = out.size();
        replay_modeac_block(g0, blk_start, blk_start + block_len_, ac, ia, stats_, out);
        counts[b] = uint32_t(out.size() - before);
    }
}

}  // namespace rsb
