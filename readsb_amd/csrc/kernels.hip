// kernels.hip -- gfx950 kernels of the receive path.
//
//   k_scan      persistent workgroups; per 8192-offset tile: convert.c (IQ -> magnitude through a
//               folded 32 KB LUT in LDS, magnitudes live in LDS only) -> demod_2400.c:109-155 as
//               bit-parallel wavefront ballots (edge patterns of 64 lanes x 8 offsets per scalar op)
//               -> :158-172 on the ~5 % that survive -> :179-286 five trial phases, first the DF
//               field, then dense 56-/112-bit slicing by length class -> crc.c:67-82 / :383-406 ->
//               the filter-independent half of mode_s.c:311-409.  Emits one 8-byte item per trial
//               phase that can still score >= 0 under some filter state, and marks every address the
//               filter could ever learn (mode_s.c:717-726) in a 2^24-bit set.
//   k_prune     drops items that need a filter member which is not in that superset (safe: such a
//               phase scores < 0 whatever happened before it; demod_2400.c:290,304-310).
//   sort/unique hipCUB radix sort of the kept items by (offset, phase); distinct offsets ascending.
//   k_finalize  re-slices those few offsets into full records (5 x 14 message bytes + the signal
//               power sums of demod_2400.c:345-351) for the host replay.
//   k_block_sums / k_modeac   convert.c:71-83 block statistics and demod_2400.c:476-622.
//   k_convert   the stand-alone iq_convert_fn (magnitudes written to HBM) for the drop-in converter.
//
// Compiled with -ffp-contract=off: every float/double product and sum below rounds on its own, as
// in the reference's x86-64 build.
#include "kernels.hpp"

#include <hipcub/hipcub.hpp>

#include <type_traits>

namespace rsb {
namespace {

constexpr int kHalo = 304;  // look-ahead past the last offset of a tile: 290 needed, 297 used
constexpr int kThreads = 256;

// ------------------------------------------------------------------------------------------------
// sample -> magnitude

__device__ __forceinline__ uint32_t float_mag(float fI, float fQ) {
    // convert.c:236-243 (and :353-360): separately rounded products and sum, clamp, IEEE sqrt
    float magsq = fI * fI + fQ * fQ;
    if (magsq > 1.0f) magsq = 1.0f;
    const float mag = sqrtf(magsq);
    return (uint32_t)(mag * 65535.0f + 0.5f);
}

template <int FMT>
__device__ __forceinline__ uint32_t mag_of(const void *base, uint64_t i, const uint16_t *lut) {
    if constexpr (FMT == kFmtUC8) {
        return lut[((const uint16_t *)base)[i]];  // convert.c:80
    } else if constexpr (FMT == kFmtMag16) {
        return ((const uint16_t *)base)[i];
    } else {
        const short2 v = ((const short2 *)base)[i];
        constexpr float scale = (FMT == kFmtSC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f);  // exact: powers of two
        return float_mag((float)v.x * scale, (float)v.y * scale);
    }
}

// S[s] of the batch stream S = carry(326) ++ iq(n); zero outside
template <int FMT>
__device__ __forceinline__ uint32_t stream_mag(const BatchArgs &a, int64_t s) {
    if (s < 0) return 0;
    if (s < (int64_t)kTrailing) return a.carry_valid ? mag_of<FMT>(a.carry, (uint64_t)s, a.tab.uc8_lut) : 0u;
    s -= kTrailing;
    if ((uint64_t)s >= a.n) return 0;
    return mag_of<FMT>(a.iq, (uint64_t)s, a.tab.uc8_lut);
}

// ------------------------------------------------------------------------------------------------
// demod_2400.c:109-172, the whole gate on one offset

__device__ __forceinline__ bool preamble_gate(const uint16_t *p) {
    if (!(p[0] < p[1] && p[12] > p[13])) return false;
    int high;
    uint32_t sig, noise;
    if (p[1] > p[2] && p[2] < p[3] && p[3] > p[4] && p[8] < p[9] && p[9] > p[10] && p[10] < p[11]) {
        high = (p[1] + p[3] + p[9] + p[11] + p[12]) / 4;
        sig = p[1] + p[3] + p[9];
        noise = p[5] + p[6] + p[7];
    } else if (p[1] > p[2] && p[2] < p[3] && p[3] > p[4] && p[8] < p[9] && p[9] > p[10] && p[11] < p[12]) {
        high = (p[1] + p[3] + p[9] + p[12]) / 4;
        sig = p[1] + p[3] + p[9] + p[12];
        noise = p[5] + p[6] + p[7] + p[8];
    } else if (p[1] > p[2] && p[2] < p[3] && p[4] > p[5] && p[8] < p[9] && p[10] > p[11] && p[11] < p[12]) {
        high = (p[1] + p[3] + p[4] + p[9] + p[10] + p[12]) / 4;
        sig = p[1] + p[12];
        noise = p[6] + p[7];
    } else if (p[1] > p[2] && p[3] < p[4] && p[4] > p[5] && p[9] < p[10] && p[10] > p[11] && p[11] < p[12]) {
        high = (p[1] + p[4] + p[10] + p[12]) / 4;
        sig = p[1] + p[4] + p[10] + p[12];
        noise = p[5] + p[6] + p[7] + p[8];
    } else if (p[2] > p[3] && p[3] < p[4] && p[4] > p[5] && p[9] < p[10] && p[10] > p[11] && p[11] < p[12]) {
        high = (p[1] + p[2] + p[4] + p[10] + p[12]) / 4;
        sig = p[4] + p[10] + p[12];
        noise = p[6] + p[7] + p[8];
    } else {
        return false;
    }
    if (sig * 2 < 3 * noise) return false;
    if ((int)p[5] >= high || (int)p[6] >= high || (int)p[7] >= high || (int)p[8] >= high || (int)p[14] >= high ||
        (int)p[15] >= high || (int)p[16] >= high || (int)p[17] >= high || (int)p[18] >= high)
        return false;
    return true;
}

// ------------------------------------------------------------------------------------------------
// demod_2400.c:48-66 and :179-286 in closed form: bit k of trial phase tp is the sign of correlator
// (tp + 12k) % 5 at sample 19 + (tp + 12k) / 5 after the preamble start.

template <int PH>
__device__ __forceinline__ int corr(const uint16_t *m) {
    const int a = m[0], b = m[1], c = m[2];
    if constexpr (PH == 0) return 5 * a - 3 * b - 2 * c;
    if constexpr (PH == 1) return 4 * a - b - 3 * c;
    if constexpr (PH == 2) return 3 * a + b - 4 * c;
    if constexpr (PH == 3) return 2 * a + 3 * b - 5 * c;
    return a + 5 * b - 5 * c - (int)m[3];
}

__device__ __forceinline__ int slice_corr(const uint16_t *m, int phase) {
    switch (phase) {
        case 0: return corr<0>(m);
        case 1: return corr<1>(m);
        case 2: return corr<2>(m);
        case 3: return corr<3>(m);
        default: return corr<4>(m);
    }
}

// Generic (per-thread, any phase) slicer: d = &S[g + 19]. Bits are packed MSB-first into w[0..3]
// (message byte i = bits 31-8*(i&3).. of w[i>>2]). Returns the number of bytes sliced: 14, 7, or 1
// when the DF is one the demodulator gives up on (demod_2400.c:272-285).
__device__ int slice_phase(const uint16_t *d, int tp, uint32_t w[4]) {
    int s = tp / 5, ph = tp % 5;
    w[0] = w[1] = w[2] = w[3] = 0;
    int nbits = 8;
    for (int k = 0; k < nbits; ++k) {
        const uint32_t bit = slice_corr(d + s, ph) > 0;
        w[k >> 5] |= bit << (31 - (k & 31));
        ph += 2;
        s += 2;
        if (ph >= 5) {
            ph -= 5;
            s += 1;
        }
        if (k == 7) {
            switch (w[0] >> 27) {
                case 0: case 4: case 5: case 11:
                    nbits = 56;
                    break;
                case 16: case 17: case 18: case 20: case 21: case 24:
                    nbits = 112;
                    break;
                default:
                    break;
            }
        }
    }
    return nbits >> 3;
}

// The dense slicer: within any 5 consecutive bits every correlator is used exactly once, so a lane
// can evaluate "correlator 0, 1, 2, 3, 4" in that fixed order -- the same instruction stream for
// every trial phase -- and only the sample offset and the bit each result lands on differ per lane.
struct RotSlicer {
    int off[5];       // sample offset (relative to the group base) at which correlator q is evaluated
    uint32_t bit[5];  // the bit of the 5-bit group that correlator q decides (MSB = earliest bit)
    __device__ __forceinline__ explicit RotSlicer(int tp) {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int j = ((q + 10 - tp) * 3) % 5;  // bit j of the group uses correlator (tp + 2j) % 5 == q
            off[q] = (tp + 12 * j) / 5;
            bit[q] = 16u >> j;
        }
    }
    // five bits starting at bit 5*G of the message; d = &S[g + 19]
    __device__ __forceinline__ uint32_t group(const uint16_t *d, int G) const {
        const uint16_t *b = d + 12 * G;
        uint32_t r = 0;
        r |= (corr<0>(b + off[0]) > 0) ? bit[0] : 0u;
        r |= (corr<1>(b + off[1]) > 0) ? bit[1] : 0u;
        r |= (corr<2>(b + off[2]) > 0) ? bit[2] : 0u;
        r |= (corr<3>(b + off[3]) > 0) ? bit[3] : 0u;
        r |= (corr<4>(b + off[4]) > 0) ? bit[4] : 0u;
        return r;
    }
};

__device__ __forceinline__ uint32_t msg_byte(const uint32_t w[4], int i) { return (w[i >> 2] >> (24 - 8 * (i & 3))) & 0xffu; }

// modesChecksum, crc.c:67-82, on the packed words; crcb = crc_table[256] (in LDS)
template <int NBYTES>
__device__ __forceinline__ uint32_t syndrome_of(const uint32_t w[4], const uint32_t *crcb) {
    uint32_t rem = 0;
#pragma unroll
    for (int i = 0; i < NBYTES - 3; ++i) rem = ((rem << 8) ^ crcb[msg_byte(w, i) ^ (rem >> 16)]) & 0xffffffu;
    const uint32_t tail = (NBYTES == 7) ? ((w[1] >> 8) & 0xffffffu) : (((w[2] & 0xffu) << 16) | (w[3] >> 16));
    return rem ^ tail;
}

// modesChecksumDiagnose for a non-zero syndrome: errors | bit0 << 8 | bit1 << 16, or 0xffffffff
__device__ uint32_t diagnose(const DeviceTables &t, int which, uint32_t syn) {
    const uint32_t mask = t.syn_mask[which];
    const uint32_t *keys = t.syn_keys[which];
    if (!keys) return 0xffffffffu;
    uint32_t h = syndrome_hash(syn, mask);
    for (;;) {
        const uint32_t k = keys[h];
        if (k == syn) return t.syn_vals[which][h];
        if (k == 0) return 0xffffffffu;
        h = (h + 1) & mask;
    }
}

__device__ __forceinline__ uint32_t repair_aa(uint32_t addr, uint32_t info) {  // mode_s.c:266-281
    const uint32_t n = info & 0xff;
    for (uint32_t e = 0; e < n; ++e) {
        const uint32_t bit = (info >> (8 + 8 * e)) & 0xff;
        if (bit >= 8 && bit <= 31) addr ^= 1u << (31 - bit);
    }
    return addr;
}

constexpr uint32_t kDeadItem = 0xffffffffu;
constexpr uint32_t kShortDfMask = (1u << 0) | (1u << 4) | (1u << 5) | (1u << 11);                          // demod_2400.c:274
constexpr uint32_t kLongDfMask = (1u << 16) | (1u << 17) | (1u << 18) | (1u << 20) | (1u << 21) | (1u << 24);  // demod_2400.c:278

// The filter-independent half of scoreModesMessage (mode_s.c:311-409): which branch does this phase
// take, and which 24-bit value would it ask the filter about?  combo << 27 | value, or kDeadItem.
template <int NBYTES>
__device__ uint32_t classify(const uint32_t w[4], const DeviceTables &t, const uint32_t *crcb, uint32_t *adders) {
    const uint32_t df = w[0] >> 27;
    if (NBYTES == 7 && w[0] == 0 && (w[1] >> 8) == 0) return kDeadItem;  // all-zero message, mode_s.c:325
    const uint32_t syn = syndrome_of<NBYTES>(w, crcb);
    const uint32_t aa = w[0] & 0xffffffu;
    if (NBYTES == 7) {
        if (df != 11) return (kComboApShort << 27) | syn;  // DF0/4/5: Address/Parity
        const uint32_t s = syn & 0xffff80u, iid = syn & 0x7fu;
        uint32_t info = 0;
        if (s) {
            info = diagnose(t, 0, s);
            if (info == kDeadItem || (info & 0xff) > 1) return kDeadItem;  // mode_s.c:350-358
        }
        const uint32_t addr = repair_aa(aa, info);
        if (iid == 0) {
            if (s == 0) atomicOr(&adders[addr >> 5], 1u << (addr & 31));  // mode_s.c:717-726
            return ((kComboDf11Iid0 + (info & 0xff)) << 27) | addr;
        }
        return ((kComboDf11IidX + (info & 0xff)) << 27) | addr;
    } else {
        if (df == 16 || df == 24) return (kComboApLong << 27) | syn;
        if (df == 20 || df == 21) return (kComboCommB << 27) | syn;  // exact match or nothing (mode_s.c:391-403)
        uint32_t info = 0;  // DF17/18
        if (syn) {
            info = diagnose(t, 1, syn);
            if (info == kDeadItem) return kDeadItem;
        }
        const uint32_t addr = repair_aa(aa, info);
        if (df == 17 && syn == 0) atomicOr(&adders[addr >> 5], 1u << (addr & 31));
        const uint32_t e = info & 0xff;
        const uint32_t sub = e ? (2 * e - 1 + (addr != aa)) : 0;
        return (((df == 17 ? kComboDf17 : kComboDf18) + sub) << 27) | addr;
    }
}

__device__ __forceinline__ void emit_item(const BatchArgs &a, uint32_t g, uint32_t phase, uint32_t cls) {
    if (cls == kDeadItem) return;
    const uint32_t slot = atomicAdd(&a.counters[kCntItems], 1u);
    if (slot < a.item_cap) a.items[slot] = ItemRec{g, cls | (phase << 24)};
}

// per-thread path for one trial phase (any length); used when an LDS queue is full
__device__ void slow_item(const BatchArgs &a, const uint16_t *d, uint32_t g, int phase, const uint32_t *crcb) {
    uint32_t w[4];
    const int nbytes = slice_phase(d, phase + 4, w);
    if (nbytes == 7)
        emit_item(a, g, phase, classify<7>(w, a.tab, crcb, a.adder_bitmap));
    else if (nbytes == 14)
        emit_item(a, g, phase, classify<14>(w, a.tab, crcb, a.adder_bitmap));
}

// ------------------------------------------------------------------------------------------------
// the scan kernel

constexpr int kScanT = 8192;                      // scan offsets per tile
constexpr int kScanThreads = 512;
constexpr int kScanVecs = (kScanT + kHalo) / 8;   // 16-byte vectors of 8 magnitudes per tile
constexpr int kTileShift = 2;   // tiles start at S index t*T - 2: then their first IQ sample, t*T - 328, is a multiple of 8
constexpr int kEdgeCap = 4096;  // edge-pattern candidates of half a tile (at most 2 of any 3 consecutive offsets can pass)
constexpr int kGateCap = 768;   // gated offsets per tile that go through the dense stages; more take the per-thread path
constexpr int kListCap = 1536;  // short / long items per tile in the dense stages

struct __attribute__((aligned(16))) ScanShared {
    uint16_t m[kScanT + kHalo];
    uint16_t lut[16384];
    uint16_t eq[kEdgeCap];
    uint16_t gq[kGateCap];
    uint16_t lst[2][kListCap];
    uint32_t crcb[256];
    uint32_t cnt[6];  // edge candidates (first half), gated, short items, long items, edge candidates (second half)
};

__device__ __forceinline__ uint32_t uc8_pair(uint32_t x, const uint16_t *lut) {
    // two UC8 samples (I0 Q0 I1 Q1) -> two magnitudes. |v - 127.5| - 0.5 per byte, then the folded
    // 128 x 128 table; the XOR spreads rows over LDS banks (small signals otherwise hit few banks)
    const uint32_t t = (~x >> 7) & 0x01010101u;
    const uint32_t f = (x ^ ((t << 7) - t)) & 0x7f7f7f7fu;
    uint32_t i0 = (f & 0x7f) | ((f >> 1) & 0x3f80);
    uint32_t i1 = ((f >> 16) & 0x7f) | ((f >> 17) & 0x3f80);
    i0 ^= (i0 >> 6) & 0x3e;
    i1 ^= (i1 >> 6) & 0x3e;
    return (uint32_t)lut[i0] | ((uint32_t)lut[i1] << 16);
}

template <int FMT>
__device__ __forceinline__ uint32_t sc16_pair(uint4 v, int which) {
    constexpr float scale = (FMT == kFmtSC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
    const uint32_t a = which ? v.z : v.x, b = which ? v.w : v.y;
    const uint32_t m0 = float_mag((float)(short)(a & 0xffff) * scale, (float)(short)(a >> 16) * scale);
    const uint32_t m1 = float_mag((float)(short)(b & 0xffff) * scale, (float)(short)(b >> 16) * scale);
    return m0 | (m1 << 16);
}

template <int FMT>
__global__ __launch_bounds__(kScanThreads) void k_scan(const BatchArgs a) {
    __shared__ ScanShared sh;
    const uint32_t tid = threadIdx.x, lane = tid & 63;

    if (FMT == kFmtUC8)
        for (uint32_t i = tid; i < 16384; i += kScanThreads) {
            const uint32_t ai = i & 127, aq = i >> 7;
            sh.lut[i ^ ((aq & 31) << 1)] = a.tab.uc8_lut[((128 + aq) << 8) | (128 + ai)];
        }
    if (tid < 256) sh.crcb[tid] = a.tab.crc_bytes[tid];

    const uint64_t ntiles = (a.n + kTileShift + kScanT - 1) / kScanT;
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t s0 = (int64_t)(t * kScanT) - kTileShift;  // S index of m[0]
        __syncthreads();  // previous tile fully consumed; LUT ready
        if (tid < 6) sh.cnt[tid] = 0;

        // ---- convert.c: tile + halo -> magnitudes in LDS
        for (int v = tid; v < kScanVecs; v += kScanThreads) {
            const int64_t s = s0 + 8 * v, q = s - (int64_t)kTrailing;
            uint4 o;
            if ((FMT == kFmtUC8) && a.iq_aligned && q >= 0 && q + 8 <= (int64_t)a.n) {
                const uint4 x = *(const uint4 *)((const uint8_t *)a.iq + 2 * q);
                o = make_uint4(uc8_pair(x.x, sh.lut), uc8_pair(x.y, sh.lut), uc8_pair(x.z, sh.lut), uc8_pair(x.w, sh.lut));
            } else if ((FMT == kFmtSC16 || FMT == kFmtSC16Q11) && a.iq_aligned && q >= 0 && q + 8 <= (int64_t)a.n) {
                const uint4 *src = (const uint4 *)((const uint8_t *)a.iq + 4 * q);
                const uint4 x0 = src[0], x1 = src[1];
                o = make_uint4(sc16_pair<FMT>(x0, 0), sc16_pair<FMT>(x0, 1), sc16_pair<FMT>(x1, 0), sc16_pair<FMT>(x1, 1));
            } else {
                uint32_t e[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) e[k] = stream_mag<FMT>(a, s + k);
                o = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
            }
            ((uint4 *)sh.m)[v] = o;
        }
        __syncthreads();

        for (int h = 0; h < 2; ++h) {
            // ---- demod_2400.c:109-155, edge patterns only, 64 lanes x 8 consecutive offsets per ballot
            {
                const int base = h * (kScanT / 2) + (int)tid * 8;
                uint32_t w[22];
                const uint4 *src = (const uint4 *)&sh.m[base];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const uint4 v = src[q];
                    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (q * 8 + e * 2 < 22) w[q * 8 + e * 2] = d[e] & 0xffff;
                        if (q * 8 + e * 2 + 1 < 22) w[q * 8 + e * 2 + 1] = d[e] >> 16;
                    }
                }
                uint64_t L[21], G[21];  // L[j]: S[base+j] < S[base+j+1] per lane, G[j]: >
#pragma unroll
                for (int j = 0; j < 21; ++j) {
                    L[j] = __ballot(w[j] < w[j + 1]);
                    G[j] = __ballot(w[j] > w[j + 1]);
                }
                uint64_t E[8];
                uint32_t total = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint64_t *l = &L[k], *g = &G[k];
                    // the five peak patterns of :112-151 OR-ed together, gated by the quick check of :109
                    const uint64_t A = g[1] & l[2], B = l[3] & g[4], C = l[8] & g[9], D = g[10] & l[11];
                    const uint64_t p12 = A & g[3] & C & (l[10] | l[11]);
                    const uint64_t p345 = ((A & g[4] & l[8]) | (B & l[9] & (g[1] | g[2]))) & D;
                    E[k] = (p12 | p345) & l[0] & g[12];
                    total += __popcll(E[k]);
                }
                if (total) {
                    uint32_t wbase = 0;
                    if (lane == 0) wbase = atomicAdd(&sh.cnt[4 * h], total);
                    wbase = __builtin_amdgcn_readfirstlane(wbase);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if ((E[k] >> lane) & 1) {
                            const uint32_t pos = wbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(E[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)E[k], 0));
                            if (pos < (uint32_t)kEdgeCap) sh.eq[pos] = (uint16_t)(base + k);
                        }
                        wbase += __popcll(E[k]);
                    }
                }
            }
            __syncthreads();

            // ---- the whole gate (:109-172) on the edge candidates; survivors go to the gated queue
            {
                const uint32_t neq = min(sh.cnt[4 * h], (uint32_t)kEdgeCap);
                for (uint32_t c = tid; c < neq; c += kScanThreads) {
                    const uint32_t j = sh.eq[c];
                    const int64_t g = s0 + j;
                    if (g < 0 || (uint64_t)g >= a.n) continue;
                    if (!preamble_gate(&sh.m[j])) continue;
                    if (a.emit_dead) {
                        const uint32_t slot = atomicAdd(&a.counters[kCntItems], 1u);
                        if (slot < a.item_cap) a.items[slot] = ItemRec{(uint32_t)g, (kComboMarker << 27) | (7u << 24)};
                    }
                    const uint32_t pos = atomicAdd(&sh.cnt[1], 1u);
                    if (pos < (uint32_t)kGateCap) {
                        sh.gq[pos] = (uint16_t)j;
                    } else {
                        atomicAdd(&a.counters[kCntSlowPath], 1u);
                        for (int p = 0; p < 5; ++p) slow_item(a, &sh.m[j + 19], (uint32_t)g, p, sh.crcb);
                    }
                }
            }
            __syncthreads();
        }

        const uint32_t ngated = sh.cnt[1];
        const uint32_t ngq = min(ngated, (uint32_t)kGateCap);
        if (tid == 0 && ngated) atomicAdd(&a.counters[kCntGated], ngated);

        // ---- first five bits (the DF) of every (offset, trial phase): which length class? (:272-285)
        for (uint32_t i = tid; i < 5 * ngq; i += kScanThreads) {
            const uint32_t cand = i / 5, phase = i - 5 * cand;
            const uint32_t j = sh.gq[cand];
            const RotSlicer rs((int)phase + 4);
            const uint32_t df = rs.group(&sh.m[j + 19], 0);
            const uint32_t cls = ((kShortDfMask >> df) & 1) ? 0u : (((kLongDfMask >> df) & 1) ? 1u : 2u);
            if (cls < 2) {
                const uint32_t pos = atomicAdd(&sh.cnt[2 + cls], 1u);
                if (pos < (uint32_t)kListCap) {
                    sh.lst[cls][pos] = (uint16_t)i;
                } else {
                    atomicAdd(&a.counters[kCntSlowPath], 1u);
                    slow_item(a, &sh.m[j + 19], (uint32_t)(s0 + j), (int)phase, sh.crcb);
                }
            }
        }
        __syncthreads();
        const uint32_t ns = min(sh.cnt[2], (uint32_t)kListCap), nl = min(sh.cnt[3], (uint32_t)kListCap);

        // ---- 56-bit frames: 12 groups of 5 bits
        {
            for (uint32_t c = tid; c < ns; c += kScanThreads) {
                const uint32_t i = sh.lst[0][c], cand = i / 5, phase = i - 5 * cand;
                const uint32_t j = sh.gq[cand];
                const RotSlicer rs((int)phase + 4);
                const uint16_t *d = &sh.m[j + 19];
                uint64_t acc = 0;
#pragma unroll
                for (int G = 0; G < 12; ++G) acc = (acc << 5) | rs.group(d, G);
                acc <<= 4;  // message bit 0 -> bit 63; bits 7..4 hold the four extra bits sliced
                uint32_t w[4] = {(uint32_t)(acc >> 32), (uint32_t)acc & 0xffffff00u, 0, 0};
                emit_item(a, (uint32_t)(s0 + j), phase, classify<7>(w, a.tab, sh.crcb, a.adder_bitmap));
            }
        }
        // ---- 112-bit frames: 23 groups of 5 bits
        {
            for (uint32_t c = tid; c < nl; c += kScanThreads) {
                const uint32_t i = sh.lst[1][c], cand = i / 5, phase = i - 5 * cand;
                const uint32_t j = sh.gq[cand];
                const RotSlicer rs((int)phase + 4);
                const uint16_t *d = &sh.m[j + 19];
                uint64_t hi = 0, lo = 0;  // 128-bit shift register
#pragma unroll
                for (int G = 0; G < 23; ++G) {
                    hi = (hi << 5) | (lo >> 59);
                    lo = (lo << 5) | rs.group(d, G);
                }
                hi = (hi << 13) | (lo >> 51);  // message bit 0 -> bit 127; bits 15..13 hold the three extra bits
                lo <<= 13;
                uint32_t w[4] = {(uint32_t)(hi >> 32), (uint32_t)hi, (uint32_t)(lo >> 32), (uint32_t)lo & 0xffff0000u};
                emit_item(a, (uint32_t)(s0 + j), phase, classify<14>(w, a.tab, sh.crcb, a.adder_bitmap));
            }
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_prune(const BatchArgs a) {
    uint32_t n = a.counters[kCntItems];
    if (n > a.item_cap) n = a.item_cap;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const ItemRec rec = a.items[i];
        const uint32_t combo = item_combo(rec.info), v = item_value(rec.info);
        bool keep = a.keep_all != 0 || combo == kComboMarker || !combo_needs_member(combo);
        if (!keep) keep = (a.adder_bitmap[v >> 5] >> (v & 31)) & 1u;
        if (keep) {
            const uint32_t slot = atomicAdd(&a.counters[kCntSurvItems], 1u);
            if (slot < a.item_cap) {
                a.skeys[0][slot] = ((uint64_t)rec.g << 3) | item_phase(rec.info);
                a.sinfo[0][slot] = rec.info;
            }
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_keys_to_g(const uint64_t *keys, uint32_t *g, uint32_t n) {
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) g[i] = (uint32_t)(keys[i] >> 3);
}

// one wavefront per distinct surviving offset
template <int FMT>
__global__ __launch_bounds__(64) void k_finalize(const BatchArgs a) {
    __shared__ uint16_t m[kHalo];
    uint32_t nsurv = a.counters[kCntSurvivors];
    if (nsurv > a.surv_cap) nsurv = a.surv_cap;
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = blockIdx.x; i < nsurv; i += gridDim.x) {
        const uint32_t g = a.surv_g[i];
        for (uint32_t k = lane; k < (uint32_t)kHalo; k += 64) m[k] = (uint16_t)stream_mag<FMT>(a, (int64_t)g + k);
        __syncthreads();
        Survivor *out = &a.survivors[i];
        if (lane < 5) {
            uint32_t w[4];
            const int nbytes = slice_phase(&m[19], (int)lane + 4, w);
            out->vbytes[lane] = (uint8_t)nbytes;
            for (int b = 0; b < 14; ++b) out->msg[lane][b] = (uint8_t)msg_byte(w, b);
        }
        // demod_2400.c:345-351 for both possible frame lengths
        uint64_t p134 = 0, p268 = 0;
        for (uint32_t k = lane; k < 268; k += 64) {
            const uint64_t v = (uint64_t)m[19 + k] * m[19 + k];
            p268 += v;
            if (k < 134) p134 += v;
        }
        for (int off = 32; off > 0; off >>= 1) {
            p134 += __shfl_down(p134, off, 64);
            p268 += __shfl_down(p268, off, 64);
        }
        if (lane == 0) {
            out->g = g;
            out->pad = 0;
            out->pow134 = p134;
            out->pow268 = p268;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// Mode A/C

constexpr int kTile = 4096;

// convert.c:71-83: per 131072-block sums over the block's NEW samples (integer converters only)
template <int FMT>
__global__ __launch_bounds__(kThreads) void k_block_sums(const BatchArgs a) {
    const uint64_t base = (uint64_t)blockIdx.x * kTile;  // iq index
    uint64_t sl = 0, sp = 0;
    for (uint32_t i = threadIdx.x; i < (uint32_t)kTile; i += kThreads) {
        const uint64_t idx = base + i;
        if (idx < a.n) {
            const uint32_t v = mag_of<FMT>(a.iq, idx, a.tab.uc8_lut);
            sl += v;
            sp += (uint64_t)v * v;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        sl += __shfl_down(sl, off, 64);
        sp += __shfl_down(sp, off, 64);
    }
    if ((threadIdx.x & 63) == 0 && (sl | sp)) {
        const uint64_t blk = base / kBlock;
        atomicAdd((unsigned long long *)&a.block_sums[2 * blk], (unsigned long long)sl);
        atomicAdd((unsigned long long *)&a.block_sums[2 * blk + 1], (unsigned long long)sp);
    }
}

constexpr int kAcBack = 1;    // m[f1_sample - 1]
constexpr int kAcAhead = 72;  // bit 19 is sliced at f1_sample + 67 and looks 2 further (demod_2400.c:587-608)

// demod_2400.c:476-622 with one thread per f1_sample; the greedy skip (:659) is applied on the host.
template <int FMT>
__global__ __launch_bounds__(kThreads) void k_modeac(const BatchArgs a) {
    __shared__ uint16_t mt[kAcBack + kTile + kAcAhead];
    const uint64_t t0 = (uint64_t)blockIdx.x * kTile;
    const uint16_t *m = mt + kAcBack;  // m[i] = S[t0 + i]
    for (uint32_t i = threadIdx.x; i < kAcBack + kTile + kAcAhead; i += kThreads)
        mt[i] = (uint16_t)stream_mag<FMT>(a, (int64_t)(t0 + i) - kAcBack);
    __syncthreads();

    // the tile lies inside one block (kBlock % kTile == 0)
    const uint64_t blk = t0 / kBlock;
    const uint64_t blk_start = blk * kBlock;
    const uint64_t blk_len = (a.n - blk_start < (uint64_t)kBlock) ? (a.n - blk_start) : (uint64_t)kBlock;
    // convert.c:104-110 then demod_2400.c:484-485
    const double mean_level = (double)a.block_sums[2 * blk] / 65536.0 / (double)(unsigned)blk_len;
    const double mean_power = (double)a.block_sums[2 * blk + 1] / 65535.0 / 65535.0 / (double)(unsigned)blk_len;
    const double noise_stddev = sqrt(mean_power - mean_level * mean_level);
    const unsigned noise_level = (unsigned)((mean_power + noise_stddev) * 65535 + 0.5);

    for (uint32_t j = threadIdx.x; j < (uint32_t)kTile; j += kThreads) {
        const uint64_t g = t0 + j;
        if (g >= a.n) break;
        const unsigned f1_sample = (unsigned)(g - blk_start);
        if (f1_sample == 0) continue;  // the loop starts at 1 (demod_2400.c:487)

        if (!(m[j - 1] < m[j])) continue;
        if (m[j + 2] > m[j] || m[j + 2] > m[j + 1]) continue;
        const unsigned f1_level = ((unsigned)m[j] + m[j + 1]) / 2;
        if (noise_level * 2 > f1_level) continue;

        const float f1a_power = (float)m[j] * (float)m[j];
        const float f1b_power = (float)m[j + 1] * (float)m[j + 1];
        const float fraction = f1b_power / (f1a_power + f1b_power);
        const float clockf = 25.0f * ((float)f1_sample + fraction * fraction);
        const unsigned f1_clock = (unsigned)((double)clockf + 0.5);

        const unsigned f2_clock = f1_clock + (87 * 14);
        const unsigned f2_sample = f2_clock / 25;
        const int k2 = (int)j + (int)(f2_sample - f1_sample);
        if (!(m[k2 - 1] < m[k2])) continue;
        if (m[k2 + 2] > m[k2] || m[k2 + 2] > m[k2 + 1]) continue;
        const unsigned f2_level = ((unsigned)m[k2] + m[k2 + 1]) / 2;
        if (noise_level * 2 > f2_level) continue;

        const unsigned f1f2_level = f1_level > f2_level ? f1_level : f2_level;
        const float midpoint = sqrtf((float)(noise_level * f1f2_level));
        const unsigned signal_threshold = (unsigned)((double)midpoint * 1.41421356237309504880 + 0.5);
        const unsigned noise_threshold = (unsigned)((double)midpoint / 1.41421356237309504880 + 0.5);

        unsigned uncertain_bits = 0, noisy_bits = 0, bits = 0, clock = f1_clock;
        for (int bit = 0; bit < 20; ++bit, clock += 87) {
            const int s = (int)j + (int)(clock / 25 - f1_sample);
            bits <<= 1;
            noisy_bits <<= 1;
            uncertain_bits <<= 1;
            if (m[s + 2] >= signal_threshold) noisy_bits |= 1;
            if (m[s] >= signal_threshold || m[s + 1] >= signal_threshold)
                bits |= 1;
            else if (m[s] > noise_threshold && m[s + 1] > noise_threshold)
                uncertain_bits |= 1;
        }
        if ((bits & 0x80020) != 0x80020) continue;
        if ((bits & 0x0101B) != 0) continue;
        if (noisy_bits || uncertain_bits) continue;

        const unsigned modeac =  // demod_2400.c:626-639
            ((bits & 0x40000) ? 0x0010 : 0) | ((bits & 0x20000) ? 0x1000 : 0) | ((bits & 0x10000) ? 0x0020 : 0) |
            ((bits & 0x08000) ? 0x2000 : 0) | ((bits & 0x04000) ? 0x0040 : 0) | ((bits & 0x02000) ? 0x4000 : 0) |
            ((bits & 0x00800) ? 0x0100 : 0) | ((bits & 0x00400) ? 0x0001 : 0) | ((bits & 0x00200) ? 0x0200 : 0) |
            ((bits & 0x00100) ? 0x0002 : 0) | ((bits & 0x00080) ? 0x0400 : 0) | ((bits & 0x00040) ? 0x0004 : 0) |
            ((bits & 0x00004) ? 0x0080 : 0);

        const uint32_t slot = atomicAdd(&a.counters[kCntAc], 1u);
        if (slot < a.ac_cap) {
            AcCand c;
            c.g = (uint32_t)g;
            c.f2_clock = f2_clock;
            c.modeac = modeac;
            a.ac[slot] = c;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// stand-alone converter: magnitudes to HBM (+ exact integer sums when asked)

template <int FMT>
__global__ __launch_bounds__(kThreads) void k_convert(const void *iq, uint16_t *mag, uint64_t n, const uint16_t *lut, uint64_t *sums) {
    uint64_t sl = 0, sp = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kThreads) {
        const uint32_t v = mag_of<FMT>(iq, i, lut);
        mag[i] = (uint16_t)v;
        sl += v;
        sp += (uint64_t)v * v;
    }
    if (sums) {
        for (int off = 32; off > 0; off >>= 1) {
            sl += __shfl_down(sl, off, 64);
            sp += __shfl_down(sp, off, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd((unsigned long long *)&sums[0], (unsigned long long)sl);
            atomicAdd((unsigned long long *)&sums[1], (unsigned long long)sp);
        }
    }
}

template <typename F>
hipError_t by_format(int fmt, F &&f) {
    switch (fmt) {
        case kFmtUC8: return f(std::integral_constant<int, kFmtUC8>());
        case kFmtSC16: return f(std::integral_constant<int, kFmtSC16>());
        case kFmtSC16Q11: return f(std::integral_constant<int, kFmtSC16Q11>());
        case kFmtMag16: return f(std::integral_constant<int, kFmtMag16>());
        default: return hipErrorInvalidValue;
    }
}

inline unsigned tiles_of(uint64_t n) { return (unsigned)((n + kTile - 1) / kTile); }

}  // namespace

int scan_grid_size(int device) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return 512;
    return 2 * p.multiProcessorCount;  // two 512-thread workgroups (~64 KB of LDS each) per CU
}

hipError_t launch_scan(const BatchArgs &a, int grid, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    const uint64_t ntiles = (a.n + kTileShift + kScanT - 1) / kScanT;
    if ((uint64_t)grid > ntiles) grid = (int)ntiles;
    return by_format(a.fmt, [&](auto F) {
        hipLaunchKernelGGL(k_scan<decltype(F)::value>, dim3(grid), dim3(kScanThreads), 0, s, a);
        return hipGetLastError();
    });
}

hipError_t launch_prune(const BatchArgs &a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_prune, dim3(1024), dim3(kThreads), 0, s, a);
    return hipGetLastError();
}

size_t sort_temp_bytes(uint32_t max_items) {
    size_t sort_bytes = 0, uniq_bytes = 0;
    hipcub::DoubleBuffer<uint64_t> k(nullptr, nullptr);
    hipcub::DoubleBuffer<uint32_t> v(nullptr, nullptr);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, k, v, (int)max_items, 0, 35);
    (void)hipcub::DeviceSelect::Unique(nullptr, uniq_bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (int)max_items);
    return (sort_bytes > uniq_bytes ? sort_bytes : uniq_bytes) + 256;
}

hipError_t launch_sort_unique_finalize(const BatchArgs &a, uint32_t n_items, void *temp, size_t temp_bytes, int *sorted_buf, hipStream_t s) {
    *sorted_buf = 0;
    if (n_items == 0) return hipMemsetAsync(&a.counters[kCntSurvivors], 0, sizeof(uint32_t), s);
    hipcub::DoubleBuffer<uint64_t> k(a.skeys[0], a.skeys[1]);
    hipcub::DoubleBuffer<uint32_t> v(a.sinfo[0], a.sinfo[1]);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, k, v, (int)n_items, 0, 35, s);
    if (e != hipSuccess) return e;
    *sorted_buf = (k.Current() == a.skeys[1]) ? 1 : 0;
    // distinct offsets, ascending: the other key buffer is free now and serves as scratch for g
    uint32_t *gtmp = (uint32_t *)a.skeys[*sorted_buf ^ 1];
    hipLaunchKernelGGL(k_keys_to_g, dim3(512), dim3(kThreads), 0, s, k.Current(), gtmp, n_items);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = hipcub::DeviceSelect::Unique(temp, temp_bytes, gtmp, a.surv_g, &a.counters[kCntSurvivors], (int)n_items, s);
    if (e != hipSuccess) return e;
    return by_format(a.fmt, [&](auto F) {
        hipLaunchKernelGGL(k_finalize<decltype(F)::value>, dim3(8192), dim3(64), 0, s, a);
        return hipGetLastError();
    });
}

hipError_t launch_modeac(const BatchArgs &a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    return by_format(a.fmt, [&](auto F) {
        constexpr int fmt = decltype(F)::value;
        hipLaunchKernelGGL(k_block_sums<fmt>, dim3(tiles_of(a.n)), dim3(kThreads), 0, s, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_modeac<fmt>, dim3(tiles_of(a.n)), dim3(kThreads), 0, s, a);
        return hipGetLastError();
    });
}

hipError_t launch_convert(int fmt, const void *iq, uint16_t *mag, uint64_t n, const uint16_t *lut, uint64_t *sums, hipStream_t s) {
    if (n == 0) return hipSuccess;
    return by_format(fmt, [&](auto F) {
        uint64_t blocks = (n + kThreads - 1) / kThreads;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_convert<decltype(F)::value>, dim3((unsigned)blocks), dim3(kThreads), 0, s, iq, mag, n, lut, sums);
        return hipGetLastError();
    });
}

}  // namespace rsb
