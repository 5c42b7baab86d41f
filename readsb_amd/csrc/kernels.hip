// kernels.hip -- gfx950 kernels of the receive path.
//
//   k_scan      persistent grid, one 1024-thread workgroup per CU; its 16 wavefronts work alone (no workgroup barrier after the
//               table set-up) on strips of 2768 scan offsets, handed out in chunks of consecutive strips from a two-level
//               counter: convert.c (IQ -> magnitude; UC8 through a 'radial' table in LDS indexed by one v_dot4, magnitudes
//               live in LDS only) -> demod_2400.c:109-155 as bit-sliced edge patterns (4 offsets x 6 passes per lane and
//               instruction) -> :158-172 on the ~3 % that survive -> :179-286 five trial phases, first the DF field, then
//               dense 56-/112-bit slicing, one lane per frame (v_dot2 on biased sample pairs) -> crc.c:67-82 / :383-406 by
//               linearity -> the filter-independent half of mode_s.c:311-409.  Emits one 8-byte item per trial phase that can
//               still score >= 0 under some filter state, sorted by (offset, phase) within a chunk, and marks every address the
//               filter could ever learn (mode_s.c:717-726) in a 2^24-bit set.  k_scan_many: the channels of rsb_submit_many.
//   k_prune_*   the ordered prune: drops items that need a filter member which is not in that superset (safe: such a phase
//               scores < 0 whatever happened before it; demod_2400.c:290,304-310) and packs the chunks' runs in stream order
//               (summary of the adder bitmap in LDS -> keep masks -> chunk counts -> offsets -> write): no sort.
//   k_prune + rocPRIM radix sort   the fallback for a batch in which a scan wavefront overflowed an LDS queue.
//   k_replay_*  the sequential rules on the device (demod_2400.c:289-374): per-block greedy walk, the ICAO filter as a
//               fixed point over a "first added at" table, k_replay_emit writes the rsb_message records.
//   k_finalize  (host-replay path) re-slices the kept offsets into full records (5 x 14 message bytes + the signal
//               power sums of demod_2400.c:345-351).
//   k_modeac    one workgroup per 131072-sample block: convert.c:71-83 block sums, then demod_2400.c:476-622 in three
//               compacted stages; k_block_means_float / k_block_means_group: the float converters' sequential sums;
//               k_merge_modeac: Mode A/C messages into the Mode S list.
//   k_convert   the stand-alone iq_convert_fn (magnitudes written to HBM) for the drop-in converter;
//   k_dc_*, k_convert_dc   the --dcfilter converters as a chunked linear-recurrence scan.
//   k_wire_*    Beast / raw encoders.
//
// Compiled with -ffp-contract=off: every float/double product and sum below rounds on its own, as
// in the reference's x86-64 build.
#include <cstdlib>

#include "kernels.hpp"
#include "wire.hpp"

#ifndef RSB_SLICE_PREFETCH
#define RSB_SLICE_PREFETCH 1
#endif
#ifndef RSB_RANK_CHEAP
#define RSB_RANK_CHEAP 1
#endif
#ifdef RSB_STAGE_CLOCKS
#define DBG_DECL uint32_t dbg_clk[5] = {0, 0, 0, 0, 0}, dbg_cnt[7] = {0, 0, 0, 0, 0, 0, 0}; uint64_t dbg_t = __builtin_readcyclecounter();
#define DBG_STAGE(k) { const uint64_t t_ = __builtin_readcyclecounter(); dbg_clk[k] += (uint32_t)((t_ - dbg_t) >> 4); dbg_t = t_; }
#define DBG_COUNT(k, n) dbg_cnt[k] += (n);
#define DBG_FLUSH if (lane == 0) { for (int k_ = 0; k_ < 5; ++k_) atomicAdd(&a.counters[kDbgClkConvert + k_], dbg_clk[k_]); for (int k_ = 0; k_ < 7; ++k_) atomicAdd(&a.counters[kDbgStrips + k_], dbg_cnt[k_]); }
#else
#define DBG_DECL
#define DBG_STAGE(k)
#define DBG_COUNT(k, n)
#define DBG_FLUSH
#endif
#ifndef RSB_RADIAL_DEPTH
#define RSB_RADIAL_DEPTH 1
#endif
#ifndef RSB_SLICE_AHEAD
#define RSB_SLICE_AHEAD 1
#endif
#ifndef RSB_LATE_PASSES
#define RSB_LATE_PASSES 0
#endif
#ifndef RSB_PREFETCH_EARLY
#define RSB_PREFETCH_EARLY 1
#endif

#include <rocprim/rocprim.hpp>

#include <cmath>
#include <type_traits>

namespace rsb {
namespace {

constexpr int kHalo = 304;  // look-ahead past the last offset of a tile: 290 needed, 297 used
constexpr int kThreads = 256;

// ------------------------------------------------------------------------------------------------
// sample -> magnitude

// sqrtf, correctly rounded, without the compiler's provisions for arguments this path never sees (it scales arguments below
// 2^-96 up and back and passes infinities through: 7 of its 16 instructions).  v_sqrt_f32 is within one ulp; the two
// residuals of its neighbours decide the rounding (the compiler's own sequence for the normal range).  Arguments here are 0
// or in [2^-30, 1] -- or, behind the DC filter, possibly tiny, where any result below 2^-17 gives magnitude 0 like the exact one.
__device__ __forceinline__ float sqrt_rn(float x) {
    const float r = __builtin_amdgcn_sqrtf(x);
    const float r_dn = __uint_as_float(__float_as_uint(r) - 1u), r_up = __uint_as_float(__float_as_uint(r) + 1u);
    const float e_dn = __builtin_fmaf(-r_dn, r, x), e_up = __builtin_fmaf(-r_up, r, x);
    float y = (e_dn <= 0.0f) ? r_dn : r;  // (r = 0: r_dn is a NaN, the comparison fails, 0 stays)
    y = (e_up > 0.0f) ? r_up : y;
    return y;
}

__device__ __forceinline__ uint32_t float_mag(float fI, float fQ) {
    // convert.c:236-243 (and :353-360): separately rounded products and sum, clamp, IEEE sqrt
    float magsq = fI * fI + fQ * fQ;
    if (magsq > 1.0f) magsq = 1.0f;
    const float mag = sqrt_rn(magsq);
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

// ---- the DC block (convert.c:113-162) inside the kernels that read raw IQ.  A pass in front of the scan (launch_dc_states) leaves
// the filter's state before every 8-sample chunk -- evaluated as a chunked scan in double, as the converter kernel does -- and the
// chunk's samples then go through the reference's own float recurrence from that state.  Every kernel that needs a magnitude
// (scan, general strips, candidate records, message records, carry) computes it this way, so all of them see the same values.
__device__ __forceinline__ float2 dc_state_before_chunk(const BatchArgs &a, uint64_t chunk) {
    const float2 l = a.dc_local[chunk];
    const double2 z0 = a.dc_tile_start[chunk >> 9];
    const double p = a.dc_pow[chunk & 511u];
    return make_float2((float)((double)l.x + z0.x * p), (float)((double)l.y + z0.y * p));
}
// convert.c:131-132, ((float)v - 127.5f) / 127.5f, without the division: the product with the rounded reciprocal and one
// correction step give the correctly rounded quotient for all 256 values of v (checked exhaustively: tests/test_uc8_scaling.py)
__device__ __forceinline__ float uc8_scaled(uint32_t v) {
    constexpr float c = 1.0f / 127.5f;
    const float n = (float)v - 127.5f;
    float q = n * c;
    const float r = __builtin_fmaf(-q, 127.5f, n);
    return __builtin_fmaf(r, c, q);
}
// one sample (SC16 / SC16Q11: a dword, UC8: its two bytes) through the filter (state z) -> its magnitude
template <int FMT>
__device__ __forceinline__ uint32_t dc_sample(uint32_t w, float dc_a, float dc_b, float2 &z) {
    float xI, xQ;
    if constexpr (FMT == kFmtUC8) {
        xI = uc8_scaled(w & 0xffu);
        xQ = uc8_scaled((w >> 8) & 0xffu);
    } else {
        constexpr float scale = (FMT == kFmtSC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f);  // exact: a power of two
        xI = (float)(short)(w & 0xffffu) * scale;
        xQ = (float)(short)(w >> 16) * scale;
    }
    z.x = xI * dc_a + z.x * dc_b;  // convert.c:137-140
    z.y = xQ * dc_a + z.y * dc_b;
    xI -= z.x;
    xQ -= z.y;
    return float_mag(xI, xQ);
}
template <int FMT>
__device__ __forceinline__ uint4 dc_chunk_from(const BatchArgs &a, const uint4 x0, const uint4 x1, float2 z);
template <int FMT>
__device__ __forceinline__ uint4 dc_chunk(const BatchArgs &a, const uint4 x0, const uint4 x1, uint64_t chunk) {  // -> 8 biased magnitudes
    return dc_chunk_from<FMT>(a, x0, x1, dc_state_before_chunk(a, chunk));
}
template <int FMT>
__device__ __forceinline__ uint4 dc_chunk_from(const BatchArgs &a, const uint4 x0, const uint4 x1, float2 z) {
    uint32_t W[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    if constexpr (FMT == kFmtUC8) {  // 8 samples in x0 alone
        const uint32_t X[4] = {x0.x, x0.y, x0.z, x0.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) W[k] = (X[k >> 1] >> (16 * (k & 1))) & 0xffffu;
    }
    uint32_t m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = dc_sample<FMT>(W[k], a.dc_a, a.dc_b, z) ^ 0x8000u;  // (the LDS arrays' bias, kBias below)
    return make_uint4(m[0] | (m[1] << 16), m[2] | (m[3] << 16), m[4] | (m[5] << 16), m[6] | (m[7] << 16));
}

// S[s] of the batch stream S = carry(326) ++ iq(n); zero outside.  The *_run form keeps the DC filter's state between calls for
// ascending s (zpos = the iq index the state z is valid for; -1: none): one step per sample instead of a walk from the chunk's start.
// (MAYBE_DC = false: for kernels that are never launched on a batch with the filter -- the scan has a variant of its own for those,
// and the plain one keeps its register allocation)
template <int FMT, bool MAYBE_DC = true>
__device__ __forceinline__ uint32_t stream_mag_run(const BatchArgs &a, int64_t s, float2 &z, int64_t &zpos) {
    if (s < 0) return 0;
    if (s < (int64_t)kTrailing) {
        if (!a.carry_valid) return 0u;
        return a.carry_is_mag ? (uint32_t)((const uint16_t *)a.carry)[s] : mag_of<FMT>(a.carry, (uint64_t)s, a.tab.uc8_lut);
    }
    s -= kTrailing;
    if ((uint64_t)s >= a.n) return 0;
    if constexpr (FMT != kFmtMag16 && MAYBE_DC) {
        if (a.dc_local) {
            const auto raw = [&](int64_t j) -> uint32_t { return FMT == kFmtUC8 ? (uint32_t)((const uint16_t *)a.iq)[j] : ((const uint32_t *)a.iq)[j]; };
            if (s != zpos || (s & 7) == 0) {
                z = dc_state_before_chunk(a, (uint64_t)s >> 3);
                for (int64_t j = s & ~(int64_t)7; j < s; ++j) (void)dc_sample<FMT>(raw(j), a.dc_a, a.dc_b, z);
            }
            zpos = s + 1;
            return dc_sample<FMT>(raw(s), a.dc_a, a.dc_b, z);
        }
    }
    return mag_of<FMT>(a.iq, (uint64_t)s, a.tab.uc8_lut);
}
template <int FMT>
__device__ __forceinline__ uint32_t stream_mag(const BatchArgs &a, int64_t s) {
    float2 z = make_float2(0.0f, 0.0f);
    int64_t zpos = -1;
    return stream_mag_run<FMT>(a, s, z, zpos);
}

// LDS magnitude arrays hold "biased" magnitudes, mag ^ 0x8000, i.e. mag - 32768 as int16.  The taps of
// every correlator sum to zero (5-3-2, 4-1-3, 3+1-4, 2+3-5, 1+5-5-1), so the correlators give exactly the
// same values on biased samples, and the signed 16-bit dot product v_dot2_i32_i16 applies to sample
// pairs as they lie in LDS.  Comparisons between samples become signed comparisons.
constexpr uint32_t kBias = 0x8000u, kBias2 = 0x80008000u;
__device__ __forceinline__ int biased(const uint16_t *m, int i) { return (int)(int16_t)m[i]; }

// ------------------------------------------------------------------------------------------------
// demod_2400.c:109-172, the whole gate on one offset

__device__ __forceinline__ bool preamble_gate(const uint16_t *m, int e) {
    // all 19 samples first (independent LDS reads), then straight-line selects: the first matching
    // pattern of :112-151 decides high / signal / noise
    int p[19];
    {
        // 19 samples from 11 aligned dwords (the array base is 4-byte aligned; e = sample index)
        const uint32_t *q = (const uint32_t *)m + (e >> 1);
        const uint32_t sh = (uint32_t)(e & 1) << 4;
        uint32_t d[11];
#pragma unroll
        for (int i = 0; i < 11; ++i) d[i] = q[i];
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const uint32_t x = __builtin_amdgcn_alignbit(d[i + 1], d[i], sh);
            p[2 * i] = (int)(int16_t)(x & 0xffffu);  // biased samples: signed
            if (2 * i + 1 < 19) p[2 * i + 1] = (int)x >> 16;
        }
    }
    const bool l0 = p[0] < p[1], g1 = p[1] > p[2], l2 = p[2] < p[3], g2 = p[2] > p[3], g3 = p[3] > p[4], l3 = p[3] < p[4],
               g4 = p[4] > p[5], l8 = p[8] < p[9], g9 = p[9] > p[10], l9 = p[9] < p[10], l10 = p[10] < p[11], g10 = p[10] > p[11],
               l11 = p[11] < p[12], g12 = p[12] > p[13];
    const bool c1 = g1 & l2 & g3 & l8 & g9 & l10;
    const bool c2 = g1 & l2 & g3 & l8 & g9 & l11;
    const bool c3 = g1 & l2 & g4 & l8 & g10 & l11;
    const bool c4 = g1 & l3 & g4 & l9 & g10 & l11;
    const bool c5 = g2 & l3 & g4 & l9 & g10 & l11;
    // sums of k magnitudes = sums of k biased samples + k * 32768
    constexpr int B = (int)kBias;
    const int h1 = p[1] + p[3] + p[9] + p[11] + p[12] + 5 * B, s1 = p[1] + p[3] + p[9] + 3 * B, n1 = p[5] + p[6] + p[7] + 3 * B;
    const int h2 = p[1] + p[3] + p[9] + p[12] + 4 * B, s2 = h2, n2 = n1 + p[8] + B;
    const int h3 = p[1] + p[3] + p[4] + p[9] + p[10] + p[12] + 6 * B, s3 = p[1] + p[12] + 2 * B, n3 = p[6] + p[7] + 2 * B;
    const int h4 = p[1] + p[4] + p[10] + p[12] + 4 * B, s4 = h4, n4 = n2;
    const int h5 = p[1] + p[2] + p[4] + p[10] + p[12] + 5 * B, s5 = p[4] + p[10] + p[12] + 3 * B, n5 = p[6] + p[7] + p[8] + 3 * B;
    const int hs = c1 ? h1 : c2 ? h2 : c3 ? h3 : c4 ? h4 : h5;
    const int sig = c1 ? s1 : c2 ? s2 : c3 ? s3 : c4 ? s4 : s5;
    const int noise = c1 ? n1 : c2 ? n2 : c3 ? n3 : c4 ? n4 : n5;
    const int high = hs / 4;
    const int quiet = max(max(max(p[5], p[6]), max(p[7], p[8])), max(max(max(p[14], p[15]), max(p[16], p[17])), p[18])) + B;
    return l0 & g12 & (c1 | c2 | c3 | c4 | c5) & (sig * 2 >= 3 * noise) & (quiet < high);
}

// ------------------------------------------------------------------------------------------------
// demod_2400.c:48-66 and :179-286 in closed form: bit k of trial phase tp is the sign of correlator
// (tp + 12k) % 5 at sample 19 + (tp + 12k) / 5 after the preamble start.

template <int PH>
__device__ __forceinline__ int corr(const uint16_t *m) {
    const int a = biased(m, 0), b = biased(m, 1), c = biased(m, 2);
    if constexpr (PH == 0) return 5 * a - 3 * b - 2 * c;
    if constexpr (PH == 1) return 4 * a - b - 3 * c;
    if constexpr (PH == 2) return 3 * a + b - 4 * c;
    if constexpr (PH == 3) return 2 * a + 3 * b - 5 * c;
    return a + 5 * b - 5 * c - biased(m, 3);
}

typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int sdot2(uint32_t a, uint32_t b, int c) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b), c, false);
}
constexpr uint32_t pk16(int lo, int hi) { return (uint32_t)(uint16_t)(int16_t)lo | ((uint32_t)(uint16_t)(int16_t)hi << 16); }
using lds_dword = const __attribute__((address_space(3))) uint32_t;

// MINUS correlator PH at the sample whose LDS byte address is A (even): two or three aligned dwords
// (the LDS pipe charges per instruction, and unaligned ds_reads run ~20x slower), v_alignbyte to the
// sample pairs, two dot products.  Negated so that "bit is 1" (correlator > 0) is the result's sign bit.
template <int PH>
__device__ __forceinline__ int neg_corr_lds(uint32_t A) {
    lds_dword *q = (lds_dword *)(uintptr_t)(A & ~3u);
    const uint32_t d0 = q[0], d1 = q[1];
    const uint32_t x = __builtin_amdgcn_alignbyte(d1, d0, A);  // samples e, e+1
    if constexpr (PH < 4) {
        const uint32_t t = __builtin_amdgcn_alignbyte(d1, d1, A);  // low half: sample e+2
        constexpr int c[4][3] = {{5, -3, -2}, {4, -1, -3}, {3, 1, -4}, {2, 3, -5}};
        return sdot2(x, pk16(-c[PH][0], -c[PH][1]), sdot2(t, pk16(-c[PH][2], 0), 0));
    } else {
        const uint32_t y = __builtin_amdgcn_alignbyte(q[2], d1, A);  // samples e+2, e+3
        return sdot2(x, pk16(-1, -5), sdot2(y, pk16(5, 1), 0));
    }
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
struct RotTables {
    uint32_t off[5], bit[5];  // per trial phase 4..8: five nibbles / five 5-bit masks, one per correlator
};
constexpr RotTables make_rot_tables() {
    RotTables t{};
    for (int tp = 4; tp <= 8; ++tp) {
        uint32_t o = 0, b = 0;
        for (int q = 0; q < 5; ++q) {
            const int j = ((q + 10 - tp) * 3) % 5;  // bit j of the group uses correlator (tp + 2j) % 5 == q
            o |= (uint32_t)((tp + 12 * j) / 5) << (4 * q);
            b |= (16u >> j) << (5 * q);
        }
        t.off[tp - 4] = o;
        t.bit[tp - 4] = b;
    }
    return t;
}
constexpr RotTables kRot = make_rot_tables();

struct RotSlicer {
    uint32_t off2[5];  // byte offset (relative to the group base) of the sample at which correlator q is evaluated
    uint32_t bit[5];   // the bit of the 5-bit group that correlator q decides (MSB = earliest bit)
    __device__ __forceinline__ explicit RotSlicer(uint32_t phase /* try_phase - 4 */) {
        const uint32_t o = phase == 0 ? kRot.off[0] : phase == 1 ? kRot.off[1] : phase == 2 ? kRot.off[2] : phase == 3 ? kRot.off[3] : kRot.off[4];
        const uint32_t b = phase == 0 ? kRot.bit[0] : phase == 1 ? kRot.bit[1] : phase == 2 ? kRot.bit[2] : phase == 3 ? kRot.bit[3] : kRot.bit[4];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            off2[q] = ((o >> (4 * q)) & 15u) * 2;
            bit[q] = (b >> (5 * q)) & 31u;
        }
    }
    // five bits starting at bit 5*G of the message; m = biased magnitudes in LDS, e = index of S[g + 19]
    __device__ __forceinline__ uint32_t group(const uint16_t *m, int e, int G) const {
        const uint32_t A = (uint32_t)(uintptr_t)(m + e) + 24u * (uint32_t)G;
        uint32_t r = (uint32_t)(neg_corr_lds<0>(A + off2[0]) >> 31) & bit[0];
        r |= (uint32_t)(neg_corr_lds<1>(A + off2[1]) >> 31) & bit[1];
        r |= (uint32_t)(neg_corr_lds<2>(A + off2[2]) >> 31) & bit[2];
        r |= (uint32_t)(neg_corr_lds<3>(A + off2[3]) >> 31) & bit[3];
        r |= (uint32_t)(neg_corr_lds<4>(A + off2[4]) >> 31) & bit[4];
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

__device__ __forceinline__ void mark_adder(uint32_t *adders, uint32_t addr) {  // mode_s.c:717-726
    const uint32_t bit = 1u << (addr & 31);
    if (!(adders[addr >> 5] & bit)) atomicOr(&adders[addr >> 5], bit);
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
            if (s == 0) mark_adder(adders, addr);
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
        if (df == 17 && syn == 0) mark_adder(adders, addr);
        const uint32_t e = info & 0xff;
        const uint32_t sub = e ? (2 * e - 1 + (addr != aa)) : 0;
        return (((df == 17 ? kComboDf17 : kComboDf18) + sub) << 27) | addr;
    }
}

// Items leave the scan kernel without contended atomics: every wavefront owns a region of the item
// array (a.items + wave * a.item_region) and counts its items in a register; only the rare
// per-thread paths share one extra region behind an atomic counter.
__device__ __forceinline__ void emit_item_shared(const BatchArgs &a, uint32_t g, uint32_t info) {
    const uint32_t slot = atomicAdd(&a.counters[kCntItems], 1u);
    if (slot < a.shared_cap) a.items[(size_t)a.n_regions * a.item_region + slot] = ItemRec{g, info};
}
__device__ __forceinline__ void emit_item(const BatchArgs &a, uint32_t g, uint32_t phase, uint32_t cls) {
    if (cls != kDeadItem) emit_item_shared(a, g, cls | (phase << 24));
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
// the scan kernel: wave-autonomous strips
//
// One 1024-thread workgroup per CU shares the UC8 radial table (32.3 KB), the CRC group-syndrome rows, the repair tables of
// --fix and the Bloom filter over them; after that set-up each of its 16 wavefronts works alone on strips of 2768 scan offsets
// (kScanStrip) with its own 7 KB of LDS (WaveShared: 3072 magnitudes = the strip + a 304-sample halo, and three small queues).
// A strip goes through all stages inside its wavefront, so no workgroup barrier is ever executed: while one wavefront is in a
// stage that keeps few lanes busy, the other three wavefronts of its SIMD are in others.  Strips are handed out in chunks
// (scan_body: grab_chunk), whichever wavefront is free next takes the next chunk.
//
//   1. convert: kScanPasses passes, lane l of pass t owns the 8 samples of chunk 64t + l (one 16-byte load of raw IQ per pass,
//      fetched one strip ahead into registers).  While the chunk is still in registers the comparisons m[s] < m[s+1] and
//      m[s] > m[s+1] are taken with packed saturating subtracts and their sign bits shifted into bit-sliced words: byte k of a
//      lane's word, bit t = sample k of chunk 64t + l.
//   2. the edge patterns of demod_2400.c:109-155 on those words: "sample s + j" is byte (k + j) % 8 of this lane, the next lane
//      or the one after, so a shift is a v_alignbyte and every logic instruction decides 4 offsets x kScanPasses passes per lane.
//   3. the ~3 % of offsets that survive are compacted into an LDS queue and take the arithmetic gate (:158-172), one lane per
//      offset; the gated offsets (~0.9 %) are ranked into ascending order.
//   4. (gated offset, trial phase) pairs, one lane each: the first five bits (the DF) decide the length class (:272-285).
//   5. one lane per frame: all 56 / 112 bits in groups of five, CRC syndrome by linearity, classification -> item.
//
// Slicing: bit k = 5u + v of trial phase tp is the sign of correlator (tp + 12v) % 5 at sample
// e0 + 12u + (tp + 12v) / 5.  Whatever tp (4..8) and the parity of e0, the taps of bit v of a group
// fall into three consecutive dwords of LDS at a fixed position relative to dword (e0 >> 1) + 6u:
// {0,1,2}, {1,2,3}, {2,3,4}, {4,5,6}, {5,6,7} for v = 0..4.  So every lane runs the same three
// v_dot2_i32_i16 per bit on statically indexed registers, and phase + alignment only select the
// lane's 15 packed coefficient pairs (a 10-row table in LDS).

constexpr int kBloomWords = 512;
__device__ __forceinline__ uint32_t bloom_bit(uint32_t syn) { return (syn * 0x85EBCA6Bu) >> 18; }  // 14 bits

constexpr int kPasses = kScanPasses;             // 16-byte chunks per lane and strip
constexpr int kRegion = kPasses * 64 * 8;        // magnitudes in LDS per strip
constexpr int kStrip = kScanStrip;               // scan offsets per strip
static_assert(kStrip == kRegion - kHalo && kStrip % 8 == 0, "strip + halo fill the region; strips start on 16-byte IQ boundaries");
constexpr int kScanWaves = kScanWavesPerGroup;
constexpr int kScanThreads = 64 * kScanWaves;
constexpr int kTileShift = 2;   // strips start at S index k*kStrip - 2: then their first IQ sample, k*kStrip - 328, is a multiple of 8
constexpr int kEdgeCap = 64 * kPasses - 128;   // edge-pattern candidates per strip that go through the queue; more take the per-thread path
constexpr int kGateCap = 128;   // gated offsets per strip that go through the dense stages; more take the per-thread path
constexpr int kListCap = 128;   // frames per strip in the dense slicing stage
constexpr int kGsynShort = 0, kGsynLong = 12 * 32;  // short first: a finished 56-bit lane keeps reading entry 0 of long groups (zero)

struct __attribute__((aligned(16))) WaveShared {
    uint16_t m[kRegion];
    uint16_t eq[kEdgeCap];
    uint16_t gq[kGateCap];
    uint16_t fl[kListCap];
};
static_assert(sizeof(WaveShared) % 16 == 0, "per-wave LDS regions must keep 16-byte alignment");

// UC8 magnitudes by "radial" index.  convert.c:35-61 tabulates a function of (2 I - 255)^2 + (2 Q - 255)^2, and
//   P(I, Q) = I (255 - I) + Q (255 - Q)  =  (130050 - that sum) / 4,  even, 0 .. 32512,
// is ONE v_dot4_u32_u8 of the sample's two bytes with their complements: P is the byte offset of the sample's entry in a
// table of uint16.  (The folded 128 x 128 table needed ~7 instructions per sample for its address.)  Small signals land
// in a few neighbouring dwords: distinct banks or the same dword, both conflict-free.
// The reference's float arithmetic breaks the "function of the sum" rule for a few (I, Q): two values of P hold two different
// magnitudes each (60465/60466 and 49681/49682, 48 of the 65536 byte pairs).  The host's table build (modes_host.cpp:
// uc8_radial_lut) finds such entries by comparing and stores magnitude 0 there, which no sample has (the smallest is 363); a
// strip that reads one patches those samples from the full table.
constexpr uint32_t kRadialSentinel = 0u ^ 0x8000u;  // magnitude 0, biased: the smallest int16

// The repair tables of --fix (51 + 107 single-bit syndromes, crc.c:286-372) as the same open-addressed hash in LDS: 128 + 256 slots.
// --aggressive's tables (1326 + 3831 entries) do not fit beside the strips and stay in global memory behind the Bloom filter.
constexpr int kSynLdsSlots = 128 + 256;
struct SynLds {
    uint32_t keys[kSynLdsSlots], vals[kSynLdsSlots];
};

struct __attribute__((aligned(16))) ScanShared {
    uint16_t lut[kRadialTotal];
    uint32_t coef[10][16];           // [parity of e0 * 5 + trial phase - 4][3 v + dword]: negated tap pairs
    uint32_t bloom[2][kBloomWords];  // one bit per hashed syndrome of the 56- / 112-bit repair table: most lookups end here
    uint32_t gsyn[(12 + 23) * 32];   // per 5-bit group of a 56- / 112-bit frame: XOR of the single-bit syndromes of the set bits
    SynLds syn;                      // --fix / --no-fix: the repair tables themselves
    unsigned long long pool;         // chunk numbers for the workgroup's wavefronts: first chunk of the super-chunk << 32 | chunks taken
    unsigned long long pad_;
    WaveShared w[kScanWaves];
};
static_assert(kPasses >= 2 && kPasses <= 8 && kStrip / 8 > 64 * (kPasses - 1), "bit-sliced bytes hold up to 8 passes; the last pass must hold scan offsets");
static_assert(sizeof(ScanShared) <= 160 * 1024, "LDS budget of one CU");

// compiler-level ordering of this wavefront's own LDS traffic (the hardware executes a wave's LDS
// instructions in order; nothing here is shared with another wavefront)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {  // set bits of mask in lanes below this one
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// 8 UC8 samples -> 8 biased magnitudes through the radial table (ScanShared::lut), in two halves so that the table reads of
// one chunk fly while the previous chunk is worked on.  The reads are written out by hand to control exactly that; the
// compiler does not count them in its s_waitcnt bookkeeping, so the waits are ours: LDS operations of a wavefront complete
// in order, lgkmcnt(8) means "everything but the 8 most recent", i.e. the chunk issued before the latest one.
// (ds_read_u16_d16_hi puts the second sample of a pair into the upper half of its register; whether it keeps or clears the
// lower half depends on the SRAM-ECC mode, so the pair is joined with a v_perm that takes only the halves that were loaded.)
struct RadialChunk {
    uint32_t lo[4], hi[4];
};
__device__ __forceinline__ void radial_issue(const uint4 x, uint32_t lut_lds, RadialChunk &c) {  // lut_lds: LDS byte address of the table
    const uint32_t X[4] = {x.x, x.y, x.z, x.w};
    uint32_t P[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // P = I (255 - I) + Q (255 - Q) per sample; the table's address rides in the accumulator
        const uint32_t nx = ~X[k];
        P[2 * k] = __builtin_amdgcn_udot4(X[k] & 0x0000ffffu, nx, lut_lds, false);
        P[2 * k + 1] = __builtin_amdgcn_udot4(X[k] & 0xffff0000u, nx, lut_lds, false);
    }
    asm volatile(
        "ds_read_u16 %0, %8\n\tds_read_u16 %1, %9\n\tds_read_u16 %2, %10\n\tds_read_u16 %3, %11\n\t"
        "ds_read_u16_d16_hi %4, %12\n\tds_read_u16_d16_hi %5, %13\n\tds_read_u16_d16_hi %6, %14\n\tds_read_u16_d16_hi %7, %15"
        : "=&v"(c.lo[0]), "=&v"(c.lo[1]), "=&v"(c.lo[2]), "=&v"(c.lo[3]), "=&v"(c.hi[0]), "=&v"(c.hi[1]), "=&v"(c.hi[2]), "=&v"(c.hi[3])
        : "v"(P[0]), "v"(P[2]), "v"(P[4]), "v"(P[6]), "v"(P[1]), "v"(P[3]), "v"(P[5]), "v"(P[7])
        : "memory");
}
template <int NEWER>  // 8: one younger chunk is in flight; 0: none
__device__ __forceinline__ uint4 radial_take(RadialChunk &c, uint32_t &mn) {
    if constexpr (NEWER == 8)
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(c.lo[0]), "+v"(c.lo[1]), "+v"(c.lo[2]), "+v"(c.lo[3]), "+v"(c.hi[0]), "+v"(c.hi[1]), "+v"(c.hi[2]), "+v"(c.hi[3]) : : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c.lo[0]), "+v"(c.lo[1]), "+v"(c.lo[2]), "+v"(c.lo[3]), "+v"(c.hi[0]), "+v"(c.hi[1]), "+v"(c.hi[2]), "+v"(c.hi[3]) : : "memory");
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        o[k] = __builtin_amdgcn_perm(c.hi[k], c.lo[k], 0x07060100u);
        mn = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(short2v, mn), __builtin_bit_cast(short2v, o[k])));
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// The radial table into a workgroup's LDS: a plain copy of the table the host built (uc8_radial_lut), 16 bytes per load,
// biased on the way (bias2 = 0x80008000 for kernels that keep magnitudes biased, 0 otherwise).  Ends with a workgroup barrier.
__device__ __forceinline__ void load_radial_table(uint16_t *lut, const uint16_t *radial, uint32_t bias2, uint32_t tid, uint32_t nthreads) {
    static_assert(kRadialTotal % 8 == 0, "whole 16-byte pieces");
    for (uint32_t i = tid; i < (uint32_t)kRadialTotal / 8; i += nthreads) {
        uint4 v = ((const uint4 *)radial)[i];
        v.x ^= bias2;
        v.y ^= bias2;
        v.z ^= bias2;
        v.w ^= bias2;
        ((uint4 *)lut)[i] = v;
    }
    __syncthreads();
}

// 8 UC8 samples -> 8 magnitudes (^ bias_xor) through the radial table, exact on the spot: a chunk that meets a sentinel entry
// takes those samples from the full table in global memory (the kernels that convert a few hundred samples per lane; k_scan
// checks once per strip instead)
template <uint32_t BIAS_XOR>
__device__ __forceinline__ uint4 uc8_chunk_radial_exact(const uint4 x, uint32_t lut_lds, const uint32_t patch_p1) {
    using lds_u16 = const __attribute__((address_space(3))) uint16_t;
    const uint32_t X[4] = {x.x, x.y, x.z, x.w};
    typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t nx = ~X[k];
        const uint32_t lo = *(lds_u16 *)(uintptr_t)__builtin_amdgcn_udot4(X[k] & 0x0000ffffu, nx, lut_lds, false);
        const uint32_t hi = *(lds_u16 *)(uintptr_t)__builtin_amdgcn_udot4(X[k] & 0xffff0000u, nx, lut_lds, false);
        o[k] = lo | (hi << 16);
    }
    // the sentinel is the smallest value a table entry can have (magnitude 0; biased: the smallest int16): one packed minimum
    uint32_t mn = o[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        mn = BIAS_XOR ? __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(short2v, mn), __builtin_bit_cast(short2v, o[k])))
                      : __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(ushort2v, mn), __builtin_bit_cast(ushort2v, o[k])));
    const bool hit = (mn & 0xffffu) == BIAS_XOR || (mn >> 16) == BIAS_XOR;
    if (__builtin_expect(hit, 0)) {
        // the patch entries behind the table in LDS (kernels.hpp: kRadialPatch), biased like the table.  (From the full table in
        // global memory this was two dependent round trips to memory for the whole wavefront whenever one lane met a sentinel:
        // on strong signals a third of the chunks, and most of the time of the kernels that use this function.)
        auto patch = [&](uint32_t w) -> uint32_t {  // w = the sample's two bytes, I | Q << 8
            const uint32_t I = w & 0xffu, Q = w >> 8;
            const uint32_t aidx = I >= 128u ? I - 128u : 127u - I, P = I * (255u - I) + Q * (255u - Q);
            return *(lds_u16 *)(uintptr_t)(lut_lds + 2u * ((uint32_t)kRadialEntries + (P == patch_p1 ? 128u : 0u) + aidx));
        };
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((o[k] & 0xffffu) == BIAS_XOR) o[k] = (o[k] & 0xffff0000u) | patch(X[k] & 0xffffu);
            if ((o[k] >> 16) == BIAS_XOR) o[k] = (o[k] & 0xffffu) | (patch(X[k] >> 16) << 16);
        }
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

template <int FMT>
__device__ __forceinline__ uint32_t sc16_pair(uint4 v, int which) {
    constexpr float scale = (FMT == kFmtSC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
    const uint32_t a = which ? v.z : v.x, b = which ? v.w : v.y;
    const uint32_t m0 = float_mag((float)(short)(a & 0xffff) * scale, (float)(short)(a >> 16) * scale);
    const uint32_t m1 = float_mag((float)(short)(b & 0xffff) * scale, (float)(short)(b >> 16) * scale);
    return (m0 | (m1 << 16)) ^ kBias2;
}

// Which branch of scoreModesMessage does a sliced frame take (mode_s.c:330-408)?  Works from the
// pieces the group-parallel slicer reduces: DF + AA (the first 32 bits), the CRC syndrome, and
// whether any message bit is set.  combo << 27 | value, or kDeadItem.
// One code path for 56- and 112-bit frames (lng is per lane): a wavefront that holds both kinds -- nearly every
// one does -- then runs the Bloom test, the table lookup and the address repair once, not once per kind.
__device__ __forceinline__ bool syn_tables_fit_lds(const DeviceTables &t) {
    return (!t.syn_keys[0] || t.syn_mask[0] < 128u) && (!t.syn_keys[1] || t.syn_mask[1] < 256u);
}
__device__ __forceinline__ uint32_t diagnose_lds(const SynLds &T, const DeviceTables &t, int which, uint32_t syn) {
    if (!t.syn_keys[which]) return 0xffffffffu;
    const uint32_t mask = t.syn_mask[which], base = which ? 128u : 0u;
    uint32_t h = syndrome_hash(syn, mask);
    for (;;) {
        const uint32_t k = T.keys[base + h];
        if (k == syn) return T.vals[base + h];
        if (k == 0) return 0xffffffffu;
        h = (h + 1) & mask;
    }
}

__device__ __forceinline__ uint32_t classify_fields(bool lng, uint32_t w0, uint32_t syn, bool nonzero, const DeviceTables &t, uint32_t *adders,
                                                     const uint32_t (*bloom)[kBloomWords], const SynLds *syn_lds) {
    const uint32_t df = w0 >> 27, aa = w0 & 0xffffffu;
    if (!lng && !nonzero) return kDeadItem;  // all-zero message, mode_s.c:325
    // Address/Parity (DF0/4/5, DF16/24) and Comm-B (DF20/21: exact match or nothing, mode_s.c:391-403): the filter decides
    const bool checked = lng ? (df == 17 || df == 18) : (df == 11);
    if (!checked) return ((lng ? ((df == 20 || df == 21) ? kComboCommB : kComboApLong) : kComboApShort) << 27) | syn;
    // DF11: the interrogator id rides in the low 7 syndrome bits (mode_s.c:350-358); DF17/18: the whole syndrome
    const uint32_t s = lng ? syn : (syn & 0xffff80u), iid = lng ? 0u : (syn & 0x7fu);
    uint32_t info = 0;
    if (s) {
        const uint32_t hb = bloom_bit(s);
        if (!((bloom[lng ? 1 : 0][hb >> 5] >> (hb & 31)) & 1u)) return kDeadItem;  // not in the table for sure
        info = syn_lds ? diagnose_lds(*syn_lds, t, lng ? 1 : 0, s) : diagnose(t, lng ? 1 : 0, s);
        if (info == kDeadItem || (!lng && (info & 0xff) > 1)) return kDeadItem;
    }
    const uint32_t addr = repair_aa(aa, info), e = info & 0xff;
    if (s == 0 && (lng ? df == 17 : iid == 0)) mark_adder(adders, addr);  // mode_s.c:717-726
    uint32_t combo;
    if (lng)
        combo = (df == 17 ? kComboDf17 : kComboDf18) + (e ? (2 * e - 1 + (addr != aa)) : 0);
    else
        combo = (iid == 0 ? kComboDf11Iid0 : kComboDf11IidX) + e;
    return (combo << 27) | addr;
}

// lane i <- lane (i + 1) % 64 (DPP wave rotate; no LDS traffic)
__device__ __forceinline__ uint32_t from_next_lane(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x134 /* wave_rol:1 */, 0xf, 0xf, false);
}

__device__ __forceinline__ uint32_t pk_add_sat_u16(uint32_t x, uint32_t y) {  // v_pk_add_u16 clamp: each half min(x + y, 65535)
    typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(ushort2v, x), __builtin_bit_cast(ushort2v, y)));
}
__device__ __forceinline__ uint32_t pk_sub_sat(uint32_t x, uint32_t y) {  // v_pk_sub_i16 clamp: the sign bits are x < y per half
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(short2v, x), __builtin_bit_cast(short2v, y)));
}

// The comparisons of one chunk (8 biased samples in p, the next chunk's first pair in nx) shifted into the bit-sliced
// words: after the passes of a strip, byte 1 of acc[i] holds m[s] < m[s+1] for sample 2i of the lane's chunks (bit t of the
// byte, counted from bit 8 - kPasses = pass t), byte 0 the same for '>', bytes 3 and 2 those of sample 2i + 1.
__device__ __forceinline__ void edge_accumulate(const uint4 &p, uint32_t nx, uint32_t (&acc)[4]) {
    const uint32_t d[5] = {p.x, p.y, p.z, p.w, nx};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t n = __builtin_amdgcn_alignbit(d[i + 1], d[i], 16);  // samples 2i+1, 2i+2
        // v_pk_sub_i16 clamp: sign bits = '<' per half; the sign bytes of both differences side by side
        const uint32_t z = __builtin_amdgcn_perm(pk_sub_sat(d[i], n), pk_sub_sat(n, d[i]), 0x07030501u);
        acc[i] = ((acc[i] >> 1) & 0x7f7f7f7fu) | (z & 0x80808080u);
    }
}

// bytes [R3 R2 R1 R0] of a lane -> the same for the lane's next chunk: the next lane, except that the
// chunk after 64t + 63 is chunk 64(t + 1) + 0: lane 0, one pass later
__device__ __forceinline__ uint32_t next_chunk(uint32_t x, uint32_t lane) {
    const uint32_t r = from_next_lane(x);
    return lane == 63 ? ((r >> 1) & 0x7f7f7f7fu) : r;
}

// word X[r] holds samples 4r..4r+3 of (this lane's chunk ++ the next ++ the one after); the four
// samples starting at sample j
template <int J>
__device__ __forceinline__ uint32_t quad_at(const uint32_t (&X)[5]) {
    if constexpr (J % 4 == 0) return X[J / 4];
    else return __builtin_amdgcn_alignbyte(X[J / 4 + 1], X[J / 4], J % 4);
}

// demod_2400.c:109-151 for the four offsets 4Q..4Q+3 of each of the lane's 8 chunks: the quick check
// and the five peak patterns OR-ed together
template <int Q>
__device__ __forceinline__ uint32_t edge_patterns(const uint32_t (&L)[5], const uint32_t (&G)[5]) {
    constexpr int o = 4 * Q;
    const uint32_t l0 = quad_at<o + 0>(L), l2 = quad_at<o + 2>(L), l3 = quad_at<o + 3>(L), l8 = quad_at<o + 8>(L), l9 = quad_at<o + 9>(L),
                   l10 = quad_at<o + 10>(L), l11 = quad_at<o + 11>(L);
    const uint32_t g1 = quad_at<o + 1>(G), g2 = quad_at<o + 2>(G), g3 = quad_at<o + 3>(G), g4 = quad_at<o + 4>(G), g9 = quad_at<o + 9>(G),
                   g10 = quad_at<o + 10>(G), g12 = quad_at<o + 12>(G);
    const uint32_t A = g1 & l2, B = l3 & g4, C = l8 & g9, D = g10 & l11;
    const uint32_t p12 = A & g3 & C & (l10 | l11);
    const uint32_t p345 = ((A & g4 & l8) | (B & l9 & (g1 | g2))) & D;
    return (p12 | p345) & l0 & g12;
}

constexpr int kGroupDword[5] = {0, 1, 2, 4, 5};  // first of the three dwords that hold the taps of bit v of a 5-bit group

// one entry of the coefficient table: the NEGATED taps (so that "bit is 1" is the sign bit of the sum) of
// bit v that fall into dword kGroupDword[v] + t, for a frame whose first data sample has parity `par`
__device__ uint32_t coef_entry(int par, int phase, int v, int t) {
    const int taps[5][4] = {{5, -3, -2, 0}, {4, -1, -3, 0}, {3, 1, -4, 0}, {2, 3, -5, 0}, {1, 5, -5, -1}};  // demod_2400.c:48-66
    const int P = phase + 4 + 12 * v, base = par + P / 5, phi = P % 5;
    int lo = 0, hi = 0;
    for (int x = 0; x < 4; ++x) {
        const int r = base + x;
        if ((r >> 1) - kGroupDword[v] != t) continue;
        if (r & 1) hi = -taps[phi][x]; else lo = -taps[phi][x];
    }
    return pk16(lo, hi);
}

// the five bits of one group: D = the 8 dwords at (e0 >> 1) + 6u, c = the lane's 15 coefficient pairs
__device__ __forceinline__ int sdot2_first(uint32_t a, uint32_t b) {  // VOP3 form with a literal 0 accumulator
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t slice_group(const uint32_t (&D)[8], const uint32_t (&c)[15]) {
    uint32_t b = 0;
#pragma unroll
    for (int v = 0; v < 5; ++v) {
        const int d = kGroupDword[v];
        const int r = sdot2(D[d], c[3 * v], sdot2(D[d + 1], c[3 * v + 1], sdot2_first(D[d + 2], c[3 * v + 2])));
        b = __builtin_amdgcn_alignbit(b, (uint32_t)r, 31);  // b << 1 | sign
    }
    return b;
}

// LDS window of a frame: dword q of the window that starts at the dword holding sample e0.  The byte address
// goes through an empty asm so that the compiler keeps it in one register and reaches the window with the
// instruction's immediate offsets (it otherwise folds the region's base into a fresh v_add per access).
__device__ __forceinline__ lds_dword *frame_window(const uint16_t *m, uint32_t e0) {
    uint32_t A = (uint32_t)(uintptr_t)m + ((e0 >> 1) << 2);
    asm volatile("" : "+v"(A));
    return (lds_dword *)(uintptr_t)A;
}

__device__ __forceinline__ void load_coefs(const uint32_t (*coef)[16], uint32_t e0, uint32_t phase, uint32_t (&c)[15]) {
    const uint4 *row = (const uint4 *)coef[(e0 & 1u) * 5u + phase];
    const uint4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
    c[0] = r0.x; c[1] = r0.y; c[2] = r0.z; c[3] = r0.w; c[4] = r1.x; c[5] = r1.y; c[6] = r1.z; c[7] = r1.w;
    c[8] = r2.x; c[9] = r2.y; c[10] = r2.z; c[11] = r2.w; c[12] = r3.x; c[13] = r3.y; c[14] = r3.z;
}

// the whole gate on one offset, then every phase on the per-thread path (queue overflow only)
__device__ void slow_offset(const BatchArgs &a, const uint16_t *m, uint32_t j, int64_t s0) {
    const int64_t g = s0 + j;
    if (g < 0 || (uint64_t)g >= a.n || !preamble_gate(m, (int)j)) return;
    atomicAdd(&a.counters[kCntGated], 1u);
    atomicAdd(&a.counters[kCntSlowPath], 1u);
    if (a.emit_dead) emit_item_shared(a, (uint32_t)g, (kComboMarker << 27) | (7u << 24));
    for (int p = 0; p < 5; ++p) slow_item(a, &m[j + 19], (uint32_t)g, p, a.tab.crc_bytes);
}

// magnitudes of a whole region the slow way (per sample, any alignment, zero outside the stream): the first strip
// of a batch (carry), the last ones (end of the stream) and unaligned input
template <int FMT, bool DC>
__device__ void fill_region_general(const BatchArgs &a, uint16_t *m, int64_t s0, uint32_t lane) {
    for (int v = (int)lane; v < kRegion / 8; v += 64) {
        const int64_t s = s0 + 8 * v;
        uint32_t e[8];
        float2 z = make_float2(0.0f, 0.0f);
        int64_t zpos = -1;
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = stream_mag_run<FMT, DC>(a, s + k, z, zpos) ^ kBias;
        ((uint4 *)m)[v] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
    }
}

// The scan of one batch by the workgroups [0, n_regions / 16) given to it; `group` = this workgroup's number among them.
template <int FMT, bool DC = false>
__device__ __forceinline__ void scan_body(const BatchArgs &a, const uint32_t group) {
    static_assert(!DC || FMT != kFmtMag16, "the DC block inside the scan: raw sample formats");
    extern __shared__ __attribute__((aligned(16))) unsigned char scan_lds[];
    ScanShared &sh = *reinterpret_cast<ScanShared *>(scan_lds);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));  // wave-uniform: SGPR
    const uint64_t wave_t0 = a.wave_times ? wall_clock64() : 0;

    if (FMT == kFmtUC8 && !DC) load_radial_table(sh.lut, a.tab.uc8_radial, kBias2, tid, kScanThreads);
    for (uint32_t i = tid; i < 2 * kBloomWords; i += kScanThreads) sh.bloom[0][i] = 0;
    for (uint32_t i = tid; i < 160; i += kScanThreads) {
        const uint32_t row = i >> 4, e = i & 15;
        sh.coef[row][e] = e < 15 ? coef_entry((int)(row / 5), (int)(row % 5), (int)(e / 3), (int)(e % 3)) : 0u;
    }
    for (uint32_t i = tid; i < (12 + 23) * 32; i += kScanThreads) {
        // crc.c:59-64: bit k of an n-bit frame has the syndrome of bit k + 112 - n of a 112-bit frame
        const bool lng = i >= (uint32_t)kGsynLong;
        const uint32_t e = lng ? i - kGsynLong : i, G = e >> 5, v = e & 31, nbits = lng ? 112 : 56;
        uint32_t x = 0;
        for (uint32_t j = 0; j < 5; ++j) {
            const uint32_t k = 5 * G + j;
            if (((v >> (4 - j)) & 1) && k < nbits) x ^= a.tab.sbs[k + 112 - nbits];
        }
        sh.gsyn[i] = x;
    }
    __syncthreads();
    const bool syn_in_lds = syn_tables_fit_lds(a.tab);  // uniform over the grid
    const SynLds *syn_lds = syn_in_lds ? &sh.syn : nullptr;
    for (int which = 0; which < 2; ++which) {
        const uint32_t *keys = a.tab.syn_keys[which];
        if (!keys) continue;  // --no-fix: empty table, every non-zero syndrome is uncorrectable
        for (uint32_t i = tid; i <= a.tab.syn_mask[which]; i += kScanThreads) {
            const uint32_t k = keys[i];
            if (k) atomicOr(&sh.bloom[which][bloom_bit(k) >> 5], 1u << (bloom_bit(k) & 31));
            if (syn_in_lds) {
                sh.syn.keys[(which ? 128u : 0u) + i] = k;
                sh.syn.vals[(which ? 128u : 0u) + i] = a.tab.syn_vals[which][i];
            }
        }
    }
    if (tid == 0) sh.pool = (unsigned long long)(group * (uint32_t)kScanWaves) << 32;
    __syncthreads();  // the last workgroup barrier: from here on every wavefront works alone

    WaveShared &W = sh.w[wave];
    const uint32_t gwave = group * kScanWaves + wave;
    ItemRec *region = a.items + (size_t)gwave * a.item_region;
    uint32_t nitems = 0, gated_total = 0, slow_total = 0, general_strips = 0, fixed_strips = 0, split_strips = 0;  // wave-uniform
    const uint32_t nstrips = (uint32_t)((a.n + kTileShift + kStrip - 1) / kStrip);  // (2^32 strips = 10^13 samples: beyond any HBM)
    // The raw IQ of a strip is fetched one strip ahead into registers (8 x 16 bytes per lane and raw vector), so
    // the HBM latency of strip k + 1 hides behind the gate / slicing stages of strip k.
    constexpr bool kVecFmt = true;
    constexpr int kRawVecs = (FMT == kFmtUC8 || FMT == kFmtMag16) ? 1 : 2;  // 16-byte loads per 8 samples
    constexpr int kBps = 2 * kRawVecs;                                       // bytes per sample
    uint4 pre[kPasses][kRawVecs];
    auto interior = [&](uint32_t strip) {  // wave-uniform: every chunk of the strip is aligned raw IQ (all strips but the first and the last few)
        const int64_t q = (int64_t)((uint64_t)strip * kStrip) - kTileShift - (int64_t)kTrailing;
        return kVecFmt && a.iq_aligned && strip < nstrips && q >= 0 && q + kRegion <= (int64_t)a.n;
    };
    // The last kLatePasses chunks of a strip are fetched at the top of the strip itself (they are converted last, behind the table
    // reads of the others): 4 registers per raw vector and pass that the slicing stage of the strip before does not have to carry.
    constexpr int kLatePasses = RSB_LATE_PASSES;
    auto fetch_passes = [&](uint32_t strip, int t_lo, int t_hi) {
        if (!interior(strip)) return;
        const int64_t q = (int64_t)((uint64_t)strip * kStrip) - kTileShift - (int64_t)kTrailing;
        const uint4 *src0 = (const uint4 *)((const uint8_t *)a.iq + kBps * q) + kRawVecs * (int)lane;
#pragma unroll
        for (int t = 0; t < kPasses; ++t)
            if (t >= t_lo && t < t_hi) {
#pragma unroll
                for (int r = 0; r < kRawVecs; ++r) pre[t][r] = src0[kRawVecs * 64 * t + r];
            }
    };
    auto prefetch = [&](uint32_t strip) { fetch_passes(strip, 0, kPasses - kLatePasses); };
    uint32_t lut_min = 0x7fff7fffu;  // UC8: packed minimum of the strip's table reads
    auto direct_chunk = [&](int t) {
        if constexpr (FMT == kFmtUC8) {
            return pre[t][0];  // (not used: UC8 strips go through radial_issue / radial_take)
        } else if constexpr (FMT == kFmtMag16) {
            const uint4 x = pre[t][0];
            return make_uint4(x.x ^ kBias2, x.y ^ kBias2, x.z ^ kBias2, x.w ^ kBias2);
        } else {
            const uint4 x0 = pre[t][0], x1 = pre[t][kRawVecs - 1];
            return make_uint4(sc16_pair<FMT>(x0, 0), sc16_pair<FMT>(x0, 1), sc16_pair<FMT>(x1, 0), sc16_pair<FMT>(x1, 1));
        }
    };
    // The strips are handed out in CHUNKS of a.chunk_strips consecutive strips, to whichever wavefront is free next: traffic is
    // uneven along a recording (on the benchmark's, equal runs of strips took between 0.88 and 1.58 ms: the average wavefront
    // was resident for 75 % of the launch, profiles/r3_wave_times_before.json).  A wavefront walks its chunk's strips in ascending
    // order, so a chunk's items are one sorted run; chunk_dir says where each chunk's run lies, and the ordered prune packs the runs
    // in chunk order = stream order.  Chunk numbers come from a two-level counter: the workgroup's pool in LDS holds one
    // super-chunk (kScanWaves chunks); the wavefront that finds it used up takes the next super-chunk number from the batch's
    // counter in memory (same-address atomics retire at ~0.1 per ns: one per chunk and wavefront would be too many) and refills
    // the pool.  Every workgroup's first super-chunk is its own number: no atomic before the first strip.
    const uint32_t n_groups = a.n_regions / (uint32_t)kScanWaves;
    auto grab_chunk = [&]() -> uint32_t {  // wave-uniform; >= a.n_chunks: nothing is left
        for (;;) {
            unsigned long long v = 0;
            if (lane == 0) v = __hip_atomic_fetch_add(&sh.pool, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
            const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
            if (idx < (uint32_t)kScanWaves) return base + idx;
            if (idx == (uint32_t)kScanWaves) {  // the first to find the pool empty refills it (and takes the new super-chunk's first chunk)
                uint32_t sc = 0;
                if (lane == 0) sc = atomicAdd(&a.counters[kCntSuperChunks], 1u);
                sc = (uint32_t)__builtin_amdgcn_readfirstlane((int)sc) + n_groups;
                const uint32_t nb = sc < 0x08000000u ? sc * (uint32_t)kScanWaves : 0xffffff00u;  // (past the end: stays past the end)
                if (lane == 0) __hip_atomic_store(&sh.pool, ((unsigned long long)nb << 32) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return nb;
            }
            __builtin_amdgcn_s_sleep(16);  // another wavefront is refilling
        }
    };
    const uint32_t n_chunks = a.stage_limit < 1 ? 0u : a.n_chunks;  // (ablation: the table set-up alone)
    uint32_t chunk = grab_chunk(), chunk_items0 = 0, strips_taken = 0;
    uint32_t strip = chunk < n_chunks ? chunk * a.chunk_strips : nstrips;
    uint32_t chunk_end = strip < nstrips ? min(strip + a.chunk_strips, nstrips) : nstrips;
    uint32_t next_strip = nstrips, chunk_next = chunk;
    // closes the chunk when its last strip is done, and steps to the strip chosen at the top of the body
    auto advance = [&]() {
        ++strips_taken;
        if (chunk_next != chunk || next_strip >= nstrips) {
            const uint32_t cap = a.item_region, lo = min(chunk_items0, cap), hi = min(nitems, cap);
            if (lane == 0) a.chunk_dir[chunk] = make_uint2(gwave * cap + lo, hi - lo);
            chunk_items0 = nitems;
            chunk = chunk_next;
            chunk_end = next_strip < nstrips ? min(next_strip + a.chunk_strips, nstrips) : nstrips;
        }
        strip = next_strip;
    };
    if (RSB_PREFETCH_EARLY && strip < nstrips) prefetch(strip);
    DBG_DECL
    for (; strip < nstrips; advance()) {
        // which strip comes next (needed early: its raw IQ is requested while this strip is worked on)
        if (strip + 1 < chunk_end) {
            next_strip = strip + 1;
        } else {
            chunk_next = grab_chunk();
            next_strip = chunk_next < n_chunks ? chunk_next * a.chunk_strips : nstrips;
        }
        const int64_t s0 = (int64_t)((uint64_t)strip * kStrip) - kTileShift;  // S index of m[0]
        if (!RSB_PREFETCH_EARLY) prefetch(strip);
        if (kLatePasses) fetch_passes(strip, kPasses - kLatePasses, kPasses);
        wave_sync();

        // ---- convert.c: strip + halo -> magnitudes in LDS; comparisons of neighbours -> bit-sliced words
        uint32_t acc[4] = {0, 0, 0, 0};
        uint4 prev = make_uint4(0, 0, 0, 0);
        auto take_chunk = [&](int t, const uint4 o) {
            // the chunk after this lane's chunk of pass t - 1 is the next lane's, or (lane 63) lane 0's of this pass
            if (t > 0) edge_accumulate(prev, from_next_lane(lane == 0 ? o.x : prev.x), acc);
            prev = o;
        };
        bool general = !interior(strip);  // wave-uniform
        if (!general) {
            lut_min = 0x7fff7fffu;
            if constexpr (DC) {
                // (an interior strip: its first raw sample is a multiple of 8, so a lane's 8 samples are one chunk of the filter's states)
                const uint64_t chunk0 = (uint64_t)(((int64_t)((uint64_t)strip * kStrip) - kTileShift - (int64_t)kTrailing) >> 3) + lane;
#pragma unroll
                for (int t = 0; t < kPasses; ++t) {
                    const uint4 o = dc_chunk<FMT>(a, pre[t][0], pre[t][kRawVecs - 1], chunk0 + 64u * (uint32_t)t);
                    ((uint4 *)W.m)[(int)lane + 64 * t] = o;
                    take_chunk(t, o);
                }
            } else if constexpr (FMT == kFmtUC8) {
                const uint32_t lut_lds = (uint32_t)(uintptr_t)sh.lut;
                RadialChunk rc[2];
                if (RSB_RADIAL_DEPTH == 2) radial_issue(pre[0][0], lut_lds, rc[0]);
#pragma unroll
                for (int t = 0; t < kPasses; ++t) {
                    uint4 o;
                    if (RSB_RADIAL_DEPTH == 1) {  // the table reads of a chunk are waited for at once (16 registers less in this stage)
                        radial_issue(pre[t][0], lut_lds, rc[0]);
                        o = radial_take<0>(rc[0], lut_min);
                    } else if (t + 1 < kPasses) {
                        radial_issue(pre[t + 1][0], lut_lds, rc[(t + 1) & 1]);
                        o = radial_take<8>(rc[t & 1], lut_min);
                    } else {
                        o = radial_take<0>(rc[t & 1], lut_min);
                    }
                    ((uint4 *)W.m)[(int)lane + 64 * t] = o;
                    take_chunk(t, o);
                }
            } else {
#pragma unroll
                for (int t = 0; t < kPasses; ++t) {
                    const uint4 o = direct_chunk(t);
                    ((uint4 *)W.m)[(int)lane + 64 * t] = o;
                    take_chunk(t, o);
                }
            }
            if constexpr (FMT == kFmtUC8 && !DC) {
                // a sample of the strip read an entry the radial table cannot hold (about one strip in fifteen on strong signals):
                // those samples get their magnitude from the full table in global memory, the comparisons are taken again from LDS
                const bool hit = (int16_t)lut_min == (int16_t)kRadialSentinel || (int16_t)(lut_min >> 16) == (int16_t)kRadialSentinel;
                if (__ballot(hit) != 0) {  // wave-uniform
                    wave_sync();
                    if (hit) {
                        // (the raw bytes are read again from memory, in a rolled loop: keeping the prefetch registers alive for
                        // this rare path cost the kernel 11 VGPRs and made it spill)
                        const int64_t q = (int64_t)((uint64_t)strip * kStrip) - kTileShift - (int64_t)kTrailing;
                        const uint4 *src0 = (const uint4 *)((const uint8_t *)a.iq + kBps * q) + (int)lane;
#pragma unroll 1
                        for (int t = 0; t < kPasses; ++t) {
                            const uint4 raw = src0[64 * t];
                            uint4 o = ((const uint4 *)W.m)[(int)lane + 64 * t];
                            const uint32_t X[4] = {raw.x, raw.y, raw.z, raw.w};
                            uint32_t O[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                if ((O[k] & 0xffffu) == kRadialSentinel) O[k] = (O[k] & 0xffff0000u) | (a.tab.uc8_lut[X[k] & 0xffffu] ^ kBias);
                                if ((O[k] >> 16) == kRadialSentinel) O[k] = (O[k] & 0xffffu) | ((a.tab.uc8_lut[X[k] >> 16] ^ kBias) << 16);
                            }
                            ((uint4 *)W.m)[(int)lane + 64 * t] = make_uint4(O[0], O[1], O[2], O[3]);
                        }
                    }
                    wave_sync();
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = 0;
#pragma unroll
                    for (int t = 0; t < kPasses; ++t) take_chunk(t, ((const uint4 *)W.m)[(int)lane + 64 * t]);
                    fixed_strips += 1;
                }
            }
        }
        if (general) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = 0;
            general_strips += 1;
            wave_sync();
            fill_region_general<FMT, DC>(a, W.m, s0, lane);
            wave_sync();
#pragma unroll
            for (int t = 0; t < kPasses; ++t) take_chunk(t, ((const uint4 *)W.m)[(int)lane + 64 * t]);
        }
        edge_accumulate(prev, from_next_lane(lane == 0 ? 0u : prev.x), acc);  // chunk 511 has no successor: not a scan offset
        if (RSB_PREFETCH_EARLY && next_strip < nstrips) prefetch(next_strip);
        wave_sync();
        DBG_STAGE(0) DBG_COUNT(0, 1)
        if (a.stage_limit < 2) continue;

        // ---- demod_2400.c:109-155, edge patterns only, bit-parallel: 4 offsets x 8 passes per lane and instruction
        uint32_t E[2];  // per lane: the candidates of its chunks, 4 offsets x kPasses passes per word
        {
            uint32_t L[5], G[5];
            L[0] = __builtin_amdgcn_perm(acc[1], acc[0], 0x07050301u);  // samples 0..3 of the lane's chunks
            L[1] = __builtin_amdgcn_perm(acc[3], acc[2], 0x07050301u);  // samples 4..7
            G[0] = __builtin_amdgcn_perm(acc[1], acc[0], 0x06040200u);
            G[1] = __builtin_amdgcn_perm(acc[3], acc[2], 0x06040200u);
            L[2] = next_chunk(L[0], lane);
            L[3] = next_chunk(L[1], lane);
            L[4] = next_chunk(L[2], lane);
            G[2] = next_chunk(G[0], lane);
            G[3] = next_chunk(G[1], lane);
            G[4] = next_chunk(G[2], lane);
            // bit 8 - kPasses + t of a byte = pass t.  The last pass holds chunks 64 (kPasses - 1) + lane: scan offsets only
            // below chunk kStrip / 8; the bits below the first pass are shifted-in leftovers of next_chunk()
            constexpr uint32_t kPassBits = ((0xffu << (8 - kPasses)) & 0xffu) * 0x01010101u;
            const uint32_t live = lane < (uint32_t)(kStrip / 8 - 64 * (kPasses - 1)) ? kPassBits : (kPassBits & 0x7f7f7f7fu);
            E[0] = edge_patterns<0>(L, G) & live;
            E[1] = edge_patterns<1>(L, G) & live;
        }
        // The stages below keep their candidates in LDS queues of fixed size.  A strip that overflows one (strong bursts: a few
        // strips per Gsample) is taken again in parts -- 2, 3, then kPasses runs of passes, i.e. ranges of 512 consecutive offsets
        // and more, in ascending order -- each part with the whole queues to itself, so the wavefront's items still leave sorted
        // (the ordered prune depends on it; one item on the per-thread path sends the whole batch through the radix sort).  Only
        // when single passes overflow do the per-thread paths run.  Nothing of a strip is written before its last stage, and
        // what that writes (items, adder marks) is written again identically.
        // The code of the stages exists twice: once for the whole strip with nothing of the splitting in it (every strip runs
        // it first; folding the general case into that one cost 4 % of the kernel), once for a strip in parts.
        uint32_t nparts = 1;  // wave-uniform
        const uint32_t nitems0 = nitems;
        uint32_t gated_strip = 0, slow_strip = 0;
        auto attempt = [&](auto whole_strip) -> bool {  // -> a queue overflowed: take the strip again in more parts
        constexpr bool kWhole = decltype(whole_strip)::value;
        nitems = nitems0;
        gated_strip = slow_strip = 0;
        const uint32_t np = kWhole ? 1u : nparts;
        for (uint32_t part = 0; part < np; ++part) {
        const bool last_resort = !kWhole && np >= (uint32_t)kPasses;
        const uint32_t p_lo = part * kPasses / np, p_hi = (part + 1) * kPasses / np;
        const uint32_t part_bits = kWhole ? 0xffffffffu : (((1u << p_hi) - (1u << p_lo)) << (8 - kPasses)) * 0x01010101u;
        uint32_t neq = 0;  // wave-uniform
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            uint32_t e = E[q] & part_bits;
            for (;;) {
                const uint64_t bm = __ballot(e != 0);
                if (!bm) break;
                if (e) {
                    const uint32_t b = (uint32_t)__builtin_ctz(e);
                    e &= e - 1;
                    const uint32_t j = (((b & 7u) - (8 - kPasses)) << 9) | (lane << 3) | (4u * q + (b >> 3));  // pass, lane, sample of the chunk
                    const uint32_t pos = neq + lanes_below(bm);
                    if (pos < (uint32_t)kEdgeCap)
                        W.eq[pos] = (uint16_t)j;
                    else if (last_resort)
                        slow_offset(a, W.m, j, s0);
                }
                neq += __popcll(bm);
            }
        }
        if (neq > (uint32_t)kEdgeCap && !last_resort) return true;
        wave_sync();
        DBG_STAGE(1) DBG_COUNT(4, neq)
        if (a.stage_limit < 3) continue;

        // ---- the whole gate (:109-172) on the edge candidates; survivors go to the gated queue
        uint32_t ngated = 0;  // wave-uniform
        neq = min(neq, (uint32_t)kEdgeCap);
        for (uint32_t cb = 0; cb < neq; cb += 64) {
            const uint32_t c = cb + lane;
            uint32_t j = 0;
            bool pass = false;
            if (c < neq) {
                j = W.eq[c];
                const int64_t g = s0 + j;
                pass = g >= 0 && (uint64_t)g < a.n && preamble_gate(W.m, (int)j);
            }
            const uint64_t pm = __ballot(pass);
            if (pass) {
                const uint32_t g = (uint32_t)(s0 + j);
                const uint32_t pos = ngated + lanes_below(pm);
                if (a.emit_dead) {
                    const uint32_t ipos = nitems + lanes_below(pm);
                    if (ipos < a.item_region) region[ipos] = ItemRec{g, (kComboMarker << 27) | (7u << 24)};
                }
                if (pos < (uint32_t)kGateCap) {
                    W.gq[pos] = (uint16_t)j;
                } else if (last_resort) {
                    for (int p = 0; p < 5; ++p) slow_item(a, &W.m[j + 19], g, p, a.tab.crc_bytes);
                }
            }
            if (a.emit_dead) nitems += __popcll(pm);
            ngated += __popcll(pm);
        }
        wave_sync();
        if (ngated > (uint32_t)kGateCap && !last_resort) return true;
        gated_strip += ngated;
        DBG_STAGE(2) DBG_COUNT(1, (neq + 63) / 64)
        if (ngated > (uint32_t)kGateCap) slow_strip += ngated - kGateCap;
        if (a.stage_limit < 4) continue;
        const uint32_t ngq = min(ngated, (uint32_t)kGateCap);
        // The gated offsets in ascending order (the edge queue is filled bit by bit, not offset by offset): rank by counting, the
        // queue is short (about 25 entries).  From here on everything keeps that order -- (offset, phase) pairs, the frame list,
        // the ballot compactions -- so the wavefront's items leave sorted by (offset, phase).
        if (RSB_RANK_CHEAP && ngq > 1 && ngq <= 64) {
            // one register per lane, and the comparison as arithmetic on the cheap VALU operations (offsets are below 4096:
            // (mine + 4095 - other) >> 12 is 1 exactly when other < mine) instead of v_cmp + v_addc through VCC
            const uint32_t mine = lane < ngq ? W.gq[lane] : 0xfffu, mine_p = mine + 4095u;
            uint32_t rank = 0;
            for (uint32_t k = 0; k < ngq; ++k) rank += (mine_p - (uint32_t)W.gq[k]) >> 12;  // same address in every lane: a broadcast read
            wave_sync();
            if (lane < ngq) W.gq[rank] = (uint16_t)mine;  // offsets are distinct: so are the ranks
            wave_sync();
        } else if (ngq > 1 && !(a.debug_flags & 8)) {
            uint32_t mine[kGateCap / 64], rank[kGateCap / 64];
#pragma unroll
            for (int h = 0; h < kGateCap / 64; ++h) {
                mine[h] = (64u * h + lane < ngq) ? W.gq[64 * h + lane] : 0xffffu;
                rank[h] = 0;
            }
            for (uint32_t k = 0; k < ngq; ++k) {
                const uint32_t other = W.gq[k];  // same address in every lane: a broadcast read
#pragma unroll
                for (int h = 0; h < kGateCap / 64; ++h) rank[h] += other < mine[h] ? 1u : 0u;
            }
            wave_sync();
#pragma unroll
            for (int h = 0; h < kGateCap / 64; ++h)
                if (64u * h + lane < ngq) W.gq[rank[h]] = (uint16_t)mine[h];  // offsets are distinct: so are the ranks
            wave_sync();
        }

        // ---- first five bits (the DF) of every (offset, trial phase): which length class? (:272-285)
        uint32_t nlist = 0;
        for (uint32_t ib = 0; ib < 5 * ngq; ib += 64) {
            const uint32_t i = ib + lane;
            uint32_t cls = 2, j = 0, phase = 0;
            if (i < 5 * ngq) {
                const uint32_t cand = i / 5;
                phase = i - 5 * cand;
                j = W.gq[cand];
                const uint32_t e0 = j + 19;
                uint32_t c[15], D[8];
                load_coefs(sh.coef, e0, phase, c);
                lds_dword *mw = frame_window(W.m, e0);
#pragma unroll
                for (int k = 0; k < 8; ++k) D[k] = mw[k];
                const uint32_t df = slice_group(D, c);
                cls = ((kShortDfMask >> df) & 1) ? 0u : (((kLongDfMask >> df) & 1) ? 1u : 2u);
            }
            const uint64_t cm = __ballot(cls < 2);
            if (cls < 2) {
                const uint32_t pos = nlist + lanes_below(cm);
                if (pos < (uint32_t)kListCap) {
                    W.fl[pos] = (uint16_t)(i | (cls << 15));
                } else if (last_resort) {
                    slow_item(a, &W.m[j + 19], (uint32_t)(s0 + j), (int)phase, a.tab.crc_bytes);
                }
            }
            nlist += __popcll(cm);
        }
        if (nlist > (uint32_t)kListCap && !last_resort) return true;
        if (nlist > (uint32_t)kListCap) slow_strip += nlist - kListCap;
        wave_sync();
        DBG_STAGE(3) DBG_COUNT(2, (5 * ngq + 63) / 64)
        if (a.stage_limit < 5) continue;

        // ---- one lane per frame: 12 (56 bits) or 23 (112 bits) groups of five bits
        const uint32_t nfl = min(nlist, (uint32_t)kListCap);
        for (uint32_t cb = 0; cb < nfl; cb += 64) {
            const uint32_t ci = cb + lane;
            const bool valid = ci < nfl;
            uint32_t j = 0, phase = 0;
            bool lng = false;
            if (valid) {
                const uint32_t ent = W.fl[ci], i = ent & 0x7fffu, cand = i / 5;
                lng = (ent >> 15) != 0;
                phase = i - 5 * cand;
                j = W.gq[cand];
            }
            const bool any_long = __ballot(lng) != 0;
            DBG_COUNT(3, 1) DBG_COUNT(6, (uint32_t)__popcll(__ballot(lng)))
            const uint32_t e0 = j + 19;
            uint32_t c[15], D[8], NX[6], NY[6];
            load_coefs(sh.coef, e0, phase, c);
            uint32_t A = (uint32_t)(uintptr_t)W.m + ((e0 >> 1) << 2);  // LDS byte address of the window's first dword
            uint32_t T = (uint32_t)(uintptr_t)sh.gsyn + 4u * (lng ? kGsynLong : kGsynShort);  // ... of this group's syndrome row
            constexpr int kAhead = RSB_SLICE_AHEAD;  // how many groups ahead of its use a window's six new dwords are requested
            static_assert(kAhead == 1 || kAhead == 2, "");
            {
                lds_dword *mw = (lds_dword *)(uintptr_t)A;
#pragma unroll
                for (int k = 0; k < 8; ++k) D[k] = mw[k];
                if (kAhead == 2) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) NY[k] = mw[8 + k];  // group 1's new dwords
                }
            }
            // Runtime loops (not 23 unrolled groups: the register allocator then keeps the whole stage in ~70 VGPRs).
            // Per group: the six new dwords of a later group go in flight first (kAhead groups later: with 2 the wavefronts wait
            // less in this stage, but the kernel then needs 128 VGPRs and spills a prefetch register to scratch: RSB_SLICE_AHEAD),
            // the group's syndrome (crc.c:59-64,140: the CRC is linear) is looked up now and folded in one group later.
            uint32_t syn = 0, look = 0, hi = 0, lo = 0;  // hi:lo = the message bits so far (only the first 55 matter)
            auto group_io = [&](uint32_t (&load_into)[6], const uint32_t (&take_from)[6], bool more, uint32_t keep) {
                asm volatile("" : "+v"(A), "+v"(T));  // one address register each, the rest in the instructions' offsets
                lds_dword *mw = (lds_dword *)(uintptr_t)A;
                if (more) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) load_into[k] = mw[2 + 6 * kAhead + k];
                }
                const uint32_t b = slice_group(D, c) & keep;
                syn ^= look;
                look = *(lds_dword *)(uintptr_t)(T + 4u * b);
                D[0] = D[6];
                D[1] = D[7];
#pragma unroll
                for (int k = 0; k < 6; ++k) D[2 + k] = take_from[k];
                A += 24;
                T += 128;
                return b;
            };
            // group g: with kAhead == 2 even groups load into NX and take NY, odd groups the other way round; with 1 everything
            // goes through NY.  Group g + kAhead is fetched if the frame (or, past group 11, any frame of the batch) has it: the
            // windows up to group 13 lie inside the region whatever the frame's length (13 * 12 + 19 + 12 < kHalo).
            auto group = [&](int g, uint32_t keep) {
                const bool more = g + kAhead <= 11 || (any_long && g + kAhead <= 22);
                if (kAhead == 1) return group_io(NY, NY, more, keep);
                return (g & 1) ? group_io(NY, NX, more, keep) : group_io(NX, NY, more, keep);
            };
            auto message_bits = [&](uint32_t b) {
                hi = __builtin_amdgcn_alignbit(hi, lo, 27);
                lo = (lo << 5) | b;
            };
            lo = group(0, 31u);
            for (int u = 1; u < 9; u += 2) {  // (g + kAhead <= 11 for all of these: the flag folds)
                message_bits(group(1, 31u));
                message_bits(group(2, 31u));
            }
            message_bits(group(9, 31u));
            message_bits(group(10, 31u));
            const uint32_t w0 = __builtin_amdgcn_alignbit(hi, lo, 23);  // message bits 0..31: DF + AA
            // group 11: a 56-bit frame ends with its first bit (bit 55)
            const uint32_t b11 = group(11, lng ? 31u : 16u);
            const uint32_t nz = hi | lo | b11;  // only 56-bit frames are checked for all-zero (mode_s.c:325)
            uint32_t syn_short = syn ^ look;
            if (any_long) {  // wave-uniform; lanes with 56-bit frames run along and are discarded below
                for (int u = 12; u < 20; u += 2) {
                    group(12, 31u);
                    group(13, 31u);
                }
                group(20, 31u);
                group(21, 31u);
                group(22, 24u);  // group 22 holds bits 110..114: 110 and 111 belong to the frame
                syn ^= look;
            }
            syn = lng ? syn : syn_short;
            uint32_t cls = kDeadItem;
            if (valid) cls = classify_fields(lng, w0, syn, nz != 0, a.tab, a.adder_bitmap, sh.bloom, syn_lds);
            const uint64_t em = __ballot(cls != kDeadItem);
            if (cls != kDeadItem) {
                const uint32_t pos = nitems + lanes_below(em);
                if (pos < a.item_region) region[pos] = ItemRec{(uint32_t)(s0 + j), cls | (phase << 24)};
            }
            nitems += __popcll(em);
        }
        DBG_STAGE(4) DBG_COUNT(5, nfl)
        }  // parts of the strip
        return false;
        };
        bool redo = attempt(std::true_type{});
        while (redo) {
            wave_sync();
            nparts = nparts == 1 ? 2u : (nparts == 2 && kPasses > 3 ? 3u : (uint32_t)kPasses);
            split_strips += 1;
            redo = attempt(std::false_type{});
        }
        gated_total += gated_strip;
        slow_total += slow_strip;
    }
    DBG_FLUSH
    if (a.wave_times && lane == 0) {
        uint64_t *wt = a.wave_times + 4 * (size_t)gwave;
        wt[0] = wave_t0;
        wt[1] = wall_clock64();
        wt[2] = ((uint64_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) | (uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4);  // XCC_ID, HW_ID
        wt[3] = strips_taken;
    }
    if (lane == 0) {
        a.region_counts[gwave] = nitems;
        if (nitems > a.item_region) atomicAdd(&a.counters[kCntOverflow], 1u);
        if (gated_total) atomicAdd(&a.counters[kCntGated], gated_total);
        if (slow_total) atomicAdd(&a.counters[kCntSlowPath], slow_total);
        if (general_strips) atomicAdd(&a.counters[kCntGeneralStrips], general_strips);
        if (fixed_strips) atomicAdd(&a.counters[kCntFixedStrips], fixed_strips);
        if (split_strips) atomicAdd(&a.counters[kCntSplitStrips], split_strips);
    }
}

template <int FMT, bool DC = false>
__global__ __launch_bounds__(kScanThreads) void k_scan(const BatchArgs a) {
    scan_body<FMT, DC>(a, blockIdx.x);
}

// Many batches (receiver channels) in one launch: batch i has the workgroups [i * groups_per_batch, (i + 1) * groups_per_batch) and its
// arguments -- built for that many workgroups -- in args[i].  A channel of a few ten Msamples alone gives a wavefront two or three
// strips and pays the kernel's set-up and ramp for them (33 us per strip against 17 in a long run); side by side the channels'
// wavefronts each walk a long run of their own channel.
template <int FMT>
__global__ __launch_bounds__(kScanThreads) void k_scan_many(const BatchArgs *__restrict__ args, const uint32_t groups_per_batch) {
    const uint32_t batch = blockIdx.x / groups_per_batch;  // workgroup-uniform: the arguments arrive by scalar loads
    const BatchArgs a = args[batch];
    const uint32_t group = blockIdx.x - batch * groups_per_batch;
    if (a.n == 0 || group * (uint32_t)kScanWaves >= a.n_regions) return;  // (a short batch was given fewer workgroups)
    scan_body<FMT>(a, group);
}

// one workgroup per item region (n_regions wave regions + the shared region)
__global__ __launch_bounds__(kThreads) void k_prune(const BatchArgs a) {
    const uint32_t r = blockIdx.x;
    uint32_t n, cap;
    if (r < a.n_regions) {
        n = a.region_counts[r];
        cap = a.item_region;
    } else {
        n = a.counters[kCntItems];
        cap = a.shared_cap;
    }
    if (n > cap) {
        if (threadIdx.x == 0) atomicAdd(&a.counters[kCntOverflow], 1u);
        n = cap;
    }
    const ItemRec *items = a.items + (size_t)r * a.item_region;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto keeps = [&](const ItemRec &rec) {
        const uint32_t combo = item_combo(rec.info), v = item_value(rec.info);
        if (rec.info == kDeadItem) return false;  // padding between the chunks of a region
        if (a.keep_all != 0 || combo == kComboMarker || !combo_needs_member(combo)) return true;
        return ((a.adder_bitmap[v >> 5] >> (v & 31)) & 1u) != 0;
    };
    // Two passes, so that the shared output counter is bumped once per workgroup and not once per item or
    // wavefront (same-address atomics retire at ~0.1 per ns): each wavefront owns a quarter of the region,
    // counts its kept items, then writes them behind the workgroup's reservation.
    __shared__ uint32_t wave_count[kThreads / 64];
    __shared__ uint32_t block_base;
    const uint32_t per_wave = (n + kThreads / 64 - 1) / (kThreads / 64);
    const uint32_t w_lo = min(n, wave * per_wave), w_hi = min(n, w_lo + per_wave);
    __shared__ uint32_t wave_live[kThreads / 64];
    uint32_t mine = 0, live = 0;
    for (uint32_t base = w_lo; base < w_hi; base += 64) {
        const uint32_t i = base + lane;
        const ItemRec rec = i < w_hi ? items[i] : ItemRec{0, kDeadItem};
        mine += (uint32_t)__popcll(__ballot(keeps(rec)));
        live += (uint32_t)__popcll(__ballot(rec.info != kDeadItem));
    }
    if (lane == 0) {
        wave_count[wave] = mine;
        wave_live[wave] = live;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        n = 0;
        for (uint32_t w = 0; w < kThreads / 64; ++w) {
            total += wave_count[w];
            n += wave_live[w];
        }
        // kept items (the reservation) and all items (a statistic) in one atomic: same-address atomics are what this kernel waits for
        static_assert(kCntSurvItems % 2 == 0 && kCntItemsTotal == kCntSurvItems + 1, "the two counters form one aligned 64-bit word");
        block_base = n ? (uint32_t)atomicAdd((unsigned long long *)&a.counters[kCntSurvItems], (unsigned long long)total | ((unsigned long long)n << 32)) : 0u;
    }
    __syncthreads();
    uint32_t slot0 = block_base;
    for (uint32_t w = 0; w < wave; ++w) slot0 += wave_count[w];
    for (uint32_t base = w_lo; base < w_hi; base += 64) {
        const uint32_t i = base + lane;
        ItemRec rec{0, 0};
        bool keep = false;
        if (i < w_hi) {
            rec = items[i];
            keep = keeps(rec);
        }
        const uint64_t km = __ballot(keep);
        const uint32_t slot = slot0 + lanes_below(km);
        if (keep && slot < a.surv_item_cap) {
            a.skeys[0][slot] = ((uint64_t)rec.g << 3) | item_phase(rec.info);
            a.sinfo[0][slot] = rec.info;
        }
        slot0 += (uint32_t)__popcll(km);
    }
}

// ---- the prune as an ordered compaction.  The scan leaves every wavefront's items sorted by (offset, phase) and the regions in
// stream order (contiguous strip ranges), so the kept items only have to be packed region after region: count per region,
// exclusive sum over the regions, write.  No atomics on shared counters and no sort afterwards.  One wavefront per region.
// The ordered prune asks the 2 MiB adder bitmap about every item, 13 M random 4-byte reads per Gsample, each of which pulls a
// whole cache line into L1 (1.7 GB: what k_prune_count's 70 us were).  A few hundred to a few thousand bits are set.  So, for a
// large batch, a 4 KB hashed summary of the bitmap is rebuilt in front of the prune (the bitmap is final then: every scan of the
// batch has run, the caller's additions are in), every workgroup copies it into LDS, and only the items whose summary bit is set
// -- the members and a few per cent of the others -- go to the bitmap.
__device__ __forceinline__ uint32_t summary_bit(uint32_t v) { return (v ^ (v >> 9)) & (32u * kSummaryWords - 1u); }

__device__ __forceinline__ void prune_summary_body(const uint32_t *bitmap, uint32_t *summary) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;  // one 16-byte piece of the bitmap per thread
    const uint4 w = ((const uint4 *)bitmap)[i];
    const uint32_t W[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        for (uint32_t x = W[k]; x; x &= x - 1) {
            const uint32_t h = summary_bit(((4u * i + (uint32_t)k) << 5) | (uint32_t)__builtin_ctz(x));
            atomicOr(&summary[h >> 5], 1u << (h & 31));
        }
}

__device__ __forceinline__ bool prune_keeps(const BatchArgs &a, const ItemRec &rec) {
    const uint32_t combo = item_combo(rec.info), v = item_value(rec.info);
    if (rec.info == kDeadItem) return false;  // padding between the chunks of a region
    if (a.keep_all != 0 || combo == kComboMarker || !combo_needs_member(combo)) return true;
    return ((a.adder_bitmap[v >> 5] >> (v & 31)) & 1u) != 0;
}

// One workgroup per region.  Pass 1 decides every item (one 64-bit keep mask per 64 items, several item loads and set lookups in
// flight per wavefront) and counts; a small kernel turns the regions' counts into offsets; pass 2 reads the masks and moves only the
// kept items (6 % on the benchmark's traffic: 13 M items per Gsample are decided, 0.8 M move).
#ifndef RSB_PRUNE_UNROLL
#define RSB_PRUNE_UNROLL 4
#endif
constexpr int kPruneUnroll = RSB_PRUNE_UNROLL;

__device__ __forceinline__ void prune_count_body(const BatchArgs &a) {
    if (blockIdx.x >= a.n_regions) return;  // (a launch for several batches: the grid is the largest batch's)
    __shared__ uint32_t wave_kept[kThreads / 64];
    __shared__ __attribute__((aligned(16))) uint32_t summary[kSummaryWords];
    static_assert(kSummaryWords == 4 * kThreads, "one 16-byte piece per thread");
    const bool summarised = a.adder_summary != nullptr;  // uniform over the grid
    if (summarised) {
        ((uint4 *)summary)[threadIdx.x] = ((const uint4 *)a.adder_summary)[threadIdx.x];
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = blockIdx.x;
    uint32_t n = a.region_counts[r];
    if (n > a.item_region) {
        if (threadIdx.x == 0) atomicAdd(&a.counters[kCntOverflow], 1u);
        n = a.item_region;
    }
    const ItemRec *items = a.items + (size_t)r * a.item_region;
    uint64_t *masks = a.region_masks + (size_t)r * a.mask_stride;
    const uint32_t nchunks = (n + 63) / 64;
    uint32_t kept = 0;
    for (uint32_t c0 = wave * kPruneUnroll; c0 < nchunks; c0 += (kThreads / 64) * kPruneUnroll) {
        ItemRec rec[kPruneUnroll];
#pragma unroll
        for (int u = 0; u < kPruneUnroll; ++u) {
            const uint32_t i = (c0 + u) * 64 + lane;
            rec[u] = i < n ? items[i] : ItemRec{0, kDeadItem};
        }
#pragma unroll
        for (int u = 0; u < kPruneUnroll; ++u) {
            const uint32_t i = (c0 + u) * 64 + lane;
            bool keep;
            if (summarised) {
                const uint32_t combo = item_combo(rec[u].info), v = item_value(rec[u].info), h = summary_bit(v);
                keep = (combo == kComboMarker || !combo_needs_member(combo)) && rec[u].info != kDeadItem;
                if (!keep && combo_needs_member(combo) && ((summary[h >> 5] >> (h & 31)) & 1u)) keep = ((a.adder_bitmap[v >> 5] >> (v & 31)) & 1u) != 0;
            } else {
                keep = prune_keeps(a, rec[u]);
            }
            const uint64_t km = __ballot(i < n && keep);
            if (lane == 0 && c0 + u < nchunks) masks[c0 + u] = km;
            kept += (uint32_t)__popcll(km);
        }
    }
    if (lane == 0) wave_kept[wave] = kept;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < kThreads / 64; ++w) total += wave_kept[w];
        a.region_kept[r] = total;
    }
}

// The kept items leave in CHUNK order (= stream order; the chunks' runs lie wherever their wavefronts put them).  Per chunk: kept
// items = the popcounts of its keep masks; exclusive sums in two levels -- within groups of 256 chunks here, over the groups in
// k_prune_offsets.
constexpr int kChunkGroup = 256;
// the keep mask word that holds item slot `slot` (item_region % 64 == 0: region r's words start at r * mask_stride)
__device__ __forceinline__ const uint64_t *slot_mask_word(const BatchArgs &a, uint32_t slot) { return a.region_masks + (slot >> 6); }
// the bits of mask word number w (counted from the word that holds the chunk's first slot) that belong to the chunk [first, first + n)
__device__ __forceinline__ uint64_t chunk_word_bits(uint32_t first, uint32_t n, uint32_t w) {
    const uint32_t lo = first & 63u, end = lo + n;  // the chunk's slots, counted from its first word's bit 0
    uint64_t bits = ~0ull;
    if (w == 0) bits &= ~0ull << lo;
    if (end < 64u * (w + 1)) bits &= end > 64u * w ? ~0ull >> (64u * (w + 1) - end) : 0ull;
    return bits;
}

__device__ __forceinline__ void prune_chunks_body(const BatchArgs &a) {
    if (blockIdx.x * kChunkGroup >= a.n_chunks) return;
    __shared__ uint32_t wave_sum[kChunkGroup / 64], wave_all[kChunkGroup / 64];
    const uint32_t c = blockIdx.x * kChunkGroup + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t kept = 0, all = 0;
    if (c < a.n_chunks) {
        const uint2 d = a.chunk_dir[c];
        const uint64_t *masks = slot_mask_word(a, d.x);
        all = d.y;
        const uint32_t nwords = d.y ? ((d.x & 63u) + d.y + 63u) / 64u : 0u;
        for (uint32_t w = 0; w < nwords; ++w) kept += (uint32_t)__popcll(masks[w] & chunk_word_bits(d.x, d.y, w));
    }
    uint32_t incl = kept, incl_all = all;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64), up_all = (uint32_t)__shfl_up((int)incl_all, off, 64);
        if ((int)lane >= off) {
            incl += up;
            incl_all += up_all;
        }
    }
    if (lane == 63) {
        wave_sum[wave] = incl;
        wave_all[wave] = incl_all;
    }
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += wave_sum[w];
    if (c < a.n_chunks) a.chunk_kept[c] = before + incl - kept;
    if (threadIdx.x == kChunkGroup - 1) {
        uint32_t total_all = 0;
        for (uint32_t w = 0; w < kChunkGroup / 64; ++w) total_all += wave_all[w];
        a.chunk_group[blockIdx.x] = before + incl;
        atomicAdd(&a.counters[kCntItemsTotal], total_all);  // a statistic (one atomic per 256 chunks)
    }
}

// exclusive sum over the groups' totals (one workgroup; a Gsample has a few hundred groups); the total goes to the counters
__device__ __forceinline__ void prune_offsets_body(const BatchArgs &a) {
    __shared__ uint32_t part[1024];
    const uint32_t ngroups = (a.n_chunks + kChunkGroup - 1) / kChunkGroup;
    const uint32_t t = threadIdx.x, per = (ngroups + 1023) / 1024, lo = min(ngroups, t * per), hi = min(ngroups, lo + per);
    uint32_t sum = 0;
    for (uint32_t r = lo; r < hi; ++r) sum += a.chunk_group[r];
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t add = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;  // exclusive
    for (uint32_t r = lo; r < hi; ++r) {
        const uint32_t k = a.chunk_group[r];
        a.chunk_group[r] = run;
        run += k;
    }
    if (t == 1023) a.counters[kCntSurvItems] = part[1023];
}

// Eight lanes per chunk: its kept items to their places.  Sub-lane l takes keep mask word l of the chunk (a chunk of two strips has
// two or three), so the words and then the kept items of eight chunks are fetched side by side in a wavefront.
#ifndef RSB_WRITE_LANES
#define RSB_WRITE_LANES 8
#endif
constexpr int kWriteLanes = RSB_WRITE_LANES;
__device__ __forceinline__ void prune_write_body(const BatchArgs &a) {
    const uint32_t sub = threadIdx.x & (kWriteLanes - 1), c = (blockIdx.x * kThreads + threadIdx.x) / kWriteLanes;
    const bool live = c < a.n_chunks;
    const uint2 d = live ? a.chunk_dir[c] : make_uint2(0, 0);
    uint32_t base_slot = live ? a.chunk_group[c / kChunkGroup] + a.chunk_kept[c] : 0u;
    const uint32_t nwords = d.y ? ((d.x & 63u) + d.y + 63u) / 64u : 0u;
    const ItemRec *items = a.items + (d.x & ~63u);
    const uint64_t *masks = slot_mask_word(a, d.x);
    // (the loop bound is the largest of the wavefront's eight chunks: the cross-lane steps below need every lane)
    uint32_t most = nwords;
    for (int off = kWriteLanes; off < 64; off <<= 1) most = max(most, (uint32_t)__shfl_xor((int)most, off, 64));
    for (uint32_t w0 = 0; w0 < most; w0 += kWriteLanes) {
        const uint32_t w = w0 + sub;
        uint64_t km = w < nwords ? masks[w] & chunk_word_bits(d.x, d.y, w) : 0ull;
        const uint32_t cnt = (uint32_t)__popcll(km);
        uint32_t incl = cnt;
        for (int off = 1; off < kWriteLanes; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, off, kWriteLanes);
            if ((int)sub >= off) incl += up;
        }
        uint32_t slot = base_slot + incl - cnt;
        while (km) {  // four kept items at a time: their loads fly together
            ItemRec rec[4];
            bool have[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                have[u] = km != 0;
                const uint32_t bit = have[u] ? (uint32_t)__builtin_ctzll(km) : 0u;
                km &= km - 1;  // (0 stays 0)
                if (have[u]) rec[u] = items[(size_t)w * 64 + bit];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (have[u] && slot + u < a.surv_item_cap) {
                    a.skeys[0][slot + u] = ((uint64_t)rec[u].g << 3) | item_phase(rec[u].info);
                    a.sinfo[0][slot + u] = rec[u].info;
                }
            slot += 4;
        }
        base_slot += (uint32_t)__shfl((int)incl, kWriteLanes - 1, kWriteLanes);
    }
}

// ---- The ordered prune in ONE pass (round 3, late).  The four kernels above read the items, wrote a keep mask per 64 of them, summed
// the masks per chunk and per group of chunks, and read masks and kept items again: 41 + 13 + 5 + 47 us per Gsample.  Here a workgroup
// takes a TILE of 32 consecutive chunks, eight lanes per chunk: it decides and counts its items, learns how many kept items lie
// before the tile by looking back over the tiles in front of it (each publishes first its own count, then its running total, in one
// 64-bit word: flag, kept items, all items), and then takes the items once more -- from the cache -- and puts the kept ones in place.
// A tile's number is its workgroup's: workgroups start in ascending order (per XCD, which take them in turn), so the lowest unfinished
// tile is always resident and waits for nobody.  (Numbers from a counter, as rocPRIM's look-back scan takes them, cost more than the
// whole prune: 6000 atomics on one address, one after the other.)  The wait is bounded all the same: a tile that gives up raises
// kCntOverflow by kPruneStallUnit, the batch ends with RSB_EOVERFLOW (and a message that says so) instead of a hung queue.
#ifndef RSB_TILE_THREADS
#define RSB_TILE_THREADS 1024
#endif
constexpr int kTileThreads = RSB_TILE_THREADS, kTileChunks = kTileThreads / 8;
constexpr uint64_t kTileOwn = 1ull << 62, kTileTotal = 2ull << 62;
constexpr uint32_t kTileGaveUp = kPruneStallUnit;  // what a tile that gave up adds to kCntOverflow (the host tells it from full item regions)
#ifndef RSB_TILE_LOADS
#define RSB_TILE_LOADS 4
#endif
constexpr int kTileLoads = RSB_TILE_LOADS;  // items per lane and round (one 64-byte run)
constexpr int kVerdictRounds = 32 / kTileLoads, kHeadRounds = 3;
static_assert(kHeadRounds <= kVerdictRounds, "the head's verdicts fit the register");
__device__ __forceinline__ uint64_t tile_pack(uint32_t kept, uint32_t all) { return ((uint64_t)kept << 31) | all; }

// the verdicts on kTileLoads items as a bit mask: first who has to ask the bitmap at all, then those reads together, then the answers
template <int N>
__device__ __forceinline__ uint32_t tile_keeps(const BatchArgs &a, const uint32_t *summary, bool summarised, const ItemRec (&rec)[N]) {
    uint32_t keep = 0, ask = 0, word[N];
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const uint32_t combo = item_combo(rec[u].info), v = item_value(rec[u].info), h = summary_bit(v);
        const bool dead = rec[u].info == kDeadItem;  // padding between the chunks of a region
        const bool free = a.keep_all != 0 || combo == kComboMarker || !combo_needs_member(combo);
        keep |= (!dead && free ? 1u : 0u) << u;
        ask |= (!dead && !free && (!summarised || ((summary[h >> 5] >> (h & 31)) & 1u)) ? 1u : 0u) << u;
    }
#pragma unroll
    for (int u = 0; u < N; ++u) word[u] = (ask >> u) & 1u ? a.adder_bitmap[item_value(rec[u].info) >> 5] : 0u;
#pragma unroll
    for (int u = 0; u < N; ++u) keep |= ((ask >> u) & (word[u] >> (item_value(rec[u].info) & 31)) & 1u) << u;
    return keep;
}

__device__ __forceinline__ void prune_onepass_body(const BatchArgs &a) {
    __shared__ __attribute__((aligned(16))) uint32_t summary[kSummaryWords];
    __shared__ uint32_t s_before, wave_kept[kTileThreads / 64], wave_all[kTileThreads / 64];
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6, sub = t & 7;
    const uint32_t ntiles = (a.n_chunks + kTileChunks - 1) / kTileChunks, tile = blockIdx.x;
    if (tile >= ntiles) return;  // (a launch for several batches: the grid is the largest batch's)
    uint64_t *status = a.region_masks;  // (the keep masks' memory: this prune writes none)
    const bool summarised = a.adder_summary != nullptr;  // uniform over the grid
    if (summarised && t < kSummaryWords / 4) ((uint4 *)summary)[t] = ((const uint4 *)a.adder_summary)[t];
    __syncthreads();
    if (tile == 0) {  // the scan clamps what it writes into a full region and goes on counting: that batch is incomplete
        bool over = false;
        for (uint32_t r = t; r < a.n_regions; r += kTileThreads) over |= a.region_counts[r] > a.item_region;
        if (over) atomicAdd(&a.counters[kCntOverflow], 1u);
    }
    const uint32_t c = tile * kTileChunks + (t >> 3);
    const uint2 d = c < a.n_chunks ? a.chunk_dir[c] : make_uint2(0, 0);
    const ItemRec *items = a.items + d.x;
    // the lane's items of round k: k * 8 * kTileLoads + sub + 8 u (a load instruction takes a run of eight items per chunk); the verdicts of
    // the first rounds stay in a register
    // The first kHeadRounds rounds (96 items: most chunks end there) go as one: their loads together, then the verdicts, then the
    // bitmap reads of all of them -- two trips to memory where round after round made six.
    uint32_t kept = 0, verdicts = 0, round = kHeadRounds;
    {
        ItemRec rec[kHeadRounds * kTileLoads];
#pragma unroll
        for (int u = 0; u < kHeadRounds * kTileLoads; ++u) rec[u] = sub + 8 * u < d.y ? items[sub + 8 * u] : ItemRec{0, kDeadItem};
        verdicts = tile_keeps(a, summary, summarised, rec);
        kept = (uint32_t)__popc(verdicts);
    }
    for (uint32_t k0 = sub + 8 * kHeadRounds * kTileLoads; k0 < d.y; k0 += 8 * kTileLoads, ++round) {
        ItemRec rec[kTileLoads];
#pragma unroll
        for (int u = 0; u < kTileLoads; ++u) rec[u] = k0 + 8 * u < d.y ? items[k0 + 8 * u] : ItemRec{0, kDeadItem};
        const uint32_t keep = tile_keeps(a, summary, summarised, rec);
        kept += (uint32_t)__popc(keep);
        if (round < (uint32_t)kVerdictRounds) verdicts |= keep << (round * kTileLoads);
    }
    // the chunk's count (in all eight lanes), the chunks before it in the wavefront, the wavefronts before it in the tile
    uint32_t chunk_keeps = kept;
    for (int off = 1; off < 8; off <<= 1) chunk_keeps += (uint32_t)__shfl_xor((int)chunk_keeps, off, 64);
    uint32_t incl = chunk_keeps, incl_all = d.y;
    for (int off = 8; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64), up_all = (uint32_t)__shfl_up((int)incl_all, off, 64);
        if ((int)lane >= off) {
            incl += up;
            incl_all += up_all;
        }
    }
    if (lane == 63) {
        wave_kept[wave] = incl;
        wave_all[wave] = incl_all;
    }
    __syncthreads();
    if (wave == 0) {  // the look-back, 64 tiles at a time
        uint32_t own_kept = 0, own_all = 0;
        for (uint32_t w = 0; w < kTileThreads / 64; ++w) {
            own_kept += wave_kept[w];
            own_all += wave_all[w];
        }
        if (tile && lane == 0) __hip_atomic_store(&status[tile], kTileOwn | tile_pack(own_kept, own_all), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t run_kept = 0, run_all = 0, spins = 0;
        bool gave_up = false;
        for (int64_t p = (int64_t)tile - 1; p >= 0;) {
            const int64_t q = p - (int64_t)lane;
            const uint64_t st = q >= 0 ? __hip_atomic_load(&status[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kTileTotal;  // (in front of tile 0: nothing)
            const uint64_t totals = __ballot((st >> 62) == 2), blanks = __ballot((st >> 62) == 0);
            const uint32_t upto = (uint32_t)__builtin_ctzll(totals | (1ull << 63));  // the nearest running total, or all 64
            const uint64_t span = upto == 63 ? ~0ull : (2ull << upto) - 1ull;
            if (blanks & span) {  // someone in between has not published yet
                if (++spins > (1u << 16)) {  // (a load is about a microsecond: tens of milliseconds, a thousand times what a tile takes)
                    gave_up = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            uint32_t k = (span >> lane) & 1ull ? (uint32_t)(st >> 31) & 0x7fffffffu : 0u, al = (span >> lane) & 1ull ? (uint32_t)st & 0x7fffffffu : 0u;
            for (int off = 32; off > 0; off >>= 1) {
                k += (uint32_t)__shfl_xor((int)k, off, 64);
                al += (uint32_t)__shfl_xor((int)al, off, 64);
            }
            run_kept += k;
            run_all += al;
            if (totals & span) break;
            p -= 64;
        }
        if (lane == 0) {
            __hip_atomic_store(&status[tile], kTileTotal | tile_pack(run_kept + own_kept, run_all + own_all), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (gave_up) atomicAdd(&a.counters[kCntOverflow], kTileGaveUp);
            if (tile == ntiles - 1) {
                a.counters[kCntSurvItems] = run_kept + own_kept;
                a.counters[kCntItemsTotal] = run_all + own_all;
            }
            s_before = run_kept;
        }
    }
    __syncthreads();
    uint32_t slot = s_before + incl - chunk_keeps;  // the chunk's first place
    for (uint32_t w = 0; w < wave; ++w) slot += wave_kept[w];
    // once more, in order: round after round, within a round load after load, within a load lane after lane.  (The rounds of a wavefront
    // run to its longest chunk: the shuffles need every lane.  Only rounds with a kept item read their items again.)
    uint32_t most = d.y;
    for (int off = 8; off < 64; off <<= 1) most = max(most, (uint32_t)__shfl_xor((int)most, off, 64));
    ItemRec head[kHeadRounds * kTileLoads];
#pragma unroll
    for (int u = 0; u < kHeadRounds * kTileLoads; ++u)
        if ((verdicts >> u) & 1u) head[u] = items[sub + 8 * u];
    round = 0;
    for (uint32_t r0 = 0; r0 < most; r0 += 8 * kTileLoads, ++round) {
        const uint32_t k0 = r0 + sub;
        ItemRec rec[kTileLoads];
        uint32_t keep;
        if (round < (uint32_t)kHeadRounds) {
            keep = (verdicts >> (round * kTileLoads)) & ((1u << kTileLoads) - 1u);
#pragma unroll
            for (int h = 0; h < kHeadRounds; ++h)  // (static indices into head[]: a register array)
                if (round == (uint32_t)h) {
#pragma unroll
                    for (int u = 0; u < kTileLoads; ++u) rec[u] = head[h * kTileLoads + u];
                }
        } else if (round < (uint32_t)kVerdictRounds) {
            keep = (verdicts >> (round * kTileLoads)) & ((1u << kTileLoads) - 1u);
#pragma unroll
            for (int u = 0; u < kTileLoads; ++u)
                if ((keep >> u) & 1u) rec[u] = items[k0 + 8 * u];
        } else {
#pragma unroll
            for (int u = 0; u < kTileLoads; ++u) rec[u] = k0 + 8 * u < d.y ? items[k0 + 8 * u] : ItemRec{0, kDeadItem};
            keep = tile_keeps(a, summary, summarised, rec);
        }
        // load u of the eight lanes is one run of eight items: the kept ones of run 0 come first, then run 1's ...; the four counts
        // travel through the eight-lane sum as the bytes of one word
        static_assert(kTileLoads <= 8, "one byte per load in a 64-bit word");
        unsigned long long mine = 0;
#pragma unroll
        for (int u = 0; u < kTileLoads; ++u) mine |= (unsigned long long)((keep >> u) & 1u) << (8 * u);
        unsigned long long upto = mine;
        for (int off = 1; off < 8; off <<= 1) {
            const unsigned long long up = __shfl_up(upto, off, 8);
            if ((int)sub >= off) upto += up;
        }
        const unsigned long long all = __shfl(upto, 7, 8), before = upto - mine;
#pragma unroll
        for (int u = 0; u < kTileLoads; ++u) {
            const uint32_t at = slot + ((uint32_t)(before >> (8 * u)) & 0xffu);
            if (((keep >> u) & 1u) && at < a.surv_item_cap) {
                a.skeys[0][at] = ((uint64_t)rec[u].g << 3) | item_phase(rec[u].info);
                a.sinfo[0][at] = rec[u].info;
            }
            slot += (uint32_t)(all >> (8 * u)) & 0xffu;
        }
    }
}
__global__ __launch_bounds__(kTileThreads) void k_prune_onepass(const BatchArgs a) { prune_onepass_body(a); }
// in front of it: the tiles' words and the ticket back to zero (and, for batches that use it, the summary's words: the memset's place)
__device__ __forceinline__ void prune_tiles_clear_body(const BatchArgs &a) {
    const uint32_t ntiles = (a.n_chunks + kTileChunks - 1) / kTileChunks;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < ntiles; i += gridDim.x * kThreads) a.region_masks[i] = 0;
    static_assert(kSummaryWords == 4 * kThreads, "one 16-byte piece per thread");
    if (blockIdx.x == 0 && a.adder_summary) ((uint4 *)a.adder_summary)[threadIdx.x] = make_uint4(0, 0, 0, 0);
}
__global__ __launch_bounds__(kThreads) void k_prune_tiles_clear(const BatchArgs a) { prune_tiles_clear_body(a); }

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
        for (uint32_t k = lane; k < (uint32_t)kHalo; k += 64) m[k] = (uint16_t)(stream_mag<FMT>(a, (int64_t)g + k) ^ kBias);
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
            const uint64_t mag = m[19 + k] ^ kBias, v = mag * mag;
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
// device replay: the sequential rules of demod_2400.c:289-374 on the sorted kept items

// scoreModesMessage's result for an item given the filter's answer (mode_s.c:336-407)
__device__ __forceinline__ int combo_score(uint32_t combo, bool member) {
    if (combo <= kComboApLong) return member ? 1000 : -1;
    if (combo == kComboCommB) return member ? 1000 : -2;
    if (combo <= kComboDf11Iid0 + 1) return (member ? 1600 : 750) / (int)(combo - kComboDf11Iid0 + 1);
    if (combo <= kComboDf11IidX + 1) return member ? 1000 / (int)(combo - kComboDf11IidX + 1) : -1;
    const uint32_t sub = (combo - kComboDf17) % 5u;  // 0: clean, 1/2: one bit, 3/4: two bits
    return (member ? 1800 : 1400) / (int)((sub + 1) / 2 + 1);
}
// does decodeModesMessage accept the winning phase (mode_s.c:461-466, 495-498, 524-526, 541-550)?
__device__ __forceinline__ bool combo_decodes(uint32_t combo, bool member) {
    if (combo <= kComboCommB) return member;
    if (combo == kComboDf11Iid0 || combo == kComboDf11IidX) return true;
    if (combo == kComboDf11Iid0 + 1 || combo == kComboDf11IidX + 1) return member;
    const uint32_t sub = (combo - kComboDf17) % 5u;
    return (sub == 2 || sub == 4) ? member : true;  // repaired AA differs from the received one: must be known
}
__device__ __forceinline__ bool combo_is_short(uint32_t combo) { return combo == kComboApShort || (combo >= kComboDf11Iid0 && combo <= kComboDf11IidX + 1); }
__device__ __forceinline__ bool combo_adds(uint32_t combo) { return combo == kComboDf11Iid0 || combo == kComboDf17; }  // mode_s.c:717-726

// The number of kept items is the prune's result and stays on the device (counters[kCntSurvItems]): the replay kernels are
// enqueued behind the prune without the host learning it first, with grids sized from the host's bound r.n_items.
__device__ __forceinline__ uint32_t items_of(const ReplayArgs &r) { return r.n_dev ? min(*r.n_dev, r.n_cap) : r.n_items; }

// first item of every 131072-sample block in the sorted item list (blk_start[nblocks] = n_items); also clears the counters the
// replay's later kernels add to
__device__ __forceinline__ void replay_bounds_body(const ReplayArgs &r, uint32_t block, uint32_t nblocks, uint32_t nthreads) {
    const uint32_t n_items = items_of(r);
    if (block == 0 && threadIdx.x < 6 && r.counters) {
        static_assert(kCntNewAdders == kCntAccepted + 1 && kCntAccE0 == kCntAccepted + 2 && kCntAccE2 == kCntAccepted + 4, "one run of counters");
        if (threadIdx.x < 5) r.counters[kCntAccepted + threadIdx.x] = 0;
        else r.counters[kCntChanged] = 0;
    }
    for (uint32_t i = block * nthreads + threadIdx.x; i <= n_items; i += nblocks * nthreads) {
        const int64_t prev = i == 0 ? -1 : (int64_t)min((uint64_t)r.nblocks, (r.keys[i - 1] >> 3) / r.block_len);
        const int64_t cur = i == n_items ? (int64_t)r.nblocks : (int64_t)min((uint64_t)r.nblocks, (r.keys[i] >> 3) / r.block_len);
        for (int64_t b = prev + 1; b <= cur; ++b) r.blk_start[b] = i;
    }
}

// One thread per item: score it against the current first-add table; the thread on the first item of an offset picks
// that offset's best phase by strict '>' (demod_2400.c:290-301; at most five items, one per trial phase) and asks whether
// decodeModesMessage would take it.  cand[i] = 0, or for such an offset: (best score + 1) | best item's distance << 16 |
// (112-bit frame) << 20.  Fully parallel: the sequential part is left to k_replay_walk.
// Every thread scores ITS item (key, info, then the table entry: two round trips to memory) and leaves the result in LDS, where the
// offset's first thread reads its up to four followers; the workgroup's last offsets reach into the next workgroup's items, which
// four threads score as well.  (One thread walking the five items by itself was fifteen dependent round trips: 41 us per Gsample.)
#ifndef RSB_SCORE_SERIAL
#define RSB_SCORE_SERIAL 0
#endif
// an item's score given the first-add table's entry for its address: (score + 4) | decodeModesMessage takes it << 12 | 112-bit frame << 13
// (score -3 .. 1800: never 0, which stands for a marker item)
__device__ __forceinline__ uint32_t score_word(uint32_t info, uint32_t at, uint32_t first_add) {
    const uint32_t combo = item_combo(info);
    const bool member = first_add <= at;  // added by a message at an earlier offset
    return (uint32_t)(combo_score(combo, member) + 4) | (combo_decodes(combo, member) ? 1u << 12 : 0u) | (combo_is_short(combo) ? 0u : 1u << 13);
}
__device__ __forceinline__ void replay_score_body(const ReplayArgs &r) {
    const uint32_t n_items = items_of(r);
    if (blockIdx.x == 0 && threadIdx.x == 0 && r.counters) r.counters[kCntChanged] = 0;  // (k_replay_compare of this round raises it)
#if RSB_SCORE_SERIAL
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n_items; i += gridDim.x * kThreads) {
        r.acc[i] = 0;
        const uint32_t g = (uint32_t)(r.keys[i] >> 3);
        uint32_t out = 0;
        if (i == 0 || (uint32_t)(r.keys[i - 1] >> 3) != g) {
            int best = -3;
            uint32_t best_off = 0, best_info = 0;
            bool best_member = false;
#pragma unroll
            for (uint32_t d = 0; d < 5; ++d) {
                const uint32_t e = i + d;
                if (e >= n_items || (d && (uint32_t)(r.keys[e] >> 3) != g)) break;
                const uint32_t info = r.info[e], combo = item_combo(info);
                if (combo == kComboMarker) continue;
                const bool member = r.t_cur[item_value(info)] <= g;  // added by a message at an earlier offset
                const int sc = combo_score(combo, member);
                if (sc > best) {
                    best = sc;
                    best_off = d;
                    best_info = info;
                    best_member = member;
                }
            }
            if (best >= 0 && combo_decodes(item_combo(best_info), best_member))
                out = ((uint32_t)best + 1u) | (best_off << 16) | (combo_is_short(item_combo(best_info)) ? 0u : 1u << 20);
        }
        r.cand[i] = out;
    }
#else
    // per item in LDS: its offset, and (score + 4) | takes << 12 | long << 13, 0 for a marker item (score -3 .. 1800: never 0 otherwise)
    __shared__ uint32_t s_g[kThreads + 4], s_w[kThreads + 4];
    const uint32_t t = threadIdx.x;
    for (uint32_t base = blockIdx.x * kThreads; base < n_items; base += gridDim.x * kThreads) {  // (uniform over the workgroup)
        const uint32_t i = base + t, j = base + kThreads + t;  // j: the four items behind the workgroup's (threads 0-3)
        const bool own = i < n_items, tail = t < 4 && j < n_items;
        uint64_t key = 0, key_t = 0, key_before = ~0ull;
        uint32_t info = 0, info_t = 0;
        if (own) {
            key = r.keys[i];
            info = r.info[i];
        }
        if (tail) {
            key_t = r.keys[j];
            info_t = r.info[j];
        }
        if (t == 0 && base) key_before = r.keys[base - 1];
        const uint32_t g = (uint32_t)(key >> 3), g_t = (uint32_t)(key_t >> 3);
        const bool live = own && item_combo(info) != kComboMarker, live_t = tail && item_combo(info_t) != kComboMarker;
        uint32_t first = kNever, first_t = kNever;
        if (live) first = r.t_cur[item_value(info)];
        if (live_t) first_t = r.t_cur[item_value(info_t)];
        if (own) r.acc[i] = 0;
        s_g[t] = own ? g : kNever;
        s_w[t] = live ? score_word(info, g, first) : 0u;
        if (t < 4) {
            s_g[kThreads + t] = tail ? g_t : kNever;
            s_w[kThreads + t] = live_t ? score_word(info_t, g_t, first_t) : 0u;
        }
        __syncthreads();
        if (own) {
            uint32_t out = 0;
            const bool starts = t ? s_g[t - 1] != g : (base == 0 || (uint32_t)(key_before >> 3) != g);
            if (starts) {
                int best = -3;
                uint32_t best_off = 0, best_w = 0;
#pragma unroll
                for (uint32_t d = 0; d < 5; ++d) {
                    if (i + d >= n_items || s_g[t + d] != g) break;  // (d == 0: the item itself)
                    const uint32_t w = s_w[t + d];
                    if (w == 0) continue;
                    const int sc = (int)(w & 0xfffu) - 4;
                    if (sc > best) {
                        best = sc;
                        best_off = d;
                        best_w = w;
                    }
                }
                if (best >= 0 && (best_w & (1u << 12))) out = ((uint32_t)best + 1u) | (best_off << 16) | ((best_w >> 13) << 20);
            }
            r.cand[i] = out;
        }
        __syncthreads();
    }
#endif
}

// One wavefront per 131072-sample block (one demodulate2400 call): the offsets that could be accepted, 64 items at a
// time, are chained in ascending order through the greedy skip (demod_2400.c:371) -- the only sequential part.  The candidates go
// through an LDS queue (offset, where the scan goes on if taken); the chain reads them eight at a time ahead of the comparisons and
// keeps its verdicts in a 64-bit mask -- a store per step (round 2's form) made every step wait for its own LDS reads.  (The chain on
// the scalar unit, v_readlane of the candidates' lanes in ballot order: 33 us against 27.)
__device__ __forceinline__ void replay_walk_body(const ReplayArgs &r) {
    __shared__ uint2 cand_q[kThreads / 64][64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x * (kThreads / 64) + wave;
    if (b >= r.nblocks) return;  // whole wavefronts leave; no workgroup barrier below
    const uint32_t lo = r.blk_start[b], hi = r.blk_start[b + 1];
    uint32_t next_j = 0, accepted = 0;  // wave-uniform
    // the next 64 items are loaded while the current ones are chained
    uint32_t c_next = 0, g_next = 0;
    if (lo + lane < hi) {
        c_next = r.cand[lo + lane];
        g_next = (uint32_t)(r.keys[lo + lane] >> 3);
    }
    for (uint32_t base = lo; base < hi; base += 64) {
        const uint32_t i = base + lane, c = c_next, g = g_next;
        c_next = 0;
        if (i + 64 < hi) {
            c_next = r.cand[i + 64];
            g_next = (uint32_t)(r.keys[i + 64] >> 3);
        }
        const bool cand = c != 0;
        const uint64_t cm = __ballot(cand);
        const uint32_t pos = lanes_below(cm), nc = (uint32_t)__popcll(cm);
        wave_sync();
        if (cand) cand_q[wave][pos] = make_uint2(g, g + ((c >> 20) ? 268u : 134u) + 1);
        wave_sync();
        uint64_t taken = 0;  // by queue position
        for (uint32_t k0 = 0; k0 < nc; k0 += 8) {  // every lane runs the same chain
            uint2 q[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) q[u] = cand_q[wave][(k0 + u) & 63u];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                const bool take = k0 + u < nc && q[u].x >= next_j;
                next_j = take ? q[u].y : next_j;
                taken |= (uint64_t)take << (k0 + u);
            }
        }
        if (cand && ((taken >> pos) & 1ull)) {
            r.acc[i + ((c >> 16) & 15u)] = c & 0xffffu;
            r.acc_rank[i + ((c >> 16) & 15u)] = accepted + (uint32_t)__popcll(taken & ((1ull << pos) - 1ull));  // k_replay_emit: the record's place = the block's base + this
        }
        accepted += (uint32_t)__popcll(taken);
    }
    if (lane == 0) r.blk_counts[b] = accepted;
}

// accepted messages before every 131072-sample block (blk_base[nblocks] = all of the batch's, also to counters[kCntAccepted])
__device__ __forceinline__ void replay_blkbase_body(const ReplayArgs &r) {
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x, per = (r.nblocks + 1023) / 1024, lo = min(r.nblocks, t * per), hi = min(r.nblocks, lo + per);
    if (t < 4) r.counters[kCntNewAdders + t] = 0;  // new adders, accepted by errors: k_replay_list / k_replay_emit count them (a repeated round starts over)
    uint32_t sum = 0;
    for (uint32_t b = lo; b < hi; ++b) sum += r.blk_counts[b];
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t add = t >= d ? part[t - d] : 0u;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (uint32_t b = lo; b < hi; ++b) {
        r.blk_base[b] = run;
        run += r.blk_counts[b];
    }
    if (t == 1023) {
        r.blk_base[r.nblocks] = part[1023];
        r.counters[kCntAccepted] = part[1023];
    }
}

// min(offset + 1) per address over the items that add their address to the filter (mode_s.c:717-726), into a "first added at"
// table.  Thousands of messages share an aircraft's address, and same-address global atomics retire one after the other; so a
// workgroup first folds its run of the (sorted) item list into an open-addressed table in LDS and then issues one global
// atomicMin per distinct address it saw.
//   SEED: every item that COULD add its address (the optimistic starting point of the fixed-point iteration: the first clean
//         DF17 / DF11 of an aircraft is accepted unless it lies in another message's skip window, so the first walk usually
//         reproduces this table and the iteration ends after one round.  Any fixed point is the reference's result -- by induction
//         over the offsets: what happens at offset g depends only on entries below g -- so the start affects the rounds only);
//   else: the items the walk accepted, into the next table.
#ifndef RSB_MIN_ITEMS
#define RSB_MIN_ITEMS 8192  // eight per thread of a 1024-thread workgroup.  Where the time goes (ablations, 0.8 M items, 40 new aircraft, per pass):
                            // launch + loads 5 us, the LDS table 8 us, the global atomicMin 16 us -- 400 workgroups of 256 threads each
                            // sent one per aircraft, and same-address atomics are served one after the other at the memory side
                            // (~40 ns).  A quarter of the workgroups, four times the threads: 28 -> 21.5 us.  LDS atomics are not it
                            // (plain reads in front of them: no change); nor are idle workgroups of the oversized grid.
#endif
#ifndef RSB_MIN_BATCH_ITEMS
#define RSB_MIN_BATCH_ITEMS 1  // a thread's items loaded together: -3 us per pass (the same for the flush of the LDS table: +17 us, not kept)
#endif
#ifndef RSB_MIN_PLAIN_READS
#define RSB_MIN_PLAIN_READS 1
#endif
#ifndef RSB_MIN_THREADS
#define RSB_MIN_THREADS 1024
#endif
#ifndef RSB_MIN_SLOTS
#define RSB_MIN_SLOTS 4096
#endif
constexpr int kMinSlots = RSB_MIN_SLOTS, kMinThreads = RSB_MIN_THREADS, kMinItemsPerGroup = RSB_MIN_ITEMS;
template <bool SEED>
__device__ __forceinline__ void replay_table_min_body(const ReplayArgs &r, uint32_t nruns) {  // nruns: the workgroups that take runs (they stride by it)
    __shared__ uint32_t key[kMinSlots], val[kMinSlots];
    uint32_t *table = SEED ? r.t_cur : r.t_next;
    // (the grid is sized from the host's bound on the kept items, ten times the count on ordinary traffic: most workgroups have no run)
    if (blockIdx.x * (uint32_t)kMinItemsPerGroup >= items_of(r)) return;
    for (uint32_t i = threadIdx.x; i < (uint32_t)kMinSlots; i += kMinThreads) {
        key[i] = kNever;
        val[i] = kNever;
    }
    __syncthreads();
    const uint32_t n_items = items_of(r);
    for (uint32_t lo = blockIdx.x * (uint32_t)kMinItemsPerGroup; lo < n_items; lo += nruns * (uint32_t)kMinItemsPerGroup) {  // (one run per workgroup when the host's bound holds)
        const uint32_t hi = min(n_items, lo + (uint32_t)kMinItemsPerGroup);
#if RSB_MIN_BATCH_ITEMS
        // (a thread's items are all loaded before the first of them goes to the LDS table: one round trip to memory instead of one per item)
        constexpr int kPer = kMinItemsPerGroup / kMinThreads;
        static_assert(kMinItemsPerGroup % kMinThreads == 0 && kMinSlots % kMinThreads == 0, "whole rounds of the workgroup");
        uint32_t info[kPer], g1[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const uint32_t i = lo + threadIdx.x + (uint32_t)k * kMinThreads;
            info[k] = kDeadItem;
            g1[k] = 0;
            if (i < hi) {
                const uint32_t inf = r.info[i], acc = SEED ? 1u : (uint32_t)r.acc[i];
                g1[k] = (uint32_t)(r.keys[i] >> 3) + 1;
                if (acc != 0 && combo_adds(item_combo(inf))) info[k] = inf;
            }
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            if (info[k] == kDeadItem) continue;
            const uint32_t v = item_value(info[k]);
            uint32_t h = ((v * 0x9E3779B1u) >> 16) & (kMinSlots - 1);
            bool placed = false;
            for (int probe = 0; probe < 16 && !placed; ++probe) {
                // (plain reads first: a slot's key never changes once set, its value only falls -- a few dozen aircraft share a
                // run's 2048 items, and an LDS atomic takes its lanes one by one)
                uint32_t q = __atomic_load_n(&key[h], __ATOMIC_RELAXED);
                if (RSB_MIN_PLAIN_READS == 0 || q == kNever) q = atomicCAS(&key[h], kNever, v);
                if (q == kNever || q == v) {
                    if (RSB_MIN_PLAIN_READS == 0 || __atomic_load_n(&val[h], __ATOMIC_RELAXED) > g1[k]) atomicMin(&val[h], g1[k]);
                    placed = true;
                }
                h = (h + 1) & (kMinSlots - 1);
            }
            if (!placed && table[v] > g1[k]) atomicMin(&table[v], g1[k]);  // (a run with more distinct addresses than the table holds)
        }
#else
        for (uint32_t i = lo + threadIdx.x; i < hi; i += kMinThreads) {
            if (!SEED && r.acc[i] == 0) continue;
            const uint32_t info = r.info[i];
            if (!combo_adds(item_combo(info))) continue;
            const uint32_t v = item_value(info), g1 = (uint32_t)(r.keys[i] >> 3) + 1;
            uint32_t h = ((v * 0x9E3779B1u) >> 16) & (kMinSlots - 1);
            bool placed = false;
            for (int probe = 0; probe < 16 && !placed; ++probe) {
                const uint32_t k = atomicCAS(&key[h], kNever, v);
                if (k == kNever || k == v) {
                    atomicMin(&val[h], g1);
                    placed = true;
                }
                h = (h + 1) & (kMinSlots - 1);
            }
            if (!placed && table[v] > g1) atomicMin(&table[v], g1);  // (a run with more distinct addresses than the table holds)
        }
#endif
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < (uint32_t)kMinSlots; i += kMinThreads) {
            const uint32_t v = key[i];
            if (v != kNever && table[v] > val[i]) atomicMin(&table[v], val[i]);
            key[i] = kNever;
            val[i] = kNever;
        }
        __syncthreads();
    }
}

// over the items that can add an address: did the table move / reset / list / fold
__device__ __forceinline__ void replay_compare_body(const ReplayArgs &r, uint32_t *changed) {
    const uint32_t n_items = items_of(r);
    bool any = false;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n_items; i += gridDim.x * kThreads) {
        const uint32_t info = r.info[i];
        if (!combo_adds(item_combo(info))) continue;
        const uint32_t v = item_value(info);
        any |= r.t_cur[v] != r.t_next[v];
    }
    if (any) *changed = 1;
}
__global__ __launch_bounds__(kThreads) void k_replay_reset(const ReplayArgs r, uint32_t *table) {
    const uint32_t n_items = items_of(r);
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n_items; i += gridDim.x * kThreads) {
        const uint32_t info = r.info[i];
        if (!combo_adds(item_combo(info))) continue;
        const uint32_t v = item_value(info);
        if (table[v] != 0) table[v] = kNever;  // 0 = known before this batch: stays
    }
}
// message number -> item for the record kernel, and the batch's new filter members.
// CHECK (the first round of a chain): also "did the table move", without the next table.  t_next[v] would be the smallest offset + 1
// over the ACCEPTED items that add v; it equals t_cur[v] = c for every v iff no accepted item of v lies below c and -- where c names an
// item of this batch (neither 0 = known before nor kNever) -- the item at c is accepted (or a sibling at the same offset that adds
// the same address: the phases of one offset are neighbours in the list).  That saves the chain k_replay_table_min<false> and
// k_replay_compare; only a batch whose first round was not the fixed point builds the next table (launch_replay_next_table).
template <bool CHECK>
__device__ __forceinline__ void replay_list_body(const ReplayArgs &r, uint32_t *counters) {
    const uint32_t n_items = items_of(r);
    bool moved = false;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n_items; i += gridDim.x * kThreads) {
        const bool accepted = r.acc[i] != 0;
        if (!CHECK && !accepted) continue;
        const uint32_t info = r.info[i];
        const bool adds = combo_adds(item_combo(info));
        if (!accepted && !adds) continue;
        const uint32_t g = (uint32_t)(r.keys[i] >> 3);
        if (accepted) {
            // accepted messages before the item's block + its rank there
            const uint32_t place = r.blk_base[g / r.block_len] + r.acc_rank[i];
            if (place < n_items) r.acc_idx[place] = i;
        }
        if (!adds) continue;
        const uint32_t v = item_value(info), g1 = g + 1, c = r.t_cur[v];
        if (accepted) {
            if (c == g1) {
                const uint32_t slot = atomicAdd(&counters[kCntNewAdders], 1u);
                if (slot < r.new_adders_cap) r.new_adders[slot] = make_uint2(g1, v);
            } else if (CHECK && g1 < c) {
                moved = true;
            }
        } else if (CHECK && c == g1) {
            bool sibling = false;
            for (int d = -4; d <= 4; ++d) {
                const int64_t j = (int64_t)i + d;
                if (d == 0 || j < 0 || j >= (int64_t)n_items || (uint32_t)(r.keys[j] >> 3) != g || r.acc[j] == 0) continue;
                const uint32_t other = r.info[j];
                sibling |= combo_adds(item_combo(other)) && item_value(other) == v;
            }
            moved |= !sibling;
        }
    }
    if (CHECK && moved) counters[kCntChanged] = 1;
}
__device__ __forceinline__ void replay_fold_body(const ReplayArgs &r) {
    const uint32_t n_items = items_of(r);
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n_items; i += gridDim.x * kThreads) {
        const uint32_t info = r.info[i];
        if (!combo_adds(item_combo(info))) continue;
        const uint32_t v = item_value(info);
        if (r.t_cur[v] != kNever) {
            r.t_cur[v] = 0;
            r.t_next[v] = 0;
        }
    }
}
__global__ __launch_bounds__(kThreads) void k_filter_fill(uint32_t *t0, uint32_t *t1, const uint32_t *addrs, uint32_t n) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i == 0) t0[0] = t1[0] = 0;  // icaoFilterTest(0) hits an empty slot (icao_filter.c:101-107)
    if (i < n) t0[addrs[i]] = t1[addrs[i]] = 0;
}

// One LANE per accepted message: the lane streams the raw IQ from the dword that holds S[g + 19] (16-byte loads at
// 4-byte alignment, a few in flight), converts it in registers -- every row of two magnitudes sits at a fixed place in
// the instruction stream, so no LDS window is needed -- re-slices the winning phase with the scan kernel's per-lane
// coefficient slicer, repairs it (crc.c:410-418), fills in the fields of mode_s.c:424-562 and the signal level of
// demod_2400.c:342-354, and writes its 56-byte record.  LDS holds only the tables: 16 wavefronts per CU.
#ifndef RSB_EMIT_WAVES
#define RSB_EMIT_WAVES 8
#endif
#ifndef RSB_EMIT_FLIGHT
#define RSB_EMIT_FLIGHT 4
#endif
#ifndef RSB_AC_BITS_AT_ONCE
#define RSB_AC_BITS_AT_ONCE 5
#endif
#ifndef RSB_EMIT_BURST
#define RSB_EMIT_BURST 1
#endif
#ifndef RSB_EMIT_GROUPS
#define RSB_EMIT_GROUPS 256  // one workgroup per CU (180-235 VGPRs: one fits): a second round pays the set-up again, 33 us
#endif
constexpr int kEmitGroups = RSB_EMIT_GROUPS;
constexpr int kEmitWaves = RSB_EMIT_WAVES;   // per workgroup (16 capped the kernel at 128 VGPRs and made it spill); the grid holds kEmitGroups of them
constexpr int kEmitTrios = 12;       // 3 chunks = 12 rows (24 samples) = 2 groups of five bits per step; 24 groups >= 23

struct __attribute__((aligned(16))) EmitShared {
    uint16_t lut[kRadialTotal];
    uint32_t coef[10][16];
    uint32_t gsyn[(12 + 24) * 32];  // + one zero row: the 24th step
    SynLds syn;                      // --fix / --no-fix: the repair tables (as in the scan kernel)
    uint32_t rec[kEmitWaves][64 * 14];  // per wavefront: its 64 records, so that they leave as contiguous 16-byte stores
};

typedef uint4 __attribute__((aligned(4))) uint4_a4;

template <int FMT, bool DC = false>
__device__ __forceinline__ void replay_emit_body(const BatchArgs &a, const ReplayArgs &r) {
    static_assert(!DC || FMT != kFmtMag16, "the DC block inside the kernels: raw sample formats");
    extern __shared__ __attribute__((aligned(16))) unsigned char emit_lds[];  // 66 KB: over the static limit
    EmitShared &sh = *reinterpret_cast<EmitShared *>(emit_lds);
    constexpr uint32_t kT = 64 * kEmitWaves;
    if (FMT == kFmtUC8 && !DC) load_radial_table(sh.lut, a.tab.uc8_radial, kBias2, threadIdx.x, kT);  // the scan kernel's radial table
    for (uint32_t i = threadIdx.x; i < 160; i += kT) {
        const uint32_t row = i >> 4, e = i & 15;
        sh.coef[row][e] = e < 15 ? coef_entry((int)(row / 5), (int)(row % 5), (int)(e / 3), (int)(e % 3)) : 0u;
    }
    // the 112 single-bit syndromes go through LDS first (sh.rec is free until the first record): taken from global memory inside
    // the loop below they were up to five dependent loads per entry, 35 us of set-up per workgroup
    uint32_t *sbs_lds = &sh.rec[0][0];
    for (uint32_t i = threadIdx.x; i < 112; i += kT) sbs_lds[i] = a.tab.sbs[i];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (12 + 24) * 32; i += kT) {  // crc.c:59-64, as in the scan kernel
        const bool lng = i >= (uint32_t)kGsynLong;
        const uint32_t e = lng ? i - kGsynLong : i, G = e >> 5, v = e & 31, nbits = lng ? 112 : 56;
        uint32_t x = 0;
        for (uint32_t j = 0; j < 5; ++j) {
            const uint32_t k = 5 * G + j;
            if (((v >> (4 - j)) & 1) && k < nbits) x ^= sbs_lds[k + 112 - nbits];
        }
        sh.gsyn[i] = x;
    }
    // the repair tables of --fix into LDS: from global memory the lookup behind the slicing was two dependent trips to memory
    // per 64 messages, 31 % of this kernel's time (RSB_STAGE_CLOCKS: 22.7 k of 73 k cycles per iteration)
    const bool syn_in_lds = syn_tables_fit_lds(a.tab);
    if (syn_in_lds)
        for (int which = 0; which < 2; ++which) {
            if (!a.tab.syn_keys[which]) continue;
            for (uint32_t i = threadIdx.x; i <= a.tab.syn_mask[which]; i += kT) {
                sh.syn.keys[(which ? 128u : 0u) + i] = a.tab.syn_keys[which][i];
                sh.syn.vals[(which ? 128u : 0u) + i] = a.tab.syn_vals[which][i];
            }
        }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
#ifdef RSB_STAGE_CLOCKS
    uint32_t eclk[5] = {0, 0, 0, 0, 0}, eiters = 0;
    uint64_t et = __builtin_readcyclecounter();
#define EMIT_STAMP(k) { const uint64_t t_ = __builtin_readcyclecounter(); eclk[k] += (uint32_t)((t_ - et) >> 4); et = t_; }
#else
#define EMIT_STAMP(k)
#endif
    uint32_t nacc = r.counters[kCntAccepted];
    const bool folds = r.fold_with_records != 0 && r.counters[kCntChanged] == 0;  // (the list kernel's verdict on this round)
    if (nacc > r.out_cap) nacc = r.out_cap;
    if (r.host_out && nacc > r.host_cap) nacc = r.host_cap;
    uint32_t by_errors[3] = {0, 0, 0};  // wave-uniform
    constexpr int kRawVecs = (FMT == kFmtUC8 || FMT == kFmtMag16) ? 1 : 2;
    constexpr int kBps = 2 * kRawVecs;
    const uint32_t nthreads = gridDim.x * kT;
    for (uint32_t base = (blockIdx.x * kT + threadIdx.x) & ~63u; base < nacc; base += nthreads) {
        EMIT_STAMP(0)  // (set-up before the first iteration; the loop's own bookkeeping afterwards)
        const uint32_t mi = base + lane;
        const bool valid = mi < nacc;
        uint32_t g = 0, phase = 0, info = kComboDf17 << 27, score = 0;
        if (valid) {
            const uint32_t idx = r.acc_idx[mi];  // k_replay_list: message number -> item
            const uint64_t key = r.keys[idx];
            g = (uint32_t)(key >> 3);
            phase = (uint32_t)key & 7u;
            info = r.info[idx];
            score = r.acc[idx] - 1;
            // the batch's new members become "known before the next batch" (k_replay_fold, here for the price of two stores: at a
            // fixed point every address the batch taught the filter has an accepted message that adds it)
            if (folds && combo_adds(item_combo(info))) {
                const uint32_t v = item_value(info);
                r.t_cur[v] = 0;
                r.t_next[v] = 0;
            }
        }
        const uint32_t combo = item_combo(info);
        const bool lng = !combo_is_short(combo);
        const uint32_t nbits = lng ? 112 : 56;
        const bool any_long = __ballot(lng) != 0;
        EMIT_STAMP(1)  // the message's item: three dependent gathers
        const int ntrios = any_long ? kEmitTrios : 8;  // wave-uniform and even: 56-bit frames end in group 11 (rows 66..73), their power sum in row 67
        // row 0 = the dword that holds S[g + 19] (an even stream index); par = which half of it
        const int64_t e0 = (int64_t)g + 19 - (int64_t)kTrailing;  // index into iq of the first data sample (may be < 0: the overlap)
        const uint32_t par = (uint32_t)e0 & 1u;
        const int64_t sd = e0 - (int64_t)par;
        // With the DC block (DC: the kernel's own variant) the window is fetched from the 8-sample chunk that holds its first sample:
        // a trio is then three whole chunks of the filter's states plus the first chunk of the next trio, and the lane picks its 12
        // rows out of those 16 by its offset into the chunk (0..3 rows).  Rows 140.. of the window are never used, so the last
        // trio does without a successor.  (A batch with the filter in a kernel built without DC takes the sample-by-sample path.)
        const int64_t sda = DC ? (sd & ~(int64_t)7) : sd;
        const bool direct = a.iq_aligned && sd >= 0 && sda + 24 * ntrios + (DC ? 8 : 0) <= (int64_t)a.n && (DC || !a.dc_local);
        float2 zrun = make_float2(0.0f, 0.0f);
        int64_t zpos = -1;
        [[maybe_unused]] uint32_t dc_carry[4] = {0, 0, 0, 0};
        // (lanes that cannot read their window directly read the head of the UC8 table instead -- always there, 128 KB -- and
        // ignore it: behind a branch the compiler cannot count the loads in flight and waits for ALL of them at every use,
        // refills included, so nothing of the prefetch overlapped anything)
        const uint4_a4 *src = direct ? (const uint4_a4 *)((const uint8_t *)a.iq + kBps * sda) : (const uint4_a4 *)a.tab.uc8_lut;
        constexpr int kFlight = DC ? 2 : RSB_EMIT_FLIGHT;  // trios in flight (both trio counts, 12 and 8, are multiples of it; DC: four would spill)
        uint4 raw[kFlight][3][kRawVecs];
        // DC: the chunks' local filter states travel with the raw samples; the tile's starting state (the window crosses into the next
        // tile at most once) and the power of dc_b are kept running -- dc_pow[j + 1] = dc_pow[j] * dc_pow[1] is how the table was made
        [[maybe_unused]] float2 lz[DC ? kFlight : 1][3];
        [[maybe_unused]] const float2 *lsrc = a.dc_local;
        [[maybe_unused]] double2 z0a = make_double2(0, 0), z0b = make_double2(0, 0);
        [[maybe_unused]] double pj = 1.0, m8 = 1.0;
        [[maybe_unused]] uint32_t jj = 0;
        if constexpr (DC) {
            if (direct) {
                const uint64_t cf = (uint64_t)sda >> 3;
                lsrc = a.dc_local + cf;
                z0a = a.dc_tile_start[cf >> 9];
                z0b = a.dc_tile_start[(cf >> 9) + 1];
                jj = (uint32_t)cf & 511u;
                pj = a.dc_pow[jj];
            }
            m8 = a.dc_pow[1];
        }
        auto next_state = [&](const float2 l) -> float2 {  // the state before the next chunk of the window, in ascending order
            const float2 z = make_float2((float)((double)l.x + z0a.x * pj), (float)((double)l.y + z0a.y * pj));
            pj *= m8;
            if (++jj == 512u) {
                jj = 0;
                pj = 1.0;
                z0a = z0b;
            }
            return z;
        };
        auto issue = [&](int slot, int t) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int v = 0; v < kRawVecs; ++v) raw[slot][k][v] = src[kRawVecs * (3 * t + k) + v];
            if constexpr (DC) {
#pragma unroll
                for (int k = 0; k < 3; ++k) lz[slot][k] = lsrc[3 * t + k];
            }
        };
        auto rows_of = [&](int slot, int t, uint32_t (&R)[12]) {  // 24 samples from row 12 t on -> 12 rows of biased magnitude pairs
            if (DC && direct) {
                if constexpr (DC) {
                    uint32_t E[16];
                    if (t == 0) {  // (every later trio has its first chunk from the trio before: dc_carry)
                        const uint4 o = dc_chunk_from<FMT>(a, raw[slot][0][0], raw[slot][0][kRawVecs - 1], next_state(lz[slot][0]));
                        dc_carry[0] = o.x;
                        dc_carry[1] = o.y;
                        dc_carry[2] = o.z;
                        dc_carry[3] = o.w;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) E[k] = dc_carry[k];
#pragma unroll
                    for (int k = 1; k < 4; ++k) {
                        const int sl = k < 3 ? slot : (slot + 1) % kFlight, kk = k < 3 ? k : 0;  // the trio after this one is in the next slot
                        const uint4 o = dc_chunk_from<FMT>(a, raw[sl][kk][0], raw[sl][kk][kRawVecs - 1], next_state(lz[sl][kk]));
                        E[4 * k] = o.x;
                        E[4 * k + 1] = o.y;
                        E[4 * k + 2] = o.z;
                        E[4 * k + 3] = o.w;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) dc_carry[k] = E[12 + k];
                    const uint32_t shift = ((uint32_t)sd & 7u) >> 1;
#pragma unroll
                    for (int k = 0; k < 12; ++k) R[k] = shift == 0 ? E[k] : (shift == 1 ? E[k + 1] : (shift == 2 ? E[k + 2] : E[k + 3]));
                }
            } else if (direct) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    uint4 o;
                    if constexpr (FMT == kFmtUC8) {
                        o = uc8_chunk_radial_exact<kBias>(raw[slot][k][0], (uint32_t)(uintptr_t)sh.lut, a.tab.uc8_patch_p1);
                    } else if constexpr (FMT == kFmtMag16) {
                        const uint4 x = raw[slot][k][0];
                        o = make_uint4(x.x ^ kBias2, x.y ^ kBias2, x.z ^ kBias2, x.w ^ kBias2);
                    } else {
                        const uint4 x0 = raw[slot][k][0], x1 = raw[slot][k][kRawVecs - 1];
                        o = make_uint4(sc16_pair<FMT>(x0, 0), sc16_pair<FMT>(x0, 1), sc16_pair<FMT>(x1, 0), sc16_pair<FMT>(x1, 1));
                    }
                    R[4 * k] = o.x;
                    R[4 * k + 1] = o.y;
                    R[4 * k + 2] = o.z;
                    R[4 * k + 3] = o.w;
                }
            } else {  // near the ends of the stream, or unaligned input: sample by sample
                const int64_t s = (int64_t)g + 19 - (int64_t)par + 24 * t;  // S index of the trio's first sample
                for (int k = 0; k < 12; ++k) {
                    const uint32_t lo = stream_mag_run<FMT>(a, s + 2 * k, zrun, zpos), hi = stream_mag_run<FMT>(a, s + 2 * k + 1, zrun, zpos);
                    R[k] = ((lo ^ kBias) & 0xffffu) | ((hi ^ kBias) << 16);
                }
            }
        };
        uint32_t c[15];
        load_coefs(sh.coef, par, phase, c);
        const uint32_t *gs = sh.gsyn + (lng ? kGsynLong : kGsynShort);
        const uint32_t last_group = lng ? 22u : 11u, last_keep = lng ? 24u : 16u;  // bits 110, 111 / bit 55
        const uint32_t half_len = nbits * 6 / 5;  // rows of the power sum: signal_len = nbits * 12 / 5 samples (demod_2400.c:342)
        uint32_t X[4] = {0, 0, 0, 0};             // the message as a 128-bit shift register, 5 bits in per group
        uint32_t syn = 0, last_lo = 0, first_lo = 0;
        uint64_t pw = 0;                          // sum of d^2 (see below), turned into the sum of mag^2 after the loop
        int sum_d = 0;
        uint32_t P[6] = {0, 0, 0, 0, 0, 0};       // rows 6..11 of the previous trio
        auto take_group = [&](uint32_t u, const uint32_t (&D)[8]) {
            uint32_t b = slice_group(D, c);
            b = u < last_group ? b : (u == last_group ? (b & last_keep) : 0u);
            syn ^= gs[u * 32 + b];  // the CRC is linear: crc.c:59-64,140
            X[3] = __builtin_amdgcn_alignbit(X[3], X[2], 27);
            X[2] = __builtin_amdgcn_alignbit(X[2], X[1], 27);
            X[1] = __builtin_amdgcn_alignbit(X[1], X[0], 27);
            X[0] = (X[0] << 5) | b;
        };
#pragma unroll
        for (int slot = 0; slot < kFlight; ++slot) issue(slot, slot);
        for (int t2 = 0; t2 < ntrios; t2 += kFlight) {
#pragma unroll
            for (int slot = 0; slot < kFlight; ++slot) {
                const int t = t2 + slot;
                uint32_t R[12];
                rows_of(slot, t, R);
                // refills go out in pairs of trios, 96 contiguous bytes back to back: the pieces of a cache line then arrive from one
                // fetch (a trio at a time, the next piece of a line came a few hundred instructions of seven other wavefronts later)
                // (the refill is unconditional -- past the frame's end it reads the last trio again and nobody uses it: after a
                // conditional load the compiler no longer knows how many loads are in flight and waits for all of them)
                if (RSB_EMIT_BURST && !DC) {  // (DC, two in flight: the slot after this one must hold the next trio already)
                    if (slot & 1) {
                        issue(slot - 1, min(t + kFlight - 1, ntrios - 1));
                        issue(slot, min(t + kFlight, ntrios - 1));
                    }
                } else {
                    issue(slot, min(t + kFlight, ntrios - 1));
                }
                // demod_2400.c:345-351: sum of mag^2 over rows [0, half_len) (corrected below when the frame starts in an upper half).
                // On the biased pairs d = mag - 32768: sum mag^2 = sum d^2 + 65536 sum d + 2^30 n, both sums one v_dot2_i32_i16
                // per row (two squares are at most 2^31: the dword is read as unsigned).  half_len is 67 (56 bits) or 134 rows:
                // trios 0..4 lie inside for every lane; from trio 5 on a lane masks the rows it does not count.
                if (t < 5) {  // wave-uniform
#pragma unroll
                    for (int k = 0; k < 12; ++k) {
                        pw += (uint32_t)sdot2_first(R[k], R[k]);
                        sum_d = sdot2(R[k], 0x00010001u, sum_d);
                    }
                } else {
                    const int rows_in = (int)half_len - 12 * t;  // rows of this trio inside the sum (<= 0: none, >= 12: all)
#pragma unroll
                    for (int k = 0; k < 12; ++k) {
                        const uint32_t x = k < rows_in ? R[k] : 0u;  // d = 0 adds nothing to either sum
                        pw += (uint32_t)sdot2_first(x, x);
                        sum_d = sdot2(x, 0x00010001u, sum_d);
                    }
                }
                // row half_len (the frame's last half sample when it starts in an upper half): row 67 = trio 5, row 7; row 134 = trio 11, row 2
                if (t == 5 && !lng) last_lo = (R[7] ^ kBias2) & 0xffffu;
                if (t == 11 && lng) last_lo = (R[2] ^ kBias2) & 0xffffu;
                if (t == 0) first_lo = (R[0] ^ kBias2) & 0xffffu;
                if (t > 0) {  // group 2t - 1: rows 12t - 6 .. 12t + 1
                    const uint32_t D[8] = {P[0], P[1], P[2], P[3], P[4], P[5], R[0], R[1]};
                    take_group(2u * (uint32_t)t - 1u, D);
                }
                {  // group 2t: rows 12t .. 12t + 7
                    const uint32_t D[8] = {R[0], R[1], R[2], R[3], R[4], R[5], R[6], R[7]};
                    take_group(2u * (uint32_t)t, D);
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) P[k] = R[6 + k];
            }
        }
        EMIT_STAMP(2)  // windows, conversion, slicing, power
        // the 128-bit register has taken 2 ntrios - 1 groups; as if it had taken 24 (message bit 0 at bit 119)
        for (int u = 2 * ntrios - 1; u < 24; ++u) {
            X[3] = __builtin_amdgcn_alignbit(X[3], X[2], 27);
            X[2] = __builtin_amdgcn_alignbit(X[2], X[1], 27);
            X[1] = __builtin_amdgcn_alignbit(X[1], X[0], 27);
            X[0] <<= 5;
        }
        uint32_t w[4] = {(X[3] << 8) | (X[2] >> 24), (X[2] << 8) | (X[1] >> 24), (X[1] << 8) | (X[0] >> 24), X[0] << 8};
        const uint32_t df = w[0] >> 27;
        pw = (uint64_t)((int64_t)pw + 65536ll * (int64_t)sum_d + ((int64_t)(2u * half_len) << 30));  // sum of mag^2 over 2 half_len samples
        if (par) pw += (uint64_t)last_lo * last_lo - (uint64_t)first_lo * first_lo;  // starts in the upper half of row 0, ends in the lower half of row half_len
        const uint32_t signal_len = nbits * 12 / 5;

        // ---- repair: DF11 on the syndrome without its IID bits (mode_s.c:475-489), DF17/18 on all of it (:508-518)
        uint32_t fix = 0;
        if (valid && combo >= kComboDf11Iid0) {
            const uint32_t sy = (combo <= kComboDf11IidX + 1) ? (syn & 0xffff80u) : syn;
            if (sy) fix = syn_in_lds ? diagnose_lds(sh.syn, a.tab, lng ? 1 : 0, sy) : diagnose(a.tab, lng ? 1 : 0, sy);
            if (fix == 0xffffffffu) fix = 0;  // cannot happen for an item the scan classified; keep the frame as sliced
        }
        const uint32_t nerr = fix & 0xffu;
        for (uint32_t e = 0; e < nerr; ++e) {
            const uint32_t p = (fix >> (8 + 8 * e)) & 0xffu;  // crc.c:410-418
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((p >> 5) == (uint32_t)k) w[k] ^= 1u << (31 - (p & 31));
        }
        EMIT_STAMP(3)  // repair
        if (valid) {
            const uint64_t ts = 5 * (r.g0 + g) + (8 + 56) * 12 + (phase + 4);  // demod_2400.c:320
            const bool parity_addr = combo <= kComboCommB;
            const bool df11 = combo >= kComboDf11Iid0 && combo <= kComboDf11IidX + 1;
            const double signal_power = (double)pw / 65535.0 / 65535.0;
            const double level = signal_power / (double)signal_len;
            uint32_t *o = &sh.rec[threadIdx.x >> 6][lane * 14];  // rsb_message, packed: 14 dwords
            o[0] = (uint32_t)ts;
            o[1] = (uint32_t)(ts >> 32);
            o[2] = __builtin_bswap32(w[0]);  // msg[0..3]
            o[3] = __builtin_bswap32(w[1]);
            o[4] = __builtin_bswap32(w[2]);
            o[5] = (__builtin_bswap32(w[3]) & 0xffffu) | (nbits << 16) | (df << 24);  // msg[12..13], msgbits, msgtype
            o[6] = syn;                                                             // crc
            o[7] = nerr;                                                            // correctedbits
            o[8] = parity_addr ? syn : (w[0] & 0xffffffu);                          // addr
            o[9] = score;
            o[10] = parity_addr ? 3u : (df11 ? 4u : 7u);                            // datasource_t
            o[11] = df11 ? (syn & 0x7fu) : 0u;                                      // iid
            const uint64_t lv = __builtin_bit_cast(uint64_t, level);
            o[12] = (uint32_t)lv;
            o[13] = (uint32_t)(lv >> 32);
        }
        {
            // The wavefront's records [base, base + nv) are contiguous in the list: 3584 bytes as 224 x 16-byte stores, each
            // instruction covering 1 KB without holes.  That is what makes writing straight into the caller's pinned host
            // buffer (r.host_out: no device -> host copy afterwards) run at PCIe rate; dword stores at a 56-byte stride do not.
            wave_sync();
            const uint32_t nv = min(64u, nacc - base);
            rsb_message *dst = (r.host_out ? r.host_out : r.out) + base;
            const uint32_t *src = sh.rec[threadIdx.x >> 6];
            if (nv == 64 && ((uintptr_t)dst & 15u) == 0) {
#pragma unroll
                for (uint32_t t = 0; t < 4; ++t) {
                    const uint32_t q = t * 64 + lane;
                    if (q < 224) ((uint4 *)dst)[q] = ((const uint4 *)src)[q];
                }
            } else {
                for (uint32_t q = lane; q < nv * 14; q += 64) ((uint32_t *)dst)[q] = src[q];
            }
            wave_sync();
        }
        EMIT_STAMP(4)  // record (level in double) into LDS, records out
#ifdef RSB_STAGE_CLOCKS
        ++eiters;
#endif
        by_errors[0] += (uint32_t)__popcll(__ballot(valid && nerr == 0));
        by_errors[1] += (uint32_t)__popcll(__ballot(valid && nerr == 1));
        by_errors[2] += (uint32_t)__popcll(__ballot(valid && nerr >= 2));
    }
    if (lane == 0)
        for (int e = 0; e < 3; ++e)
            if (by_errors[e]) atomicAdd(&a.counters[kCntAccE0 + e], by_errors[e]);
#ifdef RSB_STAGE_CLOCKS
    if (lane == 0) {
        for (int k = 0; k < 5; ++k) atomicAdd(&a.counters[kDbgEmitSetup + k], eclk[k]);
        atomicAdd(&a.counters[kDbgEmitIters], eiters);
    }
#endif
}

// ---- the kernels of the prune / replay chain: one batch per launch (arguments by value), or -- the *_many forms -- the batches of
// several channels in one launch: blockIdx.y picks the channel, whose arguments arrive by scalar loads from a device array, and the
// grid's x extent is the largest channel's (every body strides or checks its own batch's sizes).  Thirty-two channels' chains are
// then 16 launches instead of 500, and their small kernels fill the chip together.
__device__ __forceinline__ bool chain_idle(const ChainArgs &c) { return c.a.n == 0 || c.r.nblocks == 0; }

__global__ __launch_bounds__(kThreads) void k_prune_summary(const uint32_t *bitmap, uint32_t *summary) { prune_summary_body(bitmap, summary); }
__global__ __launch_bounds__(kThreads) void k_prune_count(const BatchArgs a) { prune_count_body(a); }
__global__ __launch_bounds__(kChunkGroup) void k_prune_chunks(const BatchArgs a) { prune_chunks_body(a); }
__global__ __launch_bounds__(1024) void k_prune_offsets(const BatchArgs a) { prune_offsets_body(a); }
__global__ __launch_bounds__(kThreads) void k_prune_write(const BatchArgs a) { prune_write_body(a); }
__global__ __launch_bounds__(kThreads) void k_replay_bounds(const ReplayArgs r) { replay_bounds_body(r, blockIdx.x, gridDim.x, kThreads); }
__global__ __launch_bounds__(kThreads) void k_replay_score(const ReplayArgs r) { replay_score_body(r); }
__global__ __launch_bounds__(kThreads) void k_replay_walk(const ReplayArgs r) { replay_walk_body(r); }
__global__ __launch_bounds__(1024) void k_replay_blkbase(const ReplayArgs r) { replay_blkbase_body(r); }
template <bool SEED>
__global__ __launch_bounds__(kMinThreads) void k_replay_table_min(const ReplayArgs r) { replay_table_min_body<SEED>(r, gridDim.x); }
// the optimistic table and the blocks' first items in one launch (neither needs the other: workgroups [0, runs) / [runs, gridDim.x))
__global__ __launch_bounds__(kMinThreads) void k_replay_seed_bounds(const ReplayArgs r, uint32_t runs) {
    if (blockIdx.x < runs)
        replay_table_min_body<true>(r, runs);
    else
        replay_bounds_body(r, blockIdx.x - runs, gridDim.x - runs, kMinThreads);
}
__global__ __launch_bounds__(kThreads) void k_replay_compare(const ReplayArgs r, uint32_t *changed) { replay_compare_body(r, changed); }
template <bool CHECK>
__global__ __launch_bounds__(kThreads) void k_replay_list(const ReplayArgs r, uint32_t *counters) { replay_list_body<CHECK>(r, counters); }
__global__ __launch_bounds__(kThreads) void k_replay_fold(const ReplayArgs r) { replay_fold_body(r); }
template <int FMT, bool DC = false>
__global__ __launch_bounds__(64 * kEmitWaves) void k_replay_emit(const BatchArgs a, const ReplayArgs r) { replay_emit_body<FMT, DC>(a, r); }

// (the summary words start at zero: this kernel stands where launch_prune_ordered has its memset)
__global__ __launch_bounds__(kThreads) void k_prune_summary_clear_many(const ChainArgs *__restrict__ m) {
    const ChainArgs &c = m[blockIdx.y];
    if (chain_idle(c) || !c.a.adder_summary) return;
    static_assert(kSummaryWords == 4 * kThreads, "one 16-byte piece per thread");
    ((uint4 *)c.a.adder_summary)[threadIdx.x] = make_uint4(0, 0, 0, 0);
}
__global__ __launch_bounds__(kThreads) void k_prune_summary_many(const ChainArgs *__restrict__ m) {
    const ChainArgs &c = m[blockIdx.y];
    if (chain_idle(c) || !c.a.adder_summary) return;
    prune_summary_body(c.a.adder_bitmap, c.a.adder_summary);
}
__global__ __launch_bounds__(kThreads) void k_prune_tiles_clear_many(const ChainArgs *__restrict__ m) {
    const ChainArgs &c = m[blockIdx.y];
    if (!chain_idle(c)) prune_tiles_clear_body(c.a);
}
__global__ __launch_bounds__(kTileThreads) void k_prune_onepass_many(const ChainArgs *__restrict__ m) {
    const ChainArgs &c = m[blockIdx.y];
    if (!chain_idle(c)) prune_onepass_body(c.a);
}
__global__ __launch_bounds__(kThreads) void k_prune_count_many(const ChainArgs *__restrict__ m) {
    const ChainArgs &c = m[blockIdx.y];
    if (!chain_idle(c)) prune_count_body(c.a);
}
__global__ __launch_bounds__(kChunkGroup) void k_prune_chunks_many(const ChainArgs *__restrict__ m) {
    const ChainArgs &c = m[blockIdx.y];
    if (!chain_idle(c)) prune_chunks_body(c.a);
}
__global__ __launch_bounds__(1024) void k_prune_offsets_many(const ChainArgs *__restrict__ m) {
    const ChainArgs &c = m[blockIdx.y];
    if (!chain_idle(c)) prune_offsets_body(c.a);
}
__global__ __launch_bounds__(kThreads) void k_prune_write_many(const ChainArgs *__restrict__ m) {
    const ChainArgs &c = m[blockIdx.y];
    if (!chain_idle(c)) prune_write_body(c.a);
}
__global__ __launch_bounds__(kThreads) void k_replay_bounds_many(const ChainArgs *__restrict__ m) {
    const ChainArgs &c = m[blockIdx.y];
    if (!chain_idle(c)) replay_bounds_body(c.r, blockIdx.x, gridDim.x, kThreads);
}
__global__ __launch_bounds__(kMinThreads) void k_replay_seed_bounds_many(const ChainArgs *__restrict__ m, uint32_t runs) {
    const ChainArgs &c = m[blockIdx.y];
    if (chain_idle(c)) return;
    if (blockIdx.x < runs)
        replay_table_min_body<true>(c.r, runs);
    else
        replay_bounds_body(c.r, blockIdx.x - runs, gridDim.x - runs, kMinThreads);
}
template <bool SEED>
__global__ __launch_bounds__(kMinThreads) void k_replay_table_min_many(const ChainArgs *__restrict__ m) {
    const ChainArgs &c = m[blockIdx.y];
    if (!chain_idle(c)) replay_table_min_body<SEED>(c.r, gridDim.x);
}
__global__ __launch_bounds__(kThreads) void k_replay_score_many(const ChainArgs *__restrict__ m) {
    const ChainArgs &c = m[blockIdx.y];
    if (!chain_idle(c)) replay_score_body(c.r);
}
__global__ __launch_bounds__(kThreads) void k_replay_walk_many(const ChainArgs *__restrict__ m) {
    const ChainArgs &c = m[blockIdx.y];
    if (!chain_idle(c)) replay_walk_body(c.r);
}
__global__ __launch_bounds__(kThreads) void k_replay_compare_many(const ChainArgs *__restrict__ m) {
    const ChainArgs &c = m[blockIdx.y];
    if (!chain_idle(c)) replay_compare_body(c.r, &c.r.counters[kCntChanged]);
}
__global__ __launch_bounds__(1024) void k_replay_blkbase_many(const ChainArgs *__restrict__ m) {
    const ChainArgs &c = m[blockIdx.y];
    if (chain_idle(c)) return;
    replay_blkbase_body(c.r);
}
template <bool CHECK>
__global__ __launch_bounds__(kThreads) void k_replay_list_many(const ChainArgs *__restrict__ m) {
    const ChainArgs &c = m[blockIdx.y];
    if (!chain_idle(c)) replay_list_body<CHECK>(c.r, c.r.counters);
}
template <int FMT>
__global__ __launch_bounds__(64 * kEmitWaves) void k_replay_emit_many(const ChainArgs *__restrict__ m) {
    const ChainArgs &c = m[blockIdx.y];
    if (!chain_idle(c)) replay_emit_body<FMT>(c.a, c.r);
}
// A channel whose first round was the fixed point (nearly every one) folds its new members into the tables here and now; the
// others are iterated on by the host, channel by channel, which folds when it is through.  Then what the host wants to see of every
// channel, gathered into one block that crosses PCIe in one copy:
//   [changed | counters (kCntCount) | accepted messages per block (max_nblocks) | the first kPinAdders new adders]
__global__ __launch_bounds__(kThreads) void k_replay_fold_many(const ChainArgs *__restrict__ m) {
    const ChainArgs &c = m[blockIdx.y];
    if (chain_idle(c) || c.r.counters[kCntChanged] != 0) return;
    replay_fold_body(c.r);
}
__global__ __launch_bounds__(kThreads) void k_chain_readback_many(const ChainArgs *__restrict__ m, uint32_t *__restrict__ out, uint32_t stride, uint32_t max_nblocks,
                                                                  uint32_t pin_adders) {
    const ChainArgs &c = m[blockIdx.y];
    uint32_t *o = out + (size_t)blockIdx.y * stride;
    if (chain_idle(c)) {
        for (uint32_t i = threadIdx.x; i < 1 + (uint32_t)kCntCount; i += kThreads) o[i] = 0;
        return;
    }
    const uint32_t *cnt = c.r.counters;
    if (threadIdx.x == 0) o[0] = cnt[kCntChanged];
    for (uint32_t i = threadIdx.x; i < (uint32_t)kCntCount; i += kThreads) o[1 + i] = cnt[i];
    for (uint32_t i = threadIdx.x; i < c.r.nblocks; i += kThreads) o[1 + kCntCount + i] = c.r.blk_counts[i];
    const uint32_t nadd = min(min(cnt[kCntNewAdders], c.r.new_adders_cap), pin_adders);
    const uint32_t *src = (const uint32_t *)c.r.new_adders;
    for (uint32_t i = threadIdx.x; i < 2 * nadd; i += kThreads) o[1 + kCntCount + max_nblocks + i] = src[i];
}

// ------------------------------------------------------------------------------------------------
// Mode A/C

// The same sums for many blocks at once.  The sequential part is two float additions per sample and block; k_block_means_float
// spends a whole wavefront on one block's chain (v_readlane + v_add per sample: 64 blocks' chains would fit its lanes).  Here a
// workgroup takes 16 blocks: wavefront 0 holds one block per lane and adds, in sample order, what wavefronts 1..4 computed 64
// samples ahead into LDS.  What bounds the chain is the rate at which ONE wavefront issues instructions (one per 5-9 cycles,
// profiles/r2_valu_issue_cost.txt), not its lanes, so: both sums of a sample in one v_pk_add_f32, two samples' (mag, magsq) pairs
// per ds_read_b128 -- 1.5 instructions per sample in place of 4 -- and small groups, so that 512 chains run side by side (two
// workgroups per CU) and the converting wavefronts (the VALU work: ~20 instructions per sample) have every SIMD of the chip.
// LDS layout: [pair of samples][block] units of 16 bytes {mag0, magsq0, mag1, magsq1}, 17 units per row: the adder's 16 lanes
// read one contiguous row, a converting wavefront's lane pairs write 8-byte halves 272 bytes apart (every bank once per 8 pairs).
constexpr int kMeansGroup = 16, kMeansChunk = 256, kMeansRow = 17, kMeansProducers = kMeansGroup / 4, kMeansThreads = 64 * (1 + kMeansProducers);
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));
constexpr size_t kMeansLds = 2 * (size_t)(kMeansChunk / 2) * kMeansRow * sizeof(float4v);
template <int FMT>
__global__ __launch_bounds__(kMeansThreads) void k_block_means_group(const void *iq, uint64_t n, uint64_t blk_len, double *means, uint32_t nblocks) {
    if constexpr (FMT == kFmtSC16 || FMT == kFmtSC16Q11) {
        extern __shared__ __attribute__((aligned(16))) unsigned char means_lds[];
        float4v (*buf)[(kMeansChunk / 2) * kMeansRow] = reinterpret_cast<float4v (*)[(kMeansChunk / 2) * kMeansRow]>(means_lds);
        const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        const uint32_t b0 = blockIdx.x * kMeansGroup;
        const uint32_t nchunks = (uint32_t)((blk_len + kMeansChunk - 1) / kMeansChunk);
        constexpr float scale = (FMT == kFmtSC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
        auto length_of = [&](uint32_t blk) -> uint32_t {  // samples of block blk (0 beyond the last block)
            if (blk >= nblocks) return 0u;
            const uint64_t start = (uint64_t)blk * blk_len;
            return (uint32_t)((n - start < blk_len) ? (n - start) : blk_len);
        };
        // wavefronts 1..4, four blocks each, four consecutive samples per lane and block in one 16-byte load (1 KB per wavefront
        // and block: every trip to HBM brings 256 samples of a block), fetched two chunks ahead; zero samples (beyond a block's
        // end) give mag = magsq = 0
        const bool vec = (uintptr_t)iq % 16 == 0 && blk_len % 4 == 0;
        uint4 pre[2][4];
        auto fetch = [&](uint32_t c, int slot) {
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const uint32_t blk = b0 + (wave - 1) * 4 + q, s = c * kMeansChunk + 4 * lane, len = length_of(blk);
                const uint32_t *src = (const uint32_t *)iq + (uint64_t)blk * blk_len + s;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (vec && s + 4 <= len) {
                    v = *(const uint4 *)src;
                } else if (s < len) {
                    v.x = src[0];
                    if (s + 1 < len) v.y = src[1];
                    if (s + 2 < len) v.z = src[2];
                    if (s + 3 < len) v.w = src[3];
                }
                pre[slot][q] = v;
            }
        };
        auto mags = [&](uint32_t v) {
            const float fI = (float)(short)(v & 0xffffu) * scale, fQ = (float)(short)(v >> 16) * scale;
            float magsq = fI * fI + fQ * fQ;  // convert.c:236-243
            if (magsq > 1.0f) magsq = 1.0f;
            float2v o;
            o.x = sqrt_rn(magsq);
            o.y = magsq;
            return o;
        };
        auto emit = [&](int slot, uint32_t which) {
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const uint32_t lb = (wave - 1) * 4 + q;
                const uint4 v = pre[slot][q];
                float4v lo, hi;
                lo.xy = mags(v.x);
                lo.zw = mags(v.y);
                hi.xy = mags(v.z);
                hi.zw = mags(v.w);
                buf[which][(2 * lane) * kMeansRow + lb] = lo;
                buf[which][(2 * lane + 1) * kMeansRow + lb] = hi;
            }
        };
        const uint32_t my_len = length_of(b0 + lane);  // wavefront 0: lane = block
        float2v sums = {0.0f, 0.0f};                   // {sum_level, sum_power}: v_pk_add_f32 rounds each half like v_add_f32
        auto consume = [&](uint32_t c) {
            const float4v *row = &buf[c & 1][lane];
            const uint32_t s0 = c * kMeansChunk;
            if (s0 + kMeansChunk <= my_len) {
                // convert.c:226-262: sample order.  Reads run kDepth pairs ahead of the additions (left to the compiler, every
                // read is waited for right before its two additions: one LDS round trip per pair of samples)
                constexpr int kDepth = 8;
                float4v q[kDepth];
#pragma unroll
                for (int k = 0; k < kDepth; ++k) q[k] = row[k * kMeansRow];
                for (uint32_t k0 = 0; k0 < (uint32_t)kMeansChunk / 2; k0 += kDepth) {
#pragma unroll
                    for (int k = 0; k < kDepth; ++k) {
                        const float4v x = q[k];
                        if (k0 + kDepth < (uint32_t)kMeansChunk / 2) q[k] = row[(k0 + kDepth + k) * kMeansRow];
                        sums += x.xy;
                        sums += x.zw;
                    }
                }
            } else {
                for (uint32_t k = 0; k < (uint32_t)kMeansChunk && s0 + k < my_len; ++k) {
                    const float4v v = row[(k >> 1) * kMeansRow];
                    sums += (k & 1) ? v.zw : v.xy;
                }
            }
        };
        if (wave >= 1) {
            fetch(0, 0);
            fetch(1, 1);
            emit(0, 0);
            fetch(2, 0);
        }
        __syncthreads();
        for (uint32_t c = 0; c < nchunks; c += 2) {  // iteration i = c + u: chunk i + 1 goes to LDS while chunk i is added
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t i = c + (uint32_t)u;
                if (wave >= 1) {
                    emit((u + 1) & 1, (i + 1) & 1);
                    fetch(i + 3, (u + 1) & 1);
                } else if (lane < (uint32_t)kMeansGroup && i < nchunks) {
                    consume(i);
                }
                __syncthreads();
            }
        }
        if (wave == 0 && lane < (uint32_t)kMeansGroup && b0 + lane < nblocks) {  // "sum / nsamples" is a float division there, widened afterwards
            means[2 * (b0 + lane)] = (double)(sums.x / (float)my_len);
            means[2 * (b0 + lane) + 1] = (double)(sums.y / (float)my_len);
        }
    }
}

// ---- Mode A/C and the block statistics it needs: one 1024-thread workgroup per 131072-sample block.
// Pass 1 (integer formats without caller-supplied means): the block's new samples -> sum(mag), sum(mag^2) exactly as the
// integer converters take them (convert.c:71-72,82-83,104-110).  Pass 2: the block once more (L2-resident: 256-512 KB), in
// LDS tiles, demod_2400.c:476-622 with one thread per f1_sample.  Raw IQ comes in 16-byte loads and, for UC8, goes through
// the folded 128 x 128 magnitude table in LDS (as in k_scan, here without the bias).
constexpr int kAcThreads = 1024;
constexpr int kAcTile = 32768;                           // f1 positions per LDS tile (kBlock % kAcTile == 0); 16384 (two workgroups per CU) measured slower
constexpr int kAcQ1 = 6144, kAcQ2 = 2048;                // workgroup queues per tile: F1 survivors, F2 survivors (more run on the spot)
constexpr int kAcChunks = (kAcTile + 74 + 7) / 8;        // 8-sample chunks per tile: 2 samples before the tile's first position, 72 after its last
// (m[f1_sample - 1] is read behind a position; bit 19 is sliced at f1_sample + 67 and looks 2 further, demod_2400.c:587-608)
static_assert(kBlock % kAcTile == 0 && kTrailing % 8 == 6, "tiles start 328 samples into 16-byte aligned IQ");

template <int FMT>
__device__ __forceinline__ uint4 ac_chunk(const void *iq, int64_t chunk, const uint16_t *lut, const uint32_t patch_p1) {  // 8 samples from iq index 8 * chunk -> 8 magnitudes
    if constexpr (FMT == kFmtUC8) {
        return uc8_chunk_radial_exact<0u>(((const uint4 *)iq)[chunk], (uint32_t)(uintptr_t)lut, patch_p1);
    } else if constexpr (FMT == kFmtMag16) {
        return ((const uint4 *)iq)[chunk];
    } else {
        const uint4 x0 = ((const uint4 *)iq)[2 * chunk], x1 = ((const uint4 *)iq)[2 * chunk + 1];
        return make_uint4(sc16_pair<FMT>(x0, 0) ^ kBias2, sc16_pair<FMT>(x0, 1) ^ kBias2, sc16_pair<FMT>(x1, 0) ^ kBias2, sc16_pair<FMT>(x1, 1) ^ kBias2);
    }
}

// the three tests both framing pulses must pass (demod_2400.c:535-546 for F1 at f1_sample, :560-573 for F2): rising edge,
// quiet third sample, at least 6 dB over the noise
__device__ __forceinline__ bool ac_pulse_tests(const uint16_t *m, int k, unsigned noise_level) {
    if (!(m[k - 1] < m[k])) return false;
    if (m[k + 2] > m[k] || m[k + 2] > m[k + 1]) return false;
    const unsigned level = ((unsigned)m[k] + m[k + 1]) / 2;
    return noise_level * 2 <= level;
}
// :551-558: F1's position on the 60 MHz clock from the power split of its two samples (float32 throughout), F2 14 bit times later
__device__ __forceinline__ unsigned ac_f2_clock(const uint16_t *m, int j, unsigned f1_sample) {
    const float f1a_power = (float)m[j] * (float)m[j];
    const float f1b_power = (float)m[j + 1] * (float)m[j + 1];
    const float fraction = f1b_power / (f1a_power + f1b_power);
    const float clockf = 25.0f * ((float)f1_sample + fraction * fraction);
    const unsigned f1_clock = (unsigned)((double)clockf + 0.5);
    return f1_clock + (87 * 14);
}

struct __attribute__((aligned(16))) AcShared {
    uint16_t lut[kRadialTotal];
    uint16_t mt[8 * kAcChunks];
    uint16_t q1[kAcQ1], q2[kAcQ2];
    uint64_t red[2][kAcThreads / 64];
    uint32_t nq1, nq2;
};
static_assert(sizeof(AcShared) <= 160 * 1024, "LDS budget of one CU");

// thresholds, the 20 bits, Gillham reordering (demod_2400.c:575-639) for a position whose F1 and F2 passed
// (out of line, called from the dense stage and from the two queue-overflow paths: inlined three times it made the kernel spill)
struct AcOut {
    uint32_t *count;
    AcCand *ac;
    uint32_t cap;
};
__device__ __noinline__ void ac_bits(const AcOut a, const uint16_t *m, uint32_t j, unsigned f1_sample, uint64_t blk_start, unsigned noise_level) {
    const unsigned f1_level = ((unsigned)m[j] + m[j + 1]) / 2;
    const unsigned f2_clock = ac_f2_clock(m, (int)j, f1_sample), f1_clock = f2_clock - (87 * 14);
    const int k2 = (int)j + (int)(f2_clock / 25 - f1_sample);
    const unsigned f2_level = ((unsigned)m[k2] + m[k2 + 1]) / 2;

    const unsigned f1f2_level = f1_level > f2_level ? f1_level : f2_level;
    const float midpoint = sqrtf((float)(noise_level * f1f2_level));
    const unsigned signal_threshold = (unsigned)((double)midpoint * 1.41421356237309504880 + 0.5);
    const unsigned noise_threshold = (unsigned)((double)midpoint / 1.41421356237309504880 + 0.5);

    // The 20 bits (:587-608), five at a time: the five windows m[s .. s + 2] are read first -- each as the two aligned dwords that
    // hold it, one ds_read2_b32 -- and then tested.  (Bit by bit, every step waited for its own three LDS reads: twenty round
    // trips in a row on the few lanes that get here.)
    unsigned uncertain_bits = 0, noisy_bits = 0, bits = 0, clock = f1_clock;
    constexpr int kBitsAtOnce = RSB_AC_BITS_AT_ONCE;
    static_assert(20 % kBitsAtOnce == 0, "whole groups");
    for (int bit0 = 0; bit0 < 20; bit0 += kBitsAtOnce) {
        uint32_t w0[kBitsAtOnce], w1[kBitsAtOnce], odd[kBitsAtOnce];
#pragma unroll
        for (int b = 0; b < kBitsAtOnce; ++b, clock += 87) {
            const int s = (int)j + (int)(clock / 25 - f1_sample);
            const uint32_t *w = (const uint32_t *)(m + (s & ~1));  // m = mt + 2 and s & ~1 are even: 4-byte aligned
            w0[b] = w[0];
            w1[b] = w[1];
            odd[b] = ((uint32_t)s & 1u) * 16u;
        }
#pragma unroll
        for (int b = 0; b < kBitsAtOnce; ++b) {
            const uint32_t pair = __builtin_amdgcn_alignbit(w1[b], w0[b], odd[b]);  // m[s] | m[s + 1] << 16
            const uint32_t ms = pair & 0xffffu, ms1 = pair >> 16, ms2 = (w1[b] >> odd[b]) & 0xffffu;
            bits <<= 1;
            noisy_bits <<= 1;
            uncertain_bits <<= 1;
            if (ms2 >= signal_threshold) noisy_bits |= 1;
            if (ms >= signal_threshold || ms1 >= signal_threshold)
                bits |= 1;
            else if (ms > noise_threshold && ms1 > noise_threshold)
                uncertain_bits |= 1;
        }
    }
    if ((bits & 0x80020) != 0x80020) return;
    if ((bits & 0x0101B) != 0) return;
    if (noisy_bits || uncertain_bits) return;

    const unsigned modeac =  // demod_2400.c:626-639
        ((bits & 0x40000) ? 0x0010 : 0) | ((bits & 0x20000) ? 0x1000 : 0) | ((bits & 0x10000) ? 0x0020 : 0) |
        ((bits & 0x08000) ? 0x2000 : 0) | ((bits & 0x04000) ? 0x0040 : 0) | ((bits & 0x02000) ? 0x4000 : 0) |
        ((bits & 0x00800) ? 0x0100 : 0) | ((bits & 0x00400) ? 0x0001 : 0) | ((bits & 0x00200) ? 0x0200 : 0) |
        ((bits & 0x00100) ? 0x0002 : 0) | ((bits & 0x00080) ? 0x0400 : 0) | ((bits & 0x00040) ? 0x0004 : 0) |
        ((bits & 0x00004) ? 0x0080 : 0);

    const uint32_t slot = atomicAdd(a.count, 1u);
    if (slot < a.cap) {
        AcCand c;
        c.g = (uint32_t)(blk_start + f1_sample);
        c.f2_clock = f2_clock;
        c.modeac = modeac;
        a.ac[slot] = c;
    }
}

__device__ __forceinline__ bool ac_f2_tests(const uint16_t *m, uint32_t j, unsigned f1_sample, unsigned noise_level) {  // :551-573
    const unsigned f2_clock = ac_f2_clock(m, (int)j, f1_sample);
    return ac_pulse_tests(m, (int)j + (int)(f2_clock / 25 - f1_sample), noise_level);
}

template <int FMT>
__global__ __launch_bounds__(kAcThreads) void k_modeac(const BatchArgs a, const int sums_only) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ac_lds[];
    AcShared &sh = *reinterpret_cast<AcShared *>(ac_lds);
    uint16_t *lut = sh.lut, *mt = sh.mt;
    uint64_t (*red)[kAcThreads / 64] = sh.red;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    if constexpr (FMT == kFmtUC8) load_radial_table(lut, a.tab.uc8_radial, 0u, tid, kAcThreads);
    // persistent: a workgroup takes every gridDim.x-th block, so the table is set up once per CU and not once per block
    const uint64_t nblk = (a.n + a.block_len - 1) / a.block_len;
    for (uint64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const uint64_t blk_start = blk * a.block_len;  // (a multiple of 8: the tiles below start on 16-byte boundaries of the IQ)
    const uint32_t blk_len = (uint32_t)((a.n - blk_start < (uint64_t)a.block_len) ? (a.n - blk_start) : (uint64_t)a.block_len);

    // ---- pass 1: sum(mag), sum(mag^2) over the block's new samples iq[blk_start .. blk_start + blk_len)
    if (!a.block_means && !(a.debug_flags & 16)) {
        uint32_t sl = 0;
        uint64_t sp = 0;
        auto take = [&](uint32_t v) {
            sl += v;  // at most 128 samples per thread: no overflow
            sp += (uint64_t)v * v;
        };
        uint32_t done = 0;
        if (a.iq_aligned) {
            const uint32_t nch = blk_len / 8;
            auto take8 = [&](const uint4 o) {
                take(o.x & 0xffffu); take(o.x >> 16); take(o.y & 0xffffu); take(o.y >> 16);
                take(o.z & 0xffffu); take(o.z >> 16); take(o.w & 0xffffu); take(o.w >> 16);
            };
            constexpr int kFull = kBlock / 8 / kAcThreads;  // chunks per thread of a whole block
            if ((FMT == kFmtUC8 || FMT == kFmtMag16) && blk_len == (uint32_t)kBlock) {
                // a whole block of a 2-byte format: all of the thread's 16-byte loads go out before the first is used (in the
                // loop below every load is waited for at its use: sixteen trips to memory one after the other)
                const uint4 *src = (const uint4 *)a.iq + blk_start / 8 + tid;
                constexpr int kBatch = 8;  // (all sixteen at once: 64 registers, and the 1024-thread workgroup's 128 run out)
                static_assert(kFull % kBatch == 0, "whole batches");
#pragma unroll 1
                for (int k0 = 0; k0 < kFull; k0 += kBatch) {
                    uint4 raw[kBatch];
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) raw[k] = src[(k0 + k) * kAcThreads];
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) {
                        if constexpr (FMT == kFmtUC8)
                            take8(uc8_chunk_radial_exact<0u>(raw[k], (uint32_t)(uintptr_t)lut, a.tab.uc8_patch_p1));
                        else
                            take8(raw[k]);
                    }
                }
            } else {
                for (uint32_t c = tid; c < nch; c += kAcThreads) take8(ac_chunk<FMT>(a.iq, (int64_t)(blk_start / 8 + c), lut, a.tab.uc8_patch_p1));
            }
            done = nch * 8;
        }
        for (uint32_t i = done + tid; i < blk_len; i += kAcThreads) take(mag_of<FMT>(a.iq, blk_start + i, a.tab.uc8_lut));
        uint64_t tl = sl;
        for (int off = 32; off > 0; off >>= 1) {
            tl += __shfl_down(tl, off, 64);
            sp += __shfl_down(sp, off, 64);
        }
        if ((tid & 63) == 0) {
            red[0][tid >> 6] = tl;
            red[1][tid >> 6] = sp;
        }
        __syncthreads();
        if (tid == 0) {
            uint64_t x = 0, y = 0;
            for (int w = 0; w < kAcThreads / 64; ++w) {
                x += red[0][w];
                y += red[1][w];
            }
            red[0][0] = x;
            red[1][0] = y;
            a.block_sums[2 * blk] = x;
            a.block_sums[2 * blk + 1] = y;
        }
        __syncthreads();
    }
    if (sums_only || (a.debug_flags & 8)) continue;

    // convert.c:104-110 then demod_2400.c:484-485
    const double mean_level = a.block_means ? a.block_means[2 * blk] : (double)red[0][0] / 65536.0 / (double)blk_len;
    const double mean_power = a.block_means ? a.block_means[2 * blk + 1] : (double)red[1][0] / 65535.0 / 65535.0 / (double)blk_len;
    const double noise_stddev = sqrt(mean_power - mean_level * mean_level);
    const unsigned noise_level = (unsigned)((mean_power + noise_stddev) * 65535 + 0.5);

    // ---- pass 2: tiles of kAcTile f1 positions; mt[i] = S[blk_start + t0 - 2 + i], i.e. m[j] = S[blk_start + t0 + j] = mt[j + 2]
    const uint16_t *m = mt + 2;
    const AcOut out{&a.counters[kCntAc], a.ac, a.ac_cap};
    // F1 on packed pairs unless the block is so noisy that 4 * noise_level leaves 16 bits (workgroup-uniform)
    const bool packed_f1 = noise_level <= 16383u && !(a.debug_flags & 512);
    const uint32_t level_thr = ((4u * noise_level) ^ kBias) * 0x00010001u;
    for (uint32_t t0 = 0; t0 < blk_len; t0 += kAcTile) {
        const int64_t q0 = (int64_t)(blk_start + t0) - 2 - (int64_t)kTrailing;  // iq index of mt[0]: a multiple of 8
        if (a.iq_aligned && q0 >= 0 && q0 + 8 * kAcChunks <= (int64_t)a.n) {
            if constexpr (FMT == kFmtUC8 || FMT == kFmtMag16) {  // the thread's loads of the tile first, then the conversions
                constexpr int kLoads = (kAcChunks + kAcThreads - 1) / kAcThreads;
                const uint4 *src = (const uint4 *)a.iq + q0 / 8;
                uint4 raw[kLoads];
#pragma unroll
                for (int k = 0; k < kLoads; ++k) {
                    const uint32_t c = tid + (uint32_t)k * kAcThreads;
                    raw[k] = src[c < (uint32_t)kAcChunks ? c : (uint32_t)kAcChunks - 1];
                }
#pragma unroll
                for (int k = 0; k < kLoads; ++k) {
                    const uint32_t c = tid + (uint32_t)k * kAcThreads;
                    if (c >= (uint32_t)kAcChunks) break;
                    if constexpr (FMT == kFmtUC8)
                        ((uint4 *)mt)[c] = uc8_chunk_radial_exact<0u>(raw[k], (uint32_t)(uintptr_t)lut, a.tab.uc8_patch_p1);
                    else
                        ((uint4 *)mt)[c] = raw[k];
                }
            } else {
                for (uint32_t c = tid; c < (uint32_t)kAcChunks; c += kAcThreads) ((uint4 *)mt)[c] = ac_chunk<FMT>(a.iq, q0 / 8 + c, lut, a.tab.uc8_patch_p1);
            }
        } else {  // the batch's first tile (carry), its last ones (end of the stream), unaligned input
            for (uint32_t i = tid; i < 8u * kAcChunks; i += kAcThreads) mt[i] = (uint16_t)stream_mag<FMT>(a, q0 + (int64_t)kTrailing + i);
        }
        __syncthreads();
        if (a.debug_flags & 32) continue;
        // Three stages with compaction into workgroup queues in between: on noise the F1 tests pass at a few per cent of the
        // positions and F2 at under one per cent, and a lane that goes on alone drags its whole wavefront along -- so the
        // float steps (F2) and the 20-bit loop run on densely packed survivors of the whole tile.
        if (tid == 0) sh.nq1 = sh.nq2 = 0;
        __syncthreads();
        // F1 (:535-546), four consecutive positions per lane and step from one 16-byte window mt[4p .. 4p + 8) = m[4p - 2 .. 4p + 6).
        // All of a lane's windows of the tile are read first and its 32 results kept in one word; then ONE queue reservation per
        // wavefront and tile (round 2 reserved per step: eight LDS atomics with their round trips in a row, 0.27 ms per Gsample).
        constexpr int kF1Steps = kAcTile / 4 / kAcThreads;
        static_assert(kF1Steps * 4 == 32 && kF1Steps * 4 * kAcThreads == kAcTile, "a lane's F1 results of a tile fill one 32-bit word");
        auto f1_pass = [&](const uint2 lo, const uint2 hi, const uint32_t p) -> uint32_t {
            uint32_t pass = 0;  // v[k + 2] = m[4p + k]
            if (packed_f1) {
                // two positions per instruction: D = sample pairs (v0,v1) .. (v6,v7), N = the pairs one sample on; '<' is the sign of a
                // saturating packed subtraction of the biased values, the level test a saturating packed add against 4 * noise_level
                // ((x + y) / 2 >= 2 n  <=>  x + y >= 4 n; the clamp at 65535 keeps that exact while 4 n <= 65535)
                const uint32_t d0 = lo.x ^ kBias2, d1 = lo.y ^ kBias2, d2 = hi.x ^ kBias2, d3 = hi.y ^ kBias2;
                const uint32_t n0 = __builtin_amdgcn_alignbit(d1, d0, 16), n1 = __builtin_amdgcn_alignbit(d2, d1, 16), n2 = __builtin_amdgcn_alignbit(d3, d2, 16);
                const uint32_t s01 = pk_add_sat_u16(lo.y, n1 ^ kBias2) ^ kBias2, s23 = pk_add_sat_u16(hi.x, n2 ^ kBias2) ^ kBias2;
                // positions 4p, 4p + 1: m[j-1] < m[j]: n0 < d1;  m[j+2] <= m[j]: !(d1 < d2);  m[j+2] <= m[j+1]: !(n1 < d2);  level: !(s01 < T)
                const uint32_t r01 = pk_sub_sat(n0, d1) & ~(pk_sub_sat(d1, d2) | pk_sub_sat(n1, d2) | pk_sub_sat(s01, level_thr));
                const uint32_t r23 = pk_sub_sat(n1, d2) & ~(pk_sub_sat(d2, d3) | pk_sub_sat(n2, d3) | pk_sub_sat(s23, level_thr));
                pass = ((r01 >> 15) & 1u) | ((r01 >> 30) & 2u) | ((r23 >> 13) & 4u) | ((r23 >> 28) & 8u);
                const unsigned f1_first = t0 + 4 * p;
                if (f1_first == 0) pass &= ~1u;  // the loop starts at 1 (:487)
                if (f1_first + 4 > blk_len) pass &= f1_first < blk_len ? (1u << (blk_len - f1_first)) - 1u : 0u;
            } else {
                const uint32_t v[8] = {lo.x & 0xffffu, lo.x >> 16, lo.y & 0xffffu, lo.y >> 16, hi.x & 0xffffu, hi.x >> 16, hi.y & 0xffffu, hi.y >> 16};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned f1_sample = t0 + 4 * p + k;
                    const bool ok = f1_sample != 0 && f1_sample < blk_len  // the loop starts at 1 (:487)
                                    && v[k + 1] < v[k + 2] && v[k + 4] <= v[k + 2] && v[k + 4] <= v[k + 3] && noise_level * 2 <= (v[k + 2] + v[k + 3]) / 2;
                    pass |= (uint32_t)ok << k;
                }
            }
            return pass;
        };
        uint32_t f1_bits = 0;  // bit 4 s + k: position 4 (tid + s * kAcThreads) + k of the tile
        {
            uint2 win[kF1Steps][2];
#pragma unroll
            for (int st = 0; st < kF1Steps; ++st) {
                win[st][0] = ((const uint2 *)mt)[tid + st * kAcThreads];
                win[st][1] = ((const uint2 *)mt)[tid + st * kAcThreads + 1];
            }
#pragma unroll
            for (int st = 0; st < kF1Steps; ++st) f1_bits |= f1_pass(win[st][0], win[st][1], tid + st * kAcThreads) << (4 * st);
        }
        if (!(a.debug_flags & 1024)) {
            // this lane's first slot: the counts (0..32) of the lanes below, bit plane by bit plane
            const uint32_t cnt = (uint32_t)__popc(f1_bits);
            uint32_t below = 0, total = 0;
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const uint64_t plane = __ballot((cnt >> b) & 1u);
                below += lanes_below(plane) << b;
                total += (uint32_t)__popcll(plane) << b;
            }
            if (total) {  // wave-uniform
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&sh.nq1, total);
                uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)base) + below;
                for (uint32_t rest = f1_bits; rest; rest &= rest - 1, ++slot) {
                    const uint32_t b = (uint32_t)__builtin_ctz(rest), j = 4 * (tid + (b >> 2) * kAcThreads) + (b & 3u);
                    if (slot < (uint32_t)kAcQ1)
                        sh.q1[slot] = (uint16_t)j;
                    else if (ac_f2_tests(m, j, t0 + j, noise_level))  // queue full: on the spot
                        ac_bits(out, m, j, t0 + j, blk_start, noise_level);
                }
            }
        }
        if ((a.debug_flags & 1024) && f1_bits == 0xdeadbeefu) sh.q2[tid] = 1;  // (ablation: the tests alone)
        __syncthreads();
        if (a.debug_flags & 64) continue;
        const uint32_t nq1 = min(sh.nq1, (uint32_t)kAcQ1);
        for (uint32_t cb = (tid & ~63u); cb < nq1; cb += kAcThreads) {  // F2 (:551-573)
            const uint32_t ci = cb + lane;
            uint32_t j = 0;
            bool pass = false;
            if (ci < nq1) {
                j = sh.q1[ci];
                pass = ac_f2_tests(m, j, t0 + j, noise_level);
            }
            const uint64_t pm = __ballot(pass);
            if (pm == 0) continue;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&sh.nq2, (uint32_t)__popcll(pm));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (pass) {
                const uint32_t slot = base + lanes_below(pm);
                if (slot < (uint32_t)kAcQ2)
                    sh.q2[slot] = (uint16_t)j;
                else
                    ac_bits(out, m, j, t0 + j, blk_start, noise_level);
            }
        }
        __syncthreads();
        if (a.debug_flags & 128) continue;
        const uint32_t nq2 = min(sh.nq2, (uint32_t)kAcQ2);
        for (uint32_t ci = tid; ci < nq2; ci += kAcThreads) {  // thresholds and the 20 bits (:575-639)
            const uint32_t j = sh.q2[ci];
            ac_bits(out, m, j, t0 + j, blk_start, noise_level);
        }
        __syncthreads();
    }
    }
}

// convert.c:226-262 / :343-379: the float converters accumulate float32 sums sample by sample, and float
// addition does not reassociate.  One wavefront per block: 64 lanes compute 64 samples' mag / magsq at
// a time, then the two running sums take them in sample order (v_readlane + v_add, the only serial part).
template <int FMT>
__global__ __launch_bounds__(64) void k_block_means_float(const void *iq, uint64_t n, uint64_t blk_len, double *means) {
    if constexpr (FMT == kFmtSC16 || FMT == kFmtSC16Q11) {
        const uint64_t start = (uint64_t)blockIdx.x * blk_len;
        const uint32_t len = (uint32_t)((n - start < blk_len) ? (n - start) : blk_len), lane = threadIdx.x;
        constexpr float scale = (FMT == kFmtSC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
        float sum_level = 0, sum_power = 0;
        for (uint32_t base = 0; base < len; base += 64) {
            float mag = 0, magsq = 0;
            if (base + lane < len) {
                const short2 v = ((const short2 *)iq)[start + base + lane];
                const float fI = (float)v.x * scale, fQ = (float)v.y * scale;
                magsq = fI * fI + fQ * fQ;
                if (magsq > 1.0f) magsq = 1.0f;
                mag = sqrt_rn(magsq);
            }
            const uint32_t cnt = (len - base < 64u) ? (len - base) : 64u;
            if (cnt == 64) {
#pragma unroll
                for (int k = 0; k < 64; ++k) {
                    sum_power += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, magsq), k));
                    sum_level += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mag), k));
                }
            } else {
                for (uint32_t k = 0; k < cnt; ++k) {
                    sum_power += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, magsq), (int)k));
                    sum_level += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mag), (int)k));
                }
            }
        }
        if (lane == 0) {  // "sum / nsamples" is a float division there (float / unsigned), widened afterwards
            means[2 * blockIdx.x] = (double)(sum_level / (float)len);
            means[2 * blockIdx.x + 1] = (double)(sum_power / (float)len);
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

// convert_sc16q11_table (convert.c:297-328), the SC16Q11 converter of a reference built with -DSC16Q11_TABLE_BITS=use_bits:
// |I|, |Q| masked to 11 bits, their low 11 - use_bits bits dropped, magnitude from a table whose entries are the float32
// formula of convert.c:281-291 at the kept bits -- computed here directly (float_mag is that formula, op for op; i / 2048.0
// is exact), so no table has to live anywhere.  Exact integer sums when asked, as launch_convert.
__global__ __launch_bounds__(kThreads) void k_convert_q11t(const void *iq, uint16_t *mag, uint64_t n, int lose_bits, uint64_t *sums) {
    uint64_t sl = 0, sp = 0;
    const uint32_t *in = (const uint32_t *)iq;  // one sample = I | Q << 16
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kThreads) {
        const uint32_t w = in[i];
        int I = (int)(short)(w & 0xffffu), Q = (int)(short)(w >> 16);
        I = ((I < 0 ? -I : I) & 2047) >> lose_bits << lose_bits;  // abs((int16_t) x) & 2047: -32768 and +-2048 give 0 (convert.c:313-314)
        Q = ((Q < 0 ? -Q : Q) & 2047) >> lose_bits << lose_bits;
        const uint32_t v = float_mag((float)I * (1.0f / 2048.0f), (float)Q * (1.0f / 2048.0f));
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

// ------------------------------------------------------------------------------------------------
// --dcfilter: the generic converters of convert.c:113-162 / :164-213 / :374-423.  The DC block is a first-order
// recurrence z[n] = x[n] * dc_a + z[n-1] * dc_b per component with the state carried across calls (:152-153), which the
// reference runs sample by sample in float32.  Here it is a chunked linear-recurrence scan: (1) every tile of 4096
// samples is run from a zero state, 16 samples per thread, and reduced with the powers of dc_b to the tile's
// contribution; (2) one workgroup scans the tiles; (3) every thread gets its exact starting state from its tile's and
// runs its 16 samples like the reference does (float32, every product and sum rounded on its own).  The scan itself
// is in double, so a thread's starting state differs from the reference's float32 history only in the last bits:
// magnitudes agree to +-1 LSB (tests/test_gpu_parity.py states the tolerance), not bit for bit.

constexpr int kDcPerThread = 16, kDcTile = kThreads * kDcPerThread;  // 4096 samples; kBlock % kDcTile == 0

// scan over the workgroup's threads of the per-thread contributions: incl[t] = sum_{u <= t} m^(t-u) * own[u] (m = dc_b^16), and
// excl[t] = incl[t - 1] (0 for t = 0).  Through LDS with barriers: a shuffle-based scan (64-bit __shfl_up = two ds_bpermute
// each) measured three times slower than this on MI355X.
struct DcScan {
    double2 incl, excl;
};
__device__ __forceinline__ DcScan dc_block_scan(double2 own, double m, double2 *sh) {
    const uint32_t t = threadIdx.x;
    sh[t] = own;
    __syncthreads();
    double w = m;
    for (uint32_t d = 1; d < (uint32_t)kThreads; d <<= 1) {
        double2 add = make_double2(0, 0);
        if (t >= d) add = sh[t - d];
        __syncthreads();
        if (t >= d) {
            own.x += w * add.x;
            own.y += w * add.y;
            sh[t] = own;
        }
        __syncthreads();
        w *= w;
    }
    DcScan r;
    r.incl = own;
    r.excl = t ? sh[t - 1] : make_double2(0, 0);
    return r;
}

__device__ __forceinline__ void dc_uc8_table(float *uc8f) {  // before the first __syncthreads of the kernel
    for (uint32_t v = threadIdx.x; v < 256; v += kThreads) uc8f[v] = ((float)v - 127.5f) / 127.5f;
}

// A thread keeps its 16 samples as the raw words it loaded and decodes them in both loops: holding them as 32 floats, with the
// 16 square roots scheduled side by side, took 248 VGPRs (one wavefront per SIMD).
template <int FMT>
struct DcRaw {
    static constexpr int kWords = (FMT == kFmtUC8) ? kDcPerThread / 2 : kDcPerThread;
    uint32_t w[kWords];
    __device__ __forceinline__ void load(const void *iq, uint64_t first, uint64_t n, bool aligned) {
        constexpr int bps = (FMT == kFmtUC8) ? 2 : 4;
        if (aligned && first + kDcPerThread <= n) {
            const uint4 *p = (const uint4 *)((const uint8_t *)iq + bps * first);
#pragma unroll
            for (int v = 0; v < kWords / 4; ++v) {
                const uint4 x = p[v];
                w[4 * v] = x.x; w[4 * v + 1] = x.y; w[4 * v + 2] = x.z; w[4 * v + 3] = x.w;
            }
        } else {  // the recording's last tile, unaligned input: sample by sample (what lies beyond n is never used)
#pragma unroll
            for (int k = 0; k < kWords; ++k) w[k] = 0;
#pragma unroll
            for (int k = 0; k < kDcPerThread; ++k) {
                if (first + k >= n) continue;
                if constexpr (FMT == kFmtUC8)
                    w[k >> 1] |= (uint32_t)((const uint16_t *)iq)[first + k] << (16 * (k & 1));
                else
                    w[k] = ((const uint32_t *)iq)[first + k];
            }
        }
    }
    // uc8f: the 256 values ((float)v - 127.5f) / 127.5f of convert.c:131-132 in LDS (dc_uc8_table): a float division costs more than a lookup
    __device__ __forceinline__ void sample(int k, const float *uc8f, float &fI, float &fQ) const {
        if constexpr (FMT == kFmtUC8) {
            const uint32_t x = w[k >> 1] >> (16 * (k & 1));
            fI = uc8f[x & 0xffu];
            fQ = uc8f[(x >> 8) & 0xffu];
        } else {
            constexpr float scale = (FMT == kFmtSC16) ? (1.0f / 32768.0f) : (1.0f / 2048.0f);  // exact: a power of two
            fI = (float)(short)(w[k] & 0xffffu) * scale;
            fQ = (float)(short)(w[k] >> 16) * scale;
        }
    }
};

// phase 1: the state each tile would end in when started from zero
template <int FMT>
__global__ __launch_bounds__(kThreads) void k_dc_tile_sums(const void *iq, uint64_t n, double dc_a, double dc_b, double2 *tile_sums, float2 *local) {
    __shared__ double2 sh[kThreads];
    __shared__ float uc8f[256];
    if constexpr (FMT == kFmtUC8) {
        dc_uc8_table(uc8f);
        __syncthreads();
    }
    const uint64_t first = (uint64_t)blockIdx.x * kDcTile + (uint64_t)threadIdx.x * kDcPerThread;
    DcRaw<FMT> raw;
    raw.load(iq, first, n, (uintptr_t)iq % 16 == 0);
    double zI = 0, zQ = 0, midI = 0, midQ = 0;
#pragma unroll
    for (int k = 0; k < kDcPerThread; ++k) {  // (samples beyond the end of the call only reach a sum that no later tile uses)
        float fI, fQ;
        raw.sample(k, uc8f, fI, fQ);
        zI = (double)fI * dc_a + zI * dc_b;
        zQ = (double)fQ * dc_a + zQ * dc_b;
        if (k == 7) {
            midI = zI;
            midQ = zQ;
        }
    }
    double m8 = dc_b;
    for (int k = 0; k < 3; ++k) m8 *= m8;  // dc_b^8
    const double m16 = m8 * m8;
    const DcScan v = dc_block_scan(make_double2(zI, zQ), m16, sh);
    if (threadIdx.x == kThreads - 1) tile_sums[blockIdx.x] = v.incl;
    if (local) {
        // the DC block inside the scan (dc_state_before_chunk): the state before each of the thread's two 8-sample chunks, had the
        // tile started from zero; what the tile really starts with is added where the state is used
        static_assert(kDcPerThread == 16, "two chunks per thread");
        const float2 c0 = make_float2((float)v.excl.x, (float)v.excl.y);
        const float2 c1 = make_float2((float)(v.excl.x * m8 + midI), (float)(v.excl.y * m8 + midQ));
        ((float4 *)local)[(size_t)blockIdx.x * kThreads + threadIdx.x] = make_float4(c0.x, c0.y, c1.x, c1.y);
    }
}

// phase 2: state at the start of every tile; state[0] = the carried-in state, state[ntiles] = the state to carry out.
// Two launches over chunks of 1024 tiles (one workgroup each; round 2's single workgroup walked 256 tiles per thread with
// 4 KB between neighbouring lanes' loads and took 0.56 ms per Gsample): (a) scan of the chunk's tiles from a zero state ->
// tile_start[i] relative to the chunk's start, the chunk's total into the chunk's first tile_sums entry (already consumed);
// (b) every workgroup runs the (<= 1024) chunk totals before it to its own starting state and adds that state's decay.
constexpr int kDcChunkTiles = 1024;
__device__ __forceinline__ double dc_tile_multiplier(double dc_b) {
    double mt = dc_b;
    for (int k = 0; k < 12; ++k) mt *= mt;  // dc_b^4096: one tile
    return mt;
}
static_assert(kDcTile == 4096, "dc_tile_multiplier squares twelve times");

__global__ __launch_bounds__(kDcChunkTiles) void k_dc_scan_local(double2 *tile_sums, double2 *tile_start, uint32_t ntiles, double dc_b) {
    __shared__ double2 sh[kDcChunkTiles];
    const uint32_t t = threadIdx.x, first = blockIdx.x * kDcChunkTiles, i = first + t;
    double2 own = i < ntiles ? tile_sums[i] : make_double2(0, 0);
    sh[t] = own;
    __syncthreads();
    double w = dc_tile_multiplier(dc_b);
    for (uint32_t d = 1; d < (uint32_t)kDcChunkTiles; d <<= 1) {  // own = sum_{u <= t} mt^(t-u) sums[first + u]
        double2 add = make_double2(0, 0);
        if (t >= d) add = sh[t - d];
        __syncthreads();
        if (t >= d) {
            own.x += w * add.x;
            own.y += w * add.y;
            sh[t] = own;
        }
        __syncthreads();
        w *= w;
    }
    if (i < ntiles) tile_start[i] = t ? sh[t - 1] : make_double2(0, 0);
    const uint32_t live = ntiles - first < (uint32_t)kDcChunkTiles ? ntiles - first : (uint32_t)kDcChunkTiles;
    if (t == live - 1) tile_sums[first] = own;  // the chunk from a zero state, after its last tile
}

__global__ __launch_bounds__(kDcChunkTiles) void k_dc_scan_finish(const double2 *tile_sums, double2 *tile_start, uint32_t ntiles, const double2 *z_in, double dc_b) {
    __shared__ double2 tot[kDcChunkTiles];
    __shared__ double2 zc;
    const uint32_t t = threadIdx.x, g = blockIdx.x, first = g * kDcChunkTiles, i = first + t;
    if (t < g) tot[t] = tile_sums[(size_t)t * kDcChunkTiles];  // (at most 1024 chunks: 2^32 samples)
    __syncthreads();
    const double mt = dc_tile_multiplier(dc_b);
    if (t == 0) {
        double mc = mt;
        for (int k = 0; k < 10; ++k) mc *= mc;  // mt^1024: one chunk
        double2 z = *z_in;
        for (uint32_t u = 0; u < g; ++u) {
            z.x = z.x * mc + tot[u].x;
            z.y = z.y * mc + tot[u].y;
        }
        zc = z;
    }
    __syncthreads();
    double carry = 1.0, sq = mt;  // mt^t by the bits of t
    for (uint32_t k = t; k; k >>= 1, sq *= sq)
        if (k & 1) carry *= sq;
    const double2 z0 = zc;
    if (i < ntiles) {
        const double2 local = tile_start[i];
        tile_start[i] = make_double2(z0.x * carry + local.x, z0.y * carry + local.y);
    }
    if (g == gridDim.x - 1 && t == ntiles - first - 1) {  // the state to carry out: after the last tile
        const double2 last = tile_sums[first];
        tile_start[ntiles] = make_double2(z0.x * carry * mt + last.x, z0.y * carry * mt + last.y);
    }
}
static_assert(0xfffff000ull / kDcTile / kDcChunkTiles < (unsigned long long)kDcChunkTiles, "k_dc_scan_finish holds every chunk total of the largest batch");

// phase 3: magnitudes.  sums (optional): per 131072-sample block of this call, sum(mag) and sum(magsq) as doubles.
template <int FMT>
__global__ __launch_bounds__(kThreads) void k_convert_dc(const void *iq, uint16_t *mag, uint64_t n, float dc_a, float dc_b, const double2 *tile_start, double *sums,
                                                         double2 *tile_part, double2 *z_out, int abl) {
    __shared__ double2 sh[kThreads];
    __shared__ uint4 stage[kDcTile / 8];
    __shared__ float uc8f[256];
    if constexpr (FMT == kFmtUC8) {
        dc_uc8_table(uc8f);
        __syncthreads();
    }
    const uint64_t tile_first = (uint64_t)blockIdx.x * kDcTile, first = tile_first + (uint64_t)threadIdx.x * kDcPerThread;
    DcRaw<FMT> raw;
    raw.load(iq, first, n, (uintptr_t)iq % 16 == 0);
    double zI = 0, zQ = 0;
#pragma unroll
    for (int k = 0; k < kDcPerThread; ++k) {
        float fI, fQ;
        raw.sample(k, uc8f, fI, fQ);
        zI = (double)fI * (double)dc_a + zI * (double)dc_b;
        zQ = (double)fQ * (double)dc_a + zQ * (double)dc_b;
    }
    double m16 = (double)dc_b;
    for (int k = 0; k < 4; ++k) m16 *= m16;
    DcScan sc;
    sc.excl = sc.incl = make_double2(zI, zQ);
    if (!(abl & 1)) sc = dc_block_scan(make_double2(zI, zQ), m16, sh);  // tile started from zero: state after / before this thread's samples
    // state before this thread's samples: the tile's start carried over 16 t samples, plus the threads before
    const double2 z0 = tile_start[blockIdx.x];
    double carry = 1.0, sq = m16;  // m16^t by the bits of t
    for (uint32_t t = threadIdx.x; t; t >>= 1, sq *= sq)
        if (t & 1) carry *= sq;
    const double2 before = make_double2(z0.x * carry + sc.excl.x, z0.y * carry + sc.excl.y);
    float z1_I = (float)before.x, z1_Q = (float)before.y;
    double sl = 0, sp = 0;
    // Full tiles leave through LDS so that a wavefront's stores cover whole cache lines: the tile's 8 KB go out as 2 x 256
    // coalesced 16-byte stores.  The recording's last tile stores sample by sample.
    const bool tile_out = (uintptr_t)mag % 16 == 0 && tile_first + kDcTile <= n;  // workgroup-uniform
    const int live = (n - first >= (uint64_t)kDcPerThread || first >= n) ? (first >= n ? 0 : kDcPerThread) : (int)(n - first);  // samples of this thread inside the call
    const int last = (first < n && n - first <= (uint64_t)kDcPerThread) ? (int)(n - 1 - first) : -1;                       // ... and the call's last sample, if it is here
    uint32_t packed[kDcPerThread / 2];
#pragma unroll
    for (int k = 0; k < kDcPerThread; ++k) {
        float xI, xQ;
        raw.sample(k, uc8f, xI, xQ);
        z1_I = xI * dc_a + z1_I * dc_b;  // convert.c:137-140
        z1_Q = xQ * dc_a + z1_Q * dc_b;
        xI -= z1_I;
        xQ -= z1_Q;
        float magsq = xI * xI + xQ * xQ;
        if (magsq > 1.0f) magsq = 1.0f;
        const float m = (abl & 4) ? magsq : sqrt_rn(magsq);
        const uint32_t mv = (uint32_t)(uint16_t)(m * 65535.0f + 0.5f);
        packed[k >> 1] = (k & 1) ? (packed[k >> 1] | (mv << 16)) : mv;
        if (!tile_out && k < live) mag[first + k] = (uint16_t)mv;
        if (sums && k < live) {
            sl += (double)m;
            sp += (double)magsq;
        }
        if (k == last) *z_out = make_double2((double)z1_I, (double)z1_Q);  // converter_state carried to the next call
        __builtin_amdgcn_sched_barrier(0);  // one sample after the other (register pressure, see above)
    }
    if (tile_out && !(abl & 2)) {
        stage[2 * threadIdx.x] = make_uint4(packed[0], packed[1], packed[2], packed[3]);
        stage[2 * threadIdx.x + 1] = make_uint4(packed[4], packed[5], packed[6], packed[7]);
        __syncthreads();
        uint4 *o = (uint4 *)(mag + tile_first);
        o[threadIdx.x] = stage[threadIdx.x];
        o[threadIdx.x + kThreads] = stage[threadIdx.x + kThreads];
    }
    if (sums) {
        // the tile's two sums in a fixed order (lanes by halving, then the four wavefronts in turn): k_dc_means adds the tiles of a
        // block in ascending order, so a block's means are the same bits on every run (atomics here added in whatever order the
        // wavefronts arrived)
        for (int off = 32; off > 0; off >>= 1) {
            sl += __shfl_down(sl, off, 64);
            sp += __shfl_down(sp, off, 64);
        }
        __syncthreads();  // (sh: the scan above is done with it)
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = make_double2(sl, sp);
        __syncthreads();
        if (threadIdx.x == 0) {
            double2 t = sh[0];
            for (int w = 1; w < kThreads / 64; ++w) {
                t.x += sh[w].x;
                t.y += sh[w].y;
            }
            tile_part[blockIdx.x] = t;
        }
    }
}

// block sums -> the converter's per-call means (float / unsigned division widened afterwards, convert.c:155-161)
__global__ void k_dc_means(double *sums, const double2 *tile_part, uint32_t ntiles, uint64_t n, uint32_t nblocks) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const uint64_t start = (uint64_t)b * kBlock;
    const double len = (double)((n - start < (uint64_t)kBlock) ? (n - start) : (uint64_t)kBlock);
    constexpr uint32_t per = kBlock / kDcTile;  // 32 tiles per block
    double sl = 0, sp = 0;
    for (uint32_t t = b * per; t < (b + 1) * per && t < ntiles; ++t) {
        sl += tile_part[t].x;
        sp += tile_part[t].y;
    }
    sums[2 * b] = sl / len;
    sums[2 * b + 1] = sp / len;
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


}  // namespace

int scan_grid_size(int device) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return 256;
    return p.multiProcessorCount;  // one 1024-thread workgroup (~158 KB of LDS) per CU
}

hipError_t launch_scan(const BatchArgs &a, int grid, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    const uint64_t nstrips = (a.n + kTileShift + kStrip - 1) / kStrip;
    const uint64_t need = (nstrips + kScanWaves - 1) / kScanWaves;
    if ((uint64_t)grid > need) grid = (int)need;
    return by_format(a.fmt, [&](auto F) {
        constexpr int f = decltype(F)::value;
        if constexpr (f != kFmtMag16) {
            if (a.dc_local) {  // the DC block inside the scan
                auto kern = k_scan<f, true>;
                hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ScanShared));
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(kern, dim3(grid), dim3(kScanThreads), sizeof(ScanShared), s, a);
                return hipGetLastError();
            }
        }
        if (a.dc_local) return hipErrorInvalidValue;
        auto kern = k_scan<f>;
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ScanShared));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kScanThreads), sizeof(ScanShared), s, a);
        return hipGetLastError();
    });
}

hipError_t launch_scan_many(int fmt, const BatchArgs *args_dev, uint32_t nbatches, uint32_t groups_per_batch, hipStream_t s) {
    if (nbatches == 0 || groups_per_batch == 0) return hipSuccess;
    return by_format(fmt, [&](auto F) {
        auto kern = k_scan_many<decltype(F)::value>;
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(ScanShared));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(nbatches * groups_per_batch), dim3(kScanThreads), sizeof(ScanShared), s, args_dev, groups_per_batch);
        return hipGetLastError();
    });
}

// what rsb_submit's memsets and carry copy do, for every batch of a launch_scan_many at once (64 memsets and 32 copies otherwise)
__device__ __forceinline__ void submit_prepare_body(const BatchArgs &a);
__global__ __launch_bounds__(kThreads) void k_submit_prepare_many(const BatchArgs *__restrict__ args) { submit_prepare_body(args[blockIdx.y]); }
// (one batch, arguments by value: rsb_submit's two fills and its carry copy as one launch -- three operations fewer in the queue between
// the chain of the batch before and this batch's scan)
__global__ __launch_bounds__(kThreads) void k_submit_prepare(const BatchArgs a) { submit_prepare_body(a); }
__device__ __forceinline__ void submit_prepare_body(const BatchArgs &a) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i < (uint32_t)kCntAlloc) a.counters[i] = 0;
    for (uint32_t r = i; r < a.n_regions; r += gridDim.x * kThreads) a.region_counts[r] = 0;
    if (a.carry_out && a.n >= (uint64_t)kTrailing) {  // sdr_ifile.c:199-201: the next batch starts with this one's last 326 samples
        const uint32_t halves = (a.fmt == kFmtUC8 || a.fmt == kFmtMag16) ? kTrailing : 2 * kTrailing;  // 16-bit units
        const uint16_t *src = (const uint16_t *)a.iq + (a.n - kTrailing) * (halves / kTrailing);
        for (uint32_t k = i; k < halves; k += gridDim.x * kThreads) ((uint16_t *)a.carry_out)[k] = src[k];
    }
}
hipError_t launch_submit_prepare(const BatchArgs &a, hipStream_t s) {
    unsigned gx = (a.n_regions + kThreads - 1) / kThreads;
    if (gx < 1) gx = 1;
    if (gx > 4) gx = 4;
    hipLaunchKernelGGL(k_submit_prepare, dim3(gx), dim3(kThreads), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_submit_prepare_many(const BatchArgs *args_dev, uint32_t nbatches, uint32_t max_regions, hipStream_t s) {
    if (nbatches == 0) return hipSuccess;
    unsigned gx = (max_regions + kThreads - 1) / kThreads;
    if (gx < 1) gx = 1;
    if (gx > 4) gx = 4;
    hipLaunchKernelGGL(k_submit_prepare_many, dim3(gx, nbatches), dim3(kThreads), 0, s, args_dev);
    return hipGetLastError();
}

hipError_t launch_prune(const BatchArgs &a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_prune, dim3(a.n_regions + 1), dim3(kThreads), 0, s, a);
    return hipGetLastError();
}

// the kept items packed in stream order (already sorted: no launch_sort behind it).  Valid when nothing took the scan kernel's
// per-thread paths (counters[kCntItems] == 0: the shared region is empty) and no marker items were asked for.
// RSB_PRUNE_FOUR_PASSES=1 (A/B runs): keep masks, counts per chunk, offsets, write -- as until late in round 3
static bool prune_one_pass() {
    const char *e = std::getenv("RSB_PRUNE_FOUR_PASSES");
    return !(e && std::atoi(e) != 0);
}
static unsigned tiles_of(uint32_t n_chunks) { return n_chunks ? (n_chunks + kTileChunks - 1) / kTileChunks : 1u; }

hipError_t launch_prune_ordered(const BatchArgs &a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    if (prune_one_pass()) {
        const unsigned tiles = tiles_of(a.n_chunks), clear = (tiles + kThreads - 1) / kThreads;
        hipLaunchKernelGGL(k_prune_tiles_clear, dim3(clear > 64 ? 64 : clear), dim3(kThreads), 0, s, a);
        if (a.adder_summary) hipLaunchKernelGGL(k_prune_summary, dim3((1u << 24) / 32 / 4 / kThreads), dim3(kThreads), 0, s, a.adder_bitmap, a.adder_summary);
        hipLaunchKernelGGL(k_prune_onepass, dim3(tiles), dim3(kTileThreads), 0, s, a);
        return hipGetLastError();
    }
    if (a.adder_summary) {
        const hipError_t e = hipMemsetAsync(a.adder_summary, 0, kSummaryWords * sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_prune_summary, dim3((1u << 24) / 32 / 4 / kThreads), dim3(kThreads), 0, s, a.adder_bitmap, a.adder_summary);
    }
    hipLaunchKernelGGL(k_prune_count, dim3(a.n_regions), dim3(kThreads), 0, s, a);
    hipLaunchKernelGGL(k_prune_chunks, dim3((a.n_chunks + kChunkGroup - 1) / kChunkGroup), dim3(kChunkGroup), 0, s, a);
    hipLaunchKernelGGL(k_prune_offsets, dim3(1), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(k_prune_write, dim3((a.n_chunks + kThreads / kWriteLanes - 1) / (kThreads / kWriteLanes)), dim3(kThreads), 0, s, a);
    return hipGetLastError();
}

size_t sort_temp_bytes(uint32_t max_items) {
    // work memory of the fallback sort (rocPRIM radix sort of 35-bit keys), the unique step behind it, and the wire encoder's prefix sum
    size_t sort_bytes = 0, uniq_bytes = 0, scan_bytes = 0;
    rocprim::double_buffer<uint64_t> k(nullptr, nullptr);
    rocprim::double_buffer<uint32_t> v(nullptr, nullptr);
    (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, k, v, (size_t)max_items, 0u, 35u);
    (void)rocprim::unique(nullptr, uniq_bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)max_items);
    (void)rocprim::exclusive_scan(nullptr, scan_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, 0u, (size_t)max_items + 1, rocprim::plus<uint32_t>());
    size_t m = sort_bytes > uniq_bytes ? sort_bytes : uniq_bytes;
    if (scan_bytes > m) m = scan_bytes;
    return m + 256;
}

hipError_t launch_sort(const BatchArgs &a, uint32_t n_items, void *temp, size_t temp_bytes, int *sorted_buf, hipStream_t s) {
    *sorted_buf = 0;
    if (n_items == 0) return hipSuccess;
    rocprim::double_buffer<uint64_t> k(a.skeys[0], a.skeys[1]);
    rocprim::double_buffer<uint32_t> v(a.sinfo[0], a.sinfo[1]);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, k, v, (size_t)n_items, 0u, 35u, s);
    if (e != hipSuccess) return e;
    *sorted_buf = (k.current() == a.skeys[1]) ? 1 : 0;
    return hipSuccess;
}

hipError_t launch_unique_finalize(const BatchArgs &a, uint32_t n_items, int sorted_buf, void *temp, size_t temp_bytes, hipStream_t s) {
    if (n_items == 0) return hipMemsetAsync(&a.counters[kCntSurvivors], 0, sizeof(uint32_t), s);
    // distinct offsets, ascending: the other key buffer is free now and serves as scratch for g
    uint32_t *gtmp = (uint32_t *)a.skeys[sorted_buf ^ 1];
    hipLaunchKernelGGL(k_keys_to_g, dim3(512), dim3(kThreads), 0, s, a.skeys[sorted_buf], gtmp, n_items);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = rocprim::unique(temp, temp_bytes, gtmp, a.surv_g, &a.counters[kCntSurvivors], (size_t)n_items, rocprim::equal_to<uint32_t>(), s);
    if (e != hipSuccess) return e;
    return by_format(a.fmt, [&](auto F) {
        hipLaunchKernelGGL(k_finalize<decltype(F)::value>, dim3(8192), dim3(64), 0, s, a);
        return hipGetLastError();
    });
}

namespace {
inline unsigned item_grid(uint32_t n) {
    const unsigned g = (n + kThreads - 1) / kThreads;
    return g < 1 ? 1 : (g > 2048 ? 2048 : g);
}
}  // namespace

namespace {
inline unsigned run_grid(uint32_t n) {  // k_replay_table_min: one workgroup per run of kMinItemsPerGroup items
    const unsigned g = (n + kMinItemsPerGroup - 1) / kMinItemsPerGroup;
    return g < 1 ? 1 : g;
}
}  // namespace

// RSB_PESSIMISTIC_START=1 (tests): the fixed-point iteration starts from the filter as it was before the batch instead of from the
// optimistic seed, so every batch that teaches the filter an address goes through the correction rounds
static bool pessimistic_start() {
    const char *e = std::getenv("RSB_PESSIMISTIC_START");
    return e && std::atoi(e) != 0;
}

namespace {
inline unsigned bounds_blocks(uint32_t n) {  // k_replay_seed_bounds: 1024-thread workgroups for the blocks' first items
    const unsigned g = (n + kMinThreads) / kMinThreads;
    return g > 512 ? 512 : g;
}
}  // namespace

hipError_t launch_replay_bounds(const ReplayArgs &r, hipStream_t s) {
    if (r.n_items && r.t_cur && !pessimistic_start()) {
        const unsigned runs = run_grid(r.n_items);
        hipLaunchKernelGGL(k_replay_seed_bounds, dim3(runs + bounds_blocks(r.n_items)), dim3(kMinThreads), 0, s, r, runs);
    } else {
        hipLaunchKernelGGL(k_replay_bounds, dim3(item_grid(r.n_items + 1)), dim3(kThreads), 0, s, r);
    }
    return hipGetLastError();
}

hipError_t launch_replay_iteration(const BatchArgs &a, const ReplayArgs &r, hipStream_t s) {
    if (r.n_items) hipLaunchKernelGGL(k_replay_score, dim3(item_grid(r.n_items)), dim3(kThreads), 0, s, r);  // (clears counters[kCntChanged])
    hipLaunchKernelGGL(k_replay_walk, dim3((r.nblocks + kThreads / 64 - 1) / (kThreads / 64)), dim3(kThreads), 0, s, r);
    if (r.n_items) hipLaunchKernelGGL(k_replay_table_min<false>, dim3(run_grid(r.n_items)), dim3(kMinThreads), 0, s, r);
    if (r.n_items) hipLaunchKernelGGL(k_replay_compare, dim3(item_grid(r.n_items)), dim3(kThreads), 0, s, r, &a.counters[kCntChanged]);
    return hipGetLastError();
}

// RSB_NO_LIST_CHECK=1 (A/B runs): the first round of a chain builds the next table and compares, as every later round does
static bool list_checks() {
    const char *e = std::getenv("RSB_NO_LIST_CHECK");
    return !(e && std::atoi(e) != 0);
}

// The first round of a chain, with the list behind it: "did the table move" is answered by k_replay_list<true> from the walk's result
// and the current table alone, so the next table is not built unless the answer is yes (launch_replay_next_table).
hipError_t launch_replay_first_round(const BatchArgs &a, const ReplayArgs &r, hipStream_t s) {
    if (!list_checks()) {
        const hipError_t e = launch_replay_iteration(a, r, s);
        return e != hipSuccess ? e : launch_replay_select(a, r, nullptr, 0, s);
    }
    if (r.n_items) hipLaunchKernelGGL(k_replay_score, dim3(item_grid(r.n_items)), dim3(kThreads), 0, s, r);
    hipLaunchKernelGGL(k_replay_walk, dim3((r.nblocks + kThreads / 64 - 1) / (kThreads / 64)), dim3(kThreads), 0, s, r);
    hipLaunchKernelGGL(k_replay_blkbase, dim3(1), dim3(1024), 0, s, r);
    if (r.n_items) hipLaunchKernelGGL(k_replay_list<true>, dim3(item_grid(r.n_items)), dim3(kThreads), 0, s, r, a.counters);
    return hipGetLastError();
}

// the table the first round's walk produces (r.acc still holds it), for a batch whose first round was not the fixed point
hipError_t launch_replay_next_table(const ReplayArgs &r, hipStream_t s) {
    if (r.n_items) hipLaunchKernelGGL(k_replay_table_min<false>, dim3(run_grid(r.n_items)), dim3(kMinThreads), 0, s, r);
    return hipGetLastError();
}

hipError_t launch_replay_reset(const BatchArgs &, const ReplayArgs &r, uint32_t *table, hipStream_t s) {
    if (r.n_items) hipLaunchKernelGGL(k_replay_reset, dim3(item_grid(r.n_items)), dim3(kThreads), 0, s, r, table);
    return hipGetLastError();
}

hipError_t launch_replay_select(const BatchArgs &a, const ReplayArgs &r, void *, size_t, hipStream_t s) {
    hipLaunchKernelGGL(k_replay_blkbase, dim3(1), dim3(1024), 0, s, r);
    if (r.n_items) hipLaunchKernelGGL(k_replay_list<false>, dim3(item_grid(r.n_items)), dim3(kThreads), 0, s, r, a.counters);
    return hipGetLastError();
}

hipError_t launch_replay_fold(const ReplayArgs &r, hipStream_t s) {
    if (r.n_items) hipLaunchKernelGGL(k_replay_fold, dim3(item_grid(r.n_items)), dim3(kThreads), 0, s, r);
    return hipGetLastError();
}

hipError_t launch_replay_emit(const BatchArgs &a, const ReplayArgs &r, hipStream_t s) {
    if (r.n_items == 0) return hipSuccess;
    return by_format(a.fmt, [&](auto F) {
        constexpr int f = decltype(F)::value;
        // r.n_items bounds the KEPT ITEMS; messages are a third of them or fewer, and every workgroup pays a 32 KB table copy: a
        // grid for a quarter of the bound (the kernel strides over whatever the count turns out to be)
        const unsigned g = (r.n_items / 4 + 64 * kEmitWaves - 1) / (64 * kEmitWaves);
        const dim3 grid(g < 1 ? 1 : (g > (unsigned)kEmitGroups ? (unsigned)kEmitGroups : g));
        if constexpr (f != kFmtMag16) {
            if (a.dc_local) {  // the DC block inside the kernel
                auto kern = k_replay_emit<f, true>;
                const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(EmitShared));
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(kern, grid, dim3(64 * kEmitWaves), sizeof(EmitShared), s, a, r);
                return hipGetLastError();
            }
        }
        auto kern = k_replay_emit<f>;
        const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(EmitShared));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(64 * kEmitWaves), sizeof(EmitShared), s, a, r);
        return hipGetLastError();
    });
}

// One launch per stage for all the channels' batches (args_dev[i] = channel i's {BatchArgs, ReplayArgs}; d gives the grids' extents).
hipError_t launch_chain_prune_many(const ChainArgs *args_dev, const ChainDims &d, hipStream_t s) {
    if (d.nch == 0) return hipSuccess;
    if (prune_one_pass()) {
        const unsigned tiles = tiles_of(d.max_chunks), clear = (tiles + kThreads - 1) / kThreads;
        hipLaunchKernelGGL(k_prune_tiles_clear_many, dim3(clear > 64 ? 64 : clear, d.nch), dim3(kThreads), 0, s, args_dev);
        if (d.any_summary) hipLaunchKernelGGL(k_prune_summary_many, dim3((1u << 24) / 32 / 4 / kThreads, d.nch), dim3(kThreads), 0, s, args_dev);
        hipLaunchKernelGGL(k_prune_onepass_many, dim3(tiles, d.nch), dim3(kTileThreads), 0, s, args_dev);
        return hipGetLastError();
    }
    if (d.any_summary) {
        hipLaunchKernelGGL(k_prune_summary_clear_many, dim3(1, d.nch), dim3(kThreads), 0, s, args_dev);
        hipLaunchKernelGGL(k_prune_summary_many, dim3((1u << 24) / 32 / 4 / kThreads, d.nch), dim3(kThreads), 0, s, args_dev);
    }
    const unsigned chunks = d.max_chunks ? d.max_chunks : 1u;
    hipLaunchKernelGGL(k_prune_count_many, dim3(d.max_regions ? d.max_regions : 1u, d.nch), dim3(kThreads), 0, s, args_dev);
    hipLaunchKernelGGL(k_prune_chunks_many, dim3((chunks + kChunkGroup - 1) / kChunkGroup, d.nch), dim3(kChunkGroup), 0, s, args_dev);
    hipLaunchKernelGGL(k_prune_offsets_many, dim3(1, d.nch), dim3(1024), 0, s, args_dev);
    hipLaunchKernelGGL(k_prune_write_many, dim3((chunks + kThreads / kWriteLanes - 1) / (kThreads / kWriteLanes), d.nch), dim3(kThreads), 0, s, args_dev);
    return hipGetLastError();
}

hipError_t launch_chain_replay_many(int fmt, const ChainArgs *args_dev, const ChainDims &d, uint32_t *readback_dev, uint32_t readback_stride, uint32_t pin_adders,
                                    hipStream_t s) {
    if (d.nch == 0) return hipSuccess;
    const unsigned items = item_grid(d.max_items), items1 = item_grid(d.max_items + 1), runs = run_grid(d.max_items);
    if (!pessimistic_start())
        hipLaunchKernelGGL(k_replay_seed_bounds_many, dim3(runs + bounds_blocks(d.max_items), d.nch), dim3(kMinThreads), 0, s, args_dev, runs);
    else
        hipLaunchKernelGGL(k_replay_bounds_many, dim3(items1, d.nch), dim3(kThreads), 0, s, args_dev);
    hipLaunchKernelGGL(k_replay_score_many, dim3(items, d.nch), dim3(kThreads), 0, s, args_dev);
    hipLaunchKernelGGL(k_replay_walk_many, dim3((d.max_nblocks + kThreads / 64 - 1) / (kThreads / 64), d.nch), dim3(kThreads), 0, s, args_dev);
    if (!list_checks()) {
        hipLaunchKernelGGL(k_replay_table_min_many<false>, dim3(runs, d.nch), dim3(kMinThreads), 0, s, args_dev);
        hipLaunchKernelGGL(k_replay_compare_many, dim3(items, d.nch), dim3(kThreads), 0, s, args_dev);
    }
    hipLaunchKernelGGL(k_replay_blkbase_many, dim3(1, d.nch), dim3(1024), 0, s, args_dev);
    if (list_checks())
        hipLaunchKernelGGL(k_replay_list_many<true>, dim3(items, d.nch), dim3(kThreads), 0, s, args_dev);
    else
        hipLaunchKernelGGL(k_replay_list_many<false>, dim3(items, d.nch), dim3(kThreads), 0, s, args_dev);
    hipError_t e = by_format(fmt, [&](auto F) {
        auto kern = k_replay_emit_many<decltype(F)::value>;
        const hipError_t e2 = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(EmitShared));
        if (e2 != hipSuccess) return e2;
        // every workgroup pays the table set-up (33 us): the chip's worth of workgroups shared out over the channels
        unsigned g = (d.max_items / 4 + 64 * kEmitWaves - 1) / (64 * kEmitWaves), share = (unsigned)kEmitGroups / d.nch;
        if (share < 1) share = 1;
        if (g > share) g = share;
        hipLaunchKernelGGL(kern, dim3(g < 1 ? 1 : g, d.nch), dim3(64 * kEmitWaves), sizeof(EmitShared), s, args_dev);
        return hipGetLastError();
    });
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_replay_fold_many, dim3(items, d.nch), dim3(kThreads), 0, s, args_dev);
    hipLaunchKernelGGL(k_chain_readback_many, dim3(1, d.nch), dim3(kThreads), 0, s, args_dev, readback_dev, readback_stride, d.max_nblocks, pin_adders);
    return hipGetLastError();
}

hipError_t launch_replay_finish(const BatchArgs &a, const ReplayArgs &r, void *temp, size_t temp_bytes, hipStream_t s) {
    hipError_t e = launch_replay_select(a, r, temp, temp_bytes, s);
    if (e != hipSuccess) return e;
    e = launch_replay_fold(r, s);
    if (e != hipSuccess) return e;
    return launch_replay_emit(a, r, s);
}

// ---- Mode A/C merge: one thread per dword of a 56-byte record
__global__ __launch_bounds__(kThreads) void k_merge_modeac(const MergeArgs m) {
    constexpr uint32_t kDw = sizeof(rsb_message) / 4;
    const uint32_t *src = (const uint32_t *)m.modes, *acs = (const uint32_t *)m.ac_msgs;
    uint32_t *dst = (uint32_t *)m.dst;
    const uint64_t total = (uint64_t)(m.n_modes + m.n_ac) * kDw;
    for (uint64_t idx = (uint64_t)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * kThreads) {
        const uint32_t k = (uint32_t)(idx / kDw), d = (uint32_t)(idx - (uint64_t)k * kDw);
        if (k < m.n_modes) {
            uint32_t lo = 0, hi = m.nblocks;  // the block b with blk_first[b] <= k < blk_first[b + 1]
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (m.blk_first[mid] <= k) lo = mid; else hi = mid;
            }
            dst[(uint64_t)(k + m.ac_before[lo]) * kDw + d] = src[(uint64_t)k * kDw + d];
        } else {
            const uint32_t t = k - m.n_modes;
            dst[(uint64_t)m.ac_dst[t] * kDw + d] = acs[(uint64_t)t * kDw + d];
        }
    }
}

hipError_t launch_merge_modeac(const MergeArgs &m, hipStream_t s) {
    const uint64_t total = (uint64_t)(m.n_modes + m.n_ac) * (sizeof(rsb_message) / 4);
    if (total == 0) return hipSuccess;
    const uint64_t g = (total + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k_merge_modeac, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(kThreads), 0, s, m);
    return hipGetLastError();
}

// ---- wire formats: one thread per message record (wire.hpp holds the byte-level description)
__device__ __forceinline__ rsb_message load_message(const rsb_message *p) {
    // packed 56-byte records: 8-byte aligned, seven 64-bit loads
    union {
        rsb_message m;
        uint64_t q[7];
    } u;
    const uint64_t *src = (const uint64_t *)p;
#pragma unroll
    for (int k = 0; k < 7; ++k) u.q[k] = src[k];
    return u.m;
}

__global__ void k_wire_sizes(const WireArgs w) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > w.n) return;
    uint32_t len = 0;
    if (i < w.n) wire_bytes(load_message(w.msgs + i), w.wire, w.flags, [&](uint8_t) { ++len; });
    w.off[i] = len;  // entry n: 0, so that the exclusive sum leaves the total there
}

__global__ void k_wire_write(const WireArgs w) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w.n) return;
    uint8_t *dst = w.out + w.off[i];
    wire_bytes(load_message(w.msgs + i), w.wire, w.flags, [&](uint8_t b) { *dst++ = b; });
}

hipError_t launch_wire_offsets(const WireArgs &w, void *temp, size_t temp_bytes, hipStream_t s) {
    size_t need = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, need, w.off, w.off, 0u, (size_t)w.n + 1, rocprim::plus<uint32_t>(), s);
    if (e != hipSuccess) return e;
    if (need > temp_bytes) return hipErrorOutOfMemory;
    hipLaunchKernelGGL(k_wire_sizes, dim3((w.n + kThreads) / kThreads), dim3(kThreads), 0, s, w);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return rocprim::exclusive_scan(temp, need, w.off, w.off, 0u, (size_t)w.n + 1, rocprim::plus<uint32_t>(), s);
}

hipError_t launch_wire_write(const WireArgs &w, hipStream_t s) {
    if (w.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_wire_write, dim3((w.n + kThreads - 1) / kThreads), dim3(kThreads), 0, s, w);
    return hipGetLastError();
}

__global__ __launch_bounds__(kThreads) void k_filter_clear(uint32_t *t0, uint32_t *t1, const uint32_t *addrs, uint32_t n) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) t0[addrs[i]] = t1[addrs[i]] = kNever;
}

hipError_t launch_filter_update(uint32_t *t0, uint32_t *t1, const uint32_t *stale_dev, uint32_t n_stale, const uint32_t *addrs_dev, uint32_t n, hipStream_t s) {
    if (n_stale) hipLaunchKernelGGL(k_filter_clear, dim3((n_stale + kThreads - 1) / kThreads), dim3(kThreads), 0, s, t0, t1, stale_dev, n_stale);
    hipLaunchKernelGGL(k_filter_fill, dim3((n + kThreads) / kThreads), dim3(kThreads), 0, s, t0, t1, addrs_dev, n);  // (also sets entry 0)
    return hipGetLastError();
}

__global__ __launch_bounds__(kThreads) void k_filter_clear_many(const FilterUpdate *__restrict__ u) {
    const FilterUpdate &f = u[blockIdx.y];
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < f.n_stale; i += gridDim.x * kThreads) f.t0[f.stale[i]] = f.t1[f.stale[i]] = kNever;
}
__global__ __launch_bounds__(kThreads) void k_filter_fill_many(const FilterUpdate *__restrict__ u) {
    const FilterUpdate &f = u[blockIdx.y];
    if (blockIdx.x == 0 && threadIdx.x == 0) f.t0[0] = f.t1[0] = 0;  // icaoFilterTest(0) hits an empty slot (icao_filter.c:101-107)
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < f.n_members; i += gridDim.x * kThreads) f.t0[f.members[i]] = f.t1[f.members[i]] = 0;
}
hipError_t launch_filter_update_many(const FilterUpdate *u_dev, uint32_t nch, uint32_t max_stale, uint32_t max_members, hipStream_t s) {
    if (nch == 0) return hipSuccess;
    const auto grid = [](uint32_t n) { const unsigned g = (n + kThreads - 1) / kThreads; return g < 1 ? 1u : (g > 64 ? 64u : g); };
    if (max_stale) hipLaunchKernelGGL(k_filter_clear_many, dim3(grid(max_stale), nch), dim3(kThreads), 0, s, u_dev);
    hipLaunchKernelGGL(k_filter_fill_many, dim3(grid(max_members), nch), dim3(kThreads), 0, s, u_dev);
    return hipGetLastError();
}

// ---- the adder bitmap (the prune's superset) worked on where it lies: 2 MiB that used to cross PCIe for every change or export
__global__ __launch_bounds__(kThreads) void k_bitmap_set(uint32_t *bitmap, const uint32_t *addrs, uint32_t n) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) atomicOr(&bitmap[(addrs[i] & 0xffffffu) >> 5], 1u << (addrs[i] & 31u));
}
// out[0] = number of set bits, out[1 ..] = their addresses (in no particular order; the first cap are stored)
__global__ __launch_bounds__(kThreads) void k_bitmap_list(const uint32_t *__restrict__ bitmap, uint32_t *out, uint32_t cap) {
    for (uint32_t w = blockIdx.x * kThreads + threadIdx.x; w < (1u << 24) / 32; w += gridDim.x * kThreads)
        for (uint32_t x = bitmap[w]; x; x &= x - 1) {
            const uint32_t slot = atomicAdd(out, 1u);
            if (slot < cap) out[1 + slot] = (w << 5) | (uint32_t)__builtin_ctz(x);
        }
}
// how many of a batch's kept items ask the filter about (or would teach it) one of the given addresses
__global__ __launch_bounds__(kThreads) void k_items_reference(const uint32_t *__restrict__ info, uint32_t n, const uint32_t *__restrict__ addrs, uint32_t na,
                                                               uint32_t *count) {
    uint32_t hits = 0;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const uint32_t v = item_value(info[i]);
        for (uint32_t k = 0; k < na; ++k) hits += addrs[k] == v ? 1u : 0u;
    }
    if (hits) atomicAdd(count, hits);
}
hipError_t launch_bitmap_set(uint32_t *bitmap, const uint32_t *addrs_dev, uint32_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_bitmap_set, dim3((n + kThreads - 1) / kThreads), dim3(kThreads), 0, s, bitmap, addrs_dev, n);
    return hipGetLastError();
}
hipError_t launch_bitmap_list(const uint32_t *bitmap, uint32_t *out, uint32_t cap, hipStream_t s) {
    const hipError_t e = hipMemsetAsync(out, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_bitmap_list, dim3(512), dim3(kThreads), 0, s, bitmap, out, cap);
    return hipGetLastError();
}
hipError_t launch_items_reference(const uint32_t *info, uint32_t n, const uint32_t *addrs_dev, uint32_t na, uint32_t *count, hipStream_t s) {
    const hipError_t e = hipMemsetAsync(count, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    if (n && na) hipLaunchKernelGGL(k_items_reference, dim3(item_grid(n)), dim3(kThreads), 0, s, info, n, addrs_dev, na, count);
    return hipGetLastError();
}

hipError_t launch_filter_rebuild(uint32_t *t0, uint32_t *t1, const uint32_t *addrs_dev, uint32_t n, hipStream_t s) {
    hipError_t e = hipMemsetAsync(t0, 0xff, sizeof(uint32_t) << 24, s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(t1, 0xff, sizeof(uint32_t) << 24, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_filter_fill, dim3((n + kThreads) / kThreads), dim3(kThreads), 0, s, t0, t1, addrs_dev, n);
    return hipGetLastError();
}

hipError_t launch_block_means_float(int fmt, const void *iq, uint64_t n, uint64_t blk_len, double *means, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const uint32_t nblocks = (uint32_t)((n + blk_len - 1) / blk_len);
    return by_format(fmt, [&](auto F) {
        if (nblocks >= 16) {  // many blocks: one lane per block does the sequential sums, the other wavefronts feed it
            auto kern = k_block_means_group<decltype(F)::value>;
            const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMeansLds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3((nblocks + kMeansGroup - 1) / kMeansGroup), dim3(kMeansThreads), kMeansLds, s, iq, n, blk_len, means, nblocks);
        } else
            hipLaunchKernelGGL(k_block_means_float<decltype(F)::value>, dim3(nblocks), dim3(64), 0, s, iq, n, blk_len, means);
        return hipGetLastError();
    });
}

// one thread per block: the block's sub-block means added up in order, divided by their number (sdr_bladerf.c:389-390, 403-404)
__global__ void k_means_average(const double *sub, uint32_t nsub, uint32_t per_block, double *means, uint32_t nblocks) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const uint32_t lo = b * per_block, hi = min(nsub, lo + per_block);
    double level = 0, power = 0;
    for (uint32_t k = lo; k < hi; ++k) {
        level += sub[2 * k];
        power += sub[2 * k + 1];
    }
    means[2 * b] = level / (double)(hi - lo);  // (the reference divides by an unsigned: the same conversion)
    means[2 * b + 1] = power / (double)(hi - lo);
}

hipError_t launch_means_average(const double *sub_means, uint32_t nsub, uint32_t per_block, double *means, uint32_t nblocks, hipStream_t s) {
    if (!nblocks) return hipSuccess;
    hipLaunchKernelGGL(k_means_average, dim3((nblocks + 255) / 256), dim3(256), 0, s, sub_means, nsub, per_block, means, nblocks);
    return hipGetLastError();
}

template <int FMT>
static hipError_t launch_ac_kernel(const BatchArgs &a, int sums_only, hipStream_t s) {
    auto kern = k_modeac<FMT>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(AcShared));
    if (e != hipSuccess) return e;
    // one workgroup per CU (its LDS allows no second one), each walking the blocks with that stride
    int dev = 0, cus = 0;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
    static const int forced = std::getenv("RSB_AC_GRID") ? std::atoi(std::getenv("RSB_AC_GRID")) : 0;  // A/B timing only
    const uint64_t nblk = (a.n + a.block_len - 1) / a.block_len;
    uint64_t grid = forced > 0 ? (uint64_t)forced : (uint64_t)(cus > 0 ? cus : 256);
    if (grid > nblk) grid = nblk;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kAcThreads), sizeof(AcShared), s, a, sums_only);
    return hipGetLastError();
}

hipError_t launch_block_sums(const BatchArgs &a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    return by_format(a.fmt, [&](auto F) {
        return launch_ac_kernel<decltype(F)::value>(a, 1, s);
    });
}

hipError_t launch_modeac(const BatchArgs &a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    return by_format(a.fmt, [&](auto F) {
        return launch_ac_kernel<decltype(F)::value>(a, 0, s);
    });
}

void dc_coefficients(float *dc_a, float *dc_b) {
    // convert.c:480-488: dc_b = (float)exp(-2 pi / sample_rate), dc_a = (float)(1.0 - dc_b)
    *dc_b = (float)std::exp(-2.0 * 3.14159265358979323846 * 1.0 / 2400000.0);
    *dc_a = (float)(1.0 - (double)*dc_b);
}

void dc_pow_table(double *out512) {
    float dc_a, dc_b;
    dc_coefficients(&dc_a, &dc_b);
    double m8 = (double)dc_b;
    for (int k = 0; k < 3; ++k) m8 *= m8;  // dc_b^8, as k_dc_tile_sums forms it
    double p = 1.0;
    for (int j = 0; j < 512; ++j, p *= m8) out512[j] = p;
}

namespace {
template <int FMT>
__global__ __launch_bounds__(kThreads) void k_dc_carry(const BatchArgs a, uint16_t *carry_out) {  // the batch's last 326 magnitudes
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i < (uint32_t)kTrailing) carry_out[i] = (uint16_t)stream_mag<FMT>(a, (int64_t)a.n + (int64_t)i);
}
template <int FMT>
__global__ __launch_bounds__(kThreads) void k_dc_materialize(const BatchArgs a, uint16_t *mag) {  // one thread per 8-sample chunk
    const uint64_t c = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (8 * c >= a.n) return;
    float2 z = make_float2(0.0f, 0.0f);
    int64_t zpos = -1;
    for (uint64_t i = 8 * c; i < 8 * c + 8 && i < a.n; ++i) mag[i] = (uint16_t)stream_mag_run<FMT>(a, (int64_t)i + (int64_t)kTrailing, z, zpos);
}
}  // namespace

hipError_t launch_dc_carry(const BatchArgs &a, uint16_t *carry_out, hipStream_t s) {
    return by_format(a.fmt, [&](auto F) {
        constexpr int f = decltype(F)::value;
        if constexpr (f != kFmtMag16) {
            hipLaunchKernelGGL(k_dc_carry<f>, dim3((kTrailing + kThreads - 1) / kThreads), dim3(kThreads), 0, s, a, carry_out);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    });
}

hipError_t launch_dc_materialize(const BatchArgs &a, uint16_t *mag, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    return by_format(a.fmt, [&](auto F) {
        constexpr int f = decltype(F)::value;
        if constexpr (f != kFmtMag16) {
            hipLaunchKernelGGL(k_dc_materialize<f>, dim3((unsigned)((a.n / 8 + kThreads) / kThreads)), dim3(kThreads), 0, s, a, mag);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    });
}

hipError_t launch_dc_states(int fmt, const void *iq, uint64_t n, double2 *tile_sums, double2 *tile_start, float2 *local, double2 *z_state, hipStream_t s) {
    if (n == 0) return hipSuccess;
    float dc_a, dc_b;
    dc_coefficients(&dc_a, &dc_b);
    const uint32_t ntiles = (uint32_t)((n + kDcTile - 1) / kDcTile);
    return by_format(fmt, [&](auto F) {
        constexpr int f = decltype(F)::value;
        if constexpr (f != kFmtMag16) {
            hipLaunchKernelGGL(k_dc_tile_sums<f>, dim3(ntiles), dim3(kThreads), 0, s, iq, n, (double)dc_a, (double)dc_b, tile_sums, local);
            const uint32_t nchunks = (ntiles + kDcChunkTiles - 1) / kDcChunkTiles;
            hipLaunchKernelGGL(k_dc_scan_local, dim3(nchunks), dim3(kDcChunkTiles), 0, s, tile_sums, tile_start, ntiles, (double)dc_b);
            hipLaunchKernelGGL(k_dc_scan_finish, dim3(nchunks), dim3(kDcChunkTiles), 0, s, tile_sums, tile_start, ntiles, z_state, (double)dc_b);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
            // the state after the batch's last sample, for the next batch (a batch that is not a whole number of tiles ends its stream)
            if (n % kDcTile == 0) e = hipMemcpyAsync(z_state, tile_start + ntiles, sizeof(double2), hipMemcpyDeviceToDevice, s);
            return e;
        } else {
            return hipErrorInvalidValue;
        }
    });
}

hipError_t launch_convert_dc(int fmt, const void *iq, uint16_t *mag, uint64_t n, double2 *tile_sums, double2 *tile_start, double2 *z_state, double *block_means,
                             hipStream_t s) {
    if (n == 0) return hipSuccess;
    float dc_a, dc_b;
    dc_coefficients(&dc_a, &dc_b);
    const uint32_t ntiles = (uint32_t)((n + kDcTile - 1) / kDcTile), nblocks = (uint32_t)((n + kBlock - 1) / kBlock);
    return by_format(fmt, [&](auto F) {
        constexpr int f = decltype(F)::value;
        if constexpr (f == kFmtMag16) {
            return hipErrorInvalidValue;
        } else {
            hipLaunchKernelGGL(k_dc_tile_sums<f>, dim3(ntiles), dim3(kThreads), 0, s, iq, n, (double)dc_a, (double)dc_b, tile_sums, (float2 *)nullptr);
            const uint32_t nchunks = (ntiles + kDcChunkTiles - 1) / kDcChunkTiles;
            hipLaunchKernelGGL(k_dc_scan_local, dim3(nchunks), dim3(kDcChunkTiles), 0, s, tile_sums, tile_start, ntiles, (double)dc_b);
            hipLaunchKernelGGL(k_dc_scan_finish, dim3(nchunks), dim3(kDcChunkTiles), 0, s, tile_sums, tile_start, ntiles, z_state, (double)dc_b);
            static const int abl = std::getenv("RSB_DC_ABL") ? std::atoi(std::getenv("RSB_DC_ABL")) : 0;  // ablation timing only
            // (tile_sums is free again: the scan over the tiles has consumed it -- the tiles' partial sums of mag and mag^2 go there)
            hipLaunchKernelGGL(k_convert_dc<f>, dim3(ntiles), dim3(kThreads), 0, s, iq, mag, n, dc_a, dc_b, tile_start, block_means, tile_sums, z_state, abl);
            if (block_means) hipLaunchKernelGGL(k_dc_means, dim3((nblocks + 255) / 256), dim3(256), 0, s, block_means, (const double2 *)tile_sums, ntiles, n, nblocks);
            return hipGetLastError();
        }
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

hipError_t launch_convert_q11t(const void *iq, uint16_t *mag, uint64_t n, int use_bits, uint64_t *sums, hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + kThreads - 1) / kThreads;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_convert_q11t, dim3((unsigned)blocks), dim3(kThreads), 0, s, iq, mag, n, 11 - use_bits, sums);
    return hipGetLastError();
}

}  // namespace rsb
