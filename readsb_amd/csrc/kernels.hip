// kernels.hip -- gfx950 kernels of the receive path.
//
//   k_scan      convert.c (IQ -> magnitude, in registers/LDS only) + demod_2400.c:109-172 (preamble
//               gate, one thread per scan offset) + :179-286 (five trial phases, bit slicing) +
//               crc.c:67-82 / :383-406 (syndrome, error-table lookup) + the filter-independent half
//               of mode_s.c:311-409.  Emits one 24-byte key record per gated offset that can still
//               score >= 0 under some filter state, and marks every address the filter could ever
//               learn (mode_s.c:717-726) in a 2^24-bit set.
//   k_prune     drops key records whose every live phase needs a filter member that is not in that
//               superset (safe: such an offset is rejected by demod_2400.c:304-310 whatever happened
//               before it, and a rejected offset neither skips nor adds).
//   k_finalize  re-slices the few survivors into full records (5 x 14 message bytes + the signal
//               power sums of demod_2400.c:345-351) for the host replay.
//   k_block_sums / k_modeac   convert.c:71-83 block statistics and demod_2400.c:476-622.
//   k_convert   the stand-alone iq_convert_fn (magnitudes written to HBM) for the drop-in converter.
//
// Compiled with -ffp-contract=off: every float/double product and sum below rounds on its own, as
// in the reference's x86-64 build.
#include "kernels.hpp"

#include <type_traits>

namespace rsb {
namespace {

constexpr int kTile = 4096;   // scan offsets per workgroup
constexpr int kHalo = 304;    // look-ahead past the last offset of a tile: 290 needed
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
// demod_2400.c:109-172

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

__device__ __forceinline__ int slice_corr(const uint16_t *m, int phase) {
    const int a = m[0], b = m[1], c = m[2];
    switch (phase) {
        case 0: return 5 * a - 3 * b - 2 * c;
        case 1: return 4 * a - b - 3 * c;
        case 2: return 3 * a + b - 4 * c;
        case 3: return 2 * a + 3 * b - 5 * c;
        default: return a + 5 * b - 5 * c - (int)m[3];
    }
}

// Slices one trial phase starting at d = &S[g + 19]. Bits are packed MSB-first into w[0..3]
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

__device__ __forceinline__ uint32_t msg_byte(const uint32_t w[4], int i) { return (w[i >> 2] >> (24 - 8 * (i & 3))) & 0xffu; }

// modesChecksum, crc.c:67-82, on the packed words; crcb = crc_table[256] (in LDS)
__device__ uint32_t syndrome_of(const uint32_t w[4], int nbytes, const uint32_t *crcb) {
    uint32_t rem = 0;
    for (int i = 0; i < nbytes - 3; ++i) rem = ((rem << 8) ^ crcb[msg_byte(w, i) ^ (rem >> 16)]) & 0xffffffu;
    const uint32_t tail = (nbytes == 7) ? ((w[1] >> 8) & 0xffffffu) : (((w[2] & 0xffu) << 16) | (w[3] >> 16));
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

// The filter-independent half of scoreModesMessage (mode_s.c:311-409): what can this phase's score be?
__device__ uint32_t classify_phase(const uint32_t w[4], int nbytes, const DeviceTables &t, const uint32_t *crcb, uint32_t *adders) {
    if (nbytes < 7) return kDead << 24;  // validbits < 56
    const uint32_t df = w[0] >> 27;
    if (nbytes == 7 && w[0] == 0 && (w[1] >> 8) == 0) return kDead << 24;  // all-zero message, mode_s.c:325
    const uint32_t syn = syndrome_of(w, nbytes, crcb);
    switch (df) {
        case 0: case 4: case 5: case 16: case 24:  // Address/Parity (25..31 never get this far)
        case 20: case 21:                          // Comm-B: exact match or nothing (mode_s.c:391-403)
            return (kNeedCrc << 24) | syn;
        case 11: {
            const uint32_t s = syn & 0xffff80u, iid = syn & 0x7fu;
            uint32_t info = 0;
            if (s) {
                info = diagnose(t, 0, s);
                if (info == 0xffffffffu || (info & 0xff) > 1) return kDead << 24;  // mode_s.c:350-358
            }
            const uint32_t addr = repair_aa(w[0] & 0xffffffu, info);
            if (iid == 0) {
                if (s == 0) atomicOr(&adders[addr >> 5], 1u << (addr & 31));  // mode_s.c:717-726
                return (kAlive << 24) | addr;
            }
            return (kNeedAddr << 24) | addr;
        }
        case 17: case 18: {
            uint32_t info = 0;
            if (syn) {
                info = diagnose(t, 1, syn);
                if (info == 0xffffffffu) return kDead << 24;
            }
            const uint32_t addr = repair_aa(w[0] & 0xffffffu, info);
            if (df == 17 && syn == 0) atomicOr(&adders[addr >> 5], 1u << (addr & 31));
            return (kAlive << 24) | addr;
        }
        default:
            return kDead << 24;
    }
}

// ------------------------------------------------------------------------------------------------

template <int FMT>
__global__ __launch_bounds__(kThreads) void k_scan(const BatchArgs a) {
    __shared__ uint16_t m[kTile + kHalo];
    __shared__ uint16_t queue[kTile];
    __shared__ uint32_t crcb[256];
    __shared__ uint32_t qn;

    const uint32_t tid = threadIdx.x;
    const uint64_t t0 = (uint64_t)blockIdx.x * kTile;
    const uint32_t nvalid = (uint32_t)((a.n - t0 < (uint64_t)kTile) ? (a.n - t0) : (uint64_t)kTile);

    for (uint32_t i = tid; i < kTile + kHalo; i += kThreads) m[i] = (uint16_t)stream_mag<FMT>(a, (int64_t)(t0 + i));
    crcb[tid] = a.tab.crc_bytes[tid];
    if (tid == 0) qn = 0;
    __syncthreads();

    for (uint32_t j = tid; j < nvalid; j += kThreads)
        if (preamble_gate(&m[j])) queue[atomicAdd(&qn, 1u)] = (uint16_t)j;
    __syncthreads();

    const uint32_t ncand = qn;
    if (tid == 0 && ncand) atomicAdd(&a.counters[kCntGated], ncand);

    for (uint32_t c = tid; c < ncand; c += kThreads) {
        const uint32_t j = queue[c];
        KeyRec rec;
        rec.g = (uint32_t)(t0 + j);
        bool live = false;
        for (int p = 0; p < 5; ++p) {
            uint32_t w[4];
            const int nbytes = slice_phase(&m[j + 19], p + 4, w);
            rec.key[p] = classify_phase(w, nbytes, a.tab, crcb, a.adder_bitmap);
            live |= (rec.key[p] >> 24) != kDead;
        }
        if (live || a.emit_dead) {
            const uint32_t slot = atomicAdd(&a.counters[kCntKeys], 1u);
            if (slot < a.key_cap) a.keys[slot] = rec;
        }
    }
}

__global__ __launch_bounds__(kThreads) void k_prune(const BatchArgs a) {
    uint32_t nkeys = a.counters[kCntKeys];
    if (nkeys > a.key_cap) nkeys = a.key_cap;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < nkeys; i += gridDim.x * kThreads) {
        const KeyRec rec = a.keys[i];
        bool keep = a.keep_all != 0;
        for (int p = 0; p < 5 && !keep; ++p) {
            const uint32_t cls = rec.key[p] >> 24, v = rec.key[p] & 0xffffffu;
            if (cls == kAlive)
                keep = true;
            else if (cls == kNeedCrc || cls == kNeedAddr)
                keep = (a.adder_bitmap[v >> 5] >> (v & 31)) & 1u;
        }
        if (keep) {
            const uint32_t slot = atomicAdd(&a.counters[kCntSurvivors], 1u);
            if (slot < a.surv_cap) a.surv_g[slot] = rec.g;
        }
    }
}

// one wavefront per survivor
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

hipError_t launch_scan(const BatchArgs &a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    return by_format(a.fmt, [&](auto F) {
        hipLaunchKernelGGL(k_scan<decltype(F)::value>, dim3(tiles_of(a.n)), dim3(kThreads), 0, s, a);
        return hipGetLastError();
    });
}

hipError_t launch_prune_finalize(const BatchArgs &a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_prune, dim3(1024), dim3(kThreads), 0, s, a);
    hipError_t e = hipGetLastError();
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
