/*
 * modes_oracle.c -- TEST INFRASTRUCTURE ONLY (see modes_oracle.h).
 *
 * Scalar CPU restatement of: convert.c (all six default-build converters), crc.c,
 * icao_filter.c, mode_s.c:scoreModesMessage + the CRC-dependent part of
 * decodeModesMessage, demod_2400.c:demodulate2400 / demodulate2400AC,
 * mode_ac.c:decodeModeAMessage (message fields only) and the sdr_ifile.c block loop.
 *
 * Compile with -ffp-contract=off: the reference is built by x86-64 gcc -O2 without FMA
 * (SURVEY.md Appendix B.5), so every float product and sum rounds separately.
 */
#define _GNU_SOURCE
#include "modes_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ CRC (crc.c) */

#define GENERATOR_POLY 0xfff409u /* crc.c:31 */

static uint32_t crc_table[256];
static uint32_t single_bit_syndrome[112];
static int crc_tables_ready;

/* crc.c:67-82 */
uint32_t orc_checksum(const uint8_t *message, int bits) {
    uint32_t rem = 0;
    int n = bits / 8;
    for (int i = 0; i < n - 3; ++i) {
        rem = (rem << 8) ^ crc_table[message[i] ^ ((rem & 0xff0000) >> 16)];
        rem &= 0xffffff;
    }
    return rem ^ ((uint32_t) message[n - 3] << 16) ^ ((uint32_t) message[n - 2] << 8) ^ message[n - 1];
}

/* crc.c:42-65 */
static void init_crc_tables(void) {
    if (crc_tables_ready)
        return;
    for (int i = 0; i < 256; ++i) {
        uint32_t c = (uint32_t) i << 16;
        for (int j = 0; j < 8; ++j)
            c = (c & 0x800000) ? ((c << 1) ^ GENERATOR_POLY) : (c << 1);
        crc_table[i] = c & 0xffffff;
    }
    uint8_t msg[14];
    memset(msg, 0, sizeof (msg));
    for (int i = 0; i < 112; ++i) {
        msg[i / 8] ^= 1 << (7 - (i & 7));
        single_bit_syndrome[i] = orc_checksum(msg, 112);
        msg[i / 8] ^= 1 << (7 - (i & 7));
    }
    crc_tables_ready = 1;
}

static int ei_compare(const void *x, const void *y) {
    return (int) ((const orc_errorinfo *) x)->syndrome - (int) ((const orc_errorinfo *) y)->syndrome; /* crc.c:91-95 */
}

/* crc.c:98-113 */
static int combinations(int n, int k) {
    int result = 1;
    if (k == 0 || k == n) return 1;
    if (k > n) return 0;
    for (int i = 1; i <= k; ++i) {
        result = result * n / i;
        n = n - 1;
    }
    return result;
}

/* crc.c:130-149 */
static int prepare_subtable(orc_errorinfo *table, int n, int offset, int startbit, int endbit,
        const orc_errorinfo *base, int error_bit, int max_errors) {
    if (error_bit >= max_errors)
        return n;
    for (int i = startbit; i < endbit; ++i) {
        table[n] = *base;
        table[n].syndrome ^= single_bit_syndrome[i + offset];
        table[n].errors = error_bit + 1;
        table[n].bit[error_bit] = (int8_t) i;
        ++n;
        orc_errorinfo cur = table[n - 1];
        n = prepare_subtable(table, n, offset, i + 1, endbit, &cur, error_bit + 1, max_errors);
    }
    return n;
}

/* crc.c:151-175 */
static int flag_collisions(orc_errorinfo *table, int tablesize, int offset, int startbit, int endbit,
        uint32_t base_syndrome, int error_bit, int first_error, int last_error) {
    int count = 0;
    if (error_bit > last_error)
        return 0;
    for (int i = startbit; i < endbit; ++i) {
        orc_errorinfo ei;
        ei.syndrome = base_syndrome ^ single_bit_syndrome[i + offset];
        if (error_bit >= first_error) {
            orc_errorinfo *hit = bsearch(&ei, table, tablesize, sizeof (orc_errorinfo), ei_compare);
            if (hit && hit->errors != -1) {
                ++count;
                hit->errors = -1;
            }
        }
        count += flag_collisions(table, tablesize, offset, i + 1, endbit, ei.syndrome, error_bit + 1, first_error, last_error);
    }
    return count;
}

/* crc.c:180-350 */
static orc_errorinfo *prepare_error_table(int bits, int max_correct, int max_detect, int *size_out) {
    if (!max_correct) {
        *size_out = 0;
        return NULL;
    }
    int maxsize = 0;
    for (int i = 1; i <= max_correct; ++i)
        maxsize += combinations(bits - 5, i);
    orc_errorinfo *table = malloc((size_t) (maxsize + 1) * sizeof (orc_errorinfo));
    orc_errorinfo base;
    memset(&base, 0, sizeof (base));
    base.bit[0] = base.bit[1] = -1;
    /* the first five bits (the DF field) are never corrected, crc.c:210-211 */
    int used = prepare_subtable(table, 0, 112 - bits, 5, bits, &base, 0, max_correct);
    qsort(table, used, sizeof (orc_errorinfo), ei_compare);

    /* drop every syndrome produced by more than one pattern, crc.c:243-263 */
    int j = 0;
    for (int i = 0; i < used; ++i) {
        if (i < used - 1 && table[i + 1].syndrome == table[i].syndrome) {
            while (i < used - 1 && table[i + 1].syndrome == table[i].syndrome)
                ++i;
            continue;
        }
        if (i != j)
            table[j] = table[i];
        ++j;
    }
    used = j;

    /* drop syndromes that a (max_correct+1 .. max_detect)-bit pattern also produces, crc.c:266-294 */
    if (max_detect > max_correct) {
        int flagged = flag_collisions(table, used, 112 - bits, 5, bits, 0, 1, max_correct + 1, max_detect);
        if (flagged > 0) {
            j = 0;
            for (int i = 0; i < used; ++i)
                if (table[i].errors != -1) {
                    if (i != j)
                        table[j] = table[i];
                    ++j;
                }
            used = j;
        }
    }
    *size_out = used;
    return table;
}

/* ------------------------------------------------------------------ ICAO filter (icao_filter.c) */

#define FILTER_SIZE 4096 /* icao_filter.c:27 */

typedef struct {
    uint32_t a[FILTER_SIZE], b[FILTER_SIZE];
    uint32_t *active;
} icao_filter;

/* icao_filter.c:42-63 */
static uint32_t icao_hash(uint32_t a) {
    uint32_t hash = 0;
    hash += a & 0xff;
    hash += hash << 10;
    hash ^= hash >> 6;
    hash += (a >> 8) & 0xff;
    hash += hash << 10;
    hash ^= hash >> 6;
    hash += (a >> 16) & 0xff;
    hash += hash << 10;
    hash ^= hash >> 6;
    hash += hash << 3;
    hash ^= hash >> 11;
    hash += hash << 15;
    return hash & (FILTER_SIZE - 1);
}

static void filter_init(icao_filter *f) { /* icao_filter.c:65-69 */
    memset(f->a, 0, sizeof (f->a));
    memset(f->b, 0, sizeof (f->b));
    f->active = f->a;
}

static void filter_add(icao_filter *f, uint32_t addr) { /* icao_filter.c:71-95 */
    uint32_t h, h0;
    h0 = h = icao_hash(addr);
    while (f->active[h] && f->active[h] != addr) {
        h = (h + 1) & (FILTER_SIZE - 1);
        if (h == h0)
            return;
    }
    if (!f->active[h])
        f->active[h] = addr;
    h0 = h = icao_hash(addr & 0x00ffff);
    while (f->active[h] && (f->active[h] & 0x00ffff) != (addr & 0x00ffff)) {
        h = (h + 1) & (FILTER_SIZE - 1);
        if (h == h0)
            return;
    }
    if (!f->active[h])
        f->active[h] = addr;
}

static int filter_test(const icao_filter *f, uint32_t addr) { /* icao_filter.c:97-119 */
    uint32_t h, h0;
    h0 = h = icao_hash(addr);
    while (f->a[h] && f->a[h] != addr) {
        h = (h + 1) & (FILTER_SIZE - 1);
        if (h == h0)
            break;
    }
    if (f->a[h] == addr)
        return 1;
    h = h0;
    while (f->b[h] && f->b[h] != addr) {
        h = (h + 1) & (FILTER_SIZE - 1);
        if (h == h0)
            break;
    }
    return f->b[h] == addr;
}

/* ------------------------------------------------------------------ context */

struct orc_ctx {
    int format, dc_filter, nfix, modeac;
    orc_errorinfo *tab_short, *tab_long;
    int n_short, n_long;
    icao_filter filter;
    /* struct converter_state, convert.c:26-31 */
    float dc_a, dc_b, z1_I, z1_Q;
    int q11_table_bits;          /* 0: the default build; n: the reference built with -DSC16Q11_TABLE_BITS=n (convert.c:264-328) */
    uint16_t *q11_lookup;
    orc_stats stats;
    /* message sink */
    orc_msg *out;
    size_t cap, nout;
    orc_candidate *cand_out;
    size_t cand_cap, ncand;
    orc_ac_candidate *ac_out;
    size_t ac_cap, nac;
    /* the two alternating slice buffers live across calls in the reference only as stack
     * garbage; we keep them zero-initialised per demod call (compared bytes are msgbits/8 only) */
};

static uint16_t *uc8_lookup;

/* convert.c:35-61 */
static void init_uc8_lookup(void) {
    if (uc8_lookup)
        return;
    uc8_lookup = malloc(sizeof (uint16_t) * 65536);
    for (int i = 0; i <= 255; i++)
        for (int q = 0; q <= 255; q++) {
            float fI = (i - 127.5) / 127.5;
            float fQ = (q - 127.5) / 127.5;
            float magsq = fI * fI + fQ * fQ;
            if (magsq > 1)
                magsq = 1;
            float mag = sqrtf(magsq);
            uc8_lookup[i * 256 + q] = (uint16_t) (mag * 65535.0f + 0.5f); /* le16toh is the identity on the x86-64 oracle host */
        }
}

const uint16_t *orc_uc8_lut(void) {
    init_uc8_lookup();
    return uc8_lookup;
}

orc_ctx *orc_create(int format, int dc_filter, int nfix, int modeac) {
    if (format < ORC_UC8 || format > ORC_SC16Q11)
        return NULL; /* convert.c:460-464: "no suitable converter" */
    orc_ctx *c = calloc(1, sizeof (*c));
    c->format = format;
    c->dc_filter = dc_filter;
    c->nfix = nfix;
    c->modeac = modeac;
    init_crc_tables();
    init_uc8_lookup();
    /* crc.c:353-378 */
    switch (nfix) {
        case 0:
            break;
        case 1:
            c->tab_short = prepare_error_table(56, 1, 1, &c->n_short);
            c->tab_long = prepare_error_table(112, 1, 1, &c->n_long);
            break;
        default:
            c->tab_short = prepare_error_table(56, 2, 4, &c->n_short);
            c->tab_long = prepare_error_table(112, 2, 4, &c->n_long);
            break;
    }
    orc_reset(c);
    return c;
}

void orc_reset(orc_ctx *c) {
    filter_init(&c->filter);
    memset(&c->stats, 0, sizeof (c->stats));
    c->z1_I = c->z1_Q = 0;
    if (c->dc_filter) { /* convert.c:480-488 */
        c->dc_b = exp(-2.0 * M_PI * 1.0 / 2400000.0);
        c->dc_a = 1.0 - c->dc_b;
    } else {
        c->dc_b = 1.0;
        c->dc_a = 0.0;
    }
}

void orc_destroy(orc_ctx *c) {
    if (!c)
        return;
    free(c->tab_short);
    free(c->tab_long);
    free(c->q11_lookup);
    free(c);
}

void orc_get_stats(const orc_ctx *c, orc_stats *st) { *st = c->stats; }
int orc_table_size(const orc_ctx *c, int bits) { return bits == 56 ? c->n_short : c->n_long; }
const orc_errorinfo *orc_table(const orc_ctx *c, int bits) { return bits == 56 ? c->tab_short : c->tab_long; }
void orc_filter_add(orc_ctx *c, uint32_t addr) { filter_add(&c->filter, addr); }
int orc_filter_test(const orc_ctx *c, uint32_t addr) { return filter_test(&c->filter, addr); }

static const orc_errorinfo NO_ERRORS; /* crc.c:28: all zero, bit[] = 0 not -1 */

/* crc.c:383-406 */
const orc_errorinfo *orc_diagnose(const orc_ctx *c, uint32_t syndrome, int bits) {
    if (syndrome == 0)
        return &NO_ERRORS;
    const orc_errorinfo *table = (bits == 56) ? c->tab_short : c->tab_long;
    int n = (bits == 56) ? c->n_short : c->n_long;
    if (!table)
        return NULL;
    orc_errorinfo key;
    key.syndrome = syndrome;
    return bsearch(&key, table, n, sizeof (orc_errorinfo), ei_compare);
}

/* crc.c:410-418 */
static void checksum_fix(uint8_t *msg, const orc_errorinfo *info) {
    if (!info)
        return;
    for (int i = 0; i < info->errors; ++i)
        msg[info->bit[i] >> 3] ^= 1 << (7 - (info->bit[i] & 7));
}

/* ------------------------------------------------------------------ converters (convert.c) */

/* convert.c:264-295 with SC16Q11_TABLE_BITS = bits: the build of the reference that some packages ship (armhf: 8) */
void orc_set_sc16q11_table_bits(orc_ctx *c, int bits) {
    free(c->q11_lookup);
    c->q11_lookup = NULL;
    c->q11_table_bits = (bits >= 1 && bits <= 11) ? bits : 0;
    if (!c->q11_table_bits)
        return;
    const int use_bits = c->q11_table_bits, lose_bits = 11 - use_bits;
    c->q11_lookup = malloc(sizeof (uint16_t) * ((size_t) 1 << (use_bits * 2)));
    for (int i = 0; i < 2048; i += (1 << lose_bits))
        for (int q = 0; q < 2048; q += (1 << lose_bits)) {
            float fI = i / 2048.0, fQ = q / 2048.0;
            float magsq = fI * fI + fQ * fQ;
            if (magsq > 1)
                magsq = 1;
            float mag = sqrtf(magsq);
            unsigned index = ((i >> lose_bits) << use_bits) | (q >> lose_bits);
            c->q11_lookup[index] = (uint16_t) (mag * 65535.0f + 0.5f);
        }
}

/* one routine per table row of convert.c:431-444 */
void orc_convert(orc_ctx *c, const void *iq, uint16_t *mag_data, unsigned nsamples, double *out_mean_level, double *out_mean_power) {
    if (c->format == ORC_UC8 && !c->dc_filter) {
        /* convert_uc8_nodc, convert.c:63-111: the 16-bit little-endian word (I | Q<<8) indexes the LUT */
        const uint16_t *in = iq;
        uint64_t sum_level = 0, sum_power = 0;
        for (unsigned i = 0; i < nsamples; ++i) {
            uint16_t mag = uc8_lookup[in[i]];
            mag_data[i] = mag;
            sum_level += mag;
            sum_power += (uint32_t) mag * (uint32_t) mag;
        }
        if (out_mean_level)
            *out_mean_level = sum_level / 65536.0 / nsamples;
        if (out_mean_power)
            *out_mean_power = sum_power / 65535.0 / 65535.0 / nsamples;
        return;
    }

    if (c->format == ORC_SC16Q11 && !c->dc_filter && c->q11_table_bits) {
        /* convert_sc16q11_table, convert.c:297-328 */
        const int use_bits = c->q11_table_bits, lose_bits = 11 - use_bits;
        const int16_t *in = iq;
        uint64_t sum_level = 0, sum_power = 0;
        for (unsigned i = 0; i < nsamples; ++i) {
            uint16_t I = abs((int16_t) in[2 * i]) & 2047;
            uint16_t Q = abs((int16_t) in[2 * i + 1]) & 2047;
            uint16_t mag = c->q11_lookup[((I >> lose_bits) << use_bits) | (Q >> lose_bits)];
            mag_data[i] = mag;
            sum_level += mag;
            sum_power += (uint32_t) mag * (uint32_t) mag;
        }
        if (out_mean_level)
            *out_mean_level = sum_level / 65536.0 / nsamples;
        if (out_mean_power)
            *out_mean_power = sum_power / 65535.0 / 65535.0 / nsamples;
        return;
    }

    /* float paths: convert.c:113-162 (uc8 generic), 164-253 (sc16), 332-423 (sc16q11) */
    float z1_I = c->z1_I, z1_Q = c->z1_Q;
    const float dc_a = c->dc_a, dc_b = c->dc_b;
    float sum_level = 0, sum_power = 0;
    const uint8_t *in8 = iq;
    const int16_t *in16 = iq;
    for (unsigned i = 0; i < nsamples; ++i) {
        float fI, fQ;
        if (c->format == ORC_UC8) {
            fI = (in8[2 * i] - 127.5f) / 127.5f;
            fQ = (in8[2 * i + 1] - 127.5f) / 127.5f;
        } else if (c->format == ORC_SC16) {
            fI = in16[2 * i] / 32768.0f;
            fQ = in16[2 * i + 1] / 32768.0f;
        } else {
            fI = in16[2 * i] / 2048.0f;
            fQ = in16[2 * i + 1] / 2048.0f;
        }
        if (c->dc_filter) {
            z1_I = fI * dc_a + z1_I * dc_b;
            z1_Q = fQ * dc_a + z1_Q * dc_b;
            fI -= z1_I;
            fQ -= z1_Q;
        }
        float magsq = fI * fI + fQ * fQ;
        if (magsq > 1)
            magsq = 1;
        float mag = sqrtf(magsq);
        sum_power += magsq;
        sum_level += mag;
        mag_data[i] = (uint16_t) (mag * 65535.0f + 0.5f);
    }
    if (c->dc_filter) {
        c->z1_I = z1_I;
        c->z1_Q = z1_Q;
    }
    /* float / unsigned -> float division, widened afterwards (convert.c:155-161) */
    if (out_mean_level)
        *out_mean_level = sum_level / nsamples;
    if (out_mean_power)
        *out_mean_power = sum_power / nsamples;
}

/* ------------------------------------------------------------------ scoring / decode (mode_s.c) */

static unsigned getbits(const uint8_t *data, unsigned firstbit, unsigned lastbit) { /* mode_s.h:58-115, generic form */
    unsigned v = 0;
    for (unsigned b = firstbit; b <= lastbit; ++b)
        v = (v << 1) | ((data[(b - 1) >> 3] >> (7 - ((b - 1) & 7))) & 1);
    return v;
}

static int msglen_by_type(int type) { return (type & 0x10) ? 112 : 56; } /* mode_s.c:81-83 */

/* mode_s.c:266-281 */
static int correct_aa_field(uint32_t *addr, const orc_errorinfo *ei) {
    int addr_errors = 0;
    if (!ei)
        return 0;
    for (int i = 0; i < ei->errors; ++i)
        if (ei->bit[i] >= 8 && ei->bit[i] <= 31) {
            *addr ^= 1u << (31 - ei->bit[i]);
            ++addr_errors;
        }
    return addr_errors;
}

/* mode_s.c:311-409 */
int orc_score(orc_ctx *c, const uint8_t *msg, int validbits) {
    static const uint8_t all_zeros[14];
    if (validbits < 56)
        return -2;
    int msgtype = getbits(msg, 1, 5);
    int msgbits = msglen_by_type(msgtype);
    if (validbits < msgbits)
        return -2;
    if (!memcmp(all_zeros, msg, msgbits / 8))
        return -2;
    uint32_t crc = orc_checksum(msg, msgbits);
    const orc_errorinfo *ei;
    uint32_t addr;
    switch (msgtype) {
        case 0: case 4: case 5: case 16:
        case 24: case 25: case 26: case 27: case 28: case 29: case 30: case 31:
            return filter_test(&c->filter, crc) ? 1000 : -1;
        case 11: {
            int iid = crc & 0x7f;
            crc &= 0xffff80;
            addr = getbits(msg, 9, 32);
            ei = orc_diagnose(c, crc, msgbits);
            if (!ei)
                return -2;
            if (ei->errors > 1)
                return -2;
            correct_aa_field(&addr, ei);
            if (iid == 0)
                return filter_test(&c->filter, addr) ? 1600 / (ei->errors + 1) : 750 / (ei->errors + 1);
            return filter_test(&c->filter, addr) ? 1000 / (ei->errors + 1) : -1;
        }
        case 17: case 18:
            ei = orc_diagnose(c, crc, msgbits);
            if (!ei)
                return -2;
            addr = getbits(msg, 9, 32);
            correct_aa_field(&addr, ei);
            return filter_test(&c->filter, addr) ? 1800 / (ei->errors + 1) : 1400 / (ei->errors + 1);
        case 20: case 21:
            return filter_test(&c->filter, crc) ? 1000 : -2;
        default:
            return -2;
    }
}

/* the CRC-dependent part of decodeModesMessage, mode_s.c:424-555 and 717-726.
 * m->msg must already hold the 14 sliced bytes. Returns 0 / -1 / -2. */
static int decode_crc_part(orc_ctx *c, orc_msg *m, uint8_t msg[14]) {
    static const uint8_t all_zeros[7];
    if (!memcmp(all_zeros, msg, 7))
        return -2;
    m->msgtype = getbits(msg, 1, 5);
    m->msgbits = msglen_by_type(m->msgtype);
    m->crc = orc_checksum(msg, m->msgbits);
    m->correctedbits = 0;
    m->addr = 0;
    m->iid = 0;
    switch (m->msgtype) {
        case 0: case 4: case 5: case 16:
        case 24: case 25: case 26: case 27: case 28: case 29: case 30: case 31:
            if (!filter_test(&c->filter, m->crc))
                return -1;
            m->source = 3; /* SOURCE_MODE_S, readsb.h:131-141 */
            m->addr = m->crc;
            break;
        case 11:
            m->iid = m->crc & 0x7f;
            if (m->crc & 0xffff80) {
                const orc_errorinfo *ei = orc_diagnose(c, m->crc & 0xffff80, m->msgbits);
                if (!ei)
                    return -2;
                if (ei->errors > 1)
                    return -2;
                m->correctedbits = ei->errors;
                checksum_fix(msg, ei);
                if (!filter_test(&c->filter, getbits(msg, 9, 32)))
                    return -1;
            }
            m->source = 4; /* SOURCE_MODE_S_CHECKED */
            break;
        case 17: case 18:
            if (m->crc != 0) {
                const orc_errorinfo *ei = orc_diagnose(c, m->crc, m->msgbits);
                if (!ei)
                    return -2;
                uint32_t addr1 = getbits(msg, 9, 32);
                m->correctedbits = ei->errors;
                checksum_fix(msg, ei);
                uint32_t addr2 = getbits(msg, 9, 32);
                if (addr1 != addr2 && !filter_test(&c->filter, addr2))
                    return -1;
            }
            m->source = 7; /* SOURCE_ADSB */
            break;
        case 20: case 21:
            if (filter_test(&c->filter, m->crc)) {
                m->source = 3;
                m->addr = m->crc;
                break;
            }
            return -1;
        default:
            return -2;
    }
    if (m->msgtype == 11 || m->msgtype == 17 || m->msgtype == 18)
        m->addr = getbits(msg, 9, 32); /* mode_s.c:560-562 */
    if (!m->correctedbits && (m->msgtype == 17 || (m->msgtype == 11 && m->iid == 0)))
        filter_add(&c->filter, m->addr); /* mode_s.c:717-726: the only add site */
    return 0;
}

static void emit(orc_ctx *c, const orc_msg *m) {
    if (c->nout < c->cap)
        c->out[c->nout] = *m;
    ++c->nout;
}

/* ------------------------------------------------------------------ demodulate2400 (demod_2400.c:72-383) */

static inline int slice_phase0(const uint16_t *m) { return 5 * m[0] - 3 * m[1] - 2 * m[2]; }     /* :48-50 */
static inline int slice_phase1(const uint16_t *m) { return 4 * m[0] - m[1] - 3 * m[2]; }         /* :52-54 */
static inline int slice_phase2(const uint16_t *m) { return 3 * m[0] + m[1] - 4 * m[2]; }         /* :56-58 */
static inline int slice_phase3(const uint16_t *m) { return 2 * m[0] + 3 * m[1] - 5 * m[2]; }     /* :60-62 */
static inline int slice_phase4(const uint16_t *m) { return m[0] + 5 * m[1] - 5 * m[2] - m[3]; }  /* :64-66 */

static int slice_bit(const uint16_t *p, int phase) {
    switch (phase) {
        case 0: return slice_phase0(p) > 0;
        case 1: return slice_phase1(p) > 0;
        case 2: return slice_phase2(p) > 0;
        case 3: return slice_phase3(p) > 0;
        default: return slice_phase4(p) > 0;
    }
}

/* demod_2400.c:109-172 */
static int preamble_gate(const uint16_t *p) {
        int high;
        uint32_t base_signal, base_noise;

        if (!(p[0] < p[1] && p[12] > p[13])) /* :109 */
            return 0;
        if (p[1] > p[2] && p[2] < p[3] && p[3] > p[4] && p[8] < p[9] && p[9] > p[10] && p[10] < p[11]) { /* :112-119 */
            high = (p[1] + p[3] + p[9] + p[11] + p[12]) / 4;
            base_signal = p[1] + p[3] + p[9];
            base_noise = p[5] + p[6] + p[7];
        } else if (p[1] > p[2] && p[2] < p[3] && p[3] > p[4] && p[8] < p[9] && p[9] > p[10] && p[11] < p[12]) { /* :120-127 */
            high = (p[1] + p[3] + p[9] + p[12]) / 4;
            base_signal = p[1] + p[3] + p[9] + p[12];
            base_noise = p[5] + p[6] + p[7] + p[8];
        } else if (p[1] > p[2] && p[2] < p[3] && p[4] > p[5] && p[8] < p[9] && p[10] > p[11] && p[11] < p[12]) { /* :128-135 */
            high = (p[1] + p[3] + p[4] + p[9] + p[10] + p[12]) / 4;
            base_signal = p[1] + p[12];
            base_noise = p[6] + p[7];
        } else if (p[1] > p[2] && p[3] < p[4] && p[4] > p[5] && p[9] < p[10] && p[10] > p[11] && p[11] < p[12]) { /* :136-143 */
            high = (p[1] + p[4] + p[10] + p[12]) / 4;
            base_signal = p[1] + p[4] + p[10] + p[12];
            base_noise = p[5] + p[6] + p[7] + p[8];
        } else if (p[2] > p[3] && p[3] < p[4] && p[4] > p[5] && p[9] < p[10] && p[10] > p[11] && p[11] < p[12]) { /* :144-151 */
            high = (p[1] + p[2] + p[4] + p[10] + p[12]) / 4;
            base_signal = p[4] + p[10] + p[12];
            base_noise = p[6] + p[7] + p[8];
        } else {
            return 0;
        }
        if (base_signal * 2 < 3 * base_noise) /* :158 */
            return 0;
        if (p[5] >= high || p[6] >= high || p[7] >= high || p[8] >= high || p[14] >= high ||
                p[15] >= high || p[16] >= high || p[17] >= high || p[18] >= high) /* :162-172 */
            return 0;
        return 1;
}

/* demod_2400.c:179-286 for one trial phase, closed form: bit k is sliced at
 * P = try_phase + 12k, sample 19 + P/5, correlator P%5. Returns bytes sliced (14, 7 or 1). */
static int slice_message(const uint16_t *pre, int try_phase, uint8_t *msg) {
    int bytelen = 14, i;
    for (i = 0; i < bytelen; ++i) {
        uint8_t b = 0;
        for (int bit = 0; bit < 8; ++bit) {
            int P = try_phase + 12 * (i * 8 + bit);
            b = (uint8_t) ((b << 1) | slice_bit(&pre[19 + P / 5], P % 5));
        }
        msg[i] = b;
        if (i == 0) { /* :272-285 */
            switch (msg[0] >> 3) {
                case 0: case 4: case 5: case 11:
                    bytelen = 7;
                    break;
                case 16: case 17: case 18: case 20: case 21: case 24:
                    break;
                default:
                    bytelen = 1;
                    break;
            }
        }
    }
    return i;
}

static void demod2400(orc_ctx *c, const uint16_t *m, uint32_t mlen, uint64_t sampleTimestamp) {
    uint8_t msg1[14], msg2[14], *msg = msg1;
    memset(msg1, 0, sizeof (msg1));
    memset(msg2, 0, sizeof (msg2));

    for (uint32_t j = 0; j < mlen; j++) {
        if (c->cand_out) { /* candidate dump mode: no scoring, no skipping */
            if (preamble_gate(&m[j])) {
                if (c->ncand < c->cand_cap) {
                    orc_candidate *r = &c->cand_out[c->ncand];
                    memset(r, 0, sizeof (*r));
                    r->g = (uint32_t) (sampleTimestamp / 5 + j);
                    for (int p = 0; p < 5; ++p)
                        r->vbytes[p] = (uint8_t) slice_message(&m[j], p + 4, r->msg[p]);
                    for (int k = 0; k < 268; ++k) {
                        uint64_t v = (uint64_t) m[j + 19 + k] * m[j + 19 + k];
                        r->pow268 += v;
                        if (k < 134)
                            r->pow134 += v;
                    }
                }
                c->ncand++;
            }
            continue;
        }
        if (!preamble_gate(&m[j]))
            continue;

        c->stats.preambles++;
        uint8_t *bestmsg = NULL;
        int bestscore = -2, bestphase = -1;
        for (int try_phase = 4; try_phase <= 8; ++try_phase) { /* :179-301 */
            int i = slice_message(&m[j], try_phase, msg);
            int score = orc_score(c, msg, i * 8);
            if (score > bestscore) { /* :290-300 */
                bestmsg = msg;
                bestscore = score;
                bestphase = try_phase;
                msg = (msg == msg1) ? msg2 : msg1;
            }
        }

        if (bestscore < 0) { /* :304-310 */
            if (bestscore == -1)
                c->stats.rejected_unknown++;
            else
                c->stats.rejected_bad++;
            continue;
        }

        int msglen = msglen_by_type(bestmsg[0] >> 3);
        orc_msg mm;
        memset(&mm, 0, sizeof (mm));
        mm.ts = sampleTimestamp + (uint64_t) j * 5 + (8 + 56) * 12 + bestphase; /* :320 */
        mm.score = bestscore;
        uint8_t work[14];
        memcpy(work, bestmsg, 14); /* mode_s.c:426 */
        int result = decode_crc_part(c, &mm, work);
        if (result < 0) { /* :330-335 */
            if (result == -1)
                c->stats.rejected_unknown++;
            else
                c->stats.rejected_bad++;
            continue;
        }
        c->stats.accepted[mm.correctedbits]++;
        memcpy(mm.msg, work, mm.msgbits / 8);

        /* :342-363 */
        uint64_t scaled_signal_power = 0;
        int signal_len = msglen * 12 / 5;
        for (int k = 0; k < signal_len; ++k) {
            uint32_t mag = m[j + 19 + k];
            scaled_signal_power += mag * mag;
        }
        double signal_power = scaled_signal_power / 65535.0 / 65535.0;
        mm.signal = signal_power / signal_len;

        j += msglen * 12 / 5; /* :371 */
        emit(c, &mm);
    }
}

/* ------------------------------------------------------------------ demodulate2400AC (demod_2400.c:476-662) */

static void demod2400AC(orc_ctx *c, const uint16_t *m, uint32_t mlen, uint64_t sampleTimestamp, double mean_level, double mean_power) {
    double noise_stddev = sqrt(mean_power - mean_level * mean_level);                 /* :484 */
    unsigned noise_level = (unsigned) ((mean_power + noise_stddev) * 65535 + 0.5);    /* :485 */

    for (unsigned f1_sample = 1; f1_sample < mlen; ++f1_sample) {
        if (!(m[f1_sample - 1] < m[f1_sample + 0]))
            continue;
        if (m[f1_sample + 2] > m[f1_sample + 0] || m[f1_sample + 2] > m[f1_sample + 1])
            continue;
        unsigned f1_level = (m[f1_sample + 0] + m[f1_sample + 1]) / 2;
        if (noise_level * 2 > f1_level)
            continue;

        float f1a_power = (float) m[f1_sample] * m[f1_sample];           /* :551 */
        float f1b_power = (float) m[f1_sample + 1] * m[f1_sample + 1];   /* :552 */
        float fraction = f1b_power / (f1a_power + f1b_power);
        unsigned f1_clock = (unsigned) (25 * (f1_sample + fraction * fraction) + 0.5); /* :554 */

        unsigned f2_clock = f1_clock + (87 * 14);
        unsigned f2_sample = f2_clock / 25;
        if (!(m[f2_sample - 1] < m[f2_sample + 0]))
            continue;
        if (m[f2_sample + 2] > m[f2_sample + 0] || m[f2_sample + 2] > m[f2_sample + 1])
            continue;
        unsigned f2_level = (m[f2_sample + 0] + m[f2_sample + 1]) / 2;
        if (noise_level * 2 > f2_level)
            continue;

        unsigned f1f2_level = (f1_level > f2_level ? f1_level : f2_level);
        float midpoint = sqrtf(noise_level * f1f2_level);                        /* :577 (unsigned product -> float) */
        unsigned signal_threshold = (unsigned) (midpoint * M_SQRT2 + 0.5);       /* :578 */
        unsigned noise_threshold = (unsigned) (midpoint / M_SQRT2 + 0.5);        /* :579 */

        unsigned uncertain_bits = 0, noisy_bits = 0, bits = 0, bit, clock;
        for (bit = 0, clock = f1_clock; bit < 20; ++bit, clock += 87) {
            unsigned sample = clock / 25;
            bits <<= 1;
            noisy_bits <<= 1;
            uncertain_bits <<= 1;
            if (m[sample + 2] >= signal_threshold)
                noisy_bits |= 1;
            if (m[sample + 0] >= signal_threshold || m[sample + 1] >= signal_threshold)
                bits |= 1;
            else if (m[sample + 0] > noise_threshold && m[sample + 1] > noise_threshold)
                uncertain_bits |= 1;
        }
        if ((bits & 0x80020) != 0x80020)
            continue;
        if ((bits & 0x0101B) != 0)
            continue;
        if (noisy_bits || uncertain_bits)
            continue;

        unsigned modeac = /* :626-639 */
                ((bits & 0x40000) ? 0x0010 : 0) | ((bits & 0x20000) ? 0x1000 : 0) |
                ((bits & 0x10000) ? 0x0020 : 0) | ((bits & 0x08000) ? 0x2000 : 0) |
                ((bits & 0x04000) ? 0x0040 : 0) | ((bits & 0x02000) ? 0x4000 : 0) |
                ((bits & 0x00800) ? 0x0100 : 0) | ((bits & 0x00400) ? 0x0001 : 0) |
                ((bits & 0x00200) ? 0x0200 : 0) | ((bits & 0x00100) ? 0x0002 : 0) |
                ((bits & 0x00080) ? 0x0400 : 0) | ((bits & 0x00040) ? 0x0004 : 0) |
                ((bits & 0x00004) ? 0x0080 : 0);

        if (c->ac_out) { /* candidate dump mode: every passing offset, no skip */
            if (c->nac < c->ac_cap) {
                c->ac_out[c->nac].g = (uint32_t) (sampleTimestamp / 5 + f1_sample);
                c->ac_out[c->nac].f2_clock = f2_clock;
                c->ac_out[c->nac].modeac = modeac;
            }
            c->nac++;
            continue;
        }

        orc_msg mm;
        memset(&mm, 0, sizeof (mm));
        mm.ts = sampleTimestamp + f2_clock / 5; /* :649 */
        /* decodeModeAMessage, mode_ac.c:165-199 (message fields only) */
        mm.source = 1; /* SOURCE_MODE_AC */
        mm.msgtype = 32;
        mm.msgbits = 16;
        mm.msg[0] = (uint8_t) (modeac >> 8);
        mm.msg[1] = (uint8_t) modeac;
        mm.addr = (modeac & 0x0000FF7F) | (1u << 24);
        mm.correctedbits = 0;
        emit(c, &mm);

        f1_sample += (20 * 87 / 25); /* :659 */
        c->stats.modeac++;
    }
}

/* ------------------------------------------------------------------ the sdr_ifile.c block loop (sdr_ifile.c:184-228) */

/* mags != NULL: the magnitudes (and block_means[2k], [2k+1] = mean level / power of block k) come from the caller
 * instead of orc_convert -- the demodulators on somebody else's converter output */
static size_t run_blocks(orc_ctx *c, const void *iq, const uint16_t *mags, const double *block_means, uint64_t nsamples, orc_msg *out, size_t cap) {
    const unsigned bps = (c->format == ORC_UC8) ? 2 : 4;
    uint16_t *buf[2];
    uint32_t length[2] = {0, 0};
    for (int i = 0; i < 2; ++i)
        buf[i] = calloc(ORC_BLOCK_SAMPLES + ORC_TRAILING, sizeof (uint16_t));
    c->out = out;
    c->cap = cap;
    c->nout = 0;

    uint64_t pos = 0, sampleCounter = 0;
    int cur = 0, eof = 0;
    while (!eof) {
        uint16_t *data = buf[cur], *last = buf[cur ^ 1];
        uint64_t sampleTimestamp = sampleCounter * 5; /* sampleCounter*12e6/2.4e6, exact */
        sampleCounter += ORC_BLOCK_SAMPLES;
        if (length[cur ^ 1] >= ORC_TRAILING)
            memcpy(data, last + length[cur ^ 1], ORC_TRAILING * sizeof (uint16_t));
        else
            memset(data, 0, ORC_TRAILING * sizeof (uint16_t));
        uint64_t want = ORC_BLOCK_SAMPLES, have = nsamples - pos;
        if (have < want) {
            want = have;
            eof = 1;
        }
        double mean_level = 0, mean_power = 0;
        length[cur] = (uint32_t) want;
        if (mags) {
            memcpy(&data[ORC_TRAILING], mags + pos, want * sizeof (uint16_t));
            if (block_means) {
                mean_level = block_means[2 * (pos / ORC_BLOCK_SAMPLES)];
                mean_power = block_means[2 * (pos / ORC_BLOCK_SAMPLES) + 1];
            }
        } else {
            orc_convert(c, (const uint8_t *) iq + pos * bps, &data[ORC_TRAILING], (unsigned) want, &mean_level, &mean_power);
        }
        pos += want;
        demod2400(c, data, (uint32_t) want, sampleTimestamp);
        if (c->modeac)
            demod2400AC(c, data, (uint32_t) want, sampleTimestamp, mean_level, mean_power);
        cur ^= 1;
    }
    free(buf[0]);
    free(buf[1]);
    c->out = NULL;
    return c->nout;
}

/* ------------------------------------------------------------------ a chain of mag_bufs as a hardware driver builds it
 * (sdr_bladerf.c:322-404, without the USB framing): every buffer has its own length (<= 131072) and sampleTimestamp, its
 * overlap is ALWAYS the 326 samples behind the previous buffer's data (:323-324: data + length of the ring slot before it,
 * zeros before the first -- the ring is calloc'ed, readsb.c:198-207 -- and part of that buffer's own overlap when it was
 * shorter than 326), the converter runs once per sub-block of sub_len samples and the buffer's means are the averages of
 * the sub-blocks' means (:385-390, 403-404).  Both demodulators run once per buffer (readsb.c:883-886). */
size_t orc_run_bufs(orc_ctx *c, const void *iq, const orc_buf *bufs, size_t nbufs, uint32_t sub_len, orc_msg *out, size_t cap) {
    const unsigned bps = (c->format == ORC_UC8) ? 2 : 4;
    uint16_t *buf[2];
    uint32_t length[2] = {0, 0};
    for (int i = 0; i < 2; ++i)
        buf[i] = calloc(ORC_BLOCK_SAMPLES + ORC_TRAILING, sizeof (uint16_t));
    c->out = out;
    c->cap = cap;
    c->nout = 0;
    int cur = 0;
    for (size_t k = 0; k < nbufs; ++k) {
        uint16_t *data = buf[cur], *last = buf[cur ^ 1];
        const uint32_t len = bufs[k].length > ORC_BLOCK_SAMPLES ? ORC_BLOCK_SAMPLES : bufs[k].length;
        memcpy(data, last + length[cur ^ 1], ORC_TRAILING * sizeof (uint16_t));
        double mean_level = 0, mean_power = 0;
        unsigned blocks = 0;
        const uint32_t step = sub_len ? sub_len : len;
        for (uint32_t off = 0; off < len; off += step, ++blocks) {
            const uint32_t n = len - off < step ? len - off : step;
            double ml = 0, mp = 0;
            orc_convert(c, (const uint8_t *) iq + (bufs[k].start + off) * bps, &data[ORC_TRAILING + off], n, &ml, &mp);
            mean_level += ml;
            mean_power += mp;
        }
        if (!blocks)
            continue; /* nothing converted: the buffer is not published (:398) */
        mean_level /= blocks;
        mean_power /= blocks;
        length[cur] = len;
        demod2400(c, data, len, bufs[k].sample_timestamp);
        if (c->modeac)
            demod2400AC(c, data, len, bufs[k].sample_timestamp, mean_level, mean_power);
        cur ^= 1;
    }
    free(buf[0]);
    free(buf[1]);
    c->out = NULL;
    return c->nout;
}

size_t orc_run_stream(orc_ctx *c, const void *iq, uint64_t nsamples, orc_msg *out, size_t cap) {
    return run_blocks(c, iq, NULL, NULL, nsamples, out, cap);
}

size_t orc_run_mag_stream(orc_ctx *c, const uint16_t *mags, const double *block_means, uint64_t nsamples, orc_msg *out, size_t cap) {
    return run_blocks(c, NULL, mags, block_means, nsamples, out, cap);
}

size_t orc_gated_candidates(orc_ctx *c, const void *iq, uint64_t nsamples, orc_candidate *out, size_t cap) {
    int modeac = c->modeac;
    c->modeac = 0;
    c->cand_out = out ? out : (orc_candidate *) (uintptr_t) 1;
    c->cand_cap = out ? cap : 0;
    c->ncand = 0;
    orc_run_stream(c, iq, nsamples, NULL, 0);
    c->cand_out = NULL;
    c->modeac = modeac;
    return c->ncand;
}

size_t orc_ac_candidates(orc_ctx *c, const void *iq, uint64_t nsamples, orc_ac_candidate *out, size_t cap) {
    int modeac = c->modeac;
    c->modeac = 1;
    c->ac_out = out ? out : (orc_ac_candidate *) (uintptr_t) 1;
    c->ac_cap = out ? cap : 0;
    c->nac = 0;
    orc_msg dummy;
    orc_run_stream(c, iq, nsamples, &dummy, 0);
    c->ac_out = NULL;
    c->modeac = modeac;
    return c->nac;
}

/* ------------------------------------------------------------------------------------------------
 * wire formats
 * ---------------------------------------------------------------------------------------------- */

#include <math.h>

static size_t put_byte(uint8_t *out, size_t cap, size_t pos, uint8_t b) {
    if (pos < cap) out[pos] = b;
    return pos + 1;
}

static size_t put_escaped(uint8_t *out, size_t cap, size_t pos, uint8_t b) { /* net_io.c:783-786 */
    pos = put_byte(out, cap, pos, b);
    if (b == 0x1a) pos = put_byte(out, cap, pos, b);
    return pos;
}

size_t orc_beast_encode(const orc_msg *m, size_t n, uint8_t *out, size_t cap) {
    size_t pos = 0;
    for (size_t i = 0; i < n; ++i) {
        if (m[i].correctedbits >= 2) continue; /* net_io.c:1264 with net_verbatim off */
        const int len = m[i].msgbits / 8;      /* net_io.c:761 */
        char type;
        if (len == 7) type = '2';              /* net_io.c:772-780 */
        else if (len == 14) type = '3';
        else if (len == 2) type = '1';
        else continue;
        pos = put_byte(out, cap, pos, 0x1a);
        pos = put_byte(out, cap, pos, (uint8_t) type);
        for (int sh = 40; sh >= 0; sh -= 8) pos = put_escaped(out, cap, pos, (uint8_t) (m[i].ts >> sh)); /* :783-806 */
        int sig = (int) round(sqrt(m[i].signal) * 255); /* :808-812 */
        if (m[i].signal > 0 && sig < 1) sig = 1;
        if (sig > 255) sig = 255;
        pos = put_escaped(out, cap, pos, (uint8_t) sig);
        for (int j = 0; j < len; ++j) pos = put_escaped(out, cap, pos, m[i].msg[j]); /* :818-823 */
    }
    return pos;
}

size_t orc_raw_encode(const orc_msg *m, size_t n, int mlat, uint8_t *out, size_t cap) {
    static const char hex[] = "0123456789ABCDEF"; /* net_io.c:849 */
    size_t pos = 0;
    for (size_t i = 0; i < n; ++i) {
        if (m[i].correctedbits >= 2) continue; /* net_io.c:1258 */
        const int len = m[i].msgbits / 8;
        if (mlat && m[i].ts) { /* net_io.c:868-872 */
            pos = put_byte(out, cap, pos, '@');
            for (int sh = 44; sh >= 0; sh -= 4) pos = put_byte(out, cap, pos, (uint8_t) hex[(m[i].ts >> sh) & 15]);
        } else {
            pos = put_byte(out, cap, pos, '*');
        }
        for (int j = 0; j < len; ++j) { /* :876-879 */
            pos = put_byte(out, cap, pos, (uint8_t) hex[m[i].msg[j] >> 4]);
            pos = put_byte(out, cap, pos, (uint8_t) hex[m[i].msg[j] & 15]);
        }
        pos = put_byte(out, cap, pos, ';');
        pos = put_byte(out, cap, pos, '\n');
    }
    return pos;
}

size_t orc_forward_order(const orc_msg *m, size_t n, uint32_t *idx) {
    /* per address: number of messages so far and the index of the first one (a->messages, a->first_message) */
    size_t cap = 64, used = 0, k = 0;
    while (cap < 4 * n + 16) cap <<= 1;
    uint32_t *key = calloc(cap, sizeof (uint32_t)), *cnt = calloc(cap, sizeof (uint32_t)), *first = calloc(cap, sizeof (uint32_t));
    for (size_t i = 0; i < n; ++i) {
        if (m[i].msgtype == 32 || m[i].addr == 0) { /* track.c:962-971: no aircraft -> mode_s.c:2164 sends unconditionally */
            idx[k++] = (uint32_t) i;
            continue;
        }
        size_t h = (m[i].addr * 2654435761u) & (cap - 1);
        while (cnt[h] && key[h] != m[i].addr) h = (h + 1) & (cap - 1);
        if (!cnt[h]) {
            key[h] = m[i].addr;
            first[h] = (uint32_t) i;
            ++used;
        }
        ++cnt[h];                       /* track.c:988 */
        if (cnt[h] > 1) {               /* mode_s.c:2167-2173 */
            if (cnt[h] == 2) idx[k++] = first[h];
            idx[k++] = (uint32_t) i;
        }
    }
    (void) used;
    free(key);
    free(cnt);
    free(first);
    return k;
}

/* ---- field decode: the floating-point members of decodeESAirborneVelocity ---- */

/* mode_s.c:831 (sqrtf of a double expression narrowed to float) and :835-839 (atan2 in double, degrees, narrowed to float,
 * negative angles moved up by 360 in float) */
void orc_es_velocity(const int32_t *ew, const int32_t *ns, size_t n, float *gs, float *track) {
    for (size_t i = 0; i < n; ++i) {
        int ew_vel = ew[i], ns_vel = ns[i];
        gs[i] = sqrtf((ns_vel * ns_vel) + (ew_vel * ew_vel) + 0.5);
        float ground_track = atan2(ew_vel, ns_vel) * 180.0 / M_PI;
        if (ground_track < 0)
            ground_track += 360;
        track[i] = ground_track;
    }
}
