/*
 * modes_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C11) of readsb's receive-side hot path, used solely as the
 * checker for the HIP path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 * Nothing under readsb_amd/ may include, link or call this.
 *
 * Parity pin: this restatement is validated message-for-message (and counter-for-counter)
 * against the UNMODIFIED reference compiled into oracle/_ref/ref_harness, and against the
 * golden fixtures under tests/golden/ that ref_harness produced (tests/test_oracle_*.py).
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference tree, Mictronics/readsb v3.9.0).
 */
#ifndef MODES_ORACLE_H
#define MODES_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_BLOCK_SAMPLES 131072u /* MODES_MAG_BUF_SAMPLES, readsb.h:96-97 */
#define ORC_TRAILING 326u         /* Modes.trailing_samples, readsb.c:196 */

enum { ORC_UC8 = 0, ORC_SC16 = 1, ORC_SC16Q11 = 2 }; /* input_format_t, convert.h:29-32 */

/* same on-disk layout as oracle/ref_harness.c struct rec (packed, 51 bytes) */
#pragma pack(push, 1)
typedef struct {
    uint64_t ts;       /* timestampMsg, 12 MHz */
    uint8_t msg[14];   /* first msgbits/8 bytes valid, rest zero */
    uint8_t msgbits;   /* 56 / 112 / 16 (Mode A/C) */
    uint8_t msgtype;   /* DF, or 32 for Mode A/C */
    uint32_t crc;
    int32_t correctedbits;
    uint32_t addr;
    int32_t score;
    int32_t source;
    uint32_t iid;
    double signal;     /* signalLevel */
} orc_msg;
#pragma pack(pop)

typedef struct {
    uint64_t preambles, rejected_bad, rejected_unknown, accepted[3], modeac;
} orc_stats;

typedef struct {
    uint32_t syndrome;
    int32_t errors;
    int8_t bit[2];
    uint16_t pad;
} orc_errorinfo; /* struct errorinfo, crc.h:32-38 */

typedef struct orc_ctx orc_ctx;

orc_ctx *orc_create(int format, int dc_filter, int nfix, int modeac);
void orc_destroy(orc_ctx *c);
void orc_set_sc16q11_table_bits(orc_ctx *c, int bits); /* 0 = the default build of the reference */
void orc_reset(orc_ctx *c); /* filter, converter state, stats, stream position */

/* whole-stream run in the sdr_ifile.c block model; returns number of messages (may exceed cap;
 * only the first cap are stored) */
size_t orc_run_stream(orc_ctx *c, const void *iq, uint64_t nsamples, orc_msg *out, size_t cap);
/* the same block loop on magnitudes the caller converted (block_means: mean level / power per 131072-sample block, or NULL) */
size_t orc_run_mag_stream(orc_ctx *c, const uint16_t *mags, const double *block_means, uint64_t nsamples, orc_msg *out, size_t cap);
/* a chain of mag_bufs of the caller's geometry (a hardware driver's: sdr_bladerf.c:322-404): buffer k = samples
 * [start, start + length) of iq with its own sampleTimestamp; converter calls per sub_len samples (0: one per buffer),
 * the buffer's means = the average of the calls' means */
typedef struct {
    uint64_t start;
    uint64_t length;
    uint64_t sample_timestamp;
} orc_buf;
size_t orc_run_bufs(orc_ctx *c, const void *iq, const orc_buf *bufs, size_t nbufs, uint32_t sub_len, orc_msg *out, size_t cap);
void orc_get_stats(const orc_ctx *c, orc_stats *st);

/* pieces, for unit tests */
void orc_convert(orc_ctx *c, const void *iq, uint16_t *mag, unsigned nsamples, double *mean_level, double *mean_power);
uint32_t orc_checksum(const uint8_t *msg, int bits);
const orc_errorinfo *orc_diagnose(const orc_ctx *c, uint32_t syndrome, int bits);
int orc_table_size(const orc_ctx *c, int bits);
const orc_errorinfo *orc_table(const orc_ctx *c, int bits);
int orc_score(orc_ctx *c, const uint8_t *msg, int validbits);
void orc_filter_add(orc_ctx *c, uint32_t addr);
int orc_filter_test(const orc_ctx *c, uint32_t addr);
const uint16_t *orc_uc8_lut(void);

/* Every scan offset of the stream that passes the preamble gate (demod_2400.c:109-172), with the
 * five sliced trial messages and the two signal-power sums -- i.e. what the greedy loop would see
 * if it never skipped.  Layout = the product's 96-byte candidate record, so tests can push the
 * oracle's candidates through the product's host replay without a GPU. */
typedef struct {
    uint32_t g;
    uint8_t vbytes[5];
    uint8_t pad;
    uint8_t msg[5][14];
    uint64_t pow134, pow268;
} orc_candidate;
size_t orc_gated_candidates(orc_ctx *c, const void *iq, uint64_t nsamples, orc_candidate *out, size_t cap);
/* Mode A/C offsets passing every per-offset test (demod_2400.c:535-622), before the skip of :659 */
typedef struct {
    uint32_t g, f2_clock, modeac;
} orc_ac_candidate;
size_t orc_ac_candidates(orc_ctx *c, const void *iq, uint64_t nsamples, orc_ac_candidate *out, size_t cap);

/* ---- wire formats ---- */
/* modesSendBeastOutput (net_io.c:760-826) for each record that modesQueueOutput forwards with net_verbatim off
 * (correctedbits < 2, :1264); returns the number of bytes (may exceed cap; only the first cap are stored) */
size_t orc_beast_encode(const orc_msg *m, size_t n, uint8_t *out, size_t cap);
/* modesSendRawOutput (net_io.c:859-885); mlat != 0 = Modes.mlat ("@" + 12 hex digits of the timestamp) */
size_t orc_raw_encode(const orc_msg *m, size_t n, int mlat, uint8_t *out, size_t cap);
/* the forwarding rule of useModesMessage (mode_s.c:2163-2175) over trackUpdateFromMessage's message count
 * (track.c:958-988; aircraft never expire here): idx receives the indices of the records handed to
 * modesQueueOutput, in order (a first message goes out just before the second); returns their number (<= 2n) */
size_t orc_forward_order(const orc_msg *m, size_t n, uint32_t *idx);

/* ---- field decode ---- */
/* The two floating-point members of decodeESAirborneVelocity (mode_s.c:826-842) for n pairs of signed velocity components:
 * gs = sqrtf(ns^2 + ew^2 + 0.5) and the ground track atan2(ew, ns) in degrees, 0 .. 360, through this host's libm -- the same glibc
 * the reference links.  (Everything else of the field decode is integer work and is checked against the reference's own structs.) */
void orc_es_velocity(const int32_t *ew, const int32_t *ns, size_t n, float *gs, float *track);

#ifdef __cplusplus
}
#endif
#endif
