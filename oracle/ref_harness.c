/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * A single-threaded driver around the UNMODIFIED reference objects
 * (/root/reference/{convert,demod_2400,mode_s,crc,icao_filter,mode_ac,...}.c,
 * compiled where they lie by oracle/Makefile into oracle/_ref/).  It replaces
 * only readsb.c's main(): it reproduces modesInit (readsb.c:187-251) and the
 * block/overlap logic of ifileRun (sdr_ifile.c:184-228) on one thread and
 * captures every message the reference hands to useModesMessage through
 * -Wl,--wrap=useModesMessage (recipe: SURVEY.md Appendix C).
 *
 * icaoFilterExpire() is never called, so runs are deterministic
 * (SURVEY.md Appendix B.2).
 *
 * Sub-commands:
 *   demod   <fmt> <nfix> <modeac> <dcfilter> <in.iq> <out.bin>   message list + stats
 *   demodv  (same arguments)                                       the same with Modes.net_verbatim set; behind the records: msg[14] as
 *                                                                 delivered and verbatim[14] (mode_s.c:427-430) of every message
 *
 *   demodfull (same arguments as demod)                            behind the records: the whole struct modesMessage (416 bytes) of every message as
 *                                                                 useModesMessage receives it, WITHOUT decodeCommB's work (the harness is linked
 *                                                                 with -Wl,--wrap=decodeCommB and skips the call in this mode and in decodemsgs: comm_b.c
 *                                                                 is outside SURVEY.md section 8; every other mode calls the real one)
 *   demodfullb (same arguments)                                    the whole struct as well, decodeCommB included: what an unmodified readsb hands on
 *   timedecode <in.bin> <reps>                                     seconds the reference's decodeModesMessage (the real decodeCommB included) takes for the
 *                                                                 frames of a decodemsgs input file, reps passes, one thread: the host-callback baseline
 *   decodemsgs <in.bin> <out.bin>                                  decodeModesMessage (mode_s.c:424-734) on frames from a file: in = {u32 naddr, u32 nframes,
 *                                                                 u32 addr[naddr] (icaoFilterAdd'ed first), u8 frame[nframes][14]}; out = per frame
 *                                                                 {i32 result, struct modesMessage}, then u32 stats_current.cpr_filtered
 *
 *   demodbufs <fmt> <nfix> <modeac> <dcfilter> <sublen> <in.iq> <bufs.bin> <out.bin>
 *                                                                 a chain of mag_bufs of the caller's geometry, as a hardware driver
 *                                                                 builds it (sdr_bladerf.c:322-404 without the USB framing): bufs.bin =
 *                                                                 {u64 start, u64 length, u64 sampleTimestamp} per buffer; the converter is
 *                                                                 called per sublen samples and the buffer's means are the calls' averages
 *
 * Built a second time as oracle/_ref/ref_harness_gpu (-DGPU_PATH): the SAME harness and reference objects, but with
 * demod_2400.o and convert.o replaced by integration/gpu_path.c + libreadsb_amd.so -- the link-level drop-in test.
 *   convert <fmt> <dcfilter> <chunk> <in.iq> <out.bin>           magnitudes + per-call means
 *   crctab  <nfix> <out.bin>                                     syndrome -> errorinfo, all 2^24 syndromes folded to a digest + table dump
 *   time    <fmt> <nfix> <modeac> <dcfilter> <in.iq> <reps>      prints seconds of convert+demod (CPU baseline, kind="reference")
 *   stats   <fmt> <nfix> <modeac> <dcfilter> <in.iq>             prints every demodulator member of Modes.stats_current (stats.h:61-79) as
 *                                                                 one JSON line, doubles as C99 hex floats (bit-exact round trip)
 *   wire    <fmt> <nfix> <modeac> <forward> <in.iq> <out.bin>    the bytes the reference's own Beast and raw writers (net_io.c:760-826,
 *                                                                 859-885) produce for the stream, through modesQueueOutput; forward=1 goes
 *                                                                 through the real useModesMessage (tracker + two-message rule, mode_s.c:2163-2175)
 */
#include "readsb.h"
#include <inttypes.h>

struct _Modes Modes;

void receiverPositionChanged(float lat, float lon, float alt) {
    (void) lat; (void) lon; (void) alt;
}

/* on-disk record; must match oracle/oracle.py REC_DTYPE */
struct rec {
    uint64_t ts;
    uint8_t msg[14];
    uint8_t msgbits;
    uint8_t msgtype;
    uint32_t crc;
    int32_t correctedbits;
    uint32_t addr;
    int32_t score;
    int32_t source;
    uint32_t iid;
    double signal;
} __attribute__((packed));

static struct rec *recs;
static size_t nrecs, caprecs;
static int capture = 1;
static uint8_t (*verb)[2][14]; /* demodv: msg, verbatim of every captured message */
static int capture_verbatim;
static struct modesMessage *full; /* demodfull: the whole struct of every captured message */
static int capture_full;
static int skip_commb;

void __real_decodeCommB(struct modesMessage *mm);
void __wrap_decodeCommB(struct modesMessage *mm) {
    if (!skip_commb)
        __real_decodeCommB(mm);
}

/* wire capture: both writers point at a service with one pretend connection and no clients; the
 * harness drains their shared buffers after every message, so flushWrites never runs */
static struct net_service wire_service[2];
static uint8_t *wire_bytes[2];
static size_t wire_len[2], wire_cap[2];
static int wire_mode; /* 0 off, 1 modesQueueOutput per message, 2 the real useModesMessage */
void __real_useModesMessage(struct modesMessage *mm);

static void wire_drain(int k, struct net_writer *w) {
    if (wire_len[k] + w->dataUsed > wire_cap[k]) {
        wire_cap[k] = (wire_len[k] + w->dataUsed) * 2 + 65536;
        wire_bytes[k] = realloc(wire_bytes[k], wire_cap[k]);
    }
    memcpy(wire_bytes[k] + wire_len[k], w->data, w->dataUsed);
    wire_len[k] += w->dataUsed;
    w->dataUsed = 0;
}

void __wrap_useModesMessage(struct modesMessage *mm) {
    if (wire_mode) {
        if (wire_mode == 2)
            __real_useModesMessage(mm);
        else
            modesQueueOutput(mm, NULL);
        wire_drain(0, &Modes.beast_out);
        wire_drain(1, &Modes.raw_out);
    }
    if (!capture)
        return;
    if (nrecs == caprecs) {
        caprecs = caprecs ? caprecs * 2 : 4096;
        recs = realloc(recs, caprecs * sizeof (*recs));
        if (capture_verbatim) verb = realloc(verb, caprecs * sizeof (*verb));
        if (capture_full) full = realloc(full, caprecs * sizeof (*full));
    }
    if (capture_full)
        full[nrecs] = *mm;
    if (capture_verbatim) {
        memcpy(verb[nrecs][0], mm->msg, 14);
        memcpy(verb[nrecs][1], mm->verbatim, 14);
        if (mm->msgbits < 112) { /* bytes past a short frame are stale stack contents in the reference */
            memset(verb[nrecs][0] + mm->msgbits / 8, 0, 14 - mm->msgbits / 8);
            memset(verb[nrecs][1] + mm->msgbits / 8, 0, 14 - mm->msgbits / 8);
        }
    }
    struct rec *r = &recs[nrecs++];
    memset(r, 0, sizeof (*r));
    r->ts = mm->timestampMsg;
    memcpy(r->msg, mm->msg, (mm->msgbits + 7) / 8);
    r->msgbits = mm->msgbits;
    r->msgtype = mm->msgtype;
    r->crc = mm->crc;
    r->correctedbits = mm->correctedbits;
    r->addr = mm->addr;
    r->score = mm->score;
    r->source = mm->source;
    r->iid = mm->IID;
    r->signal = mm->signalLevel;
}

static input_format_t parse_fmt(const char *s) {
    if (!strcasecmp(s, "uc8")) return INPUT_UC8;
    if (!strcasecmp(s, "sc16")) return INPUT_SC16;
    if (!strcasecmp(s, "sc16q11")) return INPUT_SC16Q11;
    fprintf(stderr, "bad format %s\n", s);
    exit(2);
}

static void *slurp(const char *path, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void *p = malloc(n ? n : 1);
    if (fread(p, 1, n, f) != (size_t) n) { perror("read"); exit(2); }
    fclose(f);
    *len = n;
    return p;
}

static void init_modes(int nfix, int modeac, int dcfilter) {
    memset(&Modes, 0, sizeof (Modes));
    Modes.sample_rate = 2400000.0;
    Modes.nfix_crc = nfix;
    Modes.check_crc = 1;
    Modes.mode_ac = modeac;
    Modes.dc_filter = dcfilter;
    Modes.quiet = 1;
    Modes.trailing_samples = (MODES_PREAMBLE_US + MODES_LONG_MSG_BITS + 16) * 1e-6 * Modes.sample_rate;
    modesChecksumInit(Modes.nfix_crc);
    icaoFilterInit();
    modeACInit();
}

/* the ifileRun block loop, single threaded, two alternating mag_bufs */
static void run_stream(input_format_t fmt, const uint8_t *iq, size_t nbytes, int modeac) {
    unsigned bps = (fmt == INPUT_UC8) ? 2 : 4;
    struct converter_state *st = NULL;
    iq_convert_fn conv = init_converter(fmt, Modes.sample_rate, Modes.dc_filter, &st);
    if (!conv) exit(2);

    struct mag_buf bufs[2];
    memset(bufs, 0, sizeof (bufs));
    for (int i = 0; i < 2; ++i)
        bufs[i].data = calloc(MODES_MAG_BUF_SAMPLES + Modes.trailing_samples, sizeof (uint16_t));

    size_t total = nbytes / bps, pos = 0;
    uint64_t sampleCounter = 0;
    int cur = 0, eof = 0;
    void *readbuf = malloc(MODES_MAG_BUF_SAMPLES * bps);
    while (!eof) {
        struct mag_buf *outbuf = &bufs[cur], *lastbuf = &bufs[cur ^ 1];
        outbuf->sampleTimestamp = sampleCounter * 12e6 / Modes.sample_rate;
        sampleCounter += MODES_MAG_BUF_SAMPLES;
        if (lastbuf->length >= Modes.trailing_samples)
            memcpy(outbuf->data, lastbuf->data + lastbuf->length, Modes.trailing_samples * sizeof (uint16_t));
        else
            memset(outbuf->data, 0, Modes.trailing_samples * sizeof (uint16_t));
        outbuf->sysTimestamp = 0;

        size_t want = MODES_MAG_BUF_SAMPLES, have = total - pos;
        if (have < want) { want = have; eof = 1; }
        memcpy(readbuf, iq + pos * bps, want * bps);
        pos += want;
        outbuf->length = want;
        conv(readbuf, &outbuf->data[Modes.trailing_samples], want, st, &outbuf->mean_level, &outbuf->mean_power);

        demodulate2400(outbuf);
        if (modeac)
            demodulate2400AC(outbuf);
        cur ^= 1;
    }
    free(readbuf);
    free(bufs[0].data);
    free(bufs[1].data);
#ifdef GPU_PATH
    cleanup_converter(st);
#else
    free(st); /* not cleanup_converter: it frees the global LUT without clearing it (convert.c:493-499) */
#endif
}

/* sdr_bladerf.c:322-404 without the framing: the reference's converter and demodulators on buffers of any length */
static void run_bufs(input_format_t fmt, const uint8_t *iq, const uint64_t *bufs, size_t nbufs, unsigned sublen, int modeac) {
    unsigned bps = (fmt == INPUT_UC8) ? 2 : 4;
    struct converter_state *st = NULL;
    iq_convert_fn conv = init_converter(fmt, Modes.sample_rate, Modes.dc_filter, &st);
    if (!conv) exit(2);
    struct mag_buf ring[2];
    memset(ring, 0, sizeof (ring));
    for (int i = 0; i < 2; ++i)
        ring[i].data = calloc(MODES_MAG_BUF_SAMPLES + Modes.trailing_samples, sizeof (uint16_t));
    int cur = 0;
    for (size_t k = 0; k < nbufs; ++k) {
        struct mag_buf *outbuf = &ring[cur], *lastbuf = &ring[cur ^ 1];
        uint64_t start = bufs[3 * k], len = bufs[3 * k + 1];
        if (len > MODES_MAG_BUF_SAMPLES) len = MODES_MAG_BUF_SAMPLES;
        memcpy(outbuf->data, lastbuf->data + lastbuf->length, Modes.trailing_samples * sizeof (uint16_t)); /* :324 */
        outbuf->dropped = 0;
        outbuf->length = 0;
        outbuf->mean_level = outbuf->mean_power = 0;
        outbuf->sampleTimestamp = bufs[3 * k + 2];
        unsigned blocks_processed = 0, step = sublen ? sublen : (unsigned) len;
        while (outbuf->length < len) {
            unsigned n = (len - outbuf->length < step) ? (unsigned) (len - outbuf->length) : step;
            double mean_level, mean_power;
            conv((void *) (iq + (start + outbuf->length) * bps), &outbuf->data[Modes.trailing_samples + outbuf->length], n, st, &mean_level, &mean_power); /* :387 */
            outbuf->length += n;
            outbuf->mean_level += mean_level;
            outbuf->mean_power += mean_power;
            ++blocks_processed;
        }
        if (!blocks_processed)
            continue;
        outbuf->mean_level /= blocks_processed; /* :403-404 */
        outbuf->mean_power /= blocks_processed;
        demodulate2400(outbuf);
        if (modeac)
            demodulate2400AC(outbuf);
        cur ^= 1;
    }
    free(ring[0].data);
    free(ring[1].data);
#ifdef GPU_PATH
    cleanup_converter(st);
#else
    free(st);
#endif
}

static int write_records(const char *path) {
    FILE *f = fopen(path, "wb");
    if (!f) { perror(path); return 2; }
    uint64_t hdr[8] = {nrecs,
        Modes.stats_current.demod_preambles,
        Modes.stats_current.demod_rejected_bad,
        Modes.stats_current.demod_rejected_unknown_icao,
        Modes.stats_current.demod_accepted[0],
        Modes.stats_current.demod_accepted[1],
        Modes.stats_current.demod_accepted[2],
        Modes.stats_current.demod_modeac};
    fwrite(hdr, sizeof (hdr), 1, f);
    fwrite(recs, sizeof (*recs), nrecs, f);
    if (capture_verbatim) fwrite(verb, sizeof (*verb), nrecs, f);
    fclose(f);
    return 0;
}

static int cmd_demodbufs(int argc, char **argv) {
    if (argc != 10) return 2;
    input_format_t fmt = parse_fmt(argv[2]);
    int nfix = atoi(argv[3]), modeac = atoi(argv[4]), dc = atoi(argv[5]);
    unsigned sublen = (unsigned) atoi(argv[6]);
    size_t nbytes, lbytes;
    uint8_t *iq = slurp(argv[7], &nbytes);
    uint64_t *bufs = slurp(argv[8], &lbytes);
    init_modes(nfix, modeac, dc);
    run_bufs(fmt, iq, bufs, lbytes / 24, sublen, modeac);
    return write_records(argv[9]);
}

static int cmd_demod(int argc, char **argv) {
    if (argc != 8) return 2;
    input_format_t fmt = parse_fmt(argv[2]);
    int nfix = atoi(argv[3]), modeac = atoi(argv[4]), dc = atoi(argv[5]);
    size_t nbytes;
    uint8_t *iq = slurp(argv[6], &nbytes);
    init_modes(nfix, modeac, dc);
    capture_verbatim = Modes.net_verbatim = !strcmp(argv[1], "demodv");
    skip_commb = !strcmp(argv[1], "demodfull");
    capture_full = skip_commb || !strcmp(argv[1], "demodfullb");
    run_stream(fmt, iq, nbytes, modeac);
    FILE *f = fopen(argv[7], "wb");
    if (!f) { perror(argv[7]); return 2; }
    uint64_t hdr[8] = {nrecs,
        Modes.stats_current.demod_preambles,
        Modes.stats_current.demod_rejected_bad,
        Modes.stats_current.demod_rejected_unknown_icao,
        Modes.stats_current.demod_accepted[0],
        Modes.stats_current.demod_accepted[1],
        Modes.stats_current.demod_accepted[2],
        Modes.stats_current.demod_modeac};
    fwrite(hdr, sizeof (hdr), 1, f);
    fwrite(recs, sizeof (*recs), nrecs, f);
    if (capture_verbatim) fwrite(verb, sizeof (*verb), nrecs, f);
    if (capture_full) fwrite(full, sizeof (*full), nrecs, f);
    fclose(f);
    return 0;
}

/* the reference's decodeModesMessage on frames from a file */
static int cmd_decodemsgs(int argc, char **argv) {
    if (argc != 4) return 2;
    size_t nbytes;
    uint8_t *in = slurp(argv[2], &nbytes);
    uint32_t naddr, nframes;
    memcpy(&naddr, in, 4);
    memcpy(&nframes, in + 4, 4);
    if (nbytes != 8 + 4 * (size_t) naddr + 14 * (size_t) nframes) return 2;
    init_modes(1, 0, 0);
    skip_commb = 1;
    for (uint32_t k = 0; k < naddr; ++k) {
        uint32_t a;
        memcpy(&a, in + 8 + 4 * (size_t) k, 4);
        icaoFilterAdd(a);
    }
    const uint8_t *frames = in + 8 + 4 * (size_t) naddr;
    FILE *f = fopen(argv[3], "wb");
    if (!f) { perror(argv[3]); return 2; }
    for (uint32_t k = 0; k < nframes; ++k) {
        struct modesMessage mm;
        unsigned char frame[MODES_LONG_MSG_BYTES];
        memset(&mm, 0, sizeof (mm));
        memcpy(frame, frames + 14 * (size_t) k, 14);
        int32_t result = decodeModesMessage(&mm, frame);
        fwrite(&result, 4, 1, f);
        fwrite(&mm, sizeof (mm), 1, f);
    }
    uint32_t filtered = Modes.stats_current.cpr_filtered; /* mode_s.c:991-998 */
    fwrite(&filtered, 4, 1, f);
    fclose(f);
    return 0;
}

static int cmd_timedecode(int argc, char **argv) {
    if (argc != 4) return 2;
    size_t nbytes;
    uint8_t *in = slurp(argv[2], &nbytes);
    int reps = atoi(argv[3]);
    uint32_t naddr, nframes;
    memcpy(&naddr, in, 4);
    memcpy(&nframes, in + 4, 4);
    if (nbytes != 8 + 4 * (size_t) naddr + 14 * (size_t) nframes) return 2;
    init_modes(1, 0, 0);
    for (uint32_t k = 0; k < naddr; ++k) {
        uint32_t a;
        memcpy(&a, in + 8 + 4 * (size_t) k, 4);
        icaoFilterAdd(a);
    }
    const uint8_t *frames = in + 8 + 4 * (size_t) naddr;
    struct timespec t0, t1;
    uint64_t accepted = 0, sink = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int r = 0; r < reps; ++r)
        for (uint32_t k = 0; k < nframes; ++k) {
            struct modesMessage mm;
            unsigned char frame[MODES_LONG_MSG_BYTES];
            memset(&mm, 0, sizeof (mm));
            memcpy(frame, frames + 14 * (size_t) k, 14);
            if (decodeModesMessage(&mm, frame) == 0) {
                ++accepted;
                sink += mm.addr + mm.squawk + mm.cpr_lat; /* keep the work alive */
            }
        }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double s = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    printf("{\"seconds\": %.6f, \"messages\": %" PRIu64 ", \"accepted\": %" PRIu64 ", \"sink\": %" PRIu64 "}\n", s, (uint64_t) nframes * reps, accepted, sink);
    return 0;
}

static int cmd_stats(int argc, char **argv) {
    if (argc != 7) return 2;
    input_format_t fmt = parse_fmt(argv[2]);
    int nfix = atoi(argv[3]), modeac = atoi(argv[4]), dc = atoi(argv[5]);
    size_t nbytes;
    uint8_t *iq = slurp(argv[6], &nbytes);
    capture = 0;
    init_modes(nfix, modeac, dc);
    run_stream(fmt, iq, nbytes, modeac);
    const struct stats *st = &Modes.stats_current;
    printf("{\"demod_preambles\": %u, \"demod_rejected_bad\": %u, \"demod_rejected_unknown_icao\": %u, \"demod_accepted\": [%u, %u, %u], "
            "\"demod_modeac\": %u, \"strong_signal_count\": %u, \"noise_power_sum\": \"%a\", \"noise_power_count\": %" PRIu64 ", "
            "\"signal_power_sum\": \"%a\", \"signal_power_count\": %" PRIu64 ", \"peak_signal_power\": \"%a\"}\n",
            st->demod_preambles, st->demod_rejected_bad, st->demod_rejected_unknown_icao, st->demod_accepted[0], st->demod_accepted[1],
            st->demod_accepted[2], st->demod_modeac, st->strong_signal_count, st->noise_power_sum, st->noise_power_count,
            st->signal_power_sum, st->signal_power_count, st->peak_signal_power);
    return 0;
}

static int cmd_time(int argc, char **argv) {
    if (argc != 8) return 2;
    input_format_t fmt = parse_fmt(argv[2]);
    int nfix = atoi(argv[3]), modeac = atoi(argv[4]), dc = atoi(argv[5]);
    size_t nbytes;
    uint8_t *iq = slurp(argv[6], &nbytes);
    int reps = atoi(argv[7]);
    capture = 0;
    init_modes(nfix, modeac, dc);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int r = 0; r < reps; ++r)
        run_stream(fmt, iq, nbytes, modeac);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double s = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    unsigned bps = (fmt == INPUT_UC8) ? 2 : 4;
    printf("{\"seconds\": %.6f, \"samples\": %" PRIu64 ", \"accepted\": %u}\n", s,
            (uint64_t) (nbytes / bps) * reps,
            Modes.stats_current.demod_accepted[0] + Modes.stats_current.demod_accepted[1] + Modes.stats_current.demod_accepted[2]);
    return 0;
}

static int cmd_convert(int argc, char **argv) {
    if (argc != 7) return 2;
    input_format_t fmt = parse_fmt(argv[2]);
    int dc = atoi(argv[3]);
    unsigned chunk = atoi(argv[4]);
    size_t nbytes;
    uint8_t *iq = slurp(argv[5], &nbytes);
    unsigned bps = (fmt == INPUT_UC8) ? 2 : 4;
    size_t total = nbytes / bps;
    struct converter_state *st = NULL;
    iq_convert_fn conv = init_converter(fmt, 2400000.0, dc, &st);
    if (!conv) return 2;
    uint16_t *mag = malloc(total * 2 + 2);
    size_t ncalls = (total + chunk - 1) / chunk;
    double *means = calloc(ncalls * 2 + 2, sizeof (double));
    size_t c = 0;
    for (size_t pos = 0; pos < total; pos += chunk, ++c) {
        unsigned n = (total - pos < chunk) ? (unsigned) (total - pos) : chunk;
        conv(iq + pos * bps, mag + pos, n, st, &means[2 * c], &means[2 * c + 1]);
    }
    FILE *f = fopen(argv[6], "wb");
    uint64_t hdr[2] = {total, ncalls};
    fwrite(hdr, sizeof (hdr), 1, f);
    fwrite(mag, 2, total, f);
    fwrite(means, sizeof (double), ncalls * 2, f);
    fclose(f);
    return 0;
}

static int cmd_wire(int argc, char **argv) {
    if (argc != 8) return 2;
    input_format_t fmt = parse_fmt(argv[2]);
    int nfix = atoi(argv[3]), modeac = atoi(argv[4]), forward = atoi(argv[5]);
    size_t nbytes;
    uint8_t *iq = slurp(argv[6], &nbytes);
    init_modes(nfix, modeac, 0);
    Modes.net = 1;
    Modes.mlat = 1; /* raw output with @timestamps (net_io.c:868) */
    Modes.net_output_flush_size = 1 << 30;
    struct net_writer *w[2] = {&Modes.beast_out, &Modes.raw_out};
    for (int k = 0; k < 2; ++k) {
        wire_service[k].connections = 1;
        wire_service[k].writer = w[k];
        w[k]->service = &wire_service[k];
        w[k]->data = malloc(MODES_OUT_BUF_SIZE);
        w[k]->dataUsed = 0;
    }
    wire_mode = forward ? 2 : 1;
    run_stream(fmt, iq, nbytes, modeac);
    FILE *f = fopen(argv[7], "wb");
    if (!f) { perror(argv[7]); return 2; }
    uint64_t hdr[3] = {nrecs, wire_len[0], wire_len[1]};
    fwrite(hdr, sizeof (hdr), 1, f);
    fwrite(recs, sizeof (*recs), nrecs, f);
    fwrite(wire_bytes[0], 1, wire_len[0], f);
    fwrite(wire_bytes[1], 1, wire_len[1], f);
    fclose(f);
    return 0;
}

/* dump both syndrome tables by probing modesChecksumDiagnose over all 2^24 syndromes */
static int cmd_crctab(int argc, char **argv) {
    if (argc != 4) return 2;
    int nfix = atoi(argv[2]);
    modesChecksumInit(nfix);
    FILE *f = fopen(argv[3], "wb");
    for (int bits = 56; bits <= 112; bits += 56) {
        uint32_t count = 0;
        long at = ftell(f);
        fwrite(&count, 4, 1, f);
        for (uint32_t s = 1; s < (1u << 24); ++s) {
            struct errorinfo *ei = modesChecksumDiagnose(s, bits);
            if (ei) {
                int32_t row[4] = {(int32_t) s, ei->errors, ei->bit[0], ei->bit[1]};
                fwrite(row, sizeof (row), 1, f);
                ++count;
            }
        }
        long end = ftell(f);
        fseek(f, at, SEEK_SET);
        fwrite(&count, 4, 1, f);
        fseek(f, end, SEEK_SET);
    }
    /* known answers for modesChecksum on fixed vectors */
    uint8_t m[14];
    for (int i = 0; i < 14; ++i) m[i] = (uint8_t) (i * 37 + 11);
    uint32_t ka[2] = {modesChecksum(m, 56), modesChecksum(m, 112)};
    fwrite(ka, sizeof (ka), 1, f);
    fclose(f);
    return 0;
}

int main(int argc, char **argv) {
    int rc = 2;
    if (argc >= 2) {
        if (!strcmp(argv[1], "demod") || !strcmp(argv[1], "demodv") || !strcmp(argv[1], "demodfull") || !strcmp(argv[1], "demodfullb")) rc = cmd_demod(argc, argv);
        else if (!strcmp(argv[1], "decodemsgs")) rc = cmd_decodemsgs(argc, argv);
        else if (!strcmp(argv[1], "timedecode")) rc = cmd_timedecode(argc, argv);
        else if (!strcmp(argv[1], "demodbufs")) rc = cmd_demodbufs(argc, argv);
        else if (!strcmp(argv[1], "time")) rc = cmd_time(argc, argv);
        else if (!strcmp(argv[1], "stats")) rc = cmd_stats(argc, argv);
        else if (!strcmp(argv[1], "convert")) rc = cmd_convert(argc, argv);
        else if (!strcmp(argv[1], "crctab")) rc = cmd_crctab(argc, argv);
        else if (!strcmp(argv[1], "wire")) rc = cmd_wire(argc, argv);
    }
    if (rc == 2)
        fprintf(stderr, "usage: see header of oracle/ref_harness.c\n");
    return rc;
}
