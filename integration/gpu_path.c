/*
 * gpu_path.c -- the shim that makes libreadsb_amd.so a LINK-LEVEL drop-in for readsb's receive path.
 *
 * Compiled with the reference's own headers and linked INSTEAD OF demod_2400.o and convert.o (plus -lreadsb_amd), it
 * exports the reference's exact symbols, so an unmodified readsb.c:883-886 / sdr_ifile.c:149,228 link against it:
 *
 *   void demodulate2400(struct mag_buf *mag);      demod_2400.h:31, demod_2400.c:72-383
 *   void demodulate2400AC(struct mag_buf *mag);    demod_2400.h:32, demod_2400.c:476-662
 *   iq_convert_fn init_converter(input_format_t, double, int, struct converter_state **);   convert.h:41-44
 *   void cleanup_converter(struct converter_state *);                                       convert.h:46
 *
 * and hands a REAL struct modesMessage (readsb.h:421-635) to useModesMessage (mode_s.h:37).  How the struct is filled:
 * the GPU path decides which (offset, phase) the reference's loop would accept and returns the repaired frame, the
 * syndrome and the number of corrected bits; the shim flips the corrected bits back (crc.c:410-418 applied twice is the
 * identity), which gives the frame exactly as the reference's slicer had it, and passes THAT to the reference's own
 * decodeModesMessage (mode_s.c:424-734): it repeats the CRC repair, keeps `verbatim` (mode_s.c:427-430), runs every
 * field decode that stays on the host, and teaches readsb's own ICAO filter the addresses it would have learned
 * (mode_s.c:717-726).  Both filters -- the library's, which the device replay uses, and readsb's global one -- therefore
 * see the same sequence of additions; the two hooks at the end keep them equal for --show-only and the 60 s expiry.
 *
 * With RSB_GPU_FIELDS=1 in the environment the field decode comes from the device as well: the library's
 * rsb_decoded IS struct modesMessage (the asserts below), so the shim copies it over `mm`, adds what the record leaves to the
 * caller -- sysTimestampMsg, verbatim[], decodeCommB for DF20 / DF21, the filter addition, the cpr_filtered counter -- and never
 * calls decodeModesMessage.
 *
 * What stays in readsb: everything else (mode_s.c, crc.c, icao_filter.c, mode_ac.c, track.c, net_io.c ... unchanged).
 * Per-block cost of this compatibility path: one host->device copy of the magnitudes and about 0.3 ms of launches and
 * waits per call; the batched rsb_* entry points are the fast path.
 */
#include "readsb.h"

#include <stddef.h>

#include "readsb_dropin.h"

/* the layouts the library assumes are the reference's */
_Static_assert(sizeof(struct mag_buf) == sizeof(struct rsbd_mag_buf), "struct mag_buf");
_Static_assert(offsetof(struct mag_buf, sampleTimestamp) == offsetof(struct rsbd_mag_buf, sampleTimestamp), "mag_buf.sampleTimestamp");
_Static_assert(offsetof(struct mag_buf, mean_level) == offsetof(struct rsbd_mag_buf, mean_level), "mag_buf.mean_level");
_Static_assert(offsetof(struct mag_buf, mean_power) == offsetof(struct rsbd_mag_buf, mean_power), "mag_buf.mean_power");
_Static_assert(offsetof(struct mag_buf, dropped) == offsetof(struct rsbd_mag_buf, dropped), "mag_buf.dropped");
_Static_assert(offsetof(struct mag_buf, length) == offsetof(struct rsbd_mag_buf, length), "mag_buf.length");
_Static_assert(offsetof(struct mag_buf, sysTimestamp) == offsetof(struct rsbd_mag_buf, sysTimestamp), "mag_buf.sysTimestamp");
_Static_assert(offsetof(struct mag_buf, data) == offsetof(struct rsbd_mag_buf, data), "mag_buf.data");
_Static_assert(sizeof(struct errorinfo) == sizeof(struct rsbd_errorinfo) && offsetof(struct errorinfo, bit) == offsetof(struct rsbd_errorinfo, bit),
               "struct errorinfo");
_Static_assert(MODES_MAG_BUF_SAMPLES == RSB_BLOCK_SAMPLES, "block size");
_Static_assert((int) INPUT_UC8 == RSB_UC8 && (int) INPUT_SC16 == RSB_SC16 && (int) INPUT_SC16Q11 == RSB_SC16Q11, "input_format_t");
_Static_assert(sizeof(((struct modesMessage *) 0)->msg) == sizeof(((rsb_message *) 0)->msg), "message bytes");

/* rsb_decoded is struct modesMessage, member by member (bit-fields: by the byte or word they start in) */
_Static_assert(sizeof(struct modesMessage) == sizeof(rsb_decoded) && sizeof(rsb_decoded) == RSB_DECODED_BYTES, "struct modesMessage");
#define SAME(member, ours) _Static_assert(offsetof(struct modesMessage, member) == offsetof(rsb_decoded, ours), #member)
SAME(timestampMsg, timestampMsg); SAME(sysTimestampMsg, sysTimestampMsg); SAME(msg, msg); SAME(verbatim, verbatim); SAME(msgbits, msgbits);
SAME(msgtype, msgtype); SAME(crc, crc); SAME(correctedbits, correctedbits); SAME(addr, addr); SAME(addrtype, addrtype); SAME(remote, remote);
SAME(score, score); SAME(sbs_in, sbs_in); SAME(reduce_forward, reduce_forward); SAME(source, source); SAME(signalLevel, signalLevel);
SAME(IID, IID); SAME(AA, AA); SAME(AC, AC); SAME(CA, CA); SAME(CC, CC); SAME(CF, CF); SAME(DR, DR); SAME(FS, FS); SAME(ID, ID); SAME(KE, KE);
SAME(ND, ND); SAME(RI, RI); SAME(SL, SL); SAME(UM, UM); SAME(VS, VS); SAME(metype, metype); SAME(mesub, mesub); SAME(MB, MB); SAME(MD, MD);
SAME(ME, ME); SAME(MV, MV); SAME(altitude_baro, altitude_baro); SAME(altitude_baro_unit, altitude_baro_unit); SAME(altitude_geom, altitude_geom);
SAME(altitude_geom_unit, altitude_geom_unit); SAME(geom_delta, geom_delta); SAME(heading, heading); SAME(heading_type, heading_type);
SAME(track_rate, track_rate); SAME(roll, roll); SAME(gs.v0, gs_v0); SAME(gs.v2, gs_v2); SAME(gs.selected, gs_selected); SAME(ias, ias); SAME(tas, tas);
SAME(mach, mach); SAME(baro_rate, baro_rate); SAME(geom_rate, geom_rate); SAME(squawk, squawk); SAME(callsign, callsign); SAME(category, category);
SAME(emergency, emergency); SAME(cpr_type, cpr_type); SAME(cpr_lat, cpr_lat); SAME(cpr_lon, cpr_lon); SAME(cpr_nucp, cpr_nucp); SAME(airground, airground);
SAME(decoded_lat, decoded_lat); SAME(decoded_lon, decoded_lon); SAME(decoded_nic, decoded_nic); SAME(decoded_rc, decoded_rc); SAME(commb_format, commb_format);
SAME(accuracy.sil_type, accuracy_sil_type); SAME(opstatus.sil_type, opstatus_sil_type); SAME(opstatus.tah, opstatus_tah); SAME(opstatus.hrd, opstatus_hrd);
SAME(opstatus.track_angle, opstatus_track_angle); SAME(opstatus.cc_lw, opstatus_cc_lw); SAME(opstatus.cc_antenna_offset, opstatus_cc_antenna_offset);
SAME(nav.fms_altitude, nav_fms_altitude); SAME(nav.mcp_altitude, nav_mcp_altitude); SAME(nav.qnh, nav_qnh); SAME(nav.heading, nav_heading);
SAME(nav.heading_type, nav_heading_type); SAME(nav.altitude_source, nav_altitude_source); SAME(nav.modes, nav_modes);
_Static_assert(offsetof(rsb_decoded, valid_bits) == offsetof(struct modesMessage, MV) + 7, "the one-bit members start in the byte behind MV[]");
_Static_assert(offsetof(rsb_decoded, accuracy_bits) + 4 == offsetof(struct modesMessage, accuracy.sil_type), "accuracy bit-fields");
_Static_assert(offsetof(rsb_decoded, opstatus_bits) == offsetof(struct modesMessage, opstatus.cc_antenna_offset) + 4, "opstatus bit-fields");
_Static_assert(offsetof(rsb_decoded, nav_bits) == offsetof(struct modesMessage, nav.heading) + 4, "nav bit-fields");

static rsbd_modes *G; /* the demodulator's share of `Modes` that lives in the library */
static double block_signal_power; /* of the block being demodulated: for the noise power statistic */

static rsbd_modes *handle(void) {
    if (!G) {
        const char *dev = getenv("RSB_DEVICE");
        G = rsbd_modesInit(Modes.nfix_crc, dev ? atoi(dev) : 0, MODES_MAG_BUF_SAMPLES);
        if (!G) {
            fprintf(stderr, "gpu_path: no usable HIP device; the GPU receive path has no CPU fallback\n");
            exit(1);
        }
    }
    return G;
}

/* rsb_message -> struct modesMessage -> useModesMessage, as demod_2400.c:313-374 / :645-660 do */
static void deliver(void *user, const rsb_message *r) {
    const struct mag_buf *mag = (const struct mag_buf *) user;
    struct modesMessage mm;
    memset(&mm, 0, sizeof (mm));
    mm.timestampMsg = r->timestamp; /* demod_2400.c:320 / :649 */
    mm.sysTimestampMsg = mag->sysTimestamp + receiveclock_ms_elapsed(mag->sampleTimestamp, mm.timestampMsg); /* :323 / :652 */

    if (r->msgtype == 32) { /* Mode A/C */
        decodeModeAMessage(&mm, (r->msg[0] << 8) | r->msg[1]); /* :654 */
        useModesMessage(&mm);
        Modes.stats_current.demod_modeac++; /* :660 */
        return;
    }

    mm.score = r->score; /* :325 */
    unsigned char frame[MODES_LONG_MSG_BYTES];
    memcpy(frame, r->msg, MODES_LONG_MSG_BYTES);
    if (r->correctedbits) {
        /* the frame as sliced = the repaired frame with the repaired bits flipped back (DF11 repairs on the syndrome
         * without its interrogator-id bits, mode_s.c:475-489) */
        struct errorinfo *ei = modesChecksumDiagnose(r->msgtype == 11 ? (r->crc & 0xffff80) : r->crc, r->msgbits);
        modesChecksumFix(frame, ei);
    }
    const int result = decodeModesMessage(&mm, frame); /* :329 -- the reference's own, unchanged */
    if (result < 0) { /* cannot happen while both filters are kept equal (hooks below) */
        if (result == -1)
            Modes.stats_current.demod_rejected_unknown_icao++;
        else
            Modes.stats_current.demod_rejected_bad++;
        return;
    }
    Modes.stats_current.demod_accepted[mm.correctedbits]++; /* :337 */

    /* :342-362; the sum over the samples was taken on the device, signalLevel is the reference's bit for bit */
    const int signal_len = modesMessageLenByType(mm.msgtype) * 12 / 5;
    mm.signalLevel = r->signal_level;
    Modes.stats_current.signal_power_sum += mm.signalLevel * signal_len;
    Modes.stats_current.signal_power_count += signal_len;
    block_signal_power += mm.signalLevel * signal_len;
    if (mm.signalLevel > Modes.stats_current.peak_signal_power)
        Modes.stats_current.peak_signal_power = mm.signalLevel;
    if (mm.signalLevel > 0.50119)
        Modes.stats_current.strong_signal_count++;

    useModesMessage(&mm); /* :374 */
}

/* the same with the struct filled on the device (RSB_GPU_FIELDS=1) */
static void deliver_decoded(void *user, const rsb_message *r, const rsb_decoded *fields) {
    const struct mag_buf *mag = (const struct mag_buf *) user;
    struct modesMessage mm;
    memcpy(&mm, fields, sizeof (mm));
    mm.sysTimestampMsg = mag->sysTimestamp + receiveclock_ms_elapsed(mag->sampleTimestamp, mm.timestampMsg); /* demod_2400.c:323 / :652 */
    if (r->msgtype == 32) {
        useModesMessage(&mm);
        Modes.stats_current.demod_modeac++; /* :660 */
        return;
    }
    memcpy(mm.msg, r->msg, MODES_LONG_MSG_BYTES);
    if (Modes.net_verbatim) { /* mode_s.c:427-430: the frame before the repair = the repaired bits flipped back */
        memcpy(mm.verbatim, r->msg, MODES_LONG_MSG_BYTES);
        if (r->correctedbits)
            modesChecksumFix(mm.verbatim, modesChecksumDiagnose(r->msgtype == 11 ? (r->crc & 0xffff80) : r->crc, r->msgbits));
    }
    if (mm.msgtype == 20 || mm.msgtype == 21)
        decodeCommB(&mm); /* mode_s.c:669 (comm_b.c stays on the host) */
    if (mm.msgtype == 17 || mm.msgtype == 18) { /* mode_s.c:991-998: the counter behind a position the decode refuses */
        const unsigned cf_decodes = mm.msgtype == 17 || mm.CF == 0 || mm.CF == 1 || mm.CF == 2 || mm.CF == 5 || mm.CF == 6;
        const unsigned ac12 = ((mm.ME[1] << 4) | (mm.ME[2] >> 4)) & 0xfff;
        if (cf_decodes && mm.metype == 15 && ac12 == 0 && mm.cpr_lon == 0 && (mm.cpr_lat & 0x0fff) == 0)
            Modes.stats_current.cpr_filtered++;
    }
    if (!mm.correctedbits && (mm.msgtype == 17 || (mm.msgtype == 11 && mm.IID == 0)))
        icaoFilterAdd(mm.addr); /* mode_s.c:717-726: readsb's own filter learns what the library's has learned */
    Modes.stats_current.demod_accepted[mm.correctedbits]++; /* :337 */
    const int signal_len = modesMessageLenByType(mm.msgtype) * 12 / 5;
    Modes.stats_current.signal_power_sum += mm.signalLevel * signal_len;
    Modes.stats_current.signal_power_count += signal_len;
    block_signal_power += mm.signalLevel * signal_len;
    if (mm.signalLevel > Modes.stats_current.peak_signal_power)
        Modes.stats_current.peak_signal_power = mm.signalLevel;
    if (mm.signalLevel > 0.50119)
        Modes.stats_current.strong_signal_count++;
    useModesMessage(&mm); /* :374 */
}

static int gpu_fields(void) {
    static int on = -1;
    if (on < 0) {
        const char *e = getenv("RSB_GPU_FIELDS");
        on = e && atoi(e) != 0;
    }
    return on;
}

void demodulate2400(struct mag_buf *mag) {
    block_signal_power = 0;
    if (gpu_fields())
        rsbd_demodulate2400_decoded(handle(), (struct rsbd_mag_buf *) mag, deliver_decoded, mag);
    else
        rsbd_demodulate2400(handle(), (struct rsbd_mag_buf *) mag, deliver, mag);
    /* :378-382 (exact counters and sums: rsb_get_stats on an RSB_FLAG_EXACT_STATS context) */
    Modes.stats_current.noise_power_sum += (mag->mean_power * mag->length - block_signal_power);
    Modes.stats_current.noise_power_count += mag->length;
}

void demodulate2400AC(struct mag_buf *mag) {
    if (gpu_fields())
        rsbd_demodulate2400AC_decoded(handle(), (struct rsbd_mag_buf *) mag, deliver_decoded, mag);
    else
        rsbd_demodulate2400AC(handle(), (struct rsbd_mag_buf *) mag, deliver, mag);
}

iq_convert_fn init_converter(input_format_t format, double sample_rate, int filter_dc, struct converter_state **out_state) {
    return (iq_convert_fn) rsbd_init_converter((int) format, sample_rate, filter_dc, (struct rsbd_converter_state **) out_state);
}

void cleanup_converter(struct converter_state *state) {
    rsbd_cleanup_converter((struct rsbd_converter_state *) state);
}

/* The two places where readsb changes its ICAO filter from OUTSIDE the receive path; a maintainer adds one call each:
 *   readsb.c:249-250   icaoFilterAdd(Modes.show_only);   ->  + gpuPathFilterAdd(Modes.show_only);
 *   readsb.c:337       icaoFilterExpire();               ->  + gpuPathFilterExpire();   (only when that call flips: icao_filter.c:150-160) */
void gpuPathFilterAdd(uint32_t addr) {
    rsbd_icaoFilterAdd(handle(), addr);
}

void gpuPathFilterExpire(void) {
    rsbd_icaoFilterExpire(handle());
}

void gpuPathCleanup(void) {
    rsbd_modesCleanup(G);
    G = NULL;
}
