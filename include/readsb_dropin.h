/*
 * readsb_dropin.h -- the reference's own entry points for the receive path, one for one, on top of
 * libreadsb_amd.so.  Where readsb_amd.h is the batched fast path (whole recordings / many blocks
 * per call, data resident on the GPU), these are the calls readsb itself makes per 131072-sample
 * block, with the reference's argument meaning and error behaviour, so that a maintainer can
 * bind them where the C functions are called today (INTEGRATION.md shows the shim).
 *
 *   reference (Mictronics/readsb v3.9.0)                  here
 *   ---------------------------------------------------   --------------------------------------
 *   init_converter            convert.h:41-44, convert.c:446-491   rsbd_init_converter
 *   iq_convert_fn             convert.h:34-39, convert.c:63-423    rsbd_iq_convert_fn (returned)
 *   cleanup_converter         convert.h:46,    convert.c:493-499   rsbd_cleanup_converter
 *   -DSC16Q11_TABLE_BITS=n    convert.c:264-328, 451-456           rsbd_set_sc16q11_table_bits
 *   modesChecksumInit         crc.h:40,        crc.c:353-378       rsbd_modesChecksumInit
 *   modesChecksum             crc.h:41,        crc.c:67-82         rsbd_modesChecksum
 *   modesChecksumDiagnose     crc.h:42,        crc.c:383-406       rsbd_modesChecksumDiagnose
 *   modesChecksumFix          crc.h:43,        crc.c:410-418       rsbd_modesChecksumFix
 *   crcCleanupTables          crc.h:44,        crc.c:424-430       rsbd_crcCleanupTables
 *   demodulate2400            demod_2400.h:31, demod_2400.c:72-383 rsbd_demodulate2400
 *   demodulate2400AC          demod_2400.h:32, demod_2400.c:476-662 rsbd_demodulate2400AC
 *   icaoFilterInit/Add/Expire icao_filter.h,   icao_filter.c       rsbd_icaoFilter*
 *   ifileOpen / ifileRun / ifileClose  sdr_ifile.c:115-272 (sdr.c:94)  rsbd_ifileOpen / rsbd_ifileRun / rsbd_ifileClose
 *
 * The reference keeps the demodulator's state in globals (Modes.*, the ICAO filter tables, the CRC
 * tables); a library cannot, so the two demodulator calls and the filter calls take a handle that
 * stands for "this process's Modes" (rsbd_modesInit).  The decoded message is handed to a callback
 * that stands for useModesMessage(&mm) (demod_2400.c:374 / :657); it receives the fields of struct
 * modesMessage that the path itself fills (rsb_message, readsb_amd.h).
 *
 * Error behaviour as in the reference: init_converter prints "no suitable converter for format=%d
 * dc=%d" and returns NULL (convert.c:460-464); the data-path calls return nothing.  Additionally, a
 * missing or failing HIP device makes the init calls return NULL with a message on stderr; there is
 * no CPU fallback.
 */
#ifndef READSB_DROPIN_H
#define READSB_DROPIN_H

#include <stdint.h>

#include "readsb_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- convert.h ------------------------------------------------------------------------------ */

struct rsbd_converter_state; /* struct converter_state */

/* input_format_t: RSB_UC8 / RSB_SC16 / RSB_SC16Q11 (same values as INPUT_UC8 / INPUT_SC16 / INPUT_SC16Q11) */
typedef void (*rsbd_iq_convert_fn)(void *iq_data, uint16_t *mag_data, unsigned nsamples, struct rsbd_converter_state *state,
                                   double *out_mean_level, double *out_mean_power);

/* filter_dc != 0 selects the DC-block converters (convert.c:113-162, :164-213, :374-423) as a chunked
 * linear-recurrence scan: magnitudes within +-1 LSB of the sequential float32 IIR, state carried across
 * calls (DESIGN.md "DC filter"); its coefficients are those of sample_rate = 2400000 (convert.c:480-488), any other rate
 * with filter_dc is refused like an unknown format.  The function returned takes
 * HOST buffers, as the reference's do (it stages through device memory owned by the state).
 * out_mean_level / out_mean_power are the reference's bit for bit, also for the SC16 formats, whose
 * float32 running sums (convert.c:226-262) are taken in sample order on the device. */
rsbd_iq_convert_fn rsbd_init_converter(int format, double sample_rate, int filter_dc, struct rsbd_converter_state **out_state);
void rsbd_cleanup_converter(struct rsbd_converter_state *state);

/* The reference's build option -DSC16Q11_TABLE_BITS=bits (convert.c:264-328, 451-456) as a run-time, process-wide setting: with
 * bits in 1..11, every rsbd_init_converter(INPUT_SC16Q11, ..., filter_dc = 0) and rsbd_ifileOpen of that format made AFTERWARDS
 * converts as convert_sc16q11_table does (bit for bit, integer block means), as a shim for a readsb built with that macro must.
 * 0 (the default) = convert_sc16q11_nodc.  Returns 0, or -1 for a value outside 0..11. */
int rsbd_set_sc16q11_table_bits(int bits);

/* ---- crc.h (host tables; the kernels carry their own copies) ------------------------------------ */

struct rsbd_errorinfo { /* struct errorinfo, crc.h:32-38 */
    uint32_t syndrome;
    int errors;
    int8_t bit[2];
};

void rsbd_modesChecksumInit(int fixBits);
uint32_t rsbd_modesChecksum(uint8_t *msg, int bitlen);
struct rsbd_errorinfo *rsbd_modesChecksumDiagnose(uint32_t syndrome, int bitlen);
void rsbd_modesChecksumFix(uint8_t *msg, struct rsbd_errorinfo *info);
void rsbd_crcCleanupTables(void);

/* ---- demod_2400.h ----------------------------------------------------------------------------- */

struct rsbd_mag_buf { /* struct mag_buf, readsb.h:298-311 (non-ARM layout) */
    uint64_t sampleTimestamp;
    double mean_level;
    double mean_power;
    uint32_t dropped;
    unsigned length;
    uint64_t sysTimestamp;
    uint16_t *data; /* HOST: trailing_samples (326) of overlap, then `length` samples */
};

typedef struct rsbd_modes rsbd_modes; /* what the reference keeps in `Modes` + icao_filter.c + crc.c statics */
typedef void (*rsbd_use_message_fn)(void *user, const rsb_message *mm); /* useModesMessage(&mm) */

/* modesInit's share of the path (readsb.c:187-251): nfix_crc, the CRC tables, an empty ICAO filter.
 * max_length = the largest mag_buf.length that will be passed (MODES_MAG_BUF_SAMPLES = 131072). */
rsbd_modes *rsbd_modesInit(int nfix_crc, int device, unsigned max_length);
void rsbd_modesCleanup(rsbd_modes *m);

void rsbd_icaoFilterInit(rsbd_modes *m);
void rsbd_icaoFilterAdd(rsbd_modes *m, uint32_t addr);
void rsbd_icaoFilterExpire(rsbd_modes *m); /* one flip; the reference's 60 s timer (icao_filter.c:147-161) is the caller's */

/* One block, exactly as the reference scans it: every offset j in [0, mag->length), messages to
 * `use` in the reference's order with timestampMsg = mag->sampleTimestamp + j*5 + (8+56)*12 + phase.
 * demodulate2400AC reads mag->mean_level / mag->mean_power as the reference does (demod_2400.c:484-485). */
void rsbd_demodulate2400(rsbd_modes *m, struct rsbd_mag_buf *mag, rsbd_use_message_fn use, void *user);
void rsbd_demodulate2400AC(rsbd_modes *m, struct rsbd_mag_buf *mag, rsbd_use_message_fn use, void *user);

/* The same two calls with the field decode done on the device as well: `use` also receives the message's
 * struct modesMessage image (rsb_decoded, include/readsb_amd.h) as decodeModesMessage (mode_s.c:557-716, 736-1474) /
 * decodeModeAMessage (mode_ac.c:165-199) leave it, timestampMsg on the block's clock; what stays with the caller is listed at
 * rsb_decoded (sysTimestampMsg, verbatim[], decodeCommB, the filter addition of mode_s.c:717-726 in ITS filter). */
typedef void (*rsbd_use_decoded_fn)(void *user, const rsb_message *mm, const rsb_decoded *fields);
void rsbd_demodulate2400_decoded(rsbd_modes *m, struct rsbd_mag_buf *mag, rsbd_use_decoded_fn use, void *user);
void rsbd_demodulate2400AC_decoded(rsbd_modes *m, struct rsbd_mag_buf *mag, rsbd_use_decoded_fn use, void *user);

/* ---- sdr_ifile.c: the "ifile" input driver -------------------------------------------------------- */

/* What --device-type ifile --ifile <filename|-> --iformat <fmt> [--dcfilter] does on the reader thread
 * (ifileOpen sdr_ifile.c:115-160, ifileRun :162-254) together with the main thread's demodulator loop
 * (readsb.c:847-911), as one call: the file is read in batches of blocks_per_batch x 131072 samples (each read
 * repeated until the batch is full or EOF, :209-223), every batch goes through the converter and both
 * demodulators on the device with the 326-sample overlap, the DC state and the ICAO filter carried across, and
 * the messages are handed to `use` in the reference's order.  --throttle (:230-238) has no counterpart.  */
typedef struct rsbd_ifile rsbd_ifile;

/* NULL + a message on stderr if the file cannot be opened (:119-127: "ifile: could not open %s: %s"), the format
 * is unknown, or no HIP device is usable.  filename "-" = standard input. */
rsbd_ifile *rsbd_ifileOpen(const char *filename, int input_format, int filter_dc, int nfix_crc, int mode_ac, int device, unsigned blocks_per_batch);
/* Runs to EOF; returns the number of samples consumed (a trailing partial sample is dropped, as :225 does), or -1
 * after a read or device error (message on stderr).  rsbd_ifileFilterAdd before the run = --show-only (readsb.c:249-250). */
long long rsbd_ifileRun(rsbd_ifile *f, rsbd_use_message_fn use, void *user);
void rsbd_ifileFilterAdd(rsbd_ifile *f, uint32_t addr);
void rsbd_ifileClose(rsbd_ifile *f);

#ifdef __cplusplus
}
#endif
#endif /* READSB_DROPIN_H */
