/*
 * readsb_amd.h -- C ABI of libreadsb_amd.so: an MI355X (gfx950) implementation of readsb's
 * receive-side hot path (IQ -> magnitude -> Mode-S / Mode A/C demodulation -> CRC repair).
 *
 * Plain C, plain pointers and sizes; HIP stays behind this boundary.  Two groups:
 *
 *   1. The batched fast path (rsb_*): device- or host-resident IQ for a whole recording or a
 *      stream of 131072-sample blocks in, the reference's message list out.  This is our own
 *      entry (the reference has no batched call); what it replaces is the per-block loop
 *      readsb.c:847-911 -> sdr_ifile.c:184-228 -> convert.c -> demod_2400.c:72-383 / :476-662.
 *
 *   2. Reference-signature drop-ins (include/readsb_dropin.h): init_converter / iq_convert_fn /
 *      cleanup_converter, demodulate2400(struct mag_buf*), demodulate2400AC(struct mag_buf*),
 *      modesChecksum*.
 *
 * All citations are file:line in Mictronics/readsb v3.9.0.
 *
 * Error behaviour mirrors the reference's (SURVEY.md section 8b "Errors"): no error codes on
 * the data path of the drop-ins (NULL / false + a message on stderr at init); the rsb_* calls
 * return 0 on success and a negative RSB_E* code otherwise, with rsb_last_error() for text.
 * The library never falls back to a CPU implementation: without a usable HIP device every
 * compute call fails with RSB_ENODEV.
 */
#ifndef READSB_AMD_H
#define READSB_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSB_BLOCK_SAMPLES 131072u /* MODES_MAG_BUF_SAMPLES, readsb.h:96-97 */
#define RSB_TRAILING_SAMPLES 326u /* Modes.trailing_samples, readsb.c:196 */

/* input_format_t, convert.h:29-32 */
enum { RSB_UC8 = 0, RSB_SC16 = 1, RSB_SC16Q11 = 2 };

enum {
    RSB_OK = 0,
    RSB_EINVAL = -1,   /* bad argument / unsupported combination */
    RSB_ENODEV = -2,   /* no HIP device, or a HIP call failed */
    RSB_ENOMEM = -3,
    RSB_EOVERFLOW = -4,/* a device queue overflowed (raise rsb_config.queue_div), or the prune's bounded wait ran out (rsb_last_error says which); no partial results */
    RSB_ESTATE = -5    /* call sequence error (collect without submit, unaligned stream position...) */
};

/* What --device-type ifile --iformat X [--dcfilter] [--fix|--no-fix|--aggressive] [--modeac] select
 * (sdr_ifile.c:82-106, readsb.c:485-509, 538-540). */
typedef struct rsb_config {
    int32_t input_format;      /* RSB_UC8 / RSB_SC16 / RSB_SC16Q11 */
    int32_t dc_filter;         /* --dcfilter (readsb.c:485-487): the DC-block IIR of convert.c:137-140 as a chunked scan; results
                                  follow the reference within a tolerance, not bit for bit (DESIGN.md sections 5c, 7).  Without mode_ac
                                  and RSB_FLAG_EXACT_STATS the filter runs inside the scan / candidate / record kernels (no magnitude
                                  buffer, two batches in flight); with them a converter writes the batch's magnitudes first and one
                                  batch is in flight.  rsb_seek is refused either way (the filter's state does not travel) */
    int32_t fix_bits;          /* Modes.nfix_crc: 0 (--no-fix), 1 (--fix, default), 2 (--aggressive) */
    int32_t mode_ac;           /* --modeac */
    int32_t device;            /* HIP device ordinal */
    uint32_t queue_div;        /* device queues hold max_batch_samples/queue_div candidates (0 -> 32) */
    uint64_t max_batch_samples;/* largest nsamples a single rsb_submit will be given */
    uint32_t flags;            /* RSB_FLAG_* */
    uint32_t scan_workgroups;  /* workgroups (= CUs) the scan kernel's persistent grid may take; 0 = one per CU of the device.
                                  Fewer leave CUs to the replay kernels of the previous batch (two batches in flight), and a
                                  very small number makes every wavefront walk many strips of a small input (tests) */
    uint32_t block_samples;    /* samples per demodulator call, i.e. mag_buf.length of a full buffer: 0 = 131072 (MODES_MAG_BUF_SAMPLES,
                                  what sdr_ifile.c / sdr_rtlsdr.c / sdr_plutosdr.c build).  A bladeRF buffer holds 524288 / block_size
                                  sub-blocks of (block_size - 16) / 4 samples (sdr_bladerf.c:335-339): 129024 or 130048.  The greedy
                                  skip, the Mode A/C scan and its noise level start fresh per buffer (demod_2400.c:88,371,484-487), so
                                  the buffer length is part of the result.  A multiple of 8 in 4096..131072; batches are whole
                                  numbers of such blocks; not with dc_filter or the SC16Q11 table converter */
    uint32_t mean_sub_samples; /* 0, or: the converter is called once per this many samples and a buffer's mean level / power are
                                  the averages of the calls' means (sdr_bladerf.c:385-390, 403-404).  A divisor of block_samples;
                                  SC16 / SC16Q11 only */
} rsb_config;

/* Run the sequential half (best phase, decode, greedy skip, filter updates) on the host from the
 * candidate records instead of on the device.  Same results; kept as the cross-check and as the
 * path a batch falls back to when the ICAO filter is close to full (icao_filter.c:76-79). */
#define RSB_FLAG_HOST_REPLAY 1u
/* Exact demodulator statistics (struct stats, stats.h:61-79, as demod_2400.c:175,304-310,329-337,355-362,378-382
 * updates them): every offset that passes the preamble gate is carried to the host replay, which then counts
 * demod_preambles / demod_rejected_* and sums signal and noise power exactly as the reference's loop does.  Same
 * message list; much slower (about 1 % of all offsets cross PCIe as candidate records): a statistics mode, not the
 * fast path.  Implies RSB_FLAG_HOST_REPLAY. */
#define RSB_FLAG_EXACT_STATS 2u

/* rsb_submit runs the scan only; the filter-superset prune waits for rsb_collect, so that the caller can widen the superset
 * first (rsb_superset_add).  For decoding ONE recording in shards on several GPUs; see rsb_seek. */
#define RSB_FLAG_DEFER_PRUNE 4u

/* SC16Q11 input as a reference BUILT with -DSC16Q11_TABLE_BITS=bits converts it (convert_sc16q11_table, convert.c:264-328, chosen
 * by convert.c:451-456 in place of convert_sc16q11_nodc): |I| and |Q| masked to 11 bits, the low 11 - bits of each dropped, the
 * magnitude looked up in a 2^(2 bits)-entry table.  The device computes each entry's formula (convert.c:283-290) instead of
 * holding the table, so any 1..11 bits cost the same.  Needs input_format RSB_SC16Q11 and no dc_filter (as the reference's
 * converter list); the block means are the integer ones of convert.c:321-327.  The context holds the batch's magnitudes,
 * so one batch is in flight at a time and rsb_seek is not available, as with dc_filter.  0 bits = the float converter. */
#define RSB_FLAG_SC16Q11_TABLE(bits) (((uint32_t)(bits) & 15u) << 8)
#define RSB_FLAG_SC16Q11_TABLE_BITS_OF(flags) (((uint32_t)(flags) >> 8) & 15u)

/* One decoded message: the fields of struct modesMessage (readsb.h:421-439) that the hot path
 * itself produces.  Packed, 56 bytes. */
#pragma pack(push, 1)
typedef struct rsb_message {
    uint64_t timestamp;     /* timestampMsg, 12 MHz: demod_2400.c:320 / :649 */
    uint8_t msg[14];        /* corrected message, first msgbits/8 bytes (rest zero) */
    uint8_t msgbits;        /* 56, 112, or 16 for Mode A/C (mode_ac.c:171) */
    uint8_t msgtype;        /* DF, or 32 for Mode A/C (mode_ac.c:168) */
    uint32_t crc;           /* mm->crc: syndrome before repair (mode_s.c:440) */
    int32_t correctedbits;  /* mode_s.c:489,518 */
    uint32_t addr;          /* as left by mode_s.c:424-562 / mode_ac.c:176 (before ES field decode) */
    int32_t score;          /* scoreModesMessage result of the winning phase (demod_2400.c:325) */
    int32_t source;         /* datasource_t as left by mode_s.c:464-543 / mode_ac.c:166 */
    uint32_t iid;           /* DF11 interrogator id (mode_s.c:475) */
    double signal_level;    /* demod_2400.c:342-354 */
} rsb_message;
#pragma pack(pop)

/* Counters.  demod_accepted[] / demod_modeac are always exact.  demod_preambles, demod_rejected_* and the
 * power members below (stats.h:62-79) are the reference's only on a context created with RSB_FLAG_EXACT_STATS
 * (they stay 0 otherwise: the fast path never sees the offsets the greedy skip hides).  One difference is kept on
 * purpose: for a recording whose length is a multiple of 131072 the reference's reader publishes a final
 * zero-length block (sdr_ifile.c:213-225) whose mean_power is 0/0, which turns its noise_power_sum into NaN; this
 * library scans no such block. */
typedef struct rsb_stats {
    uint64_t samples_processed;
    uint64_t demod_preambles, demod_rejected_bad, demod_rejected_unknown_icao;
    uint64_t demod_accepted[3];
    uint64_t demod_modeac;
    uint64_t gated_offsets;     /* offsets passing the preamble gate (before the greedy skip) */
    uint64_t live_candidates;   /* ... with at least one phase that is not "-2 under any filter" */
    uint64_t survivors;         /* ... left after the filter-superset prune (replayed on the host) */
    uint64_t modeac_candidates;
    /* RSB_FLAG_EXACT_STATS only (demod_2400.c:353-362, 378-382) */
    double signal_power_sum;        /* sum over accepted messages of sum(mag^2) / 65535^2 */
    uint64_t signal_power_count;    /* ... and of the samples summed (134 / 268 per message) */
    double noise_power_sum;         /* per block: mean_power * length - signal power of its messages */
    uint64_t noise_power_count;
    double peak_signal_power;       /* largest signalLevel */
    uint64_t strong_signal_count;   /* signalLevel > 0.50119 (-3 dBFS) */
    /* how the scan kernel converted its strips of 2768 offsets (diagnostics; no counterpart in the reference) */
    uint64_t general_strips;        /* sample by sample: first / last strips of a batch, unaligned input */
    uint64_t patched_strips;        /* UC8: strips with samples whose magnitude came from the full 256 x 256 table */
    uint64_t split_strips;          /* times a strip was taken again in more parts because one of the kernel's LDS queues overflowed */
    uint64_t sort_fallbacks;        /* batches whose items had to be sorted (a single pass of 512 offsets overflowed a queue) */
} rsb_stats;

/* Durations of the last rsb_submit, measured with HIP events on the submit stream. */
typedef struct rsb_timing {
    float scan_ms;      /* the fused convert+gate+slice+CRC kernel (dominant kernel) */
    float prune_ms;     /* filter-superset prune + survivor re-slice */
    float modeac_ms;    /* block statistics + Mode A/C kernel (0 if mode_ac off) */
    float total_ms;     /* first kernel start to last kernel end */
    float replay_ms;    /* host: sequential replay / filter bookkeeping / Mode A/C merge of the last rsb_collect */
    float sort_ms;      /* radix sort of the kept items */
    float device_replay_ms; /* device replay: fixed-point iterations + message emission (0 on the host path) */
    float copy_ms;      /* device -> host copy of the results */
    int32_t replay_iterations; /* fixed-point iterations of the device replay (0 on the host path) */
    int32_t replay_on_device;  /* 1 if the last batch was replayed on the device */
} rsb_timing;

typedef struct rsb_ctx rsb_ctx;

/* One context = one receiver channel: its own stream position, ICAO filter (icao_filter.c) and
 * device work buffers.  Replaces modesInit (readsb.c:187-251) + ifileOpen (sdr_ifile.c:115-160). */
int rsb_create(const rsb_config *cfg, rsb_ctx **out);
void rsb_destroy(rsb_ctx *ctx);

/* Start a new recording: stream position 0, zero overlap (sdr_ifile.c:200-204), empty filter
 * (icaoFilterInit, icao_filter.c:65-69), DC-block state zero (init_converter, convert.c:477), counters cleared.  The device side of
 * it is enqueued on the context's own stream without a host wait; the next rsb_submit orders itself behind it. */
int rsb_reset(rsb_ctx *ctx);

/* icaoFilterAdd (icao_filter.c:71-95): what --show-only does at start-up (readsb.c:249-250). */
int rsb_filter_add(rsb_ctx *ctx, uint32_t addr);
/* The same for n addresses in one go, in the order given (what a shard does with the filter contents handed over by the shard before it). */
int rsb_filter_add_many(rsb_ctx *ctx, const uint32_t *addrs, size_t n);
/* One icaoFilterExpire flip (icao_filter.c:147-161) as an explicit event between submits; the
 * reference drives it from wall-clock time, which a reproducible batch run cannot. */
int rsb_filter_expire(rsb_ctx *ctx);

/* ---- one recording in shards (no counterpart in the reference, which is one process on one core) ----
 * A shard is a run of whole 131072-sample blocks starting anywhere in the recording.  Its scan needs nothing but its own
 * samples and the 326 before them (the overlap of sdr_ifile.c:199-204); what couples shards is the ICAO filter
 * (icao_filter.c; mode_s.c:717-726).  Exact decoding on N GPUs, contexts created with RSB_FLAG_DEFER_PRUNE:
 *   1. every rank: rsb_seek to its shard, rsb_submit (scan), rsb_export_adders;
 *   2. all-gather of the adder lists (the one collective of the path); every rank: rsb_superset_add of the union;
 *   3. in stream order: a rank receives the filter's contents from the rank before it (rsb_filter_add each), runs
 *      rsb_collect, and passes rsb_filter_export on.
 * readsb_amd/dist.py: decode_recording_sharded does this over torch.distributed (RCCL).  Not available with dc_filter
 * (the converter's DC state would have to travel as well). */

/* Continue at stream position `stream_pos` (a multiple of RSB_BLOCK_SAMPLES) with `carry` = the RSB_TRAILING_SAMPLES samples
 * before it, in the input format (host or device memory; NULL = zeros, the start of a recording).  Nothing may be pending. */
int rsb_seek(rsb_ctx *ctx, uint64_t stream_pos, const void *carry, int carry_on_device);
/* Every address the submitted scans (and rsb_filter_add / rsb_superset_add calls) marked as "the filter may learn this"
 * (mode_s.c:717-726), ascending.  Waits for the pending scans.  *n_out = their number (RSB_EOVERFLOW if > cap). */
int rsb_export_adders(rsb_ctx *ctx, uint32_t *out, size_t cap, size_t *n_out);
/* Widen the prune's superset by these addresses (the other shards' adders).  Does not touch the filter itself. */
int rsb_superset_add(rsb_ctx *ctx, const uint32_t *addrs, size_t n);
/* How many of the LAST collected batch's kept items ask the filter about -- or would teach it -- one of `addrs` (n <= 4096)?  0 means
 * that the batch's messages cannot depend on whether those addresses were in the filter when it was replayed: what a shard needs to
 * know when the filter it assumed turns out to differ from the true one by these addresses (readsb_amd/dist.py:
 * decode_recording_parallel).  RSB_ESTATE when that batch was not replayed on the device.  No counterpart in the reference. */
int rsb_batch_references(rsb_ctx *ctx, const uint32_t *addrs, size_t n, size_t *count);

/* The filter's present contents (both generations, icao_filter.c:97-119), for handing to the next shard's rsb_filter_add. */
int rsb_filter_export(rsb_ctx *ctx, uint32_t *out, size_t cap, size_t *n_out);

/* Enqueue the scan of the next `nsamples` IQ samples of this channel, resident in device memory
 * at `iq_dev` (2 bytes/sample for UC8, 4 for SC16/SC16Q11, interleaved I,Q as convert.c reads
 * them), on HIP stream `stream` (a hipStream_t passed as void*, NULL = default stream).
 * Asynchronous.  The stream position must be a multiple of RSB_BLOCK_SAMPLES at entry (every
 * batch but the last of a recording is a whole number of blocks).  A batch that is short OR EMPTY ends
 * the recording: nothing may be submitted after it before rsb_seek / rsb_reset.  The buffer must stay
 * valid until rsb_collect returns. */
int rsb_submit(rsb_ctx *ctx, const void *iq_dev, uint64_t nsamples, void *stream);

/* Wait for the submitted batch, run the surviving candidates through the reference's sequential
 * rules (on the device; on the host with RSB_FLAG_HOST_REPLAY) and store the decoded messages, in
 * the reference's emission order, to out[0..cap).  `out` may be pinned host memory (faster copy).  *n_out receives the number of messages of this batch (if it exceeds
 * cap only the first cap are stored and RSB_EOVERFLOW is returned). */
int rsb_collect(rsb_ctx *ctx, rsb_message *out, size_t cap, size_t *n_out);

/* rsb_collect in two halves, for callers that keep the GPU busy while a batch's records cross PCIe (41 MB per Gsample of the
 * benchmark's traffic: 0.73 ms at link speed).  rsb_collect_begin returns once the batch's messages are final on the device and
 * their copy into out[] (pinned host memory, or the call simply blocks) is under way on the library's copy stream: *n_out is
 * valid, out[] is not yet.  The caller may rsb_submit the next batch -- its scan then runs beside the copy -- and calls
 * rsb_collect_wait before reading out[].  Any later rsb_collect* / rsb_reset / rsb_destroy on the context waits by itself.
 * Paths that finish on the host (RSB_FLAG_HOST_REPLAY, Mode A/C lists merged on the host) deliver in _begin; _wait is then a no-op.
 * No counterpart in the reference (readsb.c:883-886 hands messages over one by one). */
/* Many channels on one GPU: rsb_collect spends most of its time waiting for a short chain of kernels (prune, replay, emit).
 * rsb_collect_prepare enqueues that chain for the oldest submitted batch on the context's own stream and returns without waiting;
 * called for every channel before the first rsb_collect, the chains of all channels run side by side and the collects find their
 * results ready.  Optional (rsb_collect does it otherwise) and a no-op for batches that need the host's step-by-step path.
 * A filter or superset change made from outside (rsb_filter_add*, rsb_filter_expire, rsb_superset_add) while a chain is enqueued -- by
 * this call, or by the library itself for a batch submitted ahead -- discards that chain: the batch is pruned and replayed again when
 * it is collected, against the filter as it is then (what the reference's loop sees between two demodulate2400 calls). */
int rsb_collect_prepare(rsb_ctx *ctx);
/* The same with the chain on a stream of the caller's -- normally the one the batch was submitted on: the chain then follows the
 * batch's scan in that stream with no hop to another queue, and whatever the caller submits on the stream next (the next batch's
 * scan) follows the chain, so the GPU goes from scan to chain to scan without waiting for the host or for an event of another stream
 * (measured: a chain on the context's own stream starts 25 us after its scan when the submit stream is idle by then, and ~0.5 ms
 * after it when that stream already holds the next scan).  rsb_collect* of the batch waits for the chain's own
 * event, not for the stream.  What a collect has to enqueue after the chain in unusual cases (further replay rounds, the Mode A/C
 * merge on the device) goes to the same stream, behind whatever the caller has queued there meanwhile: correct, but late. */
int rsb_collect_prepare_on(rsb_ctx *ctx, void *stream);
/* N blocks x C channels in one call (last row): channel i's batch iq_dev[i][0 .. nsamples[i]) is submitted on
 * `stream` for every i, and rsb_collect_many fetches every channel's messages into out[i][0 .. cap[i]) (n_out[i] of them).
 * The channels are independent receivers (own filter, own stream position: ctx[i] from rsb_create), so this is rsb_submit /
 * rsb_collect per channel with the order that keeps the GPU busy: all scans queued first, all replay chains enqueued before the
 * first wait (they run side by side on the contexts' own streams), every channel's records crossing PCIe while the next
 * channel's results are looked at.  out[i] in pinned memory for that last part.  The first error is returned after every channel
 * has been handled (rsb_last_error on the contexts says which).
 * When the channels are distinct plain contexts of one device and format (no dc_filter, no RSB_FLAG_SC16Q11_TABLE), and there are
 * at most as many as the scan has workgroups, their scans are ONE kernel launch: every channel gets an equal share of the
 * workgroups and its wavefronts walk a long run of that channel's samples, where a channel of a few ten Msamples alone would pay
 * the kernel's set-up and ramp for two or three strips per wavefront.  Nothing is submitted then unless every channel can be; the
 * launch is timed once: rsb_timing.scan_ms is the whole launch on ctx[0] and 0 on the other channels.
 * Likewise rsb_collect_many: when every channel's batch is of the usual kind (no host replay, no exact statistics, no deferred
 * prune, nothing of it prepared yet), the prune / replay / emit chains of all channels are ONE chain of launches on ctx[0]'s
 * stream (every kernel takes all channels, one channel per blockIdx.y) with one host wait, and the filter tables of the channels
 * whose filter moved from outside are brought up to date together.  A channel whose first replay round was not the fixed point
 * (rare) goes on by itself behind the shared chain.  rsb_timing.prune_ms / device_replay_ms are then the shared chain's, the same
 * on every channel.  32 channels x 32 Msamples: 3.3 ms per call pair instead of 8.0 (profiles/r3_bench_channels32_*.json).
 * rsb_collect_many_begin is rsb_collect_many without the last step, the wait for the record copies: it returns when every
 * channel's n_out is known and its copy is under way, and the caller ends each channel with rsb_collect_wait (before reading
 * out[i]) -- meanwhile it can submit the next batches, whose scans then run beside the copies (as rsb_collect_begin does for one
 * channel). */
int rsb_submit_many(rsb_ctx *const *ctx, size_t nch, const void *const *iq_dev, const uint64_t *nsamples, void *stream);
int rsb_collect_many(rsb_ctx *const *ctx, size_t nch, rsb_message *const *out, const size_t *cap, size_t *n_out);
int rsb_collect_many_begin(rsb_ctx *const *ctx, size_t nch, rsb_message *const *out, const size_t *cap, size_t *n_out);
/* rsb_collect_prepare_on for all channels at once: the shared chain (or, for channel sets it does not take, every channel's own chain)
 * is enqueued on `stream` -- normally the one the batches were submitted on -- and the call returns; submit the next batches on the
 * stream, then rsb_collect_many[_begin], which waits for the chain's event and hands out the lists.  The host's work per channel (filter
 * bookkeeping, starting the record copies, resets, the next submit) then runs beside the next scans instead of between chain and scan. */
int rsb_collect_many_prepare_on(rsb_ctx *const *ctx, size_t nch, void *stream);

/* Device-side ordering without the host: work enqueued on `stream` after this call starts only when the chain that
 * rsb_collect_prepare put on the context's own stream has finished (without a prepared chain: when the batch's scan has).  For a
 * caller that wants the NEXT scan -- of this or another context; a scan takes every CU and nothing runs beside it -- queued
 * behind this batch's prune / replay / emit before the host has even looked at the results:
 *   rsb_submit(A, ..., s); rsb_collect_prepare(A); rsb_collect_fence(A, s); rsb_submit(B, ..., s); rsb_collect_begin(A, ...); ... */
int rsb_collect_fence(rsb_ctx *ctx, void *stream);
int rsb_collect_begin(rsb_ctx *ctx, rsb_message *out, size_t cap, size_t *n_out);
int rsb_collect_wait(rsb_ctx *ctx);

/* rsb_submit + rsb_collect */
int rsb_process(rsb_ctx *ctx, const void *iq_dev, uint64_t nsamples, void *stream,
                rsb_message *out, size_t cap, size_t *n_out);

/* Same, from a host buffer (what ifileRun's read() fills, sdr_ifile.c:209-223): stages the IQ
 * into a device buffer owned by the context first. */
int rsb_process_host(rsb_ctx *ctx, const void *iq_host, uint64_t nsamples,
                     rsb_message *out, size_t cap, size_t *n_out);

/* --dcfilter and RSB_FLAG_SC16Q11_TABLE contexts only: the magnitudes the converter produced for the last collected batch (what
 * iq_convert_fn would have written, convert.c:113-423) and, if block_means != NULL and the context runs Mode A/C, mean level / mean
 * power of each of its 131072-sample blocks.  For checking the demodulators apart from the converter's tolerance.
 * A --dcfilter context whose kernels filter for themselves holds no magnitudes: they are made by this call, from the batch's samples
 * (the caller's iq_dev, which must still be valid) and the filter states in the batch's slot, by the routine every kernel of the batch
 * used; call it before the slot is used again (two batches later). */
int rsb_get_batch_magnitudes(rsb_ctx *ctx, uint16_t *mag_host, uint64_t nsamples, double *block_means);

/* ---- live sources: the producer half of the mag_buf ring ---------------------- */

/* A live SDR handler hands over one block of raw IQ at a time: plutosdrCallback (sdr_plutosdr.c:168-244) and
 * ifileRun (sdr_ifile.c:176-253) fill one mag_buf of 131072 samples per call, copy the previous buffer's last 326
 * magnitudes in front of it -- or zeros when samples were dropped in between (sdr_plutosdr.c:224-228) -- stamp it with
 * sampleCounter * 12e6 / sample_rate and signal the demodulator thread (readsb.c:847-911).  rsb_stream is that hand-over
 * for the GPU consumer: blocks are staged in pinned host memory, every blocks_per_batch of them go to the device as one
 * batch (host -> device copy and scan of batch k run while the caller fills batch k + 1, two batches in flight), and
 * the messages of every finished batch reach `sink` in stream order, exactly those rsb_process_host returns for the
 * same samples.  Latency bound: blocks_per_batch x 54.6 ms of samples (at 2.4 MS/s) + one batch's processing;
 * rsb_stream_flush submits what is staged and delivers everything outstanding (a timer's job).
 *   nsamples        131072, or fewer for the last block of the stream (as rsb_submit)
 *   dropped_before  samples the source lost between the previous block and this one (mag_buf.dropped), whole blocks
 *                   only: the staged blocks are flushed, the stream position moves on by that many samples (timestamps
 *                   jump as sampleCounter does, sdr_plutosdr.c:208-213) and this block starts from a zero carry.
 *                   The ICAO filter is kept.  Not available with dc_filter (rsb_seek).
 * The context must have max_batch_samples >= blocks_per_batch * 131072 and nothing pending; while the stream is open
 * the caller does not call rsb_submit / rsb_collect on it.  sink runs on the caller's thread, inside push / flush /
 * close; the records it is given are valid during the call only. */
typedef struct rsb_stream rsb_stream;
typedef void (*rsb_stream_sink)(void *user, const rsb_message *msgs, size_t n);
int rsb_stream_open(rsb_ctx *ctx, uint32_t blocks_per_batch, rsb_stream_sink sink, void *user, rsb_stream **out);
int rsb_stream_push(rsb_stream *st, const void *iq_host, uint32_t nsamples, uint64_t dropped_before);
/* handle_bladerf_samples (sdr_bladerf.c:283-422): one transfer buffer as the bladeRF stream callback receives it, `bytes` long, made of
 * sub-blocks of block_size bytes = a 16-byte metadata header (magic 0x12344321, 64-bit hardware timestamp at +4, status flags at
 * +12, little endian) followed by (block_size - 16) / 4 SC16Q11 samples.  The context must have the driver's buffer geometry:
 * rsb_config.block_samples = the samples of one transfer buffer (bytes / block_size sub-blocks: 129024 for block_size 1024, 130048
 * for 2048) and, with mode_ac, mean_sub_samples = those of one sub-block -- every transfer buffer then is one struct mag_buf of
 * the reference: the greedy skip and the Mode A/C scan start fresh per buffer, and the Mode A/C noise level comes from the average
 * of the sub-blocks' means (:385-390, 403-404).  The handler's state is kept as the reference keeps it: the expected timestamp
 * starts at 0 = "none" (:289, 366), so unless the hardware counter starts at 0 the first sub-block ever is dropped by the mismatch the
 * second one raises; a timestamp that is not the expected one (forwards or backwards) discards what THIS buffer held so far and
 * restarts the buffer there (:366-375); a wrong magic value ends the buffer (:347-353).  Every buffer's overlap is the 326 samples
 * behind the previously published buffer's data, whatever was lost in between (:323-324: this driver never zeroes it).  A buffer
 * that comes out short ends the run of blocks: it is decoded as the last block of a ragged batch, and the next buffer starts a new
 * run at its own position (rsb_seek with that overlap as the carry).  Message timestamps follow the hardware timestamp:
 * sampleTimestamp = timestamp * 12e6 / sample_rate / decimation (:382; timestamps that are not multiples of the decimation are
 * rounded down to a sample).  *overruns receives the number of sub-blocks flagged BLADERF_META_STATUS_OVERRUN (only reported by
 * the reference).  rsb_stream_flush on such a stream submits whole blocks only.
 * Parity: the reference's converter and demodulators, unmodified, run on the chain of buffers the handler builds for a transfer
 * sequence with all of these cases (oracle/_ref/ref_harness demodbufs -> tests/golden/demod_bladerf_chain.npz); the handler's own
 * framing loop needs libbladeRF and cannot be built here, so the chain itself comes from a test-side restatement of :339-404
 * (tests/scenarios.py: bladerf_model).  Not available with dc_filter (rsb_seek is not). */
int rsb_stream_push_bladerf(rsb_stream *st, const void *buf, size_t bytes, uint32_t block_size, uint32_t decimation, uint32_t *overruns);
int rsb_stream_flush(rsb_stream *st);
int rsb_stream_close(rsb_stream *st);   /* flush, then free the stream (not the context) */

/* ---- the rest of decodeModesMessage on the device ------------------------------ */

/* What decodeModesMessage (mode_s.c:557-716) and decodeExtendedSquitter (mode_s.c:736-1474) make of an accepted message, and
 * decodeModeAMessage (mode_ac.c:165-199) of a Mode A/C reply: every field of struct modesMessage (readsb.h:421-635) in the
 * reference's own x86-64 layout, 416 bytes, so that a caller copies the record over its struct modesMessage and hands it to
 * useModesMessage (integration/gpu_path.c does with RSB_GPU_FIELDS=1; its _Static_asserts hold every member's offset against the
 * reference's header).  Bit-fields are given as the bytes / 32-bit words they occupy (gcc allocates them from bit 0 up, in
 * declaration order).
 * Left for the caller: sysTimestampMsg (wall clock, demod_2400.c:323), verbatim[] (only kept with Modes.net_verbatim,
 * mode_s.c:427-430), decodeCommB for DF20/21 (comm_b.c: the BDS heuristics, out of scope -- MB[] is filled), and the tracker's
 * fields (reduce_forward, cpr_decoded ...).  msg[] holds the first msgbits / 8 bytes, the rest is zero (the reference leaves stale
 * stack bytes there).  The records are read and written as 32-bit words: msgs_dev and out_dev must be 4-byte aligned.
 * Mode A/C: the reference zeroes its struct once per block, not per message (demod_2400.c:482), so an altitude decoded for one
 * reply stays in the struct for the later replies of the same block that decode none: reproduced.
 * Why on the device: the reference's decodeModesMessage handles 28 M messages per second and core (oracle/_ref/ref_harness
 * timedecode on the GPU box's host), a benchmark step delivers 0.73 M messages in 2.1 ms = 350 M per second; the kernel decodes
 * 6.2 G messages per second (profiles/r3_decode_rate.json). */
typedef struct rsb_decoded {
    uint64_t timestampMsg;             /*   0 */
    uint64_t sysTimestampMsg;          /*   8  0: the caller's */
    uint8_t msg[14];                   /*  16 */
    uint8_t verbatim[14];              /*  30  0: the caller's (Mode A/C: msg[0..1], mode_ac.c:172-173) */
    int32_t msgbits, msgtype;          /*  44  48 */
    uint32_t crc;                      /*  52 */
    int32_t correctedbits;             /*  56 */
    uint32_t addr;                     /*  60  with MODES_NON_ICAO_ADDRESS where the extended squitter says so */
    int32_t addrtype, remote, score, sbs_in, reduce_forward, source; /* 64 68 72 76 80 84 */
    double signalLevel;                /*  88 */
    uint32_t IID, AA, AC, CA, CC, CF, DR, FS, ID, KE, ND, RI, SL, UM, VS, metype, mesub; /* 96 .. 160 */
    uint8_t MB[7], MD[10], ME[7], MV[7];  /* 164 171 181 188 */
    uint8_t valid_bits[5];             /* 195  the one-bit members of readsb.h:466-492 in declaration order (gcc starts them in the byte behind
                                        *      MV[]): member k is bit k % 8 of valid_bits[k / 8] -- altitude_baro_valid 0, altitude_geom_valid 1,
                                        *      track_valid 2, track_rate_valid 3, heading_valid 4, roll_valid 5, gs_valid 6, ias_valid 7, tas_valid 8,
                                        *      mach_valid 9, baro_rate_valid 10, geom_rate_valid 11, squawk_valid 12, callsign_valid 13, cpr_valid 14,
                                        *      cpr_odd 15, cpr_decoded 16, cpr_relative 17, category_valid 18, geom_delta_valid 19, from_mlat 20,
                                        *      from_tisb 21, spi_valid 22, spi 23, alert_valid 24, alert 25, emergency_valid 26 */
    int32_t altitude_baro, altitude_baro_unit, altitude_geom, altitude_geom_unit, geom_delta; /* 200 .. 216 */
    float heading;                     /* 220 */
    int32_t heading_type;              /* 224 */
    float track_rate, roll;            /* 228 232 */
    float gs_v0, gs_v2, gs_selected;   /* 236 240 244 */
    uint32_t ias, tas;                 /* 248 252 */
    double mach;                       /* 256 */
    int32_t baro_rate, geom_rate;      /* 264 268 */
    uint32_t squawk;                   /* 272 */
    char callsign[16];                 /* 276 */
    uint32_t category;                 /* 292 */
    int32_t emergency, cpr_type;       /* 296 300 */
    uint32_t cpr_lat, cpr_lon, cpr_nucp; /* 304 308 312 */
    int32_t airground;                 /* 316 */
    double decoded_lat, decoded_lon;   /* 320 328 */
    uint32_t decoded_nic, decoded_rc;  /* 336 340 */
    int32_t commb_format;              /* 344 */
    uint32_t accuracy_bits;            /* 348  nic_a_valid = bit 0 ... sda = bits 23-24 (readsb.h:551-572) */
    int32_t accuracy_sil_type;         /* 352 */
    int32_t opstatus_sil_type, opstatus_tah, opstatus_hrd, opstatus_track_angle; /* 356 .. 368 */
    uint32_t opstatus_cc_lw, opstatus_cc_antenna_offset; /* 372 376 */
    uint32_t opstatus_bits;            /* 380  valid = bit 0, version = bits 1-3 ... cc_lw_valid = bit 18 (readsb.h:592-609) */
    uint32_t nav_fms_altitude, nav_mcp_altitude; /* 384 388 */
    float nav_qnh, nav_heading;        /* 392 396 */
    uint32_t nav_bits;                 /* 400  heading_valid = bit 0 ... modes_valid = bit 4 (readsb.h:623-627) */
    int32_t nav_heading_type, nav_altitude_source, nav_modes; /* 404 408 412 */
} rsb_decoded;
#define RSB_DECODED_BYTES 416

/* n message records on the device -> n rsb_decoded on the device, asynchronously on `stream` (one thread per message; no PCIe).
 * batch_pos / block_samples: the stream position of the batch the records came from and the context's block length -- they only
 * serve the Mode A/C quirk above (0, 0: every Mode A/C reply decoded on its own).  *cpr_filtered_dev (optional, device memory) is
 * raised once per airborne position that mode_s.c:991-998 refuses (Modes.stats_current.cpr_filtered). */
int rsb_decode_fields_device(rsb_ctx *ctx, const rsb_message *msgs_dev, size_t n, uint64_t batch_pos, uint32_t block_samples, rsb_decoded *out_dev,
                             uint32_t *cpr_filtered_dev, void *stream);
/* The same from and to host memory (blocking: upload, kernel, download). */
int rsb_decode_fields(rsb_ctx *ctx, const rsb_message *msgs, size_t n, uint64_t batch_pos, rsb_decoded *out);
/* rsb_collect + the decoded structs of the same messages: out[i] and dec[i] belong together (dec may be NULL: plain rsb_collect) */
int rsb_collect_decoded(rsb_ctx *ctx, rsb_message *out, rsb_decoded *dec, size_t cap, size_t *n_out);

/* ---- wire formats of the message list ---------------------------------------- */

/* What modesQueueOutput (net_io.c:1249-1276) writes for Beast-binary clients (modesSendBeastOutput, :760-826:
 * 0x1a, type '1'/'2'/'3', 48-bit 12 MHz timestamp, RSSI byte round(sqrt(signalLevel)*255), message; every 0x1a
 * after the type doubled) and for raw clients (modesSendRawOutput, :859-885: "@<12 hex digits>" or "*", hex
 * message, ";\n").  net_verbatim off: messages with two corrected bits are not forwarded (:1258,1264). */
enum { RSB_WIRE_BEAST = 0, RSB_WIRE_RAW = 1 };
#define RSB_WIRE_MLAT 1u       /* raw: Modes.mlat set, "@" + timestamp (net_io.c:868) */
#define RSB_WIRE_MAX_BYTES 44u /* per message: Beast 2 + 2 * (7 + 14) (net_io.c:762); raw 14 * 2 + 15 = 43 (:861) */

/* Host: records -> bytes.  *nbytes = bytes the list needs; RSB_EOVERFLOW if that exceeds cap (the first cap are stored). */
int rsb_wire_encode(int wire, uint32_t flags, const rsb_message *msgs, size_t n, uint8_t *out, size_t cap, size_t *nbytes);

/* rsb_collect with the batch delivered as that byte stream.  The records are encoded on the device (one size pass, a
 * prefix sum, one write pass), so only the wire bytes cross PCIe (about 40 % of the records' size); batches that the
 * host has to merge (Mode A/C) or replay are encoded on the host from the same records.  *n_msgs = messages of the
 * batch, forwarded or not. */
int rsb_collect_wire(rsb_ctx *ctx, int wire, uint32_t flags, uint8_t *out, size_t cap, size_t *nbytes, size_t *n_msgs);

/* The forwarding rule of useModesMessage (mode_s.c:2163-2175): the first message of an address is held back; when the
 * second arrives both go out, later ones pass.  Mode A/C messages and address 0 have no aircraft record (track.c:962-971)
 * and pass.  Keyed on rsb_message.addr, i.e. without the non-ICAO flag the reference's DF18 field decode adds
 * (mode_s.c:1387,1412,1425), and without the tracker's expiry (track.c:1447).  out needs room for 2n messages. */
typedef struct rsb_forwarder rsb_forwarder;
rsb_forwarder *rsb_forwarder_create(void);
void rsb_forwarder_destroy(rsb_forwarder *f);
int rsb_forwarder_push(rsb_forwarder *f, const rsb_message *in, size_t n, rsb_message *out, size_t cap, size_t *n_out);

/* Host merge of several channels' (or shards') message lists into ONE list in timestamp order (BASELINE north star: "results are
 * merged on the host in timestamp order").  The reference has one channel and no such step; the order defined here: ascending
 * timestampMsg, ties in list order and, within a list, in emission order (stable).  *n_out = total; RSB_EOVERFLOW if > cap. */
int rsb_merge_by_timestamp(const rsb_message *const *lists, const size_t *counts, size_t nlists, rsb_message *out, size_t cap, size_t *n_out);

int rsb_get_stats(const rsb_ctx *ctx, rsb_stats *st);
int rsb_get_timing(const rsb_ctx *ctx, rsb_timing *t);
const char *rsb_last_error(const rsb_ctx *ctx);
const char *rsb_version(void);

/* ---- stand-alone pieces of the path, for callers that bind them one by one -------------- */

/* iq_convert_fn on device buffers (convert.c:63-423): nsamples IQ -> uint16 magnitudes.
 * mean_level / mean_power may be NULL; they are the reference's, bit for bit: integer sums for UC8
 * (convert.c:104-110), float32 running sums taken in sample order for SC16 / SC16Q11 (:254-260). */
int rsb_convert_device(rsb_ctx *ctx, const void *iq_dev, void *mag_dev, uint64_t nsamples, void *stream,
                       double *mean_level, double *mean_power);

/* rsb_convert_device for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  Same kernels and same results as rsb_convert_device on a
 * context of that format / dc_filter, but the work memory (UC8 table, integer sums, the --dcfilter scan's tile arrays and block
 * means) is one fixed array in the library's code object, about 0.66 MB on every GPU the library runs on, held until the
 * process ends; the call allocates nothing, frees nothing and does not wait for the device: everything is enqueued on `stream`.
 *   nsamples      at most RSB_NOALLOC_MAX_SAMPLES per call (RSB_EINVAL and a message on stderr beyond that)
 *   dc_state_dev  filter_dc only: two doubles in DEVICE memory, the converter_state (z1_I, z1_Q) before the call's first sample
 *                 on entry and after its last on exit (zero them for a new stream, convert.c:476-478)
 *   means_dev     NULL, or two doubles in DEVICE memory that receive mean_level, mean_power (valid once the stream has run)
 * Runs on the CURRENT device.  Set-up (address look-up of the array and the UC8 table upload, a blocking copy) happens in
 * rsb_create for its device, otherwise in the first call on a device: make that call outside stream capture.  All calls on
 * one GPU share the array: calls that overlap in time on one GPU (other streams, other threads) race. */
#define RSB_NOALLOC_MAX_SAMPLES (1ull << 26)
int rsb_convert_noalloc(int input_format, int filter_dc, const void *iq_dev, uint16_t *mag_dev, uint64_t nsamples, void *stream,
                        double *dc_state_dev, double *means_dev);

/* crc.h:40-44 on the host tables the context was built with. */
uint32_t rsb_checksum(const uint8_t *msg, int bits);
/* modesChecksumDiagnose: returns number of bit errors (0..2) and their positions, or -1 if the
 * syndrome is not correctable with this context's fix_bits. */
int rsb_diagnose(const rsb_ctx *ctx, uint32_t syndrome, int bits, int bitpos[2]);

/* ---- host-only pieces (no device needed) ---------------------------------------------------- */

/* The syndrome tables modesChecksumInit(fix_bits) builds (crc.c:353-378): number of entries for
 * 56- / 112-bit frames, and entry `idx` in ascending syndrome order. */
int rsb_crc_table_size(int fix_bits, int bits);
int rsb_crc_table_entry(int fix_bits, int bits, int idx, uint32_t *syndrome, int *errors, int bitpos[2]);

/* The 65536-entry UC8 magnitude table of convert.c:35-61, indexed by the 16-bit sample word. */
const uint16_t *rsb_uc8_lut(void);

/* A gated scan offset as the finalize kernel hands it to the replay: the five sliced trial
 * messages (try_phase 4..8, demod_2400.c:179-286), how many bytes each holds (14, 7, or 1 when the
 * slicer gave up, :272-285) and sum(mag^2) over 134 / 268 samples from S[g+19] (:345-351). */
typedef struct rsb_candidate {
    uint32_t g;            /* scan offset relative to the batch start */
    uint8_t vbytes[5];
    uint8_t pad;
    uint8_t msg[5][14];
    uint64_t pow134, pow268;
} rsb_candidate;           /* 96 bytes */

typedef struct rsb_ac_candidate {
    uint32_t g;            /* f1_sample, relative to the batch start */
    uint32_t f2_clock;     /* block-local 60 MHz clock of F2 (demod_2400.c:558) */
    uint32_t modeac;       /* Gillham-reordered code (demod_2400.c:626-639) */
} rsb_ac_candidate;

/* The sequential half of the path on its own: score every candidate's phases against a fresh
 * ICAO filter, pick the best (demod_2400.c:289-301), run the CRC part of decodeModesMessage,
 * apply the greedy skips (:371, :659) per 131072-sample block, emit in the reference's order.
 * Candidates may be any superset of the offsets the reference would accept, in any order.
 * g0 = stream position of candidate offset 0 (a multiple of RSB_BLOCK_SAMPLES). */
int rsb_host_replay(int fix_bits, uint64_t g0, const rsb_candidate *cands, size_t ncands,
                    const rsb_ac_candidate *ac, size_t nac, rsb_message *out, size_t cap, size_t *n_out);

#ifdef __cplusplus
}
#endif
#endif /* READSB_AMD_H */
