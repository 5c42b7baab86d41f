// kernels.hpp -- launch interface between the C ABI (capi.cpp) and the HIP kernels (kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "modes_host.hpp"

namespace rsb {

// sample source formats a kernel can read; the first three are input_format_t (convert.h:29-32)
enum : int { kFmtUC8 = 0, kFmtSC16 = 1, kFmtSC16Q11 = 2, kFmtMag16 = 3 };

// geometry of the scan kernel that the host needs for sizing the item regions: scan offsets per strip, and
// wavefronts (= private item regions) per workgroup
#ifndef RSB_SCAN_WAVES
#define RSB_SCAN_WAVES 16
#endif
#ifndef RSB_SCAN_PASSES
#define RSB_SCAN_PASSES 6
#endif
constexpr int kScanPasses = RSB_SCAN_PASSES;                 // 512-sample passes per strip (LDS region = 512 * passes magnitudes)
constexpr int kScanStrip = 512 * kScanPasses - 304;          // ... of which the last 304 are look-ahead only
constexpr int kScanWavesPerGroup = RSB_SCAN_WAVES;

// UC8 magnitudes by radial index P = I (255 - I) + Q (255 - Q) (even, 0 .. 32512): entry P / 2; padded to a multiple of 8 entries
constexpr int kSummaryWords = 1024;  // 2^15 bits
constexpr int kRadialEntries = 16264;
// behind the radial entries: the magnitudes of the byte pairs the radial table cannot hold (two values of P, 'classes'), by
// a = |2 I - 255| (index (a - 1) / 2): [class][128].  modes_host.cpp: uc8_radial_lut / uc8_radial_patch_p1
constexpr int kRadialPatch = 256, kRadialTotal = kRadialEntries + kRadialPatch;

// device counters (one uint32 each)
enum : int {
    kCntItems = 0,     // item records appended by the scan kernel
    kCntAc,            // Mode A/C candidates
    kCntSurvItems,     // items kept by the prune kernel   } one 8-byte pair: k_prune bumps both with a single
    kCntItemsTotal,    // all items (sum over regions)     } 64-bit atomic per workgroup
    kCntGated,         // offsets that passed the preamble gate
    kCntSurvivors,     // distinct offsets among the kept items (written by the unique step)
    kCntSlowPath,      // candidates / items that overflowed an LDS queue and took the per-thread path
    kCntOverflow,      // item regions that overflowed (results incomplete -> RSB_EOVERFLOW)
    kCntChanged,       // device replay: the first-add table moved in this iteration
    kCntAccepted,      // device replay: messages accepted
    kCntNewAdders,     // device replay: addresses the filter learned in this batch
    kCntAccE0,         // device replay: accepted messages by corrected bits (0, 1, 2)
    kCntAccE1,
    kCntAccE2,
    kCntGeneralStrips, // scan: strips converted sample by sample (first / last strips, unaligned input)
    kCntFixedStrips,   // scan, UC8: strips with samples whose magnitude the radial table cannot hold (patched from the full table)
    kCntSplitStrips,   // scan: times a strip was taken again in more parts because an LDS queue overflowed
    kCntSuperChunks,   // scan: super-chunks (kScanWavesPerGroup chunks each) handed out beyond every workgroup's first one
    kCntCount = 20,
    // debug builds (RSB_STAGE_CLOCKS) and ablation runs only
    kDbgClkConvert = 20, kDbgClkEdge, kDbgClkGate, kDbgClkDf, kDbgClkSlice, kDbgStrips, kDbgGateBatches, kDbgDfBatches, kDbgSliceBatches,
    kDbgEdgeCands, kDbgFrames, kDbgLongFrames,
    // k_replay_emit, RSB_STAGE_CLOCKS builds: ticks / 16 summed over wavefronts per stage of an iteration of 64 messages, and the iterations
    kDbgEmitSetup = 32, kDbgEmitGather, kDbgEmitTrios, kDbgEmitRepair, kDbgEmitStore, kDbgEmitIters,
    kCntAlloc = 40
};

struct DeviceTables {      // read-only, built once per context
    const uint16_t *uc8_lut;       // 65536 x u16
    const uint16_t *uc8_radial;    // kRadialTotal x u16: the same magnitudes by radial index + the patch entries (modes_host.cpp: uc8_radial_lut)
    uint32_t uc8_patch_p1;         // the P = I (255 - I) + Q (255 - Q) of patch class 1 (any other sentinel entry is class 0)
    const uint32_t *crc_bytes;     // 256 x u32
    const uint32_t *sbs;           // 112 x u32: syndrome of each single set bit of a 112-bit frame (crc.c:40)
    const uint32_t *syn_keys[2];   // [0] = 56-bit frames, [1] = 112-bit frames
    const uint32_t *syn_vals[2];
    uint32_t syn_mask[2];
    int nfix;
};

// One batch = the scan offsets [0, n) of stream S = carry(326) ++ iq(n); see DESIGN.md "stream model".
struct BatchArgs {
    const void *iq;        // n samples
    const void *carry;     // 326 samples of the same format (the tail of the previous batch)
    int carry_valid;       // 0: the 326 leading magnitudes are zero (sdr_ifile.c:200-204)
    int fmt;
    int iq_aligned;        // iq is 16-byte aligned: the scan kernel may use vector loads
    uint64_t n;
    DeviceTables tab;
    ItemRec *items;          // n_regions wave-private regions of item_region records, then one shared region
    uint32_t item_region;
    uint32_t n_regions;
    uint32_t shared_cap;
    uint32_t *region_counts; // items written per wave region (the runs of its chunks one behind the other)
    uint32_t *region_kept;   // (kept items per region: a by-product of k_prune_count)
    uint64_t *region_masks;  // ordered prune: one keep mask per 64 item slots, mask_stride (= item_region / 64) masks per region
    uint32_t mask_stride;
    // The scan hands out the strips in CHUNKS of chunk_strips consecutive strips (chunk c = strips [c * chunk_strips, ...)), taken by
    // whichever wavefront is free next; a chunk's items are one sorted run inside its wavefront's region, found through chunk_dir.
    uint32_t chunk_strips, n_chunks;
    uint2 *chunk_dir;        // per chunk: {first item slot (global index into items), items}
    uint32_t *chunk_kept;    // ordered prune: kept items of the chunks before this one within its group of 256 chunks
    uint32_t *chunk_group;   // ... and kept items before every such group (n_chunks / 256 + 1 entries)
    uint32_t surv_item_cap;
    uint64_t *skeys[2];    // kept items: sort key g << 3 | phase (ping-pong buffers for the radix sort)
    uint32_t *sinfo[2];    // ... and their info words
    uint32_t *surv_g;      // distinct offsets of the kept items, ascending
    Survivor *survivors;
    uint32_t surv_cap;
    AcCand *ac;
    uint32_t ac_cap;
    uint32_t *adder_bitmap;  // 2^24 bits: superset of every address the filter can hold
    uint32_t *adder_summary; // kSummaryWords x u32 or null: a hashed summary of the bitmap, rebuilt in front of the ordered prune of a large batch
    uint32_t *counters;
    uint64_t *block_sums;    // per 131072-block: sum(mag), sum(mag^2) over its new samples
    const double *block_means; // or, when the caller already has them (struct mag_buf): mean_level, mean_power per block
    int emit_dead;           // also emit a marker item per gated offset (exact demod statistics)
    int keep_all;            // prune keeps every item (filter owned by the caller)
    int debug_flags;         // ablation only (RSB_ABL): 1 skip classify, 2 skip slicing loads, 4 skip reductions
    int stage_limit;         // debug/ablation only (RSB_STAGE_LIMIT): stop the scan kernel after stage k
    uint32_t block_len;      // samples per demodulator call (mag_buf.length of a full buffer): 131072 unless the source builds other buffers
    // --dcfilter inside the scan (DESIGN.md "DC block, fused"): the filter state before every 8-sample chunk of iq is
    // (float)(dc_local[chunk] + dc_tile_start[chunk / 512] * dc_pow[chunk % 512]); the samples of a chunk then go through the
    // reference's float recurrence (convert.c:137-140) from there.  Null dc_local: no DC filter in the scan.
    const float2 *dc_local;        // per chunk: the state before it had the chunk's 4096-sample tile started from zero
    const double2 *dc_tile_start;  // per tile: the state at its start
    const double *dc_pow;          // dc_b^(8 j), j < 512
    float dc_a, dc_b;
    int carry_is_mag;              // the carry holds 326 magnitudes (u16), not samples of fmt
    void *carry_out;         // rsb_submit_many's one launch: where the batch's last 326 samples go (the next batch's carry), or null
    uint64_t *wave_times;    // debug only (RSB_WAVE_TIMES): per scan wavefront {start, end (100 MHz ticks), XCC id << 32 | HW_ID, strips taken}
};

// The sequential half on the device (DESIGN.md "device replay"): the greedy scan of demod_2400.c:289-374
// walks each 131072-sample block with one thread over the sorted kept items, and the coupling through
// the ICAO filter (mode_s.c:717-726) is resolved by iterating a "first added at" table to its fixed point.
struct ReplayArgs {
    const uint64_t *keys;    // sorted: g << 3 | phase
    const uint32_t *info;    // the items' info words, same order
    uint32_t n_items;        // the host's BOUND on the number of items (sizes the grids); the kernels take the count itself from ...
    const uint32_t *n_dev;   // ... here (the prune's counters[kCntSurvItems]), clamped to n_cap; nullptr: n_items is exact
    uint32_t n_cap;
    uint32_t *counters;      // the batch's device counters (kernels.hpp enum)
    uint32_t *acc;           // per item: score + 1 if this (offset, phase) is decoded and accepted, else 0
    uint32_t fold_with_records;  // k_replay_emit also does k_replay_fold's work for the messages it writes (a chain whose first round holds)
    uint32_t *cand;          // per item: k_replay_score's verdict for the offset that starts there (scratch)
    uint32_t *acc_rank;      // per accepted item: its rank among the accepted messages of its 131072-sample block (k_replay_walk)
    uint32_t *blk_base;      // accepted messages before every block, nblocks + 1 entries (k_replay_blkbase)
    uint32_t *acc_idx;       // message number -> item (k_replay_list; takes cand's place: the walk is done with it by then)
    uint32_t *t_cur;         // 2^24 x u32: 0 = in the filter before this batch, g + 1 = first added by the message at
    uint32_t *t_next;        //             offset g of this batch, 0xffffffff = never
    uint32_t *blk_start;     // first item of every 131072-sample block, nblocks + 1 entries (launch_replay_bounds)
    uint32_t *blk_counts;    // accepted messages per 131072-sample block
    uint32_t nblocks;
    uint32_t block_len;      // samples per block (BatchArgs::block_len)
    uint2 *new_adders;       // (g + 1, address) of every address learned in this batch
    uint32_t new_adders_cap;
    rsb_message *out;
    uint32_t out_cap;
    uint64_t g0;             // stream position of the batch
    rsb_message *host_out;   // k_replay_emit: if set, the records go here (the caller's pinned host buffer, device-mapped) instead of `out`
    uint32_t host_cap;       // ... which holds this many
};
constexpr uint32_t kNever = 0xffffffffu;
// counters[kCntOverflow]: full item regions in the low 16 bits; the one-pass prune adds this per tile that gave up waiting for the tiles before it
constexpr uint32_t kPruneStallUnit = 1u << 16;

// One channel's batch in a launch that takes the batches of several channels (rsb_collect_many)
struct ChainArgs {
    BatchArgs a;
    ReplayArgs r;
};
struct ChainDims {
    uint32_t nch;          // channels (the grids' y extent)
    uint32_t max_regions, max_chunks;   // the largest BatchArgs::n_regions / n_chunks
    uint32_t max_items;    // the largest bound on the kept items (ReplayArgs::n_items)
    uint32_t max_nblocks;  // the largest ReplayArgs::nblocks
    bool any_summary;      // some channel's prune uses the hashed summary of the bitmap
};

int scan_grid_size(int device);
hipError_t launch_scan(const BatchArgs &a, int grid, hipStream_t s);
// nbatches batches of one format in one launch: args_dev[i] built for groups_per_batch workgroups (n_regions = 16 x that), n = 0 allowed
hipError_t launch_scan_many(int fmt, const BatchArgs *args_dev, uint32_t nbatches, uint32_t groups_per_batch, hipStream_t s);
// in front of launch_scan_many: every batch's counters and region counts to zero and its carry_out filled, in one launch
hipError_t launch_submit_prepare(const BatchArgs &a, hipStream_t s);  // one batch (a.carry_out set: its overlap for the next batch is copied as well)
hipError_t launch_submit_prepare_many(const BatchArgs *args_dev, uint32_t nbatches, uint32_t max_regions, hipStream_t s);
hipError_t launch_prune(const BatchArgs &a, hipStream_t s);
hipError_t launch_prune_ordered(const BatchArgs &a, hipStream_t s);
// sorts the n_items kept items by (g, phase), derives the ascending list of distinct offsets
// (count -> counters[kCntSurvivors]) and re-slices each of them into a Survivor record
size_t sort_temp_bytes(uint32_t max_items);
hipError_t launch_sort(const BatchArgs &a, uint32_t n_items, void *temp, size_t temp_bytes, int *sorted_buf, hipStream_t s);
hipError_t launch_unique_finalize(const BatchArgs &a, uint32_t n_items, int sorted_buf, void *temp, size_t temp_bytes, hipStream_t s);
// device replay: one fixed-point iteration (lookup, walk, compare; counters[kCntChanged] tells whether to go on),
// the reset of the table that becomes t_next, and the final list / fold / compact / emit step
hipError_t launch_replay_bounds(const ReplayArgs &r, hipStream_t s);  // once per batch, before the iterations
hipError_t launch_replay_iteration(const BatchArgs &a, const ReplayArgs &r, hipStream_t s);
hipError_t launch_replay_first_round(const BatchArgs &a, const ReplayArgs &r, hipStream_t s);  // a chain's first round + the list: no next table
hipError_t launch_replay_next_table(const ReplayArgs &r, hipStream_t s);                          // ... which this builds when that round moved the table
hipError_t launch_replay_reset(const BatchArgs &a, const ReplayArgs &r, uint32_t *table, hipStream_t s);
hipError_t launch_replay_finish(const BatchArgs &a, const ReplayArgs &r, void *temp, size_t temp_bytes, hipStream_t s);
// the two halves of launch_replay_finish on their own: where every accepted message goes (block bases; total ->
// counters[kCntAccepted]; the addresses learned -> r.new_adders), and the records (to r.out, or straight to r.host_out)
hipError_t launch_replay_select(const BatchArgs &a, const ReplayArgs &r, void *temp, size_t temp_bytes, hipStream_t s);
hipError_t launch_replay_emit(const BatchArgs &a, const ReplayArgs &r, hipStream_t s);
// (launch_replay_select leaves the tables alone; this folds the batch's new members into them once the fixed point is certain)
hipError_t launch_replay_fold(const ReplayArgs &r, hipStream_t s);
// The first-add tables of several channels brought up to their host filters in two launches (entries listed in stale -> "never",
// then entries listed in members, and entry 0, -> "in the filter before the batch"); u_dev[i] holds device pointers
struct FilterUpdate {
    uint32_t *t0, *t1;
    const uint32_t *stale, *members;
    uint32_t n_stale, n_members;
};
hipError_t launch_filter_update_many(const FilterUpdate *u_dev, uint32_t nch, uint32_t max_stale, uint32_t max_members, hipStream_t s);
// The whole chain for several channels' batches, one launch per stage (kernels.hip "the *_many forms"): the ordered prune, then
// bounds / seed / score / walk / adds / compare / places / list / records / fold-if-converged, then every channel's
// [changed | counters | block counts | first pin_adders new adders] gathered to readback_dev + i * readback_stride (words).
hipError_t launch_chain_prune_many(const ChainArgs *args_dev, const ChainDims &d, hipStream_t s);
hipError_t launch_chain_replay_many(int fmt, const ChainArgs *args_dev, const ChainDims &d, uint32_t *readback_dev, uint32_t readback_stride, uint32_t pin_adders,
                                    hipStream_t s);
// wire formats (wire.hpp) of n message records on the device: launch_wire_offsets leaves the byte offset of message i in
// off[i] and the total in off[n]; launch_wire_write stores the bytes at out + off[i]
struct WireArgs {
    const rsb_message *msgs;
    uint32_t n;
    int wire;
    uint32_t flags;
    uint32_t *off;  // n + 1 entries
    uint8_t *out;
};
// Mode A/C messages into the Mode S list on the device (readsb.c:883-886: each block's Mode S messages, then its Mode A/C
// messages).  blk_first[b] = Mode S messages before block b (nblocks + 1 entries), ac_before[b] = Mode A/C messages before
// block b, ac_dst[t] = final position of Mode A/C message t.  dst must not overlap modes.
struct MergeArgs {
    const rsb_message *modes;
    uint32_t n_modes;
    const uint32_t *blk_first;
    const uint32_t *ac_before;
    uint32_t nblocks;
    const rsb_message *ac_msgs;
    const uint32_t *ac_dst;
    uint32_t n_ac;
    rsb_message *dst;
};
hipError_t launch_merge_modeac(const MergeArgs &m, hipStream_t s);
// decode_fields.hip: n message records -> n rsb_decoded (struct modesMessage as decodeModesMessage / decodeModeAMessage leave it)
hipError_t launch_decode_fields(const rsb_message *msgs_dev, uint32_t n, uint64_t batch_pos, uint32_t block_samples, rsb_decoded *out_dev,
                                uint32_t *cpr_filtered_dev, hipStream_t s);
hipError_t launch_wire_offsets(const WireArgs &w, void *temp, size_t temp_bytes, hipStream_t s);
hipError_t launch_wire_write(const WireArgs &w, hipStream_t s);
// first-add tables := "never" everywhere, 0 for address 0 and the n given addresses
hipError_t launch_bitmap_set(uint32_t *bitmap, const uint32_t *addrs_dev, uint32_t n, hipStream_t s);
hipError_t launch_bitmap_list(const uint32_t *bitmap, uint32_t *out, uint32_t cap, hipStream_t s);
hipError_t launch_items_reference(const uint32_t *info, uint32_t n, const uint32_t *addrs_dev, uint32_t na, uint32_t *count, hipStream_t s);
hipError_t launch_filter_rebuild(uint32_t *t0, uint32_t *t1, const uint32_t *addrs_dev, uint32_t n, hipStream_t s);
// the same without refilling the tables: stale_dev[0..n_stale) are the entries that are 0 now and go back to "never" first
hipError_t launch_filter_update(uint32_t *t0, uint32_t *t1, const uint32_t *stale_dev, uint32_t n_stale, const uint32_t *addrs_dev, uint32_t n, hipStream_t s);
hipError_t launch_modeac(const BatchArgs &a, hipStream_t s);
// integer formats: sum(mag), sum(mag^2) per 131072-sample block into a.block_sums (zeroed by the caller); launch_modeac
// runs it itself when it is given no block means
hipError_t launch_block_sums(const BatchArgs &a, hipStream_t s);
// SC16 / SC16Q11: per block of blk_len samples the reference's float32 running sums, as means (level, power)
hipError_t launch_block_means_float(int fmt, const void *iq, uint64_t n, uint64_t blk_len, double *means, hipStream_t s);
// means[b] := the average of the per_block consecutive entries of sub_means that belong to block b (a driver that converts a buffer in
// sub-blocks and averages their means, sdr_bladerf.c:385-390, 403-404); the last block may hold fewer
hipError_t launch_means_average(const double *sub_means, uint32_t nsub, uint32_t per_block, double *means, uint32_t nblocks, hipStream_t s);
// --dcfilter converters (chunked linear-recurrence scan): n samples -> n magnitudes.  tile_sums / tile_start hold one double2 per 4096
// samples (+1); z_state is the carried DC state (in: before sample 0, out: after sample n-1); block_means (optional) receives
// the mean level / mean power of every 131072-sample block of the call
constexpr int kDcTileSamples = 4096;
// The fused form's share of launch_convert_dc: tile sums (one read of the IQ), the scan over them, and -- new -- the per-chunk local
// states; no magnitudes are written.  z_state: carried in, and out when n is a whole number of tiles.
hipError_t launch_dc_states(int fmt, const void *iq, uint64_t n, double2 *tile_sums, double2 *tile_start, float2 *local, double2 *z_state, hipStream_t s);
// dc_b^(8 j), j < 512, as the kernels use it (512 doubles on the host)
void dc_pow_table(double *out512);
void dc_coefficients(float *dc_a, float *dc_b);
// the last 326 magnitudes of a fused-DC batch -> carry_out (u16); the batch's magnitudes [0, n) -> mag (tests: rsb_get_batch_magnitudes)
hipError_t launch_dc_carry(const BatchArgs &a, uint16_t *carry_out, hipStream_t s);
hipError_t launch_dc_materialize(const BatchArgs &a, uint16_t *mag, hipStream_t s);
hipError_t launch_convert_dc(int fmt, const void *iq, uint16_t *mag, uint64_t n, double2 *tile_sums, double2 *tile_start, double2 *z_state, double *block_means,
                             hipStream_t s);
hipError_t launch_convert(int fmt, const void *iq, uint16_t *mag, uint64_t n, const uint16_t *lut, uint64_t *sums, hipStream_t s);
// SC16Q11 as a reference built with -DSC16Q11_TABLE_BITS=use_bits converts it (convert.c:297-328); sums as launch_convert's
hipError_t launch_convert_q11t(const void *iq, uint16_t *mag, uint64_t n, int use_bits, uint64_t *sums, hipStream_t s);

}  // namespace rsb
