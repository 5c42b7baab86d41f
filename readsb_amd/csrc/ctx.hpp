// ctx.hpp -- the context behind the opaque rsb_ctx of include/readsb_amd.h.
#pragma once

#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "modes_host.hpp"

// Everything one batch in flight owns: the scan kernel's outputs, its counters and events.  Two slots let the scan of
// batch i + 1 run while batch i is sorted, replayed and copied out (rsb_submit may be called once more before rsb_collect).
struct rsb_slot {
    rsb::ItemRec *d_items = nullptr;
    uint32_t *d_region_counts = nullptr;
    uint32_t *d_region_kept = nullptr;
    uint64_t *d_region_masks = nullptr;
    uint2 *d_chunk_dir = nullptr;
    uint32_t *d_chunk_kept = nullptr, *d_chunk_group = nullptr;
    bool pruned = false;              // the filter-superset prune of the pending batch has been enqueued
    bool pruned_in_order = false;     // ... as launch_prune_ordered: its kept items need no sort
    uint64_t *d_skeys[2] = {nullptr, nullptr};
    uint32_t *d_sinfo[2] = {nullptr, nullptr};
    rsb::AcCand *d_ac = nullptr;
    uint32_t *d_counters = nullptr;
    uint64_t *d_block_sums = nullptr;
    double *d_block_means = nullptr;  // SC16 / SC16Q11 with Mode A/C: the float converters' per-block means
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // scan start / end, prune end, Mode A/C end
    hipEvent_t rev[3] = {nullptr, nullptr, nullptr};          // collect side: sort start / end, replay end
    hipStream_t chain_on = nullptr;   // the stream the enqueued chain is on: the context's own, or -- rsb_collect_prepare_on -- the caller's
    bool chain_foreign = false;
    // device replay of this batch (one set per slot: the chain of batch k + 1 is enqueued while batch k's records still leave)
    rsb_message *d_out = nullptr;     // the batch's message records
    uint32_t *d_blk_counts = nullptr, *d_blk_start = nullptr, *d_blk_base = nullptr;
    uint2 *d_new_adders = nullptr;
    uint32_t *h_pin = nullptr;        // pinned: what the host reads back per batch (changed flag, counters, block counts, first new adders)
    bool chain_enqueued = false;      // the first replay round, the records and the readbacks are on the replay stream already
    bool chain_count_on_device = false, chain_to_host = false;
    bool chain_folded = false;        // rsb_collect_many's shared chain folded the batch's new members into the tables already
    const rsb_slot *time_from = nullptr;  // ... and its stage times are in that slot's events
    rsb::ReplayArgs chain_r;
    bool pending = false;
    float2 *d_dc_local = nullptr;     // the DC block inside the scan: the filter's per-chunk states of this slot's batch (kernels.hpp: BatchArgs::dc_local)
    double2 *d_dc_tile_start = nullptr;
    bool scan_timed = true, ac_timed = true;  // ev[0..1] / ev[2] were recorded for this batch (not for every channel of a one-launch submit)
    uint64_t submit_seq = 0;  // the order in which the batches of all contexts were submitted
    float scan_share = 1.0f;  // rsb_submit_many: this channel's share of the one scan launch (timing.scan_ms)
    const double *means_on_device = nullptr;  // per-block (mean_level, mean_power) of the pending batch, when taken from float means
    uint64_t pending_n = 0, g0 = 0;   // samples of the batch, its stream position
    hipStream_t pending_stream = nullptr;
    rsb::BatchArgs pending_args;
};

struct rsb_ctx {
    explicit rsb_ctx(const rsb_config &c) : cfg(c), replay(c.fix_bits) {
        run_ac = c.mode_ac != 0;
        block_len = c.block_samples ? c.block_samples : rsb::kBlock;
        mean_sub = c.mean_sub_samples;
        replay.set_block_len(block_len);
        std::memset(&timing, 0, sizeof(timing));
    }

    rsb_config cfg;
    rsb::Replayer replay;
    uint32_t block_len = rsb::kBlock;  // samples per demodulator call (rsb_config.block_samples)
    uint32_t mean_sub = 0;             // rsb_config.mean_sub_samples: the converter is called per this many samples, a block's means are the calls' averages
    double *d_sub_means = nullptr;     // ... the calls' means of the pending batch

    // read-only device tables
    uint16_t *d_lut = nullptr;
    uint16_t *d_radial = nullptr;
    uint32_t *d_crc_bytes = nullptr;
    uint32_t *d_sbs = nullptr;
    uint32_t *d_syn_keys[2] = {nullptr, nullptr};
    uint32_t *d_syn_vals[2] = {nullptr, nullptr};
    uint32_t syn_mask[2] = {0, 0};

    // device work buffers
    uint32_t *d_surv_g = nullptr;
    rsb::Survivor *d_surv = nullptr;
    void *d_sort_temp = nullptr;
    size_t sort_temp_bytes = 0;
    uint32_t item_cap = 0, surv_item_cap = 0, surv_cap = 0, ac_cap = 0;
    int scan_grid = 512;
    int sorted_buf = 0;
    bool force_sort = false;          // RSB_FORCE_SORT: always the atomic prune + radix sort (A/B runs)
    unsigned sort_fallbacks = 0;      // batches that had to be pruned again with the sort (a scan wavefront overflowed an LDS queue)
    int stage_limit = 99;
    uint32_t chunk_strips_forced = 0;  // RSB_CHUNK_STRIPS (tuning runs)
    size_t chunks_cap = 0;
    int debug_flags = 0;
    uint64_t *d_wave_times = nullptr;  // RSB_WAVE_TIMES=<file>: the scan wavefronts' start / end stamps of the last collected batch go there
    uint32_t *d_bitmap = nullptr;
    size_t nblocks_cap = 0;
    // a slot's block of [counters (kCntAlloc) | messages per block (nblocks_cap, padded to an even count) | new filter members]: word offsets
    size_t back_blk() const { return rsb::kCntAlloc; }
    size_t back_adders() const { return rsb::kCntAlloc + ((nblocks_cap + 1) & ~(size_t)1); }
    void *d_stage = nullptr;
    size_t stage_bytes = 0;
    uint16_t *d_mag = nullptr;  // drop-in block path
    // device replay
    uint32_t *d_first_add[2] = {nullptr, nullptr};  // 2 x 2^24 "first added at" tables (kernels.hpp ReplayArgs)
    int first_cur = 0;
    bool filter_dirty = true;       // the tables do not mirror replay.filter() (reset, external add / expire, host replay)
    // tables_valid: both tables are "never" everywhere except the entries listed in table_members (and address 0), which are 0.
    // A mirror then only undoes / sets those entries instead of refilling 2 x 64 MB.  False after anything that left an
    // iteration's intermediate values in them.
    bool tables_valid = false;
    std::vector<uint32_t> table_members;
    bool host_replay_only = false;  // RSB_FLAG_HOST_REPLAY
    bool defer_prune = false;       // RSB_FLAG_DEFER_PRUNE: k_prune runs at collect time
    bool exact_stats = false;       // RSB_FLAG_EXACT_STATS: every gated offset goes to the host replay, block means are kept
    std::vector<double> h_blk_mean_power;
    std::vector<uint64_t> h_blk_sums;
    static constexpr uint32_t kPinAdders = 1024;
    uint32_t *d_addr_upload = nullptr;
    uint32_t *h_addr_upload = nullptr;    // pinned twin of d_addr_upload: the filter mirror goes up without a host wait
    hipEvent_t addr_upload_done = nullptr; // the last upload from h_addr_upload has been read
    uint32_t out_cap = 0;
    uint32_t *d_wire_off = nullptr;  // rsb_collect_wire: byte offset per message (grown on demand)
    uint8_t *d_wire_out = nullptr;   // ... and the encoded stream
    size_t wire_off_cap = 0, wire_out_cap = 0;
    // rsb_decode_fields / rsb_collect_decoded: staging for records that come from the host, the decoded structs (grown on demand)
    rsb_message *d_dec_in = nullptr;
    rsb_decoded *d_decoded = nullptr;
    size_t dec_in_cap = 0, decoded_cap = 0;
    uint32_t *d_addr_list = nullptr;             // address lists from the caller (superset / filter additions, reference queries), grown on demand
    size_t addr_list_cap = 0;
    const uint32_t *last_items_info = nullptr;   // the kept items of the last collected batch (device replay): what rsb_batch_references asks
    uint32_t last_items_n = 0;
    const rsb_message *last_list_dev = nullptr;  // collect_impl: where the batch's final list lies on the device (nullptr: only on the host)
    uint64_t last_g0 = 0;                        // ... and the batch's stream position
    static constexpr uint32_t kNewAddersCap = 8192;
    std::vector<uint32_t> h_blk_counts, h_addrs;
    std::vector<uint2> h_new_adders;
    uint64_t *d_sums = nullptr;
    // --dcfilter: magnitudes of the batch (the scan then reads them instead of IQ), the scan's tile arrays, the carried state
    uint16_t *d_magbuf = nullptr;
    int q11_table_bits = 0;  // RSB_FLAG_SC16Q11_TABLE: the converter of a reference built with -DSC16Q11_TABLE_BITS (0 = the float one)
    double2 *d_dc_tile_sums = nullptr, *d_dc_tile_start = nullptr, *d_dc_state = nullptr;
    size_t dc_tiles_cap = 0;
    bool dc_fused = false;            // --dcfilter without a magnitude buffer: Mode S only (the scan, the candidate and the record kernels filter for themselves)
    double *d_dc_pow = nullptr;
    uint16_t *d_dc_mags = nullptr;    // rsb_get_batch_magnitudes on such a context: the last batch's magnitudes, made on request
    size_t dc_mags_cap = 0;
    rsb::BatchArgs last_args;         // of the last collected batch

    // drop-in block mode (dropin.cpp): every call is one struct mag_buf -- magnitudes in, overlap included
    bool block_mode = false;
    bool run_modes = true, run_ac = false;
    double *d_means = nullptr;

    // stream state
    static constexpr int kMaxSlots = 4;
    rsb_slot slots[kMaxSlots];
    // the 326 samples before batch k (the tail of batch k - 1, input format) live in d_carry[k % (slots + 1)]: written when batch k - 1 is
    // submitted, read by the scan and the replay of batch k, free again once batch k is collected
    static constexpr int kCarries = kMaxSlots + 1;
    void *d_carry[kCarries] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int carry_valid[kCarries] = {0, 0, 0, 0, 0};
    int nslots = 1;                   // kMaxSlots unless --dcfilter (one magnitude buffer)
    unsigned submitted = 0, collected = 0;  // batches so far: slot = count % nslots; submitted - collected <= nslots
    rsb_slot &submit_slot() { return slots[submitted % nslots]; }
    rsb_slot &collect_slot() { return slots[collected % nslots]; }
    bool any_pending() const { return submitted != collected; }
    uint64_t last_n = 0;              // samples of the last collected batch
    hipStream_t copy_stream = nullptr;    // the finished record list leaves on this one (a DMA copy beside whatever the kernels do)
    bool defer_copy_wait = false;         // rsb_collect_begin: leave the record copy in flight ...
    bool copy_in_flight = false;          // ... until rsb_collect_wait (or the next call that needs the copy stream idle)
    hipEvent_t reset_done = nullptr;      // behind rsb_reset's clearing of the bitmap on the replay stream
    bool reset_pending = false;           // ... which the next scan has not been made to wait for yet
    hipEvent_t copy_done = nullptr;       // behind the last record copy on the copy stream
    const rsb_slot *copy_slot = nullptr;  // the slot whose d_out that copy reads: a chain that writes it again waits for copy_done
    hipStream_t replay_stream = nullptr;  // sort / replay / copy-out run here, behind the batch's scan (event) and beside the next one's
    uint64_t stream_pos = 0;          // of the next batch to collect
    uint64_t submit_pos = 0;          // of the next batch to submit
    bool finished = false, submit_finished = false;

    rsb_timing timing;
    std::string err;
    std::vector<rsb::Survivor> h_surv;
    std::vector<rsb::AcCand> h_ac;
    std::vector<rsb_message> msgs, dev_msgs, ac_msgs;
    std::vector<uint32_t> h_ac_counts;
    std::vector<uint8_t> h_merge;     // device Mode A/C merge: staging of the tables and records it uploads
    uint8_t *d_merge = nullptr;
    uint32_t *d_summary = nullptr;          // hashed summary of d_bitmap for the ordered prune of large batches (kernels.hip: k_prune_summary)
    uint64_t summary_from = 1ull << 27;     // samples: below, two more launches cost more than the bitmap reads they save (RSB_SUMMARY_FROM: tests)
    rsb::BatchArgs *d_many_args = nullptr;  // rsb_submit_many: the channels' scan arguments for the one launch (held by the first channel)
    rsb::BatchArgs *h_many_args = nullptr;  // pinned, two halves used in turn
    size_t many_args_cap = 0;
    hipEvent_t many_args_read[2] = {nullptr, nullptr};
    bool many_args_pending[2] = {false, false};
    unsigned many_calls = 0;
    hipStream_t tables_stream = nullptr;    // the stream that last worked on the first-add tables (mirror, seed, fold), until a host wait covered it
    hipEvent_t tables_ev = nullptr;         // ... and behind its last operation there: a chain on another stream waits for it
    size_t many_prepared = 0;               // rsb_collect_many: channels whose shared chain is enqueued and not yet waited for (held by the first channel)
    hipStream_t many_stream = nullptr;      // ... the stream it is on
    bool many_foreign = false;              // ... which is the caller's (rsb_collect_many_prepare_on)
    uint32_t many_stride = 0, many_nblocks = 0;
    hipStream_t chain_stream = nullptr;     // set while rsb_collect_many collects this channel behind a chain shared with other channels
    rsb::ChainArgs *d_chain_args = nullptr, *h_chain_args = nullptr;  // rsb_collect_many: the channels' chain arguments (held by the first channel)
    uint32_t *d_chain_back = nullptr, *h_chain_back = nullptr;        // ... and what comes back of every channel, gathered
    size_t chain_args_cap = 0, chain_back_cap = 0;
    uint32_t *d_mirror = nullptr, *h_mirror = nullptr;  // ... and the table updates of the channels whose filter moved (FilterUpdate headers, then address lists)
    size_t mirror_cap = 0;  // words
    size_t merge_cap = 0;

    // grid_limit: the scan workgroups this batch may have (0 = the context's scan_grid; rsb_submit_many gives every channel its share)
    rsb::BatchArgs batch_args(const rsb_slot &b, const void *iq, uint64_t n, int fmt, const void *carry = nullptr, int grid_limit = 0) const {
        rsb::BatchArgs a;
        std::memset(&a, 0, sizeof(a));
        a.iq = iq;
        a.carry = carry ? carry : d_carry[submitted % kCarries];
        a.carry_valid = carry ? 1 : carry_valid[submitted % kCarries];
        a.fmt = fmt;
        a.n = n;
        a.tab.uc8_lut = d_lut;
        a.tab.uc8_radial = d_radial;
        a.tab.uc8_patch_p1 = rsb::uc8_radial_patch_p1();
        a.tab.crc_bytes = d_crc_bytes;
        a.tab.sbs = d_sbs;
        for (int w = 0; w < 2; ++w) {
            a.tab.syn_keys[w] = d_syn_keys[w];
            a.tab.syn_vals[w] = d_syn_vals[w];
            a.tab.syn_mask[w] = syn_mask[w];
        }
        a.tab.nfix = cfg.fix_bits;
        a.iq_aligned = ((uintptr_t)iq % 16 == 0) ? 1 : 0;
        a.items = b.d_items;
        {
            // one item region per wavefront that will actually run (launch_scan shrinks the grid for small batches)
            const uint64_t nstrips = (n + 2 + rsb::kScanStrip - 1) / rsb::kScanStrip, need = (nstrips + rsb::kScanWavesPerGroup - 1) / rsb::kScanWavesPerGroup;
            const uint64_t most = (uint64_t)(grid_limit > 0 && grid_limit < scan_grid ? grid_limit : scan_grid);
            const uint32_t grid = (uint32_t)(need < most ? (need ? need : 1) : most);
            a.n_regions = grid * (uint32_t)rsb::kScanWavesPerGroup;
            // chunks of two strips once every wavefront has a long run of them (the last chunks bound the imbalance at the end of the
            // launch: two strips are ~35 us of 1.3 ms per Gsample), single strips for small batches
            a.chunk_strips = chunk_strips_forced ? chunk_strips_forced : (nstrips / a.n_regions >= 16 ? 2u : 1u);
            a.n_chunks = (uint32_t)((nstrips + a.chunk_strips - 1) / a.chunk_strips);
        }
        a.shared_cap = item_cap / 16;
        a.item_region = (item_cap - a.shared_cap) / a.n_regions / 64 * 64;  // whole keep masks
        a.chunk_dir = b.d_chunk_dir;
        a.chunk_kept = b.d_chunk_kept;
        a.chunk_group = b.d_chunk_group;
        a.region_counts = b.d_region_counts;
        a.region_kept = b.d_region_kept;
        a.region_masks = b.d_region_masks;
        a.mask_stride = a.item_region / 64;
        a.surv_item_cap = surv_item_cap;
        for (int w = 0; w < 2; ++w) {
            a.skeys[w] = b.d_skeys[w];
            a.sinfo[w] = b.d_sinfo[w];
        }
        a.surv_g = d_surv_g;
        a.survivors = d_surv;
        a.surv_cap = surv_cap;
        a.ac = b.d_ac;
        a.ac_cap = ac_cap;
        a.adder_bitmap = d_bitmap;
        a.adder_summary = (n >= summary_from && !exact_stats) ? d_summary : nullptr;  // (keep_all batches ask nothing)
        a.counters = b.d_counters;
        a.block_sums = b.d_block_sums;
        a.block_means = nullptr;
        a.emit_dead = exact_stats ? 1 : 0;
        a.keep_all = exact_stats ? 1 : 0;
        a.stage_limit = stage_limit;
        a.debug_flags = debug_flags;
        a.wave_times = d_wave_times;
        a.block_len = block_len;
        return a;
    }
};

// noalloc.hip: per-device set-up of rsb_convert_noalloc's fixed work memory (the device must be current)
int rsb_internal_noalloc_setup(int device);
// internal entry points behind include/readsb_dropin.h (dropin.cpp)
int rsb_internal_create_block_ctx(int fix_bits, int device, unsigned max_length, rsb_ctx **out);
// one struct mag_buf: mag_dev = 326 overlap samples + length samples, already on the device
int rsb_internal_submit_block(rsb_ctx *c, const uint16_t *mag_dev, unsigned length, bool modes, bool modeac, double mean_level, double mean_power);
