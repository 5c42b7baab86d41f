// capi.cpp -- the C ABI of include/readsb_amd.h over the HIP kernels and the host replay.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <string>
#include <vector>

#include "ctx.hpp"
#include "wire.hpp"

using namespace rsb;

namespace {

bool fail(rsb_ctx *c, const char *what, hipError_t e) {
    char buf[256];
    std::snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    if (c) c->err = buf;
    return false;
}

#define HIP_TRY(ctx, call)                      \
    do {                                        \
        hipError_t e_ = (call);                 \
        if (e_ != hipSuccess) {                 \
            fail(ctx, #call, e_);               \
            return RSB_ENODEV;                  \
        }                                       \
    } while (0)

template <typename T>
int upload(rsb_ctx *c, T **dst, const T *src, size_t count) {
    HIP_TRY(c, hipMalloc((void **)dst, count * sizeof(T)));
    HIP_TRY(c, hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
    return RSB_OK;
}

int setup_device(rsb_ctx *c) {
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        c->err = "no HIP device available (this library has no CPU fallback)";
        return RSB_ENODEV;
    }
    if (c->cfg.device < 0 || c->cfg.device >= ndev) {
        c->err = "rsb_config.device out of range";
        return RSB_EINVAL;
    }
    HIP_TRY(c, hipSetDevice(c->cfg.device));

    int rc;
    if ((rc = rsb_internal_noalloc_setup(c->cfg.device)) != RSB_OK) {
        c->err = "set-up of rsb_convert_noalloc's work memory failed (see stderr)";
        return rc;
    }
    if ((rc = upload(c, &c->d_lut, uc8_magnitude_lut(), 65536)) != RSB_OK) return rc;
    if ((rc = upload(c, &c->d_radial, uc8_radial_lut(), (size_t)kRadialTotal)) != RSB_OK) return rc;
    if ((rc = upload(c, &c->d_crc_bytes, CrcTables::byte_table(), 256)) != RSB_OK) return rc;
    if ((rc = upload(c, &c->d_sbs, CrcTables::single_bit_syndromes(), 112)) != RSB_OK) return rc;
    for (int w = 0; w < 2; ++w) {
        DevSyndromeTable t = c->replay.crc().device_table(w == 0 ? 56 : 112);
        c->syn_mask[w] = t.mask;
        if (c->replay.crc().table(w == 0 ? 56 : 112).empty()) continue;  // --no-fix: no table (crc.c:358-360)
        if ((rc = upload(c, &c->d_syn_keys[w], t.keys.data(), t.keys.size())) != RSB_OK) return rc;
        if ((rc = upload(c, &c->d_syn_vals[w], t.vals.data(), t.vals.size())) != RSB_OK) return rc;
    }

    const uint64_t nmax = c->cfg.max_batch_samples;
    const uint32_t div = c->cfg.queue_div ? c->cfg.queue_div : 32;
    uint64_t cap = nmax / div;
    if (cap < 4096) cap = 4096;
    if (cap > 0x7fffffffull) cap = 0x7fffffffull;
    // scan output: wave-private regions need head-room for uneven traffic (4x the kept-item capacity)
    c->surv_item_cap = (uint32_t)cap;
    c->item_cap = (uint32_t)(cap * 4 > 0x7fffffffull ? 0x7fffffffull : cap * 4);
    if (c->item_cap < 65536) c->item_cap = 65536;
    c->surv_cap = (uint32_t)(cap / 4 < 4096 ? 4096 : cap / 4);
    c->exact_stats = (c->cfg.flags & RSB_FLAG_EXACT_STATS) != 0;
    c->defer_prune = (c->cfg.flags & RSB_FLAG_DEFER_PRUNE) != 0;
    if (c->exact_stats) c->surv_cap = (uint32_t)cap;  // every gated offset (about 1 % of the samples on noise) becomes a candidate record
    c->ac_cap = (uint32_t)(cap / 4 < 4096 ? 4096 : cap / 4);
    c->nblocks_cap = (nmax + c->block_len - 1) / c->block_len + 1;
    c->scan_grid = scan_grid_size(c->cfg.device);
    if (const char *e = std::getenv("RSB_STAGE_LIMIT")) c->stage_limit = std::atoi(e);  // ablation timing only
    if (const char *e = std::getenv("RSB_ABL")) c->debug_flags = std::atoi(e);
    if (const char *e = std::getenv("RSB_SUMMARY_FROM")) c->summary_from = std::strtoull(e, nullptr, 10);  // tests: 0 = always
    if (const char *e = std::getenv("RSB_FORCE_SORT")) c->force_sort = std::atoi(e) != 0;  // A/B and tests: the atomic prune + radix sort
    if (const char *e = std::getenv("RSB_SCAN_GRID")) c->scan_grid = std::atoi(e);
    if (c->cfg.scan_workgroups && (int)c->cfg.scan_workgroups < c->scan_grid) c->scan_grid = (int)c->cfg.scan_workgroups;
    if (c->scan_grid < 1) c->scan_grid = 1;
    if (const char *e = std::getenv("RSB_CHUNK_STRIPS")) c->chunk_strips_forced = (uint32_t)std::atoi(e);
    c->chunks_cap = (size_t)((nmax + 2 + kScanStrip - 1) / kScanStrip) + 1;  // chunks of one strip: the most a batch can have
    if (std::getenv("RSB_WAVE_TIMES")) {  // measurement runs only (tools/wave_times.py)
        HIP_TRY(c, hipMalloc((void **)&c->d_wave_times, (size_t)c->scan_grid * kScanWavesPerGroup * 4 * sizeof(uint64_t)));
        HIP_TRY(c, hipMemset(c->d_wave_times, 0, (size_t)c->scan_grid * kScanWavesPerGroup * 4 * sizeof(uint64_t)));
    }
    // --dcfilter (and the SC16Q11 table converter) keeps one magnitude buffer, which the replay of a batch still reads: one batch in flight there
    c->q11_table_bits = (int)RSB_FLAG_SC16Q11_TABLE_BITS_OF(c->cfg.flags);
    // ... unless the kernels filter for themselves (Mode S only, no exact statistics: those two take the converter's
    // block means): then a batch's filter states live in its slot and batches overlap as without the filter
    c->dc_fused = c->cfg.dc_filter && !c->q11_table_bits && !c->run_ac && !c->exact_stats && !std::getenv("RSB_DC_UNFUSED");
    c->nslots = c->dc_fused ? 2 : ((c->cfg.dc_filter || c->q11_table_bits) ? 1 : rsb_ctx::kMaxSlots);
    for (int k = 0; k < c->nslots; ++k) {
        rsb_slot &b = c->slots[k];
        HIP_TRY(c, hipMalloc((void **)&b.d_items, (size_t)c->item_cap * sizeof(ItemRec)));
        HIP_TRY(c, hipMalloc((void **)&b.d_region_counts, (size_t)c->scan_grid * kScanWavesPerGroup * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc((void **)&b.d_region_kept, (size_t)c->scan_grid * kScanWavesPerGroup * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc((void **)&b.d_region_masks, ((size_t)c->item_cap / 64 + (size_t)c->scan_grid * kScanWavesPerGroup) * sizeof(uint64_t)));
        HIP_TRY(c, hipMalloc((void **)&b.d_chunk_dir, c->chunks_cap * sizeof(uint2)));
        HIP_TRY(c, hipMalloc((void **)&b.d_chunk_kept, c->chunks_cap * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc((void **)&b.d_chunk_group, (c->chunks_cap / 256 + 2) * sizeof(uint32_t)));
        for (int w = 0; w < 2; ++w) {
            HIP_TRY(c, hipMalloc((void **)&b.d_skeys[w], (size_t)c->surv_item_cap * sizeof(uint64_t)));
            HIP_TRY(c, hipMalloc((void **)&b.d_sinfo[w], (size_t)c->surv_item_cap * sizeof(uint32_t)));
        }
        HIP_TRY(c, hipMalloc((void **)&b.d_ac, (size_t)c->ac_cap * sizeof(AcCand)));
        // counters, the device replay's per-block message counts and the batch's new filter members lie in ONE block: what the host
        // wants to see of a replayed batch crosses PCIe in one copy (three before)
        HIP_TRY(c, hipMalloc((void **)&b.d_counters, (c->back_adders() + 2 * (size_t)rsb_ctx::kNewAddersCap) * sizeof(uint32_t)));
        b.d_blk_counts = b.d_counters + c->back_blk();
        b.d_new_adders = (uint2 *)(b.d_counters + c->back_adders());
        HIP_TRY(c, hipMalloc((void **)&b.d_block_sums, c->nblocks_cap * 2 * sizeof(uint64_t)));
        HIP_TRY(c, hipMalloc((void **)&b.d_block_means, c->nblocks_cap * 2 * sizeof(double)));
        for (hipEvent_t &ev : b.ev) HIP_TRY(c, hipEventCreate(&ev));
        for (hipEvent_t &ev : b.rev) HIP_TRY(c, hipEventCreate(&ev));
    }
    HIP_TRY(c, hipStreamCreateWithFlags(&c->replay_stream, hipStreamNonBlocking));
    HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->copy_done, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->reset_done, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->tables_ev, hipEventDisableTiming));
    for (void *&p : c->d_carry) HIP_TRY(c, hipMalloc(&p, kTrailing * 4));
    c->sort_temp_bytes = sort_temp_bytes(c->surv_item_cap);
    HIP_TRY(c, hipMalloc(&c->d_sort_temp, c->sort_temp_bytes));
    HIP_TRY(c, hipMalloc((void **)&c->d_surv_g, (size_t)c->surv_item_cap * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc((void **)&c->d_surv, (size_t)c->surv_cap * sizeof(Survivor)));
    HIP_TRY(c, hipMalloc((void **)&c->d_bitmap, (1u << 24) / 8));
    HIP_TRY(c, hipMalloc((void **)&c->d_summary, kSummaryWords * sizeof(uint32_t)));
    if (c->mean_sub) HIP_TRY(c, hipMalloc((void **)&c->d_sub_means, ((size_t)(nmax / c->mean_sub) + 2) * 2 * sizeof(double)));
    HIP_TRY(c, hipMalloc((void **)&c->d_dc_state, sizeof(double2)));
    HIP_TRY(c, hipMemset(c->d_dc_state, 0, sizeof(double2)));
    if ((c->cfg.dc_filter && !c->dc_fused) || c->q11_table_bits) HIP_TRY(c, hipMalloc((void **)&c->d_magbuf, (size_t)nmax * sizeof(uint16_t) + 16));
    if (c->cfg.dc_filter) {
        c->dc_tiles_cap = (size_t)((nmax + kDcTileSamples - 1) / kDcTileSamples) + 1;
        HIP_TRY(c, hipMalloc((void **)&c->d_dc_tile_sums, c->dc_tiles_cap * sizeof(double2)));
        HIP_TRY(c, hipMalloc((void **)&c->d_dc_tile_start, c->dc_tiles_cap * sizeof(double2)));
    }
    if (c->dc_fused) {
        double pw[512];
        dc_pow_table(pw);
        if ((rc = upload(c, &c->d_dc_pow, pw, 512)) != RSB_OK) return rc;
        for (int k = 0; k < c->nslots; ++k) {
            HIP_TRY(c, hipMalloc((void **)&c->slots[k].d_dc_local, c->dc_tiles_cap * (kDcTileSamples / 8) * sizeof(float2)));
            HIP_TRY(c, hipMalloc((void **)&c->slots[k].d_dc_tile_start, c->dc_tiles_cap * sizeof(double2)));
        }
    }
    c->host_replay_only = (c->cfg.flags & (RSB_FLAG_HOST_REPLAY | RSB_FLAG_EXACT_STATS)) != 0;
    if (!c->host_replay_only) {
        for (int w = 0; w < 2; ++w) HIP_TRY(c, hipMalloc((void **)&c->d_first_add[w], sizeof(uint32_t) << 24));
        // the message records of the device replay: as many as the candidate records' memory would hold
        c->out_cap = (uint32_t)((size_t)c->surv_cap * sizeof(Survivor) / sizeof(rsb_message));
        for (int k = 0; k < c->nslots; ++k) {
            rsb_slot &b = c->slots[k];
            HIP_TRY(c, hipMalloc((void **)&b.d_out, (size_t)c->out_cap * sizeof(rsb_message)));
            HIP_TRY(c, hipMalloc((void **)&b.d_blk_start, (c->nblocks_cap + 1) * sizeof(uint32_t)));
            HIP_TRY(c, hipMalloc((void **)&b.d_blk_base, (c->nblocks_cap + 1) * sizeof(uint32_t)));
            HIP_TRY(c, hipHostMalloc((void **)&b.h_pin, (1 + c->back_adders() + 2 * rsb_ctx::kPinAdders) * sizeof(uint32_t), hipHostMallocDefault));
        }
        HIP_TRY(c, hipMalloc((void **)&c->d_addr_upload, 8 * IcaoFilter::kSlots * sizeof(uint32_t)));  // 2 x kSlots members, up to 6 x kSlots stale entries
        HIP_TRY(c, hipHostMalloc((void **)&c->h_addr_upload, 8 * IcaoFilter::kSlots * sizeof(uint32_t), hipHostMallocDefault));
        HIP_TRY(c, hipEventCreateWithFlags(&c->addr_upload_done, hipEventDisableTiming));
    }
    return RSB_OK;
}

int mirror_filter(rsb_ctx *c, hipStream_t s);
int order_tables(rsb_ctx *c, hipStream_t s);
int tables_touched(rsb_ctx *c, hipStream_t s);

int reset_device(rsb_ctx *c) {
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    // On the context's own stream, without a host wait (a blocking memset and copy were 45 us per reset: 1.4 ms for 32 channels):
    // whatever reads or writes the bitmap next is either on that stream (prune, rsb_superset_add, rsb_export_adders) or waits for
    // reset_done first (the next scan: enqueue_before_scan; fetch_bitmap).
    HIP_TRY(c, hipMemsetAsync(c->d_bitmap, 0, (1u << 24) / 8, c->replay_stream));
    // icaoFilterTest(0) is true on an empty slot (icao_filter.c:101-107): address 0 is always a "member"
    HIP_TRY(c, hipMemsetD32Async((hipDeviceptr_t)c->d_bitmap, 1, 1, c->replay_stream));
    if (c->d_dc_state) HIP_TRY(c, hipMemsetAsync(c->d_dc_state, 0, sizeof(double2), c->replay_stream));  // a new recording: the DC block starts from zero, as init_converter leaves it
    HIP_TRY(c, hipEventRecord(c->reset_done, c->replay_stream));
    c->reset_pending = true;
    c->filter_dirty = true;
    // The first-add tables follow the (now empty) filter here and now, on the same stream, instead of at the head of the next batch's
    // chain: the entries the old recording left are few and listed, and the chain is what a step waits for.
    if (c->d_first_add[0] && c->tables_valid && c->table_members.size() <= 6 * IcaoFilter::kSlots) {
        int rc = order_tables(c, c->replay_stream);
        if (rc == RSB_OK) rc = mirror_filter(c, c->replay_stream);
        if (rc == RSB_OK) rc = tables_touched(c, c->replay_stream);
        if (rc != RSB_OK) return rc;
    }
    return RSB_OK;
}

// first-add tables := the host filter's contents (both generations, icao_filter.c:97-119)
int mirror_filter(rsb_ctx *c, hipStream_t s) {
    c->replay.filter().contents(c->h_addrs);
    const uint32_t n = (uint32_t)c->h_addrs.size(), n_stale = (uint32_t)c->table_members.size();
    uint32_t *members_dev = c->d_addr_upload, *stale_dev = c->d_addr_upload + 2 * IcaoFilter::kSlots;
    // through pinned staging: no host wait here, so the rest of the batch's chain is enqueued while the scan still runs (with a
    // wait in this place the dozen short kernels behind it were launched one by one onto an idle GPU: ~0.1 ms per batch)
    uint32_t *members_pin = c->h_addr_upload, *stale_pin = c->h_addr_upload + 2 * IcaoFilter::kSlots;
    HIP_TRY(c, hipEventSynchronize(c->addr_upload_done));  // (the previous upload: long done)
    if (n) {
        std::memcpy(members_pin, c->h_addrs.data(), n * sizeof(uint32_t));
        HIP_TRY(c, hipMemcpyAsync(members_dev, members_pin, n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    }
    if (c->tables_valid && n_stale <= 6 * IcaoFilter::kSlots) {
        // the tables only differ from "never" at the listed entries: undo those, set the present members
        if (n_stale) {
            std::memcpy(stale_pin, c->table_members.data(), n_stale * sizeof(uint32_t));
            HIP_TRY(c, hipMemcpyAsync(stale_dev, stale_pin, n_stale * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        }
        HIP_TRY(c, launch_filter_update(c->d_first_add[0], c->d_first_add[1], stale_dev, n_stale, members_dev, n, s));
    } else {
        HIP_TRY(c, launch_filter_rebuild(c->d_first_add[0], c->d_first_add[1], members_dev, n, s));
    }
    HIP_TRY(c, hipEventRecord(c->addr_upload_done, s));
    c->table_members = c->h_addrs;
    c->tables_valid = true;
    c->first_cur = 0;
    c->filter_dirty = false;
    return RSB_OK;
}

constexpr int kMaxReplayIterations = 16;

bool emit_direct_enabled() {
    static const bool on = std::getenv("RSB_EMIT_DIRECT") && std::atoi(std::getenv("RSB_EMIT_DIRECT")) != 0;
    return on;
}

// The filter-superset prune of a scanned batch.  Normally the ordered compaction (the scan's item regions are sorted runs in
// stream order: no sort needed afterwards); the atomic prune + radix sort remain for batches with marker items (exact
// statistics) and are fallen back to at collect time when a scan wavefront overflowed one of its LDS queues (then some items
// sit in the shared region, out of order).
bool prune_is_ordered(const rsb_ctx *c, const BatchArgs &a) { return !a.emit_dead && !c->force_sort; }

int enqueue_prune(rsb_ctx *c, rsb_slot &b, const BatchArgs &a, hipStream_t s) {
    b.pruned = true;
    b.pruned_in_order = prune_is_ordered(c, a);
    if (b.pruned_in_order)
        HIP_TRY(c, launch_prune_ordered(a, s));
    else
        HIP_TRY(c, launch_prune(a, s));
    return RSB_OK;
}

// ---- the sequential half on the device, in two steps so that the chain of batch k + 1 can be put on the replay stream while the
// host still waits for batch k's records to leave (collect_ahead).

// the readbacks of one replay round's results into the slot's pinned block: [changed | counters | block counts | first new adders]
int enqueue_readbacks(rsb_ctx *c, rsb_slot &b, const ReplayArgs &r, hipStream_t s) {
    uint32_t *pin = b.h_pin;
    (void)r;
    HIP_TRY(c, hipMemcpyAsync(pin + 1, b.d_counters, (c->back_adders() + 2 * rsb_ctx::kPinAdders) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    return RSB_OK;
}

// Step 1: everything of the first round -- table mirror if needed, block bounds, optimistic seed, score / walk / adds / compare,
// the places of the accepted messages, their records, the readbacks -- goes onto stream s; nothing is waited for.
// n_items: the number of kept items, or -- count_on_device -- the host's bound on it (the kernels then read the count the prune
// left in the batch's counters, and replay_finish checks what the host could not know beforehand).
// direct_out (optional, RSB_EMIT_DIRECT=1): the caller's record buffer.  If the device can address it (pinned host memory), the
// emit kernel writes the records there itself and no device -> host copy follows.
// The first-add tables are worked on by one batch after the other, but not always on one stream (the context's own; ctx[0]'s for
// rsb_collect_many; the caller's for rsb_collect_prepare_on): whoever touches them next on another stream waits on the device.
int order_tables(rsb_ctx *c, hipStream_t s) {
    if (c->tables_stream && c->tables_stream != s) HIP_TRY(c, hipStreamWaitEvent(s, c->tables_ev, 0));
    return RSB_OK;
}
int tables_touched(rsb_ctx *c, hipStream_t s) {
    HIP_TRY(c, hipEventRecord(c->tables_ev, s));
    c->tables_stream = s;
    return RSB_OK;
}

// the replay's arguments for the slot's pending batch (the tables t_cur / t_next are the caller's to set: the mirror may swap them)
void fill_replay_args(rsb_ctx *c, rsb_slot &b, uint32_t n_items, bool count_on_device) {
    const BatchArgs &a = b.pending_args;
    ReplayArgs &r = b.chain_r;
    std::memset(&r, 0, sizeof(r));
    r.keys = a.skeys[c->sorted_buf];
    r.info = a.sinfo[c->sorted_buf];
    r.n_items = n_items;
    r.n_dev = count_on_device ? b.d_counters + kCntSurvItems : nullptr;
    r.n_cap = c->surv_item_cap;
    r.counters = b.d_counters;
    r.acc = a.sinfo[c->sorted_buf ^ 1];
    r.acc_rank = (uint32_t *)a.skeys[c->sorted_buf ^ 1];  // the spare key buffer holds 2 x 32 bits per item: the accepted items' ranks and cand
    r.cand = r.acc_rank + c->surv_item_cap;
    r.acc_idx = r.cand;
    r.blk_base = b.d_blk_base;
    r.blk_counts = b.d_blk_counts;
    r.blk_start = b.d_blk_start;
    r.nblocks = (uint32_t)((b.pending_n + c->block_len - 1) / c->block_len);
    r.block_len = c->block_len;
    r.new_adders = b.d_new_adders;
    r.new_adders_cap = rsb_ctx::kNewAddersCap;
    r.out = b.d_out;
    r.out_cap = c->out_cap;
    r.g0 = b.g0;
    b.chain_enqueued = true;
    b.chain_count_on_device = count_on_device;
    b.chain_to_host = false;
    b.chain_folded = false;
    b.time_from = nullptr;
}

int replay_enqueue(rsb_ctx *c, rsb_slot &b, uint32_t n_items, bool count_on_device, hipStream_t s, rsb_message *direct_out = nullptr, size_t direct_cap = 0) {
    const BatchArgs &a = b.pending_args;
    int rc;
    if ((rc = order_tables(c, s)) != RSB_OK) return rc;
    if (c->filter_dirty && (rc = mirror_filter(c, s)) != RSB_OK) return rc;
    fill_replay_args(c, b, n_items, count_on_device);
    b.chain_on = s;
    ReplayArgs &r = b.chain_r;
    // A record copy that rsb_collect_begin left in flight reads the d_out of ITS slot; this chain writes this slot's.  The same
    // slot only on one-slot contexts (dc_filter, SC16Q11 table): begin(k), submit(k + 1), prepare(k + 1) -- the chain then waits
    // on the device for the copy, the host does not.
    if (c->copy_in_flight && c->copy_slot == &b) HIP_TRY(c, hipStreamWaitEvent(s, c->copy_done, 0));
    HIP_TRY(c, hipEventRecord(b.rev[1], s));
    if (r.nblocks == 0) {
        HIP_TRY(c, hipEventRecord(b.rev[2], s));
        return RSB_OK;
    }
    r.t_cur = c->d_first_add[c->first_cur];  // seeded with the optimistic "first added at" entries (k_replay_table_min<true>)
    r.t_next = c->d_first_add[c->first_cur ^ 1];
    HIP_TRY(c, launch_replay_bounds(r, s));
    // Records straight into the caller's buffer when that is pinned host memory the device can address (hipHostMalloc /
    // hipHostRegister, e.g. torch's pin_memory): the emit kernel's stores then are the device -> host transfer.
    // (measured on MI355X: a DMA copy of the finished list moves 41 MB at 56 GB/s beside a running scan kernel without slowing it,
    // the kernel's own stores over PCIe reach 50 GB/s and hold CUs meanwhile -- so the direct path is opt-in: RSB_EMIT_DIRECT=1)
    if (emit_direct_enabled() && direct_out && direct_cap) {
        const uint32_t hcap = (uint32_t)(direct_cap < 0xffffffffull ? direct_cap : 0xffffffffull);
        hipPointerAttribute_t lo_attr, hi_attr;
        const bool lo_ok = hipPointerGetAttributes(&lo_attr, direct_out) == hipSuccess && lo_attr.type == hipMemoryTypeHost && lo_attr.devicePointer;
        const bool hi_ok = lo_ok && hipPointerGetAttributes(&hi_attr, (const char *)(direct_out + hcap) - 1) == hipSuccess &&
                           hi_attr.type == hipMemoryTypeHost && hi_attr.devicePointer;
        if (!lo_ok || !hi_ok) (void)hipGetLastError();  // pageable memory: not an error, the copy path takes it
        if (hi_ok && (const char *)hi_attr.devicePointer - (const char *)lo_attr.devicePointer == (ptrdiff_t)((size_t)hcap * sizeof(rsb_message) - 1)) {
            r.host_out = (rsb_message *)lo_attr.devicePointer;
            r.host_cap = hcap;
            b.chain_to_host = true;
        }
    }
    // The first round is followed at once by the list and the records (the optimistic start is nearly always the fixed
    // point): one wait for "did the table move" and the counters together.  If it did move, those records are discarded,
    // the iteration goes on and they are written again.
    // ("did the table move" is the list kernel's answer and comes back with the counters; if it did not, the record kernel folds
    // the new members into the tables as it goes -- complete when every accepted message got its record, which the host checks)
    r.fold_with_records = 1;
    HIP_TRY(c, launch_replay_first_round(a, r, s));
    HIP_TRY(c, launch_replay_emit(a, r, s));
    if ((rc = enqueue_readbacks(c, b, r, s)) != RSB_OK) return rc;
    HIP_TRY(c, hipEventRecord(b.rev[2], s));
    return RSB_OK;
}

// Step 2: wait for the chain, look at what came back, go on iterating in the rare case that the optimistic table was not the
// fixed point, fold the batch's new members into the tables and the host's filter.
// Returns 1 when the batch was replayed on the device (messages in b.d_out as rsb_message records, counters in cnt), 0 when the
// host path has to take it, -1000 when the chain ran on a count the host must not trust (cnt[] says why), < 0 on error.
int replay_finish(rsb_ctx *c, rsb_slot &b, hipStream_t s, uint32_t cnt[kCntCount], int *direct = nullptr) {
    const BatchArgs &a = b.pending_args;
    ReplayArgs &r = b.chain_r;
    b.chain_enqueued = false;
    c->timing.replay_iterations = 0;
    std::memset(cnt + kCntChanged, 0, (kCntGeneralStrips - kCntChanged) * sizeof(uint32_t));  // the replay's own counters
    if (r.nblocks == 0) return 1;
    uint32_t *pin = b.h_pin;
    bool converged = false, emitted = false;
    for (int it = 0; it < kMaxReplayIterations && !converged; ++it) {
        if (it > 0) {
            r.t_cur = c->d_first_add[c->first_cur];
            r.t_next = c->d_first_add[c->first_cur ^ 1];
            HIP_TRY(c, launch_replay_iteration(a, r, s));
            pin[0] = 0;
            HIP_TRY(c, hipMemcpyAsync(&pin[0], b.d_counters + kCntChanged, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        }
        // (a chain on the caller's stream: whatever the caller queued behind it -- the next scan -- is not waited for)
        if (it == 0 && b.chain_foreign)
            HIP_TRY(c, hipEventSynchronize(b.rev[2]));
        else
            HIP_TRY(c, hipStreamSynchronize(s));
        const uint32_t changed = it == 0 ? pin[1 + kCntChanged] : pin[0];
        if (it == 0) {
            std::memcpy(cnt, pin + 1, kCntCount * sizeof(uint32_t));
            std::memcpy(c->h_blk_counts.data(), pin + 1 + c->back_blk(), r.nblocks * sizeof(uint32_t));
            if (b.chain_count_on_device) {
                // now the host sees what the prune left: anything the kernels' clamps papered over ends the batch here
                if (cnt[kCntOverflow] || cnt[kCntSurvItems] > c->surv_item_cap || cnt[kCntItems] != 0) {
                    c->filter_dirty = true;
                    c->tables_valid = false;  // the round above ran on a list that does not count
                    return -1000;             // the caller looks at cnt[]
                }
                r.n_items = cnt[kCntSurvItems];  // exact from here on
                r.n_dev = nullptr;
            }
        }
        ++c->timing.replay_iterations;
        if (!changed) {
            converged = true;
            emitted = it == 0;
            if (emitted && r.fold_with_records) b.chain_folded = cnt[kCntAccepted] <= r.out_cap && (!r.host_out || cnt[kCntAccepted] <= r.host_cap);
        } else {
            // the table this walk produced becomes the current one (a chain's first round has not built it: now); the other goes
            // back to "before this batch"
            if (it == 0) HIP_TRY(c, launch_replay_next_table(r, s));
            c->first_cur ^= 1;
            HIP_TRY(c, launch_replay_reset(a, r, c->d_first_add[c->first_cur ^ 1], s));
        }
    }
    if (!converged) {
        c->filter_dirty = true;
        c->tables_valid = false;  // an iteration's intermediate values are in them
        return 0;
    }
    if (!emitted) {
        int rc;
        r.fold_with_records = 0;  // (launch_replay_fold below)
        HIP_TRY(c, launch_replay_select(a, r, c->d_sort_temp, c->sort_temp_bytes, s));
        HIP_TRY(c, launch_replay_emit(a, r, s));
        if ((rc = enqueue_readbacks(c, b, r, s)) != RSB_OK) return rc;
        HIP_TRY(c, hipStreamSynchronize(s));
        std::memcpy(cnt, pin + 1, kCntCount * sizeof(uint32_t));
        std::memcpy(c->h_blk_counts.data(), pin + 1 + c->back_blk(), r.nblocks * sizeof(uint32_t));
    }
    // the batch's new members become "in the filter before the next batch" (asynchronous; a chain for many channels has done it
    // on the device for the channels whose first round was the fixed point)
    if (!(emitted && b.chain_folded)) HIP_TRY(c, launch_replay_fold(r, s));
    // The tables answer "is this address in the filter" as a set.  The reference's open-addressing table does
    // the same until it is full (icao_filter.c:76-79, 101-106): stay well clear of that, else replay on the host.
    const uint32_t nnew = cnt[kCntNewAdders];
    if (nnew > rsb_ctx::kNewAddersCap || c->replay.filter().active_slots_used() + 2 * (uint64_t)nnew + 1 >= IcaoFilter::kSlots) {
        c->filter_dirty = true;
        c->tables_valid = false;  // entries folded to 0 that the host does not know of
        return 0;
    }
    // with chain_to_host the emit kernel has written the first min(accepted, host_cap) records into the caller's buffer: they are
    // delivered (rsb_collect's contract for a short buffer: the first cap records and RSB_EOVERFLOW); nothing of this batch is in r.out
    if (direct) *direct = b.chain_to_host ? 1 : 0;
    if (nnew) {
        c->h_new_adders.resize(nnew);
        if (nnew <= rsb_ctx::kPinAdders)  // came along with the counters
            std::memcpy(c->h_new_adders.data(), pin + 1 + c->back_adders(), nnew * sizeof(uint2));
        else
            HIP_TRY(c, hipMemcpy(c->h_new_adders.data(), b.d_new_adders, nnew * sizeof(uint2), hipMemcpyDeviceToHost));
        std::sort(c->h_new_adders.begin(), c->h_new_adders.end(), [](const uint2 &x, const uint2 &y) { return x.x < y.x; });
        for (const uint2 &e : c->h_new_adders) {
            c->replay.filter().add(e.y);  // in the order the reference learns them
            c->table_members.push_back(e.y);  // k_replay_fold has set its entries to 0
        }
    }
    return 1;
}

// which batches need no word from the device before their replay chain is enqueued
bool chain_needs_no_host(const rsb_ctx *c, const rsb_slot &b) {
#ifdef RSB_STAGE_CLOCKS
    return false;
#else
    const BatchArgs &a = b.pending_args;
    return c->run_modes && !c->block_mode && (b.pruned ? b.pruned_in_order : prune_is_ordered(c, a)) && !c->host_replay_only && !c->defer_prune && !a.emit_dead &&
           !a.keep_all && !emit_direct_enabled() &&
           !(c->debug_flags & 256);
#endif
}

uint32_t kept_items_bound(const rsb_ctx *c, const rsb_slot &b) {  // sizes the grids only: the kernels loop over whatever the count turns out to be
    uint64_t bound = b.pending_n / 128;
    if (bound < 4096) bound = 4096;
    if (bound > c->surv_item_cap) bound = c->surv_item_cap;
    return (uint32_t)bound;
}

// While the host waits for batch k's records to leave, the replay chain of the next scanned batch goes onto the replay stream
// (its scan may still be running: the chain waits for its event).  Only when nothing forces the host to look first.
void collect_ahead(rsb_ctx *c) {
    if (!c->any_pending() || c->filter_dirty) return;
    rsb_slot &nb = c->collect_slot();
    if (!nb.pending || nb.chain_enqueued || !chain_needs_no_host(c, nb)) return;
    hipStream_t s = c->replay_stream;
    if (hipStreamWaitEvent(s, nb.ev[3], 0) != hipSuccess) return;
    if (hipEventRecord(nb.rev[0], s) != hipSuccess) return;
    if (!nb.pruned && enqueue_prune(c, nb, nb.pending_args, s) != RSB_OK) return;
    c->sorted_buf = 0;
    if (replay_enqueue(c, nb, kept_items_bound(c, nb), true, s) != RSB_OK) nb.chain_enqueued = false;  // collect will try again and report
}

// A change of the filter or of the prune's superset from outside while the next batch's chain is already on the replay stream
// (collect_ahead / rsb_collect_prepare put it there): that chain ran against the old state.  Drop it -- the collect then prunes and
// replays the batch again, against the state as it is when the batch is collected, which is what a caller that changes the
// filter between two collects means.
int drop_enqueued_chain(rsb_ctx *c) {
    if (!c->any_pending()) return RSB_OK;
    rsb_slot &nb = c->collect_slot();
    if (!nb.pending || !nb.chain_enqueued) return RSB_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, hipStreamSynchronize(nb.chain_foreign ? nb.chain_on : c->replay_stream));
    nb.chain_enqueued = false;
    nb.chain_foreign = false;
    nb.pruned = false;  // the superset may have grown: the prune runs again at collect time (it rebuilds everything it writes)
    c->filter_dirty = true;
    c->tables_valid = false;  // the dropped round's seed entries are in the tables
    return RSB_OK;
}

// a caller's address list on the device (one word of room in front of it for a kernel's counter)
int upload_addrs(rsb_ctx *c, const uint32_t *addrs, size_t n, uint32_t **dev) {
    if ((n + 1) * sizeof(uint32_t) > c->addr_list_cap) {
        if (c->d_addr_list) (void)hipFree(c->d_addr_list);
        c->d_addr_list = nullptr;
        c->addr_list_cap = 0;
        const size_t bytes = (n + 1) * sizeof(uint32_t) * 2 + 4096;
        HIP_TRY(c, hipMalloc((void **)&c->d_addr_list, bytes));
        c->addr_list_cap = bytes;
    }
    if (n) HIP_TRY(c, hipMemcpy(c->d_addr_list + 1, addrs, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    *dev = c->d_addr_list + 1;
    return RSB_OK;
}

// bits of the prune's superset set where it lies (the scans of pending batches set theirs with atomics too: nothing to wait for);
// on the replay stream, i.e. in front of the prune of whatever is collected next
int superset_set_bits(rsb_ctx *c, const uint32_t *addrs, size_t n) {
    uint32_t *dev = nullptr;
    const int rc = upload_addrs(c, addrs, n, &dev);
    if (rc != RSB_OK) return rc;
    HIP_TRY(c, launch_bitmap_set(c->d_bitmap, dev, (uint32_t)n, c->replay_stream));
    HIP_TRY(c, hipStreamSynchronize(c->replay_stream));  // (the list buffer is reused by the next call)
    return RSB_OK;
}

const CrcTables &tables_for(int fix_bits) {
    static const CrcTables t0(0), t1(1), t2(2);
    return fix_bits <= 0 ? t0 : (fix_bits == 1 ? t1 : t2);
}

}  // namespace

extern "C" {

const char *rsb_version(void) { return "readsb_amd 0.1 (gfx950)"; }

int rsb_create(const rsb_config *cfg, rsb_ctx **out) {
    if (!cfg || !out) return RSB_EINVAL;
    *out = nullptr;
    if (cfg->input_format < RSB_UC8 || cfg->input_format > RSB_SC16Q11) {
        // convert.c:460-464
        std::fprintf(stderr, "no suitable converter for format=%d dc=%d\n", cfg->input_format, cfg->dc_filter);
        return RSB_EINVAL;
    }
    if (cfg->fix_bits < 0 || cfg->max_batch_samples == 0 || cfg->max_batch_samples > 0xfffff000ull) return RSB_EINVAL;
    if (const unsigned tb = RSB_FLAG_SC16Q11_TABLE_BITS_OF(cfg->flags)) {
        // convert.c:451-456: the table converter stands in the converter list for SC16Q11 without the DC filter only
        if (tb > 11 || cfg->input_format != RSB_SC16Q11 || cfg->dc_filter) {
            std::fprintf(stderr, "readsb_amd: RSB_FLAG_SC16Q11_TABLE(%u) needs input_format RSB_SC16Q11, no dc_filter and 1..11 bits\n", tb);
            return RSB_EINVAL;
        }
    }
    if (cfg->block_samples || cfg->mean_sub_samples) {
        // the buffer geometry of a hardware driver (sdr_bladerf.c:335-404): whole 16-byte pieces of IQ per buffer, at most MODES_MAG_BUF_SAMPLES
        const uint32_t L = cfg->block_samples ? cfg->block_samples : kBlock, sub = cfg->mean_sub_samples;
        const bool float_fmt = cfg->input_format == RSB_SC16 || cfg->input_format == RSB_SC16Q11;
        if (L > kBlock || L < 4096 || L % 8 != 0 || cfg->dc_filter || RSB_FLAG_SC16Q11_TABLE_BITS_OF(cfg->flags) ||
            (sub && (!float_fmt || L % sub != 0 || sub < 8))) {
            std::fprintf(stderr, "readsb_amd: rsb_config.block_samples must be a multiple of 8 in 4096..131072 (no dc_filter, no SC16Q11 table), "
                                 "mean_sub_samples a divisor of it (SC16 / SC16Q11 only)\n");
            return RSB_EINVAL;
        }
    }
    rsb_ctx *c = new (std::nothrow) rsb_ctx(*cfg);
    if (!c) return RSB_ENOMEM;
    int rc = setup_device(c);
    if (rc == RSB_OK) rc = reset_device(c);
    if (rc != RSB_OK) {
        std::fprintf(stderr, "readsb_amd: %s\n", c->err.c_str());
        rsb_destroy(c);
        return rc;
    }
    *out = c;
    return RSB_OK;
}

void rsb_destroy(rsb_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->copy_in_flight && c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->h_addr_upload) (void)hipHostFree(c->h_addr_upload);
    if (c->h_many_args) (void)hipHostFree(c->h_many_args);
    if (c->d_many_args) (void)hipFree(c->d_many_args);
    if (c->h_chain_args) (void)hipHostFree(c->h_chain_args);
    if (c->d_chain_args) (void)hipFree(c->d_chain_args);
    if (c->h_chain_back) (void)hipHostFree(c->h_chain_back);
    if (c->h_mirror) (void)hipHostFree(c->h_mirror);
    if (c->d_mirror) (void)hipFree(c->d_mirror);
    if (c->d_chain_back) (void)hipFree(c->d_chain_back);
    for (hipEvent_t ev : c->many_args_read)
        if (ev) (void)hipEventDestroy(ev);
    if (c->addr_upload_done) (void)hipEventDestroy(c->addr_upload_done);
    if (c->copy_done) (void)hipEventDestroy(c->copy_done);
    if (c->reset_done) (void)hipEventDestroy(c->reset_done);
    if (c->tables_ev) (void)hipEventDestroy(c->tables_ev);
    void *ptrs[] = {c->d_lut, c->d_radial, c->d_crc_bytes, c->d_sbs, c->d_syn_keys[0], c->d_syn_keys[1], c->d_syn_vals[0], c->d_syn_vals[1], c->d_sort_temp, c->d_surv_g, c->d_surv,
                    c->d_bitmap, c->d_summary, c->d_stage, c->d_mag, c->d_sums, c->d_first_add[0], c->d_first_add[1], c->d_addr_upload, c->d_means,
                    c->d_magbuf, c->d_sub_means, c->d_wave_times, c->d_dc_tile_sums, c->d_dc_tile_start, c->d_dc_state, c->d_wire_off, c->d_wire_out, c->d_merge, c->d_dec_in, c->d_decoded, c->d_addr_list};
    for (rsb_slot &b : c->slots) {
        void *sp[] = {b.d_out, b.d_blk_start, b.d_blk_base, b.d_items, b.d_region_counts, b.d_region_kept, b.d_region_masks, b.d_chunk_dir, b.d_chunk_kept, b.d_chunk_group, b.d_skeys[0], b.d_skeys[1], b.d_sinfo[0], b.d_sinfo[1], b.d_ac, b.d_counters, b.d_block_sums, b.d_block_means};
        for (void *p : sp)
            if (p) (void)hipFree(p);
        for (hipEvent_t ev : b.ev)
            if (ev) (void)hipEventDestroy(ev);
        for (hipEvent_t ev : b.rev)
            if (ev) (void)hipEventDestroy(ev);
        if (b.h_pin) (void)hipHostFree(b.h_pin);
    }
    if (c->replay_stream) (void)hipStreamDestroy(c->replay_stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (rsb_slot &b : c->slots) {
        if (b.d_dc_local) (void)hipFree(b.d_dc_local);
        if (b.d_dc_tile_start) (void)hipFree(b.d_dc_tile_start);
    }
    if (c->d_dc_pow) (void)hipFree(c->d_dc_pow);
    if (c->d_dc_mags) (void)hipFree(c->d_dc_mags);
    for (void *p : c->d_carry)
        if (p) (void)hipFree(p);
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete c;
}

int rsb_reset(rsb_ctx *c) {
    if (!c) return RSB_EINVAL;
    (void)rsb_collect_wait(c);
    if (c->any_pending()) {  // a batch still in flight: wait for its kernels, then forget it
        (void)hipSetDevice(c->cfg.device);
        (void)hipDeviceSynchronize();
    }
    c->replay.reset();
    c->stream_pos = c->submit_pos = 0;
    c->submitted = c->collected = 0;
    for (rsb_slot &b : c->slots) {
        b.pending = false;
        b.chain_enqueued = false;
        b.chain_foreign = false;
    }
    for (int &v : c->carry_valid) v = 0;
    c->finished = c->submit_finished = false;
    c->last_n = 0;
    std::memset(&c->timing, 0, sizeof(c->timing));
    return reset_device(c);
}

int rsb_filter_add(rsb_ctx *c, uint32_t addr) { return rsb_filter_add_many(c, &addr, 1); }

int rsb_filter_add_many(rsb_ctx *c, const uint32_t *addrs, size_t n) {
    if (!c || (!addrs && n)) return RSB_EINVAL;
    if (n == 0) return RSB_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    {
        const int rc = drop_enqueued_chain(c);
        if (rc != RSB_OK) return rc;
    }
    for (size_t i = 0; i < n; ++i) c->replay.filter().add(addrs[i] & 0xffffffu);
    {
        const int rc = superset_set_bits(c, addrs, n);  // (the two blocking 2 MiB copies per call this replaces cost a shard hand-over 1 ms)
        if (rc != RSB_OK) return rc;
    }
    c->filter_dirty = true;
    return RSB_OK;
}

int rsb_filter_expire(rsb_ctx *c) {
    if (!c) return RSB_EINVAL;
    // the device bitmap stays a superset (it only ever grows), which is all the prune needs
    {
        const int rc = drop_enqueued_chain(c);
        if (rc != RSB_OK) return rc;
    }
    c->replay.filter().expire();
    c->filter_dirty = true;
    return RSB_OK;
}

// the adder bitmap on the host, once every pending scan has finished writing it
static int fetch_bitmap(rsb_ctx *c, std::vector<uint32_t> &words) {
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    for (rsb_slot &b : c->slots)
        if (b.pending) HIP_TRY(c, hipEventSynchronize(b.ev[3]));
    if (c->reset_pending) HIP_TRY(c, hipEventSynchronize(c->reset_done));
    words.resize((1u << 24) / 32);
    HIP_TRY(c, hipMemcpy(words.data(), c->d_bitmap, words.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return RSB_OK;
}

int rsb_seek(rsb_ctx *c, uint64_t stream_pos, const void *carry, int carry_on_device) {
    if (!c || c->block_mode) return RSB_EINVAL;
    if (c->cfg.dc_filter || c->q11_table_bits) {
        c->err = c->cfg.dc_filter ? "rsb_seek: not available with dc_filter (the converter's state does not travel)"
                                  : "rsb_seek: not available with RSB_FLAG_SC16Q11_TABLE (the carry is kept as magnitudes of the batch before)";
        return RSB_EINVAL;
    }
    if (c->any_pending()) {
        c->err = "rsb_seek: a batch is pending";
        return RSB_ESTATE;
    }
    // (a source with its own buffer geometry restarts its buffers wherever its driver does, sdr_bladerf.c:366-375: any position)
    if (!c->cfg.block_samples && stream_pos % kBlock != 0) {
        c->err = "rsb_seek: stream position is not a multiple of 131072";
        return RSB_EINVAL;
    }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const unsigned bps = (c->cfg.input_format == RSB_UC8) ? 2 : 4, k = c->submitted % rsb_ctx::kCarries;
    if (carry) HIP_TRY(c, hipMemcpy(c->d_carry[k], carry, (size_t)kTrailing * bps, carry_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    c->carry_valid[k] = carry ? 1 : 0;
    c->stream_pos = c->submit_pos = stream_pos;
    c->finished = c->submit_finished = false;
    return RSB_OK;
}

int rsb_export_adders(rsb_ctx *c, uint32_t *out, size_t cap, size_t *n_out) {
    if (!c || (!out && cap) || !n_out) return RSB_EINVAL;
    {
        // the set bits listed on the device (a few hundred addresses instead of 2 MiB across PCIe), behind every pending scan
        HIP_TRY(c, hipSetDevice(c->cfg.device));
        hipStream_t s = c->replay_stream;
        for (rsb_slot &b : c->slots)
            if (b.pending) HIP_TRY(c, hipStreamWaitEvent(s, b.ev[3], 0));
        uint32_t *list = (uint32_t *)c->d_sort_temp;
        const uint32_t room = (uint32_t)std::min<size_t>(c->sort_temp_bytes / sizeof(uint32_t) - 1, 1u << 20);
        HIP_TRY(c, launch_bitmap_list(c->d_bitmap, list, room, s));
        uint32_t found = 0;
        HIP_TRY(c, hipMemcpyAsync(&found, list, sizeof(found), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        if (found <= room) {
            *n_out = found;
            if (found > cap) {
                c->err = "rsb_export_adders: output buffer too small";
                return RSB_EOVERFLOW;
            }
            if (found) HIP_TRY(c, hipMemcpy(out, list + 1, found * sizeof(uint32_t), hipMemcpyDeviceToHost));
            std::sort(out, out + found);
            return RSB_OK;
        }
    }
    std::vector<uint32_t> words;
    const int rc = fetch_bitmap(c, words);
    if (rc != RSB_OK) return rc;
    size_t n = 0;
    for (uint32_t w = 0; w < words.size(); ++w)
        for (uint32_t x = words[w]; x; x &= x - 1) {
            if (n < cap) out[n] = (w << 5) | (uint32_t)__builtin_ctz(x);
            ++n;
        }
    *n_out = n;
    if (n > cap) {
        c->err = "rsb_export_adders: output buffer too small";
        return RSB_EOVERFLOW;
    }
    return RSB_OK;
}

int rsb_superset_add(rsb_ctx *c, const uint32_t *addrs, size_t n) {
    if (!c || (!addrs && n) || n > 0xffffffffull) return RSB_EINVAL;
    if (n == 0) return RSB_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    const int rc = drop_enqueued_chain(c);
    if (rc != RSB_OK) return rc;
    return superset_set_bits(c, addrs, n);
}

int rsb_batch_references(rsb_ctx *c, const uint32_t *addrs, size_t n, size_t *count) {
    if (!c || (!addrs && n) || !count || n > 4096) return RSB_EINVAL;
    *count = 0;
    if (!c->last_items_info) {
        c->err = "rsb_batch_references: the last collected batch was not replayed on the device (its kept items are not at hand)";
        return RSB_ESTATE;
    }
    if (n == 0 || c->last_items_n == 0) return RSB_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    uint32_t *dev = nullptr;
    const int rc = upload_addrs(c, addrs, n, &dev);
    if (rc != RSB_OK) return rc;
    HIP_TRY(c, launch_items_reference(c->last_items_info, c->last_items_n, dev, (uint32_t)n, dev - 1, c->replay_stream));
    uint32_t hits = 0;
    HIP_TRY(c, hipMemcpyAsync(&hits, dev - 1, sizeof(hits), hipMemcpyDeviceToHost, c->replay_stream));
    HIP_TRY(c, hipStreamSynchronize(c->replay_stream));
    *count = hits;
    return RSB_OK;
}

int rsb_filter_export(rsb_ctx *c, uint32_t *out, size_t cap, size_t *n_out) {
    if (!c || (!out && cap) || !n_out) return RSB_EINVAL;
    c->replay.filter().contents(c->h_addrs);
    *n_out = c->h_addrs.size();
    if (c->h_addrs.size() > cap) {
        c->err = "rsb_filter_export: output buffer too small";
        return RSB_EOVERFLOW;
    }
    if (!c->h_addrs.empty()) std::memcpy(out, c->h_addrs.data(), c->h_addrs.size() * sizeof(uint32_t));
    return RSB_OK;
}

// a batch onto the caller's stream in three parts, so that rsb_submit_many can put ONE scan launch between the first and the last
// part of all its channels
static std::atomic<uint64_t> g_submit_seq{0};  // orders the ev[3] records of all contexts (rsb_collect_many: which of one stream's is the last)

// in_one_launch: the batch is one of rsb_submit_many's single scan launch, whose prepare kernel zeroes the counters of all its batches
static int enqueue_before_scan(rsb_ctx *c, rsb_slot &b, const BatchArgs &a, hipStream_t s, bool in_one_launch = false) {
    // (one-slot contexts: a record copy still in flight may read this slot's item buffer -- a list merged with Mode A/C messages)
    if (c->copy_in_flight && c->copy_slot == &b) HIP_TRY(c, hipStreamWaitEvent(s, c->copy_done, 0));
    if (c->reset_pending) {  // rsb_reset cleared the prune's superset on the context's own stream
        HIP_TRY(c, hipStreamWaitEvent(s, c->reset_done, 0));
        c->reset_pending = false;
    }
    b.scan_share = 1.0f;
    b.scan_timed = b.ac_timed = true;
    if (in_one_launch) return RSB_OK;
    HIP_TRY(c, launch_submit_prepare(a, s));  // counters and region counts to zero, the next batch's overlap copied (a.carry_out): one launch
    HIP_TRY(c, hipEventRecord(b.ev[0], s));
    return RSB_OK;
}

static int enqueue_after_scan(rsb_ctx *c, rsb_slot &b, const BatchArgs &a, uint64_t nsamples, hipStream_t s, bool in_one_launch = false);

static int enqueue_batch(rsb_ctx *c, rsb_slot &b, const BatchArgs &a_in, uint64_t nsamples, hipStream_t s) {
    BatchArgs a = a_in;
    // the next batch's 326-sample overlap (sdr_ifile.c:199-201) leaves with the prepare kernel in front of the scan (with the DC block
    // inside the kernels it is made behind the scan, as magnitudes: enqueue_after_scan)
    if (!c->block_mode && nsamples && nsamples % c->block_len == 0 && !a.dc_local) a.carry_out = c->d_carry[(c->submitted + 1) % rsb_ctx::kCarries];
    int rc = enqueue_before_scan(c, b, a, s);
    if (rc != RSB_OK) return rc;
    if (c->run_modes) HIP_TRY(c, launch_scan(a, c->scan_grid, s));
    return enqueue_after_scan(c, b, a, nsamples, s);
}

static int enqueue_after_scan(rsb_ctx *c, rsb_slot &b, const BatchArgs &a, uint64_t nsamples, hipStream_t s, bool in_one_launch) {
    if (!in_one_launch) HIP_TRY(c, hipEventRecord(b.ev[1], s));
    // The prune runs at the head of the batch's replay chain, on the context's own stream (rsb_collect, or collect_ahead while
    // the batch before is being delivered), not here: the caller's stream then carries the scans of successive batches back to
    // back, and the prune sees whatever the caller has added to the superset in between (RSB_FLAG_DEFER_PRUNE relies on that).
    b.pruned = false;
    b.pruned_in_order = false;
    const bool ac_work = (c->run_ac || c->exact_stats) && nsamples;
    // (the one launch for many channels: no event per channel where nothing lies between it and the next one)
    if (ac_work) HIP_TRY(c, hipEventRecord(b.ev[2], s));
    else b.ac_timed = false;
    b.means_on_device = nullptr;
    if (ac_work) {
        // mag_buf.mean_level / mean_power of every block: demodulate2400AC's noise level (demod_2400.c:484-485) and the
        // noise power statistic (:378-382)
        const size_t nblk = (nsamples + c->block_len - 1) / c->block_len;
        BatchArgs ac = a;
        if (!ac.block_means && (a.fmt == kFmtSC16 || a.fmt == kFmtSC16Q11)) {
            // the float converters' block means (convert.c:226-262): exact only in sample order
            if (c->mean_sub) {
                // ... of a driver that converts a buffer in sub-blocks and averages their means (sdr_bladerf.c:385-390, 403-404)
                const uint32_t nsub = (uint32_t)((nsamples + c->mean_sub - 1) / c->mean_sub);
                HIP_TRY(c, launch_block_means_float(a.fmt, a.iq, nsamples, c->mean_sub, c->d_sub_means, s));
                HIP_TRY(c, launch_means_average(c->d_sub_means, nsub, c->block_len / c->mean_sub, b.d_block_means, (uint32_t)nblk, s));
            } else {
                HIP_TRY(c, launch_block_means_float(a.fmt, a.iq, nsamples, c->block_len, b.d_block_means, s));
            }
            ac.block_means = b.d_block_means;
        }
        if (!ac.block_means) HIP_TRY(c, hipMemsetAsync(b.d_block_sums, 0, nblk * 2 * sizeof(uint64_t), s));
        if (c->run_ac)
            HIP_TRY(c, launch_modeac(ac, s));
        else if (!ac.block_means)
            HIP_TRY(c, launch_block_sums(ac, s));
        b.means_on_device = ac.block_means;
    }
    if (!c->block_mode && nsamples % c->block_len == 0 && nsamples) {
        // the next batch's 326-sample overlap (sdr_ifile.c:199-201)
        const unsigned bps = (a.fmt == kFmtUC8 || a.fmt == kFmtMag16) ? 2 : 4;
        const unsigned k = (c->submitted + 1) % rsb_ctx::kCarries;
        if (a.dc_local)  // the DC block inside the scan: the next batch's overlap as magnitudes (the filter's states of this batch do not outlive its slot)
            HIP_TRY(c, launch_dc_carry(a, (uint16_t *)c->d_carry[k], s));
        else if (!a.carry_out)  // (else the one launch's prepare kernel has copied it)
            HIP_TRY(c, hipMemcpyAsync(c->d_carry[k], (const char *)a.iq + (nsamples - kTrailing) * bps, (size_t)kTrailing * bps, hipMemcpyDeviceToDevice, s));
        c->carry_valid[k] = 1;
    }
    HIP_TRY(c, hipEventRecord(b.ev[3], s));
    b.submit_seq = ++g_submit_seq;
    b.pending = true;
    b.pending_n = nsamples;
    b.pending_stream = s;
    b.pending_args = a;
    b.g0 = c->submit_pos;
    ++c->submitted;
    if (!c->block_mode) {
        c->submit_pos += nsamples;
        // a short batch ends the recording, and so does an empty one (the reference publishes a zero-length block only at EOF,
        // sdr_ifile.c:213-225): nothing may follow it without rsb_seek / rsb_reset, so no carry is needed behind it
        if (nsamples == 0 || nsamples % c->block_len != 0) c->submit_finished = true;
    }
    return RSB_OK;
}

static int submit_checks(rsb_ctx *c, const void *iq_dev, uint64_t nsamples) {
    if (!c || (!iq_dev && nsamples) || c->block_mode) return RSB_EINVAL;
    if (c->submitted - c->collected >= (unsigned)c->nslots) {
        c->err = c->nslots == 1 ? "rsb_submit: previous batch not collected" : "rsb_submit: four batches already in flight: collect one first";
        return RSB_ESTATE;
    }
    if (c->submit_finished) {
        c->err = "rsb_submit: the last batch was not a whole number of blocks (131072 samples, or rsb_config.block_samples): only the last batch of a recording may be";
        return RSB_ESTATE;
    }
    if (nsamples > c->cfg.max_batch_samples) {
        c->err = "rsb_submit: nsamples exceeds rsb_config.max_batch_samples";
        return RSB_EINVAL;
    }
    return RSB_OK;
}

int rsb_submit(rsb_ctx *c, const void *iq_dev, uint64_t nsamples, void *stream) {
    const int ok = submit_checks(c, iq_dev, nsamples);
    if (ok != RSB_OK) return ok;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    rsb_slot &b = c->submit_slot();
    hipStream_t s = (hipStream_t)stream;
    if (c->reset_pending) {  // (the converters below run in front of enqueue_before_scan, which waits for this otherwise)
        HIP_TRY(c, hipStreamWaitEvent(s, c->reset_done, 0));
        c->reset_pending = false;
    }
    if (c->dc_fused) {
        // --dcfilter, Mode S only: one pass over the IQ leaves the filter's state before every 8-sample chunk; the scan (and
        // whatever else needs a magnitude) takes the samples through the filter itself.  No magnitude touches memory.
        HIP_TRY(c, launch_dc_states(c->cfg.input_format, iq_dev, nsamples, c->d_dc_tile_sums, b.d_dc_tile_start, b.d_dc_local, c->d_dc_state, s));
        BatchArgs a = c->batch_args(b, iq_dev, nsamples, c->cfg.input_format);
        a.dc_local = b.d_dc_local;
        a.dc_tile_start = b.d_dc_tile_start;
        a.dc_pow = c->d_dc_pow;
        dc_coefficients(&a.dc_a, &a.dc_b);
        a.carry_is_mag = 1;
        return enqueue_batch(c, b, a, nsamples, s);
    }
    if (c->cfg.dc_filter) {
        // --dcfilter: the chunked-scan converter writes the batch's magnitudes; the scan reads those (carry = the previous batch's last 326)
        const bool want_means = c->run_ac || c->exact_stats;
        HIP_TRY(c, launch_convert_dc(c->cfg.input_format, iq_dev, c->d_magbuf, nsamples, c->d_dc_tile_sums, c->d_dc_tile_start, c->d_dc_state,
                                     want_means ? b.d_block_means : nullptr, s));
        BatchArgs a = c->batch_args(b, c->d_magbuf, nsamples, kFmtMag16);
        if (want_means) a.block_means = b.d_block_means;
        return enqueue_batch(c, b, a, nsamples, s);
    }
    if (c->q11_table_bits) {
        // a reference built with -DSC16Q11_TABLE_BITS: magnitudes from the table converter, the block means from their integer
        // sums (convert.c:321-327, the same expressions as the UC8 converter's, which is what the magnitude path computes)
        HIP_TRY(c, launch_convert_q11t(iq_dev, c->d_magbuf, nsamples, c->q11_table_bits, nullptr, s));
        return enqueue_batch(c, b, c->batch_args(b, c->d_magbuf, nsamples, kFmtMag16), nsamples, s);
    }
    return enqueue_batch(c, b, c->batch_args(b, iq_dev, nsamples, c->cfg.input_format), nsamples, s);
}

}  // extern "C"

namespace {

// a batch asked for as wire bytes (rsb_collect_wire) instead of records
struct WireRequest {
    int wire;
    uint32_t flags;
    uint8_t *out;
    size_t cap;
    size_t nbytes;
};

// records on the device -> wire bytes on the device -> the caller's buffer
int device_wire(rsb_ctx *c, const rsb_message *msgs_dev, uint32_t nmsgs, WireRequest &w, hipStream_t s) {
    if ((size_t)nmsgs + 1 > c->wire_off_cap) {
        if (c->d_wire_off) (void)hipFree(c->d_wire_off);
        c->d_wire_off = nullptr;
        c->wire_off_cap = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_wire_off, ((size_t)nmsgs + 1) * sizeof(uint32_t)));
        c->wire_off_cap = (size_t)nmsgs + 1;
    }
    WireArgs a{msgs_dev, nmsgs, w.wire, w.flags, c->d_wire_off, nullptr};
    HIP_TRY(c, launch_wire_offsets(a, c->d_sort_temp, c->sort_temp_bytes, s));
    uint32_t total = 0;
    HIP_TRY(c, hipMemcpyAsync(&total, c->d_wire_off + nmsgs, sizeof(total), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    w.nbytes = total;
    if (total > c->wire_out_cap) {
        if (c->d_wire_out) (void)hipFree(c->d_wire_out);
        c->d_wire_out = nullptr;
        c->wire_out_cap = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_wire_out, (size_t)total + total / 4));
        c->wire_out_cap = (size_t)total + total / 4;
    }
    a.out = c->d_wire_out;
    HIP_TRY(c, launch_wire_write(a, s));
    const size_t ncopy = total < w.cap ? total : w.cap;
    if (ncopy) HIP_TRY(c, hipMemcpyAsync(w.out, c->d_wire_out, ncopy, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return RSB_OK;
}

// After a device replay of a batch that also has Mode A/C candidates: the greedy skip of demod_2400.c:659 over the (few)
// candidates on the host, then each block's Mode A/C messages placed behind its Mode S messages (readsb.c:883-886) by a
// kernel, into the slot's item buffer (free once the batch is pruned).  Returns 1 and the merged list (*merged, *ntotal),
// or 0 when that buffer is too small (the caller then merges on the host), < 0 on error.
int device_merge_modeac(rsb_ctx *c, rsb_slot &b, uint32_t nacc, size_t nblocks, hipStream_t s, const rsb_message **merged, size_t *ntotal) {
    if (((size_t)nacc + c->h_ac.size()) * sizeof(rsb_message) > (size_t)c->item_cap * sizeof(ItemRec)) return 0;  // before anything is counted
    c->ac_msgs.clear();
    c->h_ac_counts.assign(nblocks + 1, 0);
    c->replay.modeac_messages(b.g0, nblocks, c->h_ac, c->ac_msgs, c->h_ac_counts.data());
    const uint32_t nacm = (uint32_t)c->ac_msgs.size();
    // one upload: blk_first[nblocks + 1], ac_before[nblocks + 1], ac_dst[nacm], then the records (8-byte aligned)
    const size_t nwords = 2 * (nblocks + 1) + nacm, rec_off = (nwords * 4 + 7) / 8 * 8, bytes = rec_off + (size_t)nacm * sizeof(rsb_message);
    c->h_merge.resize(bytes);
    uint32_t *blk_first = (uint32_t *)c->h_merge.data(), *ac_before = blk_first + nblocks + 1, *ac_dst = ac_before + nblocks + 1;
    uint32_t mfirst = 0, afirst = 0;
    for (size_t k = 0; k < nblocks; ++k) {
        blk_first[k] = mfirst;
        ac_before[k] = afirst;
        mfirst += c->h_blk_counts[k];
        for (uint32_t t = 0; t < c->h_ac_counts[k]; ++t) ac_dst[afirst + t] = mfirst + afirst + t;  // behind the block's Mode S messages
        afirst += c->h_ac_counts[k];
    }
    blk_first[nblocks] = mfirst;
    ac_before[nblocks] = afirst;
    if (nacm) std::memcpy(c->h_merge.data() + rec_off, c->ac_msgs.data(), (size_t)nacm * sizeof(rsb_message));
    if (bytes > c->merge_cap) {
        if (c->d_merge) (void)hipFree(c->d_merge);
        c->d_merge = nullptr;
        c->merge_cap = 0;
        HIP_TRY(c, hipMalloc((void **)&c->d_merge, bytes + bytes / 2));
        c->merge_cap = bytes + bytes / 2;
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_merge, c->h_merge.data(), bytes, hipMemcpyHostToDevice, s));
    MergeArgs m;
    m.modes = b.d_out;
    m.n_modes = nacc;
    m.blk_first = (const uint32_t *)c->d_merge;
    m.ac_before = m.blk_first + nblocks + 1;
    m.nblocks = (uint32_t)nblocks;
    m.ac_dst = m.ac_before + nblocks + 1;
    m.ac_msgs = (const rsb_message *)(c->d_merge + rec_off);
    m.n_ac = nacm;
    m.dst = (rsb_message *)b.d_items;
    HIP_TRY(c, launch_merge_modeac(m, s));
    HIP_TRY(c, hipStreamSynchronize(s));  // h_merge is pageable
    *merged = m.dst;
    *ntotal = (size_t)nacc + nacm;
    return 1;
}

// One rsb_collect, in phases that hand their results to each other through this:
//   replay()      prune + replay of the batch -- the one-wait chain, or step by step (sort fallback, records straight to the caller's
//                 buffer), or the candidate records for the host replay -- and the overflow checks
//   fetch()       the finished list to the caller (device merge with Mode A/C, wire encoding, record copy), or the candidates to the host
//   stage_times() the batch's events -> rsb_timing
//   host_list()   what the device did not deliver: Mode A/C merge on the host, host replay
//   account()     statistics, stream position, the caller's buffer
struct CollectRun {
    rsb_ctx *c;
    rsb_slot &b;
    hipStream_t s;
    rsb_message *out;
    size_t cap;
    WireRequest *wire;
    const BatchArgs &a;
    size_t nblocks;
    int on_device = 0, direct_parts = 0;
    uint32_t cnt[kCntCount] = {};
    bool replayed = false, chain_ran = false;
    uint32_t n_items = 0, nsurv = 0, nac = 0;
    size_t n = 0;            // messages of this batch
    bool delivered = false;  // already in out[]

    int replay();
    int fetch();
    void stage_times();
    int host_list();
    int account(size_t *n_out);
};

int CollectRun::replay() {
    // The usual batch needs no word from the device before its replay is enqueued: the prune leaves the kept items packed in
    // order and their number in the counters, the replay kernels read it there, and the host waits ONCE, for the first
    // round's results and the counters together.  Anything unusual (a scan wavefront overflowed an LDS queue, a device queue
    // is full, marker items, host replay, records written straight to the caller's buffer) takes the step-by-step path below.
    if (b.chain_enqueued || chain_needs_no_host(c, b)) {
        chain_ran = true;
        if (!b.chain_enqueued) {
            c->sorted_buf = 0;
            const int rc = replay_enqueue(c, b, kept_items_bound(c, b), true, s);
            if (rc != RSB_OK) return rc;
        }
        on_device = replay_finish(c, b, s, cnt);
        // (the fold, or whatever an unfinished round left in the tables; rsb_collect_many ends with a host wait for its shared stream instead)
        if (!c->chain_stream) (void)tables_touched(c, s);
        if (on_device == -1000)
            on_device = 0;  // cnt[] says why: looked at below
        else if (on_device < 0)
            return on_device;
        else
            replayed = true;
    }
    if (!replayed && !(cnt[kCntOverflow] | cnt[kCntItems] | cnt[kCntSurvItems])) {
        HIP_TRY(c, hipMemcpyAsync(cnt, b.d_counters, sizeof(cnt), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
    }
    if (c->debug_flags & 256) {  // ablation runs only (RSB_ABL): survivors of the Mode A/C kernel's first two stages
        uint32_t dbg[kCntAlloc];
        HIP_TRY(c, hipMemcpy(dbg, b.d_counters, sizeof(dbg), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "modeac stages: F1 survivors %u, F2 survivors %u, candidates %u\n", dbg[kDbgStrips], dbg[kDbgFrames], dbg[kCntAc]);
    }
#ifdef RSB_STAGE_CLOCKS
    {
        uint32_t dbg[kCntAlloc];
        HIP_TRY(c, hipMemcpy(dbg, b.d_counters, sizeof(dbg), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "stage clocks (ticks/16 summed over waves): convert %u edge %u gate %u df %u slice %u | strips %u gate batches %u df batches %u slice batches %u | edge cands %u frames %u long %u gated %u\n",
                     dbg[kDbgClkConvert], dbg[kDbgClkEdge], dbg[kDbgClkGate], dbg[kDbgClkDf], dbg[kDbgClkSlice], dbg[kDbgStrips], dbg[kDbgGateBatches], dbg[kDbgDfBatches],
                     dbg[kDbgSliceBatches], dbg[kDbgEdgeCands], dbg[kDbgFrames], dbg[kDbgLongFrames], dbg[kCntGated]);
    }
#endif
    if (cnt[kCntOverflow] || cnt[kCntSurvItems] > c->surv_item_cap || cnt[kCntAc] > c->ac_cap) {
        char buf[240];
        if (cnt[kCntOverflow] >= kPruneStallUnit)
            std::snprintf(buf, sizeof(buf), "ordered prune: %u tile(s) gave up waiting for the tiles before them (a stalled device queue?); the batch is lost",
                          cnt[kCntOverflow] / kPruneStallUnit);
        else
            std::snprintf(buf, sizeof(buf), "device queue overflow: %u item regions, kept items %u/%u, modeac %u/%u (lower rsb_config.queue_div)",
                          cnt[kCntOverflow], cnt[kCntSurvItems], c->surv_item_cap, cnt[kCntAc], c->ac_cap);
        c->err = buf;
        c->finished = true;
        return RSB_EOVERFLOW;
    }
    n_items = cnt[kCntSurvItems];
    if (!replayed) {
        if (b.pruned_in_order && cnt[kCntItems] != 0 && c->run_modes) {
            // a scan wavefront overflowed an LDS queue: its per-thread path put items into the shared region, out of order.  Prune once
            // more, this time with the sort behind it.
            HIP_TRY(c, hipMemsetAsync(b.d_counters + kCntSurvItems, 0, 2 * sizeof(uint32_t), s));
            HIP_TRY(c, launch_prune(a, s));
            HIP_TRY(c, hipMemcpyAsync(cnt, b.d_counters, sizeof(cnt), hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            b.pruned_in_order = false;
            ++c->sort_fallbacks;
            if (std::getenv("RSB_VERBOSE")) std::fprintf(stderr, "readsb_amd: batch pruned again with the sort (%u items took a per-thread path)\n", cnt[kCntItems]);
            if (cnt[kCntOverflow] || cnt[kCntSurvItems] > c->surv_item_cap) {
                c->err = "device queue overflow: kept items (lower rsb_config.queue_div)";
                c->finished = true;
                return RSB_EOVERFLOW;
            }
        }
        n_items = cnt[kCntSurvItems];
        if (b.pruned_in_order)
            c->sorted_buf = 0;  // packed in stream order by the prune itself
        else
            HIP_TRY(c, launch_sort(a, n_items, c->d_sort_temp, c->sort_temp_bytes, &c->sorted_buf, s));
        HIP_TRY(c, hipEventRecord(b.rev[1], s));

        // the sequential half: on the device, unless asked otherwise or the device path declines the batch
        if (!c->run_modes) {
            on_device = 1;  // nothing to replay: a Mode A/C only pass of the drop-in
            std::memset(cnt + kCntChanged, 0, (kCntGeneralStrips - kCntChanged) * sizeof(uint32_t));
            c->h_blk_counts.assign(nblocks + 1, 0);
        } else if (!c->host_replay_only && !a.emit_dead && !a.keep_all) {
            // records wanted and nothing to interleave: the list can leave in parts while it is still being written
            const bool parts_ok = !wire && out && cnt[kCntAc] == 0;
            int rc = replay_enqueue(c, b, n_items, false, s, parts_ok ? out : nullptr, cap);
            chain_ran = true;
            if (rc != RSB_OK) return rc;
            on_device = replay_finish(c, b, s, cnt, &direct_parts);
            if (!c->chain_stream) (void)tables_touched(c, s);
            if (on_device < 0) return on_device;
            if (!on_device) direct_parts = 0;  // (whatever the kernel wrote to the buffer, the host path rewrites it)
        }
    }
    if (!on_device) {
        // kept items -> distinct offsets -> full candidate records for the host replay
        HIP_TRY(c, launch_unique_finalize(a, n_items, c->sorted_buf, c->d_sort_temp, c->sort_temp_bytes, s));
        HIP_TRY(c, hipMemcpyAsync(&nsurv, b.d_counters + kCntSurvivors, sizeof(nsurv), hipMemcpyDeviceToHost, s));
    }
    if (!on_device || !chain_ran) HIP_TRY(c, hipEventRecord(b.rev[2], s));  // (the device replay's chain records it behind its emit kernel)
    // (a chain on the caller's stream that replayed the batch on the device has been waited for by its event, and nothing the host needs
    // has been enqueued since: the stream itself may hold the caller's next scan by now)
    if (!(on_device && chain_ran && b.chain_foreign)) HIP_TRY(c, hipStreamSynchronize(s));
    if (!on_device && nsurv > c->surv_cap) {
        char buf[200];
        std::snprintf(buf, sizeof(buf), "device queue overflow: surviving offsets %u/%u (lower rsb_config.queue_div)", nsurv, c->surv_cap);
        c->err = buf;
        c->finished = true;
        return RSB_EOVERFLOW;
    }
    if (on_device && cnt[kCntAccepted] > c->out_cap) {
        char buf[200];
        std::snprintf(buf, sizeof(buf), "device queue overflow: messages %u/%u (lower rsb_config.queue_div)", cnt[kCntAccepted], c->out_cap);
        c->err = buf;
        c->finished = true;
        c->filter_dirty = true;
        c->tables_valid = false;
        return RSB_EOVERFLOW;
    }

    return RSB_OK;
}

int CollectRun::fetch() {
    const auto t_copy = std::chrono::steady_clock::now();
    nac = cnt[kCntAc];
    c->h_ac.resize(nac);
    if (nac) HIP_TRY(c, hipMemcpyAsync(c->h_ac.data(), b.d_ac, nac * sizeof(AcCand), hipMemcpyDeviceToHost, s));
    rsb_stats &st = c->replay.stats();
    if (on_device) {
        const uint32_t nacc = cnt[kCntAccepted];
        const rsb_message *list = b.d_out;  // the batch's messages on the device
        size_t nlist = nacc;
        bool on_dev_list = true;
        if (nac) {  // interleave the Mode A/C messages on the device as well
            HIP_TRY(c, hipStreamSynchronize(s));  // h_ac
            const int rc = device_merge_modeac(c, b, nacc, nblocks, s, &list, &nlist);
            if (rc < 0) return rc;
            on_dev_list = rc == 1;
        }
        if (direct_parts) {
            // the emit kernel wrote the records into the caller's buffer itself
            n = nlist;
            delivered = true;
        } else if (on_dev_list && wire) {
            // encode where the records are, only the wire bytes cross PCIe
            n = nlist;
            const int rc = device_wire(c, list, (uint32_t)nlist, *wire, s);
            if (rc != RSB_OK) return rc;
            delivered = true;
        } else if (on_dev_list) {
            // straight into the caller's buffer
            // straight into the caller's buffer, on the copy stream: everything the list depends on has been waited for, and the
            // replay stream is free to take the next batch's chain meanwhile
            n = nlist;
            c->last_list_dev = list;
            const size_t ncopy = n < cap ? n : cap;
            if (list != b.d_out) HIP_TRY(c, hipStreamSynchronize(s));  // (a merged list: its kernel ran on the replay stream)
            if (ncopy) {
                HIP_TRY(c, hipMemcpyAsync(out, list, ncopy * sizeof(rsb_message), hipMemcpyDeviceToHost, c->copy_stream));
                HIP_TRY(c, hipEventRecord(c->copy_done, c->copy_stream));
                c->copy_slot = &b;
            }
            if (!c->chain_stream) collect_ahead(c);
            if (c->defer_copy_wait && ncopy)
                c->copy_in_flight = true;  // rsb_collect_begin: the caller comes back with rsb_collect_wait
            else
                HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
            delivered = true;
        } else {
            c->dev_msgs.resize(nacc);
            if (nacc) HIP_TRY(c, hipMemcpyAsync(c->dev_msgs.data(), b.d_out, (size_t)nacc * sizeof(rsb_message), hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
        }
        for (int e = 0; e < 3; ++e) st.demod_accepted[e] += cnt[kCntAccE0 + e];
    } else {
        cnt[kCntSurvivors] = nsurv;
        c->h_surv.resize(nsurv);
        if (nsurv) HIP_TRY(c, hipMemcpyAsync(c->h_surv.data(), c->d_surv, nsurv * sizeof(Survivor), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
    }
    c->timing.copy_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_copy).count();

    return RSB_OK;
}

// RSB_GAP_TRACE=1 (measurement): per collected batch, device time between the stages' events -- also across contexts: from the end of
// the chain collected before (whichever context's) to this batch's scan start
static void gap_trace(const rsb_slot &b) {
    static const bool on = std::getenv("RSB_GAP_TRACE") != nullptr;
    if (!on) return;
    static hipEvent_t last_chain_end = nullptr;
    static double sum[4] = {0, 0, 0, 0};
    static int n = 0;
    float v = 0;
    if (last_chain_end && hipEventElapsedTime(&v, last_chain_end, b.ev[0]) == hipSuccess) sum[0] += v;   // previous chain end -> scan start
    if (hipEventElapsedTime(&v, b.ev[0], b.ev[1]) == hipSuccess) sum[1] += v;                            // scan
    if (hipEventElapsedTime(&v, b.ev[1], b.rev[0]) == hipSuccess) sum[2] += v;                           // scan end -> chain start
    if (hipEventElapsedTime(&v, b.rev[0], b.rev[2]) == hipSuccess) sum[3] += v;                          // chain
    last_chain_end = b.rev[2];
    if (++n % 8 == 0) {
        std::fprintf(stderr, "gap trace (ms, mean of 8): chain end -> scan start %.3f | scan %.3f | scan end -> chain start %.3f | chain %.3f\n", sum[0] / 8, sum[1] / 8,
                     sum[2] / 8, sum[3] / 8);
        sum[0] = sum[1] = sum[2] = sum[3] = 0;
    }
}

void CollectRun::stage_times() {
    float ms = 0;
    gap_trace(b);
    c->timing.scan_ms = c->timing.modeac_ms = c->timing.total_ms = 0;
    if (b.scan_timed) {  // (a channel of rsb_submit_many's one launch other than the first: the launch's time is reported there)
        (void)hipEventElapsedTime(&c->timing.scan_ms, b.ev[0], b.ev[1]);
        c->timing.scan_ms *= b.scan_share;
        (void)hipEventElapsedTime(&c->timing.total_ms, b.ev[0], b.ev[3]);
    }
    c->timing.prune_ms = 0;
    if (b.ac_timed) (void)hipEventElapsedTime(&c->timing.modeac_ms, b.ev[2], b.ev[3]);
    const rsb_slot &tb = b.time_from ? *b.time_from : b;  // (a chain shared by many channels: its times, the same for every channel)
    (void)hipEventElapsedTime(&c->timing.prune_ms, tb.rev[0], tb.rev[1]);  // prune (+ the radix sort in the fallback cases)
    c->timing.sort_ms = 0;
    (void)hipEventElapsedTime(&ms, tb.rev[1], tb.rev[2]);
    b.time_from = nullptr;
    c->timing.device_replay_ms = on_device ? ms : 0.0f;
    if (!on_device) c->timing.prune_ms += ms;  // the candidate re-slice belongs to the prune stage of the host path
    c->timing.total_ms += c->timing.sort_ms + ms;
    c->timing.replay_on_device = on_device;

}

int CollectRun::host_list() {
    const auto t0 = std::chrono::steady_clock::now();
    if (!delivered) {
        c->msgs.clear();
        if (on_device)
            c->replay.merge_modeac(b.g0, c->dev_msgs, c->h_blk_counts.data(), nblocks, c->h_ac, c->msgs);
        else if (c->exact_stats && b.pending_n) {
            // mag_buf.mean_power of every block of the batch: from the float means, or from the integer sums (convert.c:108-110)
            c->h_blk_mean_power.resize(nblocks);
            if (b.means_on_device) {
                std::vector<double> means(2 * nblocks);
                HIP_TRY(c, hipMemcpy(means.data(), b.means_on_device, means.size() * sizeof(double), hipMemcpyDeviceToHost));
                for (size_t k = 0; k < nblocks; ++k) c->h_blk_mean_power[k] = means[2 * k + 1];
            } else {
                c->h_blk_sums.resize(2 * nblocks);
                HIP_TRY(c, hipMemcpy(c->h_blk_sums.data(), b.d_block_sums, c->h_blk_sums.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
                for (size_t k = 0; k < nblocks; ++k) {
                    const uint64_t len = std::min<uint64_t>(c->block_len, b.pending_n - k * c->block_len);
                    c->h_blk_mean_power[k] = (double)c->h_blk_sums[2 * k + 1] / 65535.0 / 65535.0 / (double)(unsigned)len;
                }
            }
            c->replay.run_batch(b.g0, c->h_surv, c->h_ac, c->msgs, c->h_blk_mean_power.data(), b.pending_n);
        } else
            c->replay.run_batch(b.g0, c->h_surv, c->h_ac, c->msgs);
        n = c->msgs.size();
        if (!on_device && !c->host_replay_only) c->filter_dirty = true;  // the host moved the filter on
    }
    c->timing.replay_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
#ifdef RSB_STAGE_CLOCKS
    {
        uint32_t dbg[kCntAlloc];
        HIP_TRY(c, hipDeviceSynchronize());
        HIP_TRY(c, hipMemcpy(dbg, b.d_counters, sizeof(dbg), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "emit clocks (ticks/16 summed over waves): set-up + loop %u gather %u trios %u repair %u store %u | iterations %u\n", dbg[kDbgEmitSetup],
                     dbg[kDbgEmitGather], dbg[kDbgEmitTrios], dbg[kDbgEmitRepair], dbg[kDbgEmitStore], dbg[kDbgEmitIters]);
    }
#endif

    return RSB_OK;
}

int CollectRun::account(size_t *n_out) {
    b.chain_foreign = false;
    rsb_stats &st = c->replay.stats();
    if (on_device && chain_ran && c->run_modes) {  // the kept items the replay worked on stay where they are until this slot is scanned again
        c->last_items_info = b.chain_r.info;
        c->last_items_n = b.chain_r.n_items;
    }
    st.samples_processed += b.pending_n;
    st.gated_offsets += cnt[kCntGated];
    st.general_strips += cnt[kCntGeneralStrips];
    st.patched_strips += cnt[kCntFixedStrips];
    st.split_strips += cnt[kCntSplitStrips];
    st.sort_fallbacks = c->sort_fallbacks;
    if (std::getenv("RSB_VERBOSE")) std::fprintf(stderr, "readsb_amd: %u strips converted on the general path, %u patched from the full UC8 table, %u split, %u gated offsets\n", cnt[kCntGeneralStrips], cnt[kCntFixedStrips], cnt[kCntSplitStrips], cnt[kCntGated]);
    if (c->d_wave_times) {
        std::vector<uint64_t> wt((size_t)a.n_regions * 4);
        HIP_TRY(c, hipMemcpy(wt.data(), c->d_wave_times, wt.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
        if (FILE *f = std::fopen(std::getenv("RSB_WAVE_TIMES"), "wb")) {
            std::fwrite(wt.data(), sizeof(uint64_t), wt.size(), f);
            std::fclose(f);
        }
    }
    st.live_candidates += cnt[kCntItemsTotal];
    st.survivors += on_device ? cnt[kCntAccepted] : cnt[kCntSurvivors];
    st.modeac_candidates += cnt[kCntAc];

    if (!c->block_mode) {
        c->stream_pos += b.pending_n;
        if (b.pending_n == 0 || b.pending_n % c->block_len != 0) c->finished = true;
    }

    if (n_out) *n_out = n;
    if (wire) {
        if (!delivered) wire->nbytes = wire_encode_host(wire->wire, wire->flags, c->msgs.data(), n, wire->out, wire->cap);
        if (wire->nbytes > wire->cap) {
            c->err = "rsb_collect_wire: output buffer too small";
            return RSB_EOVERFLOW;
        }
        return RSB_OK;
    }
    if (!delivered) {
        const size_t ncopy = n < cap ? n : cap;
        if (ncopy) std::memcpy(out, c->msgs.data(), ncopy * sizeof(rsb_message));
    }
    if (n > cap) {
        c->err = "rsb_collect: output buffer too small";
        return RSB_EOVERFLOW;
    }
    return RSB_OK;
}

int collect_impl(rsb_ctx *c, rsb_message *out, size_t cap, size_t *n_out, WireRequest *wire) {
    if (n_out) *n_out = 0;
    if (!c->any_pending()) {
        c->err = "rsb_collect: nothing submitted";
        return RSB_ESTATE;
    }
    rsb_slot &b = c->collect_slot();
    b.pending = false;
    ++c->collected;
    c->last_n = b.pending_n;
    c->last_args = b.pending_args;
    c->last_list_dev = nullptr;
    c->last_items_info = nullptr;
    c->last_g0 = b.g0;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    // everything from here on runs on the context's own stream, behind this batch's scan and beside the next batch's
    // (rsb_collect_many: the channels' shared chain ran on one stream, and whatever is left to do for this channel follows it there)
    const bool foreign = b.chain_enqueued && b.chain_foreign;  // rsb_collect_prepare_on: the chain is on the caller's stream, the rest follows it there
    hipStream_t s = c->chain_stream ? c->chain_stream : (foreign ? b.chain_on : c->replay_stream);
    HIP_TRY(c, hipStreamWaitEvent(s, b.ev[3], 0));
    if (!b.chain_enqueued) HIP_TRY(c, hipEventRecord(b.rev[0], s));  // (collect_ahead has recorded it otherwise)
    if (c->run_modes && !b.pruned && !b.chain_enqueued) {  // against the superset as it is now (the caller may have widened it)
        const int rc = enqueue_prune(c, b, b.pending_args, s);
        if (rc != RSB_OK) return rc;
    }
    CollectRun run{c, b, s, out, cap, wire, b.pending_args, (size_t)((b.pending_n + c->block_len - 1) / c->block_len)};
    c->h_blk_counts.resize(run.nblocks + 1);
    c->timing.replay_iterations = 0;
    int rc = run.replay();         // the batch's messages decided: on the device (records in b.d_out), or candidates for the host
    if (rc != RSB_OK) return rc;
    rc = run.fetch();              // ... and brought to where they are wanted
    if (rc != RSB_OK) return rc;
    run.stage_times();
    rc = run.host_list();          // (only what the device did not deliver)
    if (rc != RSB_OK) return rc;
    return run.account(n_out);
}

}  // namespace

extern "C" {

static int collect_prepare_impl(rsb_ctx *c, hipStream_t s, bool foreign);

int rsb_collect_prepare(rsb_ctx *c) {
    if (!c) return RSB_EINVAL;
    return collect_prepare_impl(c, c->replay_stream, false);
}

int rsb_collect_prepare_on(rsb_ctx *c, void *stream) {
    if (!c) return RSB_EINVAL;
    return collect_prepare_impl(c, (hipStream_t)stream, true);
}

static int collect_prepare_impl(rsb_ctx *c, hipStream_t s, bool foreign) {
    if (!c->any_pending()) {
        c->err = "rsb_collect_prepare: nothing submitted";
        return RSB_ESTATE;
    }
    rsb_slot &nb = c->collect_slot();
    if (nb.chain_enqueued || !chain_needs_no_host(c, nb)) return RSB_OK;  // (the second case: rsb_collect does it step by step)
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    nb.chain_foreign = foreign;
    HIP_TRY(c, hipStreamWaitEvent(s, nb.ev[3], 0));
    HIP_TRY(c, hipEventRecord(nb.rev[0], s));
    if (!nb.pruned) {
        const int rc = enqueue_prune(c, nb, nb.pending_args, s);
        if (rc != RSB_OK) return rc;
    }
    c->sorted_buf = 0;
    const int rc = replay_enqueue(c, nb, kept_items_bound(c, nb), true, s);
    if (rc != RSB_OK) nb.chain_enqueued = nb.chain_foreign = false;
    return rc;
}

int rsb_collect_fence(rsb_ctx *c, void *stream) {
    if (!c) return RSB_EINVAL;
    if (!c->any_pending()) {
        c->err = "rsb_collect_fence: nothing submitted";
        return RSB_ESTATE;
    }
    rsb_slot &nb = c->collect_slot();
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    // behind the prepared chain's last kernel and readbacks; without a prepared chain, behind the batch's scan
    HIP_TRY(c, hipStreamWaitEvent((hipStream_t)stream, nb.chain_enqueued ? nb.rev[2] : nb.ev[3], 0));
    return RSB_OK;
}

int rsb_collect_wait(rsb_ctx *c) {
    if (!c) return RSB_EINVAL;
    if (c->copy_in_flight) {
        c->copy_in_flight = false;
        HIP_TRY(c, hipSetDevice(c->cfg.device));
        HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    }
    return RSB_OK;
}

int rsb_collect(rsb_ctx *c, rsb_message *out, size_t cap, size_t *n_out) {
    if (!c || (!out && cap)) return RSB_EINVAL;
    int rc = rsb_collect_wait(c);  // a copy left in flight by rsb_collect_begin
    if (rc != RSB_OK) return rc;
    return collect_impl(c, out, cap, n_out, nullptr);
}

// ---- the rest of decodeModesMessage on the device (decode_fields.hip) ----

static int grow(rsb_ctx *c, void **p, size_t *cap, size_t want_bytes) {
    if (want_bytes <= *cap) return RSB_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t bytes = want_bytes + want_bytes / 4;
    HIP_TRY(c, hipMalloc(p, bytes));
    *cap = bytes;
    return RSB_OK;
}

int rsb_decode_fields_device(rsb_ctx *c, const rsb_message *msgs_dev, size_t n, uint64_t batch_pos, uint32_t block_samples, rsb_decoded *out_dev,
                             uint32_t *cpr_filtered_dev, void *stream) {
    if (!c || (n && (!msgs_dev || !out_dev)) || n > 0xffffffffull || (uintptr_t)msgs_dev % 4 || (uintptr_t)out_dev % 4) return RSB_EINVAL;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    HIP_TRY(c, launch_decode_fields(msgs_dev, (uint32_t)n, batch_pos, block_samples, out_dev, cpr_filtered_dev, (hipStream_t)stream));
    return RSB_OK;
}

// records (host or device) -> c->d_decoded -> dec (host), on the copy stream
static int decode_to_host(rsb_ctx *c, const rsb_message *msgs, bool msgs_on_device, size_t n, uint64_t batch_pos, rsb_decoded *dec) {
    if (!n) return RSB_OK;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    hipStream_t s = c->copy_stream;
    int rc = grow(c, (void **)&c->d_decoded, &c->decoded_cap, n * sizeof(rsb_decoded));
    if (rc != RSB_OK) return rc;
    if (!msgs_on_device) {
        rc = grow(c, (void **)&c->d_dec_in, &c->dec_in_cap, n * sizeof(rsb_message));
        if (rc != RSB_OK) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->d_dec_in, msgs, n * sizeof(rsb_message), hipMemcpyHostToDevice, s));
        msgs = c->d_dec_in;
    }
    HIP_TRY(c, launch_decode_fields(msgs, (uint32_t)n, batch_pos, c->block_len, c->d_decoded, nullptr, s));
    HIP_TRY(c, hipMemcpyAsync(dec, c->d_decoded, n * sizeof(rsb_decoded), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return RSB_OK;
}

int rsb_decode_fields(rsb_ctx *c, const rsb_message *msgs, size_t n, uint64_t batch_pos, rsb_decoded *out) {
    if (!c || (n && (!msgs || !out)) || n > 0xffffffffull) return RSB_EINVAL;
    int rc = rsb_collect_wait(c);  // the copy stream is about to be used
    if (rc != RSB_OK) return rc;
    return decode_to_host(c, msgs, false, n, batch_pos, out);
}

int rsb_collect_decoded(rsb_ctx *c, rsb_message *out, rsb_decoded *dec, size_t cap, size_t *n_out) {
    size_t n = 0;
    int rc = rsb_collect(c, out, cap, &n);
    if (n_out) *n_out = n;
    if (rc != RSB_OK || !dec) return rc;
    // the list as it lies on the device when the batch was delivered from there; otherwise (host replay, host merge) the caller's copy
    return decode_to_host(c, c->last_list_dev ? c->last_list_dev : out, c->last_list_dev != nullptr, n, c->last_g0, dec);
}

int rsb_collect_begin(rsb_ctx *c, rsb_message *out, size_t cap, size_t *n_out) {
    if (!c || (!out && cap)) return RSB_EINVAL;
    int rc = rsb_collect_wait(c);
    if (rc != RSB_OK) return rc;
    c->defer_copy_wait = true;
    rc = collect_impl(c, out, cap, n_out, nullptr);
    c->defer_copy_wait = false;
    if (rc != RSB_OK && rc != RSB_EOVERFLOW) (void)rsb_collect_wait(c);
    return rc;
}

// All channels' scans as one launch (k_scan_many): same device and format, plain contexts (the converters of dc_filter and the
// SC16Q11 table write a context-owned buffer first), every channel distinct, at most one channel per scan workgroup.
static bool one_scan_launch_possible(rsb_ctx *const *ctx, size_t nch, const uint64_t *nsamples) {
    if (nch < 2 || std::getenv("RSB_NO_SCAN_MANY")) return false;
    const rsb_ctx *c0 = ctx[0];
    if (nch > (size_t)c0->scan_grid) return false;
    for (size_t i = 0; i < nch; ++i) {
        const rsb_ctx *c = ctx[i];
        if (!c || c->cfg.device != c0->cfg.device || c->cfg.input_format != c0->cfg.input_format || c->cfg.dc_filter || c->q11_table_bits ||
            c->block_mode || !c->run_modes || c->scan_grid != c0->scan_grid)
            return false;
        for (size_t k = 0; k < i; ++k)
            if (ctx[k] == c) return false;
    }
    (void)nsamples;
    return true;
}

int rsb_submit_many(rsb_ctx *const *ctx, size_t nch, const void *const *iq_dev, const uint64_t *nsamples, void *stream) {
    if (!ctx || !iq_dev || !nsamples) return RSB_EINVAL;
    if (one_scan_launch_possible(ctx, nch, nsamples)) {
        for (size_t i = 0; i < nch; ++i) {  // nothing is submitted unless every channel can be
            const int rc = submit_checks(ctx[i], iq_dev[i], nsamples[i]);
            if (rc != RSB_OK) return rc;
        }
        rsb_ctx *c0 = ctx[0];
        hipStream_t s = (hipStream_t)stream;
        HIP_TRY(c0, hipSetDevice(c0->cfg.device));
        if (nch > c0->many_args_cap) {
            if (c0->d_many_args) (void)hipFree(c0->d_many_args);
            if (c0->h_many_args) (void)hipHostFree(c0->h_many_args);
            c0->d_many_args = nullptr;
            c0->h_many_args = nullptr;
            c0->many_args_cap = 0;
            HIP_TRY(c0, hipMalloc((void **)&c0->d_many_args, 2 * nch * sizeof(BatchArgs)));  // two halves, as the pinned staging
            HIP_TRY(c0, hipHostMalloc((void **)&c0->h_many_args, 2 * nch * sizeof(BatchArgs), hipHostMallocDefault));
            c0->many_args_cap = nch;
            for (hipEvent_t &ev : c0->many_args_read)
                if (!ev) HIP_TRY(c0, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        }
        // the halves (pinned staging and device copy) used two calls ago: their scan has long run, but the host may be that far ahead
        // of the stream -- or the calls may go to different streams, where the device copy of one call must not be overwritten while
        // the scan of the other still loads its arguments: the event is recorded behind the scan launch
        const unsigned half = c0->many_calls++ & 1u;
        if (c0->many_args_pending[half]) HIP_TRY(c0, hipEventSynchronize(c0->many_args_read[half]));
        BatchArgs *h = c0->h_many_args + half * c0->many_args_cap;
        const uint32_t groups = (uint32_t)c0->scan_grid / (uint32_t)nch;  // >= 1: one_scan_launch_possible
        uint32_t max_regions = 0;
        for (size_t i = 0; i < nch; ++i) {
            rsb_ctx *c = ctx[i];
            h[i] = c->batch_args(c->submit_slot(), iq_dev[i], nsamples[i], c->cfg.input_format, nullptr, (int)groups);
            if (nsamples[i] && nsamples[i] % c->block_len == 0) h[i].carry_out = c->d_carry[(c->submitted + 1) % rsb_ctx::kCarries];
            max_regions = std::max(max_regions, h[i].n_regions);
            const int rc = enqueue_before_scan(c, c->submit_slot(), h[i], s, true);
            if (rc != RSB_OK) return rc;
        }
        BatchArgs *d_args = c0->d_many_args + half * c0->many_args_cap;
        HIP_TRY(c0, hipMemcpyAsync(d_args, h, nch * sizeof(BatchArgs), hipMemcpyHostToDevice, s));
        HIP_TRY(c0, launch_submit_prepare_many(d_args, (uint32_t)nch, max_regions, s));
        // the one launch is timed once, on the first channel's events: rsb_timing.scan_ms is the whole launch there and 0 elsewhere
        for (size_t i = 0; i < nch; ++i) {
            rsb_slot &b = ctx[i]->submit_slot();
            b.scan_share = i == 0 ? 1.0f : 0.0f;
            b.scan_timed = i == 0;
        }
        HIP_TRY(c0, hipEventRecord(c0->submit_slot().ev[0], s));
        HIP_TRY(c0, launch_scan_many(c0->cfg.input_format, d_args, (uint32_t)nch, groups, s));
        HIP_TRY(c0, hipEventRecord(c0->many_args_read[half], s));
        c0->many_args_pending[half] = true;
        HIP_TRY(c0, hipEventRecord(c0->submit_slot().ev[1], s));  // the scan's end, before any channel's Mode A/C kernels
        for (size_t i = 0; i < nch; ++i) {
            rsb_ctx *c = ctx[i];
            const int rc = enqueue_after_scan(c, c->submit_slot(), h[i], nsamples[i], s, true);
            if (rc != RSB_OK) return rc;
        }
        return RSB_OK;
    }
    for (size_t i = 0; i < nch; ++i) {
        if (!ctx[i]) return RSB_EINVAL;
        const int rc = rsb_submit(ctx[i], iq_dev[i], nsamples[i], stream);
        if (rc != RSB_OK) return rc;  // (channels before i stay submitted: collect them)
    }
    return RSB_OK;
}

// All channels' prune / replay / emit chains as ONE chain of launches (kernels.hip "the *_many forms") on one stream and one host
// wait: possible when every channel's batch is the usual kind (chain_needs_no_host), nothing of it is enqueued yet, and the channels
// are distinct contexts of one device and one sample format.
static bool one_chain_possible(rsb_ctx *const *ctx, size_t nch) {
    if (nch < 2 || std::getenv("RSB_NO_CHAIN_MANY")) return false;
    const rsb_ctx *c0 = ctx[0];
    for (size_t i = 0; i < nch; ++i) {
        rsb_ctx *c = ctx[i];
        if (c->cfg.device != c0->cfg.device || c->cfg.input_format != c0->cfg.input_format || !c->any_pending() || c->cfg.dc_filter) return false;
        const rsb_slot &b = c->collect_slot();
        if (!b.pending || b.chain_enqueued || b.pruned || !chain_needs_no_host(c, b) || b.pending_args.fmt != ctx[0]->collect_slot().pending_args.fmt) return false;
        for (size_t k = 0; k < i; ++k)
            if (ctx[k] == c) return false;
    }
    return true;
}

// enqueues the shared chain and waits for it; afterwards every channel's slot looks as if its own chain had been enqueued and
// had run (b.chain_enqueued, the readbacks in b.h_pin): rsb_collect_begin takes it from there, on the shared stream
// (s: ctx[0]'s own stream, or -- foreign -- the caller's, rsb_collect_many_prepare_on; the wait and the hand-over of what came back are
// finish_one_chain's)
static int run_one_chain(rsb_ctx *const *ctx, size_t nch, hipStream_t s, bool foreign) {
    rsb_ctx *c0 = ctx[0];
    HIP_TRY(c0, hipSetDevice(c0->cfg.device));
    ChainDims d;
    std::memset(&d, 0, sizeof(d));
    d.nch = (uint32_t)nch;
    for (size_t i = 0; i < nch; ++i) {
        const int rc = rsb_collect_wait(ctx[i]);  // (a record copy left in flight reads the d_out this chain writes, on one-slot contexts)
        if (rc != RSB_OK) return rc;
        // a channel collected by itself before has left its last kernels (the fold into its filter tables) on another stream -- its
        // own, or the caller's -- which the shared stream knows nothing of: wait for them on the device
        const int rc2 = order_tables(ctx[i], s);
        if (rc2 != RSB_OK) return rc2;
    }
    if (nch > c0->chain_args_cap) {
        if (c0->d_chain_args) (void)hipFree(c0->d_chain_args);
        if (c0->h_chain_args) (void)hipHostFree(c0->h_chain_args);
        c0->d_chain_args = nullptr;
        c0->h_chain_args = nullptr;
        c0->chain_args_cap = 0;
        HIP_TRY(c0, hipMalloc((void **)&c0->d_chain_args, nch * sizeof(ChainArgs)));
        HIP_TRY(c0, hipHostMalloc((void **)&c0->h_chain_args, nch * sizeof(ChainArgs), hipHostMallocDefault));
        c0->chain_args_cap = nch;
    }
    // behind every channel's scan (channels submitted on one stream: behind the last one's)
    bool one_stream = true;
    for (size_t i = 1; i < nch; ++i) one_stream = one_stream && ctx[i]->collect_slot().pending_stream == ctx[0]->collect_slot().pending_stream;
    size_t last = 0;
    for (size_t i = 1; i < nch; ++i)
        if (ctx[i]->collect_slot().submit_seq > ctx[last]->collect_slot().submit_seq) last = i;
    for (size_t i = 0; i < nch; ++i)
        if (!one_stream || i == last) HIP_TRY(ctx[i], hipStreamWaitEvent(s, ctx[i]->collect_slot().ev[3], 0));
    rsb_slot &b0 = c0->collect_slot();
    HIP_TRY(c0, hipEventRecord(b0.rev[0], s));
    {
        // the tables of the channels whose filter moved outside the device replay: the small updates (mirror_filter's first case)
        // together -- one upload, two launches; a channel whose tables must be rebuilt from scratch goes by itself
        std::vector<size_t> dirty;
        size_t words = 0;
        for (size_t i = 0; i < nch; ++i) {
            rsb_ctx *c = ctx[i];
            if (!c->filter_dirty) continue;
            if (!c->tables_valid || c->table_members.size() > 6 * IcaoFilter::kSlots) {
                const int rc = mirror_filter(c, s);
                if (rc != RSB_OK) return rc;
                continue;
            }
            c->replay.filter().contents(c->h_addrs);
            words += c->h_addrs.size() + c->table_members.size();
            dirty.push_back(i);
        }
        if (!dirty.empty()) {
            const size_t head = (dirty.size() * sizeof(FilterUpdate) + 15) / 16 * 4;  // words
            if (head + words > c0->mirror_cap) {
                if (c0->d_mirror) (void)hipFree(c0->d_mirror);
                if (c0->h_mirror) (void)hipHostFree(c0->h_mirror);
                c0->d_mirror = c0->h_mirror = nullptr;
                c0->mirror_cap = 0;
                const size_t cap = 2 * (head + words) + 4096;
                HIP_TRY(c0, hipMalloc((void **)&c0->d_mirror, cap * sizeof(uint32_t)));
                HIP_TRY(c0, hipHostMalloc((void **)&c0->h_mirror, cap * sizeof(uint32_t), hipHostMallocDefault));
                c0->mirror_cap = cap;
            }
            // (the staging is free: every call of this function ends with a wait for the stream)
            FilterUpdate *u = (FilterUpdate *)c0->h_mirror;
            size_t at = head;
            uint32_t max_stale = 0, max_members = 0;
            for (size_t k = 0; k < dirty.size(); ++k) {
                rsb_ctx *c = ctx[dirty[k]];
                const size_t ns = c->table_members.size(), nm = c->h_addrs.size();
                u[k].t0 = c->d_first_add[0];
                u[k].t1 = c->d_first_add[1];
                u[k].stale = c0->d_mirror + at;
                u[k].n_stale = (uint32_t)ns;
                if (ns) std::memcpy(c0->h_mirror + at, c->table_members.data(), ns * sizeof(uint32_t));
                at += ns;
                u[k].members = c0->d_mirror + at;
                u[k].n_members = (uint32_t)nm;
                if (nm) std::memcpy(c0->h_mirror + at, c->h_addrs.data(), nm * sizeof(uint32_t));
                at += nm;
                max_stale = std::max(max_stale, (uint32_t)ns);
                max_members = std::max(max_members, (uint32_t)nm);
                c->table_members = c->h_addrs;
                c->tables_valid = true;
                c->first_cur = 0;
                c->filter_dirty = false;
            }
            HIP_TRY(c0, hipMemcpyAsync(c0->d_mirror, c0->h_mirror, at * sizeof(uint32_t), hipMemcpyHostToDevice, s));
            HIP_TRY(c0, launch_filter_update_many((const FilterUpdate *)c0->d_mirror, (uint32_t)dirty.size(), max_stale, max_members, s));
        }
    }
    for (size_t i = 0; i < nch; ++i) {
        rsb_ctx *c = ctx[i];
        rsb_slot &b = c->collect_slot();
        b.pruned = true;
        b.pruned_in_order = true;  // (chain_needs_no_host: prune_is_ordered)
        c->sorted_buf = 0;
        fill_replay_args(c, b, kept_items_bound(c, b), true);
        ReplayArgs &r = b.chain_r;
        r.t_cur = c->d_first_add[c->first_cur];
        r.t_next = c->d_first_add[c->first_cur ^ 1];
        b.chain_folded = true;
        b.time_from = &b0;
        b.h_pin[0] = 0;
        c0->h_chain_args[i].a = b.pending_args;
        c0->h_chain_args[i].r = r;
        // Many channels' bitmaps together do not stay in the L2s (2 MiB each): the prune then looks into each channel's hashed 4 KB
        // summary first, which a single small batch does not bother to build (rsb_ctx::summary_from)
        if (nch >= 4 && c->d_summary && !std::getenv("RSB_NO_CHAIN_SUMMARY")) c0->h_chain_args[i].a.adder_summary = c->d_summary;
        const BatchArgs &a = c0->h_chain_args[i].a;
        if (a.n) {
            d.max_regions = std::max(d.max_regions, a.n_regions);
            d.max_chunks = std::max(d.max_chunks, a.n_chunks);
            d.max_items = std::max(d.max_items, r.n_items);
            d.max_nblocks = std::max(d.max_nblocks, r.nblocks);
            d.any_summary = d.any_summary || a.adder_summary != nullptr;
        }
    }
    const uint32_t stride = 1 + kCntCount + d.max_nblocks + 2 * rsb_ctx::kPinAdders;
    if ((size_t)stride * nch > c0->chain_back_cap) {
        if (c0->d_chain_back) (void)hipFree(c0->d_chain_back);
        if (c0->h_chain_back) (void)hipHostFree(c0->h_chain_back);
        c0->d_chain_back = nullptr;
        c0->h_chain_back = nullptr;
        c0->chain_back_cap = 0;
        const size_t words = (size_t)stride * nch + 4096;
        HIP_TRY(c0, hipMalloc((void **)&c0->d_chain_back, words * sizeof(uint32_t)));
        HIP_TRY(c0, hipHostMalloc((void **)&c0->h_chain_back, words * sizeof(uint32_t), hipHostMallocDefault));
        c0->chain_back_cap = words;
    }
    if (d.max_nblocks) {
        HIP_TRY(c0, hipMemcpyAsync(c0->d_chain_args, c0->h_chain_args, nch * sizeof(ChainArgs), hipMemcpyHostToDevice, s));
        HIP_TRY(c0, launch_chain_prune_many(c0->d_chain_args, d, s));
        HIP_TRY(c0, hipEventRecord(b0.rev[1], s));
        HIP_TRY(c0, launch_chain_replay_many(c0->collect_slot().pending_args.fmt, c0->d_chain_args, d, c0->d_chain_back, stride, rsb_ctx::kPinAdders, s));
        HIP_TRY(c0, hipMemcpyAsync(c0->h_chain_back, c0->d_chain_back, (size_t)stride * nch * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    } else {
        HIP_TRY(c0, hipEventRecord(b0.rev[1], s));
    }
    HIP_TRY(c0, hipEventRecord(b0.rev[2], s));
    for (size_t i = 0; i < nch; ++i) {
        rsb_slot &b = ctx[i]->collect_slot();
        b.chain_foreign = foreign;
        b.chain_on = s;
    }
    c0->many_prepared = nch;
    c0->many_stream = s;
    c0->many_foreign = foreign;
    c0->many_stride = stride;
    c0->many_nblocks = d.max_nblocks;
    return RSB_OK;
}

static int finish_one_chain(rsb_ctx *const *ctx, size_t nch) {
    rsb_ctx *c0 = ctx[0];
    const uint32_t stride = c0->many_stride;
    ChainDims d;
    d.max_nblocks = c0->many_nblocks;
    c0->many_prepared = 0;
    // (on the caller's stream: the chain's last event, not the stream -- it may hold the caller's next scans by now)
    if (c0->many_foreign)
        HIP_TRY(c0, hipEventSynchronize(c0->collect_slot().rev[2]));
    else
        HIP_TRY(c0, hipStreamSynchronize(c0->many_stream));
    for (size_t i = 0; i < nch && d.max_nblocks; ++i) {  // every channel's part to where replay_finish looks for it
        rsb_ctx *c = ctx[i];
        rsb_slot &b = c->collect_slot();
        const uint32_t *src = c0->h_chain_back + (size_t)i * stride;
        const uint32_t nblk = b.chain_r.nblocks;
        std::memcpy(b.h_pin, src, (1 + kCntCount) * sizeof(uint32_t));
        std::memcpy(b.h_pin + 1 + c->back_blk(), src + 1 + kCntCount, nblk * sizeof(uint32_t));
        const uint32_t nadd = std::min<uint32_t>(src[1 + kCntNewAdders], rsb_ctx::kPinAdders);
        std::memcpy(b.h_pin + 1 + c->back_adders(), src + 1 + kCntCount + d.max_nblocks, (size_t)nadd * sizeof(uint2));
    }
    return RSB_OK;
}

static int collect_many_impl(rsb_ctx *const *ctx, size_t nch, rsb_message *const *out, const size_t *cap, size_t *n_out, bool wait_for_copies);

int rsb_collect_many(rsb_ctx *const *ctx, size_t nch, rsb_message *const *out, const size_t *cap, size_t *n_out) {
    return collect_many_impl(ctx, nch, out, cap, n_out, true);
}

int rsb_collect_many_prepare_on(rsb_ctx *const *ctx, size_t nch, void *stream) {
    if (!ctx) return RSB_EINVAL;
    for (size_t i = 0; i < nch; ++i)
        if (!ctx[i]) return RSB_EINVAL;
    if (nch && ctx[0]->many_prepared) return RSB_OK;
    if (one_chain_possible(ctx, nch)) return run_one_chain(ctx, nch, (hipStream_t)stream, true);
    int first = RSB_OK;
    for (size_t i = 0; i < nch; ++i) {  // (channels the shared chain does not take: each one's own chain, on the same stream)
        const int rc = ctx[i]->any_pending() ? rsb_collect_prepare_on(ctx[i], stream) : RSB_OK;
        if (rc != RSB_OK && first == RSB_OK) first = rc;
    }
    return first;
}

int rsb_collect_many_begin(rsb_ctx *const *ctx, size_t nch, rsb_message *const *out, const size_t *cap, size_t *n_out) {
    return collect_many_impl(ctx, nch, out, cap, n_out, false);
}

static int collect_many_impl(rsb_ctx *const *ctx, size_t nch, rsb_message *const *out, const size_t *cap, size_t *n_out, bool wait_for_copies) {
    if (!ctx || !out || !cap || !n_out) return RSB_EINVAL;
    for (size_t i = 0; i < nch; ++i)
        if (!ctx[i] || (!out[i] && cap[i])) return RSB_EINVAL;
    int first = RSB_OK;
    const bool prepared = ctx[0]->many_prepared == nch && nch > 0;  // rsb_collect_many_prepare_on has enqueued the shared chain
    if (prepared || one_chain_possible(ctx, nch)) {
        int rc = prepared ? RSB_OK : run_one_chain(ctx, nch, ctx[0]->replay_stream, false);
        if (rc == RSB_OK) rc = finish_one_chain(ctx, nch);
        if (rc != RSB_OK) {
            // (nothing was collected: every channel's batch is prepared again from scratch by whoever collects it)
            for (size_t i = 0; i < nch; ++i) {
                rsb_slot &b = ctx[i]->collect_slot();
                (void)hipStreamSynchronize(ctx[0]->many_stream ? ctx[0]->many_stream : ctx[0]->replay_stream);
                b.chain_enqueued = false;
                b.chain_foreign = false;
                b.pruned = false;
                b.time_from = nullptr;
                ctx[i]->filter_dirty = true;
                ctx[i]->tables_valid = false;
            }
            return rc;
        }
        hipStream_t s = ctx[0]->many_stream;
        bool tail_on_shared_stream = false;
        for (size_t i = 0; i < nch; ++i) {
            n_out[i] = 0;
            rsb_ctx *c = ctx[i];
            c->chain_stream = s;
            rc = rsb_collect_begin(c, out[i], cap[i], &n_out[i]);
            c->chain_stream = nullptr;
            // a channel that went on iterating (or fell back to the host path) has left work for its tables on the shared stream
            if (c->timing.replay_iterations != 1 || !c->timing.replay_on_device) tail_on_shared_stream = true;
            if (rc != RSB_OK && first == RSB_OK) first = rc;
        }
        // ... which that channel's own stream knows nothing of: wait for it here (rare)
        if (tail_on_shared_stream && hipStreamSynchronize(s) != hipSuccess && first == RSB_OK) first = RSB_ENODEV;
        for (size_t i = 0; i < nch; ++i) ctx[i]->tables_stream = nullptr;  // (every table operation of this call has been waited for on the host)
        for (size_t i = 0; i < nch && (wait_for_copies || (first != RSB_OK && first != RSB_EOVERFLOW)); ++i) {
            rc = rsb_collect_wait(ctx[i]);
            if (rc != RSB_OK && first == RSB_OK) first = rc;
        }
        return first;
    }
    // every channel's prune / replay / emit chain onto its own stream first: they run side by side ...
    for (size_t i = 0; i < nch; ++i) {
        const int rc = rsb_collect_prepare(ctx[i]);
        if (rc != RSB_OK && first == RSB_OK) first = rc;
    }
    // ... then the host part channel by channel, each channel's records leaving while the next channel is looked at
    for (size_t i = 0; i < nch; ++i) {
        n_out[i] = 0;
        const int rc = rsb_collect_begin(ctx[i], out[i], cap[i], &n_out[i]);
        if (rc != RSB_OK && first == RSB_OK) first = rc;
    }
    for (size_t i = 0; i < nch && (wait_for_copies || (first != RSB_OK && first != RSB_EOVERFLOW)); ++i) {
        const int rc = rsb_collect_wait(ctx[i]);
        if (rc != RSB_OK && first == RSB_OK) first = rc;
    }
    return first;
}

int rsb_collect_wire(rsb_ctx *c, int wire, uint32_t flags, uint8_t *out, size_t cap, size_t *nbytes, size_t *n_msgs) {
    if (!c || (!out && cap) || !nbytes || (wire != RSB_WIRE_BEAST && wire != RSB_WIRE_RAW)) return RSB_EINVAL;
    *nbytes = 0;
    if (rsb_collect_wait(c) != RSB_OK) return RSB_ENODEV;
    WireRequest w{wire, flags, out, cap, 0};
    const int rc = collect_impl(c, nullptr, 0, n_msgs, &w);
    *nbytes = w.nbytes;
    return rc;
}

int rsb_process(rsb_ctx *c, const void *iq_dev, uint64_t nsamples, void *stream, rsb_message *out, size_t cap, size_t *n_out) {
    const int rc = rsb_submit(c, iq_dev, nsamples, stream);
    if (rc != RSB_OK) return rc;
    return rsb_collect(c, out, cap, n_out);
}

int rsb_process_host(rsb_ctx *c, const void *iq_host, uint64_t nsamples, rsb_message *out, size_t cap, size_t *n_out) {
    if (!c || (!iq_host && nsamples)) return RSB_EINVAL;
    const unsigned bps = (c->cfg.input_format == RSB_UC8) ? 2 : 4;
    const size_t bytes = (size_t)nsamples * bps;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (bytes > c->stage_bytes) {
        if (c->d_stage) (void)hipFree(c->d_stage);
        c->d_stage = nullptr;
        c->stage_bytes = 0;
        HIP_TRY(c, hipMalloc(&c->d_stage, bytes));
        c->stage_bytes = bytes;
    }
    if (bytes) HIP_TRY(c, hipMemcpy(c->d_stage, iq_host, bytes, hipMemcpyHostToDevice));
    return rsb_process(c, c->d_stage ? c->d_stage : (const void *)c->d_carry[0], nsamples, nullptr, out, cap, n_out);
}

int rsb_get_batch_magnitudes(rsb_ctx *c, uint16_t *mag_host, uint64_t nsamples, double *block_means) {
    if (!c || !mag_host) return RSB_EINVAL;
    if ((!c->d_magbuf && !c->dc_fused) || c->any_pending() || nsamples > c->last_n) {
        c->err = "rsb_get_batch_magnitudes: only after rsb_collect on a --dcfilter or RSB_FLAG_SC16Q11_TABLE context, at most the batch's samples";
        return RSB_ESTATE;
    }
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (c->dc_fused) {
        // no magnitude buffer exists: the batch's magnitudes are made now, from the batch's samples (still the caller's to keep valid)
        // and the filter states in its slot, by the routine every kernel of the batch used
        if (nsamples * sizeof(uint16_t) > c->dc_mags_cap) {
            if (c->d_dc_mags) (void)hipFree(c->d_dc_mags);
            c->d_dc_mags = nullptr;
            c->dc_mags_cap = 0;
            HIP_TRY(c, hipMalloc((void **)&c->d_dc_mags, nsamples * sizeof(uint16_t) + 16));
            c->dc_mags_cap = nsamples * sizeof(uint16_t);
        }
        HIP_TRY(c, launch_dc_materialize(c->last_args, c->d_dc_mags, nullptr));
        HIP_TRY(c, hipMemcpy(mag_host, c->d_dc_mags, (size_t)nsamples * sizeof(uint16_t), hipMemcpyDeviceToHost));
        return RSB_OK;
    }
    HIP_TRY(c, hipMemcpy(mag_host, c->d_magbuf, (size_t)nsamples * sizeof(uint16_t), hipMemcpyDeviceToHost));
    if (block_means && c->run_ac) {
        const size_t nblk = (size_t)((c->last_n + c->block_len - 1) / c->block_len);
        if (c->cfg.dc_filter) {
            HIP_TRY(c, hipMemcpy(block_means, c->slots[0].d_block_means, nblk * 2 * sizeof(double), hipMemcpyDeviceToHost));
        } else {  // the table converter: from the blocks' integer sums (convert.c:321-327)
            std::vector<uint64_t> sums(2 * nblk);
            HIP_TRY(c, hipMemcpy(sums.data(), c->slots[0].d_block_sums, sums.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
            for (size_t k = 0; k < nblk; ++k) {
                const uint64_t len = std::min<uint64_t>(c->block_len, c->last_n - k * c->block_len);
                block_means[2 * k] = (double)sums[2 * k] / 65536.0 / (double)(unsigned)len;
                block_means[2 * k + 1] = (double)sums[2 * k + 1] / 65535.0 / 65535.0 / (double)(unsigned)len;
            }
        }
    }
    return RSB_OK;
}

int rsb_get_stats(const rsb_ctx *c, rsb_stats *st) {
    if (!c || !st) return RSB_EINVAL;
    *st = const_cast<rsb_ctx *>(c)->replay.stats();
    return RSB_OK;
}

int rsb_get_timing(const rsb_ctx *c, rsb_timing *t) {
    if (!c || !t) return RSB_EINVAL;
    *t = c->timing;
    return RSB_OK;
}

const char *rsb_last_error(const rsb_ctx *c) { return c ? c->err.c_str() : "null context"; }

int rsb_convert_device(rsb_ctx *c, const void *iq_dev, void *mag_dev, uint64_t nsamples, void *stream, double *mean_level,
                       double *mean_power) {
    if (!c || !iq_dev || !mag_dev) return RSB_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    if (c->reset_pending) HIP_TRY(c, hipStreamWaitEvent(s, c->reset_done, 0));  // (rsb_reset zeroes the DC state on the context's own stream)
    const bool want = mean_level || mean_power;
    if (c->cfg.dc_filter) {
        // convert.c:113-162 / :164-213 / :374-423 as a chunked scan; the DC state lives in the context, as in struct converter_state
        const size_t ntiles = (size_t)((nsamples + kDcTileSamples - 1) / kDcTileSamples) + 1, nblk = (size_t)((nsamples + kBlock - 1) / kBlock);
        if (ntiles > c->dc_tiles_cap || nblk + 1 > c->nblocks_cap) {
            c->err = "rsb_convert_device: nsamples exceeds rsb_config.max_batch_samples";
            return RSB_EINVAL;
        }
        HIP_TRY(c, launch_convert_dc(c->cfg.input_format, iq_dev, (uint16_t *)mag_dev, nsamples, c->d_dc_tile_sums, c->d_dc_tile_start, c->d_dc_state,
                                     want ? c->slots[0].d_block_means : nullptr, s));
        if (want) {
            std::vector<double> m(2 * nblk);
            HIP_TRY(c, hipMemcpyAsync(m.data(), c->slots[0].d_block_means, m.size() * sizeof(double), hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            double lvl = 0, pwr = 0;  // one call = one mean (float sum / nsamples in the reference); blocks weighted by their length
            for (size_t b = 0; b < nblk; ++b) {
                const double len = (double)(b + 1 < nblk ? (uint64_t)kBlock : nsamples - (uint64_t)b * kBlock);
                lvl += m[2 * b] * len;
                pwr += m[2 * b + 1] * len;
            }
            if (mean_level) *mean_level = lvl / (double)nsamples;
            if (mean_power) *mean_power = pwr / (double)nsamples;
        }
        return RSB_OK;
    }
    const bool float_path = c->cfg.input_format != RSB_UC8 && !c->q11_table_bits;
    if (want && !c->d_sums) HIP_TRY(c, hipMalloc((void **)&c->d_sums, 2 * sizeof(uint64_t)));
    if (want && !float_path) HIP_TRY(c, hipMemsetAsync(c->d_sums, 0, 2 * sizeof(uint64_t), s));
    if (c->q11_table_bits)  // convert.c:297-328
        HIP_TRY(c, launch_convert_q11t(iq_dev, (uint16_t *)mag_dev, nsamples, c->q11_table_bits, want ? c->d_sums : nullptr, s));
    else
        HIP_TRY(c, launch_convert(c->cfg.input_format, iq_dev, (uint16_t *)mag_dev, nsamples, c->d_lut, (want && !float_path) ? c->d_sums : nullptr, s));
    if (want && float_path) HIP_TRY(c, launch_block_means_float(c->cfg.input_format, iq_dev, nsamples, nsamples, (double *)c->d_sums, s));
    if (want) {
        uint64_t sums[2];
        HIP_TRY(c, hipMemcpyAsync(sums, c->d_sums, sizeof(sums), hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        if (float_path) {  // convert.c:254-260 / :371-377
            double means[2];
            std::memcpy(means, sums, sizeof(means));
            if (mean_level) *mean_level = means[0];
            if (mean_power) *mean_power = means[1];
        } else {  // convert.c:104-110 / :321-327
            if (mean_level) *mean_level = sums[0] / 65536.0 / (unsigned)nsamples;
            if (mean_power) *mean_power = sums[1] / 65535.0 / 65535.0 / (unsigned)nsamples;
        }
    }
    return RSB_OK;
}

uint32_t rsb_checksum(const uint8_t *msg, int bits) {
    if (!msg || (bits != 56 && bits != 112)) return 0xffffffffu;  // crc.c:72-73 asserts instead
    return CrcTables::checksum(msg, bits);
}

int rsb_diagnose(const rsb_ctx *c, uint32_t syndrome, int bits, int bitpos[2]) {
    if (!c || (bits != 56 && bits != 112)) return -1;
    const SyndromeEntry *e = c->replay.crc().diagnose(syndrome, bits);
    if (!e) return -1;
    if (bitpos) {
        bitpos[0] = e->errors > 0 ? e->bit[0] : -1;
        bitpos[1] = e->errors > 1 ? e->bit[1] : -1;
    }
    return e->errors;
}

static_assert(sizeof(rsb_candidate) == sizeof(Survivor) && sizeof(rsb_ac_candidate) == sizeof(AcCand), "public record layouts");

int rsb_crc_table_size(int fix_bits, int bits) {
    if (bits != 56 && bits != 112) return -1;
    return (int)tables_for(fix_bits).table(bits).size();
}

int rsb_crc_table_entry(int fix_bits, int bits, int idx, uint32_t *syndrome, int *errors, int bitpos[2]) {
    if (bits != 56 && bits != 112) return RSB_EINVAL;
    const std::vector<SyndromeEntry> &t = tables_for(fix_bits).table(bits);
    if (idx < 0 || (size_t)idx >= t.size()) return RSB_EINVAL;
    if (syndrome) *syndrome = t[idx].syndrome;
    if (errors) *errors = t[idx].errors;
    if (bitpos) {
        bitpos[0] = t[idx].bit[0];
        bitpos[1] = t[idx].bit[1];
    }
    return RSB_OK;
}

const uint16_t *rsb_uc8_lut(void) { return uc8_magnitude_lut(); }

int rsb_host_replay(int fix_bits, uint64_t g0, const rsb_candidate *cands, size_t ncands, const rsb_ac_candidate *ac, size_t nac,
                    rsb_message *out, size_t cap, size_t *n_out) {
    if ((!cands && ncands) || (!ac && nac) || (!out && cap) || g0 % kBlock) return RSB_EINVAL;
    Replayer r(fix_bits);
    std::vector<Survivor> s(ncands);
    std::vector<AcCand> a(nac);
    if (ncands) std::memcpy(s.data(), cands, ncands * sizeof(Survivor));
    if (nac) std::memcpy(a.data(), ac, nac * sizeof(AcCand));
    std::vector<rsb_message> msgs;
    r.run_batch(g0, s, a, msgs);
    if (n_out) *n_out = msgs.size();
    const size_t ncopy = msgs.size() < cap ? msgs.size() : cap;
    if (ncopy) std::memcpy(out, msgs.data(), ncopy * sizeof(rsb_message));
    return msgs.size() > cap ? RSB_EOVERFLOW : RSB_OK;
}

}  // extern "C"

// ---- behind include/readsb_dropin.h ------------------------------------------------------------------

int rsb_internal_create_block_ctx(int fix_bits, int device, unsigned max_length, rsb_ctx **out) {
    rsb_config cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.input_format = RSB_UC8;  // placeholder: block contexts are fed magnitudes
    cfg.fix_bits = fix_bits;
    cfg.mode_ac = 1;
    cfg.device = device;
    cfg.queue_div = 8;
    cfg.max_batch_samples = max_length ? max_length : kBlock;
    const int rc = rsb_create(&cfg, out);
    if (rc != RSB_OK) return rc;
    rsb_ctx *c = *out;
    c->block_mode = true;
    if (hipMalloc((void **)&c->d_means, 2 * sizeof(double)) != hipSuccess) {
        rsb_destroy(c);
        *out = nullptr;
        return RSB_ENOMEM;
    }
    return RSB_OK;
}

int rsb_internal_submit_block(rsb_ctx *c, const uint16_t *mag_dev, unsigned length, bool modes, bool modeac, double mean_level, double mean_power) {
    if (!c || !c->block_mode || !mag_dev || length > c->cfg.max_batch_samples) return RSB_EINVAL;
    if (c->any_pending()) return RSB_ESTATE;
    HIP_TRY(c, hipSetDevice(c->cfg.device));
    c->run_modes = modes;
    c->run_ac = modeac;
    rsb_slot &b = c->submit_slot();
    BatchArgs a = c->batch_args(b, mag_dev + kTrailing, length, kFmtMag16, mag_dev);
    a.iq_aligned = 0;  // magnitudes go through the per-sample path of the scan kernel
    if (modeac) {
        const double means[2] = {mean_level, mean_power};
        HIP_TRY(c, hipMemcpy(c->d_means, means, sizeof(means), hipMemcpyHostToDevice));
        a.block_means = c->d_means;
    }
    return enqueue_batch(c, b, a, length, nullptr);
}
