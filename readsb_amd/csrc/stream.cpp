// rsb_stream: block-at-a-time hand-over of a live source to the batched device path (include/readsb_amd.h,
// "live sources").  Built on the public entry points only (rsb_submit / rsb_collect / rsb_seek); what it adds is the
// staging: two pinned host buffers and two device buffers of blocks_per_batch blocks, filled and consumed in turn.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <vector>

#include "ctx.hpp"

using namespace rsb;

struct rsb_stream {
    rsb_ctx *c = nullptr;
    rsb_stream_sink sink = nullptr;
    void *user = nullptr;
    uint32_t blocks = 0;        // per batch
    unsigned bps = 2;           // bytes per sample
    uint8_t *h_stage[2] = {nullptr, nullptr};  // pinned
    void *d_stage[2] = {nullptr, nullptr};
    rsb_message *h_out = nullptr;              // pinned: the emit kernel writes the records here itself
    size_t out_cap = 0;
    hipStream_t copy_stream = nullptr;
    int cur = 0;                // buffer being filled
    uint64_t filled = 0;        // samples staged in h_stage[cur]
    unsigned in_flight = 0;     // batches submitted and not collected
    bool ended = false;         // a short block was pushed
    // rsb_stream_push_bladerf (the state of handle_bladerf_samples, sdr_bladerf.c:289-290, 322-327):
    uint64_t bladerf_next = 0;  // nextTimestamp: the hardware timestamp the next sub-block should carry (0 = none seen yet)
    bool bladerf = false;       // the stream is fed bladeRF buffers
    bool reanchor = false;      // the last buffer was short: the next one starts a new run of blocks whatever its timestamp
    std::vector<uint8_t> tail;  // the 326 samples behind the last published buffer's data (raw IQ): the next buffer's overlap (:323-324)
};

namespace {

#define ST_TRY(st, expr)                                                              \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            (st)->c->err = std::string("rsb_stream: " #expr ": ") + hipGetErrorString(e_); \
            return RSB_ENODEV;                                                          \
        }                                                                             \
    } while (0)

int collect_one(rsb_stream *st) {
    size_t n = 0;
    const int rc = rsb_collect(st->c, st->h_out, st->out_cap, &n);
    --st->in_flight;
    if (rc != RSB_OK) return rc;
    if (n && st->sink) st->sink(st->user, st->h_out, n);
    return RSB_OK;
}

// what is staged -> one batch; keeps at most one older batch in flight behind it
int submit_staged(rsb_stream *st) {
    if (!st->filled) return RSB_OK;
    const int k = st->cur;
    // a --dcfilter context holds one batch at a time (one magnitude buffer): finish the batch in flight first
    if (st->in_flight >= (unsigned)st->c->nslots) {
        const int rc = collect_one(st);
        if (rc != RSB_OK) return rc;
    }
    ST_TRY(st, hipMemcpyAsync(st->d_stage[k], st->h_stage[k], (size_t)st->filled * st->bps, hipMemcpyHostToDevice, st->copy_stream));
    const int rc = rsb_submit(st->c, st->d_stage[k], st->filled, st->copy_stream);
    if (rc != RSB_OK) return rc;
    ++st->in_flight;
    st->filled = 0;
    st->cur ^= 1;
    // the buffers now current belong to the batch before this one: finished with once that batch is collected
    return st->in_flight == 2 ? collect_one(st) : RSB_OK;
}

// append samples to the staging buffer, submitting every time it is full (any count: sub-blocks of a framed transfer)
int stage_samples(rsb_stream *st, const uint8_t *src, uint64_t n) {
    const uint64_t cap = (uint64_t)st->blocks * st->c->block_len;
    while (n) {
        if (st->filled >= cap) {  // an earlier submit failed and left the buffer full
            st->c->err = "rsb_stream: the staging buffer is full (an earlier push failed)";
            return RSB_ESTATE;
        }
        const uint64_t k = n < cap - st->filled ? n : cap - st->filled;
        if (src) {
            std::memcpy(st->h_stage[st->cur] + (size_t)st->filled * st->bps, src, (size_t)k * st->bps);
            src += (size_t)k * st->bps;
        } else {
            std::memset(st->h_stage[st->cur] + (size_t)st->filled * st->bps, 0, (size_t)k * st->bps);  // zero IQ (SC16 formats: magnitude 0)
        }
        st->filled += k;
        n -= k;
        if (st->filled == cap) {
            const int rc = submit_staged(st);
            if (rc != RSB_OK) return rc;
        }
    }
    return RSB_OK;
}

int drain(rsb_stream *st) {
    while (st->in_flight) {
        const int rc = collect_one(st);
        if (rc != RSB_OK) return rc;
    }
    return RSB_OK;
}

void release(rsb_stream *st) {
    for (int k = 0; k < 2; ++k) {
        if (st->h_stage[k]) (void)hipHostFree(st->h_stage[k]);
        if (st->d_stage[k]) (void)hipFree(st->d_stage[k]);
    }
    if (st->h_out) (void)hipHostFree(st->h_out);
    if (st->copy_stream) (void)hipStreamDestroy(st->copy_stream);
    delete st;
}

}  // namespace

extern "C" {

int rsb_stream_open(rsb_ctx *c, uint32_t blocks_per_batch, rsb_stream_sink sink, void *user, rsb_stream **out) {
    if (!c || !out || !blocks_per_batch) return RSB_EINVAL;
    *out = nullptr;
    if (c->block_mode) return RSB_EINVAL;
    if (c->any_pending()) {
        c->err = "rsb_stream_open: a batch is pending";
        return RSB_ESTATE;
    }
    if ((uint64_t)blocks_per_batch * c->block_len > c->cfg.max_batch_samples) {
        c->err = "rsb_stream_open: blocks_per_batch blocks (131072 samples, or rsb_config.block_samples) exceed rsb_config.max_batch_samples";
        return RSB_EINVAL;
    }
    rsb_stream *st = new (std::nothrow) rsb_stream;
    if (!st) return RSB_ENOMEM;
    st->c = c;
    st->sink = sink;
    st->user = user;
    st->blocks = blocks_per_batch;
    st->bps = (c->cfg.input_format == RSB_UC8) ? 2 : 4;
    const size_t bytes = (size_t)blocks_per_batch * c->block_len * st->bps;
    // at most one message per 64 samples: a 56-bit frame and its preamble span 154 samples, a Mode A/C reply skips 69
    st->out_cap = (size_t)blocks_per_batch * (c->block_len / 64);
    hipError_t e = hipSetDevice(c->cfg.device);
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        e = hipHostMalloc((void **)&st->h_stage[k], bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(&st->d_stage[k], bytes);
    }
    if (e == hipSuccess) e = hipHostMalloc((void **)&st->h_out, st->out_cap * sizeof(rsb_message), hipHostMallocDefault);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st->copy_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        c->err = std::string("rsb_stream_open: ") + hipGetErrorString(e);
        release(st);
        return e == hipErrorOutOfMemory ? RSB_ENOMEM : RSB_ENODEV;
    }
    *out = st;
    return RSB_OK;
}

int rsb_stream_push(rsb_stream *st, const void *iq_host, uint32_t nsamples, uint64_t dropped_before) {
    if (!st || (!iq_host && nsamples) || nsamples > st->c->block_len) return RSB_EINVAL;
    rsb_ctx *c = st->c;
    const uint32_t L = c->block_len;
    if (st->ended) {
        c->err = "rsb_stream_push: the stream ended with a short block";
        return RSB_ESTATE;
    }
    ST_TRY(st, hipSetDevice(c->cfg.device));
    if (dropped_before) {
        if (dropped_before % L != 0) {
            c->err = "rsb_stream_push: only whole blocks can be dropped";
            return RSB_EINVAL;
        }
        int rc = submit_staged(st);
        if (rc == RSB_OK) rc = drain(st);
        if (rc != RSB_OK) return rc;
        rc = rsb_seek(c, c->submit_pos + dropped_before, nullptr, 0);  // zero carry: sdr_plutosdr.c:226-228
        if (rc != RSB_OK) return rc;
    }
    if (!nsamples) return RSB_OK;
    if (st->filled + nsamples > (uint64_t)st->blocks * L) {  // an earlier submit failed and left the buffer full
        c->err = "rsb_stream_push: the staging buffer is full (an earlier push failed)";
        return RSB_ESTATE;
    }
    std::memcpy(st->h_stage[st->cur] + (size_t)st->filled * st->bps, iq_host, (size_t)nsamples * st->bps);
    st->filled += nsamples;
    if (nsamples < L) {
        st->ended = true;
        const int rc = submit_staged(st);
        return rc == RSB_OK ? drain(st) : rc;
    }
    return st->filled == (uint64_t)st->blocks * L ? submit_staged(st) : RSB_OK;
}

// handle_bladerf_samples (sdr_bladerf.c:283-422) on one transfer buffer, sub-block by sub-block, with the reference's own state:
// nextTimestamp starts at 0 ("none", :289, 366), so the first sub-block ever is taken unchecked and the second one -- unless the
// hardware counter started at 0 -- is a mismatch that restarts the buffer there; a mismatch drops what the buffer held so far
// (:370-373) and the buffer's sampleTimestamp becomes the new timestamp's (:382); the published mag_buf holds the sub-blocks from the
// last restart on; its overlap is the 326 samples behind the previous published buffer's data whatever happened in between
// (:323-324: outbuf->dropped is 0 for every slot this driver publishes, :413).  A published buffer is one block of the context:
// full ones are staged and leave in batches; a short one (restart, wrong magic) ends the run of blocks -- it leaves as the last
// block of a ragged batch, and the next buffer starts a new run (rsb_seek to its position, carry = that overlap).
int rsb_stream_push_bladerf(rsb_stream *st, const void *buf, size_t bytes, uint32_t block_size, uint32_t decimation, uint32_t *overruns) {
    if (!st || !buf || block_size <= 16 || (block_size - 16) % 4 != 0 || !decimation) return RSB_EINVAL;
    rsb_ctx *c = st->c;
    if (c->cfg.input_format != RSB_SC16Q11 && c->cfg.input_format != RSB_SC16) {
        c->err = "rsb_stream_push_bladerf: the context must take 4-byte samples (SC16Q11 is what the bladeRF delivers)";
        return RSB_EINVAL;
    }
    const uint32_t per_block = (block_size - 16) / 4, L = c->block_len;  // sdr_bladerf.c:335
    const size_t nsub = bytes / block_size;
    if ((uint64_t)nsub * per_block != L || (c->run_ac && c->mean_sub != per_block)) {
        c->err = "rsb_stream_push_bladerf: rsb_config.block_samples must be the samples of one transfer buffer (524288 / block_size sub-blocks of "
                 "(block_size - 16) / 4 samples: 129024 or 130048) and, with mode_ac, mean_sub_samples those of one sub-block";
        return RSB_EINVAL;
    }
    if (st->ended) {
        c->err = "rsb_stream_push_bladerf: the stream has ended";
        return RSB_ESTATE;
    }
    ST_TRY(st, hipSetDevice(c->cfg.device));
    st->bladerf = true;
    if (st->tail.empty()) st->tail.assign((size_t)kTrailing * 4, 0);  // the ring is calloc'ed (readsb.c:198-207): zeros before the first buffer
    const uint8_t *p = (const uint8_t *)buf;
    auto rd32 = [](const uint8_t *q) { return (uint32_t)q[0] | (uint32_t)q[1] << 8 | (uint32_t)q[2] << 16 | (uint32_t)q[3] << 24; };
    size_t first_kept = 0, nblocks = 0;  // the published buffer = sub-blocks [first_kept, nblocks)
    uint64_t next = st->bladerf_next, seg_ts = 0;
    uint32_t nover = 0;
    for (size_t off = 0; off + block_size <= bytes; off += block_size, ++nblocks) {
        const uint8_t *h = p + off;
        if (rd32(h) != 0x12344321u) break;  // :347-353
        const uint64_t ts = (uint64_t)rd32(h + 4) | (uint64_t)rd32(h + 8) << 32;
        if (rd32(h + 12) & 1u) ++nover;  // BLADERF_META_STATUS_OVERRUN
        if (next && next != ts) {        // :366-375, backwards as well as forwards
            first_kept = nblocks;
            next = ts;
        }
        if (nblocks == first_kept) seg_ts = next;  // :380-383
        next += (uint64_t)per_block * decimation;  // :391
    }
    if (overruns) *overruns = nover;
    st->bladerf_next = next;
    if (nblocks == first_kept) return RSB_OK;  // nothing converted: nothing published (:398)
    const uint64_t seg_len = (uint64_t)(nblocks - first_kept) * per_block;
    // sampleTimestamp = nextTimestamp * 12e6 / sample_rate / decimation (:382) = 5 x the position in samples (hardware timestamps
    // that are not multiples of the decimation would land between two samples: the position is rounded down)
    const uint64_t pos = seg_ts / decimation;
    int rc;
    if (st->reanchor || pos != c->submit_pos + st->filled) {
        rc = submit_staged(st);
        if (rc == RSB_OK) rc = drain(st);
        if (rc != RSB_OK) return rc;
        rc = rsb_seek(c, pos, st->tail.data(), 0);
        if (rc != RSB_OK) return rc;
        st->reanchor = false;
    }
    for (size_t k = first_kept; k < nblocks; ++k) {
        rc = stage_samples(st, p + k * (size_t)block_size + 16, per_block);
        if (rc != RSB_OK) return rc;
    }
    // the next buffer's overlap: the last 326 samples of (this buffer's overlap ++ its data)
    {
        const size_t tb = st->tail.size();
        std::vector<uint8_t> joined(st->tail);
        for (size_t k = first_kept; k < nblocks; ++k)
            joined.insert(joined.end(), p + k * (size_t)block_size + 16, p + k * (size_t)block_size + 16 + (size_t)per_block * 4);
        std::memcpy(st->tail.data(), joined.data() + joined.size() - tb, tb);
    }
    if (seg_len < L) {  // a short buffer: the last block of this run
        rc = submit_staged(st);
        if (rc == RSB_OK) rc = drain(st);
        if (rc != RSB_OK) return rc;
        st->reanchor = true;
    }
    return RSB_OK;
}

// everything == false: only the whole blocks that are staged go (a framed source stages sub-blocks of any length, and a batch
// that is not a whole number of blocks would end the recording); the rest moves to the front of the other staging buffer
static int flush_impl(rsb_stream *st, bool everything) {
    ST_TRY(st, hipSetDevice(st->c->cfg.device));
    const uint64_t whole = st->filled / st->c->block_len * st->c->block_len, rest = everything ? 0 : st->filled - whole;
    const uint8_t *tail = st->h_stage[st->cur] + (size_t)whole * st->bps;
    st->filled -= rest;
    int rc = submit_staged(st);  // flips st->cur (the copy that reads the old buffer does not mind our reading it too)
    if (rc == RSB_OK) rc = drain(st);
    if (rest) {
        if (rc == RSB_OK && whole) {
            std::memcpy(st->h_stage[st->cur], tail, (size_t)rest * st->bps);
            st->filled = rest;
        } else {
            st->filled += rest;  // nothing was submitted (or it failed): the samples stay where they are
        }
    }
    return rc;
}

int rsb_stream_flush(rsb_stream *st) {
    if (!st) return RSB_EINVAL;
    return flush_impl(st, !st->bladerf);
}

int rsb_stream_close(rsb_stream *st) {
    if (!st) return RSB_EINVAL;
    const int rc = flush_impl(st, true);
    if (rc != RSB_OK) {  // leave nothing pending on the context
        size_t n = 0;
        while (st->c->any_pending() && rsb_collect(st->c, st->h_out, st->out_cap, &n) == RSB_OK) {}
    }
    release(st);
    return rc;
}

}  // extern "C"
