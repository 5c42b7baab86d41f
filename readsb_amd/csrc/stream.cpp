// rsb_stream: block-at-a-time hand-over of a live source to the batched device path (include/readsb_amd.h,
// "live sources").  Built on the public entry points only (rsb_submit / rsb_collect / rsb_seek); what it adds is the
// staging: two pinned host buffers and two device buffers of blocks_per_batch blocks, filled and consumed in turn.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>

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
    if ((uint64_t)blocks_per_batch * kBlock > c->cfg.max_batch_samples) {
        c->err = "rsb_stream_open: blocks_per_batch * 131072 exceeds rsb_config.max_batch_samples";
        return RSB_EINVAL;
    }
    rsb_stream *st = new (std::nothrow) rsb_stream;
    if (!st) return RSB_ENOMEM;
    st->c = c;
    st->sink = sink;
    st->user = user;
    st->blocks = blocks_per_batch;
    st->bps = (c->cfg.input_format == RSB_UC8) ? 2 : 4;
    const size_t bytes = (size_t)blocks_per_batch * kBlock * st->bps;
    // at most one message per 64 samples: a 56-bit frame and its preamble span 154 samples, a Mode A/C reply skips 69
    st->out_cap = (size_t)blocks_per_batch * (kBlock / 64);
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
    if (!st || (!iq_host && nsamples) || nsamples > kBlock) return RSB_EINVAL;
    rsb_ctx *c = st->c;
    if (st->ended) {
        c->err = "rsb_stream_push: the stream ended with a short block";
        return RSB_ESTATE;
    }
    ST_TRY(st, hipSetDevice(c->cfg.device));
    if (dropped_before) {
        if (dropped_before % kBlock != 0) {
            c->err = "rsb_stream_push: only whole 131072-sample blocks can be dropped";
            return RSB_EINVAL;
        }
        int rc = submit_staged(st);
        if (rc == RSB_OK) rc = drain(st);
        if (rc != RSB_OK) return rc;
        rc = rsb_seek(c, c->submit_pos + dropped_before, nullptr, 0);  // zero carry: sdr_plutosdr.c:226-228
        if (rc != RSB_OK) return rc;
    }
    if (!nsamples) return RSB_OK;
    if (st->filled + nsamples > (uint64_t)st->blocks * kBlock) {  // an earlier submit failed and left the buffer full
        c->err = "rsb_stream_push: the staging buffer is full (an earlier push failed)";
        return RSB_ESTATE;
    }
    std::memcpy(st->h_stage[st->cur] + (size_t)st->filled * st->bps, iq_host, (size_t)nsamples * st->bps);
    st->filled += nsamples;
    if (nsamples < kBlock) {
        st->ended = true;
        const int rc = submit_staged(st);
        return rc == RSB_OK ? drain(st) : rc;
    }
    return st->filled == (uint64_t)st->blocks * kBlock ? submit_staged(st) : RSB_OK;
}

int rsb_stream_flush(rsb_stream *st) {
    if (!st) return RSB_EINVAL;
    ST_TRY(st, hipSetDevice(st->c->cfg.device));
    const int rc = submit_staged(st);
    return rc == RSB_OK ? drain(st) : rc;
}

int rsb_stream_close(rsb_stream *st) {
    if (!st) return RSB_EINVAL;
    const int rc = rsb_stream_flush(st);
    if (rc != RSB_OK) {  // leave nothing pending on the context
        size_t n = 0;
        while (st->c->any_pending() && rsb_collect(st->c, st->h_out, st->out_cap, &n) == RSB_OK) {}
    }
    release(st);
    return rc;
}

}  // extern "C"
