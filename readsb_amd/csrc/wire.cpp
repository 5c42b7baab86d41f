// Host side of the wire formats: Beast / raw encoders over records already on the host, and the
// two-message forwarding rule of useModesMessage (mode_s.c:2163-2175).  No device code here.
#include "wire.hpp"

#include <algorithm>
#include <vector>

#include <new>
#include <unordered_map>
#include <vector>

namespace rsb {

size_t wire_encode_host(int wire, uint32_t flags, const rsb_message *msgs, size_t n, uint8_t *out, size_t cap) {
    size_t pos = 0;
    for (size_t i = 0; i < n; ++i)
        wire_bytes(msgs[i], wire, flags, [&](uint8_t b) {
            if (pos < cap) out[pos] = b;
            ++pos;
        });
    return pos;
}

}  // namespace rsb

// What trackUpdateFromMessage keeps per aircraft for the rule (track.c:102,988): the message count and the first message.
struct rsb_forwarder {
    struct Seen {
        uint32_t messages;
        rsb_message first;
    };
    std::unordered_map<uint32_t, Seen> seen;
};

extern "C" {

int rsb_wire_encode(int wire, uint32_t flags, const rsb_message *msgs, size_t n, uint8_t *out, size_t cap, size_t *nbytes) {
    if ((wire != RSB_WIRE_BEAST && wire != RSB_WIRE_RAW) || (!msgs && n) || (!out && cap) || !nbytes) return RSB_EINVAL;
    *nbytes = rsb::wire_encode_host(wire, flags, msgs, n, out, cap);
    return *nbytes > cap ? RSB_EOVERFLOW : RSB_OK;
}

rsb_forwarder *rsb_forwarder_create(void) { return new (std::nothrow) rsb_forwarder(); }

void rsb_forwarder_destroy(rsb_forwarder *f) { delete f; }

int rsb_forwarder_push(rsb_forwarder *f, const rsb_message *in, size_t n, rsb_message *out, size_t cap, size_t *n_out) {
    if (!f || (!in && n) || (!out && cap) || !n_out) return RSB_EINVAL;
    size_t k = 0;
    auto emit = [&](const rsb_message &m) {
        if (k < cap) out[k] = m;
        ++k;
    };
    for (size_t i = 0; i < n; ++i) {
        const rsb_message &m = in[i];
        if (m.msgtype == 32 || m.addr == 0) {  // track.c:962-971: no aircraft record -> sent unconditionally (mode_s.c:2164)
            emit(m);
            continue;
        }
        auto it = f->seen.find(m.addr);
        if (it == f->seen.end()) it = f->seen.emplace(m.addr, rsb_forwarder::Seen{0, m}).first;  // trackCreateAircraft, track.c:102
        rsb_forwarder::Seen &a = it->second;
        ++a.messages;  // track.c:988
        if (a.messages > 1) {  // mode_s.c:2167-2173
            if (a.messages == 2) emit(a.first);
            emit(m);
        }
    }
    *n_out = k;
    return k > cap ? RSB_EOVERFLOW : RSB_OK;
}

int rsb_merge_by_timestamp(const rsb_message *const *lists, const size_t *counts, size_t nlists, rsb_message *out, size_t cap, size_t *n_out) {
    if ((!lists || !counts) && nlists) return RSB_EINVAL;
    if ((!out && cap) || !n_out) return RSB_EINVAL;
    size_t total = 0;
    for (size_t l = 0; l < nlists; ++l) {
        if (!lists[l] && counts[l]) return RSB_EINVAL;
        total += counts[l];
    }
    *n_out = total;
    if (total > cap) return RSB_EOVERFLOW;
    // A channel's list is in the reference's emission order: per 131072-sample block its Mode S messages, then its Mode A/C
    // messages (readsb.c:883-886), i.e. sorted runs but not one sorted list.  Stable sort of (timestamp, list, position): messages
    // with equal timestamps keep channel order, and within a channel the order they were emitted in.
    struct Ref {
        uint64_t ts;
        uint32_t list;
        uint32_t pos_hi;
        uint64_t pos;
    };
    std::vector<Ref> refs;
    refs.reserve(total);
    bool sorted = true;
    uint64_t last = 0;
    for (size_t l = 0; l < nlists; ++l)
        for (size_t i = 0; i < counts[l]; ++i) {
            const uint64_t ts = lists[l][i].timestamp;
            sorted = sorted && ts >= last;
            last = ts;
            refs.push_back(Ref{ts, (uint32_t)l, 0, i});
        }
    if (!sorted) std::stable_sort(refs.begin(), refs.end(), [](const Ref &a, const Ref &b) { return a.ts < b.ts; });
    for (size_t k = 0; k < total; ++k) out[k] = lists[refs[k].list][refs[k].pos];
    return RSB_OK;
}

}  // extern "C"
