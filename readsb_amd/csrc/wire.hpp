// Wire formats of the message list: the bytes modesQueueOutput (net_io.c:1249-1276) hands to
// Beast and raw clients, produced straight from rsb_message records.  One byte-level description, used by the
// host encoder (wire.cpp) and by the device kernels (kernels.hip): a message is a sequence of put(byte) calls.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>

#include "../../include/readsb_amd.h"

#ifdef __HIPCC__
#define RSB_WIRE_FN __host__ __device__ inline __attribute__((always_inline))
#else
#define RSB_WIRE_FN inline
#endif

namespace rsb {

// net_io.c:808-812
RSB_WIRE_FN int beast_rssi(double level) {
    int sig = (int)round(sqrt(level) * 255);
    if (level > 0 && sig < 1) sig = 1;
    if (sig > 255) sig = 255;
    return sig;
}

// Calls put(b) for every byte of message m in format `wire`; nothing for a message the reference does not
// forward: two corrected bits with net_verbatim off (net_io.c:1258,1264), a length the writer does not know (:772-780).
template <class Put>
RSB_WIRE_FN void wire_bytes(const rsb_message &m, int wire, uint32_t flags, Put &&put) {
    if (m.correctedbits >= 2) return;
    const int len = m.msgbits / 8;  // net_io.c:761,860
    if (wire == RSB_WIRE_BEAST) {
        uint8_t type;
        if (len == 7) type = '2';
        else if (len == 14) type = '3';
        else if (len == 2) type = '1';
        else return;
        auto esc = [&](uint8_t b) {  // net_io.c:783-786
            put(b);
            if (b == 0x1a) put(b);
        };
        put((uint8_t)0x1a);
        put(type);
        for (int sh = 40; sh >= 0; sh -= 8) esc((uint8_t)(m.timestamp >> sh));
        esc((uint8_t)beast_rssi(m.signal_level));
        for (int j = 0; j < len; ++j) esc(m.msg[j]);
    } else {
        auto hex = [](uint32_t v) { return (uint8_t)(v < 10 ? '0' + v : 'A' + v - 10); };  // net_io.c:848-852
        if ((flags & RSB_WIRE_MLAT) && m.timestamp) {                                         // :868-872
            put((uint8_t)'@');
            for (int sh = 44; sh >= 0; sh -= 4) put(hex((uint32_t)(m.timestamp >> sh) & 15u));
        } else {
            put((uint8_t)'*');
        }
        for (int j = 0; j < len && j < 14; ++j) {
            put(hex(m.msg[j] >> 4));
            put(hex(m.msg[j] & 15u));
        }
        put((uint8_t)';');
        put((uint8_t)'\n');
    }
}

size_t wire_encode_host(int wire, uint32_t flags, const rsb_message *msgs, size_t n, uint8_t *out, size_t cap);

}  // namespace rsb
