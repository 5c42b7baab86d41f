// dropin.cpp -- include/readsb_dropin.h: the reference's per-block entry points on top of the kernels.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <unistd.h>

#include <cerrno>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/readsb_dropin.h"
#include "ctx.hpp"

using namespace rsb;

// Layouts of the reference's x86-64 build: struct mag_buf 48 bytes {sampleTimestamp 0, mean_level 8, mean_power 16,
// dropped 24, length 28, sysTimestamp 32, data 40}, struct errorinfo 12 bytes.  integration/gpu_path.c asserts the same against the
// reference's own headers where those are present.
#include <cstddef>
static_assert(sizeof(rsbd_mag_buf) == 48 && offsetof(rsbd_mag_buf, sampleTimestamp) == 0 && offsetof(rsbd_mag_buf, mean_level) == 8 &&
                  offsetof(rsbd_mag_buf, mean_power) == 16 && offsetof(rsbd_mag_buf, dropped) == 24 && offsetof(rsbd_mag_buf, length) == 28 &&
                  offsetof(rsbd_mag_buf, sysTimestamp) == 32 && offsetof(rsbd_mag_buf, data) == 40,
              "struct mag_buf (readsb.h:298-311)");
static_assert(sizeof(rsbd_errorinfo) == 12 && offsetof(rsbd_errorinfo, errors) == 4 && offsetof(rsbd_errorinfo, bit) == 8, "struct errorinfo (crc.h:32-38)");
static_assert(sizeof(rsb_message) == 56 && offsetof(rsb_message, msg) == 8 && offsetof(rsb_message, msgbits) == 22 && offsetof(rsb_message, crc) == 24 &&
                  offsetof(rsb_message, score) == 36 && offsetof(rsb_message, signal_level) == 48,
              "rsb_message: packed, 14 dwords (k_replay_emit writes it as such)");

// ================================================================================== convert.h

struct rsbd_converter_state {
    int format;
    int device;
    rsb_ctx *ctx;  // owns the tables (UC8 LUT) and the sums buffer
    void *d_iq = nullptr;
    uint16_t *d_mag = nullptr;
    size_t cap = 0;  // samples
};

namespace {

// what -DSC16Q11_TABLE_BITS=n is to the reference's convert.c (a build option, so process-wide here): 0 = the float converter
int g_sc16q11_table_bits = 0;

uint32_t table_flag(int format, int filter_dc) {
    // convert.c:451-456: the table converter takes the place of convert_sc16q11_nodc only
    return (format == RSB_SC16Q11 && !filter_dc) ? RSB_FLAG_SC16Q11_TABLE(g_sc16q11_table_bits) : 0u;
}

void convert_entry(void *iq_data, uint16_t *mag_data, unsigned nsamples, rsbd_converter_state *st, double *out_mean_level, double *out_mean_power) {
    // convert.c:63-423: nsamples IQ pairs in, nsamples magnitudes out, optional block means
    if (!st || !iq_data || !mag_data || nsamples == 0) {
        // convert.c:104-110 divides by nsamples; the reference is never called with 0
        if (out_mean_level) *out_mean_level = 0;
        if (out_mean_power) *out_mean_power = 0;
        return;
    }
    if (hipSetDevice(st->device) != hipSuccess) return;
    const unsigned bps = st->format == RSB_UC8 ? 2 : 4;
    if (nsamples > st->cap) {
        if (st->d_iq) (void)hipFree(st->d_iq);
        if (st->d_mag) (void)hipFree(st->d_mag);
        st->d_iq = nullptr;
        st->d_mag = nullptr;
        st->cap = 0;
        if (hipMalloc(&st->d_iq, (size_t)nsamples * bps) != hipSuccess || hipMalloc((void **)&st->d_mag, (size_t)nsamples * 2) != hipSuccess) {
            std::fprintf(stderr, "readsb_amd: converter: device allocation of %u samples failed\n", nsamples);
            return;
        }
        st->cap = nsamples;
    }
    if (hipMemcpy(st->d_iq, iq_data, (size_t)nsamples * bps, hipMemcpyHostToDevice) != hipSuccess) return;
    double lvl = 0, pwr = 0;
    const bool want = out_mean_level || out_mean_power;
    if (rsb_convert_device(st->ctx, st->d_iq, st->d_mag, nsamples, nullptr, want ? &lvl : nullptr, want ? &pwr : nullptr) != RSB_OK) {
        std::fprintf(stderr, "readsb_amd: converter: %s\n", rsb_last_error(st->ctx));
        return;
    }
    if (hipMemcpy(mag_data, st->d_mag, (size_t)nsamples * 2, hipMemcpyDeviceToHost) != hipSuccess) return;
    if (out_mean_level) *out_mean_level = lvl;
    if (out_mean_power) *out_mean_power = pwr;
}

}  // namespace

extern "C" rsbd_iq_convert_fn rsbd_init_converter(int format, double sample_rate, int filter_dc, rsbd_converter_state **out_state) {
    if (out_state) *out_state = nullptr;
    if (filter_dc && sample_rate != 2400000.0) {
        // convert.c:480-488 derives dc_a / dc_b from the sample rate; the device converters carry the 2.4 MS/s constants
        // (the only rate readsb runs this path at, readsb.c:193)
        std::fprintf(stderr, "no suitable converter for format=%d dc=%d at sample rate %.0f (the DC filter is built for 2400000)\n", format, filter_dc, sample_rate);
        return nullptr;
    }
    if (format < RSB_UC8 || format > RSB_SC16Q11 || !out_state) {
        std::fprintf(stderr, "no suitable converter for format=%d dc=%d\n", format, filter_dc);  // convert.c:460-464
        return nullptr;
    }
    rsb_config cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.input_format = format;
    cfg.dc_filter = filter_dc ? 1 : 0;
    cfg.fix_bits = 0;
    cfg.max_batch_samples = filter_dc ? (2u << 20) : 4096;  // the context is used for its tables (and the DC filter's scan arrays) only
    cfg.flags = RSB_FLAG_HOST_REPLAY | table_flag(format, filter_dc);
    rsb_ctx *ctx = nullptr;
    if (rsb_create(&cfg, &ctx) != RSB_OK) return nullptr;  // message already printed
    rsbd_converter_state *st = new (std::nothrow) rsbd_converter_state;
    if (!st) {
        rsb_destroy(ctx);
        return nullptr;
    }
    st->format = format;
    st->device = 0;
    st->ctx = ctx;
    *out_state = st;
    return convert_entry;
}

extern "C" int rsbd_set_sc16q11_table_bits(int bits) {
    if (bits < 0 || bits > 11) return -1;
    g_sc16q11_table_bits = bits;
    return 0;
}

extern "C" void rsbd_cleanup_converter(rsbd_converter_state *st) {
    if (!st) return;
    (void)hipSetDevice(st->device);
    if (st->d_iq) (void)hipFree(st->d_iq);
    if (st->d_mag) (void)hipFree(st->d_mag);
    rsb_destroy(st->ctx);
    delete st;
}

// ================================================================================== crc.h

namespace {

struct CrcGlobals {  // the statics of crc.c
    std::mutex mu;
    CrcTables *tables = nullptr;
    std::vector<rsbd_errorinfo> short_, long_;
    rsbd_errorinfo no_errors = {0, 0, {0, 0}};  // crc.c:28
};
CrcGlobals &crc_globals() {
    static CrcGlobals g;
    return g;
}

void fill(std::vector<rsbd_errorinfo> &dst, const std::vector<SyndromeEntry> &src) {
    dst.clear();
    for (const SyndromeEntry &e : src) dst.push_back(rsbd_errorinfo{e.syndrome, e.errors, {e.bit[0], e.bit[1]}});
}

}  // namespace

extern "C" void rsbd_modesChecksumInit(int fixBits) {
    CrcGlobals &g = crc_globals();
    std::lock_guard<std::mutex> lock(g.mu);
    delete g.tables;
    g.tables = new CrcTables(fixBits);
    fill(g.short_, g.tables->table(56));
    fill(g.long_, g.tables->table(112));
}

extern "C" uint32_t rsbd_modesChecksum(uint8_t *msg, int bitlen) {
    if (!msg || (bitlen != 56 && bitlen != 112)) return 0xffffffffu;  // the reference asserts (crc.c:72-73)
    return CrcTables::checksum(msg, bitlen);
}

extern "C" rsbd_errorinfo *rsbd_modesChecksumDiagnose(uint32_t syndrome, int bitlen) {
    CrcGlobals &g = crc_globals();
    if (syndrome == 0) return &g.no_errors;  // crc.c:389-390
    if (!g.tables || (bitlen != 56 && bitlen != 112)) return nullptr;
    const SyndromeEntry *e = g.tables->diagnose(syndrome, bitlen);
    if (!e) return nullptr;
    const std::vector<SyndromeEntry> &t = g.tables->table(bitlen);
    std::vector<rsbd_errorinfo> &mine = bitlen == 56 ? g.short_ : g.long_;
    return &mine[(size_t)(e - t.data())];
}

extern "C" void rsbd_modesChecksumFix(uint8_t *msg, rsbd_errorinfo *info) {
    if (!info || !msg) return;  // crc.c:413-414
    for (int i = 0; i < info->errors; ++i) msg[info->bit[i] >> 3] ^= uint8_t(1u << (7 - (info->bit[i] & 7)));
}

extern "C" void rsbd_crcCleanupTables(void) {
    CrcGlobals &g = crc_globals();
    std::lock_guard<std::mutex> lock(g.mu);
    delete g.tables;
    g.tables = nullptr;
    g.short_.clear();
    g.long_.clear();
}

// ================================================================================== demod_2400.h

struct rsbd_modes {
    rsb_ctx *ctx = nullptr;
    uint16_t *d_mag = nullptr;
    unsigned max_length = 0;
    std::vector<rsb_message> out;
    std::vector<rsb_decoded> fields;  // rsbd_demodulate2400*_decoded
};

extern "C" rsbd_modes *rsbd_modesInit(int nfix_crc, int device, unsigned max_length) {
    if (max_length == 0) max_length = kBlock;
    if (max_length > kBlock) {
        std::fprintf(stderr, "readsb_amd: mag_buf.length above MODES_MAG_BUF_SAMPLES (%u) is not supported\n", kBlock);
        return nullptr;
    }
    rsbd_modes *m = new (std::nothrow) rsbd_modes;
    if (!m) return nullptr;
    if (rsb_internal_create_block_ctx(nfix_crc, device, max_length, &m->ctx) != RSB_OK ||
        hipMalloc((void **)&m->d_mag, ((size_t)max_length + kTrailing) * sizeof(uint16_t)) != hipSuccess) {
        std::fprintf(stderr, "readsb_amd: cannot set up the demodulator on HIP device %d (there is no CPU fallback)\n", device);
        if (m->ctx) rsb_destroy(m->ctx);
        delete m;
        return nullptr;
    }
    m->max_length = max_length;
    m->out.resize(max_length / 64 + 64);
    return m;
}

extern "C" void rsbd_modesCleanup(rsbd_modes *m) {
    if (!m) return;
    if (m->d_mag) (void)hipFree(m->d_mag);
    rsb_destroy(m->ctx);
    delete m;
}

extern "C" void rsbd_icaoFilterInit(rsbd_modes *m) {
    if (m) (void)rsb_reset(m->ctx);
}
extern "C" void rsbd_icaoFilterAdd(rsbd_modes *m, uint32_t addr) {
    if (m) (void)rsb_filter_add(m->ctx, addr);
}
extern "C" void rsbd_icaoFilterExpire(rsbd_modes *m) {
    if (m) (void)rsb_filter_expire(m->ctx);
}

namespace {

void run_block(rsbd_modes *m, rsbd_mag_buf *mag, bool modes, bool modeac, rsbd_use_message_fn use, void *user, rsbd_use_decoded_fn use_decoded = nullptr) {
    if (!m || !mag || !mag->data) return;
    if (mag->length > m->max_length) {
        std::fprintf(stderr, "readsb_amd: mag_buf.length %u exceeds the %u given to rsbd_modesInit: block dropped\n", mag->length, m->max_length);
        return;
    }
    if (hipMemcpy(m->d_mag, mag->data, ((size_t)mag->length + kTrailing) * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess) return;
    size_t n = 0;
    int rc = rsb_internal_submit_block(m->ctx, m->d_mag, mag->length, modes, modeac, mag->mean_level, mag->mean_power);
    if (rc == RSB_OK) {
        if (use_decoded) {
            m->fields.resize(m->out.size());
            rc = rsb_collect_decoded(m->ctx, m->out.data(), m->fields.data(), m->out.size(), &n);
        } else
            rc = rsb_collect(m->ctx, m->out.data(), m->out.size(), &n);
        if (rc == RSB_EOVERFLOW && n > m->out.size()) {
            std::fprintf(stderr, "readsb_amd: %zu messages in one block exceed the %zu provided for; the rest is dropped\n", n, m->out.size());
            n = m->out.size();
            rc = RSB_OK;
        }
    }
    if (rc != RSB_OK) {
        std::fprintf(stderr, "readsb_amd: demodulator: %s\n", rsb_last_error(m->ctx));
        return;
    }
    for (size_t i = 0; i < n; ++i) {
        rsb_message &mm = m->out[i];
        mm.timestamp += mag->sampleTimestamp;  // demod_2400.c:320 / :649 are relative to the block's own clock
        if (use_decoded) {
            m->fields[i].timestampMsg = mm.timestamp;
            use_decoded(user, &mm, &m->fields[i]);
        } else if (use)
            use(user, &mm);
    }
}

}  // namespace

extern "C" void rsbd_demodulate2400(rsbd_modes *m, rsbd_mag_buf *mag, rsbd_use_message_fn use, void *user) { run_block(m, mag, true, false, use, user); }
extern "C" void rsbd_demodulate2400AC(rsbd_modes *m, rsbd_mag_buf *mag, rsbd_use_message_fn use, void *user) { run_block(m, mag, false, true, use, user); }
extern "C" void rsbd_demodulate2400_decoded(rsbd_modes *m, rsbd_mag_buf *mag, rsbd_use_decoded_fn use, void *user) { run_block(m, mag, true, false, nullptr, user, use); }
extern "C" void rsbd_demodulate2400AC_decoded(rsbd_modes *m, rsbd_mag_buf *mag, rsbd_use_decoded_fn use, void *user) { run_block(m, mag, false, true, nullptr, user, use); }

// ================================================================================== sdr_ifile.c

struct rsbd_ifile {
    int fd = -1;
    bool close_fd = false;
    unsigned bps = 2;
    uint64_t batch_samples = 0;
    rsb_ctx *ctx = nullptr;
    std::vector<uint8_t> readbuf;
    std::vector<rsb_message> out;
};

extern "C" rsbd_ifile *rsbd_ifileOpen(const char *filename, int input_format, int filter_dc, int nfix_crc, int mode_ac, int device, unsigned blocks_per_batch) {
    if (!filename) {
        std::fprintf(stderr, "SDR type 'ifile' requires an --ifile argument\n");  // sdr_ifile.c:116-119
        return nullptr;
    }
    rsbd_ifile *f = new (std::nothrow) rsbd_ifile;
    if (!f) return nullptr;
    if (std::strcmp(filename, "-") == 0) {
        f->fd = STDIN_FILENO;  // sdr_ifile.c:121-123
    } else if ((f->fd = open(filename, O_RDONLY)) < 0) {
        std::fprintf(stderr, "ifile: could not open %s: %s\n", filename, std::strerror(errno));  // sdr_ifile.c:124-127
        delete f;
        return nullptr;
    } else {
        f->close_fd = true;
    }
    if (blocks_per_batch == 0) blocks_per_batch = 64;
    f->bps = (input_format == RSB_UC8) ? 2 : 4;  // sdr_ifile.c:129-136
    f->batch_samples = (uint64_t)blocks_per_batch * kBlock;
    rsb_config cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.input_format = input_format;
    cfg.dc_filter = filter_dc ? 1 : 0;
    cfg.fix_bits = nfix_crc;
    cfg.mode_ac = mode_ac ? 1 : 0;
    cfg.device = device;
    cfg.max_batch_samples = f->batch_samples;
    cfg.flags = table_flag(input_format, filter_dc);
    if (rsb_create(&cfg, &f->ctx) != RSB_OK) {  // unknown format: "no suitable converter ..." (sdr_ifile.c:149-153)
        rsbd_ifileClose(f);
        return nullptr;
    }
    f->readbuf.resize((size_t)f->batch_samples * f->bps);
    f->out.resize((size_t)(f->batch_samples / 32) + 4096);
    return f;
}

extern "C" void rsbd_ifileFilterAdd(rsbd_ifile *f, uint32_t addr) {
    if (f && f->ctx) (void)rsb_filter_add(f->ctx, addr);
}

extern "C" long long rsbd_ifileRun(rsbd_ifile *f, rsbd_use_message_fn use, void *user) {
    if (!f || !f->ctx) return -1;
    long long total = 0;
    bool eof = false;
    while (!eof) {
        // sdr_ifile.c:209-223: read until the buffer is full or the file ends
        size_t have = 0;
        const size_t want = f->readbuf.size();
        while (have < want) {
            const ssize_t nread = read(f->fd, f->readbuf.data() + have, want - have);
            if (nread < 0) {
                if (errno == EINTR) continue;
                std::fprintf(stderr, "ifile: error reading input file: %s\n", std::strerror(errno));
                return -1;
            }
            if (nread == 0) {
                eof = true;
                break;
            }
            have += (size_t)nread;
        }
        const uint64_t nsamples = have / f->bps;  // sdr_ifile.c:225: whole samples only
        // a short or empty batch ends the recording (the reference publishes a last, possibly empty, block too)
        size_t n = 0;
        int rc = rsb_process_host(f->ctx, f->readbuf.data(), nsamples, f->out.data(), f->out.size(), &n);
        if (rc != RSB_OK) {
            std::fprintf(stderr, "ifile: %s\n", rsb_last_error(f->ctx));
            return -1;
        }
        if (use)
            for (size_t i = 0; i < n; ++i) use(user, &f->out[i]);
        total += (long long)nsamples;
    }
    return total;
}

extern "C" void rsbd_ifileClose(rsbd_ifile *f) {
    if (!f) return;
    if (f->close_fd && f->fd >= 0) close(f->fd);  // sdr_ifile.c:256-272
    if (f->ctx) rsb_destroy(f->ctx);
    delete f;
}
