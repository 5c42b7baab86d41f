"""Seeded scenarios shared by the CPU and GPU tests (sizes the oracle finishes in well under a second)."""
from readsb_amd.synth import Scenario

B = 131072

SCENARIOS = {
    # BASELINE config 1 at reduced size: DF11/DF17 traffic, mixed SNR, every 7th frame with a 1-bit error
    "cfg1_df11_df17": (Scenario(2_400_000, seed=1, mix={"df17": 0.7, "df11": 0.3}, collide_every=0, df11_iid_every=0),
                       dict(nfix=1, modeac=False)),
    # all DFs, near-collisions, DF11 with interrogator ids, 1-bit errors
    "mixed_fix1": (Scenario(2_000_000, seed=7), dict(nfix=1, modeac=False)),
    # BASELINE config 5 at reduced size: dense, low SNR, 2-bit errors, Mode A/C, --aggressive
    "dense_fix2_ac": (Scenario(3_000_000, seed=2, frames_per_sec=4000, err1_every=7, err2_every=10, modeac_per_sec=300,
                               amp_lo=0.04, amp_hi=0.54), dict(nfix=2, modeac=True)),
    "dense_fix1_ac": (Scenario(3_000_000, seed=2, frames_per_sec=4000, err1_every=7, err2_every=10, modeac_per_sec=300,
                               amp_lo=0.04, amp_hi=0.54), dict(nfix=1, modeac=True)),
    "nofix": (Scenario(1_000_000, seed=4), dict(nfix=0, modeac=False)),
    "noise_ac": (Scenario(4_000_000, seed=3, frames_per_sec=0, noise_lsb=3.0), dict(nfix=1, modeac=True)),
    # ragged sizes: shorter than one block, shorter than the overlap, exact multiples of a block
    "tiny_200": (Scenario(200, seed=8, frames_per_sec=0), dict(nfix=1, modeac=True)),
    "short_5000": (Scenario(5000, seed=9, frames_per_sec=20000), dict(nfix=1, modeac=True)),
    "exact_2_blocks": (Scenario(2 * B, seed=10, frames_per_sec=3000, modeac_per_sec=300), dict(nfix=1, modeac=True)),
    "block_plus_1": (Scenario(B + 1, seed=12, frames_per_sec=3000), dict(nfix=1, modeac=False)),
    # BASELINE config 3 (no-DC part): bladeRF / Pluto formats
    "sc16": (Scenario(1_000_000, seed=5, fmt="SC16"), dict(nfix=1, modeac=False)),
    "sc16q11": (Scenario(1_000_000, seed=5, fmt="SC16Q11"), dict(nfix=1, modeac=False)),
}

# --dcfilter scenarios (float IIR path: chunked scan on the GPU, compared under a stated tolerance)
SCENARIOS.update({
    "uc8_dc": (Scenario(600_000, seed=6, dc_offset=0.03), dict(nfix=1, modeac=False, dc_filter=True)),
    "sc16_dc": (Scenario(600_000, seed=5, fmt="SC16", dc_offset=0.02), dict(nfix=1, modeac=False, dc_filter=True)),
    "sc16q11_dc_ac": (Scenario(600_000, seed=5, fmt="SC16Q11", dc_offset=0.02, modeac_per_sec=300), dict(nfix=1, modeac=True, dc_filter=True)),
    # --dcfilter at high SNR (SURVEY B.7 ii): strong clean frames only, so no frame sits at the slicer's decision margin and the
    # device's message SET must be the reference's (fields that depend on magnitudes keep the test's tolerance)
    "sc16_dc_highsnr": (Scenario(900_000, seed=9, fmt="SC16", dc_offset=0.02, amp_lo=0.3, amp_hi=0.8, noise_lsb=1.0, err1_every=0,
                                 collide_every=0, df11_iid_every=0, mix={"df17": 0.7, "df11": 0.3}),
                        dict(nfix=1, modeac=False, dc_filter=True, same_set=True)),
    "sc16_ac": (Scenario(600_000, seed=13, fmt="SC16", modeac_per_sec=300), dict(nfix=1, modeac=True)),
})

# the reference built with -DSC16Q11_TABLE_BITS=8 (convert.c:264-328; what some armhf packages ship).  Golden
# lists from oracle/_ref/ref_harness_t8 (the same harness over convert.c compiled with the macro).  Kept apart from SCENARIOS:
# every consumer passes q11_table_bits explicitly.
TABLE_SCENARIOS = {
    "sc16q11_table8_ac": (Scenario(900_000, seed=11, fmt="SC16Q11", modeac_per_sec=300), dict(nfix=1, modeac=True, q11_table_bits=8)),
    "sc16q11_table8_fix2": (Scenario(700_000, seed=12, fmt="SC16Q11", err2_every=9), dict(nfix=2, modeac=False, q11_table_bits=8)),
}
TABLE_CONVERT_CASES = {"sc16q11_table8": ("SC16Q11", False, 100_000, 30_000, 8)}

# scenarios the GPU path reproduces bit for bit
GPU_SCENARIOS = [k for k, (sc, kw) in SCENARIOS.items() if not kw.get("dc_filter")]  # bit-exact ones; --dcfilter has its own test

# converter known-answer cases: name -> (format, dc_filter, samples, samples per converter call)
CONVERT_CASES = {
    "uc8": ("UC8", False, 200_000, 131072),
    "uc8_dc": ("UC8", True, 100_000, 30_000),
    "sc16": ("SC16", False, 100_000, 30_000),
    "sc16_dc": ("SC16", True, 100_000, 30_000),
    "sc16q11": ("SC16Q11", False, 100_000, 30_000),
    "sc16q11_dc": ("SC16Q11", True, 100_000, 30_000),
}


def convert_input(fmt, n):
    """Uniform random IQ over the full code range plus the corner values (convert_benchmark.c:63-80 style)."""
    import numpy as np

    rng = np.random.Generator(np.random.PCG64(99))
    if fmt == "UC8":
        iq = rng.integers(0, 256, size=(n, 2), dtype=np.uint8)
        iq[:6] = [[127, 127], [128, 128], [130, 127], [200, 100], [255, 128], [0, 0]]
        return iq
    lim = 32768 if fmt == "SC16" else 2048
    iq = rng.integers(-lim, lim, size=(n, 2), dtype=np.int64).astype("<i2")
    iq[:6] = [[1000, -2000], [lim - 1, 0], [-lim, -lim], [123, 456], [0, 0], [-1, 1]]
    return iq

# scenarios whose reference Beast / raw byte streams are kept as fixtures (tests/golden/wire_*.npz)
WIRE_SCENARIOS = ["cfg1_df11_df17", "mixed_fix1", "dense_fix2_ac", "sc16_ac"]


# ---- a bladeRF transfer sequence (sdr_bladerf.c:283-422).  block_size 1024: sub-blocks of a 16-byte metadata header +
# 252 SC16Q11 samples, 512 of them per transfer buffer (129024 samples).  The sequence exercises every branch of the handler:
# a hardware counter that does not start at 0 (the reference then drops the first sub-block), buffers in sequence, a forward jump in
# the middle of a buffer, a backward jump, a wrong magic value in the middle of a buffer and the jump that follows it.
BLADERF_BLOCK_SIZE, BLADERF_PER, BLADERF_NSUB = 1024, 252, 512
BLADERF_BUF = BLADERF_PER * BLADERF_NSUB  # 129024
BLADERF_SCENARIO = Scenario(8 * BLADERF_BUF, seed=61, fmt="SC16Q11", frames_per_sec=4000, modeac_per_sec=400, err1_every=6)
BLADERF_KW = dict(nfix=1, modeac=True)


def bladerf_transfers(iq, decimation=1):
    """-> [(first sub-block number, timestamps[512], flags[512], magic_ok[512])] per transfer buffer; sub-block k of the whole
    sequence carries samples iq[k * 252 : (k + 1) * 252]."""
    import numpy as np

    per, nsub, d = BLADERF_PER, BLADERF_NSUB, decimation
    step = per * d
    out = []
    ts = 1_000_000 * d  # the counter has been running before the stream starts
    for b in range(8):
        t = ts + np.arange(nsub, dtype=np.uint64) * np.uint64(step)
        magic = np.ones(nsub, dtype=bool)
        flags = np.zeros(nsub, dtype=np.uint32)
        if b == 1:
            flags[[3, 9]] = 1                       # overruns are only counted
        if b == 3:
            t[100:] += np.uint64(10_000 * d)        # 10000 samples lost in the middle of the buffer
        if b == 5:
            t -= np.uint64(300_000 * d)             # the counter went backwards (device restart)
        if b == 6:
            magic[300] = False                      # the rest of the buffer is skipped
        out.append((b * nsub, t, flags, magic))
        ts = int(t[-1]) + step
    return out


def bladerf_bytes(iq, transfer):
    """one transfer buffer as the bytes libbladeRF hands to the callback (sdr_bladerf.c:339-345)"""
    import numpy as np

    first, t, flags, magic = transfer
    nsub, per = t.size, BLADERF_PER
    out = np.zeros((nsub, BLADERF_BLOCK_SIZE), dtype=np.uint8)
    hdr = np.zeros(nsub, dtype=[("magic", "<u4"), ("ts", "<u8"), ("flags", "<u4")])
    hdr["magic"] = np.where(magic, 0x12344321, 0xdeadbeef)
    hdr["ts"] = t
    hdr["flags"] = flags
    out[:, :16] = hdr.view(np.uint8).reshape(nsub, 16)
    out[:, 16:] = np.ascontiguousarray(iq[first * per:(first + nsub) * per]).view(np.uint8).reshape(nsub, per * 4)
    return out.reshape(-1)


def bladerf_model(transfers, decimation=1):
    """The mag_bufs handle_bladerf_samples publishes for the sequence -- test-side restatement of sdr_bladerf.c:339-404 (the
    handler itself needs libbladeRF and cannot be built here) -> [(first sample in iq, length, sampleTimestamp)]"""
    per = BLADERF_PER
    nxt = 0
    bufs = []
    for first, t, flags, magic in transfers:
        length, start, stamp = 0, 0, 0
        for k in range(t.size):
            if not magic[k]:
                break
            ts = int(t[k])
            if nxt and nxt != ts:  # :366-375
                length = 0
                nxt = ts
            if length == 0:
                stamp = int(nxt * 12e6 / 2.4e6 / decimation)  # :382
                start = (first + k) * per
            length += per
            nxt += per * decimation
        if length:
            bufs.append((start, length, stamp))
    return bufs


# ---- frames for the field decode ------------------------------------------------------------------

DECODE_POOL = [0x4840D6, 0xA1B2C3, 0x3C6DD0, 0x780123, 0x000001, 0xFFFFFE, 0x06A0F2, 0xE48A11]


def crc24(frame, nbytes):
    """Mode S parity (polynomial 0xFFF409) of the first nbytes - 3 bytes, bitwise (test helper; the library's is rsb_checksum)"""
    rem = 0
    for b in bytes(frame[:nbytes - 3]):
        rem ^= b << 16
        for _ in range(8):
            rem = ((rem << 1) ^ 0xFFF409) & 0xFFFFFF if rem & 0x800000 else (rem << 1) & 0xFFFFFF
    return rem


def decode_frames(n, seed=21):
    """n frames [n][14] that exercise every branch of decodeModesMessage's field decode (mode_s.c:557-716, 736-1474): every downlink
    format the path accepts, DF17 / DF18 with every ME type (and every CF), Comm-D, single-bit damage on some DF11 / DF17 / DF18
    frames, the 'cpr_filtered' position (mode_s.c:991-998), legal and illegal callsigns.  AP formats carry addresses of DECODE_POOL."""
    import numpy as np

    rng = np.random.default_rng(seed)
    frames = rng.integers(0, 256, (n, 14), dtype=np.uint8)
    dfs = [17] * 10 + [18] * 6 + [11] * 2 + [0, 4, 5, 16, 20, 21, 24, 27, 31]
    for i in range(n):
        f = frames[i]
        df = dfs[int(rng.integers(len(dfs)))]
        f[0] = (df << 3) | (f[0] & 7) if df < 24 else (df << 3) | (f[0] & 7)
        nbytes = 14 if df & 0x10 else 7
        if df in (11, 17):  # these teach the reference's filter their address (mode_s.c:717-726), which holds about 2000: a small pool
            a = DECODE_POOL[int(rng.integers(len(DECODE_POOL)))] ^ int(rng.integers(64))
            f[1], f[2], f[3] = a >> 16, (a >> 8) & 255, a & 255
        if df in (17, 18):
            metype = int(rng.integers(32))
            f[4] = (metype << 3) | (f[4] & 7)
            kind = int(rng.integers(12))
            if metype in (1, 2, 3, 4) and kind < 6:  # a callsign from the legal alphabet
                chars = rng.choice(np.array([1, 2, 3, 26, 32, 48, 49, 57], dtype=np.int64), 8)
                bits = 0
                for c in chars:
                    bits = (bits << 6) | int(c)
                f[5:11] = np.frombuffer(bits.to_bytes(6, "big"), dtype=np.uint8)
            if metype == 15 and kind < 4:  # altitude 0, longitude 0, latitude's low 12 bits 0
                me = int.from_bytes(bytes(f[4:11]), "big")
                me &= ~(((1 << 12) - 1) << 36)   # bits 9-20
                me &= ~((1 << 17) - 1)          # bits 40-56
                me &= ~(((1 << 12) - 1) << 17)  # low 12 bits of bits 23-39
                f[4:11] = np.frombuffer(me.to_bytes(7, "big"), dtype=np.uint8)
            if metype == 19 and kind < 2:
                f[5] &= 0xFC  # east-west raw 0 in some velocity messages
                f[6] = 0
            if metype in (29, 31) and kind < 6:
                pass  # random bits already reach every version / subtype
        if df in (11, 17, 18):
            crc = crc24(f, nbytes)
            iid = 0
            if df == 11 and rng.integers(3) == 0:  # DF11 with an interrogator id: needs a known address
                a = DECODE_POOL[int(rng.integers(len(DECODE_POOL)))]
                f[1], f[2], f[3] = a >> 16, (a >> 8) & 255, a & 255
                crc = crc24(f, nbytes)
                iid = int(rng.integers(1, 128))
            crc ^= iid
            f[nbytes - 3], f[nbytes - 2], f[nbytes - 1] = crc >> 16, (crc >> 8) & 255, crc & 255
            if rng.integers(5) == 0 and not iid:  # one damaged bit outside the DF field and (so that the address stays known) the address
                bit = int(rng.integers(32, nbytes * 8))
                f[bit // 8] ^= 0x80 >> (bit % 8)
        else:  # Address/Parity: parity XOR address of a known aircraft
            a = DECODE_POOL[int(rng.integers(len(DECODE_POOL)))]
            crc = crc24(f, nbytes) ^ a
            f[nbytes - 3], f[nbytes - 2], f[nbytes - 1] = crc >> 16, (crc >> 8) & 255, crc & 255
    return frames


def records_from_structs(mm):
    """rsb_message records (the decode kernel's input) from the reference's struct modesMessage images: exactly the members the
    receive path itself produces (include/readsb_amd.h: rsb_message), the address and source as they are BEFORE the field decode."""
    import numpy as np

    import readsb_amd

    d = mm.view(readsb_amd.DECODED_DTYPE).reshape(-1)
    r = np.zeros(d.size, dtype=readsb_amd.MSG_DTYPE)
    r["ts"] = d["timestampMsg"]
    nbytes = d["msgbits"] // 8
    r["msg"] = d["msg"]
    for k in range(14):
        r["msg"][:, k][nbytes <= k] = 0
    r["msgbits"], r["msgtype"], r["crc"], r["correctedbits"], r["score"] = d["msgbits"], d["msgtype"], d["crc"], d["correctedbits"], d["score"]
    df = d["msgtype"]
    aa = (d["msg"][:, 1].astype(np.uint32) << 16) | (d["msg"][:, 2].astype(np.uint32) << 8) | d["msg"][:, 3]
    es = (df == 17) | (df == 18)
    modeac = df == 32
    r["addr"] = np.where(es | (df == 11), aa, np.where(modeac, d["addr"], d["crc"]))
    r["source"] = np.where(es, 7, np.where(df == 11, 4, np.where(modeac, 1, 3)))
    r["iid"] = d["IID"]
    r["signal"] = d["signalLevel"]
    return r


def mask_struct_images(mm):
    """What a comparison of struct modesMessage images leaves out: msg[] past the frame (stale stack bytes in the reference, SURVEY.md
    A.3) and sysTimestampMsg (wall clock)."""
    import numpy as np

    import readsb_amd

    out = mm.copy().view(readsb_amd.DECODED_DTYPE).reshape(-1)
    nbytes = out["msgbits"] // 8
    for k in range(14):
        out["msg"][:, k][nbytes <= k] = 0
    out["sysTimestampMsg"] = 0
    return out
