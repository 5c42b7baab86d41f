"""Seeded synthetic 2.4 MS/s IQ with Mode-S / Mode A/C traffic.

Own code (the reference ships no signal generator and no IQ fixtures).  Everything that
reaches the output goes through IEEE +,-,*,/ and floor only (no libm transcendentals, no
ziggurat normals), so the same seed gives the same bytes on any host with the same numpy
bit-generator -- the golden fixtures under tests/golden/ store a SHA-256 of the IQ to
prove it.

Waveforms are built on a 60 MHz tick grid (25 ticks per 2.4 MS/s sample, the same virtual
clock demod_2400.c:468-475 uses for Mode A/C) and box-integrated per sample, so frames
start at arbitrary 1/25-sample offsets and exercise all five slicer phases.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field

import numpy as np

TPS = 25          # ticks per sample
TPU = 60          # ticks per microsecond
POLY = 0xFFF409   # Mode S CRC-24 generator

_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            c = i << 16
            for _ in range(8):
                c = ((c << 1) ^ POLY) if (c & 0x800000) else (c << 1)
            t.append(c & 0xFFFFFF)
        _CRC_TABLE = t
    return _CRC_TABLE


def crc24(data: bytes) -> int:
    """CRC-24 remainder of `data` (what a transponder puts in the 24-bit PI/AP field)."""
    t = _crc_table()
    rem = 0
    for b in data:
        rem = ((rem << 8) ^ t[b ^ (rem >> 16)]) & 0xFFFFFF
    return rem


def _with_parity(body: bytes, overlay: int = 0) -> bytes:
    p = crc24(body) ^ overlay
    return body + bytes([(p >> 16) & 255, (p >> 8) & 255, p & 255])


def df17(addr: int, me: bytes, ca: int = 5, df: int = 17) -> bytes:
    return _with_parity(bytes([(df << 3) | ca, (addr >> 16) & 255, (addr >> 8) & 255, addr & 255]) + bytes(me))


def df11(addr: int, ca: int = 5, iid: int = 0) -> bytes:
    return _with_parity(bytes([(11 << 3) | ca, (addr >> 16) & 255, (addr >> 8) & 255, addr & 255]), iid)


def ap_short(df: int, addr: int, payload27: int) -> bytes:
    """DF0/4/5: 5-bit DF + 27 payload bits + Address/Parity."""
    w = ((df & 31) << 27) | (payload27 & 0x7FFFFFF)
    return _with_parity(w.to_bytes(4, "big"), addr)


def ap_long(df: int, addr: int, payload27: int, mb: bytes) -> bytes:
    """DF16/20/21: 32 header bits + 56-bit MB/MV + Address/Parity."""
    w = ((df & 31) << 27) | (payload27 & 0x7FFFFFF)
    return _with_parity(w.to_bytes(4, "big") + bytes(mb), addr)


def flip_bits(msg: bytes, positions) -> bytes:
    b = bytearray(msg)
    for p in positions:
        b[p >> 3] ^= 1 << (7 - (p & 7))
    return bytes(b)


def modes_pulse_starts(msg: bytes) -> np.ndarray:
    """Tick offsets of the 0.5 us pulses of a Mode-S frame (preamble 0/1.0/3.5/4.5 us, PPM data from 8 us)."""
    bits = np.unpackbits(np.frombuffer(msg, dtype=np.uint8))
    data = 8 * TPU + np.arange(bits.size) * TPU + np.where(bits == 1, 0, TPU // 2)
    return np.concatenate([np.array([0, 60, 210, 270]), data])


def modeac_pulse_starts(code: int, spi: bool = False) -> np.ndarray:
    """Tick offsets of the 0.45 us pulses of a Mode A/C reply; `code` is the 4-digit octal squawk ABCD as
    0xABCD (hex-coded octal, same as mm->squawk).  Pulse order F1 C1 A1 C2 A2 C4 A4 X B1 D1 B2 D2 B4 D4 F2 . . SPI
    at 1.45 us pitch (demod_2400.c:488-511)."""
    a, b, c, d = (code >> 12) & 7, (code >> 8) & 7, (code >> 4) & 7, code & 7
    slots = [1, c & 1, a & 1, (c >> 1) & 1, (a >> 1) & 1, (c >> 2) & 1, (a >> 2) & 1, 0,
             b & 1, d & 1, (b >> 1) & 1, (d >> 1) & 1, (b >> 2) & 1, (d >> 2) & 1, 1, 0, 0, 1 if spi else 0]
    return np.array([87 * i for i, s in enumerate(slots) if s], dtype=np.int64)


def _envelope(starts: np.ndarray, width: int, t0: int):
    """Box-integrate unit pulses -> (first sample index, per-sample amplitude in [0,1])."""
    s0 = t0 // TPS
    lead = t0 - s0 * TPS
    span = int(starts.max()) + width + lead
    nsmp = (span + TPS - 1) // TPS
    d = np.zeros(nsmp * TPS + 1, dtype=np.int32)
    np.add.at(d, starts + lead, 1)
    np.add.at(d, starts + lead + width, -1)
    ticks = np.cumsum(d[:-1])
    return s0, ticks.reshape(nsmp, TPS).sum(axis=1).astype(np.float64) / TPS


@dataclass
class Scenario:
    n_samples: int
    seed: int = 1
    fmt: str = "UC8"               # UC8 | SC16 | SC16Q11
    frames_per_sec: float = 2353.0  # Mode-S frame rate (2.4e6 samples = 1 s)
    n_aircraft: int = 40
    amp_lo: float = 0.05           # burst amplitude range, full scale = 1.0
    amp_hi: float = 0.8
    noise_lsb: float = 2.0         # noise sigma in UC8 LSBs (of 127.5 full scale), per I and Q
    mix: dict = field(default_factory=lambda: {"df17": 0.55, "df11": 0.25, "df4": 0.06, "df5": 0.03,
                                                "df20": 0.04, "df21": 0.02, "df0": 0.03, "df16": 0.01, "df18": 0.01})
    err1_every: int = 7            # every k-th frame gets one flipped bit (0 = never)
    err2_every: int = 0            # every k-th frame gets two flipped bits
    df11_iid_every: int = 5        # every k-th DF11 carries a non-zero interrogator id
    collide_every: int = 25        # every k-th frame is followed by one that clobbers its tail
    modeac_per_sec: float = 0.0
    dc_offset: float = 0.0         # complex DC (same on I and Q), full-scale units


def generate(sc: Scenario, return_truth: bool = False):
    """-> IQ array, shape (n_samples, 2); uint8 for UC8, little-endian int16 otherwise."""
    rng = np.random.Generator(np.random.PCG64(sc.seed))
    n = sc.n_samples
    xi = np.full(n, sc.dc_offset, dtype=np.float64)
    xq = np.full(n, sc.dc_offset, dtype=np.float64)
    truth = []

    pool = rng.integers(0x100000, 0xFFFFFF, size=sc.n_aircraft, dtype=np.int64)
    kinds = list(sc.mix.keys())
    cum = np.cumsum([sc.mix[k] for k in kinds])
    cum = cum / cum[-1]
    total_ticks = n * TPS
    nframes = int(sc.frames_per_sec * n / 2.4e6)
    tail = 130 * TPU

    def add_burst(starts, width, t0):
        if t0 < 0 or t0 + int(starts.max()) + width + TPS >= total_ticks:
            return False
        s0, env = _envelope(starts, width, t0)
        amp = sc.amp_lo + (sc.amp_hi - sc.amp_lo) * rng.random()
        t = 4.0 * rng.random() - 2.0            # carrier phase from a rational point on the circle
        c, s = (1 - t * t) / (1 + t * t), 2 * t / (1 + t * t)
        if rng.random() < 0.5:
            c, s = -c, -s
        xi[s0:s0 + env.size] += amp * c * env
        xq[s0:s0 + env.size] += amp * s * env
        return True

    def make_frame(i):
        kind = kinds[int(np.searchsorted(cum, rng.random(), side="right").clip(0, len(kinds) - 1))]
        addr = int(pool[int(rng.integers(0, sc.n_aircraft))])
        if kind in ("df17", "df18"):
            msg = df17(addr, rng.integers(0, 256, size=7, dtype=np.uint8).tobytes(), ca=int(rng.integers(0, 8)),
                       df=17 if kind == "df17" else 18)
        elif kind == "df11":
            iid = 0
            if sc.df11_iid_every and i % sc.df11_iid_every == 0:
                iid = int(rng.integers(1, 128))
            msg = df11(addr, ca=int(rng.integers(0, 8)), iid=iid)
        elif kind in ("df0", "df4", "df5"):
            msg = ap_short(int(kind[2:]), addr, int(rng.integers(0, 1 << 27)))
        else:
            msg = ap_long(int(kind[2:]), addr, int(rng.integers(0, 1 << 27)),
                          rng.integers(0, 256, size=7, dtype=np.uint8).tobytes())
        nbits = len(msg) * 8
        if sc.err2_every and i % sc.err2_every == sc.err2_every - 1:
            msg = flip_bits(msg, sorted(int(x) for x in rng.choice(np.arange(5, nbits), size=2, replace=False)))
        elif sc.err1_every and i % sc.err1_every == sc.err1_every - 1:
            msg = flip_bits(msg, [int(rng.integers(5, nbits))])
        return kind, msg

    if nframes:
        t0s = np.sort(rng.integers(0, max(1, total_ticks - tail), size=nframes, dtype=np.int64))
        for i, t0 in enumerate(t0s):
            kind, msg = make_frame(i)
            if add_burst(modes_pulse_starts(msg), TPU // 2, int(t0)):
                truth.append((int(t0), kind, msg))
            if sc.collide_every and i % sc.collide_every == sc.collide_every - 1:
                # a second frame whose preamble lands on the last few bits of this one (demod_2400.c:365-371)
                kind2, msg2 = make_frame(i + 1000003)
                t1 = int(t0) + (8 + len(msg) * 8 - int(rng.integers(0, 7))) * TPU + int(rng.integers(0, TPS))
                if add_burst(modes_pulse_starts(msg2), TPU // 2, t1):
                    truth.append((t1, kind2, msg2))

    nac = int(sc.modeac_per_sec * n / 2.4e6)
    if nac:
        for t0 in np.sort(rng.integers(0, max(1, total_ticks - tail), size=nac, dtype=np.int64)):
            code = (int(rng.integers(0, 8)) << 12) | (int(rng.integers(0, 8)) << 8) | (int(rng.integers(0, 8)) << 4) | int(rng.integers(0, 8))
            spi = bool(rng.random() < 0.05)
            if add_burst(modeac_pulse_starts(code, spi), 27, int(t0)):
                truth.append((int(t0), "modeac", (code, spi)))

    # Irwin-Hall(4) noise from integer draws: exactly reproducible, bell-shaped, sigma = noise_lsb / 127.5
    if sc.noise_lsb > 0:
        scale = (sc.noise_lsb / 127.5) / (65536.0 * 0.5773502691896258)
        step = 1 << 22
        for a in range(0, n, step):
            b = min(n, a + step)
            u = rng.integers(0, 65536, size=(2, b - a, 4), dtype=np.int64).sum(axis=2) - 131070
            xi[a:b] += u[0] * scale
            xq[a:b] += u[1] * scale

    if sc.fmt == "UC8":
        iq = np.empty((n, 2), dtype=np.uint8)
        iq[:, 0] = np.clip(np.floor(127.5 + 127.5 * xi + 0.5), 0, 255).astype(np.uint8)
        iq[:, 1] = np.clip(np.floor(127.5 + 127.5 * xq + 0.5), 0, 255).astype(np.uint8)
    else:
        fs = 32767.0 if sc.fmt == "SC16" else 2047.0
        iq = np.empty((n, 2), dtype="<i2")
        iq[:, 0] = np.clip(np.floor(fs * xi + 0.5), -fs - 1, fs).astype(np.int16)
        iq[:, 1] = np.clip(np.floor(fs * xq + 0.5), -fs - 1, fs).astype(np.int16)
    return (iq, truth) if return_truth else iq


def iq_sha256(iq: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(iq).view(np.uint8).tobytes()).hexdigest()
