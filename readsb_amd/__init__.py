"""readsb_amd -- MI355X-native receive-side hot path of readsb behind a C ABI.

The product is ``libreadsb_amd.so`` (HIP kernels + host replay, see include/readsb_amd.h).
This package is only the ctypes plumbing the tests and bench.py use to call that C ABI; it
holds no algorithmic code and has no CPU fallback: if the shared library is missing, or no
HIP device is present, the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libreadsb_amd.so")

FORMATS = {"UC8": 0, "SC16": 1, "SC16Q11": 2}
BLOCK = 131072
TRAILING = 326

# rsb_message, packed (include/readsb_amd.h)
MSG_DTYPE = np.dtype([
    ("ts", "<u8"), ("msg", "u1", (14,)), ("msgbits", "u1"), ("msgtype", "u1"),
    ("crc", "<u4"), ("correctedbits", "<i4"), ("addr", "<u4"), ("score", "<i4"),
    ("source", "<i4"), ("iid", "<u4"), ("signal", "<f8"),
])
CAND_DTYPE = np.dtype([("g", "<u4"), ("vbytes", "u1", (5,)), ("pad", "u1"), ("msg", "u1", (5, 14)),
                       ("pow134", "<u8"), ("pow268", "<u8")])
AC_DTYPE = np.dtype([("g", "<u4"), ("f2_clock", "<u4"), ("modeac", "<u4")])
assert MSG_DTYPE.itemsize == 56 and CAND_DTYPE.itemsize == 96 and AC_DTYPE.itemsize == 12
# rsb_decoded = struct modesMessage (readsb.h:421-635) in its x86-64 layout
DECODED_DTYPE = np.dtype([
    ("timestampMsg", "<u8"), ("sysTimestampMsg", "<u8"), ("msg", "u1", (14,)), ("verbatim", "u1", (14,)), ("msgbits", "<i4"), ("msgtype", "<i4"),
    ("crc", "<u4"), ("correctedbits", "<i4"), ("addr", "<u4"), ("addrtype", "<i4"), ("remote", "<i4"), ("score", "<i4"), ("sbs_in", "<i4"),
    ("reduce_forward", "<i4"), ("source", "<i4"), ("signalLevel", "<f8"),
    ("IID", "<u4"), ("AA", "<u4"), ("AC", "<u4"), ("CA", "<u4"), ("CC", "<u4"), ("CF", "<u4"), ("DR", "<u4"), ("FS", "<u4"), ("ID", "<u4"), ("KE", "<u4"),
    ("ND", "<u4"), ("RI", "<u4"), ("SL", "<u4"), ("UM", "<u4"), ("VS", "<u4"), ("metype", "<u4"), ("mesub", "<u4"),
    ("MB", "u1", (7,)), ("MD", "u1", (10,)), ("ME", "u1", (7,)), ("MV", "u1", (7,)), ("valid_bits", "u1", (5,)),
    ("altitude_baro", "<i4"), ("altitude_baro_unit", "<i4"), ("altitude_geom", "<i4"), ("altitude_geom_unit", "<i4"), ("geom_delta", "<i4"),
    ("heading", "<f4"), ("heading_type", "<i4"), ("track_rate", "<f4"), ("roll", "<f4"), ("gs_v0", "<f4"), ("gs_v2", "<f4"), ("gs_selected", "<f4"),
    ("ias", "<u4"), ("tas", "<u4"), ("mach", "<f8"), ("baro_rate", "<i4"), ("geom_rate", "<i4"), ("squawk", "<u4"), ("callsign", "S16"),
    ("category", "<u4"), ("emergency", "<i4"), ("cpr_type", "<i4"), ("cpr_lat", "<u4"), ("cpr_lon", "<u4"), ("cpr_nucp", "<u4"), ("airground", "<i4"),
    ("decoded_lat", "<f8"), ("decoded_lon", "<f8"), ("decoded_nic", "<u4"), ("decoded_rc", "<u4"), ("commb_format", "<i4"),
    ("accuracy_bits", "<u4"), ("accuracy_sil_type", "<i4"), ("opstatus_sil_type", "<i4"), ("opstatus_tah", "<i4"), ("opstatus_hrd", "<i4"),
    ("opstatus_track_angle", "<i4"), ("opstatus_cc_lw", "<u4"), ("opstatus_cc_antenna_offset", "<u4"), ("opstatus_bits", "<u4"),
    ("nav_fms_altitude", "<u4"), ("nav_mcp_altitude", "<u4"), ("nav_qnh", "<f4"), ("nav_heading", "<f4"), ("nav_bits", "<u4"),
    ("nav_heading_type", "<i4"), ("nav_altitude_source", "<i4"), ("nav_modes", "<i4"),
])
assert DECODED_DTYPE.itemsize == 416 and DECODED_DTYPE.fields["valid_bits"][1] == 195 and DECODED_DTYPE.fields["nav_modes"][1] == 412


class Config(C.Structure):
    _fields_ = [("input_format", C.c_int32), ("dc_filter", C.c_int32), ("fix_bits", C.c_int32), ("mode_ac", C.c_int32),
                ("device", C.c_int32), ("queue_div", C.c_uint32), ("max_batch_samples", C.c_uint64),
                ("flags", C.c_uint32), ("scan_workgroups", C.c_uint32), ("block_samples", C.c_uint32), ("mean_sub_samples", C.c_uint32)]


FLAG_HOST_REPLAY = 1
FLAG_EXACT_STATS = 2
FLAG_DEFER_PRUNE = 4


class Stats(C.Structure):
    _fields_ = [("samples_processed", C.c_uint64), ("demod_preambles", C.c_uint64), ("demod_rejected_bad", C.c_uint64),
                ("demod_rejected_unknown_icao", C.c_uint64), ("demod_accepted", C.c_uint64 * 3), ("demod_modeac", C.c_uint64),
                ("gated_offsets", C.c_uint64), ("live_candidates", C.c_uint64), ("survivors", C.c_uint64),
                ("modeac_candidates", C.c_uint64),
                ("signal_power_sum", C.c_double), ("signal_power_count", C.c_uint64), ("noise_power_sum", C.c_double),
                ("noise_power_count", C.c_uint64), ("peak_signal_power", C.c_double), ("strong_signal_count", C.c_uint64),
                ("general_strips", C.c_uint64), ("patched_strips", C.c_uint64), ("split_strips", C.c_uint64), ("sort_fallbacks", C.c_uint64)]


class Timing(C.Structure):
    _fields_ = [("scan_ms", C.c_float), ("prune_ms", C.c_float), ("modeac_ms", C.c_float), ("total_ms", C.c_float),
                ("replay_ms", C.c_float), ("sort_ms", C.c_float), ("device_replay_ms", C.c_float), ("copy_ms", C.c_float),
                ("replay_iterations", C.c_int32), ("replay_on_device", C.c_int32)]


# every symbol include/readsb_amd.h declares: name -> (restype, argtypes)
_VP, _SZ = C.c_void_p, C.c_size_t
SYMBOLS = {
    "rsb_create": (C.c_int, [C.POINTER(Config), C.POINTER(_VP)]),
    "rsb_destroy": (None, [_VP]),
    "rsb_reset": (C.c_int, [_VP]),
    "rsb_filter_add": (C.c_int, [_VP, C.c_uint32]),
    "rsb_filter_add_many": (C.c_int, [_VP, _VP, _SZ]),
    "rsb_filter_expire": (C.c_int, [_VP]),
    "rsb_submit": (C.c_int, [_VP, _VP, C.c_uint64, _VP]),
    "rsb_seek": (C.c_int, [_VP, C.c_uint64, _VP, C.c_int]),
    "rsb_export_adders": (C.c_int, [_VP, _VP, _SZ, C.POINTER(_SZ)]),
    "rsb_superset_add": (C.c_int, [_VP, _VP, _SZ]),
    "rsb_batch_references": (C.c_int, [_VP, _VP, _SZ, C.POINTER(_SZ)]),
    "rsb_filter_export": (C.c_int, [_VP, _VP, _SZ, C.POINTER(_SZ)]),
    "rsb_collect": (C.c_int, [_VP, _VP, _SZ, C.POINTER(_SZ)]),
    "rsb_collect_begin": (C.c_int, [_VP, _VP, _SZ, C.POINTER(_SZ)]),
    "rsb_collect_wait": (C.c_int, [_VP]),
    "rsb_collect_prepare": (C.c_int, [_VP]),
    "rsb_collect_prepare_on": (C.c_int, [_VP, _VP]),
    "rsb_collect_fence": (C.c_int, [_VP, _VP]),
    "rsb_submit_many": (C.c_int, [_VP, _SZ, _VP, _VP, _VP]),
    "rsb_collect_many": (C.c_int, [_VP, _SZ, _VP, _VP, _VP]),
    "rsb_collect_many_begin": (C.c_int, [_VP, _SZ, _VP, _VP, _VP]),
    "rsb_collect_many_prepare_on": (C.c_int, [_VP, _SZ, _VP]),
    "rsb_process": (C.c_int, [_VP, _VP, C.c_uint64, _VP, _VP, _SZ, C.POINTER(_SZ)]),
    "rsb_process_host": (C.c_int, [_VP, _VP, C.c_uint64, _VP, _SZ, C.POINTER(_SZ)]),
    "rsb_get_stats": (C.c_int, [_VP, C.POINTER(Stats)]),
    "rsb_get_batch_magnitudes": (C.c_int, [_VP, _VP, C.c_uint64, _VP]),
    "rsb_get_timing": (C.c_int, [_VP, C.POINTER(Timing)]),
    "rsb_last_error": (C.c_char_p, [_VP]),
    "rsb_version": (C.c_char_p, []),
    "rsb_convert_device": (C.c_int, [_VP, _VP, _VP, C.c_uint64, _VP, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rsb_convert_noalloc": (C.c_int, [C.c_int, C.c_int, _VP, _VP, C.c_uint64, _VP, _VP, _VP]),
    "rsb_checksum": (C.c_uint32, [_VP, C.c_int]),
    "rsb_diagnose": (C.c_int, [_VP, C.c_uint32, C.c_int, C.POINTER(C.c_int * 2)]),
    "rsb_crc_table_size": (C.c_int, [C.c_int, C.c_int]),
    "rsb_crc_table_entry": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int), C.POINTER(C.c_int * 2)]),
    "rsb_uc8_lut": (_VP, []),
    "rsb_wire_encode": (C.c_int, [C.c_int, C.c_uint32, _VP, _SZ, _VP, _SZ, C.POINTER(_SZ)]),
    "rsb_collect_wire": (C.c_int, [_VP, C.c_int, C.c_uint32, _VP, _SZ, C.POINTER(_SZ), C.POINTER(_SZ)]),
    "rsb_forwarder_create": (_VP, []),
    "rsb_forwarder_destroy": (None, [_VP]),
    "rsb_forwarder_push": (C.c_int, [_VP, _VP, _SZ, _VP, _SZ, C.POINTER(_SZ)]),
    "rsb_stream_open": (C.c_int, [_VP, C.c_uint32, _VP, _VP, C.POINTER(_VP)]),
    "rsb_stream_push": (C.c_int, [_VP, _VP, C.c_uint32, C.c_uint64]),
    "rsb_stream_push_bladerf": (C.c_int, [_VP, _VP, _SZ, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "rsb_stream_flush": (C.c_int, [_VP]),
    "rsb_stream_close": (C.c_int, [_VP]),
    "rsb_decode_fields_device": (C.c_int, [_VP, _VP, _SZ, C.c_uint64, C.c_uint32, _VP, _VP, _VP]),
    "rsb_decode_fields": (C.c_int, [_VP, _VP, _SZ, C.c_uint64, _VP]),
    "rsb_collect_decoded": (C.c_int, [_VP, _VP, _VP, _SZ, C.POINTER(_SZ)]),
    "rsb_merge_by_timestamp": (C.c_int, [_VP, _VP, _SZ, _VP, _SZ, C.POINTER(_SZ)]),
    "rsb_host_replay": (C.c_int, [C.c_int, C.c_uint64, _VP, _SZ, _VP, _SZ, _VP, _SZ, C.POINTER(_SZ)]),
}

_lib = None


def lib():
    """The C ABI. Raises if libreadsb_amd.so has not been built (python __graft_entry__.py builds it)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C readsb_amd/csrc` "
                               "(there is no CPU fallback for the HIP path)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


class RsbError(RuntimeError):
    pass


STREAM_SINK = C.CFUNCTYPE(None, _VP, _VP, _SZ)


def submit_many(chans, dev_ptrs, nsamples, stream: int = 0):
    """rsb_submit_many: one batch per channel"""
    n = len(chans)
    ctx = (_VP * n)(*[c.h for c in chans])
    iq = (_VP * n)(*[int(p) for p in dev_ptrs])
    ns = (C.c_uint64 * n)(*[int(x) for x in nsamples])
    rc = chans[0].l.rsb_submit_many(ctx, n, iq, ns, stream)
    if rc != 0:
        raise RsbError(f"rsb_submit_many: {rc}")


def collect_many_prepare(chans, stream: int = 0):
    """rsb_collect_many_prepare_on: the channels' shared chain onto `stream` (the submit stream), without waiting"""
    n = len(chans)
    ctx = (_VP * n)(*[c.h for c in chans])
    rc = chans[0].l.rsb_collect_many_prepare_on(ctx, n, int(stream))
    if rc != 0:
        raise RsbError(f"rsb_collect_many_prepare_on: {rc}")


def collect_many(chans, outs, wait=True):
    """rsb_collect_many: outs[i] = a (pinned) MSG_DTYPE array per channel -> list of views holding each channel's messages.
    wait=False: rsb_collect_many_begin -- the record copies are still under way; Channel.collect_wait() on every channel before reading"""
    n = len(chans)
    ctx = (_VP * n)(*[c.h for c in chans])
    op = (_VP * n)(*[o.ctypes.data for o in outs])
    cap = (_SZ * n)(*[o.size for o in outs])
    cnt = (_SZ * n)()
    rc = (chans[0].l.rsb_collect_many if wait else chans[0].l.rsb_collect_many_begin)(ctx, n, op, cap, cnt)
    if rc != 0:
        for c in chans:  # (a copy left in flight must not outlive the buffers the exception lets go of)
            c.l.rsb_collect_wait(c.h)
        raise RsbError(f"rsb_collect_many: {rc}")
    return [o[:int(k)] for o, k in zip(outs, cnt)]


class Stream:
    """rsb_stream over a Channel: blocks of a live source in, messages out (collected in .messages())."""

    def __init__(self, ch: "Channel", blocks_per_batch: int):
        self.ch, self.l = ch, ch.l
        self.parts = []

        def sink(_user, msgs, n):
            self.parts.append(np.ctypeslib.as_array(C.cast(msgs, C.POINTER(C.c_uint8)), shape=(n * MSG_DTYPE.itemsize,)).view(MSG_DTYPE).copy())

        self._sink = STREAM_SINK(sink)  # kept alive as long as the stream
        h = _VP()
        ch._check(self.l.rsb_stream_open(ch.h, blocks_per_batch, C.cast(self._sink, _VP), None, C.byref(h)), "rsb_stream_open")
        self.h = h

    def push(self, block: np.ndarray, dropped_before: int = 0):
        raw = np.ascontiguousarray(block).view(np.uint8).reshape(-1)
        self.ch._check(self.l.rsb_stream_push(self.h, raw.ctypes.data if raw.size else None, raw.size // self.ch.bps, dropped_before), "rsb_stream_push")

    def push_bladerf(self, buf: np.ndarray, block_size: int, decimation: int = 1) -> int:
        """One bladeRF transfer buffer (16-byte metadata header per sub-block); returns the number of overrun flags seen."""
        raw = np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
        over = C.c_uint32(0)
        self.ch._check(self.l.rsb_stream_push_bladerf(self.h, raw.ctypes.data, raw.size, block_size, decimation, C.byref(over)), "rsb_stream_push_bladerf")
        return over.value

    def flush(self):
        self.ch._check(self.l.rsb_stream_flush(self.h), "rsb_stream_flush")

    def close(self):
        if self.h:
            h, self.h = self.h, None
            self.ch._check(self.l.rsb_stream_close(h), "rsb_stream_close")

    def messages(self) -> np.ndarray:
        return np.concatenate(self.parts) if self.parts else np.zeros(0, dtype=MSG_DTYPE)


class Channel:
    """One receiver channel (rsb_ctx)."""

    def __init__(self, fmt="UC8", fix_bits=1, mode_ac=False, dc_filter=False, device=0, max_batch_samples=1 << 24, queue_div=0,
                 host_replay=False, exact_stats=False, defer_prune=False, scan_workgroups=0, q11_table_bits=0, block_samples=0, mean_sub_samples=0):
        self.l = lib()
        self.fmt = fmt
        self.bps = 2 if fmt == "UC8" else 4
        cfg = Config(FORMATS[fmt], int(dc_filter), int(fix_bits), int(mode_ac), int(device), int(queue_div), int(max_batch_samples),
                     (FLAG_HOST_REPLAY if host_replay else 0) | (FLAG_EXACT_STATS if exact_stats else 0) | (FLAG_DEFER_PRUNE if defer_prune else 0)
                     | ((int(q11_table_bits) & 15) << 8),  # RSB_FLAG_SC16Q11_TABLE(bits): a reference built with -DSC16Q11_TABLE_BITS
                     int(scan_workgroups), int(block_samples), int(mean_sub_samples))
        h = _VP()
        rc = self.l.rsb_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise RsbError(f"rsb_create failed: {rc}")
        self.h = h
        self.max_batch = max_batch_samples

    def close(self):
        if getattr(self, "h", None):
            self.l.rsb_destroy(self.h)
            self.h = None

    __del__ = close

    def _check(self, rc, what):
        if rc != 0:
            raise RsbError(f"{what}: rc={rc}: {self.l.rsb_last_error(self.h).decode()}")

    def reset(self):
        self._check(self.l.rsb_reset(self.h), "rsb_reset")

    def filter_add(self, addr):
        self._check(self.l.rsb_filter_add(self.h, addr), "rsb_filter_add")

    def filter_add_many(self, addrs):
        addrs = np.ascontiguousarray(addrs, dtype=np.uint32)
        self._check(self.l.rsb_filter_add_many(self.h, addrs.ctypes.data if addrs.size else None, addrs.size), "rsb_filter_add_many")

    def filter_expire(self):
        self._check(self.l.rsb_filter_expire(self.h), "rsb_filter_expire")

    # ---- one recording in shards (include/readsb_amd.h, "one recording in shards")
    def seek(self, stream_pos: int, carry: np.ndarray | None = None):
        """continue at stream_pos with carry = the 326 samples before it (host array in the input format; None = zeros)"""
        if carry is not None:
            carry = np.ascontiguousarray(carry).view(np.uint8).reshape(-1)
            assert carry.size == TRAILING * self.bps
        self._check(self.l.rsb_seek(self.h, stream_pos, carry.ctypes.data if carry is not None else None, 0), "rsb_seek")

    def export_adders(self) -> np.ndarray:
        out = np.zeros(1 << 16, dtype=np.uint32)
        n = _SZ()
        rc = self.l.rsb_export_adders(self.h, out.ctypes.data, out.size, C.byref(n))
        if rc == -4:
            out = np.zeros(n.value, dtype=np.uint32)
            rc = self.l.rsb_export_adders(self.h, out.ctypes.data, out.size, C.byref(n))
        self._check(rc, "rsb_export_adders")
        return out[:n.value].copy()

    def superset_add(self, addrs: np.ndarray):
        addrs = np.ascontiguousarray(addrs, dtype=np.uint32)
        self._check(self.l.rsb_superset_add(self.h, addrs.ctypes.data if addrs.size else None, addrs.size), "rsb_superset_add")

    def batch_references(self, addrs) -> int:
        """rsb_batch_references: kept items of the last collected batch that involve one of addrs; -1 = not known (host replay)"""
        addrs = np.ascontiguousarray(addrs, dtype=np.uint32)
        n = _SZ()
        rc = self.l.rsb_batch_references(self.h, addrs.ctypes.data if addrs.size else None, addrs.size, C.byref(n))
        if rc == -5:  # RSB_ESTATE
            return -1
        self._check(rc, "rsb_batch_references")
        return int(n.value)

    def filter_export(self) -> np.ndarray:
        out = np.zeros(2 * 4096, dtype=np.uint32)
        n = _SZ()
        self._check(self.l.rsb_filter_export(self.h, out.ctypes.data, out.size, C.byref(n)), "rsb_filter_export")
        return out[:n.value].copy()

    def submit(self, dev_ptr: int, nsamples: int, stream: int = 0):
        self._check(self.l.rsb_submit(self.h, dev_ptr, nsamples, stream), "rsb_submit")

    def collect(self, cap: int = 1 << 16, out: np.ndarray | None = None) -> np.ndarray:
        if out is None:
            out = np.zeros(cap, dtype=MSG_DTYPE)
        cap = out.size
        n = _SZ()
        rc = self.l.rsb_collect(self.h, out.ctypes.data, cap, C.byref(n))
        if rc == -4 and n.value > cap:
            raise RsbError(f"rsb_collect: {n.value} messages exceed cap {cap}")
        self._check(rc, "rsb_collect")
        return out[:n.value]

    def collect_decoded(self, cap: int = 1 << 16):
        """rsb_collect_decoded -> (records, their decoded structs)"""
        out = np.zeros(cap, dtype=MSG_DTYPE)
        dec = np.zeros(cap, dtype=DECODED_DTYPE)
        n = _SZ()
        rc = self.l.rsb_collect_decoded(self.h, out.ctypes.data, dec.ctypes.data, cap, C.byref(n))
        if rc == -4 and n.value > cap:
            raise RsbError(f"rsb_collect_decoded: {n.value} messages exceed cap {cap}")
        self._check(rc, "rsb_collect_decoded")
        return out[:n.value], dec[:n.value]

    def decode_fields(self, msgs: np.ndarray, batch_pos: int = 0) -> np.ndarray:
        """rsb_decode_fields: records on the host -> struct modesMessage images on the host (through the device)"""
        msgs = np.ascontiguousarray(msgs, dtype=MSG_DTYPE)
        dec = np.zeros(msgs.size, dtype=DECODED_DTYPE)
        self._check(self.l.rsb_decode_fields(self.h, msgs.ctypes.data if msgs.size else None, msgs.size, batch_pos, dec.ctypes.data if msgs.size else None), "rsb_decode_fields")
        return dec

    def collect_begin(self, out: np.ndarray) -> np.ndarray:
        """rsb_collect_begin: the returned slice of `out` (pinned memory) holds the records only after collect_wait()."""
        n = _SZ()
        rc = self.l.rsb_collect_begin(self.h, out.ctypes.data, out.size, C.byref(n))
        if rc != 0:
            self.l.rsb_collect_wait(self.h)  # (on overflow the library leaves the copy of the first cap records in flight: not into a buffer we let go of)
        if rc == -4 and n.value > out.size:
            raise RsbError(f"rsb_collect_begin: {n.value} messages exceed cap {out.size}")
        self._check(rc, "rsb_collect_begin")
        return out[:n.value]

    def collect_prepare(self, on_stream=None):
        """rsb_collect_prepare; on_stream (a stream handle, 0 = the default stream): rsb_collect_prepare_on, the chain on that stream"""
        if on_stream is None:
            self._check(self.l.rsb_collect_prepare(self.h), "rsb_collect_prepare")
        else:
            self._check(self.l.rsb_collect_prepare_on(self.h, int(on_stream)), "rsb_collect_prepare_on")

    def collect_fence(self, stream: int = 0):
        self._check(self.l.rsb_collect_fence(self.h, stream), "rsb_collect_fence")

    def collect_wait(self):
        self._check(self.l.rsb_collect_wait(self.h), "rsb_collect_wait")

    def collect_wire(self, wire: str = "beast", mlat: bool = True, cap_bytes: int = 1 << 20, out: np.ndarray | None = None):
        """rsb_collect_wire: the submitted batch as Beast / raw bytes -> (bytes, number of messages in the batch)"""
        if out is None:
            out = np.zeros(cap_bytes, dtype=np.uint8)
        nb, nm = _SZ(), _SZ()
        rc = self.l.rsb_collect_wire(self.h, WIRES[wire], WIRE_MLAT if mlat else 0, out.ctypes.data, out.size, C.byref(nb), C.byref(nm))
        if rc == -4 and nb.value > out.size:
            raise RsbError(f"rsb_collect_wire: {nb.value} bytes exceed cap {out.size}")
        self._check(rc, "rsb_collect_wire")
        return out[:nb.value], nm.value

    def process_host(self, iq: np.ndarray, cap: int | None = None) -> np.ndarray:
        raw = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
        ns = raw.size // self.bps
        cap = cap or max(4096, ns // 64)
        out = np.zeros(cap, dtype=MSG_DTYPE)
        n = _SZ()
        self._check(self.l.rsb_process_host(self.h, raw.ctypes.data, ns, out.ctypes.data, cap, C.byref(n)), "rsb_process_host")
        return out[:n.value].copy()

    def process_stream_host(self, iq: np.ndarray, batch_blocks: int, block: int = BLOCK) -> np.ndarray:
        """Feed a recording in batches of batch_blocks x 131072 samples (last one partial); block: rsb_config.block_samples of the context."""
        raw = np.ascontiguousarray(iq).view(np.uint8).reshape(-1, self.bps)
        step = batch_blocks * block
        parts = []
        pos = 0
        while True:
            chunk = raw[pos:pos + step]
            parts.append(self.process_host(chunk))
            pos += step
            if chunk.shape[0] < step:  # the reference also runs a final (possibly empty) block at EOF
                break
        return np.concatenate(parts)

    def batch_magnitudes(self, nsamples: int, want_means: bool = False):
        """--dcfilter contexts: the converter's output for the last collected batch (and its per-block means)."""
        mag = np.zeros(nsamples, dtype=np.uint16)
        means = np.zeros(2 * ((nsamples + BLOCK - 1) // BLOCK), dtype=np.float64) if want_means else None
        self._check(self.l.rsb_get_batch_magnitudes(self.h, mag.ctypes.data, nsamples, means.ctypes.data if want_means else None), "rsb_get_batch_magnitudes")
        return mag, means

    def stats(self) -> dict:
        st = Stats()
        self.l.rsb_get_stats(self.h, C.byref(st))
        d = {k: getattr(st, k) for k, _ in Stats._fields_ if k != "demod_accepted"}
        d["demod_accepted"] = list(st.demod_accepted)
        return d

    def timing(self) -> dict:
        t = Timing()
        self.l.rsb_get_timing(self.h, C.byref(t))
        return {k: getattr(t, k) for k, _ in Timing._fields_}


WIRES = {"beast": 0, "raw": 1}
WIRE_MLAT = 1
WIRE_MAX_BYTES = 44


def wire_encode(msgs: np.ndarray, wire: str = "beast", mlat: bool = True) -> np.ndarray:
    """rsb_wire_encode (host): records -> Beast / raw bytes (net_io.c:760-826, 859-885)"""
    msgs = np.ascontiguousarray(msgs, dtype=MSG_DTYPE)
    out = np.zeros(WIRE_MAX_BYTES * msgs.size + 1, dtype=np.uint8)
    nb = _SZ()
    rc = lib().rsb_wire_encode(WIRES[wire], WIRE_MLAT if mlat else 0, msgs.ctypes.data, msgs.size, out.ctypes.data, out.size, C.byref(nb))
    if rc != 0:
        raise RsbError(f"rsb_wire_encode rc={rc}")
    return out[:nb.value].copy()


def merge_by_timestamp(lists) -> np.ndarray:
    """rsb_merge_by_timestamp: several channels' / shards' message lists -> one list in timestamp order (stable)"""
    lists = [np.ascontiguousarray(m, dtype=MSG_DTYPE) for m in lists]
    n = len(lists)
    ptrs = (_VP * max(1, n))(*[m.ctypes.data for m in lists])
    counts = (_SZ * max(1, n))(*[m.size for m in lists])
    total = sum(m.size for m in lists)
    out = np.zeros(max(1, total), dtype=MSG_DTYPE)
    got = _SZ()
    rc = lib().rsb_merge_by_timestamp(ptrs, counts, n, out.ctypes.data, out.size, C.byref(got))
    if rc != 0:
        raise RsbError(f"rsb_merge_by_timestamp rc={rc}")
    return out[:got.value]


class Forwarder:
    """rsb_forwarder: the two-message rule of useModesMessage (mode_s.c:2163-2175)"""

    def __init__(self):
        self.l = lib()
        self.h = self.l.rsb_forwarder_create()
        if not self.h:
            raise RsbError("rsb_forwarder_create failed")

    def close(self):
        if getattr(self, "h", None):
            self.l.rsb_forwarder_destroy(self.h)
            self.h = None

    __del__ = close

    def push(self, msgs: np.ndarray) -> np.ndarray:
        msgs = np.ascontiguousarray(msgs, dtype=MSG_DTYPE)
        out = np.zeros(2 * msgs.size + 1, dtype=MSG_DTYPE)
        n = _SZ()
        rc = self.l.rsb_forwarder_push(self.h, msgs.ctypes.data, msgs.size, out.ctypes.data, out.size, C.byref(n))
        if rc != 0:
            raise RsbError(f"rsb_forwarder_push rc={rc}")
        return out[:n.value].copy()


def host_replay(fix_bits: int, cands: np.ndarray, ac: np.ndarray | None = None, g0: int = 0) -> np.ndarray:
    l = lib()
    cands = np.ascontiguousarray(cands, dtype=CAND_DTYPE)
    ac = np.zeros(0, dtype=AC_DTYPE) if ac is None else np.ascontiguousarray(ac, dtype=AC_DTYPE)
    cap = cands.size + ac.size + 1
    out = np.zeros(cap, dtype=MSG_DTYPE)
    n = _SZ()
    rc = l.rsb_host_replay(fix_bits, g0, cands.ctypes.data, cands.size, ac.ctypes.data, ac.size, out.ctypes.data, cap, C.byref(n))
    if rc != 0:
        raise RsbError(f"rsb_host_replay rc={rc}")
    return out[:n.value].copy()


def crc_table(fix_bits: int, bits: int) -> np.ndarray:
    """rows of (syndrome, errors, bit0, bit1)"""
    l = lib()
    n = l.rsb_crc_table_size(fix_bits, bits)
    rows = np.zeros((n, 4), dtype=np.int64)
    syn, err, pos = C.c_uint32(), C.c_int(), (C.c_int * 2)()
    for i in range(n):
        l.rsb_crc_table_entry(fix_bits, bits, i, C.byref(syn), C.byref(err), C.byref(pos))
        rows[i] = (syn.value, err.value, pos[0], pos[1])
    return rows


def uc8_lut() -> np.ndarray:
    p = lib().rsb_uc8_lut()
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint16)), shape=(65536,)).copy()
