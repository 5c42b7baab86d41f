"""ctypes plumbing for include/readsb_dropin.h (the reference-signature entry points); no algorithmic code."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import BLOCK, MSG_DTYPE, TRAILING, lib

_VP = C.c_void_p
CONVERT_FN = C.CFUNCTYPE(None, _VP, _VP, C.c_uint, _VP, C.POINTER(C.c_double), C.POINTER(C.c_double))
USE_FN = C.CFUNCTYPE(None, _VP, _VP)


class ErrorInfo(C.Structure):
    _fields_ = [("syndrome", C.c_uint32), ("errors", C.c_int), ("bit", C.c_int8 * 2)]


class MagBuf(C.Structure):  # struct mag_buf, readsb.h:298-311
    _fields_ = [("sampleTimestamp", C.c_uint64), ("mean_level", C.c_double), ("mean_power", C.c_double), ("dropped", C.c_uint32),
                ("length", C.c_uint), ("sysTimestamp", C.c_uint64), ("data", _VP)]


SYMBOLS = {
    "rsbd_init_converter": (_VP, [C.c_int, C.c_double, C.c_int, C.POINTER(_VP)]),
    "rsbd_cleanup_converter": (None, [_VP]),
    "rsbd_set_sc16q11_table_bits": (C.c_int, [C.c_int]),
    "rsbd_modesChecksumInit": (None, [C.c_int]),
    "rsbd_modesChecksum": (C.c_uint32, [_VP, C.c_int]),
    "rsbd_modesChecksumDiagnose": (C.POINTER(ErrorInfo), [C.c_uint32, C.c_int]),
    "rsbd_modesChecksumFix": (None, [_VP, C.POINTER(ErrorInfo)]),
    "rsbd_crcCleanupTables": (None, []),
    "rsbd_modesInit": (_VP, [C.c_int, C.c_int, C.c_uint]),
    "rsbd_modesCleanup": (None, [_VP]),
    "rsbd_icaoFilterInit": (None, [_VP]),
    "rsbd_icaoFilterAdd": (None, [_VP, C.c_uint32]),
    "rsbd_icaoFilterExpire": (None, [_VP]),
    "rsbd_demodulate2400": (None, [_VP, C.POINTER(MagBuf), USE_FN, _VP]),
    "rsbd_demodulate2400AC": (None, [_VP, C.POINTER(MagBuf), USE_FN, _VP]),
    "rsbd_demodulate2400_decoded": (None, [_VP, C.POINTER(MagBuf), _VP, _VP]),
    "rsbd_demodulate2400AC_decoded": (None, [_VP, C.POINTER(MagBuf), _VP, _VP]),
    "rsbd_ifileOpen": (_VP, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint]),
    "rsbd_ifileRun": (C.c_longlong, [_VP, USE_FN, _VP]),
    "rsbd_ifileFilterAdd": (None, [_VP, C.c_uint32]),
    "rsbd_ifileClose": (None, [_VP]),
}

_bound = False


def dlib():
    global _bound
    l = lib()
    if not _bound:
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _bound = True
    return l


class Converter:
    """init_converter / iq_convert_fn / cleanup_converter"""

    def __init__(self, fmt: int, dc_filter: bool = False, sample_rate: float = 2.4e6):
        self.l = dlib()
        self.state = _VP()
        fn = self.l.rsbd_init_converter(fmt, sample_rate, int(dc_filter), C.byref(self.state))
        if not fn:
            raise ValueError("no suitable converter")
        self.fn = CONVERT_FN(fn)
        self.bps = 2 if fmt == 0 else 4

    def __call__(self, iq: np.ndarray):
        raw = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
        n = raw.size // self.bps
        mag = np.zeros(n, dtype=np.uint16)
        lvl, pwr = C.c_double(), C.c_double()
        self.fn(raw.ctypes.data, mag.ctypes.data, n, self.state, C.byref(lvl), C.byref(pwr))
        return mag, lvl.value, pwr.value

    def close(self):
        if self.state:
            self.l.rsbd_cleanup_converter(self.state)
            self.state = _VP()

    __del__ = close


class Modes:
    """rsbd_modesInit + the two demodulators; messages are collected through the useModesMessage callback."""

    def __init__(self, nfix_crc=1, device=0, max_length=BLOCK):
        self.l = dlib()
        self.h = self.l.rsbd_modesInit(nfix_crc, device, max_length)
        if not self.h:
            raise RuntimeError("rsbd_modesInit failed (no HIP device?)")
        self.msgs = []
        self._cb = USE_FN(self._use)

    def _use(self, user, mm):
        self.msgs.append(np.ctypeslib.as_array(C.cast(mm, C.POINTER(C.c_uint8)), shape=(MSG_DTYPE.itemsize,)).copy())

    def filter_add(self, addr):
        self.l.rsbd_icaoFilterAdd(self.h, addr)

    def block(self, data: np.ndarray, length: int, sample_timestamp: int, mean_level: float, mean_power: float, modeac: bool):
        """data = 326 overlap samples + length samples (uint16). Runs demodulate2400 then, if asked, demodulate2400AC (readsb.c:883-886)."""
        data = np.ascontiguousarray(data, dtype=np.uint16)
        assert data.size >= length + TRAILING
        mb = MagBuf(sample_timestamp, mean_level, mean_power, 0, length, 0, data.ctypes.data)
        self.l.rsbd_demodulate2400(self.h, C.byref(mb), self._cb, None)
        if modeac:
            self.l.rsbd_demodulate2400AC(self.h, C.byref(mb), self._cb, None)

    def messages(self) -> np.ndarray:
        if not self.msgs:
            return np.zeros(0, dtype=MSG_DTYPE)
        return np.concatenate(self.msgs).view(MSG_DTYPE)

    def close(self):
        if getattr(self, "h", None):
            self.l.rsbd_modesCleanup(self.h)
            self.h = None

    __del__ = close


def run_ifile(path: str, fmt: int = 0, dc_filter: bool = False, nfix_crc: int = 1, mode_ac: bool = False, device: int = 0, blocks_per_batch: int = 64):
    """--device-type ifile --ifile path: -> (samples consumed, messages)."""
    l = dlib()
    h = l.rsbd_ifileOpen(path.encode(), fmt, int(dc_filter), nfix_crc, int(mode_ac), device, blocks_per_batch)
    if not h:
        raise RuntimeError("rsbd_ifileOpen failed")
    msgs = []
    cb = USE_FN(lambda user, mm: msgs.append(np.ctypeslib.as_array(C.cast(mm, C.POINTER(C.c_uint8)), shape=(MSG_DTYPE.itemsize,)).copy()))
    try:
        n = l.rsbd_ifileRun(h, cb, None)
    finally:
        l.rsbd_ifileClose(h)
    out = np.concatenate(msgs).view(MSG_DTYPE) if msgs else np.zeros(0, dtype=MSG_DTYPE)
    return n, out
