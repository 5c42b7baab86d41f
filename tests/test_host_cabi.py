"""CPU tests of the product library: it loads, exports every symbol include/*.h declares, and its
host-side logic (CRC tables, UC8 table, ICAO filter + scoring + decode + greedy replay) reproduces
the reference's fixtures.  No compute kernel is launched here."""
import ctypes as C
import hashlib
import os
import re

import numpy as np
import pytest

import readsb_amd
from oracle import oracle as orc
from oracle.oracle import Oracle, REC_DTYPE, first_mismatch
from readsb_amd.synth import generate

from scenarios import SCENARIOS
from test_oracle import GOLD, load_demod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for fn in os.listdir(inc):
        text = open(os.path.join(inc, fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"^[A-Za-z_][\w \*]*?\b(\w+)\s*\(", text, flags=re.M):
            if not m.group(0).lstrip().startswith(("typedef", "#", "return")):
                names.add(m.group(1))
    return names


def test_library_loads_and_exports_every_declared_symbol():
    lib = C.CDLL(readsb_amd.LIB_PATH)
    names = declared_symbols()
    assert {"rsb_create", "rsb_submit", "rsb_collect", "rsb_host_replay"} <= names
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, f"declared in include/ but not exported: {missing}"
    assert set(readsb_amd.SYMBOLS) <= names
    assert b"gfx950" in readsb_amd.lib().rsb_version()


@pytest.mark.parametrize("nfix", [1, 2])
def test_crc_tables_match_reference_fixture(nfix):
    z = np.load(os.path.join(GOLD, f"crctab_fix{nfix}.npz"))
    assert np.array_equal(readsb_amd.crc_table(nfix, 56), z["short"])
    assert np.array_equal(readsb_amd.crc_table(nfix, 112), z["long"])
    m = bytes((i * 37 + 11) & 255 for i in range(14))
    l = readsb_amd.lib()
    buf = (C.c_uint8 * 14).from_buffer_copy(m)
    assert [l.rsb_checksum(buf, 56), l.rsb_checksum(buf, 112)] == list(z["known"])
    assert readsb_amd.lib().rsb_crc_table_size(0, 112) == 0  # --no-fix: no table (crc.c:357-360)


def test_uc8_table_matches_reference_hash():
    assert hashlib.sha256(readsb_amd.uc8_lut().tobytes()).hexdigest() == \
        "cfdb26c107e964c91edce2069e3f6aec56a274d0378feddbd7dec87e39d619df"


def oracle_candidates(iq, fmt, nfix, modeac):
    lib = orc._lib()
    lib.orc_gated_candidates.restype = C.c_size_t
    lib.orc_gated_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t]
    lib.orc_ac_candidates.restype = C.c_size_t
    lib.orc_ac_candidates.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t]
    o = Oracle(fmt=fmt, nfix=nfix, modeac=modeac)
    raw = orc.iq_bytes(iq)
    n = raw.size // o.bps
    cnt = lib.orc_gated_candidates(o.h, raw.ctypes.data, n, None, 0)
    cands = np.zeros(cnt, dtype=readsb_amd.CAND_DTYPE)
    o.reset()
    lib.orc_gated_candidates(o.h, raw.ctypes.data, n, cands.ctypes.data, cnt)
    ac = np.zeros(0, dtype=readsb_amd.AC_DTYPE)
    if modeac:
        o.reset()
        nac = lib.orc_ac_candidates(o.h, raw.ctypes.data, n, None, 0)
        ac = np.zeros(nac, dtype=readsb_amd.AC_DTYPE)
        o.reset()
        lib.orc_ac_candidates(o.h, raw.ctypes.data, n, ac.ctypes.data, nac)
    return cands, ac


@pytest.mark.parametrize("name", ["cfg1_df11_df17", "mixed_fix1", "dense_fix2_ac", "dense_fix1_ac", "nofix", "exact_2_blocks", "sc16"])
def test_host_replay_reproduces_reference_messages(name, built):
    """Every gated offset (what the scan kernel could at most hand over), shuffled, through the
    product's host replay == the reference's message list."""
    sc, kw = SCENARIOS[name]
    iq = generate(sc)
    want, _, _ = load_demod(name)
    cands, ac = oracle_candidates(iq, sc.fmt, kw["nfix"], kw["modeac"])
    rng = np.random.default_rng(1)
    got = readsb_amd.host_replay(kw["nfix"], cands[rng.permutation(cands.size)], ac[rng.permutation(ac.size)])
    assert first_mismatch(want, got.view(REC_DTYPE)) is None


def test_host_replay_rejects_unaligned_origin():
    l = readsb_amd.lib()
    n = C.c_size_t()
    assert l.rsb_host_replay(1, 5, None, 0, None, 0, None, 0, C.byref(n)) == -1


def test_create_validates_arguments():
    l = readsb_amd.lib()
    h = C.c_void_p()
    bad_fmt = readsb_amd.Config(7, 0, 1, 0, 0, 0, 1 << 20)
    assert l.rsb_create(C.byref(bad_fmt), C.byref(h)) == -1 and not h.value  # "no suitable converter", convert.c:460-464
    neg_fix = readsb_amd.Config(0, 0, -1, 0, 0, 0, 1 << 20)
    assert l.rsb_create(C.byref(neg_fix), C.byref(h)) == -1 and not h.value
    assert l.rsb_create(None, C.byref(h)) == -1


def test_no_silent_cpu_fallback():
    """Without a HIP device the compute entry points must fail loudly, never compute on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    l = readsb_amd.lib()
    h = C.c_void_p()
    cfg = readsb_amd.Config(0, 0, 1, 0, 0, 0, 1 << 20)
    assert l.rsb_create(C.byref(cfg), C.byref(h)) == -2 and not h.value  # RSB_ENODEV
    with pytest.raises(readsb_amd.RsbError):
        readsb_amd.Channel()


def test_stream_entry_points_reject_bad_arguments_without_a_device():
    l = readsb_amd.lib()
    h = C.c_void_p()
    assert l.rsb_stream_open(None, 4, None, None, C.byref(h)) == -1 and not h.value
    assert l.rsb_stream_push(None, None, 0, 0) == -1
    assert l.rsb_stream_flush(None) == -1
    assert l.rsb_stream_close(None) == -1


def test_merge_by_timestamp_is_a_stable_merge():
    """rsb_merge_by_timestamp: lists that are sorted runs (Mode A/C messages behind each block's Mode S messages), equal timestamps
    across lists, an empty list; too small an output is refused."""
    import ctypes as C

    rng = np.random.Generator(np.random.PCG64(5))
    lists = []
    for k, n in enumerate((700, 0, 1300, 5)):
        m = np.zeros(n, dtype=readsb_amd.MSG_DTYPE)
        m["ts"] = np.sort(rng.integers(0, 4000, size=n))  # many ties
        if n > 100:
            m["ts"][50:80] = m["ts"][50:80][::-1]          # a run out of order, as a block's Mode A/C tail is
        m["addr"] = k
        m["crc"] = np.arange(n)
        lists.append(m)
    got = readsb_amd.merge_by_timestamp(lists)
    both = np.concatenate(lists)
    assert got.tobytes() == both[np.argsort(both["ts"], kind="stable")].tobytes()
    assert readsb_amd.merge_by_timestamp([]).size == 0
    out = np.zeros(3, dtype=readsb_amd.MSG_DTYPE)
    n = C.c_size_t()
    ptrs, counts = (C.c_void_p * 1)(lists[0].ctypes.data), (C.c_size_t * 1)(lists[0].size)
    assert readsb_amd.lib().rsb_merge_by_timestamp(ptrs, counts, 1, out.ctypes.data, 3, C.byref(n)) == -4 and n.value == 700
