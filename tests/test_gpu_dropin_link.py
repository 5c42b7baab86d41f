"""The link-level drop-in: oracle/_ref/ref_harness_gpu is the reference's own objects (mode_s, crc,
icao_filter, mode_ac, track, net_io ...) and the harness's unchanged block loop, linked with integration/gpu_path.c +
libreadsb_amd.so IN PLACE OF demod_2400.o and convert.o.  It calls init_converter / demodulate2400(struct mag_buf *) /
demodulate2400AC by the reference's own names and captures the real struct modesMessage in useModesMessage.  What it
captures must be what the unmodified reference produced for the same input (tests/golden/demodv_*.npz, made here by
tests/golden/make_golden.py with Modes.net_verbatim set): every compared field of every message, msg[] and verbatim[]
(the frame before the CRC repair, mode_s.c:427-430), and the accepted / Mode A/C counters of Modes.stats_current."""
import os

import numpy as np
import pytest

from oracle.oracle import REC_DTYPE, REF_GPU_BIN, STAT_NAMES, RefHarness, first_mismatch
from readsb_amd.synth import generate

from scenarios import SCENARIOS

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_dropin_binary_exports_the_reference_symbols():
    """no GPU needed: the shim defines the reference's exact entry points and the binary takes them from there"""
    import subprocess

    if not os.access(REF_GPU_BIN, os.X_OK):
        pytest.skip("oracle/_ref/ref_harness_gpu not built (needs /root/reference: make -C oracle ref)")
    syms = subprocess.run(["nm", "--defined-only", REF_GPU_BIN], capture_output=True, text=True, check=True).stdout
    for name in ("demodulate2400", "demodulate2400AC", "init_converter", "cleanup_converter"):
        assert f" T {name}\n" in syms, name
    undefined = subprocess.run(["nm", "--undefined-only", REF_GPU_BIN], capture_output=True, text=True, check=True).stdout
    assert "rsbd_demodulate2400" in undefined and "rsbd_init_converter" in undefined  # resolved by libreadsb_amd.so at load time


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mixed_fix1", "dense_fix2_ac"])
def test_reference_linked_against_the_library_reproduces_the_reference(name, built):
    assert os.access(REF_GPU_BIN, os.X_OK), "oracle/_ref/ref_harness_gpu is missing (built by __graft_entry__.build() where /root/reference exists)"
    sc, kw = SCENARIOS[name]
    z = np.load(os.path.join(GOLD, f"demodv_{name}.npz"))
    want = z["msgs"].reshape(-1).view(REC_DTYPE)
    want_stats = dict(zip(STAT_NAMES, (int(x) for x in z["stats"])))
    got, st, mv = RefHarness.demod(generate(sc), fmt=sc.fmt, nfix=kw["nfix"], modeac=kw["modeac"], verbatim=True, binary=REF_GPU_BIN)
    assert first_mismatch(want, got) is None
    assert np.array_equal(mv[:, 0], z["msg"])
    assert np.array_equal(mv[:, 1], z["verbatim"])
    assert (mv[:, 0] != mv[:, 1]).any(axis=1).sum() > 100  # repaired frames are in there
    for k in ("accepted0", "accepted1", "accepted2", "modeac"):
        assert st[k] == want_stats[k], k


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mixed_fix1", "dense_fix2_ac"])
def test_linked_reference_with_device_field_decode(name, built, monkeypatch):
    """RSB_GPU_FIELDS=1: the shim no longer calls decodeModesMessage; the struct modesMessage it hands to useModesMessage is the
    library's device-made image plus decodeCommB.  All 416 bytes of every message must be what the unmodified
    reference hands on (tests/golden/fullb_*.npz), msg[] past the frame and the wall-clock member aside."""
    from scenarios import mask_struct_images

    assert os.access(REF_GPU_BIN, os.X_OK)
    sc, kw = SCENARIOS[name]
    z = np.load(os.path.join(GOLD, f"fullb_{name}.npz"))
    want_stats = dict(zip(STAT_NAMES, (int(x) for x in z["stats"])))
    monkeypatch.setenv("RSB_GPU_FIELDS", "1")
    got, st, mm = RefHarness.demod(generate(sc), fmt=sc.fmt, nfix=kw["nfix"], modeac=kw["modeac"], full="commb", binary=REF_GPU_BIN)
    a, b = mask_struct_images(z["structs"]), mask_struct_images(mm)
    assert a.size == b.size
    for field in a.dtype.names:
        bad = np.nonzero((a[field] != b[field]).reshape(a.size, -1).any(axis=1))[0] if a[field].dtype.kind != "f" else \
            np.nonzero(a[field].view(f"u{a[field].dtype.itemsize}") != b[field].view(f"u{a[field].dtype.itemsize}"))[0]
        assert not bad.size, f"{field}: {bad.size} messages differ, first {bad[0]}: want {a[field][bad[0]]!r} got {b[field][bad[0]]!r} (DF {a['msgtype'][bad[0]]})"
    assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    for k in ("accepted0", "accepted1", "accepted2", "modeac"):
        assert st[k] == want_stats[k], k
    df = a["msgtype"]
    assert ((df == 20) | (df == 21)).sum() > 10 and (df == 17).sum() > 100
