"""include/readsb_dropin.h: the reference-signature entry points.  CRC calls on the CPU; converter and the two
per-block demodulators on the GPU, driven block by block the way sdr_ifile.c:184-228 drives the reference."""
import ctypes as C
import os

import numpy as np
import pytest

import readsb_amd
from readsb_amd import dropin
from oracle.oracle import Oracle, REC_DTYPE, first_mismatch
from readsb_amd.synth import generate

from scenarios import CONVERT_CASES, SCENARIOS, TABLE_CONVERT_CASES, TABLE_SCENARIOS, convert_input
from test_oracle import GOLD

B, T = readsb_amd.BLOCK, readsb_amd.TRAILING


def test_symbols_bound():
    l = dropin.dlib()
    for name in dropin.SYMBOLS:
        assert hasattr(l, name)


@pytest.mark.parametrize("nfix", [1, 2])
def test_crc_calls_match_reference_tables(nfix):
    l = dropin.dlib()
    z = np.load(os.path.join(GOLD, f"crctab_fix{nfix}.npz"))
    l.rsbd_modesChecksumInit(nfix)
    for bits, key in ((56, "short"), (112, "long")):
        rows = z[key]
        for syn, errors, b0, b1 in rows[:: max(1, len(rows) // 97)]:
            ei = l.rsbd_modesChecksumDiagnose(int(syn), bits)
            assert ei and (ei.contents.syndrome, ei.contents.errors) == (syn, errors)
            assert ei.contents.bit[0] == b0 and (errors < 2 or ei.contents.bit[1] == b1)
            # a frame with exactly those bits flipped has that syndrome, and Fix restores it (crc.c:410-418)
            msg = bytearray((i * 29 + 5) & 255 for i in range(bits // 8))
            buf = (C.c_uint8 * 14).from_buffer_copy(bytes(msg) + bytes(14 - len(msg)))
            crc = l.rsbd_modesChecksum(buf, bits)
            n = bits // 8
            buf[n - 3] ^= (crc >> 16) & 255
            buf[n - 2] ^= (crc >> 8) & 255
            buf[n - 1] ^= crc & 255
            assert l.rsbd_modesChecksum(buf, bits) == 0
            clean = bytes(buf)
            for b in (b0, b1)[:errors]:
                buf[b >> 3] ^= 1 << (7 - (b & 7))
            assert l.rsbd_modesChecksum(buf, bits) == syn
            l.rsbd_modesChecksumFix(buf, ei)
            assert bytes(buf) == clean
    none = l.rsbd_modesChecksumDiagnose(0, 112)
    assert none and none.contents.errors == 0  # NO_ERRORS, crc.c:389-390
    assert not l.rsbd_modesChecksumDiagnose(0x123457, 56) or nfix  # uncorrectable syndromes give NULL
    l.rsbd_crcCleanupTables()
    assert not l.rsbd_modesChecksumDiagnose(int(z["long"][0][0]), 112)


def test_init_converter_rejects_like_the_reference(capfd):
    l = dropin.dlib()
    st = C.c_void_p()
    assert not l.rsbd_init_converter(7, 2.4e6, 0, C.byref(st)) and not st.value
    assert "no suitable converter for format=7 dc=0" in capfd.readouterr().err  # convert.c:460-464


@pytest.mark.gpu
@pytest.mark.parametrize("case", [k for k, v in CONVERT_CASES.items() if not v[1]])
def test_converter_matches_oracle(case, built):
    fmt, dc, n, chunk = CONVERT_CASES[case]
    iq = convert_input(fmt, n)
    conv = dropin.Converter(readsb_amd.FORMATS[fmt])
    orc = Oracle(fmt=fmt)
    for lo in range(0, n, chunk):
        part = iq[lo:lo + chunk]
        mag, lvl, pwr = conv(part)
        wmag, wlvl, wpwr = orc.convert(part)
        assert np.array_equal(mag, wmag)
        assert (lvl, pwr) == (wlvl, wpwr)  # SC16: float32 running sums taken in sample order on the device
    conv.close()


@pytest.mark.gpu
def test_table_converter_build_option(built, tmp_path):
    """rsbd_set_sc16q11_table_bits(8) = -DSC16Q11_TABLE_BITS=8 (convert.c:264-328): init_converter(INPUT_SC16Q11, dc = 0) then hands
    out convert_sc16q11_table -- magnitudes and integer means bit for bit against the oracle and against the golden output of the
    reference's convert.c compiled with the macro; the other converters are untouched; the ifile driver follows the setting."""
    import hashlib

    l = dropin.dlib()
    assert l.rsbd_set_sc16q11_table_bits(12) == -1 and l.rsbd_set_sc16q11_table_bits(-1) == -1
    for case, (fmt, dc, n, chunk, bits) in TABLE_CONVERT_CASES.items():
        assert l.rsbd_set_sc16q11_table_bits(bits) == 0
        try:
            iq = convert_input(fmt, n)
            conv = dropin.Converter(readsb_amd.FORMATS[fmt])
            orc = Oracle(fmt=fmt, q11_table_bits=bits)
            mags, means = [], []
            for lo in range(0, n, chunk):
                mag, lvl, pwr = conv(iq[lo:lo + chunk])
                wmag, wlvl, wpwr = orc.convert(iq[lo:lo + chunk])
                assert np.array_equal(mag, wmag) and (lvl, pwr) == (wlvl, wpwr)
                mags.append(mag)
                means.append((lvl, pwr))
            conv.close()
            z = np.load(os.path.join(GOLD, f"convert_{case}.npz"))
            assert hashlib.sha256(np.concatenate(mags).tobytes()).digest() == bytes(z["mag_sha256"])
            assert np.array_equal(np.array(means), z["means"])
            # SC16 keeps its float converter under the same setting (convert.c:451-456 lists the table for SC16Q11 only)
            sc16 = convert_input("SC16", 4096)
            conv = dropin.Converter(readsb_amd.FORMATS["SC16"])
            assert np.array_equal(conv(sc16)[0], Oracle(fmt="SC16").convert(sc16)[0])
            conv.close()
            name = "sc16q11_table8_ac"
            sc, kw = TABLE_SCENARIOS[name]
            rec = generate(sc)
            path = tmp_path / "rec.iq"
            np.ascontiguousarray(rec).view(np.uint8).reshape(-1).tofile(path)
            want, _ = Oracle(fmt=sc.fmt, nfix=kw["nfix"], modeac=kw["modeac"], q11_table_bits=bits).run_stream(rec)
            nread, got = dropin.run_ifile(str(path), readsb_amd.FORMATS[sc.fmt], False, kw["nfix"], kw["modeac"], 0, 3)
            assert nread == sc.n_samples
            assert first_mismatch(want, got.view(REC_DTYPE)) is None
        finally:
            assert l.rsbd_set_sc16q11_table_bits(0) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", [k for k, v in CONVERT_CASES.items() if v[1]])
def test_dc_converter_within_tolerance(case, built):
    """init_converter(filter_dc=1) (convert.c:113-162, :164-213, :374-423).  The reference runs the DC block as a float32
    recurrence sample by sample; the device runs a chunked scan in double with the reference's float32 steps inside each
    16-sample run (tests/test_gpu_parity.py::test_dcfilter_stream explains why bit-exactness is out of reach).
    TOLERANCE: on these 100 000-sample inputs every magnitude within +-1 LSB of the reference's and at least 99.9 % of them
    identical; the means within 1e-4 relative (the reference's are float32 running sums, which carry about that much
    rounding noise themselves).  The DC state is carried from call to call like struct converter_state."""
    fmt, dc, n, chunk = CONVERT_CASES[case]
    iq = convert_input(fmt, n)
    conv = dropin.Converter(readsb_amd.FORMATS[fmt], dc_filter=True)
    orc = Oracle(fmt=fmt, dc_filter=True)
    same = total = 0
    for lo in range(0, n, chunk):
        part = iq[lo:lo + chunk]
        mag, lvl, pwr = conv(part)
        wmag, wlvl, wpwr = orc.convert(part)
        d = np.abs(mag.astype(np.int64) - wmag.astype(np.int64))
        assert d.max() <= 1, (lo, int(d.argmax()), int(d.max()))
        same += int((d == 0).sum())
        total += d.size
        assert abs(lvl - wlvl) <= 1e-4 * abs(wlvl) and abs(pwr - wpwr) <= 1e-4 * abs(wpwr)
    assert same >= 0.999 * total
    conv.close()


def feed_blocks(modes, orc, iq, modeac):
    """sdr_ifile.c:184-228: 131072-sample reads, 326 samples of overlap from the previous buffer, and the
    final short (possibly empty) block at EOF."""
    iq = iq.reshape(-1, 2)
    n = iq.shape[0]
    prev_tail = np.zeros(T, dtype=np.uint16)
    pos = 0
    while True:
        part = iq[pos:pos + B]
        length = part.shape[0]
        if length:
            mag, lvl, pwr = orc.convert(part)
        else:
            mag, lvl, pwr = np.zeros(0, dtype=np.uint16), 0.0, 0.0
        data = np.concatenate([prev_tail, mag, np.zeros(T, dtype=np.uint16)])  # the buffer is allocated with the overlap room
        modes.block(data, length, pos * 5, lvl, pwr, modeac)
        if length < B:
            break
        prev_tail = data[length:length + T].copy()  # lastbuf->data + lastbuf->length
        pos += B
    return modes.messages()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mixed_fix1", "dense_fix2_ac", "exact_2_blocks", "short_5000", "sc16_ac"])
def test_block_demodulators_match_oracle(name, built):
    sc, kw = SCENARIOS[name]
    iq = generate(sc)
    want, _ = Oracle(fmt=sc.fmt, nfix=kw["nfix"], modeac=kw["modeac"]).run_stream(iq)
    modes = dropin.Modes(nfix_crc=kw["nfix"])
    got = feed_blocks(modes, Oracle(fmt=sc.fmt), iq, kw["modeac"])
    assert first_mismatch(want, got.view(REC_DTYPE)) is None
    modes.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,blocks", [("dense_fix1_ac", 4), ("exact_2_blocks", 1), ("short_5000", 64), ("sc16", 3)])
def test_ifile_driver_matches_oracle(name, blocks, built, tmp_path):
    """sdr_ifile.c:115-254 + readsb.c:847-911 as one call: a recording on disk, read batch by batch (ragged end, a file
    that is an exact multiple of the batch, a file shorter than one block), == the oracle's block loop on the same bytes."""
    sc, kw = SCENARIOS[name]
    iq = generate(sc)
    path = tmp_path / "rec.iq"
    raw = np.ascontiguousarray(iq).view(np.uint8).reshape(-1)
    (raw if name != "short_5000" else np.concatenate([raw, raw[:1]])).tofile(path)  # a trailing partial sample is dropped (sdr_ifile.c:225)
    want, _ = Oracle(fmt=sc.fmt, nfix=kw["nfix"], modeac=kw["modeac"]).run_stream(iq)
    n, got = dropin.run_ifile(str(path), readsb_amd.FORMATS[sc.fmt], False, kw["nfix"], kw["modeac"], 0, blocks)
    assert n == sc.n_samples
    assert first_mismatch(want, got.view(REC_DTYPE)) is None


def test_ifile_open_errors(capfd):
    l = dropin.dlib()
    assert not l.rsbd_ifileOpen(b"/nonexistent/file.iq", 0, 0, 1, 0, 0, 1)
    assert "ifile: could not open /nonexistent/file.iq" in capfd.readouterr().err  # sdr_ifile.c:124-127
