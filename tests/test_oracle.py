"""CPU tests: the oracle (oracle/modes_oracle.c) is pinned to the unmodified reference.

* against the golden fixtures that oracle/_ref/ref_harness (the reference compiled from its own
  sources) produced for the seeded scenarios -- message lists, demod counters, converter outputs,
  CRC syndrome tables;
* live against ref_harness on further seeds, wherever that binary exists (it is built only where
  /root/reference is present, and travels as a binary).
"""
import hashlib
import os

import numpy as np
import pytest

from oracle.oracle import Oracle, REC_DTYPE, STAT_NAMES, RefHarness, first_mismatch
from readsb_amd.synth import Scenario, crc24, df11, df17, generate, iq_sha256

from scenarios import CONVERT_CASES, SCENARIOS, convert_input

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_demod(name):
    z = np.load(os.path.join(GOLD, f"demod_{name}.npz"))
    return z["msgs"].reshape(-1).view(REC_DTYPE), dict(zip(STAT_NAMES, (int(x) for x in z["stats"]))), bytes(z["iq_sha256"]).hex()


@pytest.mark.parametrize("name", list(SCENARIOS))
def test_oracle_matches_golden_messages(name, built):
    sc, kw = SCENARIOS[name]
    iq = generate(sc)
    want, want_stats, sha = load_demod(name)
    assert iq_sha256(iq) == sha, "synthetic IQ generator is not reproducing the fixture's input"
    got, stats = Oracle(fmt=sc.fmt, nfix=kw["nfix"], modeac=kw["modeac"], dc_filter=kw.get("dc_filter", False)).run_stream(iq)
    assert first_mismatch(want, got) is None
    assert stats == want_stats


@pytest.mark.parametrize("name", list(CONVERT_CASES))
def test_oracle_converters_match_golden(name, built):
    fmt, dc, n, chunk = CONVERT_CASES[name]
    iq = convert_input(fmt, n)
    z = np.load(os.path.join(GOLD, f"convert_{name}.npz"))
    o = Oracle(fmt=fmt, dc_filter=dc)
    mags, means = [], []
    for pos in range(0, n, chunk):  # converter state (DC filter) carries across calls
        m, ml, mp = o.convert(iq[pos:pos + chunk])
        mags.append(m)
        means.append((ml, mp))
    mag = np.concatenate(mags)
    assert np.array_equal(mag[:4096], z["mag_head"])
    assert hashlib.sha256(mag.tobytes()).digest() == bytes(z["mag_sha256"])
    assert np.array_equal(np.array(means), z["means"])  # bit-exact doubles (float32 division)


def test_uc8_lut_known_answers(built):
    lut = Oracle().uc8_lut()
    # SURVEY.md Appendix B.5 [measured on the reference]: table in (I<<8|Q) order == (Q<<8|I) order by symmetry
    assert hashlib.sha256(lut.tobytes()).hexdigest() == "cfdb26c107e964c91edce2069e3f6aec56a274d0378feddbd7dec87e39d619df"
    for (i, q), v in {(127, 127): 363, (128, 128): 363, (130, 127): 1310, (200, 100): 39856, (255, 128): 65535}.items():
        assert lut[(i << 8) | q] == v and lut[(q << 8) | i] == v


def test_sc16_known_answers(built):
    # SURVEY.md Appendix B.6 [measured on the reference]
    for fmt, cases in {"SC16": {(1000, -2000): 4472, (32767, 0): 65533, (-32768, -32768): 65535, (123, 456): 945},
                       "SC16Q11": {(1000, -2000): 65535, (2047, 0): 65503, (123, 456): 15113}}.items():
        iq = np.array(list(cases.keys()), dtype="<i2")
        mag, _, _ = Oracle(fmt=fmt).convert(iq)
        assert list(mag) == list(cases.values())


@pytest.mark.parametrize("nfix", [1, 2])
def test_crc_tables_match_golden(nfix, built):
    z = np.load(os.path.join(GOLD, f"crctab_fix{nfix}.npz"))
    o = Oracle(nfix=nfix)
    for bits, key in ((56, "short"), (112, "long")):
        t = o.table(bits)
        rows = np.stack([t["syndrome"].astype(np.int64), t["errors"], t["bit"][:, 0], t["bit"][:, 1]], axis=1)
        assert np.array_equal(rows, z[key])
    assert {1: (51, 107), 2: (1326, 3831)}[nfix] == (z["short"].shape[0], z["long"].shape[0])  # crctests, SURVEY Appendix C
    m = bytes((i * 37 + 11) & 255 for i in range(14))
    assert [o.checksum(m[:7]), o.checksum(m)] == list(z["known"])


def test_crc_linearity_and_encoder(built):
    o = Oracle()
    rng = np.random.default_rng(5)
    for _ in range(50):
        a = rng.integers(0, 256, 14, dtype=np.uint8).tobytes()
        b = rng.integers(0, 256, 14, dtype=np.uint8).tobytes()
        x = bytes(p ^ q for p, q in zip(a, b))
        assert o.checksum(x) == o.checksum(a) ^ o.checksum(b)
        assert o.checksum(df17(0xABCDEF, a[:7])) == 0
        assert o.checksum(df11(0x123456, iid=9)) == 9
        assert o.checksum(a[:11] + b"\0\0\0") == crc24(a[:11])


def test_filter_semantics(built):
    o = Oracle()
    assert o.filter_test(0)  # an empty slot equals address 0 (icao_filter.c:101-107)
    assert not o.filter_test(0x4840D6)
    o.filter_add(0x4840D6)
    assert o.filter_test(0x4840D6)
    assert not o.filter_test(0x0040D6)  # the second key stores the full address (icao_filter.c:84-94)
    o.reset()
    assert not o.filter_test(0x4840D6)


def test_score_table(built):
    o = Oracle(nfix=1)
    good = df17(0x4840D6, bytes(range(7)))
    assert o.score(good, 112) == 1400          # mode_s.c:386-389
    assert o.score(good, 56) == -2             # validbits < msgbits
    bad1 = bytes([good[0], good[1], good[2], good[3], good[4] ^ 0x10]) + good[5:]
    assert o.score(bad1, 112) == 700
    o.filter_add(0x4840D6)
    assert o.score(good, 112) == 1800 and o.score(bad1, 112) == 900
    assert o.score(df11(0x4840D6), 56) == 1600 and o.score(df11(0x4840D7), 56) == 750
    assert o.score(df11(0x4840D6, iid=5), 56) == 1000 and o.score(df11(0x4840D7, iid=5), 56) == -1
    assert o.score(bytes(7), 56) == -2         # all zeros
    assert o.score(bytes([19 << 3]) + bytes(13), 112) == -2


@pytest.mark.skipif(not RefHarness.available(), reason="oracle/_ref/ref_harness not built here")
@pytest.mark.parametrize("seed,fmt,nfix,modeac,dc", [(21, "UC8", 1, True, False), (22, "UC8", 2, True, False),
                                                     (23, "SC16Q11", 1, True, True), (24, "UC8", 0, False, True)])
def test_oracle_matches_live_reference(seed, fmt, nfix, modeac, dc, built):
    iq = generate(Scenario(700_000, seed=seed, fmt=fmt, frames_per_sec=3500, err2_every=9, modeac_per_sec=250, dc_offset=0.02 if dc else 0.0))
    want, wst = RefHarness.demod(iq, fmt=fmt, nfix=nfix, modeac=modeac, dc_filter=dc)
    got, gst = Oracle(fmt=fmt, nfix=nfix, modeac=modeac, dc_filter=dc).run_stream(iq)
    assert first_mismatch(want, got) is None and wst == gst


def test_full_stats_fixture_agrees_with_message_fixtures():
    """tests/golden/stats_full.json (reference's struct stats incl. power sums) vs the counters stored with the message lists"""
    import json
    import math

    with open(os.path.join(GOLD, "stats_full.json")) as f:
        full = json.load(f)
    assert set(full) == set(SCENARIOS)
    for name in SCENARIOS:
        msgs, st, _ = load_demod(name)
        ref = full[name]
        assert [ref["demod_preambles"], ref["demod_rejected_bad"], ref["demod_rejected_unknown_icao"], *ref["demod_accepted"], ref["demod_modeac"]] == \
            [st[k] for k in STAT_NAMES]
        modes = msgs[msgs["msgtype"] != 32]
        assert ref["signal_power_count"] == int((modes["msgbits"].astype(np.int64) * 12 // 5).sum())  # demod_2400.c:345,356
        peak = float.fromhex(ref["peak_signal_power"])
        assert peak == (float(modes["signal"].max()) if modes.size else 0.0)  # :359-360
        assert math.isnan(float.fromhex(ref["noise_power_sum"])) == (SCENARIOS[name][0].n_samples % 131072 == 0)


def test_reference_cli_prints_the_golden_lists(built):
    """the golden lists come from oracle/_ref/ref_harness (the reference's objects behind a capture of
    useModesMessage); this pins that harness to the reference's own program: `readsb --device-type ifile --raw --mlat` prints
    the same (timestamp, frame) lines.  oracle/pin_cli.py; all six UC8 scenarios recorded in profiles/r2_cli_pin.txt."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "readsb")):
        pytest.skip("oracle/_ref/readsb not built (make -C oracle ref, build container only)")
    out = subprocess.run([sys.executable, os.path.join(root, "oracle", "pin_cli.py"), "cfg1_df11_df17", "nofix"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("IDENTICAL") == 2


def test_oracle_sc16q11_table_build_matches_the_reference_built_that_way(built):
    """convert_sc16q11_table (convert.c:264-328) exists only in a reference built with -DSC16Q11_TABLE_BITS=n.  The
    fixtures come from oracle/_ref/ref_harness_t8 (n = 8): converter output and means bit for bit, and the message lists of two
    scenarios (Mode A/C on: its noise thresholds use the integer means of that converter)."""
    from scenarios import TABLE_CONVERT_CASES, TABLE_SCENARIOS

    for name, (fmt, dc, n, chunk, bits) in TABLE_CONVERT_CASES.items():
        iq = convert_input(fmt, n)
        z = np.load(os.path.join(GOLD, f"convert_{name}.npz"))
        o = Oracle(fmt=fmt, dc_filter=dc, q11_table_bits=bits)
        parts = [o.convert(iq[pos:pos + chunk]) for pos in range(0, n, chunk)]
        mag = np.concatenate([p[0] for p in parts])
        assert hashlib.sha256(mag.tobytes()).digest() == bytes(z["mag_sha256"])
        assert np.array_equal(np.array([(p[1], p[2]) for p in parts]), z["means"])
        # and it is not the default build's converter
        assert not np.array_equal(mag, np.concatenate([Oracle(fmt=fmt).convert(iq[pos:pos + chunk])[0] for pos in range(0, n, chunk)]))
    for name, (sc, kw) in TABLE_SCENARIOS.items():
        iq = generate(sc)
        want, want_stats, sha = load_demod(name)
        assert iq_sha256(iq) == sha
        got, stats = Oracle(fmt=sc.fmt, nfix=kw["nfix"], modeac=kw["modeac"], q11_table_bits=kw["q11_table_bits"]).run_stream(iq)
        assert first_mismatch(want, got) is None
        assert stats == want_stats


def test_oracle_bladerf_buffer_chain_matches_golden(built):
    """mag_bufs of a bladeRF transfer sequence (129024-sample buffers and shorter ones after timestamp jumps / a wrong
    magic, converter per 252-sample sub-block, means averaged, overlap always from the buffer before: sdr_bladerf.c:322-404).  The
    fixture is the unmodified reference's converter + demodulators run on that chain by oracle/_ref/ref_harness demodbufs."""
    from scenarios import BLADERF_KW, BLADERF_PER, BLADERF_SCENARIO, bladerf_model, bladerf_transfers

    iq = generate(BLADERF_SCENARIO)
    z = np.load(os.path.join(GOLD, "demod_bladerf_chain.npz"))
    assert iq_sha256(iq) == bytes(z["iq_sha256"]).hex()
    bufs = bladerf_model(bladerf_transfers(iq))
    assert np.array_equal(np.array(bufs, dtype=np.uint64), z["bufs"])
    assert len(bufs) == 8 and sorted({b[1] for b in bufs}) == [75600, 103824, 128772, 129024]
    want = z["msgs"].reshape(-1).view(REC_DTYPE)
    got, stats = Oracle(fmt="SC16Q11", nfix=BLADERF_KW["nfix"], modeac=BLADERF_KW["modeac"]).run_bufs(iq, bufs, sub_len=BLADERF_PER)
    assert first_mismatch(want, got) is None
    assert stats == dict(zip(STAT_NAMES, (int(x) for x in z["stats"])))
    # the geometry matters: the same samples in 131072-sample blocks decode differently
    plain, _ = Oracle(fmt="SC16Q11", nfix=BLADERF_KW["nfix"], modeac=BLADERF_KW["modeac"]).run_stream(iq)
    assert first_mismatch(plain, got) is not None
