"""the rest of decodeModesMessage on the device (readsb_amd/csrc/decode_fields.hip) against the reference's own
struct modesMessage -- golden images made by tests/golden/make_golden.py `decode` from the unmodified reference (oracle/_ref/ref_harness
decodemsgs / demodfull), and, where that binary travelled to the box, the reference run live on a larger set.  Bar: every byte of the 416
equal, except msg[] past the frame (stale stack bytes in the reference) and sysTimestampMsg (wall clock); floats bit for bit."""
import hashlib
import os

import numpy as np
import pytest

import readsb_amd
from oracle import oracle as orc
from readsb_amd.synth import generate
from scenarios import DECODE_POOL, SCENARIOS, decode_frames, mask_struct_images, records_from_structs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def first_difference(want, got):
    """None, or the first record and member where two arrays of struct images differ"""
    a, b = mask_struct_images(want), mask_struct_images(got)
    if a.size != b.size:
        return f"{a.size} != {b.size} records"
    raw_a, raw_b = a.view(np.uint8).reshape(a.size, -1), b.view(np.uint8).reshape(b.size, -1)
    bad = np.nonzero((raw_a != raw_b).any(axis=1))[0]
    if not bad.size:
        return None
    i = int(bad[0])
    names = [n for n in a.dtype.names if a[n][i].tobytes() != b[n][i].tobytes()]
    return (f"{bad.size} of {a.size} records differ; record {i} (DF {a['msgtype'][i]}, ME type {a['metype'][i]}): " +
            "; ".join(f"{n}: want {a[n][i]!r} got {b[n][i]!r}" for n in names))


def test_random_frames_match_reference_structs(built):
    z = np.load(os.path.join(GOLD, "decode_fields.npz"))
    frames = decode_frames(z["result"].size)
    assert hashlib.sha256(frames.tobytes()).digest() == z["frames_sha256"].tobytes(), "frame generator drifted from the fixture"
    ok = z["result"] == 0
    want = z["structs"][ok]
    ch = readsb_amd.Channel(max_batch_samples=1 << 20)
    got = ch.decode_fields(records_from_structs(want))
    assert first_difference(want, got) is None, first_difference(want, got)
    d = want.view(readsb_amd.DECODED_DTYPE).reshape(-1)
    assert set(np.unique(d["metype"][d["msgtype"] == 17])) == set(range(32)) and set(np.unique(d["CF"][d["msgtype"] == 18])) == set(range(8))
    assert {0, 4, 5, 11, 16, 17, 18, 20, 21, 24, 27, 31} <= set(np.unique(d["msgtype"]))


def test_cpr_filtered_counter_on_device(built):
    import torch

    z = np.load(os.path.join(GOLD, "decode_fields.npz"))
    want = z["structs"][z["result"] == 0]
    recs = records_from_structs(want)
    ch = readsb_amd.Channel(max_batch_samples=1 << 20)
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()).to(dev)
    d_out = torch.zeros(recs.size * 416, dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ch._check(ch.l.rsb_decode_fields_device(ch.h, d_in.data_ptr(), recs.size, 0, 131072, d_out.data_ptr(), d_cnt.data_ptr(), None), "rsb_decode_fields_device")
    torch.cuda.synchronize()
    assert int(d_cnt.item()) == int(z["cpr_filtered"][0]) > 0
    got = d_out.cpu().numpy().view(readsb_amd.DECODED_DTYPE)
    assert first_difference(want, got) is None


@pytest.mark.parametrize("name", ["mixed_fix1", "dense_fix2_ac"])
def test_collect_decoded_matches_reference_structs(built, name):
    """the whole path: IQ -> messages -> struct modesMessage, against what the reference's demodulators hand to useModesMessage
    (Mode A/C replies included: their altitude carries over inside a block, demod_2400.c:482)"""
    sc, kw = SCENARIOS[name]
    want = np.load(os.path.join(GOLD, f"full_{name}.npz"))["structs"]
    iq = generate(sc)
    for host_replay in (False, True):
        ch = readsb_amd.Channel(fmt=sc.fmt, fix_bits=kw["nfix"], mode_ac=kw["modeac"], max_batch_samples=1 << 22, host_replay=host_replay)
        import torch

        d_iq = torch.from_numpy(np.ascontiguousarray(iq)).to(torch.device("cuda", 0))
        torch.cuda.synchronize()
        ch.submit(d_iq.data_ptr(), iq.shape[0])
        recs, got = ch.collect_decoded(1 << 15)
        assert recs.size == want.shape[0]
        assert first_difference(want, got) is None, first_difference(want, got)


def test_collect_decoded_in_batches(built):
    """batches of four blocks: the Mode A/C carry-over follows the blocks of each batch"""
    name = "dense_fix2_ac"
    sc, kw = SCENARIOS[name]
    want = np.load(os.path.join(GOLD, f"full_{name}.npz"))["structs"]
    iq = generate(sc)
    import torch

    ch = readsb_amd.Channel(fmt=sc.fmt, fix_bits=kw["nfix"], mode_ac=kw["modeac"], max_batch_samples=1 << 20)
    d_iq = torch.from_numpy(np.ascontiguousarray(iq)).to(torch.device("cuda", 0))
    torch.cuda.synchronize()
    step, parts, pos = 4 * 131072, [], 0
    while True:
        n = min(step, iq.shape[0] - pos)
        ch.submit(d_iq.data_ptr() + pos * 2, n)
        parts.append(ch.collect_decoded(1 << 14)[1].copy())
        pos += n
        if n < step:
            break
    got = np.concatenate(parts)
    assert first_difference(want, got) is None, first_difference(want, got)


def velocity_frames(ew, ns, mesub=1):
    """DF17 airborne-velocity frames (ME type 19) with the given signed velocity components; parity left zero (the decode kernel does
    not look at it)"""
    n = ew.size
    scale = 4 if mesub == 2 else 1
    ew_raw, ns_raw = np.abs(ew) // scale + 1, np.abs(ns) // scale + 1
    me = np.zeros(n, dtype=np.uint64)
    me |= np.uint64(19) << np.uint64(51)
    me |= np.uint64(mesub) << np.uint64(48)
    me |= (ew < 0).astype(np.uint64) << np.uint64(56 - 14)
    me |= ew_raw.astype(np.uint64) << np.uint64(56 - 24)
    me |= (ns < 0).astype(np.uint64) << np.uint64(56 - 25)
    me |= ns_raw.astype(np.uint64) << np.uint64(56 - 35)
    r = np.zeros(n, dtype=readsb_amd.MSG_DTYPE)
    r["msg"][:, 0] = (17 << 3) | 5
    r["msg"][:, 1:4] = (0x48, 0x40, 0xD6)
    for k in range(7):
        r["msg"][:, 4 + k] = ((me >> np.uint64(8 * (6 - k))) & np.uint64(255)).astype(np.uint8)
    r["msgbits"], r["msgtype"], r["addr"], r["source"] = 112, 17, 0x4840D6, 7
    return r


def test_ground_track_and_speed_exhaustive(built):
    """every pair of velocity components a subtype-1 message can carry (1023 x 1023 magnitudes x 4 sign pairs, negative zero folded),
    and a sample of subtype 2: gs and heading bit for bit against glibc's sqrtf / atan2 as the reference calls them"""
    ch = readsb_amd.Channel(max_batch_samples=1 << 20)
    mags = np.arange(0, 1023, dtype=np.int32)
    worst = 0
    for sew, sns in ((1, 1), (-1, 1), (1, -1), (-1, -1)):
        ew = np.repeat(mags * sew, mags.size)
        ns = np.tile(mags * sns, mags.size)
        if sew < 0:
            keep = ew != 0  # (a set sign bit on magnitude 0 decodes to the same 0)
            ew, ns = ew[keep], ns[keep]
        if sns < 0:
            keep = ns != 0
            ew, ns = ew[keep], ns[keep]
        want_gs, want_track = orc.es_velocity(ew, ns)
        got = ch.decode_fields(velocity_frames(ew, ns))
        assert (got["valid_bits"][:, 0] & 0x50 == 0x50).all()  # heading_valid, gs_valid
        assert np.array_equal(got["gs_selected"].view(np.uint32), want_gs.view(np.uint32))
        bad = got["heading"].view(np.uint32) != want_track.view(np.uint32)
        worst = max(worst, int(bad.sum()))
        assert not bad.any(), f"{int(bad.sum())} ground tracks differ, first: ew {ew[bad][0]} ns {ns[bad][0]} want {want_track[bad][0]!r} got {got['heading'][bad][0]!r}"
    rng = np.random.default_rng(5)
    ew = (rng.integers(-1022, 1023, 200000) * 4).astype(np.int32)
    ns = (rng.integers(-1022, 1023, 200000) * 4).astype(np.int32)
    want_gs, want_track = orc.es_velocity(ew, ns)
    got = ch.decode_fields(velocity_frames(ew, ns, mesub=2))
    assert np.array_equal(got["gs_v0"].view(np.uint32), want_gs.view(np.uint32)) and np.array_equal(got["heading"].view(np.uint32), want_track.view(np.uint32))


@pytest.mark.skipif(not orc.RefHarness.available(), reason="oracle/_ref/ref_harness did not travel")
def test_random_frames_against_live_reference(built):
    frames = decode_frames(150000, seed=99)
    res, want, filtered = orc.RefHarness.decodemsgs(frames, DECODE_POOL)
    want = want[res == 0]
    assert want.shape[0] > 140000 and filtered > 0
    ch = readsb_amd.Channel(max_batch_samples=1 << 20)
    got = ch.decode_fields(records_from_structs(want))
    assert first_difference(want, got) is None, first_difference(want, got)


def test_empty_and_bad_arguments(built):
    ch = readsb_amd.Channel(max_batch_samples=1 << 20)
    assert ch.decode_fields(np.zeros(0, dtype=readsb_amd.MSG_DTYPE)).size == 0
    assert ch.l.rsb_decode_fields(ch.h, None, 5, 0, None) == -1  # RSB_EINVAL
