"""CPU side of SURVEY.md 8 (f)4: the field-decode fixtures are what the unmodified reference produces (where its harness is built),
the frame generator is reproducible, and the record <-> struct helpers the GPU test relies on are consistent."""
import hashlib
import os

import numpy as np
import pytest

import readsb_amd
from oracle import oracle as orc
from scenarios import DECODE_POOL, decode_frames, mask_struct_images, records_from_structs

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_frame_generator_matches_fixture():
    z = np.load(os.path.join(GOLD, "decode_fields.npz"))
    frames = decode_frames(z["result"].size)
    assert hashlib.sha256(frames.tobytes()).digest() == z["frames_sha256"].tobytes()
    assert z["structs"].shape == (frames.shape[0], readsb_amd.DECODED_DTYPE.itemsize)


@pytest.mark.skipif(not orc.RefHarness.available(), reason="oracle/_ref/ref_harness not built (needs /root/reference)")
def test_fixture_is_the_reference_output():
    z = np.load(os.path.join(GOLD, "decode_fields.npz"))
    res, mm, filtered = orc.RefHarness.decodemsgs(decode_frames(z["result"].size), DECODE_POOL)
    assert np.array_equal(res, z["result"]) and filtered == int(z["cpr_filtered"][0])
    ok = res == 0
    assert np.array_equal(mask_struct_images(mm[ok]), mask_struct_images(z["structs"][ok]))


def test_records_from_structs_round_trip():
    z = np.load(os.path.join(GOLD, "decode_fields.npz"))
    want = z["structs"][z["result"] == 0]
    r = records_from_structs(want)
    d = want.view(readsb_amd.DECODED_DTYPE).reshape(-1)
    assert np.array_equal(r["msgtype"], d["msgtype"]) and np.array_equal(r["crc"], d["crc"])
    # the address before the field decode: without the non-ICAO flag the extended squitter decode may add (mode_s.c:771-792, 1385-1425)
    assert np.array_equal(r["addr"] & 0xFFFFFF, d["addr"] & 0xFFFFFF)
    l = readsb_amd.lib()
    for i in range(0, r.size, 97):  # the repaired frames carry a valid parity / a known address
        m = np.ascontiguousarray(r["msg"][i])
        syn = l.rsb_checksum(m.ctypes.data, int(r["msgbits"][i]))
        df = int(r["msgtype"][i])
        if df in (17, 18):
            assert syn == 0
        elif df == 11:
            assert syn & 0xFFFF80 == 0
        else:
            assert syn in DECODE_POOL


def test_es_velocity_known_values():
    gs, track = orc.es_velocity(np.array([0, 100, -100, 0, 3], dtype=np.int32), np.array([100, 0, 0, -100, 4], dtype=np.int32))
    assert track.tolist()[:4] == [0.0, 90.0, 270.0, 180.0]
    assert abs(gs[4] - np.sqrt(25.5)) < 1e-6
