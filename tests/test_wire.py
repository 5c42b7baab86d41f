"""Wire formats of the message list: Beast binary and raw (AVR) output, and the two-message
forwarding rule.  The fixtures tests/golden/wire_*.npz hold the bytes the UNMODIFIED reference's own writers
(net_io.c:760-826, 859-885, through modesQueueOutput / the real useModesMessage + tracker) produced for seeded
scenarios.  CPU tests: oracle restatement and the product's host encoder against those bytes.  GPU tests: the
device encoder (rsb_collect_wire) against the oracle."""
import os

import numpy as np
import pytest

import readsb_amd
from oracle import oracle as orc
from oracle.oracle import Oracle, REC_DTYPE, first_mismatch
from readsb_amd.synth import Scenario, generate

from scenarios import SCENARIOS, WIRE_SCENARIOS
from test_oracle import GOLD


def load_wire(name):
    z = np.load(os.path.join(GOLD, f"wire_{name}.npz"))
    return z["msgs"].reshape(-1).view(REC_DTYPE), z


@pytest.mark.parametrize("name", WIRE_SCENARIOS)
def test_oracle_writers_match_reference_bytes(name):
    recs, z = load_wire(name)
    assert np.array_equal(orc.beast_encode(recs), z["beast"])
    assert np.array_equal(orc.raw_encode(recs, mlat=True), z["raw"])
    fwd = recs[orc.forward_order(recs)]
    assert np.array_equal(orc.beast_encode(fwd), z["beast_fwd"])
    assert np.array_equal(orc.raw_encode(fwd, mlat=True), z["raw_fwd"])


@pytest.mark.parametrize("name", WIRE_SCENARIOS)
def test_host_encoder_matches_reference_bytes(name):
    recs, z = load_wire(name)
    msgs = recs.view(readsb_amd.MSG_DTYPE)
    assert np.array_equal(readsb_amd.wire_encode(msgs, "beast"), z["beast"])
    assert np.array_equal(readsb_amd.wire_encode(msgs, "raw", mlat=True), z["raw"])
    # the forwarding rule, fed in three uneven pieces (its state carries over)
    f = readsb_amd.Forwarder()
    cuts = [0, msgs.size // 3, msgs.size // 3 + 1, msgs.size]
    fwd = np.concatenate([f.push(msgs[cuts[k]:cuts[k + 1]]) for k in range(3)])
    f.close()
    assert np.array_equal(readsb_amd.wire_encode(fwd, "beast"), z["beast_fwd"])
    assert np.array_equal(readsb_amd.wire_encode(fwd, "raw", mlat=True), z["raw_fwd"])


def synthetic_records(n=4000, seed=5):
    """records that hit every branch of the writers: 0x1a in every escaped position, all three lengths, zero
    timestamps, two corrected bits, signal levels at both clamps and on rounding boundaries"""
    rng = np.random.Generator(np.random.PCG64(seed))
    m = np.zeros(n, dtype=REC_DTYPE)
    m["ts"] = rng.integers(0, 1 << 48, n, dtype=np.uint64)
    esc = rng.random(n) < 0.3
    m["ts"][esc] = (m["ts"][esc] & np.uint64(0xFFFF00FFFF00)) | np.uint64(0x00001A00001A)
    m["ts"][::97] = 0
    m["msg"] = rng.integers(0, 256, (n, 14), dtype=np.uint8)
    m["msg"][rng.random((n, 14)) < 0.1] = 0x1A
    kind = rng.integers(0, 3, n)
    m["msgbits"] = np.array([56, 112, 16], dtype=np.uint8)[kind]
    m["msgtype"] = np.where(kind == 2, 32, m["msg"][:, 0] >> 3)
    m["correctedbits"] = rng.integers(0, 3, n)
    m["addr"] = rng.integers(0, 40, n)
    lvl = rng.random(n) ** 4
    lvl[::11] = 0.0
    lvl[1::11] = 1e-9                                           # > 0 but rounds to 0 -> 1
    lvl[2::11] = ((rng.integers(0, 255, lvl[2::11].size) + 0.5) / 255.0) ** 2  # sqrt * 255 near k + 0.5
    lvl[3::11] = (26.0 / 255.0) ** 2                             # RSSI byte 0x1a: escaped
    lvl[4::11] = 1.7
    m["signal"] = lvl
    return m


@pytest.mark.parametrize("wire,mlat", [("beast", True), ("raw", True), ("raw", False)])
def test_host_encoder_matches_oracle_on_branch_cases(wire, mlat):
    m = synthetic_records()
    want = orc.beast_encode(m) if wire == "beast" else orc.raw_encode(m, mlat=mlat)
    got = readsb_amd.wire_encode(m.view(readsb_amd.MSG_DTYPE), wire, mlat=mlat)
    assert np.array_equal(got, want)
    assert want.size > 0
    # forwarding rule on the same list (Mode A/C and address 0 pass, the rest wait for a second message)
    f = readsb_amd.Forwarder()
    got_f = f.push(m.view(readsb_amd.MSG_DTYPE))
    assert np.array_equal(got_f.view(REC_DTYPE), m[orc.forward_order(m)])


def test_host_encoder_edges():
    import ctypes as C
    l = readsb_amd.lib()
    nb = C.c_size_t(7)
    assert l.rsb_wire_encode(0, 0, None, 0, None, 0, C.byref(nb)) == 0 and nb.value == 0  # empty list
    m = synthetic_records(50).view(readsb_amd.MSG_DTYPE)
    full = readsb_amd.wire_encode(m, "beast")
    small = np.zeros(full.size - 5, dtype=np.uint8)
    rc = l.rsb_wire_encode(0, 0, m.ctypes.data, m.size, small.ctypes.data, small.size, C.byref(nb))
    assert rc == -4 and nb.value == full.size and np.array_equal(small, full[:small.size])  # RSB_EOVERFLOW, size reported
    assert l.rsb_wire_encode(2, 0, m.ctypes.data, m.size, small.ctypes.data, small.size, C.byref(nb)) == -1  # unknown format
    assert readsb_amd.wire_encode(m[:0], "raw").size == 0


# ---------------------------------------------------------------------------------------------- GPU

def _submit(ch, iq):
    import torch
    raw = np.ascontiguousarray(iq).view(np.uint8).reshape(-1, ch.bps)
    dev = torch.from_numpy(raw).cuda()
    ch.submit(dev.data_ptr(), raw.shape[0], None)
    return dev


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mixed_fix1", "dense_fix2_ac", "sc16", "nofix", "tiny_200"])
@pytest.mark.parametrize("wire", ["beast", "raw"])
def test_collect_wire_matches_oracle(name, wire, built):
    """device encoder (batches without Mode A/C) and host merge + encoder (with): same bytes as the oracle's writers
    over the oracle's message list"""
    sc, kw = SCENARIOS[name]
    iq = generate(sc)
    want_msgs, _ = Oracle(fmt=sc.fmt, nfix=kw["nfix"], modeac=kw["modeac"]).run_stream(iq)
    want = orc.beast_encode(want_msgs) if wire == "beast" else orc.raw_encode(want_msgs, mlat=True)
    ch = readsb_amd.Channel(fmt=sc.fmt, fix_bits=kw["nfix"], mode_ac=kw["modeac"], max_batch_samples=1 << 22)
    dev = _submit(ch, iq)
    got, nmsgs = ch.collect_wire(wire, mlat=True, cap_bytes=44 * want_msgs.size + 64)
    assert nmsgs == want_msgs.size
    assert np.array_equal(got, want)
    del dev
    ch.close()


@pytest.mark.gpu
def test_collect_wire_small_buffer_and_then_records(built):
    """a buffer that is too small reports the size needed; the channel goes on with the next batch as usual"""
    sc = Scenario(4 * 131072, seed=77, frames_per_sec=4000)
    iq = generate(sc)
    want_msgs, _ = Oracle(fmt="UC8", nfix=1).run_stream(iq)
    want = orc.beast_encode(want_msgs)
    half = 2 * 131072
    ch = readsb_amd.Channel(fmt="UC8", fix_bits=1, max_batch_samples=half)
    import torch
    dev = torch.from_numpy(np.ascontiguousarray(iq).view(np.uint8).reshape(-1, 2)).cuda()
    ch.submit(dev.data_ptr(), half, None)
    with pytest.raises(readsb_amd.RsbError, match="exceed cap"):
        ch.collect_wire("beast", cap_bytes=16)
    ch.submit(dev[half:].data_ptr(), half, None)
    second = ch.collect(cap=1 << 16)
    assert 0 < second.size < want_msgs.size and want.size > 16
    assert first_mismatch(want_msgs[want_msgs.size - second.size:], second.view(REC_DTYPE)) is None
    ch.close()
