"""GPU parity of rsb_stream (include/readsb_amd.h, "live sources"): a recording handed over block by block, as
plutosdrCallback / ifileRun do (sdr_plutosdr.c:168-244, sdr_ifile.c:176-253), must give the oracle's message list."""
import numpy as np
import pytest

import readsb_amd
from oracle.oracle import Oracle, REC_DTYPE, first_mismatch
from readsb_amd.synth import Scenario, generate

pytestmark = pytest.mark.gpu

BLOCK = 131072


def push_blocks(st, iq, flush_after=()):
    n = iq.shape[0]
    for k, pos in enumerate(range(0, n, BLOCK)):
        st.push(iq[pos:pos + BLOCK])
        if k in flush_after:
            st.flush()


@pytest.mark.parametrize("fmt,modeac,nfix,blocks,tail", [("UC8", True, 1, 3, 5000), ("SC16", False, 2, 2, 0), ("SC16Q11", True, 1, 4, 77)])
def test_blocks_in_batches_match_the_oracle(fmt, modeac, nfix, blocks, tail, built):
    sc = Scenario(7 * BLOCK + tail, seed=23, fmt=fmt, frames_per_sec=4000, modeac_per_sec=300 if modeac else 0, err2_every=11)
    iq = generate(sc)
    want, _ = Oracle(fmt=fmt, nfix=nfix, modeac=modeac).run_stream(iq)
    assert want.size > 50
    ch = readsb_amd.Channel(fmt=fmt, fix_bits=nfix, mode_ac=modeac, max_batch_samples=blocks * BLOCK)
    st = readsb_amd.Stream(ch, blocks)
    push_blocks(st, iq, flush_after=(4,))  # a timer's flush in the middle: a batch of fewer blocks
    st.close()
    assert len(st.parts) >= 3              # delivered batch by batch, not at the end
    assert first_mismatch(want, st.messages().view(REC_DTYPE)) is None
    assert ch.stats()["samples_processed"] == sc.n_samples
    ch.close()


def test_dropped_blocks_move_the_clock_and_zero_the_carry(built):
    """mag_buf.dropped (sdr_plutosdr.c:208-228): whole blocks lost between two blocks.  Parity unpinned against the
    reference (its live handlers cannot run here); pinned against the same context driven by hand: seek + zero carry,
    filter kept."""
    a = generate(Scenario(3 * BLOCK, seed=31, frames_per_sec=4000))
    b = generate(Scenario(2 * BLOCK + 999, seed=32, frames_per_sec=4000))
    ch = readsb_amd.Channel(max_batch_samples=4 * BLOCK)
    st = readsb_amd.Stream(ch, 4)
    push_blocks(st, a)
    st.push(b[:BLOCK], dropped_before=2 * BLOCK)
    push_blocks(st, b[BLOCK:])
    st.close()
    got = st.messages()
    ch.close()

    ref = readsb_amd.Channel(max_batch_samples=4 * BLOCK)
    part_a = ref.process_host(a)
    ref.seek(5 * BLOCK, None)
    part_b = ref.process_host(b)
    ref.close()
    want = np.concatenate([part_a, part_b])
    assert part_b.size > 20 and int(part_b["ts"].min()) >= 5 * (5 * BLOCK)
    assert first_mismatch(want.view(REC_DTYPE), got.view(REC_DTYPE)) is None
    # the first part alone is the oracle's list for those blocks
    oa, _ = Oracle(fmt="UC8", nfix=1, modeac=False).run_stream(a)
    assert first_mismatch(oa, got[:part_a.size].view(REC_DTYPE)) is None


def test_call_sequence_errors(built):
    ch = readsb_amd.Channel(max_batch_samples=2 * BLOCK)
    with pytest.raises(readsb_amd.RsbError):
        readsb_amd.Stream(ch, 3)                      # batch larger than the context allows
    st = readsb_amd.Stream(ch, 2)
    iq = generate(Scenario(BLOCK + 100, seed=5))
    with pytest.raises(readsb_amd.RsbError):
        st.push(iq[:BLOCK], dropped_before=1000)      # not whole blocks
    st.push(iq[:BLOCK])
    st.push(iq[BLOCK:])                               # short block: ends the stream
    with pytest.raises(readsb_amd.RsbError):
        st.push(iq[:BLOCK])
    st.close()
    want, _ = Oracle(fmt="UC8", nfix=1, modeac=False).run_stream(iq)
    assert first_mismatch(want, st.messages().view(REC_DTYPE)) is None
    ch.close()


@pytest.mark.parametrize("fmt", ["UC8", "SC16"])
def test_dcfilter_stream_equals_batched_host_calls(fmt, built):
    """A --dcfilter context holds ONE batch at a time (one magnitude buffer): the stream has to collect the batch in flight
    before it submits the next.  Five batches + a ragged tail; the same blocks through rsb_process_host on a second context
    of the same kind give the same messages, record for record (same kernels, same DC state carried from batch to batch)."""
    sc = Scenario(9 * BLOCK + 4321, seed=41, fmt=fmt, frames_per_sec=3000, dc_offset=0.02)
    iq = generate(sc)
    ch = readsb_amd.Channel(fmt=fmt, fix_bits=1, dc_filter=True, max_batch_samples=2 * BLOCK)
    st = readsb_amd.Stream(ch, 2)
    push_blocks(st, iq)
    st.close()
    got = st.messages()
    assert len(st.parts) >= 5 and ch.stats()["samples_processed"] == sc.n_samples
    ch.close()
    ref = readsb_amd.Channel(fmt=fmt, fix_bits=1, dc_filter=True, max_batch_samples=2 * BLOCK)
    want = ref.process_stream_host(iq, 2)
    ref.close()
    assert want.size > 100
    assert first_mismatch(want.view(REC_DTYPE), got.view(REC_DTYPE)) is None


def test_sc16q11_table_context_streams_one_batch_at_a_time(built):
    """RSB_FLAG_SC16Q11_TABLE contexts hold one magnitude buffer like --dcfilter ones: the stream collects before it submits.  Bit for
    bit against the oracle built the same way (convert.c:264-328), Mode A/C on (integer block means of that converter)."""
    sc = Scenario(7 * BLOCK + 999, seed=43, fmt="SC16Q11", frames_per_sec=3000, modeac_per_sec=300)
    iq = generate(sc)
    ch = readsb_amd.Channel(fmt="SC16Q11", fix_bits=1, mode_ac=True, q11_table_bits=8, max_batch_samples=2 * BLOCK)
    st = readsb_amd.Stream(ch, 2)
    push_blocks(st, iq)
    st.close()
    got = st.messages()
    assert len(st.parts) >= 4 and ch.stats()["samples_processed"] == sc.n_samples
    ch.close()
    want, _ = Oracle(fmt="SC16Q11", nfix=1, modeac=True, q11_table_bits=8).run_stream(iq)
    assert want.size > 100
    assert first_mismatch(want, got.view(REC_DTYPE)) is None


def _bladerf_frames(iq, block_size, ts0, decimation=1, flags=None):
    """SC16Q11 samples -> the byte stream of bladeRF metadata sub-blocks (sdr_bladerf.c:335-345): magic, timestamp, flags, samples"""
    per = (block_size - 16) // 4
    assert iq.shape[0] % per == 0
    nb = iq.shape[0] // per
    out = np.zeros((nb, block_size), dtype=np.uint8)
    hdr = np.zeros(nb, dtype=[("magic", "<u4"), ("ts", "<u8"), ("flags", "<u4")])
    hdr["magic"] = 0x12344321
    hdr["ts"] = ts0 + np.arange(nb, dtype=np.uint64) * np.uint64(per * decimation)
    if flags is not None:
        hdr["flags"] = flags
    out[:, :16] = hdr.view(np.uint8).reshape(nb, 16)
    out[:, 16:] = np.ascontiguousarray(iq).view(np.uint8).reshape(nb, per * 4)
    return out.reshape(-1)


def test_bladerf_framing_is_stripped(built):
    """rsb_stream_push_bladerf (sdr_bladerf.c:335-394).  PARITY UNPINNED against the reference (the handler needs libbladeRF);
    pinned here: (a) framed transfer buffers decode exactly as the oracle decodes the de-framed samples, overrun flags are
    counted, a timer's flush in the middle only sends whole blocks; (b) a jump of the hardware timestamp drops what the buffer
    held before it, restarts from a zero carry and moves the message timestamps by the samples lost."""
    block_size, per = 1024, 252
    nbuf = 131072 * 4 // block_size * per          # samples per transfer buffer: 129024
    sc = Scenario(5 * nbuf, seed=41, fmt="SC16Q11", frames_per_sec=4000)
    iq = generate(sc)
    want, _ = Oracle(fmt="SC16Q11", nfix=1).run_stream(iq)
    assert want.size > 500
    ch = readsb_amd.Channel(fmt="SC16Q11", fix_bits=1, max_batch_samples=2 * BLOCK)
    st = readsb_amd.Stream(ch, 2)
    ts0 = 123456789
    flags = np.zeros(nbuf // per, dtype=np.uint32)
    flags[[3, 9]] = 1
    for k in range(5):
        over = st.push_bladerf(_bladerf_frames(iq[k * nbuf:(k + 1) * nbuf], block_size, ts0 + k * nbuf, flags=flags if k == 1 else None), block_size)
        assert over == (2 if k == 1 else 0)
        if k == 2:
            st.flush()
    st.close()
    assert first_mismatch(want, st.messages().view(REC_DTYPE)) is None
    assert ch.stats()["samples_processed"] == sc.n_samples
    ch.close()

    # (b) two buffers, then a buffer whose sub-blocks 0..99 are in sequence and whose sub-block 100 comes 10000 samples late
    a, b = iq[:2 * nbuf + 100 * per], iq[3 * nbuf:4 * nbuf]
    lost = 10000
    ch = readsb_amd.Channel(fmt="SC16Q11", fix_bits=1, max_batch_samples=2 * BLOCK)
    st = readsb_amd.Stream(ch, 2)
    st.push_bladerf(_bladerf_frames(a[:nbuf], block_size, ts0), block_size)
    st.push_bladerf(_bladerf_frames(a[nbuf:2 * nbuf], block_size, ts0 + nbuf), block_size)
    third = np.concatenate([_bladerf_frames(a[2 * nbuf:], block_size, ts0 + 2 * nbuf),
                            _bladerf_frames(b[:nbuf - 100 * per], block_size, ts0 + 2 * nbuf + 100 * per + lost)])
    st.push_bladerf(third, block_size)
    st.close()
    got = st.messages()
    ch.close()
    pos_b = 2 * nbuf + 100 * per + lost          # stream position of b's first sample
    ref = readsb_amd.Channel(fmt="SC16Q11", fix_bits=1, max_batch_samples=8 * BLOCK)
    part_a = ref.process_host(a[:2 * nbuf])      # the third buffer's first 100 sub-blocks are dropped with the jump
    ref.seek(pos_b // BLOCK * BLOCK, None)
    padded = np.concatenate([np.zeros((pos_b % BLOCK, 2), dtype=iq.dtype), b[:nbuf - 100 * per]])
    part_b = ref.process_host(padded)
    ref.close()
    assert part_b.size > 50 and int(part_b["ts"].min()) >= 5 * pos_b
    assert first_mismatch(np.concatenate([part_a, part_b]).view(REC_DTYPE), got.view(REC_DTYPE)) is None
