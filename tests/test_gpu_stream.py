"""GPU parity of rsb_stream (include/readsb_amd.h, "live sources"): a recording handed over block by block, as
plutosdrCallback / ifileRun do (sdr_plutosdr.c:168-244, sdr_ifile.c:176-253), must give the oracle's message list."""
import numpy as np
import pytest

import readsb_amd
from oracle.oracle import Oracle, REC_DTYPE, first_mismatch
from readsb_amd.synth import Scenario, generate

pytestmark = pytest.mark.gpu

REPLAY = pytest.mark.parametrize("host_replay", [False, True], ids=["device_replay", "host_replay"])

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


def test_bladerf_transfer_sequence_matches_the_reference_chain(built):
    """SURVEY 8 f3, rsb_stream_push_bladerf = handle_bladerf_samples (sdr_bladerf.c:283-422): framed transfer buffers in, the mag_buf
    chain of the reference out -- first sub-block dropped (the counter does not start at 0), a forward jump inside a buffer, a backward
    jump, a wrong magic inside a buffer; buffers of 129024 samples (block_samples) whose means are the averages of the 252-sample
    sub-blocks' means (mean_sub_samples).  Compared with (a) the golden list the unmodified reference's converter and demodulators
    produced for that chain (ref_harness demodbufs; the handler's framing logic itself needs libbladeRF: the chain is
    tests/scenarios.py's restatement of it) and (b) the oracle."""
    import os

    from scenarios import (BLADERF_BLOCK_SIZE, BLADERF_BUF, BLADERF_KW, BLADERF_PER, BLADERF_SCENARIO, bladerf_bytes, bladerf_model,
                           bladerf_transfers)

    iq = generate(BLADERF_SCENARIO)
    transfers = bladerf_transfers(iq)
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demod_bladerf_chain.npz"))
    gold = z["msgs"].reshape(-1).view(REC_DTYPE)
    want, wst = Oracle(fmt="SC16Q11", nfix=BLADERF_KW["nfix"], modeac=True).run_bufs(iq, bladerf_model(transfers), sub_len=BLADERF_PER)
    assert first_mismatch(gold, want) is None and want.size > 500
    for blocks in (1, 3):
        ch = readsb_amd.Channel(fmt="SC16Q11", fix_bits=BLADERF_KW["nfix"], mode_ac=True, max_batch_samples=blocks * BLADERF_BUF,
                                block_samples=BLADERF_BUF, mean_sub_samples=BLADERF_PER)
        st = readsb_amd.Stream(ch, blocks)
        overruns = [st.push_bladerf(bladerf_bytes(iq, t), BLADERF_BLOCK_SIZE) for t in transfers]
        st.close()
        assert overruns == [0, 2, 0, 0, 0, 0, 0, 0]
        assert first_mismatch(gold, st.messages().view(REC_DTYPE)) is None
        stats = ch.stats()
        assert stats["demod_modeac"] == wst["modeac"] and stats["demod_accepted"] == [wst["accepted0"], wst["accepted1"], wst["accepted2"]]
        ch.close()
    # a context of another geometry is refused, with the reason
    ch = readsb_amd.Channel(fmt="SC16Q11", max_batch_samples=2 * BLOCK)
    st = readsb_amd.Stream(ch, 2)
    with pytest.raises(readsb_amd.RsbError, match="block_samples must be the samples of one transfer buffer"):
        st.push_bladerf(bladerf_bytes(iq, transfers[0]), BLADERF_BLOCK_SIZE)
    st.close()
    ch.close()


@REPLAY
def test_block_samples_through_plain_batches(host_replay, built):
    """rsb_config.block_samples / mean_sub_samples without the stream layer: a recording submitted in batches of whole 129024-sample
    blocks decodes as the oracle's chain of such buffers does (skip windows, Mode A/C runs and noise levels per buffer), on the device
    replay and on the host replay."""
    from scenarios import BLADERF_BUF, BLADERF_PER, BLADERF_SCENARIO

    iq = generate(BLADERF_SCENARIO)[:5 * BLADERF_BUF + 7 * BLADERF_PER]
    n = iq.shape[0]
    bufs = [(p, min(BLADERF_BUF, n - p), 5 * p) for p in range(0, n, BLADERF_BUF)]
    want, _ = Oracle(fmt="SC16Q11", nfix=2, modeac=True).run_bufs(iq, bufs, sub_len=BLADERF_PER)
    ch = readsb_amd.Channel(fmt="SC16Q11", fix_bits=2, mode_ac=True, host_replay=host_replay, max_batch_samples=2 * BLADERF_BUF,
                            block_samples=BLADERF_BUF, mean_sub_samples=BLADERF_PER)
    got = ch.process_stream_host(iq, 2, block=BLADERF_BUF)
    ch.close()
    assert want.size > 300 and first_mismatch(want, got.view(REC_DTYPE)) is None


def test_block_geometry_is_validated(built, capfd):
    for kw in (dict(block_samples=129025), dict(block_samples=2048), dict(block_samples=129024, dc_filter=True),
               dict(block_samples=129024, mean_sub_samples=250), dict(fmt="UC8", block_samples=129024, mean_sub_samples=252)):
        kw.setdefault("fmt", "SC16Q11")
        with pytest.raises(readsb_amd.RsbError):
            readsb_amd.Channel(**kw)
    assert "block_samples must be a multiple of 8" in capfd.readouterr().err
