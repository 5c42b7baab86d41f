"""GPU parity: the HIP path through the C ABI vs the oracle, message for message."""
import numpy as np
import pytest

import readsb_amd
from oracle.oracle import Oracle, REC_DTYPE, first_mismatch
from readsb_amd.synth import generate

from scenarios import GPU_SCENARIOS, SCENARIOS
from test_oracle import load_demod

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", GPU_SCENARIOS)
def test_stream_matches_oracle_and_golden(name, built):
    sc, kw = SCENARIOS[name]
    iq = generate(sc)
    want, _ = Oracle(fmt=sc.fmt, nfix=kw["nfix"], modeac=kw["modeac"]).run_stream(iq)
    gold, _, _ = load_demod(name)  # what the unmodified reference produced for this input
    assert first_mismatch(gold, want) is None
    ch = readsb_amd.Channel(fmt=sc.fmt, fix_bits=kw["nfix"], mode_ac=kw["modeac"], max_batch_samples=1 << 23)
    got = ch.process_host(iq)
    assert first_mismatch(want, got.view(REC_DTYPE)) is None
    st = ch.stats()
    assert st["samples_processed"] == sc.n_samples
    ch.close()


@pytest.mark.parametrize("batch_blocks", [1, 3])
def test_batched_stream_equals_one_shot(batch_blocks, built):
    """Feeding the recording block-batch by block-batch (overlap carried on the device) changes nothing."""
    sc, kw = SCENARIOS["dense_fix1_ac"]
    iq = generate(sc)
    want, _ = Oracle(fmt=sc.fmt, nfix=kw["nfix"], modeac=kw["modeac"]).run_stream(iq)
    ch = readsb_amd.Channel(fmt=sc.fmt, fix_bits=kw["nfix"], mode_ac=kw["modeac"], max_batch_samples=batch_blocks * 131072)
    got = ch.process_stream_host(iq, batch_blocks)
    assert first_mismatch(want, got.view(REC_DTYPE)) is None
    ch.close()
