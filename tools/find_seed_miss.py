"""Search (CPU, oracle) for a pair of frames where the SECOND, a clean DF17 of an aircraft the filter does not know, starts inside the
greedy skip of the first (demod_2400.c:365-371) and would decode by itself: the device replay's optimistic first-add table then names
an item the walk does not accept, so the first round is not the fixed point.  Prints the parameters tests/test_gpu_parity.py uses."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.oracle import Oracle
from readsb_amd import synth
from readsb_amd.synth import TPU, TPS, df17, ap_short, modes_pulse_starts, _envelope


def compose(n, bursts, noise_seed=5):
    """bursts: (tick, msg, amp); all on the I axis (same carrier phase: amplitudes add)"""
    rng = np.random.Generator(np.random.PCG64(noise_seed))
    xi = np.zeros(n)
    for t0, msg, amp in bursts:
        s0, env = _envelope(modes_pulse_starts(msg), TPU // 2, int(t0))
        xi[s0:s0 + env.size] += amp * env
    xi += rng.normal(0, 1.0 / 127.5, n)
    xq = rng.normal(0, 1.0 / 127.5, n)
    iq = np.empty((n, 2), dtype=np.uint8)
    iq[:, 0] = np.clip(np.floor(127.5 + 127.5 * xi + 0.5), 0, 255).astype(np.uint8)
    iq[:, 1] = np.clip(np.floor(127.5 + 127.5 * xq + 0.5), 0, 255).astype(np.uint8)
    return iq


def main():
    rng = np.random.Generator(np.random.PCG64(11))
    X, Y = 0x4CA123, 0x3C6589
    o = Oracle(fmt="UC8", nfix=1, modeac=False)
    n = 4096
    for trial in range(20000):
        me_a = rng.integers(0, 256, size=7, dtype=np.uint8).tobytes()
        me_b = rng.integers(0, 256, size=7, dtype=np.uint8).tobytes()
        a, b = df17(Y, me_a), df17(X, me_b)
        back = int(rng.integers(9, 20))          # B starts this many bit periods before A's end
        sub = int(rng.integers(0, TPU))
        ta = 1000 * TPS + int(rng.integers(0, TPS))
        tb = ta + (8 + 112 - back) * TPU + sub
        amp_a, amp_b = 0.25 + 0.5 * rng.random(), 0.25 + 0.5 * rng.random()
        iq = compose(n, [(ta, a, amp_a), (tb, b, amp_b)])
        full, _ = o.run_stream(iq)
        got_a = any(bytes(r["msg"][:14]) == a for r in full)
        got_b = any(bytes(r["msg"][:14]) == b for r in full)
        if not got_a or got_b:
            continue
        cut = tb // TPS - 3
        alone, _ = o.run_stream(np.ascontiguousarray(iq[cut:]))
        if any(bytes(r["msg"][:14]) == b and r["correctedbits"] == 0 for r in alone):
            print("found", dict(trial=trial, me_a=me_a.hex(), me_b=me_b.hex(), back=back, sub=sub, ta=ta, amp_a=round(amp_a, 4), amp_b=round(amp_b, 4)))
            return
    print("none")


if __name__ == "__main__":
    main()
