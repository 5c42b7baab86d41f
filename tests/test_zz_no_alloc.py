"""rsb_convert_noalloc (include/readsb_amd.h): the converter entry point that allocates nothing, against rsb_convert_device
(same kernels, work memory from the context) through the C ABI: magnitudes and means bit for bit, on two calls in a row, the
second with a smaller input (the fixed array keeps the first call's contents), the DC state carried from call to call."""
import ctypes as C

import numpy as np
import pytest

import readsb_amd

from scenarios import convert_input

pytestmark = pytest.mark.gpu

NA_MAX = 1 << 26


def _both(fmt, dc, sizes):
    import torch

    l = readsb_amd.lib()
    ch = readsb_amd.Channel(fmt=fmt, dc_filter=dc, max_batch_samples=max(sizes))
    state = torch.zeros(2, dtype=torch.float64, device="cuda")   # converter_state z1_I, z1_Q of the no-alloc calls
    iq_all = convert_input(fmt, sum(sizes))
    lo = 0
    for n in sizes:
        iq = torch.from_numpy(np.ascontiguousarray(iq_all[lo:lo + n])).cuda()
        lo += n
        want = torch.zeros(n, dtype=torch.int16, device="cuda")
        lvl, pwr = C.c_double(), C.c_double()
        rc = l.rsb_convert_device(ch.h, iq.data_ptr(), want.data_ptr(), n, None, C.byref(lvl), C.byref(pwr))
        assert rc == 0, l.rsb_last_error(ch.h)
        got = torch.full((n + 64,), 0x5555, dtype=torch.int16, device="cuda")  # guard words behind the output
        means = torch.full((2,), -1.0, dtype=torch.float64, device="cuda")
        rc = l.rsb_convert_noalloc(readsb_amd.FORMATS[fmt], int(dc), iq.data_ptr(), got.data_ptr(), n, None,
                                   state.data_ptr() if dc else None, means.data_ptr())
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(got[:n], want)
        assert bool((got[n:] == 0x5555).all())
        m = means.cpu().numpy()
        if dc:
            # the DC converters' block sums are double atomics taken in whatever order the wavefronts finish (k_convert_dc), in
            # both entry points alike: equal to rounding of a sum of doubles, 1e-12 relative (floating point: tolerance stated)
            assert abs(m[0] - lvl.value) <= 1e-12 * abs(lvl.value) and abs(m[1] - pwr.value) <= 1e-12 * abs(pwr.value)
        else:
            assert (float(m[0]), float(m[1])) == (lvl.value, pwr.value)
    ch.close()


@pytest.mark.parametrize("dc", [False, True], ids=["nodc", "dc"])
@pytest.mark.parametrize("fmt", ["UC8", "SC16", "SC16Q11"])
def test_noalloc_converter_equals_context_converter(fmt, dc, built):
    _both(fmt, dc, [300_000, 70_001])


def test_noalloc_without_means_and_limits(built):
    import torch

    l = readsb_amd.lib()
    n = 50_000
    iq = torch.from_numpy(convert_input("UC8", n)).cuda()
    ch = readsb_amd.Channel(fmt="UC8", max_batch_samples=n)
    want = torch.zeros(n, dtype=torch.int16, device="cuda")
    assert l.rsb_convert_device(ch.h, iq.data_ptr(), want.data_ptr(), n, None, None, None) == 0
    got = torch.zeros(n, dtype=torch.int16, device="cuda")
    assert l.rsb_convert_noalloc(0, 0, iq.data_ptr(), got.data_ptr(), n, None, None, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    # beyond the fixed array: refused before anything is launched
    assert l.rsb_convert_noalloc(0, 0, iq.data_ptr(), got.data_ptr(), NA_MAX + 1, None, None, None) == -1
    assert l.rsb_convert_noalloc(0, 1, iq.data_ptr(), got.data_ptr(), n, None, None, None) == -1   # dc without a state
    assert l.rsb_convert_noalloc(7, 0, iq.data_ptr(), got.data_ptr(), n, None, None, None) == -1   # unknown format
    ch.close()
