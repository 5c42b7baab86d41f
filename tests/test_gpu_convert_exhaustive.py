"""The float magnitude (convert.c:226-262, 343-379) on the device uses its own correctly rounded square root (kernels.hip
float_mag).  Bit for bit against the oracle's sqrtf on EVERY SC16Q11 sample value (by symmetry |I|, |Q| in 0..2048) and on a
dense set of SC16 values: every |I| against 256 values of |Q|, the neighbourhood of the clamp at magsq = 1, and 16 M random pairs."""
import ctypes as C

import numpy as np
import pytest

import readsb_amd
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def _device_convert(fmt, iq):
    import torch

    l = readsb_amd.lib()
    n = iq.shape[0]
    ch = readsb_amd.Channel(fmt=fmt, max_batch_samples=131072)
    d_iq = torch.from_numpy(np.ascontiguousarray(iq)).cuda()
    d_mag = torch.zeros(n, dtype=torch.int16, device="cuda")
    assert l.rsb_convert_device(ch.h, d_iq.data_ptr(), d_mag.data_ptr(), n, None, None, None) == 0, l.rsb_last_error(ch.h)
    torch.cuda.synchronize()
    ch.close()
    return d_mag.cpu().numpy().view(np.uint16)


def _check(fmt, iq):
    got = _device_convert(fmt, iq)
    orc = Oracle(fmt=fmt)
    step = 1 << 22
    for lo in range(0, iq.shape[0], step):
        want = orc.convert(iq[lo:lo + step])[0]
        bad = np.flatnonzero(want != got[lo:lo + step])
        assert bad.size == 0, (fmt, iq[lo + bad[0]].tolist(), int(want[bad[0]]), int(got[lo + bad[0]]))


def test_every_sc16q11_magnitude(built):
    v = np.arange(0, 2049, dtype=np.int64)
    v[-1] = -2048  # |-2048| = 2048: the only way to reach it in 12 bits
    i, q = np.meshgrid(v, v, indexing="ij")
    _check("SC16Q11", np.stack([i.ravel(), q.ravel()], axis=1).astype("<i2"))


def test_dense_sc16_magnitudes(built):
    rng = np.random.Generator(np.random.PCG64(2024))
    allv = np.arange(0, 32769, dtype=np.int64)
    allv[-1] = -32768
    qs = np.unique(np.concatenate([np.array([0, 1, 2, 3, 255, 256, 4095, 4096, 16384, 23170, 23171, 32767, -32768]), rng.integers(0, 32768, 243)]))
    i, q = np.meshgrid(allv, qs, indexing="ij")
    grid = np.stack([i.ravel(), q.ravel()], axis=1)
    # around the clamp: I^2 + Q^2 within a few counts of 2^30
    ci = rng.integers(0, 32768, 1 << 22)
    cq = np.sqrt(np.maximum(0.0, 2.0 ** 30 - ci.astype(np.float64) ** 2)).astype(np.int64) + rng.integers(-2, 3, ci.size)
    circle = np.stack([ci, np.clip(cq, 0, 32767)], axis=1)
    rnd = rng.integers(-32768, 32768, size=(1 << 24, 2))
    _check("SC16", np.concatenate([grid, circle, rnd]).astype("<i2"))
