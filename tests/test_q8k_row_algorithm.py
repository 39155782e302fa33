"""The arithmetic of the streamed kernel's Q8_K activation quantizer (csrc/mmvq_stream.h st_prologue_q8k16: a 16-lane row per block, 16 elements per lane) restated in
numpy and held against the oracle's quantize_row_q8_K on random and adversarial blocks — the three places where the kernel's form differs from the reference's loop:
  * the first element of largest magnitude found by SIGN: a lane's largest / smallest element say whether +max, -max or both occur among its 16 (both: a scan);
  * round-to-nearest-even by adding 1.5 * 2^23 and taking the low byte of the bit pattern (the reference's own nearest_int), without MIN(127, .);
  * 16-element sums per lane.
No GPU: this pins the algorithm; tests/test_gpu_mul_mat.py pins the kernel's products against the same oracle."""
import numpy as np

import oracle as orc


def _row16(x):
    v = x.reshape(16, 16)
    pmax, nmin = v.max(1), v.min(1)
    amax = np.maximum(pmax, -nmin); rmax = amax.max()
    if rmax == 0:
        return np.zeros(256, np.int8), np.float32(0), np.zeros(16, np.int16)
    has_pos, has_neg = pmax == rmax, -nmin == rmax
    mx = np.where(has_neg, -rmax, rmax).astype(np.float32)
    if (has_pos & has_neg).any():
        for l in range(16):
            m = v[l, 15]
            for e in range(14, -1, -1):
                if abs(v[l, e]) == rmax:
                    m = v[l, e]
            mx[l] = m
    maxv = np.float32(mx[np.nonzero(amax == rmax)[0][0]])
    iscale = np.float32(-127.0) / maxv
    t = (iscale * x).astype(np.float32) + np.float32(12582912.0)
    q = (t.view(np.uint32) & 0xFF).astype(np.uint8).view(np.int8)
    return q, np.float32(1.0) / iscale, q.reshape(16, 16).astype(np.int32).sum(1).astype(np.int16)


def test_row16_quantizer_equals_reference_q8_K():
    rng = np.random.default_rng(0)
    for trial in range(1500):
        x = rng.standard_normal(256).astype(np.float32) * np.float32(10 ** rng.uniform(-3, 3))
        if trial % 4 == 0:        # equal magnitudes of both signs, in one lane's 16 elements or across lanes
            m = np.abs(x).max(); idx = rng.integers(0, 256, 4) if trial % 8 else rng.integers(0, 16, 4) + 16 * rng.integers(0, 16)
            x[idx] = m * np.array([1, -1, -1, 1], np.float32)[rng.permutation(4)]
        if trial % 7 == 0:        # rounding ties
            x = (np.round(x * 2) / 2).astype(np.float32)
        if trial == 11:
            x[:] = 0
        ref = orc.quantize(x[None, :], orc.Q8_K)[0].view(np.uint8)
        q, d, bs = _row16(x)
        assert np.array_equal(q, ref[4:260].view(np.int8)), trial
        assert d.view(np.uint32) == ref[0:4].view(np.float32)[0].view(np.uint32), trial
        assert np.array_equal(bs, ref[260:292].view(np.int16)), trial
