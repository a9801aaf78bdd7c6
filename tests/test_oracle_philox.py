"""oracle/philox_oracle.py against the Random123 known-answer vectors for philox4x32_10 (kat_vectors), + mask shape."""
import numpy as np

from oracle import philox_oracle as P


def _run(ctr, key):
    return [int(x) for x in P.philox4x32_10(np.array(ctr, dtype=np.uint32), np.array(key, dtype=np.uint32))]


def test_philox4x32_10_known_answers():
    assert _run([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _run([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _run([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_keep_mask_convention():
    assert P.keep_threshold(0.1) == 6554 and P.keep_threshold(0.0) == 0
    m = P.keep_mask(8 * 4096, 0.1, seed=7, offset=1)
    assert m.shape == (8 * 4096,) and abs(m.mean() - 0.9) < 0.01
    assert not np.array_equal(m, P.keep_mask(8 * 4096, 0.1, seed=7, offset=2))
    assert P.keep_mask(64, 0.0, 1, 1).all()
    r = P.philox4x32_10(np.array([5, 0, 1, 0], dtype=np.uint32), np.array([7, 0], dtype=np.uint32))
    thr = P.keep_threshold(0.1)
    exp = []
    for w in r:
        exp += [(int(w) & 0xffff) >= thr, (int(w) >> 16) >= thr]
    assert list(P.keep_mask(48, 0.1, 7, 1)[40:48]) == exp
