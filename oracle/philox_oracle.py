"""CPU restatement of the dropout RNG of the C ABI (TEST INFRASTRUCTURE ONLY): Philox4x32-10 (Salmon et al., "Parallel
random numbers: as easy as 1, 2, 3", SC'11 -- the generator family torch's CUDA dropout also uses) and the keep-mask
convention of include/dle_mi355x.h: counter = (chunk_lo, chunk_hi, offset_lo, offset_hi), key = (seed_lo, seed_hi),
one call per 8-element chunk, element 2k / 2k+1 of the chunk kept iff the low / high 16 bits of output word k are
>= thr = floor(p * 65536 + 0.5).  Pinned by the Random123 known-answer vectors (tests/test_oracle_philox.py)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: uint32 [..., 4], key: uint32 [..., 2] (broadcastable) -> uint32 [..., 4]."""
    c = [np.asarray(ctr[..., i], dtype=np.uint64) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint64)
    k1 = np.asarray(key[..., 1], dtype=np.uint64)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = [(hi1 ^ c[1] ^ k0) & MASK, lo1, (hi0 ^ c[3] ^ k1) & MASK, lo0]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return np.stack(c, axis=-1).astype(np.uint32)


def keep_threshold(p):
    return int(np.float32(p) * np.float32(65536.0) + np.float32(0.5))


def keep_mask(n_elements, p, seed, offset):
    """bool [n_elements] keep mask of one dropout call (n_elements % 8 == 0)."""
    chunks = np.arange(n_elements // 8, dtype=np.uint64)
    ctr = np.stack([chunks & MASK, chunks >> np.uint64(32),
                    np.full_like(chunks, np.uint64(offset) & MASK), np.full_like(chunks, np.uint64(offset) >> np.uint64(32))],
                   axis=-1).astype(np.uint32)
    key = np.array([np.uint64(seed) & MASK, np.uint64(seed) >> np.uint64(32)], dtype=np.uint32)
    r = philox4x32_10(ctr, key[None, :])                      # [chunks, 4]
    thr = keep_threshold(p)
    lo, hi = (r & np.uint32(0xFFFF)) >= thr, (r >> np.uint32(16)) >= thr
    return np.stack([lo, hi], axis=-1).reshape(-1)             # element 2k <- low half of word k, 2k+1 <- high half


def inv_keep(p):
    return np.float32(65536.0) / np.float32(65536 - keep_threshold(p))
