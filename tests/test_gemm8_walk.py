"""The walk order of the persistent ping-pong GEMM (deeplearningexamples_amd/csrc/gemm8_walk.h: the function the kernel fills
its per-workgroup LDS table with) compiled for the HOST and checked as plain integer logic:
  * every (tile, K slice) item is visited exactly once over all walk positions, for shapes with ragged tile grids, item counts that
    are not multiples of 8 (XCD chunks of unequal length) or of the workgroup count, and every K split;
  * the K slices of a tile partition its K tiles, in order, none empty while splitk <= ktiles (the launcher's envelope);
  * locality, the reason for the order: the positions of one XCD (vb & 7) are a CONTIGUOUS run of the item list, and the items
    the CUs of an XCD run at the same time (one round of the walk) touch at most gm + ceil(32 / gm) + 1 distinct tile rows +
    columns' worth of operand panels instead of up to 32 + 1.
CPU only (g++); the GPU side of the same function is covered by tests/test_gpu_gemm8.py."""
import ctypes
import os
import subprocess
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER_DIR = os.path.join(os.path.dirname(HERE), "deeplearningexamples_amd", "csrc")

SRC = r"""
#include "gemm8_walk.h"
extern "C" void walk_all(int nitems, int tiles_m, int tiles_n, int ktiles, int splitk, int gm, int* out) {
  for (int vb = 0; vb < nitems; ++vb) {
    const G8WalkItem w = g8_walk_item(vb, nitems, tiles_m, tiles_n, ktiles, splitk, gm);
    out[4 * vb + 0] = w.m0; out[4 * vb + 1] = w.n0; out[4 * vb + 2] = w.kt0; out[4 * vb + 3] = w.slice_and_tiles;
  }
}
"""


@pytest.fixture(scope="module")
def walk():
    d = tempfile.mkdtemp(prefix="g8walk_")
    src, so = os.path.join(d, "walk.cpp"), os.path.join(d, "walk.so")
    open(src, "w").write(SRC)
    subprocess.run(["g++", "-O1", "-shared", "-fPIC", "-I", HEADER_DIR, src, "-o", so], check=True)
    lib = ctypes.CDLL(so)
    lib.walk_all.argtypes = [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_int)]

    def run(tiles_m, tiles_n, ktiles, splitk, gm=8):
        n = tiles_m * tiles_n * splitk
        buf = (ctypes.c_int * (4 * n))()
        lib.walk_all(n, tiles_m, tiles_n, ktiles, splitk, gm, buf)
        return [(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3] >> 16, buf[4 * i + 3] & 0xFFFF) for i in range(n)]
    return run


SHAPES = [  # tiles_m, tiles_n, ktiles, splitk, gm
    (128, 16, 16, 1, 8),        # BERT FFN forward: 32768 x 4096 x 1024
    (16, 4, 512, 4, 8),         # BERT FFN weight gradient, 4 slices
    (4, 4, 512, 16, 8),         # attention-output weight gradient
    (256, 4, 16, 1, 8),         # DLRM top MLP
    (23, 13, 3, 1, 8),          # 299 items: not a multiple of 8, tile rows not a multiple of gm
    (32, 10, 2, 1, 8),
    (5, 7, 9, 3, 8),            # slices of unequal length (9 K tiles in 3 ... and below in 2 / 4 / 5)
    (5, 7, 9, 2, 8), (5, 7, 9, 4, 8), (5, 7, 9, 5, 8), (5, 7, 9, 9, 8),
    (1, 1, 40, 7, 8), (1, 9, 2, 2, 8), (9, 1, 2, 1, 8),
    (20, 120, 16, 1, 4), (20, 120, 16, 1, 16), (3, 3, 2, 1, 1),
]


@pytest.mark.parametrize("tiles_m,tiles_n,ktiles,splitk,gm", SHAPES)
def test_every_item_once_and_slices_partition_k(walk, tiles_m, tiles_n, ktiles, splitk, gm):
    items = walk(tiles_m, tiles_n, ktiles, splitk, gm)
    assert len(items) == tiles_m * tiles_n * splitk
    seen = {}
    for (m0, n0, kt0, ky, nkt) in items:
        assert m0 % 256 == 0 and n0 % 256 == 0 and 0 <= m0 < 256 * tiles_m and 0 <= n0 < 256 * tiles_n
        assert 0 <= ky < splitk and nkt >= 1 and 0 <= kt0 and kt0 + nkt <= ktiles
        key = (m0, n0, ky)
        assert key not in seen, "item visited twice"
        seen[key] = (kt0, nkt)
    for tm in range(tiles_m):
        for tn in range(tiles_n):
            at = 0
            for ky in range(splitk):
                kt0, nkt = seen[(256 * tm, 256 * tn, ky)]
                assert kt0 == at
                at += nkt
            assert at == ktiles


@pytest.mark.parametrize("tiles_m,tiles_n,ktiles,splitk,gm", SHAPES)
def test_xcd_chunks_are_contiguous_runs_of_the_list(walk, tiles_m, tiles_n, ktiles, splitk, gm):
    """List index of an item = ((slice * groups + group) ...) in the documented order; reconstruct it and check that XCD x's
    positions x, x + 8, x + 16, ... map to consecutive list indices, the chunks tile the list in XCD order."""
    items = walk(tiles_m, tiles_n, ktiles, splitk, gm)
    n = len(items)

    def list_index(m0, n0, ky):
        tm, tn = m0 // 256, n0 // 256
        g = tm // gm
        rows = min(gm, tiles_m - g * gm)
        return ky * tiles_m * tiles_n + g * gm * tiles_n + tn * rows + (tm - g * gm)
    idx = [list_index(m0, n0, ky) for (m0, n0, kt0, ky, nkt) in items]
    assert sorted(idx) == list(range(n))
    start = 0
    for x in range(8):
        chunk = idx[x::8]
        assert chunk == list(range(start, start + len(chunk)))
        start += len(chunk)
    assert start == n


def test_one_round_of_an_xcd_is_a_compact_block(walk):
    """BERT's 32768 x 4096 forward on 256 CUs: the 32 items the CUs of one XCD run at the same time (walk round r: positions
    x + 8 j + 256 r, j < 32) cover at most gm = 8 tile rows and 32 / 8 + 1 tile columns."""
    items = walk(128, 16, 16, 1, 8)
    grid = 256
    for r in range(len(items) // grid):
        for x in range(8):
            block = [items[x + 8 * j + grid * r] for j in range(32)]
            rows = {b[0] for b in block}
            cols = {b[1] for b in block}
            assert len(rows) <= 8 + 8 and len(cols) <= 5        # (a round may straddle two row groups)
            assert len(rows) * len(cols) <= 80                  # against 32 x 1 + 1 x 32 for a row-major walk: 33 panels -> <= 21
            assert len(rows) + len(cols) <= 21
