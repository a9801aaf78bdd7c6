// The walk order of the persistent ping-pong GEMM (gemm8_kernel.h), as one function the kernel and a host-side test share.
//
// A launch has nitems = tiles_m x tiles_n x splitk (tile, K slice) items and grid = min(nitems, #CUs) workgroups; workgroup b
// walks positions vb = b, b + grid, b + 2 grid, ...  Position -> item:
//   * workgroup b runs on XCD b % 8 (private L2), so position vb belongs to XCD vb & 7; each XCD owns a CONTIGUOUS chunk of the
//     item list (the first nitems % 8 XCDs one item more), walked in order: what an XCD's CUs run at the same time are
//     neighbours in the list;
//   * the list orders K slices slowest, then groups of gm tile rows, and inside a group column by column (tile rows fastest):
//     the ~32 tiles an XCD runs at once form a gm x (32 / gm) block of the output -- per K step they pull gm A half-tiles and a
//     few B ones through L2 instead of 1 + 32;
//   * K slice ky of a tile covers K tiles [ky * ktiles / splitk, (ky + 1) * ktiles / splitk).
// Record: {first row, first column, first K tile, (K slice << 16) | number of K tiles}.
#pragma once
#if defined(__HIPCC__)
#define G8_WALK_FN __host__ __device__ __forceinline__
#else
#define G8_WALK_FN static inline
#endif

struct G8WalkItem { int m0, n0, kt0, slice_and_tiles; };

G8_WALK_FN G8WalkItem g8_walk_item(int vb, int nitems, int tiles_m, int tiles_n, int ktiles, int splitk, int gm) {
  const int ntiles = tiles_m * tiles_n, q8 = nitems >> 3, r8 = nitems & 7;
  const int xcd = vb & 7, loc = vb >> 3;
  int id = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  const int ky = id / ntiles;
  id -= ky * ntiles;
  const int per_group = gm * tiles_n, g = id / per_group, r = id - g * per_group;
  const int rows = (tiles_m - g * gm) < gm ? (tiles_m - g * gm) : gm;
  const int tn = r / rows, tm = g * gm + (r - tn * rows);
  const int kt0 = (int)((unsigned)ky * (unsigned)ktiles / (unsigned)splitk);
  const int kt1 = (int)((unsigned)(ky + 1) * (unsigned)ktiles / (unsigned)splitk);
  G8WalkItem it;
  it.m0 = tm * 256;
  it.n0 = tn * 256;
  it.kt0 = kt0;
  it.slice_and_tiles = (ky << 16) | (kt1 - kt0);
  return it;
}
