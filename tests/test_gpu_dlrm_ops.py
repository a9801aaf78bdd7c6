"""HIP kernels (through the C ABI) vs the CPU oracle + the reference-generated fixtures. GPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import dlrm_oracle as O

pytestmark = pytest.mark.gpu


def _F():
    from deeplearningexamples_amd import functional as F
    return F


TOL = {torch.float16: dict(rtol=1e-3, atol=1e-3), torch.bfloat16: dict(rtol=8e-3, atol=8e-3),
       torch.float32: dict(rtol=1e-4, atol=1e-4)}
NP16 = {torch.float16: np.float16}


def _round(x, dtype):
    return torch.from_numpy(x).to(dtype)


# (batch, rows, cols): the reference's test grid (dot_based_interact_ops_test.py:89-112) + Criteo shape
SHAPES = [(16, 32, 32), (17, 31, 37), (15, 31, 37), (16, 31, 33), (16, 32, 31), (8, 27, 128), (5, 27, 128),
          (3, 2, 8), (4, 27, 16), (1, 1, 16), (2048, 27, 128),
          (4099, 27, 128), (9001, 20, 64)]      # >= 4096 samples: the persistent forward walk (ragged last round of wavefronts)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("force_generic", [False, True])
def test_dot_interact_fwd_bwd(cuda, shape, dtype, force_generic):
    F = _F()
    b, r, c = shape
    g = torch.Generator().manual_seed(b * 1000 + r * 10 + c)
    x = torch.rand(b, r, c, generator=g).to(dtype)
    ow = O.interact_out_width(r, c)
    assert F.dot_interact_out_width(r, c) == ow
    ug = torch.rand(b, ow, generator=g).to(dtype)
    y = F.dot_interact_fwd(x.to(cuda), force_generic).cpu()
    ref = O.dot_interact_fwd(x.float().numpy(), np.float32)
    np.testing.assert_allclose(y.float().numpy(), _round(ref, dtype).float().numpy(), **TOL[dtype])
    # layout facts that must hold exactly
    assert torch.equal(y[:, :c], x[:, 0, :])
    assert (y[:, c + r * (r - 1) // 2:] == 0).all()
    gx, gm = F.dot_interact_bwd(x.to(cuda), ug.to(cuda), force_generic)
    rgx, rgm = O.dot_interact_bwd(x.float().numpy(), ug.float().numpy(), np.float32)
    tol = dict(TOL[dtype])
    tol["atol"] = tol["atol"] * max(1.0, r / 4)       # sums of r products of O(1) values
    np.testing.assert_allclose(gx.cpu().float().numpy(), _round(rgx, dtype).float().numpy(), **tol)
    assert torch.equal(gm.cpu(), ug[:, :c])


def test_dot_interact_golden_fp32(cuda, golden_dir):
    F = _F()
    z = np.load(os.path.join(golden_dir, "dlrm_dot_interact.npz"))
    for k in sorted({k.rsplit("_", 1)[0] for k in z.files if k.endswith("_x")}):
        x = torch.from_numpy(z[k + "_x"]).to(cuda)
        y = F.dot_interact_fwd(x)
        np.testing.assert_allclose(y.cpu().numpy(), z[k + "_y"], rtol=1e-5, atol=1e-5)
        gx, gm = F.dot_interact_bwd(x, torch.from_numpy(z[k + "_ug"]).to(cuda))
        tot = gx.cpu().numpy()
        tot[:, 0, :] += gm.cpu().numpy()
        np.testing.assert_allclose(tot, z[k + "_gx_total"], rtol=1e-4, atol=1e-4)


def test_dot_interact_empty_and_errors(cuda):
    F = _F()
    y = F.dot_interact_fwd(torch.empty(0, 27, 128, dtype=torch.float16, device=cuda))
    assert y.shape == (0, 480)
    with pytest.raises(ValueError):
        F.dot_interact_fwd(torch.empty(4, 128, dtype=torch.float16, device=cuda))
    with pytest.raises(ValueError):
        F.dot_interact_bwd(torch.zeros(4, 27, 128, dtype=torch.float16, device=cuda),
                           torch.zeros(4, 479, dtype=torch.float16, device=cuda))
    with pytest.raises(RuntimeError):
        F.dot_interact_fwd(torch.zeros(4, 27, 128, dtype=torch.float16))     # CPU tensor: no fallback


def test_embedding_golden_bit_exact(cuda, golden_dir):
    F = _F()
    z = np.load(os.path.join(golden_dir, "dlrm_embedding.npz"))
    w = torch.from_numpy(z["w0"]).to(cuda)
    idx = torch.from_numpy(z["idx_in"]).to(cuda)
    off = torch.from_numpy(z["offsets"]).to(cuda)
    sizes = torch.from_numpy(z["sizes"]).to(cuda)
    rows = F.emb_offset_indices(idx, off, sizes)
    exp_rows = O.offset_indices(z["idx_hashed"], z["offsets"])
    assert np.array_equal(rows.cpu().numpy(), exp_rows)                      # int64, bit exact
    out = F.emb_gather_fwd(w, idx, off, sizes)
    assert np.array_equal(out.cpu().numpy(), z["out"])                        # fp32 copy, bit exact
    out2 = F.emb_gather_fwd(w, rows)                                          # joint table, pre-offset indices
    assert np.array_equal(out2.cpu().numpy(), z["out"])
    out16 = F.emb_gather_fwd(w, rows, out_dtype=torch.float16)
    assert torch.equal(out16.cpu(), torch.from_numpy(z["out"]).half())
    ug = torch.from_numpy(z["ug"]).to(cuda)
    vals = F.emb_grad_values(ug)
    assert torch.equal(vals, ug)
    F.emb_sparse_sgd_(w, rows, ug, float(z["lr"]))
    np.testing.assert_allclose(w.cpu().numpy(), z["w1"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dim", [128, 16, 4, 64])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_embedding_gather_random(cuda, dim, out_dtype):
    F = _F()
    sizes = [7, 1, 1000, 33, 5000, 2][: 6 if dim != 128 else 5]
    rng = np.random.default_rng(dim)
    off = O.table_offsets(sizes)
    w = rng.standard_normal((int(off[-1]), dim)).astype(np.float32)
    b = 777
    idx = np.stack([rng.integers(0, s, b) for s in sizes], axis=1).astype(np.int64)
    out = F.emb_gather_fwd(torch.from_numpy(w).to(cuda), torch.from_numpy(idx).to(cuda),
                           torch.from_numpy(off).to(cuda), None, out_dtype)
    exp = torch.from_numpy(O.embedding_gather(w, O.offset_indices(idx, off))).to(out_dtype)
    assert torch.equal(out.cpu(), exp)


def test_embedding_sgd_duplicates_scale_skip(cuda):
    F = _F()
    rng = np.random.default_rng(3)
    w0 = rng.standard_normal((50, 128)).astype(np.float32)
    rows = rng.integers(0, 50, (4096,)).astype(np.int64)          # heavy duplication -> atomics
    g = (rng.standard_normal((4096, 128)) * 0.01).astype(np.float16)
    w = torch.from_numpy(w0.copy()).to(cuda)
    inv = torch.tensor([0.5], device=cuda)
    lr = torch.tensor(2.0, device=cuda)
    F.emb_sparse_sgd_(w, torch.from_numpy(rows).to(cuda), torch.from_numpy(g).to(cuda), lr, scale=inv)
    exp = O.sparse_sgd(w0, rows, g.astype(np.float32) * 0.5, 2.0)
    np.testing.assert_allclose(w.cpu().numpy(), exp, rtol=1e-4, atol=1e-4)
    before = w.clone()
    F.emb_sparse_sgd_(w, torch.from_numpy(rows).to(cuda), torch.from_numpy(g).to(cuda), lr,
                      skip_flag=torch.ones(1, device=cuda))
    assert torch.equal(w, before)


def test_embedding_full_size_roundtrip(cuda):
    """BASELINE-size property test: gather(W, idx) after W[idx] -= lr*g on unique rows changes exactly
    those rows by exactly lr*g (size-independent property, 64k x 26 x 128)."""
    F = _F()
    torch.manual_seed(0)
    rows_total, b, t, d = 2_000_000, 65536, 26, 128
    w = torch.randn(rows_total, d, device=cuda)
    perm = torch.randperm(rows_total, device=cuda)[: b * t].view(b, t)       # unique rows
    before = F.emb_gather_fwd(w, perm)
    g = torch.randn(b, t, d, device=cuda).half()
    F.emb_sparse_sgd_(w, perm, g, 0.25)
    after = F.emb_gather_fwd(w, perm)
    assert torch.equal(after, before - 0.25 * g.float())


def test_cuda_ext_autograd_functions_match_torch(cuda):
    """The reference's dlrm.cuda_ext boundary (autograd Functions) on the HIP kernels vs plain torch ops."""
    from deeplearningexamples_amd.dlrm import cuda_ext as X
    g = torch.Generator().manual_seed(21)
    # dot interaction through autograd: both returned gradients land on the same leaf
    x = torch.rand(6, 27, 128, generator=g).half().to(cuda).requires_grad_()
    y = X.dotBasedInteract(x, x[:, 0, :])
    ug = torch.rand(y.shape, generator=g).half().to(cuda)
    y.backward(ug)
    xr = x.detach().float().cpu().requires_grad_()
    z = torch.bmm(xr, xr.transpose(1, 2))
    ri, ci = [torch.from_numpy(a) for a in O.tril_pairs(27)]
    yr = torch.cat([xr[:, 0, :], z[:, ri, ci], torch.zeros(6, 1)], dim=1)
    yr.backward(ug.float().cpu())
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), yr.detach().numpy(), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(x.grad.float().cpu().numpy(), xr.grad.numpy(), rtol=1e-2, atol=3e-2)
    # fused gather: sparse COO weight gradient like the reference's
    sizes = [7, 50, 3]
    off = torch.tensor([0] + sizes).cumsum(0).to(cuda)
    w = torch.randn(60, 16, generator=g).to(cuda).requires_grad_()
    idx = torch.stack([torch.randint(0, s, (9,), generator=g) for s in sizes], 1).to(cuda)
    out = X.buckle_embedding_fused_gather(w, idx, off, False)
    ug = torch.randn(out.shape, generator=g).to(cuda)
    out.backward(ug)
    assert w.grad.is_sparse
    wr = w.detach().cpu().requires_grad_()
    outr = wr[(idx.cpu() + off.cpu()[:-1])]
    outr.backward(ug.cpu())
    assert torch.equal(out.detach().cpu(), outr.detach())
    np.testing.assert_allclose(w.grad.to_dense().cpu().numpy(), wr.grad.numpy(), rtol=1e-6, atol=1e-6)
    emb = X.JointSparseEmbedding(sizes, 16, device=cuda)
    o2 = emb(idx)
    o2.sum().backward()
    assert emb.weights.grad.is_sparse and o2.shape == (9, 3, 16)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("force_generic", [False, True])
def test_dot_interact_bwd_reports_nonfinite(cuda, dtype, force_generic):
    """dle_dot_interact_bwd_checked: the flag GradScaler.unscale_ would raise on this gradient, set by the kernel that writes
    it (MFMA path) or by the plain sweep (generic path); same gradient bits as the unchecked entry point; untouched when finite."""
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(3)
    b, r, c = 37, 27, 128
    x = (torch.randn(b, r, c, generator=g) * 0.5).to(dtype).to(cuda)
    up = (torch.randn(b, F.dot_interact_out_width(r, c), generator=g) * 0.1).to(dtype).to(cuda)
    flag = torch.zeros(1, device=cuda)
    g0, _ = F.dot_interact_bwd(x, up, force_generic=force_generic, fuse_mlp_grad=True)
    g1, _ = F.dot_interact_bwd(x, up, force_generic=force_generic, fuse_mlp_grad=True, found_inf=flag)
    assert torch.equal(g0, g1) and flag.item() == 0.0
    flag.fill_(0.25)                                            # left untouched (not cleared) when everything is finite
    F.dot_interact_bwd(x, up, force_generic=force_generic, fuse_mlp_grad=True, found_inf=flag)
    assert flag.item() == 0.25
    for bad in (float("inf"), float("nan")):
        up2 = up.clone()
        up2[b - 1, c + 5] = bad                                 # one pairwise-product gradient of the LAST sample
        flag.zero_()
        g2, _ = F.dot_interact_bwd(x, up2, force_generic=force_generic, fuse_mlp_grad=True, found_inf=flag)
        assert flag.item() == 1.0 and not bool(torch.isfinite(g2.float()).all())
    if dtype == torch.float16:                                  # overflow of the 16-bit conversion alone
        flag.zero_()
        F.dot_interact_bwd(x * 200, up * 200, force_generic=force_generic, fuse_mlp_grad=True, found_inf=flag)
        big, _ = F.dot_interact_bwd(x * 200, up * 200, force_generic=force_generic, fuse_mlp_grad=True)
        assert flag.item() == (0.0 if bool(torch.isfinite(big.float()).all()) else 1.0)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_a2a_blocks_pack_and_unpack(cuda, dtype):
    """The bottom -> top exchange buffer <-> interaction input (dlrm/model/distributed.py:32-98: torch.cat(dim=1) of the received
    blocks forward, the split of the gradient backward) as ONE launch per direction, bit exact."""
    from deeplearningexamples_amd import functional as F
    g = torch.Generator().manual_seed(3)
    rows, d = 37, 128
    for vectors in ([1, 4, 4, 4, 4, 4, 4, 2], [0, 3, 5], [7]):
        widths = [v * d for v in vectors]
        blocks = [torch.randn(rows, w, generator=g).to(dtype) for w in widths]
        flat = torch.cat([b.reshape(-1) for b in blocks]).to(cuda)
        x = torch.empty(rows, sum(widths), dtype=dtype, device=cuda)
        F.a2a_blocks(flat, x, rows, widths, pack=False)
        assert torch.equal(x.cpu(), torch.cat(blocks, dim=1))
        back = torch.zeros_like(flat)
        F.a2a_blocks(back, x, rows, widths, pack=True)
        assert torch.equal(back, flat)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("batch", [64, 1000, 4099, 65536])
def test_tiny_tables_onehot_mfma_vs_oracle(cuda, batch, dtype):
    """Tables of <= 128 rows through the one-hot MFMA segment sum (csrc/emb_onehot.hip), tables of <= 4096 rows through eight lists
    per row + the fold pass, vs the float64 oracle: every lookup of the batch lands on a handful of rows (gather_gpu_fused.cu:161-202 semantics), ragged batch tails, scale, skip flag, untouched rows
    keep their bits, two runs are bit-identical (fixed fold order)."""
    F = _F()
    rng = np.random.default_rng(batch + (1 if dtype == torch.float16 else 2))
    sizes = [4, 128, 1, 97, 300, 11, 63, 5000, 104, 35, 2209]  # 300 / 2209 rows: eight lists per row; 5000: one list per row
    dim = 128
    off = O.table_offsets(sizes)
    w = rng.standard_normal((int(off[-1]), dim)).astype(np.float32)
    idx = np.stack([rng.integers(0, max(s - 1, 1), batch) for s in sizes], 1).astype(np.int64)   # the LAST row of a table is never hit
    rows = O.offset_indices(idx, off)
    g16 = torch.from_numpy(rng.standard_normal((batch, len(sizes) + 1, dim)).astype(np.float32) * 0.05).to(dtype)
    gd = g16.to(cuda)
    gf = g16[:, 1:, :].to(torch.float32).numpy()
    rd = torch.from_numpy(rows).to(cuda)
    inv = torch.tensor([0.25], device=cuda)
    lr = 0.5
    exp = O.sparse_sgd(w, rows, gf * 0.25, lr)

    def run():
        wd = torch.from_numpy(w).to(cuda)
        ws = F.EmbUpdateWorkspace(off, dim, cuda)
        assert ws.n_onehot == 8
        F.emb_sgd_dedup_(wd, rd, gd[:, 1:, :], ws, lr, scale=inv, grad_batch_stride=(len(sizes) + 1) * dim)
        assert int((ws.head != -1).sum().item()) == 0
        return wd
    a = run()
    # fp32 sums of up to `batch` 16-bit values per row: error ~ sqrt(n) * 2^-24 * sum|g|
    mag = np.zeros_like(w, dtype=np.float64)
    np.add.at(mag, rows.reshape(-1), np.abs(gf.reshape(-1, dim)) * 0.25)
    err = np.abs(a.cpu().numpy().astype(np.float64) - exp)
    assert np.all(err <= 2e-6 * lr * mag + 1e-6 * np.abs(exp) + 1e-7), float((err - 2e-6 * lr * mag).max())
    untouched = mag.sum(1) == 0
    assert untouched.any() and np.array_equal(a.cpu().numpy()[untouched], w[untouched])
    # bit-reproducible on the one-hot tables (fixed fold order; the list path's order of duplicates follows the atomics)
    b2 = run()
    for ti, sz in enumerate(sizes):
        if sz <= 128:
            assert torch.equal(a[int(off[ti]):int(off[ti + 1])], b2[int(off[ti]):int(off[ti + 1])])
    # skip flag: nothing moves
    wd = torch.from_numpy(w).to(cuda)
    ws = F.EmbUpdateWorkspace(off, dim, cuda)
    F.emb_sgd_dedup_(wd, rd, gd[:, 1:, :], ws, lr, scale=inv, skip_flag=torch.ones(1, device=cuda),
                     grad_batch_stride=(len(sizes) + 1) * dim)
    assert torch.equal(wd.cpu(), torch.from_numpy(w))
