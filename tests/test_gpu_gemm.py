"""MFMA GEMM + epilogues vs a float64 reference computed from the same 16-bit-rounded inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _F():
    from deeplearningexamples_amd import functional as F, _cabi as C
    return F, C


def _mk(shape, dtype, gen, scale=1.0):
    return (torch.randn(*shape, generator=gen) * scale).to(dtype)


CASES = [
    # m, n, k
    (128, 128, 64), (256, 384, 512), (100, 70, 40), (1, 1, 8), (300, 1, 256), (129, 257, 72),
    (512, 1024, 480), (64, 512, 13), (2048, 256, 1024), (33, 65, 1000),
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("mnk", CASES)
def test_gemm_three_layouts(cuda, mnk, dtype):
    F, C = _F()
    m, n, k = mnk
    gen = torch.Generator().manual_seed(m * 7 + n * 3 + k)
    a = _mk((m, k), dtype, gen)          # A[m][k]
    b = _mk((n, k), dtype, gen)          # B[n][k]
    ref = (a.double() @ b.double().T)
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    scale = np.sqrt(k)

    def chk(out, what):
        err = (out.cpu().double() - ref).abs().max().item()
        assert err <= tol * scale, "%s: max err %g (m,n,k=%s)" % (what, err, mnk)

    # forward layout: both k-contiguous, fp32 and 16-bit outputs
    chk(F.gemm(a.to(cuda), b.to(cuda), m, n, k, True, True, out_dtype=torch.float32), "kc/kc f32")
    chk(F.gemm(a.to(cuda), b.to(cuda), m, n, k, True, True), "kc/kc 16")
    # dgrad layout: B stored [k][n]
    bt = b.T.contiguous()
    chk(F.gemm(a.to(cuda), bt.to(cuda), m, n, k, True, False, out_dtype=torch.float32), "kc/nc")
    # wgrad layout: A stored [k][m], B stored [k][n]; also split-K
    at = a.T.contiguous()
    chk(F.gemm(at.to(cuda), bt.to(cuda), m, n, k, False, False, out_dtype=torch.float32), "mc/nc")
    chk(F.gemm(at.to(cuda), bt.to(cuda), m, n, k, False, False, out_dtype=torch.float32, splitk=3), "mc/nc splitk")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_gemm_epilogues(cuda, dtype):
    F, C = _F()
    m, n, k = 200, 136, 96
    gen = torch.Generator().manual_seed(5)
    a, b = _mk((m, k), dtype, gen, 0.5), _mk((n, k), dtype, gen, 0.5)
    bias = torch.randn(n, generator=gen)
    pre_ref = a.double() @ b.double().T + bias.double()
    tol = dict(rtol=2e-2, atol=2e-2) if dtype == torch.bfloat16 else dict(rtol=3e-3, atol=3e-3)
    y, pre = F.linear_fwd(a.to(cuda), b.to(cuda), bias.to(cuda), C.ACT_RELU, want_pre=True)
    np.testing.assert_allclose(pre.cpu().double().numpy(), pre_ref.numpy(), **tol)
    np.testing.assert_allclose(y.cpu().double().numpy(), pre_ref.clamp(min=0).numpy(), **tol)
    y, pre = F.linear_fwd(a.to(cuda), b.to(cuda), bias.to(cuda), C.ACT_GELU, want_pre=True)
    gelu = torch.nn.functional.gelu(pre_ref, approximate="tanh")
    np.testing.assert_allclose(y.cpu().double().numpy(), gelu.numpy(), **tol)
    # relu-backward masking in the dgrad epilogue
    gy = _mk((m, n), dtype, gen, 0.5)
    act_prev = _mk((m, k), dtype, gen)          # forward activation of the previous layer
    dx = F.linear_dgrad(gy.to(cuda), b.to(cuda), mask_src=act_prev.to(cuda))
    ref = (gy.double() @ b.double()) * (act_prev.double() > 0)
    np.testing.assert_allclose(dx.cpu().double().numpy(), ref.numpy(), **tol)
    # wgrad with accumulate and bias grad
    dw0 = torch.randn(n, k, generator=gen)
    dw = F.linear_wgrad(gy.to(cuda), a.to(cuda), out=dw0.clone().to(cuda), accumulate=True)
    np.testing.assert_allclose(dw.cpu().double().numpy(), (dw0.double() + gy.double().T @ a.double()).numpy(),
                               rtol=2e-2, atol=5e-2)
    db = F.colsum(gy.to(cuda))
    np.testing.assert_allclose(db.cpu().double().numpy(), gy.double().sum(0).numpy(), rtol=1e-3, atol=1e-2)


def test_gemm_strided_and_errors(cuda):
    F, C = _F()
    gen = torch.Generator().manual_seed(9)
    big = _mk((64, 300), torch.float16, gen).to(cuda)
    a = big[:, 10:10 + 128]                      # lda = 300, misaligned start -> scalar loader path
    b = _mk((32, 128), torch.float16, gen).to(cuda)
    out = F.gemm(a, b, 64, 32, 128, True, True, out_dtype=torch.float32)
    ref = a.cpu().double() @ b.cpu().double().T
    assert (out.cpu().double() - ref).abs().max() < 0.05
    with pytest.raises(ValueError):
        F.gemm(a.float(), b.float(), 64, 32, 128, True, True)
    with pytest.raises(ValueError):
        F.gemm(a, b, 64, 32, 128, False, True)


# Shapes the launcher routes to the 256x256 eight-wave tile (>= 160 tiles x slices, >= 4 K tiles per slice): full and
# ragged tiles, K tail, all three operand layouts, split-K slabs, and the fused epilogues.
BIG_CASES = [(4096, 2560, 256), (2824, 3848, 328), (1024, 1024, 4096)]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("mnk", BIG_CASES)
def test_gemm_big_tile(cuda, mnk, dtype):
    F, C = _F()
    m, n, k = mnk
    gen = torch.Generator().manual_seed(m + n + k)
    a = _mk((m, k), dtype, gen).to(cuda)
    b = _mk((n, k), dtype, gen).to(cuda)
    bias = torch.randn(n, generator=gen).to(cuda)
    ref = a.double() @ b.double().T                                    # fp64 on the GPU through torch (orientation only)
    tol = (2e-3 if dtype == torch.float16 else 1.6e-2) * np.sqrt(k)
    sk = 16 if k >= 4096 else 1                                         # 16 tiles x 16 slices for the K-heavy case

    def chk(out, what, r=ref):
        err = (out.double() - r).abs().max().item()
        assert err <= tol, "%s: max err %g (m,n,k=%s)" % (what, err, mnk)

    chk(F.gemm(a, b, m, n, k, True, True, out_dtype=torch.float32, splitk=sk), "kc/kc f32")
    if sk == 1:
        y = F.gemm(a, b, m, n, k, True, True, bias=bias, act=C.ACT_RELU)
        chk(y, "kc/kc 16 bias relu", (ref + bias.double()).clamp(min=0))
    bt = b.T.contiguous()
    chk(F.gemm(a, bt, m, n, k, True, False, out_dtype=torch.float32, splitk=sk), "kc/nc")
    at = a.T.contiguous()
    chk(F.gemm(at, bt, m, n, k, False, False, out_dtype=torch.float32, splitk=sk), "mc/nc")
    chk(F.gemm(at, bt, m, n, k, False, False, out_dtype=torch.float32, splitk=max(sk, 2)), "mc/nc splitk")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(200, 136, 96), (4096, 2560, 256)])
def test_gemm_masked_add_mul_and_gelu_derivative_epilogues(cuda, dtype, shape):
    """DLE_ACT_ADD_MASKED (residual-branch gradient dy * (y > 0) added under its bit mask, never materialised),
    DLE_ACT_MUL and DLE_ACT_GELU_DAUX (the forward GEMM leaves gelu'(v) behind) on the 128x128 and the 256x256 tile."""
    F, C = _F()
    m, n, k = shape
    gen = torch.Generator().manual_seed(17)
    a, b = _mk((m, k), dtype, gen, 0.5), _mk((n, k), dtype, gen, 0.5)
    tol = dict(rtol=2e-2, atol=3e-2) if dtype == torch.bfloat16 else dict(rtol=3e-3, atol=4e-3)
    prod = a.double() @ b.double().T
    add = _mk((m, n), dtype, gen)
    keep = torch.rand(m, n, generator=gen) > 0.4
    bits = (keep.reshape(-1, 8).to(torch.int32) << torch.arange(8, dtype=torch.int32)).sum(1).to(torch.uint8)
    y = F.gemm(a.to(cuda), b.to(cuda), m, n, k, True, True, act=C.ACT_ADD_MASKED, mask_src=add.to(cuda), aux=bits.to(cuda))
    ref = prod + torch.where(keep, add.double(), torch.zeros((), dtype=torch.float64))
    np.testing.assert_allclose(y.cpu().double().numpy(), ref.numpy(), **tol)
    y = F.gemm(a.to(cuda), b.to(cuda), m, n, k, True, True, act=C.ACT_MUL, mask_src=add.to(cuda))
    np.testing.assert_allclose(y.cpu().double().numpy(), (prod * add.double()).numpy(), **tol)
    bias = torch.randn(n, generator=gen)
    daux = torch.empty((m, n), dtype=dtype, device=cuda)
    y = F.gemm(a.to(cuda), b.to(cuda), m, n, k, True, True, bias=bias.to(cuda), act=C.ACT_GELU_DAUX, aux=daux)
    pre = (prod + bias.double()).requires_grad_(True)
    g = torch.nn.functional.gelu(pre, approximate="tanh")
    g.sum().backward()
    np.testing.assert_allclose(y.cpu().double().numpy(), g.detach().numpy(), **tol)
    np.testing.assert_allclose(daux.cpu().double().numpy(), pre.grad.numpy(), **tol)


WALK_CASES = [(131072, 256, 64), (70005, 192, 64), (66000, 128, 200)]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("mnk", WALK_CASES)
def test_gemm_persistent_tile_walk(cuda, mnk, dtype):
    """More 128x128 tiles than resident workgroups (2 per CU): every workgroup walks several tiles, the next tile's first
    K tile is prefetched under the current epilogue and waited for with a COUNTED vmcnt that leaves the epilogue's stores in
    flight.  Store-only epilogue (the PLAIN instantiation) and the general one, all three operand layouts, interior tiles,
    ragged M / N edges and a ragged K; every output element against fp64."""
    F, C = _F()
    m, n, k = mnk
    gen = torch.Generator(device=cuda).manual_seed(m + n + k)
    a = (torch.randn(m, k, generator=gen, device=cuda) * 0.5).to(dtype)
    b = (torch.randn(n, k, generator=gen, device=cuda) * 0.5).to(dtype)
    ref = a.double() @ b.double().T
    tol = (2e-3 if dtype == torch.float16 else 1.6e-2) * np.sqrt(k)

    def chk(out, what, r=ref):
        err = (out.double() - r).abs().max().item()
        assert err <= tol, "%s: max err %g (m,n,k=%s)" % (what, err, mnk)

    bt, at = b.T.contiguous(), a.T.contiguous()
    for rep in range(2):                                     # back to back: tiles of one launch still draining under the next
        chk(F.gemm(a, b, m, n, k, True, True), "kc/kc store-only")
        chk(F.gemm(a, bt, m, n, k, True, False), "kc/nc store-only")
        chk(F.gemm(at, bt, m, n, k, False, False), "mc/nc store-only")
    bias = torch.randn(n, generator=gen, device=cuda)
    chk(F.gemm(a, b, m, n, k, True, True, bias=bias, act=C.ACT_RELU), "bias relu", (ref + bias.double()).clamp(min=0))
    add = (torch.randn(m, n, generator=gen, device=cuda)).to(dtype)
    chk(F.gemm(a, bt, m, n, k, True, False, act=C.ACT_ADD, mask_src=add), "kc/nc + addend", ref + add.double())
    if (m * n) % 8 == 0:
        keep = torch.rand(m, n, generator=gen, device=cuda) > 0.4
        bits = (keep.reshape(-1, 8).to(torch.int32) << torch.arange(8, dtype=torch.int32, device=cuda)).sum(1).to(torch.uint8)
        y = F.gemm(a, bt, m, n, k, True, False, act=C.ACT_ADD_MASKED, mask_src=add, aux=bits)
        chk(y, "masked addend", ref + torch.where(keep, add.double(), torch.zeros((), dtype=torch.float64, device=cuda)))
    # the fp32-output path is not store-only (different output type): general epilogue on the same walk
    chk(F.gemm(a, b, m, n, k, True, True, out_dtype=torch.float32), "f32 out")


SMALLM = [
    # m, n, k: the per-step products of the recurrent loops (Tacotron2 gates / data gradients at batch 128, the encoder LSTM, the
    # query layer), heads with few rows, edge sizes of the 64 x {32, 16} tile and of the 128-element K chunk
    (128, 4096, 1536), (128, 4096, 2560), (128, 2560, 4096), (128, 1536, 4096), (128, 128, 1024), (128, 1024, 128),
    (128, 1024, 256), (3, 384, 128), (4, 4096, 1536), (256, 1000, 2048), (70, 40, 264), (200, 8, 1024), (1, 16, 8),
    (129, 36, 136),
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("mnk", SMALLM)
def test_gemm_few_rows_weight_streaming_kernel(cuda, mnk, dtype):
    """csrc/gemm_smallm.hip (M <= 256, both operands k-contiguous): plain / bias / 16-bit addend / fp32 accumulate epilogues,
    fp32 and 16-bit outputs, row-strided A and C -- against float64 on the same 16-bit-rounded inputs, and against the
    128x128-tile kernel it replaces (DLE_GEMM_SMALLM=0 is read once per process, so the comparison is with float64 only)."""
    F, C = _F()
    m, n, k = mnk
    gen = torch.Generator().manual_seed(m * 11 + n * 5 + k)
    wide_a = _mk((m, k + 24), dtype, gen)                     # A = a column slice of a wider matrix (lda > K)
    a = wide_a[:, 8:8 + k]
    b = _mk((n, k), dtype, gen)
    bias = torch.randn(n, generator=gen)
    src = _mk((m, n), dtype, gen)
    ref = a.double() @ b.double().T
    tol = (2e-3 if dtype == torch.float16 else 1.6e-2) * np.sqrt(k)
    ad, bd = wide_a.to(cuda)[:, 8:8 + k], b.to(cuda)

    def chk(out, r, what, extra=0.0):
        err = (out.cpu().double() - r).abs().max().item()
        assert err <= tol + extra, "%s: max err %g (m,n,k=%s)" % (what, err, mnk)
    chk(F.gemm(ad, bd, m, n, k, True, True, out_dtype=torch.float32), ref, "f32")
    chk(F.gemm(ad, bd, m, n, k, True, True, out_dtype=torch.float32, bias=bias.to(cuda), alpha=0.5), 0.5 * ref + bias.double(),
        "f32 + bias, alpha")
    r16 = ref + src.double()
    chk(F.gemm(ad, bd, m, n, k, True, True, act=C.ACT_ADD, mask_src=src.to(cuda)), r16, "16-bit + addend",
        extra=float(r16.abs().max()) * (1e-3 if dtype == torch.float16 else 8e-3))
    # fp32 accumulate into a row-strided C
    base = torch.randn(m, n + 12, generator=gen)
    cw = base.to(cuda)
    F.gemm(ad, bd, m, n, k, True, True, out=cw[:, 4:4 + n], accumulate=True)
    chk(cw[:, 4:4 + n], ref + base[:, 4:4 + n].double(), "accumulate, strided C")
    assert torch.equal(cw[:, :4].cpu(), base[:, :4]) and torch.equal(cw[:, 4 + n:].cpu(), base[:, 4 + n:])


EXPAND = [  # m, n, k, epilogue, weights k-contiguous: the envelope of csrc/gemm_expand.hip (m >= 4096, k in 64 / 128 / 256, n >= 2 k, n % 128 == 0)
    (4096, 128, 64, "masked", False), (4101, 256, 64, "masked", False), (8191, 256, 128, "add", False), (5000, 512, 256, "masked", False),
    (4097, 512, 128, "none", True), (6001, 1024, 256, "none", True), (4160, 256, 64, "add", True), (70001, 256, 64, "masked", True)]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", EXPAND)
def test_gemm_streaming_expand_kernel(cuda, case, dtype, monkeypatch):
    """The channel-widening 1x1-convolution shapes (many rows, K <= 256, N >= 2 K) go to gemm_expand.hip: ragged and ODD row counts
    (its stores pair neighbouring rows), both weight layouts, the three epilogues -- against float64 and, bit for bit, against the
    tile kernel (DLE_GEMM_EXPAND=0 pins it)."""
    F, C = _F()
    m, n, k, epi, b_kc = case
    gen = torch.Generator().manual_seed(m + n + k)
    a = _mk((m, k), dtype, gen, 0.5)
    w = _mk((n, k) if b_kc else (k, n), dtype, gen, 0.1)
    src = _mk((m, n), dtype, gen) if epi != "none" else None
    bits = torch.randint(0, 256, (m * n // 8,), generator=gen, dtype=torch.uint8) if epi == "masked" else None
    ref = a.double() @ (w.double().T if b_kc else w.double())
    if epi == "add":
        ref = ref + src.double()
    elif epi == "masked":
        keep = ((bits.to(torch.int32).unsqueeze(1) >> torch.arange(8, dtype=torch.int32)) & 1).reshape(m, n)
        ref = ref + src.double() * keep
    act = {"none": C.ACT_NONE, "add": C.ACT_ADD, "masked": C.ACT_ADD_MASKED}[epi]
    dev = lambda t: None if t is None else t.to(cuda)
    ad, wd, sd, bd = dev(a), dev(w), dev(src), dev(bits)
    outs = {}
    for pin in ("1", "0"):
        monkeypatch.setenv("DLE_GEMM_EXPAND", pin)
        guard = torch.full((m + 2, n), 7.0, dtype=dtype, device=cuda)             # rows past m must stay untouched
        F.gemm(ad, wd, m, n, k, True, b_kc, out=guard[:m], act=act, mask_src=sd, aux=bd)
        assert bool((guard[m:] == 7.0).all())
        outs[pin] = guard[:m].clone()
    err = (outs["1"].cpu().double() - ref).abs().max().item()
    assert err <= (2e-3 if dtype == torch.float16 else 1.6e-2) * max(1.0, float(ref.abs().max())), err
    if not (epi == "none" and k < 128):                   # (store-only K = 64 products stay on the tile kernel either way)
        assert torch.equal(outs["1"], outs["0"]), "expand and tile kernels round differently"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 56, 64, 256), (3, 28, 128, 512), (9, 23, 256, 1024), (5, 29, 64, 128)])
def test_conv1x1_statistics_from_the_expand_kernel(cuda, shape, dtype, monkeypatch):
    """dle_conv2d_fwd_colstats on channel-widening 1x1 convolutions: output identical to the tile kernel's, BatchNorm statistics
    of the STORED output (one partial row per workgroup group of gemm_expand.hip, folded by dle_bn_stats_from_partials)."""
    F, C = _F()
    nb, hw, c, ko = shape
    gen = torch.Generator().manual_seed(nb + hw + c)
    x = _mk((nb, hw, hw, c), dtype, gen, 0.5).to(cuda)
    w = _mk((ko, 1, 1, c), dtype, gen, 0.1).to(cuda)
    res = {}
    for pin in ("1", "0"):
        monkeypatch.setenv("DLE_GEMM_EXPAND", pin)
        rm, rv = torch.zeros(ko, device=cuda), torch.ones(ko, device=cuda)
        y, mean, rstd = F.conv2d_fwd_bnstats(x, w, 1, 0, rm, rv)
        res[pin] = (y.clone(), mean.clone(), rstd.clone(), rm, rv)
    y1, mean1, rstd1, rm1, rv1 = res["1"]
    assert torch.equal(y1, res["0"][0])
    yf = y1.double().view(-1, ko)
    ref_mean, ref_var = yf.mean(0), yf.var(0, unbiased=False)
    assert float((mean1.double() - ref_mean).abs().max()) <= 1e-5 * float(ref_mean.abs().max()) + 1e-6
    assert float((rstd1.double() - (ref_var + 1e-5).rsqrt()).abs().max()) <= 1e-4 * float((ref_var + 1e-5).rsqrt().max())
    for a, b in zip(res["1"][1:], res["0"][1:]):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-7


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 1000, 512), (40, 10000, 512), (5, 77, 24), (1, 4096, 1024)])
def test_colsum_batched(cuda, shape, dtype):
    """dle_colsum_batched: n same-shaped matrices -> n fp32 destinations (views of one flat buffer, as the gradient slots are),
    against float64 sums; the same matrix may appear several times (WaveGlow's d_skip feeds every layer of a flow)."""
    F, C = _F()
    n, m, cols = shape
    gen = torch.Generator().manual_seed(n + m + cols)
    xs = [_mk((m, cols), dtype, gen).to(cuda) for _ in range(n)]
    flat = torch.full(((n + 2) * cols + 3,), 7.0, dtype=torch.float32, device=cuda)
    entries = [(xs[i], flat[1 + i * cols:1 + (i + 1) * cols] if cols % 4 == 0 else flat[i * cols:(i + 1) * cols]) for i in range(n)]
    entries.append((xs[0], flat[1 + n * cols:1 + (n + 1) * cols] if cols % 4 == 0 else flat[n * cols:(n + 1) * cols]))
    F.colsum_batched(F.ColsumTable(entries), m, cols, cols, dtype)
    for x, out in entries:
        ref = x.double().sum(0)
        assert float((out.double() - ref).abs().max()) <= 2e-6 * float(x.double().abs().sum(0).max()) + 1e-6
    assert float(flat[-1]) == 7.0
    with pytest.raises(ValueError):
        F.ColsumTable([(xs[0], flat[:cols]), (xs[0][:-1], flat[:cols])])
