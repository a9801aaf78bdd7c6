"""The persistent ping-pong GEMM (csrc/gemm8_kernel.h) behind dle_gemm / dle_gemm_colsum: every layout and epilogue class of the
linear layers it serves (forward X W^T + bias / ReLU / tanh-GELU (+ pre-activation or derivative side output): BERT/modeling.py:
130-160,340-384, dlrm/nn/mlps.py:38-43; data gradient dY W with ReLU mask / addend / stored derivative (+ column sums); weight
gradient dY^T X with split-K slabs, fp32 accumulate) on small, RAGGED shapes (edge tiles, K tails, K slices of unequal length),
against
  * the tile kernels of gemm_dma.hip (dle_gemm8_mode(0)): BIT-IDENTICAL outputs and side outputs, the contract the loss-level
    parity bars rest on (column sums: the same rounded values folded in another fp32 order, 4e-6 of the column's magnitude),
  * an fp32 torch product with the epilogue applied in fp32: within the rounding of the 16-bit output (tolerances below),
  * itself (run-to-run: a race in the LDS stream or the epilogue exchange shows up as a difference).
dle_gemm8_min_items(1) routes the small shapes to the kernel; dle_gemm8_launch_count() proves they were not declined.  GPU only."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def g8(cuda):
    from deeplearningexamples_amd import _cabi as C
    lib = C.lib()
    prev_items, prev_mode = lib.dle_gemm8_min_items(1), lib.dle_gemm8_mode(1)
    yield lib
    lib.dle_gemm8_min_items(prev_items)
    lib.dle_gemm8_mode(prev_mode)


def _operands(layout, m, n, k, dtype, dev, gen, scale=1.0):
    if layout == "nt":                     # forward: A [m, k], B [n, k]
        a = (torch.randn(m, k, generator=gen) * scale).to(dtype).to(dev)
        b = (torch.randn(n, k, generator=gen) * scale).to(dtype).to(dev)
        return a, b, True, True, a.float() @ b.float().t()
    if layout == "nn":                     # data gradient: A [m, k], B [k, n]
        a = (torch.randn(m, k, generator=gen) * scale).to(dtype).to(dev)
        b = (torch.randn(k, n, generator=gen) * scale).to(dtype).to(dev)
        return a, b, True, False, a.float() @ b.float()
    a = (torch.randn(k, m, generator=gen) * scale).to(dtype).to(dev)      # "tn" weight gradient: A [k, m], B [k, n]
    b = (torch.randn(k, n, generator=gen) * scale).to(dtype).to(dev)
    return a, b, False, False, a.float().t() @ b.float()


def _gelu(x):
    return torch.nn.functional.gelu(x, approximate="tanh")


def _gelu_d(x):
    k0, k1 = 0.7978845608028654, 0.044715
    th = torch.tanh(k0 * (x + k1 * x ** 3))
    return 0.5 * (1 + th) + 0.5 * x * (1 - th * th) * k0 * (1 + 3 * k1 * x * x)


CASES = [
    # name, layout, (m, n, k), dtype, kwargs
    ("plain", "nt", (1000, 520, 256), torch.bfloat16, {}),
    ("bias_gelu_side", "nt", (777, 776, 320), torch.bfloat16, dict(bias=True, act="gelu", aux=True)),
    ("bias_gelu_derivative_side", "nt", (520, 512, 192), torch.bfloat16, dict(bias=True, act="gelu_daux", aux=True)),
    ("bias_relu_f16_ktail", "nt", (1000, 1024, 480), torch.float16, dict(bias=True, act="relu")),
    ("bias_tanh", "nt", (512, 264, 128), torch.bfloat16, dict(bias=True, act="tanh")),
    ("bias_fp32_out", "nt", (515, 1032, 192), torch.bfloat16, dict(bias=True, out_f32=True)),
    ("addend", "nn", (900, 520, 384), torch.bfloat16, dict(act="add", src=True)),
    ("addend_ktail", "nn", (777, 520, 1000), torch.bfloat16, dict(act="add", src=True)),
    ("relu_mask_f16", "nn", (1031, 264, 128), torch.float16, dict(act="relu_bwd", src=True)),
    # the conv1 data gradient of a bottleneck: + the residual-branch gradient under the block's keep bits (1 bit per element)
    ("addend_under_keep_bits", "nn", (900, 528, 384), torch.bfloat16, dict(act="add_masked", src=True, bits=True)),
    ("addend_under_keep_bits_f16", "nn", (1024, 768, 512), torch.float16, dict(act="add_masked", src=True, bits=True)),
    ("stored_derivative_colsum", "nn", (1024, 512, 256), torch.bfloat16, dict(act="mul", src=True, colsum=True)),
    ("relu_mask_colsum_f16", "nn", (768, 1024, 512), torch.float16, dict(act="relu_bwd", src=True, colsum=True)),
    ("wgrad_splitk3", "tn", (520, 776, 1536), torch.bfloat16, dict(splitk=3)),
    ("wgrad_splitk2_accumulate", "tn", (264, 1000, 2048), torch.bfloat16, dict(splitk=2, accumulate=True)),
    ("wgrad_fp32_accumulate", "tn", (512, 768, 640), torch.bfloat16, dict(out_f32=True, accumulate=True)),
    ("wgrad_16bit_out_f16", "tn", (512, 512, 1024), torch.float16, {}),
    # K slices of ONE K tile (the first K tile of an item is also its last: stream hand-over between back-to-back epilogues)
    ("wgrad_single_tile_slices", "tn", (768, 512, 256), torch.bfloat16, dict(splitk=4)),
    ("wgrad_one_and_two_tile_slices", "tn", (512, 768, 192), torch.bfloat16, dict(splitk=2)),
    ("two_k_tiles_bias_relu", "nt", (2048, 1280, 128), torch.bfloat16, dict(bias=True, act="relu")),
]


@pytest.mark.parametrize("name,layout,mnk,dtype,kw", CASES, ids=[c[0] for c in CASES])
def test_against_tile_kernels_fp32_and_itself(g8, cuda, name, layout, mnk, dtype, kw):
    from deeplearningexamples_amd import functional as F
    from deeplearningexamples_amd import _cabi as C
    m, n, k = mnk
    gen = torch.Generator().manual_seed(1234 + m + 7 * n + 13 * k)
    a, b, a_kc, b_kc, ref = _operands(layout, m, n, k, dtype, cuda, gen, scale=0.5)
    bias = torch.randn(n, generator=gen).to(cuda) if kw.get("bias") else None
    src = torch.randn(m, n, generator=gen).to(dtype).to(cuda) if kw.get("src") else None
    splitk, accumulate, colsum = kw.get("splitk", 1), kw.get("accumulate", False), kw.get("colsum", False)
    odt = torch.float32 if (kw.get("out_f32") or splitk > 1) else dtype
    init = torch.randn(m, n, generator=gen).to(cuda) if accumulate else None
    act = {None: C.ACT_NONE, "relu": C.ACT_RELU, "gelu": C.ACT_GELU, "gelu_daux": C.ACT_GELU_DAUX, "tanh": C.ACT_TANH,
           "add": C.ACT_ADD, "relu_bwd": C.ACT_RELU_BWD, "mul": C.ACT_MUL, "add_masked": C.ACT_ADD_MASKED}[kw.get("act")]
    bits = torch.randint(0, 256, (m * n // 8,), generator=gen, dtype=torch.uint8).to(cuda) if kw.get("bits") else None

    def run(mode):
        g8.dle_gemm8_mode(mode)
        out = init.clone() if accumulate else torch.empty(m, n, dtype=odt, device=cuda)
        aux = torch.empty(m, n, dtype=dtype, device=cuda) if kw.get("aux") else bits
        cs = torch.zeros(n, dtype=torch.float32, device=cuda) if colsum else None
        if colsum:
            out = F.gemm_colsum(a, b, m, n, k, src, cs, act=act)
            assert out is not None
        else:
            F.gemm(a, b, m, n, k, a_kc, b_kc, out=out, bias=bias, act=act, aux=aux, mask_src=src, splitk=splitk,
                   accumulate=accumulate)
        torch.cuda.synchronize()
        return out, aux, cs

    o_tile, x_tile, c_tile = run(0)
    before = g8.dle_gemm8_launch_count()
    o_new, x_new, c_new = run(1)
    assert g8.dle_gemm8_launch_count() > before, "the ping-pong kernel declined the shape: nothing was tested"
    # bit-identical to the tile kernels
    assert torch.equal(o_new, o_tile)
    if x_new is not None and bits is None:
        assert torch.equal(x_new, x_tile)
    if c_new is not None:          # (column sums: the two kernels fold the same rounded values in different fp32 orders)
        mag_t = o_tile.to(torch.float64).abs().sum(0)
        assert torch.all((c_new.to(torch.float64) - c_tile.to(torch.float64)).abs() <= 4e-6 * mag_t + 1e-9)
    # fp32 reference of the product + epilogue
    pre = ref + bias if bias is not None else ref
    want = pre
    if kw.get("act") == "relu":
        want = torch.relu(pre)
    elif kw.get("act") in ("gelu", "gelu_daux"):
        want = _gelu(pre)
    elif kw.get("act") == "tanh":
        want = torch.tanh(pre)
    elif kw.get("act") == "add":
        want = pre + src.float()
    elif kw.get("act") == "add_masked":
        keep = ((bits.view(-1, 1) >> torch.arange(8, device=cuda, dtype=torch.uint8)) & 1).view(m, n).float()
        want = pre + src.float() * keep
    elif kw.get("act") == "relu_bwd":
        want = pre * (src.float() > 0)
    elif kw.get("act") == "mul":
        want = pre * src.float()
    if accumulate:
        want = want + init
    # 16-bit output: half an ulp of the largest magnitude (bf16 2^-9, fp16 2^-12) + the fp32 accumulation order; fp32 output: 1e-5
    tol = 1e-5 if odt == torch.float32 else (4e-3 if dtype == torch.bfloat16 else 6e-4)
    scale = float(want.abs().max()) + 1e-30
    assert float((o_new.float() - want).abs().max()) <= tol * scale
    if x_new is not None and bits is None:
        side = _gelu_d(pre) if kw.get("act") == "gelu_daux" else pre
        assert float((x_new.float() - side).abs().max()) <= tol * (float(side.abs().max()) + 1e-30)
    if c_new is not None:
        exact = o_new.to(torch.float64).sum(0)
        mag = o_new.to(torch.float64).abs().sum(0)
        assert torch.all((c_new.to(torch.float64) - exact).abs() <= 2e-6 * mag + 1e-9)
    # run-to-run
    for _ in range(3):
        o2, x2, c2 = run(1)
        assert torch.equal(o2, o_new)
        assert x2 is None or bits is not None or torch.equal(x2, x_new)
        assert c2 is None or torch.equal(c2, c_new)


def test_slab_rows_of_both_half_blocks(g8, cuda):
    """fp32 slabs leave through a 16-row exchange per 32-row block: every row of the second half-block must be its own (a
    compiler-forwarded load once returned rows 0-15 for the lanes that had not written)."""
    from deeplearningexamples_amd import functional as F
    m, n, k = 512, 512, 1024
    gen = torch.Generator().manual_seed(5)
    a = torch.randn(k, m, generator=gen).bfloat16().to(cuda)
    b = torch.randn(k, n, generator=gen).bfloat16().to(cuda)
    out = torch.empty(m, n, dtype=torch.float32, device=cuda)
    before = g8.dle_gemm8_launch_count()
    F.gemm(a, b, m, n, k, False, False, out=out, splitk=2)
    torch.cuda.synchronize()
    assert g8.dle_gemm8_launch_count() > before
    ref = a.float().t() @ b.float()
    err = (out - ref).abs().amax(dim=1)                    # per row
    assert float(err.max()) <= 1e-4 * float(ref.abs().max())


def test_many_items_per_workgroup_and_odd_grids(g8, cuda):
    """More items than CUs (the persistent walk, the item switch inside the stream) with item counts that do not divide by the
    workgroup count or by 8 (XCD chunks of unequal length)."""
    from deeplearningexamples_amd import functional as F
    for (m, n, k) in [(256 * 23, 256 * 13, 192), (256 * 31 + 40, 256 * 9 + 8, 128)]:
        gen = torch.Generator().manual_seed(m + n)
        a = (torch.randn(m, k, generator=gen) * 0.5).bfloat16().to(cuda)
        b = (torch.randn(n, k, generator=gen) * 0.5).bfloat16().to(cuda)
        outs = []
        for mode in (0, 1):
            g8.dle_gemm8_mode(mode)
            o = torch.empty(m, n, dtype=torch.bfloat16, device=cuda)
            before = g8.dle_gemm8_launch_count()
            F.gemm(a, b, m, n, k, True, True, out=o)
            torch.cuda.synchronize()
            assert (g8.dle_gemm8_launch_count() > before) == (mode == 1)
            outs.append(o)
        assert torch.equal(outs[0], outs[1])
        ref = a.float() @ b.float().t()
        assert float((outs[1].float() - ref).abs().max()) <= 4e-3 * float(ref.abs().max())


def test_switches(g8):
    assert g8.dle_gemm8_min_items(-1) == 1 and g8.dle_gemm8_mode(-1) == 1
    assert g8.dle_gemm8_min_items(64) == 1 and g8.dle_gemm8_min_items(1) == 64
    assert g8.dle_gemm8_mode(0) == 1 and g8.dle_gemm8_mode(1) == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nhw,c,ko", [((4, 16, 16), 256, 1024), ((8, 8, 8), 1024, 256), ((5, 16, 16), 512, 520)])
def test_conv1x1_forward_with_batchnorm_statistics(g8, cuda, dtype, nhw, c, ko):
    """dle_conv2d_fwd_colstats on the ping-pong kernel (gemm8_kernel.h EPI 3: the column sums / sums of squares of the rounded
    output leave from the register epilogue) vs the same entry point on the older kernels (dle_gemm8_mode(0)): the convolution
    output is BIT-IDENTICAL, the batch statistics agree to fp32 summation order, and mean / variance match a float64 reduction of
    the stored output (models/common.py:31-128: conv -> BatchNorm(train))."""
    from deeplearningexamples_amd import functional as F
    gen = torch.Generator().manual_seed(c + ko)
    x = (torch.randn(nhw + (c,), generator=gen) * 0.5).to(dtype).to(cuda)
    w = (torch.randn((ko, 1, 1, c), generator=gen) / c ** 0.5).to(dtype).to(cuda)
    res = []
    for mode in (0, 1):
        g8.dle_gemm8_mode(mode)
        before = g8.dle_gemm8_launch_count()
        rm, rv = torch.zeros(ko, device=cuda), torch.ones(ko, device=cuda)
        y, mean, rstd = F.conv2d_fwd_bnstats(x, w, 1, 0, rm, rv)
        torch.cuda.synchronize()
        assert (g8.dle_gemm8_launch_count() > before) == (mode == 1), "mode %d: wrong kernel took the launch" % mode
        res.append((y, mean, rstd, rm, rv))
    (y0, m0, r0, rm0, rv0), (y1, m1, r1, rm1, rv1) = res
    assert torch.equal(y0, y1)
    y64 = y1.double().view(-1, ko)
    mu, var = y64.mean(0), y64.var(0, unbiased=False)
    assert torch.allclose(m1.double(), mu, atol=2e-6 * float(y64.abs().max()) + 1e-7, rtol=1e-5)
    assert torch.allclose(r1.double(), 1.0 / torch.sqrt(var + 1e-5), rtol=2e-5)
    assert torch.allclose(m1, m0, atol=1e-6, rtol=1e-5) and torch.allclose(r1, r0, rtol=1e-5)
    assert torch.allclose(rm1, rm0, atol=1e-6, rtol=1e-5) and torch.allclose(rv1, rv0, rtol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("m,n,k", [(1024, 512, 256), (2048, 1024, 480), (768, 528, 320)])
def test_relu_as_keep_bits_forward_and_masked_data_gradient(g8, cuda, dtype, m, n, k):
    """One (Linear + ReLU) layer of an MLP with its ReLU mask as ONE BIT per element (dlrm/nn/mlps.py:38-43): the forward epilogue
    leaves y AND the keep bits of the rounded y; the masked data gradient of the layer above reads the bits instead of y.
    y == gemm(act=ReLU) bit for bit, bits == (y > 0), dX == gemm_colsum(src=y) bit for bit, the bias gradient to fp32 order."""
    from deeplearningexamples_amd import functional as F
    from deeplearningexamples_amd import _cabi as C
    gen = torch.Generator().manual_seed(m + n + k)
    x = (torch.randn(m, k, generator=gen) * 0.5).to(dtype).to(cuda)
    w = (torch.randn(n, k, generator=gen) * 0.1).to(dtype).to(cuda)
    bias = (torch.randn(n, generator=gen) * 0.1).to(cuda)
    before = g8.dle_gemm8_launch_count()
    r = F.gemm_relu_bits(x, w, m, n, k, bias)
    assert r is not None and g8.dle_gemm8_launch_count() > before
    y, bits = r
    y_ref = F.gemm(x, w, m, n, k, True, True, bias=bias, act=C.ACT_RELU)
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref)
    keep = ((bits.view(-1, 1) >> torch.arange(8, device=cuda, dtype=torch.uint8)) & 1).view(m, n).bool()
    assert torch.equal(keep, y > 0)
    # the layer above: g [m, n2] w2 [n2, n] -> dX [m, n] under the mask of y, + column sums
    n2 = 256
    g = (torch.randn(m, n2, generator=gen) * 0.5).to(dtype).to(cuda)
    w2 = (torch.randn(n2, n, generator=gen) * 0.1).to(dtype).to(cuda)
    cs_ref, cs = torch.zeros(n, device=cuda), torch.zeros(n, device=cuda)
    dx_ref = F.gemm_colsum(g, w2, m, n, n2, y, cs_ref)
    dx = F.gemm_colsum_bits(g, w2, m, n, n2, bits, cs)
    torch.cuda.synchronize()
    assert dx_ref is not None and dx is not None
    assert torch.equal(dx, dx_ref)
    mag = dx.double().abs().sum(0)
    assert torch.all((cs.double() - cs_ref.double()).abs() <= 4e-6 * mag + 1e-9)
    assert torch.all((cs.double() - dx.double().sum(0)).abs() <= 2e-6 * mag + 1e-9)
