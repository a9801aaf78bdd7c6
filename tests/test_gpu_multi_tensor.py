"""Multi-tensor L2 norm / LAMB / SGD kernels vs oracle/lamb_oracle.py."""
import numpy as np
import pytest
import torch

from oracle import lamb_oracle as L

pytestmark = pytest.mark.gpu

SHAPES = [(7,), (33, 5), (65536,), (65537,), (3, 3, 3), (200000,), (1,), (1024, 1024)]


def _mt():
    from deeplearningexamples_amd import multi_tensor as mt
    return mt


def _rand(rng, shapes, scale=1.0, dtype=np.float32, positive=False):
    out = []
    for s in shapes:
        a = rng.standard_normal(s) * scale
        out.append((np.abs(a) if positive else a).astype(dtype))
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_l2norm(cuda, dtype):
    mt = _mt()
    rng = np.random.default_rng(0)
    xs = [torch.from_numpy(a).to(dtype) for a in _rand(rng, SHAPES)]
    dev = [x.to(cuda) for x in xs]
    noop = torch.zeros(1, dtype=torch.int32, device=cuda)
    tot, per = mt.l2norm(mt.TensorTable([dev]), noop, per_tensor=True)
    rt, rp = L.l2norm([x.float().numpy() for x in xs])
    np.testing.assert_allclose(per.cpu().numpy(), rp, rtol=2e-5)
    np.testing.assert_allclose(tot.cpu().numpy(), [rt], rtol=2e-5)
    assert noop.item() == 0
    dev[2][5] = float("inf")
    mt.l2norm(mt.TensorTable([dev]), noop, per_tensor=False)
    assert noop.item() == 1                                       # non-finite -> noop flag (l2norm_kernel.cu:103)


@pytest.mark.parametrize("gdtype,copy", [(torch.float32, False), (torch.float16, True), (torch.bfloat16, True)])
@pytest.mark.parametrize("mode,wd", [(1, 0.01), (1, 0.0), (0, 0.01)])
def test_lamb_full_step(cuda, gdtype, copy, mode, wd):
    mt = _mt()
    rng = np.random.default_rng(1)
    npdt = {torch.float32: np.float32, torch.float16: np.float16}.get(gdtype)
    g32 = _rand(rng, SHAPES, 0.1)
    g_t = [torch.from_numpy(a).to(gdtype) for a in g32]
    g_np = [t.float().numpy() for t in g_t]
    p, m, v = _rand(rng, SHAPES), _rand(rng, SHAPES, 0.05), _rand(rng, SHAPES, 0.01, positive=True)
    lr, b1, b2, eps, step, scale = 6e-3, 0.9, 0.999, 1e-6, 4, 128.0
    gs = [(t.float() * scale).to(gdtype).to(cuda) for t in g_t]      # scaled grads, as under GradScaler
    g_np = [t.cpu().float().numpy() for t in gs]
    ps = [torch.from_numpy(a.copy()).to(cuda) for a in p]
    ms = [torch.from_numpy(a.copy()).to(cuda) for a in m]
    vs = [torch.from_numpy(a.copy()).to(cuda) for a in v]
    copies = [torch.zeros_like(x, dtype=gdtype) for x in ps] if copy else None
    noop = torch.zeros(1, dtype=torch.int32, device=cuda)
    gnorm, _ = mt.l2norm(mt.TensorTable([gs]), noop)
    max_norm = torch.tensor([1.0 * scale], device=cuda)
    inv_scale = torch.tensor([1.0 / scale], device=cuda)
    step_t = torch.tensor([step], dtype=torch.int32, device=cuda)
    lr_t = torch.tensor(lr, device=cuda)
    # host sequence of multi_tensor_lamb_cuda (multi_tensor_lamb.cu:371-500)
    _, pn = mt.l2norm(mt.TensorTable([ps]), noop, per_tensor=True)
    mt.lamb_stage1(mt.TensorTable([gs, ps, ms, vs]), noop, b1, b2, 1 - b1, step_t, True, eps, mode, wd,
                   gnorm, max_norm, inv_scale)
    _, un = mt.l2norm(mt.TensorTable([gs]), noop, per_tensor=True)
    lists = [gs, ps, copies] if copy else [gs, ps]
    mt.lamb_stage2(mt.TensorTable(lists), noop, pn, un, lr_t, wd, False)

    gn_ref, _ = L.l2norm(g_np)
    to_np = (lambda a: a.astype(np.float16)) if gdtype == torch.float16 else None
    class _BF:  # numpy has no bf16: emulate the cast through torch
        pass
    def cast(a):
        return torch.from_numpy(a).to(gdtype).float().numpy()
    upd, p2, m2, v2, _ = L.lamb_step(g_np, p, m, v, lr, b1, b2, eps, step, True, wd, True, mode, gn_ref,
                                     np.float32(scale), inv_scale=1.0 / scale,
                                     round_update=None if gdtype == torch.float32 else cast)
    for i in range(len(SHAPES)):
        np.testing.assert_allclose(ms[i].cpu().numpy(), m2[i], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(vs[i].cpu().numpy(), v2[i], rtol=1e-5, atol=1e-8)
        tol = 1e-5 if gdtype == torch.float32 else (2e-3 if gdtype == torch.float16 else 1.6e-2)
        np.testing.assert_allclose(gs[i].cpu().float().numpy(), cast(upd[i]), rtol=tol, atol=tol)
        # p moves by ratio * u with u rounded to the gradient dtype; a 1-ulp difference in that rounding is the floor
        step_mag = np.abs(p2[i] - p[i]).max()
        np.testing.assert_allclose(ps[i].cpu().numpy(), p2[i], rtol=1e-5, atol=tol * step_mag + 1e-6)
        if copy:
            assert torch.equal(copies[i], ps[i].to(gdtype))


@pytest.mark.parametrize("gdtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("mode", [1, 0])
def test_lamb_stage1_with_the_norms_inside_equals_the_three_passes(cuda, gdtype, mode):
    """dle_mt_lamb_stage1_norms == l2norm(p) -> stage 1 -> l2norm(update) (the host sequence of multi_tensor_lamb_cuda,
    multi_tensor_lamb.cu:380-420): update, m, v BIT-identical; the norms to the fp32 summation of a chunk's ragged tail (bit-identical
    for tensors whose chunks are multiples of 4 elements long); an overflowing update raises the flag in the fold and the norms
    come back 0; a flag already set leaves everything untouched."""
    mt = _mt()
    rng = np.random.default_rng(3)
    mk = lambda arrs, dt=torch.float32: [torch.from_numpy(a.copy()).to(dt).to(cuda) for a in arrs]
    g0 = _rand(rng, SHAPES, 12.8)
    p0, m0, v0 = _rand(rng, SHAPES), _rand(rng, SHAPES, 0.05), _rand(rng, SHAPES, 0.01, positive=True)
    one = torch.ones(1, device=cuda)
    args = (0.9, 0.999, 0.1, torch.tensor([4], dtype=torch.int32, device=cuda), True, 1e-6, mode, 0.01,
            torch.tensor([30.0], device=cuda), torch.tensor([128.0], device=cuda), torch.tensor([1.0 / 128.0], device=cuda))
    ga, pa, ma, va = mk(g0, gdtype), mk(p0), mk(m0), mk(v0)
    noop = torch.zeros(1, dtype=torch.int32, device=cuda)
    _, pn_ref = mt.l2norm(mt.TensorTable([pa]), noop, per_tensor=True)
    mt.lamb_stage1(mt.TensorTable([ga, pa, ma, va]), noop, *args)
    _, un_ref = mt.l2norm(mt.TensorTable([ga]), noop, per_tensor=True)
    gb, pb, mb, vb = mk(g0, gdtype), mk(p0), mk(m0), mk(v0)
    pn, un = mt.lamb_stage1_norms(mt.TensorTable([gb, pb, mb, vb]), noop, *args)
    assert noop.item() == 0
    for a, b in zip(ga + ma + va, gb + mb + vb):
        assert torch.equal(a, b)
    np.testing.assert_allclose(pn.cpu().numpy(), pn_ref.cpu().numpy(), rtol=2e-6)
    np.testing.assert_allclose(un.cpu().numpy(), un_ref.cpu().numpy(), rtol=2e-6)
    whole = [i for i, sh in enumerate(SHAPES) if int(np.prod(sh)) % 4 == 0]
    assert len(whole) >= 3 and torch.equal(pn[whole], pn_ref[whole]) and torch.equal(un[whole], un_ref[whole])
    # a non-finite parameter: the fold raises the flag (what the l2norm sweep over p would have done) and zeroes the norms
    gc, pc, mc, vc = mk(g0, gdtype), mk(p0), mk(m0), mk(v0)
    pc[5][77] = float("inf")
    pn2, un2 = mt.lamb_stage1_norms(mt.TensorTable([gc, pc, mc, vc]), noop, *args)
    assert noop.item() == 1 and float(pn2[5]) == 0.0 and float(un2[5]) == 0.0
    # the flag set beforehand: nothing moves (multi_tensor_lamb.cu:63-65)
    gd, pd, md, vd = mk(g0, gdtype), mk(p0), mk(m0), mk(v0)
    pn3, un3 = mt.lamb_stage1_norms(mt.TensorTable([gd, pd, md, vd]), noop, *args)
    assert all(torch.equal(a, b) for a, b in zip(gd + md, mk(g0, gdtype) + mk(m0))) and float(pn3.abs().sum()) == 0.0


def test_lamb_noop_skips_everything(cuda):
    mt = _mt()
    g = [torch.ones(1000, device=cuda)]
    p, m, v = [torch.ones(1000, device=cuda)], [torch.zeros(1000, device=cuda)], [torch.zeros(1000, device=cuda)]
    noop = torch.ones(1, dtype=torch.int32, device=cuda)
    one = torch.ones(1, device=cuda)
    mt.lamb_stage1(mt.TensorTable([g, p, m, v]), noop, 0.9, 0.999, 0.1, torch.ones(1, dtype=torch.int32, device=cuda),
                   True, 1e-6, 1, 0.01, one, one, one)
    assert m[0].abs().sum() == 0 and (g[0] == 1).all()


@pytest.mark.parametrize("gdtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("nesterov", [False, True])
def test_sgd_momentum(cuda, gdtype, nesterov):
    mt = _mt()
    rng = np.random.default_rng(2)
    p = _rand(rng, SHAPES)
    ps = [torch.from_numpy(a.copy()).to(cuda) for a in p]
    bufs = [torch.zeros_like(x) for x in ps]
    ref_buf = [None] * len(p)
    inv = torch.tensor([0.25], device=cuda)
    for it in range(3):
        g = [torch.from_numpy(a).to(gdtype) for a in _rand(rng, SHAPES)]
        gs = [x.to(cuda) for x in g]
        mt.sgd(mt.TensorTable([gs, ps, bufs]), torch.tensor(0.1, device=cuda), 0.875, 0.0, 3.0517578125e-05,
               nesterov, first_step=(it == 0), inv_scale=inv)
        for i in range(len(p)):
            p[i], ref_buf[i] = L.sgd_step(g[i].float().numpy(), p[i], ref_buf[i], 0.1, 0.875, 0.0,
                                          3.0517578125e-05, nesterov, first=(it == 0), inv_scale=0.25)
            np.testing.assert_allclose(ps[i].cpu().numpy(), p[i], rtol=1e-5, atol=1e-6)
    # plain SGD, host lr, skip flag
    before = [x.clone() for x in ps]
    mt.sgd(mt.TensorTable([gs, ps]), 0.1, skip_flag=torch.ones(1, device=cuda))
    assert all(torch.equal(a, b) for a, b in zip(ps, before))
    mt.sgd(mt.TensorTable([gs, ps]), 0.1)
    for a, b, gg in zip(ps, before, gs):
        np.testing.assert_allclose(a.cpu().numpy(), (b - 0.1 * gg.float()).cpu().numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("cdtype", [torch.float16, torch.bfloat16])
def test_sgd_with_an_optional_model_copy_per_tensor(cuda, cdtype):
    """One SGD launch over weights that keep a 16-bit working copy AND tensors that have none (biases, K-padded first layers): a
    None entry in the copy list is pointer 0 in the table -- the tensor is stepped, nothing else is written (apex FusedSGD with
    materialize_master_grads over a model whose biases stay fp32, dlrm/scripts/main.py:469-471)."""
    mt = _mt()
    rng = np.random.default_rng(5)
    p0, g0 = _rand(rng, SHAPES), _rand(rng, SHAPES, 0.1)
    ps = [torch.from_numpy(a.copy()).to(cuda) for a in p0]
    gs = [torch.from_numpy(a.copy()).to(cuda) for a in g0]
    guard = torch.full((4096,), 7.0, dtype=cdtype, device=cuda)                    # nothing may be written through a null pointer
    copies = [torch.zeros(s, dtype=cdtype, device=cuda) if i % 2 == 0 else None for i, s in enumerate(SHAPES)]
    table = mt.TensorTable([gs, ps, copies])
    assert table.dtypes[2] == cdtype
    mt.sgd(table, torch.tensor(0.5, device=cuda), has_momentum=False, model_copy=True)
    for i in range(len(SHAPES)):
        want = torch.from_numpy(p0[i] - 0.5 * g0[i]).to(cuda)
        np.testing.assert_allclose(ps[i].cpu().numpy(), want.cpu().numpy(), rtol=1e-6, atol=1e-7)
        if copies[i] is not None:
            assert torch.equal(copies[i], ps[i].to(cdtype))
    assert bool((guard == 7.0).all())
    with pytest.raises(ValueError):
        mt.TensorTable([[None] + gs[1:], ps])                                       # the first list holds every tensor
