"""The reference-facing stand-ins in shims/  compute what the reference's compiled modules compute:
fused_lamb_CUDA through apex's multi_tensor_applier vs the LAMB oracle; dlrm.cuda_ext pybind names vs the DLRM
oracle; apex.mlp.MLP / FusedSGD vs torch.  GPU only; does not need the reference tree."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import lamb_oracle as L
from oracle import dlrm_oracle as DO

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shims"))


def test_fused_lamb_cuda_matches_oracle(cuda):
    import fused_lamb_CUDA
    from apex.multi_tensor_apply import multi_tensor_applier
    rng = np.random.default_rng(3)
    shapes = [(1000,), (37, 129), (4096,), (3,)]
    g = [rng.standard_normal(s).astype(np.float32) * 0.1 for s in shapes]
    p = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    m = [rng.standard_normal(s).astype(np.float32) * 0.05 for s in shapes]
    v = [np.abs(rng.standard_normal(s)).astype(np.float32) * 0.01 for s in shapes]
    lr, b1, b2, eps, step, wd = 6e-3, 0.9, 0.999, 1e-6, 3, 0.01
    gs, ps, ms, vs = ([torch.from_numpy(a.copy()).to(cuda) for a in lst] for lst in (g, p, m, v))
    noop = torch.zeros(1, dtype=torch.int32, device=cuda)
    one = torch.ones(1, device=cuda)
    gnorm = multi_tensor_applier(fused_lamb_CUDA.multi_tensor_l2norm, noop, [gs], False)[0]
    gn_ref, _ = L.l2norm(g)
    np.testing.assert_allclose(float(gnorm.item()), float(gn_ref), rtol=1e-5)
    multi_tensor_applier(fused_lamb_CUDA.multi_tensor_lamb, noop, [gs, ps, ms, vs],
                         torch.tensor(lr, device=cuda), b1, b2, eps, torch.tensor([step], dtype=torch.int32, device=cuda),
                         1, wd, 1, 1, gnorm, one.clone(), False, torch.zeros(1, device=cuda), one.clone())
    upd, p2, m2, v2, _ = L.lamb_step(g, p, m, v, lr, b1, b2, eps, step, True, wd, True, 1, gn_ref, np.float32(1.0))
    for i in range(len(shapes)):
        np.testing.assert_allclose(ms[i].cpu().numpy(), m2[i], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(vs[i].cpu().numpy(), v2[i], rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(ps[i].cpu().numpy(), p2[i], rtol=1e-5, atol=1e-6)


def test_dlrm_cuda_ext_pybind_surface(cuda):
    import dle_reference_shims as S
    mods = S.cuda_ext_modules()
    inter, fused, sparse = (mods["dlrm.cuda_ext." + n] for n in ("interaction_ampere", "fused_embedding", "sparse_gather"))
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 27, 128, generator=g).half()
    out = inter.dotBasedInteractFwd(x.to(cuda), x[:, 0, :].to(cuda))
    ref = DO.dot_interact_fwd(x.float().numpy())
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=2e-2, atol=2e-2)
    up = torch.randn(out.shape, generator=g).half()
    grad, mlp_grad = inter.dotBasedInteractBwd(x.to(cuda), up.to(cuda))
    gref, mref = DO.dot_interact_bwd(x.float().numpy(), up.float().numpy())
    np.testing.assert_allclose(grad.float().cpu().numpy(), gref, rtol=3e-2, atol=3e-2)
    np.testing.assert_allclose(mlp_grad.float().cpu().numpy(), mref, rtol=1e-3, atol=1e-3)
    sizes = [50, 7, 300]
    offs = torch.tensor([0] + sizes).cumsum(0)
    w = torch.randn(int(offs[-1]), 128, generator=g)
    idx = torch.stack([torch.randint(0, s, (32,), generator=g) for s in sizes], 1)
    y = fused.gather_gpu_fused_fwd(w.to(cuda), idx.to(cuda), offs.to(cuda), True)
    assert y.dtype == torch.float16 and torch.equal(y.cpu(), w[idx + offs[:-1]].half())
    gy = torch.randn(32, 3, 128, generator=g)
    sg = fused.gather_gpu_fused_bwd(w.to(cuda), idx.to(cuda), offs.to(cuda), gy.to(cuda))
    dense = torch.zeros_like(w).index_add_(0, (idx + offs[:-1]).reshape(-1), gy.reshape(-1, 128))
    assert sg.is_sparse and torch.allclose(sg.to_dense().cpu(), dense, atol=1e-5)
    rows = (idx + offs[:-1]).to(cuda)
    y2 = sparse.gather_gpu_fwd(w.to(cuda), rows)
    assert torch.equal(y2.cpu(), w[idx + offs[:-1]])
    wd = w.clone().to(cuda)
    sparse.gather_gpu_bwd_fuse_sgd(gy.to(cuda), rows, 0.5, wd)
    assert torch.allclose(wd.cpu(), w - 0.5 * dense, atol=1e-4)
    assert torch.allclose(sparse.gather_gpu_bwd(gy.to(cuda), rows, w.shape[0]).to_dense().cpu(), dense, atol=1e-5)


def test_apex_mlp_and_fused_sgd(cuda):
    from apex.mlp import MLP
    from apex.optimizers import FusedSGD
    torch.manual_seed(0)
    mlp = MLP([16, 64, 32]).to(cuda)
    ref = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.ReLU(), torch.nn.Linear(64, 32), torch.nn.ReLU()).to(cuda)
    with torch.no_grad():
        ref[0].weight.copy_(mlp.weights[0]); ref[0].bias.copy_(mlp.biases[0])
        ref[2].weight.copy_(mlp.weights[1]); ref[2].bias.copy_(mlp.biases[1])
    x = torch.randn(256, 16, device=cuda)
    opt, ropt = FusedSGD(mlp.parameters(), lr=0.1, momentum=0.9), torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9)
    for _ in range(3):
        with torch.autocast("cuda", dtype=torch.float16):
            y = mlp(x)
        yr = ref(x)
        assert torch.allclose(y.float(), yr, atol=2e-2, rtol=2e-2)
        y.float().square().mean().backward()
        yr.square().mean().backward()
        for a, b in zip(list(mlp.weights) + list(mlp.biases), [ref[0].weight, ref[2].weight, ref[0].bias, ref[2].bias]):
            assert torch.allclose(a.grad, b.grad, atol=2e-3, rtol=5e-2), (a.grad - b.grad).abs().max()
        opt.step(); ropt.step()
        opt.zero_grad(); ropt.zero_grad()
    assert torch.allclose(mlp.weights[0], ref[0].weight, atol=5e-3)
