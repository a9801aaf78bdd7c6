"""lamb_oracle cross-check against an independent float64 closed form (parity is otherwise unpinned)."""
import numpy as np

from oracle import lamb_oracle as L


def test_lamb_oracle_vs_float64_closed_form():
    rng = np.random.default_rng(0)
    shapes = [(7,), (33, 5), (1024,), (3, 3, 3)]
    g = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    p = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    m = [np.abs(rng.standard_normal(s)).astype(np.float32) * 0.1 for s in shapes]
    v = [np.abs(rng.standard_normal(s)).astype(np.float32) * 0.01 for s in shapes]
    lr, b1, b2, eps, step, wd = 6e-3, 0.9, 0.999, 1e-6, 3, 0.01
    gnorm, _ = L.l2norm(g)
    upd, p2, m2, v2, _ = L.lamb_step(g, p, m, v, lr, b1, b2, eps, step, True, wd, True, 1, gnorm, np.float32(1.0))
    clip = max(float(gnorm), 1.0)
    for i in range(len(shapes)):
        G = g[i].astype(np.float64) / clip
        M = b1 * m[i] + (1 - b1) * G
        V = b2 * v[i] + (1 - b2) * G * G
        U = (M / (1 - b1 ** step)) / (np.sqrt(V / (1 - b2 ** step)) + eps) + wd * p[i]
        ratio = lr * np.linalg.norm(p[i].astype(np.float64)) / np.linalg.norm(U)
        np.testing.assert_allclose(m2[i], M, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(v2[i], V, rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(p2[i], p[i] - ratio * U, rtol=2e-5, atol=1e-6)


def test_sgd_oracle_matches_torch():
    import torch
    rng = np.random.default_rng(1)
    p0 = rng.standard_normal(50).astype(np.float32)
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.SGD([tp], lr=0.1, momentum=0.875, weight_decay=3.0517578125e-05, nesterov=False)
    p, buf = p0.copy(), None
    for it in range(3):
        g = rng.standard_normal(50).astype(np.float32)
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        p, buf = L.sgd_step(g, p, buf, 0.1, 0.875, 0.0, 3.0517578125e-05, False, first=(it == 0))
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=1e-6, atol=1e-7)


def test_lamb_host_sequence_reproduces_the_reference_classes():
    """tests/golden/lamb_ref_steps.npz = the reference's unmodified FusedLAMBAMP + PolyWarmUpScheduler + torch
    GradScaler stepping on CPU (oracle/make_golden.py gen_lamb; fused_lamb_CUDA bound to the numpy kernels).  The
    oracle's restatement of that host sequence must land on the same bits: step counter frozen on the overflow
    step, lr derived from group['step'] + 1, loss-scale back-off / growth, masters, moments, fp16 model copies."""
    import os
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "lamb_ref_steps.npz"))
    case = L.LAMB_GOLDEN_CASE
    params0, grads = L.lamb_golden_inputs(case)
    host = L.FusedLambHost(params0, case["groups"], case["lr"], case["warmup"], case["total_steps"],
                           init_scale=case["init_scale"], growth_interval=case["growth_interval"])
    for it, g in enumerate(grads):
        assert float(host.scale) == gold["scale"][it]
        scaled = {}
        for k, (_, half) in case["shapes"].items():
            gs = (g[k] * np.float32(host.scale)).astype(np.float16 if half else np.float32)
            if it in case["overflow_at"] and k == "w_c":
                gs = gs.copy(); gs.reshape(-1)[5] = np.inf
            scaled[k] = gs
        found = host.optimizer_step(scaled)
        assert float(found) == gold["found_inf"][it]
        assert host.step == int(gold["step"][it])
        assert np.float32(host.lr) == np.float32(gold["lr"][it])
    assert list(gold["step"]) == [1, 2, 2, 3, 4, 5, 6]          # the overflow step does not advance the counter
    assert float(host.scale) == float(gold["final_scale"])
    for k, (_, half) in case["shapes"].items():
        np.testing.assert_array_equal(host.p[k], gold["p_" + k])
        np.testing.assert_array_equal(host.m[k], gold["m_" + k])
        np.testing.assert_array_equal(host.v[k], gold["v_" + k])
        if half:
            np.testing.assert_array_equal(host.p16[k], gold["p16_" + k])
