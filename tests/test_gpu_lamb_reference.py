"""LAMB on the HIP kernels against the golden the REFERENCE'S OWN classes produced (tests/golden/lamb_ref_steps.npz:
unmodified FusedLAMBAMP + PolyWarmUpScheduler + torch GradScaler, oracle/make_golden.py gen_lamb).

 * test_hip_kernels_through_reference_host_sequence -- always runs on the GPU box: the restated host sequence
   (oracle.lamb_oracle.FusedLambHost, itself bit-identical to the reference classes on CPU) with the two
   fused_lamb_CUDA entry points of shims/ (the b2 boundary, HIP kernels underneath) substituted for the numpy kernels.
 * test_reference_class_steps_on_hip_kernels -- when the reference tree is mounted: the reference's unmodified
   FusedLAMBAMP class imported through shims/ steps CUDA tensors under torch.cuda.amp.GradScaler.
Tolerance: fp32 arithmetic in a different association order (device pow / rsqrt vs numpy) -> 2e-5 relative on masters
and moments after 6 steps; the fp16 model copies may differ by one fp16 ulp where a master sits on a rounding edge."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import lamb_oracle as L

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "shims"))
GOLD = os.path.join(ROOT, "tests", "golden", "lamb_ref_steps.npz")


def _scaled_grads(case, g, scale, it):
    out = {}
    for k, (_, half) in case["shapes"].items():
        gs = (g[k] * np.float32(scale)).astype(np.float16 if half else np.float32)
        if it in case["overflow_at"] and k == "w_c":
            gs = gs.copy(); gs.reshape(-1)[5] = np.inf
        out[k] = gs
    return out


def _check_against_gold(gold, case, p, m, v, p16):
    for k, (_, half) in case["shapes"].items():
        # half parameters: the update is rounded to fp16 inside the gradient buffer (multi_tensor_lamb.cu:163), so an
        # fp32-ulp difference upstream can flip one fp16 rounding of it: lr x trust ratio x 2^-11 |u| ~ 5e-6 absolute
        np.testing.assert_allclose(p[k], gold["p_" + k], rtol=2e-5, atol=1e-5 if half else 2e-6, err_msg=k)
        np.testing.assert_allclose(m[k], gold["m_" + k], rtol=2e-5, atol=1e-7, err_msg=k)
        np.testing.assert_allclose(v[k], gold["v_" + k], rtol=2e-5, atol=1e-8, err_msg=k)
        if half:
            a, b = np.asarray(p16[k], np.float32), gold["p16_" + k].astype(np.float32)
            assert np.all(np.abs(a - b) <= np.abs(b) * 2.0 ** -10 + 1e-7), k


def test_hip_kernels_through_reference_host_sequence(cuda):
    import fused_lamb_CUDA as FL
    from apex.multi_tensor_apply import multi_tensor_applier
    gold = np.load(GOLD)
    case = L.LAMB_GOLDEN_CASE
    params0, grads = L.lamb_golden_inputs(case)

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)

    def k_l2norm(arrs):
        noop = torch.zeros(1, dtype=torch.int32, device=cuda)
        tot, per = multi_tensor_applier(FL.multi_tensor_l2norm, noop, [[dev(a) for a in arrs]], True)
        return np.float32(tot.item()), per.cpu().numpy()

    def k_lamb(g, p, m, v, lr, b1, b2, eps, step, bias_corr, wd, grad_avg, mode, gnorm, max_norm, use_nvlamb,
               inv_scale=1.0, grad_dtype=np.float32, model_copy_dtype=None):
        gs, ps, ms, vs = ([dev(a) for a in lst] for lst in (g, p, m, v))
        lists = [gs, ps, ms, vs]
        if model_copy_dtype is not None:
            lists.append([torch.empty(a.shape, dtype=torch.float16, device=cuda) for a in p])
        noop = torch.zeros(1, dtype=torch.int32, device=cuda)
        multi_tensor_applier(FL.multi_tensor_lamb, noop, lists, torch.tensor(float(lr), device=cuda), b1, b2, eps,
                             torch.tensor([int(step)], dtype=torch.int32, device=cuda), int(bias_corr), wd,
                             int(grad_avg), int(mode), torch.tensor([float(gnorm)], device=cuda),
                             torch.tensor([float(max_norm)], device=cuda), use_nvlamb, torch.zeros(1, device=cuda),
                             torch.tensor([float(inv_scale)], device=cuda))
        cp = [c.cpu().numpy() for c in lists[4]] if model_copy_dtype is not None else [None] * len(p)
        return ([t.cpu().numpy() for t in gs], [t.cpu().numpy() for t in ps], [t.cpu().numpy() for t in ms],
                [t.cpu().numpy() for t in vs], cp)

    host = L.FusedLambHost(params0, case["groups"], case["lr"], case["warmup"], case["total_steps"],
                           init_scale=case["init_scale"], growth_interval=case["growth_interval"],
                           kernels=(k_l2norm, k_lamb))
    for it, g in enumerate(grads):
        assert float(host.scale) == gold["scale"][it]
        found = host.optimizer_step(_scaled_grads(case, g, host.scale, it))
        assert float(found) == gold["found_inf"][it] and host.step == int(gold["step"][it])
    _check_against_gold(gold, case, host.p, host.m, host.v, host.p16)


@pytest.mark.skipif(not os.path.isdir("/root/reference/PyTorch/LanguageModeling/BERT/lamb_amp_opt"),
                    reason="reference tree not mounted (GPU box): the unmodified class cannot be imported")
def test_reference_class_steps_on_hip_kernels(cuda):
    sys.path.insert(0, "/root/reference/PyTorch/LanguageModeling/BERT/lamb_amp_opt")
    sys.path.insert(0, "/root/reference/PyTorch/LanguageModeling/BERT")
    from fused_lamb.fused_lamb import FusedLAMBAMP           # unmodified; imports apex / fused_lamb_CUDA from shims/
    import schedulers
    gold = np.load(GOLD)
    case = L.LAMB_GOLDEN_CASE
    params0, grads = L.lamb_golden_inputs(case)
    tp = {k: torch.nn.Parameter(torch.from_numpy(a.copy()).to(cuda).to(torch.float16 if half else torch.float32))
          for k, (a, half) in params0.items()}
    opt = FusedLAMBAMP([{"params": [tp[k] for k in names], "weight_decay": wd} for wd, names in case["groups"]],
                       lr=case["lr"])
    opt.setup_fp32_params()
    sched = schedulers.PolyWarmUpScheduler(opt, warmup=case["warmup"], total_steps=case["total_steps"],
                                           base_lr=case["lr"], device=cuda)
    scaler = torch.cuda.amp.GradScaler(init_scale=case["init_scale"], growth_interval=case["growth_interval"])
    for it, g in enumerate(grads):
        scaler.scale(torch.zeros(1, device=cuda))
        scale = float(scaler.get_scale())
        assert scale == gold["scale"][it]
        for k, gs in _scaled_grads(case, g, scale, it).items():
            tp[k].grad = torch.from_numpy(gs.copy()).to(cuda)
        sched.step()
        scaler.step(opt)
        scaler.update()
        opt.zero_grad(set_to_none=True)
        assert int(opt.param_groups[0]["step"].item()) == int(gold["step"][it])
    names = [k for _, ns in case["groups"] for k in ns]
    flat = [p for grp in opt.param_groups for p in grp["params"]]
    flat32 = [p for grp in opt.param_groups_fp32 for p in grp["params"]]
    p, m, v, p16 = {}, {}, {}, {}
    for k, q, q32 in zip(names, flat, flat32):
        p[k] = (q32 if q32 is not None else q).detach().float().cpu().numpy()
        m[k] = opt.state[q]["exp_avg"].cpu().numpy()
        v[k] = opt.state[q]["exp_avg_sq"].cpu().numpy()
        if q.dtype == torch.float16:
            p16[k] = q.detach().cpu().numpy()
    _check_against_gold(gold, case, p, m, v, p16)


def test_reference_class_call_trace_replayed_through_the_shims(cuda):
    """The reference's unmodified FusedLAMBAMP.step CANNOT be imported on the GPU box (no reference tree there), so its calls travel
    as data: tests/golden/lamb_ref_trace.npz holds every call the class made into `fused_lamb_CUDA` over the golden scenario --
    the tensor lists and scalar arguments exactly as fused_lamb.py:147-258 passed them (7 optimizer steps incl. the overflow step:
    21 multi_tensor_l2norm + 28 multi_tensor_lamb calls) and what each call left behind (oracle/make_golden.py gen_lamb_trace,
    run where the tree is mounted).  Each call is replayed here through shims/fused_lamb_CUDA.py (argument order of
    csrc/frontend.cpp:3-32, HIP kernels underneath) on device tensors and compared with the recorded result: returned norms,
    the noop flag, and the in-place results g (-> update), p, m, v and the fp16 model copy."""
    import fused_lamb_CUDA as FL
    calls, marks = L.load_call_trace(os.path.join(ROOT, "tests", "golden", "lamb_ref_trace.npz"))
    assert len(marks) == L.LAMB_GOLDEN_CASE["steps"] and marks[-1] == len(calls)

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)

    n_l2 = n_lamb = n_skipped_by_flag = 0
    for i, c in enumerate(calls):
        lists = [[dev(a) for a in l] for l in c["lists"]]
        noop = dev(c["noop_in"])
        if c["fn"] == "l2norm":
            tot, per = FL.multi_tensor_l2norm(int(c["chunk"]), noop, lists, c["per_tensor"])
            torch.cuda.synchronize()
            np.testing.assert_allclose(tot.cpu().numpy(), c["total"], rtol=2e-6, err_msg="call %d total" % i)
            np.testing.assert_allclose(per.cpu().numpy(), c["per"], rtol=2e-6, err_msg="call %d per-tensor" % i)
            assert np.array_equal(noop.cpu().numpy(), c["noop_out"]), "call %d: noop flag" % i
            n_l2 += 1
            continue
        FL.multi_tensor_lamb(int(c["chunk"]), noop, lists, dev(c["lr"]), c["beta1"], c["beta2"], c["eps"], dev(c["step"]),
                             int(c["bias_correction"]), c["weight_decay"], int(c["grad_averaging"]), int(c["mode"]),
                             dev(c["global_grad_norm"]), dev(c["max_grad_norm"]), c["use_nvlamb"], dev(c["found_inf"]),
                             dev(c["inv_scale"]))
        torch.cuda.synchronize()
        n_lamb += 1
        n_skipped_by_flag += int(c["noop_in"].reshape(-1)[0] != 0)
        for li, (got_l, want_l) in enumerate(zip(lists, c["out"])):
            for ti, (got, want) in enumerate(zip(got_l, want_l)):
                g = got.float().cpu().numpy()
                w = want.astype(np.float32)
                if want.dtype == np.float16:                 # 16-bit results: one fp16 ulp where an fp32 value sits on a rounding edge
                    fin = np.isfinite(w)
                    assert np.array_equal(np.isfinite(g), fin), "call %d list %d tensor %d" % (i, li, ti)
                    assert np.all(np.abs(g[fin] - w[fin]) <= np.abs(w[fin]) * 2.0 ** -10 + 1e-7), "call %d list %d tensor %d" % (i, li, ti)
                else:
                    # fp32 results: device pow / rsqrt / division against numpy's, one call deep (measured <= 1.9e-5 relative).
                    # Masters (list 1) of HALF parameters: the update is rounded to fp16 inside the gradient buffer
                    # (multi_tensor_lamb.cu:163), so an fp32-ulp difference upstream can flip one fp16 rounding of it:
                    # lr x trust ratio x 2^-11 |u| ~ 5e-6 absolute (the same allowance as _check_against_gold)
                    half_call = c["lists"][0][0].dtype == np.float16
                    np.testing.assert_allclose(g, w, rtol=4e-5, atol=1e-5 if (half_call and li == 1) else 2e-7, equal_nan=True,
                                               err_msg="call %d list %d tensor %d" % (i, li, ti))
    print("replayed %d l2norm + %d lamb calls of the reference class (%d under a set noop flag)" % (n_l2, n_lamb, n_skipped_by_flag))
    assert n_l2 == 21 and n_lamb == 28
