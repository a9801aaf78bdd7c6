"""apex.optimizers.FusedSGD on dle_mt_sgd (Recommendation/DLRM/dlrm/scripts/main.py:469-471).  One multi-tensor
launch per parameter group; momentum buffers are created on the first step like apex (first_step flag)."""
import torch

from deeplearningexamples_amd import multi_tensor as _mt


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False,
                 wd_after_momentum=False, materialize_master_grads=True, set_grad_none=False):
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        if wd_after_momentum:
            raise NotImplementedError("FusedSGD shim: wd_after_momentum is not on the reference's DLRM path")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                                      nesterov=nesterov))
        self._cache = _mt.TableCache()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            if any(p.grad.is_sparse for p in ps):
                raise RuntimeError("FusedSGD does not support sparse gradients")
            mom = group["momentum"]
            first = False
            lists = [[p.grad.contiguous() for p in ps], [p.data for p in ps]]
            if mom != 0:
                bufs = []
                for p in ps:
                    st = self.state[p]
                    if "momentum_buffer" not in st:
                        st["momentum_buffer"] = torch.zeros_like(p.data)
                        first = True
                    bufs.append(st["momentum_buffer"])
                lists.append(bufs)
            table = self._cache.get("g%d" % gi, lists)
            _mt.sgd(table, group["lr"], momentum=mom, dampening=group["dampening"], weight_decay=group["weight_decay"],
                    nesterov=group["nesterov"], first_step=first, has_momentum=mom != 0)
        return loss


class FusedAdam(torch.optim.Adam):
    """Not on the measured path (DLRM --optimizer adam only): plain torch.optim.Adam semantics."""

    def __init__(self, params, lr=1e-3, bias_correction=True, betas=(0.9, 0.999), eps=1e-8, adam_w_mode=True,
                 weight_decay=0.0, amsgrad=False, set_grad_none=True):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)
