"""shims/: the reference's unmodified modules import against this library (CPU, needs /root/reference -- skipped on
the GPU box), and the compiled-module stand-ins have the pybind surface the reference calls."""
import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/PyTorch"
sys.path.insert(0, os.path.join(ROOT, "shims"))


def test_shim_surface():
    import dle_reference_shims as S
    mods = S.cuda_ext_modules()
    assert sorted(mods) == ["dlrm.cuda_ext.fused_embedding", "dlrm.cuda_ext.interaction_ampere",
                            "dlrm.cuda_ext.interaction_volta", "dlrm.cuda_ext.sparse_gather"]
    for fn in ("gather_gpu_fused_fwd", "gather_gpu_fused_bwd"):
        assert callable(getattr(mods["dlrm.cuda_ext.fused_embedding"], fn))
    for fn in ("dotBasedInteractFwd", "dotBasedInteractBwd"):
        assert callable(getattr(mods["dlrm.cuda_ext.interaction_ampere"], fn))
    for fn in ("gather_gpu_fwd", "gather_gpu_bwd", "gather_gpu_bwd_fuse_sgd"):
        assert callable(getattr(mods["dlrm.cuda_ext.sparse_gather"], fn))
    import fused_lamb_CUDA, amp_C
    from apex.multi_tensor_apply import multi_tensor_applier
    assert multi_tensor_applier.available and amp_C.multi_tensor_lamb is fused_lamb_CUDA.multi_tensor_lamb
    calls = []
    multi_tensor_applier(lambda chunk, noop, lists, *a: calls.append((chunk, noop, lists, a)), "noop", [[1]], 7)
    assert calls == [(2048 * 32, "noop", [[1]], (7,))]
    from apex.mlp import MLP
    m = MLP([13, 512, 256, 128])
    assert [tuple(w.shape) for w in m.weights] == [(512, 13), (256, 512), (128, 256)] and len(m.biases) == 3
    import pynvml
    pynvml.nvmlInit()
    assert len(pynvml.nvmlDeviceGetCpuAffinity(pynvml.nvmlDeviceGetHandleByIndex(0), 2)) == 2
    from absl import flags
    fv = flags.FlagValues()
    flags.DEFINE_integer("batch_size", 65536, "h", flag_values=fv)
    flags.DEFINE_boolean("amp", False, "h", flag_values=fv)
    flags.DEFINE_list("top_mlp_sizes", [1024, 1024, 512, 256, 1], "h", flag_values=fv)
    assert fv(["prog", "--batch_size=8", "--amp", "--top_mlp_sizes", "4,2,1", "pos"]) == ["prog", "pos"]
    assert fv.flag_values_dict() == {"batch_size": 8, "amp": True, "top_mlp_sizes": ["4", "2", "1"]}
    fv.set_default("batch_size", 16)
    assert fv.batch_size == 8
    from lddl.torch import get_bert_pretrain_data_loader
    batch = next(iter(get_bert_pretrain_data_loader("unused", data_loader_kwargs={"batch_size": 2})))
    assert sorted(batch) == ["attention_mask", "input_ids", "labels", "next_sentence_labels", "token_type_ids"]
    assert batch["input_ids"].shape == (2, 128) and int((batch["labels"] != -1).sum()) == 40


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_reference_modules_import_through_shims():
    import dle_reference_shims as S
    S.install()
    saved = list(sys.path)
    try:
        # BERT: the optimizer module binds fused_lamb_CUDA / apex.multi_tensor_apply / amp_C at import time
        sys.path.insert(0, os.path.join(REF, "LanguageModeling/BERT"))
        sys.path.insert(0, os.path.join(REF, "LanguageModeling/BERT/lamb_amp_opt"))
        fl = importlib.import_module("fused_lamb.fused_lamb")
        assert hasattr(fl, "FusedLAMBAMP")
        import fused_lamb_CUDA
        assert fl.fused_lamb_CUDA is fused_lamb_CUDA
        modeling = importlib.import_module("modeling")               # needs boto3 / botocore + the gelu patch
        import torch
        assert torch.allclose(modeling.gelu(torch.tensor([0.5, -1.0])),
                              torch.nn.functional.gelu(torch.tensor([0.5, -1.0]), approximate="tanh"))
        # RN50: training.py logs through dllogger
        sys.path.insert(0, os.path.join(REF, "Classification/ConvNets"))
        tr = importlib.import_module("image_classification.training")
        assert hasattr(tr, "Trainer") and hasattr(tr, "Executor")
        ga = importlib.import_module("image_classification.gpu_affinity")     # pynvml
        assert hasattr(ga, "set_affinity")
        # DLRM: host-side modules (the cuda_ext package itself queries the device at import: GPU only)
        sys.path.insert(0, os.path.join(REF, "Recommendation/DLRM"))
        mlps = importlib.import_module("dlrm.nn.mlps")                # apex.mlp
        assert hasattr(mlps, "CppMlp")
        du = importlib.import_module("dlrm.utils.distributed")
        assert du.get_gpu_batch_sizes(65536, 8) == (8192,) * 8
    finally:
        sys.path[:] = saved
