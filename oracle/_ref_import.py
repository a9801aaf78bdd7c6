"""Import the reference's own Python modules on CPU (build container only).

TEST INFRASTRUCTURE.  /root/reference does not exist on the GPU box, so nothing in
tests/, smoke() or bench.py calls this at run time; it is used by
oracle/make_golden.py to generate the committed fixtures under tests/golden/.
Recipe follows SURVEY.md section 8(c): stub absent third-party imports, three patches.
"""
import os
import sys
import types

REF = os.environ.get("DLE_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def have_reference():
    return os.path.isdir(os.path.join(REF, "PyTorch"))


def import_dlrm():
    import torch
    root = os.path.join(REF, "PyTorch/Recommendation/DLRM")
    if root not in sys.path:
        sys.path.insert(0, root)
    # absent: absl, apex, compiled dlrm.cuda_ext.* 
    logging = _stub("absl.logging", WARNING=30, log_first_n=lambda *a, **k: None,
                    warning=lambda *a, **k: None)
    _stub("absl", logging=logging, app=types.SimpleNamespace(), flags=types.SimpleNamespace())
    mlp = _stub("apex.mlp", MLP=torch.nn.Module, MlpFunction=torch.autograd.Function)
    _stub("apex", mlp=mlp)
    fge = _stub("dlrm.cuda_ext.fused_gather_embedding", BuckleEmbeddingFusedGatherFunction=None)
    import dlrm  # noqa: the real package (pure python part)
    ce = _stub("dlrm.cuda_ext", dotBasedInteract=None, fused_gather_embedding=fge,
               JointSparseEmbedding=None)
    dlrm.cuda_ext = ce
    # interactions.py:53 calls .cuda() unconditionally
    torch.Tensor.cuda = lambda self, *a, **k: self
    from dlrm.utils import distributed as dist_utils
    from dlrm.nn import interactions, embeddings, mlps, parts
    from dlrm.model import distributed as model_dist
    return types.SimpleNamespace(dist_utils=dist_utils, interactions=interactions,
                                 embeddings=embeddings, mlps=mlps, parts=parts,
                                 model=model_dist)


def import_bert():
    import torch
    import torch.nn.functional as F
    root = os.path.join(REF, "PyTorch/LanguageModeling/BERT")
    if root not in sys.path:
        sys.path.insert(0, root)
    exc = _stub("botocore.exceptions", ClientError=Exception)
    _stub("botocore", exceptions=exc)
    _stub("boto3")
    import modeling
    # modeling.py:122 passes approximate=True (bool), rejected by torch>=2; semantics = "tanh"
    modeling.ACT2FN["gelu"] = lambda x: F.gelu(x, approximate="tanh")
    import schedulers
    return types.SimpleNamespace(modeling=modeling, schedulers=schedulers)


def import_convnets():
    import torch
    root = os.path.join(REF, "PyTorch/Classification/ConvNets")
    if root not in sys.path:
        sys.path.insert(0, root)
    verb = types.SimpleNamespace(DEFAULT=0, VERBOSE=1)
    _stub("dllogger", Verbosity=verb, init=lambda *a, **k: None, log=lambda *a, **k: None,
          metadata=lambda *a, **k: None, flush=lambda *a, **k: None,
          StdOutBackend=object, JSONStreamBackend=object)
    torch.cuda.synchronize = lambda *a, **k: None          # training.py:181 unconditional
    from image_classification import models
    from image_classification import training, optimizers, smoothing
    return types.SimpleNamespace(models=models, training=training, optimizers=optimizers,
                                 smoothing=smoothing)


def import_fused_lamb():
    """The reference's FusedLAMBAMP class, unmodified, importable on CPU: `fused_lamb_CUDA` is bound to
    oracle/lamb_cpu_ext.py, apex's multi_tensor_applier to its two-line Python (apex/multi_tensor_apply/
    multi_tensor_apply.py: op(chunk_size, noop_flag, tensor_lists, *args), chunk 2048*32), amp_C to the same
    functions, torch.cuda.current_device() to "cpu" (fused_lamb.py:23-24 builds lr/step on that device)."""
    import torch
    from oracle import lamb_cpu_ext
    root = os.path.join(REF, "PyTorch/LanguageModeling/BERT/lamb_amp_opt")
    if root not in sys.path:
        sys.path.insert(0, root)

    class _Applier:
        available = True

        def __init__(self, chunk_size):
            self.chunk_size = chunk_size

        def __call__(self, op, noop_flag_buffer, tensor_lists, *args):
            return op(self.chunk_size, noop_flag_buffer, tensor_lists, *args)

    mta = _stub("apex.multi_tensor_apply", multi_tensor_applier=_Applier(2048 * 32), MultiTensorApply=_Applier)
    _stub("apex", multi_tensor_apply=mta)
    _stub("amp_C", multi_tensor_l2norm=lamb_cpu_ext.multi_tensor_l2norm, multi_tensor_lamb=lamb_cpu_ext.multi_tensor_lamb)
    _stub("fused_lamb_CUDA", multi_tensor_l2norm=lamb_cpu_ext.multi_tensor_l2norm,
          multi_tensor_lamb=lamb_cpu_ext.multi_tensor_lamb)
    torch.cuda.current_device = lambda: "cpu"
    sys.modules.pop("fused_lamb", None)
    sys.modules.pop("fused_lamb.fused_lamb", None)
    from fused_lamb.fused_lamb import FusedLAMBAMP
    return FusedLAMBAMP


def import_waveglow():
    """PyTorch/SpeechSynthesis/Tacotron2/waveglow/{model,loss_function}.py, loaded by file path: the
    package directory shares its name (`waveglow`) with nothing in this repo, but its parent also holds `tacotron2` / `common`
    packages that are not needed here."""
    import importlib.util
    base = os.path.join(REF, "PyTorch", "SpeechSynthesis", "Tacotron2", "waveglow")
    out = {}
    for nm in ("model", "loss_function"):
        spec = importlib.util.spec_from_file_location("_ref_waveglow_" + nm, os.path.join(base, nm + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        out[nm] = mod
    return types.SimpleNamespace(**out)


def import_tacotron2():
    """PyTorch/SpeechSynthesis/Tacotron2/tacotron2/{model,loss_function}.py (Tacotron2 half).  `librosa` (mel
    filter bank / STFT helpers of tacotron2_common.layers, never touched by the model) is absent here: stubbed."""
    import importlib.util
    root = os.path.join(REF, "PyTorch", "SpeechSynthesis", "Tacotron2")
    if root not in sys.path:
        sys.path.insert(0, root)
    util = _stub("librosa.util", pad_center=None, tiny=None, normalize=None)
    filt = _stub("librosa.filters", mel=None)
    _stub("librosa", util=util, filters=filt)
    out = {}
    for nm in ("model", "loss_function"):
        spec = importlib.util.spec_from_file_location("_ref_tacotron2_" + nm, os.path.join(root, "tacotron2", nm + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        out[nm] = mod
    return types.SimpleNamespace(**out)


def import_tacotron2_frontend():
    """The host-side input pipeline of PyTorch/SpeechSynthesis/Tacotron2: tacotron2.text.text_to_sequence,
    tacotron2.data_function.TextMelCollate, tacotron2_common.stft.STFT.  Absent third-party packages are stubbed: `inflect`
    (numbers are never spelled out in the fixtures), `librosa` (pad_center restated: centre zero padding; the mel filter bank is
    NOT provided -- nothing pinned here touches it)."""
    import numpy as np
    root = os.path.join(REF, "PyTorch", "SpeechSynthesis", "Tacotron2")
    if root not in sys.path:
        sys.path.insert(0, root)

    def pad_center(data, size, axis=-1, **kw):
        n = data.shape[axis]
        lpad = (size - n) // 2
        widths = [(0, 0)] * data.ndim
        widths[axis] = (lpad, size - n - lpad)
        return np.pad(data, widths)

    util = _stub("librosa.util", pad_center=pad_center, tiny=lambda x: np.finfo(np.float32).tiny, normalize=None)
    filt = _stub("librosa.filters", mel=None)
    _stub("librosa", util=util, filters=filt)
    _stub("inflect", engine=lambda: None)
    import importlib
    text = importlib.import_module("tacotron2.text")
    stft = importlib.import_module("tacotron2_common.stft")
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ref_tacotron2_data_function", os.path.join(root, "tacotron2", "data_function.py"))
    data = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(data)
    return types.SimpleNamespace(text=text, stft=stft, data_function=data)
