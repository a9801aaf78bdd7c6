"""CPU restatement of the reference's DLRM train step (TEST INFRASTRUCTURE ONLY -- never shipped/measured
except as bench.py's cpu_baseline leg).

Follows, in plain fp32 torch on the CPU (paths relative to /root/reference/PyTorch/Recommendation/DLRM/):
    dlrm/nn/mlps.py:78-114            TorchMlp: (Linear + ReLU) x L
    dlrm/nn/embeddings.py:102-137     JointEmbedding: W[idx + offsets[:-1]]
    dlrm/nn/interactions.py:65-82     DotInteraction.interact: bmm, strict-lower-tri gather, concat, zero pad
    dlrm/nn/parts.py:79-136           DlrmBottom / DlrmTop
    dlrm/scripts/main.py:585-608      BCEWithLogitsLoss(mean), backward, SGD on MLPs, sparse SGD on embeddings
Pinned by tests/golden/dlrm_step_*.npz, generated from the reference's own DistributedDlrm by
oracle/make_golden.py (gen_dlrm_step).
"""
import numpy as np
import torch
import torch.nn.functional as TF

from oracle.storage import q as q_store


def tril_index_pairs(n_vec):
    """interactions.py:50-53."""
    rows = [i for i in range(n_vec) for _ in range(i)]
    cols = [j for i in range(n_vec) for j in range(i)]
    return torch.tensor(rows, dtype=torch.long), torch.tensor(cols, dtype=torch.long)


class DlrmOracle:
    """state: dict name -> fp32 torch tensor with the reference's state_dict layout restricted to
    bottom_mlp.{i}.{weight,bias}, top_mlp.{i}.{weight,bias}, out.{weight,bias}, embedding (joint [sum N, D])."""

    def __init__(self, state, table_sizes, lr, storage_dtype=None):
        # storage_dtype: round what the AMP path keeps in 16 bits (MLP weights, the cast numerical input, gathered rows,
        # every MLP / interaction output incl. the logits, and their gradients) -- oracle/storage.py; tables stay fp32
        self.sd = storage_dtype
        self.p = {k: v.clone().float().requires_grad_(True) for k, v in state.items()}
        self.sizes = list(table_sizes)
        self.offsets = torch.tensor([0] + self.sizes, dtype=torch.long).cumsum(0)
        self.lr = lr
        self.n_bot = len([k for k in state if k.startswith("bottom_mlp.") and k.endswith(".weight")])
        self.n_top = len([k for k in state if k.startswith("top_mlp.") and k.endswith(".weight")])

    def forward(self, num, cat):
        p = self.p
        Q = lambda t: q_store(t, self.sd)
        h = Q(num)
        for i in range(self.n_bot):
            h = Q(torch.relu(TF.linear(h, Q(p[f"bottom_mlp.{i}.weight"]), p[f"bottom_mlp.{i}.bias"])))
        rows = cat + self.offsets[:-1]
        emb = Q(p["embedding"][rows])                                # [B, T, D]
        x = torch.cat([h.unsqueeze(1), emb], dim=1)                  # bottom MLP first (parts.py:97-99)
        z = torch.bmm(x, x.transpose(1, 2))
        ri, ci = tril_index_pairs(x.shape[1])
        flat = z[:, ri, ci]
        raw = flat.shape[1] + h.shape[1]
        pad = ((raw - 1) // 8 + 1) * 8 - raw
        inter = Q(torch.cat([h, flat, torch.zeros(h.shape[0], pad)], dim=1))
        t = inter
        for i in range(self.n_top):
            t = Q(torch.relu(TF.linear(t, Q(p[f"top_mlp.{i}.weight"]), p[f"top_mlp.{i}.bias"])))
        return Q(TF.linear(t, Q(p["out.weight"]), p["out.bias"])).squeeze(-1)

    def step(self, num, cat, click, lr=None):
        lr = self.lr if lr is None else lr
        for v in self.p.values():
            v.grad = None
        loss = TF.binary_cross_entropy_with_logits(self.forward(num, cat), click, reduction="mean")
        loss.backward()
        with torch.no_grad():
            for v in self.p.values():
                v -= lr * v.grad                                     # dense == sparse SGD (duplicates summed)
        return float(loss.detach())


def state_from_reference(model):
    """Flatten a reference DistributedDlrm(embedding_type='joint', use_cpp_mlp=False) into oracle keys."""
    s = {}
    for i, l in enumerate([m for m in model.bottom_model.mlp.layers if isinstance(m, torch.nn.Linear)]):
        s[f"bottom_mlp.{i}.weight"], s[f"bottom_mlp.{i}.bias"] = l.weight.detach().clone(), l.bias.detach().clone()
    for i, l in enumerate([m for m in model.top_model.mlp.layers if isinstance(m, torch.nn.Linear)]):
        s[f"top_mlp.{i}.weight"], s[f"top_mlp.{i}.bias"] = l.weight.detach().clone(), l.bias.detach().clone()
    s["out.weight"], s["out.bias"] = model.top_model.out.weight.detach().clone(), model.top_model.out.bias.detach().clone()
    s["embedding"] = model.bottom_model.embeddings.embedding.weight.detach().clone()
    return s


def load_into_hip_model(model, state):
    """Copy an oracle state into a deeplearningexamples_amd.dlrm.model.DistributedDlrm (any device)."""
    with torch.no_grad():
        for i, l in enumerate(model.bottom_model.mlp.linears):
            l.weight.copy_(state[f"bottom_mlp.{i}.weight"]); l.bias.copy_(state[f"bottom_mlp.{i}.bias"])
        for i, l in enumerate(model.top_model.mlp.linears):
            l.weight.copy_(state[f"top_mlp.{i}.weight"]); l.bias.copy_(state[f"top_mlp.{i}.bias"])
        model.top_model.out.weight.copy_(state["out.weight"]); model.top_model.out.bias.copy_(state["out.bias"])
        model.bottom_model.embeddings.weight.copy_(state["embedding"])
    model.refresh_working_copies()


def state_to_numpy(state):
    return {k: v.detach().cpu().numpy() for k, v in state.items()}


CRITEO_F15_SIZES = [7912889, 33823, 582469, 245828, 11, 2209, 10667, 104, 4, 968, 15, 8165896, 17139,
                    2675940, 7156453, 302516, 12022, 97, 35, 7339, 20046, 4, 7105, 1382, 63, 5554114]
"""tests/feature_specs/criteo_f15.yaml cardinalities (SURVEY.md section 8c)."""


def seeded_dlrm_state(sizes, dim, bottom, top, num, seed):
    """Initial weights from a numpy PCG64 stream (stable across machines), same distributions as the reference
    (mlps.py:92-96 normal init, parts.py:64-72 uniform(+-sqrt(1/size)) embeddings, zeroed pad column)."""
    rng = np.random.default_rng(seed)
    st = {}
    d = num
    for i, o in enumerate(bottom):
        st[f"bottom_mlp.{i}.weight"] = (rng.standard_normal((o, d)) * np.sqrt(2.0 / (d + o))).astype(np.float32)
        st[f"bottom_mlp.{i}.bias"] = (rng.standard_normal(o) * np.sqrt(1.0 / o)).astype(np.float32)
        d = o
    nvec = len(sizes) + 1
    raw = nvec * (nvec - 1) // 2 + dim
    d = ((raw - 1) // 8 + 1) * 8
    for i, o in enumerate(top[:-1]):
        st[f"top_mlp.{i}.weight"] = (rng.standard_normal((o, d)) * np.sqrt(2.0 / (d + o))).astype(np.float32)
        st[f"top_mlp.{i}.bias"] = (rng.standard_normal(o) * np.sqrt(1.0 / o)).astype(np.float32)
        d = o
    if raw != ((raw - 1) // 8 + 1) * 8:
        st["top_mlp.0.weight"][:, -1] = 0.0
    bound = 1.0 / np.sqrt(d)
    st["out.weight"] = rng.uniform(-bound, bound, (top[-1], d)).astype(np.float32)
    st["out.bias"] = rng.uniform(-bound, bound, (top[-1],)).astype(np.float32)
    st["embedding"] = np.concatenate(
        [rng.uniform(-np.sqrt(1.0 / s_), np.sqrt(1.0 / s_), (s_, dim)).astype(np.float32) for s_ in sizes], axis=0)
    return {k: torch.from_numpy(v) for k, v in st.items()}


def seeded_dlrm_batch(sizes, num, batch, seed):
    """Synthetic batch in the reference's SyntheticDataset shape (datasets.py:32-61) with LEARNABLE labels
    (a function of the inputs) so that the per-step losses actually move."""
    rng = np.random.default_rng(seed)
    x = rng.random((batch, num), dtype=np.float32)
    cat = np.stack([rng.integers(0, s_, batch) for s_ in sizes], axis=1).astype(np.int64)
    click = ((x[:, 0] + 0.5 * x[:, 1] + 0.25 * (cat[:, 0] % 2)) > 0.85).astype(np.float32)
    return torch.from_numpy(x), torch.from_numpy(cat), torch.from_numpy(click)


DLRM_STEP_CONFIGS = {
    "tiny": dict(sizes=[11, 4, 968, 15, 97, 35, 63, 104], dim=32, bottom=[64, 32], top=[64, 32, 1],
                 num=13, batch=256, lr=1.0, steps=10, seed=11),
    "criteo_shape": dict(sizes=[min(s, 3000) for s in CRITEO_F15_SIZES], dim=128, bottom=[512, 256, 128],
                         top=[1024, 1024, 512, 256, 1], num=13, batch=2048, lr=1.0, steps=8, seed=12),
    # every sparse-update path of the HIP engine inside ONE step that is compared with the reference's DistributedDlrm: the
    # one-hot MFMA segment sum (tables of <= 128 rows: 4, 97), the eight-lists-per-row form (<= 4096 rows: 968, 2209, the
    # capped tail), the one-list form with its LookupMap multiply-shift division, the end-first walk and emb_link (5000,
    # 20046, 200000, 1000000 rows) -- batch 8192 so that the small tables see ~10-2000 duplicates per row and the big ones few
    "mixed_paths": dict(sizes=[4, 97, 968, 2209, 5000, 20046, 200000, 1000000] + [min(s, 3000) for s in CRITEO_F15_SIZES[10:18]],
                        dim=128, bottom=[512, 256, 128], top=[1024, 1024, 512, 256, 1], num=13, batch=8192, lr=1.0, steps=4,
                        seed=13),
}
MIXED_PATHS_PROBE_TABLES = (0, 1, 2, 3, 4, 5, 6, 7)     # tables whose final rows the fixture keeps (<= 256 touched rows each)


def probe_rows(cfg, cat):
    """Rows of the joint embedding the mixed_paths fixture keeps: per probed table the first <= 256 distinct rows the batch looks
    up (in order of first appearance) -- touched rows of every update path, few enough for a small fixture."""
    off = np.concatenate([[0], np.cumsum(cfg["sizes"])]).astype(np.int64)
    out = []
    c = cat.numpy() if hasattr(cat, "numpy") else np.asarray(cat)
    for t in MIXED_PATHS_PROBE_TABLES:
        _, first = np.unique(c[:, t], return_index=True)
        ids = c[np.sort(first)[:256], t]
        out.append(ids + off[t])
    return np.concatenate(out).astype(np.int64)


