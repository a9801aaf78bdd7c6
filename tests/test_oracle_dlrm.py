"""Oracle vs the reference's own outputs (fixtures made by oracle/make_golden.py). CPU only."""
import json
import os

import numpy as np

from oracle import dlrm_oracle as O


def _load(golden_dir, name):
    return json.load(open(os.path.join(golden_dir, name)))


def test_device_mapping_bit_exact(golden_dir):
    g = _load(golden_dir, "dlrm_placement.json")
    assert len(g["device_mapping"]) > 20
    for case in g["device_mapping"]:
        got = O.device_mapping(case["sizes"], case["num_gpus"])
        assert got == case["result"], (case["name"], case["num_gpus"])


def test_survey_kat_criteo_8gpu():
    # SURVEY.md section 8(c) known-answer vector computed from the reference
    sizes = [7912889, 33823, 582469, 245828, 11, 2209, 10667, 104, 4, 968, 15, 8165896, 17139,
             2675940, 7156453, 302516, 12022, 97, 35, 7339, 20046, 4, 7105, 1382, 63, 5554114]
    m = O.device_mapping(sizes, 8)
    assert m["embedding"] == [[], [15, 3, 1, 20], [2, 12, 16, 6], [13, 19, 22, 5], [25, 23, 9, 7],
                              [14, 17, 24, 18], [0, 10, 4, 8], [11, 21]]
    assert m["vectors_per_gpu"] == [1, 4, 4, 4, 4, 4, 4, 2]
    assert O.gpu_batch_sizes(65536, 8) == (8192,) * 8


def test_gpu_batch_sizes_and_argsort(golden_dir):
    g = _load(golden_dir, "dlrm_placement.json")
    for case in g["gpu_batch_sizes"]:
        try:
            got = list(O.gpu_batch_sizes(case["global_batch"], case["num_gpus"]))
        except RuntimeError:
            got = None
        assert got == case["result"], case
    for case in g["argsort"]:
        assert O.stable_argsort(case["seq"]) == case["asc"]
        assert O.stable_argsort(case["seq"], True) == case["desc"]


def test_tril_and_padding(golden_dir):
    g = _load(golden_dir, "dlrm_placement.json")
    for nv, pairs in g["tril"].items():
        r, c = O.tril_pairs(int(nv))
        assert r.tolist() == pairs[0] and c.tolist() == pairs[1]
        assert O.interact_out_width(int(nv), 128) == g["padding"][nv]["num_interactions"]


def test_dot_interact_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "dlrm_dot_interact.npz"))
    keys = sorted({k.rsplit("_", 1)[0] for k in z.files if k.endswith("_x")})
    assert keys
    for k in keys:
        x, y, ug, gx = z[k + "_x"], z[k + "_y"], z[k + "_ug"], z[k + "_gx_total"]
        np.testing.assert_allclose(O.dot_interact_fwd(x), y, rtol=1e-5, atol=1e-5)
        grad, mlp = O.dot_interact_bwd(x, ug)
        total = grad.copy()
        total[:, 0, :] += mlp        # autograd sums both paths onto row 0
        np.testing.assert_allclose(total, gx, rtol=1e-4, atol=1e-4)


def test_embedding_index_math_bit_exact(golden_dir):
    z = np.load(os.path.join(golden_dir, "dlrm_embedding.npz"))
    sizes = z["sizes"].tolist()
    off = O.table_offsets(sizes)
    assert off.dtype == np.int64 and (off == z["offsets"]).all()
    hashed = O.hash_indices(z["idx_in"], sizes)
    assert (hashed == z["idx_hashed"]).all()
    rows = O.offset_indices(hashed, off)
    out = O.embedding_gather(z["w0"], rows)
    assert (out == z["out"]).all()          # pure copy: bit exact
    w1 = O.sparse_sgd(z["w0"], rows, z["ug"], float(z["lr"]))
    np.testing.assert_allclose(w1, z["w1"], rtol=1e-6, atol=1e-6)
