"""DLRM split-binary dataset (dlrm/data.py) on CPU: round trip, the reference's yaml fixtures, and -- with
the reference tree mounted -- batches identical to the reference's own ParametricDataset on the same files."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeplearningexamples_amd.dlrm import data as D  # noqa: E402
from oracle import _ref_import as R  # noqa: E402


def _restore(mods, path, saved):
    """Undo oracle/_ref_import.import_dlrm: the reference's modules leave sys.modules, the patched torch entry points come back."""
    torch.cuda.current_device, torch.Tensor.cuda, torch.cuda.synchronize = saved
    for k in list(sys.modules):
        if k not in mods and not k.startswith(("torch", "numpy", "scipy", "_pytest", "pytest")):
            del sys.modules[k]
    for k, v in mods.items():
        sys.modules[k] = v
    sys.path[:] = path


def _make(tmp_path, rows=1000, sizes=(7, 300, 40000, 5)):
    spec = D.FeatureSpec.get_default_feature_spec(13, list(sizes))
    spec.base_directory = str(tmp_path)
    rng = np.random.default_rng(0)
    num = rng.random((rows, 13)).astype(np.float16)
    cat = np.stack([rng.integers(0, s, rows) for s in sizes], axis=1)
    lab = rng.integers(0, 2, rows).astype(bool)
    for m in (D.TRAIN_MAPPING, D.TEST_MAPPING):
        D.write_split_binary(spec, m, num, cat, lab)
    spec.to_yaml()
    return spec, num, cat, lab


def test_split_binary_round_trip(tmp_path):
    spec, num, cat, lab = _make(tmp_path)
    assert [spec.feature_spec[n]["dtype"] for n in spec.get_categorical_feature_names()] == ["int8", "int16", "int32", "int8"]
    again = D.FeatureSpec.from_yaml(str(tmp_path / "feature_spec.yaml"))
    assert again.to_dict() == spec.to_dict() and again.get_categorical_sizes() == [7, 300, 40000, 5]
    order = ["cat_2.bin", "cat_0.bin"]                                    # a rank's own tables, in its device order
    ds = D.ParametricDataset(again, "train", batch_size=128, numerical_features_enabled=True,
                             categorical_features_to_read=order, prefetch_depth=3)
    assert len(ds) == 8
    got = list(ds)
    assert [b[2].shape[0] for b in got] == [128] * 7 + [104]
    assert torch.equal(torch.cat([b[0] for b in got]), torch.from_numpy(num))
    assert torch.equal(torch.cat([b[1] for b in got]), torch.from_numpy(cat[:, [2, 0]]))
    assert got[0][1].dtype == torch.int64 and got[0][2].dtype == torch.float32
    assert torch.equal(torch.cat([b[2] for b in got]), torch.from_numpy(lab).float())
    dropped = D.ParametricDataset(again, "test", batch_size=128, drop_last_batch=True)
    assert len(dropped) == 7 and dropped[0][0] is None and dropped[0][1] is None
    with pytest.raises(IndexError):
        dropped[7]


@pytest.mark.skipif(not R.have_reference(), reason="reference tree not mounted")
def test_same_batches_as_the_reference_dataset(tmp_path):
    mods, path = dict(sys.modules), list(sys.path)
    saved = (torch.cuda.current_device, torch.Tensor.cuda, torch.cuda.synchronize)
    try:
        R.import_dlrm()
        from dlrm.data.datasets import ParametricDataset as RefDataset
        from dlrm.data.feature_spec import FeatureSpec as RefSpec
        spec, num, cat, lab = _make(tmp_path, rows=777)
        rspec = RefSpec.from_yaml(str(tmp_path / "feature_spec.yaml"))
        assert rspec.to_dict() == spec.to_dict()
        # the reference's own fixture parses to the same dictionaries
        fx = os.path.join(R.REF, "PyTorch/Recommendation/DLRM/tests/feature_specs/criteo_f15.yaml")
        assert D.FeatureSpec.from_yaml(fx).get_categorical_sizes() == RefSpec.from_yaml(fx).get_categorical_sizes()
        order = ["cat_1.bin", "cat_3.bin", "cat_0.bin"]
        kw = dict(mapping="train", batch_size=100, numerical_features_enabled=True, categorical_features_to_read=order,
                  prefetch_depth=1)
        ours, ref = D.ParametricDataset(spec, **kw), RefDataset(rspec, **kw)
        assert len(ours) == len(ref) == 8
        for i in range(len(ref)):
            (n0, c0, l0), (n1, c1, l1) = ours[i], ref[i]
            assert torch.equal(n0, n1) and torch.equal(c0, c1) and torch.equal(l0, l1), i
    finally:
        _restore(mods, path, saved)


FSPEC_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "feature_specs")
FSPECS = sorted(f for f in os.listdir(FSPEC_DIR) if f.endswith(".yaml"))


def synth_from_fixture(name, out_dir, rows, seed=0):
    """dlrm/scripts/prepare_synthetic_dataset.py on one of the reference's feature-spec fixtures (DLRM/tests/feature_specs/*.yaml,
    committed as data under tests/golden/feature_specs/): uniform ids below each cardinality, uniform numericals, coin-flip labels,
    written where the spec's source_spec says.  -> (spec, numerical, categorical, labels)."""
    import shutil
    os.makedirs(out_dir, exist_ok=True)
    shutil.copy(os.path.join(FSPEC_DIR, name), os.path.join(out_dir, "feature_spec.yaml"))
    spec = D.FeatureSpec.from_yaml(os.path.join(out_dir, "feature_spec.yaml"))
    spec.check_feature_spec()
    rng = np.random.default_rng(seed)
    sizes = spec.get_categorical_sizes()
    num = rng.random((rows, spec.get_number_of_numerical_features())).astype(np.float16)
    cat = np.stack([rng.integers(0, s, rows) for s in sizes], axis=1)
    lab = rng.integers(0, 2, rows).astype(bool)
    for m in (D.TRAIN_MAPPING, D.TEST_MAPPING):
        D.write_split_binary(spec, m, num, cat, lab)
    return spec, num, cat, lab


@pytest.mark.parametrize("name", FSPECS)
def test_reference_feature_spec_fixtures_round_trip(tmp_path, name):
    """Every feature spec the reference's own DLRM tests train on (tests/test_fspecs.sh, test_all_configs.sh: 10 / 13 / 20
    numerical features, 10 / 26 / 30 tables, other feature names and file paths, int8 .. int64 storage): parsed, validated, a
    synthetic dataset written in its layout and read back batch by batch."""
    spec, num, cat, lab = synth_from_fixture(name, str(tmp_path), rows=300)
    assert len(FSPECS) == 10
    names = spec.get_categorical_feature_names()
    assert cat.shape[1] == len(names) and num.shape[1] == spec.get_number_of_numerical_features()
    ds = D.ParametricDataset(spec, "test", batch_size=128, numerical_features_enabled=True, categorical_features_to_read=names)
    got = list(ds)
    assert [b[2].shape[0] for b in got] == [128, 128, 44]
    assert torch.equal(torch.cat([b[0] for b in got]), torch.from_numpy(num))
    assert torch.equal(torch.cat([b[1] for b in got]), torch.from_numpy(cat))
    assert torch.equal(torch.cat([b[2] for b in got]), torch.from_numpy(lab).float())
    for i, n in enumerate(names):                     # each table's file holds the storage type the spec names
        _, _, paths = spec.get_mapping_paths("train")
        assert os.path.getsize(paths[n]) == 300 * np.dtype(spec.feature_spec[n]["dtype"]).itemsize
    if R.have_reference():                            # the reference's classes read the same files to the same batches
        mods, path = dict(sys.modules), list(sys.path)
        saved = (torch.cuda.current_device, torch.Tensor.cuda, torch.cuda.synchronize)
        try:
            R.import_dlrm()
            from dlrm.data.datasets import ParametricDataset as RefDataset
            from dlrm.data.feature_spec import FeatureSpec as RefSpec
            rspec = RefSpec.from_yaml(str(tmp_path / "feature_spec.yaml"))
            rspec.check_feature_spec()
            assert rspec.get_categorical_sizes() == spec.get_categorical_sizes()
            assert rspec.get_number_of_numerical_features() == spec.get_number_of_numerical_features()
            ref = RefDataset(rspec, mapping="test", batch_size=128, numerical_features_enabled=True,
                             categorical_features_to_read=names, prefetch_depth=1)
            assert len(ref) == len(ds)
            for i in range(len(ref)):
                (n0, c0, l0), (n1, c1, l1) = ds[i], ref[i]
                assert torch.equal(n0, n1) and torch.equal(c0, c1) and torch.equal(l0, l1), (name, i)
        finally:
            _restore(mods, path, saved)
