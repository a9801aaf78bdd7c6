"""Input side of the DLRM trainer: the reference's split-binary dataset.

Mirrors Recommendation/DLRM/dlrm/data/:
    feature_spec.py:31-253  FeatureSpec: feature_spec.yaml = {feature_spec: name -> {dtype, cardinality}, source_spec:
                            mapping (train / test) -> chunks {type: split_binary, features: [...], files: [...]},
                            channel_spec: {numerical, categorical, label} -> names, metadata}
    datasets.py:64-223      ParametricDataset: one raw file per chunk -- numerical.bin fp16 [rows, n_num], label.bin bool
                            [rows], one file per categorical feature in its smallest integer type; item = one BATCH:
                            (fp16 [B, n_num] or None, int64 [B, n_cat] or None, fp32 [B]), read ahead by a worker thread
    utils.py:129-145        prefetcher: host -> device copies on a side stream, one batch ahead
Written for this package (memory-mapped files, a bounded queue); the file format and the yaml schema are the reference's,
checked against its own classes in tests/test_dlrm_data.py.
"""
import math
import os
import queue
import threading
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import yaml

CATEGORICAL_CHANNEL, NUMERICAL_CHANNEL, LABEL_CHANNEL = "categorical", "numerical", "label"
SPLIT_BINARY, TRAIN_MAPPING, TEST_MAPPING = "split_binary", "train", "test"


def categorical_dtype(cardinality: int):
    """Smallest signed integer type whose maximum exceeds the cardinality (dlrm/data/defaults.py:34-43)."""
    for t in (np.int8, np.int16, np.int32):
        if cardinality < np.iinfo(t).max:
            return t
    raise RuntimeError("categorical feature of size %d is too big for the defined types" % cardinality)


class FeatureSpec:
    def __init__(self, feature_spec=None, source_spec=None, channel_spec=None, metadata=None, base_directory=None):
        self.feature_spec: Dict = feature_spec or {}
        self.source_spec: Dict = source_spec or {}
        self.channel_spec: Dict = channel_spec or {}
        self.metadata: Dict = metadata or {}
        self.base_directory = base_directory

    @classmethod
    def from_yaml(cls, path):
        with open(path) as f:
            return cls(base_directory=os.path.dirname(path), **yaml.safe_load(f))

    def to_dict(self):
        return {k: getattr(self, k) for k in ("feature_spec", "source_spec", "channel_spec", "metadata")}

    def to_yaml(self, output_path=None):
        with open(output_path or os.path.join(self.base_directory, "feature_spec.yaml"), "w") as f:
            f.write(yaml.dump(self.to_dict()))

    def get_number_of_numerical_features(self):
        return len(self.channel_spec[NUMERICAL_CHANNEL])

    def get_categorical_feature_names(self):
        return self.channel_spec[CATEGORICAL_CHANNEL]

    def get_categorical_sizes(self) -> List[int]:
        return [self.feature_spec[n]["cardinality"] for n in self.get_categorical_feature_names()]

    def check_feature_spec(self):
        if sorted(self.source_spec) != sorted([TEST_MAPPING, TRAIN_MAPPING]):
            raise ValueError("source_spec must hold exactly the train and test mappings")
        if sorted(self.channel_spec) != sorted([CATEGORICAL_CHANNEL, NUMERICAL_CHANNEL, LABEL_CHANNEL]):
            raise ValueError("channel_spec must hold exactly the numerical, categorical and label channels")
        names = [n for ch in self.channel_spec.values() for n in ch]
        for mapping in self.source_spec.values():
            seen = [f for chunk in mapping for f in chunk["features"]]
            if sorted(seen) != sorted(names):
                raise ValueError("every mapping must cover every feature of the channel spec exactly once")
            for chunk in mapping:
                if chunk["type"] != SPLIT_BINARY or len(chunk["files"]) != 1:
                    raise ValueError("chunks are split_binary with one file each")

    @staticmethod
    def get_default_feature_spec(number_of_numerical_features, categorical_feature_cardinalities):
        nums = ["num_%d" % i for i in range(number_of_numerical_features)]
        cats = ["cat_%d.bin" % i for i in range(len(categorical_feature_cardinalities))]
        feats = {n: {"dtype": str(np.dtype(categorical_dtype(int(c)))), "cardinality": c}
                 for n, c in zip(cats, categorical_feature_cardinalities)}
        feats.update({n: {"dtype": "float16"} for n in nums})
        feats["label"] = {"dtype": "bool"}
        source = {}
        for m in (TRAIN_MAPPING, TEST_MAPPING):
            source[m] = [{"type": SPLIT_BINARY, "features": nums, "files": [os.path.join(m, "numerical.bin")]},
                         {"type": SPLIT_BINARY, "features": ["label"], "files": [os.path.join(m, "label.bin")]}]
            source[m] += [{"type": SPLIT_BINARY, "features": [n], "files": [os.path.join(m, n)]} for n in cats]
        return FeatureSpec(feats, source, {CATEGORICAL_CHANNEL: cats, NUMERICAL_CHANNEL: nums, LABEL_CHANNEL: ["label"]}, {})

    def get_mapping_paths(self, mapping_name):
        label, num, cats = None, None, {}
        for chunk in self.source_spec[mapping_name]:
            path, first = os.path.join(self.base_directory, chunk["files"][0]), chunk["features"][0]
            if first in self.channel_spec[NUMERICAL_CHANNEL]:
                num = path
            elif first in self.channel_spec[CATEGORICAL_CHANNEL]:
                cats[first] = path
            elif first == self.channel_spec[LABEL_CHANNEL][0]:
                label = path
        return label, num, cats


def write_split_binary(spec: FeatureSpec, mapping: str, numerical: np.ndarray, categorical: np.ndarray, labels: np.ndarray):
    """Write one mapping of a dataset in the split-binary format (rows = samples; categorical columns in channel order)."""
    label_path, num_path, cat_paths = spec.get_mapping_paths(mapping)
    for p in [label_path, num_path] + list(cat_paths.values()):
        os.makedirs(os.path.dirname(p), exist_ok=True)
    np.asarray(numerical, dtype=np.float16).tofile(num_path)
    np.asarray(labels, dtype=bool).tofile(label_path)
    for i, name in enumerate(spec.get_categorical_feature_names()):
        np.asarray(categorical[:, i]).astype(spec.feature_spec[name]["dtype"]).tofile(cat_paths[name])


class ParametricDataset:
    """dataset[i] = batch i.  categorical_features_to_read dictates the order of the returned columns (a rank reads the
    tables it owns, in its device order)."""

    def __init__(self, feature_spec: FeatureSpec, mapping: str, batch_size: int = 1, numerical_features_enabled: bool = False,
                 categorical_features_to_read: Optional[Sequence[str]] = None, prefetch_depth: int = 10,
                 drop_last_batch: bool = False, **kwargs):
        feature_spec.check_feature_spec()
        self._batch = batch_size
        label_path, num_path, cat_paths = feature_spec.get_mapping_paths(mapping)
        self._n_num = feature_spec.get_number_of_numerical_features()
        self._label = np.memmap(label_path, dtype=bool, mode="r")
        rows = self._label.shape[0]
        self._num = None
        if numerical_features_enabled:
            self._num = np.memmap(num_path, dtype=np.float16, mode="r").reshape(-1, self._n_num)
            if self._num.shape[0] != rows:
                raise ValueError("Size mismatch in data files")
        self._cats = []
        for name in (categorical_features_to_read or []):
            a = np.memmap(cat_paths[name], dtype=feature_spec.feature_spec[name]["dtype"], mode="r")
            if a.shape[0] != rows:
                raise ValueError("Size mismatch in data files")
            self._cats.append(a)
        n = rows / batch_size
        self._len = math.floor(n) if drop_last_batch else math.ceil(n)
        self._rows = rows
        self._depth = max(1, min(prefetch_depth, self._len))

    def __len__(self):
        return self._len

    def _get_item(self, idx):
        lo, hi = idx * self._batch, min((idx + 1) * self._batch, self._rows)
        click = torch.from_numpy(np.array(self._label[lo:hi])).to(torch.float32)
        num = torch.from_numpy(np.array(self._num[lo:hi])) if self._num is not None else None
        cat = None
        if self._cats:
            cat = torch.from_numpy(np.stack([np.asarray(a[lo:hi]).astype(np.int64) for a in self._cats], axis=1))
        return num, cat, click

    def __getitem__(self, idx):
        if idx >= self._len:
            raise IndexError()
        return self._get_item(idx)

    def __iter__(self):
        """Sequential pass with a reader thread `prefetch_depth` batches ahead."""
        q: "queue.Queue" = queue.Queue(maxsize=self._depth)

        def reader():
            for i in range(self._len):
                q.put(self._get_item(i))
            q.put(None)
        threading.Thread(target=reader, daemon=True).start()
        while True:
            item = q.get()
            if item is None:
                return
            yield item


def prefetcher(load_iterator, device, stream: Optional[torch.cuda.Stream] = None):
    """Move the batches of an iterator to `device` on a side stream, one batch ahead of the consumer (data/utils.py:129-145)."""
    stream = stream or torch.cuda.Stream(device=device)

    def to_dev(batch):
        with torch.cuda.stream(stream):
            return tuple(t.pin_memory().to(device, non_blocking=True) if t is not None else None for t in batch)
    it = iter(load_iterator)
    try:
        nxt = to_dev(next(it))
    except StopIteration:
        return
    while True:
        torch.cuda.current_stream(device).wait_stream(stream)
        cur = nxt
        for t in cur:
            if t is not None:
                t.record_stream(torch.cuda.current_stream(device))
        try:
            nxt = to_dev(next(it))
        except StopIteration:
            yield cur
            return
        yield cur
