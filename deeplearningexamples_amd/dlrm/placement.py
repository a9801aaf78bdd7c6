"""Hybrid-parallel placement and batch split for DLRM (pure integer host logic, bit-exact contract).

Mirrors the results of the reference's
    get_gpu_batch_sizes / argsort / distribute_to_buckets / get_device_mapping
(Recommendation/DLRM/dlrm/utils/distributed.py:102-176) -- same names, same return shapes -- so that a
checkpoint or a rank layout produced by either side is interchangeable.  Checked against fixtures generated
by the reference itself (tests/golden/dlrm_placement.json).
"""
import bisect
import itertools
import math
from typing import Dict, List, Sequence, Tuple


def get_gpu_batch_sizes(global_batch_size: int, num_gpus: int = 4, batch_std: int = 64,
                        divisible_by: int = 64) -> Tuple[int, ...]:
    """Non-decreasing per-GPU batch sizes (multiples of `divisible_by`, within +-batch_std of the mean) that
    add up to the global batch and maximise the product; the first maximiser in enumeration order wins."""
    mean = global_batch_size // num_gpus
    allowed = [v for v in range(mean - batch_std, mean + batch_std + 1) if v % divisible_by == 0]
    winner, winner_score = None, -1
    for sizes in itertools.combinations_with_replacement(allowed, num_gpus):
        if sum(sizes) != global_batch_size:
            continue
        score = math.prod(sizes)
        if winner is None or score > winner_score:
            winner, winner_score = sizes, score
    if winner is None:
        raise RuntimeError("Could not find GPU batch sizes for a given configuration. "
                           "Please adjust global batch size or number of used GPUs.")
    return winner


def argsort(sequence: Sequence, reverse: bool = False) -> List[int]:
    """Stable argsort (ties keep input order, also when reverse=True -- Python's sort contract)."""
    return sorted(range(len(sequence)), key=sequence.__getitem__, reverse=reverse)


def distribute_to_buckets(sizes: Sequence[int], buckets_num: int) -> List[List[int]]:
    """Greedy: biggest remaining table goes to the lightest open bucket; a bucket closes when it holds
    ceil(T / buckets) tables.  Open buckets are kept ordered by load with a stable re-sort after every
    placement, which is what fixes the tie-breaking."""
    cap = math.ceil(len(sizes) / buckets_num)
    order = argsort(sizes, reverse=True)
    open_b: List[List[int]] = [[] for _ in range(buckets_num)]
    loads: List[int] = [0] * buckets_num
    done: List[List[int]] = []
    for tbl in order:
        open_b[0].append(tbl)
        loads[0] += sizes[tbl]
        if len(open_b[0]) == cap:
            done.append(open_b.pop(0))
            loads.pop(0)
        # stable sort by load == what list.sort(key=load) does
        perm = sorted(range(len(open_b)), key=loads.__getitem__)
        open_b = [open_b[i] for i in perm]
        loads = [loads[i] for i in perm]
    return done + open_b


def get_device_mapping(embedding_sizes: Sequence[int], num_gpus: int = 8) -> Dict:
    """{'bottom_mlp': 0, 'embedding': [tables of rank 0, ...], 'vectors_per_gpu': [...]}.  With more than
    four GPUs rank 0 keeps only the bottom MLP."""
    if num_gpus > 4:
        buckets = [[]] + distribute_to_buckets(embedding_sizes, num_gpus - 1)
    else:
        buckets = distribute_to_buckets(embedding_sizes, num_gpus)
    vectors = [len(b) for b in buckets]
    vectors[0] += 1
    return {"bottom_mlp": 0, "embedding": buckets, "vectors_per_gpu": vectors}


class ExchangePlan:
    """Split sizes and feature permutation of the bottom->top all-to-all for one rank
    (dlrm/model/distributed.py:32-98 BottomToTop).

    Forward: rank r holds its bottom output for the GLOBAL batch, [sum(B_p), n_r, D]; peer p receives the
    slice of its B_p samples.  The receive buffer is the concatenation over sources s of [B_r, n_s, D]
    blocks; `recv_feature_base[s]` is the first feature slot of source s in the interaction input
    [B_r, sum(n_s), D] (device order: bottom MLP first, then rank 0's tables, rank 1's, ...).
    """

    def __init__(self, batch_sizes_per_gpu: Sequence[int], vectors_per_gpu: Sequence[int], dim: int, rank: int):
        self.world = len(batch_sizes_per_gpu)
        if len(vectors_per_gpu) != self.world:
            raise ValueError("batch_sizes_per_gpu and vectors_per_gpu must have one entry per rank")
        self.rank, self.dim = rank, dim
        self.batch_sizes = list(batch_sizes_per_gpu)
        self.vectors = list(vectors_per_gpu)
        self.local_batch = self.batch_sizes[rank]
        self.global_batch = sum(self.batch_sizes)
        self.n_local = self.vectors[rank]
        self.n_total = sum(self.vectors)
        self.batch_start = [0] + list(itertools.accumulate(self.batch_sizes))
        # forward: send [B_p, n_r, D] to p, receive [B_r, n_s, D] from s   (element counts)
        self.fwd_send_splits = [b * self.n_local * dim for b in self.batch_sizes]
        self.fwd_recv_splits = [self.local_batch * n * dim for n in self.vectors]
        self.recv_feature_base = [0] + list(itertools.accumulate(self.vectors))[:-1]
        self.recv_block_start = [0] + list(itertools.accumulate(self.fwd_recv_splits))[:-1]

    def source_of_feature(self, slot: int) -> int:
        return bisect.bisect_right(self.recv_feature_base, slot) - 1


class RowShardPlan:
    """Row-wise placement of EVERY embedding table over the ranks (BASELINE.json configs[3] "embedding tables row-sharded over 8
    GPUs"; SURVEY.md 8(e): "index bucketing by row range + a2a of indices and vectors").  The reference itself places whole tables
    (get_device_mapping above); this is the opt-in alternative (`--embedding_sharding row`).

    Table t (N_t rows) is cut into `world` contiguous ranges of shard_t = ceil(N_t / world) rows; rank r owns rows
    [r * shard_t, min(N_t, (r + 1) * shard_t)) and keeps its ranges of all tables stacked in table order in ONE joint matrix
    (local_offsets[r][t] = first joint row of table t on rank r).  Pure integer logic; `route` works on any torch device."""

    def __init__(self, table_sizes: Sequence[int], world: int):
        if world < 1 or any(n < 1 for n in table_sizes):
            raise ValueError("RowShardPlan: world >= 1 and non-empty tables expected")
        self.sizes, self.world = [int(n) for n in table_sizes], int(world)
        self.shard = [-(-n // world) for n in self.sizes]
        self.local_sizes = [[max(0, min(n - r * s, s)) for n, s in zip(self.sizes, self.shard)] for r in range(world)]
        self.local_offsets = [[0] + list(itertools.accumulate(ls))[:-1] for ls in self.local_sizes]
        self.local_rows = [sum(ls) for ls in self.local_sizes]

    def rows_of(self, rank: int, table: int) -> Tuple[int, int]:
        """[first, last) GLOBAL row of `table` that `rank` owns."""
        lo = rank * self.shard[table]
        return min(lo, self.sizes[table]), min(lo + self.shard[table], self.sizes[table])

    def tensors(self, device):
        """(shard [T], local_offsets [world, T]) as int64 tensors on `device` (the operands of route())."""
        import torch
        return (torch.tensor(self.shard, dtype=torch.int64, device=device),
                torch.tensor(self.local_offsets, dtype=torch.int64, device=device))

    @staticmethod
    def route(ids, shard, local_offsets):
        """ids int64 [B, T] (0 <= ids[:, t] < N_t) -> (owner [B, T], joint row on the owner [B, T]), both int64."""
        import torch
        owner = torch.div(ids, shard, rounding_mode="floor")
        t_idx = torch.arange(ids.shape[1], device=ids.device).expand_as(ids)
        row = ids - owner * shard + local_offsets[owner, t_idx]
        return owner, row

    @staticmethod
    def bucket(owner, world):
        """owner int64 [n] -> (order, counts): a STABLE permutation that groups the lookups by owner rank (lookups bound for the
        same rank keep their (sample, table) order) and the number of lookups per rank (int64 [world])."""
        import torch
        order = torch.argsort(owner, stable=True)
        counts = torch.bincount(owner, minlength=world)
        return order, counts
