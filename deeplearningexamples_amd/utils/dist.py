"""One process per GPU, rendezvous from the environment (the reference's launchers export the same variables:
ConvNets/multiproc.py:148-210, torch.distributed.launch for BERT / DLRM, dlrm/utils/distributed.py:75-99)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend="nccl"):
    rank = int(os.environ.get("RANK", os.environ.get("OMPI_COMM_WORLD_RANK", "0")))
    world = int(os.environ.get("WORLD_SIZE", os.environ.get("OMPI_COMM_WORLD_SIZE", "1")))
    local = int(os.environ.get("LOCAL_RANK", os.environ.get("OMPI_COMM_WORLD_LOCAL_RANK", "0")))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, init_method="env://", **kw)
    return rank, world, local


def is_main_process():
    return (not dist.is_initialized()) or dist.get_rank() == 0
