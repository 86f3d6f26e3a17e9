"""Multi-GPU sharding of the restoration path (one process per GPU, RCCL via
torch.distributed).  The reference has no distributed code; images are
independent units (GroupNorm is per-sample, pnpflow/models.py:33-38; batches are
independent, pnpflow/methods/pnp_flow.py:71), so the global batch is split contiguously,
weights are replicated, and the ONLY collective on the data path is the final all_gather
of per-image PSNR (reproducing the reference's averaging order, pnpflow/utils.py:628-656).

To stay equal to a single-device run at the global batch size, the three batch-shaped
random draws of the reference are taken for the GLOBAL batch and sliced:
  measurement noise  torch.manual_seed(batch); randn(global shape)   (pnp_flow.py:79-80)
  random mask        RandomState(42).binomial(global B, H, W)          (utils.py:357-359)
  interpolation noise  one flat Philox stream per (iteration, sample); shard s uses the
                       elements [lo*n, hi*n) of it (pf_pnp_params.elem_offset; PNP_FLOW.image_offset = lo).
"""
from __future__ import annotations

from typing import Tuple

import torch


def rank_world(group=None) -> Tuple[int, int]:
    """(rank, world size) of the running job; (0, 1) when torch.distributed is not initialised."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def force_collectives() -> bool:
    import os
    return os.environ.get("PNPFLOW_DIST_FORCE", "0") == "1"


def init_from_env(device_index=None):
    """torchrun entry: joins the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (backend "nccl" = RCCL on ROCm,
    one process per GPU) and returns (rank, world, local_rank).  A plain `python main.py` run returns (0, 1, 0) untouched."""
    import os
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # PNPFLOW_FORCE_DEVICE / PNPFLOW_DIST_BACKEND=gloo: functional runs of the N > 1 path on a 1-GPU box (every rank on one device,
    # collectives staged through the host)
    local = int(os.environ.get("PNPFLOW_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0"))) if device_index is None else device_index
    backend = os.environ.get("PNPFLOW_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    # PNPFLOW_DIST_FORCE=1: a one-rank job joins a process group too and executes its collectives (the RCCL path on a 1-GPU box)
    if (world > 1 or force_collectives()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0)); os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def global_normal(key, global_shape, lo: int, hi: int) -> torch.Tensor:
    """Rows [lo, hi) of a seeded N(0,1) tensor of the GLOBAL batch shape (numpy Philox keyed by `key`): every rank
    reproduces the same global tensor and keeps its own images."""
    import numpy as np
    g = np.random.Generator(np.random.Philox(key=list(key)))
    return torch.from_numpy(g.standard_normal(size=tuple(global_shape), dtype=np.float32)[lo:hi].copy())


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank `rank`; the first (global_batch % world) ranks get one extra image."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_measurement_noise(batch: int, global_shape, lo: int, hi: int) -> torch.Tensor:
    """Rows [lo, hi) of the reference's measurement-noise draw for batch index `batch`
    (CPU generator, so every rank reproduces the same global tensor)."""
    g = torch.Generator().manual_seed(batch)
    return torch.randn(tuple(global_shape), generator=g, dtype=torch.float32)[lo:hi].contiguous()


def gather_in_image_order(local: torch.Tensor, group=None) -> torch.Tensor:
    """all_gather of a per-image vector; equal shard sizes are not required."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force_collectives()):
        return local
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo" and local.is_cuda:      # gloo collectives run on host tensors
        return gather_in_image_order(local.cpu(), group).to(local.device)
    n = torch.tensor([local.numel()], device=local.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    m = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros(m, device=local.device, dtype=local.dtype)
    pad[: local.numel()] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[: int(s.item())] for b, s in zip(bufs, sizes)])


def mean_psnr(local_psnr: torch.Tensor, group=None) -> float:
    """Mean over the global batch, summed in global image order."""
    return float(gather_in_image_order(local_psnr, group).double().mean())


def empty_shard_result(like: torch.Tensor, shape, steps: int, iter_cb=None, cb_iterations=None, first: int = 0) -> torch.Tensor:
    """What a rank whose shard of the batch is empty returns from `restore_batch`: a 0-image tensor - after calling the
    logging callback for the same iterations, in the same order, as the ranks that do restore images (the callbacks
    all_gather per-image metrics: a rank that skipped them would leave the others blocked in the collective)."""
    x = torch.empty(tuple(shape), dtype=torch.float32, device=like.device)
    if iter_cb is not None:
        its = range(first, int(steps)) if cb_iterations is None else sorted(i for i in set(cb_iterations) if first <= i < int(steps))
        for it in its:
            iter_cb(it, x)
    return x
