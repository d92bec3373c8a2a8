"""One-process-per-GPU sharded scoring: images split contiguously by rank, per-rank logits, one RCCL all-gather.

The reference evaluates unsharded (every rank scores the whole test set, dassl/data/data_manager.py:42-43 wraps
only the training sampler; there is no all_gather anywhere in it - SURVEY.md §2), so this exchange is new: the
path shards by independent images, weights are replicated (172 MB bf16), and the only collective is the
all-gather of ``[B_rank, 80]`` fp32 logits (80 KiB per rank at B_rank = 256) before the BCE / mAP step.  The
message is latency-bound on xGMI, so it is ONE collective per step on the gathered logits, never per layer.
Backend "nccl" is RCCL on ROCm; "gloo" carries the same code on CPU for the world_size-2 tests.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; initialises the default process group when
    WORLD_SIZE > 1.  One process drives one GPU: the device is chosen by LOCAL_RANK before any allocation."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # LECLIP_DIST_BACKEND=gloo: rehearse the multi-rank path with several ranks on ONE GPU (RCCL refuses two ranks on a device);
        # recorded by bench.py under env_overrides
        backend = backend or os.environ.get("LECLIP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of n items; the first n % world ranks take one extra (ragged tails allowed)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local: torch.Tensor, counts=None) -> torch.Tensor:
    """Concatenate per-rank row blocks [n_r, C] in rank order.  Equal blocks use all_gather_into_tensor (one RCCL
    call on one flat buffer); ragged blocks pad to the largest shard and trim."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    local = local.contiguous()
    if counts is None:
        return _gather_flat(local, world)
    width = max(counts)
    padded = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    out = _gather_flat(padded, world)
    return torch.cat([out[r * width:r * width + counts[r]] for r in range(world)], dim=0)


def _gather_flat(block: torch.Tensor, world: int) -> torch.Tensor:
    """Equal-size blocks -> one [world * n, ...] tensor in rank order.  RCCL ("nccl"): a single all_gather_into_tensor on
    one flat buffer - an error there is a real communicator failure and propagates.  gloo (the CPU test transport, also
    used to carry device tensors when two test ranks share one GPU) has no flat primitive: list form, chosen by backend
    name, never by catching an exception."""
    if dist.get_backend() == "gloo":
        parts = [torch.empty_like(block) for _ in range(world)]
        dist.all_gather(parts, block)
        return torch.cat(parts, dim=0)
    out = torch.empty((world * block.shape[0],) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
    dist.all_gather_into_tensor(out, block)
    return out


class ShardedScorer:
    """Scores a global batch: each rank runs ``score_fn`` on its contiguous shard, logits are all-gathered."""

    def __init__(self, score_fn, n_out: int = 80):
        self.score_fn = score_fn
        self.n_out = n_out          # logit columns (80 COCO labels); only used when a rank's shard is empty
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1

    def local_slice(self, n_global: int) -> slice:
        lo, hi = shard_bounds(n_global, self.rank, self.world)
        return slice(lo, hi)

    def score_local(self, local_images: torch.Tensor, n_global: Optional[int] = None) -> torch.Tensor:
        """``local_images`` is already this rank's shard; returns the gathered [n_global, C] logits on every rank."""
        if local_images.shape[0] == 0:   # ragged split with more ranks than images: this rank contributes no rows
            counts = [shard_bounds(n_global, r, self.world)[1] - shard_bounds(n_global, r, self.world)[0] for r in range(self.world)]
            return all_gather_rows(torch.zeros((0, self.n_out), dtype=torch.float32, device=local_images.device), counts)
        logits = self.score_fn(local_images)
        self.n_out = logits.shape[1]
        if self.world == 1:
            return logits
        if n_global is None or n_global == self.world * local_images.shape[0]:
            return all_gather_rows(logits)
        counts = [shard_bounds(n_global, r, self.world)[1] - shard_bounds(n_global, r, self.world)[0] for r in range(self.world)]
        return all_gather_rows(logits, counts)

    def score_global(self, images: torch.Tensor) -> torch.Tensor:
        """Every rank holds the global batch (e.g. from a shared loader); each scores only its slice."""
        return self.score_local(images[self.local_slice(images.shape[0])].contiguous(), images.shape[0])
