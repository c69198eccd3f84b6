"""Batch-parallel replicas: the only multi-GPU form the UNet hot path needs.

The reference has no multi-GPU support at all ("Parallelism strategies": no
NCCL / torch.distributed call site). Every image's denoise loop is independent (GroupNorm,
LayerNorm and attention are per-sample), so batched generation shards by batch across one process
per GPU with NO per-step collective: one RCCL broadcast of the flattened weights over xGMI at
start-up (and whenever rank 0's weights change), then independent per-GPU hipGraph replay loops,
and optionally one gather of the final latents.

Works on any torch.distributed backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests.
"""
from typing import Dict, List, Tuple

import torch
import torch.distributed as dist


def shard_batch(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, end) slice of a global batch for `rank`; remainders go to the first ranks."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _buckets(params: Dict[str, torch.Tensor], bucket_bytes: int) -> List[List[str]]:
    names = sorted(params)
    out, cur, size = [], [], 0
    for n in names:
        nb = params[n].numel() * params[n].element_size()
        if cur and size + nb > bucket_bytes:
            out.append(cur)
            cur, size = [], 0
        cur.append(n)
        size += nb
    if cur:
        out.append(cur)
    return out


def broadcast_parameters(params: Dict[str, torch.Tensor], src: int = 0, bucket_bytes: int = 1 << 30, group=None,
                         force: bool = False) -> int:
    """Broadcast rank `src`'s parameter values into every rank's (already allocated, same-shaped)
    parameter storage, in place. Parameters are packed into large flat buckets (default 1 GiB --
    sized for xGMI point-to-point links and 288 GB of HBM: SD1.5's 1.72 GB goes in two collectives)
    so the transfer is bandwidth- not latency-bound. In-place `copy_` keeps the pointers that captured
    hipGraphs hold valid. Returns the number of bytes broadcast."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return 0
    rank = dist.get_rank(group)
    total = 0
    by_dtype: Dict[torch.dtype, Dict[str, torch.Tensor]] = {}
    for n, p in params.items():
        by_dtype.setdefault(p.dtype, {})[n] = p
    for dtype in sorted(by_dtype, key=str):
        group_params = by_dtype[dtype]
        for names in _buckets(group_params, bucket_bytes):
            numel = sum(group_params[n].numel() for n in names)
            dev = group_params[names[0]].device
            flat = torch.empty(numel, dtype=dtype, device=dev)
            if rank == src:
                off = 0
                for n in names:
                    k = group_params[n].numel()
                    flat[off:off + k].copy_(group_params[n].reshape(-1))
                    off += k
            dist.broadcast(flat, src=src, group=group)
            if rank != src:
                off = 0
                for n in names:
                    p = group_params[n]
                    k = p.numel()
                    # logical-order copy through the parameter's own strides (channels_last included)
                    p.copy_(flat[off:off + k].view(p.shape))
                    off += k
            total += numel * flat.element_size()
    return total


def gather_latents(local: torch.Tensor, dst: int = 0, group=None):
    """Gather per-rank latents (equal shapes) on rank `dst`; returns the list there, None elsewhere."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [local]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if dist.get_backend(group) == "nccl":
        outs = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(outs, local.contiguous(), group=group)
        return outs if rank == dst else None
    outs = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local.contiguous(), outs, dst=dst, group=group)
    return outs


def share_tune_cache(src: int = 0, group=None) -> int:
    """Send rank `src`'s measured kernel choices (sfast.engine.autotune) to every other rank, once, before they build their plans:
    all replicas then run the SAME (variant, split-K) per problem -- identical arithmetic on every GPU -- and only one rank pays
    the seconds of timing. One small object broadcast; no effect on the per-step path. Returns the number of entries received."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    from . import autotune
    box = [autotune.export_cache() if dist.get_rank(group) == src else None]
    dist.broadcast_object_list(box, src=src, group=group)
    if dist.get_rank(group) == src:
        return 0
    return autotune.import_cache(box[0], overwrite=True)
