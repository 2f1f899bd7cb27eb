"""Multi-GPU: one process per GPU, rows (tokens) sharded, factor matrices broadcast once over RCCL/xGMI.

The path is embarrassingly parallel over tokens: the only shared state is read-only (left/right/hadK and two
clip scalars per layer).  So there is NO data-path collective — just a one-time ``broadcast`` of the small
matrices from rank 0 at set-up.  ``backend='nccl'`` is RCCL on ROCm; tests use ``gloo`` on CPU.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.distributed as dist


def shard_rows(total_rows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous [start, stop) block of rows owned by ``rank``; sizes differ by at most one row."""
    base, rem = divmod(total_rows, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_experts(counts, world_size: int, rank: int):
    """Expert-parallel partition of a routed-expert workload (the layout of deepseek_v3/model.py:657-660: rank r owns the
    experts [r E / W, (r + 1) E / W)): -> (e0, e1, group_offsets) where ``group_offsets`` (int64, len e1 - e0 + 1, starts
    at 0) are this rank's expert groups over ITS rows — the rows routed to its experts, expert-major. ``counts``: rows routed
    to every expert (len E, any integer sequence / tensor). Every expert and every routed row belongs to exactly one rank."""
    counts = [int(c) for c in counts]
    e0, e1 = shard_rows(len(counts), world_size, rank)
    offs = [0]
    for c in counts[e0:e1]:
        offs.append(offs[-1] + c)
    return e0, e1, torch.tensor(offs, dtype=torch.int64)


def broadcast_matrices(mats: Dict[str, torch.Tensor], src: int = 0, group=None, force: bool = False) -> Dict[str, torch.Tensor]:
    """Broadcast every tensor of ``mats`` from ``src`` as ONE flat buffer (one collective for all layers: a few
    hundred KB to a few MB — latency-bound on xGMI, so fewer, larger messages). ``force``: run the collective in a one-rank group too
    (bench.py --force-dist: the RCCL path exercised on a single-GPU box)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return mats
    keys = sorted(mats)
    if not keys:
        return mats
    # every tensor starts on a 16-byte boundary of the flat buffer: an odd-length fp16 tensor followed by an fp32 one
    # would otherwise leave the second at a 2-byte offset, which .view(dtype) rejects
    ALIGN = 16
    dev = mats[keys[0]].device
    offs, total = {}, 0
    for k in keys:
        offs[k] = total
        nbytes = mats[k].numel() * mats[k].element_size()
        total += (nbytes + ALIGN - 1) // ALIGN * ALIGN
    flat = torch.zeros(total, dtype=torch.uint8, device=dev)
    for k in keys:
        src_bytes = mats[k].contiguous().reshape(-1).view(torch.uint8)
        flat[offs[k]:offs[k] + src_bytes.numel()] = src_bytes
    dist.broadcast(flat, src=src, group=group)
    out = {}
    for k in keys:
        nbytes = mats[k].numel() * mats[k].element_size()
        out[k] = flat[offs[k]:offs[k] + nbytes].view(mats[k].dtype).reshape(mats[k].shape).clone()
    return out


def gather_rows(local: torch.Tensor, group=None) -> torch.Tensor:
    """all_gather of row shards (equal sizes) — for tests that compare with the single-GPU result only."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    parts = [torch.empty_like(local) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, local.contiguous(), group=group)
    return torch.cat(parts, dim=0)
