"""Batch sharding over the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The Encode/Decode path has no cross-clip dependency (no BatchNorm; GroupNorm / LayerNorm / RMS scale are per
sample; RVQ is per frame), so clips are split into contiguous blocks, every rank runs the single-GPU engine on its block with
replicated weights, and the only collective is the all-gather of the emitted integer codes.  Decode needs only the local
latents, so the gather is issued on a side stream and overlaps the local decode.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple


def shard_bounds(n_clips: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; the first n_clips % world ranks hold one clip more."""
    if world <= 0 or not (0 <= rank < world) or n_clips < 0:
        raise ValueError("bad shard request")
    q, r = divmod(n_clips, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_sizes(n_clips: int, world: int) -> List[int]:
    return [shard_bounds(n_clips, world, r)[1] - shard_bounds(n_clips, world, r)[0] for r in range(world)]


def all_gather_codes(codes, n_clips: int, group=None, out=None):
    """codes: this rank's [b_local, ...] integer tensor (torch).  Returns the [n_clips, ...] tensor of all ranks in clip order.

    Equal shards use one all_gather_into_tensor (a single RCCL collective); ragged shards are padded to the largest shard,
    gathered, and the padding rows are dropped.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_clips, world)
    if codes.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError(f"rank holds {codes.shape[0]} clips, expected {sizes[dist.get_rank(group)]}")
    bmax = max(sizes)
    tail = tuple(codes.shape[1:])
    if all(s == bmax for s in sizes):
        if out is None:
            out = torch.empty((n_clips,) + tail, dtype=codes.dtype, device=codes.device)
        dist.all_gather_into_tensor(out, codes.contiguous(), group=group)
        return out
    pad = torch.zeros((bmax,) + tail, dtype=codes.dtype, device=codes.device)
    pad[: codes.shape[0]] = codes
    buf = torch.empty((world * bmax,) + tail, dtype=codes.dtype, device=codes.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    parts = [buf[r * bmax: r * bmax + sizes[r]] for r in range(world)]
    return torch.cat(parts, 0)


def concat_levels(levels: Sequence, ) -> Tuple["object", List[int]]:
    """SNAC emits one code tensor per level ([B, T'/stride_i]); concatenate them along time so ONE collective moves them."""
    import torch
    widths = [int(l.shape[-1]) for l in levels]
    return torch.cat([l.reshape(l.shape[0], -1) for l in levels], dim=-1), widths


def all_gather_levels(levels: Sequence, n_clips: int, group=None, out=None):
    """SNAC.Encode's List<Tensor> (SNAC.cs:113-150) of this rank -> the [n_clips, sum(widths)] tensor of all ranks (levels of a clip
    side by side, the layout nc_snac_encode emits): one collective for all levels.  split_levels(result, widths) restores the list."""
    flat, _ = concat_levels(levels)
    return all_gather_codes(flat.contiguous(), n_clips, group=group, out=out)


def split_levels(flat, widths: Sequence[int]):
    out, o = [], 0
    for w in widths:
        out.append(flat[:, o:o + w])
        o += w
    return out
