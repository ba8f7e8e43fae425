"""Multi-GPU plumbing for the streaming path: one process per GPU, streams sharded in
contiguous blocks, the fused weight blob broadcast once from rank 0 (RCCL over xGMI on GPUs,
gloo in the CPU tests).  Streams never interact at inference (BatchNorm folded, attention is
within a frame, GRU state is per stream), so there is NO per-step collective."""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_streams: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of streams [b0, b1) owned by `rank`; sizes differ by at most one."""
    assert 0 <= rank < world and n_streams >= 0
    base, rem = divmod(n_streams, world)
    b0 = rank * base + min(rank, rem)
    return b0, b0 + base + (1 if rank < rem else 0)


def broadcast_blob(blob: torch.Tensor, src: int = 0) -> torch.Tensor:
    """ncclBroadcast of the flat fp32 weight blob (<= 4.4 MB: latency-bound)."""
    if dist.is_available() and dist.is_initialized():
        dist.broadcast(blob, src=src)
    return blob


def synthetic_streams(b0: int, b1: int, n_samples: int, sr: int, seed: int) -> torch.Tensor:
    """Synthetic input of SURVEY.md §8(d) for global streams b0..b1-1:
    clip(0.1*N(0,1) + 0.3*sin(2*pi*f_b*t), -1, 1), f_b = 100 + 13*b Hz.  Each stream has its own
    generator so that a stream's signal does not depend on the sharding."""
    t = np.arange(n_samples, dtype=np.float64) / sr
    out = np.empty((b1 - b0, n_samples), dtype=np.float32)
    for i, b in enumerate(range(b0, b1)):
        rng = np.random.Generator(np.random.PCG64([seed, b]))
        out[i] = np.clip(0.1 * rng.standard_normal(n_samples) + 0.3 * np.sin(2 * np.pi * (100.0 + 13.0 * b) * t), -1, 1)
    return torch.from_numpy(out)


def gather_frame_counts(frames_done: int, elapsed: float, device) -> Tuple[int, float]:
    """End-of-run report: SUM of frames, MAX of elapsed over ranks."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return frames_done, elapsed
    f = torch.tensor([float(frames_done)], dtype=torch.float64, device=device)
    e = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.all_reduce(e, op=dist.ReduceOp.MAX)
    return int(f.item()), float(e.item())
