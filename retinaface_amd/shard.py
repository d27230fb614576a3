"""Image-level data parallelism for one-process-per-GPU runs.

The path is embarrassingly parallel over images (per-image NMS, RetinaFace.cpp:916-918; weights are 0.84 MB and
replicated), so the data path has NO collective: every rank runs the same engine on its contiguous slice of the
batch.  The only exchange is the optional result gather -- fixed-size records (count + cap x 16 floats per image)
all-gathered with torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def shard_range(n_images: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of rank `rank`: ceil(n/world) images per rank, the tail ranks may get fewer / none."""
    per = -(-n_images // world)
    lo = min(rank * per, n_images)
    return lo, min(lo + per, n_images)


RECORD_FLOATS = 16      # 15 floats of FaceDetectInfo + the global anchor index (stored as float bits)


def pack_records(dets: Sequence[Sequence], cap: int) -> np.ndarray:
    """List (per image) of Detection-like objects -> [n_images, 1 + cap*16] float32 (count first)."""
    out = np.zeros((len(dets), 1 + cap * RECORD_FLOATS), np.float32)
    for i, dd in enumerate(dets):
        k = min(len(dd), cap)
        out[i, 0] = len(dd)
        for j in range(k):
            d = dd[j]
            row = out[i, 1 + j * RECORD_FLOATS: 1 + (j + 1) * RECORD_FLOATS]
            row[:15] = d.as_row()
            row[15:16] = np.array([d.anchor_index], np.int32).view(np.float32)
    return out


def unpack_records(buf: np.ndarray, cap: int) -> List[List[Tuple[np.ndarray, int]]]:
    res = []
    for i in range(buf.shape[0]):
        k = min(int(buf[i, 0]), cap)
        rows = buf[i, 1:1 + k * RECORD_FLOATS].reshape(k, RECORD_FLOATS)
        res.append([(rows[j, :15].copy(), int(rows[j, 15:16].view(np.int32)[0])) for j in range(k)])
    return res


def gather_records(local: np.ndarray, n_images: int, device=None):
    """All-gather the per-rank record blocks into the full batch, in image order.  `local` is this rank's
    [hi-lo, R] block; every rank contributes a ceil(n/world)-row block (zero padded) so the collective is one
    fixed-size all_gather_into_tensor."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    per = -(-n_images // world)
    block = torch.zeros((per, local.shape[1]), dtype=torch.float32, device=device)
    if local.shape[0]:
        block[:local.shape[0]] = torch.from_numpy(local).to(block.device)
    full = torch.empty((world * per, local.shape[1]), dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(full, block)
    return full[:n_images].cpu().numpy()
