"""
Multi-GPU template sharding: one process per GPU, every rank holds the whole image and a subset
of the units (templates / rotations / scales); the only exchange step is an all-gather of the
per-rank hit lists (24-byte records), after which every rank runs the same global NMS.

The reference has no distributed code at all: its only parallelism is one thread-pool task per
template (MTM/__init__.py:172-175), which is the same independence this module exploits.

Exchange backends
  "rccl"  : ncclAllGather inside libmtm_hip.so (mtm_comm_*; RCCL over xGMI).  The 128-byte unique
            id travels through the caller's bootstrap (here: torch.distributed's store).
  "torch" : torch.distributed.all_gather on CPU tensors (gloo) - used by the CPU tests and as a
            fallback where RCCL is unavailable.
"""
import contextlib
import os
import sys
from typing import List, Sequence

import numpy as np

from . import _lib
from .NMS import NMS


def unit_cost(template, image_shape, masked=False) -> float:
    """Multiply-accumulates of the direct method: out_px * w * h * C (* 2 with a mask)."""
    th, tw = template.shape[:2]
    ch = 1 if template.ndim == 2 else template.shape[2]
    out = max(image_shape[0] - th + 1, 0) * max(image_shape[1] - tw + 1, 0)
    return float(out) * th * tw * ch * (2.0 if masked else 1.0)


def shard_units(costs: Sequence[float], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first partition of unit indices over ranks (deterministic)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += costs[i]
    return [sorted(s) for s in shards]


@contextlib.contextmanager
def _stdout_to_stderr():
    """librccl prints a version banner on stdout when it is first initialised; programs whose stdout
    is a protocol (bench.py prints one JSON line) get it on stderr instead."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


class HitExchange:
    """All-gather of structured hit arrays (dtype _lib.HIT_DTYPE) across ranks."""

    def __init__(self, backend="torch", rank=0, world_size=1, context=None, group=None):
        self.backend = backend
        self.rank, self.world_size = rank, world_size
        self.group = group
        self.ctx = context
        self._slot_hits = 512
        if backend == "rccl" and world_size > 1:
            import torch.distributed as dist
            self.ctx = context or _lib.default_context()
            with _stdout_to_stderr():
                box = [_lib.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0, group=group)     # bootstrap only
                self.ctx.comm_init(box[0], world_size, rank)
        elif backend not in ("rccl", "torch"):
            raise ValueError("backend must be 'rccl' or 'torch'")

    def allgather(self, hits: np.ndarray) -> np.ndarray:
        hits = np.ascontiguousarray(hits, dtype=_lib.HIT_DTYPE)
        if self.world_size == 1:
            return hits
        if self.backend == "rccl":
            out, _ = self.ctx.allgather_hits(hits)
            return out
        # same protocol as mtm_comm_allgather_hits: ONE all-gather of fixed slots [count | records]; a second
        # one with larger slots only if some rank had more records than the slot holds.  The slot size follows
        # twice the largest count of the previous exchange (every rank sees every count, so they agree).
        import torch
        import torch.distributed as dist
        rec = _lib.HIT_DTYPE.itemsize
        slot_hits = self._slot_hits
        while True:
            buf = np.zeros(16 + slot_hits * rec, dtype=np.uint8)
            buf[:8] = np.frombuffer(np.int64(len(hits)).tobytes(), dtype=np.uint8)
            k = min(len(hits), slot_hits)
            buf[16:16 + k * rec] = hits[:k].view(np.uint8).reshape(-1)
            mine = torch.from_numpy(buf)
            parts = [torch.empty_like(mine) for _ in range(self.world_size)]
            dist.all_gather(parts, mine, group=self.group)
            parts = [p.numpy() for p in parts]
            counts = [int(np.frombuffer(p[:8].tobytes(), dtype=np.int64)[0]) for p in parts]
            mx = max(counts)
            want = 512
            while want < 2 * mx:
                want *= 2
            self._slot_hits = want
            if mx <= slot_hits:
                break
            slot_hits = mx
        out = [p[16:16 + c * rec].view(_lib.HIT_DTYPE) for p, c in zip(parts, counts)]
        return np.concatenate(out) if out else hits[:0]


def merge_and_nms(raw_all: np.ndarray, listTemplates, method, N_object, score_threshold, maxOverlap,
                  xOffset=0, yOffset=0):
    """Global NMS over the gathered hits; identical on every rank.  The gathered list is first put
    in the single-process order (template index, then the per-template order each rank produced)."""
    from . import _nms_raw, _to_hit_list
    idx = raw_all["templ_idx"]
    if len(idx) > 1 and bool((idx[1:] < idx[:-1]).any()):       # one rank: already in template order
        raw_all = raw_all[np.argsort(idx, kind="stable")]
    kept = _nms_raw(raw_all, score_threshold, method == 1, N_object, maxOverlap)
    return _to_hit_list(kept, listTemplates, xOffset, yOffset)


def matchTemplates_sharded(listTemplates, image, exchange: HitExchange, method=5, N_object=float("inf"),
                           score_threshold=0.5, maxOverlap=0.25, searchBox=None, find_local=None):
    """matchTemplates with the units sharded over exchange.world_size ranks.  Collective: every rank
    calls it with the same arguments and gets the same list back.  ``find_local`` (tests) replaces
    the GPU step: callable(sub_list, image) -> structured hits with LOCAL template indices."""
    from . import _raw_matches, _validate_search
    if maxOverlap < 0 or maxOverlap > 1:
        raise ValueError("Maximal overlap between bounding box is in range [0-1]")
    image, xOffset, yOffset = _validate_search(listTemplates, image, N_object, searchBox)
    costs = [unit_cost(t[1], image.shape, len(t) >= 3 and method in (0, 3)) for t in listTemplates]
    mine = shard_units(costs, exchange.world_size)[exchange.rank]
    sub = [listTemplates[i] for i in mine]
    if find_local is not None:
        raw = find_local(sub, image)
    elif sub:
        raw = _raw_matches(sub, image, method, N_object, score_threshold)
    else:
        raw = np.zeros(0, dtype=_lib.HIT_DTYPE)
    raw = raw.copy()
    raw["templ_idx"] = np.asarray(mine, dtype=np.int32)[raw["templ_idx"]] if len(raw) else raw["templ_idx"]
    gathered = exchange.allgather(raw)
    if method == 0:
        raise ValueError("The method TM_SQDIFF is not supported. Use TM_SQDIFF_NORMED instead.")
    return merge_and_nms(gathered, listTemplates, method, N_object, score_threshold, maxOverlap, xOffset, yOffset)
