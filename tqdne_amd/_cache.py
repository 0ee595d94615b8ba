"""Bounded caches of execution plans and per-shape scratch buffers.

Every plan owns static activation buffers (a few GB for the paper UNet at B = 64), so a cache keyed by the call's shape must not grow
with every shape a run ever sees (a loader's ragged last batch, evaluation over several batch sizes).  ``PlanCache`` keeps the
``cap`` most recently used GROUPS of keys (a group = the plans of one (batch, length, device): the lanes of a multi-stream sampler /
training step live and die together) and drops the least recently used group when a new one arrives."""

from __future__ import annotations

import os
from collections import OrderedDict

PLAN_SHAPES = max(2, int(os.environ.get("TQDNE_PLAN_CACHE_SHAPES", "6")))   # (batch, length, device) groups kept per model
SCRATCH_ENTRIES = max(16, int(os.environ.get("TQDNE_SCRATCH_CACHE_ENTRIES", "64")))


class PlanCache:
    """dict-like (``get`` / ``[]=`` / ``values`` / ``items`` / ``in`` / ``len``), least-recently-used eviction by group.

    ``group(key)``: the eviction unit of a key (default: the key itself).  ``on_evict(items)``: called with the ``(key, value)`` pairs
    of the dropped groups while the cache still holds the LAST references to them -- the owners synchronise the device there (launches
    of an evicted plan may still be in flight on a side stream, and the caching allocator only orders re-use against the stream a block
    was allocated on) and break the values' reference cycles, so that dropping them right afterwards frees their device memory.
    ``can_evict()``: False postpones the eviction to a later insertion (the cache then runs over its cap meanwhile): a device
    synchronisation inside a stream capture would invalidate the capture."""

    def __init__(self, cap: int, group=None, on_evict=None, can_evict=None):
        self.cap, self._group, self._on_evict, self._can_evict = cap, (group or (lambda k: k)), on_evict, can_evict
        self._d: "OrderedDict[object, dict]" = OrderedDict()   # group -> {key: value}, least recently used first
        self.evictions = 0

    def get(self, key, default=None):
        g = self._group(key)
        grp = self._d.get(g)
        if grp is None or key not in grp:
            return default
        self._d.move_to_end(g)
        return grp[key]

    def __setitem__(self, key, value):
        g = self._group(key)
        grp = self._d.get(g)
        if grp is None:
            if len(self._d) >= self.cap and (self._can_evict is None or self._can_evict()):
                dropped = []
                while len(self._d) >= self.cap:
                    _, old = self._d.popitem(last=False)
                    dropped.append(old)
                    self.evictions += 1
                if self._on_evict is not None:   # (synchronise first, THEN let go of the buffers)
                    self._on_evict([kv for old in dropped for kv in old.items()])
                for old in dropped:
                    old.clear()
                del dropped
            grp = self._d[g] = {}
        grp[key] = value
        self._d.move_to_end(g)

    def __contains__(self, key):
        grp = self._d.get(self._group(key))
        return grp is not None and key in grp

    def __len__(self):
        return sum(len(g) for g in self._d.values())

    def groups(self):
        return list(self._d.keys())

    def keys(self):
        return [k for g in self._d.values() for k in g]

    def values(self):
        return [v for g in self._d.values() for v in g.values()]

    def items(self):
        return [kv for g in self._d.values() for kv in g.items()]

    def clear(self):
        self._d.clear()


def _not_capturing() -> bool:
    import torch
    return not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())


def _sync_on_evict(items):
    """wait for the devices named in the evicted keys (every key carries ``str(device)``), or for the current one; then take the evicted
    values apart where they sit in reference cycles (``release()`` of an execution plan: plan <-> backward plan), so that their
    activation / gradient buffers go back to the allocator with the last reference instead of waiting for a cyclic-GC pass that the
    caching allocator never triggers"""
    import torch
    if torch.cuda.is_available():
        devs = {part for k, _ in items if isinstance(k, tuple) for part in k if isinstance(part, str) and part.startswith("cuda")}
        for d in devs or {None}:
            torch.cuda.synchronize(None if d is None else torch.device(d))
    for _, v in items:
        rel = getattr(v, "release", None)
        if callable(rel):
            rel()


def plan_cache() -> PlanCache:
    """cache of execution plans keyed (B, T, device[, lane]): grouped by the first three entries"""
    return PlanCache(PLAN_SHAPES, group=lambda k: k[:3], on_evict=_sync_on_evict, can_evict=_not_capturing)


def scratch_cache() -> PlanCache:
    """cache of per-shape scratch buffers of the EDM / consistency wrappers (scalars, noised copies, sampler state): small next to the
    plans, bounded per entry"""
    return PlanCache(SCRATCH_ENTRIES, on_evict=_sync_on_evict, can_evict=_not_capturing)
