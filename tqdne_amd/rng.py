"""Per-process random streams of the training path.

The reference trains under Lightning DDP: every rank seeds its own torch generator (``seed_everything`` + the
DistributedSampler's rank offset), so the noise levels, the noise and the dropout masks differ between ranks
(experiments/train_1d_edm.py:34-41; tqdne/edm.py:126-134; tqdne/unet.py:101).  Here

* ``seed_rank(seed, rank)`` seeds torch's generators with ``seed + rank`` (eps / noise draws) and records the rank,
* ``next_dropout_seed()`` hands the HIP kernels' counter-based dropout hash (csrc/common.hpp ``drop_key`` / ``drop_hash``) a 64-bit seed
  built from (torch's initial seed, the rank, a per-process call counter): masks differ between ranks and between
  steps, and are reproducible for a fixed (seed, rank, call number).
"""

from __future__ import annotations

import torch

_state = {"rank": 0, "counter": 0}

_M64 = 0xFFFFFFFFFFFFFFFF


def _mix(z: int) -> int:
    """splitmix64 finaliser"""
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def seed_rank(seed: int, rank: int = 0) -> None:
    """Seed this process' random streams for data-parallel rank ``rank`` and restart the dropout call counter."""
    torch.manual_seed(int(seed) + int(rank))
    _state["rank"] = int(rank)
    _state["counter"] = 0


def set_rank(rank: int) -> None:
    _state["rank"] = int(rank)


def get_rank() -> int:
    return _state["rank"]


def reset_dropout_counter(value: int = 0) -> None:
    _state["counter"] = int(value)


def dropout_seed_for(initial_seed: int, rank: int, counter: int) -> int:
    """The seed ``next_dropout_seed`` returns for call number ``counter`` (pure function; the tests rebuild masks from it)."""
    return _mix(_mix((int(initial_seed) & _M64) ^ ((int(rank) + 1) * 0xD1B54A32D192ED03 & _M64)) ^ (int(counter) & _M64))


def next_dropout_seed() -> int:
    _state["counter"] += 1
    return dropout_seed_for(torch.initial_seed(), _state["rank"], _state["counter"])
