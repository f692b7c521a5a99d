"""GPU-resident replay buffers with the reference's sampling semantics.

The reference keeps Python lists of `Transition` namedtuples on the host
(utilities/replay_buffer.py:5-58) — at most a few thousand entries, one env.  Here every field is
ONE preallocated device tensor used as a ring, B transitions (one per env of a batch) are appended
per call, and a sample is a view of the ring (two slices concatenated per field when the window wraps); nothing crosses to the host.  What is
kept from the reference, exactly:

* FIFO eviction: when the buffer is full the OLDEST entry is dropped (`offset()` = `pop(0)`,
  replay_buffer.py:12-13,25-29).
* `TransReplayBuffer.get_batch(batch_size)` returns a CONTIGUOUS WINDOW of `batch_size`
  transitions in insertion order, whose start is drawn uniformly from
  `len(buffer) - batch_size + 1` positions with `np.random.choice(sample_range, 1,
  replace=False)[0]` on numpy's global RNG (replay_buffer.py:19-23) — the same call, so a seeded
  reference run and a seeded run of this class pick the same windows.
* `EpisodeReplayBuffer.get_batch(batch_size)` draws `batch_size` distinct episodes with
  `np.random.choice(length, batch_size, replace=False)` and concatenates their transitions in the
  drawn order (replay_buffer.py:45-51).

Insertion order for a batch of envs is (step, env): `add_experience` with B transitions is
equivalent to B reference `add_experience` calls in env order.
"""
from __future__ import annotations

from typing import Dict, Optional

import os

import numpy as np
import torch

Batch = Dict[str, torch.Tensor]


class TransReplayBuffer:
    """Ring of `size` transitions; every field of a transition is a tensor [...], stored as [size (+ window), ...].

    `window` > 0 appends a MIRROR of the first `window` ring positions behind the ring (every insertion into those positions is
    written twice), so that a sampled window of up to `window` transitions is always one contiguous slice — a view, never a copy —
    also when it wraps around the ring.  The learner sets it to its batch size (at 32 x 8192 transitions of the 322-bus env a
    gathered batch is 10 GB of copies)."""

    def __init__(self, size: int, device=None, window: int = 0):
        self.size = int(size)
        self.window = max(0, min(int(window), self.size))
        if self.size <= 0:
            raise ValueError("replay buffer size must be positive")
        self.device = torch.device(device) if device is not None else None
        self._store: Dict[str, torch.Tensor] = {}
        self._side = None       # side stream of the asynchronous insertion (see _copy_in_one_launch) and the event of its last copy
        self._pending = None
        self._tail = 0          # ring position of the oldest entry
        self._len = 0

    def __len__(self) -> int:
        return self._len

    @property
    def store(self) -> Dict[str, torch.Tensor]:
        """field -> ring tensor [size (+ window), ...].  Reading it makes the current stream wait for an insertion that is still in
        flight on the side stream (sync())."""
        self.sync()
        return self._store

    def sync(self) -> None:
        """the current stream waits for the last asynchronous insertion (a no-op when none is pending)"""
        if self._pending is not None:
            torch.cuda.current_stream(self.device).wait_event(self._pending)
            self._pending = None

    @property
    def buffer(self):
        """`len(trainer.replay_buffer.buffer)` is what the reference's update condition reads
        (models/model.py:42-44); a sized view is enough."""
        return range(self._len)

    def _allocate(self, trans: Batch) -> None:
        for k, v in trans.items():
            dev = self.device if self.device is not None else v.device
            self._store[k] = torch.empty((self.size + self.window,) + tuple(v.shape[1:]), dtype=v.dtype, device=dev)
        if self.device is None:
            self.device = next(iter(self._store.values())).device

    def add_experience(self, trans: Batch) -> None:
        """trans[field]: [B, ...] — B transitions appended in env order (replay_buffer.py:25-29).  On the GPU the copy may run on a side
        stream (_copy_in_one_launch): the field tensors may be dropped right away (their memory is kept until the copy has read it), but
        must not be modified IN PLACE afterwards — the rollout hands over fresh tensors every step."""
        if not self._store:
            self._allocate(trans)
        if trans.keys() != {k for k in self._store if not k.endswith("_cached")}:
            raise KeyError(f"transition fields {sorted(trans)} != buffer fields {sorted(self._store)}")
        B = next(iter(trans.values())).shape[0]
        skip = max(0, B - self.size)            # more than a buffer-full at once: only the newest survive
        n = B - skip
        head = (self._tail + self._len) % self.size
        first = min(n, self.size - head)
        pieces = []                              # (destination slice, source slice) of every field: ring, wrap-around, mirror
        for k, v in trans.items():
            if v.shape[0] != B:
                raise ValueError(f"field {k}: leading dimension {v.shape[0]} != {B}")
            dst = self._store[k]
            pieces.append((dst[head:head + first], v[skip:skip + first]))
            if n > first:
                pieces.append((dst[0:n - first], v[skip + first:B]))
            if self.window:                       # ring positions below `window` live a second time behind the ring
                if head < self.window:
                    m = min(first, self.window - head)
                    pieces.append((dst[self.size + head:self.size + head + m], v[skip:skip + m]))
                if n > first:
                    m = min(n - first, self.window)
                    pieces.append((dst[self.size:self.size + m], v[skip + first:skip + first + m]))
        if not self._copy_in_one_launch(pieces):
            self.sync()
            for d, s_ in pieces:
                d.copy_(s_)
        overflow = max(0, self._len + n - self.size)
        self._tail = (self._tail + overflow) % self.size
        self._len = min(self.size, self._len + n)

    def _copy_in_one_launch(self, pieces) -> bool:
        """all pieces of one insertion as ONE HIP launch (libmapdn_hip.so: mapdn_copy_segments) instead of ~20 copy launches — when every
        piece is a contiguous same-dtype device-to-device copy of 16-byte granularity (the batched env's fields are); else False."""
        if not pieces or len(pieces) > 48 or os.environ.get("MAPDN_FUSED_ROLLOUT", "1") == "0":
            return False
        dev = pieces[0][0].device
        if dev.type != "cuda":
            return False
        srcs = []
        for d, s_ in pieces:
            if s_.device != dev or s_.dtype != d.dtype or s_.shape != d.shape or not d.is_contiguous():
                return False
            s_ = s_ if s_.is_contiguous() else s_.contiguous()
            nb = d.numel() * d.element_size()
            if nb % 16 or d.data_ptr() % 16 or s_.data_ptr() % 16:
                return False
            srcs.append(s_)
        from . import _lib
        C = _lib.C
        n = len(pieces)
        sp = (C.c_void_p * n)(*[t.data_ptr() for t in srcs])
        dp = (C.c_void_p * n)(*[d.data_ptr() for d, _ in pieces])
        nb = (C.c_int64 * n)(*[d.numel() * d.element_size() for d, _ in pieces])
        with torch.cuda.device(dev):
            if os.environ.get("MAPDN_REPLAY_ASYNC", "1") != "0":
                # the copy (0.6 GB per step of the 322-bus shard: HBM-bound) runs on a SIDE stream, under the next step's policy forward and
                # power flow (MFMA / latency-bound): it starts after everything enqueued so far (the producers of the sources; earlier
                # readers of the ring slots it overwrites), and whoever reads the ring next waits for it (the `store` property, get_batch)
                main = torch.cuda.current_stream(dev)
                if self._side is None:
                    self._side = torch.cuda.Stream(device=dev)
                ev = torch.cuda.Event()
                ev.record(main)
                self._side.wait_event(ev)
                _lib.check(_lib.load().mapdn_copy_segments(sp, dp, nb, n, self._side.cuda_stream))
                for t in srcs:
                    t.record_stream(self._side)        # (the allocator must not hand a source's memory out again before the copy has read it)
                self._pending = torch.cuda.Event()
                self._pending.record(self._side)
            else:
                _lib.check(_lib.load().mapdn_copy_segments(sp, dp, nb, n, torch.cuda.current_stream(dev).cuda_stream))
        return True

    def get_single(self, index: int) -> Batch:
        if index < 0:
            index += self._len
        if not 0 <= index < self._len:
            raise IndexError("list index out of range")
        pos = (self._tail + index) % self.size
        return {k: v[pos] for k, v in self.store.items()}

    def get_batch(self, batch_size: int, start: Optional[int] = None) -> Batch:
        return self.get_truncated_episodes_batch(batch_size, start)

    def get_truncated_episodes_batch(self, batch_size: int, start: Optional[int] = None) -> Batch:
        sample_range = self._len - batch_size + 1
        if sample_range <= 0:
            raise ValueError("a must be greater than 0 unless no samples are taken")   # numpy's message
        if start is None:
            start = int(np.random.choice(sample_range, 1, replace=False)[0])
        elif not 0 <= start < sample_range:
            raise IndexError(f"window start {start} outside [0, {sample_range})")
        base = (self._tail + start) % self.size
        if base + batch_size <= self.size or (base + batch_size <= self.size + self.window and batch_size <= self.window):
            # the window does not wrap around the ring: hand out VIEWS of the ring (no copy — at 32 x 8192 transitions of the 322-bus
            # env a gathered copy is 2.6 GB and 7 % of the training loop).  Valid until the next add_experience, which is how every
            # caller uses a batch (models/model.py:39-70: sample, update, drop); `.clone()` a field to keep it longer.
            return {k: v[base:base + batch_size] for k, v in self.store.items()}
        first = self.size - base                             # the window wraps (and no mirror covers it): two contiguous pieces, copied
        return {k: torch.cat((v[base:self.size], v[:batch_size - first]), dim=0) for k, v in self.store.items()}

    def clear(self) -> None:
        self._tail = 0
        self._len = 0


class EpisodeReplayBuffer:
    """Ring of `size` episodes of at most `max_steps` transitions each; fields stored [size, max_steps, ...]."""

    def __init__(self, size: int, max_steps: int, device=None):
        self.size, self.max_steps = int(size), int(max_steps)
        if self.size <= 0 or self.max_steps <= 0:
            raise ValueError("replay buffer size and max_steps must be positive")
        self.device = torch.device(device) if device is not None else None
        self.store: Dict[str, torch.Tensor] = {}
        self.lengths = np.zeros(self.size, dtype=np.int64)     # host side: needed to size a sample
        self._tail = 0
        self._len = 0

    def __len__(self) -> int:
        return self._len

    @property
    def buffer(self):
        return range(self._len)

    def add_experience(self, episodes: Batch, lengths) -> None:
        """episodes[field]: [T, B, ...] (time-major window of a batched rollout); lengths[b] = number of
        valid steps of env b's episode.  Appends B episodes in env order (replay_buffer.py:53-57)."""
        lengths = np.asarray(lengths.cpu() if isinstance(lengths, torch.Tensor) else lengths, dtype=np.int64)
        T, B = next(iter(episodes.values())).shape[:2]
        if T > self.max_steps or lengths.shape != (B,) or lengths.min() < 1 or lengths.max() > T:
            raise ValueError("episode window does not fit the buffer / bad lengths")
        if not self.store:
            for k, v in episodes.items():
                dev = self.device if self.device is not None else v.device
                self.store[k] = torch.zeros((self.size, self.max_steps) + tuple(v.shape[2:]), dtype=v.dtype, device=dev)
            if self.device is None:
                self.device = next(iter(self.store.values())).device
        skip = max(0, B - self.size)
        for b in range(skip, B, max(1, self.size)):             # chunks that never wrap more than once
            nb = min(self.size, B - b)
            head = (self._tail + self._len) % self.size
            first = min(nb, self.size - head)
            for k, v in episodes.items():
                dst = self.store[k]
                dst[head:head + first, :T].copy_(v[:, b:b + first].transpose(0, 1))
                if nb > first:
                    dst[0:nb - first, :T].copy_(v[:, b + first:b + nb].transpose(0, 1))
            pos = (head + np.arange(nb)) % self.size
            self.lengths[pos] = lengths[b:b + nb]
            overflow = max(0, self._len + nb - self.size)
            self._tail = (self._tail + overflow) % self.size
            self._len = min(self.size, self._len + nb)

    def get_single(self, index: int) -> Batch:
        if index < 0:
            index += self._len
        if not 0 <= index < self._len:
            raise IndexError("list index out of range")
        pos = (self._tail + index) % self.size
        return {k: v[pos, :self.lengths[pos]] for k, v in self.store.items()}

    def get_batch(self, batch_size: int, indices=None) -> Batch:
        if indices is None:
            indices = np.random.choice(self._len, batch_size, replace=False)
        pos = (self._tail + np.asarray(indices, dtype=np.int64)) % self.size
        flat = np.concatenate([p * self.max_steps + np.arange(self.lengths[p]) for p in pos])
        idx = torch.from_numpy(flat).to(self.device)
        return {k: v.reshape((self.size * self.max_steps,) + tuple(v.shape[2:])).index_select(0, idx)
                for k, v in self.store.items()}

    def clear(self) -> None:
        self._tail = 0
        self._len = 0
