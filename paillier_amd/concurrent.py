"""Overlapping small batches on one GPU.

A batch of a few thousand ciphertexts cannot fill an MI355X: the kernels are latency-bound (run time = the length of one
ladder, whatever the batch: tools/small_batch_sweep.py), so a second, third ... independent batch costs nothing if it runs
at the same time.  The C ABI is blocking and a context is one stream, so concurrency = several contexts (each with a stream
of its own) driven from several host threads; ctypes releases the GIL for the duration of a call.

The typical case is threshold decryption of a small batch (thresholdkey.go:192-201): the t servers' PartialDecrypt of the
same ciphertexts are t independent exponentiations -- `Lanes.map` runs them side by side.
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
from typing import Callable, List, Sequence

from .api import Context


class Lanes:
    """`width` contexts on one device, each with its own stream, and a thread pool to drive them."""

    def __init__(self, device: int = 0, width: int = 4, partition_cus: bool = True):
        """partition_cus: lane k's stream is confined to the k-th of `width` slices of the device's compute units, so that the
        lanes' kernels run side by side instead of piling up on the same CUs (measured: three concurrent 2048-ciphertext
        PartialDecrypt calls take 1.8x the time of one without the partition)."""
        self.contexts: List[Context] = [Context(device, own_stream=True) for _ in range(width)]
        if partition_cus and width > 1:
            for k, cx in enumerate(self.contexts):
                cx.set_flag("cu_partition", (width << 16) | k)
        self.pool = ThreadPoolExecutor(max_workers=width)
        self.state: List[dict] = [{} for _ in range(width)]      # per-lane cache for key handles etc.

    def map(self, fn: Callable, items: Sequence) -> list:
        """fn(ctx, state, item) for every item, item i on lane i % width; items of one lane run in order, lanes in parallel."""
        w = len(self.contexts)

        def lane(k):
            return [(i, fn(self.contexts[k], self.state[k], items[i])) for i in range(k, len(items), w)]

        out = [None] * len(items)
        for part in self.pool.map(lane, range(min(w, len(items)))):
            for i, r in part:
                out[i] = r
        return out

    def close(self):
        self.pool.shutdown(wait=True)
        self.state = []
        for c in self.contexts:
            c.close()
        self.contexts = []
