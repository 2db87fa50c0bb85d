"""Dropout seeds of the launches a hipGraph records.

A dropout launch gets its 64-bit seed as a kernel argument; recorded in a hipGraph that argument would be frozen.  While the trainer
records a graph it activates a SeedArena: `draw()` then hands out a one-element DEVICE tensor (a slot of the arena) instead of the
integer, the kernels read the seed through that pointer, and before every replay the trainer draws the slots' seeds on the host --
the same `torch.randint` calls, in the same order, the eager modules make -- and uploads them."""
import torch

active = None      # the arena being recorded into, or None (eager: seeds are plain integers)


def _one():
    return int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64))


def draw():
    """the next dropout seed: an int (eager) or a device slot (recording)"""
    s = _one()
    if active is None:
        return s
    return active.slot(s)


class SeedArena:
    def __init__(self, device, capacity=64):
        self.host = torch.zeros(capacity, dtype=torch.int64, pin_memory=True)
        self.dev = torch.zeros(capacity, dtype=torch.int64, device=device)
        self.n = 0

    def slot(self, seed):
        i = self.n
        if i >= self.host.numel():
            raise RuntimeError("SeedArena: more dropout launches than slots")
        self.n += 1
        self.host[i] = seed
        return self.dev[i:i + 1]

    def redraw(self, lo, hi):
        """new seeds for slots [lo, hi) in slot order (= the order the eager modules would draw them)"""
        for i in range(lo, hi):
            self.host[i] = _one()

    def upload(self, lo, hi):
        if hi > lo:
            self.dev[lo:hi].copy_(self.host[lo:hi], non_blocking=True)
