"""
Page-locked host memory for the results of a run on ONE rank.

Every run delivers its nav results into page-locked memory (D2H copies run at link speed only
from / to page-locked buffers; small write-once results are even written there by the kernels
themselves).  Page-locking a fresh 256 MiB buffer costs ~16 ms -- three times the C4 job -- and
torch's caching host allocator hands a freed block back only a run or two later, so the executor
keeps its own ring of buffers and reuses one as soon as nobody references the results that live in
it: the arrays a run hands out are views of a per-run owner object (`_Owner`, see
executor/nodeshared.py), a slot is free when its owner is dead.  The results behave like
caller-owned arrays: valid for as long as any view of them is referenced.
"""
import weakref

import numpy as np

from .nodeshared import _Owner


class PinnedRing:
    MAX_SLOTS = 8

    def __init__(self, torch, dev_ptr_fn=None):
        self.torch = torch
        self.dev_ptr_fn = dev_ptr_fn    # host address -> device address (raises if not mapped)
        self.slots = []                 # [tensor (uint8, pinned), owner weakref | None, dev ptr]

    def _free(self, slot):
        ref = slot[1]
        return ref is None or ref() is None

    def get(self, nbytes):
        """-> (uint8 torch tensor of >= nbytes, `_Owner` NumPy view of it, device address | None).
        Views derived from the owner keep the slot reserved."""
        nbytes = max(int(nbytes), 1)
        best = None
        for slot in self.slots:
            if self._free(slot) and slot[0].numel() >= nbytes:
                if best is None or slot[0].numel() < best[0].numel():
                    best = slot
        if best is None:
            cap = max(1 << 21, (nbytes + (1 << 21) - 1) >> 21 << 21)
            tensor = self.torch.empty((cap,), dtype=self.torch.uint8, pin_memory=True)
            dev = None
            if self.dev_ptr_fn is not None:
                try:
                    dev = self.dev_ptr_fn(tensor.data_ptr())
                except Exception:
                    dev = None          # not device-accessible: results are copied out instead
            best = [tensor, None, dev]
            # replace a free (too small) slot rather than growing without bound
            for k, slot in enumerate(self.slots):
                if self._free(slot):
                    self.slots[k] = best
                    break
            else:
                if len(self.slots) < self.MAX_SLOTS:
                    self.slots.append(best)
                # else: an untracked buffer, released by torch when its views die
        owner = best[0].numpy().view(_Owner)
        best[1] = weakref.ref(owner)
        return best[0], owner, best[2]

    def close(self):
        self.slots = []
