"""Stage timers of the drivers: `MH_TIMING=1 python PMVO.py ...` prints one line per stage (device-synchronised wall
time) on rank 0.  Off by default: no synchronisation is added to a normal run."""
import os
import sys
import time

ENABLED = bool(int(os.environ.get("MH_TIMING", "0") or 0))
totals = {}


class stage:
    def __init__(self, name, device=None):
        self.name, self.device = name, device

    def _sync(self):
        if self.device is not None:
            import torch

            if torch.cuda.is_available():
                torch.cuda.synchronize(self.device)

    def __enter__(self):
        if ENABLED:
            self._sync()
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if ENABLED:
            self._sync()
            dt = time.perf_counter() - self.t0
            totals[self.name] = totals.get(self.name, 0.0) + dt
            if int(os.environ.get("RANK", "0")) == 0:
                print("[mh-timing] %-28s %8.1f ms" % (self.name, dt * 1e3), file=sys.stderr, flush=True)
        return False
