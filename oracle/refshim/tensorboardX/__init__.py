"""Stub SummaryWriter: records scalars in memory (they are the PPO loss parity probes)."""


class SummaryWriter:
    def __init__(self, log_dir=None, **kw):
        self.log_dir = log_dir
        self.scalars = {}

    def add_scalar(self, tag, value, step=None):
        self.scalars.setdefault(tag, []).append((step, float(value)))

    def close(self):
        pass
