"""Random-draw sources for the loops.  The reference draws from torch's global generator in a fixed
order (SURVEY.md 7 "Randomness parity"); GPU and CPU streams can never match, so every loop takes a
draw source: ``DeviceRNG`` (default, torch's device generator) or ``ReplayRNG`` (tests: replays draws
recorded from the CPU oracle in the same order)."""
import torch


class DeviceRNG:
    def __init__(self, device):
        self.device = device

    def rand(self, *shape):
        return torch.rand(*shape, device=self.device)

    def randn(self, *shape):
        return torch.randn(*shape, device=self.device)


class ReplayRNG:
    def __init__(self, draws, device):
        self.draws = list(draws)
        self.device = device
        self.pos = 0

    def _next(self, shape):
        t = self.draws[self.pos]
        self.pos += 1
        assert tuple(t.shape) == tuple(shape), f'replayed draw {self.pos - 1} has shape {tuple(t.shape)}, loop asked for {tuple(shape)}'
        return t.to(self.device)

    def rand(self, *shape):
        return self._next(shape)

    def randn(self, *shape):
        return self._next(shape)
