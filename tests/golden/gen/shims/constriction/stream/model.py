"""Leaky quantised Laplace model of constriction 0.4.2, restated."""
import math

PRECISION = 24


class QuantizedLaplace:
    def __init__(self, lo, hi):
        self.lo = int(lo)
        self.hi = int(hi)
        self.free = float((1 << PRECISION) - 1 - (self.hi - self.lo))

    @staticmethod
    def _cdf(x, mu, b):
        if x <= mu:
            return 0.5 * math.exp((x - mu) / b)
        return 1.0 - 0.5 * math.exp((mu - x) / b)

    def left(self, s, mu, b):
        if s == self.lo:
            return 0
        return int(self.free * self._cdf(s - 0.5, mu, b)) + (s - self.lo)

    def right(self, s, mu, b):
        if s == self.hi:
            return 1 << PRECISION
        return int(self.free * self._cdf(s + 0.5, mu, b)) + (s - self.lo) + 1
