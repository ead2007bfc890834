"""Range coder of constriction 0.4.2 (stream.queue), restated."""
import numpy as np

P = 24
M64 = (1 << 64) - 1


class RangeDecoder:
    def __init__(self, words):
        self.w = [int(x) for x in np.asarray(words, dtype=np.uint32)]
        self.pos = 0
        self.lower = 0
        self.range = M64
        self.point = (self._next() << 32) | self._next()
        self.n_read_past_end = 0

    def _next(self):
        if self.pos < len(self.w):
            v = self.w[self.pos]
        else:
            v = 0
        self.pos += 1
        return v

    def decode(self, model, mus, scales):
        mus = np.asarray(mus, dtype=np.float32)
        scales = np.asarray(scales, dtype=np.float32)
        out = np.empty(mus.shape[0], dtype=np.int32)
        for i in range(mus.shape[0]):
            mu = float(mus[i])
            b = float(scales[i])
            scale = self.range >> P
            q = ((self.point - self.lower) & M64) // scale
            if q >= (1 << P):
                raise ValueError("invalid compressed data")
            lo, hi = model.lo, model.hi
            # bisection on the left cumulative
            a, c = lo, hi
            while a < c:
                m = (a + c + 1) >> 1
                if model.left(m, mu, b) <= q:
                    a = m
                else:
                    c = m - 1
            s = a
            left = model.left(s, mu, b)
            right = model.right(s, mu, b)
            assert left <= q < right
            self.lower = (self.lower + scale * left) & M64
            self.range = scale * (right - left)
            if self.range < (1 << 32):
                self.lower = (self.lower << 32) & M64
                self.range = (self.range << 32) & M64
                self.point = ((self.point << 32) & M64) | self._next()
            out[i] = s
        return out


class RangeEncoder:
    def __init__(self):
        self.lower = 0
        self.range = M64
        self.inv = None  # (n, first)
        self.out = []
        self.any = False

    def _flush_inv(self, carry):
        n, first = self.inv
        self.out.append((first + 1) & 0xFFFFFFFF if carry else first)
        for _ in range(n - 1):
            self.out.append(0 if carry else 0xFFFFFFFF)
        self.inv = None

    def encode(self, xs, model, mus, scales):
        xs = np.asarray(xs, dtype=np.int32)
        mus = np.asarray(mus, dtype=np.float32)
        scales = np.asarray(scales, dtype=np.float32)
        for i in range(xs.shape[0]):
            s = int(xs[i])
            mu = float(mus[i])
            b = float(scales[i])
            left = model.left(s, mu, b)
            right = model.right(s, mu, b)
            self.any = True
            scale = self.range >> P
            self.range = scale * (right - left)
            new = (self.lower + scale * left) & M64
            if self.inv is not None and ((new + self.range) & M64) > new:
                self._flush_inv(new < self.lower)
            self.lower = new
            if self.range < (1 << 32):
                word = self.lower >> 32
                self.lower = (self.lower << 32) & M64
                self.range = (self.range << 32) & M64
                if self.inv is not None:
                    self.inv = (self.inv[0] + 1, self.inv[1])
                elif ((self.lower + self.range) & M64) > self.lower:
                    self.out.append(word)
                else:
                    self.inv = (1, word)

    def get_compressed(self):
        """Seal a copy of the state and return the words (state left untouched)."""
        if not self.any:
            return np.zeros(0, dtype=np.uint32)
        out = list(self.out)
        lower, rng, inv = self.lower, self.range, self.inv
        point = (lower + (1 << 32) - 1) & M64
        if inv is not None:
            n, first = inv
            carry = point < lower
            out.append((first + 1) & 0xFFFFFFFF if carry else first)
            for _ in range(n - 1):
                out.append(0 if carry else 0xFFFFFFFF)
        pw = point >> 32
        out.append(pw)
        if (((lower + rng) & M64) >> 32) == pw:
            out.append(0)
        return np.asarray(out, dtype=np.uint32)
