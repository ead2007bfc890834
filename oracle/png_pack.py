"""CPU restatement of the on-device PNG packer (cool_chic_amd/csrc/ccd_png.hip), numpy + small Python loops.

TEST INFRASTRUCTURE: only tests/ and __graft_entry__.smoke() import this module; the product package never does.

The reference writes PNGs with PIL (coolchic/io/format/png.py:44-62: HWC uint8 -> Image.save), i.e. with zlib's
deflate.  PNG bytes are not normative - any conforming zlib stream of the filtered scanlines is the same picture - so
the device packer does not try to reproduce zlib's LZ77 choices.  Its parity bar has two parts:
  (1) the picture an independent decoder (PIL / zlib, the reference's own library) reads back is pixel-exact, and
  (2) the device bytes equal the bytes of THIS restatement (every step below is integer and deterministic).

Format produced (RFC 2083 / 1950 / 1951):
  signature, IHDR (8-bit RGB, no interlace), ONE IDAT holding a zlib stream (CMF/FLG 0x78 0x01) of the filtered
  scanlines, IEND.  Scanline filter: per row the one of None/Sub/Up/Average/Paeth with the smallest sum of absolute
  signed residuals (ties: lowest filter number).  Deflate: rows are grouped into blocks of `rows_per_block(w)` rows,
  each a dynamic-Huffman block (BTYPE 2) of literals only + end-of-block: HLIT = 257 codes, HDIST = 1 code of length 0,
  the code-length alphabet uses 4-bit codes for lengths 0..15 and no run-length symbols.  Literal code lengths: optimal
  (Moffat-Katajainen in-place construction on the symbols sorted by (count, symbol)), limited to 15 bits by moving the
  Kraft excess down from the longest codes, canonical code assignment.
"""
import struct
import zlib

import numpy as np

MAX_BITS = 15
HEADER_BITS = 3 + 5 + 5 + 4 + 19 * 3 + 258 * 4  # 1106 bits before the first literal of a block
CL_ORDER = (16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15)
BLOCK_TARGET = 32768  # bytes of filtered scanlines per deflate block (at least one row)
CRC_CHUNK = 512       # bytes per partial CRC on the device


def rows_per_block(w: int) -> int:
    return max(1, BLOCK_TARGET // (3 * w + 1))


def bound(h: int, w: int) -> int:
    """Capacity that always suffices: <= 10 bits per scanline byte + per-block header + container."""
    raw = h * (3 * w + 1)
    nblk = (h + rows_per_block(w) - 1) // rows_per_block(w)
    return (raw * 10 + 7) // 8 + nblk * 144 + 128


# ------------------------------------------------------------------------------------------------ filters
def filter_rows(img: np.ndarray):
    """img [H, W, 3] uint8 -> (filtered scanlines [H, 1 + 3W] uint8, filter type per row)."""
    h, w, _ = img.shape
    n = 3 * w
    raw = img.reshape(h, n).astype(np.int32)
    a = np.zeros_like(raw)
    a[:, 3:] = raw[:, :-3]                      # left
    b = np.zeros_like(raw)
    b[1:] = raw[:-1]                            # up
    c = np.zeros_like(raw)
    c[1:, 3:] = raw[:-1, :-3]                   # up-left
    p = a + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    paeth = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
    cands = np.stack([raw, raw - a, raw - b, raw - ((a + b) >> 1), raw - paeth]) & 255  # [5, H, n]
    cost = np.where(cands < 128, cands, 256 - cands).sum(axis=2)                          # [5, H]
    ftype = np.argmin(cost, axis=0)                                                       # first minimum = lowest number
    out = np.empty((h, n + 1), np.uint8)
    out[:, 0] = ftype
    out[:, 1:] = cands[ftype, np.arange(h)]
    return out, ftype


# ------------------------------------------------------------------------------------------------ Huffman
def code_lengths(hist) -> np.ndarray:
    """hist[257] counts (EOB included) -> code length per symbol (0 = unused), <= 15 bits, complete code."""
    used = [(int(cnt), s) for s, cnt in enumerate(hist) if cnt > 0]
    used.sort()
    m = len(used)
    assert m >= 2
    A = [cnt for cnt, _ in used]
    # Moffat-Katajainen: in-place minimum-redundancy code lengths of ascending counts
    A[0] += A[1]
    root, leaf = 0, 2
    for nxt in range(1, m - 1):
        if leaf >= m or A[root] < A[leaf]:
            A[nxt] = A[root]
            A[root] = nxt
            root += 1
        else:
            A[nxt] = A[leaf]
            leaf += 1
        if leaf >= m or (root < nxt and A[root] < A[leaf]):
            A[nxt] += A[root]
            A[root] = nxt
            root += 1
        else:
            A[nxt] += A[leaf]
            leaf += 1
    A[m - 2] = 0
    for nxt in range(m - 3, -1, -1):
        A[nxt] = A[A[nxt]] + 1
    avbl, usedn, dpth, root, nxt = 1, 0, 0, m - 2, m - 1
    while avbl > 0:
        while root >= 0 and A[root] == dpth:
            usedn += 1
            root -= 1
        while avbl > usedn:
            A[nxt] = dpth
            nxt -= 1
            avbl -= 1
        avbl = 2 * usedn
        dpth += 1
        usedn = 0
    # limit to MAX_BITS: codes longer than the limit are folded into it, then the Kraft excess is worked off
    num = [0] * (MAX_BITS + 1)
    for ln in A:
        num[min(ln, MAX_BITS)] += 1
    total = sum(num[ln] << (MAX_BITS - ln) for ln in range(1, MAX_BITS + 1))
    while total != 1 << MAX_BITS:
        num[MAX_BITS] -= 1
        for ln in range(MAX_BITS - 1, 0, -1):
            if num[ln]:
                num[ln] -= 1
                num[ln + 1] += 2
                break
        total -= 1
    lens = np.zeros(257, np.int32)
    j = 0
    for ln in range(MAX_BITS, 0, -1):  # rarest symbols take the longest codes
        for _ in range(num[ln]):
            lens[used[j][1]] = ln
            j += 1
    assert j == m
    return lens


def _rev(v: int, nbits: int) -> int:
    r = 0
    for _ in range(nbits):
        r = (r << 1) | (v & 1)
        v >>= 1
    return r


def canonical_codes(lens: np.ndarray) -> np.ndarray:
    """RFC 1951 3.2.2 codes, bit-reversed (deflate packs Huffman codes starting from their most significant bit)."""
    bl_count = np.bincount(lens, minlength=MAX_BITS + 1)
    next_code = [0] * (MAX_BITS + 2)
    code = 0
    for bits in range(1, MAX_BITS + 1):
        code = (code + (bl_count[bits - 1] if bits > 1 else 0)) << 1
        next_code[bits] = code
    codes = np.zeros(257, np.int64)
    for s in range(257):
        ln = int(lens[s])
        if ln:
            codes[s] = _rev(next_code[ln], ln)
            next_code[ln] += 1
    return codes


# ------------------------------------------------------------------------------------------------ bit packing
class _Bits:
    def __init__(self, n_bits_cap: int):
        self.bits = np.zeros(n_bits_cap, np.uint8)
        self.pos = 0

    def put(self, value: int, nbits: int):
        for k in range(nbits):
            self.bits[self.pos + k] = (value >> k) & 1
        self.pos += nbits

    def put_many(self, codes: np.ndarray, lens: np.ndarray):
        starts = self.pos + np.concatenate([[0], np.cumsum(lens)[:-1]])
        for k in range(MAX_BITS):
            sel = lens > k
            self.bits[starts[sel] + k] = (codes[sel] >> k) & 1
        self.pos += int(lens.sum())


def deflate_huffman_only(scan: np.ndarray, w: int):
    """Filtered scanlines [H, 1+3W] -> deflate bytes (padded to a byte), bits per block."""
    h = scan.shape[0]
    R = rows_per_block(w)
    nblk = (h + R - 1) // R
    out = _Bits(scan.size * 15 + nblk * (HEADER_BITS + 15) + 64)
    blk_bits = []
    for k in range(nblk):
        data = scan[k * R:(k + 1) * R].reshape(-1)
        hist = np.bincount(data, minlength=257)
        hist[256] = 1
        lens = code_lengths(hist)
        codes = canonical_codes(lens)
        p0 = out.pos
        out.put(1 if k == nblk - 1 else 0, 1)
        out.put(2, 2)
        out.put(0, 5)
        out.put(0, 5)
        out.put(15, 4)
        for s in CL_ORDER:
            out.put(0 if s >= 16 else 4, 3)
        for s in range(258):  # 257 literal/length lengths + the single distance code (length 0)
            out.put(_rev(int(lens[s]) if s < 257 else 0, 4), 4)
        out.put_many(codes[data], lens[data])
        out.put(int(codes[256]), int(lens[256]))
        blk_bits.append(out.pos - p0)
        assert blk_bits[-1] == HEADER_BITS + int((hist * lens).sum())
    n_bytes = (out.pos + 7) // 8
    return np.packbits(out.bits[: n_bytes * 8], bitorder="little").tobytes(), blk_bits


# ------------------------------------------------------------------------------------------------ checksums as the device computes them
_POLY = 0xEDB88320


def _multmodp(a: int, b: int) -> int:
    """a * b modulo the CRC-32 polynomial, reflected representation (bit 31 = x^0)."""
    m, p = 1 << 31, 0
    while True:
        if a & m:
            p ^= b
            if (a & (m - 1)) == 0:
                break
        m >>= 1
        b = (b >> 1) ^ _POLY if b & 1 else b >> 1
    return p


def x2n_table():
    t = [1 << 30]
    for _ in range(31):
        t.append(_multmodp(t[-1], t[-1]))
    return t


def _x_pow_bytes(n: int, tab) -> int:
    """x^(8 n) mod P."""
    p, k = 1 << 31, 3
    while n:
        if n & 1:
            p = _multmodp(tab[k & 31], p)
        n >>= 1
        k += 1
    return p


def crc32_chunked(data: bytes, chunk: int = CRC_CHUNK) -> int:
    """CRC-32 as XOR of the chunk CRCs, each multiplied by x^(8 * bytes behind the chunk)."""
    tab = x2n_table()
    n, total = len(data), 0
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        total ^= _multmodp(_x_pow_bytes(n - e, tab), zlib.crc32(data[s:e]))
    return total


def adler32_rows(scan: np.ndarray) -> int:
    """Adler-32 from per-row sums (a_r = sum d_i, b_r = sum (n - i) d_i), combined row after row."""
    n = scan.shape[1]
    wts = np.arange(n, 0, -1, dtype=np.int64)
    A, B = 1, 0
    for row in scan.astype(np.int64):
        a_r, b_r = int(row.sum()) % 65521, int((row * wts).sum()) % 65521
        B = (B + n * A + b_r) % 65521
        A = (A + a_r) % 65521
    return (B << 16) | A


# ------------------------------------------------------------------------------------------------ container
def _chunk(kind: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", crc32_chunked(kind + data))


def pack_rgb8(planes) -> bytes:
    """planes: [3, H, W] uint8 (r, g, b) -> PNG bytes."""
    planes = np.asarray(planes)
    assert planes.dtype == np.uint8 and planes.ndim == 3 and planes.shape[0] == 3
    _, h, w = planes.shape
    img = np.ascontiguousarray(planes.transpose(1, 2, 0))
    scan, _ = filter_rows(img)
    body, _ = deflate_huffman_only(scan, w)
    z = b"\x78\x01" + body + struct.pack(">I", adler32_rows(scan))
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", z) + _chunk(b"IEND", b"")
    assert len(png) <= bound(h, w)
    return png
