"""Minimal OpenEXR scanline reader (HALF / FLOAT channels; NONE, ZIPS/ZIP and PIZ compression) for
the ground-truth images the reference compares against (main.py:38-41: `mi.Bitmap(TungstenRender.exr)`).
No EXR library exists in the target image, so the PIZ path (Huffman + 2-D wavelet + LUT, as
specified by the OpenEXR file format / ImfPizCompressor) is implemented here with numpy.

Host-side file IO only; not on the hot path.
"""
from __future__ import annotations

import struct
import zlib
from typing import Dict, Tuple

import numpy as np

_MAGIC = 20000630
_PIXEL_SIZE = {0: 4, 1: 2, 2: 4}  # UINT, HALF, FLOAT


def _read_header(buf: bytes):
    magic, version = struct.unpack_from("<ii", buf, 0)
    if magic != _MAGIC:
        raise ValueError("not an OpenEXR file")
    if version & 0x200 or version & 0x1000 or version & 0x800:
        raise ValueError("tiled / multipart / deep EXR files are not supported")
    pos = 8
    attrs = {}
    while True:
        end = buf.index(b"\0", pos)
        name = buf[pos:end].decode()
        pos = end + 1
        if not name:
            break
        end = buf.index(b"\0", pos)
        typ = buf[pos:end].decode()
        pos = end + 1
        (size,) = struct.unpack_from("<i", buf, pos)
        pos += 4
        attrs[name] = (typ, buf[pos:pos + size])
        pos += size
    return attrs, pos


def _channels(raw: bytes):
    out, pos = [], 0
    while raw[pos] != 0:
        end = raw.index(b"\0", pos)
        name = raw[pos:end].decode()
        pos = end + 1
        ptype, _plinear, xs, ys = struct.unpack_from("<iB3xii", raw, pos)
        pos += 16
        if xs != 1 or ys != 1:
            raise ValueError("subsampled channels are not supported")
        out.append((name, ptype))
    return out


# ---- PIZ -----------------------------------------------------------------------------------------
_HUF_ENCSIZE = (1 << 16) + 1
_SHORT_ZEROCODE_RUN, _LONG_ZEROCODE_RUN = 59, 63
_SHORTEST_LONG_RUN = 2 + _LONG_ZEROCODE_RUN - _SHORT_ZEROCODE_RUN


class _Bits:
    """MSB-first bit reader."""

    def __init__(self, data: bytes, pos: int = 0):
        self.d, self.p, self.c, self.lc = data, pos, 0, 0

    def get(self, n: int) -> int:
        while self.lc < n:
            self.c = (self.c << 8) | self.d[self.p]
            self.p += 1
            self.lc += 8
        self.lc -= n
        v = (self.c >> self.lc) & ((1 << n) - 1)
        self.c &= (1 << self.lc) - 1
        return v


def _huf_uncompress(data: bytes, n_out: int) -> np.ndarray:
    im, iM, _table_len, nbits, _ = struct.unpack_from("<iiiii", data, 0)
    if not (0 <= im < _HUF_ENCSIZE and 0 <= iM < _HUF_ENCSIZE):
        raise ValueError("corrupt PIZ Huffman header")
    # packed code-length table
    br = _Bits(data, 20)
    lengths = np.zeros(_HUF_ENCSIZE, np.int64)
    s = im
    while s <= iM:
        l = br.get(6)
        if l == _LONG_ZEROCODE_RUN:
            s += br.get(8) + _SHORTEST_LONG_RUN
        elif l >= _SHORT_ZEROCODE_RUN:
            s += l - _SHORT_ZEROCODE_RUN + 2
        else:
            lengths[s] = l
            s += 1
    start = br.p  # the bit stream starts at the next byte boundary after the table
    # canonical codes (hufCanonicalCodeTable)
    n = np.bincount(lengths, minlength=59).astype(np.int64)
    n[0] = 0
    base = np.zeros(59, np.int64)
    c = 0
    for i in range(58, 0, -1):
        nc = (c + n[i]) >> 1
        base[i] = c
        c = nc
    codes = {}
    nxt = base.copy()
    for sym in np.nonzero(lengths)[0]:
        l = int(lengths[sym])
        codes[(l, int(nxt[l]))] = int(sym)
        nxt[l] += 1
    min_len = int(lengths[lengths > 0].min()) if codes else 0
    # 12-bit prefix table for speed: prefix -> (symbol, length) when a code of length <= 12 matches
    FAST = 12
    fast = [None] * (1 << FAST)
    for (l, code), sym in codes.items():
        if l <= FAST:
            lo = code << (FAST - l)
            for k in range(lo, lo + (1 << (FAST - l))):
                fast[k] = (sym, l)
    out = np.zeros(n_out, np.uint16)
    d = data
    nbytes = len(d)
    pos, acc, lc = start, 0, 0   # acc holds `lc` not yet consumed bits, MSB first
    consumed, o, rlc = 0, 0, iM
    while consumed < nbits and o < n_out:
        while lc < 58:               # keep a full look-ahead; past the end the stream is zero-padded
            acc = (acc << 8) | (d[pos] if pos < nbytes else 0)
            pos += 1
            lc += 8
        e = fast[(acc >> (lc - FAST)) & ((1 << FAST) - 1)]
        if e is not None:
            sym, l = e
        else:
            sym, l = None, FAST + 1
            while l <= 58:
                sym = codes.get((l, (acc >> (lc - l)) & ((1 << l) - 1)))
                if sym is not None:
                    break
                l += 1
            if sym is None:
                raise ValueError("corrupt PIZ Huffman stream")
        lc -= l
        acc &= (1 << lc) - 1
        consumed += l
        if sym == rlc:
            run = (acc >> (lc - 8)) & 0xFF
            lc -= 8
            acc &= (1 << lc) - 1
            consumed += 8
            if o == 0 or o + run > n_out:
                raise ValueError("corrupt PIZ run")
            out[o:o + run] = out[o - 1]
            o += run
        else:
            out[o] = sym
            o += 1
    if o != n_out:
        raise ValueError(f"PIZ Huffman stream ended early ({o} of {n_out})")
    return out


def _wdec14(l, h):
    ls = l.astype(np.int16).astype(np.int32)
    hs = h.astype(np.int16).astype(np.int32)
    ai = ls + (hs & 1) + (hs >> 1)
    a = ai.astype(np.int16)
    b = (ai - hs).astype(np.int16)
    return a.view(np.uint16), b.view(np.uint16)


def _wdec16(l, h):
    m = l.astype(np.int32)
    d = h.astype(np.int32)
    bb = (m - (d >> 1)) & 0xFFFF
    aa = (d + bb - 0x8000) & 0xFFFF
    return aa.astype(np.uint16), bb.astype(np.uint16)


def _wav2_decode(a: np.ndarray, mx: int) -> None:
    """In-place inverse wavelet of a (ny, nx) uint16 plane (ImfWav.cpp wav2Decode)."""
    ny, nx = a.shape
    dec = _wdec14 if mx < (1 << 14) else _wdec16
    n = min(nx, ny)
    p = 1
    while p <= n:
        p <<= 1
    p >>= 1
    p2 = p
    p >>= 1
    while p >= 1:
        ys = np.arange(0, ny - p2 + 1, p2) if ny - p2 >= 0 else np.arange(0)
        xs = np.arange(0, nx - p2 + 1, p2) if nx - p2 >= 0 else np.arange(0)
        if len(ys) and len(xs):
            Y, X = np.ix_(ys, xs)
            p00, p10, p01, p11 = a[Y, X], a[Y + p, X], a[Y, X + p], a[Y + p, X + p]
            i00, i10 = dec(p00, p10)
            i01, i11 = dec(p01, p11)
            a[Y, X], a[Y, X + p] = dec(i00, i01)
            a[Y + p, X], a[Y + p, X + p] = dec(i10, i11)
        if nx & p and len(ys):  # odd column: the x position right after the last full 2x2 cell
            x = len(xs) * p2
            Y = ys[:, None]
            i00, b = dec(a[Y, x], a[Y + p, x])
            a[Y + p, x] = b
            a[Y, x] = i00
        if ny & p and len(xs):  # odd row
            y = len(ys) * p2
            X = xs[None, :]
            i00, b = dec(a[y, X], a[y, X + p])
            a[y, X + p] = b
            a[y, X] = i00
        p2 = p
        p >>= 1


def _piz_block(data: bytes, chans, nx: int, ny: int) -> bytes:
    sizes = [_PIXEL_SIZE[t] // 2 for _, t in chans]  # in uint16 units
    total = sum(s * nx * ny for s in sizes)
    min_nz, max_nz = struct.unpack_from("<HH", data, 0)
    pos = 4
    bitmap = np.zeros(8192, np.uint8)
    if min_nz <= max_nz:
        cnt = max_nz - min_nz + 1
        bitmap[min_nz:min_nz + cnt] = np.frombuffer(data, np.uint8, cnt, pos)
        pos += cnt
    (length,) = struct.unpack_from("<i", data, pos)
    pos += 4
    tmp = _huf_uncompress(data[pos:pos + length], total)
    # reverse LUT
    bits = np.unpackbits(bitmap, bitorder="little").astype(bool)
    bits[0] = True
    lut = np.zeros(65536, np.uint16)
    vals = np.nonzero(bits)[0].astype(np.uint16)
    lut[:len(vals)] = vals
    max_value = len(vals) - 1
    # wavelet per channel (and per 16-bit half of 32-bit channels)
    off = 0
    planes = []
    for s in sizes:
        ch = tmp[off:off + s * nx * ny].reshape(ny, nx * s)
        for j in range(s):
            sub = np.ascontiguousarray(ch[:, j::s])
            _wav2_decode(sub, max_value)
            ch[:, j::s] = sub
        planes.append(ch)
        off += s * nx * ny
    out = np.empty((ny, sum(s * nx for s in sizes)), np.uint16)
    col = 0
    for ch, s in zip(planes, sizes):
        out[:, col:col + s * nx] = lut[ch]
        col += s * nx
    return out.astype("<u2").tobytes()


def _undo_zip_predictor(raw: bytes) -> bytes:
    a = np.frombuffer(raw, np.uint8).astype(np.int32)
    a = (np.cumsum(np.concatenate([[a[0]], a[1:] - 128])) & 0xFF).astype(np.uint8)
    half = (len(a) + 1) // 2
    out = np.empty(len(a), np.uint8)
    out[0::2] = a[:half]
    out[1::2] = a[half:]
    return out.tobytes()


def read_exr(path: str) -> Tuple[Dict[str, np.ndarray], Tuple[int, int]]:
    """Returns ({channel name: float32 (H, W)}, (width, height))."""
    buf = open(path, "rb").read()
    attrs, pos = _read_header(buf)
    chans = _channels(attrs["channels"][1])
    comp = attrs["compression"][1][0]
    x0, y0, x1, y1 = struct.unpack("<iiii", attrs["dataWindow"][1])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    lines = {0: 1, 2: 1, 3: 16, 4: 32}.get(comp)
    if lines is None:
        raise ValueError(f"EXR compression {comp} is not supported (NONE, ZIPS, ZIP, PIZ are)")
    n_chunks = (H + lines - 1) // lines
    offsets = struct.unpack_from(f"<{n_chunks}Q", buf, pos)
    line_bytes = sum(_PIXEL_SIZE[t] for _, t in chans) * W
    out = {name: np.empty((H, W), np.float32) for name, _ in chans}
    for off in offsets:
        y, size = struct.unpack_from("<ii", buf, off)
        data = buf[off + 8:off + 8 + size]
        ny = min(lines, y1 - y + 1)
        want = line_bytes * ny
        if size != want:
            if comp in (2, 3):
                data = _undo_zip_predictor(zlib.decompress(data))
            elif comp == 4:
                data = _piz_block(data, chans, W, ny)
        rows = np.frombuffer(data, np.uint8).reshape(ny, line_bytes)
        col = 0
        for name, t in chans:
            nb = _PIXEL_SIZE[t] * W
            seg = np.ascontiguousarray(rows[:, col:col + nb])
            if t == 1:
                out[name][y - y0:y - y0 + ny] = seg.view("<f2").astype(np.float32)
            elif t == 2:
                out[name][y - y0:y - y0 + ny] = seg.view("<f4")
            else:
                out[name][y - y0:y - y0 + ny] = seg.view("<u4").astype(np.float32)
            col += nb
    return out, (W, H)


def read_rgb(path: str) -> np.ndarray:
    """(H, W, 3) float32 linear RGB, as `mi.TensorXf(mi.Bitmap(path))` yields it (main.py:38-41)."""
    ch, _ = read_exr(path)
    return np.stack([ch["R"], ch["G"], ch["B"]], axis=2)


def _zip_predictor(raw: bytes) -> bytes:
    """The inverse of _undo_zip_predictor: even bytes then odd bytes, then byte differences + 128."""
    a = np.frombuffer(raw, np.uint8)
    t = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
    d = np.empty_like(t)
    d[0] = t[0]
    d[1:] = t[1:] - t[:-1] + 128
    return (d & 0xFF).astype(np.uint8).tobytes()


def _attr(name: str, typ: str, payload: bytes) -> bytes:
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload


def write_rgb(path: str, image: np.ndarray, half: bool = False, compression: str = "zip") -> None:
    """Writes an (H, W, 3) linear-RGB image as a scanline OpenEXR file, channels B, G, R (FLOAT, or HALF
    with `half`), ZIP (16 scanlines per chunk) or uncompressed -- the `.exr` that
    `mi.util.write_bitmap(imageFileName + '.exr', image)` leaves next to each `.png` (main.py:278, 401).
    read_rgb() reads it back bit for bit (FLOAT) or to half precision (HALF)."""
    img = np.ascontiguousarray(np.asarray(image, np.float32))
    if img.ndim != 3 or img.shape[2] != 3:
        raise ValueError("image must be (H, W, 3)")
    comp = {"none": 0, "zip": 3}.get(compression)
    if comp is None:
        raise ValueError("compression must be 'none' or 'zip'")
    H, W = img.shape[:2]
    ptype, dt = (1, "<f2") if half else (2, "<f4")
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", ptype, 0, 1, 1) for n in ("B", "G", "R")) + b"\0"
    box = struct.pack("<iiii", 0, 0, W - 1, H - 1)
    header = struct.pack("<ii", _MAGIC, 2)
    header += _attr("channels", "chlist", chlist)
    header += _attr("compression", "compression", bytes([comp]))
    header += _attr("dataWindow", "box2i", box)
    header += _attr("displayWindow", "box2i", box)
    header += _attr("lineOrder", "lineOrder", b"\0")
    header += _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    header += _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0))
    header += _attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    header += b"\0"
    lines = 16 if comp == 3 else 1
    chunks = []
    for y in range(0, H, lines):
        rows = img[y:y + lines]
        # per scanline: all of B, then all of G, then all of R (channels in alphabetical order)
        raw = np.concatenate([rows[:, :, c].astype(dt).view(np.uint8).reshape(rows.shape[0], -1) for c in (2, 1, 0)],
                             axis=1).tobytes()
        data = raw
        if comp == 3:
            z = zlib.compress(_zip_predictor(raw), 6)
            if len(z) < len(raw):  # (a chunk that does not shrink is stored raw, as the format prescribes)
                data = z
        chunks.append(struct.pack("<ii", y, len(data)) + data)
    table_at = len(header)
    pos = table_at + 8 * len(chunks)
    offsets = []
    for c in chunks:
        offsets.append(pos)
        pos += len(c)
    with open(path, "wb") as f:
        f.write(header)
        f.write(struct.pack(f"<{len(offsets)}Q", *offsets))
        for c in chunks:
            f.write(c)
