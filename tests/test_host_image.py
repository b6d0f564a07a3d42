"""TGA/PNG decode and encode of the host library.

Test files are produced by small encoders written here (struct + zlib), so the expected texels are
known exactly; Pillow, when importable, is a second independent decoder.  The reference's own data
files (tests/golden/images/bth.tga, assets/sword/sword.png) are decoded and compared with arrays that
tests/golden/images/make_image_golden.py stored from Pillow's decode of the same files.

Parity unpinned against the decoder the reference links (SOIL's stb_image_aug.c): it cannot be run
here, so its TGA conventions (16 bits = grey + alpha, palette depth from the colour map, the
right-to-left bit ignored) are restated from its source and tested as stated."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(HERE, "golden", "images")


@pytest.fixture(scope="module")
def host():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "opengl_raytracer_amd", "csrc", "host")])
    from opengl_raytracer_amd import host as h
    h.load()
    return h


# ---------------------------------------------------------------------------------- TGA encoder
def tga_bytes(pixels, kind, rle=False, top=False, id_field=b"", palette=None, rs=None):
    """pixels: (h, w, c) uint8 in file channel order (c = 1, 2, 3 BGR, 4 BGRA), rows top-down.
    kind 2 truecolour, 3 grey, 1 colour-mapped (pixels are indices, palette (n, 3|4) in file order)."""
    h, w, c = pixels.shape
    rows = pixels if top else pixels[::-1]
    flat = rows.reshape(-1, c)
    if rle:
        body = bytearray()
        i, n = 0, flat.shape[0]
        while i < n:  # packets may cross rows, as the format allows
            run = 1
            while i + run < n and run < 128 and (flat[i + run] == flat[i]).all():
                run += 1
            if run > 1 and (rs is None or rs.rand() < 0.9):
                body += bytes([128 | (run - 1)]) + flat[i].tobytes()
                i += run
            else:
                k = 1 if rs is None else int(rs.randint(1, 20))
                k = min(k, n - i, 128)
                body += bytes([k - 1]) + flat[i:i + k].tobytes()
                i += k
        body = bytes(body)
    else:
        body = flat.tobytes()
    cmap = b"" if palette is None else palette.tobytes()
    head = struct.pack("<BBBHHBHHHHBB", len(id_field), 0 if palette is None else 1, kind + (8 if rle else 0),
                       0, 0 if palette is None else palette.shape[0], 0 if palette is None else palette.shape[1] * 8,
                       0, 0, w, h, 8 if palette is not None else c * 8, (0x20 if top else 0) | (8 if c == 4 else 0))
    return head + id_field + cmap + body


def expected_rgba(pixels):
    h, w, c = pixels.shape
    out = np.empty((h, w, 4), dtype=np.uint8)
    if c == 1:
        out[..., :3] = pixels
        out[..., 3] = 255
    elif c == 2:
        out[..., :3] = pixels[..., :1]
        out[..., 3] = pixels[..., 1]
    elif c == 3:
        out[..., :3] = pixels[..., ::-1]
        out[..., 3] = 255
    else:
        out[..., :3] = pixels[..., 2::-1]
        out[..., 3] = pixels[..., 3]
    return out


def blocky(rs, h, w, c):
    """Random texels with runs, so RLE has both packet kinds."""
    a = rs.randint(0, 256, size=(h, w, c)).astype(np.uint8)
    for _ in range(6):
        y, x = rs.randint(0, h), rs.randint(0, w)
        a[y:y + rs.randint(1, 4), x:x + rs.randint(2, 40)] = rs.randint(0, 256, size=c)
    return a


@pytest.mark.parametrize("c", [1, 2, 3, 4])
@pytest.mark.parametrize("rle", [False, True])
@pytest.mark.parametrize("top", [False, True])
def test_tga_truecolour_and_grey(host, c, rle, top):
    rs = np.random.RandomState(c * 4 + rle * 2 + top)
    px = blocky(rs, 13, 37, c)
    data = tga_bytes(px, 3 if c <= 2 else 2, rle=rle, top=top, id_field=b"hello" if c == 3 else b"", rs=rs)
    got = host.decode_image(data, invert_y=False)
    assert np.array_equal(got, expected_rgba(px))
    assert np.array_equal(host.decode_image(data, invert_y=True), expected_rgba(px)[::-1])


@pytest.mark.parametrize("entry", [3, 4])
@pytest.mark.parametrize("rle", [False, True])
def test_tga_colour_mapped(host, entry, rle):
    rs = np.random.RandomState(entry + rle)
    pal = rs.randint(0, 256, size=(17, entry)).astype(np.uint8)
    idx = blocky(rs, 9, 21, 1) % 20  # some indices beyond the palette: they read entry 0
    data = tga_bytes(idx, 1, rle=rle, palette=pal, rs=rs)
    safe = np.where(idx[..., 0] < 17, idx[..., 0], 0)
    assert np.array_equal(host.decode_image(data, invert_y=False), expected_rgba(pal[safe]))


def test_tga_file_round_trip_and_pillow(host, tmp_path):
    rs = np.random.RandomState(5)
    img = blocky(rs, 40, 64, 4)  # RGBA, row 0 = bottom
    for rle in (False, True):
        p = str(tmp_path / f"f{int(rle)}.tga")
        host.write_tga(p, img, rle=rle)
        assert np.array_equal(host.load_image(p, invert_y=True), img)
        assert np.array_equal(host.load_image(p, invert_y=False), img[::-1])
        PIL = pytest.importorskip("PIL.Image")
        assert np.array_equal(np.array(PIL.open(p).convert("RGBA")), img[::-1])
    assert os.path.getsize(str(tmp_path / "f1.tga")) < os.path.getsize(str(tmp_path / "f0.tga")) + 700


# ---------------------------------------------------------------------------------- PNG encoder
def png_chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)


def pack_samples(samples, depth):
    """(h, w, ch) integer samples -> list of packed scanline bytes."""
    h, w, ch = samples.shape
    rows = []
    for y in range(h):
        s = samples[y].reshape(-1)
        if depth == 8:
            rows.append(s.astype(np.uint8).tobytes())
        elif depth == 16:
            rows.append(s.astype(">u2").tobytes())
        else:
            bits = np.zeros(((len(s) * depth + 7) // 8) * 8, dtype=np.uint8)
            for b in range(depth):
                bits[np.arange(len(s)) * depth + b] = (s >> (depth - 1 - b)) & 1
            rows.append(np.packbits(bits).tobytes())
    return rows


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def filter_rows(rows, bpp, rs):
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for row in rows:
        ft = int(rs.randint(0, 5))
        out.append(ft)
        for k, x in enumerate(row):
            a = row[k - bpp] if k >= bpp else 0
            b = prev[k]
            c = prev[k - bpp] if k >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, paeth(a, b, c))[ft]
            out.append((x - pred) & 255)
        prev = row
    return bytes(out)


ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]


def png_bytes(samples, color, depth, rs, interlace=False, palette=None, trns=None, level=6, split=1):
    h, w, ch = samples.shape
    bpp = max(1, ch * depth // 8)
    if interlace:
        raw = b""
        for x0, y0, dx, dy in ADAM7:
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += filter_rows(pack_samples(sub, depth), bpp, rs)
    else:
        raw = filter_rows(pack_samples(samples, depth), bpp, rs)
    z = zlib.compress(raw, level)
    out = b"\x89PNG\r\n\x1a\n" + png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, int(interlace)))
    out += png_chunk(b"gAMA", struct.pack(">I", 45455))  # an ancillary chunk to skip
    if palette is not None:
        out += png_chunk(b"PLTE", palette.astype(np.uint8).tobytes())
    if trns is not None:
        out += png_chunk(b"tRNS", trns)
    step = max(1, len(z) // split)
    for i in range(0, len(z), step):
        out += png_chunk(b"IDAT", z[i:i + step])
    return out + png_chunk(b"IEND", b"")


def to8(v, depth):
    if depth == 8:
        return v.astype(np.uint8)
    if depth == 16:
        return (v >> 8).astype(np.uint8)
    return (v.astype(np.int64) * 255 // ((1 << depth) - 1)).astype(np.uint8)


CASES = [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 1), (3, 2), (3, 4), (3, 8), (4, 8), (4, 16),
         (6, 8), (6, 16)]


@pytest.mark.parametrize("color,depth", CASES)
@pytest.mark.parametrize("interlace", [False, True])
def test_png_colour_types_depths_filters_interlace(host, color, depth, interlace):
    rs = np.random.RandomState(color * 100 + depth + interlace)
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
    for (h, w) in ((11, 19), (1, 1), (8, 3), (3, 9)):
        pal = trns = None
        hi = 1 << depth
        if color == 3:
            npal = min(hi, 23)
            pal = rs.randint(0, 256, size=(npal, 3))
            hi = npal
            trns = bytes(rs.randint(0, 256, size=npal // 2).astype(np.uint8))
        s = rs.randint(0, hi, size=(h, w, ch))
        if depth <= 8 and w > 4:
            s[:, 2:5] = s[:, 1:2]  # some repetition for the compressor
        exp = np.empty((h, w, 4), dtype=np.uint8)
        if color == 0:
            exp[..., :3] = to8(s, depth)
            exp[..., 3] = 255
            if h > 1:  # colour key
                key = int(s[0, 0, 0])
                trns = struct.pack(">H", key)
                exp[..., 3] = np.where(s[..., 0] == key, 0, 255)
        elif color == 2:
            exp[..., :3] = to8(s, depth)
            key = s[0, 0]
            trns = struct.pack(">HHH", *[int(k) for k in key])
            exp[..., 3] = np.where((s == key).all(-1), 0, 255)
        elif color == 3:
            exp[..., :3] = pal[s[..., 0]]
            alpha = np.full(len(pal), 255)
            alpha[:len(trns)] = list(trns)
            exp[..., 3] = alpha[s[..., 0]]
        elif color == 4:
            exp[..., :3] = to8(s[..., :1], depth)
            exp[..., 3] = to8(s[..., 1], depth)
        else:
            exp[...] = to8(s, depth)
        data = png_bytes(s, color, depth, rs, interlace=interlace, palette=pal, trns=trns,
                         level=int(rs.choice([0, 1, 6, 9])), split=int(rs.choice([1, 3])))
        got = host.decode_image(data, invert_y=False)
        assert got.shape == exp.shape
        assert np.array_equal(got, exp), (color, depth, interlace, h, w)
        PIL = pytest.importorskip("PIL.Image")
        if depth == 8 and color in (2, 6) :
            import io
            assert np.array_equal(np.array(PIL.open(io.BytesIO(data)).convert("RGBA")), exp)


def test_png_large_fixed_and_stored_blocks(host):
    """zlib level 0 gives stored blocks, Z_FIXED gives fixed-Huffman blocks, level 9 dynamic ones."""
    rs = np.random.RandomState(11)
    s = rs.randint(0, 256, size=(70, 300, 4))
    s[20:50] = s[19:20]
    rows = pack_samples(s, 8)
    raw = filter_rows(rows, 4, rs)
    head = b"\x89PNG\r\n\x1a\n" + png_chunk(b"IHDR", struct.pack(">IIBBBBB", 300, 70, 8, 6, 0, 0, 0))
    for strategy, level in ((zlib.Z_FIXED, 6), (zlib.Z_DEFAULT_STRATEGY, 0), (zlib.Z_DEFAULT_STRATEGY, 9),
                            (zlib.Z_HUFFMAN_ONLY, 6), (zlib.Z_RLE, 6)):
        co = zlib.compressobj(level, zlib.DEFLATED, 15, 8, strategy)
        z = co.compress(raw) + co.flush()
        data = head + png_chunk(b"IDAT", z) + png_chunk(b"IEND", b"")
        assert np.array_equal(host.decode_image(data, invert_y=False), s.astype(np.uint8))


def test_png_file_round_trip(host, tmp_path):
    rs = np.random.RandomState(6)
    img = blocky(rs, 300, 257, 4)  # > 65535 bytes of scanlines: several stored blocks
    p = str(tmp_path / "f.png")
    host.write_png(p, img)
    assert np.array_equal(host.load_image(p, invert_y=True), img)
    PIL = pytest.importorskip("PIL.Image")
    assert np.array_equal(np.array(PIL.open(p)), img[::-1])


def test_damaged_files_fail_cleanly(host):
    """Truncations and byte flips end in an error or a decoded image, never in a crash."""
    rs = np.random.RandomState(12)
    s = rs.randint(0, 256, size=(24, 31, 3))
    files = [png_bytes(s, 2, 8, rs), png_bytes(s % 16, 3, 4, rs, palette=rs.randint(0, 256, size=(16, 3)), interlace=True),
             tga_bytes(blocky(rs, 16, 16, 4), 2, rle=True), tga_bytes(blocky(rs, 16, 16, 3), 2),
             tga_bytes(blocky(rs, 9, 9, 1) % 8, 1, rle=True, palette=rs.randint(0, 256, size=(8, 3)).astype(np.uint8))]
    errors = 0
    for data in files:
        for cut in list(range(0, 60)) + [int(x) for x in rs.randint(0, len(data), size=60)]:
            try:
                host.decode_image(data[:cut], invert_y=True)
            except host.HostError:
                errors += 1
        for _ in range(300):
            b = bytearray(data)
            for _ in range(int(rs.randint(1, 4))):
                b[int(rs.randint(0, len(b)))] = int(rs.randint(0, 256))
            try:
                host.decode_image(bytes(b), invert_y=False)
            except host.HostError:
                errors += 1
    assert errors > 300
    for junk in (b"", b"\x89PNG\r\n\x1a\n", b"not an image at all, but long enough"):
        with pytest.raises(host.HostError):
            host.decode_image(junk)
    with pytest.raises(host.HostError, match="Can't open file"):
        host.load_image("/nonexistent/x.tga")


def test_reference_data_files(host):
    """bth.tga (16 x 16, 24-bit) and sword.png (1024 x 1024, 8-bit palette + tRNS) are the two
    texture files the reference ships for its scenes (main.cpp:393, bth_logo_obj_tga/)."""
    want = np.load(os.path.join(GOLDEN, "bth_tga_rgba.npy"))
    assert np.array_equal(host.load_image(os.path.join(GOLDEN, "bth.tga"), invert_y=False), want)
    assert np.array_equal(host.load_image(os.path.join(GOLDEN, "bth.tga"), invert_y=True), want[::-1])
    g = np.load(os.path.join(GOLDEN, "sword_png_digest.npz"))
    a = host.load_image(os.path.join(ROOT, "assets", "sword", "sword.png"), invert_y=True)
    assert a.shape == tuple(g["shape"])
    assert zlib.crc32(a.tobytes()) == int(g["crc32_bottom_up"])
    assert np.array_equal(a[::64, ::64], g["samples_bottom_up"])
