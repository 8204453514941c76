"""Baseline JPEG maps (rayzath_amd/csrc/image_io.cpp: decode_jpeg) — CPU only.

The files are written HERE by a small baseline encoder in pure Python / numpy (forward DCT, the Annex K Huffman tables, 4:4:4,
4:2:2 and 4:2:0 sampling, grey, restart intervals), and the expectation is computed independently from the QUANTISED coefficients the
encoder emitted: double-precision inverse DCT, stb_image's triangle-filter chroma upsampling and fixed-point YCbCr matrix — so the
library must reproduce it to within one code value (the rounding of the IDCT), and the original picture to within the quantisation."""
import numpy as np
import pytest

from rayzath_amd import scene_io
from rayzath_amd._lib import HiprzError

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56,
          57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
# ITU-T T.81 Annex K.3 typical Huffman tables
DC_LUM = ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], list(range(12)))
DC_CHR = ([0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], list(range(12)))
AC_LUM = ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d],
          [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08,
           0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
           0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
           0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
           0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
           0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
           0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])
AC_CHR = ([0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77],
          [0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91,
           0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
           0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
           0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
           0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
           0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
           0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])

_x = np.arange(8)
BASIS = np.array([[(np.sqrt(0.125) if u == 0 else 0.5) * np.cos((2 * x + 1) * u * np.pi / 16) for u in range(8)] for x in _x])  # [x][u]


def _codes(table):
    bits, values = table
    out, code, k = {}, 0, 0
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            out[values[k]] = (code, length)
            code += 1
            k += 1
        code <<= 1
    return out


class _Bits:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, value, length):
        for i in range(length - 1, -1, -1):
            self.acc = (self.acc << 1) | ((value >> i) & 1)
            self.n += 1
            if self.n == 8:
                self.out.append(self.acc)
                if self.acc == 0xFF:
                    self.out.append(0)
                self.acc, self.n = 0, 0

    def flush(self):
        while self.n:
            self.put(1, 1)


def _magnitude(v):
    size = int(abs(v)).bit_length()
    return size, (v if v >= 0 else v + (1 << size) - 1)


def _segment(marker, payload):
    return bytes([0xFF, marker]) + (len(payload) + 2).to_bytes(2, "big") + payload


def encode_jpeg(image, sampling=(1, 1), quant=None, restart=0):
    """image: (h, w) grey or (h, w, 3) RGB uint8.  sampling = chroma subsampling factors (h, v).  Returns (file bytes,
    [per component (quantised coefficient blocks (by, bx, 8, 8), quant table 8x8, (h factor, v factor))], planes before encoding)."""
    image = np.asarray(image, dtype=np.float64)
    grey = image.ndim == 2
    h, w = image.shape[:2]
    q_l = np.full((8, 8), 2, dtype=np.int64) if quant is None else np.asarray(quant[0], dtype=np.int64)
    q_c = np.full((8, 8), 3, dtype=np.int64) if quant is None else np.asarray(quant[1], dtype=np.int64)
    if grey:
        planes, factors, tables = [image], [(1, 1)], [q_l]
    else:
        r, g, b = image[..., 0], image[..., 1], image[..., 2]
        y = 0.299 * r + 0.587 * g + 0.114 * b
        cb = -0.168736 * r - 0.331264 * g + 0.5 * b + 128
        cr = 0.5 * r - 0.418688 * g - 0.081312 * b + 128
        sh, sv = sampling
        def down(p):
            ph, pw = -h % sv, -w % sh
            p = np.pad(p, ((0, ph), (0, pw)), mode="edge")
            return p.reshape(p.shape[0] // sv, sv, p.shape[1] // sh, sh).mean(axis=(1, 3))
        planes, factors, tables = [y, down(cb), down(cr)], [(sh, sv), (1, 1), (1, 1)], [q_l, q_c, q_c]
    hmax, vmax = max(f[0] for f in factors), max(f[1] for f in factors)
    mcus_x, mcus_y = -(-w // (8 * hmax)), -(-h // (8 * vmax))
    if grey:
        mcus_x, mcus_y = -(-w // 8), -(-h // 8)
    blocks = []
    for p, (fh, fv), q in zip(planes, factors, tables):
        ph, pw = mcus_y * 8 * fv, mcus_x * 8 * fh
        p = np.pad(p, ((0, ph - p.shape[0]), (0, pw - p.shape[1])), mode="edge") - 128.0
        b = p.reshape(ph // 8, 8, pw // 8, 8).transpose(0, 2, 1, 3)                       # (by, bx, y, x)
        coef = np.einsum("yv,abyx,xu->abvu", BASIS, b, BASIS)                            # forward DCT
        blocks.append(np.rint(coef / q).astype(np.int64))
    dc_codes, ac_codes = [_codes(DC_LUM), _codes(DC_CHR)], [_codes(AC_LUM), _codes(AC_CHR)]
    bits, pred, out = _Bits(), [0] * len(planes), bytearray()
    count = 0
    for my in range(mcus_y):
        for mx in range(mcus_x):
            if restart and count and count % restart == 0:
                bits.flush()
                out += bits.out + bytes([0xFF, 0xD0 + ((count // restart - 1) % 8)])
                bits, pred = _Bits(), [0] * len(planes)
            count += 1
            for c, (fh, fv) in enumerate(factors):
                t = 0 if c == 0 else 1
                for by in range(fv):
                    for bx in range(fh):
                        zz = blocks[c][my * fv + by, mx * fh + bx].reshape(64)[ZIGZAG]
                        size, extra = _magnitude(int(zz[0]) - pred[c])
                        pred[c] = int(zz[0])
                        bits.put(*dc_codes[t][size])
                        bits.put(extra, size)
                        run = 0
                        last = max([k for k in range(1, 64) if zz[k] != 0], default=0)
                        for k in range(1, last + 1):
                            if zz[k] == 0:
                                run += 1
                                continue
                            while run > 15:
                                bits.put(*ac_codes[t][0xF0])
                                run -= 16
                            size, extra = _magnitude(int(zz[k]))
                            bits.put(*ac_codes[t][(run << 4) | size])
                            bits.put(extra, size)
                            run = 0
                        if last < 63:
                            bits.put(*ac_codes[t][0x00])
    bits.flush()
    out += bits.out
    def dqt(i, q):
        return _segment(0xDB, bytes([i]) + bytes(int(v) for v in q.reshape(64)[ZIGZAG]))
    def dht(tc, th, table):
        return _segment(0xC4, bytes([(tc << 4) | th]) + bytes(table[0]) + bytes(table[1]))
    n = len(planes)
    sof = bytes([8]) + h.to_bytes(2, "big") + w.to_bytes(2, "big") + bytes([n])
    for c, (fh, fv) in enumerate(factors):
        sof += bytes([c + 1, (fh << 4) | fv, 0 if c == 0 else 1])
    sos = bytes([n]) + b"".join(bytes([c + 1, 0x00 if c == 0 else 0x11]) for c in range(n)) + bytes([0, 63, 0])
    head = b"\xFF\xD8" + _segment(0xE0, b"JFIF\0\x01\x01\0\0\x01\0\x01\0\0") + _segment(0xFE, b"written by tests/test_jpeg_io.py")
    head += dqt(0, q_l) + (dqt(1, q_c) if not grey else b"") + _segment(0xC0, sof)
    head += dht(0, 0, DC_LUM) + dht(1, 0, AC_LUM) + (dht(0, 1, DC_CHR) + dht(1, 1, AC_CHR) if not grey else b"")
    if restart:
        head += _segment(0xDD, restart.to_bytes(2, "big"))
    return head + _segment(0xDA, sos) + bytes(out) + b"\xFF\xD9", list(zip(blocks, tables, factors))


def reference_decode(components, w, h):
    """What a decoder must produce from the quantised coefficients: double IDCT rounded to nearest, stb_image's triangle-filter
    upsampling for factor 2, its 20-bit fixed-point YCbCr -> RGB."""
    planes = []
    for blocks, q, _ in components:
        px = np.einsum("yv,abvu,xu->abyx", BASIS, blocks * q, BASIS) + 128.0
        nby, nbx = px.shape[:2]
        planes.append(np.clip(np.floor(px.transpose(0, 2, 1, 3).reshape(nby * 8, nbx * 8) + 0.5), 0, 255).astype(np.int64))
    if len(planes) == 1:
        return planes[0][:h, :w].astype(np.uint8)[..., None]
    hmax, vmax = components[0][2]
    full = [planes[0][:h, :w]]
    for p in planes[1:]:
        ch, cw = -(-h // vmax), -(-w // hmax)
        p = p[:ch, :cw]
        if vmax == 2:
            near = p[np.arange(h) // 2]
            far_idx = np.where(np.arange(h) % 2 == 1, np.minimum(np.arange(h) // 2 + 1, ch - 1), np.maximum(np.arange(h) // 2 - 1, 0))
            far = p[far_idx]
        else:
            near = far = p[np.arange(h)]
        if hmax == 1:
            rows = (3 * near + far + 2) >> 2 if vmax == 2 else near
        elif vmax == 1:
            rows = np.zeros((h, 2 * cw), dtype=np.int64)
            if cw == 1:
                rows[:, 0] = rows[:, 1] = near[:, 0]
            else:
                rows[:, 0] = near[:, 0]
                rows[:, 1] = (3 * near[:, 0] + near[:, 1] + 2) >> 2
                for i in range(1, cw - 1):
                    rows[:, 2 * i] = (3 * near[:, i] + 2 + near[:, i - 1]) >> 2
                    rows[:, 2 * i + 1] = (3 * near[:, i] + 2 + near[:, i + 1]) >> 2
                rows[:, 2 * cw - 2] = (3 * near[:, cw - 2] + near[:, cw - 1] + 2) >> 2
                rows[:, 2 * cw - 1] = near[:, cw - 1]
        else:
            t = 3 * near + far
            rows = np.zeros((h, 2 * cw), dtype=np.int64)
            rows[:, 0] = (t[:, 0] + 2) >> 2
            for i in range(1, cw):
                rows[:, 2 * i - 1] = (3 * t[:, i - 1] + t[:, i] + 8) >> 4
                rows[:, 2 * i] = (3 * t[:, i] + t[:, i - 1] + 8) >> 4
            rows[:, 2 * cw - 1] = (t[:, cw - 1] + 2) >> 2
        full.append(rows[:, :w])
    y, cb, cr = full[0], full[1] - 128, full[2] - 128
    yf = (y << 20) + (1 << 19)
    r = (yf + cr * 1470208) >> 20
    g = (yf + cr * -748800 + (((cb * -360960) & 0xFFFFFFFF) & 0xFFFF0000).astype(np.int64).astype(np.uint32).astype(np.int32).astype(np.int64)) >> 20
    b = (yf + cb * 1858048) >> 20
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def _picture(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([128 + 100 * np.sin(xx / 9.0 + seed), 128 + 90 * np.cos(yy / 7.0), 60 + 1.5 * xx + yy], axis=-1)
    return np.clip(base + rng.normal(0, 6, size=base.shape), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("sampling", [(1, 1), (2, 1), (2, 2), (1, 2)])
@pytest.mark.parametrize("size", [(8, 8), (17, 23), (40, 33), (64, 48)])
def test_colour_jpeg_sampling_factors(tmp_path, sampling, size):
    h, w = size
    img = _picture(h, w, 3 * h + w)
    data, comps = encode_jpeg(img, sampling)
    (tmp_path / "t.jpg").write_bytes(data)
    got = scene_io.read_image(str(tmp_path / "t.jpg"))
    want = reference_decode(comps, w, h)
    assert got.shape == (h, w, 3)
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff == 0).mean() > 0.98                      # only a .5 tie of the IDCT may round the other way
    assert np.abs(got.astype(int) - img.astype(int)).mean() < (3.0 if sampling == (1, 1) else 8.0)   # and it is the picture


def test_grey_jpeg_restart_intervals_and_coarse_quantisation(tmp_path):
    rng = np.random.default_rng(5)
    img = _picture(50, 70, 1)[..., 1]
    for restart in (0, 1, 4):
        for q in (1, 16):
            quant = (np.full((8, 8), q), np.full((8, 8), q))
            data, comps = encode_jpeg(img, quant=quant, restart=restart)
            (tmp_path / "g.jpg").write_bytes(data)
            got = scene_io.read_image(str(tmp_path / "g.jpg"))
            want = reference_decode(comps, 70, 50)
            assert got.shape == (50, 70, 1) and np.abs(got.astype(int) - want.astype(int)).max() <= 1
            if q == 1:
                assert np.abs(got[..., 0].astype(int) - img.astype(int)).max() <= 2
    rgb = _picture(33, 47, 9)
    data, comps = encode_jpeg(rgb, (2, 2), restart=2)
    (tmp_path / "r.jpg").write_bytes(data)
    assert np.abs(scene_io.read_image(str(tmp_path / "r.jpg")).astype(int) - reference_decode(comps, 47, 33).astype(int)).max() <= 1


def test_jpeg_map_through_the_mtl_loader_and_refusals(tmp_path):
    img = _picture(16, 16, 2)
    data, comps = encode_jpeg(img)
    (tmp_path / "kd.jpg").write_bytes(data)
    (tmp_path / "m.mtl").write_text("newmtl a\nmap_Kd kd.jpg\nmap_Pr kd.jpg\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\no x\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl a\nf 1 2 3\n")
    s = scene_io.load_scene_file(str(tmp_path / "m.obj"))
    assert s.errors == 0, s.log
    f = s.flat
    mat = f.materials[2]
    tex, rough = f.textures[mat["texture"]], f.textures[mat["roughness_map"]]
    want = reference_decode(comps, 16, 16).astype(int)
    got = f.texels[tex["offset"]:tex["offset"] + 16 * 16 * 4].reshape(16, 16, 4).astype(int)
    assert np.abs(got[..., :3] - want).max() <= 1 and (got[..., 3] == 255).all()
    luma = (got[..., 0] * 77 + got[..., 1] * 150 + got[..., 2] * 29) >> 8
    assert np.array_equal(f.texels[rough["offset"]:rough["offset"] + 256].reshape(16, 16), luma)
    progressive = data.replace(b"\xFF\xC0", b"\xFF\xC2", 1)
    (tmp_path / "p.jpg").write_bytes(progressive)
    with pytest.raises(HiprzError, match="progressive"):
        scene_io.read_image(str(tmp_path / "p.jpg"))
    (tmp_path / "cut.jpg").write_bytes(data[:60])
    with pytest.raises(HiprzError):
        scene_io.read_image(str(tmp_path / "cut.jpg"))


GOLDEN_JPEG = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden", "jpeg")


@pytest.mark.parametrize("name", ["rgb444", "rgb420", "rgb422_restart", "grey", "rgb420_tiny"])
def test_progressive_jpeg_equals_the_baseline_file_of_the_same_coefficients(name):
    """Progressive JPEG maps (the reference reads them through stb_image, loader.cpp:36-144).  tests/golden/make_progressive_jpeg.py saved
    every picture twice with the same quantisation: as a progressive file — ten scans: DC with point transform, luma AC in two spectral
    bands, full-band chroma, then the refinement scans; end-of-band runs and a restart interval among them — and as a baseline file.  The two
    hold the same quantised coefficients, so the decoded pixels must be EQUAL; the baseline path is pinned above.  libjpeg's own decode of
    the progressive file (stored beside them) is a looser second anchor: its integer IDCT and colour conversion round differently from
    stb_image's, which the library follows."""
    import os
    data = open(os.path.join(GOLDEN_JPEG, name + "_progressive.jpg"), "rb").read()
    assert b"\xFF\xC2" in data and data.count(b"\xFF\xDA") >= (6 if name == "grey" else 10)
    progressive = scene_io.read_image(os.path.join(GOLDEN_JPEG, name + "_progressive.jpg"))
    baseline = scene_io.read_image(os.path.join(GOLDEN_JPEG, name + "_baseline.jpg"))
    assert np.array_equal(progressive, baseline)
    expected = np.load(os.path.join(GOLDEN_JPEG, "jpeg_expected.npz"))
    diff = np.abs(progressive.astype(int) - expected[name].astype(int))
    assert progressive.shape == expected[name].shape and diff.max() <= 4 and diff.mean() < 0.4
    assert np.abs(progressive.astype(int) - expected[name + "_source"].astype(int)).mean() < 6.0      # and it is the picture


def test_progressive_jpeg_cut_short_and_malformed(tmp_path):
    """A progressive file that ends after some of its scans still decodes — coarser, as stb_image shows it — and gets closer to the picture
    with every scan it keeps; nonsense in a scan header is refused."""
    import os
    data = open(os.path.join(GOLDEN_JPEG, "rgb420_progressive.jpg"), "rb").read()
    full = scene_io.read_image(os.path.join(GOLDEN_JPEG, "rgb420_progressive.jpg")).astype(int)
    scans = [i for i in range(len(data) - 1) if data[i] == 0xFF and data[i + 1] == 0xDA]
    errors = []
    for keep in (1, 4, 7, len(scans)):
        cut = data[:scans[keep]] if keep < len(scans) else data
        (tmp_path / "cut.jpg").write_bytes(cut)
        got = scene_io.read_image(str(tmp_path / "cut.jpg")).astype(int)
        assert got.shape == full.shape
        errors.append(np.abs(got - full).mean())
    assert errors[0] > errors[1] > errors[2] > errors[3] == 0.0
    sos = scans[1]   # second scan: luma AC band; make it claim two components (AC scans hold one)
    broken = bytearray(data)
    broken[sos + 4] = 2
    (tmp_path / "broken.jpg").write_bytes(bytes(broken))
    with pytest.raises(HiprzError):
        scene_io.read_image(str(tmp_path / "broken.jpg"))
    (tmp_path / "noscan.jpg").write_bytes(data[:scans[0]] + b"\xFF\xD9")
    with pytest.raises(HiprzError, match="scan"):
        scene_io.read_image(str(tmp_path / "noscan.jpg"))


def test_mutated_jpeg_files_are_decoded_or_refused(tmp_path):
    """Map files come from outside: 1 500 random mutations of the fixtures (bytes changed, bits flipped, runs removed or inserted, files cut
    short) must each either decode to an image of the declared size or be refused with a message — never crash, hang or read out of bounds
    (the same mutations ran clean through an -fsanitize=address,undefined build of image_io.cpp, 4 000 files)."""
    import os
    import random
    rng = random.Random(20240501)
    names = [f for f in sorted(os.listdir(GOLDEN_JPEG)) if f.endswith(".jpg")]
    decoded = refused = 0
    for _ in range(1500):
        data = bytearray(open(os.path.join(GOLDEN_JPEG, rng.choice(names)), "rb").read())
        for _ in range(rng.randint(1, 6)):
            k, mode = rng.randrange(len(data)), rng.randrange(4)
            if mode == 0:
                data[k] = rng.randrange(256)
            elif mode == 1:
                data[k] ^= 1 << rng.randrange(8)
            elif mode == 2:
                del data[k:k + rng.randint(1, 20)]
            else:
                data[k:k] = bytes(rng.randrange(256) for _ in range(rng.randint(1, 8)))
        if rng.random() < 0.2:
            data = data[:rng.randrange(2, len(data))]
        (tmp_path / "m.jpg").write_bytes(bytes(data))
        try:
            img = scene_io.read_image(str(tmp_path / "m.jpg"))
            assert img.ndim == 3 and img.shape[2] in (1, 3) and img.size > 0
            decoded += 1
        except HiprzError:
            refused += 1
    assert decoded > 50 and refused > 50
