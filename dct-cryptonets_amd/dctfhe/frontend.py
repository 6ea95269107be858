"""Plaintext client-side front-end: image -> DCT-packed tensor.

Restates the evaluation branch of the reference transform pipeline (reference data/datamgr.py:192-219):
  Resize(int(filter*S*1.15)) -> CenterCrop(filter*S) -> GetDCT(filter) -> UpScaleDCT(S) -> ToTensorDCT
  -> SubsetDCT -> Aggregate -> NormalizeDCT -> x[0]
Pinned by goldens captured from the reference's own functions (tests/golden/frontend_golden.npz,
tools/make_goldens.py): matrix2dct, SubsetDCT, Aggregate, NormalizeDCT and the statistics table.
NOT pinned (OpenCV is absent): the 8-bit YCrCb conversion and the bilinear resizes; they
follow OpenCV's documented arithmetic and are covered by self-consistency tests only.
The whole front-end runs before quantise+encrypt (the reference does it inside the Dataset), so it
never touches ciphertexts.
"""
import math
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "dct_stats.npz")

# reference data/cvtransforms.py:1601-1613 (filter 4) and :1643-1683 (default pattern): kept low-frequency indices
SUBSET_FILTER4 = {
    24: ([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13], [0, 1, 2, 4, 5, 8], [0, 1, 2, 4, 5, 8]),
    48: (list(range(16)), list(range(16)), list(range(16))),
}
SUBSET_DEFAULT = {
    6: ([0, 1, 4, 5], [0], [0]),
    12: ([0, 1, 2, 8, 9, 10, 16, 17], [0, 8], [0, 8]),
    24: ([0, 1, 2, 3, 4, 5, 8, 9, 10, 16, 17, 18, 24, 32], [0, 1, 3, 8, 24], [0, 1, 3, 8, 24]),
    48: ([0, 1, 2, 3, 4, 5, 8, 9, 10, 11, 12, 13, 16, 17, 18, 19, 20, 21, 24, 25, 26, 27, 28, 29, 32, 33, 34, 35, 40, 41, 42, 43],
         [0, 1, 2, 8, 9, 10, 16, 17], [0, 1, 2, 8, 9, 10, 16, 17]),
    64: ([0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14, 16, 17, 18, 19, 20, 21, 24, 25, 26, 27, 28, 29, 32, 33, 34, 35, 36, 37,
          40, 41, 42, 43, 44, 45, 48, 49, 50, 51, 52, 53], [0, 1, 2, 8, 9, 10, 16, 17, 24, 25], [0, 1, 2, 8, 9, 10, 16, 17, 24, 25]),
}


def load_stats():
    z = np.load(_DATA)
    return z["mean"], z["std"]


_TABLES = None


def subset_tables():
    """{pattern: {channels: (y, cb, cr) index lists}} for the reference's five SubsetDCT tables -- 'filter4', 'default', 'square',
    'learned', 'triangle' (cvtransforms.py:1600-1860) -- as data captured from the reference's constants by tools/make_goldens.py
    (dctfhe/data/subset_tables.json), like the statistics table.  The two tables every BASELINE config uses are also spelled out
    above (SUBSET_FILTER4 / SUBSET_DEFAULT) and tests/test_frontend_golden.py holds the two forms equal."""
    global _TABLES
    if _TABLES is None:
        import json
        with open(os.path.join(os.path.dirname(_DATA), "subset_tables.json")) as f:
            raw = json.load(f)
        _TABLES = {name: {int(ch): tuple(list(part) for part in v) for ch, v in d.items()} for name, d in raw.items()}
    return _TABLES


# ------------------------------------------------------------------------------------------ pinned by goldens
def matrix2dct(plane, size):
    """(pixel - 128) blockwise orthonormal DCT-II, T B T^t per size x size block, flattened row-major;
    trailing rows/cols that do not fill a block are dropped (reference cvfunctional.py:37-57)."""
    m = plane.astype(np.int16).astype(np.float64) - 128.0
    T = np.zeros((size, size))
    for i in range(size):
        for j in range(size):
            T[i, j] = 1.0 / math.sqrt(size) if i == 0 else math.sqrt(2.0 / size) * math.cos((2 * j + 1) * i * math.pi / (2 * size))
    bh, bw = m.shape[0] // size, m.shape[1] // size
    blocks = m[:bh * size, :bw * size].reshape(bh, size, bw, size).transpose(0, 2, 1, 3)    # [bh, bw, s, s]
    out = np.matmul(np.matmul(T, blocks), T.T)
    return out.reshape(bh, bw, size * size)


def subset_indices(channels, pattern="default", filter_size=8):
    """reference SubsetDCT.__init__ (cvtransforms.py:117-136): the pattern is ignored when filter_size == 4"""
    if channels == 192:
        return list(range(64)), list(range(64)), list(range(64))
    if filter_size == 4:
        table = SUBSET_FILTER4
    elif pattern == "default":
        table = SUBSET_DEFAULT if channels in SUBSET_DEFAULT else subset_tables()["default"]
    elif pattern in ("square", "learned", "triangle"):
        table = subset_tables()[pattern]
    else:
        raise ValueError(f"dct_pattern {pattern!r}: the reference knows 'default', 'square', 'learned', 'triangle'")
    if channels not in table:
        raise ValueError(f"no {'filter-4' if filter_size == 4 else pattern!r} coefficient table for {channels} channels (the reference has {sorted(table)})")
    return table[channels]


def normalize_indices(channels):
    """reference NormalizeDCT.__init__ (cvtransforms.py:168-183): ALWAYS the default (8x8) pattern -- the
    transform is built without pattern/filter_size (datamgr.py:209-216), so for filter 4 the statistics do
    not correspond to the selected coefficients.  Kept as is."""
    table = SUBSET_DEFAULT if channels in SUBSET_DEFAULT else subset_tables()["default"]
    if channels not in table:
        raise ValueError(f"NormalizeDCT indexes the statistics with the default table, which has no entry for {channels} channels "
                         "(the reference raises KeyError here)")
    y, cb, cr = table[channels]
    return list(y) + [64 + c for c in cb] + [128 + c for c in cr]


def subset_aggregate_normalize(dct_y, dct_cb, dct_cr, channels, pattern="default", filter_size=8):
    """CHW float32 planes -> [channels, S, S] (SubsetDCT + Aggregate + NormalizeDCT, cvtransforms.py:117-208)"""
    sy, scb, scr = subset_indices(channels, pattern, filter_size)
    agg = np.concatenate([dct_y[sy], dct_cb[scb], dct_cr[scr]], axis=0).astype(np.float32)
    if channels < 192:
        mean, std = load_stats()
        idx = normalize_indices(channels)
        m, s = mean[idx].astype(np.float32), std[idx].astype(np.float32)
    else:
        m, s = [a.astype(np.float32) for a in load_stats()]
    # torch: t.sub_(m).div_(s) in float32, channel by channel (cvfunctional.py:181-201)
    return (agg - m[:, None, None]) / s[:, None, None]


# ------------------------------------------------------------------------------------------ unpinned (OpenCV arithmetic)
def rgb_to_ycrcb_u8(img):
    """8-bit RGB -> Y, Cr, Cb planes with OpenCV's fixed-point coefficients (14-bit, round to nearest):
    Y = (4899 R + 9617 G + 1868 B + 8192) >> 14; Cr = ((R - Y) * 11682 + (128 << 14) + 8192) >> 14;
    Cb = ((B - Y) * 9241 + (128 << 14) + 8192) >> 14, saturated.  [K] -- not checkable here."""
    r, g, b = [img[..., i].astype(np.int64) for i in range(3)]
    y = (4899 * r + 9617 * g + 1868 * b + 8192) >> 14
    cr = ((r - y) * 11682 + (128 << 14) + 8192) >> 14
    cb = ((b - y) * 9241 + (128 << 14) + 8192) >> 14
    sat = lambda v: np.clip(v, 0, 255).astype(np.uint8)
    return sat(y), sat(cr), sat(cb)


def _bilinear(src, oh, ow):
    """bilinear resize with half-pixel centres and edge clamp (OpenCV INTER_LINEAR geometry), float64"""
    ih, iw = src.shape[:2]
    ys = (np.arange(oh) + 0.5) * (ih / oh) - 0.5
    xs = (np.arange(ow) + 0.5) * (iw / ow) - 0.5
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
    fy = ys - y0; fx = xs - x0
    y0c, y1c = np.clip(y0, 0, ih - 1), np.clip(y0 + 1, 0, ih - 1)
    x0c, x1c = np.clip(x0, 0, iw - 1), np.clip(x0 + 1, 0, iw - 1)
    s = src.astype(np.float64)
    if s.ndim == 2:
        s = s[..., None]
    top = s[y0c][:, x0c] * (1 - fx)[None, :, None] + s[y0c][:, x1c] * fx[None, :, None]
    bot = s[y1c][:, x0c] * (1 - fx)[None, :, None] + s[y1c][:, x1c] * fx[None, :, None]
    out = top * (1 - fy)[:, None, None] + bot * fy[:, None, None]
    return out if src.ndim == 3 else out[..., 0]


def resize_u8(img, oh, ow):
    if (oh, ow) == img.shape[:2]:
        return img.copy()
    return np.clip(np.floor(_bilinear(img, oh, ow) + 0.5), 0, 255).astype(np.uint8)


def halve_u8(plane):
    """cv2.resize(plane, (w//2, h//2)) on uint8: exact 2x bilinear decimation = rounded 2x2 mean [K]"""
    h, w = plane.shape
    p = plane[:h // 2 * 2, :w // 2 * 2].astype(np.int32)
    return ((p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2] + 2) >> 2).astype(np.uint8)


def center_crop(img, size):
    """reference cvfunctional.py:358-368 + :324-355; Python round() is banker's rounding"""
    h, w = img.shape[:2]
    i, j = int(round((h - size) * 0.5)), int(round((w - size) * 0.5))
    return img[i:i + size, j:j + size].copy()


def transform_dct_size(img, size):
    """reference cvfunctional.py:59-74.  OpenCV's COLOR_BGR2YCrCb yields planes Y, Cr, Cb but the code
    unpacks them as y, cb, cr: the "cb" slot carries Cr.  Reproduced: returned order is (Y, Cr, Cb)."""
    y, cr, cb = rgb_to_ycrcb_u8(img)
    slot_cb, slot_cr = halve_u8(cr), halve_u8(cb)
    return matrix2dct(y, size), matrix2dct(slot_cb, size), matrix2dct(slot_cr, size)


def _round_half_away(x):
    return np.sign(x) * np.floor(np.abs(x) + 0.5)


# libjpeg's accurate integer DCT ("islow", jfdctint.c: Loeffler-Ligtenberg-Moschytz with 13-bit constants, two extra bits carried out of
# the row pass) [K: restated from the published algorithm; libjpeg-turbo / TurboJPEG / jpeg2dct are absent -- PARITY UNPINNED].
# TurboJPEG selects it at quality >= 96 (reference data/cvfunctional.py:24 encodes at quality=100).
_ISLOW = dict(F0_298=2446, F0_390=3196, F0_541=4433, F0_765=6270, F0_899=7373, F1_175=9633, F1_501=12299, F1_847=15137, F1_961=16069,
              F2_053=16819, F2_562=20995, F3_072=25172)


def _islow_1d(d, first):
    """one pass over the last axis (8 values) of an int64 array; first: row pass (x 2^2), else column pass (removes the 2^2)"""
    c = _ISLOW
    CB, P1 = 13, 2
    descale = lambda x, n: (x + (1 << (n - 1))) >> n
    x = [d[..., i] for i in range(8)]
    t0, t7, t1, t6, t2, t5, t3, t4 = x[0] + x[7], x[0] - x[7], x[1] + x[6], x[1] - x[6], x[2] + x[5], x[2] - x[5], x[3] + x[4], x[3] - x[4]
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    sh = CB - P1 if first else CB + P1
    o = [None] * 8
    o[0] = (t10 + t11) << P1 if first else descale(t10 + t11, P1)
    o[4] = (t10 - t11) << P1 if first else descale(t10 - t11, P1)
    z1 = (t12 + t13) * c["F0_541"]
    o[2] = descale(z1 + t13 * c["F0_765"], sh)
    o[6] = descale(z1 - t12 * c["F1_847"], sh)
    z1, z2, z3, z4 = t4 + t7, t5 + t6, t4 + t6, t5 + t7
    z5 = (z3 + z4) * c["F1_175"]
    t4, t5, t6, t7 = t4 * c["F0_298"], t5 * c["F2_053"], t6 * c["F3_072"], t7 * c["F1_501"]
    z1, z2, z3, z4 = -z1 * c["F0_899"], -z2 * c["F2_562"], -z3 * c["F1_961"] + z5, -z4 * c["F0_390"] + z5
    o[7] = descale(t4 + z1 + z3, sh)
    o[5] = descale(t5 + z2 + z4, sh)
    o[3] = descale(t6 + z2 + z3, sh)
    o[1] = descale(t7 + z1 + z4, sh)
    return np.stack(o, axis=-1)


def jpeg_quantised_dct(plane_u8):
    """uint8 plane [H, W] (multiples of 8) -> int64 [H/8, W/8, 64]: level shift, libjpeg's integer forward DCT (rows, then columns;
    the result is 8x the orthonormal DCT), then its quantiser with the quality-100 tables (all ones): (|c| + 4) >> 3 with the sign put
    back, i.e. c / 8 rounded half away from zero (jcdctmgr.c).  Natural (row-major u*8+v) order, as jpeg2dct returns the blocks."""
    h, w = plane_u8.shape[0] // 8, plane_u8.shape[1] // 8
    blk = plane_u8[:h * 8, :w * 8].astype(np.int64).reshape(h, 8, w, 8).transpose(0, 2, 1, 3) - 128      # [h, w, row i, col j]
    rows = _islow_1d(blk, True)                                    # along j -> index v
    cols = _islow_1d(rows.transpose(0, 1, 3, 2), False)            # [h, w, v, u] along i -> index u
    c = cols.transpose(0, 1, 3, 2)                                 # [h, w, u, v]
    q = np.where(c < 0, -((-c + 4) >> 3), (c + 4) >> 3)
    return q.reshape(h, w, 64)


def transform_dct_jpeg(img):
    """8x8 path of the reference (cvfunctional.py:21-26): TurboJPEG.encode(img, quality=100, jpeg_subsample=2) then
    jpeg2dct.loads -> QUANTISED coefficients.  Restated from the JPEG/libjpeg definitions [K] -- PARITY UNPINNED, neither
    library is available:
      * the encoder's default pixel format is BGR while the array is RGB, so red and blue trade places inside the JPEG;
      * JFIF full-range YCbCr (libjpeg jccolor, 16-bit fixed point, rounded);
      * 4:2:0: 2x2 box average with libjpeg's alternating bias 1,2,1,2 (h2v2_downsample);
      * level shift -128, libjpeg's integer forward DCT (jfdctint "islow", what TurboJPEG runs at quality >= 96) and its quantiser with
        the quality-100 tables = all ones (jpeg_quantised_dct above): integers from the first pixel to the last coefficient, so the host
        and the device path (k_dct_frontend, round_coeffs) agree bit for bit; natural (row-major) coefficient order within a block.
    Returns (dct_y [H/8, W/8, 64], dct_cb [H/16, W/16, 64], dct_cr [H/16, W/16, 64]) as float64 holding integers."""
    h, w = (img.shape[0] // 16) * 16, (img.shape[1] // 16) * 16
    if (h, w) != img.shape[:2]:
        raise ValueError("image sides must be multiples of 16 (the reference crops to 8*S)")
    planes = jpeg_planes_u8(img)
    out = [jpeg_quantised_dct(pl).astype(np.float64) for pl in planes]
    return out[0], out[1], out[2]


def jpeg_planes_u8(img):
    """the three component planes a 4:2:0 JPEG of `img` holds before its DCT (see transform_dct_jpeg): Y [H, W], Cb, Cr [H/2, W/2]"""
    r, g, b = [img[..., i].astype(np.int64) for i in (2, 1, 0)]                 # array handed over as if it were BGR
    fix = lambda c: int(round(c * 65536))
    half = 32768
    y = (fix(0.29900) * r + fix(0.58700) * g + fix(0.11400) * b + half) >> 16
    cb = (-fix(0.16874) * r - fix(0.33126) * g + fix(0.50000) * b + (128 << 16) + half - 1) >> 16
    cr = (fix(0.50000) * r - fix(0.41869) * g - fix(0.08131) * b + (128 << 16) + half - 1) >> 16

    def down(p):
        bias = np.arange(p.shape[1] // 2) % 2 + 1                                # libjpeg: bias = 1; ...; bias ^= 3
        return (p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2] + bias[None, :]) >> 2

    return [np.clip(y, 0, 255).astype(np.uint8), down(np.clip(cb, 0, 255)).astype(np.uint8), down(np.clip(cr, 0, 255)).astype(np.uint8)]


def device_dct_batch(ctx, images_u8, filter_size=4, image_size_dct=16, channels=24, dct_pattern="default"):
    """The same evaluation transform with its DCT / subset / up-sampling / normalisation stages on the GPU
    (dctfhe_dct_frontend); colour conversion, resize, crop and chroma halving stay on the host (OpenCV-defined,
    integer).  Filter 4 (matrix2dct) path.  -> float32 [B, channels, S, S]; the numpy path stays the client-side default."""
    if filter_size not in (4, 8):
        raise ValueError("filter_size is 4 (matrix2dct path) or 8 (JPEG-domain path)")
    S = image_size_dct
    ys, c1s, c2s = [], [], []
    for img in images_u8:
        side = int(filter_size * S * 1.15)
        h, w = img.shape[:2]
        oh, ow = (int(side * h / w), side) if w <= h else (side, int(side * w / h))
        x = center_crop(resize_u8(img, oh, ow), filter_size * S)
        if filter_size == 8:        # the component planes of the 4:2:0 JPEG; the device runs libjpeg's integer DCT + quantiser on them
            y, c1, c2 = jpeg_planes_u8(x)
        else:
            y, cr, cb = rgb_to_ycrcb_u8(x)
            c1, c2 = halve_u8(cr), halve_u8(cb)                                 # the reference's name-swapped slots (transform_dct_size)
        ys.append(y); c1s.append(c1); c2s.append(c2)
    sy, scb, scr = subset_indices(channels, dct_pattern, filter_size)
    mean, std = load_stats()
    idx = normalize_indices(channels)
    return ctx.dct_frontend(np.stack(ys), np.stack(c1s), np.stack(c2s), filter_size, (sy, scb, scr), mean[idx], std[idx],
                            round_coeffs=filter_size == 8)


def dct_eval_transform(filter_size=4, image_size_dct=16, channels=24, dct_pattern="default"):
    """The composed evaluation transform of reference datamgr.py:192-219 (matrix2dct path for filter 4, JPEG-domain
    path for filter 8)."""
    S = image_size_dct

    def tf(img_u8):
        side = int(filter_size * S * 1.15)
        h, w = img_u8.shape[:2]
        if w <= h:
            ow, oh = side, int(side * h / w)
        else:
            oh, ow = side, int(side * w / h)
        x = resize_u8(img_u8, oh, ow)
        x = center_crop(x, filter_size * S)
        dy, dcb, dcr = transform_dct_jpeg(x) if filter_size == 8 else transform_dct_size(x, filter_size)
        # UpScaleDCT, cvtransforms.py:56-64.  The JPEG path hands int16 planes to cv2.resize, which rounds the
        # interpolated value back to int16 (cvRound = half to even) [K]; the matrix2dct path stays float.
        rnd = np.rint if filter_size == 8 else (lambda a: a)
        up = lambda d: d if d.shape[:2] == (S, S) else rnd(_bilinear(d, S, S))
        planes = [np.ascontiguousarray(up(d).transpose(2, 0, 1)).astype(np.float32) for d in (dy, dcb, dcr)]   # ToTensorDCT
        return subset_aggregate_normalize(*planes, channels=channels, pattern=dct_pattern, filter_size=filter_size)

    return tf


def rgb_eval_transform(image_size=32):
    """Non-DCT branch (reference datamgr.py:83-88, homomorphic_eval.py:103-107): Resize(1.15x) -> CenterCrop ->
    ToTensor -> Normalize(CIFAR mean/std)."""
    mean = np.array([0.4914, 0.4822, 0.4465], np.float32)
    std = np.array([0.2023, 0.1994, 0.2010], np.float32)

    def tf(img_u8):
        side = int(image_size * 1.15)
        x = resize_u8(img_u8, side, side)
        x = center_crop(x, image_size)
        t = x.astype(np.float32).transpose(2, 0, 1) / 255.0
        return (t - mean[:, None, None]) / std[:, None, None]

    return tf
