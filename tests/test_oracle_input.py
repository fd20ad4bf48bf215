"""Input pipeline restatement (oracle, SURVEY 8f row f1: readData.py).  cv2 is not installed anywhere in this image,
so these are known-answer / property tests of the restated cv::resize fixed-point path, not a pin against OpenCV
(DESIGN.md says "parity unpinned" for this row)."""
import numpy as np

from oracle import pifu_oracle as O


def _brute_resize(img, d):
    """Scalar transcription of the published per-pixel formulas (independent of the vectorised oracle code)."""
    H, W, Cn = img.shape
    out = np.zeros((d, d, Cn), np.uint8)
    sx_scale = 1.0 / (float(d) / W)
    sy_scale = 1.0 / (float(d) / H)
    for dy in range(d):
        fy = np.float32((dy + 0.5) * sy_scale - 0.5)
        sy = int(np.floor(fy)); fy = np.float32(fy - np.float32(sy))
        b0 = int(np.rint(np.float32(np.float32(1) - fy) * np.float32(2048))); b1 = int(np.rint(fy * np.float32(2048)))
        y0 = min(max(sy, 0), H - 1); y1 = min(max(sy + 1, 0), H - 1)
        for dx in range(d):
            fx = np.float32((dx + 0.5) * sx_scale - 0.5)
            sx = int(np.floor(fx)); fx = np.float32(fx - np.float32(sx))
            if sx < 0:
                fx = np.float32(0); sx = 0
            if sx >= W - 1:
                fx = np.float32(0); sx = W - 1
            x1 = min(sx + 1, W - 1)
            a0 = int(np.rint(np.float32(np.float32(1) - fx) * np.float32(2048))); a1 = int(np.rint(fx * np.float32(2048)))
            for c in range(Cn):
                S0 = int(img[y0, sx, c]) * a0 + int(img[y0, x1, c]) * a1
                S1 = int(img[y1, sx, c]) * a0 + int(img[y1, x1, c]) * a1
                out[dy, dx, c] = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
    return out


def test_resize_known_answers():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (23, 31, 3), dtype=np.uint8)
    sq = img[:23, :23]
    assert np.array_equal(O.cv2_resize_linear_u8(sq, 23), sq)                         # scale 1 = copy
    const = np.full((17, 29, 3), 201, np.uint8)
    for d in (5, 17, 40):
        assert (O.cv2_resize_linear_u8(const, d) == 201).all()                        # weights sum to 2048
    big = rng.integers(0, 256, (24, 24, 3), dtype=np.uint8)
    want = ((big[0::2, 0::2].astype(int) + big[0::2, 1::2] + big[1::2, 0::2] + big[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    assert np.array_equal(O.cv2_resize_linear_u8(big, 12), want)                      # exact 2x: INTER_AREA fast path
    ramp = np.repeat(np.arange(0, 250, 10, dtype=np.uint8)[None, :, None], 8, 0).repeat(3, 2)
    up = O.cv2_resize_linear_u8(ramp, 50).astype(int)
    assert (np.diff(up[0, :, 0]) >= 0).all() and up[0, 0, 0] == 0 and up[0, -1, 0] == 240


def test_resize_vectorised_equals_scalar_formulas():
    rng = np.random.default_rng(5)
    for (h, w, d) in [(9, 13, 16), (21, 8, 7), (10, 10, 32), (33, 47, 12)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(O.cv2_resize_linear_u8(img, d), _brute_resize(img, d))


def test_addrect_zero_border_and_calib():
    rng = np.random.default_rng(7)
    img = rng.integers(1, 256, (20, 30, 3), dtype=np.uint8)
    for rect in [(3, 4, 10, 12), (-5, -2, 20, 15), (15, 10, 30, 25), (-4, -4, 40, 30), (0, 0, 30, 20), (29, 19, 5, 5)]:
        x, y, w, h = rect
        got = O.addrect(img, rect)
        want = np.zeros((h, w, 3), np.uint8)
        for r in range(h):
            for c in range(w):
                if 0 <= y + r < 20 and 0 <= x + c < 30:
                    want[r, c] = img[y + r, x + c]
        assert np.array_equal(got, want), rect
    calib, world = O.calib_matrices([100, 50, 400, 600], 1000, 800)
    assert np.array_equal(calib, np.diag([1, -1, 1, 1]).astype(np.float32))
    s = 1000 / 400
    assert np.allclose(np.diag(world), [s, s, s, 1])
    assert np.isclose(world[0, 3], -s * (100 + 200 - 500) / 500) and np.isclose(world[1, 3], s * (50 + 300 - 400) / 500)


def test_preprocess_range_and_channel_order():
    img = np.zeros((8, 8, 3), np.uint8)
    img[..., 0] = 255                                   # pure blue in BGR
    t = O.preprocess_image(img, (0, 0, 8, 8), 8)
    assert t.dtype == np.float32 and t.shape == (3, 8, 8)
    assert (t[2] == 1).all() and (t[0] == -1).all() and (t[1] == -1).all()            # RGB planes: blue is plane 2
