"""`reconstruction.write_png`: the debug strip of gen_mesh (`reconstruction.py:36-54`) as a PNG whose deflate stream is
compressed in row bands on threads.  Host-only code: the file must be a well-formed PNG that decodes to the pixels given."""
import struct
import zlib

import numpy as np
import pytest

import pifuhd_amd  # noqa: F401
from pifuhd_amd import reconstruction


def _decode(path):
    blob = open(path, 'rb').read()
    assert blob[:8] == b'\x89PNG\r\n\x1a\n'
    pos, chunks = 8, []
    while pos < len(blob):
        n, tag = struct.unpack('>I4s', blob[pos:pos + 8])
        data = blob[pos + 8:pos + 8 + n]
        assert struct.unpack('>I', blob[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + data) & 0xffffffff
        chunks.append((tag, data))
        pos += 12 + n
    assert [t for t, _ in chunks] == [b'IHDR', b'IDAT', b'IEND']
    w, h, depth, ctype, comp, filt, lace = struct.unpack('>IIBBBBB', chunks[0][1])
    assert (depth, ctype, comp, filt, lace) == (8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, 1 + 3 * w)      # checks the adler32 as well
    assert not raw[:, 0].any()                                      # filter type 0 on every row
    return raw[:, 1:].reshape(h, w, 3)


@pytest.mark.parametrize('shape,bands', [((512, 1536), 8), ((512, 1536), 1), ((37, 101), 8), ((1, 1), 8), ((300, 700), 5)])
def test_png_round_trip(tmp_path, shape, bands):
    rng = np.random.default_rng(shape[0] + bands)
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    img = np.stack([(yy + xx) % 256, (3 * xx) % 256, rng.integers(0, 256, shape)], 2).astype(np.uint8)
    path = str(tmp_path / 'strip.png')
    reconstruction.write_png(path, img, bands=bands)
    assert np.array_equal(_decode(path), img)
    try:
        from PIL import Image
    except Exception:
        return
    assert np.array_equal(np.array(Image.open(path).convert('RGB')), img)       # an independent decoder agrees
