"""Device input pipeline (pifu_preprocess_image, readDataset mirror) and the standalone index() op against the
oracle restatement: uint8 arithmetic bit-exact, bilinear sampling to fp32 rounding."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ctx():
    from pifuhd_amd.nets import _Ctx
    return _Ctx('cuda')


@pytest.mark.parametrize('shape,rect,size', [
    ((37, 53), (0, 0, 53, 37), 64),            # whole image, upscale
    ((37, 53), (-7, -5, 70, 50), 32),          # rect leaves the image on every side
    ((64, 64), (0, 0, 64, 64), 32),            # exact 2x: INTER_AREA fast path
    ((64, 64), (0, 0, 64, 64), 64),            # identity
    ((120, 90), (10, 20, 61, 77), 128),        # odd crop, upscale
    ((300, 200), (150, 250, 100, 100), 48),    # crop hanging over the bottom-right corner
])
def test_preprocess_bit_exact(shape, rect, size):
    from oracle import pifu_oracle as O
    from pifuhd_amd import _lib
    from pifuhd_amd.nets import _ptr, _stream_ptr
    ctx = _ctx()
    rng = np.random.default_rng(hash((shape, rect, size)) % (2 ** 32))
    bgr = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    src = torch.from_numpy(bgr).cuda()
    out = torch.empty(3, size, size, device='cuda')
    _lib.call('pifu_preprocess_image', ctx.h, _ptr(src), shape[0], shape[1], (C.c_int32 * 4)(*rect), size, _ptr(out),
              _stream_ptr(ctx.device))
    assert np.array_equal(out.cpu().numpy(), O.preprocess_image(bgr, rect, size))


def test_index_matches_oracle():
    from oracle import pifu_oracle as O
    from pifuhd_amd import _lib
    from pifuhd_amd.nets import _ptr, _stream_ptr
    ctx = _ctx()
    g = torch.Generator().manual_seed(2)
    feat = torch.randn(1, 3, 40, 56, generator=g)
    uv = torch.rand(1, 2, 5000, generator=g) * 2.4 - 1.2            # some points outside [-1, 1]: zero padding
    uv[0, :, :4] = torch.tensor([[-1., 1., -1., 1.], [-1., -1., 1., 1.]])      # the four corners exactly
    want = O.index(feat, uv)[0].numpy()
    out = torch.empty(3, 5000, device='cuda')
    feat_d, uv_d = feat.cuda().contiguous(), uv[0].cuda().contiguous()
    _lib.call('pifu_index', ctx.h, _ptr(feat_d), 3, 40, 56, _ptr(uv_d), 5000, _ptr(out), _stream_ptr(ctx.device))
    assert np.abs(out.cpu().numpy() - want).max() < 1e-5
    from pifuhd_amd.nets import index
    assert torch.equal(index(feat_d, uv.cuda())[0], out)


def test_read_dataset_end_to_end(tmp_path):
    from PIL import Image
    from oracle import pifu_oracle as O
    from pifuhd_amd.readData import addrect, readDataset
    rng = np.random.default_rng(11)
    rgb = rng.integers(0, 256, (90, 70, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / 'person.png')
    (tmp_path / 'person_rect.txt').write_text('-6 4 80 80\n')
    Image.fromarray(rgb).save(tmp_path / 'norect.png')               # no _rect.txt: not part of the dataset
    os.makedirs(tmp_path / 'depth')
    dep = rng.integers(0, 256, (90, 70, 3), dtype=np.uint8)
    Image.fromarray(dep).save(tmp_path / 'depth' / 'depth_person.png')
    ds = readDataset(str(tmp_path), loadsize=128)
    assert len(ds) == 1 and ds.projection_mode == 'orthogonal'
    d = ds[0]
    assert d['name'] == 'person' and d['img'].shape == (1, 3, 128, 128) and d['img_512'].shape == (1, 3, 512, 512)
    bgr = np.ascontiguousarray(rgb[:, :, ::-1])
    rect = [-6, 4, 80, 80]
    assert np.array_equal(d['img'][0].cpu().numpy(), O.preprocess_image(bgr, rect, 128))
    assert np.array_equal(d['img_512'][0].cpu().numpy(), O.preprocess_image(bgr, rect, 512))
    assert np.array_equal(d['depth'][0].cpu().numpy(), O.preprocess_image(np.ascontiguousarray(dep[:, :, ::-1]), rect, 128))
    calib, world = O.calib_matrices(rect, 70, 90)
    assert np.array_equal(d['calib'][0].numpy(), calib) and np.array_equal(d['calib_world'][0].numpy(), world)
    assert np.array_equal(addrect(bgr, rect), O.addrect(bgr, rect))
    with pytest.raises(NotImplementedError):
        readDataset(str(tmp_path), use_crop=True)
