"""Input pipeline: drop-in for the reference's `readData.py` (`readDataset`, `addrect`).

Same constructor, same file-discovery rule, same dict out of `get_item` (`readData.py:106-116`), but the pixel
work - rect crop with a zero border, the two cv2.resize calls, BGR->RGB, ToTensor, Normalize - runs on the device
through `pifu_preprocess_image` (`include/pifuhd_hip.h`) on the decoded BGR bytes.  Differences, all loud:
`use_crop=True` (the reference's `crop_people`, a third-party web API) is not available; files are decoded with PIL
(OpenCV is not a dependency here), so JPEG pixels can differ from cv2.imread's by the decoder's rounding, PNGs not;
a missing `depth/depth_<name>.png` only drops the `depth*` entries (the reference crashes on it).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .nets import _Ctx, _ptr, _stream_ptr


def addrect(img, rect):
    """`addrect` (`readData.py:15-29`) on a host array: crop (x, y, w, h) with a zero border."""
    x, y, w, h = [int(v) for v in rect]
    H, W = img.shape[:2]
    left = abs(x) if x < 0 else 0
    top = abs(y) if y < 0 else 0
    right = abs(W - (x + w)) if x + w >= W else 0
    bottom = abs(H - (y + h)) if y + h >= H else 0
    new_img = np.zeros((H + top + bottom, W + left + right) + img.shape[2:], img.dtype)
    new_img[top:top + H, left:left + W] = img
    x, y = x + left, y + top
    return new_img[y:(y + h), x:(x + w)]


def _imread_bgr(path):
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert('RGB'))
    return np.ascontiguousarray(rgb[:, :, ::-1])


_IMAGE_EXTENSIONS = ('jpg', 'jpeg', 'png')


def _sidecar(path, suffix):
    """`<image>.<ext>` -> `<image><suffix>`, with the reference's string rule (`readData.py:39,55`): every occurrence of
    '.<ext>' in the path is replaced, not only the last one."""
    ext = path.split('.')[-1]
    return path.replace('.' + ext, suffix)


def discover_subjects(root):
    """Sorted paths of the images under `root` that come with a `<stem>_rect.txt` crop file (`readData.py:36-40`)."""
    found = []
    for entry in os.listdir(root):
        if entry.split('.')[-1].lower() not in _IMAGE_EXTENSIONS:
            continue
        path = os.path.join(root, entry)
        if os.path.exists(_sidecar(path, '_rect.txt')):
            found.append(path)
    return sorted(found)


def crop_to_world(width, height, rect):
    """4x4 `calib_world` of a crop (`readData.py:71-79`): uniform zoom `width / rect_w` and the offset of the crop centre
    from the image centre in NDC units of the full image (x negated, y not).  The arithmetic keeps the reference's
    order - integer halves, a reciprocal pixel-to-NDC factor - so the float64 entries are bit-identical to it."""
    x, y, rw, rh = rect
    zoom = width / rw
    ndc_per_pixel = 1.0 / float(width // 2)
    m = np.identity(4) * zoom
    m[3, 3] = 1.0
    m[0, 3] = -zoom * (x + rw // 2 - width // 2) * ndc_per_pixel
    m[1, 3] = zoom * (y + rh // 2 - height // 2) * ndc_per_pixel
    return m


class readDataset:
    """`readDataset` (`readData.py:32-129`)."""

    def __init__(self, dataroot, loadsize=1024, projection='orthogonal', use_crop=False, device='cuda'):
        if use_crop:
            raise NotImplementedError('use_crop needs the reference\'s third-party person-segmentation web API')
        self.projection_mode = projection
        self.root = dataroot
        self.img_files = discover_subjects(dataroot)
        self.load_size = loadsize
        self.use_crop = use_crop
        self._ctx = _Ctx(device)

    def _device_image(self, bgr, rect, size):
        dev = self._ctx.device
        src = torch.from_numpy(bgr).to(dev)
        out = torch.empty(3, size, size, device=dev, dtype=torch.float32)
        r = (C.c_int32 * 4)(*[int(v) for v in rect])
        _lib.call('pifu_preprocess_image', self._ctx.h, _ptr(src), bgr.shape[0], bgr.shape[1], r, int(size), _ptr(out),
                  _stream_ptr(dev))
        return out

    def get_item(self, index):
        img_path = self.img_files[index]
        name = os.path.splitext(os.path.basename(img_path))[0]
        depth_path = os.path.join(os.path.dirname(img_path), 'depth', 'depth_' + name + '.png')
        pixels = _imread_bgr(img_path)
        height, width = pixels.shape[:2]
        rect = np.atleast_2d(np.loadtxt(_sidecar(img_path, '_rect.txt'), dtype=np.int32))[0].tolist()

        flip_y = np.identity(4)
        flip_y[1, 1] = -1                                                       # the orthographic camera: y up
        item = {
            'name': name,
            'img': self._device_image(pixels, rect, self.load_size).unsqueeze(0),
            'img_512': self._device_image(pixels, rect, 512).unsqueeze(0),
        }
        if os.path.exists(depth_path):
            depth = _imread_bgr(depth_path)
            item['depth'] = self._device_image(depth, rect, self.load_size).unsqueeze(0)
            item['depth_512'] = self._device_image(depth, rect, 512).unsqueeze(0)
        item['calib'] = torch.from_numpy(flip_y).float().unsqueeze(0)
        item['calib_world'] = torch.from_numpy(crop_to_world(width, height, rect)).float().unsqueeze(0)
        item['b_min'] = np.array([-1, -1, -1])
        item['b_max'] = np.array([1, 1, 1])
        return item

    def __len__(self):
        return len(self.img_files)

    def __getitem__(self, index):
        return self.get_item(index)
