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


class readDataset:
    """`readDataset` (`readData.py:32-129`)."""

    def __init__(self, dataroot, loadsize=1024, projection='orthogonal', use_crop=False, device='cuda'):
        if use_crop:
            raise NotImplementedError('use_crop needs the reference\'s third-party person-segmentation web API')
        self.projection_mode = projection
        self.root = dataroot
        files = os.listdir(self.root)
        self.img_files = sorted([os.path.join(self.root, f) for f in files
                                 if f.split('.')[-1].lower() in ['jpg', 'jpeg', 'png'] and
                                 os.path.exists(os.path.join(self.root, f.replace('.' + f.split('.')[-1], '_rect.txt')))])
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
        img_name = os.path.splitext(os.path.basename(img_path))[0]
        ext = img_path.split('.')[-1]
        rect_path = img_path.replace('.' + ext, '_rect.txt')
        depth_path = img_path.replace(img_name + '.' + ext, 'depth/depth_' + img_name + '.png')
        im = _imread_bgr(img_path)
        h, w = im.shape[:2]
        rects = np.loadtxt(rect_path, dtype=np.int32)
        if len(rects.shape) == 1:
            rects = np.array([rects])
        rect = rects[0].tolist()

        trans_mat = np.identity(4)                                          # readData.py:71-79
        scale_im2ndc = 1.0 / float(w // 2)
        scale = w / rect[2]
        trans_mat *= scale
        trans_mat[3, 3] = 1.0
        trans_mat[0, 3] = -scale * (rect[0] + rect[2] // 2 - w // 2) * scale_im2ndc
        trans_mat[1, 3] = scale * (rect[1] + rect[3] // 2 - h // 2) * scale_im2ndc
        intrinsic = np.matmul(trans_mat, np.identity(4))
        projection_matrix = np.identity(4)
        projection_matrix[1, 1] = -1

        out = {
            'name': img_name,
            'img': self._device_image(im, rect, self.load_size).unsqueeze(0),
            'img_512': self._device_image(im, rect, 512).unsqueeze(0),
        }
        if os.path.exists(depth_path):
            depth = _imread_bgr(depth_path)
            out['depth'] = self._device_image(depth, rect, self.load_size).unsqueeze(0)
            out['depth_512'] = self._device_image(depth, rect, 512).unsqueeze(0)
        out.update({
            'calib': torch.Tensor(projection_matrix).float().unsqueeze(0),
            'calib_world': torch.Tensor(intrinsic).float().unsqueeze(0),
            'b_min': np.array([-1, -1, -1]),
            'b_max': np.array([1, 1, 1]),
        })
        return out

    def __len__(self):
        return len(self.img_files)

    def __getitem__(self, index):
        return self.get_item(index)
