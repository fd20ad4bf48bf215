"""Shape-defining configuration of the two PIFuHD levels.

The reference keeps these values as argparse defaults (`options.py:18,49,73,78,86-108,152,162-163`)
and copies the `_global` / `_local` variants onto `opt.hg_dim / mlp_dim / mlp_res_layers /
num_stack` before building each net (`train.py:102-105,115-118`).  Only the fields that shape the
hot path (filter -> query -> gen_mesh) are kept here; everything else in the reference's option
system is training / data-loading control plane and out of scope.
"""
from __future__ import annotations

import copy
from types import SimpleNamespace


def coarse_opt(**overrides) -> SimpleNamespace:
    """Options of the coarse level `PIFuNetwNML` (reference defaults, `_global` variant)."""
    opt = SimpleNamespace(
        loadSize=1024,            # options.py:18  (DepthNormalizer uses loadSize // 2)
        loadSizeBig=1024,         # options.py:152
        resolution=512,           # options.py:49
        z_size=200.0,             # options.py:73
        norm='batch',             # options.py:78  HGFilter norm: 'batch' | 'group'
        num_stack=4,              # options.py:86  (num_stack_global)
        hg_depth=2,               # options.py:89
        hg_down='ave_pool',       # options.py:90
        hg_dim=256,               # options.py:93  (hg_dim_global)
        mlp_norm='group',         # options.py:95
        mlp_dim=[257, 1024, 512, 256, 128, 1],      # options.py:96
        mlp_res_layers=[2, 3, 4],                   # options.py:102
        merge_layer=2,            # options.py:108
        use_front_normal=True,    # options.py:162
        use_back_normal=True,     # options.py:163
        train_full_pifu=False,
        in_image_channels=3,      # SURVEY C-7: 3 as committed; 6 = RGB||depth variant
    )
    for k, v in overrides.items():
        setattr(opt, k, v)
    return opt


def fine_opt(base: SimpleNamespace | None = None, **overrides) -> SimpleNamespace:
    """Options of the fine level `PIFuMRNet` (reference defaults, `_local` variant)."""
    opt = copy.deepcopy(base) if base is not None else coarse_opt()
    opt.num_stack = 1                               # options.py:87
    opt.hg_dim = 16                                 # options.py:92
    opt.mlp_dim = [272, 512, 256, 128, 1]           # options.py:98
    opt.mlp_res_layers = [1, 2]                     # options.py:104
    for k, v in overrides.items():
        setattr(opt, k, v)
    return opt
