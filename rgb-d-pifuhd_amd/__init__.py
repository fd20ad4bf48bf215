"""MI355X-native PIFuHD occupancy-query + meshing hot path (drop-in for the reference's
`filter()/query()/get_preds()/calc_normal()`, `mesh_util.reconstruction()` and
`reconstruction.gen_mesh()`).  Compute lives in `csrc/` (hand-written HIP behind a C ABI, see
`include/pifuhd_hip.h`); the Python here is the thin host side that mirrors the reference's
interface.  There is no CPU fallback: anything that computes raises if the HIP library is missing.
"""
from . import config, synth  # noqa: F401

__all__ = ['config', 'synth']
