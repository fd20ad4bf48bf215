"""MI355X-native PIFuHD occupancy-query + meshing hot path (drop-in for the reference's
`filter()/query()/get_preds()/calc_normal()`, `mesh_util.reconstruction()` and
`reconstruction.gen_mesh()`).  Compute lives in `csrc/` (hand-written HIP behind a C ABI, see
`include/pifuhd_hip.h`); the Python here is the thin host side that mirrors the reference's
interface.  There is no CPU fallback: anything that computes raises if the HIP library is missing.
"""
import os

# Multi-rank runs (pifuhd_amd.dist: one process per GPU over RCCL) exchange device buffers through IPC handles, and the host
# driver of this pool only offers the dmabuf kind: with the legacy mode left on, RCCL's first collective between two ranks
# fails in hipIpcGetMemHandle ("invalid argument").  The HSA runtime reads the variable when the first HIP call initialises
# it, so it is set here - importing the package precedes any use of the device.  A value set by the caller wins.
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

from . import config, synth  # noqa: E402,F401

__all__ = ['config', 'synth']
