"""Point the ctypes binding at the lab build (tools/lab/libpifuhd_lab.so) instead of the product library.

    import lab; lab.use()        # before the first pifuhd_amd call; (re)builds the library when it is missing or older than its sources

The lab library is the product sources compiled with -DPIFU_LAB: same ABI plus the PIFU_* environment switches, the
predecessor GEMM kernels and pifu_debug_read.  Nothing under rgb-d-pifuhd_amd/ knows about it.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LAB_PATH = os.path.join(HERE, 'libpifuhd_lab.so')


def build(experiments=False):
    cmd = ['make', '-C', HERE, '-j4'] + (['EXPERIMENTS=1'] if experiments else [])
    subprocess.check_call(cmd)


def use(experiments=False):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    build(experiments)          # make: a no-op when the library is newer than the product sources it is compiled from
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd import _lib
    prev = (_lib.LIB_PATH, _lib._lib)
    _lib.LIB_PATH, _lib._lib = LAB_PATH, None
    return prev


def restore(prev):
    from pifuhd_amd import _lib
    _lib.LIB_PATH, _lib._lib = prev
