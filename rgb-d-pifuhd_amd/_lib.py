"""ctypes binding of `csrc/libpifuhd_hip.so` (C ABI: `include/pifuhd_hip.h`).

There is deliberately no fallback: if the library is missing or a call fails, a `PifuError` is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libpifuhd_hip.so')

ABI_VERSION = 5          # PIFU_ABI_VERSION of include/pifuhd_hip.h this binding was written against
OK = 0
ERR_INVALID, ERR_HIP, ERR_STATE, ERR_NOMEM, ERR_MC_LEVEL, ERR_MC_EMPTY = -1, -2, -3, -4, -5, -6
LEVEL_COARSE, LEVEL_FINE = 0, 1
NORM_NONE, NORM_GROUP, NORM_BATCH_EVAL, NORM_BATCH_TRAIN = 0, 1, 2, 3
PROF_GEMM, PROF_SAMPLER, PROF_MC, PROF_ENC, PROF_MC_NODES, PROF_WALK, PROF_ENC_CANONICAL = 0, 1, 2, 3, 4, 5, 6
Q_COARSE_ONLY, Q_NO_FINE_MASK, Q_FULL_COARSE, Q_NO_COLUMN_SHARING, Q_NO_DEPTH_WALK, Q_NO_COARSE_MASK = 1, 2, 4, 8, 16, 32
ENC_ALL_STACKS, ENC_DEFER_JOIN, ENC_NORMALS_ONLY, ENC_NORMALS_GIVEN, ENC_NO_WINOGRAD = 1, 2, 4, 8, 16
PIX_OFF, PIX_AUTO, PIX_ALWAYS = 0, 1, 2          # pifu_set_pixel_tables
POINTWISE_CHUNK = 16384                   # PIFU_POINTWISE_CHUNK
PIXEL_TABLE_MIN_POINTS = 16384            # the automatic threshold: max(this, H * W of the coarse map) points per call
ARENA_QUERY, ARENA_MC, ARENA_OCTREE, ARENA_COLUMN, ARENA_PIXEL, ARENA_MISC = 0, 1, 2, 3, 4, 5
COMM_ID_BYTES = 128


class PifuError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f'pifuhd_hip error {code}: {message}')
        self.code = code
        self.message = message


_p = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int
_f = C.c_float
_d = C.c_double

# name -> (restype, argtypes); must list every symbol declared in include/pifuhd_hip.h
SIGNATURES = {
    'pifu_last_error': (C.c_char_p, []),
    'pifu_abi_version': (_i32, []),
    'pifu_ctx_create': (_i32, [_i32, C.POINTER(_p)]),
    'pifu_ctx_destroy': (_i32, [_p]),
    'pifu_reserve': (_i32, [_p, _i64]),
    'pifu_reserve_auto': (_i32, [_p, _i64, _d, C.POINTER(_i64)]),
    'pifu_workspace_bytes': (_i64, [_p]),
    'pifu_set_workspace_limit': (_i32, [_p, _i64]),
    'pifu_arena_bytes': (_i64, [_p, _i32]),
    'pifu_set_host_threads': (_i32, [_i32]),
    'pifu_comm_unique_id': (_i32, [_p]),
    'pifu_comm_create': (_i32, [_i32, _i32, _i32, _p, C.POINTER(_p)]),
    'pifu_comm_destroy': (_i32, [_p]),
    'pifu_allgather_f32': (_i32, [_p, _p, _p, _i64, _p]),
    'pifu_allgather_bytes': (_i32, [_p, _p, _p, _i64, _p]),
    'pifu_mlp_define': (_i32, [_p, _i32, _p, _i32, _p, _i32, _i32, _i32]),
    'pifu_mlp_set_layer': (_i32, [_p, _i32, _i32, _p, _p, _p, _p, _p, _p]),
    'pifu_set_precision': (_i32, [_p, _i32]),
    'pifu_set_pixel_tables': (_i32, [_p, _i32]),
    'pifu_drop_tables': (_i32, [_p]),
    'pifu_set_depth_normalizer': (_i32, [_p, _f, _f]),
    'pifu_set_features': (_i32, [_p, _i32, _p, _i32, _i32, _i32, _p]),
    'pifu_enc_set_param': (_i32, [_p, C.c_char_p, _p, _p, _i32]),
    'pifu_enc_set_param_device': (_i32, [_p, C.c_char_p, _p, _p, _i32, _p]),
    'pifu_enc_clear': (_i32, [_p]),
    'pifu_filter_coarse': (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _i32, _p]),
    'pifu_filter_fine': (_i32, [_p, _p, _i32, _i32, _i32, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    'pifu_query': (_i32, [_p, _p, _i64, _p, _i64, _i32, _p, _p, _p, _p]),
    'pifu_effective_chunk': (_i64, [_p, _i64, _i32, _i32]),
    'pifu_eval_lattice': (_i32, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _p, _p]),
    'pifu_octree_begin': (_i32, [_p, _i32, _p, _p, _p]),
    'pifu_octree_select': (_i32, [_p, _i32, _i32, _p, _p, _i64, C.POINTER(_i64), _p]),
    'pifu_octree_commit': (_i32, [_p, _i32, _i32, _d, _p, _p, _i64, _p, _p, _p]),
    'pifu_eval_grid': (_i32, [_p, _i32, _p, _p, _i64, _i32, _d, _i32, _p, _p, C.POINTER(_i64), _p]),
    'pifu_mc_count': (_i32, [_p, _p, _i32, _i32, _i32, _i32, _d, C.POINTER(_i64), C.POINTER(_i64), _p]),
    'pifu_mc_emit': (_i32, [_p, _p, _p, _p]),
    'pifu_transform_verts': (_i32, [_p, _p, _i64, _p, _p, _p]),
    'pifu_mc_slab_count': (_i32, [_p, _p, _i32, _i32, _i32, _i32, _i32, _d, C.POINTER(_i64), C.POINTER(_i64),
                                  C.POINTER(C.c_float), C.POINTER(C.c_float), _p]),
    'pifu_mc_slab_verts': (_i32, [_p, _p, _i64, _p, _p]),
    'pifu_mc_slab_faces': (_i32, [_p, _p, _p, _p]),
    'pifu_calc_normal': (_i32, [_p, _p, _i64, _p, _f, _p, _p, _p]),
    'pifu_calc_normal_coarse': (_i32, [_p, _p, _i64, _p, _f, _p, _p, _p]),
    'pifu_calc_normal_chunked': (_i32, [_p, _p, _i64, _i64, _p, _f, _p, _p, _p]),
    'pifu_index': (_i32, [_p, _p, _i32, _i32, _i32, _p, _i64, _p, _p]),
    'pifu_preprocess_image': (_i32, [_p, _p, _i32, _i32, _p, _i32, _p, _p]),
    'pifu_face_components': (_i32, [_p, _p, _i64, _p, _p, _p]),
    'pifu_esti_color': (_i32, [_p, _p, _i64, _p, _p]),
    'pifu_bce_loss': (_i32, [_p, _p, _p, _i32, _i64, _p, _f, _p, _i32, _i32, _p, _p]),
    'pifu_profile_enable': (_i32, [_p, _i32]),
    'pifu_profile_read': (_i32, [_p, _i32, C.POINTER(_d), C.POINTER(_i64), C.POINTER(_d), _i32]),
    'pifu_clock_probe': (_i32, [_p, _p, _p]),
    'pifu_write_obj': (_i32, [C.c_char_p, _p, _p, _i64, _p, _i64]),
    'pifu_obj_begin': (_i32, [C.c_char_p, _p, _i64, _i64, _i64, C.POINTER(_p)]),
    'pifu_obj_layout': (_i32, [_p, _p, _p, _i64]),
    'pifu_obj_ready': (_i32, [_p, _i64]),
    'pifu_obj_finish': (_i32, [_p, _p, _p, _i64]),
}

_lib = None


def load():
    """Load the HIP library (once).  Raises PifuError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PifuError(ERR_STATE, f'{LIB_PATH} is missing: build it with `python -c "import '
                                       f'__graft_entry__ as g; g.build()"` (make -C rgb-d-pifuhd_amd/csrc); '
                                       f'there is no CPU fallback')
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        built = lib.pifu_abi_version()
        if built != ABI_VERSION:
            raise PifuError(ERR_STATE, f'{LIB_PATH} was built for ABI version {built}, this binding expects {ABI_VERSION}: '
                                       f'rebuild it (make -C rgb-d-pifuhd_amd/csrc) - a stale library would misread arguments')
        _lib = lib
    return _lib


def check(code):
    if code != OK:
        raise PifuError(code, load().pifu_last_error().decode('utf-8', 'replace'))
    return code


def call(name, *args):
    return check(getattr(load(), name)(*args))
