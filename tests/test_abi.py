"""The C-ABI library builds, loads without a GPU and exports every symbol include/pifuhd_hip.h declares."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'rgb-d-pifuhd_amd', 'csrc'), '-j4'])
    return _lib


def test_header_symbols_exported_and_bound(lib):
    header = open(os.path.join(ROOT, 'include', 'pifuhd_hip.h')).read()
    declared = set(re.findall(r'\b(pifu_[a-z0-9_]+)\s*\(', header))
    declared -= {'pifu_ctx', 'pifu_stream', 'pifu_comm'}
    assert len(declared) >= 20
    handle = lib.load()
    for name in sorted(declared):
        assert hasattr(handle, name), f'{name} declared in the header but not exported'
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    assert handle.pifu_abi_version() == lib.ABI_VERSION == 5


def test_library_exports_exactly_the_header(lib):
    """No lab switches, diagnostics or C++ internals in the shipped library: its WHOLE dynamic symbol table (every defined
    symbol `nm -D` lists, not only the `pifu_*` names) is the set of declarations of the header.  The library is built
    with -fvisibility=hidden, the entry points carry PIFU_API, and the link uses csrc/exports.map."""
    header = open(os.path.join(ROOT, 'include', 'pifuhd_hip.h')).read()
    declared = set(re.findall(r'\b(pifu_[a-z0-9_]+)\s*\(', header)) - {'pifu_ctx', 'pifu_stream', 'pifu_comm'}
    # every declaration carries the export attribute
    assert len(re.findall(r'^PIFU_API\s', header, re.M)) == len(declared)
    lib.load()
    out = subprocess.check_output(['nm', '-D', '--defined-only', lib.LIB_PATH], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split()}
    assert exported == declared, exported ^ declared
    blob = open(lib.LIB_PATH, 'rb').read()
    for switch in (b'PIFU_GEMM_VARIANT', b'PIFU_COLUMN_MODE', b'PIFU_DEPTH_WALK', b'PIFU_PRECISION', b'PIFU_DEBUG_STAMPS'):
        assert switch not in blob, switch


def test_no_cpu_fallback(lib):
    """The product path refuses to run without a GPU instead of silently computing elsewhere."""
    import torch
    from pifuhd_amd import config, nets
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(Exception):
        nets.PIFuNetwNML(config.coarse_opt(), 'orthogonal', 'cuda')
    with pytest.raises(lib.PifuError):
        nets.PIFuNetwNML(config.coarse_opt(), 'orthogonal', 'cpu')


def test_binding_refuses_a_library_of_another_abi_version(lib, monkeypatch):
    """`_lib.load()` compares `pifu_abi_version()` with the version the binding was written against: a stale or newer
    library (argument meanings changed between versions, see the header) is refused instead of misread."""
    lib.load()
    header = open(os.path.join(ROOT, 'include', 'pifuhd_hip.h')).read()
    assert int(re.search(r'#define\s+PIFU_ABI_VERSION\s+(\d+)', header).group(1)) == lib.ABI_VERSION
    monkeypatch.setattr(lib, '_lib', None)
    monkeypatch.setattr(lib, 'ABI_VERSION', lib.ABI_VERSION + 1)
    with pytest.raises(lib.PifuError, match='ABI version'):
        lib.load()
    monkeypatch.undo()
    assert lib.load().pifu_abi_version() == lib.ABI_VERSION


def test_graft_entry_build_passes():
    """The driver's "does it build" check (`__graft_entry__.build()`): every library compiles for gfx950 without a GPU and
    the binding accepts what was built (its ABI version included)."""
    import importlib
    entry = importlib.import_module('__graft_entry__')
    entry.build()


def test_header_binds_from_plain_c(lib, tmp_path):
    """The boundary is a C ABI: `include/pifuhd_hip.h` compiles as C99 with warnings as errors, a C program links the library,
    gets the header's ABI version, error codes instead of crashes, and the host-only OBJ writer produces the reference's text
    (`mesh_util.py:189-198`) - no Python, no C++ and no GPU involved."""
    import shutil
    if shutil.which('gcc') is None:
        pytest.skip('no C compiler')
    lib.load()
    libdir = os.path.dirname(lib.LIB_PATH)
    exe = tmp_path / 'abi_smoke'
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Wextra', '-Werror', '-pedantic', '-I', os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'tests', 'c', 'abi_smoke.c'), '-o', str(exe), '-L', libdir, '-lpifuhd_hip',
                           '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib'])
    out = tmp_path / 'c.obj'
    r = subprocess.run([str(exe), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == 'abi %d ok' % lib.ABI_VERSION
    from oracle import pifu_oracle as O
    import numpy as np
    v = np.array([[0.0, 0.5, -1.25], [1.0, 2.0, 3.0], [-0.00004, 0.00005, 10.0]])
    c = np.array([[0.1, 0.2, 0.3], [0.5, 0.5, 0.5], [1.0, 0.0, 0.25]])
    assert O.obj_text(v, np.array([[0, 1, 2]], np.int32), c) == out.read_text() == open(str(out) + '.2').read()


def test_product_never_imports_oracle():
    """No file of the product package imports, loads or links anything under oracle/."""
    pkg = os.path.join(ROOT, 'rgb-d-pifuhd_amd')
    pat = re.compile(r'(^\s*(from|import)\s+oracle\b)|libmc_oracle|oracle/|pifu_oracle', re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.cpp', '.h', 'Makefile')):
                text = open(os.path.join(dirpath, f), errors='replace').read()
                assert not pat.search(text), f'{f} references the oracle'


def test_obj_writer_host_only(lib, tmp_path):
    """pifu_write_obj is host-only code: byte-identical to the reference's format string."""
    import numpy as np
    from oracle import pifu_oracle as O
    from pifuhd_amd import mesh_util
    rs = np.random.RandomState(0)
    # random values + exact rounding ties (k/32, odd/20000), signed zeros, tiny negatives, large magnitudes
    v = np.concatenate([rs.randn(5000, 3) * 3, rs.randint(-64, 64, size=(600, 3)) / 32.0,
                        (rs.randint(0, 20000, size=(600, 3)) * 2 + 1) / 20000.0,
                        np.array([[0.0, -0.0, 1e-9], [-1e-9, 0.00005, -0.00005], [123456.78905, -99999.99995, 1e10]]),
                        # outside the fast formatter: huge finite values (315 characters), infinities, NaNs of either sign
                        np.array([[1.7e308, -1.7e308, 1e11], [np.inf, -np.inf, np.nan], [-np.nan, 3e200, -2.5e15]])])
    c = rs.rand(len(v), 3)
    f = rs.randint(0, len(v), size=(8000, 3)).astype(np.int32)
    p = tmp_path / 'a.obj'
    mesh_util.save_obj_mesh_with_color(str(p), v, f, c)
    assert p.read_text() == O.obj_text(v, f, c)
    # several blocks of the ordered block writer (16 384 lines each), a block straddling the vertex / face boundary, blocks
    # with and without out-of-range values, an empty mesh
    v = np.concatenate([rs.randn(40000, 3) * 3, np.array([[1.7e308, -1e300, np.nan]]), rs.randn(9000, 3)])
    c = rs.rand(len(v), 3)
    f = rs.randint(0, len(v), size=(70001, 3)).astype(np.int32)
    mesh_util.save_obj_mesh_with_color(str(p), v, f, c)
    assert p.read_text() == O.obj_text(v, f, c)
    mesh_util.save_obj_mesh_with_color(str(p), v[:0], f[:0], c[:0])
    assert p.read_text() == ''
    # the two-step writer gen_mesh uses (face lines formatted while the colours are still being computed): same bytes, also
    # from the reversed view `reconstruction()` returns for a mirroring calib (no copy: strides go through the ABI)
    want = O.obj_text(v, f[:, ::-1], c)
    w = mesh_util.ObjFile(str(p), f[:, ::-1])
    w.finish(v, c)
    assert p.read_text() == want
    w = mesh_util.ObjFile(str(p), f[:0])
    w.finish(v[:0], c[:0])
    assert p.read_text() == ''
    # the streaming form of gen_mesh's colour-loop pipeline: positions complete, colours (float32, the values the reference
    # copies into its float64 colour array) arriving later.  The writer lays the file out from the positions alone - a colour
    # in [0, 10) is always 6 characters - writes the faces in place at once and the vertex blocks as their colours are
    # declared ready; same bytes as the one-call writer on the float64 copy of those colours
    vs = np.ascontiguousarray(v[np.isfinite(v).all(1) & (np.abs(v) < 1e10).all(1)])      # positions inside the fast formatter's range
    c32 = rs.rand(len(vs), 3).astype(np.float32)
    c32[::97] = 0.0
    c32[5::113] = 1.0
    fs = rs.randint(0, len(vs), size=(70001, 3)).astype(np.int32)
    want = O.obj_text(vs, fs[:, ::-1], c32.astype(np.float64))
    late = np.full_like(c32, np.nan)                               # what the buffer holds before the colours arrive
    w = mesh_util.ObjFile(str(p), fs[:, ::-1])
    w.layout(vs, late)
    for upto in (0, 1, 7, 16384, 16385, 40001, len(vs)):
        late[:upto] = c32[:upto]
        w.ready(upto)
    assert p.read_text() != want                                   # nothing under the final name before finish()
    w.finish()
    assert p.read_text() == want and not (tmp_path / 'a.obj.part').exists()
    # colours that do not print as 6 characters (NaN, negative, >= 10) and positions outside the fast range break the
    # predicted layout: the finish call notices and rewrites the file in one pass - still the one-call writer's bytes
    odd = c32.copy()
    odd[3] = [np.nan, -0.25, 12.5]
    odd[20000, 1] = -1e-9
    for verts_, cols_ in ((vs, odd), (v, rs.rand(len(v), 3).astype(np.float32))):
        ff = rs.randint(0, len(verts_), size=(50000, 3)).astype(np.int32)
        w = mesh_util.ObjFile(str(p), ff)
        w.layout(np.ascontiguousarray(verts_), cols_)
        w.ready(len(verts_) // 2)
        w.finish()
        assert p.read_text() == O.obj_text(verts_, ff, cols_.astype(np.float64))
    # an empty mesh, and a writer abandoned mid-stream: the previous file stays, the partial one goes
    w = mesh_util.ObjFile(str(p), fs[:0])
    w.layout(vs[:0], c32[:0])
    w.finish()
    assert p.read_text() == ''
    p.write_text(want)
    w = mesh_util.ObjFile(str(p), fs)
    w.layout(vs, c32)
    w.ready(100)
    del w
    assert p.read_text() == want and not (tmp_path / 'a.obj.part').exists()
    # a writer abandoned between the two steps (an exception in the colour loop) leaves the previous file as it was, and
    # creates none where there was none - the reference writes only once everything has succeeded
    p.write_text('previous result\n')
    w = mesh_util.ObjFile(str(p), f)
    del w
    assert p.read_text() == 'previous result\n'
    q = tmp_path / 'never.obj'
    w = mesh_util.ObjFile(str(q), f)
    del w
    assert not q.exists()


def test_checkpoint_io_roundtrip(tmp_path):
    """The reference's checkpoint layout loads through the weights-only unpickler (argparse.Namespace allowed)."""
    import argparse
    import torch
    import pifuhd_amd  # noqa: F401
    from pifuhd_amd import reconstruction
    p = str(tmp_path / 'ck')
    torch.save({'opt': argparse.Namespace(name='x', mlp_dim=[272, 512, 1]), 'opt_netG': argparse.Namespace(loadSize=512),
                'model_state_dict': {'mlp.filters.0.weight': torch.arange(6.).view(1, 6, 1)}}, p)
    ck = reconstruction.load_checkpoint(p)
    assert ck['opt'].name == 'x' and ck['opt_netG'].loadSize == 512
    assert torch.equal(ck['model_state_dict']['mlp.filters.0.weight'], torch.arange(6.).view(1, 6, 1))
