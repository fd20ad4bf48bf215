"""OBJ writer alone on the host (no GPU): the one-call writer `pifu_write_obj` on a synthetic mesh of gen_mesh's 128^3 size
(221 694 vertices, 410 692 faces) and of the smoother 512^3 subject's, to /dev/null (formatting only) and to a file, for a
range of writer threads (lab build: PIFU_OBJ_THREADS).  Prints ms per call and the host's view of its CPUs."""
import os
import subprocess
import sys
import time
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, os.path.join(HERE, 'lab'))
    import lab
    lab.use()
    import tempfile
    import numpy as np
    from pifuhd_amd import mesh_util
    rng = np.random.default_rng(0)
    for V, F in ((221694, 410692), (1862352, 3720250)):
        verts = rng.uniform(-1, 1, (V, 3)); faces = rng.integers(0, V, (F, 3)).astype(np.int32); col = rng.uniform(0, 1, (V, 3))
        with tempfile.TemporaryDirectory() as td:
            for target in ('/dev/null', os.path.join(td, 'm.obj')):
                ts = []
                for k in range(4):
                    t0 = time.perf_counter()
                    mesh_util.save_obj_mesh_with_color(target if target == '/dev/null' else target + str(k), verts, faces, col)
                    ts.append((time.perf_counter() - t0) * 1e3)
                print(f"  threads {os.environ.get('PIFU_OBJ_THREADS', 'default'):>7s}  V {V:8d}  {'null' if target == '/dev/null' else 'file'}  "
                      + ' '.join(f'{t:7.1f}' for t in ts) + ' ms', flush=True)
else:
    print('os.cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)),
          'cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'n/a', flush=True)
    for nt in ('1', '2', '4', '8', '16', '32'):
        subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=dict(os.environ, PIFU_OBJ_THREADS=nt))
