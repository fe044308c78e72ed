"""render_queue_kernel (jobs not tied to lanes, camera rays generated a wave at a time; the f64 default on leaf-list scenes,
opt-in for f32) against render_persist_kernel (lane = pixel): the same (pixel, sample) Philox streams, hence the same paths
and the same work counters; only the order of a pixel's sum differs.  RT_QUEUE is read once per process, so each form
renders in a child process.  (Both forms against the ORACLE: the shared parity tests run on whichever is the default.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = (
    "import sys, json, numpy as np; sys.path.insert(0, %r); import raytracer_amd as R\n"
    "sid, f64, w, h, spp, off = int(sys.argv[2]), sys.argv[3] == '1', int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])\n"
    "hs = R.HostScene(sid, w / h, scene_seed=3, n=9) if sid == 9 else R.HostScene(sid, w / h)\n"
    "ds = R.DeviceScene(hs, 0)\n"
    "a, st = ds.render(w, h, spp, 50, seed=11, sample_offset=off, stats=True, f64=f64)\n"
    "np.save(sys.argv[1], a); print(json.dumps({k: int(v) for k, v in st.items() if isinstance(v, (int, np.integer))}))\n" % ROOT)


def _render(queue, sid, f64, w, h, spp, off):
    path = f"/tmp/rt_queue_{queue}_{sid}_{int(f64)}_{os.getpid()}.npy"
    env = dict(os.environ, RT_QUEUE=str(queue))
    r = subprocess.run([sys.executable, "-c", CODE, path, str(sid), "1" if f64 else "0", str(w), str(h), str(spp), str(off)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    img = np.load(path)
    os.remove(path)
    import json
    return img, json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("f64", [True, False])
# Cornell (pinhole), ragged image, odd sample count and offset; the generated scene 9 (a lens: the queue carries origins)
@pytest.mark.parametrize("sid,w,h,spp", [(6, 75, 53, 37), (9, 64, 48, 24)])
def test_queue_form_renders_the_paths_of_the_persistent_form(sid, w, h, spp, f64):
    a, sa = _render(0, sid, f64, w, h, spp, 5)
    b, sb = _render(1, sid, f64, w, h, spp, 5)
    assert sa == sb, (sa, sb)  # every work counter: the same rays, tests, vertices
    assert np.array_equal(np.isnan(a), np.isnan(b))
    ok = np.isfinite(a)
    tol = 1e-12 if f64 else 2e-5
    assert np.allclose(a[ok], b[ok], rtol=tol, atol=tol * max(1.0, float(np.abs(a[ok]).max())))
