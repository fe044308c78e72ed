"""Which kind of decision sends pixels of rt_primary_hits (f32) to the f64 kernel?  RT_AUDIT_REASONS=1 prints the histogram."""
import os, sys
os.environ["RT_AUDIT_REASONS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import raytracer_amd as R
from conftest import SCENES
W = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for name, (sid, asp, n, depth) in SCENES.items():
    ds = R.DeviceScene(R.HostScene(sid, asp, n=n), 0)
    print(name, flush=True)
    ds.primary_hits(W, R.image_height(W, asp), seed=3, time=0.25)
