"""Wall time of Sprase_ImgAlign::Run through the C++ host layer (dsdtm_amd/host/dsdtm_host.hpp -> C ABI) on resident frames,
configs 2 / 3 / 5: what a C++ caller waits for per Run, without the Python mirror's marshalling.
    python tools/cpp_run_latency.py
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dsdtm_amd import synth  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests.test_host_cpp import build_example, dump_scene  # noqa: E402

exe = build_example("example_align")
for name, kw in (("config 2: 640x480, 300 patches", dict()),
                 ("config 3: 640x480, 1000 patches", dict(n_patches=1000, cam=synth.Camera.tum(640, 480, synth.TUM_FR3))),
                 ("config 5: 1280x960, 2000 patches", dict(width=1280, height=960, n_patches=2000, margin=60))):
    sc = synth.make_scene(**kw)
    pb, p = H.make_border_patches(sc.cur_pyr[0], [(150.3, 101.6)])
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "scene.bin")
        dump_scene(path, sc, (4, 0, 10), 15, pb[0], p[0], (151.0, 101.0))
        out = subprocess.run([exe, path], capture_output=True, text=True, check=True).stdout.split("\n")
    line = [l for l in out if l.startswith("run_resident_ms")][0].split()
    print(f"{name}: Run through the C++ layer, resident frames: median {float(line[1]):.4f} ms, min {float(line[3]):.4f} ms")
