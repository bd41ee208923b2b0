"""Wall time of Feature_detector::detect through the C++ host layer (dsdtm_host.hpp) on a resident 640x480 frame with
5 levels: the driver example_align prints the median of 21 calls (library call + the sort / mask / cap bookkeeping of
src/Feature_detection.cpp:110-150). Usage: python tools/cpp_detect.py   (MI355X)"""
import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsdtm_amd import synth
from tests import helpers as H
from tests.test_host_cpp import build_example, dump_scene

exe = build_example()
sc = synth.make_scene(width=640, height=480, levels=5, n_patches=300, seed=7, margin=30)
pb, p = H.make_border_patches(sc.cur_pyr[0], [(150.3, 101.6)])
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "scene.bin")
    dump_scene(path, sc, (5, 0, 8), 15, pb[0], p[0], np.array([151.2, 100.9]))
    out = subprocess.run([exe, path], capture_output=True, text=True, check=True).stdout.strip().split("\n")
print("C++ Feature_detector::detect, 640x480 x 5 levels, resident frame:", out[-1])
