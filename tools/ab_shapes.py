"""One line per kernel shape for tools/ab_lib.sh: 1024 pairs of 120 / 190 / 256 / 300 patches through bench.py (delivered rate on
8 streams, the kernel alone).   tools/ab_lib.sh "python tools/ab_shapes.py [patches ...]" a.so b.so"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
label = sys.argv[-1]
shapes = [int(v) for v in sys.argv[1:-1]] or [120, 190, 256, 300]
for patches in shapes:
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "200", "--warmup", "20", "--no-cpu", "--no-secondary", "--patches", str(patches)],
                       capture_output=True, text=True, timeout=280)
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print(f"{label:12s} patches {patches}: value {d['value'] / 1e6:6.3f} M/s  kernel alone {d['roofline']['kernel_ms_avg']:.4f} ms", flush=True)
