"""One line per library variant for tools/ab_lib.sh: the headline shape through bench.py (no CPU leg, no secondary entries),
delivered rate on 8 streams, the kernel alone, parity of the first pairs.   tools/ab_lib.sh "python tools/ab_headline.py" a.so b.so"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
label = sys.argv[-1] if len(sys.argv) > 1 else "lib"
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "200", "--warmup", "20", "--no-cpu", "--no-secondary"],
                   capture_output=True, text=True, timeout=280)
line = [l for l in r.stdout.splitlines() if l.startswith("{")]
if r.returncode or not line:
    print(label, "FAILED", r.stderr[-800:])
    sys.exit(1)
d = json.loads(line[-1])
rf = d["roofline"]
print(f"{label:16s} value {d['value'] / 1e6:6.3f} M/s  ms_per_step {d['ms_per_step']:.4f}  kernel alone {rf['kernel_ms_avg']:.4f} ms  frac {rf['frac']:.3f}"
      f"  pose delta vs cpu {d.get('pose_delta_vs_cpu')}", flush=True)
