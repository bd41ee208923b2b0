import json, os, subprocess, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo/tools") else os.getcwd()
label = sys.argv[-1]
for patches in (120, 190, 256, 300):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "200", "--warmup", "20", "--no-cpu", "--no-secondary", "--patches", str(patches)],
                       capture_output=True, text=True, timeout=280)
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print(f"{label:12s} patches {patches}: value {d['value'] / 1e6:6.3f} M/s  kernel alone {d['roofline']['kernel_ms_avg']:.4f} ms", flush=True)
