"""bench_line.py — what bench.py prints and where the rest goes.

The driver reads the LAST stdout line of bench.py as one JSON object; a 20 KB line (round 4) was not parsed. So:
  * the last line is the compact headline: numbers and short strings only, no prose, < COMPACT_LIMIT bytes (asserted);
  * every secondary entry is printed BEFORE it as its own short line ({"secondary": key, ...}, notes stripped);
  * the full objects, with their notes, go to bench_secondary.json beside bench.py.
What the fields mean is written down in DESIGN.md §6, not in the line.
"""
from __future__ import annotations

import json
import os

COMPACT_LIMIT = 4096
SECONDARY_LINE_LIMIT = 1024
SECONDARY_FILE = "bench_secondary.json"

# Keys of the compact line, in print order (the contract's first, then the two required objects, then the extras)
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms_avg", "kernel_ms_min",
                 "hbm_frac_measured", "frac_overlapped", "span_ms_per_step", "algorithmic_bytes_per_alignment",
                 "alignments_per_launch", "launches_in_flight")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "host_cpu", "host_threads_usable")
CONFIG_KEYS = ("workload", "pairs_per_gpu", "patches", "levels", "max_iters", "launch_streams", "parallelism", "barrier_backend")


def sig(x, digits=6):
    """Floats to `digits` significant digits (the line carries measurements, not 17-digit doubles); everything else as is."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{digits}g}")
    if isinstance(x, (list, tuple)):
        return [sig(v, digits) for v in x]
    if isinstance(x, dict):
        return {k: sig(v, digits) for k, v in x.items()}
    return x


def strip_notes(obj, max_str=96):
    """A copy of `obj` without prose: keys named `note` / `*_note` dropped, strings cut to `max_str` characters."""
    if isinstance(obj, dict):
        return {k: strip_notes(v, max_str) for k, v in obj.items() if not (k == "note" or k.endswith("_note"))}
    if isinstance(obj, (list, tuple)):
        return [strip_notes(v, max_str) for v in obj]
    if isinstance(obj, str) and len(obj) > max_str:
        return obj[:max_str - 1] + "…"
    return obj


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def secondary_line(entry):
    """One short stdout line for a secondary entry: its key, value, unit and the numbers of its roofline block."""
    e = strip_notes(entry, 64)
    out = {"secondary": e.get("key") or e.get("workload", "")[:48]}
    for k in ("value", "unit", "value_four_streams", "us_per_frame", "run_wall_ms", "run_device_ms", "run_wall_ms_cpp", "new_frame_wall_ms",
              "frame_wall_ms", "frame_wall_ms_pinned_image", "four_call_wall_ms", "frame_device_ms", "frame_cpu_oracle_ms", "cpu_oracle_ms", "h2d_achieved_GBps", "h2d_fraction_of_ceiling",
              "cpu_all_cores_alignments_per_s", "iterations"):
        if k in e:
            out[k] = e[k]
    r = e.get("roofline")
    if isinstance(r, dict):
        out["roofline"] = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms_avg", "hbm_frac_measured"))
    p = e.get("pose_delta_vs_cpu") or e.get("parity")
    if isinstance(p, dict):
        out["pose_delta_vs_cpu"] = {k: v for k, v in p.items() if not isinstance(v, (dict, list, str))}
    line = json.dumps(sig(out), separators=(",", ":"))
    if len(line) >= SECONDARY_LINE_LIMIT:                       # never let an entry grow back into a paragraph
        out.pop("pose_delta_vs_cpu", None)
        line = json.dumps(sig(out), separators=(",", ":"))
    assert len(line) < SECONDARY_LINE_LIMIT, len(line)
    return line


def compact(full):
    """The headline object of the last stdout line, built from bench.py's full result dict."""
    out = {k: full[k] for k in CONTRACT_KEYS if k in full}
    if isinstance(out.get("config"), dict):
        out["config"] = _pick(strip_notes(out["config"], 160), CONFIG_KEYS)
    if "roofline" in full:
        out["roofline"] = _pick(full["roofline"], ROOFLINE_KEYS)
    for k in ("cpu_baseline", "cpu_baseline_all_cores", "cpu_baseline_native"):
        if isinstance(full.get(k), dict):
            keys = CPU_KEYS if k == "cpu_baseline" else ("value", "unit", "cores", "kind")
            out[k] = _pick(strip_notes(full[k], 120), keys)
    if isinstance(full.get("pose_delta_vs_cpu"), dict):
        out["pose_delta_vs_cpu"] = _pick(full["pose_delta_vs_cpu"], ("max_rad", "max_m", "pairs_checked", "n_tracked_equal", "iterations_equal", "ranks_checked"))
    if isinstance(full.get("fp64"), dict):
        out["fp64"] = _pick(full["fp64"], ("bound", "achieved", "peak", "unit", "frac"))
    for k in ("parity_failed", "value_from_idle", "executed_iterations_total_mean", "n_tracked_mean", "library", "library_sha",
              "per_rank_ms_per_step", "per_rank_value", "ranks_seen", "barrier_backend", "pairs_per_step_all_ranks", "elapsed_max_s"):
        if k in full:
            out[k] = full[k]
    if isinstance(full.get("preroll"), dict):
        out["preroll"] = _pick(full["preroll"], ("launches", "ms"))
    if isinstance(full.get("err_vs_ground_truth_median"), dict):
        out["err_vs_ground_truth_median"] = full["err_vs_ground_truth_median"]
    sec = full.get("secondary")
    if sec:
        out["secondary"] = {"file": SECONDARY_FILE, "entries": len(sec),
                            "values": {(e.get("key") or e.get("workload", "")[:24]): sig(e.get("value"), 4) for e in sec}}
    if "secondary_error" in full:
        out["secondary_error"] = str(full["secondary_error"])[:160]
    return sig(out)


def compact_line(full):
    """The last stdout line. Sheds optional parts before it would ever exceed the limit; the contract's keys, `roofline` and
    `cpu_baseline` are never shed — if those alone do not fit, that is a bug and the assertion says so."""
    out = compact(full)
    line = json.dumps(out, separators=(",", ":"))
    for k in ("secondary", "err_vs_ground_truth_median", "fp64", "cpu_baseline_native", "per_rank_value"):
        if len(line) < COMPACT_LIMIT:
            break
        if k == "secondary" and isinstance(out.get(k), dict):
            out[k] = {"file": SECONDARY_FILE, "entries": out[k].get("entries")}
        else:
            out.pop(k, None)
        line = json.dumps(out, separators=(",", ":"))
    assert len(line) < COMPACT_LIMIT, f"bench line is {len(line)} bytes (limit {COMPACT_LIMIT})"
    assert "\n" not in line
    return line


def emit(full, root, stream=None):
    """Prints the secondary lines, writes bench_secondary.json, then prints the compact headline as the LAST line."""
    import sys
    stream = stream or sys.stdout
    sec = full.get("secondary") or []
    for e in sec:
        print(secondary_line(e), file=stream, flush=True)
    try:
        with open(os.path.join(root, SECONDARY_FILE), "w") as f:
            json.dump({"headline": {k: v for k, v in full.items() if k != "secondary"}, "secondary": sec}, f, indent=1)
    except OSError as e:                                     # a read-only tree: the lines above still carry the numbers
        print(json.dumps({"secondary_file_error": str(e)[:120]}), file=stream, flush=True)
    line = compact_line(full)
    print(line, file=stream, flush=True)
    return line
