#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (tools/profile_cmd.sh / tools/profile_bench.sh) into the committed summaries
profiles/<name>_kernel_stats.csv and profiles/<name>_pmc.json.

    python tools/summarise_profile.py <tag> <name>

HBM traffic per launch follows MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB of 64-byte fabric
requests; on gfx950 FETCH_SIZE reports half of the bytes of a coalesced streaming read, so reads = 2 x FETCH_SIZE;
WRITE_SIZE is exact.  The dominant kernel = the one with the largest total duration in the kernel trace; only its
launches are averaged (for the tick kernel every launch of the profiled commands advances the same number of ticks)."""
import csv, glob, json, os, shutil, sys
tag, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)  # (gpurun merges runs into one directory: take the last one)
shutil.copy(newest(os.path.join(src, "trace", "*", "*kernel_stats.csv")), os.path.join(dst, f"{name}_kernel_stats.csv"))
cmd_file = os.path.join(src, "command.txt")
out = {"command": ("python3 " + open(cmd_file).read().strip().replace(root + "/", "")) if os.path.exists(cmd_file) else
       "python3 bench.py --steps N --warmup W --no-cpu-baseline --no-ref-mode",
       "how": "rocprofv3, three separate runs of the command: --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE"}
stats = list(csv.DictReader(open(os.path.join(dst, f"{name}_kernel_stats.csv"))))
k = max(stats, key=lambda r: float(r["TotalDurationNs"]))
out["kernel"] = k["Name"]
try:  # the library the profile was taken with (bench.py quotes counters only next to a timing of the same build)
    import hashlib
    sys.path.insert(0, root)
    from cgmres_cpp_amd import build as _b
    out["library_sha256_16"] = hashlib.sha256(open(os.environ.get("CGMRES_HIP_LIB") or _b.LIB_PATH, "rb").read()).hexdigest()[:16]
except OSError:
    out["library_sha256_16"] = None
out["kernel_calls"] = int(k["Calls"])
out["kernel_avg_ns"] = float(k["AverageNs"])
for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = newest(os.path.join(src, f"pmc_{kind}", "*", "*counter_collection.csv"))
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"] == k["Name"] and r["Counter_Name"] == ctr]
    vals = [float(r["Counter_Value"]) for r in rows]
    out[f"{ctr}_KiB_per_launch"] = sum(vals) / len(vals)
    out[f"{ctr}_launches"] = len(vals)
out["read_bytes_per_launch"] = 2.0 * out["FETCH_SIZE_KiB_per_launch"] * 1024
out["write_bytes_per_launch"] = out["WRITE_SIZE_KiB_per_launch"] * 1024
out["hbm_bytes_per_launch"] = out["read_bytes_per_launch"] + out["write_bytes_per_launch"]
out["measured_GBps"] = out["hbm_bytes_per_launch"] / out["kernel_avg_ns"]
out["note"] = "reads = 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md §HBM)"
lines = []
for fn in ("out_trace.json", "bench_trace.json"):
    pth = os.path.join(src, fn)
    if os.path.exists(pth):
        lines = [json.loads(l) for l in open(pth, errors="ignore") if l.startswith("{")]
        break
if lines and "roofline" in lines[-1]:
    b = lines[-1]
    out["ticks_per_launch"] = b["roofline"].get("ticks_per_launch", 1)
    out["bench_line_under_profiler"] = {k2: b[k2] for k2 in ("value", "ms_per_step", "roofline")}
elif lines:
    out["ticks_per_launch"] = 10
    out["lines_under_profiler"] = lines
    by = lines[-1].get("algorithmic_GBps")
    if by:
        out["algorithmic_bytes_per_launch"] = by * 1e9 * lines[-1]["ms_per_tick"] * 1e-3 * out["ticks_per_launch"]
        out["algorithmic_frac_of_8TBps_from_kernel_avg"] = out["algorithmic_bytes_per_launch"] / out["kernel_avg_ns"] / 8000.0
json.dump(out, open(os.path.join(dst, f"{name}_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
