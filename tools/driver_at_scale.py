#!/usr/bin/env python3
"""The drop-in binary end to end at BASELINE size: write the 4-level M6-like hierarchy in the reference's file
formats, run euler3d_gpu_double on it (default = fused stages with the per-loop times attributed, --loop-timers = every
loop bracketed as the reference's -DTIME build does, and --no-timers),
print its own 'Total runtime' line and the wall time of the whole process (file parsing and plan building included)."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd"))
from mgcfd import meshgen
exe = os.path.join(ROOT, "mg-cfd-app-plain_amd", "csrc", "euler3d_gpu_double")
d = tempfile.mkdtemp(prefix="mgcfd_scale_")
t0 = time.time()
mg = meshgen.make_multigrid((67, 55, 48, 43), "m6wing", seed=0, jitter=0.2, area_noise=0.02, volume_noise=0.02)
meshgen.write_input(mg, d)
print(f"generated + wrote the input files in {time.time() - t0:.1f} s ({sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d)) / 1e6:.0f} MB)")
if os.environ.get("DRIVER_FLOOR"):      # tools/hip_process_floor.hip, built by the caller
    for mb in ("0", "300"):
        walls = []
        for _ in range(9):
            t1 = time.time(); subprocess.run([os.environ["DRIVER_FLOOR"], mb]); walls.append(time.time() - t1)
        print(f"a process that wakes the device{'' if mb == '0' else ', uploads ' + mb + ' MB, runs one kernel'} and exits: min {min(walls):.3f} median {sorted(walls)[4]:.3f} max {max(walls):.3f} s")
for extra in ([], ["--loop-timers"], ["--no-timers"])[:int(os.environ.get("DRIVER_RUNS", "3"))]:
    t0 = time.time()
    r = subprocess.run([exe, "-i", "input.dat", "-d", d, "-o", d + "/", "-g", "25"] + extra, capture_output=True, text=True)
    wall = time.time() - t0
    lines = [l for l in r.stdout.splitlines() if "Total runtime" in l or "RMS" in l]
    print(" ".join(["euler3d_gpu_double -g 25"] + extra), "-> rc", r.returncode, f"process wall {wall:.2f} s;", lines[-1] if lines else r.stdout[-200:], "|", (lines[-2] if len(lines) > 1 else ""))
    for ln in r.stderr.splitlines():
        if "seconds of the epoch" in ln:
            import re
            a, b = map(float, re.findall(r"(\d+\.\d+)", ln)[:2])
            print(f"    process started -> main() {a - t0:.3f} s, main() {b - a:.3f} s, main() returned -> process gone {t0 + wall - b:.3f} s")
        if "input files read" in ln or "mgcfd plan" in ln or "mgcfd create" in ln or "since main()" in ln: print("   ", ln.strip())
    if os.environ.get("DRIVER_REPEAT") and not extra:      # the same command again and again: how the process wall scatters
        for env_extra in [e.split("=") for e in os.environ.get("DRIVER_ENVS", "X=0").split(",")]:
            walls = []
            for _ in range(int(os.environ["DRIVER_REPEAT"])):
                t1 = time.time()
                subprocess.run([exe, "-i", "input.dat", "-d", d, "-o", d + "/", "-g", "25"], capture_output=True, text=True, env=dict(os.environ, **{env_extra[0]: env_extra[1]}))
                walls.append(time.time() - t1)
            print(f"    {env_extra[0]}={env_extra[1]}: process wall of {len(walls)} runs: min {min(walls):.3f} median {sorted(walls)[len(walls) // 2]:.3f} max {max(walls):.3f} s")
    if r.returncode != 0:
        print(r.stdout[-500:], r.stderr[-500:])
    elif extra != ["--no-timers"]:
        rows = [l.rstrip(",\n").split(",") for l in open(os.path.join(d, "Times.csv")) if l.strip()]
        t = dict(zip(rows[0], rows[1]))
        print("   Times.csv:", " ".join(f"{k}={float(v) * 1e3:.3f}ms" for k, v in t.items() if k[:-1] in ("flux", "compute_step", "time_step", "restrict", "prolong", "indirect_rw") and float(v) > 0),
              f"| Total={float(t['Total']) * 1e3:.3f}ms = {float(t['Total']) / 25 * 1e3:.4f} ms per cycle | Flux options: {t['Flux options']!r}")
