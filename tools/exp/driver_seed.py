"""Regenerate the input of a tools/fuzz_driver.py seed into a directory and run the drop-in on it with the given extra flags.
    python tools/exp/driver_seed.py SEED OUTDIR [flags ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("mg-cfd-app-plain_amd", "", "tests", "oracle", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import fuzz_parity
from mgcfd import meshgen
seed, d = int(sys.argv[1]), sys.argv[2]
rng = np.random.default_rng(9000 + seed)
while True:
    kind, name, mg, _ = fuzz_parity.make_case(rng)
    if max(l.nel for l in mg.levels) <= 3000 and all(mg.levels[k + 1].nel <= mg.levels[k].nel for k in range(len(mg.levels) - 1)):
        break
os.makedirs(d, exist_ok=True)
meshgen.write_input(mg, d)
cycles = int(rng.integers(1, 5)); dup = int(rng.choice([1, 1, 2, 3]))
print(kind, name, [l.nel for l in mg.levels], "cycles", cycles, "dup", dup, flush=True)
exe = os.path.join(ROOT, "mg-cfd-app-plain_amd", "csrc", "euler3d_gpu_double")
r = subprocess.run([exe, "-i", "input.dat", "-d", d, "-o", d + "/", "-g", str(cycles), "-m", str(dup), "--output-variables"] + sys.argv[3:], capture_output=True, text=True)
print("rc", r.returncode, "|", r.stdout[-300:].replace("\n", " / "), "|", r.stderr[-200:].replace("\n", " / "))
