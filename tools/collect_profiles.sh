#!/usr/bin/env bash
# Per-round evidence, collected on the GPU box into gpurun_out/<tag>/ (copy what is to be judged into profiles/):
#   kernel-trace summaries of the default bench command (without its two-in-flight leg: two solvers' launches overlapped on purpose would
#   raise the per-kernel means) and of the V-cycle bench, the un-profiled bench line,
#   FETCH_SIZE / WRITE_SIZE passes over the bench command (separate --pmc passes; no trace domain beside --kernel-trace).
# Usage (GPU box): bash tools/collect_profiles.sh <tag>
set -u
tag=${1:-r4}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp && cd "$OLDPWD"
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_bench -- python3 bench.py --no-vcycle --cpu-seconds 0 --no-two-in-flight > $out/bench_default_under_rocprof.json 2> $out/kt_bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_vcycle -- python3 tools/vcycle_bench.py --cycles 50 --repeats 2 > $out/vcycle_under_rocprof.json 2> $out/kt_vcycle.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 40 --warmup 5 --no-vcycle --cpu-seconds 0 --no-two-in-flight > $out/pmc_$c.log 2>&1
done
# the fast mode (MGCFD_OPT_EXACT = 0: order-free stages): its line, its kernel summary, its traffic
python3 bench.py --fast --steps 2000 --warmup 200 --cpu-seconds 0 > $out/bench_fast.json 2> $out/bench_fast.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_bench_fast -- python3 bench.py --fast --no-vcycle --cpu-seconds 0 --no-two-in-flight > $out/bench_fast_under_rocprof.json 2> $out/kt_bench_fast.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmcfast_$c -- python3 bench.py --fast --steps 40 --warmup 5 --no-vcycle --cpu-seconds 0 --no-two-in-flight > $out/pmcfast_$c.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
def stats(d, dst):
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        open(dst, "w").write(open(f).read())
stats(os.path.join(out, "kt_bench"), os.path.join(out, "bench_default_kernel_stats.csv"))
stats(os.path.join(out, "kt_vcycle"), os.path.join(out, "vcycle_kernel_stats.csv"))
stats(os.path.join(out, "kt_bench_fast"), os.path.join(out, "bench_fast_kernel_stats.csv"))
agg = collections.defaultdict(list)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for sub in ("pmc_", "pmcfast_"):
        for f in glob.glob(os.path.join(out, sub + c, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
lines = [f"{k:70s} {c:12s} launches={len(v):4d} mean={sum(v)/len(v):.6g}" for (k, c), v in sorted(agg.items())]
open(os.path.join(out, "pmc_bench_traffic.txt"), "w").write("\n".join(lines) + "\n")
def mean(sub, counter):
    vals = [sum(v) / len(v) for (k, c), v in agg.items() if c == counter and sub(k)]
    return sum(vals) / len(vals) if vals else None
def pack(sub):
    f, w = mean(sub, "FETCH_SIZE"), mean(sub, "WRITE_SIZE")
    return None if f is None or w is None else {"fetch_kb": round(f), "write_kb": round(w), "bytes": int(2 * f * 1024 + w * 1024)}
traffic = {"_comment": "HBM-side bytes per launch from rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes over `python3 bench.py --steps 40 --warmup 5 --no-vcycle --cpu-seconds 0`), corrected as MI355X_MICROARCH.md prescribes for gfx950: bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024.  fused_stage = mean over the three role-specialised instantiations of the stage kernel.",
           "build": os.environ.get("MGCFD_BUILD_TAG", "unknown"),
           "flux_only": pack(lambda k: "exact::k_flux_tile<" in k and ", false, false," in k),
           "fused_stage": pack(lambda k: "exact::k_flux_tile<" in k and ", true, false," in k),
           "indirect_rw_tile": pack(lambda k: "k_indirect_rw_tile" in k),
           "flux_order_free": pack(lambda k: "k_flux_free<false, false" in k),
           "fused_stage_order_free": pack(lambda k: "k_flux_free<true, false" in k)}
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=2)
print(open(os.path.join(out, "pmc_bench_traffic.txt")).read())
PY
tail -c 600 $out/bench_default.json
