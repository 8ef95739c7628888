#!/usr/bin/env python3
"""Ranks in different PROCESSES storing their halo messages straight into each other's memory through HIP IPC
(include/mgcfd.h: mgcfd_rank_ipc_*) — rehearsed on ONE GPU: this script starts N processes that all use device 0, each builds
its part of a level (local time step, or a global one whose all-reduce then goes through the same flags: no RCCL, which
refuses two ranks on one device), they hand each other their IPC exports through files, sweep, and every rank compares its
owned nodes and ghosts with the unpartitioned level it computes for itself, bit for bit.
    python tools/ipc_ranks_check.py [--ranks 2] [--lattice 14] [--sweeps 5]
(rank processes are started with --rank R --dir D)"""
import argparse, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def wait_for(path, timeout=120.0):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise TimeoutError(path)
        time.sleep(0.01)
    return open(path, "rb").read()


def publish(path, data):
    with open(path + ".tmp", "wb") as f:
        f.write(data)
    os.rename(path + ".tmp", path)


def rank_main(a):
    import numpy as np
    import mgcfd
    from mgcfd import meshgen
    from mgcfd.partition import partition_level, rcb_partition
    from conftest import perturbed_state
    if a.mesh == "fvcorr":                                  # local time step: no all-reduce in a sweep
        mg = meshgen.make_multigrid((a.lattice,), "fvcorr", seed=4, cavity_radius=0.01, volume_noise=0.02)
    elif a.mesh == "tet":                                   # unstructured (long rows), global time step; --lattice = nodes / 100
        mg = meshgen.make_tet_multigrid((a.lattice * 100,), "m6wing", seed=6)
    else:                                                   # global time step: every rank's minimum to every rank, through the flags
        mg = meshgen.make_multigrid((a.lattice,), "m6wing", seed=4, cavity_radius=0.15, jitter=0.2, area_noise=0.05, volume_noise=0.05)
    L = mgcfd.generated_to_levels(mg)[0]
    parts = partition_level(L, rcb_partition(np.asarray(L["coords"]), a.ranks))
    P = parts[a.rank]
    whole = mgcfd.Solver.from_arrays([L], mg.mesh_variant)
    q0 = perturbed_state(L["nel"], whole.far_field()[:5], seed=21)
    whole.set(0, "variables", q0)
    whole.smooth(0, a.sweeps)
    want = whole.get(0, "variables")
    whole.close()
    s = mgcfd.Solver.from_arrays([P.level], mg.mesh_variant, n_owned=[P.n_owned])
    s.set(0, "variables", q0[P.global_ids])
    s.rank_attach_plain(a.rank, a.ranks)
    if a.unsplit:
        s.set_option("rank_split", 0)                       # every tile of a stage in one launch, the message behind it
    if a.fused:
        s.set_option("rank_split", 2)                       # one launch per stage that sends its own message (boundary tiles first)
    s.rank_set_halo(0, P)
    publish(os.path.join(a.dir, f"export.{a.rank}"), s.rank_ipc_export(0))
    peers = sorted(set(P.send) | set(P.recv))
    s.rank_ipc_attach(0, [wait_for(os.path.join(a.dir, f"export.{p}")) for p in range(a.ranks) if p != a.rank])
    # nobody pushes before everybody has opened everybody's buffers
    publish(os.path.join(a.dir, f"attached.{a.rank}"), b"1")
    for p in range(a.ranks):
        wait_for(os.path.join(a.dir, f"attached.{p}"))
    s.rank_exchange(0)
    if a.one_by_one:
        for _ in range(a.sweeps):                           # (as bench.py issues them: a call per sweep)
            s.rank_sweeps(0, 1)
    else:
        s.rank_sweeps(0, a.sweeps)
    late = s.rank_ipc_status(0)                             # (first: the library refuses the state while timeouts are unacknowledged)
    got = s.get(0, "variables")
    own, gh = P.global_ids[:P.n_owned], P.global_ids[P.n_owned:]
    ok_own = bool(np.array_equal(got[:P.n_owned].view(np.int64), want[own].view(np.int64)))
    ok_gh = bool(np.array_equal(got[P.n_owned:].view(np.int64), want[gh].view(np.int64)))
    info = s.rank_halo_info(0)
    print(f"rank {a.rank}: {P.n_owned} owned + {len(gh)} ghost nodes, peers {peers}, {info['nodes_sent']} nodes per message, owned {'equal' if ok_own else 'DIFFER'}, "
          f"ghosts {'equal' if ok_gh else 'DIFFER'}, waits that gave up: {late}", flush=True)
    if a.time:
        # what a sweep costs a rank: host time to issue it, and until it is done (the ranks share ONE GPU here)
        s.rank_sweeps(0, 30); s.synchronize()
        publish(os.path.join(a.dir, f"warm.{a.rank}"), b"1")
        for p in range(a.ranks):
            wait_for(os.path.join(a.dir, f"warm.{p}"))
        t0 = time.perf_counter(); s.rank_sweeps(0, a.time); t1 = time.perf_counter(); s.synchronize(); t2 = time.perf_counter()
        print(f"rank {a.rank}: {a.time} sweeps: host {1e6 * (t1 - t0) / a.time:.1f} us per sweep to issue, {1e6 * (t2 - t0) / a.time:.1f} us until done; "
              f"waits that gave up: {s.rank_ipc_status(0)}", flush=True)
    # every rank keeps its buffers mapped until all are done
    publish(os.path.join(a.dir, f"done.{a.rank}"), b"1")
    for p in range(a.ranks):
        wait_for(os.path.join(a.dir, f"done.{p}"))
    s.close()
    return 0 if (ok_own and ok_gh and late == 0) else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--lattice", type=int, default=14)
    ap.add_argument("--sweeps", type=int, default=5)
    ap.add_argument("--mesh", default="fvcorr", choices=["fvcorr", "m6wing", "tet"], help="local (fvcorr) or global (m6wing) time step")
    ap.add_argument("--time", type=int, default=0, help="also time that many sweeps per rank")
    ap.add_argument("--one-by-one", action="store_true", help="one mgcfd_rank_sweeps call per sweep")
    ap.add_argument("--unsplit", action="store_true", help="MGCFD_OPT_RANK_SPLIT = 0")
    ap.add_argument("--fused", action="store_true", help="MGCFD_OPT_RANK_SPLIT = 2")
    ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--dir", default="")
    a = ap.parse_args()
    if a.rank >= 0:
        sys.exit(rank_main(a))
    with tempfile.TemporaryDirectory(prefix="mgcfd_ipc_") as d:
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--ranks", str(a.ranks), "--lattice", str(a.lattice), "--sweeps", str(a.sweeps), "--mesh", a.mesh,
                                   "--time", str(a.time), "--rank", str(r), "--dir", d] + (["--one-by-one"] if a.one_by_one else []) + (["--unsplit"] if a.unsplit else []) + (["--fused"] if a.fused else [])) for r in range(a.ranks)]
        rcs = [p.wait(timeout=600) for p in procs]
    print("ranks returned", rcs)
    sys.exit(0 if all(rc == 0 for rc in rcs) else 1)


if __name__ == "__main__":
    main()
