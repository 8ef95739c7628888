"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol
include/mgcfd.h declares, its file readers agree with the reference-generated golden data,
error paths report instead of exiting, and nothing computes without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = ["m6_2lvl", "m6_3lvl", "fvcorr_1lvl", "tet_2lvl", "mixed_2lvl"]


@pytest.fixture(scope="module")
def mgcfd_mod():
    import mgcfd
    if not os.path.exists(mgcfd.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return mgcfd


def test_library_exports_every_declared_symbol(mgcfd_mod):
    header = open(os.path.join(ROOT, "include", "mgcfd.h")).read()
    declared = sorted(set(re.findall(r"\b(mgcfd_[a-z_0-9]+)\s*\(", header)))
    assert len(declared) >= 45
    lib = C.CDLL(mgcfd_mod.LIB_PATH)
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, f"libmgcfd_hip.so lacks {missing}"
    # and the Python binding types every one of them
    assert sorted(mgcfd_mod.EXPORTED_SYMBOLS) == declared
    assert mgcfd_mod.load_library().mgcfd_abi_version() == 1


def test_hip_kernels_are_in_the_library(mgcfd_mod):
    """The shared object carries gfx950 code objects for both numeric flavours."""
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", mgcfd_mod.LIB_PATH], capture_output=True, text=True).stdout
    assert ".hip_fatbin" in out
    blob = open(mgcfd_mod.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_flux_tile" in blob


@pytest.mark.parametrize("case", CASES)
def test_reader_matches_reference(mgcfd_mod, case):
    d = os.path.join(GOLDEN, case)
    g = np.load(os.path.join(d, "kernels.npz"))
    mesh = mgcfd_mod.Mesh("input.dat", os.path.join(d, "input"))
    assert mesh.variant == (0 if case.startswith("fvcorr") else 4 if case.startswith("tet") else 2)      # tet_2lvl is a rotor37 input
    for l in range(mesh.num_levels):
        L = mesh.level(l)
        sizes = g[f"L{l}_sizes"].tolist()
        assert [L["nel"], L["n_edges"], L["n_internal"], L["n_boundary"], L["n_wall"], L["internal_start"],
                L["boundary_start"], L["wall_start"]] == sizes
        assert np.array_equal(L["edges"], g[f"L{l}_edges_raw"])
        assert np.array_equal(L["volumes"].view(np.int64), g[f"L{l}_volumes"].view(np.int64))
    mesh.close()


def test_in_memory_edge_builder_equals_file_reader(mgcfd_mod, tmp_path):
    from mgcfd import meshgen
    mg = meshgen.make_multigrid((6, 4), "m6wing", seed=21, cavity_radius=0.2, jitter=0.2, area_noise=0.1)
    meshgen.write_input(mg, str(tmp_path))
    mesh = mgcfd_mod.Mesh("input.dat", str(tmp_path))
    for l, gen in enumerate(mgcfd_mod.generated_to_levels(mg)):
        L = mesh.level(l)
        assert np.array_equal(L["edges"], gen["edges"])
        assert (L["n_internal"], L["n_boundary"], L["n_wall"]) == (gen["n_internal"], gen["n_boundary"], gen["n_wall"])
        if gen["mg_map"] is not None:
            assert np.array_equal(L["mg_map"], gen["mg_map"])
    mesh.close()


def test_duplication_matches_oracle(mgcfd_mod, oracle):
    d = os.path.join(GOLDEN, "m6_2lvl_dup2", "input")
    mesh = mgcfd_mod.Mesh("input.dat", d, 3)
    oc = oracle.OracleCase.from_input_dat(os.path.join(d, "input.dat"), 3)
    assert mesh.size == 3
    for l in range(mesh.num_levels):
        L = mesh.level(l)
        assert np.array_equal(L["edges"], oc.edges(l))
        assert np.array_equal(L["mg_map"], oc.mg_map(l))
        assert np.array_equal(L["coords"].ravel(), oc.array(l, "coords"))
    mesh.close()


def test_reader_errors_are_reported_not_fatal(mgcfd_mod, tmp_path):
    with pytest.raises(mgcfd_mod.MgcfdError) as e:
        mgcfd_mod.Mesh("does_not_exist.dat", str(tmp_path))
    assert e.value.code == 2 and "Could not open" in str(e.value)
    (tmp_path / "bad.dat").write_text("size = 1\nmesh_name = m6wing\n[levels]\n0 = x\n")
    with pytest.raises(mgcfd_mod.MgcfdError) as e:
        mgcfd_mod.Mesh("bad.dat", str(tmp_path))
    assert "number of levels" in str(e.value)
    (tmp_path / "bad2.dat").write_text("size = 1\nnum_levels = 1\nmesh_name = warp_drive\n")
    with pytest.raises(mgcfd_mod.MgcfdError) as e:
        mgcfd_mod.Mesh("bad2.dat", str(tmp_path))
    assert "Unknown mesh_name" in str(e.value)
    (tmp_path / "ok.dat").write_text("size = 1\nnum_levels = 1\nmesh_name = fvcorr\n[levels]\n0 = missing_mesh\n")
    with pytest.raises(mgcfd_mod.MgcfdError) as e:
        mgcfd_mod.Mesh("ok.dat", str(tmp_path))
    assert "could not open data file" in str(e.value)
    # ragged / truncated mesh file
    (tmp_path / "trunc").write_text("3 2\n1.0 1 -2 0.1 0.2\n")
    (tmp_path / "t.dat").write_text("size = 1\nnum_levels = 1\nmesh_name = fvcorr\n[levels]\n0 = trunc\n")
    with pytest.raises(mgcfd_mod.MgcfdError):
        mgcfd_mod.Mesh("t.dat", str(tmp_path))


def test_reader_accepts_what_operator_extraction_accepts(mgcfd_mod, tmp_path):
    """The reference parses the mesh, .coords and map files with operator>>: any whitespace separates tokens, numbers
    may carry '+' and a capital exponent, lines may end in CR LF.  (Checked against the real binary when this test was
    written: such files give its golden dump byte for byte — while an input.dat with CR LF line ends makes it exit
    with 1, as it does this reader.)"""
    import re
    import shutil
    src = os.path.join(GOLDEN, "m6_2lvl", "input")
    want = mgcfd_mod.Mesh("input.dat", src)
    for f in os.listdir(src):
        t = open(os.path.join(src, f)).read()
        if f != "input.dat":
            t = re.sub(r"(?<![\w.+-])(\d+\.\d+e[+-]\d+)", r"+\1", t).replace("e", "E").replace(" ", "\t").replace("\n", "\r\n")
        open(tmp_path / f, "w", newline="").write(t)
    got = mgcfd_mod.Mesh("input.dat", str(tmp_path))
    for l in range(2):
        for k in ("edges", "volumes", "coords"):
            assert np.array_equal(got.level(l)[k], want.level(l)[k]), (l, k)
    assert np.array_equal(got.level(0)["mg_map"], want.level(0)["mg_map"])
    got.close()
    crlf = tmp_path / "crlf"
    shutil.copytree(src, crlf)
    (crlf / "input.dat").write_bytes(open(os.path.join(src, "input.dat"), "rb").read().replace(b"\n", b"\r\n"))
    with pytest.raises(mgcfd_mod.MgcfdError):
        mgcfd_mod.Mesh("input.dat", str(crlf))
    want.close()


def test_reader_rounds_every_spelling_of_a_number_as_strtod_does(mgcfd_mod, tmp_path):
    """The readers convert numbers with std::from_chars (round 4: the 75 MB level file is most of the drop-in's start-up) and
    fall back to strtod for what from_chars does not take.  Every spelling must give the bits strtod gives — Python's float()
    is the same correctly rounded conversion: long mantissas, halfway cases, subnormals, overflow to infinity, a leading '+',
    capital exponents, hexadecimal floats, a file without a final newline, form feeds and vertical tabs between tokens."""
    import random
    rng = random.Random(5)
    spell = ["1", "-1", "+1.5", "1E5", "1e+5", "-1.25e-3", "0.1", "0.30000000000000004", "2.2250738585072014e-308", "4.9e-324",
             "2.4703282292062328e-324", "1.7976931348623157e308", "1e400", "-1e400", "1e-400", "9007199254740993", "9007199254740992.5",
             "0.000000000000000000000000000000123456789012345678901234567890", "123456789012345678901234567890.123456789",
             "5e-1", ".5", "5.", "0x1.8p1", "-0X1P-3", "1.0000000000000002220446049250313", "1.00000000000000011102230246251565404236316680908203125",
             "1.00000000000000011102230246251565404236316680908203126", "8.5e-1"]
    for _ in range(400):
        spell.append(f"{'-' if rng.random() < 0.5 else ''}{rng.randrange(10 ** rng.randrange(1, 25))}.{rng.randrange(10 ** rng.randrange(1, 25))}e{rng.randrange(-330, 310)}")
    want = [float.fromhex(t) if "x" in t.lower() else float(t) for t in spell]
    n = len(spell)
    # a level of n nodes without edges: the volumes carry the spellings; separators of every kind, no newline at the end
    seps = [" ", "\t", "\n", "\r\n", " \v ", "\f", "  "]
    body = f"{n} 0"
    for t in spell:
        body += rng.choice(seps) + t + rng.choice(seps) + "0"
    (tmp_path / "lvl").write_bytes(body.encode())
    (tmp_path / "in.dat").write_text("size = 1\nnum_levels = 1\nmesh_name = fvcorr\n[levels]\n0 = lvl\n")
    m = mgcfd_mod.Mesh("in.dat", str(tmp_path))
    got = m.level(0)["volumes"]
    assert got.shape == (n,)
    assert np.array_equal(got.view(np.int64), np.asarray(want, dtype=np.float64).view(np.int64)), [(t, g, w) for t, g, w in zip(spell, got, want) if not (g == w or (g != g and w != w))][:5]
    m.close()
    # integers: a leading '+', and a count that runs into the end of the file
    (tmp_path / "lvl2").write_bytes(b"+2 +0 1.0 +0 2.0 0")
    (tmp_path / "in2.dat").write_text("size = 1\nnum_levels = 1\nmesh_name = fvcorr\n[levels]\n0 = lvl2\n")
    m = mgcfd_mod.Mesh("in2.dat", str(tmp_path))
    assert list(m.level(0)["volumes"]) == [1.0, 2.0]
    m.close()


def test_dump_format_and_validation_rule(mgcfd_mod, tmp_path):
    lib = mgcfd_mod.load_library()
    a = np.array([[1.4, 1.68, 0.0, -1e-300, 3.508], [np.pi, -np.e, 1e22, 5e-324, 2.0]])
    p = str(tmp_path / "dump.txt")
    assert lib.mgcfd_write_array(p.encode(), a.ctypes.data_as(C.c_void_p), 2, 5) == 0
    lines = open(p).read().splitlines()
    assert lines[0] == "%.17e %.17e %.17e %.17e %.17e" % tuple(a[0])      # src/Base/io.cpp:223-227
    assert np.array_equal(np.loadtxt(p).view(np.int64), a.view(np.int64))
    master = np.array([[1.0, 1e-12, 0.0, -2.0, 1e-30]])
    test = master + np.array([[0.9e-8, 0.9e-20, 2e-19, -1.9e-8, 2e-19]])
    bad = C.c_int64(0)
    assert lib.mgcfd_identify_differences(test.ctypes.data_as(C.c_void_p), master.ctypes.data_as(C.c_void_p), 1, 2, C.byref(bad)) == 0
    test[0, 2] = 4e-19
    assert lib.mgcfd_identify_differences(test.ctypes.data_as(C.c_void_p), master.ctypes.data_as(C.c_void_p), 1, 2, C.byref(bad)) == 7
    assert bad.value == 2
    assert lib.mgcfd_identify_differences(test.ctypes.data_as(C.c_void_p), master.ctypes.data_as(C.c_void_p), 1, 0, C.byref(bad)) == 0


def test_no_gpu_means_no_compute(mgcfd_mod):
    """Without a HIP device the solver must refuse loudly (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    mesh = mgcfd_mod.Mesh("input.dat", os.path.join(GOLDEN, "m6_2lvl", "input"))
    with pytest.raises(mgcfd_mod.MgcfdError) as e:
        mgcfd_mod.Solver.from_mesh(mesh)
    assert e.value.code == 3 and "no CPU fallback" in str(e.value)


def test_driver_cli_errors():
    exe = os.path.join(ROOT, "mg-cfd-app-plain_amd", "csrc", "euler3d_gpu_double")
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "ERROR: input_file not set" in r.stdout       # src/euler3d_cpu_double.cpp:79-82
    r = subprocess.run([exe, "-h"], capture_output=True, text=True)
    assert r.returncode == 1 and "--input-file" in r.stderr
    r = subprocess.run([exe, "-i", "nope.dat"], capture_output=True, text=True)
    assert r.returncode != 0 and "Could not open input file" in r.stderr


def test_rcb_partition_is_balanced_and_compact():
    """Recursive coordinate bisection (the partitioner for BASELINE config 5's 8-way split): every node assigned,
    part sizes equal to within a node, the same owned/ghost bookkeeping as any partition, and on a box its
    2 x 2 x 2 blocks need far fewer ghost copies than 8 slabs."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "mg-cfd-app-plain_amd"))
    import mgcfd
    from mgcfd import meshgen
    from mgcfd.partition import halo_volume, partition_level, rcb_partition, slab_partition
    mg = meshgen.make_multigrid((16,), "m6wing", seed=2, jitter=0.2, area_noise=0.02, volume_noise=0.02)
    L = mgcfd.generated_to_levels(mg)[0]
    coords = np.asarray(L["coords"])
    for n in (1, 2, 3, 5, 8):
        part = rcb_partition(coords, n)
        counts = np.bincount(part, minlength=n)
        assert part.min() == 0 and part.max() == n - 1 and counts.max() - counts.min() <= 1
    rcb, slab = rcb_partition(coords, 8), slab_partition(coords, 8)
    assert halo_volume(L, rcb) < 0.6 * halo_volume(L, slab)
    parts = partition_level(L, rcb)
    assert sum(p.n_owned for p in parts) == L["nel"]
    owned = np.concatenate([p.global_ids[:p.n_owned] for p in parts])
    assert np.array_equal(np.sort(owned), np.arange(L["nel"]))
    for p in parts:                                   # what a part receives from a peer is what that peer sends to it
        for peer, ids in p.recv.items():
            assert np.array_equal(p.global_ids[ids], parts[peer].global_ids[parts[peer].send[p.rank]])
