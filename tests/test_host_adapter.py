"""The C++ host side (host/): World / Field / PotentialSolver surface of the reference
with MultiGridSolverHIP behind PotentialSolver::solve()."""
import os
import subprocess

import numpy as np
import pytest

from util import assert_bit_equal, rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOX = os.path.join(ROOT, "host", "box_mg")


BOX_ROWS = os.path.join(ROOT, "host", "box_mg_rows")


@pytest.fixture(scope="module")
def box():
    if not os.path.exists(BOX) or not os.path.exists(BOX_ROWS):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")], stdout=subprocess.DEVNULL)
    return BOX


def test_adapter_compiles_against_the_reference_declarations():
    """The reference declares computeEF() non-virtual (cpp/PotentialSolver.h:63) and allocates Field
    rows one by one (cpp/Field.h:65-70): the adapter must build against both, warnings as errors."""
    host = os.path.join(ROOT, "host")
    for defs in ([], ["-DMG_REFERENCE_COMPUTE_EF"], ["-DMG_REFERENCE_COMPUTE_EF", "-DMG_FIELD_ROW_PER_ALLOCATION"]):
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only"] + defs +
                              [os.path.join(host, "MultiGridSolverHIP.cpp"), os.path.join(host, "box_mg.cpp")])


REFERENCE_CPP = "/root/reference/cpp"


@pytest.mark.skipif(not os.path.exists(os.path.join(REFERENCE_CPP, "World.h")),
                    reason="the reference tree exists in the build container only")
def test_adapter_compiles_against_the_reference_headers(tmp_path):
    """host/MultiGridSolverHIP.{h,cpp} against the reference's OWN cpp/World.h and cpp/Field.h (found
    through -I, nothing of them copied): the adapter uses only what those headers offer.  The
    reference's PotentialSolver.h cannot be included here (its line 6 wants <fftw3.h>, which this
    image lacks), so the base class is declared as cpp/PotentialSolver.h:49-68 declares it:
    constructor from World&, virtual destructor, pure virtual solve(), NON-virtual computeEF(),
    protected World& world."""
    import shutil
    host, inc = tmp_path / "host", tmp_path / "include"
    host.mkdir()
    inc.mkdir()
    for f in ("MultiGridSolverHIP.h", "MultiGridSolverHIP.cpp"):
        shutil.copy(os.path.join(ROOT, "host", f), host / f)
    shutil.copy(os.path.join(ROOT, "include", "mg_hip.h"), inc / "mg_hip.h")
    (host / "PotentialSolver.h").write_text(
        "#ifndef _SOLVER_H\n#define _SOLVER_H\n#include \"World.h\"\n#include \"Field.h\"\n"
        "class PotentialSolver {\npublic:\n    PotentialSolver(World &world) : world(world) {}\n"
        "    virtual ~PotentialSolver() {}\n    virtual bool solve() = 0;\n    void computeEF();\n"
        "protected:\n    World &world;\n};\n#endif\n")
    p = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I", REFERENCE_CPP, "-H",
                        str(host / "MultiGridSolverHIP.cpp")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    used = [l.split()[-1] for l in p.stderr.splitlines() if l.startswith(".")]
    assert os.path.join(REFERENCE_CPP, "World.h") in used and os.path.join(REFERENCE_CPP, "Field.h") in used
    assert not any(u.startswith(os.path.join(ROOT, "host")) for u in used)  # none of this repository's stand-ins


@pytest.mark.gpu
@pytest.mark.parametrize("n,depth", [(41, 2), (33, 0)])
def test_reference_field_layout_gives_the_same_bits(box, gpu, tmp_path, n, depth):
    """box_mg_rows is the driver built the way the reference tree would build it: every Field row its
    own heap block (the adapter packs and unpacks row by row through its page-locked staging block)
    and PotentialSolver::computeEF non-virtual (the call through the base pointer runs the host loop
    on the downloaded phi).  phi and E must equal the contiguous-Field build's bit for bit."""
    outs = {}
    for exe in (BOX, BOX_ROWS):
        tag = os.path.basename(exe)
        phi, ef = tmp_path / f"{tag}.phi", tmp_path / f"{tag}.ef"
        p = subprocess.run([exe, str(n), str(depth), "1e-6", "--dump", str(phi), "--dump-ef", str(ef)],
                           capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        outs[tag] = (np.fromfile(phi), np.fromfile(ef), [l for l in p.stdout.splitlines() if "converged=" in l][0].split()[:2])
    a, b = outs["box_mg"], outs["box_mg_rows"]
    assert a[2] == b[2]
    assert_bit_equal(b[0], a[0], "phi through row-per-allocation Fields")
    assert_bit_equal(b[1], a[1], "E through row-per-allocation Fields")
    assert a[0].size == n ** 3 and np.abs(a[0]).max() > 0


def _tagged(stdout, tags):
    return {l.split(":")[0]: dict(kv.split("=") for kv in l.split(":")[1].split()) for l in stdout.splitlines()
            if l.startswith(tuple(t + ":" for t in tags))}


@pytest.mark.gpu
@pytest.mark.parametrize("exe", [BOX, BOX_ROWS], ids=["contiguous", "rows"])
def test_phi_is_uploaded_on_every_call_by_default(box, gpu, exe):
    """solve() and computeEF() read the current world.phi in full on every call, like the
    reference's host loops: first solve + computeEF, second solve, third solve + computeEF."""
    p = subprocess.run([exe, "41", "2", "1e-6", "--again"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    lines = _tagged(p.stdout, ("again", "reset"))
    # the rows build hides the base's non-virtual computeEF: the call through the base pointer runs
    # the host loop on the downloaded phi and uploads nothing
    second, third = ("3", "5") if exe == BOX else ("2", "3")
    assert lines["again"] == {"converged": "1", "cycles": "1", "phi_uploads": second}
    assert lines["reset"]["converged"] == "1" and lines["reset"]["phi_uploads"] == third
    assert int(lines["reset"]["cycles"]) > 10


@pytest.mark.gpu
@pytest.mark.parametrize("exe", [BOX, BOX_ROWS], ids=["contiguous", "rows"])
def test_assume_untouched_uploads_phi_only_when_it_changed(box, gpu, exe):
    """setAssumePhiUntouched(true): second solve() with world.phi untouched: no upload, converged at
    the first test.  Third after the driver zeroed world.phi: noticed, uploaded, solved again."""
    p = subprocess.run([exe, "41", "2", "1e-6", "--again", "--assume-untouched"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    lines = _tagged(p.stdout, ("again", "reset"))
    assert lines["again"] == {"converged": "1", "cycles": "1", "phi_uploads": "1"}
    assert lines["reset"]["converged"] == "1" and lines["reset"]["phi_uploads"] == "2"
    assert int(lines["reset"]["cycles"]) > 10


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["face", "interior"])
def test_an_edit_of_phi_between_two_solves_reaches_the_device(box, gpu, tmp_path, where):
    """A Dirichlet value written onto one off-centre node of a wall between two solves (an
    electrode patch; cpp/World.cpp:40-52 keeps those values in world.phi), or an edit of a single
    interior node: the second solve must start from the edited potential.  The default policy and
    an explicit invalidate() give the same bits, and for the wall node so does
    setAssumePhiUntouched(true), whose check covers every face node.  With the wall at 5 V the
    residual on that node never vanishes (the reference defines it as phi there), so the second
    solve is bounded by --max-it."""
    n = 41
    runs = {"default": [], "invalidate": ["--assume-untouched", "--invalidate"]}
    if where == "face":
        runs["faces_checked"] = ["--assume-untouched"]
    got = {}
    for tag, extra in runs.items():
        phi = tmp_path / f"{tag}.phi"
        p = subprocess.run([BOX, str(n), "2", "1e-6", "--max-it", "60", "--edit", where, "--dump", str(phi)] + extra,
                           capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        got[tag] = (np.fromfile(phi).reshape(n, n, n), _tagged(p.stdout, ("edit",))["edit"])
    ref = got["invalidate"][0]
    node = (n // 3, 0, n // 4 + 1) if where == "face" else (n // 3, n // 2 + 1, n // 4 + 1)
    if where == "face":
        assert ref[node] == 5.0                      # the wall keeps its potential
        assert abs(ref[node[0], 1, node[2]]) > 0.5   # and the solution next to it feels it
        assert got["invalidate"][1]["converged"] == "0" and got["invalidate"][1]["cycles"] == "60"
    for tag in runs:
        assert_bit_equal(got[tag][0], ref, f"phi after the edited solve, policy {tag}")
        assert got[tag][1]["cycles"] == got["invalidate"][1]["cycles"]
    assert got["invalidate"][1]["phi_uploads"] == "2"


def test_driver_fails_loudly_without_gpu(box, mg):
    if mg.device_count() > 0:
        pytest.skip("a GPU is present")
    p = subprocess.run([box, "21", "1"], capture_output=True, text=True)
    assert p.returncode == 1
    assert "no HIP device" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("n,depth", [(41, 2), (65, 3)])
def test_solve_through_potential_solver_interface(box, gpu, orc, tmp_path, n, depth):
    """PotentialSolver* -> solve() on the GPU equals the oracle: bit-identical to its red-black
    twin, <= 1e-6 relative L2 to the lexicographic (reference) ordering."""
    out = tmp_path / "phi.bin"
    p = subprocess.run([box, str(n), str(depth), "1e-6", "--dump", str(out)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert "converged=1" in p.stdout
    phi = np.fromfile(out).reshape(n, n, n)
    N = (n, n, n)
    dh = orc.main_extents_dh(N)
    # the driver fills rho with its own libm calls; rebuild it the same way the oracle does and
    # accept last-bit differences in rho by comparing through the 1e-6 bar for both orderings
    rho = orc.synthetic_rho(N)
    phi_rb = np.zeros(N)
    orc.Solver(N, dh, depth).solve(phi_rb, rho, 10000, 1e-6, rb=True)
    phi_lex = np.zeros(N)
    orc.Solver(N, dh, depth).solve(phi_lex, rho, 10000, 1e-6)
    assert rel_l2(phi, phi_rb) <= 1e-9
    assert rel_l2(phi, phi_lex) <= 1e-6
    assert not phi[0].any() and not phi[:, :, -1].any()  # Dirichlet faces untouched


@pytest.mark.gpu
def test_reference_menu_on_stdin(box, gpu, orc, tmp_path):
    """`printf '4\\n2\\n' | box_mg 41 --menu` is the reference's `printf '4\\n2\\n' | ./box 41`
    (cpp/Main.cpp:62-107): MultiGridSolverV2(world, 10000, 1e-4).  The cycle count and the potential
    equal the oracle's red-black solve at that tolerance; the menu's other solvers are refused."""
    n = 41
    out = tmp_path / "phi.bin"
    p = subprocess.run([box, str(n), "--menu", "--dump", str(out)], input="4\n2\n", capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert "Choose solver (1)Gauss-Seidel" in p.stdout and "MultiGrid V-Cycle size (1)V1" in p.stdout
    N = (n, n, n)
    ref = np.zeros(N)
    ok, cycles, _, _ = orc.Solver(N, orc.main_extents_dh(N), 2).solve(ref, orc.synthetic_rho(N), 10000, 1e-4, rb=True)
    assert ok and f"converged=1 cycles={cycles} " in p.stdout
    assert rel_l2(np.fromfile(out).reshape(N), ref) <= 1e-9
    p = subprocess.run([box, "21", "--menu"], input="2\n", capture_output=True, text=True)
    assert p.returncode == 1 and "not part of the multigrid library" in p.stderr


@pytest.mark.gpu
def test_gauss_seidel_and_field_through_the_interface(box, gpu, orc, tmp_path):
    """depth 0 = GaussSeidelSolver behind the same base pointer; computeEF() is answered from the
    device-resident potential and must equal the host formula applied to the returned phi."""
    n = 33
    out, out_ef = tmp_path / "phi.bin", tmp_path / "ef.bin"
    p = subprocess.run([box, str(n), "0", "1e-6", "--dump", str(out), "--dump-ef", str(out_ef)],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert "converged=1" in p.stdout
    N = (n, n, n)
    dh = orc.main_extents_dh(N)
    phi = np.fromfile(out).reshape(N)
    ef = np.fromfile(out_ef).reshape(N + (3,))
    assert_bit_equal(ef, orc.compute_ef(phi, dh), "E from the device-resident phi")
    lex = np.zeros(N)
    assert orc.gs_solve(lex, orc.synthetic_rho(N), dh, 100000, 1e-6)[0]
    assert rel_l2(phi, lex) <= 1e-6


def _parse_vti(path):
    import re
    text = open(path).read()
    head = re.search(r'<ImageData Origin="([^"]*)" Spacing="([^"]*)" WholeExtent="([^"]*)">', text)
    arrays = {}
    for name, comps, body in re.findall(
            r'<DataArray Name="(\w+)" NumberOfComponents="(\d)" format="ascii" type="Float64">\n(.*?)</DataArray>',
            text, flags=re.S):
        arrays[name] = (int(comps), [np.array(line.split(), dtype=float) for line in body.strip("\n").split("\n")])
    return head.groups(), arrays


def test_vti_writer_format(tmp_path):
    """The reference's .vti layout (cpp/Output.cpp:12-65, cpp/Field.h:221-228): ASCII ImageData,
    one k-plane per line, i fastest within it; vectors as three numbers per node."""
    exe = os.path.join(ROOT, "host", "vti_selftest")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "vti_selftest"], stdout=subprocess.DEVNULL)
    out = tmp_path / "t.vti"
    subprocess.check_call([exe, str(out)])
    (origin, spacing, extent), arrays = _parse_vti(out)
    assert origin == "-0.2 -0.2 -0.1" and extent == "0 2 0 3 0 4"
    assert np.allclose([float(x) for x in spacing.split()], [0.2, 0.1, 0.1])
    assert open(out).read().startswith('<VTKFile type="ImageData">\n<ImageData Origin=')
    ni, nj, nk = 3, 4, 5
    comps, lines = arrays["phi"]
    assert comps == 1 and len(lines) == nk
    for k in range(nk):
        want = [100 * i + 10 * j + k + 0.5 for j in range(nj) for i in range(ni)]
        assert lines[k].tolist() == want
    comps, lines = arrays["ef"]
    assert comps == 3 and len(lines) == nk
    assert lines[2].reshape(nj, ni, 3)[3, 1].tolist() == [1.0, 3.0, 2.0]
    assert list(arrays) == ["NodeVol", "phi", "rho", "ef"]  # the reference's order, without its per-species densities
    comps, lines = arrays["NodeVol"]
    cell = 0.2 * 0.1 * 0.1
    assert comps == 1 and np.isclose(lines[0][0], cell / 8) and np.isclose(lines[2][1 * ni + 1], cell)
    # every node: the cell volume halved once per wall the node lies on (cpp/World.cpp:74-85)
    for k in range(nk):
        want = [cell * 0.5 ** ((i in (0, ni - 1)) + (j in (0, nj - 1)) + (k in (0, nk - 1)))
                for j in range(nj) for i in range(ni)]
        assert np.allclose(lines[k], want, rtol=1e-5, atol=0)  # 6 significant digits in the file


@pytest.mark.gpu
def test_driver_writes_vti_of_the_solution(box, gpu, orc, tmp_path):
    n = 21
    out, vti = tmp_path / "phi.bin", tmp_path / "f.vti"
    p = subprocess.run([box, str(n), "2", "1e-6", "--dump", str(out), "--vti", str(vti)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    phi = np.fromfile(out).reshape(n, n, n)
    _, arrays = _parse_vti(vti)
    got = np.array(arrays["phi"][1]).reshape(n, n, n)          # [k][j][i]
    assert np.allclose(got.transpose(2, 1, 0), phi, rtol=1e-5, atol=1e-12)  # 6 significant digits in the file
    ef = np.array(arrays["ef"][1]).reshape(n, n, n, 3).transpose(2, 1, 0, 3)
    assert np.allclose(ef, orc.compute_ef(phi, orc.main_extents_dh((n, n, n))), rtol=1e-5, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("n,depth,tol,slabs", [("129", "3", "1e-6", ["--gpus", "0,0,0"]),
                                               # levels 1 and 2 partitioned, on deep halos (one exchange of R_l per cycle)
                                               ("257", "5", "1e-2", ["--gpus", "0,0", "--cut", "3"])], ids=["129", "257_deep_halos"])
def test_multi_gpu_solver_behind_the_same_interface(box, gpu, tmp_path, n, depth, tol, slabs):
    """MultiGridSolverHIP(world, max_it, tol, depth, {0, 0, 0}): three slabs (here all on GPU 0) behind
    PotentialSolver::solve() give the single-GPU solver's potential bit for bit; computeEF()
    differentiates each slab on its GPU and agrees too."""
    outs = []
    for extra in ([], slabs):
        phi, ef = tmp_path / f"phi{len(extra)}.bin", tmp_path / f"ef{len(extra)}.bin"
        p = subprocess.run([BOX, n, depth, tol, "--dump", str(phi), "--dump-ef", str(ef)] + extra,
                           capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        outs.append((np.fromfile(phi), np.fromfile(ef), [l for l in p.stdout.splitlines() if "converged=" in l][0].split()[:2]))
    assert outs[0][2] == outs[1][2]
    assert_bit_equal(outs[1][0], outs[0][0], "phi from three slabs")
    assert_bit_equal(outs[1][1], outs[0][1], "E from the slabs")


def _c_example():
    exe = os.path.join(ROOT, "host", "c_abi_example")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "c_abi_example"], stdout=subprocess.DEVNULL)
    return exe


def test_header_is_plain_c99(mg):
    """include/mg_hip.h compiled by gcc -std=c99 -pedantic -Wall -Wextra -Werror inside a C program
    that uses the boundary the way a foreign-function binding does; without a GPU that program
    stops at mg_create with the library's message."""
    exe = _c_example()
    if mg.device_count() > 0:
        pytest.skip("a GPU is present")
    p = subprocess.run([exe, "21", "1"], capture_output=True, text=True)
    assert p.returncode == 1 and "no HIP device" in p.stderr
    assert "MB of device memory" in p.stdout  # mg_device_bytes needs no device


@pytest.mark.gpu
def test_solve_from_plain_c(gpu, orc, tmp_path):
    n, depth = 41, 2
    out = tmp_path / "phi.bin"
    p = subprocess.run([_c_example(), str(n), str(depth), "1e-6", str(out)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    N = (n, n, n)
    want = np.zeros(N)
    ok, cycles, _, _ = orc.Solver(N, orc.main_extents_dh(N), depth).solve(want, orc.synthetic_rho(N), 10000, 1e-6, rb=True)
    assert ok and f"converged=1 cycles={cycles} " in p.stdout
    assert rel_l2(np.fromfile(out).reshape(N), want) <= 1e-9  # rho comes from the C library's sin(): last-bit differences
