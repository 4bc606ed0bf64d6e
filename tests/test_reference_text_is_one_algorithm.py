"""The reference's ten hand-unrolled multigrid classes, read as TEXT, are one algorithm -- TEST
INFRASTRUCTURE ONLY; runs where the reference tree is present (the build container), skipped elsewhere.

The oracle (oracle/mg_oracle.c) and the HIP library run ONE depth-generic cycle where the reference
has a separate solve() per depth and restriction variant (cpp/PotentialSolver.cpp:620-2387 the "B"
classes, :2390-4248 MultiGridSolver ... V5).  The reference cannot be compiled in this image, so this
test establishes from its source text what the generic cycle assumes:

  * every solve() is the same sequence of steps 1) 2) 2b) 3).. 4) 5) [4) 5)].. 6);
  * every restriction, coarse smoothing and interpolation block is, after renaming its level (array
    suffix _<m>h, coefficient suffix, divisor in world.ni/<m>), character for character the block of
    the two-level solver -- the one that tests/test_oracle_second_transcription.py transcribes --
    except for the interpolation's boundary test, where the deeper classes carry world.ni/4-1
    whatever the level (a condition that is false throughout the loop either way);
  * level m uses coefficients 1.0/(m*m*dh*dh) written as the literals 4, 16, 64, 256, 1024, arrays of
    world.ni/m nodes, and the sweep counts the oracle and the library's default configurations hold;
  * a "B" class differs from its namesake in the restriction expression only (and V5B in its sweep
    counts).

Nothing of the reference is executed; the file is only read.
"""
import os
import re

import pytest

REF = "/root/reference/cpp/PotentialSolver.cpp"
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="the reference tree is not present on this machine")

CLASSES = {  # class -> (depth, variant B)
    "MultiGridSolver": (1, False), "MultiGridSolverV2": (2, False), "MultiGridSolverV3": (3, False),
    "MultiGridSolverV4": (4, False), "MultiGridSolverV5": (5, False),
    "MultiGridSolverB": (1, True), "MultiGridSolverV2B": (2, True), "MultiGridSolverV3B": (3, True),
    "MultiGridSolverV4B": (4, True), "MultiGridSolverV5B": (5, True),
}


def _functions():
    """class name -> lines of its solve() body."""
    lines = open(REF).read().split("\n")
    out = {}
    for idx, line in enumerate(lines):
        m = re.match(r"bool (MultiGridSolver\w*)::solve\(\)", line)
        if m:
            end = next(e for e in range(idx + 1, len(lines)) if lines[e].startswith("}"))
            out[m.group(1)] = lines[idx + 1:end]
    return out


def _strip(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)          # block comments (incl. the dead phi_test code)
    text = "\n".join(l.split("//")[0] for l in text.split("\n"))  # line comments
    return re.sub(r"\s+", "", text)


def _blocks(body):
    """The iteration loop of a solve(): [(step label, comment text, code with comments and blanks removed)]."""
    start = next(i for i, l in enumerate(body) if "for (unsigned it=0;it<max_solver_it;it++)" in l)
    out, label, note, acc = [], "head", "", []
    for l in body[start + 1:]:
        m = re.match(r"\s*// (\d)b?\) (.*)", l)
        m2 = re.match(r"\s*// (2b)\) (.*)", l)
        if "if (!converged) cerr" in l:
            break
        if m:
            out.append((label, note, _strip("\n".join(acc))))
            label, note, acc = (m2.group(1) if m2 else m.group(1)), m.group(2), []
        else:
            acc.append(l)
    out.append((label, note, _strip("\n".join(acc))))
    return out


def _levels(code):
    """The mesh levels (1, 2, 4, ...) a block names through its array suffixes."""
    return sorted({int(m or 1) for m in re.findall(r"(?:R|eps)_(\d*)h", code)})


def _rename(code, fine, coarse=None):
    """Level names -> roles: arrays of level `fine` become F, of `coarse` C; the coefficients of
    the block's level lose their suffix; world.n?/<fine> becomes N? (the fine level itself has no
    divisor in the text)."""
    def arr(m):
        lvl = int(m.group(2) or 1)
        return m.group(1) + ("_F" if lvl == fine else "_C" if lvl == coarse else "_?%d" % lvl)
    code = re.sub(r"\b(R|eps)_(\d*)h", arr, code)
    code = re.sub(r"id([xyz])2_(\d+)h", lambda m: "id%s2" % m.group(1) if int(m.group(2)) in (fine, coarse) else m.group(0), code)
    code = re.sub(r"inner\d*h?_its", "ITS", code)
    if fine > 1:
        code = re.sub(r"world\.n([ijk])/%d(?!\d)" % fine, r"world.n\1", code)
    return code


@pytest.fixture(scope="module")
def solvers():
    f = _functions()
    assert set(CLASSES) <= set(f)
    return {name: _blocks(f[name]) for name in CLASSES}


def test_every_class_is_the_same_sequence_of_steps(solvers):
    for name, (depth, _) in CLASSES.items():
        steps = [b[0] for b in solvers[name]]
        want = ["head", "1", "2", "2b"] + ["3"] * depth + ["4"] + ["5", "4"] * (depth - 1) + ["5", "6"]
        assert steps == want, (name, steps)


def test_blocks_of_every_level_are_the_two_level_solvers_blocks_renamed(solvers):
    v2 = solvers["MultiGridSolverV2"]
    canon = {
        "1": v2[1][2], "2": v2[2][2], "2b": v2[3][2], "6": v2[-1][2],
        "3": _rename(v2[5][2], 2, 4),          # R_2h -> R_4h
        "4": _rename(v2[6][2], 4),             # smoothing on the 4h mesh
        "5": _rename(v2[7][2], 2, 4),          # eps_4h -> eps_2h
    }
    # the two-level solver's own first restriction and last interpolation are those blocks one level up
    assert _rename(v2[4][2], 1, 2) == canon["3"]
    assert _rename(v2[9][2], 1, 2) == canon["5"]
    assert _rename(v2[8][2], 2) == canon["4"]
    quirk = re.compile(r"\(([ijk])==world\.n\1/4-1\)")
    for name, (depth, variant_b) in CLASSES.items():
        if variant_b:
            continue
        down, up = 1, None
        for label, note, code in solvers[name][1:]:
            if label in ("1", "2", "2b", "6"):
                if label == "1":  # the sweep count is a local named inner_its in every class
                    assert re.sub(r"inner_its", "ITS", code) == re.sub(r"inner_its", "ITS", canon["1"]), (name, label)
                else:
                    assert code == canon[label], (name, label)
            elif label == "3":
                assert _levels(code) == [down, 2 * down] and f"{2 * down}h mesh" in note, (name, note)
                assert _rename(code, down, 2 * down) == canon["3"], (name, note)
                down *= 2
            elif label == "4":
                lvl = down if up is None else up
                assert _levels(code) == [lvl] and f"{lvl}h mesh" in note, (name, note)
                assert _rename(code, lvl) == canon["4"], (name, note)
                up = lvl
            elif label == "5":
                fine = up // 2
                assert _levels(code) == [fine, up], (name, note)
                got = _rename(code, fine, up)
                if got != canon["5"]:
                    # the boundary test names world.n?/4-1 whatever the level: false inside the loop either way
                    # (the loop stops at world.n?/<fine> - 2), so the branch it guards is the index-0 one alone
                    assert fine not in (1, 4) and len(quirk.findall(got)) == 3, (name, note)
                    got = quirk.sub(lambda m: "(%s==world.n%s-1)" % (m.group(1), m.group(1)), got)
                assert got == canon["5"], (name, note)
                up = fine
        assert down == 2 ** depth and up == 1


def test_coefficients_sizes_and_sweep_counts(solvers, orc, mg):
    f = _functions()
    for name, (depth, variant_b) in CLASSES.items():
        text = _strip("\n".join(f[name]))
        for l in range(1, depth + 1):
            m = 2 ** l
            for a, ax in enumerate("xyz"):
                assert f"doubleid{ax}2_{m}h=1.0/({m * m}*dh[{a}]*dh[{a}]);" in text, (name, m, ax)
        assert f"id x2_{2 ** (depth + 1)}h".replace(" ", "") not in text
        counts = [int(x) for x in re.findall(r"unsignedinner\d*h?_its=(\d+);", text)]
        assert tuple(counts) == orc.schedule(depth, variant_b), (name, counts)
        cfg = mg.default_config((97, 97, 97), (0.1, 0.1, 0.1), depth, 0, variant_b=variant_b) if variant_b else \
            mg.default_config((97, 97, 97), (0.1, 0.1, 0.1), depth, 0)
        assert tuple(cfg.nu[:depth + 1]) == tuple(counts), (name, counts)
        assert "1.4*(" in text and "/Const::EPS_0" in text


def test_b_classes_differ_in_the_restriction_only(solvers):
    seven = ("R_C[i/2][j/2][k/2]=(R_F[i-1][j][k]+2*R_F[i][j][k]+R_F[i+1][j][k]+R_F[i][j-1][k]+2*R_F[i][j][k]+R_F[i][j+1][k]+"
             "R_F[i][j][k-1]+2*R_F[i][j][k]+R_F[i][j][k+1])/12.0;")
    for name, (depth, variant_b) in CLASSES.items():
        if not variant_b:
            continue
        twin = solvers[name[:-1]]
        level = 1
        for (label, note, code), (label_a, _, code_a) in zip(solvers[name], twin):
            assert label == label_a
            if label == "3":
                got, ref = _rename(code, level, 2 * level), _rename(code_a, level, 2 * level)
                assert seven in got and "/64.0" in ref and "/64.0" not in got
                assert got.split("}else{")[0] == ref.split("}else{")[0]  # same loops, same injection branch
                level *= 2
            elif label == "head" and depth == 5:
                pass  # V5B's sweep counts differ (checked against the oracle's table above)
            else:
                assert code == code_a, (name, label, note)
