"""libmg_hip.so and PyTorch-ROCm in one process, in either import order: the binding maps torch's
copy of the HIP runtime before the library, so the process never holds two runtimes."""
import subprocess
import sys

import pytest

CODE = """
import sys
sys.path.insert(0, {root!r})
{first}
{second}
from multigrid_amd import libmg
import torch
libmg.load()
maps = open('/proc/self/maps').read()
runtimes = sorted({{l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l}})
print('RUNTIMES', len(runtimes), runtimes)
n = libmg.device_count()
print('DEVICES', n, torch.cuda.device_count())
if n:
    import numpy as np
    t = torch.ones(4, device='cuda')          # torch's runtime initialised
    s = libmg.op_sumsq(np.arange(10.0))       # and the library's
    print('SUMSQ', s, float(t.sum()))
"""


def _run(first, second):
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", CODE.format(root=root, first=first, second=second)],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    return dict(line.split(" ", 1) for line in p.stdout.splitlines() if line.split(" ", 1)[0].isupper())


@pytest.mark.parametrize("first,second", [("from multigrid_amd import libmg; libmg.load()", "import torch"),
                                          ("import torch", "from multigrid_amd import libmg; libmg.load()")],
                         ids=["library_first", "torch_first"])
def test_one_hip_runtime_whatever_the_import_order(first, second):
    out = _run(first, second)
    assert out["RUNTIMES"].startswith("1 "), out["RUNTIMES"]


@pytest.mark.gpu
def test_both_runtime_users_see_the_gpu_when_the_library_loads_first(gpu):
    out = _run("from multigrid_amd import libmg; libmg.load()", "import torch")
    assert out["RUNTIMES"].startswith("1 ")
    assert out["DEVICES"].split()[0] != "0" and out["DEVICES"].split()[1] != "0"
    assert out["SUMSQ"].split()[0] == "285.0"
