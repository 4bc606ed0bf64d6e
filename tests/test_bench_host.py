"""Host logic of bench.py that needs no GPU: the PMC traffic figures it quotes come from committed
profile summaries and must be found under the kernel names the library's kernels have now, and a
rank count beyond the machine's GPUs is refused in one line before anything is started."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_traffic_of_the_dominant_kernels_is_found_in_the_committed_summary():
    import bench
    for kind, lo, hi in [("smooth_fine", 25.0, 30.0), ("residual", 17.0, 21.0)]:
        nbytes, source = bench.profiled_traffic(kind)
        assert nbytes is not None and source.endswith("_summary.json"), kind
        assert lo <= nbytes / 513.0 ** 3 <= hi, (kind, nbytes / 513.0 ** 3)   # bytes per fine point, >= compulsory


def test_slab_traffic_is_found_for_every_configuration_of_the_scaling_run():
    import bench
    for n, world, planes in [(1025, 8, 128), (1025, 4, 256), (1025, 2, 512), (2049, 8, 256)]:
        nbytes, source = bench.slab_profiled_traffic(n, world)
        assert nbytes is not None and source, (n, world)
        assert 24.0 <= nbytes / (planes * float(n) ** 2) <= 31.0, (n, world, nbytes / (planes * float(n) ** 2))
    assert bench.slab_profiled_traffic(777, 3) == (None, None)


def test_more_ranks_than_gpus_is_refused_before_anything_starts():
    import torch
    have = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 3)], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
    lines = [x for x in (r.stdout + r.stderr).splitlines() if x.strip()]
    assert len(lines) == 1 and str(have + 3) in lines[0] and "Traceback" not in lines[0], lines
    # and nothing that parses as a result line
    try:
        json.loads(lines[0])
        assert False, "a refusal must not look like a benchmark line"
    except ValueError:
        pass
