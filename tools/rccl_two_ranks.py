#!/usr/bin/env python3
"""The RCCL transport between TWO processes, the smallest real check of the multi-GPU path: two
cycles of a grid cut in two, one rank per process, halo planes, all-gather and all-reduce over RCCL,
compared bit for bit with the undivided handle.  On a node with two or more GPUs rank r runs on
GPU r.  On a one-GPU box both ranks land on GPU 0, which RCCL refuses at communicator creation
("invalid usage": two ranks on one device): the tool then prints the library's message and exits 3
-- observed on the test box; what it shows there is only that the refusal comes back as an error
from mg_dist_create on both ranks and nothing hangs.

    python tools/rccl_two_ranks.py [n=129] [depth=3] [cut=2]        (257 5 3 exercises the deep halos)
"""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, n, depth, cut, out_dir):
    import torch  # before libmg_hip: both share one HIP runtime
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multigrid_amd import libmg, problem
    libmg.load()
    N = (n, n, n)
    ids = [libmg.dist_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    try:
        device = rank if torch.cuda.device_count() >= world else 0
        d = libmg.DistRank(N, problem.main_extents_dh(N), depth, rank, world, device=device, unique_id=ids[0], coarse_cut=cut)
    except libmg.MgError as e:
        print(f"rank {rank}: communicator refused: {e}", flush=True)
        dist.destroy_process_group()
        sys.exit(3)
    d.set_rho(lambda lo, hi: problem.synthetic_rho(N, (lo, hi)))
    d.set_phi(lambda lo, hi: np.zeros((hi - lo, n, n)))
    l2 = d.vcycles(2)
    lo, phi = d.get_phi_owned()
    np.save(os.path.join(out_dir, f"phi_{rank}.npy"), phi)
    np.save(os.path.join(out_dir, f"lo_{rank}.npy"), np.array([lo, l2, float(d.deep_halos), d.stats()["bytes_sent"]]))
    d.close()
    dist.barrier()
    dist.destroy_process_group()


def main():
    import tempfile
    import torch.multiprocessing as mp
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 129
    depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    cut = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tempfile.mkdtemp()
    try:
        mp.spawn(worker, args=(2, port, n, depth, cut, out), nprocs=2, join=True)
    except Exception as e:  # noqa: BLE001
        print("ranks failed:", str(e).splitlines()[-1] if str(e) else type(e).__name__)
        sys.exit(3)
    from multigrid_amd import libmg, problem
    N = (n, n, n)
    got = np.full(N, np.nan)
    for r in range(2):
        lo, l2, deep, sent = np.load(os.path.join(out, f"lo_{r}.npy"))
        part = np.load(os.path.join(out, f"phi_{r}.npy"))
        got[int(lo):int(lo) + part.shape[0]] = part
        print(f"rank {r}: planes from {int(lo)}, L2 {l2:.6e}, deep halos {bool(deep)}, bytes sent {int(sent)}")
    with libmg.Handle(N, problem.main_extents_dh(N), depth) as h:
        h.set_rho(problem.synthetic_rho(N))
        h.set_phi(np.zeros(N))
        h.vcycles(2)
        want = h.get_phi()
    same = np.array_equal(got.view(np.uint64), want.view(np.uint64))
    print("two RCCL ranks:", "phi bit-identical to the undivided handle" if same else "phi DIFFERS")
    sys.exit(0 if same else 1)


if __name__ == "__main__":
    main()
