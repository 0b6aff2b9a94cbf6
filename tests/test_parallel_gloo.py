"""Multi-process CPU tests (gloo, world_size 2) of the multi-GPU multi-try selection: score
all-gather + winner broadcast (bayesfmmm_amd/parallel.py).  No GPU needed: the per-rank chain
results are synthetic."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from bayesfmmm_amd.parallel import select_winner

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_select_winner_rules():
    assert select_winner([1.0, 3.0, 2.0], [0, 1, 2]) == 1
    assert select_winner([3.0, 3.0], [5, 2]) == 1            # tie -> lowest chain index
    assert select_winner([-np.inf, -7.0], [-1, 3]) == 1      # ranks without a chain are skipped
    assert select_winner([-5.0, -np.inf], [0, -1]) == 0


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    from bayesfmmm_amd.parallel import gather_select_broadcast
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    case = int(sys.argv[1])
    rng = np.random.default_rng(100 + rank)
    nu = np.asfortranarray(rng.standard_normal((2, 3, 4)))
    if case == 0:      # rank 1 has the better chain
        local = dict(best_score=float(-10.0 + 5 * rank), best_chain=float(rank), nu=nu, loglik=np.arange(4.0) + rank)
    elif case == 1:    # tie: lowest chain index (held by rank 1) wins
        local = dict(best_score=-3.0, best_chain=float(7 - 5 * rank), nu=nu, loglik=np.arange(4.0) + rank)
    else:              # rank 1 had no chain to run
        local = dict(best_score=-3.0, best_chain=0.0, nu=nu, loglik=np.arange(4.0)) if rank == 0 else None
    out = gather_select_broadcast(local, None)
    expect_rank = {0: 1, 1: 1, 2: 0}[case]
    ref = np.asfortranarray(np.random.default_rng(100 + expect_rank).standard_normal((2, 3, 4)))
    assert out["nu"].shape == (2, 3, 4) and np.array_equal(out["nu"], ref), (rank, case)
    assert out["loglik"][0] == (expect_rank if case < 2 else 0.0)
    dist.barrier()
    dist.destroy_process_group()
    print("ok", rank)
""") % ROOT


def test_gather_select_broadcast_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    for case in range(3):
        port = 29500 + (os.getpid() + case) % 2000
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), str(case)]
        env = dict(os.environ, OMP_NUM_THREADS="1")
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        assert res.returncode == 0, res.stdout + res.stderr
        assert res.stdout.count("ok") == 2
