"""CPU: repeat-sharded merge: vectorised replay == oracle evidence; world_size-2 gloo."""
import os
import subprocess
import sys

import numpy as np

from tests import oracle_api as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(seed, nlive=60, batch=1):
    s = orc.settings(6, 0, nlive=nlive, num_repeats=12, seed=seed, batch=batch)
    L, P, keep = orc.make_problem("gaussian", 6)
    return orc.run(s, L, P)


def test_replay_equals_engine_evidence_linear_mode():
    from polychordlite_amd.merge import evidence_replay, lived_records
    o = _run(3)
    lz, var = evidence_replay(*lived_records(o))
    assert abs(lz - o["logZ"]) < 1e-9 and abs(var - o["varlogZ"]) < 1e-9


def test_merged_runs_shrink_the_error():
    from polychordlite_amd.merge import evidence_replay, lived_records
    runs = [_run(s) for s in range(4)]
    L = np.concatenate([lived_records(r)[0] for r in runs]); B = np.concatenate([lived_records(r)[1] for r in runs])
    lz, var = evidence_replay(L, B)
    single = np.mean([r["varlogZ"] for r in runs])
    assert var < 0.4 * single                      # ~ 1/4
    assert abs(lz) < 4 * np.sqrt(var) + 0.2        # truth ~ 0 (6-D Gaussian inside the unit box)


WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from tests import oracle_api as orc
from polychordlite_amd.merge import merge_runs, evidence_replay, lived_records
dist.init_process_group("gloo")
rank = dist.get_rank()
s = orc.settings(6, 0, nlive=60, num_repeats=12, seed=10 + rank, batch=1)
L, P, keep = orc.make_problem("gaussian", 6)
o = orc.run(s, L, P)
o["entry"] = o["dead"][:, -2]
m = merge_runs(o, dist, torch, 0)
if rank == 0:
    runs = []
    for r in range(dist.get_world_size()):
        s2 = orc.settings(6, 0, nlive=60, num_repeats=12, seed=10 + r, batch=1)
        runs.append(orc.run(s2, L, P))
    Ls = np.concatenate([lived_records(x)[0] for x in runs]); Bs = np.concatenate([lived_records(x)[1] for x in runs])
    lz, var = evidence_replay(Ls, Bs)
    assert m["n_runs"] == 2 and abs(m["logZ"] - lz) < 1e-12, (m, lz)
    print("MERGE_OK", m["logZ"], lz)
dist.barrier(); dist.destroy_process_group()
"""


def test_all_gather_merge_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29611", str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=600)
    assert "MERGE_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
