"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: stream sharding + weight-blob
broadcast; and sharding invariants."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fastenhancer_amd.parallel import shard_range, synthetic_streams

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_all_streams():
    for n in (0, 1, 7, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            cuts = [shard_range(n, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_synthetic_streams_do_not_depend_on_sharding():
    full = synthetic_streams(0, 6, 500, 16000, seed=5)
    a, b = synthetic_streams(0, 3, 500, 16000, seed=5), synthetic_streams(3, 6, 500, 16000, seed=5)
    assert torch.equal(full, torch.cat([a, b]))
    assert float(full.abs().max()) <= 1.0


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from common import MODEL_KWARGS
    from fastenhancer_amd.config import FEConfig
    from fastenhancer_amd.engine import Engine
    from fastenhancer_amd.parallel import broadcast_blob, gather_frame_counts
    from fastenhancer_amd.weights import default_state_dict
    cfg = FEConfig.from_model_kwargs(**MODEL_KWARGS["fe_t"][0])
    eng = Engine(cfg, None)
    blob = torch.zeros(eng.weight_floats)
    if rank == 0:
        blob = eng.make_blob(default_state_dict(cfg, torch.Generator().manual_seed(11)))
    broadcast_blob(blob, src=0)
    b0, b1 = shard_range(9, world, rank)
    frames, elapsed = gather_frame_counts((b1 - b0) * 10, 1.0 + rank, torch.device("cpu"))
    q.put((rank, float(blob.double().sum()), int(blob.numel()), (b0, b1), frames, elapsed))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_broadcast_and_sharding():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, n0, sh0, f0, e0), (r1, s1, n1, sh1, f1, e1) = res
    assert s0 == s1 and s0 != 0.0 and n0 == n1          # every rank holds rank 0's blob
    assert sh0 == (0, 5) and sh1 == (5, 9)
    assert f0 == f1 == 90 and e0 == e1 == 2.0


def test_bench_refuses_to_measure_fewer_ranks_than_asked():
    """`python bench.py --gpus N` starts N ranks itself; with fewer than N devices it must fail loudly, never run 1 rank."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs 2 visible" in r.stderr and r.stdout.strip() == ""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_bench_launch_path_runs_end_to_end_on_two_gloo_ranks():
    """`bench.py --gpus 2 --cpu-dry-run`: the SUCCESS path of spawn_ranks - bench -> torch.distributed.run -> 2 ranks -> blob broadcast ->
    barrier-bracketed blocks -> exactly one JSON line from rank 0 - without kernels (the refusal branches are tested above)."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--cpu-dry-run", "--steps", "3", "--warmup", "1", "--workload", "fe_t",
                        "--streams", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["world_size"] == 2 and d["n_gpus"] == 2 and d["value"] is None
    assert d["streams_per_rank"] == [5, 5] and d["weight_blob_floats"] > 0 and d["blocks"] >= 1
    assert d["launch_mode"] == "hip_graph"


def test_bench_config3_command_line_dry_run_on_eight_gloo_ranks():
    """BASELINE config 3's exact command line - `bench.py --workload fe_l --gpus 8` (FastEnhancer_L, 256 streams per GPU = 2048 over the
    node) - through the launch path on eight gloo ranks: the host-thread budget is split over the ranks, every rank gets its shard,
    one JSON line comes back."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", "fe_l", "--gpus", "8", "--cpu-dry-run", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["world_size"] == 8 and d["n_gpus"] == 8
    assert d["streams_per_rank"] == [256] * 8 and sum(d["streams_per_rank"]) == 2048
    # r6: a world of more than one rank times ONE HIP graph of the K steps by default (the eager figure beside it), and the line carries the
    # spread over the ranks
    assert d["launch_mode"] == "hip_graph" and "eager_ms_per_step" in d
    sp = d["rank_spread_ms_per_step"]
    assert 0.0 <= sp["min"] <= sp["max"] and sp["widest_block_max_over_min"] >= 1.0
