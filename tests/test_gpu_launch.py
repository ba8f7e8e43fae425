"""The launched branch of bench.py on real hardware (-m gpu): `python -m torch.distributed.run --nproc-per-node=1 bench.py --gpus 1` is the
command line the driver uses for N > 1 with N = 1 - it walks init_process_group("nccl", device_id), the RCCL weight broadcast, the checksum
all-reduce, every barrier, the device-side MAX / all-gather and destroy_process_group, none of which run without a launcher.  The streams
shard with no data-path collective, so a world of one exercises every collective call the path has."""
import json
import math
import os
import socket
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _json_line(stdout: str) -> dict:
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def _run(cmd, env=None):
    p = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    return _json_line(p.stdout)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_under_torchrun_walks_the_rccl_branch_and_matches_the_plain_run():
    flags = ["--gpus", "1", "--steps", "50", "--warmup", "5", "--no-cpu-baseline"]
    plain = _run([sys.executable, "bench.py"] + flags)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    launched = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                     "--master-port", str(_free_port()), "bench.py"] + flags, env=env)
    assert plain["collective_backend"].startswith("none") and plain["rccl_world_size"] == 1
    assert launched["collective_backend"] == "nccl", launched["collective_backend"]             # RCCL on ROCm
    assert launched["rccl_world_size"] == 1 and launched["n_gpus"] == 1
    assert math.isfinite(launched["weight_broadcast_ms"]) and launched["weight_broadcast_ms"] > 0.0
    assert len(launched["per_rank_ms_per_step"]) == 1 and math.isfinite(launched["per_rank_ms_per_step"][0])
    assert launched["steps"] == 50 and launched["warmup"] == 5 and launched["config"] == plain["config"]
    # the kernel the two runs time is the same one: HIP-event time per launch within 3 %, whole-step wall rate within 5 % (a 1.6 ms block on the host clock)
    k0, k1 = plain["roofline"]["kernel_ms"], launched["roofline"]["kernel_ms"]
    assert abs(k1 - k0) <= 0.03 * k0, (k0, k1)
    assert abs(launched["value"] - plain["value"]) <= 0.05 * plain["value"], (plain["value"], launched["value"])
    # parity of what was timed (bench.py, after the timed region): fp32 rounding, far inside north_star's 1e-4
    for line in (plain, launched):
        assert line["parity_rms_rel"] is not None and line["parity_rms_rel"] < 3e-6, line["parity_rms_rel"]
        assert line["parity_cache_rms_rel"] < 3e-6, line["parity_cache_rms_rel"]
    assert launched["roofline"]["frac"] > 0.3


def test_host_contention_probe_is_labelled_as_such():
    """`--share-gpu`: N ranks launching on ONE GPU (gloo) - a reading of what N launch threads cost each other under the container's CPU
    quota, never a scaling number: the line says so and carries no roofline claim."""
    line = _run([sys.executable, "bench.py", "--gpus", "2", "--share-gpu", "--streams", "32", "--steps", "20", "--warmup", "5", "--blocks", "5",
                 "--no-cpu-baseline", "--no-parity"])
    assert line["contention_probe"] is True and line["n_gpus_really_used"] == 1 and line["roofline"] is None
    assert "NOT a scaling number" in line["metric"] and line["collective_backend"] == "gloo"
    assert len(line["per_rank_ms_per_step"]) == 2 and line["host_enqueue_us_per_step"] > 0.0
