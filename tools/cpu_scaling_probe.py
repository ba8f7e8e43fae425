#!/usr/bin/env python3
"""Host-side scaling probe of the C/OpenMP oracle (cpu_baseline): frames/s vs OpenMP width, with the cgroup CPU quota
and the OpenMP wait policy, to explain where the baseline stops scaling.  Runs on the GPU box's host (no GPU needed)."""
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
        if os.path.exists(f):
            print(f, "=", open(f).read().strip())
    print("affinity:", len(os.sched_getaffinity(0)), "cpu_count:", os.cpu_count(), "loadavg:", open("/proc/loadavg").read().strip())
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        from common import MODEL_KWARGS
        from oracle.c_oracle import COracle
        from oracle.fe_oracle import FEConfig, fold_state_dict
        from oracle.weightgen import make_input, make_training_state_dict
        cfg = FEConfig.from_model_kwargs(MODEL_KWARGS["fe_b"][0])
        fused = fold_state_dict(make_training_state_dict(cfg, 2), cfg)
        B, H = 256, cfg.hop_size
        x = make_input(B, 8 * H, 1, 16000)
        for threads in (1, 8, 16, 32, 64, 128, 256):
            co = COracle(cfg, fused, threads=threads)
            cs, ci, h = co.initialize_cache(B)
            co.step(x[:, :H], cs, ci, h)
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < 1.5:
                co.step(x[:, (n % 8) * H:(n % 8 + 1) * H], cs, ci, h)
                n += 1
            dt = time.perf_counter() - t0
            print(f"  threads {threads:3d}: {B * n / dt:10.0f} frames/s  ({dt / n * 1e3:7.2f} ms per 256-stream hop)")
        return
    for env in ({}, {"OMP_WAIT_POLICY": "ACTIVE", "OMP_PROC_BIND": "close", "OMP_PLACES": "cores"}, {"OMP_WAIT_POLICY": "ACTIVE", "OMP_PROC_BIND": "spread", "OMP_PLACES": "threads"}):
        print("env:", env)
        subprocess.run([sys.executable, __file__, "child"], env={**os.environ, **env})


if __name__ == "__main__":
    main()
