"""bench.py's launch contract without a GPU: ``python bench.py --gpus N`` (the driver's N = 1 command form with N changed)
must start N ranks by itself - /root/reference/tools/launch.py:159-192 forks the ranks the same way - and the launcher form
(``python -m torch.distributed.run ... bench.py --gpus N``) must keep working.  On this CPU-only container every rank stops at
"bench.py needs a GPU": the point is that it is the RANKS that say so (WORLD_SIZE = N reached them), not the parent."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.timeout(300)
def test_gpus_n_without_a_launcher_starts_n_ranks():
    import torch
    if torch.cuda.is_available():
        pytest.skip("the GPU form of this test is tests/test_clip_shard_gpu.py::test_bench_py_two_ranks_gloo_rehearsal")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1",
                          "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=280, env=_env())
    assert out.returncode != 0
    assert "WORLD_SIZE=1" not in out.stderr                      # the round-3 refusal
    # a rank said so (the launcher may end the other rank before it gets that far), under torch.distributed.run
    assert "bench.py needs a GPU" in out.stderr and "torch.distributed.elastic" in out.stderr, out.stderr[-1500:]


def test_mismatched_world_size_is_refused():
    env = dict(_env(), WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                         timeout=120, env=env)
    assert out.returncode != 0 and "--gpus 2 but WORLD_SIZE=4" in out.stderr


def test_bench_asks_for_eight_hardware_queues_unless_told_otherwise():
    """bench.py sets GPU_MAX_HW_QUEUES=8 before torch is imported (the HIP runtime reads it when it initialises; two eager
    pipeline lanes are five streams, profiles/r04_hw_queues.txt) and leaves a value from the environment alone."""
    code = "import os, bench; print('HWQ', os.environ['GPU_MAX_HW_QUEUES'])"
    for preset, want in ((None, "8"), ("4", "4")):
        env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
        if preset is not None:
            env["GPU_MAX_HW_QUEUES"] = preset
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
        assert out.returncode == 0 and f"HWQ {want}" in out.stdout, out.stderr[-1000:]
