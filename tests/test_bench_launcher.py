"""`python bench.py --gpus N` must start its own ranks, report the size of the process group that actually ran, and
fail loudly when a rank dies or when the group does not match --gpus (VERDICT r1 missing #1).  Exercised here without a
GPU through bench.py's --dry mode: same launcher, gloo on 127.0.0.1, the same flat gradient all-reduce."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_self_launch_two_ranks_dry():
    p = _run(["--gpus", "2", "--dry", "--steps", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, printed by rank 0"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["allreduce_ok"] is True and out["dry"] is True


def test_a_failing_rank_fails_the_launch():
    p = _run(["--gpus", "2", "--dry", "--steps", "1"], {"NFL_BENCH_FAIL_RANK": "1"})
    assert p.returncode != 0
    assert "rank 1 exited" in p.stderr


def test_refuses_a_process_group_of_another_size():
    """Under an external launcher (WORLD_SIZE set) --gpus must equal the group size: no silent n_gpus = 1 for --gpus 8."""
    p = _run(["--gpus", "8", "--dry"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                        "MASTER_PORT": "29999"})
    assert p.returncode != 0
    assert "refusing" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


import pytest


@pytest.mark.gpu
def test_two_ranks_drive_the_real_render_path():
    """The N > 1 path with the REAL kernels: `python bench.py --gpus 2` starts two ranks that share the box's one GPU
    (gloo for the collective, both on device 0: NFL_BENCH_BACKEND / NFL_BENCH_ONE_DEVICE), each renders and
    back-propagates its own ray shard, the flat gradient all-reduce averages, Adam steps -- and the two replicas'
    parameters stay bit-identical (the RCCL run itself needs the 8-GPU node)."""
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-extras"],
             {"NFL_BENCH_BACKEND": "gloo", "NFL_BENCH_ONE_DEVICE": "1"}, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["ranks"] == 2
    assert out["replica_param_max_diff"] == 0.0
    assert out["value"] > 0


@pytest.mark.gpu
def test_two_ranks_graphed_step_keeps_replicas_identical():
    """The same rehearsal with the step replayed from HIP graphs: with gloo the collective cannot be captured,
    so the step is two graphs around an eager in-place all-reduce of the gradient arena; after the replays the two replicas'
    parameters must still be bit-identical.
    (This test found, at the end of round 3, that a hipMemsetAsync captured into a graph -- the zeroing of d_gmax in
    nfl_composite_backward -- can take effect out of order with the kernel after it when a second process replays graphs on the
    same GPU: the maxima wiped, the loss scale wrong, non-finite gradients on one rank in up to half of the runs on some boxes.
    The library zeroes with a kernel since; tests/diag_two_rank_graph.py is the reproducer.)"""
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-extras", "--graph", "--sustained-seconds", "0"],
             {"NFL_BENCH_BACKEND": "gloo", "NFL_BENCH_ONE_DEVICE": "1"}, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["hip_graph"] is True and out["backend"] == "gloo"
    assert out["all_reduce"]["captured_in_graph"] is False
    assert out["replica_param_max_diff"] == 0.0
    assert out["value"] > 0
