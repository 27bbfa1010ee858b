"""The CPU baseline bench.py reports (`cpu_baseline`, kind "port") is the oracle timed on the GPU box's host cores, because
the reference itself cannot travel.  It must not be a slower program than the one it stands for (VERDICT r2 weak #9b:
it was 25-30 % slower): here, where the reference is importable, both are timed on the same weights, rays and draws,
alternating, best of N -- the oracle may be at most 10 % slower on the forward and on the train step.  Skipped where
/root/reference does not exist (the GPU box)."""
import os

import pytest

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="needs the reference (build container only)")


def test_oracle_is_as_fast_as_the_reference():
    import time_cpu_baseline as tc
    b = tc.measure(R=512, reps=3)
    print({k: round(v, 3) for k, v in b.items()})
    assert b["fwd_ratio"] <= 1.10, b
    assert b["train_ratio"] <= 1.10, b
