"""The host-side plan builder under AddressSanitizer + UBSan (GPU sanitizers are not available on this pool; the host code is
where fixed-size tables are filled by configuration-dependent loops): tests/plan_sweep.cpp builds forward and backward plans
for every encoder / latent width combination, accepted or rejected."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_plan_builder_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "plan_sweep")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", os.path.join(ROOT, "tests", "plan_sweep.cpp"),
           os.path.join(ROOT, "nerf_fl_amd", "csrc", "nfl_plan.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and "sanitize" in b.stderr and "cannot find" in b.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "plans ok" in r.stdout and "ERROR" not in r.stderr
    n_ok = int(r.stdout.split("plans ok")[1].split()[0])
    assert n_ok > 10000
