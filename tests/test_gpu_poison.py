"""The parity tests once more against the -DMYO_POISON=1 build of the same sources (libmyo_hip_poison.so, built by
__graft_entry__.build()): every word of an env's LDS slice starts as a NaN there, and before each step launch helper kernels fill the private
(scratch) memory of every wave slot, ALL vector registers of every SIMD and the scalar registers with NaN patterns (MYO_POISON_MODE, default 7).
A read of an LDS word, a scratch word or a register (lane) that the launch never wrote then turns into wrong numbers or flagged envs instead
of passing on whatever earlier workgroups and kernels left behind -- which is what two code shapes of round 2 did, order-dependently and
without any plain test noticing reliably; round 3 reproduced one of them and traced it to the contents of the vector register file
(profiles/r3_badshape_repro.txt).  Runs in a child process because the library is chosen when it is first loaded
(MYO_HIP_LIB)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parity_suites_pass_on_the_poisoned_build():
    lib = os.path.join(ROOT, "myosuite_mjx_amd", "libmyo_hip_poison.so")
    assert os.path.exists(lib), "libmyo_hip_poison.so is missing: run __graft_entry__.build()"
    env = dict(os.environ, MYO_HIP_LIB=lib)
    suites = ["tests/test_gpu_parity.py", "tests/test_gpu_legs.py", "tests/test_gpu_track.py", "tests/test_gpu_hold.py", "tests/test_gpu_terrain.py",
              "tests/test_gpu_rk4.py", "tests/test_gpu_conditions.py", "tests/test_gpu_walk.py", "tests/test_gpu_env.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", *suites], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout
