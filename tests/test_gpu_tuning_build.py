"""The second binary (`libsemabs_hip_tune.so`, the same sources with -DSEMABS_TUNING: ablation switches, raster / tile-configuration knobs,
per-workgroup traces for tools/) must compute what the production library computes: the GEMM and UNet suites are re-run against it in a
subprocess (`SEMABS_TUNE_LIB=1` selects it in semabs_amd/_lib.py), with every knob at its default."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_gemm_and_unet_suites_against_the_tuning_build():
    lib = os.path.join(ROOT, "semantic-abstraction_amd", "lib", "libsemabs_hip_tune.so")
    if not os.path.exists(lib):
        pytest.skip("tuning build not present (python semantic-abstraction_amd/build.py --tuning)")
    env = dict(os.environ, SEMABS_TUNE_LIB="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_gemm.py"), os.path.join(ROOT, "tests", "test_gpu_semabs3d.py"),
                        "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout
