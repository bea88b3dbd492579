"""Opt-in check of a kernel CANDIDATE that is not part of the product library (tools/next/): skipped unless UAV_TEST_CANDIDATE=1 and
the variant library has been built (`bash tools/next/build_variant.sh`).  The 8-phase k-loop keeps the per-accumulator K order of
the production kernel, so the outputs of seven conv shapes (tools/conv_digest.py: 3x3, temporal, GEGLU linear, up-sampler phases,
stride 2 on odd sizes, 3x3x3, two-source 1x1) must match BIT FOR BIT; run under a short timeout — a barrier-count mistake in a
never-executed kernel would hang the GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT = os.path.join(ROOT, "tools", "ab", "libuav_hip_8phase.so")
pytestmark = pytest.mark.gpu


def _digests(dmav):
    env = dict(os.environ, UAV_HIP_LIB=VARIANT, UAV_CONV_TILE="256", UAV_CONV_DMAV=str(dmav))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "conv_digest.py")], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-1500:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_8phase_k_loop_same_bits_as_production():
    if os.environ.get("UAV_TEST_CANDIDATE") != "1" or not os.path.exists(VARIANT):
        pytest.skip("candidate kernel: opt-in (UAV_TEST_CANDIDATE=1 + tools/next/build_variant.sh)")
    ref, new = _digests(1), _digests(8)
    assert ref == new, {k: (ref[k], new.get(k)) for k in ref if ref[k] != new.get(k)}
