"""A short campaign of every GPU fuzzer (tools/fuzz_*.py: random models / plans / networks through the C ABI against the oracle) inside
the -m gpu suite, so that the suite the driver runs also walks shapes nobody wrote down.  The long campaigns are run by hand
(profiles/rNN/gpu_fuzz_campaign.log); the seeds here differ from theirs."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CAMPAIGNS = [("fuzz_gmm.py", 300), ("fuzz_tied.py", 300), ("fuzz_scorers.py", 200), ("fuzz_frontends.py", 200), ("fuzz_more.py", 100),
             ("fuzz_ffnn.py", 40), ("fuzz_backend.py", 200), ("fuzz_estimate.py", 200), ("fuzz_cache.py", 30)]


@pytest.mark.gpu
@pytest.mark.parametrize("script,cases", CAMPAIGNS)
def test_fuzzer_finds_nothing(script, cases):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), str(cases), "977"], cwd=ROOT, capture_output=True, text=True,
                       timeout=900)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-8:])
    assert r.returncode == 0, tail
    assert "mismatch" not in tail.lower() or " 0 mismatches" in tail, tail
