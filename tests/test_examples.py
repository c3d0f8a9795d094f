"""The example drivers (examples/) run end to end on the GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("script,args,expect", [
    ("example.py", [], "Experiment over."),
    ("dmcts_experiment.py", ["--worlds", "8", "--steps", "3", "--Ntree", "4", "--Ncycles", "2", "--Nsims", "3"], "cumulative team reward"),
    ("vecenv_random_policy.py", [], "env-steps/s"),
])
def test_example_runs(script, args, expect):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and expect in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
