"""The committed fixtures ARE what tests/golden/make_golden.py produces from the unmodified reference: three small archives are
regenerated into a scratch directory (about two seconds each) and compared BYTE FOR BYTE with the files under tests/golden/ -
an episode group (static_mixes: 5 reference-run episodes incl. their `__coop` arrays), the adapter records and the GA3C state
vectors.  The whole recipe (`python tests/golden/make_golden.py`, ~100 s) reproduces the other archives the same way; round 4
re-ran it and re-committed the nine episode groups whose `__coop` arrays the generator had gained after they were last stored.
Needs the reference checkout (this container only): skipped where /root/reference does not exist (the GPU box)."""
import filecmp
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference checkout is only present in the development container")
@pytest.mark.parametrize("flag,archive", [("--static-mixes-only", "static_mixes.npz"), ("--adapters-only", "adapters.npz"),
                                          ("--ga3c-only", "ga3c_states.npz")])
def test_generator_reproduces_committed_fixture(tmp_path, flag, archive):
    env = dict(os.environ, CAGYM_GOLDEN_OUT=str(tmp_path))
    subprocess.run([sys.executable, os.path.join(GOLD, "make_golden.py"), flag], check=True, env=env, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, timeout=300)
    assert filecmp.cmp(str(tmp_path / archive), os.path.join(GOLD, archive), shallow=False), archive + " differs from the committed fixture"


def test_every_episode_group_carries_coop():
    """every episode fixture holds the agents' cooperation coefficients (replay passes them on: RVO worlds need them)"""
    import numpy as np
    import golden_util as gu
    for g in gu.all_groups():
        z = np.load(os.path.join(GOLD, g + ".npz"))
        cases = sorted({k.split("__")[0] for k in z.files})
        missing = [c for c in cases if c + "__coop" not in z.files]
        assert not missing, (g, missing[:3])
