"""CPU checks of the two small pieces of test / safety logic added in round 4: the LaserScan border audit the GPU parity tests
use (golden_util.laser_audit) and the bounded intra-workgroup wait of the kernels (csrc/cagym_spin.h, compiled for the host)."""
import os
import subprocess
import sys

import numpy as np

import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_laser_audit_accepts_border_beams_only():
    N, M = 2, 3
    scan = np.zeros((N, M, 16))
    px = np.full((N, M), 0.4321)
    py = np.full((N, M), -1.2345)
    h = np.full((N, M), 0.3)
    pose = (px, py, h)
    assert gu.laser_audit(scan, scan.copy(), pose, pose) == (0, [])
    # a beam that differs although no sample of it is anywhere near a cell border: an index error
    other = scan.copy()
    other[1, 2, 5] = 0.5
    n, bad = gu.laser_audit(scan, other, pose, pose)
    assert n == 1 and len(bad) == 1 and bad[0][:3] == (1, 2, 5) and bad[0][3] > 1e-3
    # the same disagreement with the ego exactly on a cell border (x * 10 integer): explained
    px2 = px.copy()
    px2[1, 2] = 0.4
    n, bad = gu.laser_audit(scan, other, (px2, py, h), (px2, py, h))
    assert n == 1 and not bad
    # ... or with the two backends' poses 1e-4 apart and a sample within that distance of a border: explained by the pose margin
    px3 = px.copy()
    px3[1, 2] = 0.40003
    px4 = px3.copy()
    px4[1, 2] += 1e-4
    n, bad = gu.laser_audit(scan, other, (px3, py, h), (px4, py, h))
    assert n == 1 and not bad


def test_bounded_wait_logic_on_the_host(tmp_path):
    """csrc/cagym_spin.h's poll loop is a template over (load, pause): the same code, compiled by g++, must return true as soon as
    the counter arrives, false after exactly `limit` pauses when it never does, and look once more after the last pause."""
    exe = str(tmp_path / "spin_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "gym-exploration-2d_amd", "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "spin_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "spin_check ok" in out.stdout
