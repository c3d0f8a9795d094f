"""AddressSanitizer + UBSan build of the CPU oracle (CPU build only: GPU ASan is not available on the pool).
Replays a mixed golden subset and the IG primitives in a subprocess with the sanitized library preloaded."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, os
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from oracle import oracle as orc
orc.LIB = os.path.join(%(root)r, "oracle", "libcagym_oracle_asan.so")
import numpy as np, importlib
import golden_util as gu
scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
for g in ("obstacles_laserscan", "static_mixes_perturbed", "action_maps"):
    for name, case in gu.load_cases(g).items():
        gu.replay(case, lambda **kw: orc.OracleEnv(**kw))
env = orc.OracleEnv(N=8, M=10, game_over_mode=1)
env.set_scenario(scen.random_worlds(8, 10, seed=5), scen.POLICY_RVO, scen.DYN_UNICYCLE, coop=np.full((8, 10), .5))
env.reset()
for _ in range(150):
    env.step()
env.ga3c_states()
env.run(40)
orc.generate_scenarios(60, 10, seed=3, n_min=2, n_max=10)
for pose in ((0.0, 0.0, 0.3), (14.9, -14.9, 2.0), (40.0, 7.0, -1.0)):
    orc.occupancy_grid(orc.rasterize([(2, 2, 10, 10), (-10, 2, -2, 10)]), *pose)
z = np.load(os.path.join(%(root)r, "tests", "golden", "ig_primitives.npz"))
edf, d2 = orc.edt(orc.rasterize(z["rects__obstacles"]))
for p in z["rects__vis_poses"][:10]:
    m = orc.visible_cells(edf, p)
    orc.rollout(np.ones((60, 60)), edf, p, m, m * 0, 4, 1, 0, 0)
print("sanitized run ok")
'''


def test_oracle_under_asan_ubsan():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "sanitized run ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
