"""The drop-in claim, held by a test: with the reference package importable (here: the unmodified tree at /root/reference
behind the import stubs of oracle/ref_harness.py), the HIP agents ARE subclasses of the reference's own
``MOAgent`` / ``MOPolicy`` (``common/morl_algorithm.py:23-337``), train through ``update()``, and exchange checkpoints with the
reference's agents in both directions (``envelope.py:230-261``; ``capql.py`` save / load).  The package decides its base
classes at import time, so the scenario runs in a fresh interpreter (emulated kernel library: no GPU needed).
Skipped where the reference tree is absent (the GPU box)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_harness  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_harness.reference_available(), reason="reference tree not mounted")

SCRIPT = r'''
import os, sys, tempfile
ROOT = sys.argv[1]
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np, torch as th
import ref_harness
ref_harness.install_stubs()                      # gymnasium / wandb / pymoo stand-ins, /root/reference on sys.path
from morl_baselines.common.morl_algorithm import MOAgent as RefMOAgent, MOPolicy as RefMOPolicy
import simlib
import morl_baselines_amd.native as native
lib = simlib.load_sim()
native.use_library(lib)
import morl_baselines_amd.api as api
assert api.HAVE_REFERENCE_API, "the reference's base classes were not picked up"
assert api.MOAgent is RefMOAgent and api.MOPolicy is RefMOPolicy
from morl_baselines_amd.envelope import Envelope
from test_host_api import _fill

cpu = th.device("cpu")
# ---- Envelope ------------------------------------------------------------------------------------------------------------
env = ref_harness.FakeEnv(obs_dim=6, n_actions=3, reward_dim=2)     # spaces of the (stubbed) gymnasium types
env.D, env.A, env.R = 6, 3, 2
th.manual_seed(0); np.random.seed(0)
ag = Envelope(env, net_arch=[32, 32], batch_size=8, num_sample_w=4, buffer_size=256, per=True, learning_starts=0, log=False,
              seed=0, device=cpu, lib=lib)
assert isinstance(ag, RefMOAgent) and isinstance(ag, RefMOPolicy)
_fill(ag.replay_buffer, 64, env.D, env.A, env.R)
ag.global_step = 5
for _ in range(3):
    ag.update(); ag.global_step += 1
assert np.isfinite(ag.last_loss())
tmp = tempfile.mkdtemp()
ag.save(save_replay_buffer=False, save_dir=tmp, filename="hip")
from morl_baselines.multi_policy.envelope.envelope import Envelope as RefEnvelope
ref = RefEnvelope(env, net_arch=[32, 32], batch_size=8, num_sample_w=4, buffer_size=256, per=True, learning_starts=0,
                  log=False, seed=1, device=cpu)
ref.load(os.path.join(tmp, "hip.tar"), load_replay_buffer=False)      # HIP checkpoint -> reference agent
ours = ag.q_net.state_dict()
for k, v in ref.q_net.state_dict().items():
    assert th.equal(v, ours[k].cpu()), k
assert len(ref.q_optim.state_dict()["state"]) == len(ours)              # Adam moments travelled too
for p_ref, (name, _) in zip(ref.q_net.parameters(), ref.q_net.named_parameters()):
    st = ref.q_optim.state[p_ref]
    assert int(float(st["step"])) == 3 and float(st["exp_avg"].abs().sum()) > 0.0
# the reference keeps training from it
ref.replay_buffer = None
# reverse direction: reference checkpoint -> HIP agent
th.manual_seed(3)
ref2 = RefEnvelope(env, net_arch=[32, 32], batch_size=8, num_sample_w=4, buffer_size=256, per=True, learning_starts=0,
                   log=False, seed=2, device=cpu)
ref2.save(save_replay_buffer=False, save_dir=tmp, filename="ref")
ag.load(os.path.join(tmp, "ref.tar"), load_replay_buffer=False)
want = ref2.q_net.state_dict()
for k, v in ag.q_net.state_dict().items():
    assert th.equal(v.cpu(), want[k]), k
for k, v in ag.target_q_net.state_dict().items():
    assert th.equal(v.cpu(), want[k]), k
ag.update()                                                            # and the HIP agent trains on from it
print("envelope drop-in ok")

# ---- one actor-critic agent: CAPQL ---------------------------------------------------------------------------------------
from morl_baselines_amd.capql import CAPQL
from morl_baselines.multi_policy.capql.capql import CAPQL as RefCAPQL
cenv = ref_harness.FakeEnv(obs_dim=5, reward_dim=2, act_dim=2, env_id="fake-hopper-v0")
th.manual_seed(0); np.random.seed(0)
cap = CAPQL(cenv, net_arch=[32, 32], batch_size=8, buffer_size=128, learning_starts=0, log=False, seed=0, device=cpu, lib=lib)
assert isinstance(cap, RefMOAgent) and isinstance(cap, RefMOPolicy)
rng = np.random.default_rng(0)
for _ in range(32):
    w = rng.dirichlet(np.ones(2)).astype(np.float32)
    cap.replay_buffer.push(rng.standard_normal(5).astype(np.float32), rng.uniform(-1, 1, 2).astype(np.float32), w,
                           rng.standard_normal(2).astype(np.float32), rng.standard_normal(5).astype(np.float32), False)
cap.update()
cap.save(save_dir=tmp, filename="cap_hip", save_replay_buffer=False)
rcap = RefCAPQL(cenv, net_arch=[32, 32], batch_size=8, buffer_size=128, learning_starts=0, log=False, seed=5, device=cpu)
rcap.load(os.path.join(tmp, "cap_hip.tar"), load_replay_buffer=False)
for mine, theirs in ((cap.policy, rcap.policy), (cap.q_nets[0], rcap.q_nets[0]), (cap.q_nets[1], rcap.q_nets[1])):
    a, b = mine.state_dict(), theirs.state_dict()
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert th.equal(a[k].cpu(), b[k]), k
print("capql drop-in ok")
'''


def test_hip_agents_are_reference_subclasses_and_exchange_checkpoints():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT, ROOT], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    assert "envelope drop-in ok" in r.stdout and "capql drop-in ok" in r.stdout
