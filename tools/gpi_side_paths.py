"""Development timing of the GPI-PD side paths on one MI355X (not a bench line): the buffer-wide priority reset
(gpi_pd.py:619-660), a Dyna roll-out (gpi_pd.py:367-414) and a dynamics fit, at reference-like sizes (minecart shapes)."""
import os
import sys
import time
import types

import numpy as np
import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import momdp  # noqa: E402
from morl_baselines_amd.gpi_pd import GPIPD  # noqa: E402


class Env(momdp.TreasureLine):
    """minecart-shaped stand-in: 7 observations, 6 actions, 3 objectives."""

    def __init__(self):
        super().__init__(0, env_id="mo-minecart-like-v0")
        self.observation_space = momdp.BoxSpace(-1.0, 1.0, (7,), 0)
        self.action_space = momdp.DiscreteSpace(6, 0)
        self.reward_space = momdp.BoxSpace(-1.0, 1.0, (3,), 0)
        self.reward_dim = 3


def main():
    n_buf, n_sup = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, 16
    dev = th.device("cuda:0")
    th.manual_seed(0); np.random.seed(0)
    ag = GPIPD(Env(), log=False, seed=0, device=dev, buffer_size=n_buf, dynamics_rollout_batch_size=25000,
               dynamics_buffer_size=200000, max_support=32, dynamics_max_rows=10000)
    rng = np.random.default_rng(0)
    b = ag.replay_buffer
    obs = rng.uniform(-1, 1, (n_buf, 7)).astype(np.float32)
    b.add_batch(obs, rng.integers(0, 6, n_buf), rng.standard_normal((n_buf, 3)).astype(np.float32),
                obs + 0.05 * rng.standard_normal((n_buf, 7)).astype(np.float32), rng.random(n_buf) < 0.02) \
        if hasattr(b, "add_batch") else None
    sup = [w.astype(np.float32) for w in rng.dirichlet(np.ones(3), n_sup)]
    ag.set_weight_support(sup)
    w = th.tensor(sup[0])
    for name, fn in (("reset_priorities", lambda: ag._reset_priorities(w)),):
        fn(); th.cuda.synchronize()
        t = time.perf_counter(); fn(); th.cuda.synchronize()
        print(f"{name}: {n_buf} records x |M|={n_sup}: {(time.perf_counter() - t) * 1e3:.1f} ms")
    X = np.hstack([obs, np.eye(6, dtype=np.float32)[rng.integers(0, 6, n_buf)]])
    Y = (0.05 * rng.standard_normal((n_buf, 10))).astype(np.float32)
    t = time.perf_counter()
    ag.dynamics.fit(X[:20000], Y[:20000], max_epochs=5)
    th.cuda.synchronize()
    print(f"dynamics.fit 20000 samples x 5 epochs: {(time.perf_counter() - t) * 1e3:.1f} ms")
    ag.global_step = 10 ** 6
    ag._rollout_dynamics(w)
    th.cuda.synchronize()
    t = time.perf_counter()
    ag._rollout_dynamics(w)
    th.cuda.synchronize()
    print(f"rollout 25000 states x |M|={n_sup}: {(time.perf_counter() - t) * 1e3:.1f} ms, {ag._last_rollout}")
    if os.environ.get("PROFILE"):
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        ag._rollout_dynamics(w)
        th.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)


if __name__ == "__main__":
    main()
