"""MORL/D on the HIP actor-critic engine (``multi_policy/morld/morld.py``).

The population's MOSAC learners are rows of ONE population engine (``ACEngine(population=pop_size)``): parameters,
target networks, Adam moments, step counters and entropy coefficients are ``[pop] ...`` device buffers and every
``Policy.wrapped`` works on its slice.  The reference's

    for i in range(update_passes):
        for p in population:
            if len(p.wrapped.get_buffer()) > 0 and p != current: p.wrapped.update()        (morld.py:423-433)

-- ``update_passes x (pop - 1)`` sequential, launch-bound updates -- becomes ``update_passes`` batched
``morl_ac_update`` calls per contiguous run of members: the sub-problems are independent, so the learner axis is just
more workgroups.  Per-member results are bit-identical to the sequential loop given the same batches and noise
(tests/test_ac_kernels_parity.py::test_population_batch_equals_independent_learners); the member batches are still
sampled one by one in population order so the host RNG stream is the reference's.

Not on the hot path and reduced here: weight initialisation uses the reference's ``equally_spaced_weights`` when
``morl_baselines`` (pymoo) is importable and a simplex lattice otherwise; the Tchebycheff scalarisation ("tch", pymoo)
is not available -- the engine scalarises with the weighted sum.
"""
from __future__ import annotations

import math
import os
import time
from typing import Callable, List, Optional, Tuple, Union

import numpy as np
import torch as th

from .acnets import noise_device, randn
from .ac_engine import ALGO_MOSAC, ALGO_SACD, ACEngine
from .api import MOAgent
from .mosac import MOSAC
from .mosac_discrete import MOSACDiscrete
from .native import NativeLib, load_library
from .pareto import ParetoArchive


def simplex_lattice_weights(dim: int, n: int) -> np.ndarray:
    """n weight vectors on the unit simplex: the first n points of the finest Das-Dennis lattice with >= n points,
    spread by striding (deterministic stand-in for pymoo's Riesz-energy directions used by ``equally_spaced_weights``)."""
    h = 1
    while math.comb(h + dim - 1, dim - 1) < n:
        h += 1
    pts = []

    def rec(prefix, left, k):
        if k == dim - 1:
            pts.append(prefix + [left])
            return
        for v in range(left + 1):
            rec(prefix + [v], left - v, k + 1)

    rec([], h, 0)
    pts = np.asarray(pts, dtype=np.float64) / h
    sel = np.round(np.linspace(0, len(pts) - 1, n)).astype(int)
    return pts[sel]


def _equally_spaced_weights(dim: int, n: int, seed: int) -> np.ndarray:
    try:
        from morl_baselines.common.weights import equally_spaced_weights  # needs pymoo
        return np.array(equally_spaced_weights(dim, n, seed))
    except Exception:
        return simplex_lattice_weights(dim, n)


class Policy:
    """``morld.py:30-45``: an individual of the population."""

    def __init__(self, id: int, weights: np.ndarray, wrapped: MOSAC):
        self.id = id
        self.weights = weights
        self.wrapped = wrapped


class MORLD(MOAgent):
    """MORL/D (Felten et al., JAIR 2024) with MOSAC sub-problem learners on one MI355X, or -- ``devices=[...]`` -- with the population
    split over several device contexts (BASELINE config 5: 64 decomposition weights over the 8 MI355X of a node; ``morld.py:423-433``
    updates the sub-problems independently, so the split needs no exchange: SURVEY 8(e) "replicas only")."""

    def __init__(self, env, scalarization_method: str = "ws", evaluation_mode: str = "ser", policy_name: str = "MOSAC",
                 policy_args: dict = {}, gamma: float = 0.995, pop_size: int = 6, seed: int = 42,
                 rng: Optional[np.random.Generator] = None, exchange_every: int = int(4e4), neighborhood_size: int = 1,
                 dist_metric: Callable[[np.ndarray, np.ndarray], float] = lambda a, b: np.sum(np.square(a - b)),
                 shared_buffer: bool = False, sharing_mechanism: List[str] = [], update_passes: int = 10,
                 weight_init_method: str = "uniform", weight_adaptation_method: Optional[str] = None,
                 project_name: str = "MORL-Baselines", experiment_name: str = "MORL-D",
                 wandb_entity: Optional[str] = None, log: bool = True, device: Union[th.device, str] = "auto",
                 weights: Optional[np.ndarray] = None, lib: Optional[NativeLib] = None,
                 devices: Optional[List[Union[th.device, str]]] = None):
        self.env = env
        super().__init__(self.env, device, seed=seed)
        self.gamma, self.seed = gamma, seed
        self.np_random = rng if rng is not None else np.random.default_rng(self.seed)
        if scalarization_method != "ws":
            # 'tch' has no behaviour to match: the reference hands common/scalarization.py:20-41's NUMPY thunk to MOSAC as
            # `scalarization` (morld.py:141-146, :198), where it is applied to batched torch tensors
            # (mosac_continuous_action.py:438-459) -- the thunk enumerates the batch rows into its reward_dim-long
            # best_so_far list and raises on the first update; every reference example and sweep uses 'ws'
            raise NotImplementedError("only the weighted-sum scalarisation ('ws') runs on the HIP engine (the reference's "
                                      "'tch' path does not run with MOSAC either: see the comment above this line)")
        if policy_name not in ("MOSAC", "MOSACDiscrete"):
            raise NotImplementedError("MOSAC / MOSACDiscrete are the sub-problem learners of the HIP engine")
        self.scalarization_method, self.scalarization = scalarization_method, np.dot
        self.evaluation_mode, self.pop_size = evaluation_mode, pop_size
        self.weight_init_method, self.weight_adaptation_method = weight_init_method, weight_adaptation_method
        self.delta = 0.1 if weight_adaptation_method == "PSA" else None
        if weights is not None:
            self.weights = np.asarray(weights, dtype=np.float64)
            assert self.weights.shape == (pop_size, self.reward_dim)
        elif weight_init_method == "uniform":
            self.weights = _equally_spaced_weights(self.reward_dim, pop_size, seed)
        elif weight_init_method == "random":
            from .envelope import random_weights
            self.weights = random_weights(self.reward_dim, n=pop_size, dist="dirichlet", rng=self.np_random)
        else:
            raise Exception("Unsupported weight init method.")
        self.exchange_every, self.shared_buffer, self.update_passes = exchange_every, shared_buffer, update_passes
        self.sharing_mechanism = sharing_mechanism
        self.transfer = "transfer" in sharing_mechanism
        self.neighborhood_size, self.dist_metric = neighborhood_size, dist_metric
        self.neighborhoods = [[] for _ in range(pop_size)]
        self.iteration = 0
        self.project_name, self.experiment_name, self.log = project_name, experiment_name, log
        self.policy_name, self.policy_args = policy_name, dict(policy_args)
        self.lib = lib or load_library()
        # ---- one engine per device context for its share of the population; members are slices of it ----------------------
        # ``devices`` = G contexts (BASELINE config 5: the 64 sub-problems over the 8 MI355X of a node): member i lives in context
        # g = i * G // pop_size -- contiguous shares of pop_size / G learners, each an ACEngine(population = share) on its device.
        # The sub-problems are independent (morld.py:423-433 updates them one after the other), so nothing is exchanged between
        # the contexts during updates: every device's launches are enqueued on its own stream and run side by side; the host only
        # merges the members' evaluations into the ParetoArchive (and copies a neighbour's actor across devices in _share).  One
        # context (the default: ``device``) is the single population engine of before; G contexts on ONE device give the same
        # learners bit for bit (tests/test_ac_agents.py).
        a = self.policy_args
        arch = a.get("net_arch", [256, 256])
        self.batch_size = a.get("batch_size", 128)
        D = int(np.prod(env.observation_space.shape))
        self.discrete = policy_name == "MOSACDiscrete"
        self.devices = [th.device(d) for d in devices] if devices else [th.device(self.device)]
        G = len(self.devices)
        if G > pop_size:
            raise ValueError(f"{G} device contexts for a population of {pop_size}")
        bounds = [k * pop_size // G for k in range(G + 1)]
        self._where = [None] * pop_size                  # member id -> (context, index inside its engine)
        self.engines = []
        for g, dev in enumerate(self.devices):
            n_g = bounds[g + 1] - bounds[g]
            for k in range(n_g):
                self._where[bounds[g] + k] = (g, k)
            if self.discrete:
                eng = ACEngine(ALGO_SACD, D, int(env.action_space.n), self.reward_dim, arch, action_low=0.0, action_high=1.0,
                               max_rows=self.batch_size, population=n_g, device=dev, lib=self.lib, device_steps=True)
            else:
                Ad = int(np.prod(env.action_space.shape))
                eng = ACEngine(ALGO_MOSAC, D, Ad, self.reward_dim, arch, action_low=np.asarray(env.action_space.low),
                               action_high=np.asarray(env.action_space.high), max_rows=self.batch_size, population=n_g,
                               device=dev, lib=self.lib, device_steps=True)
            self.engines.append(eng)
        self.engine = self.engines[0]                    # (the whole population when there is one context)
        learner = MOSACDiscrete if self.discrete else MOSAC
        self.current_policy = 0
        self.population = [
            Policy(id=i, weights=w,
                   wrapped=learner(id=i, env=self.env, weights=w, scalarization=th.matmul, gamma=gamma, log=self.log,
                                   seed=self.seed, parent_rng=self.np_random, device=self.devices[self._where[i][0]], lib=self.lib,
                                   engine=self.engines[self._where[i][0]].member(self._where[i][1]), **self.policy_args))
            for i, w in enumerate(self.weights)]
        self.archive = ParetoArchive(lib=self.lib, device=self.devices[0] if self.lib.is_device_build else None)
        self._update_neighborhoods()
        if self.log:
            self.setup_wandb(project_name=self.project_name, experiment_name=self.experiment_name, entity=wandb_entity)
        if self.shared_buffer:
            self._share_buffers()

    def get_config(self) -> dict:
        return {"env_id": self.env.unwrapped.spec.id, "scalarization_method": self.scalarization_method,
                "evaluation_mode": self.evaluation_mode, "gamma": self.gamma, "pop_size": self.pop_size,
                "exchange_every": self.exchange_every, "neighborhood_size": self.neighborhood_size,
                "shared_buffer": self.shared_buffer, "update_passes": self.update_passes, "transfer": self.transfer,
                "weight_init_method": self.weight_init_method, "weight_adapt_method": self.weight_adaptation_method,
                "delta_adapt": self.delta, "project_name": self.project_name, "experiment_name": self.experiment_name,
                "seed": self.seed, "log": self.log, "device": self.device, "devices": [str(d) for d in self.devices],
                "policy_name": self.policy_name,
                **self.population[0].wrapped.get_config(), **self.policy_args}

    # -- population plumbing (morld.py:245-330) ---------------------------------------------------------------------------
    def _share_buffers(self, neighborhood: bool = False):
        if neighborhood:
            for p in self.population:
                shared = p.wrapped.get_buffer()
                for n in self.neighborhoods[p.id]:
                    self.population[n].wrapped.set_buffer(shared)
        else:
            shared = self.population[0].wrapped.get_buffer()
            for p in self.population:
                p.wrapped.set_buffer(shared)

    def _select_candidate(self) -> Policy:
        candidate = self.population[self.current_policy]
        if self.current_policy + 1 == self.pop_size:
            self.iteration += 1
        self.current_policy = (self.current_policy + 1) % self.pop_size
        return candidate

    def _update_neighborhoods(self):
        """k closest weight vectors of every individual (``morld.py`` ``__update_neighborhoods``)."""
        self.neighborhoods = [[] for _ in range(self.pop_size)]
        if self.neighborhood_size <= 0:
            return
        ws = [p.weights for p in self.population]
        for i in range(self.pop_size):
            d = [(self.dist_metric(ws[i], ws[j]), j) for j in range(self.pop_size) if j != i]
            d.sort(key=lambda t: t[0])
            self.neighborhoods[i] = [j for _, j in d[: self.neighborhood_size]]

    def _eval_policy(self, policy: Policy, eval_env, num_eval_episodes_for_front: int) -> np.ndarray:
        acc = np.zeros(self.reward_dim)
        for _ in range(num_eval_episodes_for_front):
            # (each call is itself the mean of policy_eval's default 5 episodes, as in the reference: morld.py:280-290)
            _, _, _, disc = policy.wrapped.policy_eval(eval_env, weights=policy.weights, scalarization=self.scalarization,
                                                       log=self.log)
            acc += disc
        return acc / num_eval_episodes_for_front

    def _eval_all_policies(self, eval_env, num_eval_episodes_for_front: int, num_eval_weights_for_eval: int,
                           ref_point: np.ndarray, known_front=None):
        evals = []
        for agent in self.population:
            disc = self._eval_policy(agent, eval_env, num_eval_episodes_for_front)
            evals.append(disc)
            self.archive.add(agent, disc)            # non-dominated filtering: morl_pareto_mask on the device
        if self.log:
            from morl_baselines.common.evaluation import log_all_multi_policy_metrics
            log_all_multi_policy_metrics(self.archive.evaluations, ref_point, self.reward_dim, self.global_step,
                                         n_sample_weights=num_eval_weights_for_eval, ref_front=known_front)
        return evals

    def _share(self, last_trained: Policy):
        """``morld.py`` ``__share``: copy the trained actor to not-yet-trained neighbours (first sweep only)."""
        if self.transfer and self.iteration == 0:
            src = last_trained.id
            gs, ks = self._where[src]
            for n in self.neighborhoods[src]:
                if n > src:
                    gn, kn = self._where[n]
                    en = self.engines[gn]
                    en.pol[kn].copy_(self.engines[gs].pol[ks])    # (across devices when the neighbour lives in another context)
                    en.pol_exp_avg[kn].zero_()                    # the reference re-creates the neighbour's optimiser
                    en.pol_exp_avg_sq[kn].zero_()
                    en.pol_steps[kn] = 0
                    self.population[n].wrapped._p_step = 0

    def _adapt_weights(self, evals: List[np.ndarray]):
        """PSA weight adaptation (``morld.py`` ``__adapt_weights``)."""

        def closest_non_dominated(e: np.ndarray) -> Tuple[Optional[Policy], Optional[np.ndarray]]:
            best, who, ev = math.inf, None, None
            for cand_eval, cand in zip(self.archive.evaluations, self.archive.individuals):
                d = np.sum(np.square(e - cand_eval))
                if best > d > 0.01:
                    best, who, ev = d, cand, cand_eval
            return who, ev

        if self.weight_adaptation_method == "PSA":
            for i, p in enumerate(self.population):
                _, closest_eval = closest_non_dominated(evals[i])
                new_w = np.array(p.weights, dtype=np.float64)
                if closest_eval is not None:
                    for k in range(len(evals[i])):
                        new_w[k] = p.weights[k] * (1 + self.delta) if evals[i][k] >= closest_eval[k] \
                            else p.weights[k] / (1 + self.delta)
                normalized = new_w / np.linalg.norm(new_w, ord=1)
                p.wrapped.set_weights(normalized)
                p.weights = normalized

    # -- the batched hot path (morld.py:423-433) ----------------------------------------------------------------------------
    def _update_others(self, current: Policy):
        members = [p for p in self.population if len(p.wrapped.get_buffer()) > 0 and p is not current]
        if not members:
            return
        # contiguous runs of member ids -> one engine call each
        runs, start = [], 0
        ids = [p.id for p in members]
        for k in range(1, len(ids) + 1):
            if k == len(ids) or ids[k] != ids[k - 1] + 1:
                runs.append(ids[start:k])
                start = k
        e0 = self.engines[0]
        ref = members[0].wrapped

        def pieces(run):
            """A run of consecutive member ids cut at the context boundaries: (context, ids, offset inside the run)."""
            out, a0 = [], 0
            for k in range(1, len(run) + 1):
                if k == len(run) or self._where[run[k]][0] != self._where[run[a0]][0]:
                    out.append((self._where[run[a0]][0], run[a0:k], a0))
                    a0 = k
            return out

        for _ in range(self.update_passes):
            batches = {p.id: p.wrapped.update_inputs() for p in members}      # population order: reference RNG stream
            for run in runs:
                n, B = len(run), ref.batch_size
                cfg = ref.make_cfg()
                eps = None
                if not self.discrete:
                    # the run's noise is drawn ONCE, in population order, on the first context's device -- whatever the number of
                    # contexts, the learners see the draws a single population engine would have handed them
                    e = e0
                    pf = ref.policy_freq
                    if noise_device(e.q.device).type == "cpu":
                        # host generator: one call per tensor in the order of the reference's sequential updates (learner by
                        # learner: next-action noise, then per actor iteration pi and, with autotune, the alpha re-sample --
                        # mosac_continuous_action.py:436-468), so a seeded run consumes the stream exactly like the reference
                        eps = th.empty((1 + 2 * pf, n, B, e.Ad), dtype=th.float32)
                        for j in range(n):
                            eps[0, j].normal_()
                            if cfg.do_policy:
                                for k in range(pf):
                                    eps[1 + k, j].normal_()
                                    if ref.autotune:
                                        eps[1 + pf + k, j].normal_()
                        eps = eps.to(e.q.device)
                    else:
                        eps = randn((1 + 2 * pf, n, B, e.Ad), e.q.device)
                for g, ids, off in pieces(run):
                    e = self.engines[g]
                    stack = lambda k_: th.stack([batches[i][k_] for i in ids])  # noqa: E731
                    w = th.stack([self.population[i].wrapped.weights_tensor for i in ids])
                    first, m = self._where[ids[0]][1], len(ids)
                    if self.discrete:
                        e.update(cfg, obs=stack(0), actions=stack(1), rewards=stack(2), next_obs=stack(3), dones=stack(4), w=w,
                                 want=(), first=first, count=m)
                    else:
                        ep = eps[:, off:off + m].to(e.q.device) if (len(self.engines) > 1) else eps
                        pf = ref.policy_freq
                        e.update(cfg, obs=stack(0), actions=stack(1), rewards=stack(2), next_obs=stack(3), dones=stack(4), w=w,
                                 eps_next=ep[0].contiguous(), eps_pi=ep[1:1 + pf].contiguous(), eps_alpha=ep[1 + pf:].contiguous(),
                                 want=(), first=first, count=m)
                for i in run:
                    self.population[i].wrapped.note_update(True if self.discrete else bool(cfg.do_policy))

    # -- checkpoints ---------------------------------------------------------------------------------------------------------
    def save(self, save_dir="weights/", filename=None, save_replay_buffer=True):
        if not os.path.isdir(save_dir):
            os.makedirs(save_dir)
        filename = filename or "morld_save"
        saved = {}
        for i, policy in enumerate(self.population):
            saved[f"population_policy_{i}"] = policy.wrapped.get_save_dict(save_replay_buffer)
        for i, (policy, ev) in enumerate(zip(self.archive.individuals, self.archive.evaluations)):
            saved[f"archive_policy_{i}"] = policy.wrapped.get_save_dict(save_replay_buffer=False)
            saved[f"archive_policy_{i}_eval"] = ev
        th.save(saved, os.path.join(save_dir, filename + ".tar"))

    def load(self, path, load_replay_buffer=True):
        import copy
        params = th.load(path, map_location=self.device, weights_only=False)
        for i, policy in enumerate(self.population):
            key = f"population_policy_{i}"
            if key in params:
                policy.wrapped.load(params[key], load_replay_buffer=load_replay_buffer)
                policy.weights = policy.wrapped.weights
        self.archive.individuals, self.archive.evaluations = [], []
        i = 0
        while f"archive_policy_{i}" in params and f"archive_policy_{i}_eval" in params:
            ind = copy.deepcopy(self.population[0])
            ind.wrapped.load(params[f"archive_policy_{i}"], load_replay_buffer=False)
            ind.weights = ind.wrapped.weights
            self.archive.individuals.append(ind)
            self.archive.evaluations.append(params[f"archive_policy_{i}_eval"])
            i += 1

    def train(self, total_timesteps: int, eval_env, ref_point: np.ndarray, known_pareto_front=None,
              num_eval_episodes_for_front: int = 5, num_eval_weights_for_eval: int = 50,
              reset_num_timesteps: bool = False, checkpoints: bool = True, save_freq: int = 10000):
        """``morld.py:500-575``."""
        if self.log:
            self.register_additional_config({
                "total_timesteps": total_timesteps, "ref_point": ref_point.tolist(), "known_front": known_pareto_front,
                "num_eval_weights_for_eval": num_eval_weights_for_eval,
                "num_eval_episodes_for_front": num_eval_episodes_for_front})
        self.global_step = 0 if reset_num_timesteps else self.global_step
        self.num_episodes = 0 if reset_num_timesteps else self.num_episodes
        start_time = time.time()
        self.env.reset()
        self._eval_all_policies(eval_env, num_eval_episodes_for_front, num_eval_weights_for_eval, ref_point,
                                known_pareto_front)
        while self.global_step < total_timesteps:
            policy = self._select_candidate()
            policy.wrapped.train(self.exchange_every, eval_env=eval_env, start_time=start_time)
            self.global_step += self.exchange_every
            for p in self.population:
                p.wrapped.global_step = self.global_step
            self._update_others(policy)
            evals = self._eval_all_policies(eval_env, num_eval_episodes_for_front, num_eval_weights_for_eval,
                                            ref_point, known_pareto_front)
            self._share(policy)
            self._adapt_weights(evals)
            if checkpoints and self.global_step % save_freq == 0:
                self.save(filename=f"MORL-D step={self.global_step}", save_replay_buffer=False)
        self.close_wandb()
