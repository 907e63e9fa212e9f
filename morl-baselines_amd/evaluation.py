"""Evaluation roll-outs (``common/evaluation.py:23-67, 118-144``), sequential and in lock-step.

The reference evaluates one weight vector at a time: ``policy_evaluation_mo`` runs ``rep`` episodes, every step one
``agent.eval(obs, w)`` -- with the networks on a GPU that is one tiny launch chain plus a host synchronisation per
environment step, 100 weights x 5 episodes every ``eval_freq``.  ``policy_evaluation_mo_batched`` steps one
environment per weight in lock-step and asks the agent for all actions of a step at once (``agent.eval_batch``: one
pass of ``morl_gpi_actions_rows`` / ``morl_qnet_forward`` + ``morl_envelope_reduce_rows`` / ``morl_ac_policy_forward``
over the live rows).  Returns are accumulated exactly like ``eval_mo`` (same dtype, same order), so for deterministic
environments -- or environment copies seeded like the sequential run -- the results are identical to calling the
reference's ``policy_evaluation_mo`` per weight.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

EvalResult = Tuple[float, float, np.ndarray, np.ndarray]


def eval_mo(agent, env, w: Optional[np.ndarray] = None, scalarization=np.dot, render: bool = False) -> EvalResult:
    """One greedy episode (``evaluation.py:23-67``)."""
    obs, _ = env.reset()
    done = False
    vec_return, disc_vec_return = np.zeros_like(w), np.zeros_like(w)
    gamma = 1.0
    while not done:
        if render:
            env.render()
        obs, r, terminated, truncated, _ = env.step(agent.eval(obs, w))
        done = terminated or truncated
        vec_return += r
        disc_vec_return += gamma * r
        gamma *= agent.gamma
    if w is None:
        return scalarization(vec_return), scalarization(disc_vec_return), vec_return, disc_vec_return
    return scalarization(w, vec_return), scalarization(w, disc_vec_return), vec_return, disc_vec_return


def _mean_of(evals: Sequence[EvalResult]) -> EvalResult:
    return (np.mean([e[0] for e in evals]), np.mean([e[1] for e in evals]), np.mean([e[2] for e in evals], axis=0),
            np.mean([e[3] for e in evals], axis=0))


def policy_evaluation_mo(agent, env, w: np.ndarray, scalarization=np.dot, rep: int = 5) -> EvalResult:
    """Average of ``rep`` episodes (``evaluation.py:118-144``)."""
    return _mean_of([eval_mo(agent=agent, env=env, w=w, scalarization=scalarization) for _ in range(rep)])


def _actions_of(agent, obs: List[np.ndarray], ws: List[np.ndarray]):
    if hasattr(agent, "eval_batch"):
        return agent.eval_batch(np.stack(obs), np.stack(ws))
    return [agent.eval(o, w) for o, w in zip(obs, ws)]         # agents without a batched action pass: one row at a time


def eval_mo_batched(agent, envs: Sequence, weights: Sequence[np.ndarray], scalarization=np.dot) -> List[EvalResult]:
    """One episode per (env, weight) pair, all stepped in lock-step; entry i equals ``eval_mo(agent, envs[i], weights[i])``."""
    if len(envs) != len(weights):
        raise ValueError(f"{len(envs)} environments for {len(weights)} weight vectors")
    n = len(envs)
    obs = [env.reset()[0] for env in envs]
    vec = [np.zeros_like(w) for w in weights]
    disc = [np.zeros_like(w) for w in weights]
    gamma = [1.0] * n
    live = list(range(n))
    while live:
        actions = _actions_of(agent, [obs[i] for i in live], [weights[i] for i in live])
        still = []
        for a, i in zip(actions, live):
            obs[i], r, terminated, truncated, _ = envs[i].step(a)
            vec[i] += r
            disc[i] += gamma[i] * r
            gamma[i] *= agent.gamma
            if not (terminated or truncated):
                still.append(i)
        live = still
    return [(scalarization(w, v), scalarization(w, d), v, d) for w, v, d in zip(weights, vec, disc)]


def policy_evaluation_mo_batched(agent, envs: Sequence, weights: Sequence[np.ndarray], scalarization=np.dot,
                                 rep: int = 5) -> List[EvalResult]:
    """``policy_evaluation_mo`` for every weight at once: ``rep`` rounds of lock-step episodes, one environment per
    weight (``envs[i]`` plays all ``rep`` episodes of ``weights[i]``, like the reference's single env does per weight)."""
    rounds = [eval_mo_batched(agent, envs, weights, scalarization) for _ in range(rep)]
    return [_mean_of([rounds[k][i] for k in range(rep)]) for i in range(len(weights))]


def evaluate_front(agent, make_env: Callable[[], object], weights: Sequence[np.ndarray], rep: int = 5,
                   scalarization=np.dot) -> List[np.ndarray]:
    """Discounted vector returns of every evaluation weight -- the ``current_front`` list the training loops hand to
    ``log_all_multi_policy_metrics`` (e.g. ``envelope.py:545-557``), from lock-step roll-outs on fresh environments."""
    envs = [make_env() for _ in weights]
    return [res[3] for res in policy_evaluation_mo_batched(agent, envs, list(weights), scalarization, rep)]


def front_returns(agent, eval_env, eval_weights: Sequence[np.ndarray], rep: int = 5) -> List[np.ndarray]:
    """The ``current_front`` list of the training loops.  ``eval_env`` a single environment: the reference's loop, one
    weight after the other.  A list / tuple of environments (at least one per weight): lock-step roll-outs."""
    if isinstance(eval_env, (list, tuple)):
        if len(eval_env) < len(eval_weights):
            raise ValueError(f"{len(eval_env)} evaluation environments for {len(eval_weights)} weights")
        res = policy_evaluation_mo_batched(agent, list(eval_env[:len(eval_weights)]), list(eval_weights), rep=rep)
        return [r[3] for r in res]
    return [policy_evaluation_mo(agent, eval_env, ew, rep=rep)[3] for ew in eval_weights]


def multi_policy_metrics(current_front: Sequence[np.ndarray], hv_ref_point: np.ndarray, weights_set: Sequence[np.ndarray],
                         ref_front: Optional[Sequence[np.ndarray]] = None, mul_weights: Optional[np.ndarray] = None,
                         lib=None, device=None) -> dict:
    """The numbers ``log_all_multi_policy_metrics`` (``evaluation.py:147-198``) logs, as a dict and without pymoo / wandb:
    Pareto filter (``morl_pareto_mask``), hypervolume and expected utility on the device (``morl_hypervolume``,
    ``morl_expected_utility``), cardinality; with a known front also IGD and the maximum utility loss.  The weight sets are
    the caller's (the reference draws them with pymoo's Riesz s-energy directions)."""
    from . import performance_indicators as pi
    from .pareto import filter_pareto_dominated

    front = list(filter_pareto_dominated(current_front, lib=lib, device=device))
    out = {"eval/hypervolume": pi.hypervolume(hv_ref_point, front, lib=lib, device=device),
           "eval/eum": pi.expected_utility(front, list(weights_set), lib=lib, device=device),
           "eval/cardinality": pi.cardinality(front), "front": front}
    if ref_front is not None:
        out["eval/igd"] = pi.igd(known_front=list(ref_front), current_estimate=front)
        out["eval/mul"] = pi.maximum_utility_loss(front=front, reference_set=list(ref_front),
                                                  weights_set=weights_set if mul_weights is None else mul_weights)
    return out
