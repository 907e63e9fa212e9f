"""Generate the golden fixtures by executing the UNMODIFIED reference in the build container.

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden.py

Needs /root/reference (read-only) and the import stubs of ``oracle/ref_harness.py``; it cannot run on the
GPU box, which is why its outputs (``tests/golden/*.npz``) are committed.  For each case of ``cases.py`` it
builds the reference's real ``Envelope`` agent, loads the seeded parameters / Adam state, pins the batch and
the sampled weights by wrapping ``replay_buffer.sample`` and ``random_weights`` (instance / module attribute
wrapping only -- no reference source is modified), runs ``agent.update()`` and records what came out.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import ref_harness as rh  # noqa: E402
from cases import CASES, FULL_SIZE, Case, make_inputs, pareto_sets  # noqa: E402


def run_case(ref, c: Case) -> dict:
    th.set_num_threads(1 if c.name != "flagship_full" else 8)
    inp = make_inputs(c)
    env = rh.FakeEnv(c.D, c.A, c.R)
    ag = ref.envelope.Envelope(
        env, learning_rate=c.lr, net_arch=list(c.arch), batch_size=c.B, gamma=c.gamma,
        max_grad_norm=c.max_grad_norm, envelope=c.envelope, num_sample_w=c.W, per=True, buffer_size=64,
        initial_homotopy_lambda=c.homotopy_lambda, log=False, seed=0, device="cpu",
        target_net_update_freq=10 ** 9,
    )
    names = [n for n, _ in ag.q_net.named_parameters()]
    ag.q_net.load_state_dict({n: th.tensor(a) for n, a in zip(names, inp["online"])})
    ag.target_q_net.load_state_dict({n: th.tensor(a) for n, a in zip(names, inp["target"])})
    if c.step > 1:
        for p, m, v in zip(ag.q_net.parameters(), inp["exp_avg"], inp["exp_avg_sq"]):
            ag.q_optim.state[p] = {"step": th.tensor(float(c.step - 1)), "exp_avg": th.tensor(m),
                                   "exp_avg_sq": th.tensor(v)}
    ag.global_step = 1  # never a multiple of target_net_update_freq -> no polyak inside this update

    # pin the batch and the sampled weights
    batch = tuple(th.tensor(inp[k]) for k in ("obs", "actions", "rewards", "next_obs", "dones")) + (
        th.arange(c.B),)
    ag.replay_buffer.sample = lambda *a, **k: batch
    rec = {}
    ag.replay_buffer.update_priorities = lambda idx, pr: rec.__setitem__("priority_final", np.asarray(pr).copy())
    ag.replay_buffer.min_priority = 0.125
    orig_rw = ref.envelope.random_weights
    ref.envelope.random_weights = lambda **k: inp["sampled_w"].copy()

    # record loss / pre-clip grads / norm without touching reference code
    F = ref.envelope.F
    orig_mse = F.mse_loss
    losses = []
    F.mse_loss = lambda a, b, *x, **k: (losses.append(orig_mse(a, b, *x, **k)) or losses[-1])
    orig_clip = th.nn.utils.clip_grad_norm_

    def clip_rec(params, max_norm, *a, **k):
        params = list(params)
        rec["grads_raw"] = [p.grad.detach().clone().numpy() for p in params]
        n = orig_clip(params, max_norm, *a, **k)
        rec["grad_norm"] = np.float32(n.item())
        return n

    th.nn.utils.clip_grad_norm_ = clip_rec
    # the targets as the reference computes them (same tensors update() will build)
    sw = th.tensor(inp["sampled_w"]).float()
    w = sw.repeat_interleave(c.B, 0)
    t_nobs = batch[3].repeat(c.W, 1)
    if c.envelope:
        rec["target"] = ag.envelope_target(t_nobs, w, sw).numpy().copy()
        if c.subsample >= 200:          # full-size cases: the arg-max indices of envelope.py:420-426, from the reference's own modules
            with th.no_grad():
                Wr = sw.repeat(t_nobs.size(0), 1)
                nob = t_nobs.repeat_interleave(c.W, 0)
                nq = ag.q_net(nob, Wr).view(t_nobs.size(0), c.W, c.A, c.R)
                scal = th.einsum("br,bwar->bwa", w, nq)
                max_q, ac = th.max(scal, dim=2)
                pref = th.argmax(max_q, dim=1)
                rec["pref"] = pref.numpy().astype(np.int16)
                rec["ac"] = ac.gather(1, pref.reshape(-1, 1)).squeeze(1).numpy().astype(np.int16)
    else:
        rec["target"] = ag.ddqn_target(t_nobs, w).numpy().copy()
    try:
        ag.update()
    finally:
        F.mse_loss = orig_mse
        th.nn.utils.clip_grad_norm_ = orig_clip
        ref.envelope.random_weights = orig_rw
    if c.homotopy_lambda > 0:
        lam = c.homotopy_lambda
        loss = (1 - lam) * losses[0] + lam * losses[1]
    else:
        loss = losses[0]
    params = list(ag.q_net.parameters())
    if "grads_raw" not in rec:  # max_grad_norm None: clip never called
        rec["grad_norm"] = np.float32(-1.0)
    s = c.subsample
    sub = (lambda a: a.reshape(-1)[::s].copy())
    out = dict(
        loss=np.float32(loss.item()), grad_norm=rec["grad_norm"], target=rec["target"],
        priority_final=rec["priority_final"].astype(np.float64),
    )
    if "pref" in rec:
        out["pref"], out["ac"] = rec["pref"], rec["ac"]
        out["target"] = rec["target"][::16].copy()      # every 16th TD row is enough at this size
    for i, p in enumerate(params):
        out[f"param_after_{i}"] = sub(p.detach().numpy())
        out[f"grad_{i}"] = sub(p.grad.detach().numpy())  # post-clip
        st = ag.q_optim.state[p]
        out[f"exp_avg_{i}"] = sub(st["exp_avg"].numpy())
        out[f"exp_avg_sq_{i}"] = sub(st["exp_avg_sq"].numpy())
        if "grads_raw" in rec:
            out[f"grad_raw_{i}"] = sub(rec["grads_raw"][i])
    return out


def per_trace(ref) -> dict:
    """A short PrioritizedReplayBuffer life: adds, samples, priority updates (``prioritized_buffer.py``)."""
    np.random.seed(123)
    rng = np.random.default_rng(5)
    buf = ref.prioritized_buffer.PrioritizedReplayBuffer((3,), 1, rew_dim=2, max_size=50, action_dtype=np.uint8)
    out = {}
    for t in range(70):  # wraps around the ring
        buf.add(rng.standard_normal(3), rng.integers(4), rng.standard_normal(2), rng.standard_normal(3), t % 9 == 0)
        if t % 10 == 9:
            state = np.random.get_state()
            idx = buf.tree.sample(16)
            np.random.set_state(state)
            u = np.random.random_sample(16)     # same stream as the np.random.uniform the tree just used
            pr = (rng.random(16) + 0.01) ** 0.6
            buf.update_priorities(idx, pr)
            out[f"idx_{t}"] = idx.astype(np.int64)
            out[f"u_{t}"] = u
            out[f"pr_{t}"] = pr
            out[f"root_{t}"] = np.float64(buf.tree.nodes[0][0])
            out[f"minp_{t}"] = np.float64(buf.min_priority)
    out["leaves"] = buf.tree.nodes[-1].copy()
    out["level3"] = buf.tree.nodes[3].copy()
    return out


TRAIN_CFG = dict(learning_rate=1e-3, net_arch=[64, 64], batch_size=32, gamma=0.95, num_sample_w=4, per=True,
                 buffer_size=10000, learning_starts=100, initial_epsilon=1.0, final_epsilon=0.05,
                 epsilon_decay_steps=6000, target_net_update_freq=100)
TRAIN_STEPS, TRAIN_CHUNKS, TRAIN_SEED = 12000, 4, 0
HV_REF = (-0.5, -20.0)


def train_trace(ref) -> dict:
    """Train the reference's Envelope agent on ``momdp.TreasureLine`` and record the hypervolume of its greedy front
    after each quarter of the run -- the "HV after equal gradient steps" anchor of tests/test_train_hv.py."""
    import momdp

    th.set_num_threads(1)
    np.random.seed(TRAIN_SEED)                      # PER sampling draws from the global numpy stream
    ref.envelope.equally_spaced_weights = lambda *a, **k: None   # pymoo (absent); its result is unused w/o eval_env
    env = momdp.TreasureLine(TRAIN_SEED)
    ag = ref.envelope.Envelope(env, log=False, seed=TRAIN_SEED, device="cpu", **TRAIN_CFG)
    out = {f"init_{i}": p.detach().numpy().copy() for i, p in enumerate(ag.q_net.parameters())}
    weights = momdp.equally_spaced_weights_2d(11)
    ev = momdp.TreasureLine(TRAIN_SEED)
    hv = []
    for chunk in range(TRAIN_CHUNKS):
        ag.train(total_timesteps=TRAIN_STEPS // TRAIN_CHUNKS, reset_num_timesteps=(chunk == 0))
        front = momdp.greedy_front(ag, ev, weights)
        hv.append(momdp.hypervolume_2d(front, HV_REF))
    out["hv"] = np.asarray(hv)
    out["front"] = front
    out["actions"] = np.asarray(env.action_log[:2000], dtype=np.int8)
    return out


def main():
    ref = rh.import_reference()
    if "--train-only" in sys.argv:
        tr = train_trace(ref)
        np.savez_compressed(os.path.join(HERE, "train_trace.npz"), **tr)
        print("train trace: HV per quarter", tr["hv"])
        return
    only_new = "--only-new" in sys.argv       # add fixtures of new full-size cases without rewriting the committed ones
    for c in ([] if only_new else CASES):
        out = run_case(ref, c)
        np.savez_compressed(os.path.join(HERE, f"envelope_{c.name}.npz"), **out)
        print(f"{c.name}: loss={out['loss']:.6g} grad_norm={out['grad_norm']:.6g}")
    th.set_num_threads(8)
    for big in FULL_SIZE:
        if only_new and os.path.exists(os.path.join(HERE, f"envelope_{big.name}.npz")):
            continue
        out = run_case(ref, big)
        np.savez_compressed(os.path.join(HERE, f"envelope_{big.name}.npz"), **out)
        print(f"{big.name}: loss={out['loss']:.6g} grad_norm={out['grad_norm']:.6g}")
    if only_new:
        return
    masks = {}
    for name, pts in pareto_sets().items():
        for rd in (True, False):
            m = ref.pareto.get_non_pareto_dominated_inds(pts, remove_duplicates=rd)
            masks[f"{name}__rd{int(rd)}"] = np.atleast_1d(m).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "pareto_masks.npz"), **masks)
    np.savez_compressed(os.path.join(HERE, "per_trace.npz"), **per_trace(ref))
    # schedule + weights + huber known answers
    misc = {}
    misc["lin_decay"] = np.array([ref.utils.linearly_decaying_value(1.0, 50000, s, 100, 0.05)
                                  for s in (0, 100, 101, 25000, 50100, 90000)], dtype=np.float64)
    x = th.tensor(np.random.default_rng(9).random(1000).astype(np.float32) * 0.03)
    misc["huber"] = np.float32(ref.networks.huber(x, 0.01).item())
    misc["huber_x"] = x.numpy()
    np.savez_compressed(os.path.join(HERE, "misc.npz"), **misc)
    tr = train_trace(ref)
    np.savez_compressed(os.path.join(HERE, "train_trace.npz"), **tr)
    print("train trace: HV per quarter", tr["hv"])
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
