"""Probabilistic dynamics ensemble of GPI-PD's Dyna part on the HIP engine
(``common/model_based/probabilistic_ensemble.py``) and the model-as-environment wrapper (``model_based/utils.py``).

Same constructor arguments and methods as the reference's ``ProbabilisticEnsemble`` (``forward``, ``sample``, ``fit``,
``save`` / ``load``, ``elites``, ``inputs_mu`` / ``inputs_sigma``, ``max_logvar`` / ``min_logvar``).  ``fit`` keeps the
reference's host-side data handling -- holdout split, bootstrap indices and per-epoch row shuffles from the global numpy
RNG, early stopping, elite selection -- and runs every optimiser step as ONE ``morl_ens_train_step`` call (forward of
all members, Gaussian NLL, backward, Adam with the per-layer weight decay, the step of the log-variance bounds); the
training set lives on the device and each step only uploads the (E, batch) index block.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch as th
from torch import nn

from . import native
from .acnets import randn
from .native import EnsCfg, EnsDesc, NativeLib

DECAYS = (0.000025, 0.00005, 0.000075, 0.000075, 0.0001)      # probabilistic_ensemble.py:205


class ProbabilisticEnsemble(nn.Module):
    def __init__(self, input_dim, output_dim, ensemble_size=5, arch=[200, 200, 200, 200], activation=None,
                 learning_rate=0.001, num_elites=2, normalize_inputs=True, device="auto", lib: Optional[NativeLib] = None,
                 max_rows: int = 10000):
        super().__init__()
        if activation is not None and activation is not th.nn.functional.relu:
            raise NotImplementedError("the HIP ensemble uses ReLU (the reference default)")
        self.ensemble_size, self.input_dim, self.output_dim = ensemble_size, input_dim, output_dim * 2
        self.arch, self.num_elites, self.normalize_inputs = list(arch), num_elites, normalize_inputs
        self.elites = [i for i in range(ensemble_size)]
        self.learning_rate = learning_rate
        self.device = (th.device("cuda") if th.cuda.is_available() else th.device("cpu")) if device == "auto" \
            else th.device(device)
        self.lib = lib or native.load_library()
        d = EnsDesc()
        d.input_dim, d.output_dim, d.n_hidden, d.ensemble_size, d.max_rows = input_dim, output_dim, len(arch), \
            ensemble_size, max_rows
        for i, h in enumerate(arch):
            d.hidden[i] = int(h)
        self.desc, self.max_rows, self._O = d, max_rows, output_dim
        self.Pm = int(self.lib.lib.morl_ens_param_count(C.byref(d)))
        if self.Pm < 0:
            self.lib.check(-1)
        h = C.c_void_p()
        self.lib.check(self.lib.lib.morl_ens_create(C.byref(h), C.byref(d)))
        self._h = h.value
        E = ensemble_size
        z = lambda *s: th.zeros(*s, dtype=th.float32, device=self.device)  # noqa: E731
        self.flat, self.exp_avg, self.exp_avg_sq = z(E, self.Pm), z(E, self.Pm), z(E, self.Pm)
        # reference initialisation (EnsembleLayer.__init__, :16-19): orthogonal on the 3-D (E, in, out) tensor, zero biases
        dims = [input_dim] + list(arch) + [self.output_dim]
        gain = nn.init.calculate_gain("relu")
        for l, (w, b) in enumerate(self._layer_views(self.flat)):
            W = th.empty((E, dims[l], dims[l + 1]))
            nn.init.orthogonal_(W, gain=gain)
            w.copy_(W.transpose(1, 2))
        self.bounds = th.cat([th.ones(output_dim) / 2.0, -th.ones(output_dim) * 10.0]).to(self.device)   # max | min logvar
        self.bounds_m, self.bounds_v = z(2 * output_dim), z(2 * output_dim)
        self.inputs_mu = th.zeros((1, input_dim), device=self.device)
        self.inputs_sigma = th.zeros((1, input_dim), device=self.device)
        self._adam_step = 0
        self.lib.check_device(self.flat)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.lib.morl_ens_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- parameter views ------------------------------------------------------------------------------------------------
    def _layer_views(self, buf: th.Tensor):
        """[(W (E, out, in), b (E, out))] per layer: the engine keeps nn.Linear orientation (the reference's W[e]^T)."""
        dims = [self.input_dim] + self.arch + [self.output_dim]
        out, o = [], 0
        for l in range(len(dims) - 1):
            n = dims[l + 1] * dims[l]
            w = buf[:, o:o + n].view(self.ensemble_size, dims[l + 1], dims[l])
            o += n
            b = buf[:, o:o + dims[l + 1]]
            o += dims[l + 1]
            out.append((w, b))
        assert o == self.Pm
        return out

    @property
    def max_logvar(self) -> th.Tensor:
        return self.bounds[:self._O].view(1, -1)

    @property
    def min_logvar(self) -> th.Tensor:
        return self.bounds[self._O:].view(1, -1)

    def state_dict(self, *a, **k):
        """The reference's keys and layouts: ``layers.<l>.W`` (E, in, out), ``layers.<l>.b`` (E, 1, out), bounds, stats."""
        sd = {}
        for l, (w, b) in enumerate(self._layer_views(self.flat)):
            sd[f"layers.{l}.W"] = w.transpose(1, 2).contiguous().clone()
            sd[f"layers.{l}.b"] = b.unsqueeze(1).clone()
        if self.normalize_inputs:
            sd["inputs_mu"], sd["inputs_sigma"] = self.inputs_mu.clone(), self.inputs_sigma.clone()
        sd["max_logvar"], sd["min_logvar"] = self.max_logvar.clone(), self.min_logvar.clone()
        return sd

    def load_state_dict(self, sd, *a, **k):
        with th.no_grad():
            for l, (w, b) in enumerate(self._layer_views(self.flat)):
                w.copy_(sd[f"layers.{l}.W"].to(self.device).transpose(1, 2))
                b.copy_(sd[f"layers.{l}.b"].to(self.device).squeeze(1))
            if "inputs_mu" in sd:
                self.inputs_mu = sd["inputs_mu"].to(self.device).float().reshape(1, -1).clone()
                self.inputs_sigma = sd["inputs_sigma"].to(self.device).float().reshape(1, -1).clone()
            self.bounds[:self._O].copy_(sd["max_logvar"].to(self.device).reshape(-1))
            self.bounds[self._O:].copy_(sd["min_logvar"].to(self.device).reshape(-1))

    def save(self, path):
        os.makedirs("weights/", exist_ok=True)
        th.save({"ensemble_state_dict": self.state_dict()}, path + ".tar")

    def load(self, path):
        self.load_state_dict(th.load(path, weights_only=False)["ensemble_state_dict"])

    # -- inference -----------------------------------------------------------------------------------------------------------
    def _norm_ptrs(self):
        if not self.normalize_inputs:
            return None, None
        return self.inputs_mu.data_ptr(), self.inputs_sigma.data_ptr()

    @th.no_grad()
    def predict(self, x: th.Tensor, per_member: bool = False):
        """(mean, logvar), each (E, rows, out): ``forward(..., deterministic=True, return_dist=True)`` (:88-121)."""
        x = x.to(self.device, th.float32).contiguous()
        rows = x.shape[-2] if x.dim() >= 2 else 1
        mean = th.empty((self.ensemble_size, rows, self._O), dtype=th.float32, device=self.device)
        logvar = th.empty_like(mean)
        mu, sg = self._norm_ptrs()
        self.lib.check_device(x)
        self.lib.check(self.lib.lib.morl_ens_forward(self._h, self.flat.data_ptr(), self.bounds.data_ptr(), mu, sg,
                                                     x.data_ptr(), int(per_member), rows, mean.data_ptr(),
                                                     logvar.data_ptr(), self.lib.stream_of(self.flat)))
        return mean, logvar

    def forward(self, input, deterministic=False, return_dist=False):
        dim = input.dim()
        x = input.reshape(1, -1) if dim == 1 else input
        mean, logvar = self.predict(x, per_member=(dim == 3))
        if dim == 1:
            mean, logvar = mean.squeeze(1), logvar.squeeze(1)
        if deterministic:
            return (mean, logvar) if return_dist else mean
        std = th.exp(0.5 * logvar)
        samples = mean + std * randn(std.shape, std.device)
        return (samples, mean, logvar) if return_dist else samples

    def sample(self, input, deterministic=False):
        """``probabilistic_ensemble.py:131-154`` (elite choice from the global numpy RNG, ensemble-variance uncertainty)."""
        if not deterministic:
            samples, means, logvar = self.forward(input, deterministic=False, return_dist=True)
            samples = samples.detach().cpu().numpy()
        else:
            means, logvar = self.forward(input, deterministic=True, return_dist=True)
        means, logvar = means.detach().cpu().numpy(), logvar.detach().cpu().numpy()
        vars_ = np.exp(logvar)
        _, batch_size, _ = means.shape
        batch_inds = np.arange(0, batch_size)
        model_inds = np.random.choice(self.elites, size=batch_size)
        mean_ensemble = means.mean(axis=0)
        var_ensemble = (means ** 2 + vars_).mean(axis=0) - mean_ensemble ** 2
        uncertainties = np.sqrt(var_ensemble + 1e-12).sum(-1)
        if deterministic:
            return means[model_inds, batch_inds], vars_[model_inds, batch_inds], uncertainties
        return samples[model_inds, batch_inds], vars_[model_inds, batch_inds], uncertainties

    # -- training (probabilistic_ensemble.py:178-290) -------------------------------------------------------------------------
    def _fit_input_stats(self, data):
        mu = np.mean(data, axis=0, keepdims=True)
        sigma = np.std(data, axis=0, keepdims=True)
        sigma[sigma < 1e-12] = 1.0
        self.inputs_mu = th.tensor(mu).to(self.device).float()
        self.inputs_sigma = th.tensor(sigma).to(self.device).float()

    def train_step(self, x: th.Tensor, y: th.Tensor, want_loss: bool = False):
        """One optimiser step on the members' batches x (E, rows, in), y (E, rows, out)."""
        self._adam_step += 1
        cfg = EnsCfg()
        cfg.lr, cfg.beta1, cfg.beta2, cfg.eps, cfg.adam_step = self.learning_rate, 0.9, 0.999, 1e-8, self._adam_step
        for l in range(len(self.arch) + 1):
            cfg.weight_decay[l] = self.decays[l]
        loss = th.zeros(1, dtype=th.float32, device=self.device) if want_loss else None
        mu, sg = self._norm_ptrs()
        self.lib.check_device(x, y)
        self.lib.check(self.lib.lib.morl_ens_train_step(
            self._h, self.flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.bounds.data_ptr(),
            self.bounds_m.data_ptr(), self.bounds_v.data_ptr(), mu, sg, x.data_ptr(), y.data_ptr(), x.shape[1],
            C.byref(cfg), None if loss is None else loss.data_ptr(), self.lib.stream_of(self.flat)))
        return loss

    def holdout_mse(self, x: th.Tensor, y: th.Tensor) -> th.Tensor:
        out = th.empty(self.ensemble_size, dtype=th.float32, device=self.device)
        mu, sg = self._norm_ptrs()
        self.lib.check(self.lib.lib.morl_ens_mse(self._h, self.flat.data_ptr(), self.bounds.data_ptr(), mu, sg, x.data_ptr(),
                                                 y.data_ptr(), x.shape[0], out.data_ptr(), self.lib.stream_of(self.flat)))
        return out

    def fit(self, X, Y, batch_size=256, holdout_ratio=0.1, max_holdout_size=5000, max_epochs_no_improvement=5,
            max_epochs=200):
        if self.normalize_inputs:
            self._fit_input_stats(X)
        self.decays = list(DECAYS)
        if len(self.arch) + 1 > len(self.decays):
            raise ValueError("the reference defines weight decays for at most 5 layers")
        # the reference builds a fresh Adam in every fit(): moments and step counter restart
        self.exp_avg.zero_(); self.exp_avg_sq.zero_(); self.bounds_m.zero_(); self.bounds_v.zero_()
        self._adam_step = 0
        num_holdout = min(int(X.shape[0] * holdout_ratio), max_holdout_size)
        permutation = np.random.permutation(X.shape[0])
        tr, ho = permutation[num_holdout:], permutation[:num_holdout]
        inputs = th.from_numpy(np.ascontiguousarray(X[tr])).to(self.device).float()
        targets = th.from_numpy(np.ascontiguousarray(Y[tr])).to(self.device).float()
        holdout_inputs = th.from_numpy(np.ascontiguousarray(X[ho])).to(self.device).float()
        holdout_targets = th.from_numpy(np.ascontiguousarray(Y[ho])).to(self.device).float()
        if batch_size > self.max_rows or num_holdout > self.max_rows:
            raise ValueError(f"batch / holdout larger than the engine's max_rows={self.max_rows}")
        idxs = np.random.randint(inputs.shape[0], size=[self.ensemble_size, inputs.shape[0]])
        num_batches = int(np.ceil(idxs.shape[-1] / batch_size))

        def shuffle_rows(arr):
            order = np.argsort(np.random.uniform(size=arr.shape), axis=-1)
            return arr[np.arange(arr.shape[0])[:, None], order]

        num_epochs_no_improvement, epoch = 0, 0
        best = [float("inf") for _ in range(self.ensemble_size)]
        holdout_losses = [float("nan")] * self.ensemble_size
        while num_epochs_no_improvement < max_epochs_no_improvement and epoch < max_epochs:
            for b in range(num_batches):
                bi = th.from_numpy(idxs[:, b * batch_size:(b + 1) * batch_size]).to(self.device, non_blocking=True)
                self.train_step(inputs[bi].contiguous(), targets[bi].contiguous())     # device gather of the index block
            idxs = shuffle_rows(idxs)
            holdout_losses = [float(v) for v in self.holdout_mse(holdout_inputs, holdout_targets).cpu()]
            self.elites = np.argsort(holdout_losses)[: self.num_elites]
            improved = False
            for i in range(self.ensemble_size):
                if epoch == 0 or (best[i] - holdout_losses[i]) / (best[i]) > 0.01:
                    best[i] = holdout_losses[i]
                    num_epochs_no_improvement = 0
                    improved = True
            if not improved:
                num_epochs_no_improvement += 1
            epoch += 1
        return np.mean(holdout_losses)


# ---------------------------------------------------------------------------------------------------------------------
# model-as-environment (common/model_based/utils.py:105-186).  The termination rules are environment knowledge on the
# host side: the reference's own functions are used when ``morl_baselines`` is importable, restatements otherwise.
# ---------------------------------------------------------------------------------------------------------------------
def termination_fn_false(obs, act, next_obs, rew):
    return np.zeros((len(obs), 1), dtype=bool)


def termination_fn_mountaincar(obs, act, next_obs, rew):
    return ((next_obs[:, 0] >= 0.45) * (next_obs[:, 1] >= 0.0))[:, np.newaxis]


def termination_fn_minecart(obs, act, next_obs, rew):
    old_pos, pos = obs[:, 0:2], next_obs[:, 0:2]
    in_base = np.sqrt(np.einsum("ij,ij->i", pos, pos)) < 0.15
    was_out_base = np.sqrt(np.einsum("ij,ij->i", old_pos, old_pos)) >= 0.15
    return (was_out_base * in_base)[:, np.newaxis]


def termination_fn_hopper(obs, act, next_obs, rew):
    height, angle = next_obs[:, 0], next_obs[:, 1]
    not_done = (np.isfinite(next_obs).all(axis=-1) * np.abs(next_obs[:, 1:] < 100).all(axis=-1) * (height > 0.7)
                * (np.abs(angle) < 0.2))
    return (~not_done)[:, np.newaxis]


def termination_fn_lunarlander(obs, act, next_obs, rew):
    exited = abs(next_obs[:, 0]) >= 1.0
    landed = (rew[:, 0] != 0) & (next_obs[:, 6] >= 0.95) & (next_obs[:, 7] >= 0.95)
    return (exited | landed)[:, np.newaxis]


def termination_fn_humanoid(obs, act, next_obs, rew):
    return (~((1.0 < next_obs[:, 0]) & (next_obs[:, 0] < 2.0)))[:, np.newaxis]


def _termination_for(env_id: str):
    try:  # the reference's table, unchanged, when it is installed
        from morl_baselines.common.model_based import utils as ref_utils
        return ref_utils.ModelEnv(None, env_id, 1).termination_func
    except ImportError:
        pass
    table = (("hopper", termination_fn_hopper), ("halfcheetah", termination_fn_false),
             ("humanoid", termination_fn_humanoid), ("lunar-lander", termination_fn_lunarlander),
             ("mo-reacher", termination_fn_false), ("mountaincar", termination_fn_mountaincar),
             ("minecart", termination_fn_minecart), ("mo-highway", termination_fn_false))
    for key, fn in table:
        if key in env_id:
            return fn
    raise NotImplementedError(f"no termination rule for '{env_id}': pass termination_func=")


class ModelEnv:
    """``ModelEnv`` (model_based/utils.py:105-186): one model step for a batch of (obs, one-hot / continuous action)."""

    def __init__(self, model, env_id=None, rew_dim=1, termination_func=None):
        self.model, self.rew_dim = model, rew_dim
        self.termination_func = termination_func or _termination_for(env_id or "")

    def step(self, obs: th.Tensor, act: th.Tensor, deterministic: bool = False):
        single = obs.dim() == 1
        if single:
            obs, act = obs.unsqueeze(0), act.unsqueeze(0)
        inputs = th.cat((obs, act), dim=-1).float().to(self.model.device)
        with th.no_grad():
            samples, vars_, uncertainties = self.model.sample(inputs, deterministic=deterministic)
        obs = obs.detach().cpu().numpy()
        samples[:, self.rew_dim:] += obs
        rewards, next_obs = samples[:, :self.rew_dim], samples[:, self.rew_dim:]
        terminals = self.termination_func(obs, act, next_obs, rewards)
        var_rewards, var_obs = vars_[:, :self.rew_dim], vars_[:, self.rew_dim:]
        if single:
            next_obs, rewards, terminals = next_obs[0], rewards[0], terminals[0]
            uncertainties, var_obs, var_rewards = uncertainties[0], var_obs[0], var_rewards[0]
        return next_obs, rewards, terminals, {"uncertainty": uncertainties, "var_obs": var_obs, "var_rewards": var_rewards}
