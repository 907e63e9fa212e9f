"""Golden fixture of the dynamics ensemble: the UNMODIFIED reference ``ProbabilisticEnsemble.fit()`` on seeded data
(build container only).   PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden_ens.py"""
import os
import sys

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)

import ref_harness as rh  # noqa: E402

CFG = dict(input_dim=7, output_dim=6, ensemble_size=3, arch=[16, 16], learning_rate=1e-3, num_elites=2)
FIT = dict(batch_size=16, holdout_ratio=0.2, max_epochs=3)
N, SEED = 60, 11


def data():
    rng = np.random.default_rng(SEED)
    X = rng.standard_normal((N, CFG["input_dim"])).astype(np.float32)
    Wt = rng.standard_normal((CFG["input_dim"], CFG["output_dim"])).astype(np.float32) * 0.5
    Y = (np.tanh(X @ Wt) + 0.05 * rng.standard_normal((N, CFG["output_dim"]))).astype(np.float32)
    probe = rng.standard_normal((9, CFG["input_dim"])).astype(np.float32)
    return X, Y, probe


def main():
    rh.install_stubs()
    from morl_baselines.common.model_based.probabilistic_ensemble import ProbabilisticEnsemble
    th.set_num_threads(1)
    X, Y, probe = data()
    out = {}
    for norm in (True, False):
        th.manual_seed(SEED)
        np.random.seed(SEED)
        model = ProbabilisticEnsemble(normalize_inputs=norm, device="cpu", **CFG)
        tag = f"n{int(norm)}"
        for l, layer in enumerate(model.layers):
            out[f"{tag}_init_W{l}"] = layer.W.detach().numpy().copy()
            out[f"{tag}_init_b{l}"] = layer.b.detach().numpy().copy()
        np.random.seed(SEED + 1)
        hl = model.fit(X, Y, **FIT)
        for l, layer in enumerate(model.layers):
            out[f"{tag}_W{l}"] = layer.W.detach().numpy().copy()
            out[f"{tag}_b{l}"] = layer.b.detach().numpy().copy()
        out[f"{tag}_max_logvar"] = model.max_logvar.detach().numpy().copy()
        out[f"{tag}_min_logvar"] = model.min_logvar.detach().numpy().copy()
        out[f"{tag}_holdout"] = np.float64(hl)
        out[f"{tag}_elites"] = np.asarray(model.elites)
        with th.no_grad():
            mean, logvar = model(th.tensor(probe), deterministic=True, return_dist=True)
        out[f"{tag}_probe_mean"], out[f"{tag}_probe_logvar"] = mean.numpy().copy(), logvar.numpy().copy()
        np.random.seed(SEED + 2)
        s, v, u = model.sample(th.tensor(probe), deterministic=True)
        out[f"{tag}_sample"], out[f"{tag}_vars"], out[f"{tag}_unc"] = s, v, u
        print(tag, "holdout", hl, "elites", model.elites)
    np.savez_compressed(os.path.join(HERE, "ens_fit.npz"), **out)


if __name__ == "__main__":
    main()
