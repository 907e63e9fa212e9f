"""TEST INFRASTRUCTURE: flip-aware parity of one Envelope gradient step.

Two discrete decisions of a step depend on fp32 summation order, which no two GEMM engines share (torch's CPU BLAS, the
64 / 32-row MFMA chain, the 16-row chain, the per-layer tiles all differ):

* a hidden unit whose pre-activation is within rounding of zero lands on one side of the ReLU or the other -- and moves every
  upstream gradient of its row by that row's whole share;
* a TD row whose two best (weight, action) candidates tie to the last bit picks one or the other -- and regresses on another
  row of the target slab.

A blanket tolerance wide enough to absorb either (1e-3 of the largest gradient in round 2's sweep) also absorbs real errors.
Here the device's OWN decisions are fed to the oracle instead: the ReLU masks it applied (``morl_ctx_debug_hidden``) and the
targets it selected (``out->target``); every mask difference is asserted to sit on a pre-activation within ``ZERO_BAND`` of
zero in the oracle's forward (and every index difference to be a near-tie, in the callers), and then gradients, moments and
stepped parameters are compared with the oracle re-run under those decisions to the TIGHT contract: 5e-5 of the largest
gradient entry, and for the parameters the bound Adam's own formula gives for a gradient error of that size.
The observed maxima and flip counts are returned (and stored by the callers under gpurun_out/ -> profiles/).
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional

import numpy as np
import torch as th
import torch.nn.functional as F

import envelope_oracle as orc

ZERO_BAND = 1e-6        # |pre-activation| (relative to 1 + the row's largest |pre-activation| of that layer) of a unit that may flip
GRAD_TOL = 5e-5         # of the largest gradient entry (the fixtures' contract)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _flat(ts) -> th.Tensor:
    return th.cat([t.reshape(-1) for t in ts])


def forward_masked(params: List[th.Tensor], x: th.Tensor, masks: Optional[List[th.Tensor]]):
    """``QNet.forward`` (envelope.py:60-77) with the ReLU written as z * mask; ``masks`` None = the oracle's own (z > 0).
    Returns (output, [z_l])."""
    zs = []
    n_layers = len(params) // 2
    for l in range(n_layers):
        z = F.linear(x, params[2 * l], params[2 * l + 1])
        if l < n_layers - 1:
            zs.append(z.detach())
            m = (z > 0) if masks is None else masks[l]
            x = z * m.to(z.dtype)
        else:
            x = z
    return x, zs


def oracle_step_under(c, inp, dev_masks: Optional[List[th.Tensor]], dev_target: Optional[th.Tensor]) -> Dict[str, th.Tensor]:
    """The loss / backward / clip / Adam of ``Envelope.update`` (envelope.py:298-326) on torch-CPU with the training forward's
    ReLU masks and the (pre-TD) target rows GIVEN.  Row r = i * B + b as in the reference."""
    online = [th.tensor(a) for a in inp["online"]]
    m = [th.tensor(a) for a in inp["exp_avg"]]
    v = [th.tensor(a) for a in inp["exp_avg_sq"]]
    obs, act = th.tensor(inp["obs"]), th.tensor(inp["actions"])
    rew, done = th.tensor(inp["rewards"]), th.tensor(inp["dones"])
    sw = th.tensor(inp["sampled_w"]).float()
    B, W = obs.size(0), sw.size(0)
    w = sw.repeat_interleave(B, 0)
    t_obs, t_act, t_rew, t_done = obs.repeat(W, 1), act.repeat(W, 1), rew.repeat(W, 1), done.repeat(W, 1)
    target_q = t_rew + (1 - t_done) * c.gamma * dev_target                    # envelope.py:298
    leaf = [p.detach().clone().requires_grad_(True) for p in online]
    q, zs = forward_masked(leaf, th.cat((t_obs, w), dim=1), dev_masks)
    q_values = q.view(-1, c.A, c.R)
    q_value = q_values.gather(1, t_act.long().reshape(-1, 1, 1).expand(q_values.size(0), 1, c.R)).reshape(-1, c.R)
    loss = F.mse_loss(q_value, target_q)
    if c.homotopy_lambda > 0:
        loss = (1 - c.homotopy_lambda) * loss + c.homotopy_lambda * F.mse_loss(th.einsum("br,br->b", q_value, w),
                                                                              th.einsum("br,br->b", target_q, w))
    grads = list(th.autograd.grad(loss, leaf))
    if c.max_grad_norm is not None:
        norm = orc.clip_grad_norm(grads, c.max_grad_norm)
    else:
        norm = th.linalg.vector_norm(_flat(grads), 2.0)
    g_flat = _flat(grads).clone()
    m0, v0, p0 = _flat(m).clone(), _flat(v).clone(), _flat(online).clone()
    orc.adam_step(online, grads, m, v, c.step, c.lr)
    return {"loss": loss.detach(), "grad_norm": norm, "grads": g_flat, "zs": zs, "m": _flat(m), "v": _flat(v),
            "params": _flat(online), "m0": m0, "v0": v0, "p0": p0}


def adam_bound(g: th.Tensor, m0: th.Tensor, v0: th.Tensor, err: float, step: int, lr: float, b1=0.9, b2=0.999, eps=1e-8):
    """Element-wise bound on |p'(g + d) - p'(g)| for |d| <= err under torch's Adam: the step is evaluated in float64 on
    a grid of nine points of [g - err, g + err] and the largest deviation from the centre taken (x 1.5), plus the rounding of
    the fp32 update itself."""
    g, m0, v0 = g.double(), m0.double(), v0.double()
    bc1, bc2s = 1 - b1 ** step, (1 - b2 ** step) ** 0.5

    def upd(gg):
        mm = m0 + (gg - m0) * (1 - b1)
        vv = v0 * b2 + (1 - b2) * gg * gg
        return (lr / bc1) * mm / (vv.sqrt() / bc2s + eps)
    mid = upd(g)
    dev = th.zeros_like(mid)
    for f in (-1.0, -0.75, -0.5, -0.25, 0.25, 0.5, 0.75, 1.0):
        dev = th.maximum(dev, (upd(g + f * err) - mid).abs())
    # + the fp32 rounding of Adam's own arithmetic (five roundings on the way to the update: 6e-7 of its size)
    return 1.5 * dev + 6e-7 * mid.abs()


def check_step(c, inp, res, t, dev_hidden: List[th.Tensor], tag: str, store: bool = True) -> Dict[str, float]:
    """``res`` / ``t``: what ``test_kernels_parity.run_update(..., debug=True)`` returned; ``dev_hidden``: the device's saved
    post-ReLU activations per hidden layer, (rows, width) each.  Asserts the tight contract, returns what was observed."""
    rows = c.B * c.W
    dev_masks = [h.cpu() > 0 for h in dev_hidden]
    # the device's own (pre-TD) targets; DDQN / envelope alike
    o = oracle_step_under(c, inp, dev_masks, res["target"].cpu().view(rows, c.R))
    flips, worst_band = 0, 0.0
    for l, (z, mk) in enumerate(zip(o["zs"], dev_masks)):
        diff = (z > 0) != mk
        n = int(diff.sum())
        if n:
            scale = 1.0 + z.abs().amax(dim=1, keepdim=True)
            band = (z.abs() / scale)[diff]
            worst_band = max(worst_band, float(band.max()))
            assert float(band.max()) <= ZERO_BAND, (
                f"{tag}: hidden layer {l + 1}: {n} ReLU decisions differ from torch's and the worst sits at a pre-activation of "
                f"{float(band.max()):.3g} (relative) -- not a rounding flip")
            flips += n
    gmax = float(o["grads"].abs().max())
    g_err = float((t["g"].cpu() - o["grads"]).abs().max())
    assert g_err <= GRAD_TOL * gmax, f"{tag}: gradient off by {g_err / gmax:.3g} of its largest entry under the device's own decisions"
    obs = {"relu_flips": flips, "relu_flip_worst_band": worst_band, "hidden_units": int(sum(z.numel() for z in o["zs"])),
           "grad_err_rel_gmax": g_err / gmax if gmax > 0 else 0.0,
           "loss_rel": abs(res["loss"].item() - o["loss"].item()) / max(abs(o["loss"].item()), 1e-30)}
    assert obs["loss_rel"] <= 1e-5
    if c.max_grad_norm is not None or "grad_norm" in res:
        obs["grad_norm_rel"] = abs(res["grad_norm"].item() - o["grad_norm"].item()) / max(o["grad_norm"].item(), 1e-30)
        assert obs["grad_norm_rel"] <= 1e-5
    # moments: linear in the gradient
    m_err = float((t["m"].cpu() - o["m"]).abs().max())
    assert m_err <= 0.1 * GRAD_TOL * gmax + 2.4e-7 * float(o["m"].abs().max()) + 1e-12, f"{tag}: exp_avg off by {m_err:.3g}"
    v_err = float((t["v"].cpu() - o["v"]).abs().max())
    assert v_err <= 2 * 0.001 * GRAD_TOL * gmax * gmax + 2.4e-7 * float(o["v"].abs().max()) + 1e-20, f"{tag}: exp_avg_sq off by {v_err:.3g}"
    # stepped parameters: what Adam's formula makes of a gradient error of GRAD_TOL * gmax, + 2 ulp of the parameter
    g_in = o["grads"]          # (clipped) gradient Adam consumed
    bound = adam_bound(g_in, o["m0"], o["v0"], GRAD_TOL * gmax, c.step, c.lr) + 2.4e-7 * o["params"].abs().double() + 1e-12
    p_err = (t["po"].cpu().double() - o["params"].double()).abs()
    worst = float((p_err / bound).max())
    obs["param_err_over_adam_bound"] = worst
    obs["param_err_max_over_lr"] = float(p_err.max()) / c.lr
    assert worst <= 1.0, f"{tag}: a stepped parameter is {worst:.3g}x the Adam bound of the gradient tolerance away"
    if store:
        try:
            d = os.path.join(ROOT, "gpurun_out", "parity_observed")
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, f"{tag}.json"), "w") as fh:
                json.dump(dict(obs, case=c.name, rows=rows), fh)
        except OSError:
            pass
    return obs
