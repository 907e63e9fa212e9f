"""Parity of every C-ABI entry point against the oracle.

Each test runs twice:
  * backend "hip"  (``-m gpu``): the real gfx950 library on cuda:0 -- these are the parity tests proper;
  * backend "sim"  (CPU suite):  the same kernel sources compiled for the host with the wave-level emulator of
    tests/hipsim (checks index arithmetic / fragment maps / reductions without a GPU; small sizes only).
Tolerances: integer / boolean / index results bit-exact; fp32 results 1e-5 relative (BASELINE.json north_star).
"""
import os

import numpy as np
import pytest
import torch as th

import envelope_oracle as orc
from cases import CASES, make_inputs, pareto_sets

import morl_baselines_amd.ops as ops
from morl_baselines_amd.native import load_library

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-5


@pytest.fixture(scope="module", params=["sim", pytest.param("hip", marks=pytest.mark.gpu)])
def be(request):
    """(lib, device, is_sim)"""
    if request.param == "sim":
        import simlib
        return simlib.load_sim(), th.device("cpu"), True
    lib = load_library()
    assert lib.is_device_build
    return lib, th.device("cuda:0"), False


def flat(ps):
    return th.cat([th.as_tensor(p).reshape(-1) for p in ps])


def relmax(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def run_update(lib, dev, c, inp, apply_step=True, debug=True, fused=None, dw_mode=None, hidden=False):
    ctx = ops.QNetContext(c.D, c.R, c.A, c.arch, c.B, c.W, lib=lib, fused=fused)
    if dw_mode is not None:
        ctx.set_dw_mode(dw_mode)
    t = dict(
        po=flat(inp["online"]).to(dev), pt=flat(inp["target"]).to(dev), m=flat(inp["exp_avg"]).to(dev),
        v=flat(inp["exp_avg_sq"]).to(dev), obs=th.tensor(inp["obs"]).to(dev), nobs=th.tensor(inp["next_obs"]).to(dev),
        act=th.tensor(inp["actions"].astype(np.int32).reshape(-1)).to(dev), rew=th.tensor(inp["rewards"]).to(dev),
        done=th.tensor(inp["dones"]).reshape(-1).to(dev), sw=th.tensor(inp["sampled_w"]).float().to(dev))
    t["g"] = th.zeros_like(t["po"])
    res = ops.envelope_update(ctx, t["po"], t["pt"], t["g"], t["m"], t["v"], t["obs"], t["nobs"], t["act"], t["rew"],
                              t["done"], t["sw"], gamma=c.gamma, lr=c.lr, adam_step=c.step,
                              max_grad_norm=c.max_grad_norm, homotopy_lambda=c.homotopy_lambda, envelope=c.envelope,
                              apply_step=apply_step, debug=debug)
    if hidden:      # the ReLU decisions the device took (tests/flip_aware.py): post-ReLU activations of every hidden layer
        res["hidden"] = [ctx.debug_hidden(l, c.B * c.W, t["po"]).cpu() for l in range(1, len(c.arch) + 1)]
    res["lazy_rows"] = ctx.lazy_target_rows(t["po"])
    res["bits"] = ctx.last_step_bf16()
    res["engine"] = ctx.engine                                        # 0: the per-layer engine took the net (never lazy)
    if debug == "lazy":
        # no target slab was requested (asking for it is what switches the step to the eager form, include/morl_hip.h); the oracle
        # comparison of check_update needs one: a separate no-grad forward of the target network, row b * W + j like the slab
        assert "q_target_next" not in res
        res["q_target_next"] = ops.qnet_forward(ctx, t["pt"], t["nobs"], t["sw"], row_order=0).view(c.B, c.W, c.A, c.R)
    if dev.type == "cuda":
        th.cuda.synchronize()
    ctx.close()
    return res, t


def run_oracle(c, inp):
    th.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    online = [th.tensor(a) for a in inp["online"]]
    target = [th.tensor(a) for a in inp["target"]]
    m = [th.tensor(a) for a in inp["exp_avg"]]
    v = [th.tensor(a) for a in inp["exp_avg_sq"]]
    batch = tuple(th.tensor(inp[k]) for k in ("obs", "actions", "rewards", "next_obs", "dones"))
    sw = th.tensor(inp["sampled_w"]).float()
    o = orc.envelope_update(online, target, m, v, c.step, batch, sw, n_actions=c.A, reward_dim=c.R, gamma=c.gamma,
                            lr=c.lr, max_grad_norm=c.max_grad_norm, envelope=c.envelope,
                            homotopy_lambda=c.homotopy_lambda, dedup=c.envelope)
    return o, online, m, v


def check_update(res, t, o, online, m, v, c, param_tol_frac=0.02, grad_tol=5e-5, flip_aware=None):
    """The per-update parity contract (single update, identical parameters and batch).  ``flip_aware`` = (inputs, tag): the
    gradient / moment / parameter comparison is made against the oracle re-run under the device's own ReLU masks and targets
    (tests/flip_aware.py: tight tolerance + a derived Adam bound, every mask difference checked to be a rounding flip) instead
    of against the oracle's own decisions with ``grad_tol`` / ``param_tol_frac``."""
    assert abs(res["loss"].item() - o["loss"].item()) <= RTOL * abs(o["loss"].item())
    assert relmax(res["q_values"], o["q_values"]) <= RTOL
    if c.envelope:
        assert relmax(res["q_online_next"], o["qo"]) <= RTOL
        assert relmax(res["q_target_next"], o["qt"]) <= RTOL
        # arg-max indices are bit-exact *given identical Q inputs*: re-run the oracle reduce on the device's slabs
        tg, pref, ac = orc.envelope_reduce(res["q_online_next"].cpu(), res["q_target_next"].cpu(), t["sw"].cpu())
        assert th.equal(res["pref"].cpu().long(), pref.reshape(-1))
        assert th.equal(res["ac"].cpu().long(), ac.reshape(-1))
        if res.get("bits", 0) & 32:     # (target rows on the few-row split-bf16 chain: fp32-class, not the f32 tiles' bits)
            assert relmax(res["target"], tg.reshape(-1, c.R)) <= RTOL
        else:
            assert th.equal(res["target"].cpu(), tg.reshape(-1, c.R))
        # end-to-end index agreement with the reference arithmetic (reported; near-ties may flip on GEMM rounding)
        mism = (res["pref"].cpu().long() != o["pref"]) | (res["ac"].cpu().long() != o["ac"])
        assert mism.float().mean().item() <= 0.002
    assert relmax(res["target"], o["target"]) <= 1e-4 if c.envelope else relmax(res["target"], o["target"]) <= RTOL
    gn = o["grad_norm"].item()
    assert abs(res["grad_norm"].item() - gn) <= RTOL * gn
    assert relmax(res["priority"], o["priority_raw"]) <= 1e-4
    if flip_aware is not None:
        import flip_aware as fa
        inp, tag = flip_aware
        return fa.check_step(c, inp, res, t, res["hidden"], tag)
    assert relmax(t["g"], flat(o["grads"])) <= grad_tol
    assert relmax(t["m"], flat(m)) <= grad_tol
    assert relmax(t["v"], flat(v)) <= grad_tol
    # one Adam step moves a parameter by <= lr; the device must agree to a small fraction of that
    assert float((t["po"].cpu() - flat(online)).abs().max()) <= param_tol_frac * c.lr


def _lazy_capable(c, engine_fused):
    """Steps the library can evaluate lazily: envelope targets over >= 2 weights on the layer-fused engine with the row tile picked
    per launch (morl_hip.hip, morl_envelope_update); tests/conftest.py lowers MORL_LAZY_MIN_ROWS to 0 so that small steps qualify."""
    return c.envelope and c.W >= 2 and engine_fused == 1 and os.environ.get("MORL_LAZY_TARGETS") != "0"


@pytest.mark.parametrize("fused", [1, 2, 3, 0, "lazy"], ids=["fused_auto", "fused64", "fused32", "perlayer", "fused_auto_lazy"])
@pytest.mark.parametrize("c", CASES, ids=lambda c: c.name)
def test_envelope_update_vs_oracle(be, c, fused):
    """``debug=True`` asks for the whole target slab, which makes the step evaluate it EAGERLY; the ``lazy`` leg asks for
    everything but that slab and so runs the default pipeline (arg-max, compact rows, target network on the selected rows)."""
    lib, dev, is_sim = be
    inp = make_inputs(c)
    lazy = fused == "lazy"
    res, t = run_update(lib, dev, c, inp, fused=1 if lazy else fused, debug="lazy" if lazy else True)
    if lazy:
        assert (res["lazy_rows"] > 0) == _lazy_capable(c, res["engine"])   # (nets the per-layer engine takes stay eager)
    else:
        assert res["lazy_rows"] == 0
    o, online, m, v = run_oracle(c, inp)
    check_update(res, t, o, online, m, v, c)


@pytest.mark.parametrize("lazy", [False, True], ids=["eager", "lazy"])
@pytest.mark.parametrize("c", CASES, ids=lambda c: c.name)
def test_envelope_update_vs_reference_golden(be, c, lazy):
    """Directly against what the unmodified reference produced (tests/golden/*.npz); ``lazy``: through the default pipeline (no
    target slab requested)."""
    lib, dev, is_sim = be
    g = np.load(os.path.join(GOLD, f"envelope_{c.name}.npz"))
    inp = make_inputs(c)
    res, t = run_update(lib, dev, c, inp, debug="lazy" if lazy else True)
    if not lazy:
        assert res["lazy_rows"] == 0
    assert abs(res["loss"].item() - float(g["loss"])) <= RTOL * abs(float(g["loss"]))
    if c.max_grad_norm is not None:
        assert abs(res["grad_norm"].item() - float(g["grad_norm"])) <= RTOL * float(g["grad_norm"])
    assert relmax(res["target"], th.tensor(g["target"])) <= 1e-4
    s = c.subsample
    lay, off = [], 0
    for p in inp["online"]:
        lay.append((off, p.size))
        off += p.size
    po, gr = t["po"].cpu(), t["g"].cpu()
    gmax = max(float(np.abs(g[f"grad_{i}"]).max()) for i in range(len(lay)))
    for i, (o0, n) in enumerate(lay):
        assert float((po[o0:o0 + n][::s] - th.tensor(g[f"param_after_{i}"])).abs().max()) <= 0.02 * c.lr
        assert float((gr[o0:o0 + n][::s] - th.tensor(g[f"grad_{i}"])).abs().max()) <= 5e-5 * gmax
    pr = (res["priority"].cpu().numpy() + np.float32(0.125)) ** np.float32(0.6)
    np.testing.assert_allclose(pr, g["priority_final"], rtol=1e-4)


@pytest.mark.parametrize("dw_mode", [2, 3])     # 3 (dw_tiles.h) is the default, 2 the generic fall-back for unaligned operand rows
def test_weight_gradient_engines_agree_with_oracle(be, dw_mode):
    lib, dev, _ = be
    c = [c for c in CASES if c.name == "flagship_b32w8"][0]
    inp = make_inputs(c)
    res, t = run_update(lib, dev, c, inp, dw_mode=dw_mode)
    o, online, m, v = run_oracle(c, inp)
    check_update(res, t, o, online, m, v, c)


def test_grads_only_leaves_state_untouched(be):
    lib, dev, _ = be
    c = CASES[0]
    inp = make_inputs(c)
    res, t = run_update(lib, dev, c, inp, apply_step=False)
    assert th.equal(t["po"].cpu(), flat(inp["online"]))
    assert th.equal(t["m"].cpu(), flat(inp["exp_avg"]))
    assert float(t["g"].abs().max()) > 0


def test_envelope_reduce_ties_bit_exact(be):
    """Synthetic slabs full of exact ties: first (j, a) in flattened order must win, like th.max / th.argmax."""
    lib, dev, _ = be
    rng = np.random.default_rng(3)
    for (B, W, A, R) in [(5, 7, 3, 2), (4, 64, 6, 3), (3, 33, 5, 4), (2, 2, 1, 1)]:
        qo = th.tensor(np.round(rng.standard_normal((B, W, A, R)) * 2) / 2, dtype=th.float32)
        qt = th.tensor(rng.standard_normal((B, W, A, R)), dtype=th.float32)
        sw = np.round(np.abs(rng.standard_normal((W, R))) * 4) / 4 + 0.25
        sw = th.tensor(sw / sw.sum(1, keepdims=True), dtype=th.float32)
        for diag in (False, True):
            tg, pref, ac = ops.envelope_reduce(lib, qo.to(dev), qt.to(dev), sw.to(dev), diag_only=diag)
            if diag:
                scal = th.einsum("ir,biar->iba", sw, qo)
                ac_o = th.argmax(scal, dim=2)
                pref_o = th.arange(W).unsqueeze(1).expand(W, B)
                tg_o = qt[th.arange(B).unsqueeze(0).expand(W, B), pref_o, ac_o]
            else:
                tg_o, pref_o, ac_o = orc.envelope_reduce(qo, qt, sw)
            assert th.equal(pref.cpu().long(), pref_o.reshape(-1))
            assert th.equal(ac.cpu().long(), ac_o.reshape(-1))
            assert th.equal(tg.cpu(), tg_o.reshape(-1, R))


def test_envelope_reduce_maximum_slab_and_loud_refusal(be):
    """The LDS slab of the envelope kernel holds W * A * R <= 9216 floats (512 weights x 6 actions x 3 objectives: the
    weak-scaled candidate set of an 8-GPU job).  At the limit the arg-max is still bit-exact; one weight more is refused
    with an error, never truncated."""
    lib, dev, is_sim = be
    rng = np.random.default_rng(12)
    B, A, R = (2, 6, 3) if is_sim else (16, 6, 3)
    W = 512
    qo = th.tensor(rng.standard_normal((B, W, A, R)), dtype=th.float32)
    qt = th.tensor(rng.standard_normal((B, W, A, R)), dtype=th.float32)
    sw = th.tensor(rng.dirichlet(np.ones(R), W), dtype=th.float32)
    if not is_sim:                                        # (16 waves x 3072 candidates per row: GPU only, the emulator is slow)
        tg, pref, ac = ops.envelope_reduce(lib, qo.to(dev), qt.to(dev), sw.to(dev))
        tg_o, pref_o, ac_o = orc.envelope_reduce(qo, qt, sw)
        assert th.equal(pref.cpu().long(), pref_o.reshape(-1)) and th.equal(ac.cpu().long(), ac_o.reshape(-1))
        assert th.equal(tg.cpu(), tg_o.reshape(-1, R))
    W1 = W + 1
    qo1 = th.zeros((B, W1, A, R)); sw1 = th.full((W1, R), 1.0 / R)
    with pytest.raises(RuntimeError, match="LDS slab|exceeds|W"):
        ops.envelope_reduce(lib, qo1.to(dev), qo1.to(dev), sw1.to(dev))


@pytest.mark.parametrize("fused", [2, 3, 0], ids=["fused64", "fused32", "perlayer"])
@pytest.mark.parametrize("dims", [(6, 5, 3, 2, (16, 16)), (9, 4, 4, 3, (40,)), (130, 3, 5, 3, (200, 72, 136))])
def test_qnet_forward_row_orders(be, dims, fused):
    lib, dev, _ = be
    B, W, A, R, arch = dims
    D = 12 if arch[0] <= 32 else 11    # a narrow first layer is only fused when D + R is even
    rng = np.random.default_rng(B)
    params = orc.init_qnet_params(D, A, R, arch, generator=th.Generator().manual_seed(B))
    params = [p + 0.01 * th.randn(p.shape, generator=th.Generator().manual_seed(1)) for p in params]
    obs = th.tensor(rng.standard_normal((B, D)), dtype=th.float32)
    sw = th.tensor(orc.random_weights(R, W, "gaussian", rng=rng), dtype=th.float32).reshape(W, R)
    ctx = ops.QNetContext(D, R, A, arch, B, W, lib=lib, fused=fused)
    assert ctx.engine in (fused, 0)   # 0: architecture outside the fused engine's envelope -> per-layer GEMMs
    pf = flat(params).to(dev)
    q0 = ops.qnet_forward(ctx, pf, obs.to(dev), sw.to(dev), row_order=0).cpu()
    q1 = ops.qnet_forward(ctx, pf, obs.to(dev), sw.to(dev), row_order=1).cpu()
    # row_order 2: paired rows (obs[r], w[r]) -- QNet.forward on a batch, the evaluation / acting path
    pw = th.tensor(orc.random_weights(R, B, "gaussian", rng=rng), dtype=th.float32).reshape(B, R)
    q2 = ops.qnet_forward_rows(ctx, pf, obs.to(dev), pw.to(dev)).cpu()
    assert relmax(q2, orc.qnet_forward(params, obs, pw, A, R)) <= RTOL
    ctx.close()
    ref0 = orc.qnet_forward(params, obs.repeat_interleave(W, 0), sw.repeat(B, 1), A, R)
    ref1 = orc.qnet_forward(params, obs.repeat(W, 1), sw.repeat_interleave(B, 0), A, R)
    assert relmax(q0, ref0) <= RTOL and relmax(q1, ref1) <= RTOL
    assert th.equal(q0.view(B, W, A, R).transpose(0, 1).reshape(-1, A, R), q1)


def test_gather_batch(be):
    lib, dev, _ = be
    rng = np.random.default_rng(0)
    N, D, R, B = 300, 9, 3, 70
    rec = rng.standard_normal((N, 2 * D + R + 2)).astype(np.float32)
    rec[:, 2 * D + R] = (rng.random(N) < 0.1)
    rec[:, 2 * D + R + 1] = rng.integers(6, size=N)
    idx = rng.integers(N, size=B)
    obs, act, rew, nobs, done = ops.gather_batch(lib, th.tensor(rec).to(dev), th.tensor(idx).to(dev), D, R)
    assert np.array_equal(obs.cpu().numpy(), rec[idx, :D])
    assert np.array_equal(nobs.cpu().numpy(), rec[idx, D:2 * D])
    assert np.array_equal(rew.cpu().numpy(), rec[idx, 2 * D:2 * D + R])
    assert np.array_equal(done.cpu().numpy()[:, 0], rec[idx, 2 * D + R])
    assert np.array_equal(act.cpu().numpy(), rec[idx, 2 * D + R + 1].astype(np.int32))


@pytest.mark.parametrize("tau", [1.0, 0.005, 0.3])
def test_polyak(be, tau):
    lib, dev, _ = be
    g = th.Generator().manual_seed(0)
    src, dst = th.randn(10007, generator=g), th.randn(10007, generator=g)
    want = [dst.clone()]
    orc.polyak_update([src], want, tau)
    d = dst.clone().to(dev)
    ops.polyak(lib, src.to(dev), d, tau)
    if tau == 1.0:
        assert th.equal(d.cpu(), want[0])
    else:
        assert relmax(d, want[0]) <= 1e-6


def _pareto_both(lib, dev, pts, rd):
    got = ops.pareto_mask(lib, th.tensor(pts, dtype=th.float64).to(dev), rd).cpu().numpy().astype(bool)
    want = orc.pareto_mask(pts, rd)
    return got, want


def test_pareto_mask_adversarial_sets(be):
    lib, dev, _ = be
    g = np.load(os.path.join(GOLD, "pareto_masks.npz"))
    for name, pts in pareto_sets().items():
        for rd in (True, False):
            got, want = _pareto_both(lib, dev, pts, rd)
            assert np.array_equal(got, want), (name, rd)
            assert np.array_equal(got, g[f"{name}__rd{int(rd)}"].astype(bool)), (name, rd)  # the reference's own mask


def test_reference_pruning_known_fronts(be):
    """``tests/test_pruning.py::test_small_pf`` / ``test_large_pf`` of the reference at their full sizes (100 + 500 x 2; 1 000 +
    5 000 x 4 -- the large one on the GPU only: the emulator runs N^2 fibres): ``morl_pareto_mask`` equals the mask the unmodified
    ``get_non_pareto_dominated_inds`` produced on the same points (tests/golden/pruning_known_fronts.npz), bit for bit, and the
    kept set is the planted front -- the reference test's own assertion."""
    lib, dev, is_sim = be
    import morl_baselines_amd.pareto as par
    g = np.load(os.path.join(GOLD, "pruning_known_fronts.npz"))
    for name in ("small_pf",) if is_sim else ("small_pf", "large_pf"):
        pts, n_nd = g[f"{name}__points"], int(g[f"{name}__n_nd"])
        for rd in (True, False):
            got = ops.pareto_mask(lib, th.tensor(pts, dtype=th.float64).to(dev), rd).cpu().numpy().astype(bool)
            assert np.array_equal(got, g[f"{name}__mask_rd{int(rd)}"].astype(bool)), (name, rd)
        kept = par.filter_pareto_dominated(pts, lib=lib, device=dev)
        assert {tuple(v) for v in kept} == {tuple(v) for v in pts[:n_nd]}, name


def test_pareto_mask_single_and_tile_edges(be):
    lib, dev, is_sim = be
    rng = np.random.default_rng(11)
    sizes = [1, 2, 63, 64, 65, 255, 256, 257] + ([] if is_sim else [511, 512, 513, 1025, 3000])
    for n in sizes:
        for R in (1, 2, 3, 5, 8):
            pts = np.round(rng.random((n, R)) * 6) / 6
            got, want = _pareto_both(lib, dev, pts, True)
            assert np.array_equal(got, want), (n, R)


def test_sumtree_trace_bit_exact(be):
    """Adds, samples and priority updates against the oracle SumTree (itself pinned to the reference trace)."""
    lib, dev, _ = be
    rng = np.random.default_rng(21)
    for cap in (50, 64, 1000):
        n_levels = int(np.ceil(np.log2(cap))) + 1
        tree_o = orc.SumTree(cap)
        tree_d = th.zeros(2 ** n_levels - 1, dtype=th.float64, device=dev)
        rmax_d = th.tensor([1e-5], dtype=th.float64, device=dev)
        minp, ptr = 1e-5, 0
        for rnd in range(6):
            n_add = int(rng.integers(1, 40))
            ptrs = []
            for _ in range(n_add):
                tree_o.set(ptr, minp)
                ptrs.append(ptr)
                ptr = (ptr + 1) % cap
            ops.sumtree_set(lib, tree_d, n_levels, th.tensor(ptrs, dtype=th.int64, device=dev), None, rmax_d)
            B = int(rng.integers(1, 200))
            u = rng.random(B)
            idx_o = tree_o.sample_from_uniforms(u)
            idx_d = ops.sumtree_sample(lib, tree_d, n_levels, th.tensor(u, device=dev))
            assert np.array_equal(idx_d.cpu().numpy(), idx_o)
            raw = (rng.random(B) * 2).astype(np.float32)
            pr_o = (raw + np.float32(minp)) ** np.float32(0.6)
            pr_d = th.empty(B, dtype=th.float64, device=dev)
            ops.sumtree_update(lib, tree_d, n_levels, idx_d, th.tensor(raw, device=dev), 0.6, rmax_d, pr_d)
            # powf may differ by an ulp between libm and the device: feed the DEVICE's priorities to the oracle tree so
            # that the tree arithmetic itself is compared bit-for-bit, and bound the pow difference separately
            pr_dev = pr_d.cpu().numpy()
            np.testing.assert_allclose(pr_dev, pr_o.astype(np.float64), rtol=3e-7)
            pr32 = pr_dev.astype(np.float32)
            minp = max(minp, pr32.max())
            tree_o.batch_set(idx_o, pr32)
            assert float(rmax_d.cpu()[0]) == float(minp)
            got = tree_d.cpu().numpy()
            for l in range(n_levels):
                assert np.array_equal(got[2 ** l - 1: 2 ** (l + 1) - 1], tree_o.nodes[l]), (cap, rnd, l)


# ---- round 2: fused launches must equal the launches they replaced, bit for bit ------------------------------------------------
def _records(dev, n=300, D=5, R=2, Ad=1, seed=0):
    rng = np.random.default_rng(seed)
    rec = rng.standard_normal((n, 2 * D + R + 1 + Ad)).astype(np.float32)
    rec[:, -1] = rng.integers(0, 4, n)
    return th.tensor(rec).to(dev), D, R, Ad


def _tree(dev, n_leaves=300, seed=1):
    """A device sum tree whose nodes are exact float64 sums of small integers (any summation order gives the same bits)."""
    rng = np.random.default_rng(seed)
    n_levels = int(np.ceil(np.log2(n_leaves))) + 1
    leaves = np.zeros(2 ** (n_levels - 1))
    leaves[:n_leaves] = rng.integers(1, 50, n_leaves)
    levels = [leaves]
    while len(levels[0]) > 1:
        levels.insert(0, levels[0].reshape(-1, 2).sum(1))
    return th.tensor(np.concatenate(levels)).to(dev), n_levels


def test_sample_gather_equals_descent_plus_gather(be):
    """morl_sample_gather == morl_sumtree_sample followed by morl_gather_batch (and the aux copy is a copy); indices path too."""
    lib, dev, _ = be
    records, D, R, Ad = _records(dev)
    tree, n_levels = _tree(dev)
    B = 37
    u = th.tensor(np.random.default_rng(2).random(B)).to(dev)
    aux_src = th.arange(24, dtype=th.float32, device=dev) * 0.5
    aux_dst = th.zeros(24, dtype=th.float32, device=dev)
    got = ops.sample_gather(lib, records, B, D, R, Ad, True, tree=tree, n_levels=n_levels, u01_ptr=u.data_ptr(),
                            aux_src_ptr=aux_src.data_ptr(), aux_dst=aux_dst)
    idx = ops.sumtree_sample(lib, tree, n_levels, u)
    want = ops.gather_batch(lib, records, idx, D, R, Ad, int_actions=True)
    assert th.equal(got[5], idx) and th.equal(aux_dst, aux_src)
    for a, b in zip(got[:5], want):
        assert th.equal(a.reshape(-1), b.reshape(-1))
    inds = th.tensor(np.random.default_rng(3).integers(0, 300, B), dtype=th.int64).to(dev)
    got2 = ops.sample_gather(lib, records, B, D, R, Ad, True, idx_ptr=inds.data_ptr())
    want2 = ops.gather_batch(lib, records, inds, D, R, Ad, int_actions=True)
    assert th.equal(got2[5], inds)
    for a, b in zip(got2[:5], want2):
        assert th.equal(a.reshape(-1), b.reshape(-1))


def test_prologue_and_in_step_per_update_change_nothing(be):
    """An update whose shadow weights were made by morl_envelope_prepare and whose PER update rides in the weight-gradient
    launch == the same update with the library's own shadow launch followed by a separate morl_sumtree_update: parameters,
    moments, loss and the float64 tree are bit-identical."""
    lib, dev, _ = be
    c = [c for c in CASES if c.name == "flagship_b32w8"][0]
    inp = make_inputs(c)
    records, D, R, Ad = _records(dev, n=300, D=c.D, R=c.R)
    idx = th.tensor(np.random.default_rng(5).integers(0, 300, c.B), dtype=th.int64).to(dev)
    idx[3] = idx[1]                                        # a duplicate index: batch_set keeps the first occurrence

    def run(fused_path):
        ctx = ops.QNetContext(c.D, c.R, c.A, c.arch, c.B, c.W, lib=lib)
        t = dict(po=flat(inp["online"]).to(dev), pt=flat(inp["target"]).to(dev), m=flat(inp["exp_avg"]).to(dev),
                 v=flat(inp["exp_avg_sq"]).to(dev))
        t["g"] = th.zeros_like(t["po"])
        tree, n_levels = _tree(dev)
        rmax = th.tensor([0.125], dtype=th.float64, device=dev)
        per = None
        if fused_path:
            ops.sample_gather(lib, records, c.B, D, R, Ad, True, idx_ptr=idx.data_ptr(), prepare=(ctx, t["po"], t["pt"]))
            per = (tree, n_levels, idx, 0.6, rmax)
        res = ops.envelope_update(ctx, t["po"], t["pt"], t["g"], t["m"], t["v"], th.tensor(inp["obs"]).to(dev),
                                  th.tensor(inp["next_obs"]).to(dev),
                                  th.tensor(inp["actions"].astype(np.int32).reshape(-1)).to(dev), th.tensor(inp["rewards"]).to(dev),
                                  th.tensor(inp["dones"]).reshape(-1).to(dev), th.tensor(inp["sampled_w"]).float().to(dev),
                                  gamma=c.gamma, lr=c.lr, adam_step=c.step, max_grad_norm=c.max_grad_norm, per=per)
        if not fused_path:
            ops.sumtree_update(lib, tree, n_levels, idx, res["priority"], 0.6, rmax)
        if dev.type == "cuda":
            th.cuda.synchronize()
        out = (t["po"].clone(), t["m"].clone(), t["v"].clone(), res["loss"].clone(), tree.clone(), rmax.clone())
        ctx.close()
        return out
    for a, b in zip(run(True), run(False)):
        assert th.equal(a, b)


def test_greedy_actions_follow_the_fma_chain(be):
    """morl_envelope_greedy_actions == first arg-max of fma(w2, q2, fma(w1, q1, w0 * q0)) over the device's own Q rows
    (Envelope.max_action, envelope.py:389-402)."""
    lib, dev, _ = be
    c = [c for c in CASES if c.name == "flagship_b32w8"][0]
    inp = make_inputs(c)
    n = 48
    ctx = ops.QNetContext(c.D, c.R, c.A, c.arch, n, 1, lib=lib)
    po = flat(inp["online"]).to(dev)
    rng = np.random.default_rng(9)
    obs = th.tensor(rng.standard_normal((n, c.D)).astype(np.float32)).to(dev)
    w = th.tensor(rng.dirichlet(np.ones(c.R), n).astype(np.float32)).to(dev)
    ac = ops.envelope_greedy_actions(ctx, po, obs, w).cpu().numpy()
    q = ops.qnet_forward_rows(ctx, po, obs, w).cpu().numpy().astype(np.float64)          # (n, A, R)
    wd = w.cpu().numpy().astype(np.float64)
    f32 = lambda x: x.astype(np.float32).astype(np.float64)
    s = f32(wd[:, None, 0] * q[..., 0])
    for r in range(1, c.R):
        s = f32(q[..., r] * wd[:, None, r] + s)                                           # one rounding: an fma
    assert np.array_equal(ac, s.argmax(1))
    ctx.close()


@pytest.mark.parametrize("c", [c for c in CASES if c.envelope] +
                         [__import__("cases").Case("ragged", B=37, W=5, D=9, A=4, R=3, arch=(48, 40), homotopy_lambda=0.2, seed=21),
                          # more than 64 weight vectors: the TD rows of a transition span two workgroups of the arg-max launch, each
                          # with its own list of selected pairs (a pair selected from both is evaluated twice)
                          __import__("cases").Case("two_groups", B=3, W=80, D=5, A=3, R=2, arch=(48, 40), seed=23),
                          # 8-row tiles with a transition that selects more than 8 distinct weights, one hidden layer
                          __import__("cases").Case("wide_pick", B=8, W=40, D=9, A=4, R=3, arch=(256,), seed=4)],
                         ids=lambda c: c.name)
def test_lazy_target_evaluation_equals_the_eager_one(be, c):
    """``morl_envelope_update`` evaluates the TARGET network lazily by default: arg-max over the online slab first, then the target
    network on the distinct (transition, weight) pairs the TD rows actually selected (``envelope.py:429-439`` only ever gathers
    those).  Same targets, indices, priorities, gradients and stepped parameters, bit for bit at sizes where both forms run the
    same tiling (the emulator's), and the row count it reports is exactly the number of distinct selected pairs."""
    lib, dev, is_sim = be
    inp = make_inputs(c)
    ctx_rows = {}

    def run(debug):
        ctx = ops.QNetContext(c.D, c.R, c.A, c.arch, c.B, c.W, lib=lib)
        t = dict(po=flat(inp["online"]).to(dev), pt=flat(inp["target"]).to(dev), m=flat(inp["exp_avg"]).to(dev),
                 v=flat(inp["exp_avg_sq"]).to(dev))
        t["g"] = th.zeros_like(t["po"])
        res = ops.envelope_update(ctx, t["po"], t["pt"], t["g"], t["m"], t["v"], th.tensor(inp["obs"]).to(dev),
                                  th.tensor(inp["next_obs"]).to(dev), th.tensor(inp["actions"].astype(np.int32).reshape(-1)).to(dev),
                                  th.tensor(inp["rewards"]).to(dev), th.tensor(inp["dones"]).reshape(-1).to(dev),
                                  th.tensor(inp["sampled_w"]).float().to(dev), gamma=c.gamma, lr=c.lr, adam_step=c.step,
                                  max_grad_norm=c.max_grad_norm, homotopy_lambda=c.homotopy_lambda, envelope=True, debug=debug)
        ctx_rows[debug] = ctx.lazy_target_rows(t["po"])
        ctx_rows["engine"] = ctx.engine
        ctx_rows[("bits", debug)] = ctx.last_step_bf16()
        ctx.close()
        return res, t
    eager, te = run(True)
    lazy, tl = run("lazy")
    assert ctx_rows[True] == 0                                         # the whole slab was asked for: evaluated eagerly
    pref = lazy["pref"].cpu().long().view(c.W, c.B)                    # [i][b] -> j*
    distinct = sum(len(set(pref[:, b].tolist())) for b in range(c.B))
    if ctx_rows["engine"] > 0 and c.W <= 64:                           # (the per-layer engine of the narrowest nets stays eager)
        assert ctx_rows["lazy"] == distinct and 1 <= distinct <= c.B * c.W
    elif ctx_rows["engine"] > 0:                                       # (two row groups per transition: shared pairs counted per group)
        assert distinct <= ctx_rows["lazy"] <= 2 * distinct
    else:
        assert ctx_rows["lazy"] == 0
    # bit for bit where both forms run the 16-row tiles; with the large tiles forced (tests/test_chain_tilings.py) the eager target
    # slab comes from 64 / 32-row tiles and the lazy rows from 16-row ones, as on the GPU at the flagship size
    # ... and the lazy rows on the f32 tiles: a step on the bf16 matrix cores evaluates them on the few-row split-bf16 chain
    # (mlp_chain_bfn.h, bit 5 of last_step_bf16) -- six split products like its online passes, fp32-class, not the eager slab's bits
    bits = ctx_rows[("bits", "lazy")]
    on_bfn = bool(bits & 32)
    # (... or the EAGER step's target pass: a few-row step evaluates it as a third chain of its few-row forward launch)
    eager_on_bfn = bool(ctx_rows[("bits", True)] & 32)
    if (bits & 1) and ctx_rows["lazy"] > 0 and os.environ.get("MORL_BFN_TARGETS") == "1":
        assert on_bfn, bits                                            # (asked for: tests/test_chain_tilings.py)
    if os.environ.get("MORL_BFN_TARGETS", "0") == "0" or os.environ.get("MORL_EXACT_F32") == "1":
        assert not on_bfn, bits
    exact = is_sim and os.environ.get("MORL_CHAIN16") != "0" and not on_bfn and not eager_on_bfn
    for k in ("target", "pref", "ac", "priority", "q_values", "q_online_next"):
        if exact:
            assert th.equal(eager[k].cpu(), lazy[k].cpu()), k
        else:                                                          # (GPU: the two forms may run different row tilings)
            assert relmax(lazy[k], eager[k]) <= (0 if k in ("pref", "ac") else 1e-5), k
    if exact:
        assert th.equal(te["g"], tl["g"]) and th.equal(te["po"], tl["po"]) and eager["loss"].item() == lazy["loss"].item()
    else:
        assert relmax(tl["g"], te["g"]) <= 5e-6 and abs(eager["loss"].item() - lazy["loss"].item()) <= 1e-6 * abs(eager["loss"].item())
