"""Both tilings of the layer-fused chain on the emulator.  The library picks 16-row tiles (mlp_chain16.h) whenever a chain has
at most 4 096 rows -- which is every case the emulator can afford -- so the 64 / 32-row persistent kernel (mlp_chain2.h) would
only be exercised on the GPU.  This test re-runs the chain-centred parity tests in a fresh interpreter with MORL_CHAIN16=0 (the
switch is read once per process), i.e. the same oracle / reference-golden assertions on the large-tile kernel."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_large_tile_chain_passes_the_same_parity_tests():
    env = dict(os.environ, MORL_CHAIN16="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_kernels_parity.py"), "-x", "-q", "-m", "not gpu",
                        "-k", "envelope_update_vs_reference_golden or qnet_forward_row_orders or grads_only", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_population_sized_shadow_refresh_passes_the_same_parity_tests():
    """Populations re-make the K-major shadow weights with the tiled transpose kernel behind the Adam step instead of scattering
    them from inside it (morl_ac.hip: adam); the switch is by size, so the emulator-sized cases never take that branch unless
    forced to (MORL_AC_SCATTER_MAX=0).  With the large-tile chain as well, this is the path a 64-learner MORL/D update takes."""
    # legs: the 16-row chain (K-major shadow copies), the large-tile chain with its default N-major weight stream (no shadow
    # copies at all) and with the K-major shadow copies (MORL_AC_NMAJOR=0)
    for extra in ({}, {"MORL_CHAIN16": "0"}, {"MORL_CHAIN16": "0", "MORL_AC_NMAJOR": "0"}):
        env = dict(os.environ, MORL_AC_SCATTER_MAX="0", **extra)
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_ac_kernels_parity.py"), "-x", "-q", "-m",
                            "not gpu", "-k", "population_batch_equals_independent_learners or update_matches_oracle_and_reference", "-p",
                            "no:cacheprovider"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        assert " passed" in r.stdout


@pytest.mark.gpu
def test_large_tile_chain_passes_the_shape_sweep_on_the_gpu():
    """The seeded shape sweep (tests/test_shape_fuzz.py: ragged batches, odd widths, every update family) takes the 16-row tiles by
    size; here it runs on the MI355X with MORL_CHAIN16=0, i.e. through the 64 / 32-row persistent kernel with all three weight
    streams (generic K-major, K4 constant-stride, N-major)."""
    env = dict(os.environ, MORL_CHAIN16="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_shape_fuzz.py"), "-x", "-q", "-m", "gpu", "-p",
                        "no:cacheprovider"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_lazy_targets_with_large_tiles_forced():
    """Lazy target evaluation with the forward passes on the 64 / 32-row tiles (as at the flagship size on the GPU; the lazily
    evaluated rows always take the 16-row ones) must equal the eager evaluation."""
    env = dict(os.environ, MORL_CHAIN16="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_kernels_parity.py"), "-x", "-q", "-m", "not gpu",
                        "-k", "lazy_target_evaluation and (ragged or small_homotopy or dup_weights)", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.gpu
def test_eager_target_evaluation_on_the_gpu():
    """The agents, traces and the ``lazy`` legs of the fixture tests run the lazy pipeline (tests/conftest.py); here the same
    assertions with MORL_LAZY_TARGETS=0 -- every leg eager (the ``lazy`` legs then assert that they were NOT lazy)."""
    env = dict(os.environ, MORL_LAZY_TARGETS="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_kernels_parity.py"),
                        os.path.join(ROOT, "tests", "test_flagship_golden.py"), os.path.join(ROOT, "tests", "test_train_traces.py"),
                        "-x", "-q", "-m", "gpu", "-k", "golden or trace", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_large_tile_split_bf16_chain_passes_the_same_parity_tests():
    """Chain launches of at most 4 096 rows -- every case the emulator can afford -- take the few-row split-bf16 chain
    (csrc/mlp_chain_bfn.h: 16-row tiles whose waves split the output features); the 64 / 32-row tiles of mlp_chain_bf.h, which carry the
    flagship step on the GPU, are reached here with MORL_BFN_MAX_ROWS=0: the same fixture / oracle assertions, including the arg-max
    inside the forward launch and the TD stage inside the backward launch, which only those tiles have."""
    env = dict(os.environ, MORL_BFN_MAX_ROWS="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_kernels_parity.py"),
                        os.path.join(ROOT, "tests", "test_flagship_golden.py"), "-x", "-q", "-m", "not gpu",
                        "-k", "(flagship_b32w8 or wide_pick or flagship) and (reference_golden or lazy_target_evaluation or fused_auto or separate_launch)",
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_eager_few_row_step_with_its_target_pass_on_the_f32_tiles():
    """An eagerly evaluated few-row step runs the target network's pass as a third chain of its few-row forward launch (bit 5 of
    ``morl_ctx_last_step_bf16``); MORL_BFN_EAGER3=0 keeps that pass on the f32 tiles as a launch of its own -- the same fixture
    assertions on that leg (eager and lazy then agree bit for bit again on the emulator, which the test asserts by itself)."""
    env = dict(os.environ, MORL_BFN_EAGER3="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_kernels_parity.py"), "-x", "-q", "-m", "not gpu",
                        "-k", "(flagship_b32w8 or wide_pick) and (reference_golden or lazy_target_evaluation)", "-p",
                        "no:cacheprovider"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_target_rows_on_the_few_row_split_bf16_chain_pass_the_same_parity_tests():
    """The lazily evaluated target rows of a bf16 step run on the 8-row f32 tiles of mlp_chain4.h; MORL_BFN_TARGETS=1 puts them on the
    few-row split-bf16 chain (csrc/mlp_chain_bfn.h, in_mode 3: pair list, device-side row count, count mirror) -- measured slower on
    the MI355X and therefore not the default, but a supported pipeline: the same fixture assertions, the lazy leg asserting through
    bit 5 of ``morl_ctx_last_step_bf16`` that the rows did run there."""
    env = dict(os.environ, MORL_BFN_TARGETS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_kernels_parity.py"), "-x", "-q", "-m", "not gpu",
                        "-k", "(flagship_b32w8 or wide_pick) and (reference_golden or lazy_target_evaluation)", "-p",
                        "no:cacheprovider"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_exact_f32_arithmetic_passes_the_same_parity_tests():
    """The suite runs qualifying networks (hidden layers of 256) through the split-bf16 chain (csrc/mlp_chain_bf.h; tests/conftest.py
    sets its row threshold to zero); here the same fixture / oracle assertions with MORL_EXACT_F32=1 -- every GEMM on the f32-input
    MFMA, the arithmetic of rounds 1-3."""
    env = dict(os.environ, MORL_EXACT_F32="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_kernels_parity.py"), "-x", "-q", "-m", "not gpu",
                        "-k", "(flagship_b32w8 or wide_pick) and (reference_golden or fused_auto or lazy_target_evaluation)", "-p",
                        "no:cacheprovider"],          # (the emulator leg: the fixture + oracle + lazy-vs-eager cases; the GPU leg below runs them all)
                       capture_output=True, text=True, timeout=2400, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.gpu
def test_exact_f32_arithmetic_on_the_gpu():
    env = dict(os.environ, MORL_EXACT_F32="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_kernels_parity.py"),
                        os.path.join(ROOT, "tests", "test_flagship_golden.py"), "-x", "-q", "-m", "gpu", "-k",
                        "golden or flagship_b32w8 or wide_pick or consecutive", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=2400, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


_CHAIN4_SNIPPET = r"""
import hashlib, os, sys
import numpy as np, torch as th
ROOT = sys.argv[1]
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import morl_baselines_amd.ops as ops
if sys.argv[2] == "gpu":
    from morl_baselines_amd.native import load_library
    lib, dev = load_library(), th.device("cuda:0")
else:
    import simlib
    lib, dev = simlib.load_sim(), th.device("cpu")
h = hashlib.sha256()
g = th.Generator().manual_seed(5)
# (rows, obs, objectives, actions, arch): ragged against the 8- and 16-row tiles, one to four hidden layers, narrow and wide heads
# (round 5: inputs wider than the 64 columns the prologue fetches ahead of the weight stream, a wide head behind one hidden layer)
shapes = [(1, 7, 3, 6, (256,)), (8, 7, 3, 6, (256, 256)), (9, 5, 2, 3, (256, 256)), (23, 32, 3, 6, (256, 256, 256, 256)),
          (40, 11, 4, 12, (256, 256, 256)), (70, 3, 2, 2, (256,)), (23, 70, 3, 6, (256, 256)), (9, 100, 4, 12, (256,))]
if sys.argv[2] == "gpu":
    shapes += [(1355, 32, 3, 6, (256, 256, 256, 256)), (2048, 32, 3, 6, (256, 256, 256, 256))]
for rows, D, R, A, arch in shapes:
    ctx = ops.QNetContext(D, R, A, arch, rows, 1, lib=lib)
    p = (th.randn(ctx.n_params, generator=g) * 0.08).to(dev)
    obs, w = th.randn(rows, D, generator=g).to(dev), th.rand(rows, R, generator=g).to(dev)
    q = ops.qnet_forward_rows(ctx, p, obs, w)
    assert th.isfinite(q).all()
    h.update(q.cpu().numpy().tobytes())
    ctx.close()
print("CHAIN4_DIGEST", h.hexdigest())
"""


def _chain4_digest(mode, extra_env):
    r = subprocess.run([sys.executable, "-c", _CHAIN4_SNIPPET, ROOT, mode], capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, **extra_env), cwd=ROOT)
    assert r.returncode == 0 and "CHAIN4_DIGEST" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout.split("CHAIN4_DIGEST")[1].split()[0]


def test_eight_row_tiles_give_the_bits_of_the_sixteen_row_tiles():
    """``mlp_chain4.h`` (8-row tiles on the 16-block 4x4x1 MFMA) takes every eligible no-grad forward chain of at most 2 048 rows;
    its contraction order is the one of ``mlp_chain16.h``, so the outputs must be the same BITS with the variant switched off."""
    assert _chain4_digest("sim", {}) == _chain4_digest("sim", {"MORL_CHAIN4": "0"})


@pytest.mark.gpu
def test_eight_row_tiles_give_the_bits_of_the_sixteen_row_tiles_on_the_gpu():
    assert _chain4_digest("gpu", {}) == _chain4_digest("gpu", {"MORL_CHAIN4": "0"})


_ARGMAX_SNIPPET = r"""
import hashlib, os, sys
import numpy as np, torch as th
ROOT = sys.argv[1]
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import morl_baselines_amd.ops as ops
if sys.argv[2] == "gpu":
    from morl_baselines_amd.native import load_library
    lib, dev = load_library(), th.device("cuda:0")
else:
    import simlib
    lib, dev = simlib.load_sim(), th.device("cpu")
h = hashlib.sha256()
g = th.Generator().manual_seed(11)
# (B, W, obs, objectives, actions, arch): a row tile of the forward launch is one transition's W rows (32-row tiles: W = 32; on the
# GPU also the flagship's 64-row tiles), two whole transitions (W = 16, with a last tile that is half empty: B odd) -- and one case
# where it is neither (W = 8: the separate launch either way)
cases = [(3, 32, 7, 3, 6, (256, 256)), (2, 32, 5, 2, 4, (256,)), (3, 16, 7, 3, 6, (256, 256)), (5, 16, 7, 3, 6, (256,)), (3, 8, 7, 3, 6, (256,))]
if sys.argv[2] == "gpu":
    cases += [(256, 64, 32, 3, 6, (256, 256, 256, 256)), (256, 32, 7, 3, 6, (256, 256, 256, 256)), (255, 16, 7, 3, 6, (256, 256, 256, 256))]
if os.environ.get("TILING_DEEP"):     # three hidden layers: the backward chain has two 256 x 256 steps (a rolling step entered with a pair pending)
    cases = [(4, 32, 5, 2, 4, (256, 256, 256)), (5, 16, 7, 3, 6, (256, 256))]
if os.environ.get("TILING_CASES"):
    cases = [cases[int(i)] for i in os.environ["TILING_CASES"].split(",")]
rows_seen, dual_seen, roll_seen, pw_seen, fpw_seen = [], [], [], [], []
for B, W, D, R, A, arch in cases:
    ctx = ops.QNetContext(D, R, A, arch, B, W, lib=lib)
    ctx.set_lazy_targets(2)
    P = ctx.n_params
    po = (th.randn(P, generator=g) * 0.1).to(dev); pt = (th.randn(P, generator=g) * 0.1).to(dev)
    obs, nobs = th.randn(B, D, generator=g).to(dev), th.randn(B, D, generator=g).to(dev)
    act = th.randint(0, A, (B,), generator=g).to(th.int32).to(dev)
    rew, done = th.randn(B, R, generator=g).to(dev), (th.rand(B, generator=g) < 0.2).float().to(dev)
    w = th.rand(W, R, generator=g); w = (w / w.sum(1, keepdim=True)).to(dev)
    grads, m, v = th.zeros(P, device=dev), th.zeros(P, device=dev), th.zeros(P, device=dev)
    out = ops.envelope_update(ctx, po, pt, grads, m, v, obs, nobs, act, rew, done, w, gamma=0.98, lr=3e-4, adam_step=1,
                              max_grad_norm=1.0, homotopy_lambda=0.3, debug="lazy")
    for k in ("pref", "ac", "target", "loss", "priority"):
        h.update(out[k].cpu().numpy().tobytes())
    h.update(grads.cpu().numpy().tobytes()); h.update(po.cpu().numpy().tobytes())
    rows_seen.append(ctx.lazy_target_rows(po))
    dual_seen.append((ctx.last_step_bf16() >> 6) & 1)
    roll_seen.append((ctx.last_step_bf16() >> 7) & 1)
    pw_seen.append((ctx.last_step_bf16() >> 8) & 1)
    fpw_seen.append((ctx.last_step_bf16() >> 9) & 1)
    # the same step as the agents issue it -- no parity outputs requested: the TD stage may then run inside the backward launch
    out2 = ops.envelope_update(ctx, po, pt, grads, m, v, obs, nobs, act, rew, done, w, gamma=0.98, lr=3e-4, adam_step=2,
                               max_grad_norm=1.0, homotopy_lambda=0.3)
    for k in ("loss", "grad_norm", "priority"):
        h.update(out2[k].cpu().numpy().tobytes())
    h.update(grads.cpu().numpy().tobytes()); h.update(po.cpu().numpy().tobytes()); h.update(m.cpu().numpy().tobytes())
    ctx.close()
print("DUAL_STEPS", sum(dual_seen))
print("ROLL_STEPS", sum(roll_seen))
print("PW_STEPS", sum(pw_seen))
print("FPW_STEPS", sum(fpw_seen))
print("ARGMAX_DIGEST", h.hexdigest(), rows_seen)
"""


def _argmax_digest(mode, extra_env, dual_steps=None, roll_steps=None, pw_steps=None, fpw_steps=None):
    r = subprocess.run([sys.executable, "-c", _ARGMAX_SNIPPET, ROOT, mode], capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, MORL_BF_MIN_ROWS="0", MORL_LAZY_MIN_ROWS="0", **extra_env), cwd=ROOT)
    assert r.returncode == 0 and "ARGMAX_DIGEST" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    if dual_steps is not None:
        seen = int(r.stdout.split("DUAL_STEPS")[1].split()[0])
        assert (seen > 0) == dual_steps, (seen, dual_steps)
    if roll_steps is not None:
        seen = int(r.stdout.split("ROLL_STEPS")[1].split()[0])
        assert (seen > 0) == roll_steps, (seen, roll_steps)
    if pw_steps is not None:
        seen = int(r.stdout.split("PW_STEPS")[1].split()[0])
        assert (seen > 0) == pw_steps, (seen, pw_steps)
    if fpw_steps is not None:
        seen = int(r.stdout.split("FPW_STEPS")[1].split()[0])
        assert (seen > 0) == fpw_steps, (seen, fpw_steps)
    return r.stdout.split("ARGMAX_DIGEST")[1].strip()


def test_argmax_inside_the_forward_launch_gives_the_bits_of_the_separate_launch():
    """When a row tile of the split-bf16 forward launch is one transition's W rows, its workgroups take the transition's arg-max
    themselves (``envelope_argmax_tile`` from the head's accumulators, no ``envelope_td_kernel<1>`` launch).  Indices, targets,
    loss, priorities, gradients and stepped parameters must be the bits of the step with the separate launch
    (``MORL_ARGMAX_IN_CHAIN=0``), and the same number of target rows must have been evaluated."""
    assert _argmax_digest("sim", {}) == _argmax_digest("sim", {"MORL_ARGMAX_IN_CHAIN": "0"})


@pytest.mark.gpu
def test_argmax_inside_the_forward_launch_gives_the_bits_of_the_separate_launch_on_the_gpu():
    assert _argmax_digest("gpu", {}) == _argmax_digest("gpu", {"MORL_ARGMAX_IN_CHAIN": "0"})


def test_td_stage_inside_the_backward_launch_gives_the_bits_of_the_separate_launch():
    """Likewise the TD stage (target from the compact target rows, TD error, dLoss/dQ, loss partials, priorities): when the row
    tiles of the split-bf16 backward launch are whole transitions and no parity outputs are requested, its workgroups compute their
    rows' dLoss/dQ themselves (``BfTdArgs``, no ``envelope_td_kernel<2>`` launch).  Same loss, priorities, gradients, parameters and
    optimiser state, to the last bit, as with ``MORL_TD_IN_CHAIN=0``."""
    assert _argmax_digest("sim", {}) == _argmax_digest("sim", {"MORL_TD_IN_CHAIN": "0"})


@pytest.mark.gpu
def test_td_stage_inside_the_backward_launch_gives_the_bits_of_the_separate_launch_on_the_gpu():
    assert _argmax_digest("gpu", {}) == _argmax_digest("gpu", {"MORL_TD_IN_CHAIN": "0"})


def test_paired_forward_tiles_give_the_bits_of_the_separate_tiles():
    """``MORL_BF_DUAL=1`` runs the two online forward passes of a step as PAIRS of 64-row tiles -- one of the next-state pass, one of the
    training pass, sharing every weight fragment a wave reads (csrc/mlp_chain_bf2.h).  Every accumulator sees the products it sees in
    mlp_chain_bf.h in the same order, so the step -- arg-max inside the launch, saves, sign bits, gradients, Adam -- is the same to the
    last bit.  The emulated chip is shrunk to two CUs (``HIPSIM_CUS``) so that the 64-row tiles, which want a tile per CU, are chosen at
    row counts the emulator can afford (ragged last tiles included)."""
    base = {"HIPSIM_CUS": "2", "MORL_BFN_MAX_ROWS": "0", "TILING_CASES": "0,3"}
    assert _argmax_digest("sim", base, dual_steps=False) == _argmax_digest("sim", dict(base, MORL_BF_DUAL="1"), dual_steps=True)


@pytest.mark.gpu
def test_paired_forward_tiles_give_the_bits_of_the_separate_tiles_on_the_gpu():
    assert _argmax_digest("gpu", {"MORL_BF_DUAL": "0"}, dual_steps=False) == _argmax_digest("gpu", {"MORL_BF_DUAL": "1"}, dual_steps=True)


def test_rolling_epilogues_give_the_bits_of_the_step_end_epilogues():
    """``MORL_BF_ROLL=1``: the backward chain's 64-row launch walks every 256 x 256 step pair of feature tiles after pair (a pair-major
    weight stream) and runs a finished pair's epilogue in slices between the MFMAs of the pair that follows (csrc/mlp_chain_bf_roll.h).
    Every accumulator sees the products it sees in mlp_chain_bf.h, in the same order: gradients, optimiser state and priorities are the
    same to the last bit.  Two hidden layers (one rolling step) and three (a rolling step entered with a pair pending), ragged tiles."""
    base = {"HIPSIM_CUS": "2", "MORL_BFN_MAX_ROWS": "0", "TILING_DEEP": "1"}
    assert _argmax_digest("sim", base, roll_steps=False) == _argmax_digest("sim", dict(base, MORL_BF_ROLL="1"), roll_steps=True)


@pytest.mark.gpu
def test_rolling_epilogues_give_the_bits_of_the_step_end_epilogues_on_the_gpu():
    assert _argmax_digest("gpu", {"MORL_BF_ROLL": "0"}, roll_steps=False) == _argmax_digest("gpu", {"MORL_BF_ROLL": "1"}, roll_steps=True)


def test_producer_wave_gives_the_bits_of_the_self_fed_chain():
    """``MORL_BF_PW`` bit 0 (the mask's default is 15, every form on): the backward chain's 64-row launch runs 320 work-items -- a fifth wave issues every piece of the weight ring and
    waits for it, the four MFMA waves issue none (csrc/mlp_chain_bf.h: bf_ring_producer).  Same stream, same products, same order: the
    same bits; what the test pins is the hand-over (barrier counts, buffer reuse) on ragged tiles and on two and three hidden layers."""
    base = {"HIPSIM_CUS": "2", "MORL_BFN_MAX_ROWS": "0", "TILING_DEEP": "1"}
    assert _argmax_digest("sim", dict(base, MORL_BF_PW="14"), pw_steps=False) == _argmax_digest("sim", dict(base, MORL_BF_PW="15"), pw_steps=True)


@pytest.mark.gpu
def test_producer_wave_gives_the_bits_of_the_self_fed_chain_on_the_gpu():
    assert _argmax_digest("gpu", {"MORL_BF_PW": "0"}, pw_steps=False) == _argmax_digest("gpu", {"MORL_BF_PW": "15"}, pw_steps=True)


def test_producer_wave_on_32_row_tiles_gives_the_bits_of_the_self_fed_chain():
    """``MORL_BF_PW`` bit 1: the producer wave also on the 32-row tiles that backward launches of fewer 64-row tiles than CUs take (192
    work-items: two MFMA waves + the producer) -- every case the emulated 256-CU chip runs with ``MORL_BFN_MAX_ROWS=0``."""
    base = {"MORL_BFN_MAX_ROWS": "0", "TILING_DEEP": "1"}
    assert _argmax_digest("sim", dict(base, MORL_BF_PW="13"), pw_steps=False) == _argmax_digest("sim", dict(base, MORL_BF_PW="15"), pw_steps=True)


@pytest.mark.gpu
def test_producer_wave_on_32_row_tiles_gives_the_bits_of_the_self_fed_chain_on_the_gpu():
    # (the 256 x 64 case runs the 64-row producer form in both legs; 256 x 32 -- 8 192 rows -- is the 32-row launch)
    assert _argmax_digest("gpu", {"MORL_BF_PW": "13"}) == _argmax_digest("gpu", {"MORL_BF_PW": "15"})


def test_producer_wave_in_the_forward_launch_gives_the_bits_of_the_self_fed_chain():
    """``MORL_BF_PW`` bit 2: a forward launch of one round (at most a 64-row tile per CU: one workgroup per CU) with the producer wave --
    no-grad chain with the arg-max at its end (the producer has ended by then) and training chain.  An emulated chip of four CUs: the
    two chains' four tiles are one round."""
    base = {"HIPSIM_CUS": "4", "MORL_BFN_MAX_ROWS": "0", "TILING_DEEP": "1"}
    assert _argmax_digest("sim", dict(base, MORL_BF_PW="11"), fpw_steps=False) == _argmax_digest("sim", dict(base, MORL_BF_PW="15"), fpw_steps=True)


@pytest.mark.gpu
def test_producer_wave_in_the_forward_launch_gives_the_bits_of_the_self_fed_chain_on_the_gpu():
    # (256 x 32: the forward launch's 2 x 128 tiles are one round)
    assert _argmax_digest("gpu", {"MORL_BF_PW": "3"}, fpw_steps=False) == _argmax_digest("gpu", {"MORL_BF_PW": "15"}, fpw_steps=True)


def test_producer_wave_in_the_forward_launch_on_32_row_tiles_gives_the_bits_of_the_self_fed_chain():
    """``MORL_BF_PW`` bit 3: ... and forward launches on 32-row tiles (192 work-items), the emulated 256-CU chip's choice for every case
    here."""
    base = {"MORL_BFN_MAX_ROWS": "0", "TILING_DEEP": "1"}
    assert _argmax_digest("sim", dict(base, MORL_BF_PW="7"), fpw_steps=False) == _argmax_digest("sim", dict(base, MORL_BF_PW="15"), fpw_steps=True)


@pytest.mark.gpu
def test_producer_wave_in_the_forward_launch_on_32_row_tiles_gives_the_bits_of_the_self_fed_chain_on_the_gpu():
    assert _argmax_digest("gpu", {"MORL_BF_PW": "7"}) == _argmax_digest("gpu", {"MORL_BF_PW": "15"})
