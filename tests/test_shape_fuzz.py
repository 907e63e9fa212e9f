"""Seeded shape sweep of one Envelope gradient step: random batch / weight counts (ragged against every tile size: 16, 32, 64
rows), observation / action / objective counts and hidden widths (multiples of 4 up to 256, one to four layers), through every
MLP engine, against the oracle with the per-update parity contract of test_kernels_parity.py.  The fixtures pin the reference's
shapes; this pins the edges between them -- partial row tiles, K padding of odd input widths, narrow / wide last layers, split
counts of the weight-gradient jobs.  The emulator takes the three smallest cases, the GPU all of them."""
import numpy as np
import pytest

from cases import Case, make_inputs
from test_kernels_parity import be, check_update, run_oracle, run_update  # noqa: F401  (be: the backend fixture)


def _random_cases(n, seed=2024):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        L = int(rng.integers(1, 5))
        arch = tuple(int(4 * rng.integers(9, 65)) for _ in range(L))            # 36 .. 256, multiples of 4
        if k % 5 == 0:
            arch = tuple(256 for _ in range(L))                                  # the fast (constant-stride) weight stream
        B = int(rng.choice([1, 2, 3, 7, 15, 16, 17, 31, 33, 48, 63, 64, 65, 96, 130]))
        W = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 24, 33]))
        if B * W > 2600:
            W = max(1, 2600 // B)
        out.append(Case(f"fuzz{k}_B{B}W{W}_{'x'.join(map(str, arch))}", B=B, W=max(W, 1 if k % 7 else 2), D=int(rng.integers(1, 41)),
                        A=int(rng.integers(1, 8)), R=int(rng.integers(2, 5)), arch=arch, envelope=bool(k % 6 != 5),
                        homotopy_lambda=float(rng.choice([0.0, 0.0, 0.4])), max_grad_norm=[1.0, 0.1, None][k % 3],
                        step=int(rng.integers(1, 9)), seed=100 + k))
    return out


FUZZ = _random_cases(30)
SMALL = sorted(FUZZ, key=lambda c: c.B * c.W * sum(c.arch))[:3]


@pytest.mark.parametrize("fused", [1, 0], ids=["fused", "perlayer"])
@pytest.mark.parametrize("c", FUZZ, ids=lambda c: c.name)
def test_random_shapes_match_the_oracle(be, c, fused):
    lib, dev, is_sim = be
    if is_sim and c not in SMALL:
        pytest.skip("the emulator runs the three smallest shapes; the rest need the GPU")
    inp = make_inputs(c)
    res, t = run_update(lib, dev, c, inp, fused=fused, hidden=True)
    o, online, m, v = run_oracle(c, inp)
    # Round 2 allowed 1e-3 on gradients and 0.1 lr on parameters here: a hidden unit whose pre-activation is within rounding of
    # zero lands on the other side of the ReLU than in torch's GEMM (observed 2.7e-4 of the largest gradient entry at B130 x W20,
    # 156 x 124) and a near-tie arg-max picks another target row.  Now the oracle is re-run under the DEVICE's ReLU masks and
    # targets (tests/flip_aware.py): every mask difference must sit within 1e-6 of zero, and gradients / moments / parameters
    # are held to the fixtures' 5e-5 and to the bound Adam's formula derives from it
    check_update(res, t, o, online, m, v, c, flip_aware=(inp, f"fuzz_{'sim' if is_sim else 'hip'}_{'fused' if fused else 'perlayer'}_{c.name}"))


# ---- the actor-critic updates (CAPQL, MOSAC, MOSAC-discrete, GPI-PD continuous): the same sweep over their shapes ------------
from cases_ac import ACCase  # noqa: E402
from test_ac_kernels_parity import run_and_check_against_oracle  # noqa: E402
from test_ac_kernels_parity import be as ac_be  # noqa: E402,F401


def _random_ac_cases(n, seed=77):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        algo = ["capql", "mosac", "gpipd", "sacd"][k % 4]
        arch = tuple(int(4 * rng.integers(9, 65)) for _ in range(2))
        if k % 3 == 0:
            arch = (256, 256)
        B = int(rng.choice([1, 3, 15, 16, 17, 31, 33, 64, 100, 128, 200]))
        kw = dict(D=int(rng.integers(2, 24)), Ad=int(rng.integers(1, 7)), R=int(rng.integers(2, 5)), arch=arch, B=B,
                  step=int(rng.integers(1, 6)), seed=500 + k)
        if algo == "mosac":
            kw.update(autotune=bool(k % 8 < 4), global_step=100 + (k // 4) % 2)
        if algo == "sacd":
            kw.update(Ad=max(2, kw["Ad"]), tau=float(rng.choice([1.0, 0.3])), autotune=bool(k % 8 < 4))
        if algo == "gpipd":
            kw.update(n_support=int(rng.choice([1, 3])), per=bool(k % 8 < 4), n_updates=(k // 4) % 2,
                      layer_norm=bool(k % 16 < 12), drop_rate=0.01 if k % 16 < 12 else 0.0)
        out.append(ACCase(f"acfuzz{k}_{algo}_B{B}_{arch[0]}x{arch[1]}", algo, **kw))
    return out


AC_FUZZ = _random_ac_cases(24)
AC_SMALL = sorted(AC_FUZZ, key=lambda c: c.B * sum(c.arch))[:3]


@pytest.mark.parametrize("c", AC_FUZZ, ids=lambda c: c.name)
def test_random_actor_critic_shapes_match_the_oracle(ac_be, c):
    lib, dev = ac_be
    if dev.type == "cpu" and c not in AC_SMALL:
        pytest.skip("the emulator runs the three smallest shapes; the rest need the GPU")
    run_and_check_against_oracle(lib, dev, c)


# ---- GPIPD.update (conditioned Q-networks with Dropout / LayerNorm, envelope targets over a support set) ---------------------
from cases_gpi import GpiCase  # noqa: E402
from test_gpi_kernels_parity import run_and_check_against_oracle as gpi_check  # noqa: E402
from test_gpi_kernels_parity import be as gpi_be  # noqa: E402,F401


def _random_gpi_cases(n, seed=909):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        L = int(rng.integers(2, 5))
        arch = tuple(int(4 * rng.integers(6, 65)) for _ in range(L))
        if k % 4 == 0:
            arch = tuple(256 for _ in range(L))
        B = int(rng.choice([1, 2, 7, 16, 17, 33, 64, 100, 128]))
        out.append(GpiCase(f"gpifuzz{k}_B{B}_{'x'.join(map(str, arch))}", D=int(rng.integers(1, 20)), A=int(rng.integers(2, 8)),
                           R=int(rng.integers(2, 5)), arch=arch, B=B, n_support=int(rng.choice([1, 2, 5])),
                           gpi_pd=bool(k % 5 != 4), step=int(rng.integers(1, 6)), layer_norm=bool(k % 6 != 5),
                           drop_rate=0.01 if k % 6 != 5 else 0.0, max_grad_norm=[-1.0, 0.5][k % 2], seed=700 + k))
    return out


GPI_FUZZ = _random_gpi_cases(16)
GPI_SMALL = sorted(GPI_FUZZ, key=lambda c: c.B * c.n_support * sum(c.arch))[:2]


@pytest.mark.parametrize("c", GPI_FUZZ, ids=lambda c: c.name)
def test_random_gpi_shapes_match_the_oracle(gpi_be, c):
    lib, dev = gpi_be
    if dev.type == "cpu" and c not in GPI_SMALL:
        pytest.skip("the emulator runs the two smallest shapes; the rest need the GPU")
    gpi_check(lib, dev, c)
