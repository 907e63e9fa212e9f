"""``bench.py --gpus 2`` end to end without a GPU: the self-spawn under ``torch.distributed.run`` (rendezvous on 127.0.0.1), the
process group, the sharded agents on BOTH axes, the weak-scaled sub-record, the collectives-alone side record and the one JSON
line on stdout -- with the ranks on the host-emulated kernel library over gloo (``--cpu-emulator``).  What the driver's first
multi-GPU run exercises in set-up and line assembly, minus RCCL itself."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(transport):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import simlib
    env = dict(os.environ, MORL_HIP_LIB=simlib.build_sim(), MORL_LAZY_MIN_ROWS="0", MORL_BF_MIN_ROWS="0", OMP_NUM_THREADS="1")
    env.pop("MORL_COMM", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-emulator", "--transport", transport, "--steps", "2",
           "--warmup", "1", "--batch", "8", "--weights", "4", "--buffer-fill", "200"]
    if transport != "staged":          # (the second leg is about the communicator the line reports: no sub-record, no side record again)
        cmd += ["--no-sub-record", "--no-collective-microbench"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]                  # exactly ONE line on stdout: rank 0's JSON
    return json.loads(lines[0]), r.stderr


@pytest.mark.parametrize("transport", ["staged", "auto"])
def test_bench_two_ranks_end_to_end_on_the_emulator(transport):
    d, err = _run(transport)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["metric"].startswith("Envelope-Q TD updates/sec") and d["unit"] == "TD-updates/s" and d["value"] > 0
    assert "cpu_emulator" in d and "FUNCTIONAL" in d["cpu_emulator"]
    # both partitions of the strong-scaled job were run and are in the line; the headline is one of them
    axes = d["strong_scaling_axes"]
    assert set(axes) == {"batch", "weights"} and all("error" not in v for v in axes.values())
    assert d["config"]["shard_axis"] in axes
    for ax, v in axes.items():
        assert v["ms_per_step"] > 0 and v["weights_per_gpu"] == 2 and v["shard_axis"] == ax
        assert v["last_loss"] == axes["batch"]["last_loss"] or abs(v["last_loss"] - axes["batch"]["last_loss"]) <= 1e-4 * abs(v["last_loss"])
    # transport and rank count as the library's communicator reports them (morl_comm_size); none for the staged path
    if transport == "staged":
        assert "staged" in d["config"]["transport"] and d["config"]["comm_ranks"] is None
    else:
        assert "torch" in d["config"]["transport"] and d["config"]["comm_ranks"] == 2
    assert d["config"]["rccl_ranks"] is None                   # (gloo: RCCL never came up, and the line says so)
    assert d["config"]["strong_scaling_ceiling_emulated"]["source"].startswith("profiles/")
    if transport != "staged":
        assert "weak_scaling" not in d and "collectives_alone" not in d
        return
    # the weak-scaled sub-record (weight axis grown to 4 * 2) and the collectives alone (single-hop transport over shared memory)
    w = d["weak_scaling"]
    assert "error" not in w and w["weights"] == 8 and w["weights_per_gpu"] == 4 and w["scaling"] == "weak"
    c = d["collectives_alone"]
    assert "ipc" in c and "error" not in c["ipc"], c
    assert c["ipc"]["allreduce_us"] > 0 and c["ipc"]["allgather_us"] > 0 and c["ipc"]["allgather_correct"] is True
    assert "[bench] rank 0/2" in err and "[bench] rank 1/2" in err
