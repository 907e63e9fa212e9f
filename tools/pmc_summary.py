"""Collect the per-kernel PMC figures behind bench.py's `roofline.traffic` (run ON the GPU box):

    cd /tmp && export TMPDIR=/tmp && python $REPO/tools/pmc_summary.py $REPO/gpurun_out/pmc_summary.json

Three separate `rocprofv3 --pmc` passes of `python bench.py --steps 10 --warmup 3 --no-cpu-baseline` (FETCH_SIZE, WRITE_SIZE
and an SQ set do not fit one pass; counter passes are never combined with the trace domains), averaged per launch and kernel.
FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950 (it reports
TCC_EA0_RDREQ x 64 B for 128-B requests); WRITE_SIZE is uncalibrated and reported as counted (KB -> bytes).
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the profiled command: bench.py by default, or `pmc_summary.py out.json <script under the repo root> <its arguments...>`
COMMAND = [os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-ramp-record",
           "--no-sustained-record", "--no-exact-record"]
PASSES = {"fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"],
          "sq": ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"]}
# a second SQ pass (optional: a counter name this rocprofv3 does not know fails the pass, which is then skipped): where the waves'
# cycles go -- parked at s_waitcnt / s_barrier (WAIT_ANY), issue-stalled (WAIT_INST_ANY), issuing (ACTIVE_INST_ANY); LDS activity
EXTRA_PASSES = {"sq2": ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_IDX_ACTIVE",
                        "SQ_WAVE_CYCLES", "SQ_INSTS_VALU"]}


def run_pass(name, counters, outdir):
    d = os.path.join(outdir, name)
    cmd = ["rocprofv3", "--pmc", *counters, "--output-format", "csv", "-d", d, "--", sys.executable, *COMMAND]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    dur = collections.defaultdict(dict)           # kernel -> dispatch -> ns (when the counter pass carries timestamps)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k].add(r["Dispatch_Id"])
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                dur[k][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    out = {}
    for k, cs in acc.items():
        out[k] = {c: v / len(launches[k]) for c, v in cs.items()} | {"launches": len(launches[k])}
        if dur[k]:
            out[k]["avg_ns_in_pass"] = sum(dur[k].values()) / len(dur[k])
    return out


def run_trace(outdir):
    """Average launch duration per kernel (ns) from a plain kernel-trace pass of the same command: joined with
    GRBM_GUI_ACTIVE it gives the clock the chip sustained INSIDE each kernel (DVFS: MI355X_MICROARCH.md, 'DVFS give-back')."""
    d = os.path.join(outdir, "trace")
    cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", sys.executable, *COMMAND]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Name"].split("(")[0].replace("void ", "")] = float(r["AverageNs"])
    return out


def main():
    global COMMAND
    out_path = sys.argv[1]
    if len(sys.argv) > 2:
        COMMAND = [os.path.join(ROOT, sys.argv[2]), *sys.argv[3:]]
    work = os.path.join(os.environ.get("TMPDIR", "/tmp"), "pmc_passes")
    res = {n: run_pass(n, c, work) for n, c in PASSES.items()}
    for n, c in EXTRA_PASSES.items():
        try:
            res[n] = run_pass(n, c, work)
        except Exception as exc:                       # (unknown counter, slot overflow: the summary goes on without it)
            print(f"[pmc_summary] optional pass {n} skipped: {exc}", file=sys.stderr)
            res[n] = {}
    avg_ns = run_trace(work)
    kernels = {}
    for k in sorted(res["fetch"]):
        if not k.startswith("morl::"):
            continue
        f, w, s = res["fetch"][k], res["write"].get(k, {}), res["sq"].get(k, {})
        fetch = 2.0 * 1024.0 * f.get("FETCH_SIZE", 0.0)            # KB counted at 64 B per 128-B request -> bytes, doubled
        write = 1024.0 * w.get("WRITE_SIZE", 0.0)
        sq = {c: s.get(c, 0.0) for c in PASSES["sq"]}
        kernels[k] = {"launches": f["launches"], "fetch_bytes_corrected": fetch, "write_bytes": write,
                      "hbm_bytes": fetch + write,
                      # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs:
                      # busy fraction = busy / (active / 8 * 1024)
                      "mfma_busy_frac": (sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (sq["GRBM_GUI_ACTIVE"] * 128.0)) if sq["GRBM_GUI_ACTIVE"] else 0.0,
                      "lds_bank_conflict_over_wave_cycles": (sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_WAVE_CYCLES"]) if sq["SQ_WAVE_CYCLES"] else 0.0,
                      "avg_launch_ns_unprofiled": avg_ns.get(k),
                      # GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles per XCD / launch duration = sustained clock
                      "effective_clock_ghz": (sq["GRBM_GUI_ACTIVE"] / 8.0 / avg_ns[k]) if avg_ns.get(k) else None,
                      "avg_launch_ns_in_counter_pass": s.get("avg_ns_in_pass"),
                      "effective_clock_ghz_in_counter_pass": (sq["GRBM_GUI_ACTIVE"] / 8.0 / s["avg_ns_in_pass"]) if s.get("avg_ns_in_pass") else None,
                      "sq": sq, "sq2": {c: v for c, v in res.get("sq2", {}).get(k, {}).items() if c != "launches"}}
    json.dump({"note": __doc__.strip().split("\n\n")[-1].replace("\n", " "), "command": "python " + " ".join([os.path.relpath(COMMAND[0], ROOT)] + COMMAND[1:]), "kernels": kernels}, open(out_path, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:45s} launches {v['launches']:4d}  HBM {v['hbm_bytes'] / 1e6:9.2f} MB  mfma_busy {v['mfma_busy_frac']:.3f}")


if __name__ == "__main__":
    main()
