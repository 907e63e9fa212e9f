#!/bin/bash
# Same-box A/B of one rank of an N-rank strong-scaled job run alone (bench.py --force-shard --emulate-world N), with the few-row
# split-bf16 chain (mlp_chain_bfn.h, the default below 4 096 rows) against the round-5 engines (MORL_BFN_MAX_ROWS=0):
#   tools/rank_step_ab.sh OUTDIR
out=$1; mkdir -p "$out"
B="--no-cpu-baseline --no-ramp-record --no-sustained-record --no-exact-record --steps 100 --warmup 20"
for n in 8 4 2; do for ax in batch weights; do for leg in r5 bfn; do
  if [ $leg = r5 ]; then export MORL_BFN_MAX_ROWS=0; else unset MORL_BFN_MAX_ROWS; fi
  timeout 200 python bench.py --gpus 1 --force-shard --emulate-world $n --shard-axis $ax $B > "$out/emu${n}_${ax}_$leg.json" 2> "$out/emu${n}_${ax}_$leg.err"
done; done; done
unset MORL_BFN_MAX_ROWS
python - "$out" <<'PY'
import glob, json, os, sys
res = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "emu*.json"))):
    name = os.path.basename(f)[:-5]
    try:
        b = json.load(open(f))
        res[name] = {"ms_per_step": b["ms_per_step"], "dtype": b["dtype"][:40], "lazy_rows": b.get("lazy_target_rows_last_step"),
                     "host_enqueue_ms": b.get("host_enqueue_ms_per_step"),
                     "per_kernel_us": {k: round(v["avg_launch_us"], 1) for k, v in b["roofline"]["per_kernel"].items()}}
    except Exception as e:
        res[name] = {"error": str(e)}
    print(name, res[name])
json.dump(res, open(os.path.join(sys.argv[1], "rank_step_bfn_ab.json"), "w"), indent=1)
PY
