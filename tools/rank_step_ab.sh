#!/bin/bash
# Same-box A/B of one rank of an N-rank strong-scaled job run alone (bench.py --force-shard --emulate-world N), with the few-row
# split-bf16 chain (mlp_chain_bfn.h, the default below 4 096 rows) against the round-5 engines (MORL_BFN_MAX_ROWS=0):
#   tools/rank_step_ab.sh OUTDIR
out=$1; mkdir -p "$out"
B="--no-cpu-baseline --no-ramp-record --no-sustained-record --no-exact-record --steps 100 --warmup 20"
# legs: r5 = the round-5 engines; bfn2 = the few-row chain for the two online passes and the backward pass, the target pass of an eager step on
# the f32 tiles (MORL_BFN_EAGER3=0); bfn = the default (the target pass rides as a third few-row chain)
for n in 8 4 2; do for ax in batch weights; do for leg in r5 bfn2 bfn; do
  unset MORL_BFN_MAX_ROWS MORL_BFN_EAGER3
  if [ $leg = r5 ]; then export MORL_BFN_MAX_ROWS=0; fi
  if [ $leg = bfn2 ]; then export MORL_BFN_EAGER3=0; fi
  timeout 200 python bench.py --gpus 1 --force-shard --emulate-world $n --shard-axis $ax $B > "$out/emu${n}_${ax}_$leg.json" 2> "$out/emu${n}_${ax}_$leg.err"
done; done; done
unset MORL_BFN_MAX_ROWS MORL_BFN_EAGER3
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
