"""profiles/rNN_emulated_ceiling.json from bench lines of ONE build: python tools/emulated_ceiling.py <single.json> <dir with
emu{2,4,8}_{batch,weights}.json> <out.json>.  Strong-scaling ceiling = single-GPU step / one rank's step (run alone: no collective
costs a microsecond)."""
import json
import os
import sys


def main():
    single, d, out = sys.argv[1:4]
    t1 = json.load(open(single))["ms_per_step"]
    rec = {"what": "one rank of an N-rank strong-scaled 256 x 64 x 3 job run ALONE on one MI355X (bench.py --force-shard --emulate-world N): "
                   "single-GPU ms / rank ms = the ceiling of strong scaling before any collective",
           "single_gpu_ms": t1, "rank_ms": {"batch_axis": {}, "weight_axis": {}}, "batch_axis": {}, "weight_axis": {}, "pipeline": {}}
    for n in (2, 4, 8):
        for ax, key in (("batch", "batch_axis"), ("weights", "weight_axis")):
            r = json.load(open(os.path.join(d, f"emu{n}_{ax}.json")))
            rec["rank_ms"][key][str(n)] = r["ms_per_step"]
            rec[key][str(n)] = round(t1 / r["ms_per_step"], 3)
            rec["pipeline"][f"{ax}{n}"] = {"dtype": r["dtype"][:40], "lazy_rows": r.get("lazy_target_rows_last_step"),
                                           "host_enqueue_ms": r.get("host_enqueue_ms_per_step")}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
