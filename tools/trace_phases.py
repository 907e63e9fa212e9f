"""Per-position kernel durations of one update from a rocprofv3 kernel trace: groups the morl:: kernels by their position inside
one step (the launch whose name contains argv[2] opens a step: default step_prologue, the Envelope step; ac_inputs for the
actor-critic updates) and prints the mean duration of each position over the steps seen."""
import csv, sys, collections
OPENER = sys.argv[2] if len(sys.argv) > 2 else "step_prologue"
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    if "morl::" not in n: continue
    if OPENER in n:
        cur = []; steps.append(cur)
    if cur is not None:
        cur.append((n.split("(")[0].replace("void ", "")[:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
steps = [s for s in steps[5:-1]]
if not steps:
    sys.exit(f"no launch named *{OPENER}* opens a step in this trace")
import statistics
# the step shapes seen (a bench run mixes them: e.g. the stand-alone updates and the iterations of the prioritised loop, which carry the
# tree update and the next iteration's sampling); per position the MEDIAN duration and the mean gap without its largest twentieth -- a trace has the odd multi-hundred-microsecond host hiccup
for L, cnt in collections.Counter(len(s) for s in steps).most_common(2):
    if cnt < 5:
        continue
    sel = [s for s in steps if len(s) == L]
    print(f"{len(sel)} steps of {L} launches (medians)")
    for k in range(L):
        d = [s[k][1] for s in sel]
        gap = [(s[k][2] - s[k - 1][3]) / 1e3 for s in sel] if k else [0.0]
        gap = sorted(gap)[:max(1, len(gap) - max(1, len(gap) // 20))]          # (mean without the largest twentieth: the hiccups)
        print(f"{k:2d} {sel[0][k][0]:40s} {statistics.median(d):8.2f} us   gap before {sum(gap) / len(gap):6.2f} us")
    tot = [(s[-1][3] - s[0][2]) / 1e3 for s in sel]
    print(f"first start -> last end: {statistics.median(tot):.1f} us (median), kernels {sum(statistics.median([s[k][1] for s in sel]) for k in range(L)):.1f} us")
    print()
