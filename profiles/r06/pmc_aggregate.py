"""rocprofv3 counter_collection.csv of profiles/r06/one_forward.py -> per (kernel, grid) averages over the LAST forward's dispatches.
Usage: python profiles/r06/pmc_aggregate.py <counter_collection.csv> <out.json> [tag]      (appends / merges counters into out.json)"""
import collections, csv, json, os, re, sys

src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
# one record per dispatch
disp = collections.OrderedDict()
for r in rows:
    d = disp.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"], "grid": int(r["Grid_Size"]) if "Grid_Size" in r else 0,
                                                 "wg": int(r["Workgroup_Size"]) if "Workgroup_Size" in r else 0, "c": {}})
    d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(disp)
starts = [i for i in ids if "prep_image" in disp[i]["kernel"]]
if len(starts) < 2:
    print("pmc_aggregate: fewer than two forwards in", src); sys.exit(1)
fw = [i for i in ids if starts[-1] <= i]                      # the last forward (everything from its prep kernel on)


def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.split("(")[0]
    return n[:120]


acc = collections.OrderedDict()
for i in fw:
    d = disp[i]
    key = f"{short(d['kernel'])} | grid {d['grid'] // max(d['wg'], 1)} x {d['wg']}"
    a = acc.setdefault(key, {"dispatches_per_forward": 0, "sum": collections.defaultdict(float)})
    a["dispatches_per_forward"] += 1
    for k, v in d["c"].items():
        a["sum"][k] += v
out = json.load(open(dst)) if os.path.exists(dst) else {}
for key, a in acc.items():
    e = out.setdefault(key, {"dispatches_per_forward": a["dispatches_per_forward"]})
    for k, v in a["sum"].items():
        e[k] = round(v / a["dispatches_per_forward"], 3)
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(f"pmc_aggregate: {len(fw)} dispatches of the last forward, {len(acc)} (kernel, grid) classes -> {dst}")
