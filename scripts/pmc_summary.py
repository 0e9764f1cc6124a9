"""Per-kernel means of every counter in a rocprofv3 counter_collection.csv -> JSON.
usage: pmc_summary.py <counter_collection.csv> <out.json> [kernel-name-substring ...]"""
import collections, csv, json, re, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
    if name.startswith("at::") or "rocclr" in name:
        continue
    tot[name][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[name].add(r["Dispatch_Id"])
want = sys.argv[3:]
out = {}
for k in sorted(tot, key=lambda k: -len(disp[k])):
    if want and not any(w in k for w in want):
        continue
    n = len(disp[k])
    out[k] = {"launches": n, **{c: v / n for c, v in sorted(tot[k].items())}}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in list(out.items())[:10]:
    print(k, {c: (round(x, 1) if isinstance(x, float) else x) for c, x in v.items()})
