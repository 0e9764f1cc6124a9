"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM bytes per launch.
Corrections as /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3): counters are in KB; on gfx950
FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled; WRITE_SIZE is exact.
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections, csv, json, re, sys

def per_kernel(path, counter):
    tot = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        tot[name] += float(r["Counter_Value"]); disp[name].add(r["Dispatch_Id"])
    return {k: (tot[k] / len(disp[k]), len(disp[k])) for k in tot}

f = per_kernel(sys.argv[1], "FETCH_SIZE"); w = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(f, key=lambda k: -f[k][0] * f[k][1]):
    if k not in w or k.startswith("at::") or "rocclr" in k:
        continue
    out[k] = {"launches": f[k][1], "fetch_KB_raw_avg": round(f[k][0], 1), "write_KB_avg": round(w[k][0], 1),
              "hbm_MB_per_launch_corrected": round((2 * f[k][0] + w[k][0]) / 1024, 1)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in list(out.items())[:8]:
    print(k, v)
