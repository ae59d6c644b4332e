"""Reads the per-pass counter CSVs of scripts/pmc_run.sh and writes a per-kernel summary (JSON + text)."""
import csv, glob, json, os, re, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import sources_hash
src, out_json = sys.argv[1], sys.argv[2]
command = " ".join(sys.argv[3:])
acc = defaultdict(lambda: defaultdict(float)); calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        m = re.search(r"(k_\w+)", name)
        if not m:
            continue                # kernels of other libraries (the device-side scene generator of the largest configuration)
        k = m.group(1)
        if k in ("k_trace_persistent", "k_trace_dual", "k_wf_extend") and (k + "<true") in name:
            k += "<detailed>"
        if k == "k_refit_low" and "k_refit_low<true" in name:
            k += "<pack>"          # the default build: the refit also writes the trace kernels' leaf records
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]] += 1
res = {}
for k in acc:
    res[k] = {c: acc[k][c] / max(1, calls[k][c]) for c in acc[k]}
    res[k]["_launches_seen"] = max(calls[k].values())
    if "FETCH_SIZE" in res[k] and "WRITE_SIZE" in res[k]:
        # FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B read requests as 64 B,
        # i.e. reads exactly half of a wide streaming read: doubled here (upper bound for narrow gathers)
        res[k]["hbm_bytes_per_launch"] = (2.0 * res[k]["FETCH_SIZE"] + res[k]["WRITE_SIZE"]) * 1024.0
        res[k]["hbm_bytes_per_launch_uncorrected"] = (res[k]["FETCH_SIZE"] + res[k]["WRITE_SIZE"]) * 1024.0
# which kernel sources the counters belong to (bench.py withholds figures of a profile taken with other sources)
res["_meta"] = {"sources_sha256_16": sources_hash(), "command": command,
                "collected_by": "scripts/pmc_run.sh: one rocprofv3 --pmc run per counter group; values are averages per launch"}
json.dump(res, open(out_json, "w"), indent=1, sort_keys=True)
for k in sorted(res):
    if k != "_meta":
        print(k, {c: round(v, 1) for c, v in res[k].items()})
