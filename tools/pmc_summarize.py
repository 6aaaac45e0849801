"""Aggregate a rocprofv3 --pmc run per kernel: python tools/pmc_summarize.py <rocprof output dir> <out.json> [name filter regex]
Reads every *counter_collection.csv under the directory (one row per dispatch and counter), sums each counter and counts the
dispatches per (shortened) kernel name; raw CSVs of a whole bench run are far beyond the 64 MiB that travel back from the GPU box."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
pat = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None


def short(name: str) -> str:
    name = re.sub(r"\(.*", "", name.replace("void ", ""))
    return name[:160]


agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = short(row.get("Kernel_Name", row.get("Kernel Name", "?")))
            if pat and not pat.search(k):
                continue
            c = row["Counter_Name"]
            agg[k][c] += float(row["Counter_Value"])
            calls[k][c] += 1
res = {k: {"dispatches": max(calls[k].values()), **{c: v for c, v in cs.items()}} for k, cs in agg.items()}
json.dump(dict(sorted(res.items(), key=lambda kv: -kv[1]["dispatches"])), open(out, "w"), indent=1)
print(f"{len(res)} kernels -> {out}")
