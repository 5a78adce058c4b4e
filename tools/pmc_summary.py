"""Summarise rocprofv3 --pmc CSVs (counter_collection) per kernel: mean per dispatch."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict


def source_hash():
    """Same stamp as bench.py's: the counters are only quoted for the kernel sources they were taken from."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(root, "versalignlib_amd", "csrc", "*kernel*.hip*"))):      # as bench.py: device code only
        if os.path.isfile(path):
            h.update(os.path.basename(path).encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"]
        if "valign" not in name:
            continue
        short = name.split("(")[0].replace("void valign::", "")
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
out = {"pairs": pairs, "csrc_sha16": source_hash(), "units": "mean per dispatch; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them",
       "kernels": {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in acc.items()}}
print(json.dumps(out, indent=1))
