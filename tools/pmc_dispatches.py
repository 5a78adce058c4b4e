"""Per-dispatch counter values of the alignment-path PMC passes (tools/pmc_align.sh): traceback_kernel is one kernel
name for every mode, so the per-kernel means of tools/pmc_summary.py average linear and affine walks -- this prints
each dispatch (order of tools/align_bench.py: SW linear, NW linear, SW affine, NW affine; warm-up + timed run each).
Usage: python tools/pmc_dispatches.py gpurun_out/pmc_align"""
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    table = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU"):
        for d in sorted(glob.glob(os.path.join(root, "p*"))):
            files = glob.glob(os.path.join(d, "runc", "*counter_collection.csv"))
            if not files:
                continue
            f = max(files, key=os.path.getmtime)
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != counter:
                    continue
                name = r["Kernel_Name"]
                if "traceback" not in name and "align_fill" not in name:
                    continue
                key = (int(r["Dispatch_Id"]), name.replace("void valign::", "").split("(")[0])
                table.setdefault(key, {})[counter] = float(r["Counter_Value"])
    print("# dispatch  kernel  FETCH_SIZE[KB]  WRITE_SIZE[KB]  GRBM_GUI_ACTIVE (sum over 8 XCDs)  SQ_INSTS_VALU")
    print("# HBM bytes = 2 x FETCH_SIZE (gfx950: 64-byte units counted as 32) + WRITE_SIZE, both in KB")
    # dispatch ids differ between passes (one process per pass): align by order of appearance per pass instead
    by_counter = {}
    for (disp, name), vals in sorted(table.items()):
        for c, v in vals.items():
            by_counter.setdefault(c, []).append((disp, name, v))
    n = max(len(v) for v in by_counter.values())
    for i in range(n):
        name = None
        cols = []
        for c in ("FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU"):
            lst = by_counter.get(c, [])
            if i < len(lst):
                name = lst[i][1]
                cols.append("%.4g" % lst[i][2])
            else:
                cols.append("-")
        print("%2d  %-62s %s" % (i, name, "  ".join(cols)))


if __name__ == "__main__":
    main()
