import csv, glob, sys
root = sys.argv[1]
for path in glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(path)))
    print(path, len(rows), "copies; columns:", list(rows[0].keys()))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    big = [r for r in rows if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 100000]
    for r in big[-45:]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print("  %-28s start %9.3f ms dur %7.3f ms" % (r.get("Direction", r.get("Kind", "?")), s / 1e6, (e - s) / 1e6))
for path in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(path)))
    rows = [r for r in rows if "score_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    print(path, len(rows), "score kernels")
    for r in rows[-16:]:
        print("  kernel start %d dur %.3f ms" % (int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
