"""Device timeline of the LAST host-pointer call in a rocprofv3 --kernel-trace --memory-copy-trace output directory:
per category (fill / walk / other kernels, H2D, D2H) the busy time, and the idle gaps of the union."""
import csv
import glob
import sys

root = sys.argv[1]
span_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
ev = []
for path in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        kind = "fill" if "align_fill" in name else ("walk" if "traceback" in name else ("score" if "score_" in name else ("unpack" if "unpack" in name else "other kernel")))
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
for path in glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        d = r.get("Direction", r.get("Kind", "?"))
        kind = "H2D" if "HOST_TO_DEVICE" in d.upper() or "H2D" in d.upper() else ("D2H" if "DEVICE_TO_HOST" in d.upper() or "D2H" in d.upper() else d)
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
ev.sort()
end = ev[-1][1]
ev = [e for e in ev if e[0] >= end - span_ms * 1e6]
# the last call: events after the last idle gap longer than 5 ms
start_idx = 0
reach = ev[0][1]
for i, e in enumerate(ev):
    if e[0] - reach > 5e6:
        start_idx = i
    reach = max(reach, e[1])
ev = ev[start_idx:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
print("last call: %d device operations over %.2f ms" % (len(ev), (t1 - t0) / 1e6))
kinds = sorted({e[2] for e in ev})
for k in kinds:
    iv = sorted((e[0], e[1]) for e in ev if e[2] == k)
    busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    durs = [e - s for s, e in iv]
    print("  %-14s %4d ops, busy %7.2f ms (sum %7.2f), first at %6.2f ms, last ends %6.2f ms, mean %.3f ms" %
          (k, len(iv), busy / 1e6, sum(durs) / 1e6, (iv[0][0] - t0) / 1e6, (max(e for _, e in iv) - t0) / 1e6, sum(durs) / len(durs) / 1e6))
iv = sorted((e[0], e[1]) for e in ev)
gaps, cur_e = [], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        gaps.append((cur_e - t0, s - cur_e))
    cur_e = max(cur_e, e)
print("  idle gaps of the whole device: %d, total %.2f ms; largest: %s" %
      (len(gaps), sum(g[1] for g in gaps) / 1e6, ", ".join("%.2f ms at %.2f" % (g[1] / 1e6, g[0] / 1e6) for g in sorted(gaps, key=lambda g: -g[1])[:6])))
if len(sys.argv) > 3:
    for s, e, k in ev:
        print("    %-12s %8.3f -> %8.3f" % (k, (s - t0) / 1e6, (e - t0) / 1e6))
