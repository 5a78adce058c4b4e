#!/bin/bash
# Copy what tools/final_batch.sh <tag> left under gpurun_out/final_<tag>/ into profiles/ (the files the docs cite) and
# check that the PMC stamps equal the hash of the kernel sources in the tree.  Usage: tools/collect_final.sh r02
set -eu
TAG=$1
O=gpurun_out/final_$TAG
for f in pmc_final bench_final bench_long pmc_align pmc_long; do cp $O/$f.json profiles/${TAG}_$f.json; done
cp $O/pmc_long_affine.json profiles/${TAG}_pmc_long_affine.json
[ -f $O/pmc_long_c5.json ] && cp $O/pmc_long_c5.json profiles/${TAG}_pmc_long_c5.json
[ -f $O/pmc_long_c5_affine.json ] && cp $O/pmc_long_c5_affine.json profiles/${TAG}_pmc_long_c5_affine.json
for f in align_bench latency long_reads d2h_rows_ab; do [ -f $O/$f.txt ] && cp $O/$f.txt profiles/${TAG}_$f.txt; done
[ -f $O/pmc_align_dispatches.txt ] && cp $O/pmc_align_dispatches.txt profiles/${TAG}_pmc_align_dispatches.txt
cp "$(ls -t $O/prof_bench/runc/*kernel_stats.csv | head -1)" profiles/${TAG}_bench_kernel_stats.csv
cp "$(ls -t $O/prof_align/runc/*kernel_stats.csv | head -1)" profiles/${TAG}_align_kernel_stats.csv
ls $O/prof_long/runc/*kernel_stats.csv > /dev/null 2>&1 && cp "$(ls -t $O/prof_long/runc/*kernel_stats.csv | head -1)" profiles/${TAG}_bench_long_kernel_stats.csv
python - <<PY
import json, bench
h = bench.source_hash()
for f in ("profiles/${TAG}_pmc_final.json", "profiles/${TAG}_pmc_align.json"):
    s = json.load(open(f))["csrc_sha16"]
    print(f, s, "OK" if s == h else "STALE (sources: %s)" % h)
b = json.loads(open("profiles/${TAG}_bench_final.json").read().strip().splitlines()[-1])
print("headline", b["value"], b["ms_per_step"], "f16", b["half_float"]["kernel_ms"], "linear", b["linear_gap"]["kernel_ms"])
print("alignments", b["alignments"])
a = b["abi"]
print("abi score", a["score_alignments_sw"]["ms"], "fresh", a["compute_alignments_sw"]["ms_fresh_rows"], "bare new", a["compute_alignments_sw"]["ms_2n_fresh_new_rows_alone"],
      "recycled", a["compute_alignments_sw"]["ms_recycled_rows"], "flat", a["compute_alignments_sw_flat_buffers"]["ms"], a["reference_protocol_config0"])
l = json.loads(open("profiles/${TAG}_bench_long.json").read().strip().splitlines()[-1])
print("long", l["value"], l["ms_per_step"], "cpu", l["cpu_baseline"]["value"])
PY
cat profiles/${TAG}_align_bench.txt
