#!/bin/bash
# Round-end measurement batch (run on the GPU box through gpurun, AFTER the last kernel-source change: the PMC
# summary is stamped with a hash of versalignlib_amd/csrc/ and bench.py only quotes it for those sources).
# Everything lands under gpurun_out/final_<tag>/; copy what is to be judged into profiles/ afterwards.
# Usage: tools/final_batch.sh <tag>
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/final_$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
bash tools/pmc_passes.sh $TAG --modes sw_affine16,sw_affine,sw_linear,nw_linear || exit 1
python tools/pmc_summary.py $R/gpurun_out/pmc_$TAG > $OUT/pmc_final.json || exit 1
mkdir -p $R/profiles && cp $OUT/pmc_final.json $R/profiles/${TAG}_pmc_final.json     # on the box: lets the bench below quote it
python bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err || exit 1
echo "bench done"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-abi > $OUT/prof_bench.log 2>&1) || exit 1
echo "bench trace done"
# the long workload's kernel (BASELINE configs[4] per GPU: 32768 pairs of 10 kbp x 10 kbp, band 512, int32 cells): PMC first, so that its bench line can quote it
bash tools/pmc_long.sh c5 --R 10000 --F 10000 --n 32768 --band 512 --width 32 > $OUT/pmc_long_c5.log 2>&1
python tools/pmc_summary.py $R/gpurun_out/pmc_c5 32768 > $OUT/pmc_long_c5.json || exit 1
cp $OUT/pmc_long_c5.json $R/profiles/${TAG}_pmc_long_c5.json
bash tools/pmc_long.sh c5aff --R 10000 --F 10000 --n 32768 --band 512 --width 32 --affine 1 > $OUT/pmc_long_c5_affine.log 2>&1
python tools/pmc_summary.py $R/gpurun_out/pmc_c5aff 32768 > $OUT/pmc_long_c5_affine.json || exit 1
python bench.py --workload long --steps 5 --warmup 1 > $OUT/bench_long.json 2> $OUT/bench_long.err || exit 1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_long -- python3 $R/bench.py --workload long --steps 3 --warmup 1 --no-cpu > $OUT/prof_long.log 2>&1) || exit 1
echo "long bench done"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_align -- python3 $R/tools/align_bench.py --iters 2 > $OUT/prof_align.log 2>&1) || exit 1
python tools/align_bench.py --iters 3 2>&1 | grep -v amdgpu.ids > $OUT/align_bench.txt || exit 1
echo "align done"
bash tools/pmc_align.sh > $OUT/pmc_align.log 2>&1
python tools/pmc_summary.py $R/gpurun_out/pmc_align > $OUT/pmc_align.json || exit 1
echo "align pmc done"
{ echo "# tools/latency_phases.py (flat entry points, engine-reported phases)"; python tools/latency_phases.py;
  echo "# tools/latency_bench.py 1 (plugin ABI, num_threads = 1)"; python tools/latency_bench.py 1;
  echo "# tools/latency_bench.py 16 (plugin ABI, num_threads = 16: result rows of calls above 768 KB are copied by the pool)"; python tools/latency_bench.py 16;
  echo "# tools/first_call.py: cold process -> first results, and a second spawn in the warm process"; python tools/first_call.py; } 2>&1 | grep -v amdgpu.ids > $OUT/latency.txt || exit 1
{ echo "# long reads: 2.5k x 5k (65536 pairs), 10k x 10k (32768 pairs = one GPU's share of BASELINE config 5), banded (512), affine; then the round-3 schedules of the same (VALIGN_HIP_DEBUG=short_strips / no_band_chain)";
  python tools/geom_sweep.py --R 2500 --F 5000 --n 65536 --iters 2 --geoms 0x0;
  python tools/geom_sweep.py --R 10000 --F 10000 --n 32768 --iters 1 --geoms 0x0;
  python tools/geom_sweep.py --R 10000 --F 10000 --n 32768 --iters 1 --geoms 0x0 --band 512;
  python tools/geom_sweep.py --R 10000 --F 10000 --n 32768 --iters 1 --geoms 0x0 --affine 1;
  python tools/geom_sweep.py --R 10000 --F 10000 --n 32768 --iters 1 --geoms 0x0 --affine 1 --band 512;
  VALIGN_HIP_DEBUG=short_strips python tools/geom_sweep.py --R 10000 --F 10000 --n 32768 --iters 1 --geoms 0x0;
  VALIGN_HIP_DEBUG=short_strips python tools/geom_sweep.py --R 10000 --F 10000 --n 32768 --iters 1 --geoms 0x0 --affine 1;
  VALIGN_HIP_DEBUG=no_band_chain python tools/geom_sweep.py --R 10000 --F 10000 --n 32768 --iters 1 --geoms 0x0 --affine 1 --band 512;
  echo "# affine with four different scores (SW), NW affine, 150 x 500";
  python tools/geom_sweep.py --n 1048576 --geoms 0x0 --affine 2;
  python tools/geom_sweep.py --n 1048576 --geoms 0x0 --affine 1 --opt 1; } 2>&1 | grep -v amdgpu.ids > $OUT/long_reads.txt || exit 1
echo "sweeps done"
bash tools/pmc_long.sh long_lin > $OUT/pmc_long.log 2>&1; python tools/pmc_summary.py $R/gpurun_out/pmc_long_lin 65536 > $OUT/pmc_long.json
bash tools/pmc_long.sh long_aff --affine 1 >> $OUT/pmc_long.log 2>&1; python tools/pmc_summary.py $R/gpurun_out/pmc_long_aff 65536 > $OUT/pmc_long_affine.json
echo "long pmc done"
{ echo "# tools/align_d2h_ab.py 16 / 6 (host threads): result rows packed on the device vs whole rows vs registered destination"; python tools/align_d2h_ab.py 16; python tools/align_d2h_ab.py 6; } 2>&1 | grep -v amdgpu.ids > $OUT/d2h_rows_ab.txt
echo "all done"
