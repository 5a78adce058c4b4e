#!/bin/bash
# Round-end measurement batch (run on the GPU box through gpurun): PMC passes, bench line,
# kernel-trace stats of bench.py and of the alignment path, geometry sweeps, host-ABI rates.
# Usage: tools/final_batch.sh <tag>
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
bash tools/pmc_passes.sh $TAG --modes sw_affine,sw_linear,nw_linear || exit 1
python tools/pmc_summary.py $OUT/pmc_$TAG > $OUT/pmc_${TAG}_summary.json || exit 1
cp $OUT/pmc_${TAG}_summary.json $R/profiles/r01_pmc_final.json   # bench.py reads traffic and VALU share from here
python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err || exit 1
echo "bench done"
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu > $OUT/prof_$TAG.log 2>&1) || exit 1
echo "bench trace done"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_align_$TAG -- python3 $R/tools/align_bench.py --iters 2 > $OUT/prof_align_$TAG.log 2>&1) || exit 1
echo "align trace done"
python tools/align_bench.py --iters 3 > $OUT/align_$TAG.log 2>&1 || exit 1
{ python tools/geom_sweep.py --n 1048576 --geoms 8x20,16x10,16x12,32x8,64x12 ;
  python tools/geom_sweep.py --n 1048576 --geoms 8x20,16x10,16x12,32x8,64x12 --affine 1 ; } > $OUT/geom_$TAG.log 2>&1 || exit 1
{ echo "# long reads: 2.5k x 5k (65536 pairs), 10k x 10k (32768 pairs = one GPU's share of BASELINE config 5), the same banded (512)";
  python tools/geom_sweep.py --R 2500 --F 5000 --n 65536 --iters 2 --geoms 0x0;
  python tools/geom_sweep.py --R 10000 --F 10000 --n 32768 --iters 1 --geoms 0x0;
  python tools/geom_sweep.py --R 10000 --F 10000 --n 32768 --iters 1 --geoms 0x0 --band 512;
  echo "# affine with four different scores (SW), NW affine";
  python tools/geom_sweep.py --n 1048576 --geoms 0x0 --affine 2;
  python tools/geom_sweep.py --n 1048576 --geoms 0x0 --affine 1 --opt 1; } 2>&1 | grep -v amdgpu.ids > $OUT/long_$TAG.log || exit 1
echo "sweeps done"
python tools/abi_bench.py --pairs 1048576 --align-pairs 262144 --threads 16 > $OUT/abi_$TAG.log 2>&1 || exit 1
echo "abi done"
# host-pointer path: phases, mixed-length batches, small-call latency, raw PCIe and gather rates
{ echo "# tools/host_path_profile.py --threads 8,16"; python tools/host_path_profile.py --threads 8,16;
  echo "# tools/abi_bench.py --pairs 1048576 --threads 16 --ragged"; python tools/abi_bench.py --pairs 1048576 --threads 16 --ragged;
  echo "# tools/latency_bench.py"; python tools/latency_bench.py;
  echo "# tools/microbench/pcie_rate.py"; python tools/microbench/pcie_rate.py;
  echo "# tools/microbench/gather_rate.hip"; hipcc -O2 --offload-arch=gfx950 -o /tmp/gather_rate tools/microbench/gather_rate.hip -lpthread && /tmp/gather_rate;
  echo "# nproc / cpu.max"; nproc; cat /sys/fs/cgroup/cpu.max; } 2>&1 | grep -v amdgpu.ids > $OUT/host_path_$TAG.log || exit 1
echo "host path done"
