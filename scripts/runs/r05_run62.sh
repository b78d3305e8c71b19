#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05 gpurun_out/pf
{ echo "# k_spatial_gen: one queue atomic per block on ONE head word. Blocks of 16 x 16 px, 1 px per thread (base: 10 000 blocks at 1600 x 1600), 32 x 32 px / 1 px per thread (SG32: 2 500 blocks of"
  echo "# 1024 threads), 32 x 32 / 2 px per thread (SG32P2), 16 x 16 / 2 px per thread (SG16P2: 128 threads). Frame at 128 spp (3 + 1 frames), then the kernel's own time in a serialised 8-spp frame."
  bash scripts/dev_ab_frame.sh SG32 SG32P2 SG16P2
  echo "# lego-like"; MESH=clustered bash scripts/dev_ab_frame.sh SG32 SG32P2 SG16P2
  for v in base SG32 SG32P2 SG16P2; do
    rm -rf gpurun_out/pf/kt
    if [ $v = base ]; then L=""; else L="$PWD/ab/libmirres_$v.so"; fi
    MIRRES_LIB=$L MIRRES_STREAMS=1 timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pf/kt -o kt -- python3 bench.py --spp 8 --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-roofline > /dev/null 2>&1
    echo "$v serialised: $(grep -h 'k_spatial_gen' $(find gpurun_out/pf/kt -name '*kernel_stats.csv' | head -1) | python3 -c 'import sys,csv; r=list(csv.reader(sys.stdin))[0]; print("k_spatial_gen calls", r[1], "avg us", float(r[3])/1e3, "min", float(r[5])/1e3)')"
  done; } 2>&1 | tee gpurun_out/r05/ab_sgen_tiles.txt
rm -rf gpurun_out/pf
