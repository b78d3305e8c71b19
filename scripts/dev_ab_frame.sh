#!/bin/bash
# Dev (GPU box): frame-level A/B of library variants under ab/ (built with MIRRES_BUILD_TAG): [MESH=clustered] [SPP=128] scripts/dev_ab_frame.sh variant...   (interleaved, two rounds)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
one() { python3 bench.py --mesh ${MESH:-icosphere} --no-extras --spp ${SPP:-128} --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"; }
for i in 1 2; do
  echo "base  $(one)"
  for v in "$@"; do echo "$v  $(MIRRES_LIB=$PWD/ab/libmirres_$v.so one)"; done
done
