#!/bin/bash
# Round 4, GPU run 38: per-bounce masks cleared by their consumer instead of per batch: reuse check, frame hashes against the previous build's, parity tests, timing
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r04
out=gpurun_out/r04/mask_reuse.txt; : > $out
python3 scripts/dev_mask_reuse_check.py 2>&1 | grep -v amdgpu.ids >> $out
for mesh in icosphere clustered; do echo "hash $mesh 12 spp: $(MIRRES_MESH=$mesh python3 scripts/dev_frame_hash.py 12 2>/dev/null | tail -1)" >> $out; done
echo "(previous build: icosphere 1463d07024c2 260a3cab13f2 76fb5918851e 042703fc6d27 4d9d01b086c9 d73039a6b389, clustered 8014d2c7d1fa 4a9fa0a44b54 2e2d8b7021df 07dc6d49a048 8c63957f474b 190c57172c93)" >> $out
one() { python3 bench.py --mesh $1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms')"; }
for rep in 1 2; do for mesh in icosphere clustered; do echo "mesh $mesh 512 spp: $(one $mesh)" >> $out; done; done
cat $out
( time timeout 900 python3 -m pytest tests/test_gpu_render.py -m gpu -q ) 2>&1 | tail -4 | cut -c1-200
