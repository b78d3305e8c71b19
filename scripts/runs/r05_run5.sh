#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}" || exit 1
mkdir -p gpurun_out/r05
{ echo "## MR_GRAB_MODE 2 (load first after the first failure)"; python3 scripts/dev_wave_times.py 7 2>&1 | grep -v amdgpu.ids
  echo "## MR_GRAB_MODE 0 (blind atomics)"; MIRRES_LIB=$PWD/ab/libmirres_GRAB0.so python3 scripts/dev_wave_times.py 7 2>&1 | grep -v amdgpu.ids
  echo "## MR_GRAB_MODE 1"; MIRRES_LIB=$PWD/ab/libmirres_GRAB1.so python3 scripts/dev_wave_times.py 7 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r05/wave_times_grab.txt
cat gpurun_out/r05/wave_times_grab.txt
