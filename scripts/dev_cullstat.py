"""Experiment (library built with MIRRES_PASSES_FLAGS=-DMR_EXP_CULLSTAT, loaded through MIRRES_LIB): how many shadow rays of the spatial pass cannot change the merge.
Runs one bench frame (bench.py's arguments are passed through) and prints the counters of k_spatial_resolve."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]] + (sys.argv[1:] or ["--spp", "64", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline"])
import bench
bench.main()
from mirres_restir_nerf_mesh_amd._lib import lib
out = (C.c_ulonglong * 8)()
fn = lib().mirres_dev_cullstat
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
assert fn(out, 1) == 0
n = out[0]
names = ["neighbour merges (2 rays each)", "ray 1: target of the neighbour's sample at the canonical pixel is 0", "ray 1: ... or the neighbour's weight is 0",
         "ray 2: target of the canonical sample at the neighbour is 0", "ray 2: ... or canonical target / weight is 0", "ray 1: light below the canonical horizon",
         "ray 2: light below the neighbour's horizon", "ray 1 occluded"]
for k in range(8):
    print("%-75s %12d  %.4f" % (names[k], out[k], out[k] / max(1, n)))
