"""cProfile of config 5's pilot equaliser + filter stage (bench.pilot_chain) on the GPU."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from qampy_amd import synth, _lib
_lib.init(0)
cap = synth.make_pilot_capture(nframes=9) if "nframes" in synth.make_pilot_capture.__code__.co_varnames else synth.make_pilot_capture()
for _ in range(2):
    r = bench.pilot_chain(cap)
print(r["stages_ms"], r["frames"])
pr = cProfile.Profile(); pr.enable(); r = bench.pilot_chain(cap); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
