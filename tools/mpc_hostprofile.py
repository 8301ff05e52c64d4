"""cProfile of the config-4 host loop (512 scenarios x 50 re-solves, warm): where the wall time outside the kernels goes."""
import cProfile, pstats, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / 'ms-eetc_amd'))
from mseetc import workloads as wl
from mseetc.mpc import shrinkingHorizon
train, track, N = wl.config('c4')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = wl.c1_times(B, seed=20260615)
shrinkingHorizon(train, track, wl.options(N), T[:64], numResolves=2, noise=0.01, seed=1, warmStart=True)
pr = cProfile.Profile(); pr.enable()
shrinkingHorizon(train, track, wl.options(N), T, numResolves=50, noise=0.01, seed=1, warmStart=True)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
