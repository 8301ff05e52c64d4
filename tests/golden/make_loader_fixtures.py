"""
Generates tests/golden/loader_fixtures.json by importing the REFERENCE loaders
(/root/reference/mseetc/{train,track,utils,data}.py) in the build container.

The reference imports casadi at module level and casadi is not installable here,
so an empty stub module is registered first; only the pandas/numpy parts of the
reference (json loading, unit conversion, crop, merge, grid, clothoid sampling,
loss table) are executed.  Run once, in the build container:

    python tests/golden/make_loader_fixtures.py

The output is data only (inputs + expected outputs); it is committed, this script
never runs on the GPU box.
"""

import json
import sys
import types
import warnings
from pathlib import Path

warnings.simplefilter('ignore')

sys.modules['casadi'] = types.ModuleType('casadi')
sys.path.insert(0, '/root/reference')

import numpy as np  # noqa: E402

from mseetc.track import Track, computeDiscretizationPoints  # noqa: E402
from mseetc.train import Train  # noqa: E402
from mseetc.data import dataLosses  # noqa: E402
from mseetc.utils import convertUnit  # noqa: E402

TRAIN_ATTRS = ['g', 'mass', 'rho', 'velocityMax', 'forceMax', 'forceMin', 'forceMinPn', 'powerMax', 'powerMin',
               'accMax', 'accMin', 'r0', 'r1', 'r2', 'etaTraction', 'etaRgBrake']


def frame(df):
    return {'index': [float(x) for x in df.index.values],
            'columns': {c: [float(x) for x in df[c].values] for c in df.columns}}


def trainDict(t):
    return {a: getattr(t, a, None) for a in TRAIN_ATTRS}


out = {}

# ---- trains -----------------------------------------------------------------

out['train_default'] = trainDict(Train(config={'id': 'NL_Intercity_VIRM6'}))

cfg = {'id': 'NL_Intercity_VIRM6', 'max deceleration': None, 'max acceleration': {'unit': 'm/s^2', 'value': 0.45},
       'mass': {'unit': 't', 'value': 400.0}}
out['train_override'] = {'config': {k: v for k, v in cfg.items()}, 'attrs': trainDict(Train(config=dict(cfg)))}

# ---- tracks / merged profiles / grids ----------------------------------------

grids = []

for trackId, N, crop in [('00_var_speed_limit_100', 100, None), ('00_var_speed_limit_100', 300, None),
                         ('00_var_speed_limit_100', 50, None), ('00_var_speed_limit_100', 100, 8500),
                         ('00_var_speed_limit_100', 300, 8500), ('00_var_speed_limit_100', 300, 3475),
                         ('CH_StGallen_Wil', 200, None), ('CH_StGallen_Wil', 400, None)]:

    track = Track(config={'id': trackId})

    if crop is not None:
        track.updateLimits(positionEnd=crop)

    pts = computeDiscretizationPoints(track, N)

    grids.append({'track': trackId, 'N': N, 'crop': crop, 'length': float(track.length), 'points': frame(pts)})

out['grids'] = grids

track = Track(config={'id': 'CH_StGallen_Wil'})
out['merge_CH'] = frame(track.mergeDataFrames())

# grid error case: N too small for CH
try:
    computeDiscretizationPoints(Track(config={'id': 'CH_StGallen_Wil'}), 100)
    out['grid_CH_100_error'] = None
except ValueError as e:
    out['grid_CH_100_error'] = 'ValueError'

# crop from the middle (MPC-like): positionStart inside a section
track = Track(config={'id': 'CH_StGallen_Wil'})
track.updateLimits(positionStart=5000.0, positionEnd=20000.0)
out['crop_CH_5000_20000'] = {'length': float(track.length), 'merged': frame(track.mergeDataFrames())}

# reversed track
track = Track(config={'id': 'CH_StGallen_Wil'}).reverse()
out['reverse_CH'] = {'title': track.title, 'merged': frame(track.mergeDataFrames())}

# curvature import with clothoid sampling
track = Track(config={'id': '00_var_speed_limit_100'})
track.importCurvatureTuples([[0.0, 1000, 500], [10000.0, 500, "infinity"], [20000.0, "infinity", "infinity"], [30000.0, -400, -400]],
                            clothoidSamplingInterval=1500.0)
out['clothoid_00'] = {'tuples': [[0.0, 1000, 500], [10000.0, 500, "infinity"], [20000.0, "infinity", "infinity"], [30000.0, -400, -400]],
                      'ds': 1500.0, 'curvatures': frame(track.curvatures), 'merged': frame(track.mergeDataFrames())}

# ---- units ---------------------------------------------------------------------

units = ['m', 'km', 'km/h', 't', '%', 'kW', 'MW', 'kN', 'kN/(m/s)', 'kN/(km/h)', 'N/(km/h)', 'kN/(m/s)^2', 'kN/(km/h)^2',
         'N/(km/h)^2', 't/m', 'm/s', 'permil', 'kg', 'W', 'N', 'm/s^2', '-', 'N/(m/s)', 'N/(m/s)^2', 'kg/m']
out['units'] = {u: convertUnit(1.7, u) for u in units}

# ---- measured loss tables (data.py) -----------------------------------------------

a, b = dataLosses()
out['dataLosses'] = {'A': a, 'B': b}

dst = Path(__file__).resolve().parent / 'loader_fixtures.json'

with open(dst, 'w') as fh:
    json.dump(out, fh, indent=0)

print("wrote", dst, dst.stat().st_size, "bytes")
