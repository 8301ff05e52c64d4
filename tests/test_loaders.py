"""
Host-side loaders against fixtures captured from the reference loaders
(tests/golden/make_loader_fixtures.py) and against the reference's own stored data
(gpops/CH_StGallen_Wil.csv == merged CH profile; SURVEY.md section 8c pin 1).
"""

import copy
import json
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

from mseetc.track import Track, computeDiscretizationPoints
from mseetc.train import Train
from mseetc.utils import convertUnit, checkTTOBenchVersion, Options

GOLD = Path(__file__).resolve().parent / 'golden'

with open(GOLD / 'loader_fixtures.json') as fh:
    FIX = json.load(fh)


def assertFrame(df, ref, exact=True):
    assert list(df.columns) == list(ref['columns'].keys())
    cmp = np.array_equal if exact else np.allclose
    assert cmp(df.index.values, np.array(ref['index']))
    for c in df.columns:
        assert cmp(df[c].values, np.array(ref['columns'][c])), c


def test_train_default_attributes():
    t = Train(config={'id': 'NL_Intercity_VIRM6'})
    for k, v in FIX['train_default'].items():
        assert getattr(t, k, None) == v, k


def test_train_override_and_config_not_mutated():
    cfg = copy.deepcopy(FIX['train_override']['config'])
    t = Train(config=cfg)
    assert 'id' in cfg
    for k, v in FIX['train_override']['attrs'].items():
        assert getattr(t, k, None) == v, k


def test_train_errors():
    with pytest.raises(ValueError):
        Train(config=[])
    with pytest.raises(ValueError):
        Train(config={})
    with pytest.raises(ValueError):
        Train(config={'id': 'NL_Intercity_VIRM6', 'no such field': {'unit': 'kg', 'value': 1}})
    with pytest.raises(ValueError):
        Train(config={'id': 'NL_Intercity_VIRM6', 'mass': 3})
    t = Train(config={'id': 'NL_Intercity_VIRM6'})
    t.forceMin = 0
    t.forceMinPn = 0
    with pytest.raises(ValueError):
        t.checkFields()


@pytest.mark.parametrize('k', range(len(FIX['grids'])))
def test_grid_matches_reference(k):
    g = FIX['grids'][k]
    track = Track(config={'id': g['track']})
    if g['crop'] is not None:
        track.updateLimits(positionEnd=g['crop'])
    assert track.length == g['length']
    pts = computeDiscretizationPoints(track, g['N'])
    assert len(pts) == g['N'] + 1
    assertFrame(pts, g['points'])


def test_grid_too_coarse_is_an_error():
    assert FIX['grid_CH_100_error'] == 'ValueError'
    with pytest.raises(ValueError):
        computeDiscretizationPoints(Track(config={'id': 'CH_StGallen_Wil'}), 100)


def test_merged_profile_CH_equals_reference_and_gpops_csv():
    track = Track(config={'id': 'CH_StGallen_Wil'})
    merged = track.mergeDataFrames()
    assertFrame(merged, FIX['merge_CH'])
    csv = pd.read_csv(GOLD / 'CH_StGallen_Wil.csv')
    assert len(csv) == len(merged) == 165
    assert np.allclose(csv['position [m]'].values, merged.index.values)
    assert np.allclose(csv['gradient [permil]'].values, merged['Gradient [permil]'].values)
    assert np.allclose(csv['speed limit [m/s]'].values, merged['Speed limit [m/s]'].values)
    assert np.all(merged['Curvature [1/m]'].values == 0)


def test_crop_from_the_middle():
    track = Track(config={'id': 'CH_StGallen_Wil'})
    track.updateLimits(positionStart=5000.0, positionEnd=20000.0)
    assert track.length == FIX['crop_CH_5000_20000']['length']
    assertFrame(track.mergeDataFrames(), FIX['crop_CH_5000_20000']['merged'])
    with pytest.raises(ValueError):
        track.updateLimits(positionStart=-1)
    with pytest.raises(ValueError):
        track.updateLimits(positionEnd=1e9)


def test_reverse():
    track = Track(config={'id': 'CH_StGallen_Wil'}).reverse()
    assert track.title == FIX['reverse_CH']['title']
    assertFrame(track.mergeDataFrames(), FIX['reverse_CH']['merged'], exact=False)


def test_clothoid_fixture():
    fx = FIX['clothoid_00']
    track = Track(config={'id': '00_var_speed_limit_100'})
    track.importCurvatureTuples(fx['tuples'], clothoidSamplingInterval=fx['ds'])
    assertFrame(track.curvatures, fx['curvatures'])
    assertFrame(track.mergeDataFrames(), fx['merged'])


def test_clothoid_known_answers():
    # restates the solver-free reference unit test (unitTests/curvatureResistance/curvatureResistance.py:204-286)
    track = Track(config={'id': '00_var_speed_limit_100'})
    r0, rf = 1000, 500
    k0, kf = 1/r0, 1/rf
    col = 'Curvature [1/m]'

    track.importCurvatureTuples(tuples=[[0.0, r0, rf]])
    assert track.curvatures[col].to_dict() == {0.0: (k0 + kf)/2}

    track.importCurvatureTuples(tuples=[[0.0, r0, rf]], clothoidSamplingInterval=track.length + 1)
    assert track.curvatures[col].to_dict() == {0.0: (k0 + kf)/2}

    ds = track.length/4
    track.importCurvatureTuples(tuples=[[0.0, r0, rf]], clothoidSamplingInterval=ds)
    alpha = track.length/(kf - k0)
    k1 = (k0 + (k0 + ds*1/alpha))/2
    k2 = ((k0 + ds*1/alpha) + (k0 + ds*2/alpha))/2
    k3 = ((k0 + ds*2/alpha) + (k0 + ds*3/alpha))/2
    k4 = ((k0 + ds*3/alpha) + kf)/2
    assert track.curvatures[col].to_dict() == {0.0: k1, ds: k2, 2*ds: k3, 3*ds: k4}

    ds = track.length/4 + 1
    track.importCurvatureTuples(tuples=[[0.0, r0, rf]], clothoidSamplingInterval=ds)
    k1 = (k0 + (k0 + ds*1/alpha))/2
    k2 = ((k0 + ds*1/alpha) + (k0 + ds*2/alpha))/2
    k3 = ((k0 + ds*2/alpha) + kf)/2
    assert track.curvatures[col].to_dict() == {0.0: k1, ds: k2, 2*ds: k3}

    track.importCurvatureTuples(tuples=[[0.0, r0, "infinity"]])
    assert track.curvatures[col].to_dict() == {0.0: k0/2}

    with pytest.raises(ValueError):
        track.importCurvatureTuples(tuples=[[0.0, r0, rf]], clothoidSamplingInterval=-1)
    with pytest.raises(ValueError):
        track.importCurvatureTuples(tuples=[[0.0, 0.0, rf]])
    with pytest.raises(ValueError):
        track.importCurvatureTuples(tuples=[[500, r0, rf], [500, rf, 1 + rf]])
    with pytest.raises(ValueError):
        track.importCurvatureTuples(tuples=[[-1, r0, rf]])


def test_curvature_threshold_rejected():
    track = Track(config={'id': '00_var_speed_limit_100'})
    track.importCurvatureTuples(tuples=[[0.0, 100, 100]])
    with pytest.raises(ValueError):
        track.checkFields()


def test_units():
    for u, v in FIX['units'].items():
        assert convertUnit(1.7, u) == v, u
    with pytest.raises(ValueError):
        convertUnit(1.0, 'furlong')


def test_version_check():
    checkTTOBenchVersion({'metadata': {'library version': 'TTOBench v1.3'}}, ['1.3'])
    with pytest.raises(ValueError):
        checkTTOBenchVersion({'metadata': {'library version': 'TTOBench v9.9'}}, ['1.3'])
    with pytest.raises(ValueError):
        checkTTOBenchVersion({}, ['1.3'])
    with pytest.raises(TypeError):
        checkTTOBenchVersion({}, '1.3')


def test_options_base():
    class O(Options):
        def __init__(self, d):
            self.a = 1
            super().__init__(d)
    assert O({'a': 5}).toDict() == {'a': 5}
    with pytest.raises(ValueError):
        O({'b': 1})
