"""
Track description and shooting-grid construction for the MI355X train-control
solver.

Same public surface as the reference's `mseetc/track.py` (`Track` :110-450,
`computeDiscretizationPoints` :91-107): speed limits, gradients and curvatures are
piecewise-constant profiles kept as pandas frames indexed by 'Position [m]' (that is
what callers and `postProcessDataFrame` read), but all arithmetic on them is numpy:
profiles are (positions, values) pairs, merged with a sorted union + step lookup
instead of outer joins.  Plotting / printing / altitude helpers are out of scope.
"""

import json
import sys
from pathlib import Path

import numpy as np
import pandas as pd

from .utils import checkTTOBenchVersion, convertUnit

_DATA_DIR = Path(__file__).resolve().parent.parent / 'data'

_POS = 'Position [m]'
_GRAD = 'Gradient [permil]'
_VLIM = 'Speed limit [m/s]'
_CURV = 'Curvature [1/m]'


def _frame(positions, values, label):
    "Piecewise-constant profile as the single-column frame callers expect."

    idx = pd.Index(np.asarray(positions, dtype=float), name=_POS)

    return pd.DataFrame({label: np.asarray(values, dtype=float)}, index=idx)


def _stepLookup(positions, values, query):
    "Value of the piecewise-constant profile at each query position (NaN before the first section)."

    k = np.searchsorted(positions, query, side='right') - 1
    out = np.where(k >= 0, np.asarray(values, dtype=float)[np.maximum(k, 0)], np.nan)

    return out


def importTuples(tuples, xLabel, yLabels):
    "Validated list of (position, value...) tuples -> frame (reference: track.py:11-53)."

    if not isinstance(yLabels, list):
        yLabels = [yLabels]

    if not isinstance(tuples, list):
        raise ValueError("Input must be a list (of tuples or lists)!")

    for tup in tuples:

        if not isinstance(tup, (tuple, list)) or len(tup) != 1 + len(yLabels):
            raise ValueError("Error in list!")

    pos = np.array([tup[0] for tup in tuples])

    if np.any(pos < 0):
        raise ValueError("Position data cannot be negative!")

    if np.any(np.isinf(pos)):
        raise ValueError("Position data cannot be infinite!")

    if np.any(np.diff(pos) <= 0):
        raise ValueError("Position data must monotonically increase!")

    df = pd.DataFrame(index=pd.Index(pos, name=xLabel))

    for k, label in enumerate(yLabels):
        df[label] = [float(tup[1 + k]) for tup in tuples]

    return df


def checkDataFrame(df, trackLength):
    "First section must start at 0 and the last one before the end of the track (track.py:56-70)."

    if df.index[0] != 0:
        raise ValueError("Error in '{}': First track section must start at 0 m (beginning of track)!".format(df.columns[0]))

    if df.index[-1] > trackLength:
        raise ValueError("Error in '{}': Last track section must start before {} m (end of track)!".format(df.columns[0], trackLength))

    return True


def computeDiscretizationPoints(track, numIntervals):
    """
    Shooting grid (reference: track.py:91-107): the breakpoints of the merged
    profile plus a uniform fill-in so that there are exactly N+1 nodes; a fill-in
    node that coincides with a breakpoint leaves the grid short and is an error.
    """

    profile = track.mergeDataFrames()

    fill = np.linspace(0, track.length, numIntervals + 1 - (len(profile) - 1))

    nodes = np.union1d(fill, profile.index.values)

    if len(nodes) != numIntervals + 1:
        raise ValueError("Wrong number of computed discretization intervals!")

    cols = {label: _stepLookup(profile.index.values, profile[label].values, nodes) for label in profile.columns}

    return pd.DataFrame(cols, index=pd.Index(nodes, name='position [m]'))


class Track():

    CURVATURE_THRESHOLD = 1/150  # absolute value of maximum allowed curvature [1/m]

    def __init__(self, config, pathJSON=_DATA_DIR / 'tracks'):

        if not isinstance(config, dict):
            raise ValueError("Track configuration should be provided as a dictionary!")

        if 'id' not in config:
            raise ValueError("Track ID must be specified in configuration!")

        with open(Path(pathJSON) / (config['id'] + '.json')) as file:
            data = json.load(file)

        checkTTOBenchVersion(data, ['1.1', '1.2', '1.3'])

        stops = data['stops']

        self.length = convertUnit(stops['values'][-1], stops['unit'])
        self.altitude = convertUnit(data['altitude']['value'], data['altitude']['unit']) if 'altitude' in data else 0
        self.title = data['metadata']['id']

        self.importSpeedLimitTuples(data['speed limits']['values'], data['speed limits']['units']['velocity'])

        if 'gradients' in data:
            self.importGradientTuples(data['gradients']['values'], data['gradients']['units']['slope'])
        else:
            self.importGradientTuples([(0.0, 0.0)], 'permil')

        if 'curvatures' in data:
            units = data['curvatures']['units']
            self.importCurvatureTuples(data['curvatures']['values'], units['radius at start'], units['radius at end'],
                                       config.get('clothoidSamplingInterval'))
        else:
            self.importCurvatureTuples([(0.0, "infinity", "infinity")], "m", "m", config.get('clothoidSamplingInterval'))

        numStops = len(stops['values'])
        first = config.get('from', 0)
        last = config.get('to', numStops - 1)

        if not 0 <= first < numStops - 1:
            raise ValueError("Index of departure is out of bounds!")

        if not first < last < numStops:
            raise ValueError("Index of destination is out of bounds!")

        self.updateLimits(convertUnit(stops['values'][first], stops['unit']), convertUnit(stops['values'][last], stops['unit']))

        self.checkFields()

    # ---- validation -------------------------------------------------------

    def lengthOk(self):

        L = self.length

        return L is not None and 0 < L < np.inf

    def _profileOk(self, df):

        return bool(len(df) > 0 and checkDataFrame(df, self.length))

    def gradientsOk(self):

        return self._profileOk(self.gradients)

    def speedLimitsOk(self):

        return self._profileOk(self.speedLimits)

    def curvaturesOk(self):

        tooTight = np.abs(self.curvatures[_CURV].values) > Track.CURVATURE_THRESHOLD

        return (not tooTight.any()) and self._profileOk(self.curvatures)

    def checkFields(self):

        if not self.lengthOk():
            raise ValueError("Track length must be a strictly positive number, not {}!".format(self.length))

        if self.altitude is None or np.isinf(self.altitude):
            raise ValueError("Altitude must be a number, not {}!".format(self.altitude))

        for ok, what in ((self.gradientsOk, 'gradients'), (self.speedLimitsOk, 'speed limits'), (self.curvaturesOk, 'curvatures')):
            if not ok():
                raise ValueError("Issue with track {}!".format(what))

    # ---- importers --------------------------------------------------------

    def _requireLength(self, what):

        if not self.lengthOk():
            raise ValueError("Cannot import {} without a valid track length!".format(what))

    def importGradientTuples(self, tuples, unit='permil'):

        self._requireLength('gradients')

        if unit != 'permil':
            raise ValueError("Specified gradient unit not supported!")

        frame = importTuples(tuples, _POS, _GRAD)
        checkDataFrame(frame, self.length)
        self.gradients = frame

    def importSpeedLimitTuples(self, tuples, unit='km/h'):

        self._requireLength('speed limits')

        if unit not in ('km/h', 'm/s'):
            raise ValueError("Specified speed unit not supported!")

        frame = importTuples([(pos, convertUnit(lim, unit)) for pos, lim in tuples], _POS, _VLIM)
        checkDataFrame(frame, self.length)
        self.speedLimits = frame

    def importCurvatureTuples(self, tuples, unitRadiusStart='m', unitRadiusEnd='m', clothoidSamplingInterval=None):

        self._requireLength('curvature')

        if not {unitRadiusStart, unitRadiusEnd} <= {'m', 'km'}:
            raise ValueError("Specified curvature radius unit not supported!")

        # float("infinity") is inf, i.e. a straight section
        sections = [(pos, convertUnit(float(ra), unitRadiusStart), convertUnit(float(rb), unitRadiusEnd)) for pos, ra, rb in tuples]

        frame = importTuples(self.sampleClothoid(sections, clothoidSamplingInterval), _POS, [_CURV])
        checkDataFrame(frame, self.length)
        self.curvatures = frame

    def sampleClothoid(self, tuples, ds=None):
        """
        Piecewise-constant curvature from (position, radius at start, radius at end) sections (reference: track.py:270-348).
        Equal end radii: the section keeps its curvature.  Otherwise the section is a clothoid (curvature linear in
        position); with a sampling interval ds it is cut into floor(length/ds) pieces, each carrying the mean of the
        curvatures at its two ends, the last piece reaching to the end of the section; without ds (or ds longer than the
        section) the whole section carries the mean of its end curvatures.
        """

        starts = [sec[0] for sec in tuples]

        if any(r == 0 for sec in tuples for r in sec[1:3]):
            raise ValueError("Curvature radius cannot be 0!")

        if min(starts, default=0) < 0:
            raise ValueError("Positions cannot be negative!")

        if any(a == b for a, b in zip(starts, starts[1:])):
            raise ValueError("Positions must be monotonically increasing")

        if ds is not None and ds <= 0:
            raise ValueError("Discretization step must be greater than zero or None!")

        ends = starts[1:] + [self.length]
        out = []

        for (start, ra, rb), end in zip(tuples, ends):

            k0, k1 = 1/ra, 1/rb

            if abs(k0 - k1) <= sys.float_info.epsilon:
                out.append((start, k0))
                continue

            pieces = int((end - start)/ds) if ds is not None else 0

            if pieces == 0:
                out.append((start, (k0 + k1)/2))
                continue

            alpha = (end - start)/(k1 - k0)      # K(s) = k0 + (s - start)/alpha

            for j in range(pieces):
                here = k0 + j*ds/alpha
                mean = (here + k1)/2 if j == pieces - 1 else here + ds/(2*alpha)
                out.append((start + j*ds, mean))

        return out

    # ---- transformations --------------------------------------------------

    def reverse(self):
        "Switch to the opposite direction of travel (track.py:351-374)."

        try:
            self.checkFields()
        except ValueError as e:
            raise ValueError("Track cannot be reversed due to error: {}".format(str(e)))

        def flipped(df, sign):

            label = df.columns[0]
            ends = np.append(df.index.values[1:], self.length)

            return _frame(np.flip(self.length - ends), sign*np.flip(df[label].values), label)

        self.gradients = flipped(self.gradients, -1)
        self.speedLimits = flipped(self.speedLimits, +1)
        self.curvatures = flipped(self.curvatures, -1)

        self.title = self.title + ' (reversed)'

        return self

    def mergeDataFrames(self):
        "Sections of constant curvature, gradient and speed limit (track.py:377-383); columns in that order."

        parts = ((self.curvatures, _CURV), (self.gradients, _GRAD), (self.speedLimits, _VLIM))

        nodes = parts[0][0].index.values

        for df, _ in parts[1:]:
            nodes = np.union1d(nodes, df.index.values)

        cols = {label: _stepLookup(df.index.values, df[label].values, nodes) for df, label in parts}

        return pd.DataFrame(cols, index=pd.Index(nodes, name=_POS))

    def updateLimits(self, positionStart=None, positionEnd=None, unit='m'):
        "Truncate the track to [positionStart, positionEnd] and re-base positions (track.py:420-450)."

        positionStart = 0 if positionStart is None else positionStart
        positionEnd = self.length if positionEnd is None else positionEnd

        if (not 0 <= positionStart < self.length) or (not 0 < positionEnd <= self.length):
            raise ValueError("Given positions must be between limits of track!")

        positionStart = convertUnit(positionStart, unit)
        positionEnd = convertUnit(positionEnd, unit)

        def cropped(df):

            label = df.columns[0]
            pos = df.index.values
            nodes = np.union1d(pos, [positionStart])
            vals = _stepLookup(pos, df[label].values, nodes)
            keep = (nodes >= positionStart) & (nodes <= positionEnd)
            nodes, vals = nodes[keep], vals[keep]

            return _frame(nodes - nodes[0], vals, label)

        self.length -= positionStart + (self.length - positionEnd)

        self.speedLimits = cropped(self.speedLimits)
        self.gradients = cropped(self.gradients)
        self.curvatures = cropped(self.curvatures)
