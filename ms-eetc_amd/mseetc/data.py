"""
Measured traction-chain loss tables (reference: mseetc/data.py:1-27).  The numbers are data and live in
data/motor_losses_VIRM6.json; `dataLosses()` returns them in the reference's structure.
"""

import json
from pathlib import Path

_FILE = Path(__file__).resolve().parent.parent / 'data' / 'motor_losses_VIRM6.json'


def dataLosses():
    "(configA, configB): dicts with 'loads' [%], 'frequencies' [Hz] and 'losses' [W] (rows = loads)."

    with open(_FILE) as fh:
        raw = json.load(fh)

    return raw['A'], raw['B']
