import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

for p in (ROOT / 'ms-eetc_amd', ROOT):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
