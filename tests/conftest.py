import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    config.addinivalue_line(
        'markers', 'needs_reference: needs /root/reference (build container only)')


def pytest_collection_modifyitems(config, items):
    from oracle import refshim
    skip_ref = pytest.mark.skip(reason='/root/reference not present on this box')
    for item in items:
        if 'needs_reference' in item.keywords and not refshim.available():
            item.add_marker(skip_ref)
