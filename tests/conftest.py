import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def synth_model():
    from smplifyx_amd import synthetic
    return synthetic.make_synthetic_model(0)


def pytest_terminal_summary(terminalreporter):
    """Observed maxima of every closure-vs-oracle comparison of the session (tests/helpers.check_closure)."""
    try:
        import helpers as H
    except Exception:
        return
    lines = H.parity_log_lines()
    if lines:
        terminalreporter.write_sep("-", "closure parity: observed maxima")
        for l in lines:
            terminalreporter.write_line(l)
