import os
import sys

import pytest

# glibc writes its fatal messages ("malloc(): corrupted ...", "free(): invalid pointer", stack-protector and fortify reports) to
# the controlling TERMINAL unless this is set -- under a harness that keeps only stdout / stderr the reason of an abort is lost
# (round 5: one SIGABRT without a word in 26 soak runs).  Read by glibc when the message is printed.
os.environ.setdefault("LIBC_FATAL_STDERR_", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import orb_oracle_py as O
    O.build()
    O.lib()
    return O
