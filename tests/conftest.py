import os
import sys

import pytest

# glibc writes its fatal messages ("malloc(): corrupted ...", "free(): invalid pointer", stack-protector and fortify reports) to
# the controlling TERMINAL unless this is set -- under a harness that keeps only stdout / stderr the reason of an abort is lost
# (round 5: one SIGABRT without a word in 26 soak runs).  Read by glibc when the message is printed.
os.environ.setdefault("LIBC_FATAL_STDERR_", "1")
# ... and pytest.ini runs the suite with --capture=sys for the same reason: the HIP runtime's "Memory access fault by GPU ..." line
# (round 6 caught one inside a runtime copy that way), RCCL's warnings and libstdc++'s "terminate called ..." are
# written to fd 2 by C code, and a process that aborts mid-test never gets to replay a captured fd.
import faulthandler  # noqa: E402

if not faulthandler.is_enabled():  # (pytest's own plugin enables it on the real stderr before this file is imported)
    try:
        faulthandler.enable(file=sys.__stderr__, all_threads=True)
    except Exception:  # noqa: BLE001 (a stderr without a file descriptor)
        pass

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
