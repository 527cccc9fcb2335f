import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def c_client():
    """tests/c_client/client.c compiled as strict C99 against include/astroz_hip.h and linked to the library."""
    import subprocess
    import __graft_entry__ as g
    g.build()
    out = os.path.join(ROOT, "tests", "c_client", "client")
    src = os.path.join(ROOT, "tests", "c_client", "client.c")
    lib_dir = os.path.join(ROOT, "astroz_amd")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
                    src, "-o", out, "-L", lib_dir, "-lastroz_hip", "-Wl,-rpath," + lib_dir], check=True)
    return out
