import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_manifest():
    import json
    import orc
    with open(os.path.join(orc.GOLDEN, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture()
def workdir(tmp_path):
    """Scratch dir pre-populated with the (decompressed) golden inputs a case needs, on demand."""
    import orc

    class W:
        path = str(tmp_path)

        def need(self, name):
            """Materialise tests/golden/<name> (strip .gz) into the scratch dir; returns its path."""
            plain = name[:-3] if name.endswith(".gz") else name
            dst = os.path.join(self.path, plain)
            if not os.path.exists(dst):
                with open(dst, "wb") as f:
                    f.write(orc.read_maybe_gz(os.path.join(orc.GOLDEN, name)))
            return dst

        def file(self, name):
            return os.path.join(self.path, name)

    return W()
