import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "simple-zk-rollups_amd", "python"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the C oracle is test infrastructure: build it on demand (gcc only)
    if not os.path.exists(os.path.join(ROOT, "oracle", "libzkr_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def small_case():
    """m = 2^7 synthetic rollup-shaped circuit with oracle-generated key (shared by several tests)."""
    import groth16 as g
    circ = g.synth_circuit(128, 7, 0x5A4B0001)
    tox = g.toxic_from_seed(0x5A4B00FF)
    pk, vk = g.setup(circ, tox)
    pkb = g.binarify_proving_key(g.to_json_key(pk))
    rng = g.SplitMix64(77)
    return dict(circ=circ, tox=tox, pk=pk, vk=vk, pkb=pkb, w=circ["witness"], wb=g.binarify_witness(circ["witness"]),
                r=rng.fr(), s=rng.fr())
