import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
BCL2 = "bcl2_first25_uniprotkb_accession_O43236_OR_accession_2025_02_06.fasta.gz"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    path = os.path.join(GOLDEN, name)
    if name.endswith(".gz"):
        with gzip.open(path, "rb") as f:
            return json.loads(f.read())
    with open(path) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_sketches():
    return load_golden("sketches.json.gz")


@pytest.fixture(scope="session")
def golden_kmer_tables():
    return load_golden("kmer_tables.json.gz")


@pytest.fixture(scope="session")
def hash_kats():
    return load_golden("hash_kats.json")


@pytest.fixture(scope="session")
def index_kats():
    return load_golden("index_kats.json")


@pytest.fixture(scope="session")
def search_expected():
    return load_golden("search_expected.json")


@pytest.fixture(scope="session")
def bcl2_records():
    from oracle import oracle
    return oracle.read_fasta(os.path.join(GOLDEN, BCL2))


@pytest.fixture(scope="session")
def ced9_records():
    from oracle import oracle
    return oracle.read_fasta(os.path.join(GOLDEN, "ced9.fasta"))
