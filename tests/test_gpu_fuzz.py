"""Seeded randomised differential test: HIP path vs oracle over random k / scaled / moltype / length distributions /
alphabets (tools/fuzz_parity.py holds the generator; `python tools/fuzz_parity.py --cases 1000` runs it at length)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def _load():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [3, 11])
def test_random_cases_match_oracle(seed):
    assert _load().run(60, seed) == 0
