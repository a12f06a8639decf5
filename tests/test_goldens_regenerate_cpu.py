"""CPU suite: the committed fixtures under tests/golden/ ARE what tools/make_golden.py generates today from the PyTorch-CPU restatement
(oracle/torch_ref) -- the reduced-width cases of all three codecs are re-derived into a temporary directory and compared array by
array (integer codes exactly, floating-point arrays to 2e-6 absolute: ATen's reductions may differ in the last bit between hosts).
The C oracle is held to these same files (tests/test_oracle_*_cpu.py), so this closes the loop restatement -> fixtures -> oracle."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def regenerated(tmp_path_factory):
    import make_golden as mg
    out = str(tmp_path_factory.mktemp("golden"))
    mg.OUT = out
    mg.dac_case("dac_small", mg.SMALL, 2, 2000, 7, 11, False)
    mg.snac_case("snac_small", mg.SNAC_SMALL, 2, 3001, 5, 3, 99, False)
    mg.snac_case("snac_small_attn", mg.SNAC_SMALL_ATTN, 2, 2500, 6, 4, 98, False)
    mg.snac_case("snac_small_tensor", mg.SNAC_SMALL, 2, 3100, 5, 3, 99, False, tensor_overload=True)
    mg.encodec_case("encodec_small48", mg.ENC_SMALL48, 2, 8100, 7, 3, False)
    mg.encodec_case("encodec_small24", mg.ENC_SMALL24, 2, 3001, 8, 4, False)
    return out


@pytest.mark.parametrize("name", ["dac_small", "snac_small", "snac_small_attn", "snac_small_tensor", "encodec_small48", "encodec_small24"])
def test_committed_fixture_is_what_the_generator_writes(regenerated, name):
    new = np.load(os.path.join(regenerated, name + ".npz"))
    old = np.load(os.path.join(GOLD, name + ".npz"))
    assert sorted(new.files) == sorted(old.files), f"{name}: the generator writes {sorted(new.files)}, the fixture holds {sorted(old.files)}"
    assert json.loads(str(new["meta"])) == json.loads(str(old["meta"]))
    for k in old.files:
        if k == "meta":
            continue
        a, b = new[k], old[k]
        assert a.shape == b.shape and a.dtype == b.dtype, (name, k)
        if k.startswith("codes"):
            assert np.array_equal(a, b), f"{name}: {k} differs from the committed fixture ({int((a != b).sum())} codes)"
        elif k.startswith("gap"):
            assert np.allclose(a, b, atol=1e-4), (name, k)                  # distance gaps: differences of O(10) numbers
        else:
            assert np.allclose(a, b, atol=2e-6, rtol=0), f"{name}: {k} max-abs diff {np.abs(a - b).max()}"
