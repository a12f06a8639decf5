import json
import os
import sys

# Two OpenMP runtimes live in the test process (the C oracle's libgomp and PyTorch's): with the default active wait policy their idle
# worker threads spin against each other on many-core hosts (measured on a 256-thread GPU box: 8.5 s -> more than 10 minutes for two
# test files).  Must be set before either runtime starts its first parallel region.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(str(d["meta"]))
    return d


@pytest.fixture(scope="session")
def golden_small():
    return load_golden("dac_small")


@pytest.fixture(scope="session")
def golden_full():
    return load_golden("dac44k_b1")


def dac_cfg_from_meta(meta):
    from neuralcodecs_amd.config import DACConfig
    kw = dict(meta["cfg"])
    for k in ("encoder_rates", "decoder_rates"):
        if k in kw:
            kw[k] = tuple(kw[k])
    return DACConfig(**kw)


def audit_code_mismatches(codes, ref_codes, gap, tol):
    """Every (stage, frame) whose FIRST mismatch appears at that stage must be a near-tie of the
    golden argmin (top-2 distance gap < tol); later stages of the same frame may then differ freely
    (the residual changed).  Returns the number of frames that diverged."""
    codes = np.asarray(codes); ref = np.asarray(ref_codes)
    B, nq, T = ref.shape
    bad = 0
    for b in range(B):
        for t in range(T):
            neq = np.nonzero(codes[b, :, t] != ref[b, :, t])[0]
            if neq.size:
                bad += 1
                i = int(neq[0])
                g = float(gap[i, b * T + t])
                assert g < tol, f"code mismatch at clip {b} stage {i} frame {t} is not a near-tie (gap {g:g})"
    return bad


def snac_cfg_from_meta(meta):
    from neuralcodecs_amd.config import SNACConfig
    kw = dict(meta["cfg"])
    for k in ("encoder_rates", "decoder_rates", "vq_strides"):
        if k in kw:
            kw[k] = tuple(kw[k])
    return SNACConfig(**kw)


def audit_snac_levels(codes, golden, tol):
    """Per level: a code that differs from the golden one must be a near-tie of the golden argmin.  SNAC levels run at
    different rates, so a flip at a coarse level can change finer levels freely: stop auditing a clip at its first flip."""
    flips = 0
    B = codes[0].shape[0]
    for b in range(B):
        for i, c in enumerate(codes):
            ref = golden[f"codes{i}"][b]
            neq = np.nonzero(np.asarray(c[b]) != ref)[0]
            if neq.size:
                T = ref.shape[0]
                g = golden[f"gap{i}"][b * T + int(neq[0])]
                assert g < tol, f"SNAC code mismatch clip {b} level {i} frame {int(neq[0])} is not a near-tie (gap {g:g})"
                flips += 1
                break
    return flips


def encodec_cfg_from_meta(meta):
    from neuralcodecs_amd.config import EncodecConfig
    kw = dict(meta["cfg"])
    for k in ("ratios", "target_bandwidths"):
        if k in kw:
            kw[k] = tuple(kw[k])
    return EncodecConfig(**kw)
