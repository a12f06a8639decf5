"""CPU suite: the hot matrix-core loops of the BUILT library against the committed ISA audit (profiles/r06_isa_audit.json).

DESIGN 8 (round 4): every vector instruction a wavefront issues between its matrix-core instructions costs matrix-pipe time, and a spilled
scalar register comes back through `v_readlane` -- a vector instruction the source never asked for.  One extra int32 in ConvArgs once
shifted the argument block, put 51 of them into the k = 7 loop and cost 0.7 ms of the DAC step for several commits, unnoticed behind the
box-to-box spread (VERDICT r4 "hygiene").  This test disassembles the gfx950 code objects of neuralcodecs_amd/csrc/build/*.o (seconds, no
compilation: tools/isa_audit.py audit_object) and fails when an audited loop's in-loop `v_readlane` count RISES above the record, or its
vector-instruction count grows by more than 3 %.  A count may fall: regenerate the record with `python tools/isa_audit.py --table
profiles/r06_isa_audit.json` and commit it with the change that earned it."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
REC = os.path.join(ROOT, "profiles", "r06_isa_audit.json")
BUILD = os.path.join(ROOT, "neuralcodecs_amd", "csrc", "build")


@pytest.fixture(scope="module")
def now():
    import isa_audit
    if not os.path.isdir(BUILD) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no build directory / LLVM tools: run __graft_entry__.build() first")
    rec_cc = json.load(open(REC)).get("compiler")
    if rec_cc and rec_cc != isa_audit.compiler_version():
        pytest.skip(f"the record was made under {rec_cc!r}, this is {isa_audit.compiler_version()!r}: instruction counts are not comparable (re-record)")
    tab = isa_audit.hot_table(BUILD)
    if not tab:
        pytest.skip("the build directory holds none of the audited objects")
    return tab


def test_every_recorded_kernel_is_still_built(now):
    rec = json.load(open(REC))["kernels"]
    assert len(rec) >= 10
    missing = [k for k in rec if k not in now]
    assert not missing, f"audited instances no longer in the build (renamed template parameters? re-record): {missing}"


def test_in_loop_v_readlane_and_vector_counts_do_not_rise(now):
    rec = json.load(open(REC))["kernels"]
    worse = []
    for k, r in rec.items():
        n = now.get(k)
        if n is None:
            continue
        if n["max_in_loop_v_readlane"] > r["max_in_loop_v_readlane"]:
            worse.append((r["what"], "v_readlane", r["max_in_loop_v_readlane"], n["max_in_loop_v_readlane"]))
        rv, nv = max(lp["valu"] for lp in r["loops"]), max(lp["valu"] for lp in n["loops"])
        if nv > 1.03 * rv + 8:
            worse.append((r["what"], "valu", rv, nv))
    assert not worse, "hot loops got heavier (what, counter, recorded, now): %r" % (worse,)


def test_scratch_and_vector_register_spills_do_not_rise(now):
    """scratch was invisible to the guard (the 128-row XV-only instances spilled 88-105 vector registers / 324 B of scratch per lane
    unnoticed).  The record now carries private_segment_fixed_size and vgpr_spill_count of every audited instance; neither may rise."""
    rec = json.load(open(REC))["kernels"]
    worse = []
    for k, r in rec.items():
        n = now.get(k)
        if n is None or r.get("scratch_bytes") is None:
            continue
        for key in ("scratch_bytes", "vgpr_spill"):
            if (n.get(key) or 0) > (r.get(key) or 0):
                worse.append((r["what"], key, r.get(key), n.get(key)))
    assert not worse, "audited instances spill more than the record (what, counter, recorded, now): %r" % (worse,)
    assert any(r.get("scratch_bytes") is not None for r in rec.values()), "the record carries no scratch figures: re-record with tools/isa_audit.py --table"
