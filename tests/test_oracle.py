"""CPU tests of the oracle: it must reproduce the golden fixtures (generated from the compiled
reference), the known-answer values of the serial reference, and — when the reference build is
present — the reference itself, bit for bit."""
import json
import os
import zlib

import numpy as np
import pytest

import cases
import oracle_binding as ob

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "golden.json")))
SAMPLES = np.load(os.path.join(HERE, "golden", "golden_samples.npz"))


def check_against_golden(name, u, exact=True, l2_tol=None):
    g = GOLDEN[name]
    assert list(u.shape) == g["shape"]
    idx = SAMPLES[name + "_idx"]
    bits = SAMPLES[name + "_bits"]
    if exact:
        assert np.array_equal(u.reshape(-1)[idx].view(np.uint32), bits), f"{name}: sampled words differ"
        d = cases.digest(u)
        assert d["crc32"] == g["crc32"], f"{name}: level CRC32 differs"
        assert d["sum"] == g["sum"] and d["sumsq"] == g["sumsq"] and d["maxabs"] == g["maxabs"]
    else:
        ref = bits.view(np.float32).astype(np.float64)
        got = u.reshape(-1)[idx].astype(np.float64)
        l2 = np.sqrt(((got - ref) ** 2).sum() / ((ref ** 2).sum() + 1e-30))
        assert l2 < l2_tol, f"{name}: sampled L2 {l2} >= {l2_tol}"
        assert abs(cases.digest(u)["sumsq"] - g["sumsq"]) <= 1e-6 * g["sumsq"]


def inputs_crc(c):
    crc = 0
    for k in ("u", "m", "src", "coords"):
        if c[k] is not None:
            crc = zlib.crc32(np.ascontiguousarray(c[k]).tobytes(), crc)
    return format(crc, "08x")


@pytest.mark.parametrize("name", list(cases.CASES))
def test_oracle_matches_golden(pkg, name):
    c = cases.build(name, pkg)
    assert inputs_crc(c) == GOLDEN[name]["inputs_crc32"], "input generator drifted from the fixture"
    prof = ob.run_oracle(c["m"], c["src"], c["coords"], c["u"], **c["kw"])
    check_against_golden(name, c["u"])
    assert (prof.section0 > 0) == GOLDEN[name]["timers_nonzero"]


def test_known_answers_of_serial_reference(pkg):
    """32^3, 50 steps: sqrt(sum u^2) and max|u| of the serial reference for both input sets."""
    u, m, src, coords = pkg.standard_inputs(32, 32, 32, 50, 1)
    ob.run_oracle(m, src, coords, u)
    assert np.sqrt((u.astype(np.float64) ** 2).sum()) == pytest.approx(2.168773008e-01, rel=1e-9)
    assert float(np.abs(u).max()) == pytest.approx(1.168384925e-01, rel=1e-8)
    assert not np.isnan(u).any()
    uc, mc = pkg.fill_correctness_inputs(32)
    ob.run_oracle(mc, None, None, uc, time_M=49, h=(1.0, 1.0, 1.0))
    assert np.sqrt((uc.astype(np.float64) ** 2).sum()) == pytest.approx(1.573436133e+06, rel=1e-9)
    assert float(np.abs(uc).max()) == pytest.approx(5.609833984e+03, rel=1e-9)


def test_reference_sanity_value_max_field(pkg):
    """The driver prints 'Max field value: 0.1' after every Ricker run (main.cpp:481 with
    setprecision(1)); 64^3 gives the same maximum as 32^3."""
    u, m, src, coords = pkg.standard_inputs(64, 64, 64, 50, 1)
    ob.run_oracle(m, src, coords, u)
    assert f"{float(np.abs(u).max()):.1f}" == "0.1"


@pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref not built (needs the reference tree)")
@pytest.mark.parametrize("name", ["ricker_n32", "correctness_n32", "noncubic_aniso", "odd_sizes",
                                  "subrange_unaligned", "time_offset", "ricker30_n24_overlap"])
def test_oracle_bit_identical_to_compiled_reference(pkg, name):
    c = cases.build(name, pkg)
    u_ref = c["u"].copy()
    ob.run_oracle(c["m"], c["src"], c["coords"], c["u"], **c["kw"])
    ob.run_ref(c["m"], c["src"], c["coords"], u_ref, **c["kw"])
    assert ob.bits_equal(c["u"], u_ref)


def test_halos_are_inert_without_sources(pkg):
    """section0 writes interior cells only (openacc.cpp:152-156): halo cells of every level,
    including stale data in the level being written, come back unchanged."""
    c = cases.build("no_sources_random", pkg)
    before = c["u"].copy()
    ob.run_oracle(c["m"], None, None, c["u"], **c["kw"])
    interior = np.zeros(before.shape[1:], dtype=bool)
    interior[4:-4, 4:-4, 4:-4] = True
    for t in range(3):
        assert np.array_equal(before[t][~interior], c["u"][t][~interior])
    assert not np.array_equal(before, c["u"])


def test_thread_count_does_not_change_bits(pkg):
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import cases, oracle_binding as ob\n"
        "from __graft_entry__ import load_package\n"
        "pkg = load_package(); c = cases.build('noncubic_aniso', pkg)\n"
        "ob.run_oracle(c['m'], c['src'], c['coords'], c['u'], **c['kw'])\n"
        "print(cases.digest(c['u'])['crc32'])\n" % (os.path.dirname(HERE), HERE))
    outs = []
    for nthreads in ("1", "5"):
        env = dict(os.environ, OMP_NUM_THREADS=nthreads)
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, check=True,
                                   capture_output=True, text=True).stdout.strip())
    assert outs[0] == outs[1] == str(GOLDEN["noncubic_aniso"]["crc32"])
