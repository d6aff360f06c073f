"""Generate tests/golden/golden.json + golden_samples.npz from the COMPILED REFERENCE.

Run in the build container, where /root/reference exists:
    make -C oracle ref && python tests/golden/make_golden.py
It executes oracle/_ref/libref_openacc.so — the reference's openacc.cpp compiled with
g++ -O2 from where it lies — on the inputs of tests/cases.py and records, per case, the
CRC32 of every output level, fp64 moments, a CRC32 of the inputs (to detect generator
drift) and 4096 sampled output words.  Only data is written; no reference text is stored.
"""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402
import oracle_binding as ob  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


def main():
    assert ob.have_ref(), "build the reference first: make -C oracle ref"
    pkg = load_package()
    golden, samples = {}, {}
    for name in cases.CASES:
        c = cases.build(name, pkg)
        inputs_crc = 0
        for k in ("u", "m", "src", "coords"):
            if c[k] is not None:
                inputs_crc = zlib.crc32(np.ascontiguousarray(c[k]).tobytes(), inputs_crc)
        u = c["u"]
        prof = ob.run_ref(c["m"], c["src"], c["coords"], u, **c["kw"])
        d = cases.digest(u)
        d["inputs_crc32"] = format(inputs_crc, "08x")
        d["shape"] = list(u.shape)
        d["timers_nonzero"] = bool(prof.section0 > 0)
        idx = cases.sample_indices(u.size)
        samples[name + "_idx"] = idx.astype(np.int64)
        samples[name + "_bits"] = u.reshape(-1)[idx].view(np.uint32)
        golden[name] = d
        print(name, d["crc32"], d["maxabs"])
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(golden, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "golden_samples.npz"), **samples)


if __name__ == "__main__":
    main()
