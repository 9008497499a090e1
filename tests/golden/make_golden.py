#!/usr/bin/env python3
"""Generate the committed golden vectors from the REAL reference code that builds here.

Only the reference's header-only PCG library compiles in this image (oracle/Makefile target `ref`
-> oracle/_ref/libref_pcg.so, built from /root/reference/inst/include/pcg/*.hpp where they lie), so
that is what is captured: pcg64 output streams for fixed sub-seed rows, seeded exactly as
/root/reference/src/pcg.h:48-85 does.  The vectors travel to the GPU box; the reference does not.

Also stored: the known-answer fixtures of the reference's own sequencer tests
(/root/reference/tests/testthat/test-sequencer.R:82-161) and of its mutation-table test
(tests/testthat/test-vcf_IO.R:14-90) as plain data (inputs + expected outputs).

Run from the repo root:  python tests/golden/make_golden.py
"""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def pcg_vectors():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_pcg.so"))
    rows = [
        [0, 0, 0, 0, 0, 0, 0, 0],
        [1, 2, 3, 4, 5, 6, 7, 8],
        [0xFFFFFFFF] * 8,
        [0xDEADBEEF, 0x01234567, 0x89ABCDEF, 0x0BADF00D, 0xCAFEF00D, 0xD15EA5E5, 0x00000001, 0x80000000],
        [3141592653, 589793238, 462643383, 2795028841, 971693993, 751058209, 749445923, 78164062],
    ]
    out = []
    for r in rows:
        w = np.asarray(r, dtype=np.uint32)
        o = np.zeros(64, dtype=np.uint64)
        lib.ref_pcg64_outputs(w.ctypes.data_as(C.c_void_p), C.c_uint64(64), o.ctypes.data_as(C.c_void_p))
        out.append({"sub_seeds": [int(x) for x in r], "outputs": ["%016x" % int(x) for x in o]})
    lib.ref_pcg64_max.restype = C.c_uint64
    return {"source": "pcg64 (setseq_xsl_rr_128_64) of /root/reference/inst/include/pcg/pcg_random.hpp via "
                      "oracle/ref_pcg_driver.cpp; seeding per src/pcg.h:48-85",
            "max": "%016x" % lib.ref_pcg64_max(), "streams": out}


def known_answers():
    # test-sequencer.R:82-87: one quality (255) per nucleotide and position, count 1000
    # test-sequencer.R:91-124 (paired) and :128-161 (mate-pair)
    return {
        "source": "/root/reference/tests/testthat/test-sequencer.R:82-161",
        "chrom": "C" * 25 + "N" * 150 + "T" * 25,
        "read_length": 100, "n_reads": 10000, "frag_len_min": 200, "frag_len_max": 200,
        "profile": {"n_positions": 100, "quals": [255], "cum_counts": [1000]},
        "paired_expected_reads": sorted(["C" * 25 + "N" * 75, "A" * 25 + "N" * 75]),
        "matepair_expected_reads": sorted(["N" * 75 + "T" * 25, "N" * 75 + "G" * 25]),
    }


if __name__ == "__main__":
    with open(os.path.join(HERE, "pcg64_vectors.json"), "w") as fh:
        json.dump(pcg_vectors(), fh, indent=1)
    with open(os.path.join(HERE, "sequencer_known_answers.json"), "w") as fh:
        json.dump(known_answers(), fh, indent=1)
    print("wrote golden vectors")
