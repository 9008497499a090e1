#!/usr/bin/env python3
"""Convert ART-style Illumina quality profiles into the compact .npz form this repo bundles.

Input  : the tab-separated `<nt>\\t<pos>\\t<q...>` / `<nt>\\t<pos>\\t<cumulative counts...>` line pairs of
         an ART profile (grammar: reference R/hts_illumina.R:211-245 `read_profile`).  By default
         the profiles shipped with the reference package are read in place from
         /root/reference/inst/art_profiles (data files, not code).
Output : jackalope_amd/data/art_profiles/<name>.npz holding, for the T/C/A/G rows only,
         `n_quals[4, P]` (int32), `quals` (uint8, flat) and `cum_counts` (int64, flat) in
         nt-major, position-major order.  Probabilities are NOT stored: they are recomputed at load
         time exactly as read_profile does (successive differences / their sum), so no float
         formatting is involved.

Usage: python tools/convert_art_profiles.py [SRC_DIR] [NAME ...]
"""
import gzip
import os
import sys

import numpy as np

DEFAULT_SRC = "/root/reference/inst/art_profiles"
DEFAULT_NAMES = None   # None = every *.txt.gz in SRC_DIR


def _as_count(x):
    v = float(x)
    if v != int(v):
        raise ValueError("non-integral cumulative count %r" % x)
    return int(v)


def parse(path):
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt") as fh:
        lines = [ln.rstrip("\n") for ln in fh]
    lines = [ln for ln in lines if ln[:1] in "TCAG" and ln]
    rows = {nt: {} for nt in "TCAG"}
    for i in range(0, len(lines), 2):
        # R's strsplit() drops one trailing empty field; some profiles end lines with a tab
        a = lines[i].split("\t")
        b = lines[i + 1].split("\t")
        if a and a[-1] == "":
            a.pop()
        if b and b[-1] == "":
            b.pop()
        if a[:2] != b[:2] or len(a) != len(b):
            raise ValueError("malformed profile %s at pair %d" % (path, i))
        rows[a[0]][int(a[1])] = ([int(x) for x in a[2:]], [_as_count(x) for x in b[2:]])
    npos = max(len(rows[nt]) for nt in "TCAG")
    n_quals = np.zeros((4, npos), dtype=np.int32)
    quals, cum = [], []
    for k, nt in enumerate("TCAG"):
        if sorted(rows[nt]) != list(range(len(rows[nt]))) or len(rows[nt]) != npos:
            raise ValueError("positions of %s in %s are not 0..P-1" % (nt, path))
        for pos in range(npos):
            q, c = rows[nt][pos]
            n_quals[k, pos] = len(q)
            quals += q
            cum += c
    return n_quals, np.asarray(quals, dtype=np.uint8), np.asarray(cum, dtype=np.int64)


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_SRC
    names = sys.argv[2:] or DEFAULT_NAMES or sorted(f[:-7] for f in os.listdir(src) if f.endswith(".txt.gz"))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                           "jackalope_amd", "data", "art_profiles")
    os.makedirs(out_dir, exist_ok=True)
    for name in names:
        path = os.path.join(src, name + ".txt.gz")
        n_quals, quals, cum = parse(path)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), n_quals=n_quals, quals=quals, cum_counts=cum)
        print("%s: %d positions, %d entries" % (name, n_quals.shape[1], quals.size))


if __name__ == "__main__":
    main()
