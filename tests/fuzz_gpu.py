#!/usr/bin/env python3
"""Randomised parity sweep on the GPU (not collected by pytest; tests/test_gpu_fuzz.py runs a short seeded slice).

Every case draws a random job -- genome shape (tiny chromosomes, N runs, odd bytes), read length, profile,
single/paired/mate-pair, fragment distribution, indel and duplicate probabilities, pool size, barcodes, lane
count, optional haplotypes made by the mutation-table builder -- and demands byte-identical FASTQ and equal seed
consumption from the HIP path and the CPU oracle.  Inputs the GPU path documents as unsupported must be refused
(JK_ERR_UNSUPPORTED), never answered differently.

    python tests/fuzz_gpu.py [--seconds 300] [--seed 1] [--kind illumina|pacbio|all]
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from helpers import builder_haplotypes, first_diff, job, run_hip, run_oracle       # noqa: E402


def random_genome(ja, rng):
    n_chrom = int(rng.choice([1, 1, 2, 3, 7]))
    sizes = [int(rng.choice([rng.integers(160, 400), rng.integers(400, 3000), rng.integers(3000, 60000)])) for _ in range(n_chrom)]
    style = rng.choice(["tcag", "tcag", "with_n", "odd"])
    seqs = []
    for n in sizes:
        if style == "tcag":
            alphabet = b"TCAG"
        elif style == "with_n":
            alphabet = b"TCAGTCAGTCAGTCAGN"
        else:
            alphabet = b"TCAGTCAGTCAGNnRY-\x00\x03"
        lut = np.frombuffer(alphabet, dtype=np.uint8)
        s = lut[rng.integers(0, lut.size, size=n)].copy()
        if style == "with_n" and n > 300 and rng.random() < 0.5:
            a = int(rng.integers(0, n - 200))
            s[a:a + int(rng.integers(10, 200))] = ord("N")
        seqs.append(s)
    return ja.RefGenome(seqs)


def illumina_case(ja, O, rng, case):
    os.environ["JK_HAP_MATERIALISE"] = str(case & 1)      # haplotype cases alternate between the two read paths
    g = random_genome(ja, rng)
    # sequencing system (built-in ART profile) and a read length it covers; None = the default for the length
    systems = [(None, 150), (None, 150), ("GA1", 44), ("GA2", 75), ("NS50", 75), ("HS10", 100), ("HS20", 100), ("HS25", 150),
               ("HSXn", 150), ("HSXt", 150), ("MSv1", 250), ("MSv3", 250)]
    seq_sys, max_len = systems[int(rng.integers(0, len(systems)))]
    read_length = int(rng.choice([x for x in (36, 44, 50, 75, 100, 125, 150, 150, 200, 250) if x <= max_len]))
    kind = rng.choice(["pe", "pe", "se", "mp"])
    frag_mean = float(rng.choice([read_length * 1.2, 300.0, 400.0, 900.0]))
    frag_sd = float(frag_mean / rng.choice([1.5, 4.0, 8.0]))
    j = job(paired=kind != "se", matepair=kind == "mp", frag_mean=frag_mean, frag_sd=frag_sd,
            prob_dup=float(rng.choice([0.0, 0.02, 0.3, 0.9])), read_pool_size=int(rng.choice([1, 2, 7, 100, 1000])),
            ins_prob1=float(rng.choice([0.0, 0.00009, 0.01, 0.1])), del_prob1=float(rng.choice([0.0, 0.00011, 0.01, 0.1])),
            ins_prob2=float(rng.choice([0.0, 0.00015, 0.02])), del_prob2=float(rng.choice([0.0, 0.00023, 0.05])),
            barcode=str(rng.choice(["", "", "ACGT", "TTGACCAN"])))
    if rng.random() < 0.3:
        j["frag_len_min"] = int(rng.integers(1, read_length + 50))
    if rng.random() < 0.3:
        j["frag_len_max"] = int(max(j["frag_len_min"] or 1, rng.integers(read_length // 2, 3 * read_length)))
    if j["frag_len_max"] is not None and (j["frag_len_min"] or read_length) > j["frag_len_max"]:
        j["frag_len_min"] = int(rng.integers(1, j["frag_len_max"] + 1))       # (min > max is an argument error in R as well)
    T = int(rng.choice([1, 2, 5, 64, 65, 300]))
    ends = 1 if kind == "se" else 2
    n_reads = int(rng.integers(1, 40)) * T * ends // int(rng.choice([1, 2, 3])) + int(rng.integers(0, 3))
    n_reads = max(n_reads, ends)
    desc = "illumina case %d: L=%d %s %s chroms=%s T=%d n=%d %s" % (case, read_length, seq_sys, kind, g.sizes(), T, n_reads,
                                                                 {k: v for k, v in j.items() if k not in ("paired", "matepair")})
    use_hap = rng.random() < 0.35 and min(g.sizes()) > 400
    words = ja.seed_words(int(rng.integers(0, 2 ** 31)), 64 * T * 8 + 256)
    paired = j["paired"] or j["matepair"]
    p1 = ja.read_profile(None, seq_sys, read_length, 1)
    p2 = ja.read_profile(None, seq_sys, read_length, 2) if paired else None
    try:
        if use_hap:
            from test_gpu_hap import hip_hap, oracle_hap
            hs = builder_haplotypes(ja, g.sizes(), int(rng.choice([1, 2, 4])), int(rng.choice([5, 40, 400])), seed=int(rng.integers(0, 10 ** 6)))
            probs = [float(x) for x in rng.choice([0.0, 1.0, 2.0], size=hs.n_haps())]
            if sum(probs) == 0:
                probs[0] = 1.0
            desc += " haps=%d probs=%s" % (hs.n_haps(), probs)
            bcs = [j["barcode"]] * hs.n_haps()
            h = hip_hap(ja, hs, read_length, words, n_reads, T, j, probs, bcs, seq_sys=seq_sys)
            o = oracle_hap(O, hs, p1, p2, words, n_reads, T, j, probs, bcs)
            got, want, used_h, used_o = (h[0], h[1]), (o[0], o[1]), h[3], o[2]
        else:
            h1, h2, _, used_h = run_hip(ja, g, (None, None), read_length, words, n_reads, T, j, seq_sys=seq_sys)
            o1, o2, used_o = run_oracle(O, g, p1, p2, words, n_reads, T, j)
            got, want = (h1, h2), (o1, o2)
    except ja.JackalopeHipError as e:
        if e.code == 2:                             # JK_ERR_UNSUPPORTED: a documented refusal
            return "refused", desc + " -> " + str(e)
        raise AssertionError(desc + "\nunexpected error: %s" % e)
    assert used_h == used_o, desc + "\nseed words: HIP %d oracle %d" % (used_h, used_o)
    for e in range(2 if paired else 1):
        if got[e] != want[e]:
            raise AssertionError(desc + "\nR%d differs at byte %d:\nHIP    %r\noracle %r" % ((e + 1,) + first_diff(got[e], want[e])))
    return "ok", desc


def pacbio_case(ja, O, rng, case, round3=True):
    """round3=False: the case generator of rounds 1-2 (the regression cases of tests/test_gpu_fuzz.py name its draws)."""
    from test_gpu_pacbio import hip
    if round3:
        # chromosomes from a few hundred bases (reads as long as their chromosome, no spare bases for deletions, windows that
        # end inside the 1000-character buffer the reference's thread starts with) to a few hundred kb
        small = rng.random() < 0.35
        sizes = [int(rng.integers(300, 6000)) if small else int(rng.integers(20_000, 300_000)) for _ in range(int(rng.choice([1, 2, 4])))]
    else:
        small = False
        sizes = [int(rng.integers(20_000, 300_000)) for _ in range(int(rng.choice([1, 2, 4])))]
    g = ja.synthetic_genome(sizes, seed=int(rng.integers(0, 10 ** 6)))
    T = int(rng.choice([1, 3, 64, 130]))
    n = int(rng.integers(1, 6)) * T + int(rng.integers(0, 3))
    if round3 and rng.random() < 0.2:
        n = int(rng.integers(20, 120)) * T                    # many reads per lane: the plan kernel's waves carry few lanes
    pb = {}
    if rng.random() < 0.6:
        if round3:
            pb["custom_read_lengths"] = sorted(int(x) for x in rng.integers(100, max(min(sizes) // 3, 102) if not small or rng.random() < 0.5 else 2 * max(sizes),
                                                                            size=int(rng.integers(1, 5))))
        else:
            pb["custom_read_lengths"] = sorted(int(x) for x in rng.integers(100, min(sizes) // 3, size=int(rng.integers(1, 5))))
    if rng.random() < 0.3:
        pb["prob_dup"] = float(rng.choice([0.1, 0.5]))
    if rng.random() < 0.3:
        pb["ins_prob"], pb["del_prob"], pb["sub_prob"] = float(rng.choice([0.05, 0.2])), float(rng.choice([0.02, 0.1])), float(rng.choice([0.005, 0.05]))
    if rng.random() < 0.2:
        pb["max_passes"] = int(rng.choice([1, 4, 20]))
    desc = "pacbio case %d: chroms=%s T=%d n=%d %s" % (case, sizes, T, n, pb)
    words = ja.seed_words(int(rng.integers(0, 2 ** 31)), 64 * T * 8 + 256)
    hs = None
    if rng.random() < 0.3:
        hs = builder_haplotypes(ja, sizes, int(rng.choice([1, 2, 3])), int(rng.choice([10, 300])), seed=int(rng.integers(0, 10 ** 6)))
        probs = [float(x) for x in rng.choice([0.5, 1.0, 2.0], size=hs.n_haps())]
        desc += " haps=%d probs=%s" % (hs.n_haps(), probs)
    try:
        if hs is not None:
            h, reads, used_h = hip(ja, hs, n, T, words, dict(pb, haplotype_probs=probs))
        else:
            h, reads, used_h = hip(ja, g, n, T, words, pb)
    except ja.JackalopeHipError as e:
        if e.code == 2:
            return "refused", desc + " -> " + str(e)
        raise AssertionError(desc + "\nunexpected error: %s" % e)
    if hs is not None:
        o, used_o, _ = O.pacbio_hap(hs, pb, hap_probs=probs, n_reads=n, n_threads=T, words=words)
    else:
        o, used_o, _ = O.pacbio_ref(g, pb, n_reads=n, n_threads=T, words=words)
    assert used_h == used_o, desc + "\nseed words: HIP %d oracle %d" % (used_h, used_o)
    if h != o:
        raise AssertionError(desc + "\nFASTQ differs at byte %d:\nHIP    %r\noracle %r" % first_diff(h, o))
    return "ok", desc


def genome_case(ja, O, rng, case):
    n_chroms = int(rng.choice([1, 2, 5, 24, 100]))
    len_mean = float(rng.choice([1, 7, 100, 2047, 2048, 5000, 60000]))
    len_sd = float(rng.choice([0, 0, len_mean * 0.1, len_mean * 0.9]))
    pi = rng.choice([0.0, 0.1, 1.0, 3.0], size=4)
    if pi.sum() == 0:
        pi[int(rng.integers(0, 4))] = 1.0
    T = int(rng.choice([1, 1, 2, 7, 64]))
    desc = "genome case %d: n=%d mean=%g sd=%g pi=%s T=%d" % (case, n_chroms, len_mean, len_sd, pi.tolist(), T)
    words = ja.seed_words(int(rng.integers(0, 2 ** 31)), 8 * T)
    want, used = O.create_genome(n_chroms, len_mean, len_sd, pi.tolist(), T, words)
    g = ja.create_genome(n_chroms, len_mean, len_sd, pi.tolist(), T, seed_words=words)
    assert g.seed_words_used() == used and g.sizes() == [len(c) for c in want], desc
    for i, w in enumerate(want):
        assert g.chrom(i).tobytes() == w, desc + " chromosome %d" % i
    g.close()
    return "ok", desc


def fasta_case(ja, O, rng, case, tmp):
    import gzip
    n = int(rng.choice([1, 2, 9]))
    width = int(rng.choice([1, 7, 60, 80, 4096, 100000]))
    nl = b"\r\n" if rng.random() < 0.25 else b"\n"
    alphabet = np.frombuffer(b"TCAGTCAGTCAGNtcagnRYKM-*", dtype=np.uint8)
    fn = os.path.join(tmp, "f%d.fa" % case)
    with open(fn, "wb") as f:
        for i in range(n):
            name = ("c%d" % i) + (" some text here" if rng.random() < 0.5 else "")
            f.write(b">" + name.encode() + nl)
            L = int(rng.choice([0, 1, 79, 80, 81, 4095, 4096, 4097, rng.integers(100, 200000)]))
            seq = bytes(alphabet[rng.integers(0, alphabet.size, size=L)])
            for a in range(0, L, width):
                f.write(seq[a:a + width] + (nl if (a + width < L or rng.random() < 0.9) else b""))
    files = [fn]
    if rng.random() < 0.3:
        gz = fn + ".gz"
        open(gz, "wb").write(gzip.compress(open(fn, "rb").read()))
        files = [gz]
    cut = bool(rng.random() < 0.5)
    desc = "fasta case %d: %d chromosomes, width %d, %s, cut_names=%s, %s" % (case, n, width, "CRLF" if nl != b"\n" else "LF", cut, files[0][-3:])
    want_names, want = O.read_fasta(files, None, cut_names=cut)
    g = ja.read_fasta(files, cut_names=cut)
    assert [x.encode() for x in g.names] == want_names, desc
    assert g.sizes() == [len(c) for c in want], desc
    for i, w in enumerate(want):
        assert g.chrom(i).tobytes() == w, desc + " chromosome %d" % i
    g.close()
    return "ok", desc


def bgzf_case(ja, O, rng, case):
    import gzip
    n = int(rng.choice([0, 1, 63, 64, 65, 0xff00 - 1, 0xff00, 0xff00 + 1, rng.integers(1, 400000)]))
    style = rng.choice(["fastq", "uniform", "skewed", "runs", "binary", "records", "records", "lines"])
    if style == "fastq":
        data = bytes(np.frombuffer(b"ACGTN\n@+IIIIFFF#", dtype=np.uint8)[rng.integers(0, 16, size=n)])
    elif style == "records":
        # FASTQ-shaped: records of (nearly) equal length whose lines repeat the previous record's in the same columns -- what
        # the device matcher looks for: id prefixes and suffixes, "+", low-entropy qualities, whole duplicated records,
        # records a byte longer or shorter than their predecessor (the distance changes at a line start), empty lines
        L = int(rng.choice([1, 4, 30, 75, 150, 300, 2000]))
        out, prev = [], None
        alpha_q = np.frombuffer(b"CCCGGGGJJ=8", dtype=np.uint8)
        while sum(len(r) for r in out) < n:
            if prev is not None and rng.random() < 0.1:
                rec = prev
            else:
                ll = max(L + int(rng.choice([0, 0, 0, 1, -1])), 0)
                hdr = b"@REF-chrom%d-%d-%s/%d" % (int(rng.integers(0, 3)), int(rng.integers(0, 10 ** int(rng.integers(1, 9)))), b"FR"[int(rng.integers(0, 2)):][:1], int(rng.integers(1, 3)))
                seq = np.frombuffer(b"TCAG", dtype=np.uint8)[rng.integers(0, 4, size=ll)].tobytes()
                qual = alpha_q[rng.integers(0, alpha_q.size, size=ll)].tobytes()
                rec = hdr + b"\n" + seq + b"\n+\n" + qual + b"\n"
            out.append(rec); prev = rec
        data = b"".join(out)[:n]
    elif style == "lines":
        out = []
        while sum(len(r) for r in out) < n:
            out.append(np.frombuffer(b"ab", dtype=np.uint8)[rng.integers(0, 2, size=int(rng.integers(0, 200)))].tobytes() + b"\n")
        data = b"".join(out)[:n]
    elif style == "uniform":
        data = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
    elif style == "skewed":
        data = np.minimum(rng.geometric(0.3, size=n), 255).astype(np.uint8).tobytes()
    elif style == "runs":
        data = np.repeat(rng.integers(0, 256, size=n // 50 + 1, dtype=np.uint8), 50)[:n].tobytes()
    else:
        data = (rng.integers(0, 4, size=n, dtype=np.uint8) * 85).tobytes()
    desc = "bgzf case %d: %d bytes, %s" % (case, n, style)
    comp = ja.bgzf_deflate(data).cpu().numpy().tobytes()
    assert len(comp) <= ja.bgzf_bound(n), desc
    assert gzip.decompress(comp) == data, desc
    return "ok", desc


def run(seconds, seed, kind, max_cases=None, verbose=True, first_case=0):
    import jackalope_amd as ja
    import oracle_lib as O
    O.lib()
    t0, case, stats = time.time(), first_case, {"ok": 0, "refused": 0}
    while time.time() - t0 < seconds and (max_cases is None or case < first_case + max_cases):
        rng = np.random.default_rng([seed, case])          # every case is reproducible on its own (--first-case N --cases 1)
        which = kind if kind != "all" else ("pacbio" if rng.random() < 0.25 else "illumina")
        if which == "genome":
            res, desc = genome_case(ja, O, rng, case)
        elif which == "fasta":
            import tempfile
            with tempfile.TemporaryDirectory(prefix="jk_fuzz_") as tmp:
                res, desc = fasta_case(ja, O, rng, case, tmp)
        elif which == "bgzf":
            res, desc = bgzf_case(ja, O, rng, case)
        else:
            res, desc = (pacbio_case if which == "pacbio" else illumina_case)(ja, O, rng, case)
        stats[res] += 1
        if verbose and (res == "refused" or case % 25 == 0):
            print("[%5.0fs] %s: %s" % (time.time() - t0, res, desc[:200]), flush=True)
        case += 1
    os.environ.pop("JK_HAP_MATERIALISE", None)
    return stats


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--kind", choices=["illumina", "pacbio", "genome", "fasta", "bgzf", "all"], default="all",
                    help="all = illumina + pacbio; the genome-side kinds are run on their own")
    ap.add_argument("--first-case", type=int, default=0)
    ap.add_argument("--cases", type=int, default=None)
    a = ap.parse_args()
    print("done:", run(a.seconds, a.seed, a.kind, max_cases=a.cases, first_case=a.first_case))
