"""GPU: the streaming sink (jk_session_run, the one-shot entry points, jobs).  Every generator launch's FASTQ goes to
the files as it completes -- the reference flushes pool by pool under `omp critical`, src/hts.h:401-412 -- so no image
stays in device memory.  The files must equal the resident image of the same job (and therefore the oracle), whatever
the number of launches, the sink (plain / device BGZF / host BGZF / gzip) and the number of devices."""
import gzip
import os
import threading
import time

import numpy as np
import pytest

from helpers import job, run_oracle, open_hip

pytestmark = pytest.mark.gpu


def resident(ja, g, words, n_reads, T, **kw):
    with ja.illumina(g, None, n_reads, 150, True, n_threads=T, seed_words=words, _session=True, **kw) as s:
        s.generate()
        return s.fetch(0), s.fetch(1), s.n_batches()


def read(fn):
    with open(fn, "rb") as f:
        return f.read()


def test_streamed_files_equal_the_resident_image(ja, O, hs25, tmp_path):
    g = ja.synthetic_genome([300_000, 200_000, 50_000], seed=21)
    n_reads, T = 400_000, 4096
    words = ja.seed_words(77, 16 * T)
    r1, r2, _ = resident(ja, g, words, n_reads, T)
    o1, o2, _ = run_oracle(O, g, hs25[0], hs25[1], words, n_reads, T, job())
    assert r1 == o1 and r2 == o2
    # one-shot, default batching (one launch) and small launches (17 of them)
    for tag, mbb in (("a", 0), ("b", 4 << 20)):
        pre = str(tmp_path / tag)
        ja.illumina(g, pre, n_reads, 150, True, n_threads=T, seed_words=words, max_batch_bytes=mbb)
        assert read(pre + "_R1.fq") == r1 and read(pre + "_R2.fq") == r2
    # streaming session: batches, progress, sizes
    pre = str(tmp_path / "c")
    with ja.illumina(g, pre, n_reads, 150, True, n_threads=T, seed_words=words, max_batch_bytes=4 << 20, _session=True,
                     stream_output=True) as s:
        assert s.n_batches() > 8
        assert s.progress() == (0, n_reads)
        with pytest.raises(ja.JackalopeHipError):
            s.generate()                     # a streaming session has no resident image
        s.run()
        assert s.progress() == (n_reads, n_reads)
        sizes, reads = s.sizes()
        assert reads == n_reads and sizes == [len(r1), len(r2)]
    assert read(pre + "_R1.fq") == r1 and read(pre + "_R2.fq") == r2
    # null sink: nothing written, same counts
    with ja.illumina(g, None, n_reads, 150, True, n_threads=T, seed_words=words, max_batch_bytes=4 << 20, _session=True,
                     stream_output=True) as s:
        s.run()
        assert s.sizes() == ([len(r1), len(r2)], n_reads)


@pytest.mark.parametrize("method,level", [("bgzip", 6), ("bgzip-host", 4), ("gzip", 5)])
def test_streamed_compressed_sinks(ja, tmp_path, method, level):
    g = ja.synthetic_genome([400_000], seed=22)
    # (gzip: the R-level check allows it with one thread only, R/hts_illumina.R:648-652)
    n_reads, T = (200_000, 2048) if method != "gzip" else (3000, 1)
    words = ja.seed_words(78, 16 * T)
    r1, r2, _ = resident(ja, g, words, n_reads, T)
    pre = str(tmp_path / "z")
    s = ja.illumina(g, pre, n_reads, 150, True, n_threads=T, seed_words=words, max_batch_bytes=6 << 20, compress=level,
                    comp_method=method, _session=True, stream_output=True)
    with s:
        assert s.n_batches() > 3 or method == "gzip"
        s.run()
    for fn, want in ((pre + "_R1.fq.gz", r1), (pre + "_R2.fq.gz", r2)):
        raw = read(fn)
        assert gzip.decompress(raw) == want
        if method != "gzip":
            # BGZF container: a chain of members with a 'BC' extra field, each <= 64 KiB, ending in the EOF block
            at, n_members = 0, 0
            while at < len(raw):
                assert raw[at:at + 4] == b"\x1f\x8b\x08\x04" and raw[at + 12:at + 14] == b"BC"
                at += int.from_bytes(raw[at + 16:at + 18], "little") + 1
                n_members += 1
            assert at == len(raw) and raw[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
            assert n_members > len(want) // 0xff00


def test_two_device_slots_write_the_same_files(ja, O, hs25, tmp_path):
    """In-library fan-out of the one-shot calls (args.devices): one host thread per device, contiguous lane blocks,
    parts appended in order.  Only one GPU exists here, so both slots are device 0; the code path is the multi-GPU one."""
    from jackalope_amd.genome import random_haplotypes
    ref = ja.synthetic_genome([150_000, 90_000], seed=23)
    hs = random_haplotypes(ref, 3, seed=5)
    n_reads, T = 120_000, 1000
    words = ja.seed_words(79, hs.seed_budget(T) + 64)
    with ja.illumina(hs, None, n_reads, 150, True, n_threads=T, seed_words=words, _session=True) as s:
        s.generate()
        r1, r2 = s.fetch(0), s.fetch(1)
    one, two, three = str(tmp_path / "one"), str(tmp_path / "two"), str(tmp_path / "three")
    ja.illumina(hs, one, n_reads, 150, True, n_threads=T, seed_words=words)
    ja.illumina(hs, two, n_reads, 150, True, n_threads=T, seed_words=words, devices=[0, 0])
    ja.illumina(hs, three, n_reads, 150, True, n_threads=T, seed_words=words, devices=[0, 0, 0], compress=True)
    for e, want in ((1, r1), (2, r2)):
        assert read("%s_R%d.fq" % (one, e)) == want
        assert read("%s_R%d.fq" % (two, e)) == want
        assert gzip.decompress(read("%s_R%d.fq.gz" % (three, e))) == want
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]
    # sep_files over two device slots: one file pair per haplotype, each equal to the single-device run's
    ja.illumina(hs, str(tmp_path / "s1"), n_reads, 150, True, n_threads=T, seed_words=words, sep_files=True)
    ja.illumina(hs, str(tmp_path / "s2"), n_reads, 150, True, n_threads=T, seed_words=words, sep_files=True, devices=[0, 0])
    total = 0
    for h in hs.hap_names():
        for e in (1, 2):
            a, b = read("%s_%s_R%d.fq" % (tmp_path / "s1", h, e)), read("%s_%s_R%d.fq" % (tmp_path / "s2", h, e))
            assert a == b
            total += a.count(b"\n") // 4
    assert total == n_reads


def test_pacbio_streams_and_fans_out(ja, O, tmp_path):
    ref = ja.synthetic_genome([600_000], seed=24)
    n_reads, T = 3000, 512
    words = ja.seed_words(80, 16 * T)
    o, _, _ = O.pacbio_ref(ref, {}, n_reads=n_reads, n_threads=T, words=words)
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    ja.pacbio(ref, a, n_reads, n_threads=T, seed_words=words, max_batch_bytes=8 << 20)
    ja.pacbio(ref, b, n_reads, n_threads=T, seed_words=words, devices=[0, 0], compress=3)
    assert read(a + "_R1.fq") == o
    got = gzip.decompress(read(b + "_R1.fq.gz"))
    if got != o:
        from helpers import first_diff
        raise AssertionError("lengths %d / %d; first difference at byte %d:\nfiles  %r\noracle %r" % ((len(got), len(o)) + first_diff(got, o)))


def test_job_progress_and_abort(ja, tmp_path):
    """A job runs on a worker thread while the calling thread polls its progress and can stop it (what the Rcpp shim
    does with Progress::check_abort, src/hts.h:396-399)."""
    g = ja.synthetic_genome([2_000_000], seed=25)
    n_reads, T = 6_000_000, 1 << 16
    words = ja.seed_words(81, 16 * T)
    flag = np.zeros(1, dtype=np.int32)
    jb = ja.illumina(g, str(tmp_path / "j"), n_reads, 150, True, n_threads=T, seed_words=words, max_batch_bytes=32 << 20,
                     abort_flag=flag, _job=True)
    with jb:
        assert jb.n_files() == 1 and jb.progress() == (0, n_reads)
        jb.plan_next()
        assert jb.seed_words_used() == 16 * T
        seen, err = [], []

        def work():
            try:
                jb.run()
            except Exception as e:       # noqa: BLE001
                err.append(e)
        th = threading.Thread(target=work)
        th.start()
        while th.is_alive():
            d, t = jb.progress()
            seen.append(d)
            if d > 0:
                flag[0] = 1              # stop it as soon as some reads are out
            time.sleep(0.002)
        th.join()
        assert err and err[0].code == 6          # JK_ERR_ABORTED
        assert 0 < max(seen) < n_reads
    # the same job to completion
    flag[0] = 0
    jb = ja.illumina(g, str(tmp_path / "k"), n_reads, 150, True, n_threads=T, seed_words=words, max_batch_bytes=32 << 20,
                     abort_flag=flag, _job=True)
    with jb:
        jb.plan_next()
        jb.run()
        assert jb.progress() == (n_reads, n_reads)
    assert read(str(tmp_path / "k") + "_R1.fq").count(b"\n") == 4 * (n_reads // 2)


@pytest.mark.parametrize("compress", [False, True])
def test_streamed_pacbio_survives_a_replan(ja, O, tmp_path, compress):
    """A streaming run whose scratch (or image) turns out too small is planned again, larger, and run again INTO THE SAME
    FILES: the first attempt's sink must be quiet -- no writer thread still holding a task with the old descriptor -- before
    the files are closed and their names reopened (the sink's pipe outlives the attempt), or old bytes land in the new
    file.  Plain (pwrite at offsets) and device BGZF (compressed offsets change with the launch layout)."""
    g = ja.synthetic_genome([900_000, 300_000], seed=71)
    n_reads, T = 3000, 600
    words = ja.seed_words(91, 16 * T)
    pb = {"custom_read_lengths": [3000, 9000, 20000]}
    o, _, _ = O.pacbio_ref(g, pb, n_reads=n_reads, n_threads=T, words=words)
    for var, val in (("JK_PB_POOL_SCALE", "0.05"), ("JK_PB_IMAGE_SCALE", "0.05")):
        pre = str(tmp_path / ("p" + var[-11:-6] + str(int(compress))))
        os.environ[var] = val
        try:
            with ja.pacbio(g, pre, n_reads, n_threads=T, seed_words=words, max_batch_bytes=8 << 20, compress=compress, _session=True,
                           stream_output=True, **pb) as s:
                s.run()
                assert s.retries() >= 1, var
                assert s.sizes() == ([len(o)], n_reads)
        finally:
            del os.environ[var]
        raw = read(pre + "_R1.fq" + (".gz" if compress else ""))
        assert (gzip.decompress(raw) if compress else raw) == o, var


def test_two_jobs_in_one_process_share_the_device_arena(ja, hs25, tmp_path):
    """The per-haplotype loop of sep_files (src/hts.h:512-552) and repeated calls from one R session open session after
    session: the large device buffers of a closed session are parked and the next session of the same shape takes them
    instead of going through hipMalloc (which clears fresh VRAM: seconds for a 100 GB run).  Same files both times."""
    g = ja.synthetic_genome([2_000_000], seed=23)
    n_reads, T = 2_000_000, 65536
    words = ja.seed_words(79, 16 * T)
    ja.arena_trim()
    pre = [str(tmp_path / "j0"), str(tmp_path / "j1")]
    ja.illumina(g, pre[0], n_reads, 150, True, n_threads=T, seed_words=words)
    st0 = ja.arena_stats()
    assert st0["bytes"] > 100 << 20              # pools and image slots of the closed session are parked
    ja.illumina(g, pre[1], n_reads, 150, True, n_threads=T, seed_words=words)
    st1 = ja.arena_stats()
    assert st1["hits"] >= st0["hits"] + 4 and st1["bytes"] == st0["bytes"]      # ... and were taken again, not re-made
    for e in (1, 2):
        assert read("%s_R%d.fq" % (pre[0], e)) == read("%s_R%d.fq" % (pre[1], e))
    ja.arena_trim()
    assert ja.arena_stats()["bytes"] == 0


def test_pipelined_steps_give_the_same_image(ja, O, hs25):
    """jk_session_generate_async / jk_session_wait: two passes in flight -- the second pass's generator launches run
    beside the first pass's last compaction, on the other pool set.  Every pass must leave exactly the image a lone
    generate() leaves (and that the oracle makes)."""
    g = ja.synthetic_genome([500_000, 100_000], seed=24)
    n_reads, T = 300_000, 6000
    words = ja.seed_words(80, 16 * T)
    o1, o2, _ = run_oracle(O, g, hs25[0], hs25[1], words, n_reads, T, job())
    for mbb in (0, 3 << 20):                      # one launch per pass; many launches per pass
        with ja.illumina(g, None, n_reads, 150, True, n_threads=T, seed_words=words, max_batch_bytes=mbb, _session=True) as s:
            s.generate_async()
            s.generate_async()
            with pytest.raises(ja.JackalopeHipError):
                s.generate_async()                # two in flight at most
            s.wait()
            assert s.sizes() == ([len(o1), len(o2)], n_reads)
            s.generate_async()
            s.wait()
            s.wait()
            with pytest.raises(ja.JackalopeHipError):
                s.wait()
            assert s.fetch(0) == o1 and s.fetch(1) == o2
            s.generate()                          # and the plain call still works afterwards
            assert s.fetch(0) == o1 and s.fetch(1) == o2 and s.timing_ms()["total"] > 0


def test_pipelined_pacbio_steps_give_the_same_image(ja, O):
    """The same for a PacBio session: the first plan kernel of the second pass runs beside the last emit kernel of the
    first, on the set of records and masks that the first pass's last-but-one launch used."""
    g = ja.synthetic_genome([900_000], seed=26)
    n_reads, T = 2400, 300
    words = ja.seed_words(82, 16 * T)
    pb = {"custom_read_lengths": [400, 2500, 9000]}
    o, _, _ = O.pacbio_ref(g, pb, n_reads=n_reads, n_threads=T, words=words)
    for mbb in (0, 2 << 20):                      # one launch per pass; five or six launches per pass
        with ja.pacbio(g, None, n_reads, n_threads=T, seed_words=words, max_batch_bytes=mbb, _session=True, **pb) as s:
            assert (s.n_batches() == 1) == (mbb == 0)
            s.generate_async()
            s.generate_async()
            with pytest.raises(ja.JackalopeHipError):
                s.generate_async()                # two in flight at most
            s.wait()
            assert s.sizes() == ([len(o)], n_reads)
            s.generate_async()
            s.wait()
            s.wait()
            assert s.fetch(0) == o
            s.generate()
            assert s.fetch(0) == o
