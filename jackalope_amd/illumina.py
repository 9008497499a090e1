"""`illumina()` -- the R-level entry point of the path, mirrored in Python over the C ABI.

Argument names, defaults, checks and error texts follow /root/reference/R/hts_illumina.R:593-732
(`illumina`) and :277-396 (`check_illumina_args`); the call it ends in is
``jk_illumina_ref`` / ``jk_illumina_hap`` instead of ``illumina_ref_cpp`` / ``illumina_hap_cpp``.
Differences a user sees: ``n_threads`` counts GPU lanes (see include/jackalope_hip.h) and the
32-bit seed words come from ``seed`` (a SplitMix64 stream) or ``seed_words`` because R's RNG is
not available outside R.
"""
import ctypes as C
import os

import numpy as np

from . import _abi
from .genome import RefGenome
from .profiles import read_profile
from .rng import seed_words as _seed_words, illumina_ref_seed_budget


def _is_num(x, lo=None, hi=None):
    if isinstance(x, bool) or not isinstance(x, (int, float, np.integer, np.floating)):
        return False
    if lo is not None and x < lo:
        return False
    if hi is not None and x > hi:
        return False
    return True


def _err(arg, what):
    raise ValueError("\nFor the function `illumina`, argument `%s` must be %s." % (arg, what))


def check_illumina_args(obj, n_reads, read_length, paired, frag_mean, frag_sd, matepair, seq_sys, profile1,
                        profile2, ins_prob1, del_prob1, ins_prob2, del_prob2, frag_len_min, frag_len_max,
                        haplotype_probs, barcodes, prob_dup, sep_files, compress, comp_method, n_threads,
                        read_pool_size, show_progress):
    """R/hts_illumina.R:277-396 (the checks that do not depend on R object classes)."""
    if not isinstance(obj, RefGenome) and not hasattr(obj, "n_haps"):
        _err("obj", 'a "ref_genome" or "haplotypes" object')
    for nm, v in (("n_reads", n_reads), ("read_length", read_length), ("n_threads", n_threads),
                  ("read_pool_size", read_pool_size)):
        if not _is_num(v, 1) or int(v) != v:
            _err(nm, "a single integer >= 1")
    for nm, v in (("paired", paired), ("matepair", matepair), ("sep_files", sep_files), ("show_progress", show_progress)):
        if not isinstance(v, (bool, np.bool_)):
            _err(nm, "a single logical")
    for nm, v in (("frag_mean", frag_mean), ("frag_sd", frag_sd)):
        if not _is_num(v) or v <= 0:
            _err(nm, "a single number > 0")
    for nm, v in (("ins_prob1", ins_prob1), ("del_prob1", del_prob1), ("ins_prob2", ins_prob2),
                  ("del_prob2", del_prob2), ("prob_dup", prob_dup)):
        if not _is_num(v, 0, 1):
            _err(nm, "a single number in range [0,1]")
    if ins_prob1 + del_prob1 >= 1 or ins_prob2 + del_prob2 >= 1:
        raise ValueError("\nFor the function `illumina`, the sum of insertion and deletion probabilities must be < 1 "
                         "for each read.")
    for nm, v in (("frag_len_min", frag_len_min), ("frag_len_max", frag_len_max)):
        if v is not None and (not _is_num(v, 1) or int(v) != v):
            _err(nm, "NULL or a single integer >= 1")
    for nm, v in (("seq_sys", seq_sys), ("profile1", profile1), ("profile2", profile2)):
        if v is not None and not isinstance(v, str):
            _err(nm, "NULL or a single string")
    if profile1 is None and profile2 is not None:
        raise ValueError("\nFor the function `illumina`, if you don't provide a custom profile for read 1, you "
                         "cannot provide one for read 2.")
    if profile1 is not None and profile2 is None and paired:
        raise ValueError("\nFor the function `illumina`, if you provide a custom profile for read 1 and want "
                         "paired-end reads, you must also provide one for read 2.")
    if comp_method not in ("gzip", "bgzip", "bgzip-host"):     # "bgzip-host": this library's zlib-on-the-host variant
        _err("comp_method", '"gzip" or "bgzip"')
    if not (isinstance(compress, (bool, np.bool_)) or (_is_num(compress, 1, 9) and int(compress) == compress)):
        _err("compress", "a single logical or integer from 1 to 9")


def _check_file_existence(fns, compress, overwrite):
    """R/util.R:59-81"""
    if compress:
        fns = [f + ".gz" for f in fns]
    if not overwrite and any(os.path.exists(f) for f in fns):
        raise FileExistsError("\nOne or more of the output files already exists, and argument `overwrite` is FALSE.")


class Job:
    """A whole illumina()/pacbio() call in two steps per output file set (jk_job_*): plan_next() reads the seed words
    on the calling thread, run() generates and writes on any thread while progress() may be polled from another."""

    def __init__(self, handle, keep):
        self._h = handle
        self._keep = keep

    def n_files(self):
        return int(_abi.lib().jk_job_n_files(self._h))

    def plan_next(self):
        _abi.check(_abi.lib().jk_job_plan_next(self._h))

    def run(self):
        rc = _abi.lib().jk_job_run(self._h)          # (called on worker threads too: the error text is thread-local there)
        if rc != _abi.JK_OK:
            raise _abi.JackalopeHipError(rc, _abi.lib().jk_last_error().decode("utf-8", "replace"))

    def progress(self):
        d, t = C.c_uint64(), C.c_uint64()
        _abi.check(_abi.lib().jk_job_progress(self._h, C.byref(d), C.byref(t)))
        return int(d.value), int(t.value)

    def seed_words_used(self):
        return int(_abi.lib().jk_job_seed_words_used(self._h))

    def close(self):
        if self._h:
            _abi.lib().jk_job_free(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class IlluminaSession:
    """Generated FASTQ held in HBM (jk_session_*): open -> generate -> fetch / write."""

    def __init__(self, handle, keep):
        self._h = handle
        self._keep = keep

    def generate(self):
        _abi.check(_abi.lib().jk_session_generate(self._h))
        return self

    def generate_async(self):
        """Queue one pass (at most two may be queued); wait() completes the oldest.  Consecutive passes overlap: the next
        one's generator launches run beside this one's last compaction."""
        _abi.check(_abi.lib().jk_session_generate_async(self._h))
        return self

    def wait(self):
        _abi.check(_abi.lib().jk_session_wait(self._h))
        return self

    def run(self):
        """Streaming sessions (stream_output=True): generate and write the files (or the null sink) in one pass."""
        _abi.check(_abi.lib().jk_session_run(self._h))
        return self

    def progress(self):
        d, t = C.c_uint64(), C.c_uint64()
        _abi.check(_abi.lib().jk_session_progress(self._h, C.byref(d), C.byref(t)))
        return int(d.value), int(t.value)

    def sizes(self):
        b = (C.c_uint64 * 2)()
        r = C.c_uint64()
        ne = C.c_uint32()
        _abi.check(_abi.lib().jk_session_sizes(self._h, b, C.byref(r), C.byref(ne)))
        return [int(b[i]) for i in range(ne.value)], int(r.value)

    def fetch(self, end):
        sizes, _ = self.sizes()
        out = np.empty(sizes[end], dtype=np.uint8)
        _abi.check(_abi.lib().jk_session_fetch(self._h, end, out.ctypes.data, out.size))
        return out.tobytes()

    def fetch_range(self, end, byte_off, n):
        """Bytes [byte_off, byte_off + n) of the image of read end `end` (jk_session_fetch_range)."""
        out = np.empty(int(n), dtype=np.uint8)
        _abi.check(_abi.lib().jk_session_fetch_range(self._h, end, int(byte_off), int(n), out.ctypes.data))
        return out.tobytes()

    def write_shard(self, file_offsets):
        off = (C.c_uint64 * 2)(*([int(x) for x in file_offsets] + [0])[:2])
        _abi.check(_abi.lib().jk_session_write_shard(self._h, off))

    def shard_seed_words(self):
        b, e = C.c_uint64(), C.c_uint64()
        _abi.check(_abi.lib().jk_session_shard_seed_words(self._h, C.byref(b), C.byref(e)))
        return int(b.value), int(e.value)

    def device_ptr(self, end):
        p = C.c_void_p()
        _abi.check(_abi.lib().jk_session_device_ptr(self._h, end, C.byref(p)))
        return p.value

    def write(self):
        _abi.check(_abi.lib().jk_session_write(self._h))

    def timing_ms(self):
        ms = (C.c_double * 3)()
        _abi.check(_abi.lib().jk_session_timing(self._h, ms))
        return {"generate_kernel": ms[0], "scan_compact": ms[1], "total": ms[2]}

    def n_batches(self):
        return int(_abi.lib().jk_session_batches(self._h))

    def seed_words_used(self):
        return int(_abi.lib().jk_session_seed_words_used(self._h))

    def retries(self):
        return int(_abi.lib().jk_session_retries(self._h))

    def rare_branch_lanes(self):
        """(count, lanes): how often a cut point's low word decided an alias draw since open, and in which lanes (up to 61)."""
        n = C.c_uint32(0)
        lanes = (C.c_uint32 * 61)()
        _abi.check(_abi.lib().jk_session_rare_branch_lanes(self._h, C.byref(n), lanes, 61))
        return int(n.value), [int(x) for x in lanes[:min(61, n.value)]]

    def lane_bytes(self, end, n_lanes):
        out = np.empty(n_lanes, dtype=np.uint64)
        _abi.check(_abi.lib().jk_session_lane_bytes(self._h, end, out.ctypes.data, n_lanes))
        return out

    def close(self):
        if self._h:
            _abi.lib().jk_session_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_args(out_prefix, n_reads, paired, matepair, prof1, prof2, ins_prob1, del_prob1, ins_prob2, del_prob2,
              frag_len_shape, frag_len_scale, frag_len_min, frag_len_max, barcodes, prob_dup, n_threads,
              read_pool_size, words, compress=0, comp_method="bgzip", sep_files=False, haplotype_probs=None,
              lane_begin=0, lane_end=0, device=0, max_batch_bytes=0, seed_fn=None, abort_flag=None, seed_offset_words=None,
              devices=None, stream_output=False):
    """Assemble jk_illumina_args; returns (struct, keep-alive list)."""
    a = _abi.IlluminaArgs()
    keep = []
    a.paired, a.matepair = int(bool(paired)), int(bool(matepair))
    a.out_prefix = (out_prefix or "").encode()
    a.sep_files = int(bool(sep_files))
    a.compress = int(compress)
    a.comp_method = comp_method.encode()
    a.n_reads = int(n_reads)
    a.prob_dup = float(prob_dup)
    a.n_threads = int(n_threads)
    a.show_progress = 0
    a.read_pool_size = int(read_pool_size)
    if haplotype_probs is not None:
        hp = np.ascontiguousarray(haplotype_probs, dtype=np.float64)
        keep.append(hp)
        a.haplotype_probs = hp.ctypes.data_as(C.POINTER(C.c_double))
    a.frag_len_shape, a.frag_len_scale = float(frag_len_shape), float(frag_len_scale)
    a.frag_len_min, a.frag_len_max = int(frag_len_min), int(frag_len_max)
    a.profile1 = prof1.c_struct()
    a.ins_prob1, a.del_prob1 = float(ins_prob1), float(del_prob1)
    keep.append(prof1)
    if paired:
        a.profile2 = prof2.c_struct()
        keep.append(prof2)
    a.ins_prob2, a.del_prob2 = float(ins_prob2), float(del_prob2)
    bcs = [b.encode() for b in barcodes]
    arr = (C.c_char_p * len(bcs))(*bcs)
    keep.append(arr)
    a.barcodes = arr
    a.n_barcodes = len(bcs)
    if seed_fn is not None:
        # the form the Rcpp shim uses: a callback that fills the next 8 words (Rcpp::runif there)
        def _cb(_user, out8):
            try:
                w = seed_fn()
                for i in range(8):
                    out8[i] = int(w[i])
                return 0
            except Exception:
                return 1
        cb = _abi.SEED_FN(_cb)
        keep.append(cb)
        a.seeds.fn = cb
    else:
        words = np.ascontiguousarray(words, dtype=np.uint32)
        keep.append(words)
        a.seeds.words = words.ctypes.data_as(C.POINTER(C.c_uint32))
        a.seeds.n_words = words.size
    if abort_flag is not None:                # numpy int32 array of one element, polled between batches
        keep.append(abort_flag)
        a.abort_flag = abort_flag.ctypes.data_as(C.POINTER(C.c_int32))
    a.lane_begin, a.lane_end = int(lane_begin), int(lane_end)
    a.device = int(device)
    a.max_batch_bytes = int(max_batch_bytes)
    if seed_offset_words is not None:
        a.seed_offset_given, a.seed_offset_words = 1, int(seed_offset_words)
    if devices is not None:
        dv = np.ascontiguousarray(devices, dtype=np.int32)
        keep.append(dv)
        a.devices, a.n_devices = dv.ctypes.data_as(C.POINTER(C.c_int32)), dv.size
    a.stream_output = int(bool(stream_output))
    return a, keep


def illumina(obj, out_prefix, n_reads, read_length, paired, frag_mean=400, frag_sd=100, matepair=False,
             seq_sys=None, profile1=None, profile2=None, ins_prob1=0.00009, del_prob1=0.00011,
             ins_prob2=0.00015, del_prob2=0.00023, frag_len_min=None, frag_len_max=None, haplotype_probs=None,
             barcodes=None, prob_dup=0.02, sep_files=False, compress=False, comp_method="bgzip", n_threads=1,
             read_pool_size=1000, show_progress=False, overwrite=False,
             seed=None, seed_words=None, device=0, lane_begin=0, lane_end=0, max_batch_bytes=0, _session=False,
             seed_fn=None, abort_flag=None, seed_offset_words=None, devices=None, stream_output=False, _job=False):
    """Create and write Illumina reads (R/hts_illumina.R:593-732).

    With ``_session=True`` nothing is written: the opened `IlluminaSession` is returned instead
    (tests and bench.py use it to keep the FASTQ in HBM).
    """
    if matepair:
        paired = True
    check_illumina_args(obj, n_reads, read_length, paired, frag_mean, frag_sd, matepair, seq_sys, profile1, profile2,
                        ins_prob1, del_prob1, ins_prob2, del_prob2, frag_len_min, frag_len_max, haplotype_probs,
                        barcodes, prob_dup, sep_files, compress, comp_method, n_threads, read_pool_size, show_progress)
    out_prefix = os.path.expanduser(out_prefix) if out_prefix else out_prefix
    is_ref = isinstance(obj, RefGenome)
    if is_ref:
        sep_files = False
    ends = 2 if paired else 1
    if not _session and out_prefix:
        if not sep_files:
            fns = ["%s_R%d.fq" % (out_prefix, i + 1) for i in range(ends)]
        else:
            fns = ["%s_%s_R%d.fq" % (out_prefix, h, i + 1) for h in obj.hap_names() for i in range(ends)]
        _check_file_existence(fns, bool(compress), overwrite)
    if isinstance(compress, (bool, np.bool_)):
        compress = 6 if compress else 0
    if n_threads > 1 and compress > 0 and comp_method == "gzip":
        raise ValueError("\nCompression using gzip cannot be performed using multiple threads. "
                         "Please use bgzip compression instead.")
    frag_len_shape = (frag_mean / frag_sd) ** 2
    frag_len_scale = frag_sd ** 2 / frag_mean
    if frag_len_min is None:
        frag_len_min = read_length
    if frag_len_max is None or frag_len_max > 2 ** 32 - 1:
        frag_len_max = 2 ** 32 - 1
    if frag_len_min > frag_len_max:
        raise ValueError("\nFragment length min can't be less than the max. For computational reasons, both should "
                         "also be < 2^32, and if `frag_len_min` is not provided, it's automatically changed to the "
                         "read length.")
    if haplotype_probs is None and not is_ref:
        haplotype_probs = [1.0] * obj.n_haps()
    if barcodes is None:
        barcodes = [""] * (1 if is_ref else obj.n_haps())
    elif isinstance(barcodes, str):
        barcodes = [barcodes]
    prof1 = read_profile(profile1, seq_sys, read_length, 1)
    prof2 = read_profile(profile2, seq_sys, read_length, 2) if paired else None

    if seed_words is None and seed_fn is None:
        if seed is None:
            raise ValueError("give `seed` (SplitMix64 seed for the 32-bit sub-seed words), `seed_words` or `seed_fn`")
        budget = illumina_ref_seed_budget(n_threads) if is_ref else obj.seed_budget(n_threads)
        seed_words = _seed_words(seed, budget)

    args, keep = make_args(out_prefix, n_reads, paired, matepair, prof1, prof2, ins_prob1, del_prob1, ins_prob2,
                           del_prob2, frag_len_shape, frag_len_scale, frag_len_min, frag_len_max, barcodes, prob_dup,
                           n_threads, read_pool_size, seed_words, compress, comp_method, sep_files, haplotype_probs,
                           lane_begin, lane_end, device, max_batch_bytes, seed_fn, abort_flag, seed_offset_words, devices,
                           stream_output)
    L = _abi.lib()
    if _job:
        view, keep2 = obj._view()
        h = C.c_void_p()
        _abi.check((L.jk_illumina_ref_job if is_ref else L.jk_illumina_hap_job)(C.byref(view), C.byref(args), C.byref(h)))
        return Job(h, [keep, keep2, view, args])
    if is_ref:
        view, keep2 = obj._view()
        if _session:
            h = C.c_void_p()
            _abi.check(L.jk_illumina_ref_open(C.byref(view), C.byref(args), C.byref(h)))
            return IlluminaSession(h, [keep, keep2, view, args])
        _abi.check(L.jk_illumina_ref(C.byref(view), C.byref(args)))
    else:
        view, keep2 = obj._view()
        if _session:
            h = C.c_void_p()
            _abi.check(L.jk_illumina_hap_open(C.byref(view), C.byref(args), C.byref(h)))
            return IlluminaSession(h, [keep, keep2, view, args])
        _abi.check(L.jk_illumina_hap(C.byref(view), C.byref(args)))
    return None
