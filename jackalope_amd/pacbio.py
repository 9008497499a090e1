"""`pacbio()` -- the R-level entry point of the PacBio path, mirrored in Python over the C ABI.

Argument names, defaults and checks follow /root/reference/R/hts_pacbio.R:232-348 (`pacbio`) and :8-122
(`check_pacbio_args`); the call it ends in is ``jk_pacbio_ref`` / ``jk_pacbio_hap`` instead of
``pacbio_ref_cpp`` / ``pacbio_hap_cpp``.
"""
import ctypes as C
import os

import numpy as np

from . import _abi
from .genome import RefGenome
from .illumina import IlluminaSession, _check_file_existence, _is_num
from .rng import seed_words as _seed_words


def _err(arg, what):
    raise ValueError("\nFor the function `pacbio`, argument `%s` must be %s." % (arg, what))


def check_pacbio_args(obj, n_reads, haplotype_probs, sep_files, compress, comp_method, n_threads, read_pool_size,
                      chi2_params_s, chi2_params_n, max_passes, sqrt_params, norm_params, prob_thresh, ins_prob,
                      del_prob, sub_prob, min_read_length, lognorm_read_length, custom_read_lengths, prob_dup,
                      show_progress):
    """R/hts_pacbio.R:8-122 (the checks that do not depend on R object classes)."""
    if not isinstance(obj, RefGenome) and not hasattr(obj, "n_haps"):
        _err("obj", 'a "ref_genome" or "haplotypes" object')
    for nm, v in (("n_reads", n_reads), ("n_threads", n_threads), ("read_pool_size", read_pool_size),
                  ("max_passes", max_passes), ("min_read_length", min_read_length)):
        if not _is_num(v, 1) or int(v) != v:
            _err(nm, "a single integer >= 1")
    for nm, v, n in (("chi2_params_s", chi2_params_s, 5), ("chi2_params_n", chi2_params_n, 3),
                     ("sqrt_params", sqrt_params, 2), ("norm_params", norm_params, 2),
                     ("lognorm_read_length", lognorm_read_length, 3)):
        if len(v) != n or not all(_is_num(x) for x in v):
            _err(nm, "a numeric vector of length %d" % n)
    for nm, v in (("prob_thresh", prob_thresh), ("ins_prob", ins_prob), ("del_prob", del_prob), ("sub_prob", sub_prob),
                  ("prob_dup", prob_dup)):
        if not _is_num(v, 0, 1):
            _err(nm, "a single number in range [0,1]")
    if ins_prob + del_prob + sub_prob > 1:
        raise ValueError("\nFor the function `pacbio`, the sum of `ins_prob`, `del_prob`, and `sub_prob` must be <= 1.")
    for nm, v in (("sep_files", sep_files), ("show_progress", show_progress)):
        if not isinstance(v, (bool, np.bool_)):
            _err(nm, "a single logical")
    if comp_method not in ("gzip", "bgzip", "bgzip-host"):     # "bgzip-host": this library's zlib-on-the-host variant
        _err("comp_method", '"gzip" or "bgzip"')
    if custom_read_lengths is not None:
        crl = np.asarray(custom_read_lengths, dtype=np.float64)
        if crl.ndim not in (1, 2) or (crl.ndim == 2 and crl.shape[1] != 2) or crl.size == 0:
            _err("custom_read_lengths", "NULL, a numeric vector, or a 2-column numeric matrix")
        lens = crl[:, 0] if crl.ndim == 2 else crl
        if (lens < 1).any():
            _err("custom_read_lengths", "read lengths >= 1")


def pacbio(obj, out_prefix, n_reads,
           chi2_params_s=(0.01214, -5.12, 675, 48303.0732881, 1.4691051212330266),
           chi2_params_n=(0.00189237136, 2.53944970, 5500), max_passes=40, sqrt_params=(0.5, 0.2247),
           norm_params=(0, 0.2), prob_thresh=0.2, ins_prob=0.11, del_prob=0.04, sub_prob=0.01, min_read_length=50,
           lognorm_read_length=(0.200110276521, -10075.4363813, 17922.611306), custom_read_lengths=None,
           prob_dup=0.0, haplotype_probs=None, sep_files=False, compress=False, comp_method="bgzip", n_threads=1,
           read_pool_size=100, show_progress=False, overwrite=False,
           seed=None, seed_words=None, device=0, lane_begin=0, lane_end=0, max_batch_bytes=0, _session=False,
           seed_offset_words=None, devices=None, stream_output=False, _job=False, abort_flag=None):
    """Create and write PacBio reads (R/hts_pacbio.R:232-348).  ``_session=True`` returns the opened
    session (FASTQ stays in HBM) instead of writing ``<out_prefix>_R1.fq``."""
    check_pacbio_args(obj, n_reads, haplotype_probs, sep_files, compress, comp_method, n_threads, read_pool_size,
                      chi2_params_s, chi2_params_n, max_passes, sqrt_params, norm_params, prob_thresh, ins_prob, del_prob,
                      sub_prob, min_read_length, lognorm_read_length, custom_read_lengths, prob_dup, show_progress)
    out_prefix = os.path.expanduser(out_prefix) if out_prefix else out_prefix
    is_ref = isinstance(obj, RefGenome)
    if is_ref:
        sep_files = False
    if not _session and out_prefix:
        fns = ["%s_R1.fq" % out_prefix] if not sep_files else ["%s_%s_R1.fq" % (out_prefix, h) for h in obj.hap_names()]
        _check_file_existence(fns, bool(compress), overwrite)
    if isinstance(compress, (bool, np.bool_)):
        compress = 6 if compress else 0
    if n_threads > 1 and compress > 0 and comp_method == "gzip":
        raise ValueError("\nCompression using gzip cannot be performed using multiple threads. "
                         "Please use bgzip compression instead.")
    keep = []
    a = _abi.PacbioArgs()
    a.out_prefix = (out_prefix or "").encode()
    a.sep_files, a.compress, a.comp_method = int(bool(sep_files)), int(compress), comp_method.encode()
    a.n_reads, a.n_threads, a.show_progress, a.read_pool_size = int(n_reads), int(n_threads), 0, int(read_pool_size)
    a.prob_dup = float(prob_dup)
    a.sigma, a.loc, a.scale = float(lognorm_read_length[0]), float(lognorm_read_length[1]), float(lognorm_read_length[2])
    a.min_read_len = float(min_read_length)
    if custom_read_lengths is not None:
        crl = np.asarray(custom_read_lengths, dtype=np.float64)
        lens, probs = (crl[:, 0], crl[:, 1]) if crl.ndim == 2 else (crl, np.ones(crl.size))
        lens = np.ascontiguousarray(lens, dtype=np.uint64)
        probs = np.ascontiguousarray(probs, dtype=np.float64)
        keep += [lens, probs]
        a.read_lens = lens.ctypes.data_as(C.POINTER(C.c_uint64))
        a.read_probs = probs.ctypes.data_as(C.POINTER(C.c_double))
        a.n_read_lens = lens.size
    a.max_passes = int(max_passes)
    for name, val in (("chi2_params_n", chi2_params_n), ("chi2_params_s", chi2_params_s), ("sqrt_params", sqrt_params),
                      ("norm_params", norm_params)):
        arr = np.ascontiguousarray(val, dtype=np.float64)
        keep.append(arr)
        setattr(a, name, arr.ctypes.data_as(C.POINTER(C.c_double)))
    a.prob_thresh, a.prob_ins, a.prob_del, a.prob_subst = float(prob_thresh), float(ins_prob), float(del_prob), float(sub_prob)
    if haplotype_probs is None and not is_ref:
        haplotype_probs = [1.0] * obj.n_haps()
    if haplotype_probs is not None:
        hp = np.ascontiguousarray(haplotype_probs, dtype=np.float64)
        keep.append(hp)
        a.haplotype_probs = hp.ctypes.data_as(C.POINTER(C.c_double))
    if seed_words is None:
        if seed is None:
            raise ValueError("give `seed` (SplitMix64 seed for the 32-bit sub-seed words) or `seed_words`")
        seed_words = _seed_words(seed, 16 * int(n_threads) if is_ref else obj.seed_budget(n_threads))
    words = np.ascontiguousarray(seed_words, dtype=np.uint32)
    keep.append(words)
    a.seeds.words = words.ctypes.data_as(C.POINTER(C.c_uint32))
    a.seeds.n_words = words.size
    a.lane_begin, a.lane_end, a.device, a.max_batch_bytes = int(lane_begin), int(lane_end), int(device), int(max_batch_bytes)
    if seed_offset_words is not None:
        a.seed_offset_given, a.seed_offset_words = 1, int(seed_offset_words)
    if devices is not None:
        dv = np.ascontiguousarray(devices, dtype=np.int32)
        keep.append(dv)
        a.devices, a.n_devices = dv.ctypes.data_as(C.POINTER(C.c_int32)), dv.size
    a.stream_output = int(bool(stream_output))
    if abort_flag is not None:
        keep.append(abort_flag)
        a.abort_flag = abort_flag.ctypes.data_as(C.POINTER(C.c_int32))
    L = _abi.lib()
    view, keep2 = obj._view()
    if _job:
        from .illumina import Job
        h = C.c_void_p()
        _abi.check((L.jk_pacbio_ref_job if is_ref else L.jk_pacbio_hap_job)(C.byref(view), C.byref(a), C.byref(h)))
        return Job(h, [keep, keep2, view, a])
    if _session:
        h = C.c_void_p()
        fn = L.jk_pacbio_ref_open if is_ref else L.jk_pacbio_hap_open
        _abi.check(fn(C.byref(view), C.byref(a), C.byref(h)))
        return IlluminaSession(h, [keep, keep2, view, a])
    _abi.check((L.jk_pacbio_ref if is_ref else L.jk_pacbio_hap)(C.byref(view), C.byref(a)))
    return None
