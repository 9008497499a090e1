"""ctypes binding of libjackalope_hip.so (include/jackalope_hip.h).

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is no
CPU fallback: if the library is missing or a HIP call fails, the error is raised to the caller.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JK_HIP_LIB") or os.path.join(_HERE, "csrc", "libjackalope_hip.so")   # JK_HIP_LIB: kernel-variant experiments

JK_OK = 0
JK_ERR_ARG, JK_ERR_UNSUPPORTED, JK_ERR_DEVICE, JK_ERR_IO, JK_ERR_SEEDS, JK_ERR_ABORTED = 1, 2, 3, 4, 5, 6

(OP_PCG_STREAM, OP_RUNIF_INDEX, OP_RUNIF_DOUBLE, OP_CANONICAL, OP_N_QUAL, OP_LT_HALF, OP_FRAG_START,
 OP_LOG, OP_SQRT, OP_GAMMA_STREAM, OP_EXP, OP_POW, OP_LOG10, OP_QNORM, OP_RUNIF_AB, OP_RUNIF_INDEX32, OP_ALIAS_INDEX32) = range(17)


class JackalopeHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("jackalope_hip error %d: %s" % (code, msg))
        self.code = code
        self.message = msg


class RefGenomeView(C.Structure):
    _fields_ = [("n_chroms", C.c_uint64),
                ("chrom_names", C.POINTER(C.c_char_p)),
                ("chrom_seqs", C.POINTER(C.c_void_p)),
                ("chrom_lens", C.POINTER(C.c_uint64)),
                ("name", C.c_char_p),
                ("seqs_on_device", C.c_int32)]


class HapSetView(C.Structure):
    _fields_ = [("n_haps", C.c_uint64), ("n_chroms", C.c_uint64),
                ("hap_names", C.POINTER(C.c_char_p)),
                ("ref", RefGenomeView),
                ("chrom_size", C.POINTER(C.c_uint64)),
                ("n_mut", C.POINTER(C.c_uint64)),
                ("old_pos", C.POINTER(C.c_uint64)),
                ("new_pos", C.POINTER(C.c_uint64)),
                ("nuc_off", C.POINTER(C.c_uint64)),
                ("nuc_blob", C.c_void_p)]


class IlluminaProfile(C.Structure):
    _fields_ = [("read_length", C.c_uint32),
                ("n_quals", C.POINTER(C.c_uint32)),
                ("probs", C.POINTER(C.c_double)),
                ("quals", C.POINTER(C.c_uint8))]


SEED_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32))


class SeedSource(C.Structure):
    _fields_ = [("words", C.POINTER(C.c_uint32)),
                ("n_words", C.c_uint64),
                ("fn", SEED_FN),
                ("user", C.c_void_p)]


class IlluminaArgs(C.Structure):
    _fields_ = [("paired", C.c_int32), ("matepair", C.c_int32),
                ("out_prefix", C.c_char_p),
                ("sep_files", C.c_int32), ("compress", C.c_int32),
                ("comp_method", C.c_char_p),
                ("n_reads", C.c_uint64), ("prob_dup", C.c_double),
                ("n_threads", C.c_uint64), ("show_progress", C.c_int32),
                ("read_pool_size", C.c_uint64),
                ("haplotype_probs", C.POINTER(C.c_double)),
                ("frag_len_shape", C.c_double), ("frag_len_scale", C.c_double),
                ("frag_len_min", C.c_uint64), ("frag_len_max", C.c_uint64),
                ("profile1", IlluminaProfile), ("ins_prob1", C.c_double), ("del_prob1", C.c_double),
                ("profile2", IlluminaProfile), ("ins_prob2", C.c_double), ("del_prob2", C.c_double),
                ("barcodes", C.POINTER(C.c_char_p)), ("n_barcodes", C.c_uint64),
                ("seeds", SeedSource),
                ("abort_flag", C.POINTER(C.c_int32)),
                ("lane_begin", C.c_uint64), ("lane_end", C.c_uint64),
                ("device", C.c_int32),
                ("max_batch_bytes", C.c_uint64),
                ("seed_offset_given", C.c_int32), ("seed_offset_words", C.c_uint64),
                ("devices", C.POINTER(C.c_int32)), ("n_devices", C.c_uint32),
                ("stream_output", C.c_int32)]


class PacbioArgs(C.Structure):
    _fields_ = [("out_prefix", C.c_char_p), ("sep_files", C.c_int32), ("compress", C.c_int32),
                ("comp_method", C.c_char_p),
                ("n_reads", C.c_uint64), ("n_threads", C.c_uint64), ("show_progress", C.c_int32),
                ("read_pool_size", C.c_uint64), ("prob_dup", C.c_double),
                ("scale", C.c_double), ("sigma", C.c_double), ("loc", C.c_double), ("min_read_len", C.c_double),
                ("read_probs", C.POINTER(C.c_double)), ("read_lens", C.POINTER(C.c_uint64)), ("n_read_lens", C.c_uint64),
                ("max_passes", C.c_uint64),
                ("chi2_params_n", C.POINTER(C.c_double)), ("chi2_params_s", C.POINTER(C.c_double)),
                ("sqrt_params", C.POINTER(C.c_double)), ("norm_params", C.POINTER(C.c_double)),
                ("prob_thresh", C.c_double), ("prob_ins", C.c_double), ("prob_del", C.c_double), ("prob_subst", C.c_double),
                ("haplotype_probs", C.POINTER(C.c_double)),
                ("seeds", SeedSource),
                ("abort_flag", C.POINTER(C.c_int32)),
                ("lane_begin", C.c_uint64), ("lane_end", C.c_uint64),
                ("device", C.c_int32),
                ("max_batch_bytes", C.c_uint64),
                ("seed_offset_given", C.c_int32), ("seed_offset_words", C.c_uint64),
                ("devices", C.POINTER(C.c_int32)), ("n_devices", C.c_uint32),
                ("stream_output", C.c_int32)]


# every symbol include/jackalope_hip.h declares
EXPORTS = [
    "jk_last_error", "jk_version", "jk_device_count", "jk_device_arena_trim", "jk_device_arena_stats", "jk_illumina_ref", "jk_illumina_hap", "jk_pacbio_ref", "jk_pacbio_hap",
    "jk_illumina_ref_job", "jk_illumina_hap_job", "jk_pacbio_ref_job", "jk_pacbio_hap_job", "jk_job_n_files", "jk_job_plan_next", "jk_job_run",
    "jk_job_progress", "jk_job_seed_words_used", "jk_job_free",
    "jk_illumina_ref_open", "jk_illumina_hap_open", "jk_pacbio_ref_open", "jk_pacbio_hap_open", "jk_session_generate", "jk_session_generate_async", "jk_session_wait", "jk_session_run", "jk_session_progress", "jk_session_sizes",
    "jk_session_device_ptr", "jk_session_fetch", "jk_session_fetch_range", "jk_session_write", "jk_session_write_shard",
    "jk_session_shard_seed_words", "jk_session_timing",
    "jk_session_seed_words_used", "jk_session_rare_branch_lanes", "jk_session_retries", "jk_session_batches", "jk_session_lane_bytes", "jk_session_close",
    "jk_split_int", "jk_reads_per_group", "jk_plan_lane_quotas", "jk_alias_build", "jk_hap_chrom_full",
    "jk_host_eval", "jk_dev_eval", "jk_eval_set_gamma", "jk_x87_one_minus",
    "jk_hap_builder_new", "jk_hap_builder_from", "jk_add_substitution", "jk_add_insertion", "jk_add_deletion",
    "jk_hap_builder_view", "jk_hap_builder_free",
    "jk_bgzf_bound", "jk_bgzf_deflate",
    "jk_pcg_advance_outputs", "jk_create_genome", "jk_read_fasta", "jk_genome_view", "jk_genome_fetch", "jk_genome_seed_words_used", "jk_genome_ms", "jk_genome_free",
]

_lib = None


def lib():
    """Load the HIP library (once).  Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise JackalopeHipError(JK_ERR_DEVICE, "%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    # One HIP runtime per process: PyTorch bundles its own libamdhip64.so (same SONAME as the system
    # one).  Whichever is mapped first serves both, so let torch map its copy before ours is resolved;
    # otherwise two runtimes end up loaded and the second one to initialise sees no device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.jk_last_error.restype = C.c_char_p
    L.jk_version.restype = C.c_char_p
    L.jk_session_seed_words_used.restype = C.c_uint64
    L.jk_session_seed_words_used.argtypes = [C.c_void_p]
    L.jk_session_retries.restype = C.c_uint32
    L.jk_session_retries.argtypes = [C.c_void_p]
    L.jk_session_batches.restype = C.c_uint32
    L.jk_session_batches.argtypes = [C.c_void_p]
    L.jk_illumina_ref.argtypes = [C.POINTER(RefGenomeView), C.POINTER(IlluminaArgs)]
    L.jk_illumina_hap.argtypes = [C.POINTER(HapSetView), C.POINTER(IlluminaArgs)]
    L.jk_illumina_ref_open.argtypes = [C.POINTER(RefGenomeView), C.POINTER(IlluminaArgs), C.POINTER(C.c_void_p)]
    L.jk_illumina_hap_open.argtypes = [C.POINTER(HapSetView), C.POINTER(IlluminaArgs), C.POINTER(C.c_void_p)]
    L.jk_pacbio_ref.argtypes = [C.POINTER(RefGenomeView), C.POINTER(PacbioArgs)]
    L.jk_pacbio_hap.argtypes = [C.POINTER(HapSetView), C.POINTER(PacbioArgs)]
    L.jk_pacbio_ref_open.argtypes = [C.POINTER(RefGenomeView), C.POINTER(PacbioArgs), C.POINTER(C.c_void_p)]
    L.jk_pacbio_hap_open.argtypes = [C.POINTER(HapSetView), C.POINTER(PacbioArgs), C.POINTER(C.c_void_p)]
    L.jk_illumina_ref_job.argtypes = [C.POINTER(RefGenomeView), C.POINTER(IlluminaArgs), C.POINTER(C.c_void_p)]
    L.jk_illumina_hap_job.argtypes = [C.POINTER(HapSetView), C.POINTER(IlluminaArgs), C.POINTER(C.c_void_p)]
    L.jk_pacbio_ref_job.argtypes = [C.POINTER(RefGenomeView), C.POINTER(PacbioArgs), C.POINTER(C.c_void_p)]
    L.jk_pacbio_hap_job.argtypes = [C.POINTER(HapSetView), C.POINTER(PacbioArgs), C.POINTER(C.c_void_p)]
    L.jk_job_n_files.restype = C.c_uint32
    L.jk_job_n_files.argtypes = [C.c_void_p]
    L.jk_job_plan_next.argtypes = [C.c_void_p]
    L.jk_job_run.argtypes = [C.c_void_p]
    L.jk_job_progress.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.jk_job_seed_words_used.restype = C.c_uint64
    L.jk_job_seed_words_used.argtypes = [C.c_void_p]
    L.jk_job_free.argtypes = [C.c_void_p]
    L.jk_job_free.restype = None
    L.jk_session_generate.argtypes = [C.c_void_p]
    L.jk_session_generate_async.argtypes = [C.c_void_p]
    L.jk_session_wait.argtypes = [C.c_void_p]
    L.jk_session_run.argtypes = [C.c_void_p]
    L.jk_session_progress.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.jk_session_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    L.jk_session_device_ptr.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.jk_session_fetch.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]
    L.jk_session_write.argtypes = [C.c_void_p]
    L.jk_session_fetch_range.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p]
    L.jk_session_write_shard.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.jk_session_shard_seed_words.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.jk_session_rare_branch_lanes.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p, C.c_uint32]
    L.jk_session_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.jk_session_lane_bytes.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]
    L.jk_session_close.argtypes = [C.c_void_p]
    L.jk_session_close.restype = None
    L.jk_split_int.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p]
    L.jk_split_int.restype = None
    L.jk_reads_per_group.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(SeedSource), C.c_void_p]
    L.jk_plan_lane_quotas.argtypes = [C.c_int32, C.c_uint32, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64,
                                      C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(SeedSource), C.c_int32, C.c_uint64,
                                      C.c_void_p, C.c_void_p, C.c_void_p]
    L.jk_alias_build.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.jk_alias_build.restype = None
    L.jk_hap_chrom_full.argtypes = [C.POINTER(HapSetView), C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]
    L.jk_hap_builder_new.argtypes = [C.POINTER(RefGenomeView), C.c_uint64, C.POINTER(C.c_void_p)]
    L.jk_hap_builder_from.argtypes = [C.POINTER(HapSetView), C.POINTER(C.c_void_p)]
    L.jk_add_substitution.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_char, C.c_uint64]
    L.jk_add_insertion.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_char_p, C.c_uint64]
    L.jk_add_deletion.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    L.jk_hap_builder_view.argtypes = [C.c_void_p, C.POINTER(HapSetView)]
    L.jk_hap_builder_free.argtypes = [C.c_void_p]
    L.jk_hap_builder_free.restype = None
    L.jk_create_genome.argtypes = [C.c_uint64, C.c_double, C.c_double, C.POINTER(C.c_double), C.c_uint64,
                                   C.POINTER(SeedSource), C.c_int, C.POINTER(C.c_void_p)]
    L.jk_pcg_advance_outputs.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    L.jk_pcg_advance_outputs.restype = None
    L.jk_read_fasta.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_uint64, C.c_int32, C.c_int32, C.c_int,
                                C.POINTER(C.c_void_p)]
    L.jk_genome_view.argtypes = [C.c_void_p, C.POINTER(RefGenomeView)]
    L.jk_genome_fetch.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    L.jk_genome_seed_words_used.restype = C.c_uint64
    L.jk_genome_seed_words_used.argtypes = [C.c_void_p]
    L.jk_genome_ms.restype = C.c_double
    L.jk_genome_ms.argtypes = [C.c_void_p]
    L.jk_genome_free.argtypes = [C.c_void_p]
    L.jk_genome_free.restype = None
    L.jk_bgzf_bound.restype = C.c_uint64
    L.jk_bgzf_bound.argtypes = [C.c_uint64]
    L.jk_bgzf_deflate.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                  C.POINTER(C.c_double)]
    L.jk_host_eval.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    L.jk_dev_eval.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    L.jk_eval_set_gamma.argtypes = [C.c_double, C.c_double]
    L.jk_eval_set_gamma.restype = None
    L.jk_x87_one_minus.argtypes = [C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
    L.jk_x87_one_minus.restype = None
    _lib = L
    return L


def check(rc):
    if rc != JK_OK:
        raise JackalopeHipError(rc, lib().jk_last_error().decode("utf-8", "replace"))
