/* abi_driver.c -- the C ABI of include/jackalope_hip.h exercised from plain C99 (no ctypes, no C++): fills
 * jk_ref_genome / jk_illumina_args by hand from a job file written by tests/test_gpu_boundary.py, hands the seed
 * words over through the CALLBACK form of jk_seed_source, and calls jk_illumina_ref (mode "oneshot") or the job API
 * (mode "job": jk_illumina_ref_job, jk_job_plan_next, jk_job_run, jk_job_progress).  The test compares the FASTQ
 * files with the oracle.  Build: gcc -std=c99 -Wall -Wextra -pedantic -Iinclude tests/abi_driver.c -ljackalope_hip
 *
 *   abi_driver <job file> <out_prefix> oneshot|job
 * prints: "seed_words <n>" and, in job mode, "progress <done> <total>". */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jackalope_hip.h"

typedef struct { const uint32_t* words; uint64_t n, pos; } seed_state;

static int next_seed_words(void* user, uint32_t* out8) {
    seed_state* st = (seed_state*)user;
    int i;
    if (st->pos + 8 > st->n) return 1;
    for (i = 0; i < 8; i++) out8[i] = st->words[st->pos + (uint64_t)i];
    st->pos += 8;
    return 0;
}

static void* take(FILE* f, size_t n) {
    void* p = malloc(n ? n : 1);
    if (!p || fread(p, 1, n, f) != n) { fprintf(stderr, "short job file\n"); exit(2); }
    return p;
}
static uint64_t u64(FILE* f) { uint64_t v; if (fread(&v, 8, 1, f) != 1) { fprintf(stderr, "short job file\n"); exit(2); } return v; }
static uint32_t u32(FILE* f) { uint32_t v; if (fread(&v, 4, 1, f) != 1) { fprintf(stderr, "short job file\n"); exit(2); } return v; }
static double f64(FILE* f) { double v; if (fread(&v, 8, 1, f) != 1) { fprintf(stderr, "short job file\n"); exit(2); } return v; }
static char* str(FILE* f) {
    uint64_t n = u64(f);
    char* s = (char*)malloc((size_t)n + 1);
    if (!s || fread(s, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short job file\n"); exit(2); }
    s[n] = 0;
    return s;
}

static void read_profile(FILE* f, uint32_t L, jk_illumina_profile* p, double* ins, double* del) {
    uint64_t total;
    p->read_length = L;
    p->n_quals = (const uint32_t*)take(f, (size_t)4 * L * 4);
    total = u64(f);
    p->probs = (const double*)take(f, (size_t)total * 8);
    p->quals = (const uint8_t*)take(f, (size_t)total);
    *ins = f64(f); *del = f64(f);
}

int main(int argc, char** argv) {
    FILE* f;
    char magic[8];
    jk_ref_genome g;
    jk_illumina_args a;
    seed_state st;
    uint64_t i, n_chroms;
    const char** names; const char** seqs; uint64_t* lens;
    const char* barcode;
    uint32_t L;
    int rc;
    if (argc != 4) { fprintf(stderr, "usage: abi_driver <job file> <out_prefix> oneshot|job\n"); return 2; }
    f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "JKJOB1\0\0", 8) != 0) { fprintf(stderr, "not a job file\n"); return 2; }

    memset(&g, 0, sizeof g);
    memset(&a, 0, sizeof a);
    n_chroms = u64(f);
    names = (const char**)malloc((size_t)n_chroms * sizeof *names);
    seqs = (const char**)malloc((size_t)n_chroms * sizeof *seqs);
    lens = (uint64_t*)malloc((size_t)n_chroms * sizeof *lens);
    if (!names || !seqs || !lens) return 2;
    for (i = 0; i < n_chroms; i++) {
        names[i] = str(f);
        lens[i] = u64(f);
        seqs[i] = (const char*)take(f, (size_t)lens[i]);
    }
    g.n_chroms = n_chroms; g.chrom_names = names; g.chrom_seqs = seqs; g.chrom_lens = lens;
    g.name = NULL;                        /* NULL = "REF" */
    g.seqs_on_device = 0;

    a.paired = (int32_t)u32(f); a.matepair = (int32_t)u32(f);
    a.n_reads = u64(f); a.prob_dup = f64(f); a.n_threads = u64(f); a.read_pool_size = u64(f);
    a.frag_len_shape = f64(f); a.frag_len_scale = f64(f); a.frag_len_min = u64(f); a.frag_len_max = u64(f);
    L = u32(f);
    read_profile(f, L, &a.profile1, &a.ins_prob1, &a.del_prob1);
    if (a.paired) read_profile(f, L, &a.profile2, &a.ins_prob2, &a.del_prob2);
    barcode = str(f);
    a.barcodes = &barcode; a.n_barcodes = 1;
    st.n = u64(f);
    st.words = (const uint32_t*)take(f, (size_t)st.n * 4);
    st.pos = 0;
    fclose(f);

    a.out_prefix = argv[2];
    a.compress = 0; a.comp_method = "bgzip";
    a.seeds.words = NULL; a.seeds.n_words = 0;
    a.seeds.fn = next_seed_words; a.seeds.user = &st;
    a.device = 0;

    if (strcmp(argv[3], "oneshot") == 0) {
        rc = jk_illumina_ref(&g, &a);
        if (rc != JK_OK) { fprintf(stderr, "jk_illumina_ref: %d %s\n", rc, jk_last_error()); return 1; }
    } else {
        jk_job* job = NULL;
        uint64_t done = 0, total = 0;
        rc = jk_illumina_ref_job(&g, &a, &job);
        if (rc == JK_OK && jk_job_n_files(job) != 1) { fprintf(stderr, "expected one file set\n"); return 1; }
        if (rc == JK_OK) rc = jk_job_plan_next(job);
        if (rc == JK_OK) rc = jk_job_run(job);
        if (rc == JK_OK) rc = jk_job_progress(job, &done, &total);
        if (rc != JK_OK) { fprintf(stderr, "job: %d %s\n", rc, jk_last_error()); return 1; }
        printf("progress %llu %llu\n", (unsigned long long)done, (unsigned long long)total);
        jk_job_free(job);
    }
    printf("seed_words %llu\n", (unsigned long long)st.pos);
    return 0;
}
