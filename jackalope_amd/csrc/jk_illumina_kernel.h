// jk_illumina_kernel.h -- the Illumina per-read inner loop as a CDNA4 (gfx950) HIP kernel.
//
// One GPU thread = one "lane" = one OpenMP thread of the reference's driver
// (write_reads_one_filetype_, /root/reference/src/hts.h:357-417): its own pcg64, its own read
// quota, its own per-chromosome quotas, its own gamma state.  A lane runs the reference's
// create_reads loop (src/hts.h:254-280) to completion and appends FASTQ text to its private "pool"
// region in HBM (the analogue of ReadWriterOneThread::fastq_pools); a scan + compaction pass then
// lays the pools out lane-major, which is the file order.
//
// What stays converged: every lane of a wave is in the same phase (fragment draw -> indel draws for
// R1,R2 -> strand -> per-base quality/mismatch draws -> duplicate draw) at the same time; the rare
// data-dependent extras (gamma rejections, indel events, mismatches, 'N' bases) are short divergent
// branches.  The kernel is bound by the 128-bit pcg64 multiply (about 1200 draws per read pair),
// not by HBM: per pair it reads 300 reference bytes and writes ~660 FASTQ bytes.
//
// Pool layout (chosen from the first rocprof PMC pass, profiles/r01_v1_*: a pool region per lane
// with 4-byte stores cost 8x write amplification and byte-wise reference loads missed L2 for most
// bases): lanes are grouped in tiles of 64 (= one wave) and a tile's pools are word-interleaved,
// word k of lane l at tile_base + (k*64 + l)*4, so that the 64 lanes of a wave, which emit their
// k-th word at (nearly) the same time, store one contiguous 256-byte row.  The reference is read in
// aligned 8-byte pieces kept in a register pair.
//
// Memory plan per workgroup: the ART quality tables (alias thresholds as exact u64 cut points on
// the raw pcg output, qualities, mismatch cut points) are staged once into LDS (97.5 KB for the
// HiSeq 2500 / 150 bp pair of profiles -> one 1024-thread workgroup per CU, 4 waves per SIMD);
// profiles that do not fit in LDS are read through L2 instead (template parameter).
#pragma once
#include <type_traits>
#include "jk_math.h"

namespace jk {

constexpr int JK_MAX_BARCODE = 992;      // a barcode is shorter than the read, and reads are at most 992 long (JK_MAX_EVW_LONG)
constexpr int JK_MAX_EVW = 16;         // 64-bit words of indel-event bitmaps per read end (positions < 1024) -- kernels with the tables in LDS
constexpr int JK_MAX_EVW_LONG = 32;    // -- kernels with the tables in global memory (64-bit word masks): read lengths up to 992
constexpr uint32_t JK_HAP_BUCKET_SHIFT = 10;   // 1024 haplotype positions per bucket of the mutation index
constexpr uint32_t JK_HAP_SEGS = 3;    // segments of a read window kept in the per-lane LDS table (haplotype runs)

constexpr uint32_t JK_RARE_LOG_CAP = 61;      // lanes IlluminaKernelParams::rare_log holds
// error bits reported through IlluminaKernelParams::err
enum : uint32_t {
    JK_KERR_IMAGE_FULL = 64u,          // the compacted FASTQ image would not fit the buffer allocated for it (PacBio: see plan_pools_common)
    JK_KERR_TOO_MANY_DELETIONS = 1u,   // a read end needed more source positions than the event bitmaps hold
    JK_KERR_POOL_OVERFLOW = 2u,        // internal: a lane wrote past its pool region
    JK_KERR_EMPTY_CHROM = 256u,        // a lane holds reads for an empty chromosome (reads_per_group's last group takes the remainder whatever its probability)
    JK_KERR_GAMMA_MATH = 128u,         // gamma shape < 1: pow(u, 1/shape) left the transcribed main path of glibc's pow (|y log u| >= 512)
};

struct GenomeDev {
    const uint8_t* seq;          // all chromosomes, 1 byte per base, 64-B padded, ENCODED on upload:
                                 // T=0 C=1 A=2 G=3 (jlp::bases order), anything else keeps its ASCII value
    const uint64_t* chrom_off;   // [n_chroms] byte offset of chromosome in seq
    const uint64_t* chrom_len;   // [n_chroms]
    const uint8_t* hdr_blob;     // "@<genome>-<chrom>-" per chromosome
    const uint32_t* hdr_off;     // [n_chroms + 1]
    uint32_t n_chroms;
    // Illumina, optional (null: bytes only): the same bases at 2 bits each -- base i in bits 2*(i & 3) of packed[i >> 2],
    // code as in seq -- and one flag bit per 4096 bases, bit (i >> 12) & 7 of nflags[i >> 15], set when a chromosome
    // holds a byte that is not T, C, A or G there (pack_reference_kernel).  A read end whose source window touches no
    // flagged block takes all its bases from `packed`; any other reads `seq`.
    const uint8_t* packed;
    const uint8_t* nflags;
};

// Haplotypes: the mutation tables of all (haplotype, chromosome) cells, cell = hap * n_chroms + chrom
// (read side of AllMutations/HapChrom, /root/reference/src/hap_classes.h:100-258,314-333,439-455).
// For mutation m (absolute index) covering haplotype positions [new_pos[m], new_pos[m+1]):
//   pos - new_pos[m] <  nuc_len[m]  ->  seq[nuc_off[m] + pos - new_pos[m]]     (substituted / inserted bytes)
//   otherwise                       ->  reference chromosome at pos + ref_shift[m]
// which is get_char_ with size_modifier folded in on the host (nuc_len = max(size_mod + 1, 0),
// ref_shift = old_pos - size_mod - new_pos).  Positions before a cell's first mutation read the
// reference unshifted.  The nucleotide bytes live in the same (encoded) buffer as the genome.
struct HapMut {                    // one mutation, 32 bytes: everything a lookup needs in one cache-line access
    uint64_t new_pos;              // first haplotype position it covers
    int64_t ref_shift;             // reference position = haplotype position + ref_shift behind its nucleotides
    uint64_t nuc_off;              // offset of its nucleotides in GenomeDev::seq
    uint32_t nuc_len;              // max(size_modifier + 1, 0)
    uint32_t pad;
};
struct HapDev {
    const uint64_t* cell_mut_off;  // [n_cells + 1]
    const HapMut* mut;             // [n_mut] (the separate arrays of round 1 cost four dependent cache misses per lookup; the
                                   //  mutation tables of BASELINE configs[2..3] are 0.5-1 GB, far beyond any cache)
    const uint64_t* cell_size;     // [n_cells] haplotype chromosome sizes
    // coarse index over new_pos: for cell c and bucket j = hpos >> JK_HAP_BUCKET_SHIFT,
    // bucket[bucket_off[c] + j] = number of the cell's mutations with new_pos < (j << JK_HAP_BUCKET_SHIFT)
    const uint64_t* bucket_off;    // [n_cells + 1]
    const uint32_t* bucket;
    const uint8_t* bc_blob;        // [n_haps][JK_MAX_BARCODE] encoded barcodes
    const uint32_t* bc_len;        // [n_haps]
    uint32_t n_haps;
};

struct IlluminaKernelParams {
    GenomeDev g;
    HapDev h;                    // used by the HAP kernels only; then g.hdr_off is indexed by cell and
                                 // chrom_reads holds n_reads_vc: [cell * chrom_stride + lane]
    // lanes of this launch
    uint32_t n_lanes;
    const uint32_t* seeds;       // [n_lanes * 8] sub-seed words
    const uint64_t* lane_reads;  // [n_lanes] read quota (all ends)
    const uint32_t* chrom_reads; // per-chromosome read quotas, chromosome-major: [ci * chrom_stride + lane]
    uint32_t chrom_stride;
    const uint64_t* pool_off;    // [n_tiles + 1] byte offset of each 64-lane tile's pools (same for every end)
    uint8_t* pool[2];            // pool buffers, one per read end
    uint64_t* lane_bytes[2];     // out: bytes written per lane and end
    uint64_t* lane_made;         // out: reads made per lane
    uint64_t* evw;               // scratch: [n_ends][4 planes][ev_words][n_lanes] indel bitmaps
    uint32_t* err;
    // model
    uint32_t read_len, n_ends, paired, matepair;
    uint32_t ev_words;
    uint64_t frag_min, frag_max;
    jk_gamma_param gp;
    uint64_t th_match[2], th_del[2];   // draw x: x >= th_match -> match; else x >= th_del -> deletion; else insertion
    uint32_t never_match[2], never_del[2];
    uint64_t th_dup; uint32_t dup_all;
    uint64_t pool_size;
    uint32_t bc_len; uint8_t barcode[JK_MAX_BARCODE];      // encoded like the genome
    // tables, see IlluminaPacked in jk_host.h: mm2 [256] u64 by quality character (always staged to LDS);
    // tab = info2 [end][pos][nt] {byte offset of the first alias entry in tab, n entries} followed by the alias
    // entries {cut point's high word, 8*char kept | 8*char of the alias << 16} (staged to LDS when LDS_TAB); tab_lo: the cut
    // points' low words, one per entry, read when a draw's high word equals its entry's (global memory)
    const uint32_t* tab; const uint64_t* mm2; const uint32_t* tab_lo;
    // diagnostic: every time a cut point's low word decides (2^-32 per alias draw) the lane is noted -- [0] count, [1..JK_RARE_LOG_CAP]
    // lanes of the session's shard -- so that a test can compare exactly those lanes with the oracle
    uint32_t* rare_log; uint32_t rare_lane0;
    uint32_t n_info, n_entries;
    uint32_t lds_seg_off;                                  // HAP: byte offset of the per-lane segment table in LDS
    uint32_t lds_lut_off;                                  // packed reference: byte offset of the 512-entry expansion table in LDS
    uint32_t info_in_lds;                                  // tables in global memory: their {offset, count} part is staged at the start of LDS
    uint32_t lds_cell_off;                                 // byte offset of the per-lane chromosome cache (8 x BLOCK words: tag, offset, id-line prefix), 0xffffffff: none
};

// ---------------------------------------------------------------------------------------------
// Byte appender into a lane's pool (one column of its tile).  Bytes are gathered in a 64-bit
// shift register and leave as whole 32-bit words.  In the per-base loop every lane appends one byte
// per iteration, so after every 4th iteration each lane has a full word whatever its phase: the
// store is then issued wave-uniformly (no exec masking, one store instruction per 4 bases instead
// of one per base; first PMC pass of v3: SQ_INSTS_VMEM_WR was 600 per pair for 165 words).
// A stream that starts mid-word stores that first word with zeros in the bytes below its start:
// those bytes belong to the stream that ends there, which writes them LATER in program order,
// byte by byte (os_flush), so nothing is lost.
// ---------------------------------------------------------------------------------------------
struct OutStream {
    uint32_t acc;      // pending bytes (fewer than 4: pos & 3 of them), oldest in the low byte
    uint32_t pos;      // byte offset in the lane's stream of the next byte to append
};

constexpr uint32_t TILE_ROW = 64 * 4;      // bytes between consecutive words of one lane

// `col` = the lane's column in its tile: tile base (wave-uniform, so it lives in scalar registers) + 4 * (lane & 63);
// the word that holds stream byte `pos` is at col + (pos >> 2) * TILE_ROW
__device__ __forceinline__ uint8_t* os_word(uint8_t* col, uint32_t pos) { return col + (size_t)(pos >> 2) * TILE_ROW; }
__device__ __forceinline__ void os_begin(OutStream& s, uint32_t pos) { s.pos = pos; s.acc = 0; }
// 1..4 bytes at once (low byte first; the bytes of `word` above the n-th must be zero)
// A word of FASTQ into the pool.  (-DJK_NT_STORES: as a non-temporal store -- the pools are written once and read back by the
// compaction long after they have left L2, where they only push out the reference lines the generator re-reads.)
__device__ __forceinline__ void pool_store(uint8_t* p, uint32_t w) {
#ifdef JK_NT_STORES
    __builtin_nontemporal_store(w, reinterpret_cast<uint32_t*>(p));
#else
    *reinterpret_cast<uint32_t*>(p) = w;
#endif
}
__device__ __forceinline__ void os_put_n(OutStream& s, uint8_t* col, uint32_t word, uint32_t n) {
    const uint32_t cnt = s.pos & 3u;
    const uint64_t t = (uint64_t)word << (8u * cnt);
    const uint32_t w = s.acc | (uint32_t)t;
    if (cnt + n >= 4u) { pool_store(os_word(col, s.pos), w); s.acc = (uint32_t)(t >> 32); }
    else s.acc = w;
    s.pos += n;
}
__device__ __forceinline__ void os_put(OutStream& s, uint8_t* col, uint32_t byte) { os_put_n(s, col, byte, 1u); }
// write the pending bytes of the current word one by one; the stream is abandoned afterwards
__device__ __forceinline__ void os_flush(OutStream& s, uint8_t* col) {
    uint8_t* wp = os_word(col, s.pos);
    const uint32_t cnt = s.pos & 3u;
    for (uint32_t j = 0; j < cnt; j++) wp[j] = (uint8_t)(s.acc >> (8u * j));
    s.acc = 0;
}

// Global-memory byte pointer that stays one through an asm barrier (a pointer that went through `asm("" : "+s"(p))` is a
// generic one to the compiler otherwise: flat_load, which also counts as an LDS access, so every wait for an LDS read
// would wait for the chunk prefetch as well), and unaligned 4- and 8-byte loads from it.
typedef const __attribute__((address_space(1))) uint8_t* gbytes_t;
typedef uint32_t __attribute__((aligned(1))) u32_unaligned_t;
typedef uint64_t __attribute__((aligned(1))) u64_unaligned_t;
__device__ __forceinline__ uint32_t gload8(gbytes_t p) { return *p; }
__device__ __forceinline__ uint32_t gload32(gbytes_t p) { return *reinterpret_cast<const __attribute__((address_space(1))) u32_unaligned_t*>(p); }
__device__ __forceinline__ void gload64(gbytes_t p, uint32_t* v) {
    const uint64_t x = *reinterpret_cast<const __attribute__((address_space(1))) u64_unaligned_t*>(p);
    v[0] = (uint32_t)x; v[1] = (uint32_t)(x >> 32);
}

struct LaneRng {
    jk_pcg64d e;
    __device__ __forceinline__ uint64_t operator()() { return jk_pcg_next(e); }
};

// (uint64)(runif_01 * n) for n < 2^32: same value as jk_runif_index, two 32x32 multiply-adds.
__device__ __forceinline__ uint32_t runif_index32(uint64_t x, uint32_t n) {
    const uint64_t t0 = (uint64_t)(uint32_t)x * n + n;                  // < 2^64
    const uint64_t t1 = (uint64_t)(uint32_t)(x >> 32) * n + (t0 >> 32);  // (x+1)*n = t1 * 2^32 + lo32(t0)
    uint32_t hi = (uint32_t)(t1 >> 32);
    // x87 rounding of the product can only carry into `hi` when the low 64 bits are within 2^31 of 2^64
    // (probability 2^-32 per draw): one compare and a wave-uniform branch on the common path
    if (__builtin_amdgcn_ballot_w64((uint32_t)t1 == 0xffffffffu) != 0) {
        asm volatile("" ::: "memory");       // keeps the block from being if-converted into the common path
        if ((uint32_t)t1 == 0xffffffffu && hi != 0) {
            const uint64_t lo = (t1 << 32) | (uint32_t)t0;
            const uint64_t half = 1ULL << (31 - __builtin_clz(hi));
            if (lo + half < lo) hi++;
        }
    }
    return hi;
}

// (uint64)(runif_01 * nq) for an alias table of nq <= 255 entries (src/alias_sampler.h:55), as the quality step uses it:
// hi32(xh*nq + B) with B = hi32((xl+1)*nq) <= nq; B can only matter when the low word of xh*nq is within 256 of
// wrapping (2^-24 per draw), and then the 96-bit routine decides.
__device__ __forceinline__ uint32_t alias_index32(uint64_t x, uint32_t nq) {
    const uint32_t xh = (uint32_t)(x >> 32);
    uint32_t idx = __umulhi(xh, nq);
    const uint32_t prl = xh * nq;
    if (__builtin_amdgcn_ballot_w64(prl > 0xfffffeffu) != 0) {
        asm volatile("" ::: "memory");
        if (prl > 0xfffffeffu) idx = runif_index32(x, nq);
    }
    return idx;
}

// jk_n_qual for the device: (x+1)*10 = b * 2^32 + lo32(a); both x87 roundings (product, then + '!') move the value by
// less than 2^-58, so they can only matter when the fraction is within that of 1, i.e. when lo32(b) is all ones
// (2^-32 per draw): the common path is 33 + hi32(b), the exact routine sits behind a wave-uniform branch.
__device__ __forceinline__ uint32_t n_qual32(uint64_t x) {
    const uint64_t a = (uint64_t)(uint32_t)x * 10u + 10u;
    const uint64_t b = (uint64_t)(uint32_t)(x >> 32) * 10u + (a >> 32);
    uint32_t q = 33u + (uint32_t)(b >> 32);
    if (__builtin_amdgcn_ballot_w64((uint32_t)b == 0xffffffffu) != 0) {
        asm volatile("" ::: "memory");
        if ((uint32_t)b == 0xffffffffu) q = jk_n_qual(x);
    }
    return q;
}

// code (T0 C1 A2 G3) -> ASCII
__device__ __forceinline__ uint32_t base_char(uint32_t code) {        // code 0..3
    // byte `code` of "TCAG" by v_perm_b32 (selector bytes 1..3 = 0x0c: constant zero): two instructions instead of
    // shift, shift, mask
    return __builtin_amdgcn_perm(0u, 0x47414354u, code | 0x0c0c0c00u);
}

// the table blob either in LDS or global (byte pointer; layout in IlluminaKernelParams)
struct TabPtrs {
    const uint8_t* tab;
};

#ifdef JK_TIMELINE
// experiment builds only (tools/timeline.sh): per wave {100 MHz wall clock at start, at end, HW_ID, XCC_ID}
constexpr uint32_t JK_TIMELINE_WAVES = 1u << 15;
__device__ uint64_t g_timeline[4 * JK_TIMELINE_WAVES];
#endif

struct HapSeg { uint64_t addr, begin, end; };   // haplotype positions [begin, end) lie contiguously at seq[addr + (pos - hpos)]

// Which contiguous piece of the (encoded) buffer serves haplotype position hpos of `cell`?  `m` is the
// cell-relative index of the last mutation with new_pos <= hpos (-1: none) and is updated in place.
template <typename MIdx>
__device__ __forceinline__ HapSeg hap_resolve(const HapDev& H, uint64_t chrom_off, uint32_t cell, MIdx& m, uint64_t hpos) {
    const uint64_t mo = H.cell_mut_off[cell];
    const int64_t n = (int64_t)(H.cell_mut_off[cell + 1] - mo);
    const HapMut* M = H.mut + mo;
    while (m + 1 < n && M[m + 1].new_pos <= hpos) m++;
    while (m >= 0 && M[m].new_pos > hpos) m--;
    HapSeg sg;
    if (m < 0) {
        sg.begin = 0; sg.end = n > 0 ? M[0].new_pos : H.cell_size[cell];
        sg.addr = chrom_off + hpos;
    } else {
        const HapMut mu = M[m];
        const uint64_t np = mu.new_pos;
        const uint64_t nl = mu.nuc_len;
        if (hpos - np < nl) {
            sg.begin = np; sg.end = np + nl;
            sg.addr = mu.nuc_off + (hpos - np);
        } else {
            sg.begin = np + nl; sg.end = (m + 1 < n) ? M[m + 1].new_pos : H.cell_size[cell];
            sg.addr = chrom_off + (uint64_t)((int64_t)hpos + mu.ref_shift);
        }
    }
    return sg;
}
// last mutation of `cell` with new_pos <= hpos (cell-relative, -1 if none): the bucket index narrows it to the
// mutations that start in hpos's bucket (about one at 1.2 mutations per kb; the plain binary search took ~20 dependent
// loads from HBM tables per read end on a 125 Mbp chromosome), then a binary search among those
__device__ __forceinline__ int64_t hap_search(const HapDev& H, uint32_t cell, uint64_t hpos) {
    const uint64_t mo = H.cell_mut_off[cell];
    const uint64_t n = H.cell_mut_off[cell + 1] - mo;
    const uint64_t bo = H.bucket_off[cell], nb = H.bucket_off[cell + 1] - bo;      // nb = (cell_size >> shift) + 2 entries
    const uint64_t j = hpos >> JK_HAP_BUCKET_SHIFT;
    int64_t lo, hi;
    if (j + 1 < nb) { lo = H.bucket[bo + j]; hi = H.bucket[bo + j + 1]; }
    else { lo = nb ? H.bucket[bo + nb - 1] : 0; hi = (int64_t)n; }
    while (lo < hi) {                                          // first index with new_pos > hpos
        const int64_t mid = (lo + hi) >> 1;
        if (H.mut[mo + mid].new_pos <= hpos) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}

// HapChrom::get_chrom_full (src/hap_classes.cpp:80-116) for every cell of a haplotype set at once: the reference
// materialises a haplotype chromosome per thread each time the thread's cursor enters a cell; here every cell is
// written once into device memory (n_haplotypes x genome size bytes: 24 GB for BASELINE configs[3], of 288), after
// which a haplotype run reads plain contiguous sequences.  One thread = 16 haplotype positions: the mutation under the
// first is found through the bucket index (the 64 threads of a wave share one bucket), a run that lies in one segment
// is one unaligned 16-byte load and one aligned 16-byte store.
constexpr uint32_t JK_MAT_TILE = 4096;     // positions per 256-thread workgroup
__global__ void __launch_bounds__(256)
materialise_haps_kernel(const uint8_t* __restrict__ seq, const uint64_t* __restrict__ ref_off, uint32_t n_chroms, HapDev H, uint32_t n_cells,
                        const uint64_t* __restrict__ tile0 /* [n_cells + 1] first tile of each cell */,
                        const uint64_t* __restrict__ out_off /* [n_cells] */, uint8_t* __restrict__ out) {
    const uint64_t tile = blockIdx.x;
    uint32_t lo = 0, hi = n_cells;               // the cell this tile belongs to: last cell with tile0 <= tile
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (tile0[mid] <= tile) lo = mid; else hi = mid; }
    const uint32_t cell = lo;
    const uint64_t size = H.cell_size[cell];
    const uint64_t pos0 = (tile - tile0[cell]) * JK_MAT_TILE + (uint64_t)threadIdx.x * 16u;
    if (pos0 >= size) return;
    const uint32_t n = size - pos0 < 16u ? (uint32_t)(size - pos0) : 16u;
    const uint64_t coff = ref_off[cell % n_chroms];
    int64_t m = hap_search(H, cell, pos0);
    HapSeg sg = hap_resolve(H, coff, cell, m, pos0);
    uint8_t* dst = out + out_off[cell] + pos0;
    if (n == 16u && sg.end - pos0 >= 16u) {
        uint4 v;
        __builtin_memcpy(&v, seq + sg.addr, 16);
        *reinterpret_cast<uint4*>(dst) = v;
        return;
    }
    uint64_t at = pos0;                          // the position sg.addr stands for
    for (uint32_t k = 0; k < n; k++) {
        const uint64_t pos = pos0 + k;
        if (pos >= sg.end) { sg = hap_resolve(H, coff, cell, m, pos); at = pos; }
        dst[k] = seq[sg.addr + (pos - at)];
    }
}

// NE = number of read ends (1 single-end, 2 paired); BLOCK = workgroup size (1024 -> 4 waves/SIMD and a
// 128-VGPR budget, 512 -> 2 waves/SIMD and 256 VGPRs; with the tables in LDS one workgroup fits per CU).
// HAP = sequence a set of haplotypes (IlluminaHaplotypes, src/hts_illumina.h:509-675, .cpp:495-558):
// the lane walks (haplotype, chromosome) cells in order with per-cell quotas and reads bases through
// the mutation tables instead of materialising each haplotype chromosome as the reference does.
// (experiment switch: -DJK_GEN_WAVES=5 caps the generator at 96 VGPRs, which leaves room for other kernels' waves on
// its SIMDs -- measured in DESIGN.md section 4, not a gain)
#ifdef JK_GEN_WAVES
#define JK_GEN_ATTR __attribute__((amdgpu_waves_per_eu(JK_GEN_WAVES, JK_GEN_WAVES)))
#else
#define JK_GEN_ATTR
#endif
// SEG = the haplotype's bases are read through the mutation tables, segment by segment; HAP without SEG = every
// haplotype chromosome was materialised in device memory beforehand (materialise_haps_kernel): the cursor, quota, gamma
// and barcode logic of a haplotype run over plain contiguous sequences, g.chrom_off / hdr_off indexed by cell.
// E6 (with LDS_TAB) = the alias entries as two arrays, 6 bytes per entry: the cut points' high words, then {character kept,
// character of the alias} as byte pairs (IlluminaPacked6 in jk_host.h) -- for pairs of profiles that fit LDS that way only.
template <bool LDS_TAB, uint32_t NE, int BLOCK, bool HAP, bool SEG = HAP, bool E6 = false>
__global__ void __launch_bounds__(BLOCK) JK_GEN_ATTR
illumina_kernel(IlluminaKernelParams P) {
    extern __shared__ __align__(16) uint8_t smem[];
    TabPtrs T;
    // mm2 is the kernel's only static LDS object: its entries sit at compile-time LDS addresses 8*char
    __shared__ uint64_t s_mm[256];
    for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) s_mm[i] = P.mm2[i];
#ifndef JK_NO_PRIO_BALANCE
    __shared__ uint32_t s_prog[16];      // [SIMD][slot]: progress of the (up to 4) waves of this workgroup on a SIMD
    __shared__ uint32_t s_slots[4];
    if (threadIdx.x < 16) s_prog[threadIdx.x] = 0xffffffffu;
    if (threadIdx.x < 4) s_slots[threadIdx.x] = 0;
#endif
    if (LDS_TAB) {
        uint32_t* s_tab = reinterpret_cast<uint32_t*>(smem);
        const uint32_t n_words = 2u * P.n_info + (E6 ? P.n_entries + (P.n_entries + 1u) / 2u : 2u * P.n_entries);
        // entry offsets become absolute LDS addresses on the way in (one add less per base); E6: the second word of an
        // info entry holds the offset of the position's first character pair above its entry count
        const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
        for (uint32_t i = threadIdx.x; i < n_words; i += blockDim.x)
            s_tab[i] = P.tab[i] + (i < 2u * P.n_info ? (!(i & 1u) ? lds_base : (E6 ? lds_base << 8 : 0u)) : 0u);
        T.tab = smem;
    } else {
        // profiles whose alias entries do not fit in LDS (every built-in one but HiSeq 2500 150 bp) still have a small
        // {entry offset, entry count} table -- 8 bytes per (end, position, nucleotide): 16 KB at 250 bases -- which is
        // staged when there is room (info_in_lds): one dependent global load per base instead of two
        if (P.info_in_lds) {
            uint32_t* s_tab = reinterpret_cast<uint32_t*>(smem);
            for (uint32_t i = threadIdx.x; i < 2u * P.n_info; i += blockDim.x) s_tab[i] = P.tab[i];
        }
        T.tab = reinterpret_cast<const uint8_t*>(P.tab);
    }
    const bool info_lds = LDS_TAB || P.info_in_lds;
    // offset of the first alias entry, in the units of the info table's entry offsets (absolute LDS addresses when LDS_TAB)
    const uint32_t ent_base = 8u * P.n_info + (LDS_TAB ? (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem : 0u);
    // packed reference: 8 bits (4 bases, ascending source order) -> the 4 code bytes in read order; entries 256..511 for
    // the reverse strand (descending source order, complemented: code ^ 2)
    if (!SEG && P.lds_cell_off != 0xffffffffu) reinterpret_cast<uint32_t*>(smem + P.lds_cell_off)[threadIdx.x] = 0;     // chromosome cache: empty
    uint32_t* const s_lut = reinterpret_cast<uint32_t*>(smem + P.lds_lut_off);
    if (!SEG && P.g.packed) {
        for (uint32_t e = threadIdx.x; e < 512u; e += blockDim.x) {
            const uint32_t a = e & 3u, b = (e >> 2) & 3u, c = (e >> 4) & 3u, d = (e >> 6) & 3u;
            s_lut[e] = e < 256u ? (a | (b << 8) | (c << 16) | (d << 24))
                                : ((d ^ 2u) | ((c ^ 2u) << 8) | ((b ^ 2u) << 16) | ((a ^ 2u) << 24));
        }
    }
    __syncthreads();

    const uint32_t lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= P.n_lanes) return;

#ifdef JK_TIMELINE
    const uint64_t tl_t0 = wall_clock64();
#endif
#ifndef JK_NO_PRIO_BALANCE
    // Wave balancing.  The SIMD arbiter issues oldest-first, and the lanes of a launch all have the same amount of
    // work: left alone, the four waves of a SIMD finish at 50 %, 61 %, 76 % and 93 % of the kernel's duration
    // (tools/timeline.py), and a wave on its own only fills ~40 % of the issue slots.  Every 64 draws a wave
    // publishes its progress and takes the priority (s_setprio) of its rank among the waves of its SIMD, the one
    // furthest behind highest: all waves then end within 3 % of each other and the launch is 13 % shorter.
    const uint32_t bal_simd = (__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4)) & 3u;     // HW_ID[5:4]
    uint32_t bal_k = 0;
    if ((threadIdx.x & 63u) == 0) bal_k = atomicAdd(&s_slots[bal_simd], 1u);
    bal_k = __builtin_amdgcn_readfirstlane(bal_k);
    const bool bal_on = bal_k < 4u;
    uint32_t bal_prog = 0, bal_it = 0;
    auto bal_tick = [&]() {
        if (!bal_on) return;
        bal_prog += bal_it >> 6; bal_it &= 63u;
        s_prog[bal_simd * 4u + bal_k] = bal_prog;
        const uint4 pr = *reinterpret_cast<const uint4*>(&s_prog[bal_simd * 4u]);
        // waves that are ahead of this one (finished or absent ones read 0xffffffff: a constant offset)
        uint32_t ahead = (pr.x > bal_prog) + (pr.y > bal_prog) + (pr.z > bal_prog) + (pr.w > bal_prog);
        ahead = __builtin_amdgcn_readfirstlane(ahead);
        const uint32_t present = __builtin_amdgcn_readfirstlane((pr.x != 0xffffffffu) + (pr.y != 0xffffffffu) + (pr.z != 0xffffffffu) + (pr.w != 0xffffffffu));
        const uint32_t rank = ahead - (4u - present);           // 0 = leader .. 3 = last
        if (rank == 0) __builtin_amdgcn_s_setprio(0);
        else if (rank == 1) __builtin_amdgcn_s_setprio(1);
        else if (rank == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
    };
// n = draws just made (wave-uniform); the priorities are looked at again every 64 draws
#define JK_BAL_STEP(n) do { bal_it += (n); if (bal_it >= 64u) bal_tick(); } while (0)
#else
#define JK_BAL_STEP(n) do { } while (0)
#endif
    LaneRng rng;
    rng.e = jk_pcg_limbs(jk_pcg_seed(P.seeds + (size_t)lane * 8));
    jk_gamma_state gst; gst.saved = 0.0; gst.saved_available = 0; gst.fail = 0;

    const uint32_t L = P.read_len;
    uint32_t bc = P.bc_len;                      // HAP: per haplotype, reloaded when the cursor moves
    // (a lane's read count fits 32 bits: plan_lanes refuses more; 32-bit counters keep four more registers free)
    const uint32_t quota = (uint32_t)P.lane_reads[lane];
    uint32_t made = 0, in_pool = 0;
    const uint32_t pool_size = P.pool_size > 0xffffffffULL ? 0xffffffffu : (uint32_t)P.pool_size;

    // chromosome cursor: the reference rescans from chromosome 0 for the first non-zero quota
    // (src/hts_illumina.cpp:199-200); quotas only ever decrease, so a monotone cursor is the same.
    // HAP: `ci` is the cell index hap * n_chroms + chrom, `n_cells` the number of cells.
    const uint32_t n_cells = HAP ? P.h.n_haps * P.g.n_chroms : P.g.n_chroms;
    uint32_t ci = 0;
    uint32_t ccnt = n_cells ? P.chrom_reads[lane] : 0;
    uint32_t cur_hap = 0xffffffffu;

    // the tile is the wave's: its offset and capacity are wave-uniform (scalar registers), a lane only adds its column
    const uint32_t tile = __builtin_amdgcn_readfirstlane(lane) >> 6;
    const uint64_t tile_off = P.pool_off[tile];
    const uint64_t lane_cap = (P.pool_off[tile + 1] - tile_off) >> 6;     // bytes per lane in this tile
    const uint32_t colb = (lane & 63u) * 4u;
    OutStream os[2];
    os_begin(os[0], 0);
    os_begin(os[1], 0);     // unused when NE == 1

    uint64_t frag_len = 0, frag_start = 0;
    uint32_t err = 0;
    const uint32_t W = P.ev_words;
    const size_t ev_stride = (size_t)P.n_lanes;
    // Indel events of the current read ends live in per-lane bitmaps in HBM (they are rare: ~0.03 per
    // read): plane p of end r, word w -> evw[((r*4 + p) * W + w) * n_lanes + lane]; p: 0 insertions,
    // 1 deletions, 2/3 the two bits of each inserted base.  `evalid` says which words were written
    // for this read (bits 0-15 insertion words, 16-31 deletion words).
    auto evaddr = [&](uint32_t r, uint32_t p, uint32_t w) -> uint64_t* {
        return P.evw + ((size_t)((r * 4 + p) * W + w)) * ev_stride + lane;
    };

    bool is_dup = false;     // next pair re-reads the current fragment (re_read)
    while (made < quota) {
        if (!is_dup) {
            // ---- chrom_indels_frag: chromosome, fragment length, fragment start (hts_illumina.cpp:192-217)
            // (HAP: IlluminaHaplotypes::one_read's cursor search, hts_illumina.cpp:505-525 -- the first cell at
            //  or after the current one with reads left)
            // (four cells per look: a haplotype run has as many cells as haplotypes x chromosomes and a lane's few dozen pairs
            //  leave most of them empty, so cell by cell the cursor paid a dependent global load for every one of them)
            while (ci < n_cells && ccnt == 0) {
                uint32_t q[4];
#pragma unroll
                for (uint32_t k = 0; k < 4u; k++) q[k] = ci + 1u + k < n_cells ? P.chrom_reads[(size_t)(ci + 1u + k) * P.chrom_stride + lane] : 0u;
                if (q[0]) { ci += 1u; ccnt = q[0]; }
                else if (q[1]) { ci += 2u; ccnt = q[1]; }
                else if (q[2]) { ci += 3u; ccnt = q[2]; }
                else { ci += 4u; ccnt = q[3]; }
                if (ci > n_cells) ci = n_cells;
            }
            if (ci >= n_cells) { made = quota; break; }         // `finished`
            if (HAP) {
                const uint32_t hap = ci / P.g.n_chroms;
                if (hap != cur_hap) {       // a new IlluminaOneHaplotype: its own gamma state and barcode
                    cur_hap = hap;
                    gst.saved = 0.0; gst.saved_available = 0;
                    bc = P.h.bc_len[hap];
                }
            }
            const uint64_t chrom_len = HAP ? P.h.cell_size[ci] : P.g.chrom_len[ci];
            if (chrom_len == 0) { err |= JK_KERR_EMPTY_CHROM; break; }
            double gl = jk_gamma(P.gp, gst, rng);
            if (gst.fail) { err |= JK_KERR_GAMMA_MATH; break; }
            frag_len = (uint64_t)gl;
            if (frag_len < P.frag_min) frag_len = P.frag_min;
            if (frag_len > P.frag_max) frag_len = P.frag_max;
            if (frag_len >= chrom_len) { frag_len = chrom_len; frag_start = 0; }
            else frag_start = jk_frag_start(rng(), chrom_len - frag_len + 1);
        }
        // Per-chromosome data of the read ends below -- the chromosome's offset in the sequence buffer and the id line's
        // prefix -- come from eight LDS words per lane, refilled when the lane's cursor has moved to another chromosome
        // (s_cell; word 0 = cell index + 1).  The waves of a SIMD are kept in step (wave balancing above), so the two
        // dependent global loads the id line otherwise starts with stall all of them together: 4.7 % of the launch
        // (tools/ablate notes in DESIGN.md section 4).  Without room in LDS (lds_cell_off = 0xffffffff) every read end
        // loads them as before.
        uint32_t* const s_cell = (!SEG && P.lds_cell_off != 0xffffffffu) ? reinterpret_cast<uint32_t*>(smem + P.lds_cell_off) + threadIdx.x : nullptr;
        if (s_cell && s_cell[0] != ci + 1u) {
            const uint64_t co = P.g.chrom_off[ci];
            const uint32_t h0 = P.g.hdr_off[ci], hl = P.g.hdr_off[ci + 1] - h0;
            const uint8_t* hp = P.g.hdr_blob + h0;            // (the blob is padded: the four loads need no bounds)
            uint32_t w4[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) __builtin_memcpy(&w4[k], hp + 4u * k, 4);
            s_cell[0] = ci + 1u; s_cell[BLOCK] = (uint32_t)co; s_cell[2 * BLOCK] = (uint32_t)(co >> 32); s_cell[3 * BLOCK] = hl;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) s_cell[(4 + k) * BLOCK] = w4[k];
        }
        // ---- sample_indels + adjust_chrom_spaces (hts_illumina.cpp:117-184): one draw per fragment
        // position; anything but a match is rare and is recorded in the HBM bitmaps
        // which bitmap words a read end has written: insertion words in the low half, deletion words in the high half of a
        // 32-bit mask (16 words, reads up to 480) when the tables sit in LDS, of a 64-bit mask (32 words, up to 992) otherwise
        using ev_t = typename std::conditional<LDS_TAB, uint32_t, uint64_t>::type;
        constexpr uint32_t DSH = LDS_TAB ? 16u : 32u;
        constexpr ev_t HALF = LDS_TAB ? (ev_t)0xffffu : (ev_t)0xffffffffu;
        uint32_t space_len[2]; ev_t evalid[2];       // space | out_len << 16 (both are at most 2 * read_len + 64 < 2^16)
        const uint32_t fl32 = frag_len > 0xffffffffULL ? 0xffffffffu : (uint32_t)frag_len;
#pragma unroll
        for (uint32_t r = 0; r < NE; r++) {
            uint32_t frag_pos = 0, len_now = 0, n_ins = 0, n_del = 0; ev_t ev = 0;
            const uint64_t thm = P.th_match[r], thd = P.th_del[r];
            const bool nm = P.never_match[r], nd = P.never_del[r];
            auto indel_event = [&](uint64_t x) {     // the draw at frag_pos was not a match
                const bool is_del = !nd && x >= thd;
                if (!is_del && len_now == L - 1) { len_now++; return; }       // insertion after the last base: counted, not recorded
                const uint32_t w = frag_pos >> 6;
                if (w >= W) { err |= JK_KERR_TOO_MANY_DELETIONS; len_now = L; return; }
                const uint32_t vb = (is_del ? DSH : 0u) + w;
                uint64_t* a = evaddr(r, is_del ? 1u : 0u, w);
                const uint64_t old = ((ev >> vb) & 1u) ? *a : 0ULL;
                *a = old | (1ULL << (frag_pos & 63u));
                ev |= (ev_t)1 << vb;
                if (is_del) n_del++; else { n_ins++; len_now += 2; }
            };
            // Matches are all but ~3e-4 of the draws, and until an event the lanes of a wave are at the same
            // position: run the draws in wave-uniform stretches (no per-lane loop condition, position counters
            // added once per stretch), as long as every lane has steps left; an event ends the stretch.
            for (;;) {
                uint32_t steps = 0;       // draws this lane makes for sure if they are all matches
                if (len_now < L && frag_pos < fl32) { const uint32_t sa = L - len_now, sb = fl32 - frag_pos; steps = sa < sb ? sa : sb; }
                if (__builtin_amdgcn_ballot_w64(steps == 0) != 0) break;
                uint32_t n_uni = __builtin_amdgcn_readfirstlane(steps);        // minimum over the active lanes
                for (uint64_t less = __builtin_amdgcn_ballot_w64(steps < n_uni); less; less = __builtin_amdgcn_ballot_w64(steps < n_uni))
                    n_uni = __builtin_amdgcn_readlane(steps, (int)__builtin_ctzll(less));
                uint32_t k = 0; bool hit = false;
                uint64_t x = 0;
                // (the ballot of the compare alone is the compare's own mask; with `nm ||` inside it the compiler goes
                // through a VGPR: two more VALU instructions per draw)
                while (k < n_uni) {
                    x = rng(); k++;
                    if (nm || __builtin_amdgcn_ballot_w64(x < thm) != 0) { hit = true; break; }
                }
                if (hit && (nm || x < thm)) { frag_pos += k - 1; len_now += k - 1; indel_event(x); frag_pos++; }
                else { frag_pos += k; len_now += k; }
                JK_BAL_STEP(k);
            }
            // the remaining draws of lanes that stop at different positions, one masked step at a time
            while (len_now < L && frag_pos < fl32) {
                const uint64_t x = rng();
                if (!nm && x >= thm) len_now++; else indel_event(x);
                frag_pos++;
            }
            uint64_t sp = (uint64_t)L + n_del - n_ins;
            if (sp > frag_len) sp = frag_len;
            space_len[r] = (uint32_t)sp | (((uint32_t)sp - n_del + n_ins) << 16);
            evalid[r] = ev;
        }
        if (err) break;

        // ---- append_pools (hts_illumina.cpp:339-409)
        bool reverse = jk_runif_lt_half(rng());
#pragma unroll
        for (uint32_t i = 0; i < NE; i++) {
            const uint32_t sp = space_len[i] & 0xffffu, n_out = space_len[i] >> 16;
            ev_t ev = evalid[i];
            const uint32_t ev_any = (uint32_t)((ev | (ev >> DSH)) & HALF);      // words with any event
            const uint64_t cspace = (uint64_t)sp - bc;
            uint64_t start;
            if ((!P.matepair && !reverse) || (P.matepair && reverse)) start = frag_start;
            else start = frag_start + frag_len - cspace;

            // inserted bases are drawn first, right to left (hts_illumina.h:213-225)
            if (ev & HALF) {
                for (int w = (int)W - 1; w >= 0; w--) {
                    if (!((ev >> w) & 1u)) continue;
                    uint64_t rest = *evaddr(i, 0, w), b0 = 0, b1 = 0, nul = 0;
                    while (rest) {
                        const int bit = 63 - jk_clz64(rest);
                        rest &= ~(1ULL << bit);
                        // index 4 needs x == 2^64-1 (2^-64 per insertion): the reference then inserts "TCAG"[4], the
                        // string's NUL, which fill_read_qual turns into 'N'.  Kept as "insertion AND deletion" at the
                        // same position, a combination sample_indels never produces.
                        const uint64_t b = jk_runif_index(rng(), 4);
                        b0 |= (b & 1u) << bit; b1 |= ((b >> 1) & 1u) << bit; nul |= (b >> 2) << bit;
                    }
                    *evaddr(i, 2, w) = b0; *evaddr(i, 3, w) = b1;
                    if (nul) {
                        uint64_t* a = evaddr(i, 1, w);
                        *a = (((ev >> (DSH + w)) & 1u) ? *a : 0ULL) | nul;
                        ev |= (ev_t)1 << (DSH + w);
                    }
                }
            }

            // (the pointer as a scalar pair of its own: left as a member of the kernel-argument block, every chunk load
            // restores the block's whole 8-register tuple from its spill lanes -- v_readlane is a VALU instruction)
            gbytes_t gseq = (gbytes_t)P.g.seq;
            asm volatile("" : "+s"(gseq));
            uint32_t pf[2] = {0, 0};
            bool have_pf = false;
            uint64_t coff = 0;
            if (!SEG) coff = s_cell ? (((uint64_t)s_cell[2 * BLOCK] << 32) | s_cell[BLOCK]) : P.g.chrom_off[ci];

            // ---- FASTQ id line (fill_fq_lines, hts_illumina.cpp:286-312)
            OutStream& o = os[i];
            uint8_t* const col = P.pool[i] + tile_off + colb;
            {
                // "@<genome>-<chrom>-": 4 bytes per load and append (the blob is padded, so the first four loads need no
                // bounds; their latencies overlap)
                uint32_t h0 = 0, hlen;
                uint32_t hw[4];
                if (s_cell) {
                    hlen = s_cell[3 * BLOCK];
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) hw[k] = s_cell[(4 + k) * BLOCK];
                    if (__builtin_amdgcn_ballot_w64(hlen > 16u) != 0) h0 = P.g.hdr_off[ci];
                } else {
                    h0 = P.g.hdr_off[ci]; hlen = P.g.hdr_off[ci + 1] - h0;
                    const uint8_t* hp0 = P.g.hdr_blob + h0;
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) __builtin_memcpy(&hw[k], hp0 + 4u * k, 4);
                    // (waited for here, on this path only: left pending, the compiler guards every later reuse of these
                    // registers with a vmcnt wait on the common path too -- and vmcnt counts the pool stores as well)
                    __builtin_amdgcn_s_waitcnt(0x0f70);
                }
                const uint8_t* hp = P.g.hdr_blob + h0;
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    if (4u * k < hlen) {
                        const uint32_t n = hlen - 4u * k < 4u ? hlen - 4u * k : 4u;
                        os_put_n(o, col, n < 4u ? hw[k] & ((1u << (8u * n)) - 1u) : hw[k], n);
                    }
                }
                for (uint32_t k = 16; k < hlen; k += 4) {
                    uint32_t w; __builtin_memcpy(&w, hp + k, 4);
                    const uint32_t n = hlen - k < 4u ? hlen - k : 4u;
                    os_put_n(o, col, n < 4u ? w & ((1u << (8u * n)) - 1u) : w, n);
                }
                // decimal digits of the start position: built least significant first by shifting characters into a
                // 96-bit string, so the most significant digit ends up in the lowest byte (the first to go out)
                uint32_t s0 = 0, s1 = 0, s2 = 0, nd = 0;
                if (__builtin_amdgcn_ballot_w64(start >= 1000000000000ULL) != 0) {      // more than 12 digits: byte by byte
                    uint64_t v = start, packed_lo = 0, packed_hi = 0;
                    do {
                        const uint64_t q = v / 10, d = v - q * 10;
                        packed_hi = (packed_hi << 4) | (packed_lo >> 60);
                        packed_lo = (packed_lo << 4) | d;
                        v = q; nd++;
                    } while (v);
                    for (uint32_t d = 0; d < nd; d++) {
                        os_put(o, col, '0' + (uint32_t)(packed_lo & 15u));
                        packed_lo = (packed_lo >> 4) | (packed_hi << 60); packed_hi >>= 4;
                    }
                } else {
                    auto push_digit = [&](uint32_t d) {
                        s2 = (s2 << 8) | (s1 >> 24); s1 = (s1 << 8) | (s0 >> 24); s0 = (s0 << 8) | ('0' + d);
                        nd++;
                    };
                    uint64_t v = start;
                    if (__builtin_amdgcn_ballot_w64((v >> 32) != 0) != 0) {
                        while (v >> 32) { const uint64_t q = v / 10; push_digit((uint32_t)(v - q * 10)); v = q; }
                    }
                    uint32_t v32 = (uint32_t)v;
                    if (!(start >> 32) || v32) {        // (a 64-bit start whose low part came out as 0 has all its digits already)
                        do { const uint32_t q = v32 / 10u; push_digit(v32 - q * 10u); v32 = q; } while (v32);
                    }
                    os_put_n(o, col, s0, nd < 4u ? nd : 4u);
                    if (nd > 4u) os_put_n(o, col, s1, nd - 4u < 4u ? nd - 4u : 4u);
                    if (nd > 8u) os_put_n(o, col, s2, nd - 8u);
                }
                uint32_t sfx = '-' | ((reverse ? (uint32_t)'R' : (uint32_t)'F') << 8);
                if (P.paired) {
                    os_put_n(o, col, sfx | ((uint32_t)'/' << 16) | (('1' + i) << 24), 4);
                    os_put_n(o, col, '\n', 1);
                } else {
                    os_put_n(o, col, sfx | ((uint32_t)'\n' << 16), 3);
                }
            }
            // the quality line is produced in the same pass as the bases, by its own stream that
            // starts right after the bases with "\n+\n"
            OutStream oq;
            os_begin(oq, o.pos + n_out);
            os_put_n(oq, col, (uint32_t)'\n' | ((uint32_t)'+' << 8) | ((uint32_t)'\n' << 16), 3);

            // ---- bases + qualities (fill_read / rev_comp / fill_read_qual).
            // Source position pp of the pre-indel read: pp < bc -> barcode; else forward chrom[start + pp - bc],
            // reverse: complement of chrom[start + sp - 1 - pp]; in both cases one byte of the encoded buffer at
            // A + pp (forward) or A - pp (reverse), A being fixed per read end (per segment for haplotypes).
            //
            // The loop runs in two gears, chosen wave-uniformly:
            //  * quads: when every lane has at least 4 (8) plain TCAG source bases ahead -- no barcode, indel event,
            //    segment boundary, 'N' or read end in reach -- each lane loads its next 4 (8) source bytes with one
            //    UNALIGNED load (so all lanes refill in the same iteration, whatever their alignment), turns them
            //    into read order with v_perm (byte swap + complement for the reverse strand), maps the four codes of
            //    a word to ASCII with one more v_perm, and runs four unrolled copies of the quality/mismatch step
            //    with compile-time shifts;
            //  * single bases, the general path (barcodes, indels, 'N', segment changes, uneven read lengths), until
            //    the output position is a multiple of 4 again and no lane needs it.
            const uint8_t* const bcode = HAP ? P.h.bc_blob + (size_t)cur_hap * JK_MAX_BARCODE : P.barcode;
            uint64_t A = 0;
            // HAP: the read is served segment by segment (reference runs and mutation bytes);
            // seg_end_pp = first source position that is NOT in the current segment.  The segments of the
            // read's window are resolved ONCE here, before the per-base loop, into a small per-lane table in
            // LDS (start position, A of the segment; up to JK_HAP_SEGS of them): a boundary met inside the loop then
            // costs two LDS reads instead of a chain of dependent table loads from HBM.  Reads that cross more
            // segments than the table holds resolve the rest on the fly.
            uint32_t seg_end_pp = 0xffffffffu;
            uint32_t seg_state = 0;                 // current segment | segments in the table << 8 | first unresolved position << 16
            constexpr uint32_t NO_POS = 0xffffu;
            uint32_t* const s_seg = SEG ? reinterpret_cast<uint32_t*>(smem + P.lds_seg_off) + threadIdx.x : nullptr;
            auto seg_hpos = [&](uint32_t pp) -> uint64_t { return reverse ? (start + sp - 1 - pp) : (start + pp - bc); };
            auto seg_A = [&](uint64_t addr, uint32_t pp) -> uint64_t { return reverse ? addr + pp : addr - pp; };
            auto seg_enter = [&](uint32_t pp) {     // SEG only; pp >= seg_end_pp: move to the next segment
                const uint32_t cur = (seg_state & 0xffu) + 1u, n = (seg_state >> 8) & 0xffu, over = seg_state >> 16;
                if (cur < n) {
                    const uint32_t* e = s_seg + 3u * cur * BLOCK;
                    A = ((uint64_t)e[2 * BLOCK] << 32) | e[BLOCK];
                    seg_end_pp = cur + 1u < n ? e[3 * BLOCK] : (over != NO_POS ? over : 0xffffffffu);
                    seg_state = (seg_state & ~0xffu) | cur;
                } else {                            // beyond the table (rare): resolve from scratch
                    const uint64_t hpos = seg_hpos(pp);
                    int32_t m = (int32_t)hap_search(P.h, ci, hpos);
                    const HapSeg sg = hap_resolve(P.h, P.g.chrom_off[ci % P.g.n_chroms], ci, m, hpos);
                    A = seg_A(sg.addr, pp);
                    const uint64_t avail = reverse ? (hpos - sg.begin + 1) : (sg.end - hpos);
                    seg_end_pp = avail >= (uint64_t)(0xffffffffu - pp) ? 0xffffffffu : pp + (uint32_t)avail;
                }
            };
            if (SEG) {
                int32_t m = (int32_t)hap_search(P.h, ci, seg_hpos(bc));      // (a cell holds fewer than 2^31 mutations: checked at upload)
                const uint64_t coff = P.g.chrom_off[ci % P.g.n_chroms];
                uint32_t q = bc, n = 0;
                while (q < sp && n < JK_HAP_SEGS) {
                    const uint64_t hpos = seg_hpos(q);
                    const HapSeg sg = hap_resolve(P.h, coff, ci, m, hpos);
                    const uint64_t a = seg_A(sg.addr, q);
                    uint32_t* e = s_seg + 3u * n * BLOCK;
                    e[0] = q; e[BLOCK] = (uint32_t)a; e[2 * BLOCK] = (uint32_t)(a >> 32);
                    n++;
                    const uint64_t avail = reverse ? (hpos - sg.begin + 1) : (sg.end - hpos);
                    q = avail >= (uint64_t)(sp - q) ? sp : q + (uint32_t)avail;
                }
                seg_state = (n << 8) | ((q < sp ? q : NO_POS) << 16);
                if (n) {
                    A = ((uint64_t)s_seg[2 * BLOCK] << 32) | s_seg[BLOCK];
                    seg_end_pp = n > 1u ? s_seg[3 * BLOCK] : (q < sp ? q : 0xffffffffu);
                }
            } else {
                A = coff + (reverse ? start + sp - 1 : start - bc);
            }
            // packed reference: may this read end take its 8-base blocks from the 2-bit copy?  (Its source window
            // [lo, lo + cspace) must not touch a flagged 64-base block, and the copy's byte offsets are kept in 32 bits.)
            bool use_bytes = true;
            uint32_t pk_off = 0, pk_sh = 0;     // byte offset of the current block's dword in `packed`; bit position of its first-quad byte
            gbytes_t pk = (gbytes_t)P.g.packed;
            asm volatile("" : "+s"(pk));
            if (!SEG && P.g.packed != nullptr && sp > bc) {
                const uint64_t lo = coff + start, last = lo + cspace - 1;
                const uint64_t b0 = lo >> 12;
                const uint32_t two = (last >> 12) != b0 ? 3u : 1u;        // (a window is shorter than 4096 bases: at most two blocks)
                uint32_t fw;
                __builtin_memcpy(&fw, P.g.nflags + (b0 >> 3), 4);
                const uint32_t bits = (fw >> ((uint32_t)b0 & 7u)) & two;
                use_bytes = bits != 0u || (last >> 2) >= 0xffffff00ULL;
            }
            auto src_byte = [&](uint32_t pp) -> uint32_t {     // general path: one source base (pp >= bc)
                if (SEG) { while (pp >= seg_end_pp) seg_enter(pp); }
                uint32_t c;
                if (!SEG && !use_bytes) {
                    const uint64_t p = reverse ? A - pp : A + pp;
                    c = (gload8(pk + (uint32_t)(p >> 2)) >> (2u * ((uint32_t)p & 3u))) & 3u;
                } else c = gload8(gseq + (reverse ? A - pp : A + pp));
                if (reverse) c ^= ((~c) >> 1) & 2u;                 // codes 0..3: ^2 (T<->A, C<->G); others stay non-TCAG
                return c;
            };
            // first position of (ins|del) at or after pp, or "none"
            auto next_event = [&](uint32_t pp) -> uint32_t {
                for (uint32_t w = pp >> 6; w < W; w++) {
                    if (!((ev_any >> w) & 1u)) continue;
                    uint64_t m = (((ev >> w) & 1u) ? *evaddr(i, 0, w) : 0ULL) | (((ev >> (DSH + w)) & 1u) ? *evaddr(i, 1, w) : 0ULL);
                    if (w == (pp >> 6)) m &= ~0ULL << (pp & 63u);
                    if (m) return w * 64u + (uint32_t)__builtin_ctzll(m);
                }
                return 0xffffffffu;
            };
            // `nes`: the next source position that needs the general path (barcode, deletion, insertion, segment
            // change); 0 forces it for the next base (pending inserted base).
            // (HAP: `nev` keeps the part of it that is not a segment change, for the gathering gear below)
            uint32_t nev = 0;
            auto next_slow = [&](uint32_t pp) -> uint32_t {
                uint32_t e = ev_any ? next_event(pp) : 0xffffffffu;
                if (SEG) nev = e;
                return (SEG && seg_end_pp < e) ? seg_end_pp : e;
            };

            uint32_t pp = 0, op = 0;
            uint32_t nes = bc ? 0u : next_slow(0);
            bool pending = false; uint32_t pend_base = 0;
            // Output: the lane's phase in its 4-byte words (o.cnt / oq.cnt pending bytes, < 4) does not change while
            // whole quads are appended, so a quad is merged with one 64-bit shift and leaves as one word per stream,
            // stored by every lane of the wave in the same instruction.
            const uint32_t sh_b = 8u * (o.pos & 3u), sh_q = 8u * (oq.pos & 3u);
            uint32_t acc_b = o.acc, acc_q = oq.acc;
            uint8_t* wpb = os_word(col, o.pos); uint8_t* wpq = os_word(col, oq.pos);
            auto put_quad = [&](uint32_t gb, uint32_t gq) {
                const uint64_t tb = (uint64_t)gb << sh_b, tq = (uint64_t)gq << sh_q;
                pool_store(wpb, acc_b | (uint32_t)tb); acc_b = (uint32_t)(tb >> 32); wpb += TILE_ROW;
                pool_store(wpq, acc_q | (uint32_t)tq); acc_q = (uint32_t)(tq >> 32); wpq += TILE_ROW;
            };
            // one quality + mismatch step (IlluminaQualityError::fill_read_qual, hts_illumina.h:243-256) for a TCAG
            // base with code c (c8 = 8*c) at output position `opos`: returns 8 * quality character, sets `mism`
            // (x1 = the step's first draw: made by the caller, because a non-TCAG position uses it differently)
            auto qual_step = [&](uint32_t c8, uint32_t opos, uint64_t x1, bool& mism) -> uint32_t {
                const uint32_t inf_off = (i * L + opos) * 32u + c8;
                const uint2 inf = info_lds ? *reinterpret_cast<const uint2*>(smem + inf_off) : *reinterpret_cast<const uint2*>(T.tab + inf_off);
                const uint32_t ent_off = inf.x, nq = E6 ? (inf.y & 0xffu) : inf.y;
                const uint32_t idx = alias_index32(x1, nq);
                const uint32_t eoff = idx * (E6 ? 4u : 8u) + ent_off;
                uint32_t th_hi, qp;
                if (E6) {
                    th_hi = *(const __attribute__((address_space(3))) uint32_t*)(uintptr_t)eoff;
                    qp = *(const __attribute__((address_space(3))) uint16_t*)(uintptr_t)((inf.y >> 8) + idx * 2u);
                } else if (LDS_TAB) {
                    const __attribute__((address_space(3))) uint32_t* ep = (const __attribute__((address_space(3))) uint32_t*)(uintptr_t)eoff;
                    th_hi = ep[0]; qp = ep[1];
                } else {
                    const uint2 ev2 = *reinterpret_cast<const uint2*>(T.tab + eoff);
                    th_hi = ev2.x; qp = ev2.y;
                }
                const uint64_t x2 = rng();
                const uint64_t x3 = rng();
                // u < Prob[i]  <=>  x2 < cut point: decided by the high words unless they are equal (2^-32 per draw)
                const uint32_t x2h = (uint32_t)(x2 >> 32);
                bool keep = x2h < th_hi;
                if (__builtin_amdgcn_ballot_w64(x2h == th_hi) != 0) {
                    asm volatile("" ::: "memory");
                    if (x2h == th_hi) {
                        keep = (uint32_t)x2 < P.tab_lo[(eoff - ent_base) >> (E6 ? 2 : 3)];
                        const uint32_t k = atomicAdd(P.rare_log, 1u);
                        if (k < JK_RARE_LOG_CAP) P.rare_log[1 + k] = P.rare_lane0 + lane;
                    }
                }
                const uint32_t ch8 = E6 ? ((keep ? (qp & 0xffu) : (qp >> 8)) << 3) : (keep ? (qp & 0xffffu) : (qp >> 16));
                const uint64_t mmth = *reinterpret_cast<const uint64_t*>(reinterpret_cast<const uint8_t*>(s_mm) + ch8);
                mism = x3 < mmth;
                return ch8;
            };
            // mm_nucleos = {"CAG","TAG","TCG","TCA"} (src/hts.h:46): the m-th base that is not c
            auto mismatch_char = [&](uint32_t c) -> uint32_t {
                const uint32_t m = runif_index32(rng(), 3);
                return (m < 3u) ? base_char(m + (m >= c ? 1u : 0u)) : 0u;
            };
            uint32_t grp_b = 0, grp_q = 0;          // the quad being filled by the general path
            const uint32_t rsel = reverse ? 0x04050607u : 0x03020100u;      // v_perm selectors: read order of a chunk
            const uint32_t rcm = reverse ? 0x02020202u : 0u;                // complement of codes 0..3
            // the 8-base gear loads one block ahead: the chunk of the NEXT 8 positions is requested when this block starts,
            // so its latency (and that of the pool stores queued before it) has a whole block to pass; it is used only if
            // the next iteration takes the 8-base gear again (then every lane has advanced by exactly 8)

            while (op < n_out) {
                // ---- gear choice (wave-uniform)
                uint32_t nquads = 0, room = 0;
                bool gather_gear = false;
                if ((op & 3u) == 0) {
                    room = pending ? 0u : (nes - pp < n_out - op ? nes - pp : n_out - op);    // pp <= nes unless pending
                    if (__builtin_amdgcn_ballot_w64(room < 4u) == 0)
                        nquads = __builtin_amdgcn_ballot_w64(room < 8u) == 0 ? 2u : 1u;
                    else if (SEG) {
                        // Haplotypes: at 1.2 mutations per kb some lane of the wave meets a segment boundary in a third of
                        // all quads.  If segment changes are all that stands in the way, those lanes gather their four
                        // bases one by one across the boundary and the wave still takes the 4-base gear.
                        const uint32_t room_ev = pending ? 0u : (nev - pp < n_out - op ? nev - pp : n_out - op);
                        if (__builtin_amdgcn_ballot_w64(room_ev < 4u) == 0) { nquads = 1u; gather_gear = true; }
                    }
                }
                uint32_t wlo = 0, whi = 0, nmlo = 0, nmhi = 0;
                bool seg_moved = false, any_n = false;      // any_n: some lane's chunk holds a byte that is not TCAG (wave-uniform)
                if (nquads == 2u) {
                    if (!SEG && !use_bytes) {
                        // 16 bits of the 2-bit copy: the block's 8 source bases in ascending order, from bit pk_sh' of the
                        // dword at pk_off (2 bytes further per block, so the shift stays what it is); each half goes through
                        // the LDS table into read order.  The next block's dword is requested now, like the byte chunk below.
                        uint32_t v;
                        if (have_pf) v = pf[0];
                        else {
                            const uint64_t p0 = reverse ? A - pp - 7u : A + pp;        // lowest source base of the block
                            pk_off = (uint32_t)(p0 >> 2);
                            pk_sh = 2u * ((uint32_t)p0 & 3u) + (reverse ? 8u : 0u);
                            v = gload32(pk + pk_off);
                        }
                        const uint32_t* lut = s_lut + (reverse ? 256u : 0u);
                        const uint32_t q0 = __builtin_amdgcn_ubfe(v, pk_sh, 8u), q1 = __builtin_amdgcn_ubfe(v, reverse ? pk_sh - 8u : pk_sh + 8u, 8u);
                        pk_off += reverse ? 0xfffffffeu : 2u;
                        pf[0] = gload32(pk + pk_off);                                  // (stays inside the copy's padding)
                        wlo = lut[q0];
                        whi = lut[q1];
                    }
                    bool n_here = false;
                    if (SEG || use_bytes) {
                        uint32_t v[2];
                        if (have_pf) { v[0] = pf[0]; v[1] = pf[1]; }
                        else gload64(gseq + (reverse ? A - pp - 7u : A + pp), v);
                        gload64(gseq + (reverse ? A - pp - 15u : A + pp + 8u), pf);      // (stays inside the buffer's padding)
                        const uint32_t rlo = __builtin_amdgcn_perm(v[1], v[0], rsel), rhi = __builtin_amdgcn_perm(v[1], v[0], rsel ^ 0x04040404u);
                        wlo = rlo ^ rcm; whi = rhi ^ rcm;
                        n_here = ((v[0] | v[1]) & 0xfcfcfcfcu) != 0;
                        nmlo = rlo & 0xfcfcfcfcu; nmhi = rhi & 0xfcfcfcfcu;       // non-zero bytes: positions that are not TCAG
                    }
                    have_pf = true;
                    any_n = __builtin_amdgcn_ballot_w64(n_here) != 0;
                } else have_pf = false;
                if (nquads == 1u) {
                    if (SEG && gather_gear) {
                        const uint64_t A0 = A; const uint32_t se0 = seg_end_pp, st0 = seg_state;
                        uint32_t w, bad;
                        if (room >= 4u) {
                            uint32_t v;
                            v = gload32(gseq + (reverse ? A - pp - 3u : A + pp));
                            w = __builtin_amdgcn_perm(0u, v, rsel & 0x03030303u) ^ rcm; bad = v & 0xfcfcfcfcu;
                        } else {
                            // one 4-byte load per segment the four positions touch: each is read as if all four lay in
                            // that segment, and contributes the bytes of the positions that do
                            w = 0;
                            uint32_t filled = 0;
                            while (filled < 4u) {
                                while (pp + filled >= seg_end_pp) seg_enter(pp + filled);
                                uint32_t v;
                                v = gload32(gseq + (reverse ? A - pp - 3u : A + pp));
                                const uint32_t left = seg_end_pp - (pp + filled), n = left < 4u - filled ? left : 4u - filled;
                                const uint32_t mask = (n >= 4u ? 0xffffffffu : ((1u << (8u * n)) - 1u)) << (8u * filled);
                                w |= (__builtin_amdgcn_perm(0u, v, rsel & 0x03030303u) ^ rcm) & mask;
                                filled += n;
                            }
                            bad = w & 0xfcfcfcfcu;
                            seg_moved = true;
                        }
                        if (__builtin_amdgcn_ballot_w64(bad != 0) == 0) wlo = w;
                        else { A = A0; seg_end_pp = se0; seg_state = st0; seg_moved = false; nquads = 0; }    // not TCAG somewhere: the general path redoes it
                    } else {
                        bool n_here = false;
                        if (!SEG && !use_bytes) {
                            const uint64_t p0 = reverse ? A - pp - 3u : A + pp;
                            const uint32_t v = gload32(pk + (uint32_t)(p0 >> 2));
                            wlo = (s_lut + (reverse ? 256u : 0u))[__builtin_amdgcn_ubfe(v, 2u * ((uint32_t)p0 & 3u), 8u)];
                        } else {
                            const uint32_t v = gload32(gseq + (reverse ? A - pp - 3u : A + pp));
                            const uint32_t rlo = __builtin_amdgcn_perm(0u, v, rsel & 0x03030303u);
                            wlo = rlo ^ rcm;
                            n_here = (v & 0xfcfcfcfcu) != 0;
                            nmlo = rlo & 0xfcfcfcfcu;
                        }
                        any_n = __builtin_amdgcn_ballot_w64(n_here) != 0;
                    }
                }
                if (nquads) {
                    for (uint32_t qd = 0; qd < nquads; qd++) {
                        const uint32_t w = qd ? whi : wlo;
                        uint32_t cw = __builtin_amdgcn_perm(0u, 0x47414354u, w);      // four codes -> "TCAG" characters
                        uint32_t gq = 0;
                        if (!any_n) {
                            const uint32_t w8 = w << 3;
#pragma unroll
                            for (uint32_t j = 0; j < 4; j++) {
                                bool mism;
                                const uint32_t ch8 = qual_step((w8 >> (8u * j)) & 0xffu, op + j, rng(), mism);
                                gq |= j == 0 ? (ch8 >> 3) : (ch8 << (8u * j - 3u));
                                if (mism) cw = (cw & ~(0xffu << (8u * j))) | (mismatch_char((w >> (8u * j)) & 3u) << (8u * j));
                            }
                        } else {
                            // some lane has a non-TCAG base in this quad (the N runs of real assemblies put one into most
                            // waves): every lane makes the position's first draw; such a lane turns it into the quality
                            // of an 'N' (hts_illumina.h:237-242), the others go on with the alias step
                            const uint32_t nm = qd ? nmhi : nmlo;
                            const uint32_t w8 = (w & 0x03030303u) << 3;     // (a non-TCAG byte must not spill into its neighbour's code)
#pragma unroll
                            for (uint32_t j = 0; j < 4; j++) {
                                const uint64_t x1 = rng();
                                if ((nm >> (8u * j)) & 0xffu) {
                                    gq |= n_qual32(x1) << (8u * j);
                                    cw = (cw & ~(0xffu << (8u * j))) | ((uint32_t)'N' << (8u * j));
                                } else {
                                    bool mism;
                                    const uint32_t ch8 = qual_step((w8 >> (8u * j)) & 0xffu, op + j, x1, mism);
                                    gq |= j == 0 ? (ch8 >> 3) : (ch8 << (8u * j - 3u));
                                    if (mism) cw = (cw & ~(0xffu << (8u * j))) | (mismatch_char((w >> (8u * j)) & 3u) << (8u * j));
                                }
                            }
                        }
                        put_quad(cw, gq);
                        op += 4;
                    }
                    pp += 4u * nquads;
                    if (SEG && seg_moved) nes = seg_end_pp < nev ? seg_end_pp : nev;     // the gathering lanes' next general position moved with their segment
                    JK_BAL_STEP(12u * nquads);
                    continue;
                }
                // ---- one base the general way
                uint32_t c;
                if (pp < nes) {
                    c = src_byte(pp);
                    pp++;
                } else {
                    if (pending) {
                        c = pend_base; pending = false;
                    } else {
                        for (;;) {
                            const uint32_t w = pp >> 6, bit = pp & 63u;
                            const bool deleted = w < W && ((ev >> (DSH + w)) & 1u) && ((*evaddr(i, 1, w) >> bit) & 1ULL) &&
                                                 !(((ev >> w) & 1u) && ((*evaddr(i, 0, w) >> bit) & 1ULL));     // (both bits: a NUL insertion)
                            if (!deleted) break;
                            pp++;
                        }
                        c = (pp < bc) ? (uint32_t)bcode[pp] : src_byte(pp);
                        const uint32_t w = pp >> 6, bit = pp & 63u;
                        if (w < W && ((ev >> w) & 1u) && ((*evaddr(i, 0, w) >> bit) & 1ULL)) {
                            pending = true;
                            pend_base = (uint32_t)((*evaddr(i, 2, w) >> bit) & 1ULL) | ((uint32_t)((*evaddr(i, 3, w) >> bit) & 1ULL) << 1);
                            if (((ev >> (DSH + w)) & 1u) && ((*evaddr(i, 1, w) >> bit) & 1ULL)) pend_base = 4u;      // inserted NUL: not TCAG
                        }
                        pp++;
                    }
                    if (SEG) { while (pp >= seg_end_pp && pp < sp) seg_enter(pp); }
                    nes = pending ? 0u : (pp < bc ? pp : next_slow(pp));
                    if (SEG && (pending || pp < bc)) nev = nes;
                }
                uint32_t q, ch;
                if (c < 4u) {
                    bool mism;
                    q = qual_step(c << 3, op, rng(), mism) >> 3;
                    ch = base_char(c);
                    if (mism) ch = mismatch_char(c);
                } else {
                    q = n_qual32(rng());
                    ch = 'N';
                }
                const uint32_t bsh = 8u * (op & 3u);       // wave-uniform
                grp_b |= ch << bsh;
                grp_q |= q << bsh;
                if ((op & 3u) == 3u) { put_quad(grp_b, grp_q); grp_b = 0; grp_q = 0; }
                op++;
            }
            // the whole quads are out; the 0..3 bases of the last, partial one go through the byte appender
            o.acc = acc_b; o.pos += n_out & ~3u; os_put_n(o, col, grp_b, n_out & 3u);
            oq.acc = acc_q; oq.pos += n_out & ~3u; os_put_n(oq, col, grp_q, n_out & 3u);
            os_put(oq, col, '\n');
            os_flush(o, col);          // after every word store of oq above (see OutStream)
            o = oq;                    // the next record continues where the quality stream stopped
            reverse = !reverse;
        }
        // quota bookkeeping: reference genome hts_illumina.cpp:404-406; haplotypes :532-533 (one_read)
        // and :554-555 (re_read) -- the same "decrement, never below zero" for 1 or 2 ends
        ccnt = (ccnt < NE) ? 0 : ccnt - NE;

        // ---- ReadWriterOneThread::create_reads tail (src/hts.h:263-278)
        made += NE; in_pool += NE;
        const uint64_t xd = rng();
        const bool dup = P.dup_all || xd < P.th_dup;
        // the duplicate loop only continues while the pool has room and the quota is not met; a
        // full pool (or met quota) is flushed, which on the GPU only resets the counter
        if (dup && made < quota && in_pool < pool_size) {
            is_dup = true;
        } else {
            is_dup = false;
            if (in_pool >= pool_size || made >= quota) in_pool = 0;
        }
    }

#ifndef JK_NO_PRIO_BALANCE
    if (bal_on) s_prog[bal_simd * 4u + bal_k] = 0xffffffffu;
#endif
#pragma unroll
    for (uint32_t i = 0; i < NE; i++) {
        os_flush(os[i], P.pool[i] + tile_off + colb);
        const uint64_t nbytes = os[i].pos;
        P.lane_bytes[i][lane] = nbytes;
        if (nbytes > lane_cap) err |= JK_KERR_POOL_OVERFLOW;
    }
    P.lane_made[lane] = made;
    if (err) atomicOr(P.err, err);
#ifdef JK_TIMELINE
    if ((threadIdx.x & 63u) == 0 && (lane >> 6) < JK_TIMELINE_WAVES) {
        uint64_t* t = g_timeline + 4u * (lane >> 6);
        t[0] = tl_t0; t[1] = wall_clock64();
        t[2] = __builtin_amdgcn_s_getreg((31 << 11) | 4); t[3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
}

// ASCII -> code, in place (run once per uploaded buffer): T,C,A,G -> 0..3 (jlp::bases order, complement =
// code ^ 2); every other byte keeps its value (>= 4), which is all the Illumina path needs to know
// ("not TCAG") and what the PacBio path copies through verbatim.  Input bytes 0..3 -- 0 is what the
// reference's FASTA reader turns every non-TCAGN character into (src/str_manip.h:24-56) -- move to
// 0xfc..0xff (jk_decode_other undoes it); input bytes 0xfc..0xff cannot be represented.
constexpr uint32_t JK_LOW_BYTE_BASE = 0xfcu;
__device__ __forceinline__ uint32_t jk_decode_other(uint32_t code) {       // code >= 4
    return code >= JK_LOW_BYTE_BASE ? code - JK_LOW_BYTE_BASE : code;
}
__global__ void encode_bases_kernel(uint8_t* seq, uint64_t n, uint32_t* bad) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = i; k < n; k += stride) {
        const uint8_t c = seq[k];
        if (c >= JK_LOW_BYTE_BASE) *bad = 1;
        seq[k] = c == 'T' ? 0 : c == 'C' ? 1 : c == 'A' ? 2 : c == 'G' ? 3 : (c < 4 ? (uint8_t)(JK_LOW_BYTE_BASE + c) : c);
    }
}

// The 2-bit copy of an encoded buffer and its "not only TCAG" flags (GenomeDev::packed / nflags).  One thread per 16
// bases (one output dword), one 256-thread workgroup per 4096 bases = one flag bit (kept coarse so that the flags of a
// 3 Gbp genome, 92 KB, stay in L2: a flag per 64 bases cost every read end a cache line of its own).  Only bytes inside
// a chromosome (cell) count: the padding between them is 'N'.  nflags is zeroed by the caller.
__global__ void __launch_bounds__(256) pack_reference_kernel(const uint8_t* seq, uint64_t n, uint32_t* packed, uint32_t* nflags,
                                                             const uint64_t* cell_off, const uint64_t* cell_len, uint32_t n_cells) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;      // thread = 16 bases
    const uint64_t i0 = t * 16;
    uint32_t w = 0, bad = 0;
    if (i0 < n) {
        // the cell this group of 16 lies in (cells start at multiples of 64: a group never spans two)
        uint32_t lo = 0, hi = n_cells;
        while (hi - lo > 1u) { const uint32_t m = (lo + hi) >> 1; if (cell_off[m] <= i0) lo = m; else hi = m; }
        uint64_t in_lo = 0, in_hi = 0;
        if (n_cells && cell_off[lo] <= i0 + 15) { in_lo = cell_off[lo]; in_hi = cell_off[lo] + cell_len[lo]; }
        for (uint32_t k = 0; k < 16 && i0 + k < n; k++) {
            const uint32_t c = seq[i0 + k];
            w |= (c & 3u) << (2u * k);
            if (c > 3u && i0 + k >= in_lo && i0 + k < in_hi) bad = 1;
        }
        packed[t] = w;
    }
    if (__syncthreads_or((int)bad) && threadIdx.x == 0) atomicOr(&nflags[blockIdx.x >> 5], 1u << (blockIdx.x & 31u));
}

// ---------------------------------------------------------------------------------------------
// Pool compaction: word-interleaved tiles -> lane-major FASTQ image.  One 256-thread workgroup per
// 64-lane tile.  Per step it moves 32 words (128 B) of every lane: rows of the tile are read
// coalesced into LDS (row stride 65 words: the transposed reads below are then conflict-free), and
// each lane's 128 bytes are written by 8 consecutive threads as 16-byte pieces at the lane's final
// offset (arbitrary alignment; unaligned dwordx4 stores are legal on gfx950 global memory).
// ---------------------------------------------------------------------------------------------
// Step size and LDS buffering, measured on the headline workload (one launch = 0.82 GB in, 0.82 GB out; alone, after
// the last generator): 32 rows double-buffered 0.60 ms, 64 rows 0.44 (double) / 0.48 (single), 128 rows single 0.40,
// 128 double / 256 single 0.36-0.38.  The last two need 66 KB of LDS per workgroup and do not fit next to a generator
// workgroup (124 KB) -- which costs nothing: the generator fills the whole register file of every SIMD it runs on and,
// with its waves balanced, leaves no tail to run in, so the compaction of a batch runs between two generator launches
// either way (whole step 14.6 ms with 128 double, 14.9 ms with 128 single).
#ifndef JK_CP_ROWS
#define JK_CP_ROWS 128
#endif
#ifndef JK_CP_DB
#define JK_CP_DB 1
#endif
constexpr int CP_ROWS = JK_CP_ROWS;             // words of every lane moved per step (512 bytes per lane)
constexpr size_t CP_SLACK = (size_t)CP_ROWS * 256;   // a pool buffer is allocated this much longer: the last step of a tile loads whole rows
#ifdef JK_CP_WAVES
#define JK_CP_ATTR __attribute__((amdgpu_waves_per_eu(JK_CP_WAVES, JK_CP_WAVES)))
#else
#define JK_CP_ATTR
#endif
#ifndef JK_CP_THREADS
#define JK_CP_THREADS 256
#endif
constexpr uint32_t CP_THREADS = JK_CP_THREADS;
__global__ void __launch_bounds__(JK_CP_THREADS) JK_CP_ATTR
compact_pools_kernel(const uint8_t* __restrict__ pool, const uint64_t* __restrict__ pool_off,
                     const uint64_t* __restrict__ lane_bytes, const uint64_t* __restrict__ out_off,
                     uint8_t* __restrict__ out, const uint64_t* __restrict__ out_base, uint32_t n_lanes) {
    __shared__ uint32_t tile_lds[JK_CP_DB ? 2 : 1][CP_ROWS * 65];
    __shared__ uint32_t s_max;
    // These waves share SIMDs with the generator of the next batch, which is older and saturates the VALU issue
    // port: at default priority they get the left-over slots only.  They are few and memory-bound: let them issue first.
#ifndef JK_CP_NOPRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    const uint32_t tile = blockIdx.x, t = threadIdx.x;
    const uint32_t lane0 = tile * 64u;
    if (t == 0) s_max = 0;
    __syncthreads();
    if (t < 64 && lane0 + t < n_lanes) atomicMax(&s_max, (uint32_t)lane_bytes[lane0 + t]);
    __syncthreads();
    const uint32_t max_bytes = s_max;
    // thread -> (lane, 16-byte piece) for the write side: PIECES consecutive threads cover one lane's bytes of a step
    constexpr uint32_t PIECES = CP_ROWS / 4;                 // 16-byte pieces per lane and step
    constexpr uint32_t LANES_PER_PASS = CP_THREADS / PIECES;
    constexpr uint32_t PASSES = 64 / LANES_PER_PASS;
    const uint32_t j = t % PIECES;
    uint32_t nb[PASSES]; uint8_t* dst[PASSES];
    const uint64_t base = out_base[0];
#pragma unroll
    for (uint32_t pass = 0; pass < PASSES; pass++) {
        const uint32_t lane = lane0 + pass * LANES_PER_PASS + t / PIECES;
        nb[pass] = lane < n_lanes ? (uint32_t)lane_bytes[lane] : 0u;
        dst[pass] = out + base + (lane < n_lanes ? out_off[lane] : 0ULL) + j * 16u;
#ifdef JK_CP_ALIGN_EXPERIMENT
        dst[pass] = reinterpret_cast<uint8_t*>(reinterpret_cast<uintptr_t>(dst[pass]) & ~(uintptr_t)15);   // timing experiment only: wrong output
#endif
    }
    // read side: a step is CP_ROWS rows of 256 bytes, contiguous in the tile: 16 bytes per thread and load, all loads of a
    // step in flight together and the next step's loads issued before this step's stores
    constexpr uint32_t LOADS = (CP_ROWS * 16) / CP_THREADS;
    const uint4* src = reinterpret_cast<const uint4*>(pool + pool_off[tile]);
    // Loads run two steps ahead of the stores (two register sets, the loop is unrolled by two so that they swap roles
    // without moves): a tile-copy on a handful of CUs -- the compaction runs on the CUs a generator launch leaves free,
    // see plan_pools_and_alloc -- is bound by the bytes it keeps in flight per CU, and LDS limits the workgroups per CU to two.
    uint4 ra[LOADS], rb[LOADS];
    auto load_step = [&](uint4* r, uint32_t k0) {
        if (k0 * 4u < max_bytes) {
#pragma unroll
            for (uint32_t i = 0; i < LOADS; i++) r[i] = src[(size_t)k0 * 16u + t + i * CP_THREADS];
        }
    };
    auto do_step = [&](const uint4* r, uint32_t k0, uint32_t buf) {
        uint32_t* L = tile_lds[buf];
        if (!JK_CP_DB && k0) __syncthreads();
#pragma unroll
        for (uint32_t i = 0; i < LOADS; i++) {
            const uint32_t idx = t + i * CP_THREADS, row = idx >> 4, l4 = (idx & 15u) * 4u;
            uint32_t* w = L + row * 65u + l4;
            w[0] = r[i].x; w[1] = r[i].y; w[2] = r[i].z; w[3] = r[i].w;
        }
        __syncthreads();       // (one barrier per step: the other buffer is rewritten only after the next barrier)
    };
    auto store_step = [&](uint32_t k0, uint32_t buf) {
        const uint32_t* L = tile_lds[buf];
#pragma unroll
        for (uint32_t pass = 0; pass < PASSES; pass++) {
            const uint32_t l = pass * LANES_PER_PASS + t / PIECES;
            const uint32_t b0 = k0 * 4u + j * 16u;          // first byte of this piece in the lane's stream
            if (b0 < nb[pass]) {
                uint32_t v[4];
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) v[i] = L[(j * 4u + i) * 65u + l];
                uint8_t* d = dst[pass] + (size_t)k0 * 4u;
                if (b0 + 16u <= nb[pass]) {
                    __builtin_memcpy(d, v, 16);
                } else {
                    for (uint32_t b = 0; b < nb[pass] - b0; b++) d[b] = (uint8_t)(v[b >> 2] >> ((b & 3u) * 8u));
                }
            }
        }
    };
    load_step(ra, 0);
    load_step(rb, CP_ROWS);
    for (uint32_t k0 = 0; k0 * 4u < max_bytes; k0 += 2 * CP_ROWS) {
        do_step(ra, k0, 0);
        load_step(ra, k0 + 2 * CP_ROWS);
        store_step(k0, 0);
        if ((k0 + CP_ROWS) * 4u >= max_bytes) break;
        do_step(rb, k0 + CP_ROWS, JK_CP_DB ? 1u : 0u);
        load_step(rb, k0 + 3 * CP_ROWS);
        store_step(k0 + CP_ROWS, JK_CP_DB ? 1u : 0u);
    }
}

// Pool compaction for per-lane contiguous regions (PacBio): one 256-thread workgroup copies one lane's
// text (16-byte aligned source, arbitrary destination alignment) in 16-byte pieces.
__global__ void __launch_bounds__(256)
compact_linear_kernel(const uint8_t* __restrict__ pool, const uint64_t* __restrict__ pool_off,
                      const uint64_t* __restrict__ lane_bytes, const uint64_t* __restrict__ out_off,
                      uint8_t* __restrict__ out, const uint64_t* __restrict__ out_base, uint32_t n_lanes,
                      uint64_t out_capacity, uint32_t* __restrict__ err) {
    const uint32_t lane = blockIdx.x;
    if (lane >= n_lanes) return;
    const uint32_t tile = lane >> 6;
    const uint64_t cap = (pool_off[tile + 1] - pool_off[tile]) >> 6;
    const uint8_t* src = pool + pool_off[tile] + (uint64_t)(lane & 63u) * cap;
    const uint64_t n = lane_bytes[lane], n16 = n >> 4;
    // the image is allocated for the expected size, not for the pools' worst case: never write past it
    if (out_base[0] + out_off[lane] + n > out_capacity) { if (threadIdx.x == 0) atomicOr(err, JK_KERR_IMAGE_FULL); return; }
    uint8_t* dst = out + out_base[0] + out_off[lane];
    for (uint64_t c = threadIdx.x; c < n16; c += 256) {
        const uint4 v = *reinterpret_cast<const uint4*>(src + c * 16);
        __builtin_memcpy(dst + c * 16, &v, 16);
    }
    const uint64_t tail = n16 << 4;
    if (tail + threadIdx.x < n) dst[tail + threadIdx.x] = src[tail + threadIdx.x];
}

// ---------------------------------------------------------------------------------------------
// Exclusive scan of per-lane byte counts (u64), three small kernels.
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_BLOCK = 256;     // small blocks co-reside with the generator (which fills 16 of 32 wave slots per CU)

__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t* total) {
    __shared__ uint64_t wsum[SCAN_BLOCK / 64];
    const uint32_t lid = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    uint64_t incl = v;
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t o = __shfl_up(incl, d, 64);
        if ((int)lid >= d) incl += o;
    }
    if (lid == 63) wsum[wid] = incl;
    __syncthreads();
    if (wid == 0) {
        uint64_t s = lid < SCAN_BLOCK / 64 ? wsum[lid] : 0, si = s;
        for (int d = 1; d < SCAN_BLOCK / 64; d <<= 1) {
            uint64_t o = __shfl_up(si, d, 64);
            if ((int)lid >= d) si += o;
        }
        if (lid < SCAN_BLOCK / 64) wsum[lid] = si - s;     // exclusive prefix of wave sums
        if (lid == SCAN_BLOCK / 64 - 1) *total = si;
    }
    __syncthreads();
    const uint64_t r = wsum[wid] + incl - v;
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(SCAN_BLOCK) scan_block_kernel(const uint64_t* in, uint64_t* out, uint64_t* block_sums, uint32_t n) {
    __shared__ uint64_t total;
    __builtin_amdgcn_s_setprio(3);      // see compact_pools_kernel
    const uint32_t i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
    const uint64_t v = i < n ? in[i] : 0;
    const uint64_t ex = block_exclusive_scan(v, &total);
    if (i < n) out[i] = ex;
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
// single block: exclusive scan of block sums in place; base[1] = base[0] + grand total (the running
// byte offset of the next batch in the FASTQ image, kept on the device so no host sync is needed)
__global__ void __launch_bounds__(SCAN_BLOCK) scan_sums_kernel(uint64_t* block_sums, uint32_t nb, uint64_t* base) {
    __shared__ uint64_t total;
    __shared__ uint64_t carry;
    __builtin_amdgcn_s_setprio(3);
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < nb; base += SCAN_BLOCK) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < nb ? block_sums[i] : 0;
        const uint64_t ex = block_exclusive_scan(v, &total);
        if (i < nb) block_sums[i] = ex + carry;
        __syncthreads();
        if (threadIdx.x == 0) carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) base[1] = base[0] + carry;
}
__global__ void __launch_bounds__(SCAN_BLOCK) scan_add_kernel(uint64_t* out, const uint64_t* block_sums, uint32_t n) {
    __builtin_amdgcn_s_setprio(3);
    const uint32_t i = blockIdx.x * SCAN_BLOCK + threadIdx.x;
    if (i < n) out[i] += block_sums[blockIdx.x];
}

// End of a generate(): everything the host wants to know in one 32-byte record {error bits, bytes of end 0, bytes of
// end 1, reads made}, so that the step ends with one small copy instead of three plus the whole per-lane array.
__global__ void __launch_bounds__(256)
finish_kernel(const uint64_t* __restrict__ lane_made, uint64_t n_lanes, const uint32_t* __restrict__ err,
              const uint64_t* __restrict__ total0, const uint64_t* __restrict__ total1, uint64_t* __restrict__ out) {
    __shared__ uint64_t part[4];
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_lanes; i += (uint64_t)gridDim.x * 256) acc += lane_made[i];
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0) {
            out[0] = err[0];
            out[1] = total0[0];
            out[2] = total1 ? total1[0] : 0;
        }
        atomicAdd(reinterpret_cast<unsigned long long*>(out + 3), (unsigned long long)(part[0] + part[1] + part[2] + part[3]));     // out[3] zeroed by the host
    }
}

}  // namespace jk
