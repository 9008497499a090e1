// BGZF (blocked gzip) compression of a FASTQ image that is already in HBM.
//
// Replaces the reference's compressed sinks for comp_method = "bgzip": FileBGZF (src/io.h:150-236) and
// bgzip_file (src/hts.h:140-180), which hand <= 0xff00-byte pieces to htslib's bgzf_write.  The block
// boundaries are the same; the DEFLATE payload is this kernel's own: one dynamic-Huffman block of
// literals only (no LZ77 matches), or a stored block when that would be larger.  Any inflate
// implementation restores the identical bytes, which is the parity criterion for a compressed sink
// (compressed bytes of the reference depend on the zlib/htslib build it is linked with).
//
// One 1024-thread workgroup per BGZF block; thread t owns input bytes [64t, 64t+64) and keeps them in
// registers from the histogram to the encoding pass, so the input is read from HBM once.
//   1. byte histogram (LDS atomics into 64 per-lane-index sub-histograms, so that equal bytes in a wave never
//      meet at one address or bank -- FASTQ has 4-40 distinct symbols) and CRC-32 of each 64-byte piece
//   2. CRC-32 of the block = XOR of piece CRCs multiplied by x^(8 * bytes that follow) mod P
//   3. Huffman code lengths: rank sort of the used symbols, two-queue tree build (one thread),
//      leaf depths in parallel, limit to 15 bits, canonical codes
//   4. bit counts per thread, workgroup scan, bits OR-ed into the block image in LDS
//   5. image (18-byte BGZF header, payload, CRC-32, ISIZE) copied to the block's slot in HBM
// A second kernel gathers the slots into the contiguous file image at offsets from a scan of the sizes.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace jk {

constexpr uint32_t BGZF_BLOCK_IN = 0xff00;       // input bytes per block, htslib's BGZF_BLOCK_SIZE
constexpr uint32_t BGZF_SLOT = 65536;            // a block is at most 64 KiB (BSIZE is 16 bits)
constexpr uint32_t BGZF_THREADS = 1024;
constexpr uint32_t BGZF_PIECE = 64;              // input bytes per thread
constexpr uint32_t BGZF_NSYM = 257;              // literals 0..255 and end-of-block
constexpr uint32_t BGZF_IMG_WORDS = 16400;       // LDS image of one output block
constexpr uint32_t BGZF_HDR_BYTES = 18;
constexpr uint32_t BGZF_DYN_HDR_BITS = 17 + 57 + 4 * (BGZF_NSYM + 1);   // 1106
constexpr uint32_t CRC_POLY = 0xedb88320u;       // reflected CRC-32 polynomial (bit 31 = x^0)

// a * b mod P in the reflected representation (what zlib calls multmodp)
__host__ __device__ inline uint32_t crc_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) {
        if (a & 0x80000000u) p ^= b;
        a <<= 1;
        b = (b >> 1) ^ ((b & 1u) ? CRC_POLY : 0u);
    }
    return p;
}

struct BgzfTables {
    const uint32_t* crc_tab;     // [4][256] slicing-by-4 CRC tables ([0] = the byte-wise table)
    const uint32_t* x512;        // [1024] x^(512 j) mod P
    const uint32_t* x8;          // [64]   x^(8 r) mod P
};

__device__ __forceinline__ void img_or_bits(uint32_t* img, uint32_t bitpos, uint64_t bits) {
    // bits occupy at most 32 positions starting at bitpos (callers keep chunks <= 32 bits)
    const uint32_t w = bitpos >> 5, sh = bitpos & 31u;
    const uint64_t v = bits << sh;
    atomicOr(&img[w], (uint32_t)v);
    const uint32_t hi = (uint32_t)(v >> 32);
    if (hi) atomicOr(&img[w + 1], hi);
}
__device__ __forceinline__ void img_or_byte(uint32_t* img, uint32_t bytepos, uint32_t byte) {
    atomicOr(&img[bytepos >> 2], byte << (8u * (bytepos & 3u)));
}

__global__ void __launch_bounds__(BGZF_THREADS, 8)     // 8 waves/SIMD: two workgroups per CU (64 VGPRs, ~73 KB LDS each)
bgzf_deflate_kernel(const uint8_t* __restrict__ src, uint64_t n_total, uint8_t* __restrict__ slots,
                    uint64_t* __restrict__ sizes, BgzfTables T) {
    __shared__ uint32_t img[BGZF_IMG_WORDS];
    __shared__ uint32_t s_crc_tab[4 * 256];
    __shared__ uint64_t used_key[BGZF_NSYM];      // (freq << 16 | symbol) of the used symbols, unordered
    __shared__ uint16_t sorted[BGZF_NSYM];        // used symbols by ascending (freq, symbol)
    __shared__ uint8_t sorted_len[BGZF_NSYM + 3];
    __shared__ uint32_t node_w[2 * BGZF_NSYM];    // leaves 0..nz-1 (sorted), internal nodes nz..2nz-2
    __shared__ uint16_t node_par[2 * BGZF_NSYM];
    __shared__ uint32_t depth_cnt[32];            // leaves per depth (depths above 31 counted at 31)
    __shared__ uint32_t len_cnt[16];
    __shared__ uint32_t next_code[16];
    __shared__ uint8_t sym_len[BGZF_NSYM + 3];
    __shared__ uint32_t enc[BGZF_NSYM];           // reversed code | len << 16
    __shared__ uint32_t wave_part[BGZF_THREADS / 64];
    __shared__ uint32_t s_nz, s_crc, s_total_bits, s_last_crc;

    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint64_t blk_off = (uint64_t)blockIdx.x * BGZF_BLOCK_IN;
    const uint32_t n = (uint32_t)((n_total - blk_off) < BGZF_BLOCK_IN ? (n_total - blk_off) : BGZF_BLOCK_IN);
    const uint32_t full = n / BGZF_PIECE, rem = n % BGZF_PIECE;      // full pieces, bytes of the partial one

    // sub-histograms live in the image buffer until the image is needed: sub[l][c >> 1] holds the counts of
    // bytes c (two 16-bit halves; a sub-histogram sees at most 16 waves x 64 bytes) met by lanes with index l.
    // Row stride 129 words: lanes with equal bytes fall into different banks.
    constexpr uint32_t SUB_STRIDE = 129;
    uint32_t* const sub = img;
    for (uint32_t i = t; i < 64 * SUB_STRIDE; i += BGZF_THREADS) sub[i] = 0;
    s_crc_tab[t] = T.crc_tab[t];
    if (t < 32) depth_cnt[t] = 0;
    if (t < 16) len_cnt[t] = 0;
    if (t == 0) { s_nz = 0; s_crc = 0; s_last_crc = 0; }

    // ---- this thread's piece, kept in registers
    uint32_t piece[BGZF_PIECE / 4];
    const uint32_t my_n = t < full ? BGZF_PIECE : (t == full ? rem : 0u);
    {
        const uint8_t* p = src + blk_off + (uint64_t)t * BGZF_PIECE;
        if (my_n == BGZF_PIECE) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 v = reinterpret_cast<const uint4*>(p)[q];
                piece[4 * q] = v.x; piece[4 * q + 1] = v.y; piece[4 * q + 2] = v.z; piece[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int w = 0; w < (int)(BGZF_PIECE / 4); w++) {
                uint32_t v = 0;
                for (uint32_t b = 0; b < 4; b++)
                    if (4u * w + b < my_n) v |= (uint32_t)p[4 * w + b] << (8u * b);
                piece[w] = v;
            }
        }
    }
    __syncthreads();

    // ---- 1. histogram + CRC-32 of the piece
    uint32_t crc = 0xffffffffu;
    uint32_t* const my_sub = sub + lane * SUB_STRIDE;
    if (my_n == BGZF_PIECE) {
#pragma unroll
        for (int w = 0; w < (int)(BGZF_PIECE / 4); w++) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const uint32_t c = (piece[w] >> (8 * b)) & 0xffu;
                atomicAdd(&my_sub[c >> 1], 1u << (16u * (c & 1u)));
            }
            const uint32_t x = crc ^ piece[w];               // slicing by 4: four independent lookups per word
            crc = s_crc_tab[768 + (x & 0xffu)] ^ s_crc_tab[512 + ((x >> 8) & 0xffu)] ^
                  s_crc_tab[256 + ((x >> 16) & 0xffu)] ^ s_crc_tab[x >> 24];
        }
    } else {
#pragma unroll
        for (int w = 0; w < (int)(BGZF_PIECE / 4); w++) {       // (static register indices: no scratch)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if (4u * w + b < my_n) {
                    const uint32_t c = (piece[w] >> (8 * b)) & 0xffu;
                    atomicAdd(&my_sub[c >> 1], 1u << (16u * (c & 1u)));
                    crc = s_crc_tab[(crc ^ c) & 0xffu] ^ (crc >> 8);
                }
            }
        }
    }
    crc = ~crc;
    // ---- 2. combine: full pieces are shifted over the bytes that follow them
    {
        uint32_t part = 0;
        if (t < full) part = crc_mulmod(crc, T.x512[full - 1 - t]);
        for (int d = 32; d > 0; d >>= 1) part ^= __shfl_xor(part, d, 64);
        if (lane == 0) wave_part[wave] = part;
        if (t == full && rem) s_last_crc = crc;
    }
    __syncthreads();
    if (t == 0) {
        uint32_t s = 0;
        for (uint32_t w = 0; w < BGZF_THREADS / 64; w++) s ^= wave_part[w];
        s_crc = rem ? (crc_mulmod(s, T.x8[rem]) ^ s_last_crc) : s;
    }
    if (t < BGZF_NSYM) {
        uint32_t f = 1u;
        if (t < 256) {
            f = 0;
            for (uint32_t l = 0; l < 64; l++) f += (sub[l * SUB_STRIDE + (t >> 1)] >> (16u * (t & 1u))) & 0xffffu;
        }
        sym_len[t] = 0;
        if (f) {                                  // used symbols, in no particular order
            const uint32_t slot = atomicAdd(&s_nz, 1u);
            used_key[slot] = ((uint64_t)f << 16) | t;          // (frequency, symbol): distinct keys
        }
    }
    __syncthreads();
    for (uint32_t i = t; i < BGZF_IMG_WORDS; i += BGZF_THREADS) img[i] = 0;       // from here on the buffer is the image

    // ---- 3. code lengths.  FASTQ uses 5-45 byte values, so everything below loops over the used symbols only.
    const uint32_t nz = s_nz;                    // >= 2: at least one literal and end-of-block
    if (t < nz) {
        const uint64_t key = used_key[t];
        uint32_t rank = 0;
        for (uint32_t u = 0; u < nz; u++) rank += used_key[u] < key ? 1u : 0u;
        sorted[rank] = (uint16_t)(key & 0xffffu);
        node_w[rank] = (uint32_t)(key >> 16);
    }
    __syncthreads();
    if (t == 0) {
        // two queues: leaves in ascending weight, internal nodes in creation order (also ascending).
        // The heads of both queues are kept in registers; each take loads its successor.
        const uint32_t INF = 0xffffffffu;
        uint32_t leaf = 0, inner = nz, made = nz;
        uint32_t w_leaf = node_w[0], w_inner = INF;
        for (uint32_t k = 0; k + 1 < nz; k++) {
            uint32_t sum = 0;
            for (int j = 0; j < 2; j++) {
                if (w_leaf <= w_inner) {        // (an exhausted queue shows INF; both are never exhausted here)
                    sum += w_leaf; node_par[leaf] = (uint16_t)made; leaf++;
                    w_leaf = leaf < nz ? node_w[leaf] : INF;
                } else {
                    sum += w_inner; node_par[inner] = (uint16_t)made; inner++;
                    w_inner = inner < made ? node_w[inner] : INF;
                }
            }
            node_w[made] = sum;
            if (w_inner == INF && inner == made) w_inner = sum;      // the new node is the inner queue's head
            made++;
        }
    }
    __syncthreads();
    if (t < nz) {
        const uint32_t root = 2 * nz - 2;
        uint32_t d = 0, v = t;
        while (v != root && d < 2 * BGZF_NSYM) { v = node_par[v]; d++; }     // (bounded: every wave must leave)
        atomicAdd(&depth_cnt[d < 31u ? d : 31u], 1u);
    }
    __syncthreads();
    if (t == 0) {
        // limit to 15 bits, keeping the code complete (the counts-per-length repair used by miniz/zlib-style
        // encoders: fold deeper leaves into 15, then move leaves down until the Kraft sum is exact)
        for (int i = 16; i < 32; i++) depth_cnt[15] += depth_cnt[i];
        uint32_t total = 0;
        for (int i = 15; i >= 1; i--) total += depth_cnt[i] << (15 - i);
        for (uint32_t guard = 0; total != (1u << 15) && guard < (1u << 16); guard++) {
            depth_cnt[15]--;
            for (int i = 14; i >= 1; i--)
                if (depth_cnt[i]) { depth_cnt[i]--; depth_cnt[i + 1] += 2; break; }
            total--;
        }
        uint32_t code = 0, prev = 0;
        len_cnt[0] = 0;
        for (int i = 1; i < 16; i++) {
            code = (code + prev) << 1;
            next_code[i] = code;
            prev = depth_cnt[i];
            len_cnt[i] = prev;
        }
    }
    __syncthreads();
    if (t < nz) {
        // the q-th most frequent symbol gets the q-th shortest length
        const uint32_t q = nz - 1 - t;
        uint32_t acc = 0, len = 15;
        for (uint32_t i = 1; i < 16; i++) {
            acc += len_cnt[i];
            if (q < acc) { len = i; break; }
        }
        sorted_len[t] = (uint8_t)len;
        sym_len[sorted[t]] = (uint8_t)len;
    }
    if (t < BGZF_NSYM) enc[t] = 0;
    __syncthreads();
    if (t < nz) {
        // canonical code: symbols of one length are numbered in symbol order
        const uint32_t sym = sorted[t], len = sorted_len[t];
        uint32_t before = 0;
        for (uint32_t u = 0; u < nz; u++) before += (sorted_len[u] == len && sorted[u] < sym) ? 1u : 0u;
        const uint32_t code = next_code[len] + before;
        enc[sym] = (__brev(code) >> (32u - len)) | (len << 16);
    }
    __syncthreads();

    // ---- 4. bit counts and scan
    uint32_t my_bits = 0;
    if (my_n == BGZF_PIECE) {
#pragma unroll
        for (int w = 0; w < (int)(BGZF_PIECE / 4); w++) {
#pragma unroll
            for (int b = 0; b < 4; b++) my_bits += enc[(piece[w] >> (8 * b)) & 0xffu] >> 16;
        }
    } else {
#pragma unroll
        for (int w = 0; w < (int)(BGZF_PIECE / 4); w++) {
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (4u * w + b < my_n) my_bits += enc[(piece[w] >> (8 * b)) & 0xffu] >> 16;
        }
    }
    const uint32_t t_last = (n - 1) / BGZF_PIECE;          // owner of the last byte appends end-of-block
    if (t == t_last) my_bits += enc[256] >> 16;
    uint32_t incl = my_bits;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += o;
    }
    if (lane == 63) wave_part[wave] = incl;
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (uint32_t w = 0; w < BGZF_THREADS / 64; w++) { const uint32_t s = wave_part[w]; wave_part[w] = run; run += s; }
        s_total_bits = run + BGZF_DYN_HDR_BITS;
    }
    __syncthreads();
    const uint32_t payload_dyn = (s_total_bits + 7u) >> 3;
    const bool stored = payload_dyn > n + 5u;
    const uint32_t payload = stored ? n + 5u : payload_dyn;
    const uint32_t total = BGZF_HDR_BYTES + payload + 8u;
    const uint32_t base_bit = BGZF_HDR_BYTES * 8u;

    if (!stored) {
        if (t == 0) {
            // BFINAL=1, BTYPE=2, HLIT=0 (257 codes), HDIST=0 (1 code), HCLEN=15 (19 lengths); the code-length
            // code gives symbols 0..15 four bits each (16,17,18 unused), so its canonical codes are the symbols
            img_or_bits(img, base_bit, 1u | (2u << 1) | (15u << 13));
            uint64_t cl = 0;                                         // 16 fields of 3 bits, all = 4
            for (int i = 0; i < 16; i++) cl |= 4ull << (3 * i);
            img_or_bits(img, base_bit + 26, cl & 0xffffffu);
            img_or_bits(img, base_bit + 50, cl >> 24);
        }
        if (t <= BGZF_NSYM) {                                        // 257 literal/length lengths + 1 distance length (0)
            const uint32_t len = t < BGZF_NSYM ? sym_len[t] : 0u;
            img_or_bits(img, base_bit + 74 + 4 * t, __brev(len) >> 28);
        }
        // The thread's bits start at bit `pos`; the register starts with the (pos & 31) zero bits below them, so
        // every flush is one whole word of the image.  Two symbols add at most 30 bits to fewer than 32
        // pending ones, so the 64-bit register is checked once per pair.
        const uint32_t pos = base_bit + BGZF_DYN_HDR_BITS + wave_part[wave] + (incl - my_bits);
        uint32_t* wp = img + (pos >> 5);
        uint64_t acc = 0; uint32_t nacc = pos & 31u;
        if (my_n == BGZF_PIECE) {
#pragma unroll
            for (int w = 0; w < (int)(BGZF_PIECE / 4); w++) {
#pragma unroll
                for (int b = 0; b < 4; b += 2) {
                    const uint32_t e0 = enc[(piece[w] >> (8 * b)) & 0xffu];
                    const uint32_t e1 = enc[(piece[w] >> (8 * b + 8)) & 0xffu];
                    acc |= (uint64_t)(e0 & 0xffffu) << nacc;
                    nacc += e0 >> 16;
                    acc |= (uint64_t)(e1 & 0xffffu) << nacc;
                    nacc += e1 >> 16;
                    if (nacc >= 32u) { atomicOr(wp++, (uint32_t)acc); acc >>= 32; nacc -= 32; }
                }
            }
        } else {
#pragma unroll
            for (int w = 0; w < (int)(BGZF_PIECE / 4); w++) {
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    if (4u * w + b < my_n) {
                        const uint32_t e = enc[(piece[w] >> (8 * b)) & 0xffu];
                        acc |= (uint64_t)(e & 0xffffu) << nacc;
                        nacc += e >> 16;
                        if (nacc >= 32u) { atomicOr(wp++, (uint32_t)acc); acc >>= 32; nacc -= 32; }
                    }
                }
            }
        }
        if (t == t_last) {
            const uint32_t e = enc[256];
            acc |= (uint64_t)(e & 0xffffu) << nacc;
            nacc += e >> 16;
            if (nacc >= 32u) { atomicOr(wp++, (uint32_t)acc); acc >>= 32; nacc -= 32; }
        }
        if (nacc) atomicOr(wp, (uint32_t)acc);
    } else {
        if (t == 0) {
            img_or_byte(img, BGZF_HDR_BYTES, 1u);                    // BFINAL=1, BTYPE=0
            img_or_byte(img, BGZF_HDR_BYTES + 1, n & 0xffu);
            img_or_byte(img, BGZF_HDR_BYTES + 2, n >> 8);
            img_or_byte(img, BGZF_HDR_BYTES + 3, ~n & 0xffu);
            img_or_byte(img, BGZF_HDR_BYTES + 4, (~n >> 8) & 0xffu);
        }
        const uint32_t at = BGZF_HDR_BYTES + 5u + t * BGZF_PIECE;
#pragma unroll
        for (int w = 0; w < (int)(BGZF_PIECE / 4); w++) {
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (4u * w + b < my_n) img_or_byte(img, at + 4 * w + b, (piece[w] >> (8 * b)) & 0xffu);
        }
    }
    if (t == 0) {
        // gzip member header with the BGZF extra field 'B','C',2,0,BSIZE (total - 1)
        const uint32_t h[5] = {0x04088b1fu, 0x00000000u, 0x0006ff00u, 0x00024342u, (total - 1u) & 0xffffu};
        atomicOr(&img[0], h[0]); atomicOr(&img[1], h[1]); atomicOr(&img[2], h[2]); atomicOr(&img[3], h[3]);
        atomicOr(&img[4], h[4]);
        const uint32_t tr = BGZF_HDR_BYTES + payload;
        const uint32_t c = s_crc;
        for (uint32_t k = 0; k < 4; k++) {
            img_or_byte(img, tr + k, (c >> (8 * k)) & 0xffu);
            img_or_byte(img, tr + 4 + k, (n >> (8 * k)) & 0xffu);
        }
        sizes[blockIdx.x] = total;
    }
    __syncthreads();

    // ---- 5. image -> slot
    uint32_t* dst = reinterpret_cast<uint32_t*>(slots + (uint64_t)blockIdx.x * BGZF_SLOT);
    const uint32_t words = (total + 3u) >> 2;
    for (uint32_t i = t; i < words; i += BGZF_THREADS) dst[i] = img[i];
}

// ---------------------------------------------------------------------------------------------------------------
// The same with matches (LZ77), found the way FASTQ wants them found.
//
// One Huffman code has to serve three kinds of bytes -- id lines, bases, qualities -- and pays about a bit per byte for
// it (order-0 entropy of the headline FASTQ: 2.93 bits per byte; of its parts separately 2.1).  zlib's level 6 gets
// 0.320 out of hash chains and lazy matching; what it finds is mostly this: a record repeats the record before it --
// "@REF-chrom0-" ... "-R/1\n" in the same columns, "+\n", the occasional stretch of a quality line -- and qualities
// come in runs.  So every byte is compared with exactly two others: the byte at the same offset in the line FOUR LINES
// BACK (the same column of the previous record; the distance is the previous record's length, taken from a table of
// the block's line starts) and the byte before it (a run).  Maximal stretches of equal bytes of length >= 5 inside a
// thread's 64-byte piece become matches (length, distance); everything else stays a literal.  No hash table, no chains,
// no dependence between threads; measured on the headline FASTQ: 0.333 against 0.375 for literals only (PacBio reads,
// whose quality lines are two runs: 0.20 against 0.375).
// Block layout as above, with the second (distance) Huffman code built by the same routine.
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t LZ_NLIT = 286;                // literals, end-of-block, 29 length codes
constexpr uint32_t LZ_NDIST = 30;
constexpr uint32_t LZ_DYN_HDR_BITS = 17 + 57 + 4 * (LZ_NLIT + LZ_NDIST);      // 1338
constexpr uint32_t LZ_MIN_MATCH = 5;
constexpr uint32_t LZ_MAX_LINES = 2048;          // line starts remembered per block (more lines: runs only)
__device__ const uint16_t LZ_LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t LZ_LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t LZ_DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t LZ_DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__device__ __forceinline__ uint32_t lz_len_code(uint32_t len) {        // 3..258 -> 0..28
    if (len < 11u) return len - 3u;
    if (len == 258u) return 28u;
    const uint32_t v = len - 3u, e = 29u - (uint32_t)__builtin_clz(v);      // extra bits: floor(log2(v)) - 2
    return 4u * e + 4u + ((v >> e) & 3u);
}
__device__ __forceinline__ uint32_t lz_dist_code(uint32_t dist) {      // 1..32768 -> 0..29
    if (dist < 5u) return dist - 1u;
    const uint32_t v = dist - 1u, e = 30u - (uint32_t)__builtin_clz(v);      // extra bits: floor(log2(v)) - 1
    return 2u * e + 2u + ((v >> e) & 1u);
}

// Code lengths (<= 15 bits) and canonical codes of the `nz` used symbols listed in used_key[] as (frequency << 16 |
// symbol): the steps 3 of the kernel above as a routine of the whole workgroup.  Writes sym_len[symbol] and
// enc[symbol] = bit-reversed code | length << 16 for the used symbols (the caller has cleared both).
struct LzBuild {
    uint64_t* used_key; uint16_t* sorted; uint8_t* sorted_len; uint32_t* node_w; uint16_t* node_par;
    uint32_t* depth_cnt; uint32_t* len_cnt; uint32_t* next_code;
};
__device__ __forceinline__ void lz_build_codes(const LzBuild& B, uint32_t nz, uint8_t* sym_len, uint32_t* enc, uint32_t t) {
    if (t < 32) B.depth_cnt[t] = 0;
    if (t < 16) B.len_cnt[t] = 0;
    __syncthreads();
    if (nz == 1) {                                  // a single symbol still needs one bit
        if (t == 0) { const uint32_t sym = (uint32_t)(B.used_key[0] & 0xffffu); sym_len[sym] = 1; enc[sym] = 1u << 16; }
        __syncthreads();
        return;
    }
    if (t < nz) {
        const uint64_t key = B.used_key[t];
        uint32_t rank = 0;
        for (uint32_t u = 0; u < nz; u++) rank += B.used_key[u] < key ? 1u : 0u;
        B.sorted[rank] = (uint16_t)(key & 0xffffu);
        B.node_w[rank] = (uint32_t)(key >> 16);
    }
    __syncthreads();
    if (t == 0 && nz >= 2) {
        const uint32_t INF = 0xffffffffu;
        uint32_t leaf = 0, inner = nz, made = nz;
        uint32_t w_leaf = B.node_w[0], w_inner = INF;
        for (uint32_t k = 0; k + 1 < nz; k++) {
            uint32_t sum = 0;
            for (int j = 0; j < 2; j++) {
                if (w_leaf <= w_inner) {
                    sum += w_leaf; B.node_par[leaf] = (uint16_t)made; leaf++;
                    w_leaf = leaf < nz ? B.node_w[leaf] : INF;
                } else {
                    sum += w_inner; B.node_par[inner] = (uint16_t)made; inner++;
                    w_inner = inner < made ? B.node_w[inner] : INF;
                }
            }
            B.node_w[made] = sum;
            if (w_inner == INF && inner == made) w_inner = sum;
            made++;
        }
    }
    __syncthreads();
    if (t < nz) {
        const uint32_t root = 2 * nz - 2;
        uint32_t d = 0, v = t;
        while (v != root && d < 2 * LZ_NLIT) { v = B.node_par[v]; d++; }
        atomicAdd(&B.depth_cnt[d < 31u ? d : 31u], 1u);
    }
    __syncthreads();
    if (t == 0) {
        for (int i = 16; i < 32; i++) B.depth_cnt[15] += B.depth_cnt[i];
        uint32_t total = 0;
        for (int i = 15; i >= 1; i--) total += B.depth_cnt[i] << (15 - i);
        for (uint32_t guard = 0; total != (1u << 15) && guard < (1u << 16); guard++) {
            B.depth_cnt[15]--;
            for (int i = 14; i >= 1; i--)
                if (B.depth_cnt[i]) { B.depth_cnt[i]--; B.depth_cnt[i + 1] += 2; break; }
            total--;
        }
        uint32_t code = 0, prev = 0;
        B.len_cnt[0] = 0;
        for (int i = 1; i < 16; i++) {
            code = (code + prev) << 1;
            B.next_code[i] = code;
            prev = B.depth_cnt[i];
            B.len_cnt[i] = prev;
        }
    }
    __syncthreads();
    if (t < nz) {
        const uint32_t q = nz - 1 - t;              // the q-th most frequent symbol gets the q-th shortest length
        uint32_t acc = 0, len = 15;
        for (uint32_t i = 1; i < 16; i++) {
            acc += B.len_cnt[i];
            if (q < acc) { len = i; break; }
        }
        B.sorted_len[t] = (uint8_t)len;
        sym_len[B.sorted[t]] = (uint8_t)len;
    }
    __syncthreads();
    if (t < nz) {
        const uint32_t sym = B.sorted[t], len = B.sorted_len[t];
        uint32_t before = 0;
        for (uint32_t u = 0; u < nz; u++) before += (B.sorted_len[u] == len && B.sorted[u] < sym) ? 1u : 0u;
        const uint32_t code = B.next_code[len] + before;
        enc[sym] = (__brev(code) >> (32u - len)) | (len << 16);
    }
    __syncthreads();
}

__global__ void __launch_bounds__(BGZF_THREADS)
bgzf_deflate_lz_kernel(const uint8_t* __restrict__ src, uint64_t n_total, uint8_t* __restrict__ slots,
                       uint64_t* __restrict__ sizes, BgzfTables T) {
    __shared__ __align__(16) uint32_t img[BGZF_IMG_WORDS];     // the sub-histograms, then the block's output image
    __shared__ __align__(16) uint32_t in_words[BGZF_BLOCK_IN / 4 + 4];      // the block's input (one workgroup per CU: 147 KB of LDS)
    __shared__ uint32_t s_crc_tab[4 * 256];
    __shared__ uint16_t line_start[LZ_MAX_LINES + 1];
    __shared__ uint64_t used_key[LZ_NLIT];
    __shared__ uint16_t sorted[LZ_NLIT];
    __shared__ uint8_t sorted_len[LZ_NLIT + 2];
    __shared__ uint32_t node_w[2 * LZ_NLIT];
    __shared__ uint16_t node_par[2 * LZ_NLIT];
    __shared__ uint32_t depth_cnt[32];
    __shared__ uint32_t len_cnt[16];
    __shared__ uint32_t next_code[16];
    __shared__ uint8_t sym_len[LZ_NLIT + 2];
    __shared__ uint8_t dsym_len[LZ_NDIST + 2];
    __shared__ uint32_t enc[LZ_NLIT];
    __shared__ uint32_t denc[LZ_NDIST];
    __shared__ uint32_t hist_ld[64];              // [0..28] length codes, [32..61] distance codes
    __shared__ uint32_t wave_part[BGZF_THREADS / 64];
    __shared__ uint32_t s_nz, s_nzd, s_crc, s_total_bits, s_last_crc, s_lines;

    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint64_t blk_off = (uint64_t)blockIdx.x * BGZF_BLOCK_IN;
    const uint32_t n = (uint32_t)((n_total - blk_off) < BGZF_BLOCK_IN ? (n_total - blk_off) : BGZF_BLOCK_IN);
    const uint32_t full = n / BGZF_PIECE, rem = n % BGZF_PIECE;
    uint8_t* const inb = reinterpret_cast<uint8_t*>(in_words);

    s_crc_tab[t] = T.crc_tab[t];
    if (t < 64) hist_ld[t] = 0;
    if (t == 0) { s_nz = 0; s_nzd = 0; s_crc = 0; s_last_crc = 0; line_start[0] = 0; }

    // ---- this thread's piece: in registers for the whole kernel, and in LDS for the neighbours' comparisons
    uint32_t piece[BGZF_PIECE / 4];
    const uint32_t my_n = t < full ? BGZF_PIECE : (t == full ? rem : 0u);
    {
        const uint8_t* p = src + blk_off + (uint64_t)t * BGZF_PIECE;
        if (my_n == BGZF_PIECE) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 v = reinterpret_cast<const uint4*>(p)[q];
                piece[4 * q] = v.x; piece[4 * q + 1] = v.y; piece[4 * q + 2] = v.z; piece[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int w = 0; w < (int)(BGZF_PIECE / 4); w++) {
                uint32_t v = 0;
                for (uint32_t b = 0; b < 4; b++)
                    if (4u * w + b < my_n) v |= (uint32_t)p[4 * w + b] << (8u * b);
                piece[w] = v;
            }
        }
        if (t * BGZF_PIECE < BGZF_BLOCK_IN) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                reinterpret_cast<uint4*>(inb + t * BGZF_PIECE)[q] = make_uint4(piece[4 * q], piece[4 * q + 1], piece[4 * q + 2], piece[4 * q + 3]);
        }
    }
    // newlines of the piece, CRC-32 of the piece
    uint64_t nlm = 0;
    uint32_t crc = 0xffffffffu;
#pragma unroll
    for (int w = 0; w < (int)(BGZF_PIECE / 4); w++) {
#pragma unroll
        for (int b = 0; b < 4; b++) {
            if (4u * w + b < my_n) {
                const uint32_t c = (piece[w] >> (8 * b)) & 0xffu;
                if (c == '\n') nlm |= 1ULL << (4 * w + b);
                if (my_n != BGZF_PIECE) crc = s_crc_tab[(crc ^ c) & 0xffu] ^ (crc >> 8);
            }
        }
        if (my_n == BGZF_PIECE) {
            const uint32_t x = crc ^ piece[w];
            crc = s_crc_tab[768 + (x & 0xffu)] ^ s_crc_tab[512 + ((x >> 8) & 0xffu)] ^ s_crc_tab[256 + ((x >> 16) & 0xffu)] ^ s_crc_tab[x >> 24];
        }
    }
    crc = ~crc;
    {
        uint32_t part = 0;
        if (t < full) part = crc_mulmod(crc, T.x512[full - 1 - t]);
        for (int d = 32; d > 0; d >>= 1) part ^= __shfl_xor(part, d, 64);
        if (lane == 0) wave_part[wave] = part;
        if (t == full && rem) s_last_crc = crc;
    }
    __syncthreads();
    if (t == 0) {
        uint32_t s = 0;
        for (uint32_t w = 0; w < BGZF_THREADS / 64; w++) s ^= wave_part[w];
        s_crc = rem ? (crc_mulmod(s, T.x8[rem]) ^ s_last_crc) : s;
    }
    __syncthreads();

    // ---- line starts: line k begins at line_start[k] (line 0 at the block's first byte, wherever in a line that is)
    uint32_t k0;                                   // line of this piece's first byte
    {
        const uint32_t cnt = (uint32_t)__builtin_popcountll(nlm);
        uint32_t incl = cnt;
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if ((int)lane >= d) incl += o; }
        if (lane == 63) wave_part[wave] = incl;
        __syncthreads();
        if (t == 0) {
            uint32_t run = 0;
            for (uint32_t w = 0; w < BGZF_THREADS / 64; w++) { const uint32_t s = wave_part[w]; wave_part[w] = run; run += s; }
            s_lines = run;
        }
        __syncthreads();
        k0 = wave_part[wave] + incl - cnt;
        uint64_t m = nlm; uint32_t k = k0;
        while (m) {
            const uint32_t j = (uint32_t)__builtin_ctzll(m); m &= m - 1;
            k++;
            if (k <= LZ_MAX_LINES) line_start[k] = (uint16_t)(t * BGZF_PIECE + j + 1u);      // (a start at n is never looked at)
        }
    }
    __syncthreads();
    const bool rec_ok = s_lines < LZ_MAX_LINES;      // else: too many lines to remember, runs only
    // distance to the same column four lines back for a position in line k (0: none)
    auto rec_dist = [&](uint32_t k) -> uint32_t {
        if (!rec_ok || k < 4u) return 0u;
        const uint32_t d = (uint32_t)line_start[k] - (uint32_t)line_start[k - 4u];
        return d <= 32768u ? d : 0u;
    };

    // ---- equality masks of the piece: with the previous record (mR), with the previous byte (m1); line starts (ls).
    // Four bytes at a time: 0x80 in every zero byte of a word, gathered into a nibble
    auto zero_bytes = [](uint32_t x) -> uint32_t {
        const uint32_t u = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu) >> 7;      // bit 8 b: byte b of x is zero
        return ((u * 0x00204081u) >> 21) & 0xfu;
    };
    uint64_t mR = 0, m1 = 0, ls = 0, dchg = 0;
    {
        const uint32_t g0 = t * BGZF_PIECE;
        const uint32_t* const inw = in_words;
        uint32_t k = k0;
        uint32_t D = rec_dist(k);
        const uint32_t prev0 = g0 ? inb[g0 - 1u] : 0x100u;
        (void)ls;                                                         // dchg below: line starts at which the distance changes
        uint32_t carry = prev0 & 0xffu;                                   // the byte before the word
#pragma unroll
        for (int w = 0; w < (int)(BGZF_PIECE / 4); w++) {
            const uint32_t x = piece[w];
            uint32_t e1 = zero_bytes(x ^ ((x << 8) | carry));
            if (w == 0 && g0 == 0) e1 &= ~1u;
            carry = x >> 24;
            m1 |= (uint64_t)e1 << (4 * w);
            const uint32_t nl4 = (uint32_t)(nlm >> (4 * w)) & 0xfu;
            // the bytes of the word up to and including a newline belong to line k (distance D), those behind it to line
            // k + 1 (distance D2): both comparisons are made for the whole word and merged by position -- among the 64
            // lanes of a wave some word always holds a newline, so a per-byte path for that case ran for every word
            auto eq_word = [&](uint32_t dist) -> uint32_t {
                if (!dist || g0 + 4u * w < dist) {
                    // (the first bytes of the block may lie closer to its start than the distance: byte by byte)
                    uint32_t e = 0;
                    if (dist) for (uint32_t b = 0; b < 4; b++) { const uint32_t j = 4 * w + b; if (g0 + j >= dist && inb[g0 + j - dist] == ((x >> (8 * b)) & 0xffu)) e |= 1u << b; }
                    return e;
                }
                const uint32_t a = g0 + 4u * w - dist;
                const uint32_t lo = inw[a >> 2], hi = inw[(a >> 2) + 1u];
                return zero_bytes(x ^ (uint32_t)((((uint64_t)hi << 32) | lo) >> (8u * (a & 3u))));
            };
            if ((nl4 & (nl4 - 1u)) == 0) {                                // at most one newline in the word
                uint32_t e = eq_word(D);
                if (nl4) {
                    k++;
                    const uint32_t D2 = rec_dist(k);
                    const uint32_t first = nl4 | (nl4 - 1u);              // bytes up to and including the newline
                    e = (e & first) | (eq_word(D2) & ~first);
                    if (D2 != D) dchg |= ((uint64_t)nl4 << (4 * w)) << 1;     // a stretch does not continue into a line of another distance
                    D = D2;
                }
                mR |= (uint64_t)e << (4 * w);
            } else {
                for (uint32_t b = 0; b < 4; b++) {
                    const uint32_t j = 4 * w + b;
                    const uint32_t c = (x >> (8 * b)) & 0xffu;
                    if (D && g0 + j >= D && inb[g0 + j - D] == c) mR |= 1ULL << j;
                    if (c == '\n') { k++; const uint32_t D2 = rec_dist(k); if (D2 != D && j + 1u < 64u) dchg |= 1ULL << (j + 1u); D = D2; }
                }
            }
        }
        const uint64_t valid = my_n >= 64u ? ~0ULL : ((1ULL << my_n) - 1ULL);
        mR &= valid; m1 &= valid;
    }
    // ---- tokens of the piece: greedy, left to right.  A match is a maximal stretch of equal bytes -- with the previous
    // record: as long as the distance stays the same, which it does across a line start whenever the two records' lines
    // are equally long ("\n+\n" and the first qualities, a line's end and the next id line's prefix) -- of at least
    // LZ_MIN_MATCH bytes, or of 3 when it holds a newline (line-structure bytes are rare, hence expensive, literals:
    // measured on the headline FASTQ 0.329 against 0.337 with stretches cut at every line start); the longer kind wins
    uint64_t m_start = 0, m_cover = 0, m_rle = 0;      // first byte of each match, all bytes of matches, matches that are runs
    {
        auto starts5 = [](uint64_t m) -> uint64_t { return m & (m >> 1) & (m >> 2) & (m >> 3) & (m >> 4); };
        const uint64_t cand = (mR & (mR >> 1) & (mR >> 2)) | starts5(m1);       // the only places a match can start
        uint32_t p = 0;
        while (p < my_n) {
            const uint64_t rest = cand >> p;
            if (!rest) break;
            p += (uint32_t)__builtin_ctzll(rest);
            const uint64_t sR = ~(mR >> p) | ((dchg >> p) & ~1ULL), s1 = ~(m1 >> p);
            uint32_t lR = sR ? (uint32_t)__builtin_ctzll(sR) : 64u, l1 = s1 ? (uint32_t)__builtin_ctzll(s1) : 64u;      // (within the piece: the masks end at my_n)
            const uint64_t nl_in = (nlm >> p) & (lR >= 64u ? ~0ULL : ((1ULL << lR) - 1ULL));
            if (lR < (nl_in ? 3u : LZ_MIN_MATCH)) lR = 0;
            if (l1 < LZ_MIN_MATCH) l1 = 0;
            const bool rle = l1 > lR;
            const uint32_t len = rle ? l1 : lR;
            if (len) {
                m_start |= 1ULL << p;
                m_cover |= (len >= 64u ? ~0ULL : ((1ULL << len) - 1ULL)) << p;
                if (rle) m_rle |= 1ULL << p;
                p += len;
            } else p++;
        }
    }
    __syncthreads();

    // ---- 1. histograms.  Literals as above (64 per-lane sub-histograms); length and distance codes with plain atomics
    constexpr uint32_t SUB_STRIDE = 129;
    uint32_t* const sub = img;
    for (uint32_t i = t; i < 64 * SUB_STRIDE; i += BGZF_THREADS) sub[i] = 0;
    __syncthreads();
    // (length, distance) of the match that starts at byte p of the piece
    auto match_at = [&](uint32_t p, uint32_t* len, uint32_t* dist) {
        const uint64_t c = m_cover >> p, nxt = (m_start >> p) & ~1ULL;
        const uint64_t stop = ~c | nxt;
        *len = stop ? (uint32_t)__builtin_ctzll(stop) : 64u - p;
        if ((m_rle >> p) & 1ULL) *dist = 1u;
        else *dist = rec_dist(k0 + (uint32_t)__builtin_popcountll(nlm & ((1ULL << p) - 1ULL)));
    };
    // (the three passes over the piece -- histogram, bit count, emission -- are loops over its words, read back from LDS:
    //  unrolled over 64 register-held bytes with a match branch at every byte the kernel was 29 000 lines of code, far
    //  beyond the instruction cache)
    const uint32_t nw = (my_n + 3u) >> 2;
    const uint32_t* const myw = in_words + t * (BGZF_PIECE / 4);
    const uint64_t lit_mask = ~m_cover & (my_n >= 64u ? ~0ULL : ((1ULL << my_n) - 1ULL));      // bytes that stay literals
    {
        uint32_t* const my_sub = sub + lane * SUB_STRIDE;
        for (uint32_t w = 0; w < nw; w++) {
            const uint32_t x = myw[w];
            const uint32_t l4 = (uint32_t)(lit_mask >> (4u * w)) & 0xfu;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                if ((l4 >> b) & 1u) {
                    const uint32_t c = (x >> (8 * b)) & 0xffu;
                    atomicAdd(&my_sub[c >> 1], 1u << (16u * (c & 1u)));
                }
            }
        }
        uint64_t m = m_start;
        while (m) {
            const uint32_t p = (uint32_t)__builtin_ctzll(m); m &= m - 1;
            uint32_t len, dist;
            match_at(p, &len, &dist);
            atomicAdd(&hist_ld[lz_len_code(len)], 1u);
            atomicAdd(&hist_ld[32u + lz_dist_code(dist)], 1u);
        }
    }
    __syncthreads();
    if (t < LZ_NLIT) {
        uint32_t f;
        if (t < 256) {
            f = 0;
            for (uint32_t l = 0; l < 64; l++) f += (sub[l * SUB_STRIDE + (t >> 1)] >> (16u * (t & 1u))) & 0xffffu;
        } else f = t == 256 ? 1u : hist_ld[t - 257u];
        sym_len[t] = 0; enc[t] = 0;
        if (f) { const uint32_t slot = atomicAdd(&s_nz, 1u); used_key[slot] = ((uint64_t)f << 16) | t; }
    }
    if (t < LZ_NDIST) { dsym_len[t] = 0; denc[t] = 0; }
    __syncthreads();
    for (uint32_t i = t; i < BGZF_IMG_WORDS; i += BGZF_THREADS) img[i] = 0;       // from here on the buffer is the image

    // ---- 3. the two Huffman codes
    LzBuild B{used_key, sorted, sorted_len, node_w, node_par, depth_cnt, len_cnt, next_code};
    lz_build_codes(B, s_nz, sym_len, enc, t);
    if (t < LZ_NDIST) {
        const uint32_t f = hist_ld[32u + t];
        if (f) { const uint32_t slot = atomicAdd(&s_nzd, 1u); used_key[slot] = ((uint64_t)f << 16) | t; }
    }
    __syncthreads();
    if (s_nzd) lz_build_codes(B, s_nzd, dsym_len, denc, t);

    // ---- 4. bit counts and scan.  Tokens are placed at absolute bit positions (the image is OR-ed together anyway), so the
    // literals of a word and the matches need not be emitted in stream order: the lanes of a wave have their matches at
    // different bytes, and a byte loop with a match branch runs that branch -- ~80 instructions -- at nearly every byte
    // because SOME lane needs it (the first version: 9 600 vector instructions per wave).  Instead: per word the bit
    // length of its literals (pre[w]: prefix sums, 11 bits each, packed), a short loop over the lane's matches, and a
    // branch-free pass that writes each word's literals as one group.
    uint32_t my_bits = 0;
    uint64_t pre[4] = {0, 0, 0, 0};                // pre[w]: bits of the literals in words < w (w = 0..15), 12 bits each, five to a register
    for (uint32_t w = 0; w < nw; w++) {
        const uint32_t x = myw[w];
        const uint32_t l4 = (uint32_t)(lit_mask >> (4u * w)) & 0xfu;
        pre[w / 5u] |= (uint64_t)my_bits << (12u * (w % 5u));
        const uint32_t e0 = enc[x & 0xffu] >> 16, e1 = enc[(x >> 8) & 0xffu] >> 16, e2 = enc[(x >> 16) & 0xffu] >> 16, e3 = enc[x >> 24] >> 16;
        my_bits += ((l4 & 1u) ? e0 : 0u) + ((l4 & 2u) ? e1 : 0u) + ((l4 & 4u) ? e2 : 0u) + ((l4 & 8u) ? e3 : 0u);
    }
    auto pre_at = [&](uint32_t w) -> uint32_t { return (uint32_t)(pre[w / 5u] >> (12u * (w % 5u))) & 0xfffu; };
    // the match that starts at byte p: its code bits (<= 48) and their number
    auto match_bits = [&](uint32_t p, uint64_t* bits) -> uint32_t {
        uint32_t len, dist;
        match_at(p, &len, &dist);
        const uint32_t lc = lz_len_code(len), dc = lz_dist_code(dist);
        const uint32_t el = enc[257u + lc], ed = denc[dc];
        uint32_t nb = el >> 16;
        uint64_t v = el & 0xffffu;
        v |= (uint64_t)(len - LZ_LEN_BASE[lc]) << nb; nb += LZ_LEN_EXTRA[lc];
        v |= (uint64_t)(ed & 0xffffu) << nb; nb += ed >> 16;
        v |= (uint64_t)(dist - LZ_DIST_BASE[dc]) << nb; nb += LZ_DIST_EXTRA[dc];
        *bits = v;
        return nb;
    };
    const uint32_t lit_bits = my_bits;
    uint64_t mw[2] = {0, 0};                       // bits of the matches that start in word w (two at most: 3 bytes at byte 0, another at byte 3), 8 bits each
    {
        uint64_t m = m_start;
        while (m) {
            const uint32_t p = (uint32_t)__builtin_ctzll(m); m &= m - 1;
            uint64_t v;
            const uint32_t nb = match_bits(p, &v);
            my_bits += nb;
            mw[p >> 5] += (uint64_t)nb << (8u * ((p >> 2) & 7u));
        }
    }
    (void)lit_bits;
    const uint32_t t_last = (n - 1) / BGZF_PIECE;          // owner of the last byte appends end-of-block
    const uint32_t eob_at = my_bits;
    if (t == t_last) my_bits += enc[256] >> 16;
    uint32_t incl = my_bits;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += o;
    }
    __syncthreads();
    if (lane == 63) wave_part[wave] = incl;
    __syncthreads();
    if (t == 0) {
        uint32_t run = 0;
        for (uint32_t w = 0; w < BGZF_THREADS / 64; w++) { const uint32_t s = wave_part[w]; wave_part[w] = run; run += s; }
        s_total_bits = run + LZ_DYN_HDR_BITS;
    }
    __syncthreads();
    const uint32_t payload_dyn = (s_total_bits + 7u) >> 3;
    const bool stored = payload_dyn > n + 5u;
    const uint32_t payload = stored ? n + 5u : payload_dyn;
    const uint32_t total = BGZF_HDR_BYTES + payload + 8u;
    const uint32_t base_bit = BGZF_HDR_BYTES * 8u;

    if (!stored) {
        if (t == 0) {
            // BFINAL=1, BTYPE=2, HLIT=29 (286 codes), HDIST=29 (30 codes), HCLEN=15 (19 lengths); the code-length
            // code gives symbols 0..15 four bits each (16,17,18 unused), so its canonical codes are the symbols
            img_or_bits(img, base_bit, 1u | (2u << 1) | (29u << 3) | (29u << 8) | (15u << 13));
            uint64_t cl = 0;
            for (int i = 0; i < 16; i++) cl |= 4ull << (3 * i);
            img_or_bits(img, base_bit + 26, cl & 0xffffffu);
            img_or_bits(img, base_bit + 50, cl >> 24);
        }
        if (t < LZ_NLIT + LZ_NDIST) {
            const uint32_t len = t < LZ_NLIT ? sym_len[t] : dsym_len[t - LZ_NLIT];
            img_or_bits(img, base_bit + 74 + 4 * t, __brev(len) >> 28);
        }
        const uint32_t pos0 = base_bit + LZ_DYN_HDR_BITS + wave_part[wave] + (incl - my_bits);      // this thread's first bit
        // matches: each at the bits of the literals before it plus the bits of the matches before it
        {
            uint64_t m = m_start;
            uint32_t msum = 0;
            while (m) {
                const uint32_t p = (uint32_t)__builtin_ctzll(m); m &= m - 1;
                uint64_t v;
                const uint32_t nb = match_bits(p, &v);
                const uint32_t w = p >> 2, bq = p & 3u;
                const uint32_t x = myw[w];
                const uint32_t l4 = (uint32_t)(lit_mask >> (4u * w)) & ((1u << bq) - 1u);      // literals of the word before the match
                uint32_t before = pre_at(w);
                before += ((l4 & 1u) ? enc[x & 0xffu] >> 16 : 0u) + ((l4 & 2u) ? enc[(x >> 8) & 0xffu] >> 16 : 0u) + ((l4 & 4u) ? enc[(x >> 16) & 0xffu] >> 16 : 0u);
                const uint32_t at = pos0 + before + msum;
                img_or_bits(img, at, v & 0xffffffu);
                img_or_bits(img, at + 24u, v >> 24);
                msum += nb;
            }
        }
        // literals: the (up to four) literals of a word are consecutive in the stream
        {
            uint32_t msum = 0;
            for (uint32_t w = 0; w < nw; w++) {
                const uint32_t x = myw[w];
                const uint32_t l4 = (uint32_t)(lit_mask >> (4u * w)) & 0xfu;
                const uint32_t e0 = enc[x & 0xffu], e1 = enc[(x >> 8) & 0xffu], e2 = enc[(x >> 16) & 0xffu], e3 = enc[x >> 24];
                uint64_t v = 0; uint32_t nb = 0;
                if (l4 & 1u) { v = e0 & 0xffffu; nb = e0 >> 16; }
                if (l4 & 2u) { v |= (uint64_t)(e1 & 0xffffu) << nb; nb += e1 >> 16; }
                if (l4 & 4u) { v |= (uint64_t)(e2 & 0xffffu) << nb; nb += e2 >> 16; }
                if (l4 & 8u) { v |= (uint64_t)(e3 & 0xffffu) << nb; nb += e3 >> 16; }
                // a match that starts inside the word follows the word's literals (it covers the rest of the word); one that
                // starts at the word's first byte precedes them (a 3-byte match and a literal at byte 3)
                const uint32_t mbits = (uint32_t)(mw[w >> 3] >> (8u * (w & 7u))) & 0xffu;
                const bool first = (m_start >> (4u * w)) & 1ULL;
                const uint32_t at = pos0 + pre_at(w) + msum + (first ? mbits : 0u);
                if (nb) {
                    img_or_bits(img, at, v & 0x3fffffffu);
                    if (nb > 30u) img_or_bits(img, at + 30u, v >> 30);
                }
                msum += mbits;
            }
        }
        if (t == t_last) { const uint32_t e = enc[256]; img_or_bits(img, pos0 + eob_at, e & 0xffffu); }
    } else {
        if (t == 0) {
            img_or_byte(img, BGZF_HDR_BYTES, 1u);                    // BFINAL=1, BTYPE=0
            img_or_byte(img, BGZF_HDR_BYTES + 1, n & 0xffu);
            img_or_byte(img, BGZF_HDR_BYTES + 2, n >> 8);
            img_or_byte(img, BGZF_HDR_BYTES + 3, ~n & 0xffu);
            img_or_byte(img, BGZF_HDR_BYTES + 4, (~n >> 8) & 0xffu);
        }
        const uint32_t at = BGZF_HDR_BYTES + 5u + t * BGZF_PIECE;
#pragma unroll
        for (int w = 0; w < (int)(BGZF_PIECE / 4); w++) {
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (4u * w + b < my_n) img_or_byte(img, at + 4 * w + b, (piece[w] >> (8 * b)) & 0xffu);
        }
    }
    if (t == 0) {
        const uint32_t h[5] = {0x04088b1fu, 0x00000000u, 0x0006ff00u, 0x00024342u, (total - 1u) & 0xffffu};
        atomicOr(&img[0], h[0]); atomicOr(&img[1], h[1]); atomicOr(&img[2], h[2]); atomicOr(&img[3], h[3]);
        atomicOr(&img[4], h[4]);
        const uint32_t tr = BGZF_HDR_BYTES + payload;
        const uint32_t c = s_crc;
        for (uint32_t k = 0; k < 4; k++) {
            img_or_byte(img, tr + k, (c >> (8 * k)) & 0xffu);
            img_or_byte(img, tr + 4 + k, (n >> (8 * k)) & 0xffu);
        }
        sizes[blockIdx.x] = total;
    }
    __syncthreads();
    uint32_t* dst = reinterpret_cast<uint32_t*>(slots + (uint64_t)blockIdx.x * BGZF_SLOT);
    const uint32_t words = (total + 3u) >> 2;
    for (uint32_t i = t; i < words; i += BGZF_THREADS) dst[i] = img[i];
}

// slot b (sizes[b] bytes, 16-byte aligned) -> out[offs[b] ...): 16-byte pieces, arbitrary destination alignment
__global__ void __launch_bounds__(256)
bgzf_gather_kernel(const uint8_t* __restrict__ slots, const uint64_t* __restrict__ sizes,
                   const uint64_t* __restrict__ offs, uint8_t* __restrict__ out, const uint64_t* __restrict__ base) {
    const uint8_t* s = slots + (uint64_t)blockIdx.x * BGZF_SLOT;
    uint8_t* d = out + base[0] + offs[blockIdx.x];
    const uint32_t n = (uint32_t)sizes[blockIdx.x], n16 = n >> 4;
    for (uint32_t c = threadIdx.x; c < n16; c += 256) {
        const uint4 v = *reinterpret_cast<const uint4*>(s + c * 16);
        __builtin_memcpy(d + c * 16, &v, 16);
    }
    const uint32_t tail = n16 << 4;
    if (tail + threadIdx.x < n) d[tail + threadIdx.x] = s[tail + threadIdx.x];
}

}  // namespace jk
