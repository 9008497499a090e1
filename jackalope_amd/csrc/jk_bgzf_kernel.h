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
