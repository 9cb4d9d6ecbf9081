// Host side of the packed sequence form (pack.hip): four-state sequences as 2 bits per base plus 1
// "invalid" bit per base -- 3/8 of the bytes the reference's one-byte-per-base convention takes
// (src/record.rs:205-209 / diverse_seq/util.py:32-45) -- in the layout the kernels consume as it is:
//   codes: one uint32 per 16 bases, base 16 w at bits 31..30, ..., base 16 w + 15 at bits 1..0
//          (value = alphabet index & 3): the first base of a k-mer is its most significant digit
//          (src/record.rs:18-29), so a k-mer index is a shift and a mask of two neighbouring words;
//   mask:  one uint16 per 16 bases, bit 15 - i set when base 16 w + i is >= 4 (gap / ambiguity / filler).
// Plain C++ (no HIP): compiled for the host only, with per-function x86 targets and a run-time check, so
// the library still loads on a CPU without AVX2 / AVX-512.
#include <cstddef>
#include <cstdint>
#include <cstring>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {

// 16 bases -> one code word + one mask word; any symbol >= 4 is "invalid" (its code bits are dropped)
inline void pack16_scalar(const uint8_t *s, uint32_t *code, uint16_t *mask) {
    uint32_t c = 0, m = 0;
    for (int i = 0; i < 16; i++) {
        const uint8_t b = s[i];
        c |= uint32_t(b & 3u) << (30 - 2 * i);
        m |= uint32_t(b > 3u) << (15 - i);
    }
    *code = c;
    *mask = uint16_t(m);
}

#if defined(__x86_64__)
// Both vector forms reverse the bytes inside each 16-byte lane first (the first base of a group of 16 ends up in
// the lane's last byte): the byte mask of "symbol >= 4" is then the mask word as it is, and the code word is the
// little-endian concatenation of the bytes' low two bits -- two multiply-adds ((1, 4) over byte pairs, (1, 16) over
// the 16-bit sums) leave four bases' bits in the low byte of every 32-bit element, one byte shuffle lines the
// four bytes of a lane up as its code word.
__attribute__((target("avx2"))) void pack_avx2(const uint8_t *src, size_t n32, uint32_t *codes, uint16_t *mask) {
    const __m256i four = _mm256_set1_epi8(4), three = _mm256_set1_epi8(3);
    const __m256i rev = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0,
                                         15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    const __m256i w14 = _mm256_set1_epi16(0x0401), w116 = _mm256_set1_epi32(0x00100001);
    const __m256i low = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                         0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    for (size_t g = 0; g < n32; g++) {
        const __m256i v = _mm256_shuffle_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + g * 32)), rev);
        // (unsigned) v >= 4  <=>  min(v, 4) == 4
        const uint32_t m = uint32_t(_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_min_epu8(v, four), four)));
        const __m256i q = _mm256_madd_epi16(_mm256_maddubs_epi16(_mm256_and_si256(v, three), w14), w116);
        const __m256i c = _mm256_shuffle_epi8(q, low);
        codes[2 * g] = uint32_t(_mm256_cvtsi256_si32(c));
        codes[2 * g + 1] = uint32_t(_mm256_extract_epi32(c, 4));
        std::memcpy(mask + 2 * g, &m, 4);  // (low half: bases 0..15 of the group, high half: 16..31)
    }
}

__attribute__((target("avx512f,avx512bw"))) void pack_avx512(const uint8_t *src, size_t n64, uint32_t *codes, uint16_t *mask) {
    const __m512i three = _mm512_set1_epi8(3);
    const __m512i rev = _mm512_broadcast_i32x4(_mm_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0));
    const __m512i w14 = _mm512_set1_epi16(0x0401), w116 = _mm512_set1_epi32(0x00100001);
    const __m512i low = _mm512_broadcast_i32x4(_mm_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1));
    for (size_t g = 0; g < n64; g++) {
        const __m512i v = _mm512_shuffle_epi8(_mm512_loadu_si512(src + g * 64), rev);
        const uint64_t m = _mm512_cmpgt_epu8_mask(v, three);
        const __m512i q = _mm512_madd_epi16(_mm512_maddubs_epi16(_mm512_and_si512(v, three), w14), w116);
        const __m512i c = _mm512_maskz_compress_epi32(0x1111, _mm512_shuffle_epi8(q, low));
        _mm_storeu_si128(reinterpret_cast<__m128i *>(codes + 4 * g), _mm512_castsi512_si128(c));
        std::memcpy(mask + 4 * g, &m, 8);
    }
}
#endif

}  // namespace

// the widest form this CPU runs: 2 = AVX-512 (F + BW), 1 = AVX2, 0 = scalar
extern "C" int dvs_pack_level(void) {
#if defined(__x86_64__)
    static const int level = __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512f") ? 2
                             : __builtin_cpu_supports("avx2") ? 1 : 0;
    return level;
#else
    return 0;
#endif
}

// Packs src[0, n) into codes[ceil16(n) / 16] and mask[ceil16(n) / 16]; positions in [n, ceil16(n)) are
// marked invalid.  `level` picks the form (never above dvs_pack_level()): the CPU tests run all of them.
// (C linkage only so that the CPU tests can call it; not part of include/dvs_hip.h.)
extern "C" void dvs_pack_bases_level(const uint8_t *src, size_t n, uint32_t *codes, uint16_t *mask, int level) {
    size_t done = 0;  // groups of 32 packed so far
#if defined(__x86_64__)
    if (level > dvs_pack_level()) level = dvs_pack_level();
    if (level == 2) {
        pack_avx512(src, n / 64, codes, mask);
        done = (n / 64) * 2;
    } else if (level == 1) {
        pack_avx2(src, n / 32, codes, mask);
        done = n / 32;
    }
#endif
    size_t w = done * 2;
    for (; w < n / 16; w++) pack16_scalar(src + w * 16, codes + w, mask + w);
    if (n % 16) {
        uint8_t tail[16];
        std::memset(tail, 0xFF, sizeof tail);
        std::memcpy(tail, src + w * 16, n % 16);
        pack16_scalar(tail, codes + w, mask + w);
    }
}

extern "C" void dvs_pack_bases(const uint8_t *src, size_t n, uint32_t *codes, uint16_t *mask) {
    dvs_pack_bases_level(src, n, codes, mask, dvs_pack_level());
}
