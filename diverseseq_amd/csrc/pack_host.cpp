// Host side of the packed sequence form (pack.hip): four-state sequences as 2 bits per base plus 1
// "invalid" bit per base -- 3/8 of the bytes the reference's one-byte-per-base convention takes
// (src/record.rs:205-209 / diverse_seq/util.py:32-45) -- in the layout the kernels consume as it is:
//   codes: one uint32 per 16 bases, base 16 w at bits 31..30, ..., base 16 w + 15 at bits 1..0
//          (value = alphabet index & 3): the first base of a k-mer is its most significant digit
//          (src/record.rs:18-29), so a k-mer index is a shift and a mask of two neighbouring words;
//   mask:  one uint16 per 16 bases, bit 15 - i set when base 16 w + i is >= 4 (gap / ambiguity / filler).
// Plain C++ (no HIP): compiled for the host only, with per-function x86 targets and a run-time check, so
// the library still loads on a CPU without AVX2 / BMI2.
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
__attribute__((target("avx2,bmi2"))) void pack_avx2(const uint8_t *src, size_t n32, uint32_t *codes, uint16_t *mask) {
    const __m256i four = _mm256_set1_epi8(4);
    // bytes reversed inside each 16-byte lane: the first base of a group of 16 ends up in the lane's last
    // byte, i.e. in the top bit of that lane's half of the movemask and in the top pair of its pext
    const __m256i rev = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0,
                                         15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    const uint64_t sel = 0x0303030303030303ull;
    for (size_t g = 0; g < n32; g++) {
        const __m256i v = _mm256_shuffle_epi8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + g * 32)), rev);
        // (unsigned) v >= 4  <=>  min(v, 4) == 4
        const uint32_t m = uint32_t(_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_min_epu8(v, four), four)));
        const uint32_t c0 = uint32_t(_pext_u64(uint64_t(_mm256_extract_epi64(v, 0)), sel)) |
                            (uint32_t(_pext_u64(uint64_t(_mm256_extract_epi64(v, 1)), sel)) << 16);
        const uint32_t c1 = uint32_t(_pext_u64(uint64_t(_mm256_extract_epi64(v, 2)), sel)) |
                            (uint32_t(_pext_u64(uint64_t(_mm256_extract_epi64(v, 3)), sel)) << 16);
        codes[2 * g] = c0;
        codes[2 * g + 1] = c1;
        std::memcpy(mask + 2 * g, &m, 4);  // (low half: bases 0..15 of the group, high half: 16..31)
    }
}
#endif

}  // namespace

// Packs src[0, n) into codes[ceil16(n) / 16] and mask[ceil16(n) / 16]; positions in [n, ceil16(n)) are
// marked invalid.  (C linkage only so that the CPU tests can call it; not part of include/dvs_hip.h.)
extern "C" void dvs_pack_bases(const uint8_t *src, size_t n, uint32_t *codes, uint16_t *mask) {
    const size_t n32 = n / 32;
    size_t done = 0;  // groups of 32 packed so far
#if defined(__x86_64__)
    static const bool fast = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
    if (fast) {
        pack_avx2(src, n32, codes, mask);
        done = n32;
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
