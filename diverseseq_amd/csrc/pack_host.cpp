// Host side of the packed sequence upload (pack.hip): four-state sequences cross PCIe as 2 bits per
// base plus 1 "invalid" bit per base (3/8 of the bytes the reference's one-byte-per-base convention
// takes, src/record.rs:205-209 / diverse_seq/util.py:32-45) and are expanded again on the device.
// Plain C++ (no HIP): compiled for the host only, with per-function x86 targets and a run-time check, so
// the library still loads on a CPU without AVX2 / BMI2.
#include <cstddef>
#include <cstdint>
#include <cstring>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {

// 32 bases -> 8 code bytes + 4 mask bytes; any symbol >= 4 is "invalid" (its code bits are dropped)
inline void pack32_scalar(const uint8_t *s, uint8_t *codes, uint8_t *mask) {
    uint64_t c = 0;
    uint32_t m = 0;
    for (int i = 0; i < 32; i++) {
        const uint8_t b = s[i];
        c |= uint64_t(b & 3u) << (2 * i);
        m |= uint32_t(b > 3u) << i;
    }
    std::memcpy(codes, &c, 8);
    std::memcpy(mask, &m, 4);
}

#if defined(__x86_64__)
__attribute__((target("avx2,bmi2"))) void pack_avx2(const uint8_t *src, size_t n32, uint8_t *codes, uint8_t *mask) {
    const __m256i four = _mm256_set1_epi8(4);
    const uint64_t sel = 0x0303030303030303ull;
    for (size_t g = 0; g < n32; g++) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + g * 32));
        // (unsigned) v >= 4  <=>  min(v, 4) == 4
        const uint32_t m = uint32_t(_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_min_epu8(v, four), four)));
        const uint64_t c = _pext_u64(uint64_t(_mm256_extract_epi64(v, 0)), sel) |
                           (_pext_u64(uint64_t(_mm256_extract_epi64(v, 1)), sel) << 16) |
                           (_pext_u64(uint64_t(_mm256_extract_epi64(v, 2)), sel) << 32) |
                           (_pext_u64(uint64_t(_mm256_extract_epi64(v, 3)), sel) << 48);
        std::memcpy(codes + g * 8, &c, 8);
        std::memcpy(mask + g * 4, &m, 4);
    }
}
#endif

}  // namespace

// Packs src[0, n) into codes[ceil32(n) / 4] and mask[ceil32(n) / 8]; positions in [n, ceil32(n)) are
// marked invalid.  (C linkage only so that the CPU tests can call it; not part of include/dvs_hip.h.)
extern "C" void dvs_pack_bases(const uint8_t *src, size_t n, uint8_t *codes, uint8_t *mask) {
    const size_t n32 = n / 32;
    size_t done = 0;
#if defined(__x86_64__)
    static const bool fast = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
    if (fast) {
        pack_avx2(src, n32, codes, mask);
        done = n32;
    }
#endif
    for (size_t g = done; g < n32; g++) pack32_scalar(src + g * 32, codes + g * 8, mask + g * 4);
    if (n % 32) {
        uint8_t tail[32];
        std::memset(tail, 0xFF, sizeof tail);
        std::memcpy(tail, src + n32 * 32, n % 32);
        pack32_scalar(tail, codes + n32 * 8, mask + n32 * 4);
    }
}
