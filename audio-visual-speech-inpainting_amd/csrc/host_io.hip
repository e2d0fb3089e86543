// Host-side helper of the TFRecord reader/writer (no device code): CRC-32C (Castagnoli), the
// checksum of the TFRecord framing that TensorFlow's C++ RecordReader/RecordWriter computes
// (the reference reads its datasets through tf.data.TFRecordDataset, dataset_reader.py:24).
// SSE4.2 crc32 instruction when the CPU has it (~8 GB/s per core), else slicing-by-8 (~1-2 GB/s); tables built
// once, thread-safe via static initialisation.
#include <stddef.h>
#include <stdint.h>

#include "../../include/avsi_hip.h"

namespace {
struct Crc32cTables {
    uint32_t t[8][256];
    Crc32cTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xFF];
    }
};
__attribute__((target("sse4.2"))) uint32_t crc32c_hw(const uint8_t* p, size_t n, uint32_t c) {
    uint64_t c64 = c;
    while (n && (reinterpret_cast<uintptr_t>(p) & 7)) {
        c64 = __builtin_ia32_crc32qi((uint32_t)c64, *p++);
        --n;
    }
    while (n >= 8) {
        uint64_t v;
        __builtin_memcpy(&v, p, 8);
        c64 = __builtin_ia32_crc32di(c64, v);
        p += 8;
        n -= 8;
    }
    while (n--) c64 = __builtin_ia32_crc32qi((uint32_t)c64, *p++);
    return (uint32_t)c64;
}
}  // namespace

extern "C" uint32_t avsi_crc32c(const void* data, size_t n, uint32_t seed) {
    static const bool have_hw = __builtin_cpu_supports("sse4.2");
    if (have_hw) return ~crc32c_hw(static_cast<const uint8_t*>(data), n, ~seed);
    static const Crc32cTables T;
    const uint8_t* p = static_cast<const uint8_t*>(data);
    uint32_t c = ~seed;
    while (n && (reinterpret_cast<uintptr_t>(p) & 7)) {
        c = (c >> 8) ^ T.t[0][(c ^ *p++) & 0xFF];
        --n;
    }
    while (n >= 8) {
        uint64_t v;
        __builtin_memcpy(&v, p, 8);
        v ^= c;
        c = T.t[7][v & 0xFF] ^ T.t[6][(v >> 8) & 0xFF] ^ T.t[5][(v >> 16) & 0xFF] ^ T.t[4][(v >> 24) & 0xFF] ^
            T.t[3][(v >> 32) & 0xFF] ^ T.t[2][(v >> 40) & 0xFF] ^ T.t[1][(v >> 48) & 0xFF] ^ T.t[0][(v >> 56) & 0xFF];
        p += 8;
        n -= 8;
    }
    while (n--) c = (c >> 8) ^ T.t[0][(c ^ *p++) & 0xFF];
    return ~c;
}

// ---------------------------------------------------------------------------------------------------------------------
// WAV files of the inference driver: `wavfile.write(path, 16000, enhanced[:seq_len * 192].astype(np.int16))`
// (reference inference.py:159-162) for a batch of utterances, natively -- called from a few host threads outside the
// interpreter lock.  16-bit PCM mono with the 44-byte header scipy.io.wavfile.write produces (RIFF / WAVE / 'fmt ' of 16
// bytes / 'data'); float -> int16 as numpy's astype does it on x86 (truncation toward zero of the 32-bit conversion, low
// 16 bits kept; NaN and values beyond the int32 range give 0).
#include <errno.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>

#include <string>
#include <vector>

namespace {
bool make_parent_dirs(const char* path) {
    std::string p(path);
    const size_t last = p.rfind('/');
    if (last == std::string::npos || last == 0) return true;
    p.resize(last);
    struct stat sb;
    if (stat(p.c_str(), &sb) == 0) return S_ISDIR(sb.st_mode);
    for (size_t i = 1; i <= p.size(); ++i) {
        if (i < p.size() && p[i] != '/') continue;
        const std::string part = p.substr(0, i);
        if (mkdir(part.c_str(), 0777) != 0 && errno != EEXIST) return false;
    }
    return true;
}
inline void put_u32(unsigned char* d, uint32_t v) { d[0] = v & 0xFF, d[1] = (v >> 8) & 0xFF, d[2] = (v >> 16) & 0xFF, d[3] = (v >> 24) & 0xFF; }
inline void put_u16(unsigned char* d, uint16_t v) { d[0] = v & 0xFF, d[1] = (v >> 8) & 0xFF; }
}  // namespace

extern "C" int avsi_wav_write_batch_int16_host(const char* const* paths, const float* samples, int64_t stride,
                                               const int32_t* num_samples, int count, int sample_rate, int make_dirs) {
    if (!paths || !samples || !num_samples || count < 0 || sample_rate <= 0) return AVSI_ERR_INVALID_ARG;
    try {
        std::vector<unsigned char> buf;
        for (int i = 0; i < count; ++i) {
            const int n = num_samples[i];
            if (!paths[i] || n < 0) return AVSI_ERR_INVALID_ARG;
            buf.resize(44 + (size_t)n * 2);
            unsigned char* h = buf.data();
            memcpy(h, "RIFF", 4), put_u32(h + 4, 36 + (uint32_t)n * 2), memcpy(h + 8, "WAVEfmt ", 8), put_u32(h + 16, 16);
            put_u16(h + 20, 1), put_u16(h + 22, 1), put_u32(h + 24, (uint32_t)sample_rate), put_u32(h + 28, (uint32_t)sample_rate * 2);
            put_u16(h + 32, 2), put_u16(h + 34, 16), memcpy(h + 36, "data", 4), put_u32(h + 40, (uint32_t)n * 2);
            const float* x = samples + (int64_t)i * stride;
            unsigned char* d = h + 44;
            for (int k = 0; k < n; ++k) {
                const float v = x[k];
                const int32_t q = (v >= -2147483648.f && v < 2147483648.f) ? (int32_t)v : INT32_MIN;
                put_u16(d + 2 * k, (uint16_t)((uint32_t)q & 0xFFFFu));
            }
            if (make_dirs && !make_parent_dirs(paths[i])) return AVSI_ERR_INVALID_ARG;
            FILE* fh = fopen(paths[i], "wb");
            if (!fh) return AVSI_ERR_INVALID_ARG;
            const bool ok = fwrite(buf.data(), 1, buf.size(), fh) == buf.size();
            if (fclose(fh) != 0 || !ok) return AVSI_ERR_INVALID_ARG;
        }
    } catch (const std::bad_alloc&) {
        return AVSI_ERR_INVALID_ARG;
    }
    return AVSI_OK;
}
