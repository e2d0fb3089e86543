// Host-side decoder of the reference's TFRecord samples (no device code).  The reference reads its
// datasets through TensorFlow's C++ input pipeline (tf.data.TFRecordDataset + parse_single_sequence_example,
// dataset_reader.py:24,62-79 / dataset_reader_emb.py:63-81); this is the native counterpart for the host
// layer's DataManager: one call parses one serialized tf.train.SequenceExample of the 'fixed' schema
// (tfrecord_utils.py:19-41) straight into the caller's batch arrays.  Plain protobuf wire parsing, no
// allocation, re-entrant: the Python side calls it from a thread pool (ctypes drops the GIL).
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/avsi_hip.h"

namespace {

struct Span {
    const uint8_t* p;
    const uint8_t* end;
    bool ok() const { return p <= end; }
    bool empty() const { return p >= end; }
};

bool read_varint(Span& s, uint64_t& v) {
    v = 0;
    for (int shift = 0; shift < 64 && s.p < s.end; shift += 7) {
        const uint8_t b = *s.p++;
        v |= (uint64_t)(b & 0x7F) << shift;
        if (!(b & 0x80)) return true;
    }
    return false;
}

// next field of a message: number, wire type, and for length-delimited fields the payload span
bool next_field(Span& s, uint32_t& field, uint32_t& wt, Span& payload, uint64_t& scalar) {
    uint64_t key;
    if (!read_varint(s, key)) return false;
    field = (uint32_t)(key >> 3);
    wt = (uint32_t)(key & 7);
    payload = {nullptr, nullptr};
    scalar = 0;
    switch (wt) {
        case 0:
            return read_varint(s, scalar);
        case 1:
            if (s.end - s.p < 8) return false;
            memcpy(&scalar, s.p, 8);
            s.p += 8;
            return true;
        case 2: {
            uint64_t n;
            if (!read_varint(s, n) || n > (uint64_t)(s.end - s.p)) return false;
            payload = {s.p, s.p + n};
            s.p += n;
            return true;
        }
        case 5: {
            if (s.end - s.p < 4) return false;
            uint32_t v;
            memcpy(&v, s.p, 4);
            scalar = v;
            s.p += 4;
            return true;
        }
        default:
            return false;
    }
}

enum { KIND_NONE = 0, KIND_BYTES = 1, KIND_FLOAT = 2, KIND_INT64 = 3 };

// Feature -> which list it holds and that list's body
bool open_feature(Span feat, int& kind, Span& body) {
    kind = KIND_NONE;
    body = {nullptr, nullptr};
    uint32_t f, wt;
    uint64_t sc;
    Span pl;
    while (!feat.empty()) {
        if (!next_field(feat, f, wt, pl, sc)) return false;
        if (wt == 2 && f >= 1 && f <= 3) {
            kind = (int)f;
            body = pl;
            return true;
        }
    }
    return true;    // an empty Feature
}

// FloatList body -> out[0..cap); returns the count, -1 if malformed, -2 if it does not fit
int64_t read_floats(Span body, float* out, int64_t cap) {
    int64_t n = 0;
    uint32_t f, wt;
    uint64_t sc;
    Span pl;
    while (!body.empty()) {
        if (!next_field(body, f, wt, pl, sc)) return -1;
        if (f != 1) continue;
        if (wt == 2) {                                  // packed
            const int64_t k = (pl.end - pl.p) / 4;
            if ((pl.end - pl.p) % 4) return -1;
            if (n + k > cap) return -2;
            memcpy(out + n, pl.p, (size_t)k * 4);      // little endian host
            n += k;
        } else if (wt == 5) {
            if (n + 1 > cap) return -2;
            const uint32_t v = (uint32_t)sc;
            memcpy(out + n, &v, 4);
            ++n;
        } else {
            return -1;
        }
    }
    return n;
}

int64_t count_floats(Span body) {
    int64_t n = 0;
    uint32_t f, wt;
    uint64_t sc;
    Span pl;
    while (!body.empty()) {
        if (!next_field(body, f, wt, pl, sc)) return -1;
        if (f != 1) continue;
        if (wt == 2)
            n += (pl.end - pl.p) / 4;
        else if (wt == 5)
            ++n;
        else
            return -1;
    }
    return n;
}

bool read_first_int64(Span body, int64_t& v) {
    uint32_t f, wt;
    uint64_t sc;
    Span pl;
    while (!body.empty()) {
        if (!next_field(body, f, wt, pl, sc)) return false;
        if (f != 1) continue;
        if (wt == 0) {
            v = (int64_t)sc;
            return true;
        }
        if (wt == 2) {
            uint64_t x;
            if (!read_varint(pl, x)) return false;
            v = (int64_t)x;
            return true;
        }
        return false;
    }
    return false;
}

bool read_first_bytes(Span body, Span& out) {
    uint32_t f, wt;
    uint64_t sc;
    Span pl;
    while (!body.empty()) {
        if (!next_field(body, f, wt, pl, sc)) return false;
        if (f == 1 && wt == 2) {
            out = pl;
            return true;
        }
    }
    return false;
}

// map<string, X> entry {1: key, 2: value}
bool open_entry(Span entry, Span& key, Span& value) {
    key = value = {nullptr, nullptr};
    uint32_t f, wt;
    uint64_t sc;
    Span pl;
    while (!entry.empty()) {
        if (!next_field(entry, f, wt, pl, sc)) return false;
        if (wt != 2) continue;
        if (f == 1) key = pl;
        if (f == 2) value = pl;
    }
    return key.p != nullptr;
}

bool key_is(Span key, const char* name) {
    const size_t n = strlen(name);
    return (size_t)(key.end - key.p) == n && memcmp(key.p, name, n) == 0;
}

// FeatureList of per-frame FloatLists -> rows of `width` floats; returns frames, <0 on error
int64_t read_frames(Span list, float* out, int width, int64_t max_frames) {
    int64_t t = 0;
    uint32_t f, wt;
    uint64_t sc;
    Span pl;
    while (!list.empty()) {
        if (!next_field(list, f, wt, pl, sc)) return -1;
        if (f != 1 || wt != 2) continue;
        int kind;
        Span body;
        if (!open_feature(pl, kind, body)) return -1;
        if (out) {
            if (t >= max_frames) return -2;
            if (kind != KIND_FLOAT) return -3;
            const int64_t k = read_floats(body, out + t * width, width);
            if (k == -1) return -1;
            if (k != width) return -3;
        }
        ++t;
    }
    return t;
}

}  // namespace

extern "C" int avsi_sequence_example_shape_host(const void* buf, size_t n, int64_t* shape5) {
    if (!buf || !shape5) return AVSI_ERR_INVALID_ARG;
    for (int i = 0; i < 5; ++i) shape5[i] = 0;         // wav samples, embedding size, mask frames, video frames, labels
    Span msg{(const uint8_t*)buf, (const uint8_t*)buf + n};
    uint32_t f, wt;
    uint64_t sc;
    Span pl;
    while (!msg.empty()) {
        if (!next_field(msg, f, wt, pl, sc)) return AVSI_ERR_INVALID_ARG;
        if (wt != 2 || (f != 1 && f != 2)) continue;
        Span map = pl;
        while (!map.empty()) {
            uint32_t mf, mwt;
            Span entry, key, value;
            if (!next_field(map, mf, mwt, entry, sc)) return AVSI_ERR_INVALID_ARG;
            if (mf != 1 || mwt != 2) continue;
            if (!open_entry(entry, key, value)) return AVSI_ERR_INVALID_ARG;
            if (f == 1) {
                int kind;
                Span body;
                if (!open_feature(value, kind, body)) return AVSI_ERR_INVALID_ARG;
                const int slot = key_is(key, "target_audio_wav") ? 0 : (key_is(key, "embedding") ? 1 : -1);
                if (slot >= 0 && kind == KIND_FLOAT) {
                    const int64_t c = count_floats(body);
                    if (c < 0) return AVSI_ERR_INVALID_ARG;
                    shape5[slot] = c;
                }
            } else {
                const int slot = key_is(key, "mask") ? 2 : (key_is(key, "video_features") ? 3 : (key_is(key, "labels") ? 4 : -1));
                if (slot >= 0) {
                    const int64_t c = read_frames(value, nullptr, 0, 0);
                    if (c < 0) return AVSI_ERR_INVALID_ARG;
                    shape5[slot] = c;
                }
            }
        }
    }
    return AVSI_OK;
}

extern "C" int avsi_sequence_example_decode_fixed_host(const void* buf, size_t n, int num_audio_samples,
                                                       int audio_feat_size, int video_feat_size, int embedding_size,
                                                       int num_frames, int num_video_frames, int num_labels,
                                                       int32_t* lengths2, int32_t* wav_i32, float* embedding,
                                                       char* sample_path, int sample_path_cap, float* labels,
                                                       float* video, float* mask) {
    if (!buf || !lengths2 || !wav_i32 || !mask || !sample_path || sample_path_cap < 1 || num_audio_samples <= 0 ||
        audio_feat_size <= 0 || num_frames < 0 || (embedding_size > 0 && !embedding) || (num_labels > 0 && !labels) ||
        (num_video_frames > 0 && (!video || video_feat_size <= 0)))
        return AVSI_ERR_INVALID_ARG;
    bool have_len = false, have_lab = false, have_wav = false, have_path = false, have_emb = embedding_size <= 0;
    bool have_mask = false, have_video = num_video_frames == 0, have_labels = num_labels == 0;
    Span msg{(const uint8_t*)buf, (const uint8_t*)buf + n};
    uint32_t f, wt;
    uint64_t sc;
    Span pl;
    while (!msg.empty()) {
        if (!next_field(msg, f, wt, pl, sc)) return AVSI_ERR_INVALID_ARG;
        if (wt != 2 || (f != 1 && f != 2)) continue;
        Span map = pl;
        while (!map.empty()) {
            uint32_t mf, mwt;
            Span entry, key, value;
            if (!next_field(map, mf, mwt, entry, sc)) return AVSI_ERR_INVALID_ARG;
            if (mf != 1 || mwt != 2) continue;
            if (!open_entry(entry, key, value)) return AVSI_ERR_INVALID_ARG;
            if (f == 1) {                               // context features
                int kind;
                Span body;
                if (!open_feature(value, kind, body)) return AVSI_ERR_INVALID_ARG;
                if (key_is(key, "sequence_length") || key_is(key, "labels_length")) {
                    int64_t v;
                    if (kind != KIND_INT64 || !read_first_int64(body, v)) return AVSI_ERR_INVALID_ARG;
                    if (key_is(key, "sequence_length"))
                        lengths2[0] = (int32_t)v, have_len = true;
                    else
                        lengths2[1] = (int32_t)v, have_lab = true;
                } else if (key_is(key, "target_audio_wav")) {
                    // FloatList -> int32 with truncation toward zero (tf.to_int32, dataset_reader.py:77), in place
                    if (kind != KIND_FLOAT) return AVSI_ERR_INVALID_ARG;
                    float* tmp = reinterpret_cast<float*>(wav_i32);
                    const int64_t k = read_floats(body, tmp, num_audio_samples);
                    if (k == -1) return AVSI_ERR_INVALID_ARG;
                    if (k != num_audio_samples) return AVSI_ERR_UNSUPPORTED;
                    for (int i = 0; i < num_audio_samples; ++i) {
                        float x;
                        memcpy(&x, &tmp[i], 4);
                        // out-of-range / NaN: what x86's cvttss2si (and so TF's cast on the reference's hosts) returns
                        wav_i32[i] = (x > -2147483904.f && x < 2147483648.f) ? (int32_t)x : INT32_MIN;
                    }
                    have_wav = true;
                } else if (key_is(key, "sample_path")) {
                    Span s;
                    if (kind != KIND_BYTES || !read_first_bytes(body, s)) return AVSI_ERR_INVALID_ARG;
                    const size_t len = (size_t)(s.end - s.p);
                    if (len + 1 > (size_t)sample_path_cap) return AVSI_ERR_UNSUPPORTED;
                    memcpy(sample_path, s.p, len);
                    sample_path[len] = 0;
                    have_path = true;
                } else if (embedding_size > 0 && key_is(key, "embedding")) {
                    if (kind != KIND_FLOAT) return AVSI_ERR_INVALID_ARG;
                    const int64_t k = read_floats(body, embedding, embedding_size);
                    if (k == -1) return AVSI_ERR_INVALID_ARG;
                    if (k != embedding_size) return AVSI_ERR_UNSUPPORTED;
                    have_emb = true;
                }
            } else {                                    // feature lists
                int64_t got = 0, want = -1;
                if (key_is(key, "mask")) {
                    got = read_frames(value, mask, audio_feat_size, num_frames), want = num_frames, have_mask = true;
                } else if (key_is(key, "video_features")) {
                    if (num_video_frames > 0)
                        got = read_frames(value, video, video_feat_size, num_video_frames);
                    else
                        got = read_frames(value, nullptr, 0, 0);
                    want = num_video_frames, have_video = true;
                } else if (key_is(key, "labels")) {
                    if (num_labels > 0)
                        got = read_frames(value, labels, 1, num_labels);
                    else
                        got = read_frames(value, nullptr, 0, 0);
                    want = num_labels, have_labels = true;
                }
                if (got == -1) return AVSI_ERR_INVALID_ARG;
                if (got < 0 || (want >= 0 && got != want)) return AVSI_ERR_UNSUPPORTED;   // ragged / wrong widths
            }
        }
    }
    if (!(have_len && have_lab && have_wav && have_path && have_emb && have_mask && have_video && have_labels))
        return AVSI_ERR_INVALID_ARG;
    return AVSI_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Whole-file form for the reference's datasets (ONE record per .tfrecord file): open, read, verify the TFRecord
// framing (length + masked CRC-32C of the length, payload + masked CRC-32C of the payload), parse.  Everything
// between the path and the batch rows happens here, outside the interpreter lock, so a Python reader can run
// several files in parallel threads (the pure-Python framing loop got SLOWER with threads: 0.17 ms per file
// serial, 0.47 ms with eight).
extern "C" uint32_t avsi_crc32c(const void* data, size_t n, uint32_t seed);

namespace {

uint32_t masked_crc(const void* p, size_t n) {
    const uint32_t c = avsi_crc32c(p, n, 0);
    return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}

// 0 = ok (payload in `buf`), AVSI_ERR_INVALID_ARG = unreadable / truncated / checksum mismatch,
// AVSI_ERR_UNSUPPORTED = more than one record in the file
int read_single_record(const char* path, int verify, std::vector<unsigned char>& buf) {
    FILE* fh = fopen(path, "rb");
    if (!fh) return AVSI_ERR_INVALID_ARG;
    unsigned char head[12];
    int rc = AVSI_OK;
    uint64_t n = 0;
    // the size of the file bounds the record: a corrupt (or non-TFRecord) header must not size a buffer
    long fsize = -1;
    if (fseek(fh, 0, SEEK_END) == 0) fsize = ftell(fh);
    if (fsize < 16 || fseek(fh, 0, SEEK_SET) != 0) rc = AVSI_ERR_INVALID_ARG;
    if (rc == AVSI_OK && fread(head, 1, 12, fh) != 12) rc = AVSI_ERR_INVALID_ARG;
    if (rc == AVSI_OK) {
        memcpy(&n, head, 8);
        uint32_t hcrc;
        memcpy(&hcrc, head + 8, 4);
        // the 12-byte header checksum is always verified (it costs nothing and guards the length); `verify` adds the payload's
        if (n > (uint64_t)fsize - 16 || hcrc != masked_crc(head, 8)) rc = AVSI_ERR_INVALID_ARG;
    }
    if (rc == AVSI_OK) {
        try {
            buf.resize((size_t)n + 4);
        } catch (const std::bad_alloc&) {       // never across the extern "C" boundary
            fclose(fh);
            return AVSI_ERR_INVALID_ARG;
        }
        if (fread(buf.data(), 1, (size_t)n + 4, fh) != (size_t)n + 4) rc = AVSI_ERR_INVALID_ARG;
    }
    if (rc == AVSI_OK && verify) {
        uint32_t dcrc;
        memcpy(&dcrc, buf.data() + n, 4);
        if (dcrc != masked_crc(buf.data(), (size_t)n)) rc = AVSI_ERR_INVALID_ARG;
    }
    if (rc == AVSI_OK) {
        unsigned char extra;
        if (fread(&extra, 1, 1, fh) == 1) rc = AVSI_ERR_UNSUPPORTED;      // a second record follows
        buf.resize((size_t)n);
    }
    fclose(fh);
    return rc;
}

}  // namespace

extern "C" int avsi_tfrecord_file_shape_host(const char* path, int verify, int64_t* shape5) {
    if (!path || !shape5) return AVSI_ERR_INVALID_ARG;
    std::vector<unsigned char> buf;
    const int rc = read_single_record(path, verify, buf);
    if (rc != AVSI_OK) return rc;
    return avsi_sequence_example_shape_host(buf.data(), buf.size(), shape5);
}

extern "C" int avsi_tfrecord_file_decode_fixed_host(const char* path, int verify, int num_audio_samples, int audio_feat_size,
                                                    int video_feat_size, int embedding_size, int num_frames,
                                                    int num_video_frames, int num_labels, int32_t* lengths2, int32_t* wav_i32,
                                                    float* embedding, char* sample_path, int sample_path_cap, float* labels,
                                                    float* video, float* mask) {
    if (!path) return AVSI_ERR_INVALID_ARG;
    // one buffer per thread, kept: a record is 590 KB, and mapping fresh pages for every file was a third of the cost
    static thread_local std::vector<unsigned char> buf;
    const int rc = read_single_record(path, verify, buf);
    if (rc != AVSI_OK) return rc;
    if (buf.capacity() > ((size_t)64 << 20) && buf.size() < buf.capacity() / 4) buf.shrink_to_fit();   // one huge record must not pin its pages for ever
    return avsi_sequence_example_decode_fixed_host(buf.data(), buf.size(), num_audio_samples, audio_feat_size, video_feat_size,
                                                   embedding_size, num_frames, num_video_frames, num_labels, lengths2, wav_i32,
                                                   embedding, sample_path, sample_path_cap, labels, video, mask);
}

// `count` one-record files in ONE call (a slice of a batch: the reader gives each of its threads one such slice, so the
// interpreter is entered once per thread and batch instead of once per record -- at 1024 records per batch the per-record
// task hand-over, ~30 us under the interpreter lock, capped the reader at 14 k records/s whatever the thread count).
// Row i of every output array belongs to paths[i]; codes[i] receives that file's status; returns the first non-zero one.
extern "C" int avsi_tfrecord_files_decode_fixed_host(const char* const* paths, int count, int verify, int num_audio_samples,
                                                     int audio_feat_size, int video_feat_size, int embedding_size, int num_frames,
                                                     int num_video_frames, int num_labels, int32_t* lengths2, int32_t* wav_i32,
                                                     float* embedding, char* sample_paths, int sample_path_cap, float* labels,
                                                     float* video, float* mask, int32_t* codes) {
    if (!paths || count < 0 || !codes) return AVSI_ERR_INVALID_ARG;
    int first = AVSI_OK;
    for (int i = 0; i < count; ++i) {
        const int rc = avsi_tfrecord_file_decode_fixed_host(
            paths[i], verify, num_audio_samples, audio_feat_size, video_feat_size, embedding_size, num_frames, num_video_frames,
            num_labels, lengths2 ? lengths2 + 2 * (size_t)i : nullptr, wav_i32 ? wav_i32 + (size_t)i * num_audio_samples : nullptr,
            (embedding && embedding_size > 0) ? embedding + (size_t)i * embedding_size : nullptr,
            sample_paths ? sample_paths + (size_t)i * sample_path_cap : nullptr, sample_path_cap,
            labels ? labels + (size_t)i * num_labels : nullptr,
            video ? video + (size_t)i * num_video_frames * video_feat_size : nullptr,
            mask ? mask + (size_t)i * num_frames * audio_feat_size : nullptr);
        codes[i] = rc;
        if (rc != AVSI_OK && first == AVSI_OK) first = rc;
    }
    return first;
}
