// Sanitizer harness for the host-side SequenceExample decoder (CPU only): reads length-prefixed cases from a
// file and runs avsi_sequence_example_shape_host + avsi_sequence_example_decode_fixed_host on each; with a second
// argument (a scratch file path) every case is also written out as the content of a .tfrecord file and read back
// through avsi_tfrecord_file_shape_host / avsi_tfrecord_file_decode_fixed_host (the cases of that run are whole FILES:
// valid framing, truncations, flipped bits, a second record).
// Built by tests/test_native_sanitizers.py with g++ -fsanitize=address,undefined.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../include/avsi_hip.h"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    const int N = 2304, F = 257, V = 136, E = 512;
    int ok = 0, rejected = 0;
    for (;;) {
        uint32_t n;
        if (fread(&n, 4, 1, f) != 1) break;
        std::vector<unsigned char> buf(n ? n : 1);
        if (n && fread(buf.data(), 1, n, f) != n) return 3;
        int64_t shape[5];
        const int rc0 = avsi_sequence_example_shape_host(buf.data(), n, shape);
        // exact-size outputs (heap allocations: the sanitizer sees any write past them)
        const int T = rc0 == 0 && shape[2] >= 0 && shape[2] < 64 ? (int)shape[2] : 12;
        const int Tv = rc0 == 0 && shape[3] >= 0 && shape[3] < 64 ? (int)shape[3] : 12;
        const int L = rc0 == 0 && shape[4] >= 0 && shape[4] < 128 ? (int)shape[4] : 50;
        std::vector<int32_t> lengths(2), wav(N);
        std::vector<float> emb(E), labels(L ? L : 1), video((size_t)(Tv ? Tv : 1) * V), mask((size_t)(T ? T : 1) * F);
        std::vector<char> path(64);
        const int rc = avsi_sequence_example_decode_fixed_host(buf.data(), n, N, F, V, E, T, Tv, L, lengths.data(), wav.data(),
                                                               emb.data(), path.data(), (int)path.size(), labels.data(),
                                                               video.data(), mask.data());
        if (argc > 2) {      // the same bytes as a file
            FILE* o = fopen(argv[2], "wb");
            if (!o) return 4;
            if (n) fwrite(buf.data(), 1, n, o);
            fclose(o);
            int64_t fshape[5];
            const int frc0 = avsi_tfrecord_file_shape_host(argv[2], 1, fshape);
            const int fT = frc0 == 0 && fshape[2] >= 0 && fshape[2] < 64 ? (int)fshape[2] : 12;
            const int fTv = frc0 == 0 && fshape[3] >= 0 && fshape[3] < 64 ? (int)fshape[3] : 12;
            const int fL = frc0 == 0 && fshape[4] >= 0 && fshape[4] < 128 ? (int)fshape[4] : 50;
            std::vector<float> flabels(fL ? fL : 1), fvideo((size_t)(fTv ? fTv : 1) * V), fmask((size_t)(fT ? fT : 1) * F);
            const int frc = avsi_tfrecord_file_decode_fixed_host(argv[2], 1, N, F, V, E, fT, fTv, fL, lengths.data(), wav.data(),
                                                                 emb.data(), path.data(), (int)path.size(), flabels.data(),
                                                                 fvideo.data(), fmask.data());
            (frc == 0 ? ok : rejected)++;
            continue;
        }
        (rc == 0 ? ok : rejected)++;
    }
    fclose(f);
    printf("ok %d rejected %d\n", ok, rejected);
    return 0;
}
