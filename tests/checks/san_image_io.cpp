// san_image_io.cpp -- sanitizer harness for tools/image_io.hpp (the decoders parse untrusted files).
// Built by tests/test_sanitizers.py with -fsanitize=address,undefined -fno-sanitize-recover=all and run on the
// CPU only.  Decodes every file named on the command line; a decoder may REJECT a file (prints "reject"), it must
// never read or write out of bounds, overflow a signed integer, or allocate from a wrapped size.
#include <cstdio>

#include "image_io.hpp"

int main(int argc, char **argv)
{
    int accepted = 0;
    for (int i = 1; i < argc; ++i) {
        imgio::Image img = imgio::imread(argv[i]);
        if (img.empty()) {
            std::printf("reject %s\n", argv[i]);
            continue;
        }
        if (img.width <= 0 || img.height <= 0 || img.bgr.size() != (size_t)img.width * img.height * 3) {
            std::printf("INCONSISTENT %s: %d x %d, %zu bytes\n", argv[i], img.width, img.height, img.bgr.size());
            return 3;
        }
        unsigned long s = 0;
        for (unsigned char v : img.bgr) s += v;        // touch every byte the decoder claims to have produced
        std::printf("ok %s %dx%d sum %lu\n", argv[i], img.width, img.height, s);
        ++accepted;
    }
    std::printf("accepted %d of %d\n", accepted, argc - 1);
    return 0;
}
