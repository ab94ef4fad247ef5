/*
 * dump_weights.c -- TEST/BUILD INFRASTRUCTURE (runs only in the build container).
 *
 * Compiles the reference's parameter table IN PLACE (src/convdata.h is a
 * self-contained data header: no includes, no OpenCV) and writes the 8,129
 * float32 parameters, little-endian, in declaration order:
 *   b1[64] | W1[64][9][9] | b2[32] | W2[32][64] | b3[1] | W3[32][5][5]
 * (src/convdata.h:19-29, :32-674, :677-683, :686-976, :979, :982-1176).
 * Built by oracle/Makefile into oracle/_ref/ (git-ignored); its output is the
 * data fixture srcnn_cpp_amd/data/srcnn915_weights.f32.
 */
#include <stdio.h>
#include "convdata.h"

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s out.f32\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "wb");
    if (!f) { perror("fopen"); return 1; }
    size_t n = 0;
    n += fwrite(biases_conv1, sizeof(float), 64, f);
    n += fwrite(weights_conv1_data, sizeof(float), 64 * 81, f);
    n += fwrite(biases_conv2, sizeof(float), 32, f);
    n += fwrite(weights_conv2_data, sizeof(float), 32 * 64, f);
    n += fwrite(&biases_conv3, sizeof(float), 1, f);
    n += fwrite(weights_conv3_data, sizeof(float), 32 * 25, f);
    fclose(f);
    if (n != 8129) { fprintf(stderr, "short write: %zu\n", n); return 1; }
    if (sizeof(weights_conv1_data) != 64 * 81 * 4 || sizeof(weights_conv2_data) != 32 * 64 * 4 ||
        sizeof(weights_conv3_data) != 32 * 25 * 4) { fprintf(stderr, "table size mismatch\n"); return 1; }
    return 0;
}
