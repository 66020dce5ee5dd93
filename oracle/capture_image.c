/*
 * capture_image.c -- build-owned implementation of the three image.h entry
 * points (read_image, make_filename, write_image; /root/reference
 * src/image.h:25-27) that the reference's stereo.c / stereo-ghost.c call.
 *
 * TEST INFRASTRUCTURE ONLY (see stereo_oracle.h).  oracle/Makefile links the
 * UNMODIFIED reference pipeline objects against this file instead of the
 * reference's src/image.c so that every intermediate the reference dumps
 * (edges, matches, score_all, scores, score_best, web, output) is captured
 * as the exact u8 / i32 array instead of the lossy 0..255 PPM view
 * (SURVEY.md section 4 and 8c).  Inputs are binary PGM (P5), so no PNG
 * decoder is involved.
 *
 *   SMO_CAPTURE_DIR   directory for the dumps (default ".")
 *
 * Dump format: "<dir>/<name>-<n>.raw" =
 *   int32 width, int32 height, int32 elem_size, then width*height elements.
 */
#include "image.h" /* the reference's own header, from -I/root/reference/src */

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int read_token(FILE *f, int *out)
{
    int c = fgetc(f);
    for (;;) {
        while (c == ' ' || c == '\t' || c == '\n' || c == '\r')
            c = fgetc(f);
        if (c != '#')
            break;
        while (c != '\n' && c != EOF)
            c = fgetc(f);
    }
    if (c < '0' || c > '9')
        return 1;
    int v = 0;
    while (c >= '0' && c <= '9') {
        v = v * 10 + (c - '0');
        c = fgetc(f);
    }
    *out = v; /* the single whitespace after the token has been consumed */
    return 0;
}

int read_image(const char *name, Image *out)
{
    FILE *f = fopen(name, "rb");
    int maxval = 0;
    if (!f || fgetc(f) != 'P' || fgetc(f) != '5' ||
        read_token(f, &out->width) || read_token(f, &out->height) ||
        read_token(f, &maxval) || maxval != 255) {
        fprintf(stderr, "error reading image %s: not a binary 8-bit PGM\n", name);
        return 1;
    }
    size_t n = (size_t)out->width * out->height;
    unsigned char *raw = malloc(n);
    out->data = malloc(sizeof(double) * n);
    if (!raw || !out->data || fread(raw, 1, n, f) != n) {
        fprintf(stderr, "error reading image %s: short file\n", name);
        return 1;
    }
    for (size_t i = 0; i < n; i++)
        out->data[i] = raw[i] / 256.0; /* same scaling as src/image.c:9-15 */
    free(raw);
    fclose(f);
    return 0;
}

char *make_filename(const char *name, ImageProgramType type, int number)
{
    (void)type;
    const char *dir = getenv("SMO_CAPTURE_DIR");
    char *s = malloc(1024);
    snprintf(s, 1024, "%s/%s-%d.raw", dir ? dir : ".", name, number);
    return s;
}

void write_image(void *data, int width, int height, int ghost_size,
                 ImageType type, char *filename)
{
    int elem = type == IMTYPE_BINARY ? 1 : type == IMTYPE_GRAY_INT ? 4 : 8;
    FILE *f = fopen(filename, "wb");
    free(filename);
    if (!f)
        return;
    int hdr[3] = {width, height, elem};
    fwrite(hdr, sizeof(int), 3, f);
    /* data addresses pixel (0,0) of an image whose rows are
     * width + 2*ghost_size elements apart (src/ghost.h:54-55) */
    size_t stride = (size_t)(width + 2 * ghost_size) * elem;
    for (int y = 0; y < height; y++)
        fwrite((char *)data + (size_t)y * stride, elem, width, f);
    fclose(f);
}
