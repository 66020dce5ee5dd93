/*
 * image.c -- implementation of include/image.h (the reference's image.h API,
 * /root/reference/src/image.h:25-31, re-implemented from its behaviour).
 *
 * Reader: grayscale PNG (own decoder below, no libpng/zlib needed) and binary
 * PGM.  Writer: the reference's ASCII P3 format, byte for byte
 * (/root/reference/src/image.c:37-88): "P3\n%d %d\n255\n" then one
 * "%d %d %d\n" line per pixel; binary images map 1 -> 0 and everything else
 * -> 255; int images are rescaled by their own min/max in long arithmetic
 * with truncating division (max == min divides by zero exactly as the
 * reference does).
 */
#include "image.h"

#include <errno.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* inflate (RFC 1951), bit-serial canonical Huffman decoding           */
/* ------------------------------------------------------------------ */

typedef struct {
    const uint8_t *in;
    size_t in_len, in_pos;
    uint32_t bit_buf;
    int bit_cnt;
    uint8_t *out;
    size_t out_len, out_pos;
} Inflate;

typedef struct {
    uint16_t count[16];   /* codes of each length */
    uint16_t symbol[288]; /* symbols ordered by code */
} Huffman;

static int need_bits(Inflate *s, int n)
{
    while (s->bit_cnt < n) {
        if (s->in_pos >= s->in_len)
            return -1;
        s->bit_buf |= (uint32_t)s->in[s->in_pos++] << s->bit_cnt;
        s->bit_cnt += 8;
    }
    return 0;
}

static int get_bits(Inflate *s, int n)
{
    if (n == 0)
        return 0;
    if (need_bits(s, n))
        return -1;
    int v = (int)(s->bit_buf & ((1u << n) - 1));
    s->bit_buf >>= n;
    s->bit_cnt -= n;
    return v;
}

static int huff_build(Huffman *h, const uint8_t *lengths, int n)
{
    uint16_t offs[16];
    memset(h->count, 0, sizeof h->count);
    for (int i = 0; i < n; i++)
        h->count[lengths[i]]++;
    if (h->count[0] == n)
        return 0;
    int left = 1;
    for (int len = 1; len < 16; len++) {
        left <<= 1;
        left -= h->count[len];
        if (left < 0)
            return -1; /* over-subscribed */
    }
    offs[1] = 0;
    for (int len = 1; len < 15; len++)
        offs[len + 1] = offs[len] + h->count[len];
    for (int i = 0; i < n; i++)
        if (lengths[i])
            h->symbol[offs[lengths[i]]++] = (uint16_t)i;
    return 0;
}

static int huff_decode(Inflate *s, const Huffman *h)
{
    int code = 0, first = 0, index = 0;
    for (int len = 1; len < 16; len++) {
        int b = get_bits(s, 1);
        if (b < 0)
            return -1;
        code |= b;
        int count = h->count[len];
        if (code - count < first)
            return h->symbol[index + (code - first)];
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31,
                                      35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                      3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193,
                                       257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                       8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6,
                                       7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static int inflate_codes(Inflate *s, const Huffman *lit, const Huffman *dist)
{
    for (;;) {
        int sym = huff_decode(s, lit);
        if (sym < 0)
            return -1;
        if (sym < 256) {
            if (s->out_pos >= s->out_len)
                return -1;
            s->out[s->out_pos++] = (uint8_t)sym;
        } else if (sym == 256) {
            return 0;
        } else {
            sym -= 257;
            if (sym >= 29)
                return -1;
            int eb = get_bits(s, LEN_EXTRA[sym]);
            if (eb < 0)
                return -1;
            int len = LEN_BASE[sym] + eb;
            int ds = huff_decode(s, dist);
            if (ds < 0 || ds >= 30)
                return -1;
            eb = get_bits(s, DIST_EXTRA[ds]);
            if (eb < 0)
                return -1;
            size_t d = (size_t)DIST_BASE[ds] + (size_t)eb;
            if (d > s->out_pos || s->out_pos + (size_t)len > s->out_len)
                return -1;
            for (int i = 0; i < len; i++, s->out_pos++)
                s->out[s->out_pos] = s->out[s->out_pos - d];
        }
    }
}

static int inflate_raw(Inflate *s)
{
    static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int last;
    do {
        last = get_bits(s, 1);
        int type = get_bits(s, 2);
        if (last < 0 || type < 0)
            return -1;
        if (type == 0) {
            s->bit_buf = 0;
            s->bit_cnt = 0;
            if (s->in_pos + 4 > s->in_len)
                return -1;
            unsigned len = s->in[s->in_pos] | (s->in[s->in_pos + 1] << 8);
            s->in_pos += 4;
            if (s->in_pos + len > s->in_len || s->out_pos + len > s->out_len)
                return -1;
            memcpy(s->out + s->out_pos, s->in + s->in_pos, len);
            s->in_pos += len;
            s->out_pos += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lengths[320];
            if (type == 1) {
                int i = 0;
                for (; i < 144; i++) lengths[i] = 8;
                for (; i < 256; i++) lengths[i] = 9;
                for (; i < 280; i++) lengths[i] = 7;
                for (; i < 288; i++) lengths[i] = 8;
                huff_build(&lit, lengths, 288);
                for (i = 0; i < 30; i++) lengths[i] = 5;
                huff_build(&dist, lengths, 30);
            } else {
                int nlen = get_bits(s, 5), ndist = get_bits(s, 5), ncode = get_bits(s, 4);
                if (nlen < 0 || ndist < 0 || ncode < 0)
                    return -1;
                nlen += 257; ndist += 1; ncode += 4;
                if (nlen > 286 || ndist > 30)
                    return -1;
                memset(lengths, 0, sizeof lengths);
                for (int i = 0; i < ncode; i++) {
                    int v = get_bits(s, 3);
                    if (v < 0)
                        return -1;
                    lengths[ORDER[i]] = (uint8_t)v;
                }
                Huffman cl;
                if (huff_build(&cl, lengths, 19))
                    return -1;
                int idx = 0;
                while (idx < nlen + ndist) {
                    int sym = huff_decode(s, &cl);
                    if (sym < 0)
                        return -1;
                    if (sym < 16) {
                        lengths[idx++] = (uint8_t)sym;
                    } else {
                        int prev = 0, rep;
                        if (sym == 16) {
                            if (idx == 0)
                                return -1;
                            prev = lengths[idx - 1];
                            rep = 3 + get_bits(s, 2);
                        } else if (sym == 17) {
                            rep = 3 + get_bits(s, 3);
                        } else {
                            rep = 11 + get_bits(s, 7);
                        }
                        if (rep < 3 || idx + rep > nlen + ndist)
                            return -1;
                        while (rep--)
                            lengths[idx++] = (uint8_t)prev;
                    }
                }
                if (lengths[256] == 0)
                    return -1;
                uint8_t dl[32];
                memcpy(dl, lengths + nlen, (size_t)ndist);
                if (huff_build(&lit, lengths, nlen) || huff_build(&dist, dl, ndist))
                    return -1;
            }
            if (inflate_codes(s, &lit, &dist))
                return -1;
        } else {
            return -1;
        }
    } while (!last);
    return 0;
}

/* ------------------------------------------------------------------ */
/* PNG                                                                 */
/* ------------------------------------------------------------------ */

static uint32_t be32(const uint8_t *p)
{
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

static int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

/* returns 0 ok; 1 not decodable; 2 decodable but `*channels` != 1 */
static int decode_png(const uint8_t *buf, size_t len, uint8_t **pixels, int *w, int *h, int *channels)
{
    static const uint8_t SIG[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (len < 8 || memcmp(buf, SIG, 8))
        return 1;
    size_t pos = 8, zlen = 0;
    uint8_t *z = NULL;
    int depth = 0, ctype = -1, interlace = 0, have_trns = 0, seen_iend = 0;
    *w = *h = 0;
    while (pos + 12 <= len && !seen_iend) {
        uint32_t clen = be32(buf + pos);
        const uint8_t *tag = buf + pos + 4, *data = buf + pos + 8;
        if (clen > len - pos - 12)
            break;
        if (!memcmp(tag, "IHDR", 4) && clen >= 13) {
            *w = (int)be32(data);
            *h = (int)be32(data + 4);
            depth = data[8]; ctype = data[9]; interlace = data[12];
        } else if (!memcmp(tag, "tRNS", 4)) {
            have_trns = 1;
        } else if (!memcmp(tag, "IDAT", 4)) {
            uint8_t *nz = realloc(z, zlen + clen + 1);
            if (!nz) { free(z); return 1; }
            z = nz;
            memcpy(z + zlen, data, clen);
            zlen += clen;
        } else if (!memcmp(tag, "IEND", 4)) {
            seen_iend = 1;
        }
        pos += 12 + (size_t)clen;
    }
    if (ctype < 0 || *w <= 0 || *h <= 0 || !z || zlen < 6) { free(z); return 1; }
    /* channel count an 8-bit-per-channel loader reports for this colour type */
    switch (ctype) {
    case 0: *channels = have_trns ? 2 : 1; break;
    case 2: *channels = have_trns ? 4 : 3; break;
    case 3: *channels = have_trns ? 4 : 3; break;
    case 4: *channels = 2; break;
    case 6: *channels = 4; break;
    default: free(z); return 1;
    }
    if (*channels != 1) { free(z); return 2; }
    if (interlace || !(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) {
        free(z);
        return 1;
    }
    /* dimensions as stb_image bounds them (1 << 24 per side), and nothing is allocated for a
     * header that promises more pixels than its compressed stream can hold: deflate expands at
     * most 1032 : 1 (a 258-byte match per 2 bits), so a file of zlen bytes with larger
     * dimensions is corrupt whatever follows                                                 */
    if (*w > (1 << 24) || *h > (1 << 24)) { free(z); return 1; }
    const size_t row_bytes = ((size_t)*w * depth + 7) / 8;
    const size_t raw_len = (row_bytes + 1) * (size_t)*h;
    if (raw_len / 1032 > zlen) { free(z); return 1; }
    uint8_t *raw = malloc(raw_len);
    if (!raw) { free(z); return 1; }
    Inflate s = {z + 2, zlen - 2, 0, 0, 0, raw, raw_len, 0}; /* skip the 2-byte zlib header */
    int rc = inflate_raw(&s);
    free(z);
    if (rc || s.out_pos != raw_len) { free(raw); return 1; }

    const int bpp = depth == 16 ? 2 : 1; /* filter unit, bytes */
    uint8_t *prev = calloc(row_bytes, 1), *out = malloc((size_t)*w * *h);
    if (!prev || !out) { free(raw); free(prev); free(out); return 1; }
    for (int y = 0; y < *h; y++) {
        uint8_t *row = raw + (size_t)y * (row_bytes + 1);
        const int filter = row[0];
        uint8_t *cur = row + 1;
        for (size_t i = 0; i < row_bytes; i++) {
            int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i];
            int c = i >= (size_t)bpp ? prev[i - bpp] : 0, v = cur[i];
            switch (filter) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) >> 1; break;
            case 4: v += paeth(a, b, c); break;
            default: free(raw); free(prev); free(out); return 1;
            }
            cur[i] = (uint8_t)v;
        }
        memcpy(prev, cur, row_bytes);
        uint8_t *o = out + (size_t)y * *w;
        if (depth == 8) {
            memcpy(o, cur, (size_t)*w);
        } else if (depth == 16) {
            for (int x = 0; x < *w; x++) o[x] = cur[2 * x]; /* high byte */
        } else {
            static const uint8_t SCALE[5] = {0, 0xff, 0x55, 0, 0x11};
            for (int x = 0; x < *w; x++) {
                int per = 8 / depth, idx = x / per, sh = 8 - depth * (x % per + 1);
                o[x] = (uint8_t)(((cur[idx] >> sh) & ((1 << depth) - 1)) * SCALE[depth]);
            }
        }
    }
    free(raw);
    free(prev);
    *pixels = out;
    return 0;
}

/* ------------------------------------------------------------------ */
/* PGM                                                                 */
/* ------------------------------------------------------------------ */

static int pgm_token(const uint8_t *buf, size_t len, size_t *pos, int *out)
{
    size_t p = *pos;
    for (;;) {
        while (p < len && (buf[p] == ' ' || buf[p] == '\t' || buf[p] == '\n' || buf[p] == '\r')) p++;
        if (p < len && buf[p] == '#') {
            while (p < len && buf[p] != '\n') p++;
            continue;
        }
        break;
    }
    if (p >= len || buf[p] < '0' || buf[p] > '9')
        return 1;
    long v = 0;
    while (p < len && buf[p] >= '0' && buf[p] <= '9' && v < INT_MAX / 10) v = v * 10 + (buf[p++] - '0');
    *out = (int)v;
    *pos = p;
    return 0;
}

static int decode_pgm(const uint8_t *buf, size_t len, uint8_t **pixels, int *w, int *h)
{
    size_t pos = 2;
    int maxval = 0;
    if (len < 2 || buf[0] != 'P' || buf[1] != '5')
        return 1;
    if (pgm_token(buf, len, &pos, w) || pgm_token(buf, len, &pos, h) ||
        pgm_token(buf, len, &pos, &maxval) || maxval != 255 || *w <= 0 || *h <= 0)
        return 1;
    pos++; /* the single whitespace after maxval */
    if (pos + (size_t)*w * *h > len)
        return 1;
    *pixels = malloc((size_t)*w * *h);
    if (!*pixels)
        return 1;
    memcpy(*pixels, buf + pos, (size_t)*w * *h);
    return 0;
}

/* ------------------------------------------------------------------ */
/* the image.h entry points                                            */
/* ------------------------------------------------------------------ */

int read_image_u8(const char *name, uint8_t **data, int *width, int *height)
{
    FILE *f = fopen(name, "rb");
    uint8_t *buf = NULL;
    long size = 0;
    if (f && !fseek(f, 0, SEEK_END) && (size = ftell(f)) >= 0 && !fseek(f, 0, SEEK_SET))
        buf = malloc((size_t)size + 1);
    if (!f || !buf || fread(buf, 1, (size_t)size, f) != (size_t)size) {
        /* same shape as the reference: "error reading image NAME:" + perror("") */
        fprintf(stderr, "error reading image %s:", name);
        perror("");
        if (f) fclose(f);
        free(buf);
        return 1;
    }
    fclose(f);
    int channels = 1, rc;
    if (size >= 2 && buf[0] == 'P' && buf[1] == '5')
        rc = decode_pgm(buf, (size_t)size, data, width, height);
    else
        rc = decode_png(buf, (size_t)size, data, width, height, &channels);
    free(buf);
    if (rc == 2) {
        fprintf(stderr, "error reading image %s: wrong number of channels (%d) "
                        "(image must be grayscale)", name, channels);
        return 1;
    }
    if (rc) {
        fprintf(stderr, "error reading image %s:", name);
        errno = 0;
        perror("");
        return 1;
    }
    return 0;
}

int read_image(const char *name, Image *out)
{
    uint8_t *px = NULL;
    if (read_image_u8(name, &px, &out->width, &out->height))
        return 1;
    const size_t n = (size_t)out->width * out->height;
    out->data = calloc(n ? n : 1, sizeof(double));
    if (!out->data) {
        fprintf(stderr, "error: out of memory\n");
        exit(1);
    }
    for (size_t i = 0; i < n; i++)
        out->data[i] = px[i] / 256.0;
    free(px);
    return 0;
}

char *make_filename(const char *name, ImageProgramType type, int number)
{
    char *s = calloc(1024, 1);
    if (!s) {
        fprintf(stderr, "error: out of memory\n");
        exit(1);
    }
#ifdef DEBUG
    static const char *const DIRS[] = {"ser", "par", "sergh", "pargh"};
    snprintf(s, 1024, "%s/%s-%d.ppm", DIRS[type], name, number);
#else
    (void)type;
    snprintf(s, 1024, "%s-%d.ppm", name, number);
#endif
    return s;
}

void write_image(void *data, int width, int height, int ghost_size, ImageType type, char *filename)
{
#ifdef NO_WRITES
    (void)data; (void)width; (void)height; (void)ghost_size; (void)type; (void)filename;
#else
    FILE *f = fopen(filename, "w");
    free(filename);
    if (!f)
        return;
    const size_t stride = (size_t)width + 2 * (size_t)ghost_size;
    long lo = 0, hi = 0;
    if (type == IMTYPE_GRAY_INT) {
        /* the reference scans width*height CONTIGUOUS elements from `data`
         * (src/image.c:78-79) whatever the ghost size; int images are never
         * padded in its programs */
        const int32_t *p = data;
        lo = INT_MAX; hi = INT_MIN;
        for (size_t i = 0; i < (size_t)width * height; i++) {
            if (p[i] < lo) lo = p[i];
            if (p[i] > hi) hi = p[i];
        }
    }
    /* one big buffer: a 4K image is 8.3 M lines */
    setvbuf(f, NULL, _IOFBF, 1 << 20);
    fprintf(f, "P3\n%d %d\n255\n", width, height);
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++) {
            const size_t i = (size_t)y * stride + x;
            int v;
            switch (type) {
            case IMTYPE_BINARY: v = ((uint8_t *)data)[i] == 1 ? 0 : 255; break;
            case IMTYPE_GRAY_FLOAT: v = (int)(((double *)data)[i] * 255.0); break;
            case IMTYPE_GRAY_INT: v = (int)((((int32_t *)data)[i] - lo) * 255 / (hi - lo)); break;
            default: v = 0;
            }
            fprintf(f, "%d %d %d\n", v, v, v);
        }
    }
    fclose(f);
#endif
}
