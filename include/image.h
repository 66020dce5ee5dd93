/*
 * image.h -- the image I/O entry points the reference's programs are written
 * against, kept signature-for-signature so its main()/algorithm() code and
 * its test scripts keep working (replaces /root/reference/src/image.h:10-31;
 * the implementation in stereomatching_amd/host/image.c is new).
 *
 *   read_image      8-bit (also 1/2/4/16-bit) grayscale PNG or binary PGM ->
 *                   double brightness k/256.0; returns 0, or 1 after printing
 *                   the reference's message on stderr.  Caller frees data.
 *   make_filename   malloc'd "name-n.ppm"; with -DDEBUG it is prefixed by the
 *                   program's directory ser/ par/ sergh/ pargh/ (which must
 *                   already exist), as test/diff.sh expects.
 *   write_image     ASCII P3 PPM, byte-for-byte the reference's format; takes
 *                   ownership of and frees `filename`; a no-op under
 *                   -DNO_WRITES; returns silently if the file cannot be opened.
 *   write_gpu_image the same for an image that lives in device memory
 *                   (replaces src/image.cu:15-23): one device-to-host copy
 *                   through the C ABI of include/stereo_hip.h, then
 *                   write_image.
 */
#ifndef IMAGE_H_INCLUDED
#define IMAGE_H_INCLUDED

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    double *data;
    int width, height;
} Image;

typedef enum ImageType {
    IMTYPE_BINARY,     /* u8, 1 = black, anything else = white */
    IMTYPE_GRAY_FLOAT, /* double in 0..1 */
    IMTYPE_GRAY_INT,   /* int32, rescaled by the image's own min/max */
} ImageType;

typedef enum ProgramType {
    SER = 0, PAR, SERGHOST, PARGHOST,
} ImageProgramType;

int read_image(const char *name, Image *out);
char *make_filename(const char *name, ImageProgramType type, int number);
void write_image(void *data, int width, int height, int ghost_size, ImageType type, char *filename);
void write_gpu_image(void *device_data, int width, int height, int ghost_size, ImageType type,
                     char *filename);

/* extension used by the uint8 upload path: same decoding as read_image but
 * the raw 0..255 samples (what the reference multiplies by 1/256) */
int read_image_u8(const char *name, uint8_t **data, int *width, int *height);

#ifdef __cplusplus
}
#endif
#endif
