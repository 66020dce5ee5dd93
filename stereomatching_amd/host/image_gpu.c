/*
 * image_gpu.c -- write_gpu_image of include/image.h: dump an image that lives
 * in device memory (replaces /root/reference/src/image.cu:15-23).  One
 * device-to-host copy through the C ABI, then the ordinary write_image.  The
 * HIP layer never pads its images, so ghost_size is only honoured for
 * interface compatibility.
 */
#include "image.h"
#include "stereo_hip.h"

#include <stdio.h>
#include <stdlib.h>

#ifndef NO_WRITES
static size_t elem_size(ImageType type)
{
    switch (type) {
    case IMTYPE_BINARY: return sizeof(uint8_t);
    case IMTYPE_GRAY_FLOAT: return sizeof(double);
    case IMTYPE_GRAY_INT: return sizeof(int32_t);
    default: return 0;
    }
}
#endif

void write_gpu_image(void *device_data, int width, int height, int ghost_size, ImageType type,
                     char *filename)
{
#ifdef NO_WRITES
    (void)device_data; (void)width; (void)height; (void)ghost_size; (void)type; (void)filename;
#else
    const size_t es = elem_size(type);
    const size_t stride = (size_t)width + 2 * (size_t)ghost_size;
    const size_t count = stride * ((size_t)height + 2 * (size_t)ghost_size);
    /* device_data addresses pixel (0,0); the allocation starts ghost_size rows
     * and columns earlier (src/ghost.h:6-14) */
    const size_t lead = ((size_t)ghost_size * stride + (size_t)ghost_size) * es;
    char *host = calloc(count ? count : 1, es);
    if (!host) {
        fprintf(stderr, "error: out of memory\n");
        exit(1);
    }
    if (sm_memcpy_d2h(0, host, (char *)device_data - lead, count * es)) {
        fprintf(stderr, "%s\n", sm_last_error());
        exit(EXIT_FAILURE);
    }
    write_image(host + lead, width, height, ghost_size, type, filename);
    free(host);
#endif
}
