/* fuzz_image_reader.c -- test driver: calls read_image (include/image.h, the drop-in for
 * /root/reference/src/image.c:18-35) on every file named on the command line and prints one
 * line per file: "<rc> <width> <height>".  Built by tests/test_image_robustness.py with
 * -fsanitize=address,undefined: a corrupt file must give rc 1 and a message on stderr
 * (src/image.c:22-31), never a crash, an out-of-bounds access or a leak of the partial image. */
#include <stdio.h>
#include <stdlib.h>
#include "image.h"

int main(int argc, char **argv)
{
    for (int i = 1; i < argc; i++) {
        Image im = {0, 0, 0};
        const int rc = read_image(argv[i], &im);
        printf("%d %d %d\n", rc, rc ? 0 : im.width, rc ? 0 : im.height);
        if (!rc) {
            /* touch every pixel: the reader must have produced width * height doubles */
            double s = 0;
            for (long p = 0; p < (long)im.width * im.height; p++) s += im.data[p];
            if (s < 0) return 3;
            free(im.data);
        }
    }
    return 0;
}
