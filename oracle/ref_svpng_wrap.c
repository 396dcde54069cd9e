/* Build recipe glue for oracle/_ref/libref_svpng.so: includes the reference's svpng.inc from where it
 * lies (-I/root/reference/MTPC) -- the file is written to be #included -- and exposes one C symbol
 * that writes the PNG to a path.  No reference source is copied into this repository. */
#include <stdio.h>
#include "svpng.inc"

int ref_svpng_write(const char* path, unsigned w, unsigned h, const unsigned char* rgb)
{
    FILE* fp = fopen(path, "wb");
    if (!fp) return -1;
    svpng(fp, w, h, rgb, 0);
    fclose(fp);
    return 0;
}
