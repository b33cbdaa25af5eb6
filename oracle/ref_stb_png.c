/* oracle/ref_stb_png.c — TEST INFRASTRUCTURE, not part of the product.
 *
 * Builds the reference's OWN PNG encoder: the stb_image_write.h it vendors and calls
 * (CudaTest/src/core/render.h:6,153 -> CudaTest/stb-master/stb_image_write.h), compiled from where it
 * lies under /root/reference (the path arrives as -DREF_STB_HEADER=...; nothing of it is copied into
 * this repository), output only into oracle/_ref/ (oracle/Makefile, target ref).  tests/test_png.py
 * and tools/make_png_golden.py use it to pin csrc/crt_png.cpp byte for byte. */
#define STB_IMAGE_WRITE_IMPLEMENTATION
#define STBI_WRITE_NO_STDIO
#include REF_STB_HEADER

#include <string.h>

/* stbi_write_png(path, x, y, comp, data, 0) without the fwrite: returns the file's length and copies it
 * to `out` when it fits `capacity`; negative on failure. */
long long ref_stb_png(const unsigned char *pixels, int x, int y, int comp, unsigned char *out, long long capacity) {
    int len = 0;
    unsigned char *png = stbi_write_png_to_mem(pixels, 0, x, y, comp, &len);
    if (!png) return -1;
    if (out && len <= capacity) memcpy(out, png, (size_t)len);
    STBIW_FREE(png);
    return len;
}
