// Minimal PNG reader / writer on zlib (8-bit gray, gray+alpha, RGB, RGBA; non-interlaced), a 16-bit
// gray PNG writer and a PFM writer for disparity maps.
// The reference uses the vendored stb_image / stb_image_write for this (main.cu:57-58,162-181).
#pragma once
#define SMX_PNG_MAX_DIM 65535   /* widths / heights above this are rejected (reader and writer) */
unsigned char* smx_png_load(const char* path, int* w, int* h, int* channels);  // malloc()ed or NULL
int smx_png_write(const char* path, int w, int h, int channels, const unsigned char* data);  // 1 ok
int smx_png_write_gray16(const char* path, int w, int h, const unsigned short* data);        // 1 ok
int smx_pfm_write(const char* path, int w, int h, const float* data);                        // 1 ok
