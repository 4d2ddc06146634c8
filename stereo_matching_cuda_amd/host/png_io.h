// Minimal PNG reader / writer on zlib (8-bit gray, gray+alpha, RGB, RGBA; non-interlaced).
// The reference uses the vendored stb_image / stb_image_write for this (main.cu:57-58,162-181).
#pragma once
unsigned char* smx_png_load(const char* path, int* w, int* h, int* channels);  // malloc()ed or NULL
int smx_png_write(const char* path, int w, int h, int channels, const unsigned char* data);  // 1 ok
