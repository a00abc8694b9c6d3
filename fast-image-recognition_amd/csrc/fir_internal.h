// fir_internal.h -- what the library's translation units share (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct fir_gallery;

// Read-only view of a gallery handle for the other translation units (fir_twd.hip).
struct fir_gallery_view {
    int device;
    int cus;
    int64_t n;
    int d;
    int64_t row_offset;
    const int32_t* cls;     // device, may be NULL
    hipStream_t stream;     // the handle's own stream
};
extern "C" int fir_gallery_view_(fir_gallery* g, fir_gallery_view* out);
extern "C" void fir_set_last_error_(const char* msg);
extern "C" int fir_gallery_tiled_(fir_gallery* g, const void** gal4, int* dp4);   // the tiled f32 gallery (fir_kernels.h layout)
