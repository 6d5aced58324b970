// Input stage on the device: the reference's image transforms (sparsify_clip.py:1003-1016) from raw uint8 RGB pixels to the
// normalised fp32 [N,3,S,S] tensor the image tower reads -
//     train: RandomResizedCrop((224,224)) -> RandomHorizontalFlip -> ToTensor -> Normalize(mean, std)
//     test : Resize((224,224))                                     -> ToTensor -> Normalize(mean, std)
// torchvision applies these to PIL images, so "resize" is Pillow's antialiased two-pass resampler (ImagingResample, BILINEAR):
// a triangle filter whose support grows with the down-scaling factor, coefficients rounded to 22-bit fixed point, a horizontal
// pass to an 8-BIT intermediate image, then a vertical pass, each with Pillow's rounding.  The kernels below restate exactly
// that arithmetic (double-precision coefficient set-up, int32 accumulation), so the result is bit-identical to the reference's
// CPU pipeline for the same crop boxes / flips; the random boxes themselves are drawn on the host (input_pipeline.py).
// HBM-bound: one byte triple in, a few taps, 12 bytes out per pixel; each image is swept by consecutive lanes along x.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;   // Pillow: Resample.c

struct Taps {
    int xmin, count;
};

// Pillow's precompute_coeffs() + normalize_coeffs_8bpc() for ONE output index: fixed-point weights k[0..count) of the input
// samples xmin .. xmin + count - 1.  in_size = length of the (already cropped) input axis, out_size = S.
template <int MAXK>
__device__ __forceinline__ Taps pillow_bilinear_taps(int in_size, int out_size, int xx, int (&k)[MAXK]) {
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const double center = (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > MAXK) xmax = MAXK;   // never taken for the sizes the host admits (checked there)
    double w[MAXK];
    double ww = 0.0;
#pragma unroll 1
    for (int x = 0; x < xmax; ++x) {
        double v = (x + xmin - center + 0.5) * ss;
        v = v < 0.0 ? -v : v;
        const double f = v < 1.0 ? 1.0 - v : 0.0;
        w[x] = f;
        ww += f;
    }
#pragma unroll 1
    for (int x = 0; x < xmax; ++x) {
        double f = w[x];
        if (ww != 0.0) f /= ww;
        k[x] = (int)(0.5 + f * (double)(1 << PRECISION_BITS));
    }
    return Taps{xmin, xmax};
}

__device__ __forceinline__ unsigned char clip8(int v) {
    // Pillow's clip8 lookup: (v >> PRECISION_BITS) clamped to [0, 255]
    const int r = v >> PRECISION_BITS;
    return (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

constexpr int MAXK = 64;   // taps per output sample: 2 * ceil(in / out) + 1 <= 64 admits a 31x down-scale (6944 px -> 224)

// dims[n] = {H, W, top, left, h, w}: full image size and the crop box; tmp holds per image box_h x S x 3 bytes at tmp_off[n]
__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* src, const int64_t* offset, const int* dims, const int64_t* tmp_off,
                                                         int S, unsigned char* tmp) {
    const int n = blockIdx.y;
    const int* d = dims + 6 * n;
    const int W = d[1], top = d[2], left = d[3], bh = d[4], bw = d[5];
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)bh * S) return;
    const int y = (int)(idx / S), xo = (int)(idx - (int64_t)y * S);
    int k[MAXK];
    const Taps t = pillow_bilinear_taps<MAXK>(bw, S, xo, k);
    const unsigned char* row = src + offset[n] + ((int64_t)(top + y) * W + left + t.xmin) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
#pragma unroll 1
    for (int x = 0; x < t.count; ++x) {
        s0 += (int)row[3 * x + 0] * k[x];
        s1 += (int)row[3 * x + 1] * k[x];
        s2 += (int)row[3 * x + 2] * k[x];
    }
    unsigned char* o = tmp + tmp_off[n] + ((int64_t)y * S + xo) * 3;
    o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

__global__ __launch_bounds__(256) void resample_v_normalize_kernel(const unsigned char* tmp, const int64_t* tmp_off, const int* dims, const int* flip, int S,
                                                                   float m0, float m1, float m2, float d0, float d1, float d2, float* out) {
    const int n = blockIdx.y;
    const int bh = dims[6 * n + 4];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= S * S) return;
    const int yo = idx / S, xo = idx - yo * S;
    int k[MAXK];
    const Taps t = pillow_bilinear_taps<MAXK>(bh, S, yo, k);
    const unsigned char* col = tmp + tmp_off[n] + ((int64_t)t.xmin * S + xo) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
#pragma unroll 1
    for (int y = 0; y < t.count; ++y) {
        const unsigned char* p = col + (int64_t)y * S * 3;
        s0 += (int)p[0] * k[y];
        s1 += (int)p[1] * k[y];
        s2 += (int)p[2] * k[y];
    }
    // RandomHorizontalFlip acts on the resized image; ToTensor: uint8 / 255; Normalize: (x - mean) / std, all in fp32 (IEEE division)
    const int xw = (flip && flip[n]) ? S - 1 - xo : xo;
    float* o = out + (int64_t)n * 3 * S * S + (int64_t)yo * S + xw;
    o[0] = ((float)clip8(s0) / 255.0f - m0) / d0;
    o[(int64_t)S * S] = ((float)clip8(s1) / 255.0f - m1) / d1;
    o[(int64_t)2 * S * S] = ((float)clip8(s2) / 255.0f - m2) / d2;
}

}  // namespace

extern "C" int sc_image_resample_normalize(const uint8_t* src, const int64_t* offset, const int32_t* dims, const int32_t* flip, const int64_t* tmp_offset,
                                           int64_t n, int64_t max_box_h, int64_t out_size, float mean_r, float mean_g, float mean_b, float std_r,
                                           float std_g, float std_b, void* tmp, float* out, void* stream) {
    SC_REQUIRE(src && offset && dims && tmp_offset && tmp && out, SC_ERR_ARG, "sc_image_resample_normalize: null argument");
    SC_REQUIRE(n > 0 && n < 65536 && out_size > 0 && out_size <= 4096 && max_box_h > 0 && max_box_h <= 31 * out_size, SC_ERR_SHAPE,
               "sc_image_resample_normalize: bad sizes (n %lld, out %lld, max box height %lld: at most a 31x down-scale)", (long long)n,
               (long long)out_size, (long long)max_box_h);
    SC_REQUIRE(std_r != 0.f && std_g != 0.f && std_b != 0.f, SC_ERR_ARG, "sc_image_resample_normalize: zero std");
    hipStream_t st = (hipStream_t)stream;
    const int S = (int)out_size;
    hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)sc_cdiv(max_box_h * S, 256), (unsigned)n), dim3(256), 0, st, src, offset, dims, tmp_offset, S,
                       (unsigned char*)tmp);
    hipLaunchKernelGGL(resample_v_normalize_kernel, dim3((unsigned)sc_cdiv((int64_t)S * S, 256), (unsigned)n), dim3(256), 0, st, (const unsigned char*)tmp,
                       tmp_offset, dims, flip, S, mean_r, mean_g, mean_b, std_r, std_g, std_b, out);
    SC_CHECK_LAUNCH();
    return SC_OK;
}
