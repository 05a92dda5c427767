// Image input pipeline on device: decoded uint8 HWC image -> CLIP pixel_values in bf16, or directly the
// patch-embedding GEMM's im2col operand.  Replaces, per image, the reference's CPU chain
//   [expand2square] -> PIL bicubic resize (shortest edge) -> center crop -> * 1/255 -> (x - mean) / std -> CHW -> .to(bf16)
// (/root/reference/libra/models/clip/image_processing_clip.py:219-337, libra/data/datasets/caption_datasets.py:45-56).
// The resize is Pillow's two-pass 8-bit resampler (ImagingResample): fixed-point taps (22 fractional bits) computed on the host
// exactly as Pillow does (libra_amd/clip/image_pipeline.py), integer accumulation and clip8 here - bit-exact by construction.
// HBM-bound byte work: 3 B read per source pixel tap-row, 2 B written per output element.
#include "hip_common.hpp"
#include "../../include/libra_hip.h"

namespace libra {

constexpr int RS_BITS = 22;

struct ResampleHArgs {
    const unsigned char* in; int in_h, in_w;           // the decoded image [in_h, in_w, 3]
    int pad_y, pad_x;                                   // its offset inside the (virtual) square canvas; 0, 0 without padding
    unsigned char bg[4];                                // canvas colour
    const int* bounds; const int* coeffs; int ksize;    // [out_w][2], [out_w][ksize]
    unsigned char* out; int rows, out_w, row0;          // out [rows, out_w, 3] = canvas rows row0 .. row0 + rows
};
__global__ __launch_bounds__(256) void resample_h_kernel(const ResampleHArgs p) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)p.rows * p.out_w) return;
    const int r = (int)(i / p.out_w), xx = (int)(i - (long)r * p.out_w);
    const int y = p.row0 + r - p.pad_y;                 // row inside the image (may be outside: canvas colour)
    const int x0 = p.bounds[2 * xx], n = p.bounds[2 * xx + 1];
    const int* k = p.coeffs + (long)xx * p.ksize;
    int a0 = 1 << (RS_BITS - 1), a1 = a0, a2 = a0;
    const bool yin = y >= 0 && y < p.in_h;
    const unsigned char* row = p.in + (long)(yin ? y : 0) * p.in_w * 3;
    for (int t = 0; t < n; ++t) {
        const int x = x0 + t - p.pad_x;
        const int c = k[t];
        if (yin && x >= 0 && x < p.in_w) { a0 += row[x * 3] * c; a1 += row[x * 3 + 1] * c; a2 += row[x * 3 + 2] * c; }
        else { a0 += p.bg[0] * c; a1 += p.bg[1] * c; a2 += p.bg[2] * c; }
    }
    unsigned char* o = p.out + i * 3;
    o[0] = (unsigned char)min(max(a0 >> RS_BITS, 0), 255);
    o[1] = (unsigned char)min(max(a1 >> RS_BITS, 0), 255);
    o[2] = (unsigned char)min(max(a2 >> RS_BITS, 0), 255);
}

struct ResampleVArgs {
    const unsigned char* tmp; int tmp_w, row0;          // [rows, tmp_w, 3], first canvas row held
    const int* bounds; const int* coeffs; int ksize;    // vertical taps per OUTPUT row of the resized image
    int top, left, crop;                                // center crop window inside the resized image
    const bf16_t* lut;                                  // [3][256]: uint8 level -> normalised bf16 value
    bf16_t* out; int patch, kpad;                       // patch == 0: NCHW [3, crop, crop]; else im2col rows [(crop/patch)^2, kpad]
};
__global__ __launch_bounds__(256) void resample_v_kernel(const ResampleVArgs p) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)p.crop * p.crop) return;
    const int cy = (int)(i / p.crop), cx = (int)(i - (long)cy * p.crop);
    const int yy = cy + p.top, xx = cx + p.left;
    const int y0 = p.bounds[2 * yy], n = p.bounds[2 * yy + 1];
    const int* k = p.coeffs + (long)yy * p.ksize;
    int a0 = 1 << (RS_BITS - 1), a1 = a0, a2 = a0;
    const unsigned char* col = p.tmp + ((long)(y0 - p.row0) * p.tmp_w + xx) * 3;
    for (int t = 0; t < n; ++t) {
        const int c = k[t];
        a0 += col[0] * c; a1 += col[1] * c; a2 += col[2] * c;
        col += (long)p.tmp_w * 3;
    }
    const int v[3] = {min(max(a0 >> RS_BITS, 0), 255), min(max(a1 >> RS_BITS, 0), 255), min(max(a2 >> RS_BITS, 0), 255)};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const bf16_t val = p.lut[c * 256 + v[c]];
        if (p.patch == 0) p.out[((long)c * p.crop + cy) * p.crop + cx] = val;
        else {
            const int g = p.crop / p.patch, gy = cy / p.patch, gx = cx / p.patch;
            p.out[((long)gy * g + gx) * p.kpad + c * p.patch * p.patch + (cy - gy * p.patch) * p.patch + (cx - gx * p.patch)] = val;
        }
    }
}

}  // namespace libra

using namespace libra;

extern "C" int libra_resample_h_u8(const uint8_t* in, int64_t in_h, int64_t in_w, int64_t pad_y, int64_t pad_x, int bg_r, int bg_g,
                                   int bg_b, const int32_t* bounds, const int32_t* coeffs, int64_t ksize, uint8_t* out, int64_t rows,
                                   int64_t out_w, int64_t row0, void* stream) {
    if (rows <= 0 || out_w <= 0) return LIBRA_OK;
    if (in_h <= 0 || in_w <= 0 || ksize <= 0 || pad_y < 0 || pad_x < 0 || row0 < 0) return LIBRA_ERR_SHAPE;
    if (!in || !bounds || !coeffs || !out) return LIBRA_ERR_ALIGN;
    ResampleHArgs a;
    a.in = in; a.in_h = (int)in_h; a.in_w = (int)in_w; a.pad_y = (int)pad_y; a.pad_x = (int)pad_x;
    a.bg[0] = (unsigned char)bg_r; a.bg[1] = (unsigned char)bg_g; a.bg[2] = (unsigned char)bg_b; a.bg[3] = 0;
    a.bounds = bounds; a.coeffs = coeffs; a.ksize = (int)ksize; a.out = out; a.rows = (int)rows; a.out_w = (int)out_w; a.row0 = (int)row0;
    const long n = rows * out_w;
    hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}

extern "C" int libra_resample_v_u8_norm(const uint8_t* tmp, int64_t tmp_w, int64_t row0, const int32_t* bounds, const int32_t* coeffs,
                                        int64_t ksize, int64_t top, int64_t left, int64_t crop, const void* lut, void* out,
                                        int64_t patch, int64_t kpad, void* stream) {
    if (crop <= 0) return LIBRA_OK;
    if (tmp_w < left + crop || ksize <= 0 || top < 0 || left < 0 || row0 < 0) return LIBRA_ERR_SHAPE;
    if (patch < 0 || (patch > 0 && (crop % patch || kpad < 3 * patch * patch))) return LIBRA_ERR_SHAPE;
    if (!tmp || !bounds || !coeffs || !lut || !out) return LIBRA_ERR_ALIGN;
    ResampleVArgs a;
    a.tmp = tmp; a.tmp_w = (int)tmp_w; a.row0 = (int)row0; a.bounds = bounds; a.coeffs = coeffs; a.ksize = (int)ksize;
    a.top = (int)top; a.left = (int)left; a.crop = (int)crop; a.lut = (const bf16_t*)lut; a.out = (bf16_t*)out;
    a.patch = (int)patch; a.kpad = (int)kpad;
    const long n = crop * crop;
    hipLaunchKernelGGL(resample_v_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? LIBRA_OK : LIBRA_ERR_LAUNCH;
}
