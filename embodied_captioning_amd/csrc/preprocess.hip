// Crop + resize of object boxes on the device, bit-exact with Pillow's bicubic resample for 8-bit RGB
// (`Image.crop(box).resize((S, S), Image.BICUBIC)` - what the reference's HF image processor does to every crop:
// reference detector/pseudolabeler.py:670-675 + BlipImageProcessor.resize).  Pillow's arithmetic is integer: two separable
// passes, horizontal first, each value = clip8((2^21 + sum_k pixel_k * coeff_k) >> 22), the horizontal result rounded to
// uint8 before the vertical pass reads it.  The integer coefficient tables (round-half-away of the normalised double
// weights, Resample.c: precompute_coeffs / normalize_coeffs_8bpc) are built on the host per (crop size, S) - O(S) work in
// doubles whose rounding must match the CPU's - and everything per pixel happens here.
//
// A box may leave the frame (the reference's expand_box clamps x to the frame HEIGHT and y to its WIDTH, so on a non-square
// frame it does): Image.crop pads with zeros, and so does the kernel.
//
// One workgroup per (output row, box); a thread owns output columns.  The horizontal values a column needs are recomputed
// for every vertical tap (taps ~ (4 scale + 1)^2 per pixel): a few hundred integer MACs per output pixel, byte loads from
// a frame that sits in L2.  HBM-bound nowhere; this kernel exists to take ~0.3 ms of host PIL time per crop off the loop.
#include "common.h"
#include "ops.h"

namespace {

constexpr int PR_BITS = 22;

__device__ __forceinline__ int clip8(int v) { return min(max(v >> PR_BITS, 0), 255); }

// frames != null: box b is cut from ITS OWN frame (a list of crops of different sizes in one packed buffer): frames[b] = (byte
// offset of the frame in `frame`, its height, its width) - one upload and one launch for the whole list.
__global__ __launch_bounds__(256) void crop_resize_u8_kernel(const uint8_t* __restrict__ frame, int H, int W, int bgr,
                                                             const int* __restrict__ rects, const int* __restrict__ hb,
                                                             const int* __restrict__ hk, int KH, const int* __restrict__ vb,
                                                             const int* __restrict__ vk, int KV, int S,
                                                             uint8_t* __restrict__ out, const long long* __restrict__ frames) {
    const int yy = blockIdx.x, b = blockIdx.y;
    if (frames) {
        frame += frames[b * 3];
        H = (int)frames[b * 3 + 1];
        W = (int)frames[b * 3 + 2];
    }
    const int x1 = rects[b * 4 + 0], y1 = rects[b * 4 + 1];
    const int ymin = vb[((size_t)b * S + yy) * 2], ycnt = vb[((size_t)b * S + yy) * 2 + 1];
    const int* kv = vk + ((size_t)b * S + yy) * KV;
    const int c0 = bgr ? 2 : 0, cs = bgr ? -1 : 1;          // output channel c reads input channel c0 + cs * c
    for (int xx = threadIdx.x; xx < S; xx += blockDim.x) {
        const int xmin = hb[((size_t)b * S + xx) * 2], xcnt = hb[((size_t)b * S + xx) * 2 + 1];
        const int* kh = hk + ((size_t)b * S + xx) * KH;
        int v0 = 1 << (PR_BITS - 1), v1 = v0, v2 = v0;
        for (int ty = 0; ty < ycnt; ++ty) {
            const int y = y1 + ymin + ty;
            if (y < 0 || y >= H) continue;          // Image.crop pads a box that leaves the frame with zeros: clip8(2^21) = 0
            const int xs = x1 + xmin;
            const uint8_t* src = frame + ((ptrdiff_t)y * W + xs) * 3;
            int h0 = 1 << (PR_BITS - 1), h1 = h0, h2 = h0;
            for (int tx = 0; tx < xcnt; ++tx) {
                if (xs + tx < 0 || xs + tx >= W) continue;
                const int k = kh[tx];
                h0 += (int)src[tx * 3 + c0] * k;
                h1 += (int)src[tx * 3 + 1] * k;
                h2 += (int)src[tx * 3 + c0 + 2 * cs] * k;
            }
            const int w = kv[ty];
            v0 += clip8(h0) * w; v1 += clip8(h1) * w; v2 += clip8(h2) * w;
        }
        uint8_t* o = out + (((size_t)b * S + yy) * S + xx) * 3;
        o[0] = (uint8_t)clip8(v0); o[1] = (uint8_t)clip8(v1); o[2] = (uint8_t)clip8(v2);
    }
}

// Pillow's coefficient tables on the device (Resample.c: precompute_coeffs + normalize_coeffs_8bpc), one thread per
// (box, axis, output index).  Doubles, the same operations in the same order as the C source, and NO fused multiply-add:
// Pillow's wheels are plain x86-64 (every product and sum rounded separately), and a contracted a*b+c would round once.
// gfx950's fp64 add / mul / div are IEEE, so the tables equal the host's bit for bit (tests/test_preprocess_gpu.py).
// geom[b] = (resized width, resized height, left, top): the S x S output is the window [left, left+S) x [top, top+S) of the
// crop resized to (width, height) - (S, S, 0, 0) for the plain square resize.
__device__ __forceinline__ double pil_bicubic(double x) {
#pragma clang fp contract(off)
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

__global__ __launch_bounds__(256) void crop_resize_tables_kernel(const int* __restrict__ rects, const int* __restrict__ geom, int n,
                                                                 int S, int KH, int KV, int* __restrict__ hb, int* __restrict__ hk,
                                                                 int* __restrict__ vb, int* __restrict__ vk) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y, axis = blockIdx.z;
    if (i >= S) return;
    const int in_size = rects[b * 4 + 2 + axis] - rects[b * 4 + axis];
    const int out_size = geom[b * 4 + axis], xx = i + geom[b * 4 + 2 + axis];
    const int K = axis ? KV : KH;
    int* bounds = (axis ? vb : hb) + ((size_t)b * S + i) * 2;
    int* k = (axis ? vk : hk) + ((size_t)b * S + i) * K;
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale, ss = 1.0 / filterscale;
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > K) xmax = K;                 // cannot happen when the host sized K = 2 ceil(support) + 1; never write past the row
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += pil_bicubic((x + xmin - center + 0.5) * ss);
    for (int x = 0; x < K; ++x) {
        int v = 0;
        if (x < xmax) {
            double w = pil_bicubic((x + xmin - center + 0.5) * ss);
            if (ww != 0.0) w /= ww;
            v = w < 0 ? (int)(-0.5 + w * (double)(1 << PR_BITS)) : (int)(0.5 + w * (double)(1 << PR_BITS));
        }
        k[x] = v;
    }
    bounds[0] = xmin; bounds[1] = xmax;
}

}  // namespace

int launch_crop_resize_tables(const int* rects, const int* geom, int n, int S, int KH, int KV, int* hb, int* hk, int* vb, int* vk,
                              hipStream_t s) {
    if (!rects || !geom || !hb || !hk || !vb || !vk || n < 1 || S < 1 || S > 4096 || KH < 1 || KV < 1) {
        cap_set_error("crop_resize_tables: null pointer or bad shape (n=%d S=%d KH=%d KV=%d)", n, S, KH, KV);
        return -1;
    }
    hipLaunchKernelGGL(crop_resize_tables_kernel, dim3((S + 255) / 256, n, 2), dim3(256), 0, s, rects, geom, n, S, KH, KV, hb, hk, vb, vk);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_crop_resize_u8(const uint8_t* frame, int H, int W, int bgr, const int* rects, const int* hb, const int* hk, int KH,
                          const int* vb, const int* vk, int KV, int n, int S, uint8_t* out, hipStream_t s, const long long* frames) {
    if (!frame || !rects || !hb || !hk || !vb || !vk || !out || (!frames && (H < 1 || W < 1)) || n < 1 || S < 1 || S > 4096 || KH < 1 || KV < 1) {
        cap_set_error("crop_resize: null pointer or bad shape (H=%d W=%d n=%d S=%d KH=%d KV=%d)", H, W, n, S, KH, KV);
        return -1;
    }
    hipLaunchKernelGGL(crop_resize_u8_kernel, dim3(S, n), dim3(256), 0, s, frame, H, W, bgr, rects, hb, hk, KH, vb, vk, KV, S, out, frames);
    CAP_HIP_CHECK(hipGetLastError());
    return 0;
}
