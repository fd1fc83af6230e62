// Row a-1 of the path on the GPU: SigLipImageProcessor.preprocess (llava/model/multimodal_encoder/siglip_encoder.py:47-67)
//   uint8 [H][W][3] camera frame -> PIL bicubic resize to S x S (aspect not preserved) -> x/255 -> (x - 0.5)/0.5 -> fp32 [3][S][S]
// reproduced BIT-EXACTLY.  The resize is Pillow's two-pass fixed-point resampler (third-party: pillow==11.2.1 in the reference's
// requirements.txt:97, src/libImaging/Resample.c; restated in oracle/pil_bicubic.py): per axis, weights are computed in double
// precision on the host (build_resample_table below, same operation order as precompute_coeffs / normalize_coeffs_8bpc), rounded to
// 22-bit fixed point, and every output byte is clip8((2^21 + sum pixel * k) >> 22): horizontal pass over the input rows first, then
// the vertical pass over its uint8 result.  Rescale + normalise map each of the 256 byte levels to one fp32 value (host-built table,
// the fp64-multiply / fp32-cast / fp32-subtract-divide sequence of transformers' rescale / normalize).
//
// The frame comes up from pinned host staging through upload_kernel below (no runtime copy), then
// one launch: a workgroup owns one output row of one frame.  Its <= ks_v source rows are one contiguous byte range of the frame:
// staged into LDS with 16-byte loads, resampled horizontally into an LDS byte image [ks_v][3 S], then combined vertically; the
// three channel rows go out as full contiguous fp32 rows (CHW).  Integer work, HBM/L2-bound: 0.92 MB in, 1.77 MB out per frame.
#include <cmath>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace svln {

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;      // Resample.c

SVLN_DEV int clip8(int v) {
    v >>= PRECISION_BITS;                        // arithmetic shift, then saturate (clip8_lookups)
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

template <int MAX_V>
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ rgb, float* __restrict__ out, const int* __restrict__ hmin,
                                                         const int* __restrict__ hcnt, const int* __restrict__ hk, int ks_h,
                                                         const int* __restrict__ vmin, const int* __restrict__ vcnt, const int* __restrict__ vk,
                                                         int ks_v, const float* __restrict__ lut, int H, int W, int S, int rows_cap_bytes) {
    extern __shared__ uint4 smem[];
    uint8_t* s_rows = (uint8_t*)smem;                         // staged source rows (+ up to 15 leading bytes of alignment slack)
    uint8_t* s_hor = s_rows + rows_cap_bytes;                 // [ks_v][3 S] horizontally resampled bytes
    float* s_out = (float*)(s_hor + ((ks_v * 3 * S + 15) & ~15));   // [3][S]
    int* s_hmin = (int*)(s_out + 3 * S);                      // the horizontal tables, staged with the rows in the same round trip:
    int* s_hcnt = s_hmin + S;                                 //   a lookup per output byte from global memory would put two dependent
    int* s_hk = s_hcnt + S;                                   //   L2 round trips into every iteration of the horizontal pass
    float* s_lut = (float*)(s_hk + S * ks_h);
    const int yy = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const int S3 = 3 * S, rowb = 3 * W;
    const int y0 = vmin[yy], cnt = vcnt[yy];
    // ---- stage rows y0 .. y0+cnt-1 (contiguous in memory) and the tables
    const size_t off = ((size_t)f * H + y0) * rowb;
    const size_t a0 = off & ~(size_t)15;
    const int shift = (int)(off - a0);
    const int nchunks = (shift + cnt * rowb + 15) >> 4;
    for (int i = tid; i < nchunks; i += 256) smem[i] = *(const uint4*)(rgb + a0 + (size_t)i * 16);
    for (int i = tid; i < S; i += 256) { s_hmin[i] = hmin[i]; s_hcnt[i] = hcnt[i]; }
    for (int i = tid; i < S * ks_h; i += 256) s_hk[i] = hk[i];
    s_lut[tid] = lut[tid];
    int kvr[MAX_V];                     // this output row's vertical weights (workgroup-uniform)
#pragma unroll
    for (int r = 0; r < MAX_V; ++r) kvr[r] = r < cnt ? vk[yy * ks_v + r] : 0;
    __syncthreads();
    // ---- horizontal pass (ImagingResampleHorizontal_8bpc): a thread owns output bytes j = tid, tid + 256, ... of EVERY staged row, so
    // the byte's column, channel, window and weights are set up once and reused for the <= ks_v rows
    for (int j = tid; j < S3; j += 256) {
        const int xx = j / 3, c = j - 3 * xx;
        const uint8_t* px = s_rows + shift + s_hmin[xx] * 3 + c;
        const int* k = s_hk + xx * ks_h;
        const int n = s_hcnt[xx];
        int acc[MAX_V];
#pragma unroll
        for (int r = 0; r < MAX_V; ++r) acc[r] = 1 << (PRECISION_BITS - 1);
        for (int t = 0; t < n; ++t) {
            const int kt = k[t];
#pragma unroll
            for (int r = 0; r < MAX_V; ++r)
                if (r < cnt) acc[r] += (int)px[r * rowb + 3 * t] * kt;
        }
#pragma unroll
        for (int r = 0; r < MAX_V; ++r)
            if (r < cnt) s_hor[r * S3 + j] = (uint8_t)clip8(acc[r]);
    }
    __syncthreads();
    // ---- vertical pass (ImagingResampleVertical_8bpc) + rescale / normalise table, channel-planar
    for (int j = tid; j < S3; j += 256) {
        int acc = 1 << (PRECISION_BITS - 1);
#pragma unroll
        for (int r = 0; r < MAX_V; ++r)
            if (r < cnt) acc += (int)s_hor[r * S3 + j] * kvr[r];
        const int xx = j / 3, c = j - 3 * xx;
        s_out[c * S + xx] = s_lut[clip8(acc)];
    }
    __syncthreads();
    for (int i = tid; i < S3; i += 256) {
        const int c = i / S, xx = i - c * S;
        out[(((size_t)f * 3 + c) * S + yy) * S + xx] = s_out[i];
    }
}

// frame upload as a kernel: 16-byte loads straight from the pinned (device-mapped) staging buffer, 16-byte stores to HBM -- one launch
// instead of a runtime copy (hipMemcpyAsync from pinned memory stalled its caller for ~7 ms every few dozen calls on this runtime)
__global__ __launch_bounds__(256) void upload_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

double bicubic_filter(double x) {                            // Resample.c bicubic_filter, a = -0.5
#pragma clang fp contract(off)
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

}  // namespace

// precompute_coeffs + normalize_coeffs_8bpc of Resample.c for the whole-image box (0, in_size), bicubic.
void build_resample_table(int in_size, int out_size, ResampleAxis& ax) {
#pragma clang fp contract(off)
    const double in0 = 0.0, in1 = (double)(float)in_size;
    double scale, filterscale;
    filterscale = scale = (in1 - in0) / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 2.0 * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    ax.ksize = ksize;
    ax.xmin.assign(out_size, 0); ax.cnt.assign(out_size, 0); ax.k.assign((size_t)out_size * ksize, 0);
    std::vector<double> w(ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = in0 + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) { w[x] = bicubic_filter((x + xmin - center + 0.5) * ss); ww += w[x]; }
        for (int x = 0; x < xmax; ++x) {
            const double v = ww != 0.0 ? w[x] / ww : w[x];
            ax.k[(size_t)xx * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
        }
        ax.xmin[xx] = xmin; ax.cnt[xx] = xmax;
    }
}

// rescale (uint8 * (1/255) in fp64 -> fp32) then normalize ((v - mean) / std in fp32), siglip_encoder.py:58-60
void build_normalize_lut(float* lut256, float mean, float std) {
#pragma clang fp contract(off)
    for (int i = 0; i < 256; ++i) {
        const float a = (float)((double)i * (1.0 / 255.0));
        lut256[i] = (a - mean) / std;
    }
}

size_t preprocess_lds_bytes(int W, int S, int ks_v, int ks_h) {
    const size_t rows = (((size_t)ks_v * 3 * W + 15 + 15) & ~(size_t)15) + 16;
    return rows + (((size_t)ks_v * 3 * S + 15) & ~(size_t)15) + (size_t)3 * S * sizeof(float)       // staged rows, horizontal bytes, output row
           + (size_t)S * (2 + ks_h) * sizeof(int) + 256 * sizeof(float);                            // horizontal tables, normalise table
}

void launch_preprocess(hipStream_t s, const uint8_t* rgb, float* out, int n_frames, int H, int W, int S, const ResampleDev& t, const float* lut) {
    const int rows_cap = (int)((((size_t)t.ks_v * 3 * W + 15 + 15) & ~(size_t)15) + 16);
    const size_t lds = preprocess_lds_bytes(W, S, t.ks_v, t.ks_h);
    // the vertical window (ks_v source rows per output row) is a compile-time bound of the per-thread accumulators
    if (t.ks_v <= 8)
        hipLaunchKernelGGL(preprocess_kernel<8>, dim3(S, n_frames), dim3(256), lds, s, rgb, out, t.hmin, t.hcnt, t.hk, t.ks_h, t.vmin, t.vcnt, t.vk, t.ks_v,
                           lut, H, W, S, rows_cap);
    else if (t.ks_v <= 16)
        hipLaunchKernelGGL(preprocess_kernel<16>, dim3(S, n_frames), dim3(256), lds, s, rgb, out, t.hmin, t.hcnt, t.hk, t.ks_h, t.vmin, t.vcnt, t.vk, t.ks_v,
                           lut, H, W, S, rows_cap);
    else
        hipLaunchKernelGGL(preprocess_kernel<40>, dim3(S, n_frames), dim3(256), lds, s, rgb, out, t.hmin, t.hcnt, t.hk, t.ks_h, t.vmin, t.vcnt, t.vk, t.ks_v,
                           lut, H, W, S, rows_cap);
}

// bytes rounded up to 16: both buffers are allocated with that slack
void launch_upload(hipStream_t s, const void* src_dev_visible, void* dst, size_t bytes) {
    const size_t n16 = (bytes + 15) / 16;
    hipLaunchKernelGGL(upload_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, (const uint4*)src_dev_visible, (uint4*)dst, n16);
}

void preprocess_init_attrs() {
    set_max_lds((const void*)preprocess_kernel<8>, 160 * 1024);
    set_max_lds((const void*)preprocess_kernel<16>, 160 * 1024);
    set_max_lds((const void*)preprocess_kernel<40>, 160 * 1024);
}

}  // namespace svln
