// rotate-yolov3_amd/csrc/yolo.hip -- YOLO head decode and the small NHWC layout kernels around the conv stack.
//
// ryolo_yolo_decode replaces YOLOLayer.forward (model/models.py:183-227) + create_grids (model/model_utils.py:16-35):
//   p  = head.view(bs, na, nc+6, ny, nx).permute(0,1,3,4,2)            (models.py:189)    -> fp32 [bs,na,ny,nx,no]
//   io = p.clone(); xy = sigmoid(xy) + grid; wh = exp(wh) * anchor_wh; a = atan(a) + anchor_a;  (models.py:198-200)
//        io[..., :4] *= stride; h /= cf; w -= h*(cf-1);                                        (models.py:202-208)
//        sigmoid(io[..., 5:]) for 'default' arcs; io[..., 6] = 1 when nc == 1                  (models.py:210-221)
//   io.view(bs, -1, no)                                                                        (models.py:227)
// The head arrives as the NHWC bf16 output of the last 1x1 conv (channel = a*no + k); one thread per (pixel, anchor)
// reads its `no` contiguous channels and writes one io row (and one p row).  HBM-bound: 2*no B in, 8*no B out per row.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ryolo.h"
#include "yolo_decode.h"

using ryolo_detail::decode_row;
using ryolo_detail::sigmoidf;
using ryolo_detail::store_row;

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__global__ void yolo_decode_simple_kernel(const __bf16 *__restrict__ head, int cs, int bs, int ny, int nx, int na, int no,
                                   const float *__restrict__ anchors /* [na][3] (w_px, h_px, angle) */, float stride,
                                   float cf, int arc, float *__restrict__ io, long long io_img_rows, long long io_row0,
                                   float *__restrict__ p) {
    const long long total = (long long)bs * ny * nx * na;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int a = (int)(i % na);
        const long long pix = i / na;
        const int x = (int)(pix % nx);
        const long long t = pix / nx;
        const int y = (int)(t % ny);
        const long long n = t / ny;
        const __bf16 *src = head + pix * cs + (long long)a * no;
        const long long row = (long long)a * ny * nx + (long long)y * nx + x;
        float *o = io + ((n * io_img_rows) + io_row0 + row) * no;
        float *pp = p ? p + ((n * na * ny * nx) + row) * no : nullptr;
        float v[8];
#pragma unroll
        for (int k = 0; k < 7; k++) v[k] = (float)src[k];
        if (pp) {
#pragma unroll
            for (int k = 0; k < 7; k++) pp[k] = v[k];
        }
        // anchor_vec = anchors / stride (model_utils.py:30-31), then io[..., :4] *= stride (models.py:203)
        const float aw = anchors[a * 3 + 0] / stride, ah = anchors[a * 3 + 1] / stride, aa = anchors[a * 3 + 2];
        float bx = (sigmoidf(v[0]) + (float)x) * stride;
        float by = (sigmoidf(v[1]) + (float)y) * stride;
        float bw = (expf(v[2]) * aw) * stride;
        float bh = (expf(v[3]) * ah) * stride;
        const float ba = atanf(v[4]) + aa;
        bh = bh / cf;
        bw = bw - bh * (cf - 1.f);
        o[0] = bx; o[1] = by; o[2] = bw; o[3] = bh; o[4] = ba;
        if (arc == 0) {            // 'default' arcs: sigmoid on obj + cls
            o[5] = sigmoidf(v[5]);
            for (int k = 6; k < no; k++) o[k] = sigmoidf((float)src[k]);
        } else if (arc == 1) {     // 'BCE': sigmoid on cls, obj = 1
            o[5] = 1.f;
            for (int k = 6; k < no; k++) o[k] = sigmoidf((float)src[k]);
        } else {                   // 'CE': softmax over [obj(bg), cls...] taken from channel 4.. (models.py:216-218)
            // io[..., 5:] = softmax(io[..., 4:], dim=4)[..., 1:] is ill-formed in the reference (shape mismatch
            // unless broadcasting); not reachable with the shipped cfgs -> plain softmax over channels 5.., obj = 1
            float mx = -3.4e38f;
            for (int k = 5; k < no; k++) mx = fmaxf(mx, (float)src[k]);
            float sum = 0.f;
            for (int k = 5; k < no; k++) sum += expf((float)src[k] - mx);
            for (int k = 6; k < no; k++) o[k] = expf((float)src[k] - mx) / sum;
            o[5] = 1.f;
        }
        if (pp) for (int k = 7; k < no; k++) pp[k] = (float)src[k];
        if (no == 7) o[6] = 1.f;   // nc == 1 (models.py:220-221)
    }
}


// Tiled variant (the one that normally runs): a 256-thread workgroup owns DEC_PIX consecutive pixels of one head.
// Their na*no channels are one contiguous run in the NHWC head, so they are staged into LDS with coalesced 16-B
// loads; then thread (a, x) decodes anchor a of pixel x with x fastest, so that the 28-B io rows a wave writes are
// contiguous (rows of one anchor over consecutive pixels are adjacent in io / p).  Both sides of the transposition
// are coalesced; the simple kernel above reads or writes at a 1-KiB stride.
constexpr int DEC_PIX = 32;

// Decode + confidence filter + stream compaction: what YOLOLayer.forward (models.py:198-227) followed by
// the first half of non_max_suppression (utils/nms/nms.py:33-48) keeps of a head, without materialising `io`:
// class_conf / class = max over the class columns (first maximum), score = obj * class_conf, survivors have
// score > conf_thres, w and h > min_wh and every entry finite.  Survivor rows (x, y, w, h, a, score, class_conf, class)
// are appended with an atomic counter, tagged with their row index in the concatenated `io` (image * rows_per_image +
// row_offset + row) so the caller can restore the reference's order by sorting on the tag.
constexpr int DEC_MAX_NO = 96;
__global__ void yolo_decode_filter_kernel(const __bf16 *__restrict__ head, int cs, int bs, int ny, int nx, int na, int no,
                                          const float *__restrict__ anchors, float stride, float cf, int arc,
                                          float conf_thres, float min_wh, long long rows_per_image, long long row0,
                                          float *__restrict__ cand, long long *__restrict__ cand_row,
                                          int *__restrict__ counter, int cap) {
    const long long total = (long long)bs * ny * nx * na;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int a = (int)(i % na);
        const long long pix = i / na;
        const int x = (int)(pix % nx);
        const long long t = pix / nx;
        const int y = (int)(t % ny);
        const long long n = t / ny;
        const __bf16 *src = head + pix * cs + (long long)a * no;
        // cheap early out on the objectness alone: score = obj * class_conf <= obj for the sigmoid arcs
        if (arc == 0 && !(sigmoidf((float)src[5]) > conf_thres)) continue;
        float v[DEC_MAX_NO], o[DEC_MAX_NO];
        for (int k = 0; k < no; k++) v[k] = (float)src[k];
        const float aw = anchors[a * 3 + 0] / stride, ah = anchors[a * 3 + 1] / stride, aa = anchors[a * 3 + 2];
        decode_row(v, no, x, y, aw, ah, aa, stride, cf, arc, o);
        float best = o[6];
        int bi = 0;
        for (int k = 7; k < no; k++)
            if (o[k] > best) { best = o[k]; bi = k - 6; }
        const float score = o[5] * best;
        bool ok = score > conf_thres && o[2] > min_wh && o[3] > min_wh && isfinite(score);
        for (int k = 0; k < no; k++) ok = ok && (k == 5 || isfinite(o[k]));
        if (!ok) continue;
        const int slot = atomicAdd(counter, 1);
        if (slot >= cap) continue;
        float *c = cand + (size_t)slot * 8;
        c[0] = o[0]; c[1] = o[1]; c[2] = o[2]; c[3] = o[3]; c[4] = o[4]; c[5] = score; c[6] = best; c[7] = (float)bi;
        cand_row[slot] = n * rows_per_image + row0 + (long long)a * ny * nx + (long long)y * nx + x;
    }
}

template <int NO>   // NO == 7: single class, fully unrolled; NO == 0: generic (no <= 96)
__global__ void __launch_bounds__(256)
yolo_decode_tiled_kernel(const __bf16 *__restrict__ head, int cs, long long npix_total, int ny, int nx, int na,
                         int no_rt, const float *__restrict__ anchors, float stride, float cf, int arc,
                         float *__restrict__ io, long long io_img_rows, long long io_row0, float *__restrict__ p) {
    extern __shared__ __attribute__((aligned(16))) char smem_dec[];
    __bf16 *tile = (__bf16 *)smem_dec;
    const int no = NO ? NO : no_rt;
    const int C = na * no;                 // channels per pixel actually used (multiple of 8 by the conv contract)
    const long long pix0 = (long long)blockIdx.x * DEC_PIX;
    const int npx = (int)min((long long)DEC_PIX, npix_total - pix0);
    const int cpr = C / 8;
    for (int i = threadIdx.x; i < npx * cpr; i += 256) {
        const int px = i / cpr, ch = (i % cpr) * 8;
        *(bf16x8 *)(tile + px * C + ch) = *(const bf16x8 *)(head + (pix0 + px) * cs + ch);
    }
    __syncthreads();
    const long long hw = (long long)ny * nx;
    for (int i = threadIdx.x; i < npx * na; i += 256) {
        const int px = i % npx;                            // x fastest within the tile
        const int a = i / npx;
        const long long pix = pix0 + px;
        // 32-bit arithmetic (npix < 2^31, host check): the 64-bit divisions of the first version were ~200 instructions per 28-byte row
        const unsigned n = (unsigned)pix / (unsigned)hw, rem = (unsigned)pix - n * (unsigned)hw;
        const int y = (int)(rem / (unsigned)nx), x = (int)(rem - (unsigned)y * (unsigned)nx);
        const __bf16 *src = tile + px * C + a * no;
        float v[NO ? NO : 96];
#pragma unroll
        for (int k = 0; k < (NO ? NO : 96); k++)
            if (k < no) v[k] = (float)src[k];
        const long long row = (long long)a * hw + rem;
        if (p) store_row(p + ((n * na * hw) + row) * no, v, no);
        if (io) {                      // training passes io = NULL: only the raw head p is needed (model_utils.py:27-28)
            const float aw = anchors[a * 3 + 0] / stride, ah = anchors[a * 3 + 1] / stride, aa = anchors[a * 3 + 2];
            decode_row<(NO ? NO : 96)>(v, no, x, y, aw, ah, aa, stride, cf, arc, io + ((n * io_img_rows) + io_row0 + row) * no);
        }
    }
}

// ---- fallbacks for cfg graphs whose shortcut / upsample / route cannot be fused into a conv epilogue
__global__ void add_kernel(const __bf16 *__restrict__ a, int a_cs, const __bf16 *__restrict__ b, int b_cs,
                           __bf16 *__restrict__ y, int y_cs, long long npix, int C) {
    const int cpr = C / 8;
    const long long total = npix * cpr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long pix = i / cpr;
        const int c = (int)(i % cpr) * 8;
        const bf16x8 va = *(const bf16x8 *)(a + pix * a_cs + c);
        const bf16x8 vb = *(const bf16x8 *)(b + pix * b_cs + c);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = (__bf16)((float)va[e] + (float)vb[e]);
        *(bf16x8 *)(y + pix * y_cs + c) = o;
    }
}

// nearest upsample by `s` (s == 1: plain slice copy); y is [N, H*s, W*s] pixels
__global__ void upsample_copy_kernel(const __bf16 *__restrict__ x, int x_cs, __bf16 *__restrict__ y, int y_cs, int N,
                                     int H, int W, int C, int s) {
    const int cpr = C / 8;
    const long long total = (long long)N * H * s * W * s * cpr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpr) * 8;
        const long long opix = i / cpr;
        const int ox = (int)(opix % (W * s));
        const long long t = opix / (W * s);
        const int oy = (int)(t % (H * s));
        const long long n = t / (H * s);
        const long long ipix = (n * H + oy / s) * W + ox / s;
        *(bf16x8 *)(y + opix * y_cs + c) = *(const bf16x8 *)(x + ipix * x_cs + c);
    }
}

// maxpool k x k, stride s, darknet/ultralytics padding: pad_lo = (k-1)//2 on both sides, and for k=2,s=1 an extra
// zero row/col on the right/bottom (ZeroPad2d((0,1,0,1)) then pool with padding 0 ... models.py:79-87)
__global__ void maxpool_kernel(const __bf16 *__restrict__ x, int x_cs, __bf16 *__restrict__ y, int y_cs, int N, int H,
                               int W, int C, int k, int s, int pad_lo, int Ho, int Wo, int zero_pad_hi) {
    const int cpr = C / 8;
    const long long total = (long long)N * Ho * Wo * cpr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpr) * 8;
        const long long opix = i / cpr;
        const int ox = (int)(opix % Wo);
        const long long t = opix / Wo;
        const int oy = (int)(t % Ho);
        const long long n = t / Ho;
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; e++) m[e] = -3.4e38f;
        for (int dy = 0; dy < k; dy++)
            for (int dx = 0; dx < k; dx++) {
                const int iy = oy * s - pad_lo + dy, ix = ox * s - pad_lo + dx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                    const bf16x8 v = *(const bf16x8 *)(x + ((n * H + iy) * W + ix) * x_cs + c);
#pragma unroll
                    for (int e = 0; e < 8; e++) m[e] = fmaxf(m[e], (float)v[e]);
                } else if (zero_pad_hi && (iy >= H || ix >= W)) {
#pragma unroll
                    for (int e = 0; e < 8; e++) m[e] = fmaxf(m[e], 0.f);   // explicit zero padding takes part
                }
            }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = (__bf16)m[e];
        *(bf16x8 *)(y + opix * y_cs + c) = o;
    }
}

inline int grid_for(long long total, int tb = 256, int cap = 65536) {
    long long nb = (total + tb - 1) / tb;
    return (int)(nb < 1 ? 1 : (nb > cap ? cap : nb));
}
inline int ok_launch() { return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH; }

}  // namespace

extern "C" {

int ryolo_yolo_decode_filter(const void *head, int head_cstride, int bs, int ny, int nx, int na, int no,
                             const float *anchors, float stride, float context_factor, int arc, float conf_thres,
                             float min_wh, long long io_rows_per_image, long long io_row_offset, float *cand,
                             long long *cand_row, int *counter, int capacity, void *stream) {
    if (!head || !anchors || !cand || !cand_row || !counter || bs <= 0 || ny <= 0 || nx <= 0 || na <= 0 || no < 7 ||
        no > DEC_MAX_NO || head_cstride < na * no || capacity <= 0)
        return RYOLO_EINVAL;
    if (arc < 0 || arc > 2 || !(stride > 0.f) || !(context_factor > 0.f)) return RYOLO_EINVAL;
    const long long total = (long long)bs * ny * nx * na;
    hipLaunchKernelGGL(yolo_decode_filter_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16 *)head, head_cstride, bs, ny, nx, na, no, anchors, stride, context_factor, arc,
                       conf_thres, min_wh, io_rows_per_image, io_row_offset, cand, cand_row, counter, capacity);
    return ok_launch();
}

int ryolo_yolo_decode(const void *head, int head_cstride, int bs, int ny, int nx, int na, int no,
                      const float *anchors, float stride, float context_factor, int arc, float *io,
                      long long io_rows_per_image, long long io_row_offset, float *p, void *stream) {
    if (!head || !anchors || (!io && !p) || bs <= 0 || ny <= 0 || nx <= 0 || na <= 0 || no < 7 || head_cstride < na * no)
        return RYOLO_EINVAL;
    if (arc < 0 || arc > 2 || !(stride > 0.f) || !(context_factor > 0.f)) return RYOLO_EINVAL;
    const long long npix = (long long)bs * ny * nx;
    const int C = na * no;
    const size_t smem = (size_t)DEC_PIX * C * 2;
    if ((C & 7) == 0 && (head_cstride & 7) == 0 && (((uintptr_t)head) & 15) == 0 && smem <= 64 * 1024 && no <= 96 && npix < 0x7fffffffll) {
        const unsigned nb = (unsigned)((npix + DEC_PIX - 1) / DEC_PIX);
        if (no == 7)
            hipLaunchKernelGGL(yolo_decode_tiled_kernel<7>, dim3(nb), dim3(256), smem, (hipStream_t)stream,
                               (const __bf16 *)head, head_cstride, npix, ny, nx, na, no, anchors, stride, context_factor,
                               arc, io, io_rows_per_image, io_row_offset, p);
        else
            hipLaunchKernelGGL(yolo_decode_tiled_kernel<0>, dim3(nb), dim3(256), smem, (hipStream_t)stream,
                               (const __bf16 *)head, head_cstride, npix, ny, nx, na, no, anchors, stride, context_factor,
                               arc, io, io_rows_per_image, io_row_offset, p);
        return ok_launch();
    }
    if (!io) return RYOLO_EINVAL;          // the p-only form needs the tiled kernel's preconditions
    const long long total = npix * na;
    hipLaunchKernelGGL(yolo_decode_simple_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16 *)head, head_cstride, bs, ny, nx, na, no, anchors, stride, context_factor, arc,
                       io, io_rows_per_image, io_row_offset, p);
    return ok_launch();
}

int ryolo_add_nhwc(const void *a, int a_cstride, const void *b, int b_cstride, void *y, int y_cstride,
                   long long npix, int C, void *stream) {
    if (!a || !b || !y || npix <= 0 || C <= 0 || (C & 7) || (a_cstride & 7) || (b_cstride & 7) || (y_cstride & 7))
        return RYOLO_EINVAL;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(npix * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16 *)a, a_cstride, (const __bf16 *)b, b_cstride, (__bf16 *)y, y_cstride, npix, C);
    return ok_launch();
}

int ryolo_upsample_nhwc(const void *x, int x_cstride, void *y, int y_cstride, int N, int H, int W, int C, int scale,
                        void *stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || (x_cstride & 7) || (y_cstride & 7) || scale < 1)
        return RYOLO_EINVAL;
    const long long total = (long long)N * H * scale * W * scale * (C / 8);
    hipLaunchKernelGGL(upsample_copy_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16 *)x, x_cstride, (__bf16 *)y, y_cstride, N, H, W, C, scale);
    return ok_launch();
}

int ryolo_maxpool_nhwc(const void *x, int x_cstride, void *y, int y_cstride, int N, int H, int W, int C, int ksize,
                       int stride, void *stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || (x_cstride & 7) || (y_cstride & 7) ||
        ksize < 1 || stride < 1)
        return RYOLO_EINVAL;
    const int pad_lo = (ksize - 1) / 2;
    const int zero_hi = (ksize == 2 && stride == 1) ? 1 : 0;
    const int Hp = H + (zero_hi ? 1 : 0), Wp = W + (zero_hi ? 1 : 0);
    const int Ho = (Hp + 2 * pad_lo - ksize) / stride + 1, Wo = (Wp + 2 * pad_lo - ksize) / stride + 1;
    const long long total = (long long)N * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x,
                       x_cstride, (__bf16 *)y, y_cstride, N, H, W, C, ksize, stride, pad_lo, Ho, Wo, zero_hi);
    return ok_launch();
}

}  // extern "C"
