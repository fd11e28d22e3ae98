// rotate-yolov3_amd/csrc/conv_stem.hip -- 3x3 convolution, C_in = 32 -> C_out = 64, stride 1 or 2 (Darknet-53 layers 1 and 3, and
// their training forwards): the stem layers whose implicit-GEMM tiles are bound by the 9x tap re-fetch of 64-byte pixel rows through L2
// (6.8 GB of L2->LDS traffic per bs-64 launch for 0.76 GB of input), not by HBM and not by MFMA.
//
//   * a workgroup (4 waves) owns an 8 x 32 (stride 2: 4 x 32) block of output pixels x all 64 channels; the input patch it needs --
//     (rows-1)*S+3 x (32-1)*S+3 pixels of 32 channels, halo included -- goes HBM -> LDS ONCE (16-B buffer_load ... lds, out-of-image
//     pixels are out-of-range offsets = hardware zeros): 22 KiB for stride 1, 40 KiB for stride 2, double-buffered (the next tile's
//     patch is requested before this tile's MFMAs; one barrier per tile);
//   * C_in = 32 is exactly the K of one v_mfma_f32_16x16x32_bf16, so a filter tap is ONE MFMA per (16 pixels x 16 channels): the whole
//     3 x 3 x 32 x 64 filter lives in registers for the life of the (persistent) workgroup -- 36 A fragments = 144 VGPRs -- and the
//     B fragment of a tap is one ds_read_b128 at the tap's shifted pixel (chunk-XOR swizzle keyed on the patch column: 16 pixels
//     x one 16-B chunk cover all 64 banks);
//   * per 16-pixel group: 9 LDS reads, 36 MFMAs, then the epilogue on the accumulators (each lane: 4 consecutive channels of one
//     pixel per channel group; v_permlane16_swap regroups pairs of channel groups into 16-B runs -> two 16-B stores per lane);
//   * taps are accumulated in the order 0..8 with one K = 32 MFMA each -- the order of the implicit-GEMM kernels' K loop -- so the
//     results are bit-identical to those kernels (tests/test_conv_gpu.py).
// Two instantiations per stride: inference epilogue (scale / shift / activation / shortcut) and the training forward (z as stored +
// per-channel sums of z and z^2 into the fp64 partial rows, like every other statistics epilogue of the library).
#include <type_traits>

#include "conv_common.h"

namespace ryolo_detail {
namespace {

template <int N> using ic = std::integral_constant<int, N>;

constexpr int TW = 32;                    // output columns per workgroup tile
constexpr int CIN = 32, COUT = 64, CG = COUT / 16;

template <int S>
struct Patch {
    static constexpr int TH = S == 1 ? 8 : 4;              // output rows per tile: 8 x 32 (stride 1), 4 x 32 (stride 2)
    static constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, NPIX = PH * PW;
    static constexpr int NPIECE = (NPIX + 15) / 16;        // 1-KiB direct-to-LDS pieces (16 pixels x 64 B)
    static constexpr int PPW = (NPIECE + 3) / 4;           // pieces per wave
    static constexpr int BUF = PPW * 4 * 1024;             // one patch buffer: 22 KiB (stride 1), 40 KiB (stride 2)
    static constexpr int BYTES = 2 * BUF;                  // two of them: the next tile's patch is in flight under this tile's MFMAs
    static constexpr int GPW = TH * TW / 16 / 4;           // 16-pixel groups per wave and tile
};

__device__ __forceinline__ int chunk_swz(int pcol) { return (pcol ^ (pcol >> 2)) & 3; }

template <int S, bool STATS>
__global__ void __launch_bounds__(256, 2) conv3x3_c32_halo_kernel(const ConvParams p, int tiles_x, int tiles_y, int ntiles) {
    using P = Patch<S>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;

    // XCD-contiguous chunks of the tile list (neighbouring tiles share halo rows in one L2)
    const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, nloc = gridDim.x >> 3;
    const int start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int len = q8 + (xcd < r8 ? 1 : 0);

    // the filter: fragment (tap t, channel group cg) = rows cg*16 + fr, K columns t*32 + g*8 .. +7 of the packed image
    bf16x8 wfr[9][CG];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int cg = 0; cg < CG; cg++) wfr[t][cg] = *(const bf16x8 *)(p.w + (size_t)(cg * 16 + fr) * p.Kpad + t * CIN + g * 8);
    f32x4 sc[CG], sh[CG];
    if constexpr (!STATS) {
#pragma unroll
        for (int cg = 0; cg < CG; cg++) {
            sc[cg] = *(const f32x4 *)(p.scale + cg * 16 + g * 4);
            sh[cg] = *(const f32x4 *)(p.shift + cg * 16 + g * 4);
        }
    }
    float st_sum[CG][4], st_sq[CG][4];
#pragma unroll
    for (int cg = 0; cg < CG; cg++)
#pragma unroll
        for (int r = 0; r < 4; r++) st_sum[cg][r] = st_sq[cg][r] = 0.f;
    const float slope = p.slope;
    const int tiles_img = tiles_x * tiles_y;

    auto fill = [&](int id, char *buf) {               // the input patch of tile `id`: piece k covers patch-linear pixels 16k .. 16k+15,
        const int img = id / tiles_img, rem = id - img * tiles_img;         // 4 lanes (16-B chunks) per pixel
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int h0 = ty * P::TH * S - p.pad, w0 = tx * TW * S - p.pad;
#pragma unroll 2                                           // (fully unrolled, the pieces of the stride-2 patch spill 20 VGPRs)
        for (int j = 0; j < P::PPW; j++) {
            const int piece = wave * P::PPW + j;
            const int q = piece * 16 + (lane >> 2);
            const int prow = q / P::PW, pcol = q - prow * P::PW;
            const int hi = h0 + prow, wi = w0 + pcol;
            const bool ok = q < P::NPIX && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const int chunk = (lane & 3) ^ chunk_swz(pcol);              // the logical chunk stored at physical slot lane & 3
            const int off = (((img * p.H + hi) * p.W + wi) * p.in_cs + chunk * 8) * 2;
            buffer_load_lds16(p.x, p.x_bytes, buf + piece * 1024, ok ? off : (int)0x80000000, 0);
        }
    };
    int cur = 0;
    if (loc < len) fill(start + loc, smem);
    for (int i = loc; i < len; i += nloc) {
        const int id = start + i;
        const int img = id / tiles_img, rem = id - img * tiles_img;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int ho0 = ty * P::TH, wo0 = tx * TW;
        // this tile's patch has landed (requested one tile ago; the wait also drains the previous tile's stores) and every wave is
        // done reading the other buffer, which now receives the NEXT tile's patch under this tile's MFMAs
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (i + nloc < len) fill(id + nloc, smem + (cur ^ 1) * P::BUF);
        const char *patch = smem + cur * P::BUF;
        cur ^= 1;
        // ---- four groups of 16 output pixels per wave: row r = group / 2 of the tile, columns c0 .. c0 + 15
#pragma unroll 2
        for (int jg = 0; jg < P::GPW; jg++) {
            const int grp = wave * P::GPW + jg;
            const int r = grp >> 1, c = (grp & 1) * 16 + fr;
            const int ho = ho0 + r, wo = wo0 + c;
            const bool ok = ho < p.Ho && wo < p.Wo;
            const size_t m = ((size_t)img * p.Ho + ho) * p.Wo + wo;
            // store layout (after the lane regrouping below): for the channel-group pair (2h, 2h+1) a lane owns ONE 16-B run --
            // even g: channels 2h*16 + 8(g/2) .. +7, odd g: (2h+1)*16 + 8(g/2) .. +7
            const int run0 = ((g & 1) ? 16 : 0) + (g >> 1) * 8;
            bf16x8 rv[CG / 2];                              // shortcut rows in that layout: requested before the MFMAs
            if constexpr (!STATS) {
                if (p.res) {
#pragma unroll
                    for (int h = 0; h < CG / 2; h++)
                        rv[h] = ok ? *(const bf16x8 *)(p.res + m * p.res_cs + h * 32 + run0) : bf16x8{};
                }
            }
            f32x4 acc[CG];
#pragma unroll
            for (int cg = 0; cg < CG; cg++) acc[cg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int kh = t / 3, kw = t - 3 * kh;
                const int pcol = c * S + kw;
                const int q = (r * S + kh) * P::PW + pcol;
                const bf16x8 xf = *(const bf16x8 *)(patch + q * 64 + ((g ^ chunk_swz(pcol)) << 4));
#pragma unroll
                for (int cg = 0; cg < CG; cg++) acc[cg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[t][cg], xf, acc[cg], 0, 0, 0);
            }
            unsigned o2[CG][2];                             // this lane's 4 channels of each channel group, bf16 pairs
#pragma unroll
            for (int cg = 0; cg < CG; cg++) {
                bf16x4 o;
                if constexpr (STATS) {
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        o[rr] = (__bf16)acc[cg][rr];
                        const float qv = ok ? (float)o[rr] : 0.f;        // statistics of the values as stored
                        st_sum[cg][rr] += qv;
                        st_sq[cg][rr] += qv * qv;
                    }
                } else {
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        float v = acc[cg][rr] * sc[cg][rr] + sh[cg][rr];
                        if (p.act == RYOLO_ACT_LEAKY) v = v > 0.f ? v : v * slope;
                        else if (p.act == RYOLO_ACT_MISH) v = mish(v);
                        o[rr] = (__bf16)v;
                    }
                }
                const uint2 u = __builtin_bit_cast(uint2, o);
                o2[cg][0] = u.x;
                o2[cg][1] = u.y;
            }
            // the odd 16-lane rows of group 2h trade places with the even rows of group 2h+1: every lane then holds 8 consecutive
            // channels = one 16-B store, and the four lanes of a pixel write 64 contiguous bytes per channel-group pair
#pragma unroll
            for (int h = 0; h < CG / 2; h++) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    auto sw = __builtin_amdgcn_permlane16_swap(o2[2 * h][d], o2[2 * h + 1][d], false, false);
                    o2[2 * h][d] = sw[0];
                    o2[2 * h + 1][d] = sw[1];
                }
#endif
                u32x4 outv = u32x4{o2[2 * h][0], o2[2 * h][1], o2[2 * h + 1][0], o2[2 * h + 1][1]};
                if constexpr (!STATS) {
                    if (p.res) {
                        bf16x8 ov = __builtin_bit_cast(bf16x8, outv);
#pragma unroll
                        for (int e = 0; e < 8; e++) ov[e] = (__bf16)((float)ov[e] + (float)rv[h][e]);
                        outv = __builtin_bit_cast(u32x4, ov);
                    }
                }
                if (ok) {
                    u32x4 *dst = (u32x4 *)(p.y + m * p.out_cs + h * 32 + run0);
                    if (p.nt_out) __builtin_nontemporal_store(outv, dst);
                    else *dst = outv;
                }
            }
        }
    }
    if constexpr (STATS) {
        // the 16 lanes of a row hold the same 16 channels: DPP row sums, lane fr keeps total (cg, rr) = (fr / 4, fr % 4); the four
        // waves are combined in LDS in a fixed order and one thread per (statistic, channel) adds to the fp64 partial row
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int cg = 0; cg < CG; cg++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const float a = row16_sum(st_sum[cg][rr]), b = row16_sum(st_sq[cg][rr]);
                if (fr == cg * 4 + rr) {
                    ta = a;
                    tb = b;
                }
            }
        __syncthreads();                                   // the patch is dead
        float *slots = (float *)smem;                      // [4 waves][2][64]
        const int ch = (fr >> 2) * 16 + g * 4 + (fr & 3);
        slots[(wave * 2 + 0) * COUT + ch] = ta;
        slots[(wave * 2 + 1) * COUT + ch] = tb;
        __syncthreads();
        if (tid < 2 * COUT) {
            const int st = tid / COUT, c = tid % COUT;
            float v = slots[st * COUT + c];
#pragma unroll
            for (int w = 1; w < 4; w++) v += slots[(w * 2 + st) * COUT + c];
            atomicAdd(p.stat_part + ((size_t)(blockIdx.x % STAT_ROWS) * 2 + st) * p.stat_cpad + c, (double)v);
        }
    }
}

// ------------------------------------------------------------------------------------------------ layer 0 with its input patch in LDS
// Darknet-53 layer 0 (3x3 / 1, 3 -> 32 channels on the 8-channel padded input), forward passes (inference; y = act(BN(conv(x))) of training).  conv3x3_c8_direct_kernel
// (conv.hip) loads its B fragments -- 16 B = one tap of one pixel -- straight from global memory: every input pixel travels through the
// texture path nine times, and the forward is bound by that and by its ~80 address / masking instructions per 16 pixels (0.53 ms at bs 64
// for 1.9 GB; the statistics-only pass, 0.28 ms, is bound by its per-element arithmetic and stays on that kernel: 0.32 ms here).  Here a persistent workgroup stages the 10 x 66 pixel patch of an 8 x 64
// output tile once (16-B direct-to-LDS loads, out-of-image pixels = hardware zeros, double-buffered) and reads the fragments with
// ds_read_b128 at lane-constant offsets: 0.531 -> 0.360 ms (training forward, bs 64), 0.246 -> ~0.21 ms (inference, bs 32).  Same MFMAs in
// the same order as conv3x3_c8_direct_kernel: bit-identical outputs (tests/test_conv_gpu.py, tests/test_train_ops_gpu.py::test_layer0_*).
constexpr int C0_TH = 8, C0_TW = 64, C0_PH = C0_TH + 2, C0_PW = C0_TW + 2, C0_NPIX = C0_PH * C0_PW;
constexpr int C0_NPIECE = (C0_NPIX + 63) / 64;            // 1-KiB pieces: 64 pixels x 16 B
constexpr int C0_PPW = (C0_NPIECE + 3) / 4, C0_BUF = C0_PPW * 4 * 1024;

template <int ACT>
__global__ void __launch_bounds__(256) conv0_halo_kernel(const ConvParams p, const float *slope_dev, int round_z, int tiles_x, int tiles_y, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, nloc = gridDim.x >> 3;
    const int start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int len = q8 + (xcd < r8 ? 1 : 0);
    bf16x8 wfr[2][3];
#pragma unroll
    for (int cf = 0; cf < 2; cf++)
#pragma unroll
        for (int ks = 0; ks < 3; ks++) wfr[cf][ks] = *(const bf16x8 *)(p.w + (size_t)(cf * 16 + fr) * p.Kpad + ks * 32 + g * 8);
    f32x4 sc[2], sh[2];
#pragma unroll
    for (int cf = 0; cf < 2; cf++) {
        sc[cf] = *(const f32x4 *)(p.scale + cf * 16 + g * 4);
        sh[cf] = *(const f32x4 *)(p.shift + cf * 16 + g * 4);
    }
    const float slope = slope_dev ? slope_dev[0] : p.slope;
    // lane-constant fragment offsets of the three K steps: tap 4 ks + g of the pixel (patch origin = one pixel up / left); taps >= 9 are K padding
    int rel[3];
    bool tok[3];
#pragma unroll
    for (int ks = 0; ks < 3; ks++) {
        const int tap = ks * 4 + g, kh = (tap * 11) >> 5, kw = tap - 3 * kh;
        tok[ks] = tap < 9;
        rel[ks] = tok[ks] ? ((kh * C0_PW + kw) + fr) * 16 : 0;
    }
    const int tiles_img = tiles_x * tiles_y;
    auto fill = [&](int id, char *buf) {
        const int img = id / tiles_img, rem = id - img * tiles_img;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int h0 = ty * C0_TH - 1, w0 = tx * C0_TW - 1;
#pragma unroll
        for (int j = 0; j < C0_PPW; j++) {
            const int piece = wave * C0_PPW + j;
            const int q = piece * 64 + lane;
            const int prow = q / C0_PW, pcol = q - prow * C0_PW;
            const int hi = h0 + prow, wi = w0 + pcol;
            const bool ok = q < C0_NPIX && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const int off = ((img * p.H + hi) * p.W + wi) * 16;
            buffer_load_lds16(p.x, p.x_bytes, buf + piece * 1024, ok ? off : (int)0x80000000, 0);
        }
    };
    int cur = 0;
    if (loc < len) fill(start + loc, smem);
    for (int i = loc; i < len; i += nloc) {
        const int id = start + i;
        const int img = id / tiles_img, rem = id - img * tiles_img;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int ho0 = ty * C0_TH, wo0 = tx * C0_TW;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (i + nloc < len) fill(id + nloc, smem + (cur ^ 1) * C0_BUF);
        const char *patch = smem + cur * C0_BUF;
        cur ^= 1;
#pragma unroll 2
        for (int jg = 0; jg < C0_TH * C0_TW / 16 / 4; jg++) {
            const int grp = wave * (C0_TH * C0_TW / 16 / 4) + jg;
            const int r = grp >> 2, c0 = (grp & 3) * 16;
            const int ho = ho0 + r, wo = wo0 + c0 + fr;
            const bool mok = ho < p.Ho && wo < p.Wo;
            const size_t m = ((size_t)img * p.Ho + ho) * p.Wo + wo;
            const char *gp = patch + (r * C0_PW + c0) * 16;
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 3; ks++) {
                u4 xv = *(const u4 *)(gp + rel[ks]);
                if (!tok[ks]) xv = u4{0u, 0u, 0u, 0u};
#pragma unroll
                for (int cf = 0; cf < 2; cf++)
                    acc[cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[cf][ks], __builtin_bit_cast(bf16x8, xv), acc[cf], 0, 0, 0);
            }
            unsigned o2[2][2];
#pragma unroll
            for (int cf = 0; cf < 2; cf++) {
                bf16x4 o;
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const float a_ = round_z ? (float)(__bf16)acc[cf][rr] : acc[cf][rr];
                    float v = a_ * sc[cf][rr] + sh[cf][rr];
                    if constexpr (ACT == RYOLO_ACT_LEAKY) v = v > 0.f ? v : v * slope;
                    else if constexpr (ACT == RYOLO_ACT_MISH) v = mish(v);
                    o[rr] = (__bf16)v;
                }
                const uint2 u = __builtin_bit_cast(uint2, o);
                o2[cf][0] = u.x;
                o2[cf][1] = u.y;
            }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int d = 0; d < 2; d++) {
                auto sw = __builtin_amdgcn_permlane16_swap(o2[0][d], o2[1][d], false, false);
                o2[0][d] = sw[0];
                o2[1][d] = sw[1];
            }
#endif
            if (mok) {
                const u4 outv = u4{o2[0][0], o2[0][1], o2[1][0], o2[1][1]};
                u4 *dst = (u4 *)(p.y + m * p.out_cs + ((g & 1) ? 16 : 0) + (g >> 1) * 8);
                if (p.nt_out) __builtin_nontemporal_store(outv, dst);
                else *dst = outv;
            }
        }
    }
}

static int launch_conv0_halo_t(ConvParams &p, const float *slope_dev, int round_z, int act, int grid, int tiles_x, int tiles_y, int ntiles,
                               hipStream_t stream) {
    constexpr int smem = 2 * C0_BUF;
    if (act == RYOLO_ACT_LEAKY)
        hipLaunchKernelGGL(conv0_halo_kernel<RYOLO_ACT_LEAKY>, dim3((unsigned)grid), dim3(256), smem, stream, p, slope_dev, round_z, tiles_x, tiles_y, ntiles);
    else if (act == RYOLO_ACT_MISH)
        hipLaunchKernelGGL(conv0_halo_kernel<RYOLO_ACT_MISH>, dim3((unsigned)grid), dim3(256), smem, stream, p, slope_dev, round_z, tiles_x, tiles_y, ntiles);
    else
        hipLaunchKernelGGL(conv0_halo_kernel<RYOLO_ACT_LINEAR>, dim3((unsigned)grid), dim3(256), smem, stream, p, slope_dev, round_z, tiles_x, tiles_y, ntiles);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

// ------------------------------------------------------------------------------------------------ data gradients of layers 1 and 3
// dx[n][hi][wi][ci] = sum_{kh,kw,co} dz[n][ho][wo][co] W[co][ci][kh][kw],  hi = 2 ho - 1 + kh,  wi = 2 wo - 1 + kw   (3x3 / 2, pad 1,
// 64 -> 32 channels in the gradient's direction; autograd of model/models.py:55-60 for Darknet-53 layer 1).  On the implicit-GEMM tiles
// this is four (two x-fused) launches of 2-8 K steps each -- all prologue and epilogue: 0.94 ms at bs 64, 232 TFLOP/s, 2.3 x its HBM
// floor (1.5 GB of dx written, 0.76 GB of dz read).  Here, as in conv3x3_c32_halo_kernel:
//   * a workgroup owns an 8 x 64 block of dx; the dz patch ALL FOUR output-parity classes of that block need (5 x 33 pixels of 64
//     channels) goes HBM -> LDS once (double-buffered, 16-B direct-to-LDS loads, slot ^= (column >> 1) & 7);
//   * the filter -- 9 taps x 64 x 32, as the four class images ryolo_conv_pack_weights_dgrad already produces ([ci][tap * 64 + co]) --
//     lives in 144 VGPRs per wave for the life of the persistent workgroup;
//   * wave w takes output rows 2w (even: one filter row) and 2w + 1 (odd: two), both column parities, 16 pixels per group: a tap is
//     2 K steps x 2 channel fragments = 4 MFMAs, 72 MFMAs and 36 ds_read_b128 per wave and tile;
//   * epilogue like the first layer's (32 channels = 64 B per pixel, one 16-B run per lane); the two column parities of a row are
//     stored back to back, so that L2 sees whole 128-B lines.
// Accumulation order (taps in the class order of dgrad_classes, channels ascending) differs from the parity-class launches: results
// agree to fp32 summation order (tests/test_train_ops_gpu.py::test_stem_stride2_dgrad_*).
struct DgS2Params {
    const __bf16 *dz; unsigned dz_bytes; int dz_cs;
    const __bf16 *w;                 // the four classic class images, consecutive (conv.hip: dgrad_classic_bytes)
    __bf16 *dx; int dx_cs;
    const __bf16 *res;               // dx itself when the gradient accumulates, else nullptr
    int H, W, Ho, Wo;                // dx (= the forward input) and dz (= the forward output) extents
    int tiles_x, tiles_y, ntiles, nt_out;
};

// S = 2: the four output-parity classes of the stride-2 layer (above).  S = 1: the data gradient of the stride-1 3x3 32 -> 64 layer
// (Darknet-53 layer 3; a plain 3x3 convolution of dz with the flipped, channel-transposed filter -- ryolo_conv_pack_weights_dgrad's
// single stride-1 image, nine taps): 4 x 32 output pixels per tile, wave w takes row w.  0.44 ms on the persistent 256 x 32 tile.
template <int S>
struct DgTile {
    static constexpr int TH = S == 2 ? 8 : 4, TW = S == 2 ? 64 : 32;
    static constexpr int PH = S == 2 ? TH / 2 + 1 : TH + 2, PW = S == 2 ? TW / 2 + 1 : TW + 2, NPIX = PH * PW;
    static constexpr int NPIECE = (NPIX + 7) / 8;                // 1-KiB pieces: 8 pixels x 128 B
    static constexpr int PPW = (NPIECE + 3) / 4, BUF = PPW * 4 * 1024;
};

template <int S>
__global__ void __launch_bounds__(256, 2) dgrad3x3_c64_kernel(const DgS2Params p) {
    using T = DgTile<S>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int q8 = p.ntiles >> 3, r8 = p.ntiles & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, nloc = gridDim.x >> 3;
    const int start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int len = q8 + (xcd < r8 ? 1 : 0);

    // filter fragments, rows = ci, K = (tap, co).  S = 2: class (a, b) = image 2a + b with NTA(a) * NTB(b) taps (kh, kw descending:
    // dgrad_classes); S = 1: one image, taps 0 .. 8.  36 fragments = 144 VGPRs either way.
    bf16x8 wf[9][2][2];
    {
        constexpr int NCLS = S == 2 ? 4 : 1;
        const __bf16 *img = p.w;
        int t0 = 0;
#pragma unroll
        for (int cls = 0; cls < NCLS; cls++) {
            const int nt = S == 2 ? (cls == 0 ? 1 : (cls == 3 ? 4 : 2)) : 9, kpad = nt * 64;
#pragma unroll
            for (int t = 0; t < 4 + 5 * (S == 1); t++) {
                if (t >= nt) break;
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
#pragma unroll
                    for (int cg = 0; cg < 2; cg++)
                        wf[t0 + t][ks][cg] = *(const bf16x8 *)(img + (size_t)(cg * 16 + fr) * kpad + t * 64 + ks * 32 + g * 8);
            }
            t0 += nt;
            img += 128 * kpad + 128;
        }
    }
    const int tiles_img = p.tiles_x * p.tiles_y;
    auto fill = [&](int id, char *buf) {               // the dz patch of tile `id`: piece k covers patch-linear pixels 8k .. 8k+7, 8 lanes per pixel
        const int img = id / tiles_img, rem = id - img * tiles_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int h0 = S == 2 ? ty * (T::TH / 2) : ty * T::TH - 1, w0 = S == 2 ? tx * (T::TW / 2) : tx * T::TW - 1;
#pragma unroll 2
        for (int j = 0; j < T::PPW; j++) {
            const int piece = wave * T::PPW + j;
            const int q = piece * 8 + (lane >> 3);
            const int prow = q / T::PW, pcol = q - prow * T::PW;
            const int ho = h0 + prow, wo = w0 + pcol;
            const bool ok = q < T::NPIX && (unsigned)ho < (unsigned)p.Ho && (unsigned)wo < (unsigned)p.Wo;
            const int chunk = (lane & 7) ^ ((pcol >> 1) & 7);            // the logical 16-B chunk stored at physical slot lane & 7
            const int off = (((img * p.Ho + ho) * p.Wo + wo) * p.dz_cs + chunk * 8) * 2;
            buffer_load_lds16(p.dz, p.dz_bytes, buf + piece * 1024, ok ? off : (int)0x80000000, 0);
        }
    };
    const int run0 = ((g & 1) ? 16 : 0) + (g >> 1) * 8;   // the 16-B run of the pixel's 32 channels this lane stores (after the regrouping)
    int cur = 0;
    if (loc < len) fill(start + loc, smem);
    for (int i = loc; i < len; i += nloc) {
        const int id = start + i;
        const int img = id / tiles_img, rem = id - img * tiles_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int hi0 = ty * T::TH, wi0 = tx * T::TW;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's patch has landed (and the previous tile's stores have drained)
        __syncthreads();                                   // every wave is done reading the other buffer: it takes the NEXT tile's patch
        if (i + nloc < len) fill(id + nloc, smem + (cur ^ 1) * T::BUF);
        const char *patch = smem + cur * T::BUF;
        cur ^= 1;
        // one group of 16 output pixels of row r: S = 2: parity class (A, B), pixels wi0 + 2 (16 j + fr) + B, taps T0 .. T0 + NTA * NTB - 1;
        // S = 1 (A = B = 0 here): pixels wi0 + 16 j + fr, all nine taps
        auto group = [&](auto Ac, auto Bc, auto T0c, int r, int j) {
            constexpr int A = decltype(Ac)::value, B = decltype(Bc)::value, T0 = decltype(T0c)::value;
            constexpr int NTA = S == 2 ? (A ? 2 : 1) : 3, NTB = S == 2 ? (B ? 2 : 1) : 3;
            const int hi = hi0 + r, pxl = 16 * j + fr, wi = S == 2 ? wi0 + 2 * pxl + B : wi0 + pxl;
            const bool ok = hi < p.H && wi < p.W;
            const size_t m = ((size_t)img * p.H + hi) * p.W + wi;
            bf16x8 rv = bf16x8{};
            if (p.res && ok) rv = *(const bf16x8 *)(p.res + m * p.dx_cs + run0);
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ta = 0; ta < NTA; ta++)
#pragma unroll
                for (int tb = 0; tb < NTB; tb++) {
                    // S = 2: dz pixel (i + dy, j + dx) of the class grid; S = 1: (hi - 1 + kh, wi - 1 + kw), the patch starts one pixel up / left
                    const int prow = S == 2 ? (r - A) / 2 + (A ? ta : 0) : r + ta, pcol = S == 2 ? pxl + (B ? tb : 0) : pxl + tb;
                    const char *px = patch + (prow * T::PW + pcol) * 128;
                    const int sw = (pcol >> 1) & 7;
#pragma unroll
                    for (int ks = 0; ks < 2; ks++) {
                        const bf16x8 xf = *(const bf16x8 *)(px + (((g + 4 * ks) ^ sw) << 4));
#pragma unroll
                        for (int cg = 0; cg < 2; cg++)
                            acc[cg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[T0 + ta * NTB + tb][ks][cg], xf, acc[cg], 0, 0, 0);
                    }
                }
            unsigned o2[2][2];
#pragma unroll
            for (int cg = 0; cg < 2; cg++) {
                bf16x4 o;
#pragma unroll
                for (int rr = 0; rr < 4; rr++) o[rr] = (__bf16)acc[cg][rr];
                const uint2 u = __builtin_bit_cast(uint2, o);
                o2[cg][0] = u.x;
                o2[cg][1] = u.y;
            }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int d = 0; d < 2; d++) {                  // odd 16-lane rows of fragment 0 <-> even rows of fragment 1: one 16-B run per lane
                auto swp = __builtin_amdgcn_permlane16_swap(o2[0][d], o2[1][d], false, false);
                o2[0][d] = swp[0];
                o2[1][d] = swp[1];
            }
#endif
            u32x4 outv = u32x4{o2[0][0], o2[0][1], o2[1][0], o2[1][1]};
            if (p.res) {
                bf16x8 ov = __builtin_bit_cast(bf16x8, outv);
#pragma unroll
                for (int e = 0; e < 8; e++) ov[e] = (__bf16)((float)ov[e] + (float)rv[e]);
                outv = __builtin_bit_cast(u32x4, ov);
            }
            if (ok) {
                u32x4 *dst = (u32x4 *)(p.dx + m * p.dx_cs + run0);
                if (p.nt_out) __builtin_nontemporal_store(outv, dst);
                else *dst = outv;
            }
        };
        if constexpr (S == 2) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                group(ic<0>{}, ic<0>{}, ic<0>{}, 2 * wave, j);
                group(ic<0>{}, ic<1>{}, ic<1>{}, 2 * wave, j);
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                group(ic<1>{}, ic<0>{}, ic<3>{}, 2 * wave + 1, j);
                group(ic<1>{}, ic<1>{}, ic<5>{}, 2 * wave + 1, j);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; j++) group(ic<0>{}, ic<0>{}, ic<0>{}, wave, j);
        }
    }
}

// ------------------------------------------------------------------------------------------------ two layers, one launch (inference)
// Darknet-53 layers 2 -> 3 (+ shortcut 4): 1x1 64->32 @304^2, then 3x3 32->64 + the shortcut (model/models.py:281-282) from layer 2's own
// INPUT.  The 32-channel tensor between them (378 MB at bs 32) is written by one launch only to be read by the next.
// conv3x3_c32_halo_kernel already reads its 32-channel input patch from LDS; here the patch is not loaded but COMPUTED there:
//   stage 1  the 64-channel input patch of the tile, halo included, goes HBM -> LDS once (16-B direct-to-LDS loads, out-of-image pixels
//            = hardware zeros; [pixel][8 x 16-B slots], slot ^= (pixel >> 1) & 7 like the GEMM tiles);
//   stage 2  the 1x1 layer on the whole patch, 16 patch pixels per MFMA group, two K steps of 32 channels; scale / shift / activation;
//            pixels outside the image become the 3x3 layer's zero padding; rounded to bf16 exactly as the stored tensor would be and
//            written in the patch layout of the 3x3 stage;
//   stage 3  the 3x3 stage of conv3x3_c32_halo_kernel unchanged; the shortcut rows come from the stage-1 image in LDS.
// Same MFMAs in the same order as the two separate launches, same roundings: the output is bit-identical to running the two layers
// one after the other (tests/test_conv_gpu.py::test_stem_pair_*).  HBM traffic per bs-32 forward 1.52 -> 0.76 GB; measured in the forward
// 0.392 -> 0.305 ms (profiles/r04_bench_per_op_events.txt row L3; A/B of both fusions in profiles/r04_ab_log.txt).  (A 4 x 32 tile with the
// input patch double-buffered -- 2 x 26 + 13 KB, the next tile's patch streaming in under the two MFMA stages -- measured 0.350 ms: the
// halo recompute and three barriers per 128 pixels cost more than the exposed latency of the 8-row tile.)
// The same construction for layers 0 -> 1 (3x3 3(8)->32 computed into the patch of the 3x3/2 layer; 2.28 -> 0.57 GB of traffic) was built
// and measured SLOWER than the two launches, 0.501 vs 0.465 ms: layer 0 is bound by the ~80 VALU instructions per 16-pixel group of its
// epilogue and tap addressing, not by HBM, and inside one workgroup its stage cannot overlap the 3x3 stage's MFMAs.  Removed.
struct PairFirst {
    const __bf16 *x;        // input of the first layer, NHWC, pixel stride in_cs
    const __bf16 *w;        // its packed filter ([rows][Kpad], K index (kh*3 + kw)*8 + c for KIND 0, c for KIND 1)
    const float *scale, *shift;
    unsigned x_bytes;
    int in_cs, Kpad, act;
    float slope;
    int H, W;               // spatial size of the first layer (= of its output: stride 1)
};

struct PairGeom {
    static constexpr int S = 1;
    using P = Patch<S>;
    static constexpr int IW = P::PW, INPIX = P::NPIX;               // the 1x1 layer reads exactly the pixels of the 32-channel patch
    static constexpr int IN_PIECES = (INPIX * 128 + 1023) / 1024;   // 128 B per input pixel (64 channels)
    static constexpr int IN_BYTES = IN_PIECES * 1024;
    static constexpr int MID_GROUPS = (P::NPIX + 15) / 16;
    static constexpr int MID_BYTES = MID_GROUPS * 1024;
    static constexpr int BYTES = IN_BYTES + MID_BYTES;              // 65 KiB: two workgroups per CU
    static constexpr int C1 = 2;                                    // K steps of 32 of the 1x1 layer
};

__global__ void __launch_bounds__(256, 2) conv_stem_pair_kernel(const ConvParams p, const PairFirst f, int tiles_x, int tiles_y, int ntiles) {
    using G = PairGeom;
    using P = typename G::P;
    constexpr int S = G::S;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *const in_img = smem, *const mid = smem + G::IN_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;

    const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, nloc = gridDim.x >> 3;
    const int start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int len = q8 + (xcd < r8 ? 1 : 0);

    // the 3x3 filter of the second layer: in registers for the life of the workgroup (as in conv3x3_c32_halo_kernel)
    bf16x8 wfr[9][CG];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int cg = 0; cg < CG; cg++) wfr[t][cg] = *(const bf16x8 *)(p.w + (size_t)(cg * 16 + fr) * p.Kpad + t * CIN + g * 8);
    const float slope = p.slope, slope1 = f.slope;
    const int tiles_img = tiles_x * tiles_y;

    for (int i = loc; i < len; i += nloc) {
        const int id = start + i;
        const int img = id / tiles_img, rem = id - img * tiles_img;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int ho0 = ty * P::TH, wo0 = tx * TW;
        const int mh0 = ho0 * S - 1, mw0 = wo0 * S - 1;            // patch origin in the 1x1 layer's output (= its input: 1x1, stride 1)
        __syncthreads();                                           // every wave is done with the previous tile's LDS images
        // ---- stage 1
        for (int pc = wave; pc < G::IN_PIECES; pc += 4) {
            // 128 B per pixel: 8 pixels per piece, the 16-B slot (lane & 7) holds source chunk slot ^ key
            const int q = pc * 8 + (lane >> 3);
            const int off_in_px = ((lane & 7) ^ ((q >> 1) & 7)) * 16;
            const int prow = q / G::IW, pcol = q - prow * G::IW;
            const int hi = mh0 + prow, wi = mw0 + pcol;
            const bool ok = q < G::INPIX && (unsigned)hi < (unsigned)f.H && (unsigned)wi < (unsigned)f.W;
            const int off = ((img * f.H + hi) * f.W + wi) * f.in_cs * 2 + off_in_px;
            buffer_load_lds16(f.x, f.x_bytes, in_img + pc * 1024, ok ? off : (int)0x80000000, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- stage 2: the first layer on the 32-channel patch, 16 patch-linear pixels per group
        {
            bf16x8 w1[G::C1][2];
            f32x4 sc1[2], sh1[2];
#pragma unroll
            for (int cf = 0; cf < 2; cf++) {
#pragma unroll
                for (int ks = 0; ks < G::C1; ks++) w1[ks][cf] = *(const bf16x8 *)(f.w + (size_t)(cf * 16 + fr) * f.Kpad + ks * 32 + g * 8);
                sc1[cf] = *(const f32x4 *)(f.scale + cf * 16 + g * 4);
                sh1[cf] = *(const f32x4 *)(f.shift + cf * 16 + g * 4);
            }
            for (int grp = wave; grp < G::MID_GROUPS; grp += 4) {
                const int q = grp * 16 + fr;
                const int prow = q / P::PW, pcol = q - prow * P::PW;
                const int mh = mh0 + prow, mw = mw0 + pcol;
                const bool inside = q < P::NPIX && (unsigned)mh < (unsigned)f.H && (unsigned)mw < (unsigned)f.W;
                f32x4 a1[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int ks = 0; ks < G::C1; ks++) {
                    const bf16x8 xf = *(const bf16x8 *)(in_img + q * 128 + (((ks * 4 + g) ^ ((q >> 1) & 7)) << 4));
#pragma unroll
                    for (int cf = 0; cf < 2; cf++) a1[cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[ks][cf], xf, a1[cf], 0, 0, 0);
                }
#pragma unroll
                for (int cf = 0; cf < 2; cf++) {
                    bf16x4 o;
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        float v = a1[cf][rr] * sc1[cf][rr] + sh1[cf][rr];
                        if (f.act == RYOLO_ACT_LEAKY) v = v > 0.f ? v : v * slope1;
                        else if (f.act == RYOLO_ACT_MISH) v = mish(v);
                        o[rr] = (__bf16)(inside ? v : 0.f);        // outside the image: the second layer's zero padding
                    }
                    // channels cf*16 + g*4 .. +3 = half (g & 1) of 16-B chunk cf*2 + (g >> 1), in the 3x3 stage's patch layout
                    *(bf16x4 *)(mid + q * 64 + (((cf * 2 + (g >> 1)) ^ chunk_swz(pcol)) << 4) + (g & 1) * 8) = o;
                }
            }
        }
        __syncthreads();
        // ---- stage 3: the 3x3 stage of conv3x3_c32_halo_kernel on the computed patch
        f32x4 sc[CG], sh[CG];
#pragma unroll
        for (int cg = 0; cg < CG; cg++) {
            sc[cg] = *(const f32x4 *)(p.scale + cg * 16 + g * 4);
            sh[cg] = *(const f32x4 *)(p.shift + cg * 16 + g * 4);
        }
#pragma unroll 2
        for (int jg = 0; jg < P::GPW; jg++) {
            const int grp = wave * P::GPW + jg;
            const int r = grp >> 1, c = (grp & 1) * 16 + fr;
            const int ho = ho0 + r, wo = wo0 + c;
            const bool ok = ho < p.Ho && wo < p.Wo;
            const size_t m = ((size_t)img * p.Ho + ho) * p.Wo + wo;
            const int run0 = ((g & 1) ? 16 : 0) + (g >> 1) * 8;
            f32x4 acc[CG];
#pragma unroll
            for (int cg = 0; cg < CG; cg++) acc[cg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int kh = t / 3, kw = t - 3 * kh;
                const int pcol = c * S + kw;
                const int q = (r * S + kh) * P::PW + pcol;
                const bf16x8 xf = *(const bf16x8 *)(mid + q * 64 + ((g ^ chunk_swz(pcol)) << 4));
#pragma unroll
                for (int cg = 0; cg < CG; cg++) acc[cg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[t][cg], xf, acc[cg], 0, 0, 0);
            }
            unsigned o2[CG][2];
#pragma unroll
            for (int cg = 0; cg < CG; cg++) {
                bf16x4 o;
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    float v = acc[cg][rr] * sc[cg][rr] + sh[cg][rr];
                    if (p.act == RYOLO_ACT_LEAKY) v = v > 0.f ? v : v * slope;
                    else if (p.act == RYOLO_ACT_MISH) v = mish(v);
                    o[rr] = (__bf16)v;
                }
                const uint2 u = __builtin_bit_cast(uint2, o);
                o2[cg][0] = u.x;
                o2[cg][1] = u.y;
            }
#pragma unroll
            for (int h = 0; h < CG / 2; h++) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    auto sw = __builtin_amdgcn_permlane16_swap(o2[2 * h][d], o2[2 * h + 1][d], false, false);
                    o2[2 * h][d] = sw[0];
                    o2[2 * h + 1][d] = sw[1];
                }
#endif
                u32x4 outv = u32x4{o2[2 * h][0], o2[2 * h][1], o2[2 * h + 1][0], o2[2 * h + 1][1]};
                {
                    // shortcut = the 1x1 layer's input at this pixel: patch pixel (r + 1, c + 1) of the stage-1 image, channels
                    // h*32 + run0 .. +7 = 16-B chunk h*4 + run0/8
                    const int qc = (r + 1) * P::PW + c + 1;
                    const bf16x8 rvv = *(const bf16x8 *)(in_img + qc * 128 + (((h * 4 + (run0 >> 3)) ^ ((qc >> 1) & 7)) << 4));
                    bf16x8 ov = __builtin_bit_cast(bf16x8, outv);
#pragma unroll
                    for (int e = 0; e < 8; e++) ov[e] = (__bf16)((float)ov[e] + (float)rvv[e]);
                    outv = __builtin_bit_cast(u32x4, ov);
                }
                if (ok) {
                    u32x4 *dst = (u32x4 *)(p.y + m * p.out_cs + h * 32 + run0);
                    if (p.nt_out) __builtin_nontemporal_store(outv, dst);
                    else *dst = outv;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ 3x3 / 1, 64 -> 128 (round 6)
// Darknet-53 layers 7 and 10 (64 -> 128 @152^2, + shortcut) and their training forwards.  On the implicit-GEMM tiles these layers are all
// prologue and epilogue: K = 576 is nine K steps, and every 128-pixel tile fetches its nine shifted copies of the same input rows (144 KiB)
// AND the whole filter (147 KiB) through L2 again -- 0.29-0.31 of the MFMA peak (forward 150 us at bs 32, training forward 273 us at bs 64).
// Here, as in conv3x3_c32_halo_kernel, the input patch of a 4 x 32 output tile (6 x 34 pixels of 128 B, halo included) goes HBM -> LDS ONCE
// (double-buffered, 26 KiB per tile) and the filter never moves: it is 9 taps x 64 x 128 bf16 = 147 KiB, too much for one wave's registers,
// so the WAVES SPLIT THE OUTPUT CHANNELS -- wave w owns channels 32 w .. 32 w + 31 (9 taps x 2 K steps x 2 channel groups = 36 fragments =
// 144 VGPRs) and walks ALL eight 16-pixel groups of the tile, two at a time (four independent accumulator chains).  Every wave reads every
// B fragment: 18 ds_read_b128 per 36 MFMAs of a wave and group, 125 B/clk per CU at the full MFMA rate -- half of the LDS read rate.
// The patch pixel is 8 x 16-B slots with slot ^= (column >> 1) & 7: the 16 pixels one quarter-wave reads (consecutive columns, one chunk)
// cover all 64 banks.  Taps accumulate in the order 0..8, channels 0..31 then 32..63 inside a tap -- the K order of the implicit-GEMM
// kernels -- so outputs are bit-identical to theirs (tests/test_conv_gpu.py).
namespace c64 {
constexpr int CIN = 64, COUT = 128, TWX = 32, TH = 4;
constexpr int PH = TH + 2, PW = TWX + 2, NPIX = PH * PW;       // 6 x 34 pixels
constexpr int NPIECE = (NPIX + 7) / 8;                         // 1-KiB direct-to-LDS pieces (8 pixels x 128 B)
constexpr int PPW = (NPIECE + 3) / 4;                          // pieces per wave
constexpr int BUF = PPW * 4 * 1024;                            // one patch buffer (28 KiB)
constexpr int BYTES = 2 * BUF;
constexpr int NGRP = TH * TWX / 16;                            // 16-pixel groups per tile: every wave walks all of them
__device__ __forceinline__ int swz(int pcol) { return (pcol >> 1) & 7; }
}  // namespace c64

// ACT is a template parameter: as a run-time switch the compiler if-converts the branch and evaluates the Mish exp / divide chain for every
// element of every leaky layer -- 240 VALU instructions per 36 MFMAs, which made the inference instantiation VALU-bound (173 us vs 163).
template <bool STATS, int ACT>
__global__ void __launch_bounds__(256, 2) conv3x3_c64_halo_kernel(const ConvParams p, int tiles_x, int tiles_y, int ntiles) {
    using namespace c64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int co0 = wave * 32;                                  // this wave's 32 output channels

    const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, nloc = gridDim.x >> 3;
    const int start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int len = q8 + (xcd < r8 ? 1 : 0);

    // the filter slice: fragment (tap t, K step ks, channel group cg) = rows co0 + cg*16 + fr, K columns t*64 + ks*32 + g*8 .. +7
    bf16x8 wfr[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int cg = 0; cg < 2; cg++)
                wfr[t][ks][cg] = *(const bf16x8 *)(p.w + (size_t)(co0 + cg * 16 + fr) * p.Kpad + t * c64::CIN + ks * 32 + g * 8);
    f32x4 sc[2], sh[2];
    if constexpr (!STATS) {
#pragma unroll
        for (int cg = 0; cg < 2; cg++) {
            sc[cg] = *(const f32x4 *)(p.scale + co0 + cg * 16 + g * 4);
            sh[cg] = *(const f32x4 *)(p.shift + co0 + cg * 16 + g * 4);
        }
    }
    float st_sum[2][4], st_sq[2][4];
#pragma unroll
    for (int cg = 0; cg < 2; cg++)
#pragma unroll
        for (int r = 0; r < 4; r++) st_sum[cg][r] = st_sq[cg][r] = 0.f;
    const float slope = p.slope;
    const int tiles_img = tiles_x * tiles_y;

    auto fill = [&](int id, char *buf) {               // piece k covers patch-linear pixels 8k .. 8k+7, 8 lanes (16-B chunks) per pixel
        const int img = id / tiles_img, rem = id - img * tiles_img;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int h0 = ty * TH - p.pad, w0 = tx * TWX - p.pad;
#pragma unroll 2
        for (int j = 0; j < PPW; j++) {
            const int piece = wave * PPW + j;
            const int q = piece * 8 + (lane >> 3);
            const int prow = q / PW, pcol = q - prow * PW;
            const int hi = h0 + prow, wi = w0 + pcol;
            const bool ok = q < NPIX && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const int chunk = (lane & 7) ^ swz(pcol);                    // the logical chunk stored at physical slot lane & 7
            const int off = (((img * p.H + hi) * p.W + wi) * p.in_cs + chunk * 8) * 2;
            buffer_load_lds16(p.x, p.x_bytes, buf + piece * 1024, ok ? off : (int)0x80000000, 0);
        }
    };
    int cur = 0;
    if (loc < len) fill(start + loc, smem);
    for (int i = loc; i < len; i += nloc) {
        const int id = start + i;
        const int img = id / tiles_img, rem = id - img * tiles_img;
        const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int ho0 = ty * TH, wo0 = tx * TWX;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char *patch = smem + cur * BUF;
        char *next_buf = smem + (cur ^ 1) * BUF;
        cur ^= 1;
        const int run0 = ((g & 1) ? 16 : 0) + (g >> 1) * 8;       // this lane's 16-B run of the wave's 32 channels (after the swap below)
        if (i + nloc < len) fill(id + nloc, next_buf);
        // ---- all eight 16-pixel groups, two per trip (four accumulator chains): trip j = tile row j, columns u*16 + fr
        // (Shortcut rows: vector-memory results return IN ORDER, so a row requested behind the next patch's fills is only usable once those
        //  have landed -- the inference launches with a shortcut pay ~25 us at bs 32 for it.  Requesting a half-tile's rows before the fills
        //  and unrolling the trips was built: it spills (the filter alone is 144 VGPRs) and measured no better.)
#pragma unroll 1
        for (int jr = 0; jr < TH; jr++) {
            const size_t mbase = ((size_t)img * p.Ho + ho0 + jr) * p.Wo + wo0 + fr;
            const bool okr = ho0 + jr < p.Ho;
            bool ok_[2];
#pragma unroll
            for (int u = 0; u < 2; u++) ok_[u] = okr && wo0 + u * 16 + fr < p.Wo;
            bf16x8 rv[2];
            if constexpr (!STATS) {
                if (p.res) {
#pragma unroll
                    for (int u = 0; u < 2; u++) rv[u] = ok_[u] ? *(const bf16x8 *)(p.res + (mbase + u * 16) * p.res_cs + co0 + run0) : bf16x8{};
                }
            }
            f32x4 acc[2][2];
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int cg = 0; cg < 2; cg++) acc[u][cg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int kh = t / 3, kw = t - 3 * kh;
#pragma unroll
                for (int ks = 0; ks < 2; ks++) {
                    bf16x8 xf[2];
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int pcol = u * 16 + fr + kw;
                        const int q = (jr + kh) * PW + pcol;
                        xf[u] = *(const bf16x8 *)(patch + q * 128 + (((ks * 4 + g) ^ swz(pcol)) << 4));
                    }
#pragma unroll
                    for (int u = 0; u < 2; u++)
#pragma unroll
                        for (int cg = 0; cg < 2; cg++)
                            acc[u][cg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[t][ks][cg], xf[u], acc[u][cg], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                unsigned o2[2][2];
#pragma unroll
                for (int cg = 0; cg < 2; cg++) {
                    bf16x4 o;
                    if constexpr (STATS) {
#pragma unroll
                        for (int rr = 0; rr < 4; rr++) {
                            o[rr] = (__bf16)acc[u][cg][rr];
                            const float qv = ok_[u] ? (float)o[rr] : 0.f;        // statistics of the values as stored
                            st_sum[cg][rr] += qv;
                            st_sq[cg][rr] += qv * qv;
                        }
                    } else {
#pragma unroll
                        for (int rr = 0; rr < 4; rr++) {
                            float v = acc[u][cg][rr] * sc[cg][rr] + sh[cg][rr];
                            if constexpr (ACT == RYOLO_ACT_LEAKY) v = v > 0.f ? v : v * slope;
                            else if constexpr (ACT == RYOLO_ACT_MISH) v = mish(v);
                            o[rr] = (__bf16)v;
                        }
                    }
                    const uint2 uu = __builtin_bit_cast(uint2, o);
                    o2[cg][0] = uu.x;
                    o2[cg][1] = uu.y;
                }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    auto sw = __builtin_amdgcn_permlane16_swap(o2[0][d], o2[1][d], false, false);
                    o2[0][d] = sw[0];
                    o2[1][d] = sw[1];
                }
#endif
                u32x4 outv = u32x4{o2[0][0], o2[0][1], o2[1][0], o2[1][1]};
                if constexpr (!STATS) {
                    if (p.res) {
                        bf16x8 ov = __builtin_bit_cast(bf16x8, outv);
#pragma unroll
                        for (int e = 0; e < 8; e++) ov[e] = (__bf16)((float)ov[e] + (float)rv[u][e]);
                        outv = __builtin_bit_cast(u32x4, ov);
                    }
                }
                if (ok_[u]) {
                    u32x4 *dst = (u32x4 *)(p.y + (mbase + u * 16) * p.out_cs + co0 + run0);
                    if (p.nt_out) __builtin_nontemporal_store(outv, dst);
                    else *dst = outv;
                }
            }
        }
    }
    if constexpr (STATS) {
        // the 16 lanes of a row hold the same channels: DPP row sums; the wave owns its 32 channels outright, so lane (fr < 8, g) adds the
        // totals of channel co0 + (fr >> 2) * 16 + g * 4 + (fr & 3) straight to the fp64 partial row (no cross-wave combine)
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int cg = 0; cg < 2; cg++)
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const float a = row16_sum(st_sum[cg][rr]), b = row16_sum(st_sq[cg][rr]);
                if (fr == cg * 4 + rr) {
                    ta = a;
                    tb = b;
                }
            }
        if (fr < 8) {
            const int ch = co0 + (fr >> 2) * 16 + g * 4 + (fr & 3);
            double *row = p.stat_part + (size_t)(blockIdx.x % STAT_ROWS) * 2 * p.stat_cpad;
            atomicAdd(row + ch, (double)ta);
            atomicAdd(row + p.stat_cpad + ch, (double)tb);
        }
    }
}

template <bool STATS, int ACT>
int launch_stem64(ConvParams &p, int grid, int tiles_x, int tiles_y, int ntiles, hipStream_t stream) {
    hipLaunchKernelGGL((conv3x3_c64_halo_kernel<STATS, ACT>), dim3((unsigned)grid), dim3(256), c64::BYTES, stream, p, tiles_x, tiles_y, ntiles);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

template <int S, bool STATS>
int launch_stem(ConvParams &p, int grid, int tiles_x, int tiles_y, int ntiles, hipStream_t stream) {
    constexpr int smem = Patch<S>::BYTES;
    static bool attr_done = false;
    if (!attr_done) {
        if (smem > 64 * 1024 && hipFuncSetAttribute((const void *)conv3x3_c32_halo_kernel<S, STATS>,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return RYOLO_ELAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL((conv3x3_c32_halo_kernel<S, STATS>), dim3((unsigned)grid), dim3(256), smem, stream, p, tiles_x, tiles_y, ntiles);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

}  // namespace

bool conv_stem_eligible(const ConvParams &p, int ksize) {
    return ksize == 3 && p.Cin == CIN && p.Cout == COUT && p.pad == 1 && (p.stride == 1 || p.stride == 2) && p.fast && p.os == 1 &&
           p.ups == 1 && p.ntaps == 9 && (p.in_cs & 7) == 0 && (p.out_cs & 7) == 0 && (!p.res || (p.res_cs & 7) == 0) &&
           !(p.stat_part && p.res) && (long long)p.N * p.H * p.W * p.in_cs * 2 < 0x7fffff00ll;
}

bool conv_stem64_eligible(const ConvParams &p, int ksize) {
    return ksize == 3 && p.Cin == c64::CIN && p.Cout == c64::COUT && p.pad == 1 && p.stride == 1 && p.fast && p.os == 1 &&
           p.ups == 1 && p.ntaps == 9 && (p.in_cs & 7) == 0 && (p.out_cs & 7) == 0 && (!p.res || (p.res_cs & 7) == 0) &&
           !(p.stat_part && p.res) && (long long)p.N * p.H * p.W * p.in_cs * 2 < 0x7fffff00ll;
}

int launch_conv_stem64(ConvParams &p, int cus, hipStream_t stream) {
    const int tiles_x = (p.Wo + c64::TWX - 1) / c64::TWX, tiles_y = (p.Ho + c64::TH - 1) / c64::TH;
    const long long nt = (long long)p.N * tiles_x * tiles_y;
    if (nt > 0x7fffffffll) return RYOLO_EINVAL;
    RYOLO_CONV_DRY_RUN(RYOLO_CONV_KERNEL_STEM64);
    int grid = (2 * cus) & ~7;
    if (grid < 8) grid = 8;
    if (p.stat_part) return launch_stem64<true, RYOLO_ACT_LINEAR>(p, grid, tiles_x, tiles_y, (int)nt, stream);
    if (p.act == RYOLO_ACT_LEAKY) return launch_stem64<false, RYOLO_ACT_LEAKY>(p, grid, tiles_x, tiles_y, (int)nt, stream);
    if (p.act == RYOLO_ACT_MISH) return launch_stem64<false, RYOLO_ACT_MISH>(p, grid, tiles_x, tiles_y, (int)nt, stream);
    return launch_stem64<false, RYOLO_ACT_LINEAR>(p, grid, tiles_x, tiles_y, (int)nt, stream);
}

// The stride-2 data gradient one level down (Darknet-53 layer 5: 3x3 / 2, 64 -> 128; 128 -> 64 channels in the gradient's direction).  The filter (9 x 128 x 64 bf16 = 147 KB) only fits the registers of a workgroup if the WAVES SPLIT THE OUTPUT
// CHANNELS: wave w owns channels 16 w .. 16 w + 15 (9 taps x 4 K steps = 36 fragments = 144 VGPRs) and walks ALL pixel groups of the
// tile, so every wave reads every B fragment (4 x the LDS traffic of the 64 -> 32 kernel: still below the LDS bandwidth) and a
// pixel's 128-B row is written as four 32-B pieces (8 B per lane).  Tiles 8 x 32, patch pixels of 256 B with slot ^= column & 15.
// 488 -> 349 us.  (The stride-1 instantiation for layers 7 / 10 was built too: 286 us against the 256 x 64 tile's 278 -- no gain, and
// that tile carries the folded BatchNorm reduce; removed.)
struct DgTile128 {
    static constexpr int TH = 8, TW = 32;
    static constexpr int PH = TH / 2 + 1, PW = TW / 2 + 1, NPIX = PH * PW;
    static constexpr int NPIECE = (NPIX + 3) / 4;                // 1-KiB pieces: 4 pixels x 256 B
    static constexpr int PPW = (NPIECE + 3) / 4, BUF = PPW * 4 * 1024;
};

__global__ void __launch_bounds__(256, 2) dgrad3x3_s2_c128_kernel(const DgS2Params p) {
    using T = DgTile128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int q8 = p.ntiles >> 3, r8 = p.ntiles & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, nloc = gridDim.x >> 3;
    const int start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int len = q8 + (xcd < r8 ? 1 : 0);
    bf16x8 wf[9][4];                                   // rows = this wave's 16 output channels, K = (tap, co): 4 K steps of 32 per tap
    {
        constexpr int NCLS = 4;
        const __bf16 *img = p.w;
        int t0 = 0;
#pragma unroll
        for (int cls = 0; cls < NCLS; cls++) {
            const int nt = cls == 0 ? 1 : (cls == 3 ? 4 : 2), kpad = nt * 128;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (t >= nt) break;
#pragma unroll
                for (int ks = 0; ks < 4; ks++) wf[t0 + t][ks] = *(const bf16x8 *)(img + (size_t)(wave * 16 + fr) * kpad + t * 128 + ks * 32 + g * 8);
            }
            t0 += nt;
            img += 128 * kpad + 128;
        }
    }
    const int tiles_img = p.tiles_x * p.tiles_y;
    auto fill = [&](int id, char *buf) {               // piece k covers patch-linear pixels 4k .. 4k+3, 16 lanes (16-B chunks) per pixel
        const int img = id / tiles_img, rem = id - img * tiles_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int h0 = ty * (T::TH / 2), w0 = tx * (T::TW / 2);
#pragma unroll 2
        for (int j = 0; j < T::PPW; j++) {
            const int piece = wave * T::PPW + j;
            const int q = piece * 4 + (lane >> 4);
            const int prow = q / T::PW, pcol = q - prow * T::PW;
            const int ho = h0 + prow, wo = w0 + pcol;
            const bool ok = q < T::NPIX && (unsigned)ho < (unsigned)p.Ho && (unsigned)wo < (unsigned)p.Wo;
            const int chunk = (lane & 15) ^ (pcol & 15);
            const int off = (((img * p.Ho + ho) * p.Wo + wo) * p.dz_cs + chunk * 8) * 2;
            buffer_load_lds16(p.dz, p.dz_bytes, buf + piece * 1024, ok ? off : (int)0x80000000, 0);
        }
    };
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    int cur = 0;
    if (loc < len) fill(start + loc, smem);
    for (int i = loc; i < len; i += nloc) {
        const int id = start + i;
        const int img = id / tiles_img, rem = id - img * tiles_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int hi0 = ty * T::TH, wi0 = tx * T::TW;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (i + nloc < len) fill(id + nloc, smem + (cur ^ 1) * T::BUF);
        const char *patch = smem + cur * T::BUF;
        cur ^= 1;
        auto group = [&](auto Ac, auto Bc, auto T0c, int r) {      // 16 pixels of output row r (parity class (A, B)): wi0 + 2 fr + B
            constexpr int A = decltype(Ac)::value, B = decltype(Bc)::value, T0 = decltype(T0c)::value;
            constexpr int NTA = A ? 2 : 1, NTB = B ? 2 : 1;
            const int hi = hi0 + r, wi = wi0 + 2 * fr + B;
            const bool ok = hi < p.H && wi < p.W;
            const size_t m = ((size_t)img * p.H + hi) * p.W + wi;
            const int coff = wave * 16 + g * 4;
            u32x2 rv = u32x2{0u, 0u};
            if (p.res && ok) rv = *(const u32x2 *)(p.res + m * p.dx_cs + coff);
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ta = 0; ta < NTA; ta++)
#pragma unroll
                for (int tb = 0; tb < NTB; tb++) {
                    const int prow = (r - A) / 2 + (A ? ta : 0), pcol = fr + (B ? tb : 0);
                    const char *px = patch + (prow * T::PW + pcol) * 256;
                    const int sw = pcol & 15;
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) {
                        const bf16x8 xf = *(const bf16x8 *)(px + (((g + 4 * ks) ^ sw) << 4));
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[T0 + ta * NTB + tb][ks], xf, acc, 0, 0, 0);
                    }
                }
            bf16x4 o;
#pragma unroll
            for (int rr = 0; rr < 4; rr++) o[rr] = (__bf16)acc[rr];
            if (p.res) {
                const bf16x4 rb = __builtin_bit_cast(bf16x4, rv);
#pragma unroll
                for (int rr = 0; rr < 4; rr++) o[rr] = (__bf16)((float)o[rr] + (float)rb[rr]);
            }
            if (ok) {
                u32x2 *dst = (u32x2 *)(p.dx + m * p.dx_cs + coff);
                if (p.nt_out) __builtin_nontemporal_store(__builtin_bit_cast(u32x2, o), dst);
                else *dst = __builtin_bit_cast(u32x2, o);
            }
        };
        {
#pragma unroll
            for (int rp = 0; rp < T::TH / 2; rp++) {
                group(ic<0>{}, ic<0>{}, ic<0>{}, 2 * rp);
                group(ic<0>{}, ic<1>{}, ic<1>{}, 2 * rp);
                group(ic<1>{}, ic<0>{}, ic<3>{}, 2 * rp + 1);
                group(ic<1>{}, ic<1>{}, ic<5>{}, 2 * rp + 1);
            }
        }
    }
}

int launch_conv_stem_dgrad(int cdz, int stride, const void *dz, int dz_cs, const void *w_classes, void *dx, int dx_cs, int accumulate, int N,
                           int H, int W, int nt_out, int cus, hipStream_t stream) {
    const int Ho = stride == 2 ? (H - 1) / 2 + 1 : H, Wo = stride == 2 ? (W - 1) / 2 + 1 : W;
    const unsigned long long dzb = (((unsigned long long)N * Ho * Wo - 1) * dz_cs + cdz) * 2ull;
    if ((stride != 1 && stride != 2) || (cdz != 64 && !(cdz == 128 && stride == 2)) || dzb >= 0x7fffff00ull || (dz_cs & 7) || (dx_cs & 7) || dz_cs < cdz ||
        dx_cs < cdz / 2)
        return RYOLO_EINVAL;
    DgS2Params q;
    q.dz = (const __bf16 *)dz; q.dz_bytes = (unsigned)dzb; q.dz_cs = dz_cs; q.w = (const __bf16 *)w_classes;
    q.dx = (__bf16 *)dx; q.dx_cs = dx_cs; q.res = accumulate ? (const __bf16 *)dx : nullptr;
    q.H = H; q.W = W; q.Ho = Ho; q.Wo = Wo;
    const int th = cdz == 64 ? (stride == 2 ? DgTile<2>::TH : DgTile<1>::TH) : DgTile128::TH;
    const int tw = cdz == 64 ? (stride == 2 ? DgTile<2>::TW : DgTile<1>::TW) : DgTile128::TW;
    q.tiles_x = (W + tw - 1) / tw; q.tiles_y = (H + th - 1) / th;
    const long long nt = (long long)q.tiles_x * q.tiles_y * N;
    if (nt >= 0x7fffffff) return RYOLO_EINVAL;
    q.ntiles = (int)nt; q.nt_out = nt_out;
    int grid = (2 * cus) & ~7;
    if (grid < 8) grid = 8;
    if (cdz == 64) {
        if (stride == 2) hipLaunchKernelGGL(dgrad3x3_c64_kernel<2>, dim3((unsigned)grid), dim3(256), 2 * DgTile<2>::BUF, stream, q);
        else hipLaunchKernelGGL(dgrad3x3_c64_kernel<1>, dim3((unsigned)grid), dim3(256), 2 * DgTile<1>::BUF, stream, q);
    } else {
        hipLaunchKernelGGL(dgrad3x3_s2_c128_kernel, dim3((unsigned)grid), dim3(256), 2 * DgTile128::BUF, stream, q);
    }
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

// p: x (8-channel NHWC, x_bytes), w, scale, shift, y, out_cs, H = Ho, W = Wo, N, Kpad, act, slope, nt_out; no statistics (those passes keep
// the direct kernel: conv.hip)
int launch_conv0_halo(ConvParams &p, const float *slope_dev, int round_z, int cus, hipStream_t stream) {
    if (!p.y || p.stat_part) return RYOLO_EINVAL;
    const int tiles_x = (p.Wo + C0_TW - 1) / C0_TW, tiles_y = (p.Ho + C0_TH - 1) / C0_TH;
    const long long nt = (long long)p.N * tiles_x * tiles_y;
    if (nt > 0x7fffffffll) return RYOLO_EINVAL;
    int grid = (4 * cus) & ~7;
    if (grid < 8) grid = 8;
    return launch_conv0_halo_t(p, slope_dev, round_z, p.act, grid, tiles_x, tiles_y, (int)nt, stream);
}

int launch_conv_stem(ConvParams &p, int cus, hipStream_t stream) {
    const int th = p.stride == 1 ? Patch<1>::TH : Patch<2>::TH;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + th - 1) / th;
    const long long nt = (long long)p.N * tiles_x * tiles_y;
    if (nt > 0x7fffffffll) return RYOLO_EINVAL;
    RYOLO_CONV_DRY_RUN(RYOLO_CONV_KERNEL_STEM);
    int grid = (2 * cus) & ~7;
    if (grid < 8) grid = 8;
    if (p.stat_part) return p.stride == 1 ? launch_stem<1, true>(p, grid, tiles_x, tiles_y, (int)nt, stream)
                                          : launch_stem<2, true>(p, grid, tiles_x, tiles_y, (int)nt, stream);
    return p.stride == 1 ? launch_stem<1, false>(p, grid, tiles_x, tiles_y, (int)nt, stream)
                         : launch_stem<2, false>(p, grid, tiles_x, tiles_y, (int)nt, stream);
}

static int launch_pair(ConvParams &p, const PairFirst &f, int cus, hipStream_t stream) {
    using G = PairGeom;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + G::P::TH - 1) / G::P::TH;
    const long long nt = (long long)p.N * tiles_x * tiles_y;
    if (nt > 0x7fffffffll) return RYOLO_EINVAL;
    static bool attr_done = false;
    if (!attr_done) {
        if (G::BYTES > 64 * 1024 && hipFuncSetAttribute((const void *)conv_stem_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                        G::BYTES) != hipSuccess)
            return RYOLO_ELAUNCH;
        attr_done = true;
    }
    int grid = (2 * cus) & ~7;
    if (grid < 8) grid = 8;
    hipLaunchKernelGGL(conv_stem_pair_kernel, dim3((unsigned)grid), dim3(256), G::BYTES, stream, p, f, tiles_x, tiles_y, (int)nt);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

// kind of pair (first, second) the fused launch serves: 2 = (1x1, 64 -> 32) -> (3x3/1 pad 1, 32 -> 64) with the shortcut from the first
// layer's input; 0 = none
int conv_stem_pair_kind(const ryolo_conv_desc *a, const ryolo_conv_desc *b, int shortcut_from_input) {
    if (!a || !b || a->upsample != 1 || b->upsample != 1 || a->N != b->N) return 0;
    const int aho = (a->H + 2 * a->pad - a->ksize) / a->stride + 1, awo = (a->W + 2 * a->pad - a->ksize) / a->stride + 1;
    if (aho != b->H || awo != b->W || a->Cout != b->Cin || b->Cin != CIN || b->Cout != COUT || b->ksize != 3 || b->pad != 1) return 0;
    if ((long long)a->N * a->H * a->W * a->in_cstride * 2 >= 0x7fffff00ll) return 0;
    if (a->ksize == 1 && a->stride == 1 && a->pad == 0 && a->Cin == 64 && b->stride == 1 && shortcut_from_input && (a->in_cstride & 7) == 0) return 2;
    return 0;
}

int launch_conv_stem_pair(int kind, ConvParams &p, const void *x, unsigned x_bytes, int in_cs, int H, int W, const void *w_first, int kpad_first,
                          const float *scale_first, const float *shift_first, int act_first, float slope_first, int cus, hipStream_t stream) {
    PairFirst f;
    f.x = (const __bf16 *)x; f.w = (const __bf16 *)w_first; f.scale = scale_first; f.shift = shift_first; f.x_bytes = x_bytes;
    f.in_cs = in_cs; f.Kpad = kpad_first; f.act = act_first; f.slope = slope_first; f.H = H; f.W = W;
    (void)kind;
    return launch_pair(p, f, cus, stream);
}

}  // namespace ryolo_detail
