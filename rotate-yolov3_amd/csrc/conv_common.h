// rotate-yolov3_amd/csrc/conv_common.h -- types and device helpers shared by the convolution translation units
// (conv.hip: 128x128 / 256x64 / 256x32 tiles, first-layer direct kernel, packers; conv_mp.hip: the 256-wide
// multi-phase kernel).  Internal to libryolo_hip.so; the C ABI is include/ryolo.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/ryolo.h"

namespace ryolo_detail {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) void *lds_vp;
typedef const __attribute__((address_space(1))) void *glb_vp;

constexpr int BK = 64;   // K elements per step: one 128-B LDS row per tile row
// Outputs of this size and more are stored non-temporally (measured: layer 0 at bs 32, 757 MB of output, 0.284 -> 0.177 ms;
// smaller outputs are re-read from L2 / the Infinity Cache by the next kernel and keep the default policy)
constexpr long long NT_OUT_MIN_BYTES = 128ll << 20;
constexpr int STAT_ROWS = 128;   // partial rows the statistic atomics are spread over (tile or workgroup index mod STAT_ROWS); every flush is per
                                 // workgroup now, so 128 rows keep the fp64 atomics uncontended and bn_finalize reads a quarter of what 512 cost

struct ConvParams {
    const __bf16 *x;       // input, NHWC, pixel stride in_cs (elements); already offset to its channel slice
    const __bf16 *w;       // packed weights [Cout_pad][Kpad] + 256-B zero tail
    const float *scale;    // [Cout_pad]
    const float *shift;    // [Cout_pad]
    const __bf16 *res;     // residual (same pixel grid as the output, before upsampling) or nullptr
    __bf16 *y;             // output, NHWC, pixel stride out_cs
    int N, H, W, Cin, in_cs;
    int Ho, Wo, Cout, out_cs, res_cs;
    int stride, pad;
    int K, Kpad, M;
    int cin_log2;          // 3x3 only: log2(Cin) when Cin is a power of two, else -1 (then Cin % 64 == 0)
    int act;               // RYOLO_ACT_*
    float slope;
    int ups;               // 1, or 2 = write every output pixel to its 2x2 nearest-upsampled positions
    int nt;                // number of channel tiles
    unsigned x_bytes, w_bytes;   // FAST path buffer descriptors
    int fast;
    int taps2;             // FAST path with C_in == 32: two filter taps per 64-wide K step (3x3, regular window)
    int no_persist;        // tile bit 0x200: keep the one-tile-per-workgroup grid (tests, A/B timing)
    int force_persist;     // tile bit 0x800: persistent grid also for 3x3 (tests, A/B timing)
    // generalisations used by the training kernels (FAST path only):
    int ntaps;             // taps actually visited by the K loop (forward: KS*KS)
    int tap_dy[9], tap_dx[9];   // tap t reads input pixel (hi0 + tap_dy[t], wi0 + tap_dx[t])
    int os, ooy, oox, OH, OW;   // output pixel of grid cell (i, j): (i*os + ooy, j*osx + oox) in an [N, OH, OW] tensor
    int osx;                    // column step (== os except for the x-fused stride-2 dgrad: rows step 2, super-pixels step 1)
    int ntiles;            // persistent kernel: number of (m, n) tiles
    unsigned magic_wo, magic_ho, magic_nt;   // ceil(2^32 / d): multiply-high division by Wo, Ho, nt
    int use_magic;         // the multiply-high divisions by Wo / Ho are exact for every m < M (host check)
    unsigned y_bytes, res_bytes;   // conv_mp.hip: extents of the output / residual tensors (buffer descriptors)
    double *stat_part;     // optional [STAT_ROWS][2][Cout_pad] partial sums of z and z*z (BatchNorm statistics); zeroed by the caller.
                           // fp64: the per-wave fp32 sums are added with 64-bit atomics, so the order in which the waves arrive
                           // does not show in the fp32 mean / invstd (an fp32 accumulator made the step irreproducible at 1e-7)
    int stat_cpad;
    int nt_out;            // the output tensor is too large to stay in the caches until it is read again (>= NT_OUT_MIN_BYTES): non-temporal stores
    int reg3;              // conv_mq.hip: the taps are the regular 3x3 window, tap t = (t / 3, t % 3) (cheap border masks)
    int korder = 0;        // conv_mq.hip: K-tile visiting order, 0 tap-major, 1 channel-major (the nine taps of a 64-channel slice back to back)
    int dbg0, dbg1;        // ablation builds (-DRYOLO_MP_ABLATION) only
    int pw_nb, pw_mb;      // conv_pw.hip: channel blocks, row blocks of the launch
    unsigned *trace;       // ablation builds only (conv_pw.hip cycle stamps)
    int pw_grid_cap;       // conv_pw.hip, tests: workgroups per XCD (0 = fill the chip); lets a small tensor reach the steady state of the ring
};

// BatchNorm-backward reduce folded into the 1x1 data gradient that stores the block's final dy (conv.hip: conv_igemm_persist_kernel<..., BNRED>,
// conv_pw.hip MODE 3)
struct BnRed {
    const __bf16 *z;        // the consumer block's conv output (what its BatchNorm normalised), pixel stride z_cs
    int z_cs;
    const float *scale, *shift, *mean, *invstd;   // [C] of that BatchNorm (batch statistics of the forward)
    const float *slope;     // PReLU / leaky slope (device scalar)
    float *part;            // [gridDim.x][3][C] fp32; every workgroup zeroes its row first
    unsigned z_bytes = 0;   // conv_mq.hip: extent of z (buffer descriptor), filled by its launcher
};

__device__ __forceinline__ float mish(float v) {
    // x * tanh(softplus(x)) = x * (n - 1) / (n + 1) with n = (1 + e^x)^2; e^x clamped so n stays finite
    const float e = __expf(fminf(v, 20.f));
    const float n = (1.f + e) * (1.f + e);
    return v * (n - 1.f) / (n + 1.f);
}

// Sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), result in every lane.  Four v_add_f32 with DPP operands
// (the two quad swaps, half-row mirror, row mirror): pure VALU, where the ds_bpermute behind __shfl_xor would take LDS
// issue slots from the K loops of the co-resident workgroups.  Same pairing as the xor butterfly (1, 2, 4, 8), so the
// sums are bit-identical to it.
__device__ __forceinline__ float row16_sum(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
#endif
    return v;
}

// v(lane) + v(lane ^ O) for O = 4 / 8 (DPP rotation inside the 16-lane row: lane + O mod 16, the same partner set once both steps ran),
// 16 / 32 (v_permlane16_swap / v_permlane32_swap of the value with itself: both operands then hold the two halves side by side).  Pure
// VALU: the ds_bpermute behind __shfl_xor costs an LDS round trip per call (48 of them per output tile made the folded BatchNorm reduce
// of the one-tile-per-workgroup kernels 16 % slower than the plain data gradient).
template <int O>
__device__ __forceinline__ float lane_xor_sum(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (O == 4) {
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));   // row_ror:4
    } else if constexpr (O == 8) {
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));   // row_ror:8
    } else if constexpr (O == 16) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto s = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = __builtin_bit_cast(float, (unsigned)s[0]) + __builtin_bit_cast(float, (unsigned)s[1]);
    } else {
        static_assert(O == 32, "partner distance");
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto s = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v = __builtin_bit_cast(float, (unsigned)s[0]) + __builtin_bit_cast(float, (unsigned)s[1]);
    }
#endif
    return v;
}
// sum over the lanes that share (lane mod CPR), CPR in {4, 8, 16}: every lane of the class ends with the same total
template <int CPR>
__device__ __forceinline__ float lane_class_sum(float v) {
    if constexpr (CPR <= 4) v = lane_xor_sum<4>(v);
    if constexpr (CPR <= 8) v = lane_xor_sum<8>(v);
    v = lane_xor_sum<16>(v);
    return lane_xor_sum<32>(v);
}

// 16-B-per-lane buffer load straight into LDS (lane-linear at `lds`); lanes whose byte offset is outside
// [0, bytes) get zeros.  The builtins exist only in the device pass.
__device__ __forceinline__ void buffer_load_lds16(const void *base, unsigned bytes, char *lds, int voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_vp)lds, 16, voffset, soffset, 0, 0);
#endif
}

// 16-B buffer store whose soffset is an SGPR, safe against the store-data hazard.  A VMEM store of more than 64 bits reads
// its data VGPRs a cycle after issue, and a VALU instruction that overwrites them right behind the store races that read.
// The compiler inserts the required wait state only when soffset is NOT a register (LLVM GCNHazardRecognizer,
// createsVALUHazard); on gfx950 the race is real with a register soffset too: with `store v[138:141] ... s85` directly
// followed by `v_pk_fma_f32 v[138:139]`, lanes 12-15 of every 16-lane row of the second data dword reached memory as the
// NEXT fragment's bits (found as wrong z in the statistics instantiation; the same fault was the unexplained Mish garbage
// of conv_tw.hip).  The asm below READS the data registers, so no write to them can be scheduled in front of it, and its
// s_nop supplies the wait states; tools/check_mp_isa.py fails the build if any 12-/16-B store in the library has its data
// overwritten with fewer than two wait states in between.
template <int AUX = 0>
__device__ __forceinline__ void buffer_store16_soff(u32x4 v, __amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voffset, soffset, AUX);
    asm volatile("s_nop 1" ::"v"(v));
#endif
}

// n / d by multiply-high with magic = ceil(2^32 / d); magic == 0 encodes d == 1
__device__ __forceinline__ int udiv_magic(int n, unsigned magic) {
    return magic ? (int)__umulhi((unsigned)n, magic) : n;
}

// m -> (t = m / Wo, wo = m % Wo), t -> (img = t / Ho, ho): multiply-high when the host found it exact (use_magic), else `/`
__device__ __forceinline__ void split_pixel(int m, int Wo, int Ho, unsigned magic_wo, unsigned magic_ho, int use_magic, int &wo,
                                            int &ho, int &img) {
    if (use_magic) {
        const int t = udiv_magic(m, magic_wo);
        wo = m - t * Wo;
        img = udiv_magic(t, magic_ho);
        ho = t - img * Ho;
    } else {
        const int t = m / Wo;
        wo = m % Wo;
        ho = t % Ho;
        img = t / Ho;
    }
}


// ryolo_conv_kernel_choice() / ryolo_conv_dgrad_kernel_choice(): a dry run of the dispatch.  While this pointer is set (conv.hip),
// every launcher of the convolution units writes the code of the kernel it is about to launch there and returns RYOLO_OK instead
// of launching -- the decision is reported from the launch site itself, behind every size guard and fall-through, not from a
// parallel copy of the decision tree.
extern thread_local int *g_conv_choice;
#define RYOLO_CONV_DRY_RUN(code)          \
    do {                                  \
        if (ryolo_detail::g_conv_choice) { \
            *ryolo_detail::g_conv_choice = (code); \
            return RYOLO_OK;              \
        }                                 \
    } while (0)

// conv_mp.hip: 256-channel x BM-pixel workgroup tile, 8 waves, multi-phase K loop (FAST path only).
// Returns RYOLO_EINVAL when the shape does not qualify (caller falls back to the other tiles).
int launch_conv_mp(ConvParams &p, int bm /* 256, 192, 0 = pick */, int variant, hipStream_t stream);
int conv_mp_pick_bm(const ConvParams &p);
bool conv_mp_eligible(const ConvParams &p);
// conv_mq.hip: 128-pixel x 256-channel tile, 4 waves, two independent workgroups per CU (same eligibility as conv_mp)
int launch_conv_mq(ConvParams &p, int variant, hipStream_t stream);
// ... and its 128-CHANNEL members (round 5): bm = 128 or 64 pixels per tile; C_out % 128 == 0, otherwise conv_mq's conditions.  bnred
// (stride-1 launches without statistics): the folded BatchNorm reduce, one row of part[conv_mq128_grid()][3][C] per workgroup
bool conv_mq128_eligible(const ConvParams &p);
int conv_mq128_grid(const ConvParams &p, int bm);    // workgroups of the launch (= rows of BatchNorm-reduce partials), 0 = not served
int launch_conv_mq128(ConvParams &p, int bm, const BnRed *bnred, hipStream_t stream);
// conv_stem.hip: 3x3, C_in 32 -> C_out 64, stride 1 / 2: the input patch of an 8 x 32 output block staged once, the filter in registers
bool conv_stem_eligible(const ConvParams &p, int ksize);
int launch_conv_stem(ConvParams &p, int cus, hipStream_t stream);
bool conv_stem64_eligible(const ConvParams &p, int ksize);      // conv_stem.hip: 3x3 / 1, 64 -> 128
int launch_conv_stem64(ConvParams &p, int cus, hipStream_t stream);
// conv_stem.hip: Darknet-53 layer 0 (3x3, 8 -> 32) forward with its input patch staged in LDS (no statistics); slope_dev: optional device
// scalar for leaky / PReLU; round_z: BatchNorm applied to z rounded to bf16 (training recompute)
int launch_conv0_halo(ConvParams &p, const float *slope_dev, int round_z, int cus, hipStream_t stream);
// conv_stem.hip: the data gradients of the two 3x3 32 -> 64 stem layers (64 -> 32 channels in the gradient's direction) as one persistent
// launch each -- stride 2: all four output-parity classes from one staged dz patch, w_classes = the four classic class images of
// ryolo_conv_pack_weights_dgrad; stride 1: its single nine-tap image
// cdz = channels of dz: 64 (layers 1 / 3) or 128 (layer 5, stride 2 only: the waves split the output channels)
int launch_conv_stem_dgrad(int cdz, int stride, const void *dz, int dz_cs, const void *w_classes, void *dx, int dx_cs, int accumulate, int N,
                           int H, int W, int nt_out, int cus, hipStream_t stream);
// ... and two stem layers in one launch (inference): the 32-channel tensor between them is computed into LDS, never stored
int conv_stem_pair_kind(const ryolo_conv_desc *first, const ryolo_conv_desc *second, int shortcut_from_input);
int launch_conv_stem_pair(int kind, ConvParams &p /* the second layer */, const void *x, unsigned x_bytes, int in_cs, int H, int W,
                          const void *w_first, int kpad_first, const float *scale_first, const float *shift_first, int act_first,
                          float slope_first, int cus, hipStream_t stream);
// conv_pw.hip: 1x1 stride-1 layers (and their data gradients), weight-stationary: the filter slice in registers, rows through an LDS ring
bool conv_pw_eligible(const ConvParams &p, int ksize);
bool conv_pw_preferred(const ConvParams &p);        // the shapes on which it beats the 128 x 128 tile (auto dispatch)
int conv_pw_grid(const ConvParams &p);             // workgroups (= rows of BatchNorm-reduce partials) of the launch, 0 = not served
int launch_conv_pw(ConvParams &p, const void *bnred /* const BnRed * or nullptr */, hipStream_t stream);
// a YOLO head (1x1, K = 256, na*no <= 512 channels) decoded from the accumulators: io / p rows out, no head tensor
bool conv_pw_decode_supported(const ConvParams &p, int na, int no);
int launch_conv_pw_decode(ConvParams &p, float *io, long long io_img_rows, long long io_row0, float *pout, const float *anchors, int na, int no,
                          float stride, float cf, int arc, hipStream_t stream);
#ifdef RYOLO_MP_ABLATION
int ryolo_mp_ablation_variant(int slot);   // conv_mp.hip: VAR code stored in debug slot `slot` (ablation builds only)
#endif

// Tuning switches.  The shipped library knows SIX: the dispatch selections the tests and the A/B legs of tools/measure_round.sh use.  They are
// read from the environment ONCE (first use) and can be changed in-process through ryolo_set_tuning() (include/ryolo.h) -- no getenv on a
// launch path, no race with a setenv from another thread.  Every other environment switch of rounds 3-5 (thresholds, sweep orders, slab
// sizes, the 128-channel conv_mq family ...) exists in the measurement build only (-DRYOLO_MP_ABLATION, abl_env()).
enum TuneKey { TUNE_CONV3X3 = 0, TUNE_CONV1X1, TUNE_RNMS_TILES, TUNE_MQ_KORDER, TUNE_BN_REDUCE_TILES, TUNE_STEM_DGRAD, TUNE_COUNT };
const char *tune(TuneKey k);               // conv.hip: the switch's value, nullptr = unset
#ifdef RYOLO_MP_ABLATION
inline const char *abl_env(const char *name) { return getenv(name); }
#else
inline const char *abl_env(const char *) { return nullptr; }
#endif

}  // namespace ryolo_detail
