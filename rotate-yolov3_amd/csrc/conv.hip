// rotate-yolov3_amd/csrc/conv.hip -- NHWC bf16 implicit-GEMM convolution on MFMA for gfx950 (MI355X), with the
// Darknet block epilogue fused:  y = [upsample2x]( act( conv(x, W) * scale[c] + shift[c] ) [+ residual] ).
//
// Replaces, for the Darknet-53 stack, the reference's per-layer operator chain
//   nn.Conv2d -> nn.BatchNorm2d(eval) -> nn.PReLU            (model/models.py:49-66)
//   x + layer_outputs[from]                                   (shortcut, model/models.py:281-282)
//   nn.Upsample(nearest, x2) + torch.cat(.., 1)               (model/models.py:93-94, :269-278)
// which the reference dispatches to cuDNN / ATen as separate kernels.  scale/shift are the folded eval-mode
// BatchNorm (utils/torch_utils.py:45-69 computes the same fold into the weights; here it stays in fp32 and is
// applied to the fp32 accumulator), or (1, bias) for the three head convs.  Channel-sliced input / output /
// residual tensors (pixel stride != channel count) let route/concat layers be written in place, no copy.
//
// GEMM view:  D[c_out][pixel] = sum_k Wp[c_out][k] * X[pixel][k],  k = (kh*KS + kw)*C_in + c,  pixel = (img, ho, wo).
//   - MFMA 16x16x32 bf16, weights as the A operand so that each lane ends up with 4 CONSECUTIVE output channels
//     of one pixel (NHWC-contiguous) in its accumulator fragment.
//   - Tiles BM pixels x BN channels x BK=64; 4 waves; both operands staged HBM->LDS with 16-B direct-to-LDS loads
//     (global_load_lds_dwordx4), double buffered, one barrier per K step.  The LDS image of a tile is
//     [row][8 x 16-B slots] with slot ^= (row>>1)&7 so the ds_read_b128 fragment reads are bank-conflict free;
//     direct-to-LDS writes are lane-linear, so the XOR is applied to the per-lane SOURCE address (and again on
//     the read).  Zero padding, M/K tails: the lane's source pointer is redirected to a zero page.
//   - Epilogue: scale/shift/activation in fp32 on the accumulators -> bf16 -> LDS staging tile -> coalesced
//     16-B rows: + residual (fp32 add, one more bf16 rounding, i.e. the unfused layer-by-layer result) -> store
//     (optionally replicated 2x2 for the fused nearest upsample).
//   - Workgroup -> tile map is XCD-aware: the 8 XCDs get contiguous chunks of the tile list, channel tiles
//     fastest, so the blocks that share an activation tile run on one XCD's L2 back to back.
#include <cstring>

#include <mutex>

#include "conv_common.h"

namespace ryolo_detail {
thread_local int *g_conv_choice = nullptr;      // see conv_common.h
}

using namespace ryolo_detail;

namespace {

// GEN = false: inference instantiation (no BatchNorm-statistics epilogue, dense output placement);
// GEN = true : training instantiation (statistics partials, strided output placement for the stride-2 dgrad classes).
// BNRED (training backward, stride-1 data gradients that are the LAST writer of a BatchNorm block's output gradient): epilogue 2 --
// where a thread holds 8 consecutive channels of a pixel exactly as they are stored -- also runs the first pass of that block's
// BatchNorm/activation backward on them (see the persistent kernel below for the arithmetic), one extra 16-B read of z per chunk.
// A thread keeps the same 8 channels for all its chunks; its 24 sums are combined over the workgroup in a fixed order and stored
// (not added) into row m_tile of part[m_tiles][3][C]: every (row, channel) has exactly one writer.
template <int KS, int BM, int BN, int WGM, int WGN, int NSTAGE, bool FAST, bool GEN, bool BNRED = false>
__global__ void __launch_bounds__(WGM *WGN * 64) conv_igemm_kernel(const ConvParams p, const BnRed br = BnRed()) {
    constexpr int NW = WGM * WGN, NT = NW * 64;
    constexpr int WPIX = BM / WGM, WCH = BN / WGN, PF = WPIX / 16, CF = WCH / 16;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int A_PPW = (BM / 8) / NW, B_PPW = (BN / 8) / NW;   // 1-KiB pieces (8 tile rows) per wave
    static_assert(A_PPW >= 1 && B_PPW >= 1, "tile too small for the wave count");
    constexpr int SROW = BN * 2 + 16;                              // epilogue staging row pitch (bytes)
    static_assert(BM * SROW + WGM * 2 * BN * 4 <= NSTAGE * STAGE, "staging tile + statistics slots must fit in the operand buffers");
    constexpr int LOADS_PER_STAGE = A_PPW + B_PPW;   // direct-to-LDS instructions one wave issues per K step

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- XCD-aware tile assignment (bijective for any grid size)
    int m_tile, n_tile;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
        const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        n_tile = id % p.nt;
        m_tile = id / p.nt;
    }
    const int m0 = m_tile * BM, n0 = n_tile * BN;

    const __bf16 *zero_page = p.w + (size_t)(((p.Cout + 127) >> 7) << 7) * p.Kpad;
    const long long zero_off = zero_page - p.x;   // element distance; both bf16 arrays (any two device pointers)

    // ---- BNRED: the consumer block's z for the chunks this thread stores in epilogue 2, and its BatchNorm constants, are requested
    // FIRST -- z was written a whole forward pass ago (an HBM miss, unlike the running gradient the epilogue also reads), and loaded in
    // the epilogue its latency was exposed once per tile (+16 % on the 128 x 128 data gradients); here it hides under the K loop.
    constexpr int BR_CPR = BN / 8, BR_NIT = BM * BR_CPR / NT;
    bf16x8 zv[BNRED ? BR_NIT : 1];
    if constexpr (BNRED) {
        const int c = n0 + (tid % BR_CPR) * 8;
#pragma unroll
        for (int it = 0; it < BR_NIT; it++) {
            const int m = m0 + (it * NT + tid) / BR_CPR;
            zv[it] = *(const bf16x8 *)((m < p.M && c < p.Cout) ? br.z + (size_t)m * br.z_cs + c : zero_page);
        }
    }

    // ---- per-lane staging bookkeeping.  Lane l of a wave fills 16-B slot (l & 7) of tile row 8*piece + (l >> 3).
    long long a_base[A_PPW];   // element offset of (img, hi0, wi0, c=0); may be negative (padding)
    int a_hi0[A_PPW], a_wi0[A_PPW], a_slot[A_PPW];
#pragma unroll
    for (int j = 0; j < A_PPW; j++) {
        const int piece = wave * A_PPW + j;
        const int row = piece * 8 + (lane >> 3);
        a_slot[j] = (lane & 7) ^ ((row >> 1) & 7);
        const int m = m0 + row;
        if (m < p.M) {
            int wo, ho, img;
            split_pixel(m, p.Wo, p.Ho, p.magic_wo, p.magic_ho, p.use_magic, wo, ho, img);
            a_hi0[j] = ho * p.stride - p.pad;
            a_wi0[j] = wo * p.stride - p.pad;
            a_base[j] = (((long long)img * p.H + a_hi0[j]) * p.W + a_wi0[j]) * p.in_cs;
        } else {
            a_hi0[j] = -(1 << 28);   // always out of bounds -> zero page
            a_wi0[j] = -(1 << 28);
            a_base[j] = 0;
        }
    }
    const __bf16 *b_ptr[B_PPW];
#pragma unroll
    for (int j = 0; j < B_PPW; j++) {
        const int piece = wave * B_PPW + j;
        const int row = piece * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ ((row >> 1) & 7);
        b_ptr[j] = p.w + (size_t)(n0 + row) * p.Kpad + slot * 8;
    }

    // FAST path (Cin % 64 == 0, tensors < 2 GiB): a whole K step lies inside one filter tap, so the tap offset is a
    // SCALAR and the per-lane work per 1-KiB piece is one 32-bit add and one select.  Loads are buffer loads with a
    // per-lane byte offset: an invalid (padding / M-tail) lane gets an out-of-range offset and the hardware
    // writes zeros to LDS -- no zero page, no 64-bit address arithmetic, no branches in the K loop.
    int a_off32[A_PPW];          // byte offset of (img, hi0, wi0, slot*8); wraps for border pixels, only used when valid
    unsigned a_mask[A_PPW];      // bit t: tap t of the tap list reads inside the image for this lane's pixel
    int b_off32[B_PPW];
    if constexpr (FAST) {
#pragma unroll
        for (int j = 0; j < A_PPW; j++) {
            // taps2 (C_in == 32, one K step = two taps): the slot's low 2 bits pick the channels, bit 2 picks the tap
            a_off32[j] = (int)(a_base[j] * 2) + (p.taps2 ? (a_slot[j] & 3) : a_slot[j]) * 16;
            unsigned mk = 0;
            if constexpr (GEN) {          // arbitrary tap list (dgrad parity classes)
#pragma unroll
                for (int t = 0; t < 9; t++) {     // branch-free: 72 scalar branches per tile cost as much as a short K loop
                    const int hi = a_hi0[j] + p.tap_dy[t], wi = a_wi0[j] + p.tap_dx[t];
                    const unsigned in = (unsigned)(t < p.ntaps) & (unsigned)((unsigned)hi < (unsigned)p.H) & (unsigned)((unsigned)wi < (unsigned)p.W);
                    mk |= in << t;
                }
            } else {                      // the regular KS x KS window
#pragma unroll
                for (int t = 0; t < KS * KS; t++) {
                    const int hi = a_hi0[j] + t / KS, wi = a_wi0[j] + t % KS;
                    if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) mk |= 1u << t;
                }
            }
            a_mask[j] = mk;
        }
#pragma unroll
        for (int j = 0; j < B_PPW; j++) b_off32[j] = (int)((b_ptr[j] - p.w) * 2);
    }
    // running (scalar) position of the K loop for the FAST path: tap index, channel offset inside the tap
    int f_tap = 0, f_c0 = 0;
    // lane t keeps the byte offset of tap t; the K loop fetches the current one with v_readlane (no memory access)
    int lane_tapoff = 0;
    int f_kh = 0, f_kw = 0;       // GEN = false: plain scalar counters over the KS x KS window
    if constexpr (FAST && GEN) {
        const int t = lane < p.ntaps ? lane : 0;
        lane_tapoff = ((p.tap_dy[t] * p.W + p.tap_dx[t]) * p.in_cs) * 2;
    }

    auto stage_fast = [&](int kt, int buf) {
        char *abuf = smem + buf * STAGE;
        char *bbuf = abuf + A_BYTES;
        int tapoff;                                                                   // scalar
        if constexpr (GEN) tapoff = __builtin_amdgcn_readlane(lane_tapoff, f_tap) + f_c0 * 2;
        else tapoff = ((f_kh * p.W + f_kw) * p.in_cs + f_c0) * 2;
        if (p.taps2) {
            // C_in == 32 (regular KS x KS window only): K step kt = taps 2kt and 2kt+1 (tap 9 = K padding, its mask bit is 0);
            // two scalar tap offsets, each lane picks by bit 2 of its logical slot
            const int t0 = 2 * kt, t1 = 2 * kt + 1;
            int off0, off1;
            if constexpr (GEN) {       // the training instantiation keeps its window as a tap list (lane t: tap t's offset)
                off0 = __builtin_amdgcn_readlane(lane_tapoff, t0);
                off1 = __builtin_amdgcn_readlane(lane_tapoff, t1);
            } else {
                const int kh0 = (t0 * 11) >> 5, kw0 = t0 - 3 * kh0, kh1 = (t1 * 11) >> 5, kw1 = t1 - 3 * kh1;
                off0 = ((kh0 * p.W + kw0) * p.in_cs) * 2;
                off1 = ((kh1 * p.W + kw1) * p.in_cs) * 2;
            }
#pragma unroll
            for (int j = 0; j < A_PPW; j++) {
                const int hi = a_slot[j] >> 2;
                const bool ok = (a_mask[j] >> (t0 + hi)) & 1u;
                const int voff = ok ? a_off32[j] + (hi ? off1 : off0) : (int)0x80000000;
                buffer_load_lds16(p.x, p.x_bytes, abuf + (wave * A_PPW + j) * 1024, voff, 0);
            }
#pragma unroll
            for (int j = 0; j < B_PPW; j++)
                buffer_load_lds16(p.w, p.w_bytes, bbuf + (wave * B_PPW + j) * 1024, b_off32[j], kt * (BK * 2));
            return;
        }
#pragma unroll
        for (int j = 0; j < A_PPW; j++) {
            const bool ok = (a_mask[j] >> f_tap) & 1u;
            const int voff = ok ? a_off32[j] + tapoff : (int)0x80000000;
            buffer_load_lds16(p.x, p.x_bytes, abuf + (wave * A_PPW + j) * 1024, voff, 0);
        }
#pragma unroll
        for (int j = 0; j < B_PPW; j++)
            buffer_load_lds16(p.w, p.w_bytes, bbuf + (wave * B_PPW + j) * 1024, b_off32[j], kt * (BK * 2));
        // advance the scalar K position (stage() is called with consecutive kt)
        f_c0 += BK;
        if (f_c0 >= p.Cin) {
            f_c0 = 0;
            f_tap++;
            if constexpr (!GEN) {
                if (++f_kw == KS) { f_kw = 0; f_kh++; }
            }
        }
    };

    auto stage_slow = [&](int kt, int buf) {
        char *abuf = smem + buf * STAGE;
        char *bbuf = abuf + A_BYTES;
#pragma unroll
        for (int j = 0; j < A_PPW; j++) {
            const int k = kt * BK + a_slot[j] * 8;
            long long off;
            bool ok;
            if (KS == 1) {
                ok = (a_hi0[j] >= 0) && (k < p.K);
                off = a_base[j] + k;
            } else {
                int tap, c;
                if (p.cin_log2 >= 0) {            // power-of-two Cin: shifts
                    tap = k >> p.cin_log2;
                    c = k & (p.Cin - 1);
                } else {                          // Cin % 64 == 0: a whole K step lies inside one tap (uniform divide)
                    tap = (kt * BK) / p.Cin;
                    c = k - tap * p.Cin;
                }
                const int kh = (tap * 11) >> 5, kw = tap - 3 * kh;
                const int hi = a_hi0[j] + kh, wi = a_wi0[j] + kw;
                ok = (k < p.K) && ((unsigned)hi < (unsigned)p.H) && ((unsigned)wi < (unsigned)p.W);
                off = a_base[j] + (long long)(kh * p.W + kw) * p.in_cs + c;
            }
            const __bf16 *src = p.x + (ok ? off : zero_off);   // select, no branch
            __builtin_amdgcn_global_load_lds((glb_vp)src, (lds_vp)(abuf + (wave * A_PPW + j) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_PPW; j++) {
            __builtin_amdgcn_global_load_lds((glb_vp)(b_ptr[j] + kt * BK),
                                             (lds_vp)(bbuf + (wave * B_PPW + j) * 1024), 16, 0, 0);
        }
    };

    auto stage = [&](int kt, int buf) {
        if constexpr (FAST) stage_fast(kt, buf);
        else stage_slow(kt, buf);
    };

    // ---- fragment read offsets (bytes within a tile image)
    const int wm = wave / WGN, wn = wave % WGN;
    const int frow = lane & 15, fk = lane >> 4;
    int a_off[PF][2], b_off[CF][2];
#pragma unroll
    for (int f = 0; f < PF; f++) {
        const int row = wm * WPIX + f * 16 + frow;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) a_off[f][ks] = row * 128 + (((ks * 4 + fk) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int f = 0; f < CF; f++) {
        const int row = wn * WCH + f * 16 + frow;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) b_off[f][ks] = row * 128 + (((ks * 4 + fk) ^ ((row >> 1) & 7)) << 4);
    }

    f32x4 acc[CF][PF];
#pragma unroll
    for (int c = 0; c < CF; c++)
#pragma unroll
        for (int f = 0; f < PF; f++) acc[c][f] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int KT = p.Kpad / BK;
    auto compute = [&](int buf) {
        const char *abuf = smem + buf * STAGE;
        const char *bbuf = abuf + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            bf16x8 wf[CF], xf[PF];
#pragma unroll
            for (int c = 0; c < CF; c++) wf[c] = *(const bf16x8 *)(bbuf + b_off[c][ks]);
#pragma unroll
            for (int f = 0; f < PF; f++) xf[f] = *(const bf16x8 *)(abuf + a_off[f][ks]);
#pragma unroll
            for (int c = 0; c < CF; c++)
#pragma unroll
                for (int f = 0; f < PF; f++)
                    acc[c][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[c], xf[f], acc[c][f], 0, 0, 0);
        }
    };
    if constexpr (NSTAGE == 2) {
        stage(0, 0);
        for (int kt = 0; kt + 1 < KT; kt++) {
            __syncthreads();   // drains this wave's direct-to-LDS loads (vmcnt(0)) and orders all waves: tile kt landed,
                               // and nobody still reads the buffer that the next stage() overwrites
            stage(kt + 1, (kt + 1) & 1);
            compute(kt & 1);
        }
        __syncthreads();
        compute((KT - 1) & 1);
    } else {
        // NSTAGE-deep ring with COUNTED waits: tiles kt+1 .. kt+NSTAGE-2 stay in flight across the barrier, so the
        // HBM/L2 latency of a tile (~2000 cycles under load, longer than one K step of MFMA work) is covered by
        // NSTAGE-1 steps of compute.  hipcc's __syncthreads() would drain vmcnt to 0 while direct-to-LDS loads are
        // pending, hence the raw s_barrier + explicit s_waitcnt (cdna_hip_programming.md, "Pipelining across barriers").
#pragma unroll
        for (int t = 0; t < NSTAGE - 1; t++)
            if (t < KT) stage(t, t);
        int buf = 0;
        for (int kt = 0; kt < KT; kt++) {
            // tiles kt .. min(kt+NSTAGE-2, KT-1) have been issued; wait until only the younger ones are outstanding
            const int younger = min(NSTAGE - 2, KT - 1 - kt);
            if (younger >= NSTAGE - 2) {
                if constexpr (NSTAGE == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS_PER_STAGE) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS_PER_STAGE) : "memory");
            } else if (NSTAGE == 4 && younger == 1) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS_PER_STAGE) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();   // every wave's share of tile kt has landed; tile kt-1's buffer is free
            if (kt + NSTAGE - 1 < KT) {
                int nb = buf + NSTAGE - 1;
                if (nb >= NSTAGE) nb -= NSTAGE;
                stage(kt + NSTAGE - 1, nb);
            }
            compute(buf);
            buf = (buf + 1 == NSTAGE) ? 0 : buf + 1;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }

    // ---- epilogue 1: scale/shift/activation on the fp32 accumulators -> bf16 -> LDS staging tile
    float st_sum[CF][4], st_sq[CF][4];
#pragma unroll
    for (int c = 0; c < CF; c++)
#pragma unroll
        for (int r = 0; r < 4; r++) st_sum[c][r] = st_sq[c][r] = 0.f;
    __syncthreads();
    auto epilogue1 = [&](auto actfn) {
#pragma unroll
        for (int c = 0; c < CF; c++) {
            const int ch_local = wn * WCH + c * 16 + fk * 4;
            const f32x4 sc = *(const f32x4 *)(p.scale + n0 + ch_local);
            const f32x4 sh = *(const f32x4 *)(p.shift + n0 + ch_local);
#pragma unroll
            for (int f = 0; f < PF; f++) {
                const int pix_local = wm * WPIX + f * 16 + frow;
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; r++) o[r] = (__bf16)actfn(acc[c][f][r] * sc[r] + sh[r]);
                *(bf16x4 *)(smem + pix_local * SROW + ch_local * 2) = o;
                if (GEN && p.stat_part) {   // statistics of the values as stored (bf16); rows past M hold exact zeros
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float v = (float)o[r];
                        st_sum[c][r] += v;
                        st_sq[c][r] += v * v;
                    }
                }
            }
        }
    };
    const float slope = p.slope;
    if (p.act == RYOLO_ACT_LEAKY) epilogue1([slope](float v) { return v > 0.f ? v : v * slope; });
    else if (p.act == RYOLO_ACT_MISH) epilogue1([](float v) { return mish(v); });
    else epilogue1([](float v) { return v; });
    if (GEN && p.stat_part) {
        // the 16 lanes of a k-group hold the same channels: DPP row sum over them leaves every total in all 16 lanes; lane
        // frow then keeps total number frow (channel fragment frow / 4, register frow % 4) and parks it in this wave row's
        // slot behind the staging tile
        static_assert(CF <= 4, "one total per lane of a 16-lane row");
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int c = 0; c < CF; c++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float a = row16_sum(st_sum[c][r]), b = row16_sum(st_sq[c][r]);
                if (frow == c * 4 + r) {
                    ta = a;
                    tb = b;
                }
            }
        float *slot = (float *)(smem + BM * SROW) + wm * 2 * BN;
        const int ch = wn * WCH + (frow >> 2) * 16 + fk * 4 + (frow & 3);
        if (frow < CF * 4) {
            slot[ch] = ta;
            slot[BN + ch] = tb;
        }
    }
    __syncthreads();
    if (GEN && p.stat_part) {
        // the wave rows' totals are added in a fixed order (the fp32 sums stay reproducible), then ONE 64-bit atomic per
        // channel and statistic per workgroup into partial row (m_tile mod STAT_ROWS)
        static_assert(NT >= 2 * BN, "one thread per (statistic, channel)");
        if (tid < 2 * BN) {
            const float *slot = (const float *)(smem + BM * SROW);
            float v = slot[tid];
#pragma unroll
            for (int w = 1; w < WGM; w++) v += slot[w * 2 * BN + tid];
            const int st = tid / BN, ch = tid % BN;
            if (n0 + ch < p.Cout)
                atomicAdd(p.stat_part + (size_t)(m_tile % STAT_ROWS) * 2 * p.stat_cpad + (size_t)st * p.stat_cpad + n0 + ch, (double)v);
        }
    }

    // ---- epilogue 2: coalesced 16-B rows: (+ residual) -> global (optionally 2x2 replicated).
    // All residual loads of a thread are issued before any is consumed (NIT independent 16-B loads in flight).
    constexpr int CPR = BN / 8;              // 16-B chunks per staged row
    constexpr int NIT = BM * CPR / NT;       // chunks per thread
    static_assert(BM * CPR % NT == 0, "tile chunks must divide evenly over the threads");
    auto opix = [&](int m) -> size_t {
        int j, i, img;
        split_pixel(m, p.Wo, p.Ho, p.magic_wo, p.magic_ho, p.use_magic, j, i, img);
        return ((size_t)img * p.OH + (size_t)(i * p.os + p.ooy)) * p.OW + (size_t)(j * p.osx + p.oox);
    };
    bf16x8 rv[NIT];
    if (p.res) {
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int idx = it * NT + tid;
            const int m = m0 + idx / CPR, c = n0 + (idx % CPR) * 8;
            const bool ok = (m < p.M) && (c < p.Cout);
            rv[it] = *(const bf16x8 *)(ok ? p.res + ((!GEN || p.os == 1) ? (size_t)m : opix(m)) * p.res_cs + c : zero_page);
        }
    }
    static_assert(!BNRED || (NT % CPR == 0 && CPR <= 16 && !GEN && CPR == BR_CPR && NIT == BR_NIT), "a thread keeps one 8-channel chunk for all its pixels");
    float bn_sc[8], bn_sh[8], bn_mu[8], bn_is[8], bs1[8], bs2[8], bs3[8];      // (the BatchNorm constants of the thread's 8 channels: L2 hits, not held over the K loop)
    float bn_slope = 0.f;
    if constexpr (BNRED) {
        bn_slope = br.slope[0];
        const int c = n0 + (tid % CPR) * 8;
        const bool okc = c < p.Cout;                    // whole 8-channel chunks (C % 8 == 0)
        const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 a0 = okc ? *(const f32x4 *)(br.scale + c) : z4, a1 = okc ? *(const f32x4 *)(br.scale + c + 4) : z4;
        const f32x4 b0 = okc ? *(const f32x4 *)(br.shift + c) : z4, b1 = okc ? *(const f32x4 *)(br.shift + c + 4) : z4;
        const f32x4 u0 = okc ? *(const f32x4 *)(br.mean + c) : z4, u1 = okc ? *(const f32x4 *)(br.mean + c + 4) : z4;
        const f32x4 i0 = okc ? *(const f32x4 *)(br.invstd + c) : z4, i1 = okc ? *(const f32x4 *)(br.invstd + c + 4) : z4;   // (for the flush: its latency is not exposed there)
#pragma unroll
        for (int e = 0; e < 4; e++) {
            bn_sc[e] = a0[e]; bn_sc[4 + e] = a1[e]; bn_sh[e] = b0[e]; bn_sh[4 + e] = b1[e]; bn_mu[e] = u0[e]; bn_mu[4 + e] = u1[e];
            bn_is[e] = i0[e]; bn_is[4 + e] = i1[e];
        }
#pragma unroll
        for (int e = 0; e < 8; e++) bs1[e] = bs2[e] = bs3[e] = 0.f;
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int idx = it * NT + tid;
        const int pix = idx / CPR, ch = (idx % CPR) * 8;
        const int m = m0 + pix, c = n0 + ch;
        if (m >= p.M || c >= p.Cout) continue;
        bf16x8 v = *(const bf16x8 *)(smem + pix * SROW + ch * 2);
        if (p.res) {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = (__bf16)((float)v[e] + (float)rv[it][e]);
        }
        if constexpr (BNRED) {       // the arithmetic of bn_act_bwd_reduce_kernel<1> (train.hip) on the value as it is stored (bf16)
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float zf = (float)zv[it][e], d = (float)v[e];
                const float u = zf * bn_sc[e] + bn_sh[e];
                float g = d;
                if (u <= 0.f) { g = d * bn_slope; bs3[e] += d * u; }
                bs2[e] += g * (zf - bn_mu[e]);
                bs1[e] += g;
            }
        }
        if (p.ups == 1 && (!GEN || p.os == 1)) {
            if (p.nt_out) __builtin_nontemporal_store(v, (bf16x8 *)(p.y + (size_t)m * p.out_cs + c));
            else *(bf16x8 *)(p.y + (size_t)m * p.out_cs + c) = v;
        } else if (p.ups == 1) {     // strided placement (stride-2 dgrad parity classes)
            if (p.nt_out) __builtin_nontemporal_store(v, (bf16x8 *)(p.y + opix(m) * p.out_cs + c));
            else *(bf16x8 *)(p.y + opix(m) * p.out_cs + c) = v;
        } else {
            int wo, ho, img;
            split_pixel(m, p.Wo, p.Ho, p.magic_wo, p.magic_ho, p.use_magic, wo, ho, img);
            const size_t W2 = (size_t)p.Wo * 2;
            const size_t o00 = (((size_t)img * p.Ho * 2 + ho * 2) * W2 + wo * 2) * p.out_cs + c;
            *(bf16x8 *)(p.y + o00) = v;
            *(bf16x8 *)(p.y + o00 + p.out_cs) = v;
            *(bf16x8 *)(p.y + o00 + W2 * p.out_cs) = v;
            *(bf16x8 *)(p.y + o00 + (W2 + 1) * p.out_cs) = v;
        }
    }
    if constexpr (BNRED) {
        // lanes cch, cch + CPR, ... of a wave hold the same channels: xor-butterfly pair sums (fixed order), then the waves' totals
        // through the slots behind the staging tile (other threads may still be reading the tile itself), summed in wave order
        const int cch = tid % CPR;
        static_assert(BM * SROW + NW * CPR * 24 * 4 <= NSTAGE * STAGE, "staging tile + reduce slots must fit in the operand buffers");
        float *slots = (float *)(smem + BM * SROW);                // [NW][CPR][24] (extra LDS would cost the second workgroup per CU)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float a = bs1[e], b = bs2[e] * bn_is[e], d = bs3[e];
            a = lane_class_sum<CPR>(a); b = lane_class_sum<CPR>(b); d = lane_class_sum<CPR>(d);
            if (lane < CPR) {
                slots[(wave * CPR + cch) * 24 + e] = a;
                slots[(wave * CPR + cch) * 24 + 8 + e] = b;
                slots[(wave * CPR + cch) * 24 + 16 + e] = d;
            }
        }
        __syncthreads();
        for (int t = tid; t < CPR * 24; t += NT) {
            const int l = t / 24, k = t % 24;
            float v = slots[l * 24 + k];
#pragma unroll
            for (int w = 1; w < NW; w++) v += slots[(w * CPR + l) * 24 + k];
            const int ch = n0 + l * 8 + (k & 7);
            if (ch < p.Cout) br.part[((size_t)m_tile * 3 + (k >> 3)) * p.Cout + ch] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ persistent variant
// The FAST inference kernel as a PERSISTENT grid: two workgroups per CU walk a tile list (XCD-contiguous chunks), and
// the per-tile fixed cost is taken off the critical path --
//   * during the LAST K step of a tile the workgroup computes the next tile's lane bookkeeping and issues that tile's
//     first K step into the idle LDS stage buffer, so the loads fly while the epilogue runs;
//   * the residual rows of the current tile are requested before that prefetch (vector-memory results return in
//     order: the epilogue then waits for the residual only, not for the prefetch);
//   * the epilogue stages the bf16 tile in the stage buffer the last K step read (chunk-XOR swizzle instead of padding
//     so that it fits) and uses raw s_barrier + lgkmcnt waits, which leave the prefetch in flight.
// Pixel decomposition uses multiply-high by ceil(2^32/d) (exact while M*d < 2^32; checked by the host).
// STATS (training forward): every lane keeps the per-channel sums of z and z*z of the values it stored in registers
// over ALL the tiles the workgroup walks; the cross-lane sum, the LDS combine of the wave rows and the 64-bit atomics
// into partial row (workgroup mod STAT_ROWS) run once at the end (or when the channel tile changes), not per tile.
// BNRED (training backward, 1x1 data gradient): the tile this kernel stores is the FINAL gradient dy of the block whose output the
// 1x1 conv consumed (dx, after the accumulation into the shortcut chain's running gradient), so the first pass of that block's
// BatchNorm/activation backward -- per-channel sums of g = dy*act'(u), g*(z - mean)*invstd and dy*min(u, 0) over all pixels --
// runs here on the values in registers, for one extra read of z, instead of as a pass of its own over z AND dy.  Each lane owns
// the same 8 channels for every tile of a channel column and keeps the 24 sums in registers over ALL the tiles the workgroup
// walks; they are combined once per workgroup (fixed order) into row blockIdx.x of part[grid][3][C], which
// bn_act_bwd_finalize_kernel (train.hip) reduces exactly like the slab rows of the stand-alone pass.
template <int KS, int BM, int BN, int WGM, int WGN, bool STATS, bool BNRED = false>
__global__ void __launch_bounds__(WGM *WGN * 64, 2) conv_igemm_persist_kernel(const ConvParams p, const BnRed br = BnRed()) {
    constexpr int NW = WGM * WGN, NT = NW * 64;
    constexpr int WPIX = BM / WGM, WCH = BN / WGN, PF = WPIX / 16, CF = WCH / 16;
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int A_PPW = (BM / 8) / NW, B_PPW = (BN / 8) / NW;
    constexpr int CPR = BN / 8, NIT = BM * CPR / NT, SROW = BN * 2;
    static_assert(A_PPW >= 1 && B_PPW >= 1, "tile too small for the wave count");
    static_assert(BM * SROW <= STAGE, "the staging tile must fit in one stage buffer");
    static_assert(BM * CPR % NT == 0, "tile chunks must divide evenly over the threads");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // this workgroup's tile list: XCD x (= blockIdx & 7) owns the x-th contiguous chunk of tile ids
    const int T = p.ntiles, G = gridDim.x;
    const int q = T >> 3, r = T & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, nloc = G >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int len = q + (xcd < r ? 1 : 0);
    if constexpr (BNRED) {       // this workgroup's row of partial sums starts at zero (flush_bn adds: a row may be flushed per channel column)
        for (int t = tid; t < 3 * p.Cout; t += NT) br.part[(size_t)blockIdx.x * 3 * p.Cout + t] = 0.f;
    }
    if (loc >= len) return;

    int a_off32[A_PPW], b_off32[B_PPW];
    unsigned a_mask[A_PPW];
    unsigned a_tapsel = 0;       // taps2: bit j = which of the K step's two taps piece j of this lane stages
    int f_tap = 0, f_c0 = 0, f_kh = 0, f_kw = 0;
    int nm0 = 0, nn0 = 0;
    auto setup = [&](int id) {
        const int m_tile = udiv_magic(id, p.magic_nt);
        const int n_tile = id - m_tile * p.nt;
        nm0 = m_tile * BM;
        nn0 = n_tile * BN;
#pragma unroll
        for (int j = 0; j < A_PPW; j++) {
            const int row = (wave * A_PPW + j) * 8 + (lane >> 3);
            const int slot = (lane & 7) ^ ((row >> 1) & 7);
            const int m = nm0 + row;
            const int t = udiv_magic(m, p.magic_wo);
            const int wo = m - t * p.Wo;
            const int img = udiv_magic(t, p.magic_ho);
            const int ho = t - img * p.Ho;
            const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
            // taps2 (3x3, C_in == 32: one K step = two taps): the slot's low 2 bits pick the channels, bit 2 picks the tap
            a_off32[j] = (((img * p.H + hi0) * p.W + wi0) * p.in_cs) * 2 + ((KS == 3 && p.taps2) ? (slot & 3) : slot) * 16;
            if (KS == 3) a_tapsel = (a_tapsel & ~(1u << j)) | ((unsigned)(slot >> 2) << j);
            unsigned rb = 0, cb = 0;
#pragma unroll
            for (int k = 0; k < KS; k++) {
                rb |= ((unsigned)(hi0 + k) < (unsigned)p.H ? 1u : 0u) << k;
                cb |= ((unsigned)(wi0 + k) < (unsigned)p.W ? 1u : 0u) << k;
            }
            unsigned mk = 0;
#pragma unroll
            for (int k = 0; k < KS; k++) mk |= ((rb >> k) & 1u) ? (cb << (k * KS)) : 0u;
            a_mask[j] = m < p.M ? mk : 0u;
        }
#pragma unroll
        for (int j = 0; j < B_PPW; j++) {
            const int row = (wave * B_PPW + j) * 8 + (lane >> 3);
            const int slot = (lane & 7) ^ ((row >> 1) & 7);
            b_off32[j] = ((nn0 + row) * p.Kpad + slot * 8) * 2;
        }
        f_tap = f_c0 = f_kh = f_kw = 0;
    };

    // live = false: the same instructions with every lane out of range (zeros to LDS, no memory traffic) -- keeps the
    // instruction stream, and with it the compiler's vmcnt bookkeeping, identical whether or not a next tile exists
    auto stage = [&](int kt, int buf, bool live) {
        char *abuf = smem + buf * STAGE;
        char *bbuf = abuf + A_BYTES;
        if (KS == 3 && p.taps2) {
            // C_in == 32: K step kt = taps 2kt and 2kt+1 of the 3x3 window (tap 9 = K padding: its mask bit is 0)
            const int t0 = 2 * kt, t1 = 2 * kt + 1;
            const int kh0 = (t0 * 11) >> 5, kw0 = t0 - 3 * kh0, kh1 = (t1 * 11) >> 5, kw1 = t1 - 3 * kh1;
            const int off0 = ((kh0 * p.W + kw0) * p.in_cs) * 2, off1 = ((kh1 * p.W + kw1) * p.in_cs) * 2;
#pragma unroll
            for (int j = 0; j < A_PPW; j++) {
                const unsigned hi = (a_tapsel >> j) & 1u;
                const bool ok = ((a_mask[j] >> (t0 + hi)) & 1u) && live;
                const int voff = ok ? a_off32[j] + (hi ? off1 : off0) : (int)0x80000000;
                buffer_load_lds16(p.x, p.x_bytes, abuf + (wave * A_PPW + j) * 1024, voff, 0);
            }
#pragma unroll
            for (int j = 0; j < B_PPW; j++)
                buffer_load_lds16(p.w, p.w_bytes, bbuf + (wave * B_PPW + j) * 1024, live ? b_off32[j] : (int)0x80000000,
                                  kt * (BK * 2));
            return;
        }
        const int tapoff = ((f_kh * p.W + f_kw) * p.in_cs + f_c0) * 2;   // scalar
#pragma unroll
        for (int j = 0; j < A_PPW; j++) {
            const bool ok = ((a_mask[j] >> f_tap) & 1u) && live;
            const int voff = ok ? a_off32[j] + tapoff : (int)0x80000000;
            buffer_load_lds16(p.x, p.x_bytes, abuf + (wave * A_PPW + j) * 1024, voff, 0);
        }
#pragma unroll
        for (int j = 0; j < B_PPW; j++)
            buffer_load_lds16(p.w, p.w_bytes, bbuf + (wave * B_PPW + j) * 1024, live ? b_off32[j] : (int)0x80000000,
                              kt * (BK * 2));
        f_c0 += BK;
        if (f_c0 >= p.Cin) {
            f_c0 = 0;
            f_tap++;
            if (++f_kw == KS) { f_kw = 0; f_kh++; }
        }
    };

    const int wm = wave / WGN, wn = wave % WGN;
    const int frow = lane & 15, fk = lane >> 4;
    int a_off[PF][2], b_off[CF][2];
#pragma unroll
    for (int f = 0; f < PF; f++) {
        const int row = wm * WPIX + f * 16 + frow;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) a_off[f][ks] = row * 128 + (((ks * 4 + fk) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int f = 0; f < CF; f++) {
        const int row = wn * WCH + f * 16 + frow;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) b_off[f][ks] = A_BYTES + row * 128 + (((ks * 4 + fk) ^ ((row >> 1) & 7)) << 4);
    }

    const __bf16 *zero_page = p.w + (size_t)(((p.Cout + 127) >> 7) << 7) * p.Kpad;
    const int KT = p.Kpad / BK;
    const float slope = p.slope;
    f32x4 acc[CF][PF];

    int i = loc;
    setup(start + i);
    int m0 = nm0, n0 = nn0;
    int buf = 0;
    stage(0, buf, true);
    // folded-BN scale / shift of this lane's output channels: reloaded only when the channel tile changes (never when
    // the XCD's workgroup count is a multiple of nt, i.e. for every power-of-two nt)
    f32x4 ep_sc[CF], ep_sh[CF];
    auto load_scale_shift = [&]() {
#pragma unroll
        for (int c = 0; c < CF; c++) {
            const int ch_local = wn * WCH + c * 16 + fk * 4;
            ep_sc[c] = *(const f32x4 *)(p.scale + n0 + ch_local);
            ep_sh[c] = *(const f32x4 *)(p.shift + n0 + ch_local);
        }
    };
    load_scale_shift();
    float st_sum[CF][4], st_sq[CF][4];
#pragma unroll
    for (int c = 0; c < CF; c++)
#pragma unroll
        for (int r = 0; r < 4; r++) st_sum[c][r] = st_sq[c][r] = 0.f;
    auto flush_stats = [&]() {    // workgroup-uniform call sites only
        static_assert(!STATS || (CF <= 4 && NT >= 2 * BN), "one total per lane of a 16-lane row; one thread per (statistic, channel)");
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int c = 0; c < CF; c++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float a = row16_sum(st_sum[c][r]), b = row16_sum(st_sq[c][r]);
                if (frow == c * 4 + r) {
                    ta = a;
                    tb = b;
                }
                st_sum[c][r] = st_sq[c][r] = 0.f;
            }
        float *slots = (float *)(smem + 2 * STAGE);            // [WGM][2][BN], behind the two stage buffers
        const int ch = wn * WCH + (frow >> 2) * 16 + fk * 4 + (frow & 3);
        if (frow < CF * 4) {
            slots[wm * 2 * BN + ch] = ta;
            slots[wm * 2 * BN + BN + ch] = tb;
        }
        __syncthreads();
        if (tid < 2 * BN) {
            float v = slots[tid];
#pragma unroll
            for (int w = 1; w < WGM; w++) v += slots[w * 2 * BN + tid];     // fixed order: reproducible fp32 sums
            const int st = tid / BN, c = tid % BN;
            if (n0 + c < p.Cout)
                atomicAdd(p.stat_part + ((size_t)(blockIdx.x % STAT_ROWS) * 2 + st) * p.stat_cpad + n0 + c, (double)v);
        }
        __syncthreads();
    };
    // ---- BNRED state: this lane's 8 channels are n0 + (tid % CPR) * 8 .. + 7 in every tile of the channel column
    static_assert(!BNRED || (NT % CPR == 0 && CPR == 16 && !STATS), "a lane keeps one 8-channel chunk; 4 lanes of a wave share it");
    float bs1[8], bs2[8], bs3[8];                       // the only BNRED state that lives across the K loops (24 registers)
#pragma unroll
    for (int e = 0; e < 8; e++) bs1[e] = bs2[e] = bs3[e] = 0.f;
    const float bn_slope = BNRED ? br.slope[0] : 0.f;
    auto flush_bn = [&]() {        // workgroup-uniform call sites only
        if constexpr (BNRED) {
            const int cch = tid % CPR, c = n0 + cch * 8;
            float *slots = (float *)(smem + 2 * STAGE);        // [NW][CPR][24]
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float a = bs1[e], b = bs2[e] * (c + e < p.Cout ? br.invstd[c + e] : 0.f), d = bs3[e];
                // lanes cch, cch + 16, cch + 32, cch + 48 of a wave hold the same channels: fixed-order pair sums
                a += __shfl_xor(a, 16); b += __shfl_xor(b, 16); d += __shfl_xor(d, 16);
                a += __shfl_xor(a, 32); b += __shfl_xor(b, 32); d += __shfl_xor(d, 32);
                if (lane < CPR) {
                    slots[(wave * CPR + cch) * 24 + e] = a;
                    slots[(wave * CPR + cch) * 24 + 8 + e] = b;
                    slots[(wave * CPR + cch) * 24 + 16 + e] = d;
                }
                bs1[e] = bs2[e] = bs3[e] = 0.f;
            }
            __syncthreads();
            for (int t = tid; t < CPR * 24; t += NT) {
                const int l = t / 24, k = t % 24;
                float v = slots[l * 24 + k];
#pragma unroll
                for (int w = 1; w < NW; w++) v += slots[(w * CPR + l) * 24 + k];      // fixed order
                const int ch = n0 + l * 8 + (k & 7);
                if (ch < p.Cout) br.part[((size_t)blockIdx.x * 3 + (k >> 3)) * p.Cout + ch] += v;   // row owned by this workgroup
            }
            __syncthreads();
        }
    };
    while (true) {
        const int inext = i + nloc;
        const bool has_next = inext < len;
#pragma unroll
        for (int c = 0; c < CF; c++)
#pragma unroll
            for (int f = 0; f < PF; f++) acc[c][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto compute = [&](int cb) {
            const char *sb = smem + cb * STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                bf16x8 wf[CF], xf[PF];
#pragma unroll
                for (int c = 0; c < CF; c++) wf[c] = *(const bf16x8 *)(sb + b_off[c][ks]);
#pragma unroll
                for (int f = 0; f < PF; f++) xf[f] = *(const bf16x8 *)(sb + a_off[f][ks]);
#pragma unroll
                for (int c = 0; c < CF; c++)
#pragma unroll
                    for (int f = 0; f < PF; f++)
                        acc[c][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[c], xf[f], acc[c][f], 0, 0, 0);
            }
        };
        for (int kt = 0; kt + 1 < KT; kt++) {
            // tile step kt landed (explicit vmcnt(0): the compiler's own count does not reliably cover direct-to-LDS
            // loads across the loop back edge); nobody still reads the buffer overwritten next
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            stage(kt + 1, buf ^ 1, true);
            compute(buf);
            buf ^= 1;
        }
        // last K step (peeled): residual request, then the next tile's bookkeeping and first K step, then the MFMAs
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        bf16x8 rv[NIT];
        const bool has_res = !STATS && p.res;      // the statistics instantiation has no residual operand (registers)
        if (has_res) {
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int idx = it * NT + tid;
                const int m = m0 + idx / CPR, c = n0 + (idx % CPR) * 8;
                const bool ok = (m < p.M) && (c < p.Cout);
                rv[it] = *(const bf16x8 *)(ok ? p.res + (size_t)m * p.res_cs + c : zero_page);
            }
        }
        setup(start + (has_next ? inext : i));
        stage(0, buf ^ 1, has_next);
        compute(buf);
        buf ^= 1;
        // the last K step read buffer buf^1: it becomes the staging tile; buffer `buf` receives the next tile's step 0
        char *st = smem + (buf ^ 1) * STAGE;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        auto epilogue1 = [&](auto actfn) {
#pragma unroll
            for (int c = 0; c < CF; c++) {
                const int ch_local = wn * WCH + c * 16 + fk * 4;
                const f32x4 sc = ep_sc[c], sh = ep_sh[c];
#pragma unroll
                for (int f = 0; f < PF; f++) {
                    const int pix_local = wm * WPIX + f * 16 + frow;
                    bf16x4 o;
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) o[rr] = (__bf16)actfn(acc[c][f][rr] * sc[rr] + sh[rr]);
                    *(bf16x4 *)(st + pix_local * SROW + (((ch_local >> 3) ^ (pix_local & (CPR - 1))) << 4) + (fk & 1) * 8) = o;
                    if constexpr (STATS) {     // statistics of the values as stored (bf16); rows past M hold exact zeros
#pragma unroll
                        for (int rr = 0; rr < 4; rr++) {
                            const float q = (float)o[rr];
                            st_sum[c][rr] += q;
                            st_sq[c][rr] += q * q;
                        }
                    }
                }
            }
        };
        if (p.act == RYOLO_ACT_LEAKY) epilogue1([slope](float v) { return v > 0.f ? v : v * slope; });
        else if (p.act == RYOLO_ACT_MISH) epilogue1([](float v) { return mish(v); });
        else epilogue1([](float v) { return v; });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        bf16x8 zv[BNRED ? NIT : 1];
        float bn_sc[8], bn_sh[8], bn_mu[8];
        if constexpr (BNRED) {       // the consumer block's z for this lane's chunks (the accumulators are dead: registers are free)
            {   // and its BatchNorm constants for the lane's 8 channels: re-read per tile (L1/L2 hits) rather than held over the K loop
                const int c = n0 + (tid % CPR) * 8;
                const bool ok = c < p.Cout;             // whole 8-channel chunks (C % 8 == 0)
                const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
                const f32x4 a0 = ok ? *(const f32x4 *)(br.scale + c) : z4, a1 = ok ? *(const f32x4 *)(br.scale + c + 4) : z4;
                const f32x4 b0 = ok ? *(const f32x4 *)(br.shift + c) : z4, b1 = ok ? *(const f32x4 *)(br.shift + c + 4) : z4;
                const f32x4 m0_ = ok ? *(const f32x4 *)(br.mean + c) : z4, m1_ = ok ? *(const f32x4 *)(br.mean + c + 4) : z4;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    bn_sc[e] = a0[e]; bn_sc[4 + e] = a1[e]; bn_sh[e] = b0[e]; bn_sh[4 + e] = b1[e]; bn_mu[e] = m0_[e]; bn_mu[4 + e] = m1_[e];
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int idx = it * NT + tid;
                const int m = m0 + idx / CPR, c = n0 + (idx % CPR) * 8;
                const bool ok = (m < p.M) && (c < p.Cout);
                zv[it] = *(const bf16x8 *)(ok ? br.z + (size_t)m * br.z_cs + c : zero_page);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int idx = it * NT + tid;
            const int pix = idx / CPR, cch = idx % CPR;
            const int m = m0 + pix, c = n0 + cch * 8;
            if (m >= p.M || c >= p.Cout) continue;
            bf16x8 v = *(const bf16x8 *)(st + pix * SROW + ((cch ^ (pix & (CPR - 1))) << 4));
            if (has_res) {
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = (__bf16)((float)v[e] + (float)rv[it][e]);
            }
            if constexpr (BNRED) {   // the arithmetic of bn_act_bwd_reduce_kernel<1> on the value as it is stored (bf16)
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float zf = (float)zv[it][e], d = (float)v[e];
                    const float u = zf * bn_sc[e] + bn_sh[e];
                    float g = d;
                    if (u <= 0.f) { g = d * bn_slope; bs3[e] += d * u; }
                    bs2[e] += g * (zf - bn_mu[e]);
                    bs1[e] += g;
                }
            }
            if (p.ups == 1) {
                if (p.nt_out) __builtin_nontemporal_store(v, (bf16x8 *)(p.y + (size_t)m * p.out_cs + c));
                else *(bf16x8 *)(p.y + (size_t)m * p.out_cs + c) = v;
            } else {
                const int t = udiv_magic(m, p.magic_wo);
                const int wo = m - t * p.Wo;
                const int img = udiv_magic(t, p.magic_ho);
                const int ho = t - img * p.Ho;
                const size_t W2 = (size_t)p.Wo * 2;
                const size_t o00 = (((size_t)img * p.Ho * 2 + ho * 2) * W2 + wo * 2) * p.out_cs + c;
                *(bf16x8 *)(p.y + o00) = v;
                *(bf16x8 *)(p.y + o00 + p.out_cs) = v;
                *(bf16x8 *)(p.y + o00 + W2 * p.out_cs) = v;
                *(bf16x8 *)(p.y + o00 + (W2 + 1) * p.out_cs) = v;
            }
        }
        if (!has_next) break;
        i = inext;
        m0 = nm0;
        if (nn0 != n0) {
            if constexpr (STATS) flush_stats();
            flush_bn();
            n0 = nn0;
            load_scale_shift();
        }
    }
    if constexpr (STATS) flush_stats();
    flush_bn();
}


// ------------------------------------------------------------------------------------------------ first layer, direct
// 3x3 / stride 1 / pad 1 on the 8-channel (3 real) NHWC input -> 32 channels (Darknet-53 layer 0: 11.8 M pixels per bs-32
// batch, 0.95 GB of algorithmic traffic, HBM-bound).  K = 9 taps x 8 channels: the 16 bytes one lane needs for a B
// fragment (8 consecutive k of one pixel) are exactly the 8 channels of ONE tap of ONE input pixel, i.e. one aligned
// 16-B load -- so the fragments are loaded straight from global memory (L1/L2 serve the 9x tap re-reads), no LDS, no
// barrier, and the weights (72 x 32) live in registers for the whole kernel.  A wave walks groups of 16 consecutive output
// pixels: 3 loads, 6 MFMAs (2 channel fragments x 3 k-substeps of 32 = taps 0-3, 4-7, 8 + zeros), 1 store.
// The walk is software-pipelined and branch-free: the next group's three fragments are requested (buffer loads; padding,
// K-padding taps and the M tail are out-of-range offsets = hardware zeros) before the current group's MFMAs, the pixel
// coordinates advance incrementally (one division per wave, W_o >= 16), the activation is a template parameter, and the
// lane regrouping for the 64-B row store is two v_permlane16_swap.  (The first version branched per element on the
// activation, per load on the padding test and waited vmcnt(0) in front of its MFMAs: 2.5 TB/s.)
// Training of layer 0 WITHOUT its conv output (MODE 1..3).  z0 = conv(x) is 4 x the size of x (8 padded channels in, 32 out: 1.5 GB
// against 0.38 GB at bs 64) and costs 27 MACs per value, so it is cheaper to recompute it than to store and re-read it:
//   MODE 1  statistics only (sums of z, z^2 of the bf16-rounded values, nothing stored)         -> ryolo_bn_finalize
//   MODE 0  y = act(z * scale + shift) with the batch statistics folded in = the inference epilogue (no STATS)
//   MODE 2  backward reduce: recompute z, read dy, per-channel sums of g = dy * act'(u), g * (z - mean), dy * min(u, 0)
//   MODE 3  backward apply:  recompute z, read dy, write dz = scale * g + kb * z + kd  (kb, kd from the sums; bn_act_bwd_apply's form)
// Same recomputed bits as the stored z would hold (same MFMAs, same bf16 rounding); per step 4.9 GB less traffic than
// conv -> z, bn_act_fwd(z), bn_act_bwd(z, dy) (tools/step_ab.py).
struct C8Bwd {
    const __bf16 *dy = nullptr; int dy_cs = 0;   // gradient of the block output, NHWC
    const float *mean = nullptr;             // [32]
    const float *kb = nullptr, *kd = nullptr;    // MODE 3: per-channel constants
    const float *slope = nullptr;            // device scalar (a learnable PReLU slope) or nullptr: ConvParams::slope
    double *part = nullptr;                  // MODE 2: [STAT_ROWS][3][32] partial sums (fp64 atomics), zeroed by the caller
    unsigned dy_bytes = 0;
    int round_z = 0;                         // MODE 0 of the training engine: BatchNorm is applied to z ROUNDED to bf16, as if it had been stored
};

template <bool STATS, int ACT, int MODE = 0>   // STATS: also the per-channel sums of z and z^2 (BatchNorm batch statistics), like the GEN epilogue
__global__ void __launch_bounds__(256) conv3x3_c8_direct_kernel(const ConvParams p, int groups_per_wave, const C8Bwd bw = C8Bwd()) {
    const int lane = threadIdx.x & 63;
    const int fr = lane & 15, g = lane >> 4;
    const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long g0 = wave_id * groups_per_wave;
    if (g0 * 16 >= p.M) return;                                   // wave-uniform
    const int n_it = (int)(((long long)p.M - g0 * 16 + 15) / 16 < groups_per_wave ? ((long long)p.M - g0 * 16 + 15) / 16 : groups_per_wave);
    bf16x8 wfr[2][3];
#pragma unroll
    for (int cf = 0; cf < 2; cf++)
#pragma unroll
        for (int ks = 0; ks < 3; ks++)
            wfr[cf][ks] = *(const bf16x8 *)(p.w + (size_t)(cf * 16 + fr) * p.Kpad + ks * 32 + g * 8);
    f32x4 sc[2], sh[2];
#pragma unroll
    for (int cf = 0; cf < 2; cf++) {
        sc[cf] = *(const f32x4 *)(p.scale + cf * 16 + g * 4);
        sh[cf] = *(const f32x4 *)(p.shift + cf * 16 + g * 4);
    }
    const float slope = bw.slope ? bw.slope[0] : p.slope;
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.w_bytes, 0x00020000);   // w_bytes: bytes of y here
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(MODE >= 2 ? bw.dy : p.x), 0, MODE >= 2 ? bw.dy_bytes : 0u, 0x00020000);
#endif
    // backward modes: this lane's per-channel constants (channels cf*16 + 4g .. +3) and sums
    f32x4 mu[2], kbv[2], kdv[2];
    float b1[2][4], b2[2][4], b3[2][4];
    if constexpr (MODE >= 2) {
#pragma unroll
        for (int cf = 0; cf < 2; cf++) {
            mu[cf] = *(const f32x4 *)(bw.mean + cf * 16 + g * 4);
            if constexpr (MODE == 3) {
                kbv[cf] = *(const f32x4 *)(bw.kb + cf * 16 + g * 4);
                kdv[cf] = *(const f32x4 *)(bw.kd + cf * 16 + g * 4);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) b1[cf][r] = b2[cf][r] = b3[cf][r] = 0.f;
        }
    }
    // lane-constant tap geometry of the three k-substeps: tap = 4 ks + g (taps >= 9 are K padding)
    int dkh[3], dkw[3];
    bool tok[3];
#pragma unroll
    for (int ks = 0; ks < 3; ks++) {
        const int tap = ks * 4 + g;
        dkh[ks] = ((tap * 11) >> 5) - 1;
        dkw[ks] = tap - 3 * ((tap * 11) >> 5) - 1;
        tok[ks] = tap < 9;
    }
    float st_sum[2][4], st_sq[2][4];                             // this lane's 8 channels over all its pixels
#pragma unroll
    for (int cf = 0; cf < 2; cf++)
#pragma unroll
        for (int r = 0; r < 4; r++) st_sum[cf][r] = st_sq[cf][r] = 0.f;
    // this lane's pixel of the wave's first group (clamped into range for the coordinate split; the tail is masked by m < M)
    int m = (int)(g0 * 16) + fr;
    int wo, ho, img;
    {
        const int mc = m < p.M ? m : p.M - 1;
        const int t = mc / p.Wo;
        wo = mc - t * p.Wo;
        img = t / p.Ho;
        ho = t - img * p.Ho;
    }
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    typedef __attribute__((ext_vector_type(2))) unsigned int u2;
    auto request_dy = [&](u2(&d)[2], int mcur, bool live) {        // backward modes: dy of this lane's 8 channels of pixel mcur
        if constexpr (MODE >= 2) {
#pragma unroll
            for (int cf = 0; cf < 2; cf++) {
                const int off = (mcur * bw.dy_cs + cf * 16 + g * 4) * 2;
#if defined(__HIP_DEVICE_COMPILE__)
                d[cf] = __builtin_amdgcn_raw_buffer_load_b64(drs, (live && mcur < p.M) ? off : (int)0x80000000, 0, 0);
#endif
            }
        }
    };
    auto request = [&](u4(&x)[3], bool live) {                    // the three fragments of pixel (img, ho, wo)
#pragma unroll
        for (int ks = 0; ks < 3; ks++) {
            const int hi = ho + dkh[ks], wi = wo + dkw[ks];
            const bool ok = live && tok[ks] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const int off = ((img * p.H + hi) * p.W + wi) * 16;
#if defined(__HIP_DEVICE_COMPILE__)
            x[ks] = __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? off : (int)0x80000000, 0, 0);
#endif
        }
    };
    auto advance = [&]() {                                        // +16 pixels (W_o >= 16: at most one row wrap)
        m += 16;
        wo += 16;
        const bool wrap = wo >= p.Wo;
        wo -= wrap ? p.Wo : 0;
        ho += wrap ? 1 : 0;
        const bool wrap2 = ho >= p.Ho;
        ho = wrap2 ? 0 : ho;
        img += wrap2 ? 1 : 0;
    };
    auto finish = [&](const u4(&x)[3], const u2(&dyv)[2], int mcur, bool live) {     // MFMAs + epilogue + store of the group whose fragments are x
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 3; ks++)
#pragma unroll
            for (int cf = 0; cf < 2; cf++)
                acc[cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[cf][ks], __builtin_bit_cast(bf16x8, x[ks]), acc[cf], 0, 0, 0);
        const bool mok = live && mcur < p.M;
        unsigned o2[2][2];
        if constexpr (MODE >= 2) {
            // z as the forward stored it (bf16), u = z * scale + shift, g = dy * act'(u)
#pragma unroll
            for (int cf = 0; cf < 2; cf++) {
                const bf16x4 dq = __builtin_bit_cast(bf16x4, dyv[cf]);
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float zf = (float)(__bf16)acc[cf][r];
                    const float d = (float)dq[r];
                    const float u = zf * sc[cf][r] + sh[cf][r];
                    float gg = d;
                    if constexpr (ACT == RYOLO_ACT_LEAKY) {
                        if (u <= 0.f) {
                            gg = d * slope;
                            if (MODE == 2 && mok) b3[cf][r] += d * u;
                        }
                    } else if constexpr (ACT == RYOLO_ACT_MISH) {
                        const float e = __expf(fminf(u, 20.f)), n1 = (1.f + e) * (1.f + e), t = (n1 - 1.f) / (n1 + 1.f);
                        gg = d * (t + u * (1.f - t * t) * (e / (1.f + e)));
                    }
                    if constexpr (MODE == 2) {
                        if (mok) {
                            b1[cf][r] += gg;
                            b2[cf][r] += gg * (zf - mu[cf][r]);
                        }
                    } else {
                        o[r] = (__bf16)(sc[cf][r] * gg + (kbv[cf][r] * zf + kdv[cf][r]));
                    }
                }
                if constexpr (MODE == 3) {
                    const uint2 u_ = __builtin_bit_cast(uint2, o);
                    o2[cf][0] = u_.x;
                    o2[cf][1] = u_.y;
                }
            }
            if constexpr (MODE == 2) return;
        } else {
#pragma unroll
        for (int cf = 0; cf < 2; cf++) {
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float a_ = (MODE == 0 && bw.round_z) ? (float)(__bf16)acc[cf][r] : acc[cf][r];
                float v = a_ * sc[cf][r] + sh[cf][r];
                if constexpr (ACT == RYOLO_ACT_LEAKY) v = v > 0.f ? v : v * slope;
                else if constexpr (ACT == RYOLO_ACT_MISH) v = mish(v);
                o[r] = (__bf16)v;
                if (STATS) {                                     // statistics of the values as stored (bf16)
                    const float q = mok ? (float)o[r] : 0.f;
                    st_sum[cf][r] += q;
                    st_sq[cf][r] += q * q;
                }
            }
            const uint2 u = __builtin_bit_cast(uint2, o);
            o2[cf][0] = u.x;
            o2[cf][1] = u.y;
        }
        }
        if constexpr (MODE == 1) return;                           // statistics only: nothing is stored
        // each lane holds channels 4g..4g+3 of both channel fragments; the odd rows of fragment 0 trade places with the even
        // rows of fragment 1, after which every lane owns ONE 16-B run (even g: channels 8(g/2).. of fragment 0, odd g: of
        // fragment 1) and the four lanes of a pixel cover its 64-B row
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int d = 0; d < 2; d++) {
            auto sw = __builtin_amdgcn_permlane16_swap(o2[0][d], o2[1][d], false, false);
            o2[0][d] = sw[0];
            o2[1][d] = sw[1];
        }
        const u4 out = u4{o2[0][0], o2[0][1], o2[1][0], o2[1][1]};
        const int voff = mok ? (mcur * p.out_cs + ((g & 1) ? 16 : 0) + (g >> 1) * 8) * 2 : (int)0x80000000;
        if (p.nt_out) __builtin_amdgcn_raw_buffer_store_b128(out, yrs, voff, 0, 2);      // non-temporal
        else __builtin_amdgcn_raw_buffer_store_b128(out, yrs, voff, 0, 0);
#endif
    };
    u4 xa[3], xb[3];
    u2 da[2], db[2];
    request(xa, true);
    request_dy(da, m, true);
    for (int it = 0; it < n_it; it += 2) {
        const int m_a = m;
        advance();
        request(xb, it + 1 < n_it);
        request_dy(db, m, it + 1 < n_it);
        finish(xa, da, m_a, true);
        const int m_b = m;
        advance();
        request(xa, it + 2 < n_it);
        request_dy(da, m, it + 2 < n_it);
        finish(xb, db, m_b, it + 1 < n_it);
    }
    if constexpr (MODE == 2) {
        // 16-lane DPP row sums; lane fr < 8 keeps the totals of (fragment fr / 4, register fr % 4) and adds them to the partial row
        float t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
        for (int cf = 0; cf < 2; cf++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float a1 = row16_sum(b1[cf][r]), a2 = row16_sum(b2[cf][r]), a3 = row16_sum(b3[cf][r]);
                if (fr == cf * 4 + r) {
                    t1 = a1;
                    t2 = a2;
                    t3 = a3;
                }
            }
        if (fr < 8) {
            const int ch = (fr >> 2) * 16 + g * 4 + (fr & 3);
            double *row = bw.part + (size_t)(wave_id % STAT_ROWS) * 3 * 32;
            atomicAdd(row + ch, (double)t1);
            atomicAdd(row + 32 + ch, (double)t2);
            atomicAdd(row + 64 + ch, (double)t3);
        }
        return;
    }
    if (STATS) {
        // the 16 lanes of a k-group hold the same channels: DPP row sum over them, lane fr keeps total number fr (fragment
        // fr / 4, register fr % 4), one atomic instruction per statistic into partial row (wave id mod STAT_ROWS)
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int cf = 0; cf < 2; cf++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float a = row16_sum(st_sum[cf][r]), b = row16_sum(st_sq[cf][r]);
                if (fr == cf * 4 + r) {
                    ta = a;
                    tb = b;
                }
            }
        if (fr < 8) {
            const int ch = (fr >> 2) * 16 + g * 4 + (fr & 3);
            double *row = p.stat_part + (size_t)(wave_id % STAT_ROWS) * 2 * p.stat_cpad;
            atomicAdd(row + ch, (double)ta);
            atomicAdd(row + p.stat_cpad + ch, (double)tb);
        }
    }
}

// layer-0 backward, between the reduce and the apply pass: partial rows -> s1, s2 (x invstd), dgamma += s2, dbeta += s1, the
// slope gradient (one scalar: fixed-order sum over the 32 channels), and the apply pass's per-channel constants
// kb = -scale * invstd * s2 / M, kd = -kb * mean - scale * s1 / M.  One block of 32 threads; re-zeroes the partial rows.
__global__ void conv0_bwd_finalize_kernel(double *__restrict__ part, const float *__restrict__ scale, const float *__restrict__ mean,
                                          const float *__restrict__ invstd, float inv_count, float *__restrict__ kb, float *__restrict__ kd,
                                          float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ dslope) {
    const int c = threadIdx.x;            // 32 threads
    double a = 0.0, b = 0.0, d = 0.0;
    for (int r = 0; r < STAT_ROWS; r++) {
        double *row = part + (size_t)r * 96;
        a += row[c]; b += row[32 + c]; d += row[64 + c];
        row[c] = 0.0; row[32 + c] = 0.0; row[64 + c] = 0.0;
    }
    const float s1 = (float)a, s2 = (float)b * invstd[c];
    if (dgamma) dgamma[c] += s2;
    if (dbeta) dbeta[c] += s1;
    const float k = scale[c] * invstd[c] * s2 * inv_count;
    kb[c] = -k;
    kd[c] = k * mean[c] - scale[c] * s1 * inv_count;
    if (dslope) {
        float v = (float)d;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
        if (c == 0) dslope[0] += v;
    }
}

template <int ACT>
static int launch_c8_bwd(ConvParams &p, int gpw, unsigned nblk, const C8Bwd &bw, int mode, hipStream_t stream) {
    if (mode == 2) hipLaunchKernelGGL((conv3x3_c8_direct_kernel<false, ACT, 2>), dim3(nblk), dim3(256), 0, stream, p, gpw, bw);
    else hipLaunchKernelGGL((conv3x3_c8_direct_kernel<false, ACT, 3>), dim3(nblk), dim3(256), 0, stream, p, gpw, bw);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

template <bool STATS>
int launch_c8_direct(ConvParams &p, int gpw, unsigned nblk, hipStream_t stream) {
    if (STATS && p.y == nullptr) {        // statistics only (layer 0 of the training engine: z is recomputed, never stored)
        hipLaunchKernelGGL((conv3x3_c8_direct_kernel<true, RYOLO_ACT_LINEAR, 1>), dim3(nblk), dim3(256), 0, stream, p, gpw, C8Bwd());
        return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
    }
    if (p.act == RYOLO_ACT_LEAKY) hipLaunchKernelGGL((conv3x3_c8_direct_kernel<STATS, RYOLO_ACT_LEAKY>), dim3(nblk), dim3(256), 0, stream, p, gpw);
    else if (p.act == RYOLO_ACT_MISH) hipLaunchKernelGGL((conv3x3_c8_direct_kernel<STATS, RYOLO_ACT_MISH>), dim3(nblk), dim3(256), 0, stream, p, gpw);
    else hipLaunchKernelGGL((conv3x3_c8_direct_kernel<STATS, RYOLO_ACT_LINEAR>), dim3(nblk), dim3(256), 0, stream, p, gpw);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

__global__ void pack_weights_kernel(const float *__restrict__ w, int Cout, int Cin, int KS, int Cin_pad, int Kpad,
                                    int Cout_pad, __bf16 *__restrict__ out) {
    // out[co][ (kh*KS + kw)*Cin_pad + c ] = w[co][c][kh][kw]  (OIHW in), zero elsewhere, + 128 zero elements tail
    const size_t total = (size_t)Cout_pad * Kpad + 128;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < (size_t)Cout_pad * Kpad) {
            const int co = (int)(i / Kpad), k = (int)(i % Kpad);
            const int tap = k / Cin_pad, c = k % Cin_pad;
            if (co < Cout && tap < KS * KS && c < Cin) {
                const int kh = tap / KS, kw = tap % KS;
                v = w[(((size_t)co * Cin + c) * KS + kh) * KS + kw];
            }
        }
        out[i] = (__bf16)v;
    }
}

__global__ void nchw_f32_to_nhwc_bf16_kernel(const float *__restrict__ x, int N, int C, int H, int W, int Cpad,
                                             __bf16 *__restrict__ y) {
    // one thread per output pixel; channels C..Cpad-1 zero.  Cpad == 8: one 16-B store per pixel.
    const size_t npix = (size_t)N * H * W;
    for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (size_t)gridDim.x * blockDim.x) {
        const size_t hw = pix % ((size_t)H * W), n = pix / ((size_t)H * W);
        for (int c0 = 0; c0 < Cpad; c0 += 8) {
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int c = c0 + e;
                v[e] = (__bf16)(c < C ? x[((size_t)n * C + c) * H * W + hw] : 0.f);
            }
            *(bf16x8 *)(y + pix * Cpad + c0) = v;
        }
    }
}

__global__ void nhwc_bf16_to_nchw_f32_kernel(const __bf16 *__restrict__ x, int N, int C, int H, int W, int cs,
                                             float *__restrict__ y) {
    const size_t total = (size_t)N * C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t hw = i % ((size_t)H * W);
        const size_t t = i / ((size_t)H * W);
        const int c = (int)(t % C);
        const size_t n = t / C;
        y[i] = (float)x[(n * H * W + hw) * cs + c];
    }
}

inline int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) l++;
    return (1 << l) == v ? l : -1;
}

// set by ryolo_conv2d_dgrad_bnreduce around its dispatch: the launch must be the persistent 1x1 kernel's BNRED instantiation
static thread_local const BnRed *g_bnred = nullptr;
static thread_local int g_bnred_mode = 0;        // bnreduce_plan's choice (the partial rows are sized for that launch): 1 conv_pw.hip's MODE 3, 2 the
                                                 // persistent 2x2 tile (one row per workgroup), 3 a one-tile-per-workgroup tile (one row per pixel tile),
                                                 // 4 conv_mq.hip's 128-channel tiles (one row per workgroup)

// tile code of a (BM, BN, WGM, WGN, NSTAGE) instantiation as ryolo_conv_desc::tile / RYOLO_CONV_KERNEL_IGEMM + code name it
template <int BM, int BN, int WGM, int WGN, int NSTAGE>
constexpr int igemm_tile_code() {
    return BM == 256 ? (BN == 64 ? 2 : (BN == 32 ? 3 : 4)) : (WGM == 4 ? 6 : (WGN == 2 ? 7 : 1));
}

// the one-tile-per-workgroup instantiations that exist with the folded BatchNorm reduce (bnreduce_plan mode 3): the tiles the stride-1
// data gradients of the 3x3 layers with C_in <= 128 and of the narrow 1x1 layers take
template <int KS, int BM, int BN, int WGM, int WGN, int NSTAGE>
constexpr bool igemm_bnred_inst() {
    return NSTAGE == 2 && ((BM == 128 && BN == 128 && WGM == 2 && WGN == 4 && KS == 3) || (BM == 256 && WGM == 4 && WGN == 1 && (BN == 64 || BN == 32)));
}

template <int KS, int BM, int BN, int WGM, int WGN, int NSTAGE, bool FAST, bool GEN>
int launch_variant_impl(ConvParams &p, hipStream_t stream) {
    if (g_bnred) {
        if constexpr (FAST && !GEN && igemm_bnred_inst<KS, BM, BN, WGM, WGN, NSTAGE>()) {
            if (g_bnred_mode != 3 || p.stat_part || p.ups != 1 || p.os != 1) return RYOLO_EINVAL;
            RYOLO_CONV_DRY_RUN((RYOLO_CONV_KERNEL_IGEMM + igemm_tile_code<BM, BN, WGM, WGN, NSTAGE>()));
            constexpr size_t smem_bn = NSTAGE * (BM + BN) * BK * 2;
            static bool attr_bn = false;
            auto kbn = conv_igemm_kernel<KS, BM, BN, WGM, WGN, NSTAGE, true, false, true>;
            if (!attr_bn) {
                if (smem_bn > 64 * 1024 &&
                    hipFuncSetAttribute((const void *)kbn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bn) != hipSuccess)
                    return RYOLO_ELAUNCH;
                attr_bn = true;
            }
            const int mt = (p.M + BM - 1) / BM;
            p.nt = (p.Cout + BN - 1) / BN;
            hipLaunchKernelGGL(kbn, dim3((unsigned)(mt * p.nt)), dim3(WGM * WGN * 64), smem_bn, stream, p, *g_bnred);
            return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
        } else {
            return RYOLO_EINVAL;
        }
    }
    RYOLO_CONV_DRY_RUN((RYOLO_CONV_KERNEL_IGEMM + igemm_tile_code<BM, BN, WGM, WGN, NSTAGE>()));
    constexpr int STAGE = (BM + BN) * BK * 2;
    constexpr size_t smem = NSTAGE * STAGE;
    static bool attr_done = false;
    auto kfn = conv_igemm_kernel<KS, BM, BN, WGM, WGN, NSTAGE, FAST, GEN>;
    if (!attr_done) {
        if (smem > 64 * 1024 &&
            hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return RYOLO_ELAUNCH;
        attr_done = true;
    }
    const int mt = (p.M + BM - 1) / BM;
    p.nt = (p.Cout + BN - 1) / BN;
    hipLaunchKernelGGL(kfn, dim3((unsigned)(mt * p.nt)), dim3(WGM * WGN * 64), smem, stream, p, BnRed());
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

inline int cu_count() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    return cus;
}

inline unsigned magic_u32(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

template <int KS, int BM, int BN, int WGM, int WGN>
int launch_persist(ConvParams &p, int grid, hipStream_t stream) {
    constexpr size_t smem = 2 * (BM + BN) * BK * 2;
    if (g_conv_choice) {         // the instantiations below exist for these shapes only: report what would really be launched
        const bool ok = g_bnred ? (g_bnred_mode == 2 && KS == 1 && BM == 128 && BN == 128 && WGM == 2 && WGN == 2 && !p.stat_part && p.ups == 1)
                                : (!p.stat_part || WGM * WGN == 4);
        if (!ok) return RYOLO_EINVAL;
        RYOLO_CONV_DRY_RUN((RYOLO_CONV_KERNEL_IGEMM + igemm_tile_code<BM, BN, WGM, WGN, 2>()));
    }
    if (g_bnred) {
        if constexpr (KS == 1 && BM == 128 && BN == 128 && WGM == 2 && WGN == 2) {
            if (p.stat_part || p.ups != 1 || g_bnred_mode != 2) return RYOLO_EINVAL;
            constexpr size_t smem_bn = smem + (size_t)WGM * WGN * (BN / 8) * 24 * 4;
            static bool attr_bn = false;
            if (!attr_bn) {
                if (hipFuncSetAttribute((const void *)conv_igemm_persist_kernel<KS, BM, BN, WGM, WGN, false, true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bn) != hipSuccess)
                    return RYOLO_ELAUNCH;
                attr_bn = true;
            }
            hipLaunchKernelGGL((conv_igemm_persist_kernel<KS, BM, BN, WGM, WGN, false, true>), dim3((unsigned)grid), dim3(WGM * WGN * 64),
                               smem_bn, stream, p, *g_bnred);
            return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
        } else {
            return RYOLO_EINVAL;
        }
    }
    if constexpr (WGM * WGN == 4) {     // the statistics instantiation exists for the 4-wave tiles (1x1 layers, short-K 3x3 layers)
        if (p.stat_part) {
            constexpr size_t smem_st = smem + (size_t)WGM * 2 * BN * 4;
            static bool attr_done = false;
            if (!attr_done) {
                if (hipFuncSetAttribute((const void *)conv_igemm_persist_kernel<KS, BM, BN, WGM, WGN, true>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_st) != hipSuccess)
                    return RYOLO_ELAUNCH;
                attr_done = true;
            }
            hipLaunchKernelGGL((conv_igemm_persist_kernel<KS, BM, BN, WGM, WGN, true>), dim3((unsigned)grid), dim3(WGM * WGN * 64), smem_st, stream, p);
            return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
        }
    }
    if (p.stat_part) return RYOLO_EINVAL;
    hipLaunchKernelGGL((conv_igemm_persist_kernel<KS, BM, BN, WGM, WGN, false>), dim3((unsigned)grid), dim3(WGM * WGN * 64), smem, stream, p);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

template <int KS, int BM, int BN, int WGM, int WGN, int NSTAGE = 2>
int launch_variant(ConvParams &p, hipStream_t stream) {
    const bool gen = p.stat_part != nullptr || p.os != 1;
    {
        const long long dmax = p.Wo > p.Ho ? p.Wo : p.Ho, mpad = ((long long)p.M + BM - 1) / BM * BM;
        p.use_magic = mpad * dmax < 0x100000000ll ? 1 : 0;
        p.magic_wo = magic_u32(p.Wo);
        p.magic_ho = magic_u32(p.Ho);
    }
    // measured (tools/layer_bench.py, MI355X): the persistent grid wins on the short-K 1x1 layers (fixed per-tile cost
    // dominates: 64->32 @304 0.146 -> 0.112 ms, 256->128 @76 0.034 -> 0.031) and loses 3-5 % on the long-K 3x3 layers
    // (0.120 -> 0.127 ms), so only the 1x1 instantiations take it unless tile bit 0x800 forces it
    // (the training forward of a 1x1 layer takes it too: the 4-wave tiles have a statistics instantiation)
    const bool persist_ok = p.os == 1 && (!p.stat_part || (WGM * WGN == 4 && !p.res));
    if constexpr (NSTAGE == 2 && BM * BN * 2 <= (BM + BN) * BK * 2) if (p.fast && persist_ok && !p.no_persist && (KS == 1 || p.force_persist)) {
        // persistent grid when there is more than one round of tiles and the multiply-high divisions are exact
        const int mt = (p.M + BM - 1) / BM, nt = (p.Cout + BN - 1) / BN;
        const long long T = (long long)mt * nt;
        const int grid = (2 * cu_count()) & ~7;
        const long long dmax = p.Wo > p.Ho ? p.Wo : p.Ho;
        // ... and only when the tile list is at least 2.5 rounds deep (measured: 1444 tiles 0.030 -> 0.028 ms, 722 tiles
        // 0.021 -> 0.022), with the 4-wave layout (the 8-wave persistent variant is slower: 0.031)
        const bool deep = p.force_persist ? T > grid : 2 * T >= 5 * grid;
        if (deep && grid >= 8 && ((long long)mt * BM) * dmax < 0x100000000ll && T * nt < 0x100000000ll) {
            p.nt = nt;
            p.ntiles = (int)T;
            p.magic_wo = magic_u32(p.Wo); p.magic_ho = magic_u32(p.Ho); p.magic_nt = magic_u32(nt);
            return launch_persist<KS, BM, BN, WGM, WGN>(p, grid, stream);
        }
    }
    if (p.fast) {
        if (gen) return launch_variant_impl<KS, BM, BN, WGM, WGN, NSTAGE, true, true>(p, stream);
        return launch_variant_impl<KS, BM, BN, WGM, WGN, NSTAGE, true, false>(p, stream);
    }
    if (gen) return launch_variant_impl<KS, BM, BN, WGM, WGN, NSTAGE, false, true>(p, stream);
    return launch_variant_impl<KS, BM, BN, WGM, WGN, NSTAGE, false, false>(p, stream);
}

}  // namespace

// ---- the tuning switches of the shipped library (conv_common.h: TuneKey).  One getenv per switch, once.
static const char *const TUNE_NAMES[TUNE_COUNT] = {"RYOLO_CONV3X3", "RYOLO_CONV1X1", "RYOLO_RNMS_TILES", "RYOLO_MQ_KORDER", "RYOLO_BN_REDUCE_TILES",
                                                   "RYOLO_STEM_DGRAD"};
static char g_tune_val[TUNE_COUNT][24];
static bool g_tune_set[TUNE_COUNT];
static std::once_flag g_tune_once;
static void tune_store(int k, const char *v) {
    g_tune_set[k] = v != nullptr && v[0] != 0;
    if (g_tune_set[k]) {
        strncpy(g_tune_val[k], v, sizeof(g_tune_val[k]) - 1);
        g_tune_val[k][sizeof(g_tune_val[k]) - 1] = 0;
    }
}
static void tune_init() {
    for (int k = 0; k < TUNE_COUNT; k++) tune_store(k, getenv(TUNE_NAMES[k]));
}
namespace ryolo_detail {
const char *tune(TuneKey k) {
    std::call_once(g_tune_once, tune_init);
    return g_tune_set[k] ? g_tune_val[k] : nullptr;
}
}  // namespace ryolo_detail
extern "C" int ryolo_set_tuning(const char *name, const char *value) {
    if (!name) return RYOLO_EINVAL;
    std::call_once(g_tune_once, tune_init);
    for (int k = 0; k < TUNE_COUNT; k++)
        if (!strcmp(name, TUNE_NAMES[k])) {
            tune_store(k, value);         // (not synchronised with launches in flight on other threads: a test / A-B facility)
            return RYOLO_OK;
        }
    return RYOLO_EINVAL;
}

extern "C" {

size_t ryolo_conv_packed_weight_bytes(int Cout, int Cin_pad, int ksize) {
    if (Cout <= 0 || Cin_pad <= 0 || (ksize != 1 && ksize != 3)) return 0;
    const size_t K = (size_t)ksize * ksize * Cin_pad;
    const size_t Kpad = (K + BK - 1) / BK * BK;
    const size_t Cout_pad = ((size_t)Cout + 127) / 128 * 128;
    return (Cout_pad * Kpad + 128) * 2;
}

int ryolo_conv_pack_weights(const float *w_oihw, int Cout, int Cin, int ksize, int Cin_pad, void *packed,
                            void *stream) {
    if (!w_oihw || !packed || Cout <= 0 || Cin <= 0 || Cin_pad < Cin || (Cin_pad & 7) || (ksize != 1 && ksize != 3))
        return RYOLO_EINVAL;
    const int K = ksize * ksize * Cin_pad, Kpad = (K + BK - 1) / BK * BK, Cout_pad = (Cout + 127) / 128 * 128;
    const size_t total = (size_t)Cout_pad * Kpad + 128;
    const int nb = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, ksize,
                       Cin_pad, Kpad, Cout_pad, (__bf16 *)packed);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

int ryolo_nchw_f32_to_nhwc_bf16(const float *x, int N, int C, int H, int W, int Cpad, void *y, void *stream) {
    if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || Cpad < C || (Cpad & 7)) return RYOLO_EINVAL;
    const size_t npix = (size_t)N * H * W;
    const int nb = (int)((npix + 255) / 256 < 16384 ? (npix + 255) / 256 : 16384);
    hipLaunchKernelGGL(nchw_f32_to_nhwc_bf16_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, N, C, H, W, Cpad,
                       (__bf16 *)y);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

int ryolo_nhwc_bf16_to_nchw_f32(const void *x, int N, int C, int H, int W, int cstride, float *y, void *stream) {
    if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || cstride < C) return RYOLO_EINVAL;
    const size_t total = (size_t)N * C * H * W;
    const int nb = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(nhwc_bf16_to_nchw_f32_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)x, N,
                       C, H, W, cstride, y);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

static int pick_tile(const ryolo_conv_desc *d, int cout) {
    (void)cout;
    return d->tile & 0xff;   // 0 = auto (dispatch decides); bit 8 (0x100) forces the general (slow-address) path, for tests
}


// Which 256-channel tile a 3x3 layer takes: conv_mp.hip (one 8-wave workgroup per CU, 256 or 192 pixel rows) or conv_mq.hip (two
// 4-wave workgroups per CU, 128 rows).  Estimated launch time = (tiles of the busiest CU) x (time per tile), time per tile
// linear in the K depth; coefficients (us, bs 32 / 608^2 layers with the fused shortcut, MI355X) from tools/mp_ablate.py,
// profiles/r03_mp_vs_mq.txt.  conv_mq wins where the tile list is deep enough to keep both workgroups of every CU busy (the
// 76^2 and 38^2 layers), conv_mp's 192-row tile where one round of tiles fills the chip better (19^2: 244 tiles on 256 CUs).
// Returns 0 = conv_mq, else the BM of conv_mp.  RYOLO_CONV3X3 = mp | mq overrides (A/B timing, tests).
static int pick_wide_tile(const ConvParams &p) {
    const char *e = tune(TUNE_CONV3X3);
    const int forced = !e ? 0 : (!strcmp(e, "mp") ? 1 : (!strcmp(e, "mq") ? 2 : 0));
    if (forced == 2) return 0;
    const int bm_mp = conv_mp_pick_bm(p);
    if (forced == 1) return bm_mp;
    // Cost model in microseconds, calibrated on tools/mp_ablate.py --exp mq at bs 32 and bs 64 (K tiles 18 / 36 / 72 = the 76^2 /
    // 38^2 / 19^2 layers; piecewise linear in between, the last slope beyond):
    //   conv_mp   rounds of the persistent grid x the time of one tile;
    //   conv_mq   the busiest CU hosts workgroups `loc` and `loc + wgx/2` of an XCD with a >= b tiles: b tile slots run with both
    //             workgroups resident (c2 per slot), the a - b slots after the partner has finished run alone at 0.7 c2.
    const long long cus = cu_count() & ~7, nt = (p.Cout + 255) / 256, kt = p.Kpad / BK;
    auto interp = [&](double v18, double v36, double v72) {
        return kt <= 36 ? v18 + (kt - 18) * (v36 - v18) / 18.0 : v36 + (kt - 36) * (v72 - v36) / 36.0;
    };
    auto rounds = [&](int bm) { return ((((long long)p.M + bm - 1) / bm) * nt + cus - 1) / cus; };
    const double t256 = rounds(256) * interp(35.5, 59.5, 102.0), t192 = rounds(192) * interp(30.5, 53.0, 92.0);
    // conv_mq: XCD chunks of the tile list, 2 x cus / 8 workgroups per XCD, CU c hosts workgroups c and c + cus / 8
    const long long tq = ((((long long)p.M + 127) / 128) * nt + 7) / 8, wgx = 2 * cus / 8;
    auto ntile = [&](long long loc) { return loc < tq ? (tq - loc + wgx - 1) / wgx : 0; };
    const double c2 = interp(32.5, 59.0, 97.0);
    const double tq_ = (double)ntile(wgx / 2) * c2 + (double)(ntile(0) - ntile(wgx / 2)) * 0.7 * c2;
    if (tq_ < t256 && tq_ < t192) return 0;
    return t192 < t256 ? 192 : 256;
}

static long long g_nt_out_min = NT_OUT_MIN_BYTES;      // (settable in ablation builds: ryolo_debug_conv_nt_min)
// (measurement build: RYOLO_NT_OUT_MIN_MB overrides the threshold (MiB) per call -- the in-chain sweep of tools/step_ab.py, profiles/r05_ab_log.txt)
static inline long long nt_out_min_bytes() {
    const char *e = abl_env("RYOLO_NT_OUT_MIN_MB");
    return e ? (long long)atoll(e) << 20 : g_nt_out_min;
}
// (sc1 write-through stores for the outputs below that threshold -- no dirty lines left in L2 at the kernel boundary -- were tried in the chain
// in round 5 and lost: bs-32 forward 6.105 ms default / 6.148 from 32 MiB / 6.191 from 8 MiB / 6.130 everywhere, profiles/r05_ab_log.txt; removed)
#ifdef RYOLO_MP_ABLATION
extern "C" void ryolo_debug_conv_nt_min(long long bytes) { g_nt_out_min = bytes; }
#endif

// RYOLO_CONV1X1 = igemm keeps the 1x1 layers on the 128 x 128 tiles (A/B timing; ryolo_set_tuning)
static bool conv_pw_disabled() {
    const char *e = tune(TUNE_CONV1X1);
    return e && !strcmp(e, "igemm");
}



// conv_mq.hip's 128-channel tiles (round 5).  RYOLO_MQ128 = 0 (default): off -- the 128 x 128 / 256 x 64 tiles of this file as in round 4;
// 1: the 3x3 layers and data gradients with C_out % 256 != 0; 2: also the 1x1 layers whose 128-pixel tile list is short (38^2 / 19^2).
// OPT-IN because they measure no faster than the tiles they would replace (profiles/r05_mq128_bench.txt, r05_ab_log.txt: bs-64 step 49.27 ms
// off / 49.59 on / 49.85 with the 1x1 layers; bs-32 forward 6.08 / 6.07 / 6.30 ms): with 32 channels per wave a K tile moves 0.625 KiB of
// LDS fragments per MFMA against 0.375 for the 256-channel tile -- the LDS pipe, not the schedule, bounds these layers.
// MEASUREMENT BUILD ONLY (round 6): the shipped library has neither the switch nor the instantiations (launch_conv_mq128 returns EINVAL).
static int mq128_knob() {
    const char *e = abl_env("RYOLO_MQ128");
    return e ? atoi(e) : 0;
}
// pixels per tile: 64 when the list of 128-pixel tiles is less than 2.5 rounds of the two-workgroups-per-CU grid deep
static int mq128_pick_bm(const ConvParams &p) {
    const long long t128 = (((long long)p.M + 127) / 128) * (p.Cout / 128), grid = (2 * cu_count()) & ~7;
    return 2 * t128 >= 5 * grid ? 128 : 64;
}
// does the auto dispatch send this launch to the 128-channel family?  (3x3: every eligible shape the 256-channel tiles do not serve;
// 1x1 (knob 2): K >= 256 and a short tile list -- the 76^2 layers stay on conv_pw.hip, HBM-bound and at their floor there)
static bool mq128_auto(const ConvParams &p, int ksize) {
    const int knob = mq128_knob();
    if (knob <= 0 || !conv_mq128_eligible(p)) return false;
    if (ksize == 3) return !conv_mp_eligible(p);
    if (knob < 2 || p.Kpad < 4 * BK) return false;
    const long long t128 = (((long long)p.M + 127) / 128) * (p.Cout / 128), grid = (2 * cu_count()) & ~7;
    return t128 < 4 * grid;
}

// RYOLO_CONV0=direct keeps layer 0 on conv3x3_c8_direct_kernel (fragments from global memory); default: the LDS-staged kernel of
// conv_stem.hip (measurement build only since round 6: A/B timing)
static bool conv0_halo_on() {
    const char *e = abl_env("RYOLO_CONV0");
    return !(e && !strcmp(e, "direct"));
}

static int dispatch(ConvParams &p, int ksize, int pick, hipStream_t stream) {
    // the stem kernel (conv_stem.hip: 3x3, 32 -> 64 channels, input patch staged once): auto and pick 12
    const bool stem = (pick == 0 || pick == 12) && conv_stem_eligible(p, ksize) && !g_bnred;
    if (pick == 12 && !stem) return RYOLO_EINVAL;
    if (g_bnred && g_bnred_mode == 3) p.no_persist = 1;          // the reduce rides in the one-tile-per-workgroup kernel
    // the weight-stationary 1x1 kernel (conv_pw.hip): auto and pick 13; RYOLO_CONV1X1 = igemm keeps the 128x128 tiles (A/B timing, tests)
    const bool pw = ((pick == 0 && !conv_pw_disabled()) || pick == 13) && conv_pw_eligible(p, ksize) && (pick == 13 || (g_bnred ? g_bnred_mode == 1 : conv_pw_preferred(p)));
    if (pick == 13 && !pw) return RYOLO_EINVAL;
    if (stem) return launch_conv_stem(p, cu_count(), stream);
    // conv_stem.hip's 3x3 / 1 64 -> 128 kernel (round 6): auto and pick 17 (measurement build: RYOLO_STEM64=0 keeps the 128 x 128 tiles, A/B timing)
    {
        const char *e64 = abl_env("RYOLO_STEM64");
        const bool stem64 = ((pick == 0 && !(e64 && !strcmp(e64, "0"))) || pick == 17) && conv_stem64_eligible(p, ksize) && !g_bnred;
        if (pick == 17 && !stem64) return RYOLO_EINVAL;
        if (stem64) return launch_conv_stem64(p, cu_count(), stream);
    }
    // conv_mq.hip's 128-channel tiles: picks 15 (128 pixels) / 16 (64 pixels); auto per mq128_auto(); with the folded reduce only as
    // bnreduce_plan's mode 4 (its caller sized the partial rows for that grid)
    if (pick == 15 || pick == 16) return launch_conv_mq128(p, pick == 15 ? 128 : 64, g_bnred_mode == 4 ? g_bnred : nullptr, stream);
    if (pick == 0 && (g_bnred ? g_bnred_mode == 4 : mq128_auto(p, ksize))) {
        const int r = launch_conv_mq128(p, mq128_pick_bm(p), g_bnred, stream);
        if (r != RYOLO_EINVAL || g_bnred) return r;              // EINVAL: a size guard -- the tiles below take those
    }
    if (pw) {
        const int r = launch_conv_pw(p, g_bnred, stream);
        if (r != RYOLO_EINVAL || pick == 13 || g_bnred) return r;      // EINVAL: a size guard (2 GiB slices) -- the 128x128 tiles take those
                                                                        // (not with the folded reduce: its caller sized the partial rows for THIS grid)
    }
    if (pick == 0) {
        // auto: 3x3 layers with 256-multiple output channels take one of the persistent multi-phase tiles (the 1x1 layers are
        // faster on the 128x128 tiles, tools/mp_tune.py)
        if (ksize == 3 && conv_mp_eligible(p) && !g_bnred) {
            const int bm = pick_wide_tile(p);
            const int r = bm == 0 ? launch_conv_mq(p, 0, stream) : launch_conv_mp(p, bm, 0, stream);
            if (r != RYOLO_EINVAL) return r;      // EINVAL: a size guard of the persistent tiles (2 GiB output slices, 2^32 pixel*extent) -- the 128x128 tiles take those
        }
        pick = p.Cout <= 32 ? 3 : (p.Cout <= 64 ? 2 : 1);
        // 3x3 layers with a short K loop (C_in <= 64: at most 9 K steps) are all tile prologue / epilogue on the one-tile-per-
        // workgroup grid; the persistent 4-wave tiles prefetch the next tile's first K step under the epilogue (measured bs 32:
        // 3x3/2 32->64@304 0.339 -> 0.297 ms, 3x3 64->128@152 0.151 -> 0.136, 3x3/2 64->128@152 0.178 -> 0.165; long-K layers lose)
        // (not the statistics instantiation of the 256 x 64 tile: under the two-workgroup register cap it spills 54 VGPRs and runs
        // 2.2 x slower than the one-tile grid; with 512 registers and one workgroup per CU it is no faster than that grid either)
        if (ksize == 3 && p.os == 1 && p.fast && p.Kpad / BK <= 9 && !p.no_persist && !(p.stat_part && (p.res || pick == 2))) {
            p.force_persist = 1;
            if (pick == 1) pick = 7;
        }
    }
    // picks 8 / 11 / 14: the 256-channel multi-phase tile of conv_mp.hip with BM 256 / BM 192 / BM picked per shape (tests, A/B timing)
    if (pick == 8) return launch_conv_mp(p, 256, 0, stream);
    if (pick == 11) return launch_conv_mp(p, 192, 0, stream);
    if (pick == 14) return launch_conv_mp(p, 0, 0, stream);
    if (pick == 9) return launch_conv_mq(p, 0, stream);      // the two-workgroups-per-CU tile of conv_mq.hip
#ifdef RYOLO_MP_ABLATION
    if (pick >= 32 && pick < 64) return launch_conv_mp(p, (pick & 16) ? 192 : 256, ryolo_mp_ablation_variant(pick & 15), stream);
    if (pick >= 64 && pick < 80) return launch_conv_mq(p, ryolo_mp_ablation_variant(pick & 15), stream);
#endif
    if (ksize == 1) {
        if (pick == 1) {   // 8 waves of 64 pixels x 32 channels, except where the (4-wave) persistent grid wins
            const long long T = (((long long)p.M + 127) / 128) * ((p.Cout + 127) / 128);
            if (p.fast && p.os == 1 && !(p.stat_part && p.res) && !p.no_persist && (p.force_persist || 2 * T >= 5 * (long long)((2 * cu_count()) & ~7)))
                return launch_variant<1, 128, 128, 2, 2>(p, stream);
            p.no_persist = 1;
            return launch_variant<1, 128, 128, 2, 4>(p, stream);
        }
        if (pick == 7) return launch_variant<1, 128, 128, 2, 2>(p, stream);
        if (pick == 2) return launch_variant<1, 256, 64, 4, 1>(p, stream);
        if (pick == 3) return launch_variant<1, 256, 32, 4, 1>(p, stream);
        if (pick == 4) return launch_variant<1, 256, 128, 4, 2, 3>(p, stream);
        if (pick == 6) return launch_variant<1, 128, 128, 4, 2>(p, stream);
    } else {
        if (pick == 1) return launch_variant<3, 128, 128, 2, 4>(p, stream);
        if (pick == 7) return launch_variant<3, 128, 128, 2, 2>(p, stream);
        if (pick == 2) return launch_variant<3, 256, 64, 4, 1>(p, stream);
        if (pick == 3) return launch_variant<3, 256, 32, 4, 1>(p, stream);
        if (pick == 4) return launch_variant<3, 256, 128, 4, 2, 3>(p, stream);
        if (pick == 6) return launch_variant<3, 128, 128, 4, 2>(p, stream);
    }
    return RYOLO_EINVAL;
}

static int validate(const ryolo_conv_desc *d) {
    if (!d) return RYOLO_EINVAL;
    if (d->ksize != 1 && d->ksize != 3) return RYOLO_EINVAL;
    if (d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->stride <= 0) return RYOLO_EINVAL;
    if ((d->Cin & 7) || (d->Cout & 7) || (d->in_cstride & 7) || (d->out_cstride & 7)) return RYOLO_EINVAL;
    if (d->in_cstride < d->Cin || d->out_cstride < d->Cout) return RYOLO_EINVAL;
    if (d->upsample != 1 && d->upsample != 2) return RYOLO_EINVAL;
    if (d->ksize == 1 && d->pad != 0) return RYOLO_EINVAL;
    return RYOLO_OK;
}

int ryolo_conv_stat_rows(const ryolo_conv_desc *d) {
    if (validate(d) != RYOLO_OK) return 0;
    return STAT_ROWS;
}

int ryolo_conv2d_bn_act_stats(const ryolo_conv_desc *d, const void *x, const void *w_packed, const float *scale,
                              const float *shift, const void *residual, void *y, double *stat_part, void *stream_) {
    if (validate(d) != RYOLO_OK || !x || !w_packed || !scale || !shift) return RYOLO_EINVAL;
    if (!y && !(stat_part && ryolo_conv0_recompute_supported(d))) return RYOLO_EINVAL;     // y == NULL: statistics only, layer-0 kernel only
    if (residual && ((d->res_cstride & 7) || d->res_cstride < d->Cout)) return RYOLO_EINVAL;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w_packed | (uintptr_t)residual | (uintptr_t)scale | (uintptr_t)shift) & 15)
        return RYOLO_EINVAL;
    ConvParams p;
    p.x = (const __bf16 *)x;
    p.w = (const __bf16 *)w_packed;
    p.scale = scale;
    p.shift = shift;
    p.res = (const __bf16 *)residual;
    p.y = (__bf16 *)y;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.in_cs = d->in_cstride;
    p.stride = d->stride;
    p.pad = d->pad;
    p.Ho = (d->H + 2 * d->pad - d->ksize) / d->stride + 1;
    p.Wo = (d->W + 2 * d->pad - d->ksize) / d->stride + 1;
    if (p.Ho <= 0 || p.Wo <= 0) return RYOLO_EINVAL;
    p.Cout = d->Cout; p.out_cs = d->out_cstride; p.res_cs = d->res_cstride;
    p.K = d->ksize * d->ksize * d->Cin;
    p.Kpad = (p.K + BK - 1) / BK * BK;
    const long long M = (long long)d->N * p.Ho * p.Wo;
    if (M > 0x7fffffffLL - 512) return RYOLO_EINVAL;
    p.M = (int)M;
    p.cin_log2 = ilog2_exact(d->Cin);
    if (d->ksize == 3 && p.cin_log2 < 0 && (d->Cin % BK)) return RYOLO_EINVAL;
    p.act = d->act; p.slope = d->slope; p.ups = d->upsample; p.nt = 0;
    {
        const unsigned long long xb = (((unsigned long long)d->N * d->H * d->W - 1) * d->in_cstride + d->Cin) * 2ull;
        const unsigned long long wb = ((unsigned long long)((d->Cout + 127) / 128 * 128) * p.Kpad + 128) * 2ull;
        p.taps2 = (d->Cin == 32 && d->ksize == 3) ? 1 : 0;
        p.fast = (d->Cin % BK == 0 || p.taps2) && xb < 0x7fffff00ull && wb < 0x7fffff00ull && !(d->tile & 0x100);
        if (!p.fast) p.taps2 = 0;
        p.x_bytes = (unsigned)(p.fast ? xb : 0);
        p.w_bytes = (unsigned)(p.fast ? wb : 0);
    }
    p.ntaps = d->ksize * d->ksize;
    for (int t = 0; t < 9; t++) { p.tap_dy[t] = t / d->ksize; p.tap_dx[t] = t % d->ksize; }
    p.os = 1; p.osx = 1; p.ooy = 0; p.oox = 0; p.OH = p.Ho; p.OW = p.Wo;
    p.no_persist = (d->tile & 0x200) ? 1 : 0;
    p.force_persist = (d->tile & 0x800) ? 1 : 0;
    p.pw_grid_cap = (d->tile & 0xff) == 13 ? (d->tile >> 16) & 0xff : 0;
#ifdef RYOLO_MP_ABLATION
    if ((d->tile & 0x400) && p.fast) p.x_bytes = p.w_bytes = 0;   // timing experiment: every load out of range (zeros, no traffic)
#endif
    p.ntiles = 0; p.magic_wo = p.magic_ho = p.magic_nt = 0;
    p.nt_out = (long long)p.M * d->upsample * d->upsample * d->Cout * 2 >= nt_out_min_bytes() ? 1 : 0;
    p.stat_part = stat_part;
    p.stat_cpad = (d->Cout + 127) / 128 * 128;
    const unsigned long long c8_xb = (unsigned long long)d->N * d->H * d->W * 16ull, c8_yb = ((unsigned long long)(p.M - 1) * d->out_cstride + 32) * 2ull;
    if (d->ksize == 3 && d->stride == 1 && d->pad == 1 && d->Cin == 8 && d->in_cstride == 8 && d->Cout == 32 &&
        !residual && d->upsample == 1 && !(d->tile & 0x1ff) && p.Wo >= 16 && c8_xb < 0x7fffff00ull && c8_yb < 0x7fffff00ull) {
        // Darknet-53 layer 0: fragments straight from global memory (conv3x3_c8_direct_kernel); 32-bit buffer offsets
        p.x_bytes = (unsigned)c8_xb;
        p.w_bytes = (unsigned)c8_yb;                          // this kernel's second descriptor covers y
        const long long groups = ((long long)p.M + 15) / 16;
        const int gpw = (d->tile >> 16) ? (d->tile >> 16) : 32;   // groups of 16 pixels per wave (upper tile bits: tuning)
        const long long waves = (groups + gpw - 1) / gpw;
        const unsigned nblk = (unsigned)((waves + 3) / 4);
        const bool staged = conv0_halo_on() && !(d->tile >> 16) && y && !stat_part;
        RYOLO_CONV_DRY_RUN((staged ? RYOLO_CONV_KERNEL_STEM0 : RYOLO_CONV_KERNEL_DIRECT8));
        // (the statistics-only pass stays on the direct kernel: 284 us against 320 for the staged one -- that pass is bound by its per-element
        // arithmetic, not by the fragment path)
        // (... and so do the statistics of the stored-z form: the two statistics passes must add the same values in the same order)
        if (staged) return launch_conv0_halo(p, nullptr, 0, cu_count(), (hipStream_t)stream_);
        return stat_part ? launch_c8_direct<true>(p, gpw, nblk, (hipStream_t)stream_) : launch_c8_direct<false>(p, gpw, nblk, (hipStream_t)stream_);
    }
    return dispatch(p, d->ksize, pick_tile(d, d->Cout), (hipStream_t)stream_);
}

static bool head_params(const ryolo_conv_desc *d, ConvParams &p) {
    if (validate(d) != RYOLO_OK || d->ksize != 1 || d->stride != 1 || d->pad != 0 || d->upsample != 1 || (d->Cin % BK)) return false;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.in_cs = d->in_cstride; p.stride = 1; p.pad = 0; p.Ho = d->H; p.Wo = d->W;
    p.Cout = d->Cout; p.out_cs = d->Cout; p.res_cs = 0; p.K = d->Cin; p.Kpad = d->Cin;
    const long long M = (long long)d->N * d->H * d->W;
    if (M > 0x7fffffffLL - 512) return false;
    p.M = (int)M;
    p.act = d->act; p.slope = d->slope; p.ups = 1; p.nt = 0; p.taps2 = 0; p.os = 1; p.osx = 1; p.ooy = 0; p.oox = 0; p.OH = p.Ho; p.OW = p.Wo;
    const unsigned long long xb = (((unsigned long long)M - 1) * d->in_cstride + d->Cin) * 2ull;
    const unsigned long long wb = ((unsigned long long)((d->Cout + 127) / 128 * 128) * p.Kpad + 128) * 2ull;
    if (xb >= 0x7fffff00ull || wb >= 0x7fffff00ull) return false;
    p.fast = 1; p.x_bytes = (unsigned)xb; p.w_bytes = (unsigned)wb;
    p.res = nullptr; p.y = nullptr; p.stat_part = nullptr; p.stat_cpad = 0; p.nt_out = 0; p.pw_grid_cap = 0;
    p.no_persist = 0; p.force_persist = 0; p.ntaps = 1; p.ntiles = 0; p.magic_wo = p.magic_ho = p.magic_nt = 0; p.use_magic = 1;
    return true;
}

int ryolo_conv_head_decode_supported(const ryolo_conv_desc *d, int na, int no) {
    ConvParams p;
    return head_params(d, p) && conv_pw_decode_supported(p, na, no) ? 1 : 0;
}

int ryolo_conv_head_decode(const ryolo_conv_desc *d, const void *x, const void *w_packed, const float *scale, const float *shift,
                           const float *anchors, int na, int no, float stride, float context_factor, int arc, float *io,
                           long long io_rows_per_image, long long io_row_offset, float *pout, void *stream_) {
    ConvParams p;
    if (!x || !w_packed || !scale || !shift || !anchors || !io || !head_params(d, p) || !conv_pw_decode_supported(p, na, no)) return RYOLO_EINVAL;
    if (arc < 0 || arc > 2 || !(stride > 0.f) || !(context_factor > 0.f)) return RYOLO_EINVAL;
    if (((uintptr_t)x | (uintptr_t)w_packed) & 15) return RYOLO_EINVAL;
    p.x = (const __bf16 *)x; p.w = (const __bf16 *)w_packed; p.scale = scale; p.shift = shift;
    return launch_conv_pw_decode(p, io, io_rows_per_image, io_row_offset, pout, anchors, na, no, stride, context_factor, arc, (hipStream_t)stream_);
}

int ryolo_conv_pair_supported(const ryolo_conv_desc *first, const ryolo_conv_desc *second, int shortcut_from_input) {
    if (validate(first) != RYOLO_OK || validate(second) != RYOLO_OK) return 0;
    return conv_stem_pair_kind(first, second, shortcut_from_input) != 0 ? 1 : 0;
}

int ryolo_conv2d_bn_act_pair(const ryolo_conv_desc *a, const ryolo_conv_desc *b, const void *x, const void *w_first, const float *scale_first,
                             const float *shift_first, const void *w_second, const float *scale_second, const float *shift_second,
                             int shortcut_from_input, void *y, void *stream_) {
    if (validate(a) != RYOLO_OK || validate(b) != RYOLO_OK || !x || !w_first || !scale_first || !shift_first || !w_second || !scale_second ||
        !shift_second || !y)
        return RYOLO_EINVAL;
    const int kind = conv_stem_pair_kind(a, b, shortcut_from_input);
    if (!kind) return RYOLO_EINVAL;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w_first | (uintptr_t)w_second) & 15) return RYOLO_EINVAL;
    ConvParams p;
    p.x = nullptr; p.w = (const __bf16 *)w_second; p.scale = scale_second; p.shift = shift_second; p.res = nullptr; p.y = (__bf16 *)y;
    p.N = b->N; p.H = b->H; p.W = b->W; p.Cin = b->Cin; p.in_cs = b->Cin; p.stride = b->stride; p.pad = b->pad;
    p.Ho = (b->H + 2 * b->pad - 3) / b->stride + 1;
    p.Wo = (b->W + 2 * b->pad - 3) / b->stride + 1;
    p.Cout = b->Cout; p.out_cs = b->out_cstride; p.res_cs = 0;
    p.K = 9 * b->Cin; p.Kpad = (p.K + BK - 1) / BK * BK;
    p.M = (int)((long long)b->N * p.Ho * p.Wo);
    p.act = b->act; p.slope = b->slope; p.ups = 1;
    p.nt_out = (long long)p.M * b->Cout * 2 >= nt_out_min_bytes() ? 1 : 0;
    p.stat_part = nullptr; p.stat_cpad = 0;
    const unsigned long long xb = (((unsigned long long)a->N * a->H * a->W - 1) * a->in_cstride + a->Cin) * 2ull;
    const int kpad_first = (a->ksize * a->ksize * a->Cin + BK - 1) / BK * BK;
    return launch_conv_stem_pair(kind, p, x, (unsigned)xb, a->in_cstride, a->H, a->W, w_first, kpad_first, scale_first, shift_first, a->act,
                                 a->slope, cu_count(), (hipStream_t)stream_);
}

int ryolo_conv2d_bn_act(const ryolo_conv_desc *d, const void *x, const void *w_packed, const float *scale,
                        const float *shift, const void *residual, void *y, void *stream_) {
    return ryolo_conv2d_bn_act_stats(d, x, w_packed, scale, shift, residual, y, nullptr, stream_);
}

int ryolo_conv0_recompute_supported(const ryolo_conv_desc *d) {
    if (validate(d) != RYOLO_OK) return 0;
    const long long M = (long long)d->N * d->H * d->W;
    return d->ksize == 3 && d->stride == 1 && d->pad == 1 && d->Cin == 8 && d->in_cstride == 8 && d->Cout == 32 && d->upsample == 1 &&
           !(d->tile & 0x1ff) && d->W >= 16 && M * 16 < 0x7fffff00ll && ((M - 1) * d->out_cstride + 32) * 2 < 0x7fffff00ll;
}

int ryolo_conv0_bn_act_fwd(const ryolo_conv_desc *d, const void *x, const void *w_packed, const float *scale, const float *shift,
                           int act, const float *slope, void *y, void *stream_) {
    if (!ryolo_conv0_recompute_supported(d) || !x || !w_packed || !scale || !shift || !y) return RYOLO_EINVAL;
    if (act < 0 || act > 2 || (act == RYOLO_ACT_LEAKY && !slope)) return RYOLO_EINVAL;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w_packed) & 15) return RYOLO_EINVAL;
    ConvParams p;
    p.x = (const __bf16 *)x; p.w = (const __bf16 *)w_packed; p.scale = scale; p.shift = shift; p.res = nullptr; p.y = (__bf16 *)y;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = 8; p.in_cs = 8; p.Ho = d->H; p.Wo = d->W; p.Cout = 32; p.out_cs = d->out_cstride; p.res_cs = 0;
    p.stride = 1; p.pad = 1; p.K = 72; p.Kpad = (72 + BK - 1) / BK * BK;
    p.M = (int)((long long)d->N * d->H * d->W);
    p.act = act; p.slope = 0.f; p.ups = 1; p.stat_part = nullptr; p.stat_cpad = 0;
    p.x_bytes = (unsigned)((unsigned long long)p.M * 16ull);
    p.w_bytes = (unsigned)((((unsigned long long)p.M - 1) * d->out_cstride + 32) * 2ull);
    p.nt_out = (long long)p.M * 64 >= nt_out_min_bytes() ? 1 : 0;
    C8Bwd bw;
    bw.slope = act == RYOLO_ACT_LEAKY ? slope : nullptr;
    bw.round_z = 1;
    const long long groups = ((long long)p.M + 15) / 16;
    const int gpw = 32;
    const unsigned nblk = (unsigned)(((groups + gpw - 1) / gpw + 3) / 4);
    hipStream_t stream = (hipStream_t)stream_;
    if (conv0_halo_on()) return launch_conv0_halo(p, bw.slope, 1, cu_count(), stream);
    if (act == RYOLO_ACT_LEAKY) hipLaunchKernelGGL((conv3x3_c8_direct_kernel<false, RYOLO_ACT_LEAKY, 0>), dim3(nblk), dim3(256), 0, stream, p, gpw, bw);
    else if (act == RYOLO_ACT_MISH) hipLaunchKernelGGL((conv3x3_c8_direct_kernel<false, RYOLO_ACT_MISH, 0>), dim3(nblk), dim3(256), 0, stream, p, gpw, bw);
    else hipLaunchKernelGGL((conv3x3_c8_direct_kernel<false, RYOLO_ACT_LINEAR, 0>), dim3(nblk), dim3(256), 0, stream, p, gpw, bw);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

size_t ryolo_conv0_bn_bwd_workspace_bytes(void) { return (size_t)STAT_ROWS * 96 * 8 + 64 * 4; }

int ryolo_conv0_bn_bwd(const ryolo_conv_desc *d, const void *x, const void *w_packed, const void *dy, int dy_cstride,
                       const float *scale, const float *shift, const float *mean, const float *invstd, int act, const float *slope,
                       void *dz, int dz_cstride, float *dgamma, float *dbeta, float *dslope, void *workspace, size_t workspace_bytes,
                       int workspace_is_zero, void *stream_) {
    if (!ryolo_conv0_recompute_supported(d) || !x || !w_packed || !dy || !scale || !shift || !mean || !invstd || !dz || !workspace)
        return RYOLO_EINVAL;
    if (workspace_bytes < ryolo_conv0_bn_bwd_workspace_bytes() || (dy_cstride & 7) || (dz_cstride & 7) || dy_cstride < 32 || dz_cstride < 32)
        return RYOLO_EINVAL;
    if (act < 0 || act > 2 || (act == RYOLO_ACT_LEAKY && !slope)) return RYOLO_EINVAL;
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dz | (uintptr_t)w_packed | (uintptr_t)workspace) & 15) return RYOLO_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    ConvParams p;
    p.x = (const __bf16 *)x; p.w = (const __bf16 *)w_packed; p.scale = scale; p.shift = shift; p.res = nullptr; p.y = (__bf16 *)dz;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = 8; p.in_cs = 8; p.Ho = d->H; p.Wo = d->W; p.Cout = 32; p.out_cs = dz_cstride; p.res_cs = 0;
    p.stride = 1; p.pad = 1; p.K = 72; p.Kpad = (72 + BK - 1) / BK * BK;
    p.M = (int)((long long)d->N * d->H * d->W);
    p.act = act; p.ups = 1; p.stat_part = nullptr; p.stat_cpad = 0;
    p.x_bytes = (unsigned)((unsigned long long)p.M * 16ull);
    p.w_bytes = (unsigned)((((unsigned long long)p.M - 1) * dz_cstride + 32) * 2ull);        // the kernel's second descriptor covers dz
    p.nt_out = (long long)p.M * 64 >= nt_out_min_bytes() ? 1 : 0;
    p.slope = 0.f;
    C8Bwd bw;
    bw.slope = act == RYOLO_ACT_LEAKY ? slope : nullptr;            // the learnable slope stays on the device
    bw.dy = (const __bf16 *)dy; bw.dy_cs = dy_cstride; bw.mean = mean;
    bw.dy_bytes = (unsigned)((((unsigned long long)p.M - 1) * dy_cstride + 32) * 2ull);
    double *part = (double *)workspace;
    float *kb = (float *)(part + (size_t)STAT_ROWS * 96), *kd = kb + 32;
    bw.part = part; bw.kb = kb; bw.kd = kd;
    if (!workspace_is_zero && hipMemsetAsync(part, 0, (size_t)STAT_ROWS * 96 * 8, stream) != hipSuccess) return RYOLO_ELAUNCH;
    const long long groups = ((long long)p.M + 15) / 16;
    const int gpw = 32;
    const unsigned nblk = (unsigned)(((groups + gpw - 1) / gpw + 3) / 4);
    int rc;
#define C8_BWD(MODE_)                                                                                                      \
    rc = act == RYOLO_ACT_LEAKY ? launch_c8_bwd<RYOLO_ACT_LEAKY>(p, gpw, nblk, bw, MODE_, stream)                          \
         : (act == RYOLO_ACT_MISH ? launch_c8_bwd<RYOLO_ACT_MISH>(p, gpw, nblk, bw, MODE_, stream)                          \
                                  : launch_c8_bwd<RYOLO_ACT_LINEAR>(p, gpw, nblk, bw, MODE_, stream));
    C8_BWD(2)
    if (rc != RYOLO_OK) return rc;
    hipLaunchKernelGGL(conv0_bwd_finalize_kernel, dim3(1), dim3(32), 0, stream, part, scale, mean, invstd, 1.0f / (float)p.M, kb, kd,
                       dgamma, dbeta, act == RYOLO_ACT_LEAKY ? dslope : nullptr);
    C8_BWD(3)
#undef C8_BWD
    return rc;
}

int ryolo_conv_kernel_choice(const ryolo_conv_desc *d, int with_residual, int with_statistics) {
    int choice = -1;
    g_conv_choice = &choice;
    void *fake = (void *)(uintptr_t)4096;      // never dereferenced: the dispatch returns before any launch
    const int rc = ryolo_conv2d_bn_act_stats(d, fake, fake, (const float *)fake, (const float *)fake, with_residual ? fake : nullptr, fake,
                                             with_statistics ? (double *)fake : nullptr, nullptr);
    g_conv_choice = nullptr;
    return rc == RYOLO_OK ? choice : -1;
}

static int bnreduce_plan(const ryolo_conv_desc *d, int *mode);

int ryolo_conv_dgrad_kernel_choice(const ryolo_conv_desc *d, int with_bn_reduce) {
    int choice = -1;
    void *fake = (void *)(uintptr_t)4096;      // never dereferenced: the dispatch returns before any launch
    BnRed br{};
    int mode = 0;
    if (with_bn_reduce) {                      // ryolo_conv2d_dgrad_bnreduce: the kernel its plan picks
        if (!bnreduce_plan(d, &mode)) return -1;
        g_bnred = &br;
        g_bnred_mode = mode;
    }
    g_conv_choice = &choice;
    const int rc = ryolo_conv2d_dgrad(d, fake, d ? d->Cout : 0, fake, (const float *)fake, (const float *)fake, fake, 1, nullptr);
    g_conv_choice = nullptr;
    g_bnred = nullptr;
    return rc == RYOLO_OK ? choice : -1;
}

// ------------------------------------------------------------------------------------------------ dgrad
// dx[n, hi, wi, ci] (+)= sum_{kh,kw,co} dz[n, ho, wo, co] * W[co, ci, kh, kw],  ho*s - pad + kh = hi (same for w).
// stride 1: a plain convolution of dz with the spatially flipped, channel-transposed filter.
// stride 2 (3x3, pad 1): four output-parity classes (hi%2, wi%2) = (a, b); class (a, b) only sees the taps with
// kh = a+1 (mod 2), kw = b+1 (mod 2) -> 1, 2, 2 or 4 taps, each a stride-1 gather dz[(hi+1-kh)/2, (wi+1-kw)/2];
// each class is one launch of the same kernel with its own tap list, packed filter and strided output placement.
static int dgrad_classes(int ks, int stride, int pad, int cls, int *dy, int *dx, int *khs, int *kws) {
    // returns ntaps of class `cls` (stride 1: cls must be 0); tap t reads dz pixel (i + dy[t], j + dx[t]) * of the class grid
    if (stride == 1) {
        int n = 0;
        for (int kh = 0; kh < ks; kh++)
            for (int kw = 0; kw < ks; kw++) {   // dz pixel = hi + pad - kh' where kh' runs over the flipped filter
                dy[n] = kh; dx[n] = kw; khs[n] = ks - 1 - kh; kws[n] = ks - 1 - kw;
                n++;
            }
        return n;
    }
    const int a = cls >> 1, b = cls & 1;
    int n = 0;
    for (int kh = ks - 1; kh >= 0; kh--) {
        if (((a + pad - kh) & 1) != 0) continue;
        for (int kw = ks - 1; kw >= 0; kw--) {
            if (((b + pad - kw) & 1) != 0) continue;
            // hi = 2i + a -> ho = (2i + a + pad - kh) / 2 = i + (a + pad - kh) / 2
            dy[n] = (a + pad - kh) / 2; dx[n] = (b + pad - kw) / 2; khs[n] = kh; kws[n] = kw;
            n++;
        }
    }
    return n;
}

// x-fused stride-2 classes (3x3, pad 1, C_in a multiple of 32 and <= 64 -- the stem, where one input pixel is only 64-128 B):
// the two column parities of a row parity `a` become ONE launch whose output "pixel" (i, j) is the pair of input pixels
// (2i+a, 2j), (2i+a, 2j+1) = 2*C_in contiguous channels, so a store writes whole lines and dz is read twice, not four times.
// Taps (kh, dxo in {0,1}) read dz pixel (i + dy, j + dxo); the weight of output half b at a tap is the filter column
// b == 0 ? (dxo == 0 ? 1 : none) : (dxo == 0 ? 2 : 0).
static inline bool dgrad_xfusable(int Cout, int Cin, int ksize, int stride) {
    (void)Cout;
    return stride == 2 && ksize == 3 && Cin <= 64 && Cin % 32 == 0;
}
static int dgrad_xfused_class(int a, int *dy, int *dx, int *khs, int *kw0, int *kw1) {
    int n = 0;
    for (int kh = 2; kh >= 0; kh--) {
        if (((a + 1 - kh) & 1) != 0) continue;
        for (int dxo = 0; dxo < 2; dxo++) {
            dy[n] = (a + 1 - kh) / 2; dx[n] = dxo; khs[n] = kh;
            kw0[n] = dxo == 0 ? 1 : -1;
            kw1[n] = dxo == 0 ? 2 : 0;
            n++;
        }
    }
    return n;
}
static size_t dgrad_classic_bytes(int Cout, int Cin, int ksize, int stride) {
    const size_t rows = ((size_t)Cin + 127) / 128 * 128;
    size_t total = 0;
    int dy[9], dx[9], khs[9], kws[9];
    for (int cls = 0; cls < (stride == 1 ? 1 : 4); cls++) {
        const int nt = dgrad_classes(ksize, stride, (ksize - 1) / 2, cls, dy, dx, khs, kws);
        const size_t Kpad = ((size_t)nt * Cout + BK - 1) / BK * BK;
        total += (rows * Kpad + 128) * 2;
    }
    return total;
}

size_t ryolo_conv_packed_dgrad_bytes(int Cout, int Cin, int ksize, int stride) {
    if (Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return 0;
    size_t total = dgrad_classic_bytes(Cout, Cin, ksize, stride);
    if (dgrad_xfusable(Cout, Cin, ksize, stride)) {       // the x-fused images follow the four classic ones
        const size_t rows = ((size_t)2 * Cin + 127) / 128 * 128;
        int dy[9], dx[9], khs[9], k0[9], k1[9];
        for (int a = 0; a < 2; a++) {
            const int nt = dgrad_xfused_class(a, dy, dx, khs, k0, k1);
            const size_t Kpad = ((size_t)nt * Cout + BK - 1) / BK * BK;
            total += (rows * Kpad + 128) * 2;
        }
    }
    return total;
}

struct XfTaps { int khs[9], kw0[9], kw1[9]; };
__global__ void pack_dgrad_xfused_kernel(const float *__restrict__ w, int Cout, int Cin, int ntaps, XfTaps tp, int Kpad, int rows,
                                         __bf16 *__restrict__ out) {
    // out[b*Cin + ci][t*Cout + co] = w[co][ci][kh_t][kw_{b,t}] (0 where the half has no column at that tap)
    const size_t total = (size_t)rows * Kpad + 128;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < (size_t)rows * Kpad) {
            const int rr = (int)(i / Kpad), k = (int)(i % Kpad);
            const int t = k / Cout, co = k % Cout;
            if (rr < 2 * Cin && t < ntaps) {
                const int b = rr / Cin, ci = rr - b * Cin;
                const int kw = b ? tp.kw1[t] : tp.kw0[t];
                if (kw >= 0) v = w[(((size_t)co * Cin + ci) * 3 + tp.khs[t]) * 3 + kw];
            }
        }
        out[i] = (__bf16)v;
    }
}

__global__ void pack_dgrad_kernel(const float *__restrict__ w, int Cout, int Cin, int KS, int ntaps, const int *khs_kws,
                                  int Kpad, int rows, __bf16 *__restrict__ out) {
    // out[ci][t*Cout + co] = w[co][ci][kh_t][kw_t]
    const size_t total = (size_t)rows * Kpad + 128;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < (size_t)rows * Kpad) {
            const int ci = (int)(i / Kpad), k = (int)(i % Kpad);
            const int t = k / Cout, co = k % Cout;
            if (ci < Cin && t < ntaps) v = w[(((size_t)co * Cin + ci) * KS + khs_kws[t]) * KS + khs_kws[9 + t]];
        }
        out[i] = (__bf16)v;
    }
}

// ---- all weight packs of a training step in ONE launch (forward layout + every dgrad class of every conv) ----------
// 160 small launches per step otherwise (1.1 ms of launch-bound time at bs 32).  `jobs` is a device array built once by
// the caller with ryolo_conv_pack_job_fill; workgroup b serves the job whose [block_begin, block_end) contains b.
constexpr int PK_ROWS = 4;            // forward layout: output rows (c_out) per workgroup
constexpr int PK_CI = 32, PK_CO = 64;  // dgrad layout: (c_in rows) x (c_out columns) per workgroup
constexpr int PK_PAD = 2;             // dgrad sub-block: bf16 elements added to each c_out's run in LDS.  The transposing read walks c_out
                                      // across the lanes; 288 elements = 144 words per run put 64 lanes on 4 banks (16-way conflict),
                                      // 145 words spread them over all 64
constexpr int PK_UB = 6;              // 16-B loads a thread keeps in flight while it stages a tile (one per trip left the pass latency-bound)
constexpr int PK_LDS = PK_CO * (PK_CI * 9 + PK_PAD);   // bf16 elements: one [64 c_out][32 c_in][9 taps] sub-block, or one forward row
__global__ void __launch_bounds__(256) pack_batch_kernel(const ryolo_pack_job *__restrict__ jobs, int njobs) {
    // Both layouts are transposes of the OIHW parameter, so each workgroup moves a TILE through LDS: it reads the fp32
    // source in its own order (whole [c_in][taps] rows / 32-channel runs of them: coalesced) and writes bf16 runs that are
    // contiguous in the destination.  The element-wise gather this replaces read the 250 MB of weights through 36-byte
    // (forward) and 4.6-KB (dgrad) strides: 0.92 ms per step.
    __shared__ __bf16 sm[PK_LDS];
    int lo = 0;                             // last job with block_begin <= blockIdx.x (block_begin ascending, jobs[0] starts at 0)
    if (njobs <= 256) {                     // one round of parallel loads + a count, not eight dependent loads of a binary search
        lo = __syncthreads_count((int)threadIdx.x < njobs && jobs[threadIdx.x].block_begin <= (int)blockIdx.x) - 1;
    } else {
        int hi = njobs - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
    }
    const ryolo_pack_job j = jobs[lo];
    const int blk = (int)blockIdx.x - j.block_begin;
    const float *__restrict__ w = (const float *)j.src;
    __bf16 *__restrict__ out = (__bf16 *)j.dst;
    const int KK = j.KS * j.KS;
    const int tid = threadIdx.x;
    if (blk == 0)                           // the 128-element zero page behind the body
        for (int i = tid; i < 128; i += 256) out[(size_t)j.rows * j.Kpad + i] = (__bf16)0.f;
    if (j.kind == 0) {                      // forward: out[co][tap*Cin_pad + c] = w[co][c][tap]
        const int rowlen = j.Cin * KK;
        const int rl4 = (rowlen + 3) & ~3;                       // a row's slot in LDS (8-B aligned)
        const int fit = PK_LDS / rl4 < PK_ROWS ? (PK_LDS / rl4 < 1 ? 1 : PK_LDS / rl4) : PK_ROWS;   // rows staged together: one load
        for (int rr0 = 0; rr0 < PK_ROWS; rr0 += fit) {           // round and one barrier pair for all of them when they fit
            const int rbase = blk * PK_ROWS + rr0;
            if (rbase >= j.rows) break;
            for (int rr = 0; rr < fit && rr0 + rr < PK_ROWS; rr++) {
                const int r = rbase + rr;
                if (r >= j.Cout) break;                          // (rows past C_out are zero rows: nothing to stage)
                __bf16 *dst = sm + rr * rl4;
                if ((rowlen & 3) == 0 && ((uintptr_t)w & 15) == 0) {    // 16-B loads (a row starts at a multiple of 4 floats then, and the parameter itself at a
                                                                        // 16-B boundary: a view into a flat buffer may not -- those take the scalar path)
                    const float4 *w4 = (const float4 *)(w + (size_t)r * rowlen);
                    for (int i0 = tid; i0 < rowlen / 4; i0 += 256 * PK_UB) {     // PK_UB independent loads in flight per thread
                        float4 v[PK_UB];
#pragma unroll
                        for (int u = 0; u < PK_UB; u++)
                            if (i0 + u * 256 < rowlen / 4) v[u] = w4[i0 + u * 256];
#pragma unroll
                        for (int u = 0; u < PK_UB; u++) {
                            if (i0 + u * 256 >= rowlen / 4) break;
                            bf16x4 o;
                            o[0] = (__bf16)v[u].x; o[1] = (__bf16)v[u].y; o[2] = (__bf16)v[u].z; o[3] = (__bf16)v[u].w;
                            *(bf16x4 *)(dst + 4 * (i0 + u * 256)) = o;
                        }
                    }
                } else {
                    for (int i = tid; i < rowlen; i += 256) dst[i] = (__bf16)w[(size_t)r * rowlen + i];
                }
            }
            __syncthreads();
            for (int rr = 0; rr < fit && rr0 + rr < PK_ROWS; rr++) {
                const int r = rbase + rr;
                if (r >= j.rows) break;
                const bool real = r < j.Cout;
                const __bf16 *src = sm + rr * rl4;
                if ((j.Cin_pad & 7) == 0) {     // 8 consecutive k share a tap: one 16-B store per thread and trip (Kpad % 64 == 0)
                    for (int k8 = tid * 8; k8 < j.Kpad; k8 += 256 * 8) {
                        const int tap = k8 / j.Cin_pad, c0 = k8 - tap * j.Cin_pad;
                        bf16x8 v;
#pragma unroll
                        for (int e = 0; e < 8; e++) v[e] = (real && tap < KK && c0 + e < j.Cin) ? src[(c0 + e) * KK + tap] : (__bf16)0.f;
                        *(bf16x8 *)(out + (size_t)r * j.Kpad + k8) = v;
                    }
                } else {
                    for (int k = tid; k < j.Kpad; k += 256) {
                        const int tap = k / j.Cin_pad, c = k - tap * j.Cin_pad;
                        out[(size_t)r * j.Kpad + k] = (real && tap < KK && c < j.Cin) ? src[c * KK + tap] : (__bf16)0.f;
                    }
                }
            }
            __syncthreads();
        }
        return;
    }
    // dgrad class: out[ci][t*Cout + co] = w[co][ci][kh_t][kw_t]; kind 2 (x-fused stride-2 class): rows are (b, ci), the
    // filter column of tap t is nibble b of kws[t] minus 1 (-1: this half has no column there)
    const int cob = (j.Cout + PK_CO - 1) / PK_CO;
    const int row0 = (blk / cob) * PK_CI, co0 = (blk % cob) * PK_CO;
    const int half = (j.kind == 2 && row0 >= j.Cin) ? 1 : 0;
    const int ci0 = j.kind == 2 ? row0 - half * j.Cin : row0;
    const bool live = j.kind != 2 || row0 < 2 * j.Cin;
    const int run = PK_CI * KK;             // one c_out's share of the sub-block: 32 c_in x taps, contiguous in w
    const int pitch = run + PK_PAD;
    if ((run & 3) == 0 && (j.Cin & 3) == 0 && ci0 + PK_CI <= j.Cin && ((uintptr_t)w & 15) == 0) {
        // whole 32-channel runs: 16-B loads (a run starts at (co * Cin + ci0) * KK floats, a multiple of 4); the LDS pitch is odd in
        // words, so the four bf16 go out as two 4-B stores
        static_assert((PK_CI * 9 + PK_PAD) % 2 == 0, "4-B aligned runs in LDS");
        const int run4 = run / 4;
        for (int i0 = tid; i0 < PK_CO * run4; i0 += 256 * PK_UB) {       // PK_UB independent loads in flight per thread
            float4 v[PK_UB];
#pragma unroll
            for (int u = 0; u < PK_UB; u++) {
                const int i = i0 + u * 256;
                const int col = i / run4, q = i - col * run4;
                v[u] = float4{0.f, 0.f, 0.f, 0.f};
                if (i < PK_CO * run4 && live && co0 + col < j.Cout) v[u] = *(const float4 *)(w + ((size_t)(co0 + col) * j.Cin + ci0) * KK + 4 * q);
            }
#pragma unroll
            for (int u = 0; u < PK_UB; u++) {
                const int i = i0 + u * 256;
                if (i >= PK_CO * run4) break;
                const int col = i / run4, q = i - col * run4;
                bf16x2 lo, hi;
                lo[0] = (__bf16)v[u].x; lo[1] = (__bf16)v[u].y; hi[0] = (__bf16)v[u].z; hi[1] = (__bf16)v[u].w;
                *(bf16x2 *)(sm + col * pitch + 4 * q) = lo;
                *(bf16x2 *)(sm + col * pitch + 4 * q + 2) = hi;
            }
        }
    } else {
        for (int i = tid; i < PK_CO * run; i += 256) {
            const int col = i / run, rem = i - col * run;
            const int co = co0 + col, ci = ci0 + rem / KK;
            sm[col * pitch + rem] = (live && co < j.Cout && ci < j.Cin) ? (__bf16)w[((size_t)co * j.Cin + ci0) * KK + rem] : (__bf16)0.f;
        }
    }
    __syncthreads();
    if ((j.Cout & 7) == 0) {                // 8 consecutive c_out per thread: 16-B stores (every row offset is a multiple of 8 elements)
        for (int i = tid; i < PK_CI * j.ntaps * (PK_CO / 8); i += 256) {
            const int col = (i % (PK_CO / 8)) * 8, t = (i / (PK_CO / 8)) % j.ntaps, cil = i / ((PK_CO / 8) * j.ntaps);
            if (co0 + col < j.Cout) {
                const int kw = j.kind == 2 ? ((j.kws[t] >> (4 * half)) & 15) - 1 : j.kws[t];
                const int src = cil * KK + j.khs[t] * j.KS + kw;
                bf16x8 v;
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = kw >= 0 ? sm[(col + e) * pitch + src] : (__bf16)0.f;
                *(bf16x8 *)(out + (size_t)(row0 + cil) * j.Kpad + t * j.Cout + co0 + col) = v;
            }
        }
    } else {
        for (int i = tid; i < PK_CI * j.ntaps * PK_CO; i += 256) {
            const int col = i % PK_CO, t = (i / PK_CO) % j.ntaps, cil = i / (PK_CO * j.ntaps);
            if (co0 + col < j.Cout) {
                const int kw = j.kind == 2 ? ((j.kws[t] >> (4 * half)) & 15) - 1 : j.kws[t];
                out[(size_t)(row0 + cil) * j.Kpad + t * j.Cout + co0 + col] =
                    kw >= 0 ? sm[col * pitch + cil * KK + j.khs[t] * j.KS + kw] : (__bf16)0.f;
            }
        }
    }
    if (co0 == 0) {                         // K padding behind the last tap of these 32 rows
        const int kreal = j.ntaps * j.Cout, padn = j.Kpad - kreal;
        for (int i = tid; i < PK_CI * padn; i += 256) out[(size_t)(row0 + i / padn) * j.Kpad + kreal + i % padn] = (__bf16)0.f;
    }
}

int ryolo_conv_pack_job_fill(ryolo_pack_job *host_jobs /* room for 7 */, const float *w_oihw, int Cout, int Cin, int ksize,
                             int stride, int Cin_pad, void *packed_fwd, void *packed_dgrad /* or NULL */) {
    if (!host_jobs || !w_oihw || !packed_fwd || Cout <= 0 || Cin <= 0 || Cin_pad < Cin || (ksize != 1 && ksize != 3) ||
        (stride != 1 && stride != 2))
        return -1;
    int n = 0;
    {
        ryolo_pack_job &j = host_jobs[n++];
        j = ryolo_pack_job{};
        j.src = w_oihw; j.dst = packed_fwd; j.kind = 0; j.Cout = Cout; j.Cin = Cin; j.KS = ksize; j.Cin_pad = Cin_pad;
        j.Kpad = (ksize * ksize * Cin_pad + BK - 1) / BK * BK;
        j.rows = (Cout + 127) / 128 * 128;
        j.block_begin = 0; j.block_end = (j.rows + PK_ROWS - 1) / PK_ROWS;      // workgroups of pack_batch_kernel
        if (Cin * ksize * ksize > PK_LDS) return -1;
    }
    if (packed_dgrad) {
        if (ryolo_conv_packed_dgrad_bytes(Cout, Cin, ksize, stride) == 0) return -1;
        char *dst = (char *)packed_dgrad;
        for (int cls = 0; cls < (stride == 1 ? 1 : 4); cls++) {
            int dy[9], dx[9], khs[9], kws[9];
            const int nt = dgrad_classes(ksize, stride, (ksize - 1) / 2, cls, dy, dx, khs, kws);
            ryolo_pack_job &j = host_jobs[n++];
            j = ryolo_pack_job{};
            j.src = w_oihw; j.dst = dst; j.kind = 1; j.Cout = Cout; j.Cin = Cin; j.KS = ksize; j.ntaps = nt;
            for (int t = 0; t < nt; t++) { j.khs[t] = khs[t]; j.kws[t] = kws[t]; }
            j.Kpad = (nt * Cout + BK - 1) / BK * BK;
            j.rows = (Cin + 127) / 128 * 128;
            j.block_begin = 0; j.block_end = (j.rows / PK_CI) * ((Cout + PK_CO - 1) / PK_CO);
            dst += ((size_t)j.rows * j.Kpad + 128) * 2;
        }
        if (dgrad_xfusable(Cout, Cin, ksize, stride)) {
            for (int a = 0; a < 2; a++) {
                int dy[9], dx[9], khs[9], k0[9], k1[9];
                const int nt = dgrad_xfused_class(a, dy, dx, khs, k0, k1);
                ryolo_pack_job &j = host_jobs[n++];
                j = ryolo_pack_job{};
                j.src = w_oihw; j.dst = dst; j.kind = 2; j.Cout = Cout; j.Cin = Cin; j.KS = ksize; j.ntaps = nt;
                for (int t = 0; t < nt; t++) { j.khs[t] = khs[t]; j.kws[t] = (k0[t] + 1) | ((k1[t] + 1) << 4); }
                j.Kpad = (nt * Cout + BK - 1) / BK * BK;
                j.rows = (2 * Cin + 127) / 128 * 128;
                j.block_begin = 0; j.block_end = (j.rows / PK_CI) * ((Cout + PK_CO - 1) / PK_CO);
                dst += ((size_t)j.rows * j.Kpad + 128) * 2;
            }
        }
    }
    return n;
}

int ryolo_conv_pack_batch(const ryolo_pack_job *device_jobs, int njobs, int total_blocks, void *stream) {
    if (!device_jobs || njobs <= 0 || total_blocks <= 0) return RYOLO_EINVAL;
    hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, device_jobs, njobs);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

int ryolo_conv_dgrad_tap_table(int ksize, int stride, int *host_out /* int[72] */) {
    if (!host_out || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return RYOLO_EINVAL;
    for (int i = 0; i < 72; i++) host_out[i] = 0;
    for (int cls = 0; cls < (stride == 1 ? 1 : 4); cls++) {
        int dy[9], dx[9];
        dgrad_classes(ksize, stride, (ksize - 1) / 2, cls, dy, dx, host_out + cls * 18, host_out + cls * 18 + 9);
    }
    return RYOLO_OK;
}

int ryolo_conv_pack_weights_dgrad(const float *w_oihw, int Cout, int Cin, int ksize, int stride, void *packed,
                                  const int *taps_table /* device int[72] from ryolo_conv_dgrad_tap_table */, void *stream_) {
    if (!w_oihw || !packed || !taps_table || ryolo_conv_packed_dgrad_bytes(Cout, Cin, ksize, stride) == 0) return RYOLO_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    const int rows = (Cin + 127) / 128 * 128;
    char *dst = (char *)packed;
    for (int cls = 0; cls < (stride == 1 ? 1 : 4); cls++) {
        int dy[9], dx[9], kk[18];
        const int nt = dgrad_classes(ksize, stride, (ksize - 1) / 2, cls, dy, dx, kk, kk + 9);
        const int Kpad = (nt * Cout + BK - 1) / BK * BK;
        const size_t total = (size_t)rows * Kpad + 128;
        const int nb = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(pack_dgrad_kernel, dim3(nb), dim3(256), 0, stream, w_oihw, Cout, Cin, ksize, nt,
                           taps_table + cls * 18, Kpad, rows, (__bf16 *)dst);
        dst += total * 2;
    }
    if (dgrad_xfusable(Cout, Cin, ksize, stride)) {
        const int rows2 = (2 * Cin + 127) / 128 * 128;
        for (int a = 0; a < 2; a++) {
            int dy[9], dx[9];
            XfTaps tp;
            const int nt = dgrad_xfused_class(a, dy, dx, tp.khs, tp.kw0, tp.kw1);
            for (int t = nt; t < 9; t++) { tp.khs[t] = 0; tp.kw0[t] = -1; tp.kw1[t] = -1; }
            const int Kpad = (nt * Cout + BK - 1) / BK * BK;
            const size_t total = (size_t)rows2 * Kpad + 128;
            const int nb = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
            hipLaunchKernelGGL(pack_dgrad_xfused_kernel, dim3(nb), dim3(256), 0, stream, w_oihw, Cout, Cin, nt, tp, Kpad, rows2,
                               (__bf16 *)dst);
            dst += total * 2;
        }
    }
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

// rows of partial sums (= workgroups of the persistent grid) the fused launch writes, 0 when this conv's data gradient cannot carry
// the reduce: 1x1 stride 1, whole 128-channel tiles on both sides, dense input gradient, a tile list deep enough for the
// persistent kernel (the same tests dispatch() / launch_variant() apply)
// rows of partial sums and which kernel carries the reduce.  conv_pw.hip where it is the faster data gradient (K = 128: the 76^2
// residual blocks, 127 vs 133 us at bs 64) and where the persistent 128 x 128 tile does not apply (tile lists less than 2.5 rounds
// deep: 19^2); the persistent tile elsewhere (38^2: 68 vs 70 us, 19^2 K 512: 49 vs 54 us; tools/pw_bench.py --train).
// Mode 3 of the plan below: the stride-1 data gradients whose natural kernel is one of the one-tile-per-workgroup / narrow tiles (the
// 3x3 layers with C_in <= 128, the 1x1 layers with C_in <= 64).  Two dry runs of the dispatch decide: the kernel the plain data gradient
// takes must be one of those tiles (a layer that conv_mq / conv_mp serve keeps them: the reduce is not worth a slower conv), and the launch
// with the reduce must exist.  Returns the rows (= pixel tiles) or 0.
static int bnreduce_plan_tiles(const ryolo_conv_desc *d) {
    if (d->stride != 1 || (d->tile & 0xff) || d->in_cstride != d->Cin || (d->Cin & 7)) return 0;
    int knob = 1;
    {   // RYOLO_BN_REDUCE_TILES = 0: off (A/B timing; ryolo_set_tuning)
        const char *e = tune(TUNE_BN_REDUCE_TILES);
        if (e) knob = atoi(e);
        if (knob == 0) return 0;
    }
    void *fake = (void *)(uintptr_t)4096;      // never dereferenced: the dispatch returns before any launch
    const BnRed *sv_b = g_bnred;
    const int sv_m = g_bnred_mode;
    int *sv_c = g_conv_choice;
    BnRed br{};
    int natural = -1, fused = -1;
    g_bnred = nullptr;
    g_conv_choice = &natural;
    int rc = ryolo_conv2d_dgrad(d, fake, d->Cout, fake, (const float *)fake, (const float *)fake, fake, 1, nullptr);
    if (rc == RYOLO_OK) {
        g_bnred = &br;
        g_bnred_mode = 3;
        g_conv_choice = &fused;
        rc = ryolo_conv2d_dgrad(d, fake, d->Cout, fake, (const float *)fake, (const float *)fake, fake, 1, nullptr);
    }
    g_bnred = sv_b;
    g_bnred_mode = sv_m;
    g_conv_choice = sv_c;
    if (rc != RYOLO_OK || natural != fused) return 0;
    const int tile = fused - RYOLO_CONV_KERNEL_IGEMM;
    const int bm = tile == 1 ? 128 : ((tile == 2 || tile == 3) ? 256 : 0);
    if (!bm) return 0;
    if (tile == 3) return 0;           // 256 x 32: the plain data gradient runs on the persistent grid (short K), 233 us faster than this launch
    if (tile == 1 && knob == 2) return 0;
    const long long M = (long long)d->N * d->H * d->W;
    return (int)((M + bm - 1) / bm);
}

// mode 4: the data gradient's natural kernel is one of conv_mq.hip's 128-channel tiles -- the reduce rides in its epilogue, one row per
// workgroup.  Returns the rows (= workgroups of that launch) or 0.
static int bnreduce_plan_mq128(const ryolo_conv_desc *d) {
    if (d->stride != 1 || (d->tile & 0xff) || d->in_cstride != d->Cin || (d->Cin & 127) || mq128_knob() <= 0) return 0;
    void *fake = (void *)(uintptr_t)4096;      // never dereferenced: the dispatch returns before any launch
    const BnRed *sv_b = g_bnred;
    int *sv_c = g_conv_choice;
    int natural = -1;
    g_bnred = nullptr;
    g_conv_choice = &natural;
    const int rc = ryolo_conv2d_dgrad(d, fake, d->Cout, fake, (const float *)fake, (const float *)fake, fake, 1, nullptr);
    g_bnred = sv_b;
    g_conv_choice = sv_c;
    if (rc != RYOLO_OK || (natural != RYOLO_CONV_KERNEL_MQ128 && natural != RYOLO_CONV_KERNEL_MQ64)) return 0;
    ConvParams q;
    q.M = (int)((long long)d->N * d->H * d->W);
    q.Cout = d->Cin;
    return conv_mq128_grid(q, natural == RYOLO_CONV_KERNEL_MQ128 ? 128 : 64);
}

static int bnreduce_plan(const ryolo_conv_desc *d, int *mode) {
    *mode = 0;
    if (validate(d) != RYOLO_OK) return 0;
    {
        const int r = bnreduce_plan_mq128(d);
        if (r > 0) {
            *mode = 4;
            return r;
        }
    }
    auto tiles = [&]() {
        const int r = bnreduce_plan_tiles(d);
        if (r > 0) *mode = 3;
        return r;
    };
    if (d->ksize != 1 || d->stride != 1 || d->pad != 0) return tiles();
    if ((d->Cin & 127) || (d->Cout & 63) || (d->tile & 0xff)) return tiles();
    const long long M = (long long)d->N * d->H * d->W;
    int g_pw = 0;
    {   // the data gradient as conv_pw.hip sees it (K = the forward's C_out, channels = its C_in): its grid when it serves the shape
        ConvParams q;
        q.Cin = d->Cout; q.Kpad = (d->Cout + BK - 1) / BK * BK; q.Cout = d->Cin; q.pw_grid_cap = 0;
        const int g = conv_pw_disabled() ? 0 : conv_pw_grid(q);
        if (g > 0 && d->in_cstride == d->Cin && ((unsigned long long)(M + 1024 * 128) * d->Cout) * 2ull < 0x7fffff00ull &&
            ((unsigned long long)M * d->Cin) * 2ull < 0x7fffff00ull)
            g_pw = g;
    }
    const long long mt = (M + 127) / 128, nt = d->Cin / 128, T = mt * nt;
    const int grid = (2 * cu_count()) & ~7;
    const long long dmax = d->W > d->H ? d->W : d->H;
    const bool persist_ok = !(grid < 8 || 2 * T < 5 * (long long)grid || mt * 128 * dmax >= 0x100000000ll || T * nt >= 0x100000000ll) &&
                            ((unsigned long long)M * d->Cout) * 2ull < 0x7fffff00ull;
    if (g_pw > 0 && (d->Cout == 128 || !persist_ok)) {
        *mode = 1;
        return g_pw;
    }
    if (persist_ok) {
        *mode = 2;
        return grid;
    }
    return tiles();
}

// rows of partial sums the fused launch writes (= workgroups of a persistent launch, pixel tiles of a one-tile-per-workgroup launch), 0 when
// this conv's data gradient cannot carry the reduce: stride 1, dense input gradient, and one of the kernels with the folded pass (the
// persistent 2x2 tile / conv_pw.hip for 1x1 layers with whole 128-channel tiles, the narrow and the 128 x 128 one-tile kernels otherwise)
int ryolo_conv2d_dgrad_bnreduce_rows(const ryolo_conv_desc *d) {
    int mode;
    return bnreduce_plan(d, &mode);
}

int ryolo_conv2d_dgrad_bnreduce(const ryolo_conv_desc *d, const void *dz, int dz_cstride, const void *packed_dgrad, const float *ones,
                                const float *zeros, void *dx, int accumulate, const void *z, int z_cstride, const float *scale,
                                const float *shift, const float *mean, const float *invstd, const float *slope, float *part,
                                void *stream_) {
    int mode = 0;
    if (!bnreduce_plan(d, &mode) || !z || (z_cstride & 7) || z_cstride < d->Cin || !scale || !shift || !mean || !invstd ||
        !slope || !part || d->in_cstride != d->Cin)
        return RYOLO_EINVAL;
    BnRed br;
    br.z = (const __bf16 *)z; br.z_cs = z_cstride; br.scale = scale; br.shift = shift; br.mean = mean; br.invstd = invstd;
    br.slope = slope; br.part = part;
    g_bnred = &br;
    g_bnred_mode = mode;
    const int rc = ryolo_conv2d_dgrad(d, dz, dz_cstride, packed_dgrad, ones, zeros, dx, accumulate, stream_);
    g_bnred = nullptr;
    return rc;
}

int ryolo_conv2d_dgrad(const ryolo_conv_desc *d /* the FORWARD conv */, const void *dz, int dz_cstride,
                       const void *packed_dgrad, const float *ones, const float *zeros, void *dx, int accumulate,
                       void *stream_) {
    if (validate(d) != RYOLO_OK || !dz || !packed_dgrad || !ones || !zeros || !dx) return RYOLO_EINVAL;
    if (d->stride != 1 && !(d->stride == 2 && d->ksize == 3 && d->pad == 1)) return RYOLO_EINVAL;
    if ((dz_cstride & 7) || dz_cstride < d->Cout) return RYOLO_EINVAL;
    const int Ho = (d->H + 2 * d->pad - d->ksize) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->ksize) / d->stride + 1;
    const int rows = (d->Cin + 127) / 128 * 128;
    const char *wsrc = (const char *)packed_dgrad;
    // x-fused stride-2 classes when the input gradient is dense and its rows hold an even number of pixels (tile bit 0x8000
    // forces the four classic classes: tests / A-B)
    const bool xfuse = dgrad_xfusable(d->Cout, d->Cin, d->ksize, d->stride) && d->pad == 1 && (d->W & 1) == 0 &&
                       d->in_cstride == d->Cin && !(d->tile & 0x8000);
    if (xfuse) wsrc += dgrad_classic_bytes(d->Cout, d->Cin, d->ksize, d->stride);
    // Darknet-53 layers 1 and 3 (3x3, 32 -> 64, stride 2 / 1): one persistent launch with the dz patch staged once and the filter in
    // registers (conv_stem.hip); tile bit 0x8000 (the classic classes) and RYOLO_STEM_DGRAD=0 keep the implicit-GEMM launches (tests, A/B)
    // (the stride-1 64 -> 128 layers keep the 256 x 64 tile: it carries the folded BatchNorm reduce, and the channel-split kernel measured
    // 286 us against its 278 / 332 with the reduce -- step 48.50 vs 48.44 ms)
    if (!g_bnred && d->ksize == 3 && d->pad == 1 && ((d->Cin == 32 && d->Cout == 64) || (d->Cin == 64 && d->Cout == 128 && d->stride == 2)) &&
        !(d->tile & 0x80ff)) {
        const char *e = tune(TUNE_STEM_DGRAD);
        const int knob = e ? atoi(e) : 3;              // bit 0: the 64 -> 32 kernels, bit 1: the 128 -> 64 one (A/B; ryolo_set_tuning)
        if (knob & (d->Cout == 64 ? 1 : 2)) {
            const int nt_out = (long long)d->N * d->H * d->W * d->Cin * 2 >= nt_out_min_bytes() ? 1 : 0;
            // (its size guards -- dz of 2 GiB and more, tile counts beyond 2^31 -- answer EINVAL before anything is enqueued: those launches
            // fall through to the parity-class launches below, which served them before this kernel existed)
            const unsigned long long dzb = (((unsigned long long)d->N * Ho * Wo - 1) * dz_cstride + d->Cout) * 2ull;
            const unsigned long long dxb = (((unsigned long long)d->N * d->H * d->W - 1) * d->in_cstride + d->Cin) * 2ull;
            if (dzb < 0x7fffff00ull && dxb < 0x7fffff00ull) {
                RYOLO_CONV_DRY_RUN(RYOLO_CONV_KERNEL_STEM_DGRAD);
                const int rc = launch_conv_stem_dgrad(d->Cout, d->stride, dz, dz_cstride, packed_dgrad, dx, d->in_cstride, accumulate, d->N, d->H,
                                                      d->W, nt_out, cu_count(), (hipStream_t)stream_);
                if (rc != RYOLO_EINVAL) return rc;
            }
        }
    }
    const int ncls = d->stride == 1 ? 1 : (xfuse ? 2 : 4);
    const int wrows = xfuse ? (2 * d->Cin + 127) / 128 * 128 : rows;
    for (int cls = 0; cls < ncls; cls++) {
        int dy[9], dxx[9], khs[9], kws[9];
        const int nt = xfuse ? dgrad_xfused_class(cls, dy, dxx, khs, kws, kws)
                             : dgrad_classes(d->ksize, d->stride, d->pad, cls, dy, dxx, khs, kws);
        ConvParams p;
        p.x = (const __bf16 *)dz;
        p.w = (const __bf16 *)wsrc;
        p.scale = ones; p.shift = zeros;
        p.res = accumulate ? (const __bf16 *)dx : nullptr;
        p.y = (__bf16 *)dx;
        p.N = d->N; p.H = Ho; p.W = Wo; p.Cin = d->Cout; p.in_cs = dz_cstride;
        p.Cout = d->Cin; p.out_cs = d->in_cstride; p.res_cs = d->in_cstride;
        p.K = nt * d->Cout;
        p.Kpad = (p.K + BK - 1) / BK * BK;
        p.act = RYOLO_ACT_LINEAR; p.slope = 0.f; p.ups = 1; p.nt = 0;
        p.cin_log2 = ilog2_exact(d->Cout);
        p.ntaps = nt;
        if (d->stride == 1) {
            p.stride = 1; p.pad = d->ksize - 1 - d->pad;
            p.Ho = d->H; p.Wo = d->W;
            for (int t = 0; t < nt; t++) { p.tap_dy[t] = dy[t]; p.tap_dx[t] = dxx[t]; }
            p.os = 1; p.osx = 1; p.ooy = 0; p.oox = 0; p.OH = d->H; p.OW = d->W;
        } else if (xfuse) {
            const int a = cls;
            p.stride = 1; p.pad = 0;
            p.Cout = 2 * d->Cin; p.out_cs = 2 * d->in_cstride; p.res_cs = 2 * d->in_cstride;
            p.Ho = (d->H - a + 1) / 2; p.Wo = d->W / 2;                 // rows of this parity x pixel PAIRS
            for (int t = 0; t < nt; t++) { p.tap_dy[t] = dy[t]; p.tap_dx[t] = dxx[t]; }
            p.os = 2; p.osx = 1; p.ooy = a; p.oox = 0; p.OH = d->H; p.OW = d->W / 2;
            if (p.Ho <= 0 || p.Wo <= 0) { wsrc += ((size_t)wrows * p.Kpad + 128) * 2; continue; }
        } else {
            const int a = cls >> 1, b = cls & 1;
            p.stride = 1; p.pad = 0;
            p.Ho = (d->H - a + 1) / 2; p.Wo = (d->W - b + 1) / 2;      // grid of input pixels with this parity
            for (int t = 0; t < nt; t++) { p.tap_dy[t] = dy[t]; p.tap_dx[t] = dxx[t]; }
            p.os = 2; p.osx = 2; p.ooy = a; p.oox = b; p.OH = d->H; p.OW = d->W;
            if (p.Ho <= 0 || p.Wo <= 0) { wsrc += ((size_t)rows * p.Kpad + 128) * 2; continue; }
        }
        for (int t = nt; t < 9; t++) { p.tap_dy[t] = 0; p.tap_dx[t] = 0; }
        p.M = (int)((long long)d->N * p.Ho * p.Wo);
        const unsigned long long xb = (((unsigned long long)d->N * Ho * Wo - 1) * dz_cstride + d->Cout) * 2ull;
        const unsigned long long wb = ((unsigned long long)wrows * p.Kpad + 128) * 2ull;
        p.fast = (d->Cout % BK == 0) && xb < 0x7fffff00ull && wb < 0x7fffff00ull;
        p.taps2 = 0;
        if (!p.fast && (d->stride != 1 || (d->ksize == 3 && p.cin_log2 < 0))) return RYOLO_EINVAL;
        p.x_bytes = (unsigned)(p.fast ? xb : 0);
        p.w_bytes = (unsigned)(p.fast ? wb : 0);
        p.stat_part = nullptr; p.stat_cpad = 0;
        p.no_persist = (d->tile & 0x200) ? 1 : 0;
        p.force_persist = 0;
        p.pw_grid_cap = (d->tile & 0xff) == 13 ? (d->tile >> 16) & 0xff : 0;
        p.ntiles = 0; p.magic_wo = p.magic_ho = p.magic_nt = 0;
        p.nt_out = (long long)d->N * d->H * d->W * d->Cin * 2 >= nt_out_min_bytes() ? 1 : 0;
        const int pick = (d->tile & 0xff);   // 0 = auto
        const int rc = dispatch(p, d->ksize, pick, (hipStream_t)stream_);
        if (rc != RYOLO_OK) return rc;
        wsrc += ((size_t)wrows * p.Kpad + 128) * 2;
    }
    return RYOLO_OK;
}

}  // extern "C"
