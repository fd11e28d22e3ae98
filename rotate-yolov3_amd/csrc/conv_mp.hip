// rotate-yolov3_amd/csrc/conv_mp.hip -- the wide implicit-GEMM convolution tile for gfx950: BM pixels x 256 output
// channels per workgroup, 8 waves (2 pixel halves x 4 channel quarters, 128 x 64 outputs per wave at BM = 256), ONE
// PERSISTENT workgroup per CU (2 waves per SIMD, 128 KiB of LDS), multi-phase K loop with counted vmcnt.
//
// Same operator as conv.hip (model/models.py:49-66 conv -> BN -> PReLU, :281-282 shortcut, :93-94 upsample; and, GEN,
// the stride-1 / stride-2-parity-class dgrad + BatchNorm batch statistics of the training step); same operand layout:
//   D[c_out][pixel] = sum_k Wp[c_out][k] * X[pixel][k], weights as the MFMA A operand, 16x16x32 bf16 MFMA,
//   LDS image of a tile = [row][8 x 16-B slots], slot ^= (row >> 1) & 7, filled by 16-B direct-to-LDS buffer loads.
// What is different: the 128x128 / 2-barrier structure spends a third of its wave time in
// s_waitcnt because every K step drains its loads at a barrier and a wave has only 16 MFMAs between barriers.  Here
//   * a K tile (64 deep) is split in FOUR chunks (16 KiB each, 2 direct-to-LDS instructions per wave): XA / XB = the
//     activation rows the waves read in phase 0 / phase 2, WA / WB = the weight rows read in phase 0 / phase 1;
//     chunk s is issued 6 phases before the phase that first reads it, into the half of the double buffer that was
//     last read >= 2 phases earlier; waits are COUNTED (vmcnt(8) = four chunks stay in flight), never 0;
//   * a K tile is FOUR phases of 16 MFMAs per wave (a 64-pixel x 32-channel quadrant each); fragments of a phase are
//     read (ds_read_b128) in the phase's load segment; the 128 x 64 wave tile needs 24 reads per 64 MFMAs;
//   * the two waves of a SIMD (wave w and w + 4) run HALF A PHASE apart: while one issues its 16 MFMAs the other issues
//     its LDS reads / direct-to-LDS loads / waits -- two s_barriers per phase keep them interleaved;
//   * the grid is persistent and the chunk stream never stops: during the last two K tiles of an output tile the
//     look-ahead chunks are the first two K tiles of the workgroup's NEXT output tile, so a tile has no load prologue;
//   * the epilogue never touches LDS (the next tile's operands are landing there): scale/shift/activation on the
//     accumulators -> bf16 -> a 4x4 transpose across the four lanes of a pixel (v_permlane32_swap, v_permlane16_swap) so
//     that every lane owns 32 contiguous bytes -> 16-B global stores; the residual is fetched with the same layout into
//     the (dead) fragment registers before the arithmetic starts.  No barrier, no __syncthreads between tiles.
// Hazards (both wave groups, derived in DESIGN.md): data waited for in phase q (vmcnt before the phase's first barrier)
// is read in phase q+1 or later; a buffer is refilled >= 2 phases after its last ds_read.
// FAST path only (C_in % 64 == 0, tensors < 2 GiB, K >= 128), C_out % 256 == 0.  Everything else: conv.hip.
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"

using namespace ryolo_detail;

namespace {

constexpr int MP_BN = 256;
constexpr int MP_XB = 256 * 128, MP_WB = 256 * 128;     // bytes of one activation / weight stage (256 rows x 128 B)
constexpr int MP_OPS = 2 * MP_XB + 2 * MP_WB;             // operand stages, 4-phase schedule: [X0][X1][W0][W1]
constexpr int MP_SS = 2 * 2 * MP_BN * 4;                 // two slots of {scale[256], shift[256]} (fp32) for the epilogue
constexpr int MP_TRACE = 8 * 128 * 4;                    // debug variant: 128 time stamps per wave
constexpr int MP_LDS = MP_OPS + MP_SS + MP_TRACE;
constexpr int MP_STAT_NT = 4;                            // channel tiles whose BatchNorm statistics a workgroup carries in LDS
constexpr int MP_STAT = MP_STAT_NT * 2 * 2 * MP_BN * 4;  // [channel tile][wave row][sum | sum of squares][256] fp32
constexpr int MP_LDS_GEN = MP_LDS + MP_STAT;

template <int N> using ic = std::integral_constant<int, N>;

// VAR bits: 1 = no half-phase stagger of the two wave groups, 2 = s_setprio 1 around the MFMA clusters (measured: -3 %),
//           timing-only ablations (wrong results): 8 = no global stores in the epilogue, 16 = no epilogue at all
// GEN: 0 = inference; 1 = training forward (BatchNorm statistics of the stored values, NO residual operand); 2 = data
// gradient (strided placement of the stride-2 parity classes and / or accumulation through the residual operand, no
// statistics).  One instantiation with statistics AND residual live at once spilled 39 VGPRs into the epilogue (176 scratch
// operations per tile: the 128->256@76^2 training forward ran 52 % slower than the plain kernel); the split has none.
// KO: K-tile visiting order, 0 tap-major, 1 channel-slice-major (round 5; the switch, its rule and its reason are conv_mq.hip's: the two
// kernels keep adding in the same order and stay bit-identical)
template <int BM, int GEN, int VAR, int KO = 0>
__global__ void __launch_bounds__(512) conv_mp_kernel(const ConvParams p) {
    constexpr int OPS = MP_OPS;
    constexpr int WBASE = 2 * MP_XB, XBASE = 0;
    constexpr int HP = BM / 2;            // pixels per wave row
    constexpr int PF = HP / 16;           // pixel fragments per wave (8 at BM 256, 6 at BM 192)
    constexpr int PQ = PF / 2;            // ... per phase
    constexpr int NXP = BM / 16;          // 8-row pieces per activation chunk (16 / 12); 16 issue slots per chunk
    static_assert(BM == 256 || BM == 192, "BM");
    constexpr bool STAGGER = !(VAR & 1), PRIO = (VAR & 2) != 0;
    constexpr bool NO_STORE = (VAR & 8) != 0, NO_EPI = (VAR & 16) != 0;
    // the barrier that ends an MFMA segment is issued EARLY MFMAs before the segment's last one: the partner wave's first
    // MFMAs then queue up behind this wave's last ones instead of waiting out the barrier round trip
    constexpr int EARLY = 0;   // measured: issuing it 2 / 4 / 8 MFMAs early costs 8-10 %
    constexpr int NST = (GEN == 2 || NO_STORE || NO_EPI) ? 0 : 2 * (BM / 32);   // (GEN 1: only while the statistics stay in LDS)   // buffer stores per wave per output tile (exact)
    constexpr bool TRACE = (VAR & 128) != 0;   // debug: s_memtime stamps of K tiles 4..7 of the first output tile -> p.stat_part
    constexpr bool SKEW = (VAR & 32) != 0;     // ablation: workgroup (loc & 3) starts (loc & 3) * p.dbg0 cycles late
    constexpr bool TRACE_EPI = (VAR & 1024) != 0;   // ablation: stamps around the K loop / epilogue of the first tiles
    constexpr int STAUX = (VAR & 64) ? 2 : ((VAR & 512) ? 16 : 0);   // ablation: cache policy of the output stores (nt / sc1)

    extern __shared__ __attribute__((aligned(16))) char smem[];   // [X0][X1][W0][W1]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int frow = lane & 15, fk = lane >> 4;

    // ---- this workgroup's tile list: XCD x (= blockIdx & 7) owns the x-th contiguous chunk of tile ids (channel tile fastest)
    const int T = p.ntiles, G = gridDim.x;
    const int tq = T >> 3, tr = T & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, nloc = G >> 3;
    const int tstart = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int tlen = tq + (xcd < tr ? 1 : 0);
    if (loc >= tlen) return;
    if constexpr (SKEW) {
        const long long t_end = (long long)__builtin_readcyclecounter() + (long long)(loc & 3) * p.dbg0;
        while ((long long)__builtin_readcyclecounter() < t_end) __builtin_amdgcn_s_sleep(8);
    }
    // training instantiation: the workgroup keeps the per-channel sums of ALL its tiles in LDS and adds them to the global
    // partial rows once, at the end (the per-tile atomics were the kernel's bottleneck: 1024 atomic operations per tile kept
    // the L2 atomic units busy for longer than the tile's MFMAs -- +54 % on 128->256@76^2).  Every (wave row, channel) entry
    // is owned by ONE wave, so the accumulation order is fixed; the cross-workgroup sum stays in fp64 atomics.
    float *stat_lds = (float *)(smem + OPS + MP_SS + MP_TRACE);
    const bool stat_in_lds = GEN == 1 && p.stat_part != nullptr && p.nt <= MP_STAT_NT;
    if constexpr (GEN == 1) {
        if (stat_in_lds)
            for (int i = tid; i < MP_STAT / 4; i += 512) stat_lds[i] = 0.f;     // visible after the prologue's barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }

    // ---- staging bookkeeping.  Piece i = 2*chunk + k of this wave: 8 tile rows, lane l fills 16-B slot (l & 7) of row (l >> 3).
    int a_off32[4], b_off32[4];
    unsigned a_mask[4];
    int a_lds[4], b_lds[4];                // wave-uniform byte offsets of the pieces inside a stage / chunk slot
    int a_pr[4];                           // tile pixel row of lane 0 of activation piece i (-1: dead slot)
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = i >> 1, q = 2 * wave + (i & 1);
        // activation chunk c covers the rows of pixel quarter c of both wave rows; dead slots (BM 192) target spare rows >= BM
        const bool live = q < NXP;
        const int hw = q / (NXP / 2), idx = q % (NXP / 2);
        const int rb = live ? hw * (HP / 8) + c * (HP / 16) + idx : BM / 8 + (q - NXP);   // tile row / 8
        a_pr[i] = live ? rb * 8 : -1;
        a_lds[i] = rb * 1024;
        // weight chunk c covers channel rows [wn'*64 + c*32, +32) of the four channel quarters
        const int rbw = (q >> 2) * 8 + c * 4 + (q & 3);
        b_lds[i] = rbw * 1024;
    }
    // `ln` = the lane id behind an empty asm: keeps the compiler from hoisting these few integer operations out of the tile
    // loop into registers that then live (and spill) across the whole K loop
    auto setup_x = [&](int i, int m0, int ln, int &o_off, unsigned &o_mask) __attribute__((always_inline)) {   // m0 < 0: no such tile (every lane out of range)
        const int lrow = (a_lds[i] >> 7) + (ln >> 3);            // row inside the LDS image (swizzle key)
        const int slot = (ln & 7) ^ ((lrow >> 1) & 7);
        const int m = m0 + a_pr[i] + (ln >> 3);
        unsigned mk = 0;
        int off = 0;
        if (m0 >= 0 && a_pr[i] >= 0 && m < p.M) {
            int wo, ho, img;
            split_pixel(m, p.Wo, p.Ho, p.magic_wo, p.magic_ho, p.use_magic, wo, ho, img);
            const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
            off = (((img * p.H + hi0) * p.W + wi0) * p.in_cs) * 2 + slot * 16;
#pragma unroll
            for (int t = 0; t < 9; t++) {     // branch-free (a scalar branch per tap and piece adds up in the tile prologue)
                const int hi = hi0 + p.tap_dy[t], wi = wi0 + p.tap_dx[t];
                const unsigned in = (unsigned)(t < p.ntaps) & (unsigned)((unsigned)hi < (unsigned)p.H) & (unsigned)((unsigned)wi < (unsigned)p.W);
                mk |= in << t;
            }
        }
        o_off = off;
        o_mask = mk;
    };
    auto setup_w = [&](int i, int n0, int ln, int &o_off) __attribute__((always_inline)) {
        const int row = (b_lds[i] >> 7) + (ln >> 3);
        const int slot = (ln & 7) ^ ((row >> 1) & 7);
        o_off = ((n0 + row) * p.Kpad + slot * 8) * 2;
    };
    // lane t keeps the byte offset of tap t; the K loop fetches the current one with v_readlane
    int lane_tapoff;
    {
        const int t = lane < p.ntaps ? lane : 0;
        lane_tapoff = ((p.tap_dy[t] * p.W + p.tap_dx[t]) * p.in_cs) * 2;
    }
    const int cin_bytes = p.Cin * 2;
    const int KT = p.Kpad / BK;

    // stage toggles: activation stage s lives at byte s*32768, weight stage s at 65536 + s*32768 -> XOR 0x8000
    int xst = 0;                           // scalar: byte offset of the CURRENT K tile's stage inside its operand region
    auto issue_x = [&](int c, int stage_off, int tap, int cbyte) __attribute__((always_inline)) {   // activation chunk c of the K tile at (tap, cbyte)
        tap = __builtin_amdgcn_readfirstlane(tap);
        const int tapoff = __builtin_amdgcn_readlane(lane_tapoff, tap) + __builtin_amdgcn_readfirstlane(cbyte);
        char *base = smem + XBASE + __builtin_amdgcn_readfirstlane(stage_off);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = 2 * c + k;
            const bool ok = (a_mask[i] >> tap) & 1u;
            const int voff = ok ? a_off32[i] + tapoff : (int)0x80000000;   // out of range: the hardware writes zeros
            buffer_load_lds16(p.x, p.x_bytes, base + a_lds[i], voff, 0);
        }
    };
    auto issue_w = [&](int c, int stage_off, int kt) __attribute__((always_inline)) {
        char *base = smem + WBASE + __builtin_amdgcn_readfirstlane(stage_off);
        const int soff = __builtin_amdgcn_readfirstlane(kt) * (BK * 2);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int i = 2 * c + k;
            buffer_load_lds16(p.w, p.w_bytes, base + b_lds[i], b_off32[i], soff);
        }
    };

    // ---- fragment read addresses: one VGPR per (operand, k half); fragments are immediate offsets, the stage an XOR
    int px[2], pw[2];
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
        const int sw = (((ks * 4 + fk) ^ ((frow >> 1) & 7)) << 4) + frow * 128;
        px[ks] = XBASE + (wm * HP) * 128 + sw;
        pw[ks] = WBASE + (wn * 64) * 128 + sw;
    }

    f32x4 acc[4][PF];
    bf16x8 xf[PQ][2], wlo[2][2], whi[2][2];   // wlo / whi = channel fragments 0,1 / 2,3

    // K position (tap, channel byte offset, K tile index) of the K tiles one and two ahead of the current one, cyclic
    int tap1 = 0, cb1 = 0, kt1 = 0, tap2 = 0, cb2 = 0, kt2 = 0;
    auto advance = [&](int &tap, int &cb, int &kt) __attribute__((always_inline)) {
        if constexpr (KO == 1) {          // kt = tap * (C_in / 64) + slice: + C_in / 64 per tap, on to the next slice's tap 0 after the last tap
            tap++;
            kt += cin_bytes >> 7;
            if (tap >= p.ntaps) {
                tap = 0;
                cb += BK * 2;
                kt = cb >> 7;
                if (cb >= cin_bytes) { cb = 0; kt = 0; }
            }
        } else {
            cb += BK * 2;
            kt++;
            if (cb >= cin_bytes) { cb = 0; tap++; }
            if (kt == KT) { kt = 0; tap = 0; cb = 0; }
        }
        // wave-uniform by construction; say so (the compiler otherwise keeps them in VGPRs and wraps every
        // direct-to-LDS load that uses them as a scalar offset in a waterfall loop)
        cb = __builtin_amdgcn_readfirstlane(cb);
        kt = __builtin_amdgcn_readfirstlane(kt);
        tap = __builtin_amdgcn_readfirstlane(tap);
    };

    const int trace_off = OPS + MP_SS + wave * 512;
    if constexpr (TRACE_EPI) {
        if (lane < 32) *(unsigned *)(smem + trace_off + lane * 4) = 0u;
    }
    int tr_idx = 0;
    bool tr_on = false;
    auto stamp = [&]() __attribute__((always_inline)) {
        if constexpr (TRACE) {
            if (tr_on) {
                const unsigned tnow = (unsigned)__builtin_readcyclecounter();
                *(unsigned *)(smem + trace_off + tr_idx * 4) = tnow;
                tr_idx++;
            }
        }
    };
    int te_tile = 0;                       // TRACE_EPI: output tiles done by this workgroup
    auto stamp_e = [&](int k) __attribute__((always_inline)) {
        if constexpr (TRACE_EPI) {
            if (te_tile < 4) *(unsigned *)(smem + trace_off + (te_tile * 8 + k) * 4) = (unsigned)__builtin_readcyclecounter();
        }
    };
    bool lenient = false;                  // this K tile follows an epilogue: NST stores sit in the in-order queue
    auto phase = [&](auto PHc) __attribute__((always_inline)) {
        constexpr int PH = decltype(PHc)::value;
        stamp();                                   // (0) load segment starts
        // ---------------- load segment: this phase's fragments, one chunk of a later K tile, the counted wait
        if constexpr (PH == 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
#pragma unroll
                for (int c = 0; c < 2; c++) wlo[c][ks] = *(const bf16x8 *)(smem + pw[ks] + c * 2048);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
#pragma unroll
                for (int f = 0; f < PQ; f++) xf[f][ks] = *(const bf16x8 *)(smem + px[ks] + f * 2048);
            }
        } else if constexpr (PH == 1) {
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
#pragma unroll
                for (int c = 0; c < 2; c++) whi[c][ks] = *(const bf16x8 *)(smem + pw[ks] + (2 + c) * 2048);
            }
        } else if constexpr (PH == 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
#pragma unroll
                for (int f = 0; f < PQ; f++) xf[f][ks] = *(const bf16x8 *)(smem + px[ks] + (PQ + f) * 2048);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PH == 0) issue_w(1, xst ^ MP_XB, kt1);
        if constexpr (PH == 1) issue_x(1, xst ^ MP_XB, tap1, cb1);
        if constexpr (PH == 2) issue_x(0, xst, tap2, cb2);
        if constexpr (PH == 3) issue_w(0, xst, kt2);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PH != 2) {
            if (NST > 0 && lenient) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + NST) : "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        stamp();                                   // (1) reads + loads issued, counted wait passed
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        stamp();                                   // (2) barrier 1 released
        // ---------------- MFMA segment: one (BM/4)-pixel x 32-channel quadrant, K = 64
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
        constexpr int C0 = (PH == 0 || PH == 3) ? 0 : 2, F0 = (PH < 2) ? 0 : PQ;
        constexpr int NM = 4 * PQ;           // MFMAs of the segment
#pragma unroll
        for (int j = 0; j < NM; j++) {
            const int ks = j / (2 * PQ), c = (j / PQ) & 1, f = j % PQ;
            if (EARLY > 0 && j == NM - EARLY) {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
            const bf16x8 wv = (C0 == 0) ? wlo[c][ks] : whi[c][ks];
            acc[C0 + c][F0 + f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, xf[f][ks], acc[C0 + c][F0 + f], 0, 0, 0);
        }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        stamp();                                   // (3) MFMAs issued
        if constexpr (EARLY == 0) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- first tile: bookkeeping + chunks 0..5 = XA(0) WA(0) WB(0) XB(0) XA(1) WA(1)
    int ti = loc;                          // index inside the XCD's chunk
    int m0, n0;
    {
        const int id = tstart + ti;
        const int mt = udiv_magic(id, p.magic_nt);
        m0 = mt * BM;
        n0 = (id - mt * p.nt) * MP_BN;
#pragma unroll
        for (int i = 0; i < 4; i++) { setup_x(i, m0, lane, a_off32[i], a_mask[i]); setup_w(i, n0, lane, b_off32[i]); }
    }
    // folded-BN scale / shift of the tile's 256 channels live in LDS (slot = tile parity): thread t holds element t of
    // {scale[n0 .. n0+255], shift[n0 .. n0+255]}; the slot of tile i+1 is written at the end of tile i's epilogue
    float *ss = (float *)(smem + OPS);
    int sslot = 0;
    ss[tid] = (tid < MP_BN ? p.scale : p.shift)[n0 + (tid & (MP_BN - 1))];
    advance(tap2, cb2, kt2);               // -> K tile 1
    issue_x(0, 0, 0, 0);
    issue_w(0, 0, 0);
    issue_w(1, 0, 0);
    issue_x(1, 0, 0, 0);
    issue_x(0, MP_XB, tap2, cb2);
    issue_w(0, MP_XB, kt2);
    tap1 = tap2; cb1 = cb2; kt1 = kt2;
    advance(tap2, cb2, kt2);               // -> K tile 2 (or 0 of the next tile when KT == 2)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if constexpr (STAGGER) {
        if (wm == 1) __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);

    const int fr4 = fk * 4;
    const float slope = p.slope;
    while (true) {
        const int tnext = ti + nloc;
        const bool has_next = tnext < tlen;
        int nm0 = -1, nn0 = n0;
        if (has_next) {
            const int id = tstart + tnext;
            const int mt = udiv_magic(id, p.magic_nt);
            nm0 = __builtin_amdgcn_readfirstlane(mt * BM);
            nn0 = __builtin_amdgcn_readfirstlane((id - mt * p.nt) * MP_BN);
        }
        // bookkeeping of the NEXT output tile, computed here where registers are plentiful; the K loop only moves it in
        int xn_off[4], wn_off[4];
        unsigned xn_mask[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            setup_x(i, nm0, ln, xn_off[i], xn_mask[i]);
            setup_w(i, nn0, ln, wn_off[i]);
        }
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int f = 0; f < PF; f++) acc[c][f] = f32x4{0.f, 0.f, 0.f, 0.f};

        stamp_e(0);
#pragma clang loop unroll(disable)
        for (int t = 0; t < KT; t++) {
            int tt = t;
            asm volatile("" : "+s"(tt));
            if constexpr (TRACE) tr_on = (ti == loc) && t >= 4 && t < 12 && tr_idx < 128;   // opaque: no peeling / unswitching of the K loop on the two conditions below
            // the look-ahead chunks run into the NEXT output tile: pieces 0,1 (XA, WA: issued in phases 2, 3) switch
            // before K tile KT-2, pieces 2,3 (WB, XB: phases 0, 1) before K tile KT-1
            if (tt == KT - 2) {
#pragma unroll
                for (int i = 0; i < 2; i++) { a_off32[i] = xn_off[i]; a_mask[i] = xn_mask[i]; b_off32[i] = wn_off[i]; }
            }
            if (tt == KT - 1) {
#pragma unroll
                for (int i = 2; i < 4; i++) { a_off32[i] = xn_off[i]; a_mask[i] = xn_mask[i]; b_off32[i] = wn_off[i]; }
            }
            // first K tile of every output tile but the workgroup's first: the previous tile's 2*PF stores sit in the in-order
            // queue between the look-ahead chunks and this K tile's requests -- let them drain under these four phases
            lenient = (tt == 0) && (ti != loc) && (GEN == 0 || stat_in_lds);   // (per-tile statistic atomics would sit in the queue too)
            phase(ic<0>{});
            phase(ic<1>{});
            phase(ic<2>{});
            phase(ic<3>{});
            tap1 = tap2; cb1 = cb2; kt1 = kt2;
            advance(tap2, cb2, kt2);
            xst = __builtin_amdgcn_readfirstlane(xst ^ MP_XB);
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                px[ks] ^= MP_XB;
                pw[ks] ^= MP_WB;
            }
        }

        stamp_e(1);
        if constexpr (TRACE) {
            if (ti == loc && blockIdx.x == 0 && p.stat_part) {
                unsigned *dst = (unsigned *)p.stat_part + wave * 128;
                for (int i2 = lane; i2 < 128; i2 += 64) dst[i2] = *(const unsigned *)(smem + trace_off + i2 * 4);
            }
        }
        // ------------------------------------------------------------------ epilogue (registers -> global, no LDS)
        if constexpr (NO_EPI) {
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int f = 0; f < PF; f++) asm volatile("" ::"v"(acc[c][f]));
        } else {
          // next tile's folded-BN scale / shift element of this thread: requested first, written to its LDS slot last
          const float ss_next = (tid < MP_BN ? p.scale : p.shift)[nn0 + (tid & (MP_BN - 1))];
          auto run_epilogue = [&](auto ACTc) __attribute__((always_inline)) {
            constexpr int ACT = decltype(ACTc)::value;
            // the lane coordinates are re-derived behind an empty asm: the compiler otherwise computes the epilogue's per-lane
            // offsets before the K loop and SPILLS them across it (scratch stores at tile setup, loads here, every tile)
            // (training instantiations only: in the inference one the same trick moves MORE values to scratch -- 15 instead of 4)
            int ln_e = lane;
            if constexpr (GEN != 0) asm volatile("" : "+v"(ln_e));
            const int frow = ln_e & 15, fk = ln_e >> 4, fr4 = fk * 4;
            const int chq = n0 + wn * 64;                   // first channel of this wave's quarter
            // after the lane regrouping lane (frow, fk) owns channels chq + 8*fk .. +7 and chq + 32 + 8*fk .. +7 of pixel frow.
            // Output / residual go through buffer descriptors: ONE per-lane byte offset (pixel frow of fragment 0) plus a
            // scalar offset per fragment; lanes past M get an out-of-range offset (loads return 0, stores are dropped), so
            // every wave issues exactly 2*PF stores per tile whatever its tail.
            const int mrow = m0 + wm * HP + frow;
#if defined(__HIP_DEVICE_COMPILE__)
            const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)p.y, 0, p.y_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.res ? p.res : p.y), 0, p.res ? p.res_bytes : 0u, 0x00020000);
#endif
            auto opix = [&](int m) -> int {                 // GEN, stride-2 dgrad parity classes: strided placement
                int j, i, img;
                split_pixel(m, p.Wo, p.Ho, p.magic_wo, p.magic_ho, p.use_magic, j, i, img);
                return (img * p.OH + (i * p.os + p.ooy)) * p.OW + (j * p.osx + p.oox);
            };
            const bool strided = GEN == 2 && p.os != 1;
            const int yoff0 = (mrow * p.out_cs + chq + fk * 8) * 2, roff0 = (mrow * p.res_cs + chq + fk * 8) * 2;
            const int ystep = 16 * p.out_cs * 2, rstep = 16 * p.res_cs * 2;
            // residual rows: all requested up front into the (dead) fragment registers
            u32x4 rv[PF][2];
            const bool has_res = GEN != 1 && p.res != nullptr;
            if (has_res) {
#pragma unroll
                for (int f = 0; f < PF; f++) {
                    const int m = mrow + f * 16;
                    int voff = roff0, soff = f * rstep;
                    if (strided) { voff = (opix(m < p.M ? m : 0) * p.res_cs + chq + fk * 8) * 2; soff = 0; }
                    voff = m < p.M ? voff : (int)0x80000000;
#if defined(__HIP_DEVICE_COMPILE__)
                    rv[f][0] = __builtin_amdgcn_raw_buffer_load_b128(rrs, voff, soff, 0);
                    rv[f][1] = __builtin_amdgcn_raw_buffer_load_b128(rrs, voff + 64, soff, 0);
#endif
                }
            }
            stamp_e(2);
            const float *ssc = ss + sslot * (2 * MP_BN) + wn * 64 + fr4;   // + c*16: scale; + 256: shift
            // two passes over the pixel fragments, one per pair of channel fragments (= one 16-B store per lane and fragment):
            // the pair's scale / shift are read from LDS once per pass
#pragma unroll
            for (int h = 0; h < 2; h++) {
                f32x4 sc[2], sh[2];
#pragma unroll
                for (int cc = 0; cc < 2; cc++) {
                    sc[cc] = *(const f32x4 *)(ssc + (2 * h + cc) * 16);
                    sh[cc] = *(const f32x4 *)(ssc + MP_BN + (2 * h + cc) * 16);
                }
                float st_sum[2][4], st_sq[2][4];       // statistics of this pass's two channel fragments (16 live values, not 32)
#pragma unroll
                for (int cc = 0; cc < 2; cc++)
#pragma unroll
                    for (int r = 0; r < 4; r++) st_sum[cc][r] = st_sq[cc][r] = 0.f;
#pragma unroll
                for (int f = 0; f < PF; f++) {
                    const int m = mrow + f * 16;
                    const bool ok = m < p.M;
                    unsigned R[2][2];
#pragma unroll
                    for (int cc = 0; cc < 2; cc++) {
                        const int c = 2 * h + cc;
                        bf16x4 o;
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            float v = acc[c][f][r] * sc[cc][r] + sh[cc][r];
                            if constexpr (ACT == RYOLO_ACT_LEAKY) v = v > 0.f ? v : v * slope;
                            else if constexpr (ACT == 3) v = fmaxf(v, v * slope);   // leaky with slope <= 1: same values, one compare+select less
                            else if constexpr (ACT == RYOLO_ACT_MISH) v = mish(v);
                            o[r] = (__bf16)v;
                            if (GEN == 1 && p.stat_part && ok) {      // statistics of the values as stored (bf16)
                                const float q = (float)o[r];
                                st_sum[cc][r] += q;
                                st_sq[cc][r] += q * q;
                            }
                        }
                        const uint2 u = __builtin_bit_cast(uint2, o);
                        R[cc][0] = u.x;
                        R[cc][1] = u.y;
                    }
                    // regroup the 8-B units (channel fragment, lane group fk) of a pixel so that the four lanes of the pixel hold
                    // 16 B each of 64 CONTIGUOUS bytes: lane group a ends up with channels 32h + 8a .. 8a+7 (a store instruction
                    // then writes whole 64-B half lines instead of 16-B pieces at a 32-B stride)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                    for (int d = 0; d < 2; d++) {
                        auto s1 = __builtin_amdgcn_permlane32_swap(R[0][d], R[1][d], false, false);   // lanes 32+: R[0] <-> lanes 0-31: R[1]
                        auto s2 = __builtin_amdgcn_permlane16_swap(s1[0], s1[1], false, false);       // odd rows: R[0] <-> even rows: R[1]
                        R[0][d] = s2[0];
                        R[1][d] = s2[1];
                    }
#endif
                    u32x4 out = u32x4{R[0][0], R[0][1], R[1][0], R[1][1]};
                    if (has_res) {
                        bf16x8 a = __builtin_bit_cast(bf16x8, out);
                        const bf16x8 b = __builtin_bit_cast(bf16x8, rv[f][h]);
#pragma unroll
                        for (int e = 0; e < 8; e++) a[e] = (__bf16)((float)a[e] + (float)b[e]);
                        out = __builtin_bit_cast(u32x4, a);
                    }
                    if constexpr (NO_STORE) {
                        asm volatile("" ::"v"(out.x), "v"(out.y), "v"(out.z), "v"(out.w));
                    } else {
                        int voff = yoff0, soff = f * ystep;
                        if (strided) { voff = (opix(ok ? m : 0) * p.out_cs + chq + fk * 8) * 2; soff = 0; }
                        voff = ok ? voff : (int)0x80000000;
#if defined(__HIP_DEVICE_COMPILE__)
                        if (STAUX == 0 && p.nt_out) buffer_store16_soff<2>(out, yrs, voff + 64 * h, soff);      // large outputs: non-temporal
                        else buffer_store16_soff<STAUX>(out, yrs, voff + 64 * h, soff);
#endif
                    }
                }
                stamp_e(3 + h);
                if (GEN == 1 && p.stat_part) {
                    // the 16 lanes of a k-group hold the same channels: DPP row sum over them leaves every total in all 16
                    // lanes; lane frow < 8 keeps total number frow (fragment frow / 4, register frow % 4) and adds it to
                    // this wave's entry of the workgroup's LDS accumulator (or, for more channel tiles than it holds, one
                    // atomic instruction per statistic per wave)
                    double *row = p.stat_part + (size_t)(((m0 / BM) * 2 + wm) % STAT_ROWS) * 2 * p.stat_cpad;
                    float *slot = stat_lds + ((size_t)((n0 / MP_BN) * 2 + wm) * 2) * MP_BN + wn * 64;
                    float ta = 0.f, tb = 0.f;
#pragma unroll
                    for (int cc = 0; cc < 2; cc++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const float a = row16_sum(st_sum[cc][r]), b = row16_sum(st_sq[cc][r]);
                            if (frow == cc * 4 + r) {
                                ta = a;
                                tb = b;
                            }
                        }
                    if (frow < 8) {
                        const int cl = (2 * h + (frow >> 2)) * 16 + fr4 + (frow & 3);
                        if (stat_in_lds) {
                            slot[cl] += ta;
                            slot[MP_BN + cl] += tb;
                        } else {
                            atomicAdd(row + chq + cl, (double)ta);
                            atomicAdd(row + p.stat_cpad + chq + cl, (double)tb);
                        }
                    }
                }
            }
          };
          if (p.act == RYOLO_ACT_LEAKY && p.slope <= 1.f) run_epilogue(ic<3>{});
          else if (p.act == RYOLO_ACT_LEAKY) run_epilogue(ic<RYOLO_ACT_LEAKY>{});
          else if (p.act == RYOLO_ACT_MISH) run_epilogue(ic<RYOLO_ACT_MISH>{});
          else run_epilogue(ic<RYOLO_ACT_LINEAR>{});
          ss[(sslot ^ 1) * (2 * MP_BN) + tid] = ss_next;
        }
        stamp_e(5);
        te_tile++;
        __builtin_amdgcn_sched_barrier(0);
        if (!has_next) break;
        sslot = __builtin_amdgcn_readfirstlane(sslot ^ 1);
        ti = tnext;
        m0 = nm0;
        n0 = nn0;
    }
    if constexpr (STAGGER) {
        if (wm == 0) __builtin_amdgcn_s_barrier();
    }
    if constexpr (GEN == 1) {
        if (stat_in_lds) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                  // every wave's last accumulator update is in LDS
            double *row = p.stat_part + (size_t)(blockIdx.x % STAT_ROWS) * 2 * p.stat_cpad;
            for (int i = tid; i < p.nt * 2 * MP_BN; i += 512) {
                const int s_ = i / (2 * MP_BN), st = (i / MP_BN) & 1, ch = i % MP_BN;
                const float *q = stat_lds + ((size_t)(s_ * 2) * 2 + st) * MP_BN + ch;
                const float v = q[0] + q[2 * MP_BN];       // wave row 0 + wave row 1
                if (v != 0.f && s_ * MP_BN + ch < p.Cout) atomicAdd(row + (size_t)st * p.stat_cpad + s_ * MP_BN + ch, (double)v);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the look-ahead chunks behind the last tile
    if constexpr (TRACE_EPI) {
        if ((blockIdx.x == 0 || blockIdx.x == 8 || blockIdx.x == 129) && p.stat_part) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            unsigned *dst = (unsigned *)p.stat_part + ((blockIdx.x == 0 ? 0 : (blockIdx.x == 8 ? 1 : 2)) * 8 + wave) * 32;
            if (lane < 32) dst[lane] = *(const unsigned *)(smem + trace_off + lane * 4);
        }
    }
}

inline unsigned mp_magic_u32(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

inline int mp_cu_count() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    return cus;
}

void *g_trace_buf = nullptr;
int g_dbg[4] = {0, 0, 0, 0};
int g_var[16] = {0};

template <int BM, int GEN, int VAR, int KO = 0>
int mp_launch(ConvParams &p, hipStream_t stream) {
    if (VAR & (128 | 1024)) p.stat_part = (double *)g_trace_buf;
    static bool attr_done = false;
    constexpr int LDS = GEN == 1 ? MP_LDS_GEN : MP_LDS;
    auto kfn = conv_mp_kernel<BM, GEN, VAR, KO>;
    if (!attr_done && !g_conv_choice) {
        if (hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
            return RYOLO_ELAUNCH;
        attr_done = true;
    }
    const int mt = (p.M + BM - 1) / BM;
    p.nt = (p.Cout + MP_BN - 1) / MP_BN;
    const long long T = (long long)mt * p.nt;
    const long long dmax = p.Wo > p.Ho ? p.Wo : p.Ho, mpad = (long long)mt * BM;
    if (mpad * dmax >= 0x100000000ll || T * p.nt >= 0x100000000ll || T > 0x7fffffffll) return RYOLO_EINVAL;
    p.use_magic = 1;
    p.magic_wo = mp_magic_u32(p.Wo);
    p.magic_ho = mp_magic_u32(p.Ho);
    p.magic_nt = mp_magic_u32(p.nt);
    p.ntiles = (int)T;
    {
        const unsigned long long yb = (((unsigned long long)p.N * p.OH * p.OW - 1) * p.out_cs + p.Cout) * 2ull;
        const unsigned long long rb = p.res ? (((unsigned long long)p.N * p.OH * p.OW - 1) * p.res_cs + p.Cout) * 2ull : 0ull;
        if (yb >= 0x7fffff00ull || rb >= 0x7fffff00ull) return RYOLO_EINVAL;
        p.y_bytes = (unsigned)yb;
        p.res_bytes = (unsigned)rb;
    }
    RYOLO_CONV_DRY_RUN(BM == 192 ? RYOLO_CONV_KERNEL_MP192 : RYOLO_CONV_KERNEL_MP256);
    int cus = mp_cu_count() & ~7;
    if (g_dbg[1] >= 8) cus = g_dbg[1] & ~7;   // ablation builds: cap on the persistent grid (0 in the product)
    const int grid = T >= cus ? cus : (int)((T + 7) & ~7ll);   // a multiple of 8 (XCD chunking); surplus workgroups exit at once
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(512), LDS, stream, p);
    return hipGetLastError() == hipSuccess ? RYOLO_OK : RYOLO_ELAUNCH;
}

}  // namespace

namespace ryolo_detail {

bool conv_mp_eligible(const ConvParams &p) {
    return p.fast && !p.taps2 && p.ups == 1 && !(p.stat_part && (p.res || p.os != 1)) && (p.Cin % BK) == 0 && (p.Cout % MP_BN) == 0 && p.Kpad >= 2 * BK && p.ntaps >= 1 && p.ntaps <= 9 &&
           p.Kpad == p.ntaps * p.Cin;
}

// BM that minimises (rounds of the persistent grid) x (tile height): 256 unless 192 saves a quarter round or more
int conv_mp_pick_bm(const ConvParams &p) {
    const long long cus = mp_cu_count() & ~7, nt = (p.Cout + MP_BN - 1) / MP_BN;
    long long best = 0, best_bm = 256;
    for (int bm : {256, 192}) {
        const long long tiles = ((long long)p.M + bm - 1) / bm * nt;
        const long long cost = (tiles + cus - 1) / cus * bm;
        if (best == 0 || cost < best) { best = cost; best_bm = bm; }
    }
    return (int)best_bm;
}

int launch_conv_mp(ConvParams &p, int bm, int variant, hipStream_t stream) {
    if (!conv_mp_eligible(p)) return RYOLO_EINVAL;
    const int gen = p.stat_part != nullptr ? 1 : (p.os != 1 ? 2 : 0);
    if (bm == 0) bm = conv_mp_pick_bm(p);
    if (bm != 256 && bm != 192) return RYOLO_EINVAL;
#ifdef RYOLO_MP_ABLATION
    // schedule variants and timing-only ablations (several of them produce WRONG results): never in the shipped library
    if (gen == 0 && variant != 0) {
        p.dbg0 = g_dbg[0];
#define MP_VAR(V) case V: return bm == 256 ? mp_launch<256, 0, V>(p, stream) : mp_launch<192, 0, V>(p, stream);
        switch (variant) {
            MP_VAR(1) MP_VAR(2) MP_VAR(8) MP_VAR(16) MP_VAR(32) MP_VAR(40) MP_VAR(64) MP_VAR(512) MP_VAR(144) MP_VAR(1024) MP_VAR(1056) MP_VAR(1032)
            default: return RYOLO_EINVAL;
        }
#undef MP_VAR
    }
#endif
    if (variant != 0) return RYOLO_EINVAL;
    // K-tile order: conv_mq.hip's rule (channel-slice-major for 3x3 launches with C_in >= 512; RYOLO_MQ_KORDER = 0 | 1 forces one)
    bool cm = p.ntaps > 1 && p.Cin >= 512;
    {
        const char *e = tune(TUNE_MQ_KORDER);
        if (e) cm = atoi(e) != 0 && p.ntaps > 1;
        const char *m = abl_env("RYOLO_MQ_KORDER_MIN_CIN");     // (measurement build: the threshold itself, for the A/B that chose it)
        if (m && !e) cm = p.ntaps > 1 && p.Cin >= atoi(m);
    }
    if (bm == 256) {
        if (gen == 1) return cm ? mp_launch<256, 1, 0, 1>(p, stream) : mp_launch<256, 1, 0>(p, stream);
        if (gen == 2) return cm ? mp_launch<256, 2, 0, 1>(p, stream) : mp_launch<256, 2, 0>(p, stream);
        return cm ? mp_launch<256, 0, 0, 1>(p, stream) : mp_launch<256, 0, 0>(p, stream);
    }
    if (gen == 1) return cm ? mp_launch<192, 1, 0, 1>(p, stream) : mp_launch<192, 1, 0>(p, stream);
    if (gen == 2) return cm ? mp_launch<192, 2, 0, 1>(p, stream) : mp_launch<192, 2, 0>(p, stream);
    return cm ? mp_launch<192, 0, 0, 1>(p, stream) : mp_launch<192, 0, 0>(p, stream);
}

}  // namespace ryolo_detail

#ifdef RYOLO_MP_ABLATION
namespace ryolo_detail {
int ryolo_mp_ablation_variant(int slot) { return g_var[slot & 15]; }
}
// debug hooks of the ablation build (not part of the product ABI): a device buffer the trace variants fill with s_memtime
// stamps, and two integers (start-skew unit in cycles, cap on the persistent grid)
extern "C" void ryolo_debug_conv_trace(void *buf) { g_trace_buf = buf; }
extern "C" void ryolo_debug_conv_set(int i, int v) { if (i >= 0 && i < 4) g_dbg[i] = v; }
extern "C" void ryolo_debug_conv_variant(int slot, int var) { g_var[slot & 15] = var; }   // tile code 32 + slot (+ 16: BM 192) runs VAR `var`
#endif
