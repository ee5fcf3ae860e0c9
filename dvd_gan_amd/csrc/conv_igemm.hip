// Convolutions of the DVD-GAN step for gfx950: forward / backward-data (implicit GEMM, no im2col) and
// backward-weight, bf16 MFMA with fp32 accumulation or exact fp32 MFMA.
//
//   conv_halo_kernel       3x3 / 5x5 / 3x3x3 filters on frames >= 16 pixels wide: the input footprint of a
//                          16 x 16 pixel patch is staged in LDS once per channel chunk (LDS-DMA), every tap
//                          reads it at an immediate offset; only the weight tile moves per tap.
//   conv_igemm_kernel      1x1 filters and narrow frames: activation tile gathered per (chunk, tap).
//   conv_wgrad_row_kernel  weight gradients, one filter row (all KW taps) per workgroup, reduction over pixels
//                          through transposing LDS reads (ds_read_b64_tr_b16).
//   conv_wgrad_kernel      weight gradients, one tap per workgroup (1x1, upsampling convs, fp32 mode).
//
//   bf16 : v_mfma_f32_32x32x16_bf16   (A: lane l holds row l&31, k = 8*(l>>5)..+7)
//   f32  : v_mfma_f32_32x32x2_f32     (A: lane l holds row l&31, k = l>>5)        exact mode
//   C/D  : col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#include "conv_common.h"
#include "prof.h"

namespace {

// acc[tm][tn] += A(TM*32 rows of this wave) x B(64 cols of this wave) over one 64-byte K chunk.
template <typename T, int TM, bool RELU>
__device__ __forceinline__ void mma_swz(const char* At, const char* Bt, int arow, int brow, int lane,
                                        f32x16 (&acc)[TM][2]) {
    const int kh = lane >> 5;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int slot = kk * 2 + kh;
            bf16x8 b[2], a[TM];
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) b[tn] = *reinterpret_cast<const bf16x8*>(Bt + lds_off(brow + tn * 32, slot));
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                u32x4 raw = *reinterpret_cast<const u32x4*>(At + lds_off(arow + tm * 32, slot));
                // ReLU of the conv input (LDS-DMA cannot transform data) is applied on the fragment; it is a
                // compile-time variant: a runtime branch here makes hipcc wait lgkmcnt(0) after every ds_read
                if constexpr (RELU) raw = relu16_bf16(raw);
                a[tm] = __builtin_bit_cast(bf16x8, raw);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = kk * 2 + kh;                 // float index 0..15 inside the 64-byte row
            float b[2], a[TM];
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
                b[tn] = *reinterpret_cast<const float*>(Bt + lds_off(brow + tn * 32, k >> 2) + (k & 3) * 4);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                a[tm] = *reinterpret_cast<const float*>(At + lds_off(arow + tm * 32, k >> 2) + (k & 3) * 4);
                if constexpr (RELU) a[tm] = fmaxf(a[tm], 0.f);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
    }
}

// Workgroup = 2 x 2 waves, wave tile = (TM*32) x 64  =>  block tile BM = TM*64 rows x 128 columns.
// TM = 4 (256 x 128) is the production shape: 16 MFMAs per wave between barriers and 6 instead of 8
// fragment reads per 8 MFMAs; TM = 2 (128 x 128) serves problems with few rows.
template <typename T, int TM, int WN, bool RELU>   // waves: 2 (M) x WN (N); block tile (TM*64) x (WN*64)
__global__ __launch_bounds__(128 * WN) void conv_igemm_kernel(ConvK p) {
    constexpr int E16 = ElemTraits<T>::kPer16B;
    constexpr int BK = 4 * E16;
    constexpr int BMt = TM * 64, BNt = WN * 64;
    constexpr int NWAVE = 2 * WN;
    constexpr int NA = BMt / 16 / NWAVE;               // 16-row DMA groups of the activation tile per wave
    constexpr int NB = BNt / 16 / NWAVE;               // ... of the weight tile
    constexpr int ABYTES = BMt * 64, BBYTES = BNt * 64;
    constexpr int NSTAGE = 3;                          // LDS ring: tile k is multiplied while k+1, k+2 are in flight
    __shared__ __attribute__((aligned(16))) char smem[NSTAGE][ABYTES + BBYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch order, used for speed only);
    // give each XCD a contiguous run of tiles so the N-tiles of one M-tile share that XCD's L2.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, qd = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (qd + 1) : rr * (qd + 1) + (xcd - rr) * qd) + (bid >> 3);
    }
    // M-major (default): one XCD's contiguous run of tiles shares the activation rows.  N-major, for the recurrent convs of
    // the 4 x 4 / 8 x 8 stages whose weights (up to 26 MB, re-fetched from the Infinity Cache every time step because a
    // step's weights exceed the 4 MiB L2) dwarf the activations: a run shares the WEIGHT tile instead, so each weight block
    // enters one or two L2s rather than all eight.
    int mt, nt;
    if (p.nmajor) { const int tilesM = gridDim.x / p.tilesN; nt = bid / tilesM; mt = bid - nt * tilesM; }
    else { mt = bid / p.tilesN; nt = bid - mt * p.tilesN; }
    const int m0 = mt * BMt, n0 = nt * BNt;
    const int z = blockIdx.z;
    // Pixel-major row order (p.pm = frames F > 0; 4 x 4 / 8 x 8 recurrent convs): GEMM row m' = pixel * F + frame, so a tile
    // holds a few pixels of ONE image line for many frames instead of whole frames -- every row of the tile then has the same
    // set of filter rows that fall outside the frame (2 of 5 at the top / bottom line of a 5 x 5 conv: 30 % of all (line, row)
    // pairs at 4 x 4, 15 % at 8 x 8), and those K steps are skipped for the whole tile instead of multiplying zeros.  The
    // tensors keep their frame-major layout: only the row <-> workgroup assignment changes (memrow), results are identical.
    const int HWp = p.H * p.W;
    auto memrow = [&](int mp) __attribute__((always_inline)) -> int {
        if (!p.pm) return mp;
        const int px = mp / p.pm;
        return (mp - px * p.pm) * HWp + px;
    };
    int iy_lo = 0, iy_hi = p.kh;                       // filter rows that touch the frame for this tile's image line
    if (p.pm) {
        const int y_t = (m0 / p.pm) >> p.logW, pad_ = p.kh >> 1;
        iy_lo = max(0, pad_ - y_t); iy_hi = min(p.kh, p.H + pad_ - y_t);
    }
    const int ntaps = p.pm ? (iy_hi - iy_lo) * p.kw : p.kt * p.kh * p.kw;      // K steps per channel chunk
    const int nk = ntaps * p.kchunks;
    const int per = (nk + p.nsplit - 1) / p.nsplit;
    const int k_begin = z * per;
    const int k_end = min(nk, k_begin + per);
    const size_t esz = sizeof(T);

    // ---- staging: LDS-DMA (buffer_load ... lds).  One wave-instruction moves 64 x 16 B = 16 tile rows
    // straight from L2 into LDS (destination = wave-uniform base + lane*16, i.e. linear), so the tile
    // never passes through VGPRs and no ds_write is issued.  The XOR swizzle of lds_off() is applied
    // to the SOURCE: lane l lands on physical slot l&3 of row l>>2 and therefore fetches logical slot
    // (l&3) ^ ((row>>2)&3).  Rows outside the frame / channels past C use offset 0xFFFFFFFF, for
    // which the buffer range check stores zeros.  Wave w moves row groups w, w+4, ...
    const int wu = __builtin_amdgcn_readfirstlane(wave);
    const int lrow = lane >> 2;                               // row inside a 16-row group
    const int q = (lane & 3) ^ ((lane >> 4) & 3);             // logical 16-byte slot this lane fetches
    int am[NA], ax[NA], ay[NA], at[NA];
    bool av[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int mp = m0 + (i * NWAVE + wu) * 16 + lrow;
        const int m = memrow(mp);
        am[i] = m; av[i] = mp < p.M;
        int f_;
        grid_pos(m, p.H, p.W, p.logH, p.logW, f_, ay[i], ax[i]);
        at[i] = p.kt > 1 ? f_ % p.T : 0;
    }
    int cob[NB];
    bool cov[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) { cob[j] = n0 + (j * NWAVE + wu) * 16 + lrow; cov[j] = cob[j] < p.Cout; }
    // Offsets are 32-bit, activations can exceed 4 GiB: the descriptor of the activation tensor starts
    // at the first input row this tile can touch (wave-uniform), offsets are relative to it.
    const unsigned ldb = (unsigned)p.ldi * (unsigned)esz;        // input row pitch in bytes
    const int base_row = p.pm ? 0 : p.up2 ? (m0 / (p.H * p.W)) * p.Hin * p.Win : max(0, m0 - p.maxshift);
    const size_t base_b = (size_t)base_row * ldb;
    const size_t left_b = p.in_bytes - base_b;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.in + base_b), 0, left_b > 0xfffffffeull ? 0xfffffffeu : (unsigned)left_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    unsigned aoff[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) aoff[i] = (unsigned)(am[i] - base_row) * ldb + q * 16;
    unsigned woff[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) woff[j] = ((unsigned)cob[j] * p.C + q * E16) * (unsigned)esz;
    // wave-uniform K-step state (tap decomposition kept incrementally)
    // K order: channel chunk OUTER, taps INNER -- the 25 (9, 27) taps of one 64-byte channel chunk re-read
    // the same few input rows back to back, so they hit L1/L2 instead of coming back after a whole sweep
    // over C (which at 2 workgroups per CU is tens of MB per XCD, far beyond its 4 MiB L2).
    int tap = 0, cc = 0, it = 0, iy = iy_lo, ix = 0;       // tap = index into the weight pack: (it * kh + iy) * kw + ix
    if (k_begin < k_end) {
        cc = k_begin / ntaps; int rem = k_begin - cc * ntaps;
        if (p.pm) { iy = iy_lo + rem / p.kw; ix = rem - (iy - iy_lo) * p.kw; }
        else {
            it = rem / (p.kh * p.kw); rem -= it * p.kh * p.kw;
            iy = rem / p.kw; ix = rem - iy * p.kw;
        }
        tap = (it * p.kh + iy) * p.kw + ix;
    }
    auto dma = [&](int buf) __attribute__((always_inline)) {
        const int dt_ = it - (p.kt >> 1), dy_ = iy - (p.kh >> 1), dx_ = ix - (p.kw >> 1);
        const bool cv_ = cc * BK + q * E16 < p.C;
        // wave-uniform part of the offset: tap shift + channel chunk
        const unsigned udelta_ = (unsigned)(((dt_ * p.H + dy_) * p.W + dx_) * (int)ldb + cc * 64);
        char* abase_ = &smem[buf][wu * 1024];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int yy_ = ay[i] + dy_, xx_ = ax[i] + dx_, tt_ = at[i] + dt_;
            const bool ok_ = av[i] && cv_ && (unsigned)yy_ < (unsigned)p.H && (unsigned)xx_ < (unsigned)p.W &&
                             (unsigned)tt_ < (unsigned)p.T;
            unsigned off_ = aoff[i] + udelta_;
            if (p.up2) {
                const int f_ = p.logW >= 0 ? am[i] >> (p.logW + p.logH) : am[i] / (p.H * p.W);
                off_ = (unsigned)(((f_ + dt_) * p.Hin + (yy_ >> 1)) * p.Win + (xx_ >> 1) - base_row) * ldb + cc * 64 + q * 16;
            }
            dma16(rin, abase_ + i * (NWAVE * 1024), ok_ ? off_ : 0xffffffffu);
        }
        const unsigned uw_ = (unsigned)(tap * p.Cout) * (unsigned)p.C * (unsigned)esz + cc * 64;
        char* bbase_ = &smem[buf][ABYTES + wu * 1024];
#pragma unroll
        for (int j = 0; j < NB; ++j) dma16(rw, bbase_ + j * (NWAVE * 1024), (cov[j] && cv_) ? woff[j] + uw_ : 0xffffffffu);
        ++tap;
        if (++ix == p.kw) {
            ix = 0;
            if (++iy == iy_hi) {                        // (iy_lo, iy_hi) = (0, kh) unless pixel-major
                iy = iy_lo;
                if (++it == p.kt) { it = 0; ++cc; }
                tap = (it * p.kh + iy) * p.kw;
            }
        }
    };

    f32x16 acc[TM][2];
    {
        const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = zacc;   // (a per-element loop of 128 stores is not unrolled -> scratch)
    }

    const int arow = wm * (TM * 32) + (lane & 31), brow = wn * 64 + (lane & 31);
#ifdef DVD_EXP_NOMAIN          // compile-time measurement variant: the main loop is skipped
    constexpr bool kMain = false;
#else
    constexpr bool kMain = true;
#endif
    if (kMain && k_begin < k_end) {
        // 3-stage ring, two tiles in flight.  Waits are COUNTED (vmcnt(NA+2) keeps the youngest tile's
        // DMAs outstanding) and the barrier is the raw s_barrier: __syncthreads() would make hipcc drain
        // vmcnt(0) while an LDS-DMA is pending and collapse the pipeline to depth 1.
        constexpr int kDmaPerTile = NA + NB;
        constexpr int kWaitOne = (kDmaPerTile & 0xf) | (7 << 4) | (0xf << 8) | ((kDmaPerTile >> 4) << 14);
        constexpr int kWaitAll = 0 | (7 << 4) | (0xf << 8);
        const int nsteps = k_end - k_begin;
        dma(0);
        if (nsteps > 1) {
            dma(1);
            __builtin_amdgcn_s_waitcnt(kWaitOne);
        } else {
            __builtin_amdgcn_s_waitcnt(kWaitAll);
        }
        __builtin_amdgcn_s_barrier();
        int st = 0, st2 = 2;                                   // stage of tile s, stage of tile s+2
        constexpr bool late = true;
        for (int sidx = 0; sidx < nsteps; ++sidx) {
            const bool issue = sidx + 2 < nsteps;
            // The (slow to issue) LDS-DMAs of a step go out AFTER its MFMA block, while the matrix pipe drains, instead of in
            // front of it where they delay the first fragment reads: +2 % on the 8 x 8 recurrent convolutions (first form:
            // only the second half of the waves of the 8-wave tile did this), see also conv_halo_kernel.
            if (issue && !late) dma(st2);                      // stage st2 was last read in step sidx-1
            mma_swz<T, TM, RELU>(&smem[st][0], &smem[st][ABYTES], arow, brow, lane, acc);
            if (issue && late) dma(st2);
            if (issue) __builtin_amdgcn_s_waitcnt(kWaitOne);   // tile sidx+1 has landed (this wave's part)
            else __builtin_amdgcn_s_waitcnt(kWaitAll);
            __builtin_amdgcn_s_barrier();                      // ... and every other wave's part
            st = st == NSTAGE - 1 ? 0 : st + 1;
            st2 = st2 == NSTAGE - 1 ? 0 : st2 + 1;
        }
    }
    __syncthreads();

    // ---- epilogue: accumulators -> LDS (per wave, 32 rows x 64 columns at a time) -> 8-column vectors.
    // Going through LDS keeps the register->LDS part trivially unrollable (a large branchy epilogue
    // indexed by acc[][][r] is NOT fully unrolled by hipcc and then parks all accumulators in scratch,
    // 6x slower) and turns the global stores into coalesced 16/32-byte vectors.
    float* ep = reinterpret_cast<float*>(&smem[0][0]) + wave * (32 * 64);
    const int ecol = (lane & 7) * 8, erow = lane >> 3;
    if (p.pm) {          // rows of the tile are scattered over the tensor: offsets from its start (small tensors only, see conv_plan)
        conv_epilogue<T, TM, 0, WN == 2>(p, acc, ep, lane, n0 + wn * 64 + ecol, z, (int)blockIdx.x, 0ll, [&](int tm, int j) __attribute__((always_inline)) {
            const int mp = m0 + wm * (TM * 32) + tm * 32 + j * 8 + erow;
            return mp < p.M ? memrow(mp) : -1;
        });
        return;
    }
    // (WN == 2: the 8-wave tile has no registers to spare for the in-launch split-K combine)
    conv_epilogue<T, TM, 0, WN == 2>(p, acc, ep, lane, n0 + wn * 64 + ecol, z, (int)blockIdx.x, (long long)m0, [&](int tm, int j) __attribute__((always_inline)) {
        const int rr = wm * (TM * 32) + tm * 32 + j * 8 + erow;
        return m0 + rr < p.M ? rr : -1;
    });
}

// waves: WMV (M) x WN (N); block tile (WMV*TM*32 pixels) x (WN*64).  WMV = 4, TM = 2, WN = 1 is the 256 x 64 tile of
// the thin convs (Cout <= 64: a 128-wide N tile would multiply zeros half of the time)
// compile-time geometry of one halo-kernel variant
template <typename T, int TM, int WN, bool UP2, int WMV> struct HaloCfg {
    static constexpr int PITCH = HaloGeo<UP2>::PITCH;
    static constexpr int BMt = WMV * TM * 32, BNt = WN * 64;
    static constexpr int NWAVE = WMV * WN;
    static constexpr int PH = BMt / 16;                              // patch: PH rows x 16 columns of one frame
    static constexpr int HG = UP2 ? ((PH / 2 + 3) * PITCH + 15) / 16 // 16-row DMA groups of the largest footprint
                                  : ((PH + 4) * PITCH + 15) / 16;    //   (5 x 5 taps)
    static constexpr int HBYTES = HG * 1024, BBYTES = BNt * 64;
    // weight ring depth: NSTAGE-1 tiles in flight.  4 where LDS allows (the 256 x 128 variant runs 2 workgroups per CU)
    // (a fourth stage for the 3 x 3 filters, whose smaller footprint leaves room for it at two workgroups per CU, measured
    //  +0.6 %: the ring depth is not what limits the loop)
    static constexpr int NSTAGE = (TM == 4 && WN == 2) ? 3 : 4;
    static constexpr int EPI = NWAVE * 32 * 64 * 4;
    static constexpr int LDSB = 2 * HBYTES + 1024 + NSTAGE * BBYTES > EPI ? 2 * HBYTES + 1024 + NSTAGE * BBYTES : EPI;
};

// One output tile (M tile mt = a patch of one frame, N tile nt, K slice z) of the halo-staged convolution; `smem`: the
// workgroup's LDS block of HaloCfg::LDSB bytes (the ONLY __shared__ object of the calling kernel: a second one makes hipcc
// drain vmcnt before the LDS reads of every K step).  conv_halo_kernel calls it once per workgroup; a caller that runs
// several tiles in one workgroup (the persistent time-loop experiment of round 3) places a workgroup
// barrier between two tiles.
template <typename T, int TM, int WN, bool RELU, bool UP2, int WMV = 2>
__device__ __forceinline__ void conv_halo_tile(const ConvK& p, char* const smem, const int mt, const int nt, const int z) {
    using G = HaloGeo<UP2>;
    using Cfg = HaloCfg<T, TM, WN, UP2, WMV>;
    constexpr int PITCH = G::PITCH;
    constexpr int E16 = ElemTraits<T>::kPer16B;
    constexpr int BK = 4 * E16;
    constexpr int BMt = Cfg::BMt, BNt = Cfg::BNt;
    constexpr int NWAVE = Cfg::NWAVE;
    constexpr int PH = Cfg::PH;
    constexpr int HG = Cfg::HG;
    constexpr int NH = (HG + NWAVE - 1) / NWAVE;              // footprint DMAs per wave
    constexpr int NB = BNt / 16 / NWAVE;                      // weight-tile DMAs per wave
    constexpr int HBYTES = Cfg::HBYTES, BBYTES = Cfg::BBYTES;
    constexpr int NSTAGE = Cfg::NSTAGE;
    char* const hbuf0 = &smem[0];
    char* const dump = &smem[2 * HBYTES];                     // landing zone of the DMAs of non-existent groups
    char* const bring = &smem[2 * HBYTES + 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = nt * BNt;
    // tile -> (frame, patch origin)
    const int pw = p.W >> 4, ppf = pw * (p.H / PH);
    const int ft = mt / ppf, pidx = mt - ft * ppf;
    const int y0 = (pidx / pw) * PH, x0 = (pidx % pw) * 16;
    const int nouter = p.kchunks * p.kt;                      // outer index oc = cc * kt + it
    const int per = (nouter + p.nsplit - 1) / p.nsplit;
    const int oc_begin = z * per, oc_end = min(nouter, oc_begin + per);
    const int ntap2 = p.kh * p.kw;
    const unsigned esz = sizeof(T);
    const int pad = p.kh >> 1, cpad = (pad + 1) >> 1;
    // footprint extent in INPUT coordinates
    const int HWa = UP2 ? ((15 + pad) >> 1) + cpad + 1 : 16 + 2 * pad;
    const int HHa = UP2 ? ((PH - 1 + pad) >> 1) + cpad + 1 : PH + 2 * pad;
    const int iy_lo = UP2 ? (y0 >> 1) - cpad : y0 - pad, ix_lo = UP2 ? (x0 >> 1) - cpad : x0 - pad;

    const int wu = __builtin_amdgcn_readfirstlane(wave);
    const int lrow = lane >> 2;
    const unsigned ldb = (unsigned)p.ldi * esz;
    const int tt = p.kt > 1 ? ft % p.T : 0;                   // time index of this tile's frame
    const int base_frame = max(0, ft - (p.kt >> 1));
    const size_t fbytes = (size_t)p.Hin * p.Win * ldb;
    const size_t base_b = (size_t)base_frame * fbytes;
    const size_t left_b = p.in_bytes - base_b;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.in + base_b), 0, left_b > 0xfffffffeull ? 0xfffffffeu : (unsigned)left_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    unsigned hoff[NH];
    int hq[NH];                                               // logical 16-byte slot this lane fetches, per group
    bool hval[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int g = i * NWAVE + wu;
        const int h = g * 16 + lrow;
        const int hy = h / PITCH, hx = h - hy * PITCH;
        const int yin = iy_lo + hy, xin = ix_lo + hx;
        hq[i] = (lane & 3) ^ G::sw(hy, hx);
        hval[i] = g < HG && hy < HHa && hx < HWa && (unsigned)yin < (unsigned)p.Hin && (unsigned)xin < (unsigned)p.Win;
        hoff[i] = (unsigned)(yin * p.Win + xin) * ldb + hq[i] * 16;
    }
    const int q = (lane & 3) ^ ((lane >> 4) & 3);             // weight tile: row-index swizzle as in conv_igemm_kernel
    int cob[NB];
    bool cov[NB];
    unsigned woff[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        cob[j] = n0 + (j * NWAVE + wu) * 16 + lrow; cov[j] = cob[j] < p.Cout;
        woff[j] = ((unsigned)cob[j] * p.C + q * E16) * esz;
    }
    // footprint of the outer index (cc_, it_) -> halo buffer hb.  (The (chunk, dt) pairs are kept as running counters: an
    // integer division per K step costs ~20 scalar instructions on the wave that is about to issue its MFMAs.)
    auto dmaH = [&](int hb, int cc_, int it_) __attribute__((always_inline)) {
        const int dt_ = it_ - (p.kt >> 1);
        const bool ok_ = (unsigned)(tt + dt_) < (unsigned)p.T || p.kt == 1;
        const unsigned ud_ = (unsigned)(ft + dt_ - base_frame) * (unsigned)fbytes + cc_ * 64;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int g = i * NWAVE + wu;
            char* dst_ = g < HG ? hbuf0 + hb * HBYTES + g * 1024 : dump;
            const bool cv_ = cc_ * BK + hq[i] * E16 < p.C;
            dma16(rin, dst_, (hval[i] && ok_ && cv_) ? hoff[i] + ud_ : 0xffffffffu);
        }
    };
    // weight tile of (chunk, tap): running iterator, two steps ahead of the multiply
    int b_cc = oc_begin / p.kt, b_it = oc_begin - b_cc * p.kt, b_tap = 0;
    auto dmaB = [&](int stg) __attribute__((always_inline)) {
        const bool cv_ = b_cc * BK + q * E16 < p.C;
        const unsigned uw_ = (unsigned)((b_it * ntap2 + b_tap) * p.Cout) * (unsigned)p.C * esz + b_cc * 64;
        char* bbase_ = bring + stg * BBYTES + wu * 1024;
#pragma unroll
        for (int j = 0; j < NB; ++j) dma16(rw, bbase_ + j * (NWAVE * 1024), (cov[j] && cv_) ? woff[j] + uw_ : 0xffffffffu);
        if (++b_tap == ntap2) { b_tap = 0; if (++b_it == p.kt) { b_it = 0; ++b_cc; } }
    };

    f32x16 acc[TM][2];
    {
        const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = zacc;
    }

    // this lane's A rows: pixel (py0 + 2 tm, px) of the patch for its TM 32-row sub-tiles.  Sub-tile tm sits
    // 2 patch lines below sub-tile 0 -> a compile-time LDS offset (2 * PITCH rows; PITCH rows after the x2 fold)
    const int l31 = lane & 31, kh2 = lane >> 5;
    const int px = l31 & 15, py0 = wm * (TM * 2) + (l31 >> 4);
    constexpr int TMSTRIDE = (UP2 ? 1 : 2) * PITCH * 64;
    const int brow = wn * 64 + l31;
#ifdef DVD_EXP_NOMAIN
    const int nsteps = 0;
#else
    const int nsteps = (oc_end - oc_begin) * ntap2;
#endif
    if (nsteps > 0) {
#define VMCNT(n) (((n) & 0xf) | (7 << 4) | (0xf << 8) | (((n) >> 4) << 14))
        constexpr int FL = NSTAGE - 2;                         // weight tiles allowed to stay in flight past tile s+1
        int h_cc = b_cc, h_it = b_it;                          // outer index of the NEXT footprint to fetch
        dmaH(0, h_cc, h_it);
        if (++h_it == p.kt) { h_it = 0; ++h_cc; }
#pragma unroll
        for (int i = 0; i < NSTAGE - 1; ++i)
            if (i < nsteps) dmaB(i);
        if (nsteps > FL) __builtin_amdgcn_s_waitcnt(VMCNT(FL * NB));     // footprint + tile 0 landed
        else __builtin_amdgcn_s_waitcnt(VMCNT(0));
        __builtin_amdgcn_s_barrier();
        // The DMAs of a step go out AFTER its MFMA block (all ds_reads issued, the matrix pipe still draining): +2...7 % on the
        // 5 x 5 shapes, +0...3 % on 3 x 3 against issuing them first.  Measured alternatives: in the middle of the block (after
        // unit 1 / 3 / 5 of 8) -7 %; every other wave or every other workgroup late -4...-8 %.
        constexpr bool late = true;
        int st = 0, st2 = NSTAGE - 1, hb = 0, m_oc = oc_begin, m_tap = 0, iy = 0, ix = 0;
        for (int sidx = 0; sidx < nsteps; ++sidx) {
            const bool issueB = sidx + NSTAGE - 1 < nsteps;
            const bool issueH = m_tap == 0 && m_oc + 1 < oc_end;
            if (!late) {
                if (issueB) dmaB(st2);                         // stage st2 was last read in step sidx-1
                if (issueH) { dmaH(hb ^ 1, h_cc, h_it); if (++h_it == p.kt) { h_it = 0; ++h_cc; } }
            }
            {
                // footprint row of this lane's sub-tile 0 for tap (iy, ix)
                const int hy = UP2 ? ((py0 + iy - pad) >> 1) + cpad : py0 + iy;
                const int hx = UP2 ? ((px + ix - pad) >> 1) + cpad : px + ix;
                const int swz = G::sw(hy, hx);
                const char* Ah = hbuf0 + hb * HBYTES + (hy * PITCH + hx) * 64;
                const char* Bt = bring + st * BBYTES;
                if constexpr (sizeof(T) == 2) {
                    // Explicit software pipeline over the 2*TM (k-half, sub-tile) units: the fragment of unit
                    // u+2 is requested right before the two MFMAs of unit u, pinned with sched_barrier.  (Left
                    // alone hipcc recycles ONE A register set and waits lgkmcnt(0) before every MFMA pair;
                    // all 12 reads up front instead stall on LDS issue: SQ_WAIT_INST_LDS 7x.)
                    constexpr int NU = 2 * TM;
                    bf16x8 b[2][2], a[NU];
                    auto ldB = [&](int kk) __attribute__((always_inline)) {
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn)
                            b[kk][tn] = *reinterpret_cast<const bf16x8*>(Bt + lds_off(brow + tn * 32, kk * 2 + kh2));
                    };
                    auto ldA = [&](int u) __attribute__((always_inline)) {
                        const int kk = u / TM, tm = u % TM, slot = kk * 2 + kh2;
                        // after the x2 fold hy advances by 1 per sub-tile and SWA = 2: odd sub-tiles flip slot bit 1
                        const int sl = (UP2 && (tm & 1)) ? (slot ^ 2) : slot;
                        a[u] = *reinterpret_cast<const bf16x8*>(Ah + tm * TMSTRIDE + ((sl ^ swz) << 4));
                    };
                    ldB(0); ldA(0); ldA(1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        if (u + 2 < NU) {
                            if ((u + 2) % TM == 0) ldB((u + 2) / TM);
                            ldA(u + 2);
                        }
                        const int kk = u / TM, tm = u % TM;
                        if constexpr (RELU) a[u] = __builtin_bit_cast(bf16x8, relu16_bf16(__builtin_bit_cast(u32x4, a[u])));
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[kk][tn], acc[tm][tn], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        const int k = kk * 2 + kh2;
                        float b[2], a[TM];
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn)
                            b[tn] = *reinterpret_cast<const float*>(Bt + lds_off(brow + tn * 32, k >> 2) + (k & 3) * 4);
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm) {
                            const int sl = (UP2 && (tm & 1)) ? ((k >> 2) ^ 2) : (k >> 2);
                            a[tm] = *reinterpret_cast<const float*>(Ah + tm * TMSTRIDE + ((sl ^ swz) << 4) + (k & 3) * 4);
                            if constexpr (RELU) a[tm] = fmaxf(a[tm], 0.f);
                        }
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                            for (int tn = 0; tn < 2; ++tn)
                                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
                    }
                }
            }
#ifndef DVD_EXP_NODMA          // DVD_EXP_*: compile-time measurement variants (tools/build_variant.sh), results are garbage
            if (late) {
                if (issueB) dmaB(st2);
                if (issueH) { dmaH(hb ^ 1, h_cc, h_it); if (++h_it == p.kt) { h_it = 0; ++h_cc; } }
            }
#endif
            // In-order completion: weight tile sidx+1 has landed once at most the FL younger tiles (plus, for the
            // NSTAGE-1 steps after a footprint went out behind tile sidx+NSTAGE-1, that footprint) are outstanding.
#ifndef DVD_EXP_NOWAIT
            {
                const int younger = min(FL, nsteps - 2 - sidx);          // tiles issued beyond sidx+1
                const bool hpend = m_tap < NSTAGE - 1 && m_oc + 1 < oc_end;
                if (younger == FL) {
                    if (hpend) __builtin_amdgcn_s_waitcnt(VMCNT(FL * NB + NH));
                    else __builtin_amdgcn_s_waitcnt(VMCNT(FL * NB));
                } else if (FL == 2 && younger == 1) {
                    __builtin_amdgcn_s_waitcnt(VMCNT(NB));
                } else {
                    __builtin_amdgcn_s_waitcnt(VMCNT(0));
                }
            }
#endif
#ifndef DVD_EXP_NOBAR
            __builtin_amdgcn_s_barrier();
#endif
            st = st == NSTAGE - 1 ? 0 : st + 1;
            st2 = st2 == NSTAGE - 1 ? 0 : st2 + 1;
            ++m_tap;
            if (++ix == p.kw) { ix = 0; ++iy; }
            if (m_tap == ntap2) { m_tap = 0; iy = 0; ix = 0; ++m_oc; hb ^= 1; }
        }
#undef VMCNT
    }
    __syncthreads();

    float* ep = reinterpret_cast<float*>(&smem[0]) + wave * (32 * 64);
    const int ecol = (lane & 7) * 8, erow = lane >> 3;
    const long long frame_row0 = (long long)ft * (p.H * p.W);
    conv_epilogue<T, TM>(p, acc, ep, lane, n0 + wn * 64 + ecol, z, (int)blockIdx.x, frame_row0, [&](int tm, int j) __attribute__((always_inline)) {
        const int pi = wm * (TM * 32) + tm * 32 + j * 8 + erow;          // pixel of the patch, row-major 16 wide
        return (y0 + (pi >> 4)) * p.W + x0 + (pi & 15);
    });
}

template <typename T, int TM, int WN, bool RELU, bool UP2, int WMV = 2>
// (bf16: two workgroups per CU, i.e. at most 256 registers incl. the accumulators -- without the bound hipcc moves all 128 accumulators
//  into arch VGPRs for the in-launch split-K combine: 348 registers, one workgroup per CU)
__global__ __launch_bounds__(64 * WMV * WN, sizeof(T) == 2 ? 2 : 1) void conv_halo_kernel(ConvK p) {
    __shared__ __attribute__((aligned(16))) char smem[HaloCfg<T, TM, WN, UP2, WMV>::LDSB];
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, qd = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (qd + 1) : rr * (qd + 1) + (xcd - rr) * qd) + (bid >> 3);
    }
    int mt = bid / p.tilesN, nt = bid - mt * p.tilesN;
    if (p.nmajor) { const int tilesM = gridDim.x / p.tilesN; nt = bid / tilesM; mt = bid - nt * tilesM; }
    conv_halo_tile<T, TM, WN, RELU, UP2, WMV>(p, smem, mt, nt, blockIdx.z);
}

// ============================================================================ weight packing
struct PackK {
    const float* w; const float* sigma; char* wf; char* wd;
    int Cout, Cin, ntaps, Cip, co_off, co_tot_f, co_tot_d, kt, kh, kw, ci_off, ci_tot;
};
// one thread per (co, ci_pad, tap) of the forward pack; writes both packs
template <typename T>
__global__ void pack_weight_kernel(PackK p) {
    const long long n = (long long)p.Cout * p.Cip * p.ntaps;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int tap = (int)(i % p.ntaps);
    const long long r = i / p.ntaps;
    const int ci = (int)(r % p.Cip), co = (int)(r / p.Cip);
    float v = 0.f;
    if (ci < p.Cin) {
        v = p.w[((size_t)co * p.ci_tot + p.ci_off + ci) * p.ntaps + tap];
        if (p.sigma) v = v / *p.sigma;
    }
    if (p.wf) stf(reinterpret_cast<T*>(p.wf) + ((size_t)tap * p.co_tot_f + p.co_off + co) * p.Cip + ci, v);
    if (p.wd) {
        const int ftap = p.ntaps - 1 - tap;     // flipping every axis == reversing the flat tap index
        stf(reinterpret_cast<T*>(p.wd) + ((size_t)ftap * p.Cip + ci) * p.co_tot_d + p.co_off + co, v);
    }
}

// n packs in one launch: block b serves item j with first[j] <= b < first[j + 1] (whole blocks per item)
constexpr int kPackBatch = 24;
struct PackBatchK { PackK it[kPackBatch]; int first[kPackBatch + 1]; int n; };
template <typename T>
__global__ void pack_weight_batched_kernel(PackBatchK b) {
    int j = 0;
    while (j + 1 < b.n && (int)blockIdx.x >= b.first[j + 1]) ++j;
    const PackK& p = b.it[j];
    const long long n = (long long)p.Cout * p.Cip * p.ntaps;
    const long long i = (long long)((int)blockIdx.x - b.first[j]) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int tap = (int)(i % p.ntaps);
    const long long r = i / p.ntaps;
    const int ci = (int)(r % p.Cip), co = (int)(r / p.Cip);
    float v = 0.f;
    if (ci < p.Cin) {
        v = p.w[((size_t)co * p.ci_tot + p.ci_off + ci) * p.ntaps + tap];
        if (p.sigma) v = v / *p.sigma;
    }
    if (p.wf) stf(reinterpret_cast<T*>(p.wf) + ((size_t)tap * p.co_tot_f + p.co_off + co) * p.Cip + ci, v);
    if (p.wd) {
        const int ftap = p.ntaps - 1 - tap;
        stf(reinterpret_cast<T*>(p.wd) + ((size_t)ftap * p.Cip + ci) * p.co_tot_d + p.co_off + co, v);
    }
}

}  // namespace

// ============================================================================ optional profiling
// bench.py needs the average duration of the dominant kernel measured with HIP events on the
// launch stream.  When enabled, every conv launch is bracketed by an event pair; dvd_prof_report
// synchronises and sums them.  Off by default (no events, no global state touched).
#include <vector>
#include <mutex>
#include <cstdio>
#include <cstdlib>
namespace dvdprof {
bool g_prof = false;
std::vector<ProfRec> g_recs;
std::mutex g_prof_mu;
}  // namespace dvdprof
using namespace dvdprof;
extern "C" void dvd_prof_enable(int on) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    g_prof = on != 0;
}
// kind 0 = conv_igemm (forward / backward-data), 1 = conv_wgrad.  Drains the records of `kind` and returns the number of
// launches; n / ms / flops (each [nvar] or NULL) receive the per-variant totals -- kind 0: 1 = conv_halo 256 x 128,
// 2 = conv_halo 128 x 128, 3 = conv_halo 256 x 64 (thin outputs), 4 = conv_igemm 128 x 128, 5 = conv_igemm 256 x 128,
// 6 = conv_igemm 256 x 256 (8 waves), 7 / 8 = whole-frame footprint kernel (4 x 4 / 8 x 8 frames) 256 x 128 / 128 x 128, 9 = thin-input kernel (conv_thin.hip); kind 1: 1 = filter-row kernel, 2 = one-tap kernel, 3 = thin-end kernel (wgrad_thin.hip); index 0 = everything.
// If the environment variable DVD_PROF_CSV is set, every drained record is appended to that file.
extern "C" long long dvd_prof_report_variants(int kind, int nvar, long long* n, double* ms, double* flops) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    for (int v = 0; v < nvar; ++v) { if (n) n[v] = 0; if (ms) ms[v] = 0; if (flops) flops[v] = 0; }
    long long total = 0;
    std::vector<ProfRec> keep;
    const char* csv = getenv("DVD_PROF_CSV");
    FILE* f = csv ? fopen(csv, "a") : nullptr;
    for (auto& r : g_recs) {
        if (r.kind != kind) { keep.push_back(r); continue; }
        hipEventSynchronize(r.b);
        float t = 0; hipEventElapsedTime(&t, r.a, r.b);
        if (f) fprintf(f, "%d,%lld,%d,%d,%d,%d,%d,%.4f,%.0f,%d\n", r.kind, r.M, r.C, r.Cout, r.taps, r.split, r.flags, t, r.flops, r.variant);
        const int slots[2] = {0, r.variant};                    // slot 0 = all launches, plus the record's own slot
        for (int j = 0; j < (r.variant > 0 ? 2 : 1); ++j) {
            const int v = slots[j];
            if (v >= nvar) continue;
            if (n) ++n[v];
            if (ms) ms[v] += t;
            if (flops) flops[v] += r.flops;
        }
        ++total;
        hipEventDestroy(r.a); hipEventDestroy(r.b);
    }
    if (f) fclose(f);
    g_recs.swap(keep);
    return total;
}
extern "C" long long dvd_prof_report(int kind, double* total_ms, double* total_flops) {
    long long n = 0;
    return dvd_prof_report_variants(kind, 1, &n, total_ms, total_flops);
}

// ============================================================================ C ABI
extern "C" int dvd_conv_forward(const dvd_conv_desc* d, void* stream) { return dvd_conv_forward_gru(d, nullptr, stream); }

// Validates a forward / backward-data request and derives the kernel parameters and the variant that serves it.
struct ConvPlan { long long M; bool halo, thin, wide, big, smallf; };
static int conv_plan(const dvd_conv_desc* d, const GruEpi* g, ConvK& p, ConvPlan& pl) {
    if (!d || !d->in || !d->w || (!d->ws && (!d->out || (d->nsplit > 1 && !g)))) return DVD_E_ARG;
    if (g && (d->ws || (g->h & 7) || (d->nsplit > 1 && (!g->slabs || !g->tickets)))) return DVD_E_ARG;
    if (d->frames <= 0 || d->T <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->Cout <= 0) return DVD_E_ARG;
    int logH = ilog2_exact(d->H), logW = ilog2_exact(d->W);
    const bool pow2 = logH >= 0 && logW >= 0;
    if (!pow2) logH = logW = -1;           // any other extent: division-based indexing in the tap-by-tap kernels (grid_pos)
    if ((d->C & 7) || (d->ldi & 7) || !(d->kt & d->kh & d->kw & 1)) return DVD_E_SHAPE;
    if (d->up2 && ((d->H | d->W) & 1)) return DVD_E_SHAPE;
    if (d->kt > 1 && d->up2) return DVD_E_SHAPE;
    const long long M = (long long)d->frames * d->T * d->H * d->W;
    if (M >= (1ll << 31) - BM) return DVD_E_SHAPE;
    p.in = (const char*)d->in; p.w = (const char*)d->w; p.bias = d->bias; p.res = (const char*)d->res;
    p.mask = (const char*)d->mask; p.out = (char*)d->out; p.ws = d->ws; p.wq = (const char*)d->wq;
    p.nb32 = (d->Cout + 127) / 128 * 4;
    p.M = (int)M; p.C = d->C; p.ldi = d->ldi; p.Cout = d->Cout; p.ldo = d->ldo; p.ldres = d->ldres; p.ldmask = d->ldmask;
    p.res_up2 = d->res && d->res_up2;
    if (p.res_up2 && (d->H < 2 || d->W < 2 || ((d->H | d->W) & 1))) return DVD_E_SHAPE;
    p.T = d->T; p.H = d->H; p.W = d->W; p.logH = logH; p.logW = logW;
    p.Hin = d->up2 ? d->H / 2 : d->H; p.Win = d->up2 ? d->W / 2 : d->W;
    p.kt = d->kt; p.kh = d->kh; p.kw = d->kw;
    const int bk = d->dtype == DVD_BF16 ? 32 : 16;
    p.kchunks = (d->C + bk - 1) / bk;
    p.nk = d->kt * d->kh * d->kw * p.kchunks;
    p.nsplit = d->nsplit < 1 ? 1 : d->nsplit;
    if (p.nsplit > p.nk) p.nsplit = p.nk;
    if (p.nsplit != (d->nsplit < 1 ? 1 : d->nsplit)) return DVD_E_ARG;   // caller sized ws for d->nsplit slabs
    // 8-wave 256 x 256 tile when the output is wide enough (no more than 1/8 of the last N tile wasted)
    // and there is at least one tile per CU; else 256 x 128 (>= 2 tiles per CU) or 128 x 128.
    const long long t256 = cdiv(M, 256) * (long long)cdiv(d->Cout, 256) * p.nsplit;
    const int rem256 = d->Cout % 256;
    // halo-staged kernel: square 3x3 / 5x5 (x3) filters on frames at least one 16 x 16 patch large.  It runs as
    // 4-wave workgroups only: with the activation DMAs gone, two 256 x 128 workgroups per CU beat one 8-wave
    // 256 x 256 workgroup (1.13-1.34 vs 0.86-1.12 PF/s on the S = 16 / 32 shapes of config C2).
    const bool halo = pow2 && d->kh == d->kw && (d->kh == 3 || d->kh == 5) && d->W >= 16 && d->H >= 16 &&
                      p.nsplit <= p.kchunks * d->kt;
    // frames of 4 x 4 / 8 x 8 pixels (bf16, 2-D taps, no upsample): whole-frame footprints in LDS, weights from L2 -- needs the
    // fragment-major image
    const bool smallf = !halo && pow2 && d->dtype == DVD_BF16 && d->wq && d->kt == 1 && d->T == 1 && !d->up2 &&
                        d->kh == d->kw && (d->kh == 3 || d->kh == 5) && d->H == d->W && (d->W == 4 || d->W == 8) &&
                        p.nsplit <= p.kchunks && M < (1ll << 24);
    const bool thin = halo && d->dtype == DVD_BF16 && d->Cout <= 64 && cdiv(M, 256) * (long long)p.nsplit >= 512;
    const bool wide = !halo && !smallf && d->dtype == DVD_BF16 && d->Cout >= 256 && (rem256 == 0 || rem256 > 224) && t256 >= 256 &&
                      !(g && p.nsplit > 1);
    p.tilesN = wide ? (d->Cout + 255) / 256 : thin ? 1 : (d->Cout + BN - 1) / BN;
    p.up2 = d->up2; p.relu_in = d->relu_in; p.act = d->act; p.out_f32 = d->out_f32;
    if (g) p.g = *g; else p.g = GruEpi{};
    p.nmajor = 0;
    {   // extents of the two buffer descriptors (32-bit byte offsets): tensors must stay below 4 GiB
        const size_t esz = d->dtype == DVD_BF16 ? 2 : 4;
        const size_t rows_in = (size_t)d->frames * d->T * p.Hin * p.Win;
        const size_t inb = ((rows_in - 1) * (size_t)d->ldi + d->C) * esz;
        const size_t wb = (size_t)d->kt * d->kh * d->kw * d->Cout * d->C * esz;
        if (wb >= 0xffffffffull) return DVD_E_SHAPE;
        p.in_bytes = inb; p.w_bytes = (unsigned)wb;
        p.maxshift = ((d->kt >> 1) * d->H + (d->kh >> 1)) * d->W + (d->kw >> 1);
        p.nmajor = !halo && wb > inb && wb > (4u << 20);      // weights beyond one L2 (also for the small-frame halo kernel)
        // halo kernel: with several N tiles and weights well beyond one L2, a run of workgroups that shares the weight tile
        // (and streams the activations once per N tile) misses less than one that shares the footprint and cycles through
        // all the weights: 1272 -> 1355 TF/s on 786 k x 256 -> 1536 (19.7 MB of weights); neutral from 5 MiB, -1.5 % at 4.9 MB
        if (halo && p.tilesN > 1 && wb > ((size_t)5 << 20)) p.nmajor = 1;
        if (halo && p.tilesN >= 6 && wb > (2u << 20)) p.nmajor = 1;      // 3.1 M x 128 -> 768 (4.9 MB, 6 N tiles): +2 %
    }
    // 256-row tiles when they still give every CU work; 128-row tiles for the small recurrent convs
    constexpr long long big_thr = 512;          // (swept 128 / 256 / 512)
    // 1 x 1 filters are HBM-bound streams with 2-8 K steps: the 256-row tap-by-tap tile holds 433 registers = ONE workgroup per CU, the
    // 128-row tile 240 = two.  3.1 M x 128 -> 128: 665 -> 505 us, 128 -> 64: 691 -> 420, 64 -> 128: 574 -> 422; step 493.4 -> 490.9 ms
    const bool one_tap = d->kt * d->kh * d->kw == 1 && !halo && !smallf;
    const bool big = wide || (cdiv(M, 256) * (long long)p.tilesN * p.nsplit >= big_thr && !one_tap);   // >= 2 workgroups per CU
    p.pm = 0;
    {   // pixel-major row order for the tap-by-tap kernel on small frames: the rows of a tile must share their image line
        const int bmt = (wide || big) ? 256 : 128;
        const int F = d->frames;
        const bool lines = (F % bmt == 0) || (bmt % F == 0 && d->W % (bmt / F) == 0);
        const size_t esz = d->dtype == DVD_BF16 ? 2 : 4;
        if (!halo && !smallf && pow2 && d->kt == 1 && d->T == 1 && !d->up2 && d->kh >= 3 && d->H <= 8 && lines &&
            (size_t)M * (size_t)(d->ldi > 3 * d->Cout ? d->ldi : 3 * d->Cout) * 4 < (1ull << 31))     // every epilogue offset from row 0 fits
            p.pm = F;
    }
    pl.M = M; pl.halo = halo; pl.thin = thin; pl.wide = wide; pl.big = big && !(smallf && d->W == 4); pl.smallf = smallf;
    p.pool2 = d->pool2 != 0;
    if (p.pool2 && (!halo || p.nsplit != 1 || g || d->ws || d->res || d->act != DVD_ACT_NONE || d->kt != 1 || d->T != 1 || d->up2))
        return DVD_E_SHAPE;                   // (dvd_conv_pool2_ok: the 2 x 2 sums are an epilogue of the halo-staged kernels' patch tiles)
    return DVD_OK;
}

extern "C" int dvd_conv_pool2_ok(const dvd_conv_desc* d) {
    if (!d || !d->pool2) return 0;
    ConvK p; ConvPlan pl;
    dvd_conv_desc t = *d;
    static char dummy;
    if (!t.out) t.out = &dummy;               // (a geometry query: the output may not exist yet)
    return conv_plan(&t, nullptr, p, pl) == DVD_OK ? 1 : 0;
}

extern "C" int dvd_conv_forward_gru(const dvd_conv_desc* d, const GruEpi* g, void* stream) {
    ConvK p; ConvPlan pl;
    const int rc = conv_plan(d, g, p, pl);
    if (rc != DVD_OK) return rc;
    const long long M = pl.M;
    const bool halo = pl.halo, thin = pl.thin, wide = pl.wide, big = pl.big;
    if (d->wq_kind >= 2 && !(d->wq_kind == 2 ? dvd_conv_thin_in_ok(d) : dvd_conv_thin_out_ok(d))) return DVD_E_ARG;   // a thin image the request cannot use
    if (!g && d->wq && d->wq_kind == 2 && dvd_conv_thin_in_ok(d)) {      // the stems / the RGB layer's backward-data pass: taps folded into K (conv_thin.hip)
        ProfScope prof(0, 2.0 * (double)M * d->Cout * d->C * d->kt * d->kh * d->kw, stream, M, d->C, d->Cout, d->kt * d->kh * d->kw, 1,
                       d->relu_in << 1);
        prof.r.variant = 9;
        return dvd_conv_thin_in(d, stream);
    }
    if (!g && d->wq && d->wq_kind == 3 && dvd_conv_thin_out_ok(d)) {     // the RGB layer / the stems' backward-data pass (conv_thin.hip)
        ProfScope prof(0, 2.0 * (double)M * d->Cout * d->C * d->kt * d->kh * d->kw, stream, M, d->C, d->Cout, d->kt * d->kh * d->kw, 1,
                       d->relu_in << 1);
        prof.r.variant = 9;
        return dvd_conv_thin_out(d, stream);
    }
    dim3 grid(cdiv(M, big ? 256 : 128) * p.tilesN, 1, p.nsplit);
    if (g && p.nsplit > 1 && grid.x > DVD_GRU_TICKETS) return DVD_E_SHAPE;      // one ticket per output tile (the small-frame grid below is no larger)
    ProfScope prof(0, 2.0 * (double)M * d->Cout * d->C * d->kt * d->kh * d->kw, stream, M, d->C, d->Cout,
                   d->kt * d->kh * d->kw, p.nsplit, d->up2 | (d->relu_in << 1) | ((d->ws != nullptr) << 2));
    hipStream_t st = (hipStream_t)stream;
    prof.r.variant = halo ? (thin ? 3 : big ? 1 : 2) : pl.smallf ? (big ? 7 : 8) : (wide ? 6 : big ? 5 : 4);
    if (pl.smallf) {
        const int S_ = d->W, Gf = (big ? 256 : 128) / (S_ * S_);
        grid = dim3(cdiv(d->frames, Gf) * p.tilesN, 1, p.nsplit);
        launch_gbs(p, S_, big, d->relu_in != 0, grid, st);
        return launch_status();
    }
    if (halo) {
#define LAUNCH_HALO2(TT, TM_, WN_, RL_)                                                             \
        do { if (d->up2) conv_halo_kernel<TT, TM_, WN_, RL_, true><<<grid, 128 * WN_, 0, st>>>(p);      \
             else conv_halo_kernel<TT, TM_, WN_, RL_, false><<<grid, 128 * WN_, 0, st>>>(p); } while (0)
#define LAUNCH_HALO(TT, TM_, WN_)                                                                   \
        do { if (d->relu_in) LAUNCH_HALO2(TT, TM_, WN_, true); else LAUNCH_HALO2(TT, TM_, WN_, false); } while (0)
        if (d->dtype == DVD_BF16 && p.wq) {           // weights straight from L2 (fragment-major image supplied)
            launch_gb(p, thin ? 2 : big ? 0 : 1, d->relu_in != 0, d->up2 != 0, grid, st);
            return launch_status();
        }
        if (thin) {
            if (d->relu_in) { if (d->up2) conv_halo_kernel<bf16_t, 2, 1, true, true, 4><<<grid, 256, 0, st>>>(p);
                              else conv_halo_kernel<bf16_t, 2, 1, true, false, 4><<<grid, 256, 0, st>>>(p); }
            else            { if (d->up2) conv_halo_kernel<bf16_t, 2, 1, false, true, 4><<<grid, 256, 0, st>>>(p);
                              else conv_halo_kernel<bf16_t, 2, 1, false, false, 4><<<grid, 256, 0, st>>>(p); }
        } else if (d->dtype == DVD_BF16) { if (big) LAUNCH_HALO(bf16_t, 4, 2); else LAUNCH_HALO(bf16_t, 2, 2); }
        else if (d->dtype == DVD_F32) { if (big) LAUNCH_HALO(float, 4, 2); else LAUNCH_HALO(float, 2, 2); }
        else return DVD_E_ARG;
#undef LAUNCH_HALO
#undef LAUNCH_HALO2
        return launch_status();
    }
#define LAUNCH_CONV(TT)                                                                       \
    do {                                                                                      \
        if (big) { if (d->relu_in) conv_igemm_kernel<TT, 4, 2, true><<<grid, NT, 0, st>>>(p);   \
                   else conv_igemm_kernel<TT, 4, 2, false><<<grid, NT, 0, st>>>(p); }           \
        else     { if (d->relu_in) conv_igemm_kernel<TT, 2, 2, true><<<grid, NT, 0, st>>>(p);   \
                   else conv_igemm_kernel<TT, 2, 2, false><<<grid, NT, 0, st>>>(p); }           \
    } while (0)
    if (wide) {
        if (d->relu_in) conv_igemm_kernel<bf16_t, 4, 4, true><<<grid, 512, 0, st>>>(p);
        else conv_igemm_kernel<bf16_t, 4, 4, false><<<grid, 512, 0, st>>>(p);
    } else if (d->dtype == DVD_BF16) LAUNCH_CONV(bf16_t);
    else if (d->dtype == DVD_F32) LAUNCH_CONV(float);
    else return DVD_E_ARG;
#undef LAUNCH_CONV
    return launch_status();
}

// Several INDEPENDENT convolutions in one launch (conv_gb.hip: group_dispatch; gru.hip: the layer wavefront of a ConvGRU stack).
// Every member: bf16, fragment-major weights supplied (d[i].wq), square 3 x 3 / 5 x 5 taps, no input ReLU / upsample, all served by
// the ONE kernel `kind` names (0 / 1 = conv_halo_gb 256 x 128 / 128 x 128 tiles: frames >= 16 pixels; 2 / 3 = whole 8 x 8 frames,
// 256- / 128-row tiles; 4 = whole 4 x 4 frames).  g[i].mode: 0 = direct epilogue, 1-5 = ConvGRU gate epilogues, 6 = direct epilogue
// behind an in-launch split-K combine; nsplit > 1 needs mode != 0 (g[i].slabs; g[i].tickets = the stream's ticket buffer: member i
// takes counters [i * 1024, (i + 1) * 1024)).  run: ConvGroup::head, workgroups of the two longest members an XCD starts with (<= 0: 32).
extern "C" int dvd_conv_forward_group(const dvd_conv_desc* d, const GruEpi* g, int n, int kind, int run, void* stream) {
    if (!d || !g || n < 1 || n > kGroupMax || kind < 0 || kind > 4) return DVD_E_ARG;
    ConvGroup grp = {};
    grp.n = n;
    double flops = 0; long long Msum = 0;
    for (int i = 0; i < n; ++i) {
        ConvPlan pl;
        const GruEpi* gi = g[i].mode ? &g[i] : nullptr;
        const int rc = conv_plan(&d[i], gi, grp.c[i], pl);
        if (rc != DVD_OK) return rc;
        ConvK& p = grp.c[i];
        if (d[i].dtype != DVD_BF16 || !p.wq || d[i].wq_kind > 1 || d[i].relu_in || d[i].up2 || d[i].ws) return DVD_E_ARG;
        long long mtiles;
        if (kind <= 1) {
            if (!pl.halo || d[i].H < (kind == 0 ? 16 : 8)) return DVD_E_SHAPE;
            mtiles = cdiv(pl.M, kind == 0 ? 256 : 128);
        } else {
            const int S = kind == 4 ? 4 : 8;
            if (!pl.smallf || d[i].W != S) return DVD_E_SHAPE;
            mtiles = cdiv(d[i].frames, kind == 2 ? 4 : kind == 3 ? 2 : 8);
        }
        p.tilesN = (d[i].Cout + BN - 1) / BN;
        const long long tiles = mtiles * p.tilesN;
        if (p.nsplit > 1) {
            if (!gi || tiles > 1024) return DVD_E_SHAPE;
            p.g.tickets += i * 1024;
        }
        if (tiles * p.nsplit > (1 << 20)) return DVD_E_SHAPE;
        grp.wgs[i] = (int)(tiles * p.nsplit);
        flops += 2.0 * (double)pl.M * d[i].Cout * d[i].C * d[i].kh * d[i].kw;
        Msum += pl.M;
    }
    // slot order (ConvGroup): members by decreasing K length of a workgroup
    {
        long long len[kGroupMax];
        for (int i = 0; i < n; ++i) { grp.order[i] = i; len[i] = (long long)grp.c[i].kchunks * d[i].kh * d[i].kw / grp.c[i].nsplit; }
        for (int i = 1; i < n; ++i)
            for (int j = i; j > 0 && len[grp.order[j]] > len[grp.order[j - 1]]; --j) { const int t = grp.order[j]; grp.order[j] = grp.order[j - 1]; grp.order[j - 1] = t; }
        long long slots = 0;
        for (int i = 0; i < n; ++i) slots += (grp.wgs[i] + 7) / 8;
        grp.nslots = (int)(8 * slots);
        grp.head = run > 0 ? run : run < 0 ? 0 : 32;
    }
    ProfScope prof(0, flops, stream, Msum, d[0].C, d[0].Cout, d[0].kh * d[0].kw, n, 0);
    prof.r.variant = kind <= 1 ? 10 : 11;
    launch_group(grp, kind, (hipStream_t)stream);
    return launch_status();
}

// 1 when dvd_conv_forward would run this request through the kernel that reads a fragment-major weight image (d->wq)
extern "C" int dvd_conv_wants_fragment_major(const dvd_conv_desc* d) {
    ConvK p; ConvPlan pl;
    dvd_conv_desc t = *d;
    if (!t.w) t.w = t.in;                 // (only the geometry matters here)
    t.wq = t.w;                           // "if the image were supplied"
    if (conv_plan(&t, nullptr, p, pl) != DVD_OK) return 0;
    if (dvd_conv_thin_in_ok(d)) return 2;      // 3 (8) input channels -> 64: wants the image of dvd_conv_thin_image in `wq` instead
    if (dvd_conv_thin_out_ok(d)) return 3;     // 64 -> 3 (8) channels: dvd_conv_thin_out_image
    return ((pl.halo || pl.smallf) && d->dtype == DVD_BF16) ? 1 : 0;
}

extern "C" int dvd_pack_conv_weight(int dtype, const float* w, const float* sigma, int Cout, int Cin, int ntaps,
                                    int Cip, int co_off, int co_tot_f, int co_tot_d, void* wf, void* wd,
                                    int kt, int kh, int kw, int ci_off, int ci_tot, void* stream) {
    if (!w || (!wf && !wd) || Cout <= 0 || Cin <= 0 || ntaps != kt * kh * kw) return DVD_E_ARG;
    if (ci_tot <= 0) { ci_off = 0; ci_tot = Cin; }
    if (ci_off < 0 || ci_off + Cin > ci_tot) return DVD_E_ARG;
    if ((Cip & 7) || Cip < Cin || (wd && (co_tot_d & 7))) return DVD_E_SHAPE;
    PackK p{w, sigma, (char*)wf, (char*)wd, Cout, Cin, ntaps, Cip, co_off, co_tot_f, co_tot_d, kt, kh, kw, ci_off, ci_tot};
    const long long n = (long long)Cout * Cip * ntaps;
    if (dtype == DVD_BF16) pack_weight_kernel<bf16_t><<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(p);
    else if (dtype == DVD_F32) pack_weight_kernel<float><<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(p);
    else return DVD_E_ARG;
    return launch_status();
}

extern "C" int dvd_pack_conv_weight_batched(int dtype, const dvd_pack_item* items, int n, void* stream) {
    if (!items || n <= 0) return DVD_E_ARG;
    if (dtype != DVD_BF16 && dtype != DVD_F32) return DVD_E_ARG;
    for (int i = 0; i < n; ++i) {           // validate everything before the first launch
        const dvd_pack_item& t = items[i];
        if (!t.w || (!t.wf && !t.wd) || t.Cout <= 0 || t.Cin <= 0 || t.ntaps != t.kt * t.kh * t.kw) return DVD_E_ARG;
        const int ci_off = t.ci_tot <= 0 ? 0 : t.ci_off, ci_tot = t.ci_tot <= 0 ? t.Cin : t.ci_tot;
        if (ci_off < 0 || ci_off + t.Cin > ci_tot) return DVD_E_ARG;
        if ((t.Cip & 7) || t.Cip < t.Cin || (t.wd && (t.co_tot_d & 7))) return DVD_E_SHAPE;
    }
    for (int i0 = 0; i0 < n; i0 += kPackBatch) {
        PackBatchK b;
        b.n = n - i0 < kPackBatch ? n - i0 : kPackBatch;
        long long blocks = 0;
        for (int j = 0; j < b.n; ++j) {
            const dvd_pack_item& t = items[i0 + j];
            const int ci_off = t.ci_tot <= 0 ? 0 : t.ci_off, ci_tot = t.ci_tot <= 0 ? t.Cin : t.ci_tot;
            b.it[j] = PackK{t.w, t.sigma, (char*)t.wf, (char*)t.wd, t.Cout, t.Cin, t.ntaps, t.Cip, t.co_off, t.co_tot_f, t.co_tot_d,
                            t.kt, t.kh, t.kw, ci_off, ci_tot};
            b.first[j] = (int)blocks;
            blocks += cdiv((long long)t.Cout * t.Cip * t.ntaps, 256);
        }
        if (blocks >= (1ll << 31)) return DVD_E_SHAPE;
        b.first[b.n] = (int)blocks;
        if (dtype == DVD_BF16) pack_weight_batched_kernel<bf16_t><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(b);
        else pack_weight_batched_kernel<float><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(b);
    }
    return launch_status();
}
