// Halo-staged forward / backward-data convolutions whose weight operand is read straight from L2 in fragment-major order
// (bf16): conv_halo_gb_kernel (frames >= 16 pixels wide), conv_halo_gbs_kernel (whole 4 x 4 / 8 x 8 frames per tile), and the
// GROUPED launches of both (several independent convolutions in one grid: the layer wavefront of the ConvGRU stacks, gru.hip).
//
// Same GEMM as conv_halo_kernel (conv_igemm.hip): the input footprint of an output patch sits in LDS for all taps of a 32-channel
// chunk (LDS-DMA, zero fill by out-of-range offsets, double-buffered over chunks).  The weights exist a second time in
// fragment-major order (dvd_conv_fragment_major): record (tap, chunk, 32-column block nb, k-half pair kk) is the 1 KiB a wave needs
// for one MFMA B fragment -- lane l = (kh2 = l >> 5, col = l & 31) owns the 8 channels chunk*32 + (2 kk + kh2)*8 .. +7 of output
// column nb*32 + col at byte l*16 -- so a fragment is ONE coalesced buffer_load_dwordx4 per lane, four per wave and K step,
// prefetched one K step ahead.  Nothing is handed between waves inside a chunk: no per-tap LDS-DMA, wait or barrier; the waves of a
// workgroup meet once per chunk (footprint hand-over).
//
// Round 5: the filter size KS (3 or 5) is a template parameter.  The tap loop of a chunk is `for (row) { unrolled KS taps }`:
// the LDS offset of tap ix is an immediate, the weight record advances by a running scalar add (records are [tap][chunk], so the
// taps of a chunk are a constant stride apart), the footprint of the next chunk goes out in the first step of row 0 only, and the
// per-step tap counters / record multiplications / branchy iterator of the runtime-tap form (about 40 scalar instructions and
// three branches per 16 MFMAs) are gone.  Lane-validity of the footprint rows is folded into the offsets (an invalid row carries
// an offset beyond the descriptor's range) instead of living in lane-mask register pairs across the loop.
#include "conv_common.h"

namespace {

#ifndef DVD_GB_EPI_DEEP
#define DVD_GB_EPI_DEEP 0
#endif

// Footprint offsets: the descriptor of a tile's input starts at the first frame it can touch and is clamped to 1 GiB, a footprint row
// outside the frame (or past the group count) carries kRowOOB, a time step outside the clip adds kStepOOB: every combination lands
// at or beyond 1 GiB (no wrap-around below it), where the hardware returns zeros.  Valid offsets stay below 3 frames (< 1 GiB).
constexpr unsigned kRowOOB = 0x80000000u, kStepOOB = 0x40000000u, kRangeCap = 0x3fffffffu;

template <int TM, int WN, int WMV, bool UP2> struct HaloGbCfg {
    static constexpr int PITCH = HaloGeo<UP2>::PITCH;
    static constexpr int NWAVE = WMV * WN;
    static constexpr int PH = WMV * TM * 2;                             // patch: PH lines x 16 columns
    static constexpr int HG = UP2 ? ((PH / 2 + 3) * PITCH + 15) / 16 : ((PH + 4) * PITCH + 15) / 16;
    static constexpr int HBYTES = HG * 1024;
    static constexpr int EPI = NWAVE * 32 * 64 * 4;
    static constexpr int LDSB = 2 * HBYTES + 1024 > EPI ? 2 * HBYTES + 1024 : EPI;
};

// One K step: 2 * TM units of two MFMAs; the A fragment of unit u + 2 is requested before the MFMAs of unit u, the B fragments of
// the NEXT step (record nrec) right behind the first MFMA pair -- behind the point where the compiler waits for this step's own
// fragments -- and, when `foot` is given, the next chunk's footprint DMAs with them.  (The request is unconditional: behind a branch
// hipcc assumes the smaller outstanding count and drains the prefetch in the middle of a step.)
template <int TM, bool RELU, class LdA, class Next>
__device__ __forceinline__ void gb_step(f32x16 (&acc)[TM][2], const bf16x8 (&b)[2][2], LdA ldA, Next next) {
    constexpr int NU = 2 * TM;
#ifndef DVD_GB_ADEPTH
#define DVD_GB_ADEPTH 2
#endif
    constexpr int AD = DVD_GB_ADEPTH;                 // A fragments requested this many units ahead
    bf16x8 a[NU];
#pragma unroll
    for (int u = 0; u < AD && u < NU; ++u) a[u] = ldA(u);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        if (u + AD < NU) a[u + AD] = ldA(u + AD);
        const int kk = u / TM, tm = u % TM;
        if constexpr (RELU) a[u] = __builtin_bit_cast(bf16x8, relu16_bf16(__builtin_bit_cast(u32x4, a[u])));
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[kk][tn], acc[tm][tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (u == 0) {
#ifndef DVD_EXP_NODMA
            next();
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// waves: WMV (M) x WN (N), wave tile (TM*32 pixels) x 64 columns: 4 x 2 x 2 = the 256 x 128 tile, 2 x 2 x 2 = 128 x 128 (launches with
// few rows), 2 x 1 x 4 = 256 x 64 (thin outputs).  `ltile`: index of the output tile among the convolution's tiles (tickets / slabs
// of the in-launch split-K combine).
template <int TM, int WN, int WMV, bool RELU, bool UP2, int KS>
__device__ __forceinline__ void conv_halo_gb_tile(const ConvK& p, char* const smem, const int mt, const int nt, const int z, const int ltile) {
    using T = bf16_t;
    using G = HaloGeo<UP2>;
    using Cfg = HaloGbCfg<TM, WN, WMV, UP2>;
    constexpr int NWAVE = Cfg::NWAVE;
    constexpr int PITCH = G::PITCH;
    constexpr int BNt = WN * 64;
    constexpr int PH = Cfg::PH, HG = Cfg::HG, HBYTES = Cfg::HBYTES;
    constexpr int NH = (HG + NWAVE - 1) / NWAVE;
    constexpr int ntap2 = KS * KS;
    constexpr int pad = KS >> 1, cpad = (pad + 1) >> 1;
    constexpr int HWa = UP2 ? ((15 + pad) >> 1) + cpad + 1 : 16 + 2 * pad;
    constexpr int HHa = UP2 ? ((PH - 1 + pad) >> 1) + cpad + 1 : PH + 2 * pad;
    char* const hbuf0 = &smem[0];
    char* const dump = &smem[2 * HBYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = nt * BNt;
    const int pw = p.W >> 4, ppf = pw * (p.H / PH);
    const int ft = mt / ppf, pidx = mt - ft * ppf;
    const int y0 = (pidx / pw) * PH, x0 = (pidx % pw) * 16;
    const int nouter = p.kchunks * p.kt;
    const int per = (nouter + p.nsplit - 1) / p.nsplit;
    const int oc_begin = z * per, oc_end = min(nouter, oc_begin + per);
    constexpr unsigned esz = 2;
    const int iy_lo = UP2 ? (y0 >> 1) - cpad : y0 - pad, ix_lo = UP2 ? (x0 >> 1) - cpad : x0 - pad;

    const int wu = __builtin_amdgcn_readfirstlane(wave);
    const int lrow = lane >> 2;
    const unsigned ldb = (unsigned)p.ldi * esz;
    const int tt = p.kt > 1 ? ft % p.T : 0;
    const int base_frame = max(0, ft - (p.kt >> 1));
    const size_t fbytes = (size_t)p.Hin * p.Win * ldb;
    const size_t base_b = (size_t)base_frame * fbytes;
    const size_t left_b = p.in_bytes - base_b;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.in + base_b), 0, left_b > kRangeCap ? kRangeCap : (unsigned)left_b, 0x00020000);
    // fragment-major weights: [tap][chunk][nb32][kk][lane][16 B]; nb32 = 32-column blocks, padded to whole 128-column tiles
    const int nb32 = p.nb32;
    const __amdgpu_buffer_rsrc_t rwq = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.wq, 0, (unsigned)((size_t)p.kt * ntap2 * p.kchunks * nb32 * 2048), 0x00020000);
    // footprint row of DMA group i for this lane: its logical 16-byte slot is XOR-swizzled into the offset (see HaloGeo)
    auto hrow = [&](int i, int& hy, int& hx) __attribute__((always_inline)) {
        const int h = (i * NWAVE + wu) * 16 + lrow;
        hy = h / PITCH; hx = h - hy * PITCH;
    };
    unsigned hoff[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int g = i * NWAVE + wu;
        int hy, hx;
        hrow(i, hy, hx);
        const int yin = iy_lo + hy, xin = ix_lo + hx;
        const int hq = (lane & 3) ^ G::sw(hy, hx);
        const bool ok = g < HG && hy < HHa && hx < HWa && (unsigned)yin < (unsigned)p.Hin && (unsigned)xin < (unsigned)p.Win;
        hoff[i] = ok ? (unsigned)(yin * p.Win + xin) * ldb + hq * 16 : kRowOOB;
    }
    const bool ragged = (p.C & 31) != 0;                                 // last chunk holds fewer than 32 channels (uniform)
    auto dmaH = [&](int hb, int cc_, int it_) __attribute__((always_inline)) {
        const int dt_ = it_ - (p.kt >> 1);
        const bool ok_ = (unsigned)(tt + dt_) < (unsigned)p.T || p.kt == 1;
        const unsigned ud_ = ok_ ? (unsigned)(ft + dt_ - base_frame) * (unsigned)fbytes + cc_ * 64 : kStepOOB;
        const bool part_ = ragged && cc_ == p.kchunks - 1;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int g = i * NWAVE + wu;
            char* dst_ = g < HG ? hbuf0 + hb * HBYTES + g * 1024 : dump;
            unsigned off_ = hoff[i] + ud_;
            if (part_) {                                                 // channels past C: zeros (rare: thin / odd-width inputs)
                int hy, hx;
                hrow(i, hy, hx);
                const int hq = (lane & 3) ^ G::sw(hy, hx);
                off_ = cc_ * 32 + hq * 8 < p.C ? off_ : kRowOOB;
            }
            dma16(rin, dst_, off_);
        }
    };
    // B fragments: voffset = this wave's column half + lane slot (per lane), soffset = record of (tap, chunk, N tile) (uniform)
    const unsigned bvoff = (unsigned)(wn * 2 * 2048 + lane * 16);
    const unsigned bnt = (unsigned)(nt * (BNt / 32)) * 2048u;
    const unsigned brec = (unsigned)nb32 * 2048u;                       // bytes per (tap, chunk)
    const unsigned tapstep = (unsigned)p.kchunks * brec;                // record stride between two taps of one chunk
    auto ldBq = [&](bf16x8 (&b)[2][2], unsigned rec) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
                b[kk][tn] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rwq, bvoff, rec + (unsigned)(tn * 2048 + kk * 1024), 0));
    };
    // record of tap 0 of outer index (chunk cc, time tap it)
    auto rec0 = [&](int cc_, int it_) __attribute__((always_inline)) -> unsigned {
        return ((unsigned)(it_ * ntap2) * p.kchunks + cc_) * brec + bnt;
    };

    f32x16 acc[TM][2];
    {
        const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = zacc;
    }
    const int l31 = lane & 31, kh2 = lane >> 5;
    const int px = l31 & 15, py0 = wm * (TM * 2) + (l31 >> 4);
    constexpr int TMSTRIDE = (UP2 ? 1 : 2) * PITCH * 64;
#ifdef DVD_EXP_NOMAIN
    const int nouter_z = 0;
#else
    const int nouter_z = oc_end - oc_begin;
#endif
    if (nouter_z > 0) {
        int m_cc = oc_begin / p.kt, m_it = oc_begin - m_cc * p.kt;      // (chunk, dt) being multiplied
        int h_cc = m_cc, h_it = m_it;                                   // ... of the next footprint to fetch
        bf16x8 bq0[2][2], bq1[2][2];
        unsigned rec = rec0(m_cc, m_it);
        dmaH(0, h_cc, h_it);
        if (++h_it == p.kt) { h_it = 0; ++h_cc; }
        ldBq(bq0, rec);
        __builtin_amdgcn_s_waitcnt(0x0070 | (0xf << 8));                // vmcnt(0): footprint 0 (this wave's part) landed
        __builtin_amdgcn_s_barrier();
        // The K loop runs over filter ROWS (KS unrolled taps each), two rows per iteration: KS is odd, so a row ends on the other
        // B register set than it started on -- alternating the sets by a branch per row instead makes hipcc keep a second copy of
        // the accumulators at the join (408 spilled registers in the 256 x 128 tile).  Chunk boundaries (footprint hand-over, next
        // chunk's record) are uniform side blocks of the row that ends / starts a chunk.
        int hb = 0, iy = 0, left = nouter_z;                            // footprint buffer, filter row, chunks left incl. the current
        auto row = [&](bf16x8 (&b0)[2][2], bf16x8 (&b1)[2][2]) __attribute__((always_inline)) {
            const bool first = iy == 0, last = iy == KS - 1, more = left > 1;
            const int hy = UP2 ? ((py0 + iy - pad) >> 1) + cpad : py0 + iy;
            const char* const Arow = hbuf0 + hb * HBYTES + hy * (PITCH * 64);
#pragma unroll
            for (int ix = 0; ix < KS; ++ix) {
                const int hx = UP2 ? ((px + ix - pad) >> 1) + cpad : px + ix;
                const int swz = G::sw(hy, hx);
                const char* const Ah = Arow + hx * 64;
                auto ldA = [&](int u) __attribute__((always_inline)) -> bf16x8 {
                    const int kk = u / TM, tm = u % TM, slot = kk * 2 + kh2;
                    const int sl = (UP2 && (tm & 1)) ? (slot ^ 2) : slot;
                    return *reinterpret_cast<const bf16x8*>(Ah + tm * TMSTRIDE + ((sl ^ swz) << 4));
                };
                unsigned nrec = rec + tapstep;
                if (ix == KS - 1 && last) {                             // next chunk's first record; none left: re-read this one
                    int n_cc = m_cc, n_it = m_it + 1;
                    if (n_it == p.kt) { n_it = 0; ++n_cc; }
                    nrec = more ? rec0(n_cc, n_it) : rec;
                    m_cc = n_cc; m_it = n_it;
                }
                auto next = [&]() __attribute__((always_inline)) {
                    ldBq((ix & 1) ? b0 : b1, nrec);
                    if (ix == 0 && first && more) { dmaH(hb ^ 1, h_cc, h_it); if (++h_it == p.kt) { h_it = 0; ++h_cc; } }
                };
                gb_step<TM, RELU>(acc, (ix & 1) ? b1 : b0, ldA, next);
                rec = nrec;
            }
            ++iy;
            if (last) {
                // chunk boundary: every wave has finished reading footprint hb (it is overwritten one chunk from now) and has seen
                // its own part of footprint hb ^ 1 land (the in-order drain in front of the steps since): one barrier per chunk
                iy = 0; hb ^= 1; --left;
                if (more) {
#ifndef DVD_EXP_NOWAIT
                    __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): this wave's fragment reads are done
#endif
#ifndef DVD_EXP_NOBAR
                    __builtin_amdgcn_s_barrier();
#endif
                }
            }
        };
        const int nrows = nouter_z * KS;
        int r = 0;
#pragma unroll 1
        for (; r + 1 < nrows; r += 2) { row(bq0, bq1); row(bq1, bq0); }
        if (r < nrows) row(bq0, bq1);
    }
    __syncthreads();

    float* ep = reinterpret_cast<float*>(&smem[0]) + wave * (32 * 64);
    const int ecol = (lane & 7) * 8, erow = lane >> 3;
    const long long frame_row0 = (long long)ft * (p.H * p.W);
    conv_epilogue<T, TM, DVD_GB_EPI_DEEP>(p, acc, ep, lane, n0 + wn * 64 + ecol, z, ltile, frame_row0, [&](int tm, int j) __attribute__((always_inline)) {
        const int pi = wm * (TM * 32) + tm * 32 + j * 8 + erow;
        return (y0 + (pi >> 4)) * p.W + x0 + (pi & 15);
    });
}

// XCD-aware order: workgroup b runs on XCD b % 8 (observed dispatch order, used for speed only); each XCD gets a contiguous run
__device__ __forceinline__ int xcd_order(int bid, int nwg) {
    const int xcd = bid & 7, qd = nwg >> 3, rr = nwg & 7;
    return (xcd < rr ? xcd * (qd + 1) : rr * (qd + 1) + (xcd - rr) * qd) + (bid >> 3);
}
// tile index -> (M tile, N tile): M-major, or N-major (a run of workgroups shares the weight tile) for weight-heavy launches
struct TileMN { int mt, nt; };
__device__ __forceinline__ TileMN tile_of(int nmajor, int tilesN, int t, int ntiles) {
    const int d = nmajor ? ntiles / tilesN : tilesN;        // divisor: tiles per N tile / per M tile
    const int q = t / d, r = t - q * d;
    TileMN o;
    o.mt = nmajor ? r : q; o.nt = nmajor ? q : r;
    return o;
}

template <int TM, int WN, int WMV, bool RELU, bool UP2, int KS>
__global__ __launch_bounds__(256, 2) void conv_halo_gb_kernel(ConvK p) {
    __shared__ __attribute__((aligned(16))) char smem[HaloGbCfg<TM, WN, WMV, UP2>::LDSB];
    const int bid = xcd_order(blockIdx.x, gridDim.x);
    const TileMN t = tile_of(p.nmajor, p.tilesN, bid, gridDim.x);
    conv_halo_gb_tile<TM, WN, WMV, RELU, UP2, KS>(p, smem, t.mt, t.nt, blockIdx.z, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------- the same for frames of 8 x 8 and 4 x 4 pixels
// The recurrent convolutions of the first two generator stages (Generator.py:39,43: ConvGRUs on 4 x 4 / 8 x 8 latents): the M tile is
// G WHOLE frames (TM*64 / S^2 of them), whose zero-padded footprints ((S+4) x (S+4) rows of 64 bytes per frame, PITCH = S + 4
// whatever the filter size) sit in LDS for all 9 / 25 taps of a channel chunk; weights from L2 in fragment-major order as above.
// 16-byte slot swizzle: (line of the footprint) & 3 -- brute-forced conflict-free for both ds_read_b128 lane groups, every tap,
// S = 8 (a 32-row sub-tile = 4 lines of one frame) and S = 4 (= 2 frames).
template <int TM, int S> struct HaloGbsCfg {
    static constexpr int PITCH = S + 4, FR = PITCH * PITCH;             // rows per frame footprint
    static constexpr int G = TM * 64 / (S * S);                         // frames per tile
    static constexpr int HG = (G * FR + 15) / 16;
    static constexpr int HBYTES = HG * 1024;
    static constexpr int EPI = 4 * 32 * 64 * 4;
    static constexpr int LDSB = 2 * HBYTES + 1024 > EPI ? 2 * HBYTES + 1024 : EPI;
};

template <int TM, int S, bool RELU, int KS>
__device__ __forceinline__ void conv_halo_gbs_tile(const ConvK& p, char* const smem, const int mt, const int nt, const int z, const int ltile) {
    using T = bf16_t;
    using Cfg = HaloGbsCfg<TM, S>;
    constexpr int WN = 2, NWAVE = 4, BNt = 128;
    constexpr int PITCH = Cfg::PITCH, FR = Cfg::FR, G = Cfg::G, HG = Cfg::HG, HBYTES = Cfg::HBYTES;
    constexpr int NH = (HG + NWAVE - 1) / NWAVE;
    constexpr int SS = S * S;
    constexpr int ntap2 = KS * KS, pad = KS >> 1, ext = S + 2 * pad;   // footprint extent actually used by this filter
    char* const hbuf0 = &smem[0];
    char* const dump = &smem[2 * HBYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = nt * BNt;
    const int ft0 = mt * G, nframes = p.M / SS;
    const int per = (p.kchunks + p.nsplit - 1) / p.nsplit;
    const int cc_begin = z * per, cc_end = min(p.kchunks, cc_begin + per);
    constexpr unsigned esz = 2;

    const int wu = __builtin_amdgcn_readfirstlane(wave);
    const int lrow = lane >> 2;
    const unsigned ldb = (unsigned)p.ldi * esz;
    const size_t base_b = (size_t)ft0 * SS * ldb;
    const size_t left_b = p.in_bytes - base_b;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.in + base_b), 0, left_b > kRangeCap ? kRangeCap : (unsigned)left_b, 0x00020000);
    const int nb32 = p.nb32;
    const __amdgpu_buffer_rsrc_t rwq = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.wq, 0, (unsigned)((size_t)ntap2 * p.kchunks * nb32 * 2048), 0x00020000);
    auto hrow = [&](int i, int& fl, int& hy, int& hx) __attribute__((always_inline)) {
        const int h = (i * NWAVE + wu) * 16 + lrow;
        fl = h / FR;
        const int r = h - fl * FR;
        hy = r / PITCH; hx = r - hy * PITCH;
    };
    unsigned hoff[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int g = i * NWAVE + wu;
        int fl, hy, hx;
        hrow(i, fl, hy, hx);
        const int yin = hy - pad, xin = hx - pad;
        const int hq = (lane & 3) ^ (hy & 3);
        const bool ok = g < HG && fl < G && ft0 + fl < nframes && hy < ext && hx < ext && (unsigned)yin < (unsigned)S && (unsigned)xin < (unsigned)S;
        hoff[i] = ok ? (unsigned)((fl * S + yin) * S + xin) * ldb + hq * 16 : kRowOOB;
    }
    const bool ragged = (p.C & 31) != 0;
    auto dmaH = [&](int hb, int cc_) __attribute__((always_inline)) {
        const bool part_ = ragged && cc_ == p.kchunks - 1;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int g = i * NWAVE + wu;
            char* dst_ = g < HG ? hbuf0 + hb * HBYTES + g * 1024 : dump;
            unsigned off_ = hoff[i] + cc_ * 64;
            if (part_) {
                int fl, hy, hx;
                hrow(i, fl, hy, hx);
                const int hq = (lane & 3) ^ (hy & 3);
                off_ = cc_ * 32 + hq * 8 < p.C ? off_ : kRowOOB;
            }
            dma16(rin, dst_, off_);
        }
    };
    const unsigned bvoff = (unsigned)(wn * 2 * 2048 + lane * 16);
    const unsigned bnt = (unsigned)(nt * (BNt / 32)) * 2048u;
    const unsigned brec = (unsigned)nb32 * 2048u;
    const unsigned tapstep = (unsigned)p.kchunks * brec;
    auto ldBq = [&](bf16x8 (&b)[2][2], unsigned rec) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
                b[kk][tn] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rwq, bvoff, rec + (unsigned)(tn * 2048 + kk * 1024), 0));
    };

    f32x16 acc[TM][2];
    {
        const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = zacc;
    }
    // this lane's A rows: pixel pi = (wm*TM + tm)*32 + l31 of the tile -> (frame, y, x); sub-tile tm sits a compile-time number of
    // footprint rows further (S = 8: 4 lines or a whole frame, S = 4: two frames); y & 3 is the same for every tm
    const int l31 = lane & 31, kh2 = lane >> 5;
    const int pi0 = wm * (TM * 32) + l31;
    const int f0 = pi0 / SS, y0 = (pi0 % SS) / S, x0 = pi0 % S;
    const int arow0 = f0 * FR + y0 * PITCH + x0;
    auto tmoff = [](int tm) constexpr -> int { return S == 8 ? ((tm >> 1) * FR + (tm & 1) * 4 * PITCH) * 64 : tm * 2 * FR * 64; };
#ifdef DVD_EXP_NOMAIN
    const int nch = 0;
#else
    const int nch = cc_end - cc_begin;
#endif
    if (nch > 0) {
        unsigned rec = (unsigned)cc_begin * brec + bnt;                 // tap 0 of the first chunk; records are [tap][chunk]
        bf16x8 bq0[2][2], bq1[2][2];
        dmaH(0, cc_begin);
        ldBq(bq0, rec);
        __builtin_amdgcn_s_waitcnt(0x0070 | (0xf << 8));                // vmcnt(0)
        __builtin_amdgcn_s_barrier();
        int hb = 0, iy = 0, cc = cc_begin;                              // (row loop: see conv_halo_gb_tile)
        auto row = [&](bf16x8 (&b0)[2][2], bf16x8 (&b1)[2][2]) __attribute__((always_inline)) {
            const bool first = iy == 0, last = iy == KS - 1, more = cc + 1 < cc_end;
            const int swz = (y0 + iy) & 3;
            const char* const Arow = hbuf0 + hb * HBYTES + (arow0 + iy * PITCH) * 64;
#pragma unroll
            for (int ix = 0; ix < KS; ++ix) {
                const char* const Ah = Arow + ix * 64;
                auto ldA = [&](int u) __attribute__((always_inline)) -> bf16x8 {
                    const int kk = u / TM, tm = u % TM, slot = kk * 2 + kh2;
                    return *reinterpret_cast<const bf16x8*>(Ah + tmoff(tm) + ((slot ^ swz) << 4));
                };
                unsigned nrec = rec + tapstep;
                if (ix == KS - 1 && last) nrec = more ? (unsigned)(cc + 1) * brec + bnt : rec;
                auto next = [&]() __attribute__((always_inline)) {
                    ldBq((ix & 1) ? b0 : b1, nrec);
                    if (ix == 0 && first && more) dmaH(hb ^ 1, cc + 1);
                };
                gb_step<TM, RELU>(acc, (ix & 1) ? b1 : b0, ldA, next);
                rec = nrec;
            }
            ++iy;
            if (last) {
                iy = 0; hb ^= 1; ++cc;
                if (more) {
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_s_barrier();
                }
            }
        };
        const int nrows = nch * KS;
        int r = 0;
#pragma unroll 1
        for (; r + 1 < nrows; r += 2) { row(bq0, bq1); row(bq1, bq0); }
        if (r < nrows) row(bq0, bq1);
    }
    __syncthreads();

    float* ep = reinterpret_cast<float*>(&smem[0]) + wave * (32 * 64);
    const int ecol = (lane & 7) * 8, erow = lane >> 3;
    const long long row0 = (long long)ft0 * SS;
    conv_epilogue<T, TM, DVD_GB_EPI_DEEP>(p, acc, ep, lane, n0 + wn * 64 + ecol, z, ltile, row0, [&](int tm, int j) __attribute__((always_inline)) {
        const int pi = wm * (TM * 32) + tm * 32 + j * 8 + erow;          // the tile's rows are G consecutive whole frames
        return row0 + pi < p.M ? pi : -1;
    });
}

template <int TM, int S, bool RELU, int KS>
__global__ __launch_bounds__(256, 2) void conv_halo_gbs_kernel(ConvK p) {
    __shared__ __attribute__((aligned(16))) char smem[HaloGbsCfg<TM, S>::LDSB];
    const int bid = xcd_order(blockIdx.x, gridDim.x);
    const TileMN t = tile_of(p.nmajor, p.tilesN, bid, gridDim.x);
    conv_halo_gbs_tile<TM, S, RELU, KS>(p, smem, t.mt, t.nt, blockIdx.z, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------- grouped launches
// Several INDEPENDENT convolutions in one grid (gru.hip: the recurrent / x-part convolutions of the layers of a ConvGRU stack that
// a layer wavefront makes independent).  A one-round launch runs its workgroups in lock step: prologue, main loop and the HBM-bound
// gate epilogue of all of them coincide and overlap with nothing.  In a grouped launch the workgroup slots of a CU hold tiles of
// different convolutions with different K lengths and epilogues, so one tile's epilogue runs under another's main loop, and the
// launch has several rounds.  Slot order: see ConvGroup (conv_common.h).
template <class TileFn>
__device__ __forceinline__ void group_dispatch(const ConvGroup& grp, TileFn fn) {
    const int x = blockIdx.x & 7;
    int rem = blockIdx.x >> 3;                                           // this XCD's rem-th workgroup
    auto share = [&](int m) __attribute__((always_inline)) -> int { const int w = grp.wgs[m]; return (w >> 3) + (x < (w & 7) ? 1 : 0); };
    const int m0 = grp.order[0], m1 = grp.n > 1 ? grp.order[1] : m0;
    const int s0 = share(m0), s1 = grp.n > 1 ? share(m1) : 0;
    const int h0 = min(grp.head, s0), h1 = min(grp.head, s1);
    int g = -1, off = 0;
    if (rem < h0) { g = m0; off = rem; }
    else if ((rem -= h0) < h1) { g = m1; off = rem; }
    else if ((rem -= h1) < s0 - h0) { g = m0; off = h0 + rem; }
    else if ((rem -= s0 - h0) < s1 - h1) { g = m1; off = h1 + rem; }
    else {
        rem -= s1 - h1;
        for (int k = 2; k < grp.n; ++k) {
            const int m = grp.order[k], s = share(m);
            if (rem < s) { g = m; off = rem; break; }
            rem -= s;
        }
    }
    if (g < 0) return;
    const int w = grp.wgs[g];
    const int local = x * (w >> 3) + min(x, w & 7) + off;               // index among member g's workgroups (tiles x slices)
    const ConvK& p = grp.c[g];
    const int t = local / p.nsplit, z = local - t * p.nsplit;
    const TileMN mn = tile_of(p.nmajor, p.tilesN, t, w / p.nsplit);
    fn(p, mn.mt, mn.nt, z, t);
}

template <int TM, int WN, int WMV>
__global__ __launch_bounds__(256, 2) void conv_group_gb_kernel(ConvGroup grp) {
    __shared__ __attribute__((aligned(16))) char smem[HaloGbCfg<TM, WN, WMV, false>::LDSB];
    group_dispatch(grp, [&](const ConvK& p, int mt, int nt, int z, int t) __attribute__((always_inline)) {
        if (p.kh == 5) conv_halo_gb_tile<TM, WN, WMV, false, false, 5>(p, smem, mt, nt, z, t);
        else conv_halo_gb_tile<TM, WN, WMV, false, false, 3>(p, smem, mt, nt, z, t);
    });
}

template <int TM, int S>
__global__ __launch_bounds__(256, 2) void conv_group_gbs_kernel(ConvGroup grp) {
    __shared__ __attribute__((aligned(16))) char smem[HaloGbsCfg<TM, S>::LDSB];
    group_dispatch(grp, [&](const ConvK& p, int mt, int nt, int z, int t) __attribute__((always_inline)) {
        if (p.kh == 5) conv_halo_gbs_tile<TM, S, false, 5>(p, smem, mt, nt, z, t);
        else conv_halo_gbs_tile<TM, S, false, 3>(p, smem, mt, nt, z, t);
    });
}

// standard forward pack [tap][Cout][C] (bf16) -> fragment-major [tap][chunk][nb32][kk][lane][8] (zeros in every padded position)
struct FragK { const bf16_t* w; bf16_t* wq; int ntaps, Cout, C, kchunks, nb32; };
__global__ void fragment_major_kernel(FragK p) {
    const long long n = (long long)p.ntaps * p.kchunks * p.nb32 * 2 * 64;             // 16-byte units
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int l = (int)(i & 63);
    long long r = i >> 6;
    const int kk = (int)(r & 1); r >>= 1;
    const int nb = (int)(r % p.nb32); r /= p.nb32;
    const int cc = (int)(r % p.kchunks);
    const int tap = (int)(r / p.kchunks);
    const int co = nb * 32 + (l & 31), ci = cc * 32 + (kk * 2 + (l >> 5)) * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (co < p.Cout && ci < p.C) v = *reinterpret_cast<const u32x4*>(p.w + ((size_t)tap * p.Cout + co) * p.C + ci);   // C % 8 == 0
    *reinterpret_cast<u32x4*>(p.wq + i * 8) = v;
}
// n images in one launch (whole blocks per item, as pack_weight_batched_kernel)
constexpr int kFragBatch = 32;
struct FragBatchK { FragK it[kFragBatch]; int first[kFragBatch + 1]; int n; };
__global__ void fragment_major_batched_kernel(FragBatchK b) {
    int j = 0;
    while (j + 1 < b.n && (int)blockIdx.x >= b.first[j + 1]) ++j;
    const FragK& p = b.it[j];
    const long long n = (long long)p.ntaps * p.kchunks * p.nb32 * 2 * 64;
    const long long i = (long long)((int)blockIdx.x - b.first[j]) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int l = (int)(i & 63);
    long long r = i >> 6;
    const int kk = (int)(r & 1); r >>= 1;
    const int nb = (int)(r % p.nb32); r /= p.nb32;
    const int cc = (int)(r % p.kchunks);
    const int tap = (int)(r / p.kchunks);
    const int co = nb * 32 + (l & 31), ci = cc * 32 + (kk * 2 + (l >> 5)) * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (co < p.Cout && ci < p.C) v = *reinterpret_cast<const u32x4*>(p.w + ((size_t)tap * p.Cout + co) * p.C + ci);
    *reinterpret_cast<u32x4*>(p.wq + i * 8) = v;
}

}  // namespace

// ---------------------------------------------------------------------------- launchers (called by conv_igemm.hip's dispatch / gru.hip)
#ifdef DVD_GB_PROBE      // ISA inspection build (tools/isa_probe.sh): one instantiation of each kernel family
namespace dvdk {
void launch_gb(const ConvK& p, int, bool, bool, dim3 grid, hipStream_t st) { conv_halo_gb_kernel<4, 2, 2, false, false, 5><<<grid, 256, 0, st>>>(p); }
void launch_gbs(const ConvK& p, int, bool, bool, dim3 grid, hipStream_t st) { conv_halo_gbs_kernel<2, 8, false, 3><<<grid, 256, 0, st>>>(p); }
void launch_group(const ConvGroup& grp, int, hipStream_t st) { conv_group_gb_kernel<4, 2, 2><<<dim3(grp.nslots), 256, 0, st>>>(grp); }
}
#else
namespace dvdk {
// variant: 0 = 256 x 128 tile, 1 = 128 x 128 (launches with few rows), 2 = 256 x 64 (thin outputs); p.kh == p.kw in {3, 5}
void launch_gb(const ConvK& p, int variant, bool relu_in, bool up2, dim3 grid, hipStream_t st) {
#define LAUNCH_GB4(TM_, WN_, WMV_, KS_)                                                                       \
    do { if (relu_in) { if (up2) conv_halo_gb_kernel<TM_, WN_, WMV_, true, true, KS_><<<grid, 256, 0, st>>>(p);       \
                        else conv_halo_gb_kernel<TM_, WN_, WMV_, true, false, KS_><<<grid, 256, 0, st>>>(p); }           \
         else         { if (up2) conv_halo_gb_kernel<TM_, WN_, WMV_, false, true, KS_><<<grid, 256, 0, st>>>(p);      \
                        else conv_halo_gb_kernel<TM_, WN_, WMV_, false, false, KS_><<<grid, 256, 0, st>>>(p); } } while (0)
#define LAUNCH_GB(TM_, WN_, WMV_) do { if (p.kh == 5) LAUNCH_GB4(TM_, WN_, WMV_, 5); else LAUNCH_GB4(TM_, WN_, WMV_, 3); } while (0)
    if (variant == 2) LAUNCH_GB(2, 1, 4); else if (variant == 0) LAUNCH_GB(4, 2, 2); else LAUNCH_GB(2, 2, 2);
#undef LAUNCH_GB
#undef LAUNCH_GB4
}
// whole-frame footprints: S = 8 (256- or 128-row tiles) or S = 4 (128-row tiles)
void launch_gbs(const ConvK& p, int S, bool big, bool relu_in, dim3 grid, hipStream_t st) {
#define LAUNCH_GBS2(TM_, SZ_, KS_) do { if (relu_in) conv_halo_gbs_kernel<TM_, SZ_, true, KS_><<<grid, 256, 0, st>>>(p);   \
                                        else conv_halo_gbs_kernel<TM_, SZ_, false, KS_><<<grid, 256, 0, st>>>(p); } while (0)
#define LAUNCH_GBS(TM_, SZ_) do { if (p.kh == 5) LAUNCH_GBS2(TM_, SZ_, 5); else LAUNCH_GBS2(TM_, SZ_, 3); } while (0)
    if (S == 8) { if (big) LAUNCH_GBS(4, 8); else LAUNCH_GBS(2, 8); }
    else LAUNCH_GBS(2, 4);
#undef LAUNCH_GBS
#undef LAUNCH_GBS2
}
// grouped launch: kind 0 = conv_halo_gb 256 x 128 tiles, 1 = conv_halo_gb 128 x 128, 2 / 3 = whole 8 x 8 frames 256- / 128-row tiles,
// 4 = whole 4 x 4 frames (128-row tiles); every member: bf16, fragment-major weights, no input ReLU, no upsample
void launch_group(const ConvGroup& grp, int kind, hipStream_t st) {
    const dim3 grid(grp.nslots);
    switch (kind) {
        case 0: conv_group_gb_kernel<4, 2, 2><<<grid, 256, 0, st>>>(grp); break;
        case 1: conv_group_gb_kernel<2, 2, 2><<<grid, 256, 0, st>>>(grp); break;
        case 2: conv_group_gbs_kernel<4, 8><<<grid, 256, 0, st>>>(grp); break;
        case 3: conv_group_gbs_kernel<2, 8><<<grid, 256, 0, st>>>(grp); break;
        default: conv_group_gbs_kernel<2, 4><<<grid, 256, 0, st>>>(grp); break;
    }
}
}  // namespace dvdk
#endif

// Fragment-major image of a forward (or backward-data) pack for conv_halo_gb_kernel; see the comment at the top.
extern "C" long long dvd_conv_fragment_major_bytes(int ntaps, int Cout, int C) {
    if (ntaps <= 0 || Cout <= 0 || C <= 0) return 0;
    const long long kchunks = (C + 31) / 32, nb32 = (Cout + 127) / 128 * 4;
    return (long long)ntaps * kchunks * nb32 * 2048;
}
extern "C" int dvd_conv_fragment_major(int dtype, const void* w, void* wq, int ntaps, int Cout, int C, void* stream) {
    if (!w || !wq || ntaps <= 0 || Cout <= 0 || C <= 0) return DVD_E_ARG;
    if (dtype != DVD_BF16 || (C & 7)) return DVD_E_SHAPE;
    FragK p{(const bf16_t*)w, (bf16_t*)wq, ntaps, Cout, C, (C + 31) / 32, (Cout + 127) / 128 * 4};
    const long long n = (long long)ntaps * p.kchunks * p.nb32 * 128;
    fragment_major_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(p);
    return launch_status();
}

extern "C" int dvd_conv_fragment_major_batched(int dtype, const dvd_frag_item* items, int n, void* stream) {
    if (!items || n <= 0) return DVD_E_ARG;
    if (dtype != DVD_BF16) return DVD_E_SHAPE;
    for (int i = 0; i < n; ++i) {
        if (!items[i].w || !items[i].wq || items[i].ntaps <= 0 || items[i].Cout <= 0 || items[i].C <= 0) return DVD_E_ARG;
        if (items[i].C & 7) return DVD_E_SHAPE;
    }
    for (int i0 = 0; i0 < n; i0 += kFragBatch) {
        FragBatchK b;
        b.n = n - i0 < kFragBatch ? n - i0 : kFragBatch;
        long long blocks = 0;
        for (int j = 0; j < b.n; ++j) {
            const dvd_frag_item& t = items[i0 + j];
            b.it[j] = FragK{(const bf16_t*)t.w, (bf16_t*)t.wq, t.ntaps, t.Cout, t.C, (t.C + 31) / 32, (t.Cout + 127) / 128 * 4};
            b.first[j] = (int)blocks;
            blocks += cdiv((long long)t.ntaps * b.it[j].kchunks * b.it[j].nb32 * 128, 256);
        }
        if (blocks >= (1ll << 31)) return DVD_E_SHAPE;
        b.first[b.n] = (int)blocks;
        fragment_major_batched_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(b);
    }
    return launch_status();
}
