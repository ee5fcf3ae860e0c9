// Halo-staged forward / backward-data convolutions whose weight operand is read straight from L2 in fragment-major order
// (bf16): conv_halo_gb_kernel (frames >= 16 pixels wide) and conv_halo_gbs_kernel (whole 4 x 4 / 8 x 8 frames per tile).
#include "conv_common.h"

namespace {

// ============================================================================ forward, halo-staged, weights from L2
// conv_halo_tile with the weight operand read STRAIGHT INTO REGISTERS instead of through LDS.  The weights are kept a second
// time in fragment-major order (dvd_conv_fragment_major): record (tap, chunk, 32-column block nb, k-half pair kk) is the 1 KiB
// a wave needs for one B fragment -- lane l = (kh2 = l >> 5, col = l & 31) owns the 8 channels chunk*32 + (2 kk + kh2)*8 .. +7 of
// output column nb*32 + col at byte l*16 -- so a fragment is ONE fully coalesced buffer_load_dwordx4 per lane (8 whole 128-byte
// lines per wave instruction), four per wave and K step.  What that buys: the activation footprint of a channel chunk stays in
// LDS for all 9 / 25 taps, so with the weight tile gone from LDS nothing is handed between waves inside a chunk -- the per-tap
// LDS-DMA of the weight tile (the slowest instruction of the old loop to issue), its counted wait and the per-tap s_barrier all
// disappear; the waves of a workgroup meet once per chunk (footprint hand-over) instead of once per tap.  Price: the two waves
// that share an N half both fetch its fragments (16 KB per workgroup and K step from L1 / L2 instead of 8 KB by DMA).
// Prefetch distance one K step: the fragments of step s+1 are requested right after the first MFMA pair of step s, i.e. behind the
// point where the compiler waits for step s's own fragments (hipcc drains vmcnt(0) there while an LDS-DMA may be pending).
// waves: WMV (M) x WN (N), wave tile (TM*32 pixels) x 64 columns: 4 x 2 x 2 = the 256 x 128 tile, 2 x 2 x 2 = 128 x 128 (launches with
// few rows), 2 x 1 x 4 = 256 x 64 (thin outputs)
#ifndef DVD_GB_EPI_DEEP
#define DVD_GB_EPI_DEEP 0
#endif
template <int TM, int WN, int WMV, bool UP2> struct HaloGbCfg {
    static constexpr int PITCH = HaloGeo<UP2>::PITCH;
    static constexpr int NWAVE = WMV * WN;
    static constexpr int PH = WMV * TM * 2;                             // patch: PH lines x 16 columns
    static constexpr int HG = UP2 ? ((PH / 2 + 3) * PITCH + 15) / 16 : ((PH + 4) * PITCH + 15) / 16;
    static constexpr int HBYTES = HG * 1024;
    static constexpr int EPI = NWAVE * 32 * 64 * 4;
    static constexpr int LDSB = 2 * HBYTES + 1024 > EPI ? 2 * HBYTES + 1024 : EPI;
};

template <int TM, int WN, int WMV, bool RELU, bool UP2>
__device__ __forceinline__ void conv_halo_gb_tile(const ConvK& p, char* const smem, const int mt, const int nt, const int z) {
    using T = bf16_t;
    using G = HaloGeo<UP2>;
    using Cfg = HaloGbCfg<TM, WN, WMV, UP2>;
    constexpr int NWAVE = Cfg::NWAVE;
    constexpr int PITCH = G::PITCH;
    constexpr int BNt = WN * 64;
    constexpr int PH = Cfg::PH, HG = Cfg::HG, HBYTES = Cfg::HBYTES;
    constexpr int NH = (HG + NWAVE - 1) / NWAVE;
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
    const int ntap2 = p.kh * p.kw;
    constexpr unsigned esz = 2;
    const int pad = p.kh >> 1, cpad = (pad + 1) >> 1;
    const int HWa = UP2 ? ((15 + pad) >> 1) + cpad + 1 : 16 + 2 * pad;
    const int HHa = UP2 ? ((PH - 1 + pad) >> 1) + cpad + 1 : PH + 2 * pad;
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
        (void*)(p.in + base_b), 0, left_b > 0xfffffffeull ? 0xfffffffeu : (unsigned)left_b, 0x00020000);
    // fragment-major weights: [tap][chunk][nb32][kk][lane][16 B]; nb32 = 32-column blocks, padded to whole 128-column tiles
    const int nb32 = p.nb32;
    const __amdgpu_buffer_rsrc_t rwq = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.wq, 0, (unsigned)((size_t)p.kt * ntap2 * p.kchunks * nb32 * 2048), 0x00020000);
    unsigned hoff[NH];
    int hq[NH];
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
    auto dmaH = [&](int hb, int cc_, int it_) __attribute__((always_inline)) {
        const int dt_ = it_ - (p.kt >> 1);
        const bool ok_ = (unsigned)(tt + dt_) < (unsigned)p.T || p.kt == 1;
        const unsigned ud_ = (unsigned)(ft + dt_ - base_frame) * (unsigned)fbytes + cc_ * 64;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int g = i * NWAVE + wu;
            char* dst_ = g < HG ? hbuf0 + hb * HBYTES + g * 1024 : dump;
            const bool cv_ = cc_ * 32 + hq[i] * 8 < p.C;
            dma16(rin, dst_, (hval[i] && ok_ && cv_) ? hoff[i] + ud_ : 0xffffffffu);
        }
    };
    // B fragments: voffset = this wave's column half + lane slot (per lane), soffset = record of (tap, chunk, N tile) (uniform)
    const unsigned bvoff = (unsigned)(wn * 2 * 2048 + lane * 16);
    const unsigned bnt = (unsigned)(nt * (BNt / 32)) * 2048u;
    const unsigned brec = (unsigned)nb32 * 2048u;                       // bytes per (tap, chunk)
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
    const int l31 = lane & 31, kh2 = lane >> 5;
    const int px = l31 & 15, py0 = wm * (TM * 2) + (l31 >> 4);
    constexpr int TMSTRIDE = (UP2 ? 1 : 2) * PITCH * 64;
#ifdef DVD_EXP_NOMAIN
    const int nsteps = 0;
#else
    const int nsteps = (oc_end - oc_begin) * ntap2;
#endif
    if (nsteps > 0) {
        // running (chunk, dt) of the step being multiplied / of the next footprint, and the FETCH iterator of the weight records
        // (memory order [tap][chunk]), which runs GB_DIST K steps ahead and stops at the last record (the trailing requests re-read it:
        // they are UNCONDITIONAL so that the compiler's counted waits know them outstanding -- behind a branch hipcc assumes the
        // smaller count and drains the prefetch in the middle of a step)
#ifdef DVD_GB_DIST2
        constexpr int GB_DIST = 2;
#else
        constexpr int GB_DIST = 1;
#endif
        int m_cc = oc_begin / p.kt, m_it = oc_begin - m_cc * p.kt;
        int h_cc = m_cc, h_it = m_it;
        int f_cc = m_cc, f_it = m_it, f_tap = 0, f_left = nsteps;
        auto frec = [&]() __attribute__((always_inline)) -> unsigned {
            return ((unsigned)(f_it * ntap2 + f_tap) * p.kchunks + f_cc) * brec + bnt;
        };
        auto fadv = [&]() __attribute__((always_inline)) {
            if (f_left > 1) { --f_left; if (++f_tap == ntap2) { f_tap = 0; if (++f_it == p.kt) { f_it = 0; ++f_cc; } } }
        };
        bf16x8 bq0[2][2], bq1[2][2];
        dmaH(0, h_cc, h_it);
        if (++h_it == p.kt) { h_it = 0; ++h_cc; }
        ldBq(bq0, frec()); fadv();
#ifdef DVD_GB_DIST2
        bf16x8 bq2[2][2];
        ldBq(bq1, frec()); fadv();
#endif
        __builtin_amdgcn_s_waitcnt(0x0070 | (0xf << 8));                // vmcnt(0): footprint 0 (this wave's part) landed
        __builtin_amdgcn_s_barrier();
#ifdef DVD_GB_PRIO
        if (__builtin_amdgcn_readfirstlane((int)(blockIdx.x >> 8) & 1)) __builtin_amdgcn_s_setprio(1);   // the second workgroup of a CU
#endif
        int hb = 0, m_tap = 0, m_oc = oc_begin, iy = 0, ix = 0;
        auto step = [&](bf16x8 (&b)[2][2], bf16x8 (&bn)[2][2], int sidx) __attribute__((always_inline)) {
            const bool more = sidx + 1 < nsteps;
            const bool last_tap = m_tap + 1 == ntap2;
            const bool issueH = m_tap == 0 && m_oc + 1 < oc_end;
            const int hy = UP2 ? ((py0 + iy - pad) >> 1) + cpad : py0 + iy;
            const int hx = UP2 ? ((px + ix - pad) >> 1) + cpad : px + ix;
            const int swz = G::sw(hy, hx);
            const char* Ah = hbuf0 + hb * HBYTES + (hy * PITCH + hx) * 64;
            constexpr int NU = 2 * TM;
            bf16x8 a[NU];
            auto ldA = [&](int u) __attribute__((always_inline)) {
                const int kk = u / TM, tm = u % TM, slot = kk * 2 + kh2;
                const int sl = (UP2 && (tm & 1)) ? (slot ^ 2) : slot;
                a[u] = *reinterpret_cast<const bf16x8*>(Ah + tm * TMSTRIDE + ((sl ^ swz) << 4));
            };
            ldA(0); ldA(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (u + 2 < NU) ldA(u + 2);
                const int kk = u / TM, tm = u % TM;
                if constexpr (RELU) a[u] = __builtin_bit_cast(bf16x8, relu16_bf16(__builtin_bit_cast(u32x4, a[u])));
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[kk][tn], acc[tm][tn], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (u == 0) {
                    // behind the first MFMA pair (the compiler's wait for THIS step's fragments sits in front of it): request the
                    // fragments GB_DIST steps ahead, and at the first tap of a chunk the next chunk's footprint
#ifndef DVD_EXP_NODMA
                    ldBq(bn, frec()); fadv();
                    if (issueH) { dmaH(hb ^ 1, h_cc, h_it); if (++h_it == p.kt) { h_it = 0; ++h_cc; } }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            ++m_tap;
            if (++ix == p.kw) { ix = 0; ++iy; }
            if (last_tap) {
                // chunk boundary: every wave has finished reading footprint hb (it is overwritten one chunk from now) and has seen
                // its own part of footprint hb ^ 1 land (the in-order drain in front of the steps since): one barrier per chunk
                m_tap = 0; iy = 0; ix = 0; ++m_oc; hb ^= 1;
                if (++m_it == p.kt) { m_it = 0; ++m_cc; }
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
        int sidx = 0;
#ifdef DVD_GB_DIST2
        for (; sidx + 2 < nsteps; sidx += 3) { step(bq0, bq2, sidx); step(bq1, bq0, sidx + 1); step(bq2, bq1, sidx + 2); }
        if (sidx < nsteps) { step(bq0, bq2, sidx); ++sidx; }
        if (sidx < nsteps) { step(bq1, bq0, sidx); ++sidx; }
#else
        for (; sidx + 1 < nsteps; sidx += 2) { step(bq0, bq1, sidx); step(bq1, bq0, sidx + 1); }
        if (sidx < nsteps) step(bq0, bq1, sidx);
#endif
    }
    __syncthreads();

    float* ep = reinterpret_cast<float*>(&smem[0]) + wave * (32 * 64);
    const int ecol = (lane & 7) * 8, erow = lane >> 3;
    const long long frame_row0 = (long long)ft * (p.H * p.W);
    conv_epilogue<T, TM, DVD_GB_EPI_DEEP>(p, acc, ep, lane, n0 + wn * 64 + ecol, z, frame_row0, [&](int tm, int j) __attribute__((always_inline)) {
        const int pi = wm * (TM * 32) + tm * 32 + j * 8 + erow;
        return (y0 + (pi >> 4)) * p.W + x0 + (pi & 15);
    });
}

template <int TM, int WN, int WMV, bool RELU, bool UP2>
__global__ __launch_bounds__(256, 2) void conv_halo_gb_kernel(ConvK p) {
    __shared__ __attribute__((aligned(16))) char smem[HaloGbCfg<TM, WN, WMV, UP2>::LDSB];
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, qd = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (qd + 1) : rr * (qd + 1) + (xcd - rr) * qd) + (bid >> 3);
    }
    int mt = bid / p.tilesN, nt = bid - mt * p.tilesN;
    if (p.nmajor) { const int tilesM = gridDim.x / p.tilesN; nt = bid / tilesM; mt = bid - nt * tilesM; }
    conv_halo_gb_tile<TM, WN, WMV, RELU, UP2>(p, smem, mt, nt, blockIdx.z);
}

// ---------------------------------------------------------------------------- the same for frames of 8 x 8 and 4 x 4 pixels
// The recurrent convolutions of the first two generator stages (Generator.py:39,43: ConvGRUs on 4 x 4 / 8 x 8 latents) ran through
// the tap-by-tap kernel: per (chunk, tap) one LDS-DMA gather of the activation tile, one of the weight tile, a counted wait and a
// barrier -- K steps of pure issue / wait latency.  Here the M tile is G WHOLE frames (TM*64 / S^2 of them), whose zero-padded
// footprints ((S+4) x (S+4) rows of 64 bytes per frame, PITCH = S + 4 whatever the filter size) sit in LDS for all 9 / 25 taps of a
// channel chunk, and the weights come from L2 in fragment-major order as in conv_halo_gb_tile: no per-tap DMA, wait or barrier.
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

template <int TM, int S, bool RELU>
__device__ __forceinline__ void conv_halo_gbs_tile(const ConvK& p, char* const smem, const int mt, const int nt, const int z) {
    using T = bf16_t;
    using Cfg = HaloGbsCfg<TM, S>;
    constexpr int WN = 2, NWAVE = 4, BNt = 128;
    constexpr int PITCH = Cfg::PITCH, FR = Cfg::FR, G = Cfg::G, HG = Cfg::HG, HBYTES = Cfg::HBYTES;
    constexpr int NH = (HG + NWAVE - 1) / NWAVE;
    constexpr int SS = S * S;
    char* const hbuf0 = &smem[0];
    char* const dump = &smem[2 * HBYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = nt * BNt;
    const int ft0 = mt * G, nframes = p.M / SS;
    const int per = (p.kchunks + p.nsplit - 1) / p.nsplit;
    const int cc_begin = z * per, cc_end = min(p.kchunks, cc_begin + per);
    const int ntap2 = p.kh * p.kw;
    constexpr unsigned esz = 2;
    const int pad = p.kh >> 1, ext = S + 2 * pad;                      // footprint extent actually used by this filter

    const int wu = __builtin_amdgcn_readfirstlane(wave);
    const int lrow = lane >> 2;
    const unsigned ldb = (unsigned)p.ldi * esz;
    const size_t base_b = (size_t)ft0 * SS * ldb;
    const size_t left_b = p.in_bytes - base_b;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.in + base_b), 0, left_b > 0xfffffffeull ? 0xfffffffeu : (unsigned)left_b, 0x00020000);
    const int nb32 = p.nb32;
    const __amdgpu_buffer_rsrc_t rwq = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.wq, 0, (unsigned)((size_t)ntap2 * p.kchunks * nb32 * 2048), 0x00020000);
    unsigned hoff[NH];
    int hq[NH];
    bool hval[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int g = i * NWAVE + wu;
        const int h = g * 16 + lrow;
        const int fl = h / FR, r = h - fl * FR;
        const int hy = r / PITCH, hx = r - hy * PITCH;
        const int yin = hy - pad, xin = hx - pad;
        hq[i] = (lane & 3) ^ (hy & 3);
        hval[i] = g < HG && fl < G && ft0 + fl < nframes && hy < ext && hx < ext && (unsigned)yin < (unsigned)S && (unsigned)xin < (unsigned)S;
        hoff[i] = (unsigned)((fl * S + yin) * S + xin) * ldb + hq[i] * 16;
    }
    auto dmaH = [&](int hb, int cc_) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int g = i * NWAVE + wu;
            char* dst_ = g < HG ? hbuf0 + hb * HBYTES + g * 1024 : dump;
            const bool cv_ = cc_ * 32 + hq[i] * 8 < p.C;
            dma16(rin, dst_, (hval[i] && cv_) ? hoff[i] + cc_ * 64 : 0xffffffffu);
        }
    };
    const unsigned bvoff = (unsigned)(wn * 2 * 2048 + lane * 16);
    const unsigned bnt = (unsigned)(nt * (BNt / 32)) * 2048u;
    const unsigned brec = (unsigned)nb32 * 2048u;
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
    const int nsteps = 0;
#else
    const int nsteps = (cc_end - cc_begin) * ntap2;
#endif
    if (nsteps > 0) {
        int m_cc = cc_begin, h_cc = cc_begin;
        unsigned rec = (unsigned)m_cc * brec + bnt;                     // tap 0 of chunk m_cc; records are [tap][chunk]
        const unsigned tapstep = (unsigned)p.kchunks * brec;
        bf16x8 bq0[2][2], bq1[2][2];
        dmaH(0, h_cc); ++h_cc;
        ldBq(bq0, rec);
        __builtin_amdgcn_s_waitcnt(0x0070 | (0xf << 8));                // vmcnt(0)
        __builtin_amdgcn_s_barrier();
        int hb = 0, m_tap = 0, iy = 0, ix = 0;
        auto step = [&](bf16x8 (&b)[2][2], bf16x8 (&bn)[2][2], int sidx) __attribute__((always_inline)) {
            const bool more = sidx + 1 < nsteps;
            const bool last_tap = m_tap + 1 == ntap2;
            const bool issueH = m_tap == 0 && m_cc + 1 < cc_end;
            unsigned nrec = more ? rec + tapstep : rec;
            if (last_tap && more) nrec = (unsigned)(m_cc + 1) * brec + bnt;
            const int swz = (y0 + iy) & 3;
            const char* Ah = hbuf0 + hb * HBYTES + (arow0 + iy * PITCH + ix) * 64;
            constexpr int NU = 2 * TM;
            bf16x8 a[NU];
            auto ldA = [&](int u) __attribute__((always_inline)) {
                const int kk = u / TM, tm = u % TM, slot = kk * 2 + kh2;
                a[u] = *reinterpret_cast<const bf16x8*>(Ah + tmoff(tm) + ((slot ^ swz) << 4));
            };
            ldA(0); ldA(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (u + 2 < NU) ldA(u + 2);
                const int kk = u / TM, tm = u % TM;
                if constexpr (RELU) a[u] = __builtin_bit_cast(bf16x8, relu16_bf16(__builtin_bit_cast(u32x4, a[u])));
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[kk][tn], acc[tm][tn], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (u == 0) {
                    ldBq(bn, nrec);
                    if (issueH) { dmaH(hb ^ 1, h_cc); ++h_cc; }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            rec = nrec;
            ++m_tap;
            if (++ix == p.kw) { ix = 0; ++iy; }
            if (last_tap) {
                m_tap = 0; iy = 0; ix = 0; ++m_cc; hb ^= 1;
                if (more) {
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_s_barrier();
                }
            }
        };
        int sidx = 0;
        for (; sidx + 1 < nsteps; sidx += 2) { step(bq0, bq1, sidx); step(bq1, bq0, sidx + 1); }
        if (sidx < nsteps) step(bq0, bq1, sidx);
    }
    __syncthreads();

    float* ep = reinterpret_cast<float*>(&smem[0]) + wave * (32 * 64);
    const int ecol = (lane & 7) * 8, erow = lane >> 3;
    const long long row0 = (long long)ft0 * SS;
    conv_epilogue<T, TM, DVD_GB_EPI_DEEP>(p, acc, ep, lane, n0 + wn * 64 + ecol, z, row0, [&](int tm, int j) __attribute__((always_inline)) {
        const int pi = wm * (TM * 32) + tm * 32 + j * 8 + erow;          // the tile's rows are G consecutive whole frames
        return row0 + pi < p.M ? pi : -1;
    });
}

template <int TM, int S, bool RELU>
__global__ __launch_bounds__(256, 2) void conv_halo_gbs_kernel(ConvK p) {
    __shared__ __attribute__((aligned(16))) char smem[HaloGbsCfg<TM, S>::LDSB];
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, qd = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (qd + 1) : rr * (qd + 1) + (xcd - rr) * qd) + (bid >> 3);
    }
    int mt = bid / p.tilesN, nt = bid - mt * p.tilesN;
    if (p.nmajor) { const int tilesM = gridDim.x / p.tilesN; nt = bid / tilesM; mt = bid - nt * tilesM; }
    conv_halo_gbs_tile<TM, S, RELU>(p, smem, mt, nt, blockIdx.z);
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

}  // namespace

// ---------------------------------------------------------------------------- launchers (called by conv_igemm.hip's dispatch)
namespace dvdk {
// variant: 0 = 256 x 128 tile, 1 = 128 x 128 (launches with few rows), 2 = 256 x 64 (thin outputs)
void launch_gb(const ConvK& p, int variant, bool relu_in, bool up2, dim3 grid, hipStream_t st) {
#define LAUNCH_GB(TM_, WN_, WMV_)                                                                       \
    do { if (relu_in) { if (up2) conv_halo_gb_kernel<TM_, WN_, WMV_, true, true><<<grid, 256, 0, st>>>(p);       \
                        else conv_halo_gb_kernel<TM_, WN_, WMV_, true, false><<<grid, 256, 0, st>>>(p); }           \
         else         { if (up2) conv_halo_gb_kernel<TM_, WN_, WMV_, false, true><<<grid, 256, 0, st>>>(p);      \
                        else conv_halo_gb_kernel<TM_, WN_, WMV_, false, false><<<grid, 256, 0, st>>>(p); } } while (0)
    if (variant == 2) LAUNCH_GB(2, 1, 4); else if (variant == 0) LAUNCH_GB(4, 2, 2); else LAUNCH_GB(2, 2, 2);
#undef LAUNCH_GB
}
// whole-frame footprints: S = 8 (256- or 128-row tiles) or S = 4 (128-row tiles)
void launch_gbs(const ConvK& p, int S, bool big, bool relu_in, dim3 grid, hipStream_t st) {
#define LAUNCH_GBS(TM_, SZ_) do { if (relu_in) conv_halo_gbs_kernel<TM_, SZ_, true><<<grid, 256, 0, st>>>(p);   \
                                  else conv_halo_gbs_kernel<TM_, SZ_, false><<<grid, 256, 0, st>>>(p); } while (0)
    if (S == 8) { if (big) LAUNCH_GBS(4, 8); else LAUNCH_GBS(2, 8); }
    else LAUNCH_GBS(2, 4);
#undef LAUNCH_GBS
}
}  // namespace dvdk

// Fragment-major image of a forward (or backward-data) pack for conv_halo_gb_kernel; see the comment there.
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
