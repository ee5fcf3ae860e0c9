// Weight gradients of the convolutions: conv_wgrad_row_kernel (one filter row per workgroup, 3 x 3 / 5 x 5 / 3 x 3 x 3) and
// conv_wgrad_kernel (one tap per workgroup: 1 x 1, upsampling and exact-mode convolutions), their fixed-order reduce kernels,
// and the C ABI entry dvd_conv_wgrad.
#include "conv_common.h"
#include "prof.h"
#include <cstdlib>
#include <type_traits>

namespace {

// ============================================================================ backward-weight
struct WgK {
    const char* x; const char* dy; float* dw;
    int M, C, ldx, Cin_real, Cout, Cy, ldy;
    int T, H, W, logH, logW, Hin, Win;
    int kt, kh, kw, up2, relu_in;
    int tiles_co, tiles_ci, rows_per_split;
    long long s_co, s_ci, s_tap;
    size_t x_bytes, dy_bytes;
    int maxshift, xcd_remap;
    float* dbias;
    float* wsb;             // row kernel, workspace path: bias partial sums [slice][workgroup][BMc] behind the tile partials
    float* ws;                           // partial tiles [slice][x-block][BMc*BNc] when non-null
    int fold, cout_f;                    // conv_wgrad_row4_kernel<FOLD>: x2-upsampled input folded onto the input grid (cout_f = the layer's Cout)
};

// D[co][ci] = sum over rows m of dy[m][co] * x[pos(m)+tap][ci].  The reduction index (rows) is the
// MFMA K dimension, so both operand tiles must be K(row)-contiguous per channel in LDS:
//   bf16: tiles are staged in their natural [row][channel] image with 16-byte loads and the
//         fragments are built by the LDS transpose read ds_read_b64_tr_b16;
//   f32 : the natural [row][channel] image already matches the 32x32x2 fragment (1 float/lane).
template <typename T, int TA, int TB>      // wave tile (TA*32 out-channels) x (TB*32 in-channels); block = 2 x 2 waves
__global__ __launch_bounds__(NT) void conv_wgrad_kernel(WgK p) {
    constexpr bool kBf16 = sizeof(T) == 2;
    static_assert(kBf16 || (TA == 2 && TB == 2), "exact mode uses the 128 x 128 tile");
    constexpr int BKW = kBf16 ? 32 : 16;            // rows reduced per step
    constexpr int BMc = TA * 64, BNc = TB * 64;     // block tile: out-channels x in-channels
    constexpr int RSA = BMc * 2 + 64, RSB = BNc * 2 + 64;          // bf16 LDS row strides (bytes)
    constexpr int TA_BYTES = kBf16 ? 32 * RSA : TILEB, TB_BYTES = kBf16 ? 32 * RSB : TILEB;
    __shared__ __attribute__((aligned(16))) char smem[2][TA_BYTES + TB_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware order over (x, z): each XCD gets a contiguous run of row slices, whose workgroups (all
    // taps / channel tiles of a slice) then share the slice's rows in that XCD's L2 (hit rate 0.42 -> 0.73
    // on the 3x3 shapes, 0.81 -> 0.89 on the 5x5 ones; tools/pmc_l2.txt)
    int bx = blockIdx.x, bz = blockIdx.z;
    if (p.xcd_remap) {
        const int gx = gridDim.x, total = gx * gridDim.z, lin = bx + gx * bz;
        const int xcd = lin & 7, qd = total >> 3, rr = total & 7;
        const int lp = (xcd < rr ? xcd * (qd + 1) : rr * (qd + 1) + (xcd - rr) * qd) + (lin >> 3);
        bz = lp / gx; bx = lp - bz * gx;
    }
    const int tiles = p.tiles_co * p.tiles_ci;
    const int tap = bx / tiles;
    const int rem = bx - tap * tiles;
    const int tco = rem / p.tiles_ci, tci = rem - tco * p.tiles_ci;
    const int co0 = tco * BMc, ci0 = tci * BNc;
    const int it = tap / (p.kh * p.kw), r2 = tap - it * p.kh * p.kw;
    const int iy = r2 / p.kw, ix = r2 - iy * p.kw;
    const int dt = it - (p.kt >> 1), dy_ = iy - (p.kh >> 1), dx = ix - (p.kw >> 1);
    const int m_begin = bz * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);

    f32x16 acc[TA][TB];
    {
        const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int b = 0; b < TB; ++b) acc[a][b] = zacc;
    }

    // Buffer descriptors: invalid rows / channels use offset 0xFFFFFFFF -> hardware returns zeros.
    // (32-bit offsets relative to the first row this workgroup's row slice can touch: tensors > 4 GiB ok)
    constexpr unsigned ESZ = sizeof(T);
    const int xbase_row = p.up2 ? (m_begin / (p.H * p.W)) * p.Hin * p.Win : max(0, m_begin - p.maxshift);
    const size_t xbase_b = (size_t)xbase_row * p.ldx * ESZ, ybase_b = (size_t)m_begin * p.ldy * ESZ;
    const size_t xleft = p.x_bytes - xbase_b, yleft = p.dy_bytes > ybase_b ? p.dy_bytes - ybase_b : 0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + xbase_b), 0, xleft > 0xfffffffeull ? 0xfffffffeu : (unsigned)xleft, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.dy + ybase_b), 0, yleft > 0xfffffffeull ? 0xfffffffeu : (unsigned)yleft, 0x00020000);
    // Byte offset of the shifted x row for output row m.  Without upsampling the input grid equals
    // the output grid, so the shifted row is simply m + delta.
    const int delta = (dt * p.H + dy_) * p.W + dx;
    auto xoff = [&](int m, unsigned chan_bytes, bool cvalid) __attribute__((always_inline)) -> unsigned {
        int fm, ym, xm;
        grid_pos(m, p.H, p.W, p.logH, p.logW, fm, ym, xm);
        const int xx = xm + dx, yy = ym + dy_;
        int tt = dt;
        if (p.kt > 1) tt += fm % p.T;
        const bool ok = cvalid && m < m_end && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W &&
                        (unsigned)tt < (unsigned)p.T;
        int row = m + delta;
        if (p.up2) row = ((fm + dt) * p.Hin + (yy >> 1)) * p.Win + (xx >> 1);
        return ok ? (unsigned)(row - xbase_row) * ((unsigned)p.ldx * ESZ) + chan_bytes : 0xffffffffu;
    };
    auto yoff = [&](int m, unsigned chan_bytes, bool cvalid) __attribute__((always_inline)) -> unsigned {
        return (cvalid && m < m_end) ? (unsigned)(m - m_begin) * ((unsigned)p.ldy * ESZ) + chan_bytes : 0xffffffffu;
    };

    if constexpr (kBf16) {
        // Natural [row][channel] LDS image (row stride 320 B = 256 B of channels + 64 B: the four
        // rows a 16-lane group touches land on disjoint bank quarters).  MFMA fragments need 8
        // consecutive ROWS of one channel per lane: ds_read_b64_tr_b16 delivers exactly that
        // (each 16-lane group reads a 4-row x 16-channel block and hands lane i column i).
        // operand A = dy (BMc channels per row), operand B = x (BNc channels per row)
        constexpr int CPRA = BMc / 8, RPPA = NT / CPRA, NPA = 32 / RPPA;       // chunks/row, rows/pass, passes
        constexpr int CPRB = BNc / 8, RPPB = NT / CPRB, NPB = 32 / RPPB;
        const int rra = tid / CPRA, cka = tid % CPRA, rrb = tid / CPRB, ckb = tid % CPRB;
        const int cy = co0 + cka * 8, cx = ci0 + ckb * 8;
        const bool cyv = cy < p.Cy, cxv = cx < p.C;
        u32x4 ra[NPA], rb[NPB];
        auto wg_gload = [&](int mk) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NPA; ++i)
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rdy, yoff(mk + rra + i * RPPA, cy * 2, cyv), 0, 0);
#pragma unroll
            for (int i = 0; i < NPB; ++i)
                rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff(mk + rrb + i * RPPB, cx * 2, cxv), 0, 0);
        };
        // fused bias gradient: the centre-tap / first-ci-tile workgroups sum the dy chunks they stage
        const bool do_bias = p.dbias != nullptr && dt == 0 && dy_ == 0 && dx == 0 && tci == 0;
        float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        auto wg_lstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                *reinterpret_cast<u32x4*>(&smem[buf][(rra + i * RPPA) * RSA + cka * 16]) = ra[i];
                if (do_bias) {
                    bs[0] += __uint_as_float(ra[i].x << 16); bs[1] += __uint_as_float(ra[i].x & 0xffff0000u);
                    bs[2] += __uint_as_float(ra[i].y << 16); bs[3] += __uint_as_float(ra[i].y & 0xffff0000u);
                    bs[4] += __uint_as_float(ra[i].z << 16); bs[5] += __uint_as_float(ra[i].z & 0xffff0000u);
                    bs[6] += __uint_as_float(ra[i].w << 16); bs[7] += __uint_as_float(ra[i].w & 0xffff0000u);
                }
            }
#pragma unroll
            for (int i = 0; i < NPB; ++i)
                *reinterpret_cast<u32x4*>(&smem[buf][TA_BYTES + (rrb + i * RPPB) * RSB + ckb * 16]) =
                    p.relu_in ? relu16_bf16(rb[i]) : rb[i];
        };
#define WG_GLOAD(mk) wg_gload(mk)
#define WG_LSTORE(buf) wg_lstore(buf)
        // per-lane offset of its 8-byte chunk inside a 4-row x 16-channel block of a fragment:
        // rows (g16>>1)*8 + (i16>>2), channels (g16&1)*16 + (i16&3)*4
        const int g16 = lane >> 4, i16 = lane & 15;
        const int frow = (g16 >> 1) * 8 + (i16 >> 2), fcol2 = ((g16 & 1) * 16 + (i16 & 3) * 4) * 2;
        auto frag = [&](const char* tile, int rs, int col0, int kb) __attribute__((always_inline)) -> bf16x8 {
            typedef __attribute__((ext_vector_type(4))) short s16x4;
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            const char* pz = tile + (frow + kb) * rs + fcol2 + col0 * 2;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pz);
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pz + 4 * rs));
            s16x8 f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            return __builtin_bit_cast(bf16x8, f);
        };
        auto mma = [&](int buf) __attribute__((always_inline)) {
            const char* At = &smem[buf][0];
            const char* Bt = &smem[buf][TA_BYTES];
            // all fragment reads of the 32-row step are issued up front: the second half's transpose
            // reads land while the first half's MFMAs run
            bf16x8 fa0[TA], fb0[TB], fa1[TA], fb1[TB];
#pragma unroll
            for (int a = 0; a < TA; ++a) fa0[a] = frag(At, RSA, wm * (TA * 32) + a * 32, 0);
#pragma unroll
            for (int b = 0; b < TB; ++b) fb0[b] = frag(Bt, RSB, wn * (TB * 32) + b * 32, 0);
#pragma unroll
            for (int a = 0; a < TA; ++a) fa1[a] = frag(At, RSA, wm * (TA * 32) + a * 32, 16);
#pragma unroll
            for (int b = 0; b < TB; ++b) fb1[b] = frag(Bt, RSB, wn * (TB * 32) + b * 32, 16);
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int b = 0; b < TB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[a], fb0[b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int b = 0; b < TB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[a], fb1[b], acc[a][b], 0, 0, 0);
        };
        if (m_begin < m_end) {
            WG_GLOAD(m_begin);
            WG_LSTORE(0);
            __syncthreads();
            int buf = 0;
            for (int mk = m_begin; mk < m_end; mk += BKW, buf ^= 1) {
                const bool more = mk + BKW < m_end;
                if (more) WG_GLOAD(mk + BKW);
                mma(buf);
                if (more) WG_LSTORE(buf ^ 1);
                __syncthreads();
            }
        }
#undef WG_GLOAD
#undef WG_LSTORE
        if (do_bias) {                                   // threads sharing a channel chunk: rra = 0..RPPA-1
            float* red = reinterpret_cast<float*>(&smem[0][0]);
#pragma unroll
            for (int k = 0; k < 8; ++k) red[tid * 8 + k] = bs[k];
            __syncthreads();
            if (tid < CPRA) {
                for (int k = 0; k < 8; ++k) {
                    float a = 0.f;
                    for (int r = 0; r < RPPA; ++r) a += red[(r * CPRA + tid) * 8 + k];
                    const int co = co0 + tid * 8 + k;
                    if (p.wsb) p.wsb[((size_t)bz * p.tiles_co + tco) * BMc + tid * 8 + k] = a;     // partial of (slice, channel tile): wgrad_bias_reduce_kernel
                    else if (co < p.Cout && a != 0.f) atomicAdd(p.dbias + co, a);                  // (a single slice: one addition per channel)
                }
            }
            __syncthreads();
        }
    } else {
        // f32: LDS image [16 rows][WG_LD floats]; thread stages rows kr, kr+8, 16-byte chunk ch
        const int ch = tid & 31, kr = tid >> 5;
        const int cy = co0 + ch * 4, cx = ci0 + ch * 4;
        const bool cyv = cy < p.Cy, cxv = cx < p.C;
        u32x4 a0, a1, b0, b1;
        auto gload = [&](int mk) __attribute__((always_inline)) {
            a0 = __builtin_amdgcn_raw_buffer_load_b128(rdy, yoff(mk + kr, cy * 4, cyv), 0, 0);
            a1 = __builtin_amdgcn_raw_buffer_load_b128(rdy, yoff(mk + kr + 8, cy * 4, cyv), 0, 0);
            b0 = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff(mk + kr, cx * 4, cxv), 0, 0);
            b1 = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff(mk + kr + 8, cx * 4, cxv), 0, 0);
        };
        const bool do_bias = p.dbias != nullptr && dt == 0 && dy_ == 0 && dx == 0 && tci == 0;
        float bs[4] = {0.f, 0.f, 0.f, 0.f};
        auto lstore = [&](int buf) __attribute__((always_inline)) {
            if (do_bias) {
                bs[0] += __uint_as_float(a0.x) + __uint_as_float(a1.x); bs[1] += __uint_as_float(a0.y) + __uint_as_float(a1.y);
                bs[2] += __uint_as_float(a0.z) + __uint_as_float(a1.z); bs[3] += __uint_as_float(a0.w) + __uint_as_float(a1.w);
            }
            *reinterpret_cast<u32x4*>(&smem[buf][(kr * WG_LD + ch * 4) * 4]) = a0;
            *reinterpret_cast<u32x4*>(&smem[buf][((kr + 8) * WG_LD + ch * 4) * 4]) = a1;
            *reinterpret_cast<u32x4*>(&smem[buf][TA_BYTES + (kr * WG_LD + ch * 4) * 4]) = p.relu_in ? relu16_f32(b0) : b0;
            *reinterpret_cast<u32x4*>(&smem[buf][TA_BYTES + ((kr + 8) * WG_LD + ch * 4) * 4]) = p.relu_in ? relu16_f32(b1) : b1;
        };
        auto mma = [&](int buf) {
            const float* As = reinterpret_cast<const float*>(&smem[buf][0]) + wm * 64 + (lane & 31);
            const float* Bs = reinterpret_cast<const float*>(&smem[buf][TA_BYTES]) + wn * 64 + (lane & 31);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int k = kk * 2 + (lane >> 5);
                const float a0 = As[k * WG_LD], a1 = As[k * WG_LD + 32];
                const float b0 = Bs[k * WG_LD], b1 = Bs[k * WG_LD + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        };
        if (m_begin < m_end) {
            gload(m_begin);
            lstore(0);
            __syncthreads();
            int buf = 0;
            for (int mk = m_begin; mk < m_end; mk += BKW, buf ^= 1) {
                const bool more = mk + BKW < m_end;
                if (more) gload(mk + BKW);
                mma(buf);
                if (more) lstore(buf ^ 1);
                __syncthreads();
            }
        }
        if (do_bias) {                                   // threads sharing a channel chunk: kr = 0..7
            float* red = reinterpret_cast<float*>(&smem[0][0]);
#pragma unroll
            for (int k = 0; k < 4; ++k) red[tid * 4 + k] = bs[k];
            __syncthreads();
            if (tid < 32) {
                for (int k = 0; k < 4; ++k) {
                    float a = 0.f;
                    for (int r = 0; r < 8; ++r) a += red[(r * 32 + tid) * 4 + k];
                    const int co = co0 + tid * 4 + k;
                    if (p.wsb) p.wsb[((size_t)bz * p.tiles_co + tco) * BMc + tid * 4 + k] = a;
                    else if (co < p.Cout && a != 0.f) atomicAdd(p.dbias + co, a);
                }
            }
            __syncthreads();
        }
    }

    // epilogue: each 32x32 accumulator tile goes through a per-wave LDS block and is then added to
    // dw with a rolled loop (an unrolled 128-atomic epilogue costs ~170 VGPRs of addresses)
    float* ep = reinterpret_cast<float*>(&smem[0][0]) + wave * (32 * 32);
#pragma unroll
    for (int ta = 0; ta < TA; ++ta)
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ep[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[ta][tb][r];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            const int ci = ci0 + wn * (TB * 32) + tb * 32 + (lane & 31);
            const int cob = co0 + wm * (TA * 32) + ta * 32 + (lane >> 5);
            float* dst = p.dw + ci * p.s_ci + tap * p.s_tap;
            if (p.ws) {
                // partial tile of this row slice, tile-local [row][col] layout, coalesced plain stores
                float* wt = p.ws + ((size_t)bz * gridDim.x + bx) * (size_t)(BMc * BNc) +
                            (size_t)(wm * (TA * 32) + ta * 32) * BNc + wn * (TB * 32) + tb * 32;
#pragma unroll 1
                for (int j = 0; j < 16; ++j)
                    wt[(size_t)(2 * j + (lane >> 5)) * BNc + (lane & 31)] = ep[(2 * j + (lane >> 5)) * 32 + (lane & 31)];
            } else if (ci < p.Cin_real) {
#pragma unroll 1
                for (int j = 0; j < 16; ++j) {
                    const int co = cob + 2 * j;
                    const float v = ep[(2 * j + (lane >> 5)) * 32 + (lane & 31)];
                    if (co < p.Cout && v != 0.f) atomicAdd(dst + co * p.s_co, v);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
}

// ============================================================================ backward-weight, one filter ROW per workgroup
// Same contraction as conv_wgrad_kernel, regrouped so the staged operands are reused across taps: a workgroup owns
// (filter row iy, 64*WM out-channels, 64 in-channels) and ALL KW taps of that row.  Per 32-pixel step it stages
// the dy rows once and the input FOOTPRINT of the step -- the 32 pixels plus KW-1 halo columns (two 16-pixel lines
// when W = 16) -- once; the B fragment of tap ix is the same footprint read ix rows further on.  Bytes staged per
// MFMA drop from 384 (256 x 128 tile, one tap) to ~130, and dy / x are fetched from L2 KH instead of KH*KW times.
//   waves: WM (64 out-channels each) x 2 (32 in-channels each); wave tile 64 x (KW taps x 32) = 2 x KW accumulators
//   LDS:   dy [32 rows][64*WM ch] (+64 B pad, transposing reads as in conv_wgrad_kernel);
//          x  [2 halves of 32 ch][48 rows][64 B]: the 4 consecutive rows a ds_read_b64_tr_b16 pass touches are
//             256 contiguous bytes (conflict-free without a swizzle) and a tap is a +64-byte immediate offset.
// UP2: the convolution reads a nearest-x2 upsampled input (GResBlock.py:57-58): the footprint is kept in INPUT
// coordinates (half the columns), tap ix of step pixel pk reads row ((pk + ix - pad) >> 1) + 1 of it.
// 3 x 3 filters stay near 0.9 PF/s: with three taps per row the LDS is the limit, not the staging traffic -- per stage of the
// 256-channel tile 640 cycles of fragment reads + 624 of ds_write_b128 (13 cycles each) against 1536 MFMA cycles, 82 % busy
// (5 taps: 59 %).  A nine-tap variant (128 x 64 channel tile, dy staged once for all three filter rows, 146 staged bytes per
// MFMA instead of 212) was built and passed the tests: +3 % on 786 k x 256 -> 256, -2 % on 3.1 M x 128 -> 128 (its 10 fragment
// reads per 9 MFMAs keep the LDS as busy); removed.
// NH: 32-channel blocks of input channels per workgroup (2 = 64 channels; 4 = 128, used with WM = 2 so that the 128-channel
// output tile also runs as ONE 8-wave workgroup per CU and can stagger its two halves, see the main loop).
template <int WM, int KW, bool RELU, bool UP2 = false, int NH = 2>
__global__ __launch_bounds__(WM * NH * 64) void conv_wgrad_row_kernel(WgK p) {
    constexpr int NWAVE = WM * NH, NTt = NWAVE * 64, BMc = WM * 64, BNc = NH * 32;
    // round 6: per-thread-constant input addresses (see the loader set-up).  Measured per tile (tools/conv_microbench.py wgrad, interleaved):
    // 3 taps +1...4 % (8-wave tiles), +11 % (64-channel tile); 5 taps on the 128 x 128 tile +1.5 % but two spilled registers;
    // 5 taps on the 256 x 64 tile -4.5 % (its staggered halves lose their balance) -> the 5-tap tiles keep the round-5 form.
    constexpr bool FASTADDR = !UP2 && KW == 3;
    constexpr bool UPFAST = UP2 && KW == 3;                       // the same for the x2-upsampled input of one-segment steps (W >= 32), chosen at run time
    constexpr int RSA = BMc * 2 + 64;
    constexpr int TA_BYTES = 32 * RSA;
    constexpr int XROWS = KW == 5 ? 64 : 48, XHALF = XROWS * 64, TB_BYTES = NH * XHALF;   // W = 8, 5 taps: 4 lines x 12 rows; W = 4: 8 lines x 8 (6) rows
    constexpr int NS = 2;      // 32-pixel sub-steps per barrier (3x3: 0.72 -> 0.80 PF/s, 5x5: +3 %; three sub-steps cost occupancy)
    constexpr int SUB = TA_BYTES + TB_BYTES, STAGE = NS * SUB;
    constexpr int EPIB = NWAVE * 32 * 32 * 4;
    constexpr int REDB = NTt * 8 * 4;                            // bias partial sums
    // 8-wave tiles sit at the 256-register limit: their bias column sums live in an LDS table behind the stage buffers (ds_add_f32 on
    // the thread's own slots, same order of additions as the register form) instead of eight registers held through the main loop
    constexpr bool BLDS = false;      // (measured: ds_add_f32 on a [8][threads] table costs 40-170 % on launches WITH a bias gradient -- registers it is)
    constexpr int LDSB0 = 2 * STAGE > EPIB ? (2 * STAGE > REDB ? 2 * STAGE : REDB) : (EPIB > REDB ? EPIB : REDB);
    constexpr int LDSB = LDSB0 + (BLDS ? REDB : 0);
    __shared__ __attribute__((aligned(16))) char smem[LDSB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NH, wn = wave % NH;
    int bx = blockIdx.x, bz = blockIdx.z;
    if (p.xcd_remap) {
        const int gx = gridDim.x, total = gx * gridDim.z, lin = bx + gx * bz;
        const int xcd = lin & 7, qd = total >> 3, rr = total & 7;
        const int lp = (xcd < rr ? xcd * (qd + 1) : rr * (qd + 1) + (xcd - rr) * qd) + (lin >> 3);
        bz = lp / gx; bx = lp - bz * gx;
    }
    const int tiles = p.tiles_co * p.tiles_ci;
    const int irow = bx / tiles;                                 // filter row index: it * kh + iy
    const int rem = bx - irow * tiles;
    const int tco = rem / p.tiles_ci, tci = rem - tco * p.tiles_ci;
    const int co0 = tco * BMc, ci0 = tci * BNc;
    constexpr int pad = KW >> 1;
    const int it = irow / KW, iy = irow - it * KW;               // kh == kw
    const int dyl = iy - pad, dtl = it - (p.kt >> 1);
    const int m_begin = bz * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);

    f32x16 acc[2][KW];
    {
        const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int t = 0; t < KW; ++t) acc[a][t] = zacc;
    }
    const int xbase_row = UP2 ? (m_begin >> (p.logW + p.logH)) * p.Hin * p.Win : max(0, m_begin - p.maxshift);
    const size_t xbase_b = (size_t)xbase_row * p.ldx * 2, ybase_b = (size_t)m_begin * p.ldy * 2;
    const size_t xleft = p.x_bytes - xbase_b, yleft = p.dy_bytes > ybase_b ? p.dy_bytes - ybase_b : 0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + xbase_b), 0, xleft > 0xfffffffeull ? 0xfffffffeu : (unsigned)xleft, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.dy + ybase_b), 0, yleft > 0xfffffffeull ? 0xfffffffeu : (unsigned)yleft, 0x00020000);
    // step geometry: 32 pixels = one 32-pixel segment of a line (W >= 32), two 16-pixel lines or four 8-pixel lines
    constexpr int cpad = (pad + 1) >> 1;
    const int segw = min(p.W, 32), logsegw = min(p.logW, 5);
    const int fpr = UP2 ? (segw >> 1) + 2 : segw + KW - 1, frows = (32 >> logsegw) * fpr;
    // dy loader: NPA 16-byte chunks per thread
    constexpr int CPRA = BMc / 8, RPPA = NTt / CPRA, NPA = 32 / RPPA;
    const int rra = tid / CPRA, cka = tid % CPRA;
    const int cy = co0 + cka * 8;
    const bool cyv = cy < p.Cy;
    // x footprint loader: XROWS * CPX 16-byte chunks over NTt threads
    constexpr int CPX = NH * 4;
    constexpr int NXL = (XROWS * CPX + NTt - 1) / NTt;
    int xseg[NXL], xj[NXL], xdst[NXL];
    unsigned xcb[NXL];
    bool xcv[NXL];
    // round 6 (not UP2): a sub-step starts at a multiple of 32 pixels -- whole lines (W <= 32: 32 / W of them, starting at a line that
    // is a multiple of that; two whole frames when the frame is 4 x 4) or one 32-pixel segment of a line -- so the input row of chunk i is
    // mk + (a per-thread constant), and only its validity depends on where the sub-step sits in the frame: (y + xya) in [0, H),
    // (x0 + xxa) in [0, W).  Rebuilding line / column / frame of every chunk from mk cost ~13 VALU per chunk: 3.4 VALU per MFMA on the
    // 3-tap tiles, which are bound by instruction issue (7.5 non-MFMA instructions per MFMA and wave, two waves per SIMD).
    // At most ONE of the two checks is per thread: W > 32 -> the step is one line (every chunk has the same line offset: the line
    // check is wave-uniform, the column check per thread); W <= 32 -> x0 = 0 (the column check is a per-thread constant, folded
    // into xcv; the line check per thread).  xa = the per-thread addend of the remaining check.
    int xa[NXL];
    unsigned xld[NXL];
    const bool twofr = (32 >> logsegw) > p.H;                    // 4 x 4 frames: the step's eight lines span two frames (y = x0 = 0)
    const bool wide = p.W > 32;
    const unsigned ldx2 = (unsigned)p.ldx * 2, ldy2 = (unsigned)p.ldy * 2;
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
        const int q = tid + i * NTt, row = q / CPX, c8 = q % CPX;
        xseg[i] = row / fpr;
        xj[i] = row - xseg[i] * fpr;
        const int cx = ci0 + c8 * 8;
        xcv[i] = row < frows && cx < p.C;
        xcb[i] = (unsigned)cx * 2;
        xdst[i] = q < XROWS * CPX ? (c8 >> 2) * XHALF + row * 64 + (c8 & 3) * 16 : -1;
        if constexpr (FASTADDR) {
            const int fo = twofr ? xseg[i] >> p.logH : 0, yl = twofr ? xseg[i] & (p.H - 1) : xseg[i];
            const int xya = yl + dyl, xxa = xj[i] - pad;
            xld[i] = (unsigned)((fo << (p.logW + p.logH)) + (xya << p.logW) + xxa) * ldx2 + xcb[i];   // (wraps for negative deltas; the sum with the uniform part is exact)
            xa[i] = wide ? xxa : xya;
            if (!wide) xcv[i] = xcv[i] && (unsigned)xxa < (unsigned)p.W;
        }
        if constexpr (UPFAST) {           // x2-upsampled input, one line segment per step (W >= 32): input row = uniform + (xj - cpad)
            xa[i] = xj[i] - cpad;
            xld[i] = (unsigned)xa[i] * ldx2 + xcb[i];
        }
    }
    const bool upfast = UPFAST && p.W >= 32;
    const unsigned yld = (unsigned)rra * ldy2 + (unsigned)cy * 2;
    const int dt_rows = p.kt > 1 ? dtl << (p.logW + p.logH) : 0;
    u32x4 ra[NS][NPA], rb[NS][NXL];
    auto gload1 = [&](int mk, u32x4 (&ra)[NPA], u32x4 (&rb)[NXL]) __attribute__((always_inline)) {
        if constexpr (FASTADDR) {
            // slices and M are multiples of 32 rows: a sub-step is inside the slice or outside it as a whole
            const bool ytok = mk < m_end;
            bool tok = ytok;
            if (p.kt > 1) {
                const int tt = (mk >> (p.logW + p.logH)) % p.T + dtl;
                tok = tok && (unsigned)tt < (unsigned)p.T;
            }
            const unsigned yoff = (unsigned)(mk - m_begin) * ldy2, xoff = (unsigned)(mk + dt_rows - xbase_row) * ldx2;
            const int x0 = mk & (p.W - 1), y = twofr ? 0 : (mk >> p.logW) & (p.H - 1);
            if (wide) tok = tok && (unsigned)(y + dyl) < (unsigned)p.H;
            const int sval = wide ? x0 : y;
            const unsigned slim = wide ? p.W : p.H;
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                const unsigned off = (cyv && ytok) ? yld + yoff + (unsigned)(i * RPPA) * ldy2 : 0xffffffffu;
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rdy, off, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NXL; ++i) {
                const bool ok = xcv[i] && tok && (unsigned)(sval + xa[i]) < slim;
                rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? xld[i] + xoff : 0xffffffffu, 0, 0);
            }
            return;
        }
        if (UPFAST && upfast) {
            // one 32-pixel segment of an OUTPUT line: its input line (yy >> 1) is wave-uniform, the input column is (x0 >> 1) - cpad + xj
            const bool ytok = mk < m_end;
            const int x0 = mk & (p.W - 1), y = (mk >> p.logW) & (p.H - 1), yy = y + dyl;
            const bool tok = ytok && (unsigned)yy < (unsigned)p.H;
            const int xs = x0 >> 1;
            const unsigned yoff = (unsigned)(mk - m_begin) * ldy2;
            const unsigned xoff = (unsigned)((mk >> (p.logW + p.logH)) * (p.Hin * p.Win) + (yy >> 1) * p.Win + xs - xbase_row) * ldx2;
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                const unsigned off = (cyv && ytok) ? yld + yoff + (unsigned)(i * RPPA) * ldy2 : 0xffffffffu;
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rdy, off, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NXL; ++i) {
                const bool ok = xcv[i] && tok && (unsigned)(xs + xa[i]) < (unsigned)p.Win;
                rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? xld[i] + xoff : 0xffffffffu, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int m = mk + rra + i * RPPA;
            const unsigned off = (cyv && m < m_end) ? (unsigned)(m - m_begin) * ((unsigned)p.ldy * 2) + cy * 2 : 0xffffffffu;
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rdy, off, 0, 0);
        }
        const int x0 = mk & (p.W - 1), y = (mk >> p.logW) & (p.H - 1);
        int frow0 = mk - (y << p.logW) - x0;                       // first row of the frame
        bool tok = mk < m_end;
        if (p.kt > 1) {                                            // 3-D: the footprint comes from frame t + dt
            const int tt = (mk >> (p.logW + p.logH)) % p.T + dtl;
            tok = tok && (unsigned)tt < (unsigned)p.T;
            frow0 += dtl << (p.logW + p.logH);
        }
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            // output line the tap row looks at.  A 32-pixel step of 4 x 4 frames spans TWO frames (8 lines): line y + xseg of the
            // step is line (.. & (H-1)) of frame (.. >> logH) counted from the step's first frame
            const int lf = y + xseg[i], fo = lf >> p.logH;
            const int yy = (lf & (p.H - 1)) + dyl;
            bool ok = xcv[i] && tok && (unsigned)yy < (unsigned)p.H;
            int row;
            if (UP2) {
                const int xin = (x0 >> 1) - cpad + xj[i];
                ok = ok && (unsigned)xin < (unsigned)p.Win;
                row = ((mk >> (p.logW + p.logH)) + fo) * (p.Hin * p.Win) + (yy >> 1) * p.Win + xin;
            } else {
                const int xx = x0 + xj[i] - pad;
                ok = ok && (unsigned)xx < (unsigned)p.W;
                row = frow0 + (fo << (p.logW + p.logH)) + (yy << p.logW) + xx;
            }
            const unsigned off = ok ? (unsigned)(row - xbase_row) * ((unsigned)p.ldx * 2) + xcb[i] : 0xffffffffu;
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
        }
    };
    auto gload = [&](int mk) __attribute__((always_inline)) {
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) gload1(mk + 32 * s_, ra[s_], rb[s_]);
    };
    // Bias gradient = column sums of dy.  The dy tile is tap-independent, so ANY workgroup of a (tco, slice) can sum any of its
    // 32-pixel sub-steps.  With the slice workspace the KW * tiles_ci workgroups of the centre time slab take the sub-steps in turn
    // (sub-step n belongs to share n % nshare) and leave their partial sums in the workspace for wgrad_row_bias_reduce_kernel:
    // when only the centre-row, tci == 0 workgroups summed -- every sub-step, ~16 VALU per 16-byte chunk beside their MFMAs -- they
    // were the stragglers of the launch (+6...11 % on the large shapes), and their atomics formed one chain per channel.
    const bool spread = p.ws != nullptr;
    const bool do_bias = p.dbias != nullptr && dtl == 0 && (spread || (dyl == 0 && tci == 0));
    const int nshare = spread ? KW * p.tiles_ci : 1, myshare = spread ? iy * p.tiles_ci + tci : 0;
    int bphase = 0;                                              // share of the next sub-step to be stored
    float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float* btab = reinterpret_cast<float*>(&smem[LDSB0]) + tid;       // BLDS: [8][NTt] floats, slot k of this thread at btab[k * NTt]
    if (BLDS && do_bias) {
#pragma unroll
        for (int k = 0; k < 8; ++k) btab[k * NTt] = 0.f;
    }
    auto lstore1 = [&](char* st, const u32x4 (&ra)[NPA], const u32x4 (&rb)[NXL]) __attribute__((always_inline)) {
        const bool mine = do_bias && bphase == myshare;          // wave-uniform
        if (++bphase == nshare) bphase = 0;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            *reinterpret_cast<u32x4*>(st + (rra + i * RPPA) * RSA + cka * 16) = ra[i];
            if (mine) {
                const float v[8] = {__uint_as_float(ra[i].x << 16), __uint_as_float(ra[i].x & 0xffff0000u),
                                    __uint_as_float(ra[i].y << 16), __uint_as_float(ra[i].y & 0xffff0000u),
                                    __uint_as_float(ra[i].z << 16), __uint_as_float(ra[i].z & 0xffff0000u),
                                    __uint_as_float(ra[i].w << 16), __uint_as_float(ra[i].w & 0xffff0000u)};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if constexpr (BLDS) __hip_atomic_fetch_add(btab + k * NTt, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else bs[k] += v[k];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NXL; ++i)
            if (xdst[i] >= 0) *reinterpret_cast<u32x4*>(st + TA_BYTES + xdst[i]) = RELU ? relu16_bf16(rb[i]) : rb[i];
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) lstore1(&smem[buf * STAGE + s_ * SUB], ra[s_], rb[s_]);
    };
    // fragment addressing (ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-channel block)
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int g16 = lane >> 4, i16 = lane & 15;
    const int frow = (g16 >> 1) * 8 + (i16 >> 2), fcol2 = ((g16 & 1) * 16 + (i16 & 3) * 4) * 2;
    auto tr2 = [&](const char* lo_, const char* hi_) __attribute__((always_inline)) -> bf16x8 {
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lo_);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)hi_);
        s16x8 f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, f);
    };
    // footprint row of step pixel pk (before the tap offset): line s of a multi-line step starts at row s * fpr
    constexpr int NT_ = UP2 ? KW : 1;                            // x2 fold: the row depends on the tap's parity
    int xoffs[2][2][NT_];                                        // [k half][lo / hi 4-row block][tap]
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
#pragma unroll
            for (int t = 0; t < NT_; ++t) {
                const int pk = kb * 16 + frow + hl * 4;
                const int px_ = pk & (segw - 1);
                const int fr = (pk >> logsegw) * fpr + (UP2 ? ((px_ + t - pad) >> 1) + cpad : px_);
                xoffs[kb][hl][t] = TA_BYTES + wn * XHALF + fr * 64 + fcol2;
            }
    const int aoff = frow * RSA + fcol2 + (wm * 64) * 2;
    auto mma1 = [&](const char* st) __attribute__((always_inline)) {
        constexpr int NU = 2 * KW;                               // units: (k half, tap), 2 MFMAs each
        bf16x8 fa[2][2], fb[NU];
        auto ldA = [&](int kb) __attribute__((always_inline)) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const char* pz = st + aoff + kb * 16 * RSA + a * 64;
                fa[kb][a] = tr2(pz, pz + 4 * RSA);
            }
        };
        auto ldB = [&](int u) __attribute__((always_inline)) {
            const int kb = u / KW, t = u % KW;
            if constexpr (UP2) fb[u] = tr2(st + xoffs[kb][0][t], st + xoffs[kb][1][t]);
            else fb[u] = tr2(st + xoffs[kb][0][0] + t * 64, st + xoffs[kb][1][0] + t * 64);
        };
        ldA(0); ldB(0); ldB(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (u + 2 < NU) ldB(u + 2);
            if (u == KW - 2) ldA(1);
            const int kb = u / KW, t = u % KW;
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb][0], fb[u], acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb][1], fb[u], acc[1][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto mma = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) mma1(&smem[buf * STAGE + s_ * SUB]);
    };
    if (m_begin < m_end) {
        // One 8-wave workgroup per CU: waves w and w+4 share a SIMD and would run their load-issue / MFMA / LDS-store phases in
        // lockstep, leaving the matrix pipe idle while both issue memory operations.  The upper half of the waves therefore
        // works one stage further ahead in registers and does its LDS stores and global loads BETWEEN the two sub-steps, while
        // the lower half does them at the stage boundaries: 1.16 -> 1.29 PF/s (5 x 5), 0.86 -> 0.92 (3 x 3, 256-channel tile).
        // Measured alternatives: loads / stores compiled out 1.41 / 1.58 PF/s; both halves one stage ahead (stores at the top
        // of the stage) -1.3 %; LDS-DMA staging in a 3-stage ring (no registers, no ds_write; swizzled unpadded rows) 1.13 PF/s,
        // 1.22 staggered: a `buffer_load ... lds` costs more issue time beside MFMAs than a register load plus its ds_write.
        const bool late = NWAVE == 8 && NS == 2 && wave >= 4;
        gload(m_begin);
        lstore(0);
        if (late) gload(m_begin + 32 * NS);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        int buf = 0;
        for (int mk = m_begin; mk < m_end; mk += 32 * NS, buf ^= 1) {
            if (!late) {
                gload(mk + 32 * NS);              // past the slice: out-of-range offsets -> zeros
                mma(buf);
                lstore(buf ^ 1);
            } else {
                mma1(&smem[buf * STAGE]);
                lstore(buf ^ 1);
                gload(mk + 2 * 32 * NS);
                mma1(&smem[buf * STAGE + (NS - 1) * SUB]);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0): the upper half still has a prefetch in flight
    }
    if (do_bias) {
        float* red = reinterpret_cast<float*>(&smem[0]);
        if constexpr (!BLDS) {
#pragma unroll
            for (int k = 0; k < 8; ++k) red[tid * 8 + k] = bs[k];
        }
        __syncthreads();
        if (tid < CPRA) {
            for (int k = 0; k < 8; ++k) {
                float a = 0.f;
                for (int r = 0; r < RPPA; ++r) a += BLDS ? btab[k * NTt + r * CPRA] : red[(r * CPRA + tid) * 8 + k];
                const int co = co0 + tid * 8 + k;
                if (spread) p.wsb[((size_t)bz * gridDim.x + bx) * BMc + tid * 8 + k] = a;      // partial of (slice, workgroup)
                else if (co < p.Cout && a != 0.f) atomicAdd(p.dbias + co, a);
            }
        }
        __syncthreads();
    }
    // epilogue: 32 x 32 accumulator tiles through a per-wave LDS block
    float* ep = reinterpret_cast<float*>(&smem[0]) + wave * (32 * 32);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < KW; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ep[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[a][t][r];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            const int ci = ci0 + wn * 32 + (lane & 31);
            const int cob = co0 + wm * 64 + a * 32 + (lane >> 5);
            const int tap = irow * KW + t;
            if (p.ws) {
                // partial tile of this row slice: [slice][x-block][tap of the row][BMc][BNc], plain coalesced stores
                float* wt = p.ws + (((size_t)bz * gridDim.x + bx) * KW + t) * (size_t)(BMc * BNc) +
                            (size_t)(wm * 64 + a * 32) * BNc + wn * 32;
#pragma unroll 1
                for (int j = 0; j < 16; ++j)
                    wt[(size_t)(2 * j + (lane >> 5)) * BNc + (lane & 31)] = ep[(2 * j + (lane >> 5)) * 32 + (lane & 31)];
            } else if (ci < p.Cin_real) {
                float* dst = p.dw + ci * p.s_ci + tap * p.s_tap;
#pragma unroll 1
                for (int j = 0; j < 16; ++j) {
                    const int co = cob + 2 * j;
                    const float v = ep[(2 * j + (lane >> 5)) * 32 + (lane & 31)];
                    if (co < p.Cout && v != 0.f) atomicAdd(dst + co * p.s_co, v);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
}

// compile-time loop (the unit index of conv_wgrad_row4_kernel must be a constant expression: it selects the register class of an
// accumulator tile)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// v_mfma_f32_32x32x16_bf16 with the accumulator tile pinned to the AGPR half (V = false) or the VGPR half (V = true) of the unified
// register file.  hipcc selects ONE form for every MFMA builtin of a function -- all accumulators in AGPRs (at most 256 registers =
// 16 tiles) once the function may use more than 256 registers -- so a wave that owns 24 tiles spills 300 registers through the
// builtin (measured).  As inline assembly sixteen tiles sit in a0..a255 and eight in VGPRs.  The compiler still tracks the operands
// (s_waitcnt for the fragments' LDS reads); what it cannot see is the MFMA's result latency: the kernel never reads an accumulator
// inside the loop except as the C operand of a later MFMA on the same tile (>= 4 MFMAs later), and pads before its epilogue.
template <bool V>
__device__ __forceinline__ void mfma_pinned(f32x16& c, const bf16x8& a, const bf16x8& b) {
    if constexpr (V) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// ============================================================================ backward-weight, one filter row, ONE WAVE PER SIMD (round 6)
// The 8-wave row kernel above is held by its LDS traffic on 3-tap filters (pipe busy 0.47-0.52 with clock headroom,
// profiles/r05_pmc_wgrad.json): 229 staged bytes and 1.67 transposing reads per MFMA.  More MFMAs per staged byte need a larger
// channel tile, and the accumulators of a larger tile do not fit the 256 registers a wave has at two waves per SIMD.  This form
// runs FOUR waves per workgroup, one per SIMD, each with the whole 512-entry register file (accumulators spill over into AGPRs):
//   workgroup tile  (WCO * AW * 32) out-channels x (WCI * BW * 32) in-channels x KW taps, waves arranged WCO x WCI
//   wave tile       AW x BW x KW accumulators of 32 x 32  (3 taps, 256 x 128: 4 x 2 x 3 = 24 = 384 registers)
//   per 32-pixel sub-step and wave: 2 * KW * BW units of AW MFMAs; A fragments of a k half stay in registers for all its units,
//   B fragments are read two units ahead -> (AW + KW * BW) transposing reads per k half: 0.83 per MFMA, 146 staged bytes per MFMA.
// With one wave per SIMD nothing covers a wave that waits, so the loop is ONE software pipeline across sub-steps:
//   * staging registers are in flight all the time: chunk i of sub-step s+1 is written to the LDS ring during sub-step s and its
//     register is re-loaded at once with chunk i of sub-step s+2 (a load has a whole sub-step, ~1500 cycles, to land);
//   * the ring has three sub-step buffers; the barrier of sub-step s sits UB units into it (all stores of s+1 done), the units
//     behind it already read the first fragments of sub-step s+1, so the matrix pipe is fed across the barrier;
//   * every LDS / VMEM operation is pinned between two MFMA units (sched_barrier), ~1 staging operation per unit.
// Same LDS images, loaders, slice workspace, bias shares and reduce kernels as conv_wgrad_row_kernel (no x2-upsampled input).
#ifndef DVD_EXP_ROW4                  // timing-only experiment builds (wrong results): 1 = no global loads, 2 = no LDS stores, 4 = no barrier
#define DVD_EXP_ROW4 0
#endif
// FOLD (round 6): the weight gradient of a 3 x 3 convolution over a nearest-x2 UPSAMPLED input (GResBlock.py:57-58), worked on the
// INPUT grid.  Output pixel (2i + a, 2j + b) reads input pixel (i + ((a + ky) >> 1), j + ((b + kx) >> 1)) for tap (ky, kx) in -1..1,
// so   dw[ky][kx] = sum over phases (a, b) of  G[a][sy][b][sx],  sy = (a + ky) >> 1, sx = (b + kx) >> 1,
//      G[a][sy][b][sx][co][ci] = sum over input pixels q of  dy[2q + (a, b)][co] * x[q + (sy, sx)][ci]
// -- an ordinary 3 x 3 weight gradient on the input grid whose "output channels" are the four phases of dy side by side
// (co' = (2a + b) * Cout + co: rows of dy two apart, found by address arithmetic, nothing is re-laid-out), with only the filter rows
// sy in {a - 1, a} of phase a: 2 rows x 3 taps x 4 Cout x M / 4 = 2 / 3 of the products of the direct form (the third tap of a row is
// unused for either b: computed and dropped), on the one-wave-per-SIMD 128 x 128 tile instead of the 8-wave x2 tiles (0.43-0.9 PF/s;
// the 64-channel one, 12.5 M x 128 -> 64 on 64 x 64 frames, took 4.3 ms of a step).  Partial tiles always go through the slice
// workspace; wgrad_fold_reduce_kernel forms dw from them.  p.H / p.W / p.M are the INPUT grid's, p.Cout = p.Cy = 4 * cout_f.
template <int KW, int AW, int BW, int WCO, int WCI, bool RELU, int DEPTH = 2, bool FOLD = false>
__global__ __launch_bounds__(256) void conv_wgrad_row4_kernel(WgK p) {
    // bias column sums: eight registers.  The pinned tile with two staging sets has none left (an LDS table updated with ds_add_f32 ran
    // such launches 2x longer: that form only stays correct, the planner does not use it): launches with a bias gradient take the pinned
    // tile with ONE staging set (DEPTH 1, 222 + 256 registers) -- 786 k x 256 -> 256 incl. reduce / bias kernels 1050 -> 950 us against
    // the 128 x 128 tile they took before (depth 2 is worth 6.5 % on that tile, the larger tile 18 %)
    constexpr bool BLDS = KW == 3 && AW * BW * KW > 16 && DEPTH >= 2;        // (5 taps: 20 tiles, four of them in VGPRs -- registers to spare; DEPTH 1: one staging set less)
    static_assert(WCO * WCI == 4, "four waves, one per SIMD");
    constexpr int NTt = 256, BMc = WCO * AW * 32, NHB = WCI * BW, BNc = NHB * 32;
    constexpr int RSA = BMc * 2 + 64;
    constexpr int TA_BYTES = 32 * RSA;
    constexpr int XROWS = KW == 5 ? 64 : 48, XHALF = XROWS * 64, TB_BYTES = NHB * XHALF;
    constexpr int SUB = TA_BYTES + TB_BYTES;
    constexpr int EPIB = 4 * 32 * 32 * 4, REDB = NTt * 8 * 4;
    constexpr int LDSB0 = 3 * SUB > EPIB ? (3 * SUB > REDB ? 3 * SUB : REDB) : (EPIB > REDB ? EPIB : REDB);
    constexpr int LDSB = LDSB0 + (BLDS ? REDB : 0);              // + the bias column sums [8][256] (pinned tile)
    __shared__ __attribute__((aligned(16))) char smem[LDSB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WCI, wn = wave % WCI;
    int bx = blockIdx.x, bz = blockIdx.z;
    if (p.xcd_remap) {
        const int gx = gridDim.x, total = gx * gridDim.z, lin = bx + gx * bz;
        const int xcd = lin & 7, qd = total >> 3, rr = total & 7;
        const int lp = (xcd < rr ? xcd * (qd + 1) : rr * (qd + 1) + (xcd - rr) * qd) + (lin >> 3);
        bz = lp / gx; bx = lp - bz * gx;
    }
    const int tiles = p.tiles_co * p.tiles_ci;
    const int irow = bx / tiles;
    const int rem = bx - irow * tiles;
    const int tco = rem / p.tiles_ci, tci = rem - tco * p.tiles_ci;
    const int co0 = tco * BMc, ci0 = tci * BNc;
    constexpr int pad = KW >> 1;
    static_assert(!FOLD || (KW == 3 && BMc == 128), "the folded form runs on the 3-tap 128-channel tile");
    // FOLD: irow = 0 / 1 = filter row a - 1 / a of the tile's line phase a (a 128-channel tile lies inside one a: Cout is a multiple of 64)
    const int it = FOLD ? 0 : irow / KW, iy = FOLD ? irow : irow - it * KW;
    const int dyl = FOLD ? ((co0 / p.cout_f) >> 1) - 1 + irow : iy - pad, dtl = FOLD ? 0 : it - (p.kt >> 1);
    const int m_begin = bz * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);

    f32x16 acc[AW][BW][KW];
    {
        const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < AW; ++a)
#pragma unroll
            for (int b = 0; b < BW; ++b)
#pragma unroll
                for (int t = 0; t < KW; ++t) acc[a][b][t] = zacc;
    }
    const int xbase_row = max(0, m_begin - p.maxshift);
    // FOLD: rows of dy are on the output grid: input pixel mk = (line L, column x0) -> output row 4 * mk - 2 * x0 (+ the phase's a * 2W + b)
    const int ybase_row = FOLD ? 4 * m_begin - 2 * (p.W > 32 ? m_begin & (p.W - 1) : 0) : m_begin;
    const size_t xbase_b = (size_t)xbase_row * p.ldx * 2, ybase_b = (size_t)ybase_row * p.ldy * 2;
    const size_t xleft = p.x_bytes - xbase_b, yleft = p.dy_bytes > ybase_b ? p.dy_bytes - ybase_b : 0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + xbase_b), 0, xleft > 0xfffffffeull ? 0xfffffffeu : (unsigned)xleft, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.dy + ybase_b), 0, yleft > 0xfffffffeull ? 0xfffffffeu : (unsigned)yleft, 0x00020000);
    const int segw = min(p.W, 32), logsegw = min(p.logW, 5);
    const int fpr = segw + KW - 1, frows = (32 >> logsegw) * fpr;
    // staging chunks of one sub-step: NPA of dy, NXL of the x footprint (16 bytes each per thread)
    constexpr int CPRA = BMc / 8, RPPA = NTt / CPRA, NPA = 32 / RPPA;
    const int rra = tid / CPRA, cka = tid % CPRA;
    const int cy = co0 + cka * 8;
    const bool cyv = cy < p.Cy;
    constexpr int CPX = NHB * 4;
    constexpr int NXL = (XROWS * CPX + NTt - 1) / NTt;
    constexpr int NOPS = NPA + NXL;
    // Addresses.  A sub-step starts at a multiple of 32 pixels: it covers whole lines (W <= 32: 32 / W of them, starting at a line that
    // is a multiple of that; two whole frames when the frame is 4 x 4) or one 32-pixel segment of a line, so the input row a thread
    // fetches for chunk i is   mk + (a per-thread constant)   -- the first row of the sub-step plus the chunk's position in the
    // footprint -- and only its validity depends on where the sub-step sits in the frame: (y + xya) in [0, H) and (x0 + xxa) in [0, W).
    // (The 8-wave kernel rebuilds line / column / frame of every chunk from mk: ~14 VALU per chunk; here 6.)  Slices and M are
    // multiples of 32 rows, so a sub-step is inside the slice or outside it as a whole (`tok`, wave-uniform).
    constexpr bool XFULL = NXL * NTt == XROWS * CPX;             // every thread owns NXL chunks of the footprint image
    // At most ONE of the two checks is per thread: W > 32 -> the step is one line (the line check is wave-uniform, the column check
    // per thread); W <= 32 -> x0 = 0 (the column check is a per-thread constant, folded into xv; the line check per thread).
    int xa[NXL];
    static_assert(NTt % CPX == 0, "chunk i of a thread sits NTt / CPX footprint rows below chunk i - 1");
    const int xdst0 = ((tid % CPX) >> 2) * XHALF + (tid / CPX) * 64 + ((tid % CPX) & 3) * 16;      // chunk i: + i * (NTt / CPX) * 64 (an immediate)
    unsigned xld[NXL];
    bool xv[NXL];
    const bool twofr = (32 >> logsegw) > p.H;                    // 4 x 4 frames: the step's eight lines span two frames (y = x0 = 0)
    const bool wide = p.W > 32;
    const unsigned ldx2 = (unsigned)p.ldx * 2, ldy2 = (unsigned)p.ldy * 2;
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
        const int q = tid + i * NTt, row = q / CPX, c8 = q % CPX;
        const int seg = row / fpr, j = row - seg * fpr;
        const int cx = ci0 + c8 * 8;
        const int fo = twofr ? seg >> p.logH : 0, yl = twofr ? seg & (p.H - 1) : seg;
        const int xya = yl + dyl, xxa = j - pad;
        xa[i] = wide ? xxa : xya;
        xv[i] = row < frows && cx < p.C && (wide || (unsigned)xxa < (unsigned)p.W);
        const int delta = (fo << (p.logW + p.logH)) + (xya << p.logW) + xxa;
        xld[i] = (unsigned)delta * ldx2 + (unsigned)cx * 2;      // (wraps for negative deltas; the sum with the uniform part is exact)
    }
    unsigned yld = (unsigned)rra * ldy2 + (unsigned)cy * 2;
    unsigned ystep = (unsigned)RPPA * ldy2;                      // from staging chunk op to op + 1 of dy (RPPA rows further on)
    if constexpr (FOLD) {
        // row r = rra + 16 * op of the sub-step = input pixel (line r >> logsegw, column r & (segw - 1)) of its segment(s); phase (a, b)
        // of the thread's channel chunk: output row  2 * line * 2W + 2 * column + a * 2W + b  past the sub-step's first
        static_assert(!FOLD || RPPA == 16, "16 rows per staging pass");
        const int ph = cy / p.cout_f, c = cy - ph * p.cout_f, W2 = 2 * p.W;
        yld = (unsigned)(((rra >> logsegw) * 2 + (ph >> 1)) * W2 + 2 * (rra & (segw - 1)) + (ph & 1)) * ldy2 + (unsigned)c * 2;
        ystep = (unsigned)(segw == 32 ? 32 : (16 >> logsegw) * 2 * W2) * ldy2;
    }
    const int dt_rows = p.kt > 1 ? dtl << (p.logW + p.logH) : 0;
    u32x4 R[DEPTH][NOPS];      // staging registers: set d holds the chunks of every DEPTH-th sub-step, loaded DEPTH sub-steps before they are stored
    struct Geo { int sval; unsigned xoff, yoff; bool tok, ytok; };
    const unsigned slim = wide ? p.W : p.H;
    // geometry of the sub-step that starts at pixel mk, in four pieces (wave-uniform scalar work: inside the main loop each piece
    // sits in a slot of its own -- as one block it was a 25-instruction bubble in front of one MFMA)
    auto geo_part = [&](Geo& g, int mk, int part) __attribute__((always_inline)) {
        if (part == 0) {
            const int x0 = mk & (p.W - 1), y = twofr ? 0 : (mk >> p.logW) & (p.H - 1);
            g.sval = wide ? x0 : y;
            g.ytok = mk < m_end;
            g.tok = g.ytok && (!wide || (unsigned)(y + dyl) < (unsigned)p.H);
        } else if (part == 1) {
            if (p.kt > 1) {
                const int tt = (mk >> (p.logW + p.logH)) % p.T + dtl;
                g.tok = g.tok && (unsigned)tt < (unsigned)p.T;
            }
        } else if (part == 2) {
            g.xoff = (unsigned)(mk + dt_rows - xbase_row) * ldx2;
        } else {
            if constexpr (FOLD) g.yoff = (unsigned)(4 * mk - 2 * (wide ? mk & (p.W - 1) : 0) - ybase_row) * ldy2;
            else g.yoff = (unsigned)(mk - m_begin) * ldy2;
        }
    };
    auto geo = [&](int mk) __attribute__((always_inline)) -> Geo {
        Geo g;
#pragma unroll
        for (int part = 0; part < 4; ++part) geo_part(g, mk, part);
        return g;
    };
    auto load_op = [&](int op, const Geo& g, u32x4 (&R)[NOPS]) __attribute__((always_inline)) {
        if (op < NPA) {
            const unsigned off = (cyv && g.ytok) ? yld + g.yoff + (unsigned)op * ystep : 0xffffffffu;
            R[op] = __builtin_amdgcn_raw_buffer_load_b128(rdy, off, 0, 0);
        } else {
            const int i = op - NPA;
            const bool ok = xv[i] && g.tok && (unsigned)(g.sval + xa[i]) < slim;
            const unsigned off = ok ? xld[i] + g.xoff : 0xffffffffu;
            R[op] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
        }
    };
    // bias gradient shares: as in conv_wgrad_row_kernel (sub-step n belongs to share n % nshare)
    const bool spread = p.ws != nullptr;
    const bool do_bias = p.dbias != nullptr && dtl == 0 && (spread || (dyl == 0 && tci == 0));
    const int nshare = spread ? (FOLD ? 2 : KW) * p.tiles_ci : 1, myshare = spread ? iy * p.tiles_ci + tci : 0;
    int bphase = 0;
    float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float* btab = reinterpret_cast<float*>(&smem[BLDS ? LDSB0 : 0]) + tid;       // BLDS: [8][256] floats, slot k of this thread at btab[k * NTt]
    if (BLDS && do_bias) {
#pragma unroll
        for (int k = 0; k < 8; ++k) btab[k * NTt] = 0.f;
    }
    // 16-byte LDS stores as two 8-byte stores in two slots: 5 taps -1.5 % (3.1 M x 256 -> 256: 7540 -> 7428 us), 3 taps no change
    constexpr bool ST64 = KW == 5;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    // half: -1 = the whole chunk, 0 / 1 = its low / high 8 bytes (bias sums and ReLU with half 0 / whole)
    auto store_op = [&](int op, char* st, bool mine, const u32x4 (&R)[NOPS], int half = -1) __attribute__((always_inline)) {
        if (half >= 0) {
            const u32x4 v = (RELU && op >= NPA) ? relu16_bf16(R[op]) : R[op];
            const u32x2 h2 = half ? u32x2{v.z, v.w} : u32x2{v.x, v.y};
            char* dst = op < NPA ? st + (rra + op * RPPA) * RSA + cka * 16 : st + TA_BYTES + xdst0 + (op - NPA) * (NTt / CPX) * 64;
            if (op < NPA || XFULL || tid + (op - NPA) * NTt < XROWS * CPX) *reinterpret_cast<u32x2*>(dst + half * 8) = h2;
            if (half == 1 || op >= NPA || !mine) return;
            const float v8[8] = {__uint_as_float(R[op].x << 16), __uint_as_float(R[op].x & 0xffff0000u),
                                 __uint_as_float(R[op].y << 16), __uint_as_float(R[op].y & 0xffff0000u),
                                 __uint_as_float(R[op].z << 16), __uint_as_float(R[op].z & 0xffff0000u),
                                 __uint_as_float(R[op].w << 16), __uint_as_float(R[op].w & 0xffff0000u)};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if constexpr (BLDS) __hip_atomic_fetch_add(btab + k * NTt, v8[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else bs[k] += v8[k];
            }
            return;
        }
        if (op < NPA) {
            *reinterpret_cast<u32x4*>(st + (rra + op * RPPA) * RSA + cka * 16) = R[op];
            if (mine) {                                           // (the thread's own slots: ds_add_f32, same order of additions as a register sum)
                const float v[8] = {__uint_as_float(R[op].x << 16), __uint_as_float(R[op].x & 0xffff0000u),
                                    __uint_as_float(R[op].y << 16), __uint_as_float(R[op].y & 0xffff0000u),
                                    __uint_as_float(R[op].z << 16), __uint_as_float(R[op].z & 0xffff0000u),
                                    __uint_as_float(R[op].w << 16), __uint_as_float(R[op].w & 0xffff0000u)};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if constexpr (BLDS) __hip_atomic_fetch_add(btab + k * NTt, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else bs[k] += v[k];
                }
            }
        } else {
            const int i = op - NPA;
            if (XFULL || tid + i * NTt < XROWS * CPX)
                *reinterpret_cast<u32x4*>(st + TA_BYTES + xdst0 + i * (NTt / CPX) * 64) = RELU ? relu16_bf16(R[op]) : R[op];
        }
    };
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int g16 = lane >> 4, i16 = lane & 15;
    const int frow = (g16 >> 1) * 8 + (i16 >> 2), fcol2 = ((g16 & 1) * 16 + (i16 & 3) * 4) * 2;
    auto tr2 = [&](const char* lo_, const char* hi_) __attribute__((always_inline)) -> bf16x8 {
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lo_);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)hi_);
        s16x8 f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, f);
    };
    // B fragment addresses: the lane's pixel of k half kb / 4-row block hl is pk = kb * 16 + frow + hl * 4; its footprint row
    // (pk >> logsegw) * fpr + (pk & (segw - 1)) is  row(frow) + kb * KBROWS + hl * 4  with a wave-uniform KBROWS (frow + 4 < 16 stays
    // inside the lane's line segment for every segment width >= 8; 4-wide segments: frow and frow + 4 are one line apart)
    const int fr0 = (frow >> logsegw) * fpr + (frow & (segw - 1));
    const int hlstep = (segw >= 8 ? 4 : fpr) * 64;                                            // bytes from the lo to the hi 4-row block
    const int kbstep = ((16 >> logsegw) * fpr + (16 & (segw - 1))) * 64;                      // bytes from k half 0 to k half 1
    const int xoff0 = TA_BYTES + wn * BW * XHALF + fr0 * 64 + fcol2;
    const int aoff = frow * RSA + fcol2 + (wm * AW * 32) * 2;
    constexpr int UPK = KW * BW, NU = 2 * UPK;                   // units per k half / per sub-step; unit = (k half, tap, in-channel block)
    constexpr bool PIN = AW * BW * KW > 16;                      // more accumulator tiles than the AGPR half holds: see mfma_pinned
    constexpr int UB = NU - 2;                                   // the sub-step's barrier sits in front of unit UB (two units of fragments read ahead)
    // A fragments: ONE register set.  fa[a] is re-read for the next k half in the slot right after its last MFMA of this one
    // (AW - 1 slots, ~100 cycles, before its first use there) -- a second set would cost the 16 registers the second staging set needs.
    // B fragments: a ring read two units ahead; its length must divide the units of a sub-step (the ring index runs on across
    // sub-steps): 3 for 12 / 6 units (3 taps), 5 for 10 units (5 taps, one in-channel block per wave)
    constexpr int RING = NU % 3 == 0 ? 3 : 5;
    static_assert(NU % RING == 0, "B fragment ring");
    bf16x8 fa[AW], fb[RING];
    auto ldA = [&](const char* st, int kb, int a) __attribute__((always_inline)) {
        const char* pz = st + aoff + kb * 16 * RSA + a * 64;
        fa[a] = tr2(pz, pz + 4 * RSA);
    };
    auto ldB = [&](const char* st, int u, int slot) __attribute__((always_inline)) {
        const int kb = u / UPK, t = (u % UPK) / BW, b = u % BW;
        const char* pz = st + xoff0 + kb * kbstep + t * 64 + b * XHALF;
        fb[slot] = tr2(pz, pz + hlstep);
    };
    if (m_begin < m_end) {
        // prologue: sub-step 0 into ring slot 0, sub-steps 1 .. DEPTH into the register sets
        {
            const Geo g0 = geo(m_begin);
#pragma unroll
            for (int op = 0; op < NOPS; ++op) load_op(op, g0, R[0]);
            const bool mine = do_bias && bphase == myshare;
            if (++bphase == nshare) bphase = 0;
#pragma unroll
            for (int op = 0; op < NOPS; ++op) store_op(op, &smem[0], mine, R[0]);
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const Geo g1 = geo(m_begin + 32 * (d + 1));
#pragma unroll
                for (int op = 0; op < NOPS; ++op) load_op(op, g1, R[d]);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int a = 0; a < AW - 1; ++a) ldA(&smem[0], 0, a);       // (fa[AW - 1]: slot (0, 0) of every sub-step)
            ldB(&smem[0], 0, 0);
            ldB(&smem[0], 1, 1);
        }
        char* rd = &smem[0];
        char* nx = &smem[SUB];
        char* fr_ = &smem[2 * SUB];
        // One wave per SIMD: the matrix pipe only stays busy if the wave's other instructions sit BETWEEN its MFMAs (32 cycles each,
        // ~5 issue slots), so every MFMA is a slot of its own (sched_barrier after each) with a fixed share of the sub-step's LDS reads,
        // LDS stores, address arithmetic and global loads in front of it.  (First form: four MFMAs back to back per unit, then the
        // unit's ~20 other instructions with the pipe idle -- pipe busy 0.56.)  Slot a of unit u:
        //   a = 0: B fragment of unit u + 2;  a = 1: LDS stores of the unit's staging chunks;  a = 2 (and 3 behind the barrier): A
        //   fragments;  a = AW - 1: address + global load of the unit's staging chunks, and once per sub-step the next geometry.
        static_assert(AW == 4, "slot plan written for four A fragments per wave");
        Geo gl = geo(m_begin + 32 * (DEPTH + 1));
        for (int mk0 = m_begin; mk0 < m_end; mk0 += 32 * DEPTH) {
          static_for<0, DEPTH>([&](auto D_) __attribute__((always_inline)) {       // (an odd tail runs one sub-step of zeros)
            constexpr int dset = decltype(D_)::value;
            const int mk = mk0 + 32 * dset;
            const bool mine = do_bias && bphase == myshare;              // share of the sub-step being stored (mk + 32)
            if (++bphase == nshare) bphase = 0;
            Geo gn = gl;
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, NU * AW>([&](auto S_) __attribute__((always_inline)) {
                constexpr int s = decltype(S_)::value, u = s / AW, a = s % AW;
                constexpr int kb = u / UPK, t = (u % UPK) / BW, b = u % BW;
                if constexpr (u == UB && a == 0) {
                    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): this wave's stores of sub-step s+1 are in the LDS
                    if (!(DVD_EXP_ROW4 & 4)) __builtin_amdgcn_s_barrier();
                }
                if constexpr (a == 0) {                                  // B fragment two units ahead (the last two reach into the next sub-step)
                    if (u + 2 < NU) ldB(rd, u + 2, (u + 2) % RING);
                    else ldB(nx, u + 2 - NU, (u + 2) % RING);
                }
                if constexpr (a == 1 || (ST64 && a == 2)) {
#pragma unroll
                    for (int op = 0; op < NOPS; ++op)
                        if ((op * UB) / NOPS == u && !(DVD_EXP_ROW4 & 2)) {
                            if constexpr (ST64) store_op(op, nx, mine, R[dset], a - 1);
                            else store_op(op, nx, mine, R[dset]);
                        }
                }
                // A fragment a - 1 of the next k half, right behind its last MFMA of this one (the next sub-step's come from `nx`, behind the barrier)
                if constexpr (u == UPK - 1 && a >= 1) ldA(rd, 1, a - 1);
                if constexpr (u == UPK && a == 0) ldA(rd, 1, AW - 1);
                if constexpr (u == NU - 1 && a >= 1) ldA(nx, 0, a - 1);
                if constexpr (u == 0 && a == 0) ldA(rd, 0, AW - 1);
                if constexpr (a == AW - 1) {
#pragma unroll
                    for (int op = 0; op < NOPS; ++op)
                        if ((op * UB) / NOPS == u && !(DVD_EXP_ROW4 & 1)) load_op(op, gl, R[dset]);
                }
                // geometry of the next sub-step's loads, a piece per slot (the slots behind the barrier carry little else)
                if constexpr (u == UB - 1 && a == AW - 1) geo_part(gn, mk + 32 * (DEPTH + 2), 0);
                if constexpr (u == UB && a >= 1) geo_part(gn, mk + 32 * (DEPTH + 2), a);
                if constexpr (PIN) mfma_pinned<((a * BW + b) * KW + t >= 16)>(acc[a][b][t], fa[a], fb[u % RING]);
                else acc[a][b][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[u % RING], acc[a][b][t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            gl = gn;
            char* t_ = rd; rd = nx; nx = fr_; fr_ = t_;
          });
        }
        __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0): two sub-steps of prefetch are still in flight
        if constexpr (PIN) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results (inline assembly: latency unknown to hipcc)
        __syncthreads();
    }
    if (do_bias) {
        if constexpr (!BLDS) {
#pragma unroll
            for (int k = 0; k < 8; ++k) btab[k * NTt] = bs[k];
        }
        __syncthreads();
        if (tid < CPRA) {
            for (int k = 0; k < 8; ++k) {
                float a = 0.f;
                for (int r = 0; r < RPPA; ++r) a += btab[k * NTt + r * CPRA];
                const int co = co0 + tid * 8 + k;
                if (spread) p.wsb[((size_t)bz * gridDim.x + bx) * BMc + tid * 8 + k] = a;
                else if (co < p.Cout && a != 0.f) atomicAdd(p.dbias + co, a);
            }
        }
        __syncthreads();
    }
    float* ep = reinterpret_cast<float*>(&smem[0]) + wave * (32 * 32);
#pragma unroll
    for (int a = 0; a < AW; ++a)
#pragma unroll
        for (int b = 0; b < BW; ++b)
#pragma unroll
            for (int t = 0; t < KW; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ep[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[a][b][t][r];
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                const int ci = ci0 + (wn * BW + b) * 32 + (lane & 31);
                const int cob = co0 + (wm * AW + a) * 32 + (lane >> 5);
                const int tap = irow * KW + t;
                if (p.ws) {
                    float* wt = p.ws + (((size_t)bz * gridDim.x + bx) * KW + t) * (size_t)(BMc * BNc) +
                                (size_t)((wm * AW + a) * 32) * BNc + (wn * BW + b) * 32;
#pragma unroll 1
                    for (int j = 0; j < 16; ++j)
                        wt[(size_t)(2 * j + (lane >> 5)) * BNc + (lane & 31)] = ep[(2 * j + (lane >> 5)) * 32 + (lane & 31)];
                } else if (ci < p.Cin_real) {
                    float* dst = p.dw + ci * p.s_ci + tap * p.s_tap;
#pragma unroll 1
                    for (int j = 0; j < 16; ++j) {
                        const int co = cob + 2 * j;
                        const float v = ep[(2 * j + (lane >> 5)) * 32 + (lane & 31)];
                        if (co < p.Cout && v != 0.f) atomicAdd(dst + co * p.s_co, v);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
}

// reduction of the row kernel's partial tiles: dw[co][ci][iy*KW + t] += sum over slices
struct WgRowRedK { const float* ws; float* dw; int nslice, gx, tiles_co, tiles_ci, BMc, BNc, KW, Cout, Cin_real; long long s_co, s_ci, s_tap; int overwrite; };
// sum over the slices of four consecutive partial-tile elements (16-byte loads, four slices requested before the first addition;
// slice order kept): the reduce kernels stream the whole workspace once and were running at 2.5 TB/s with scalar loads
__device__ __forceinline__ f32x4 slice_sum4(const float* ws, int nslice, size_t stride, size_t off) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 4 <= nslice; z += 4) {
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4*>(ws + (size_t)(z + j) * stride + off);
#pragma unroll
        for (int j = 0; j < 4; ++j) a += v[j];
    }
    for (; z < nslice; ++z) a += *reinterpret_cast<const f32x4*>(ws + (size_t)z * stride + off);
    return a;
}

__global__ void wgrad_row_reduce_kernel(WgRowRedK p) {
    const int tile_elems = p.KW * p.BMc * p.BNc;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;      // 4 consecutive in-channels per thread
    if (i >= (long long)p.gx * tile_elems) return;
    const int bx = (int)(i / tile_elems), e = (int)(i - (long long)bx * tile_elems);
    const int t = e / (p.BMc * p.BNc), e2 = e - t * (p.BMc * p.BNc);
    const int r = e2 / p.BNc, c = e2 - r * p.BNc;
    const int tiles = p.tiles_co * p.tiles_ci;
    const int iy = bx / tiles, rem = bx - iy * tiles;
    const int tco = rem / p.tiles_ci, tci = rem - tco * p.tiles_ci;
    const int co = tco * p.BMc + r, ci = tci * p.BNc + c;
    if (co >= p.Cout || ci >= p.Cin_real) return;
    const f32x4 a = slice_sum4(p.ws, p.nslice, (size_t)p.gx * tile_elems, (size_t)bx * tile_elems + e);
    float* d = p.dw + co * p.s_co + ci * p.s_ci + (iy * p.KW + t) * p.s_tap;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (ci + j < p.Cin_real) d[j * p.s_ci] = p.overwrite ? a[j] : d[j * p.s_ci] + a[j];
}

// bias gradient of the row kernel's workspace path: dbias[co] += sum over (slice, filter row of the centre time slab, ci tile) of the
// partial column sums, in a fixed order (block = 32 channels x 8 partial lanes)
struct WgBiasRedK { const float* wsb; float* dbias; int nslice, gx, tiles_co, tiles_ci, KW, kt, BMc, Cout; };
__global__ __launch_bounds__(1024) void wgrad_row_bias_reduce_kernel(WgBiasRedK p) {
    // block = 32 channels x 32 partial lanes; a lane walks its items eight loads at a time (a few thousand partials per channel on
    // the large shapes: as a serial chain on 8 lanes this kernel took ~250 us)
    __shared__ float red[32][33];
    const int gpt = p.BMc / 32;
    const int tco = blockIdx.x / gpt, cg = blockIdx.x - tco * gpt;
    const int c = threadIdx.x & 31, l = threadIdx.x >> 5;
    const int tiles = p.tiles_co * p.tiles_ci, per = p.KW * p.tiles_ci, nitem = p.nslice * per;
    const int irow0 = (p.kt >> 1) * p.KW;
    auto item = [&](int i) __attribute__((always_inline)) -> float {
        if (i >= nitem) return 0.f;
        const int bz = i / per, r = i - bz * per;
        const int iy = r / p.tiles_ci, tci = r - iy * p.tiles_ci;
        const int bx = (irow0 + iy) * tiles + tco * p.tiles_ci + tci;
        return p.wsb[((size_t)bz * p.gx + bx) * p.BMc + cg * 32 + c];
    };
    float a = 0.f;
    for (int i = l; i < nitem; i += 32 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = item(i + 32 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    red[l][c] = a;
    __syncthreads();
    if (l == 0) {
        for (int j = 1; j < 32; ++j) a += red[j][c];
        const int co = tco * p.BMc + cg * 32 + c;
        if (co < p.Cout) p.dbias[co] += a;
    }
}

// bias gradient of the one-tap kernel's workspace path: dbias[co] += sum over the row slices.  Block = 64 channels x 16 slice lanes:
// lane q sums slices q, q+16, ... in order, the sixteen lane sums are added in lane order (one thread per channel walking up to
// 768 slices took 48 us per call, 1.9 ms per step).
__global__ __launch_bounds__(1024) void wgrad_bias_reduce_kernel(const float* wsb, float* dbias, int nslice, int tiles_co, int BMc, int Cout) {
    __shared__ float red[16][64];
    const int col = threadIdx.x & 63, q = threadIdx.x >> 6, co = blockIdx.x * 64 + col;
    float a = 0.f;
    if (co < Cout) {
        const int tco = co / BMc, c = co - tco * BMc;
#pragma unroll 4
        for (int z = q; z < nslice; z += 16) a += wsb[((size_t)z * tiles_co + tco) * BMc + c];
    }
    red[q][col] = a;
    __syncthreads();
    if (q == 0 && co < Cout) {
        float t = red[0][col];
#pragma unroll
        for (int i = 1; i < 16; ++i) t += red[i][col];
        dbias[co] += t;
    }
}

// Second phase of the workspace path: dw[co][ci][tap] += sum over row slices of the partial tiles.
struct WgRedK { const float* ws; float* dw; int nslice, gx, gx_per_tap, tiles_ci, BMc, BNc, Cout, Cin_real; long long s_co, s_ci, s_tap; int overwrite; };
__global__ void wgrad_reduce_kernel(WgRedK p) {
    const int tile_elems = p.BMc * p.BNc;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= (long long)p.gx * tile_elems) return;
    const int bx = (int)(i / tile_elems), e = (int)(i - (long long)bx * tile_elems);
    const int r = e / p.BNc, c = e - r * p.BNc;
    // decode bx exactly like the kernel: tap = bx / (tiles_co*tiles_ci), rem -> (tco, tci)
    const int per_tap = p.gx_per_tap;
    const int tap = bx / per_tap, rem = bx - tap * per_tap;
    const int tco = rem / p.tiles_ci, tci = rem - tco * p.tiles_ci;
    const int co = tco * p.BMc + r, ci = tci * p.BNc + c;
    if (co >= p.Cout || ci >= p.Cin_real) return;
    const f32x4 a = slice_sum4(p.ws, p.nslice, (size_t)p.gx * tile_elems, (size_t)bx * tile_elems + e);
    float* d = p.dw + co * p.s_co + ci * p.s_ci + tap * p.s_tap;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (ci + j < p.Cin_real) d[j * p.s_ci] = p.overwrite ? a[j] : d[j * p.s_ci] + a[j];
}

// Reduction of the folded form (conv_wgrad_row4_kernel<.., FOLD>): dw[co][ci][ky][kx] (+)= sum over the four phases (a, b), in that
// order, of the slice sums of G[a][sy][b][sx] -- tile (filter row irow = sy - a + 1, 128-channel block of co' = (2a + b) * Cout + co,
// ci block), tap sx + 1 of the row.  Fixed order: bit-reproducible.
struct WgFoldRedK { const float* ws; float* dw; int nslice, gx, tiles_ci, Cout, Cin_real; long long s_co, s_ci, s_tap; int overwrite; };
__global__ void wgrad_fold_reduce_kernel(WgFoldRedK p) {
    const int ci4 = (p.Cin_real + 3) / 4;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)9 * p.Cout * ci4) return;
    const int cq = (int)(i % ci4), rest = (int)(i / ci4);
    const int co = rest % p.Cout, tap = rest / p.Cout;
    const int ky = tap / 3 - 1, kx = tap % 3 - 1, ci = cq * 4;
    const int tiles = (4 * p.Cout / 128) * p.tiles_ci;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
        const int a = ph >> 1, b = ph & 1;
        const int sy = (a + ky) >> 1, sx = (b + kx) >> 1;           // (arithmetic shifts: -1 >> 1 = -1)
        const int irow = sy - a + 1, t = sx + 1;
        const int cop = ph * p.Cout + co;
        const int bx = irow * tiles + (cop >> 7) * p.tiles_ci + (ci >> 7);
        const size_t off = ((size_t)bx * 3 + t) * (128 * 128) + (size_t)(cop & 127) * 128 + (ci & 127);
        acc += slice_sum4(p.ws, p.nslice, (size_t)p.gx * 3 * (128 * 128), off);
    }
    float* d = p.dw + co * p.s_co + ci * p.s_ci + tap * p.s_tap;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (ci + j < p.Cin_real) d[j * p.s_ci] = p.overwrite ? acc[j] : d[j * p.s_ci] + acc[j];
}
// its bias gradient: dbias[co] += sum over (phase, slice, filter row, ci tile) of the workgroups' partial column sums, fixed order
// (block = 32 channels x 32 partial lanes, as wgrad_row_bias_reduce_kernel)
struct WgFoldBiasK { const float* wsb; float* dbias; int nslice, gx, tiles_ci, Cout; };
__global__ __launch_bounds__(1024) void wgrad_fold_bias_reduce_kernel(WgFoldBiasK p) {
    __shared__ float red[32][33];
    const int c = threadIdx.x & 31, l = threadIdx.x >> 5, co = blockIdx.x * 32 + c;
    const int tiles = (4 * p.Cout / 128) * p.tiles_ci, per = 2 * p.tiles_ci, nitem = 4 * p.nslice * per;
    auto item = [&](int i) __attribute__((always_inline)) -> float {
        if (i >= nitem || co >= p.Cout) return 0.f;
        const int ph = i / (p.nslice * per), r0 = i - ph * (p.nslice * per);
        const int bz = r0 / per, r = r0 - bz * per;
        const int irow = r / p.tiles_ci, tci = r - irow * p.tiles_ci;
        const int cop = ph * p.Cout + co;
        const int bx = irow * tiles + (cop >> 7) * p.tiles_ci + tci;
        return p.wsb[((size_t)bz * p.gx + bx) * 128 + (cop & 127)];
    };
    float a = 0.f;
    for (int i = l; i < nitem; i += 32 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = item(i + 32 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    red[l][c] = a;
    __syncthreads();
    if (l == 0) {
        for (int j = 1; j < 32; ++j) a += red[j][c];
        if (co < p.Cout) p.dbias[co] += a;
    }
}

}  // namespace

// Validates a weight-gradient request and derives tile shape, grid and row split.
// mode: 0 = one tap per workgroup (conv_wgrad_kernel), 1 = one filter row per workgroup (conv_wgrad_row_kernel; ta = WM)
// fold: `d` is the folded request of an x2-upsampled 3 x 3 layer (wgrad_plan below): only the one-wave-per-SIMD 128 x 128 tile serves it
static int wgrad_plan1(const dvd_wgrad_desc* d, WgK& p, dim3& grid, int& ta, int& tb, long long& msplit, int& mode, bool fold) {
    if (!d || !d->x || !d->dy || !d->dw) return DVD_E_ARG;
    int logH = ilog2_exact(d->H), logW = ilog2_exact(d->W);
    const bool pow2 = logH >= 0 && logW >= 0;
    if (!pow2) logH = logW = -1;           // division-based indexing, one-tap kernel only
    if (d->up2 && ((d->H | d->W) & 1)) return DVD_E_SHAPE;
    if ((d->C & 7) || (d->ldx & 7) || (d->Cy & 7) || (d->ldy & 7) || !(d->kt & d->kh & d->kw & 1)) return DVD_E_SHAPE;
    if (d->Cout > d->Cy || d->Cin_real > d->C) return DVD_E_ARG;
    if (d->dtype != DVD_BF16 && d->dtype != DVD_F32) return DVD_E_ARG;
    const long long M = (long long)d->frames * d->T * d->H * d->W;
    if (M >= (1ll << 31) - 64) return DVD_E_SHAPE;
    p.x = (const char*)d->x; p.dy = (const char*)d->dy; p.dw = d->dw;
    p.M = (int)M; p.C = d->C; p.ldx = d->ldx; p.Cin_real = d->Cin_real; p.Cout = d->Cout; p.Cy = d->Cy; p.ldy = d->ldy;
    p.T = d->T; p.H = d->H; p.W = d->W; p.logH = logH; p.logW = logW;
    p.Hin = d->up2 ? d->H / 2 : d->H; p.Win = d->up2 ? d->W / 2 : d->W;
    p.kt = d->kt; p.kh = d->kh; p.kw = d->kw; p.up2 = d->up2; p.relu_in = d->relu_in;
    // bf16: 256-wide tile along whichever channel axis is long enough (2x the MFMAs per barrier)
    ta = 2; tb = 2;
    if (d->dtype == DVD_BF16) {
        if (d->Cout >= 192) ta = 4;
        else if (d->Cin_real >= 192) tb = 4;
    }
    mode = (pow2 && d->dtype == DVD_BF16 && d->kh == d->kw && (d->kw == 3 || d->kw == 5) && (!d->up2 || (d->kw == 3 && d->kt == 1)) &&
            ((d->W >= 8 && (d->H * d->W) % 32 == 0) ||
             // 4 x 4 frames (round 4): a 32-pixel step = two whole frames
             (d->W == 4 && d->H == 4 && d->kt == 1 && d->T == 1 && !d->up2 && (d->frames & 1) == 0))) ? 1 : 0;
    if (mode == 1) {   // 64 channels for thin outputs, else 256 or 128, whichever pads Cout less (256 on a tie)
        const int w4 = (d->Cout + 255) / 256 * 256, w2 = (d->Cout + 127) / 128 * 128;
        ta = d->Cout <= 64 ? 1 : (w4 <= w2 ? 4 : 2); tb = 1;
        // 128-channel output tile, 5 taps: take 128 input channels too (one 8-wave workgroup per CU whose halves stagger their
        // loads, like the 256-channel tile) when that pads Cin no further: 1.20 -> 1.33 PF/s on 3.1 M x 256 -> 384; with 3 taps
        // the two 4-wave workgroups per CU of the 64-channel form stay 3 % ahead
        const bool ci128 = (d->Cin_real + 127) / 128 * 128 == (d->Cin_real + 63) / 64 * 64;      // 128-channel input tiles pad no further
        if (ta == 2 && d->kw == 5 && !d->up2 && ci128) tb = 2;
#ifndef DVD_WG_ROW4                    // 0 = 8-wave tiles only, 1 = the 128 x 128 one-wave-per-SIMD 3-tap tile only, 2 = + the pinned 256 x 128 tile, 3 = + 5 taps
#define DVD_WG_ROW4 3
#endif
#ifndef DVD_ROW4_DEPTH                 // staging depth of the 128 x 128 one-wave-per-SIMD tile (experiment knob)
#define DVD_ROW4_DEPTH 2
#endif
        // round 6: one wave per SIMD with the whole register file (conv_wgrad_row4_kernel), mode 2.
        //   3 taps: 256 x 128 channels (pinned, 24 tiles; with a bias gradient: one staging set) or 128 x 128
        //   5 taps: 256 x 64 or 128 x 128 (20 tiles)
        // Needs a round of workgroups with >= 4096 rows each (256 x 256 channels on 48 k rows ran 172 us against 101 on the 8-wave tile);
        // a caller-forced split (msplit > 0: tests) takes the tile whatever the size.
        if (DVD_WG_ROW4 && !d->up2 && ta >= 2 && (d->kw == 3 ? ci128 : (DVD_WG_ROW4 >= 3 && (ta == 4 || ci128)))) {
            int t4 = ta, b4 = 2;
#ifndef DVD_WG_PIN_BIAS                // 1 = launches with a bias gradient take the pinned tile too, with ONE staging set (DEPTH 1) so that the column sums have registers
#define DVD_WG_PIN_BIAS 1
#endif
            if (d->kw == 3) { if (DVD_WG_ROW4 == 1 || (d->dbias && !DVD_WG_PIN_BIAS) || fold) t4 = 2; }
            else if (ta == 4) b4 = 1;
            const long long wgs = (long long)((d->Cout + t4 * 64 - 1) / (t4 * 64)) * ((d->Cin_real + b4 * 64 - 1) / (b4 * 64)) * d->kt * d->kh * (M / 4096);
            if (wgs >= 224 || d->msplit > 0) { mode = 2; tb = b4; ta = t4; }
        }
    }
    if (fold && mode != 2) return DVD_E_SHAPE;
    const int nrow = fold ? 2 : d->kt * d->kh;                   // filter rows per (channel tile, slice) of the filter-row kernels
    p.tiles_co = (d->Cout + ta * 64 - 1) / (ta * 64); p.tiles_ci = (d->Cin_real + tb * 64 - 1) / (tb * 64);
    p.s_co = d->s_co; p.s_ci = d->s_ci; p.s_tap = d->s_tap; p.dbias = d->dbias; p.ws = nullptr;
    p.xcd_remap = 1;
    {
        const size_t esz = d->dtype == DVD_BF16 ? 2 : 4;
        const size_t rows_in = (size_t)d->frames * d->T * p.Hin * p.Win;
        p.x_bytes = ((rows_in - 1) * (size_t)d->ldx + d->C) * esz;
        p.dy_bytes = (((size_t)M - 1) * (size_t)d->ldy + d->Cy) * esz;
        p.maxshift = ((d->kt >> 1) * d->H + (d->kh >> 1)) * d->W + (d->kw >> 1);
    }
    const int ntaps = d->kt * d->kh * d->kw;
    msplit = d->msplit;
    if (msplit < 1) {   // auto: ~16 workgroups per CU, every workgroup keeping >= 4k rows of reduction.  Swept on the
                        // full step with the workspace reduction (ms of wgrad per step): (1024,8192) 292,
                        // (1536,8192) 267, (3072,4096) 247, (4096,4096) 243, (6144,4096) 242, (8192,2048) 252
        constexpr long long tgt = 2048;   // re-swept with whole-round grids: 4096 185.6, 2048 183.2, 1024 183.5 ms
        constexpr long long minrows = 4096;
        // (rounds 1-3, swept with whole-round grids: 1024 187.7, 1536 185.1, 2048 190.8, 3072 192.2 ms of weight gradients.  End of round 4 -- the
        //  chain's kernels no longer leave the side stream the CUs they used to -- fewer, longer slices win on the STEP: 256 503.4 / 503.8,
        //  384 498.1 / 498.1, 512 495.6 / 495.7, 768 497.7 / 498.5, 1024 498.4 / 497.3, 1536 500.4 / 500.3, 3072 501.3 ms, one box)
        constexpr long long tgt_row = 512;
        const long long base = (long long)p.tiles_co * p.tiles_ci * (mode >= 1 ? nrow : ntaps);
        // one-wave-per-SIMD tiles (one workgroup per CU at a time): ONE round of 256 workgroups -- step at C2, interleaved on one box:
        // one round 451.3 / 451.8 ms, two 452.1 / 452.3, three 454.5 / 454.2 (half the slice workspace per halving) -- two rounds
        // beyond 6 M rows (configs[3], 4x the rows per launch: one round 1802 ms, two 1739)
        const long long tgt4 = M > (6ll << 20) ? 512 : 256;
#ifndef DVD_WG_TGT1                    // 64-channel output tile: 2-wave workgroups, three per CU -> one round of 768 (tools/conv_microbench.py wgrad
#define DVD_WG_TGT1 768                // 30,32 incl. reduce + bias kernels, interleaved: 512 1736 / 314 us, 768 1537 / 264, 1536 1597 / 322, 3072 1648 / 326)
#endif
        const long long tgt1 = (mode == 1 && ta == 1) ? (long long)DVD_WG_TGT1 : tgt_row;
        msplit = ((mode == 2 ? tgt4 : mode == 1 ? tgt1 : tgt) + base - 1) / base;
        long long cap = M / minrows > 0 ? M / minrows : 1;
        if (ntaps == 1 && cap * base < 256) {
            // short 1 x 1 layers (shortcut and attention projections on <= 16-pixel maps): 16 workgroups of 4096 rows each took
            // 110-150 us for 4 GFLOP -- slices down to 256 rows until every CU has a workgroup (round 5)
            const long long want = (256 + base - 1) / base, cap2 = M / 256;
            cap = cap2 < want ? (cap2 > cap ? cap2 : cap) : want;
        }
        if (msplit > cap) msplit = cap;
        // whole rounds: the workgroups run `conc` at a time (register / LDS limited); a grid a few workgroups over a
        // multiple of that pays a full extra round (20 x 103 = 2060 workgroups = 8.05 rounds of 256 -> 9 rounds)
        const long long conc = 256ll * ((mode >= 1 && (ta == 4 || tb == 2)) ? 1 : (mode == 1 && ta == 1 && DVD_WG_TGT1 > 512) ? 3 : 2);
        if (base * msplit > conc) {
            const long long rounds = (base * msplit + conc / 2) / conc;           // nearest
            long long ms2 = rounds * conc / base;
            if (ms2 >= 1 && ms2 <= cap) msplit = ms2;
        }
    }
    long long rows = (M + msplit - 1) / msplit;
    {   // a workgroup's row slice is addressed with 32-bit byte offsets
        const long long ldy_eff = fold ? 4ll * d->ldy : d->ldy;     // (folded: a slice of input pixels spans four times as many rows of dy)
        const long long ldmax = (d->ldx > ldy_eff ? d->ldx : ldy_eff) * (d->dtype == DVD_BF16 ? 2ll : 4ll);
        const long long cap = (1ll << 31) / ldmax;
        if (rows > cap) rows = cap;
    }
    rows = (rows + 31) / 32 * 32;
    if (rows < 32) rows = 32;
    msplit = (M + rows - 1) / rows;
    if (mode == 2 && d->msplit < 1) {
        // the 32-bit offset cap can undo the whole-round rounding above (configs[3]: 3.1 M x 512 -> 512 behind a 1536-wide dy: 5 slices x 80
        // workgroups = 1.56 rounds of one workgroup per CU): fill the last round
        const long long base = (long long)p.tiles_co * p.tiles_ci * nrow, wg = base * msplit;
        if (wg > 256 && wg % 256 > 0 && wg % 256 < 192) {
            const long long ms2 = ((wg + 255) / 256 * 256) / base;
            if (ms2 > msplit && M / ms2 >= 2048) {
                rows = ((M + ms2 - 1) / ms2 + 31) / 32 * 32;
                msplit = (M + rows - 1) / rows;
            }
        }
    }
    p.rows_per_split = (int)rows;
    grid = dim3(p.tiles_co * p.tiles_ci * (mode >= 1 ? nrow : ntaps), 1, (unsigned)msplit);
    p.fold = 0; p.cout_f = d->Cout;
    return DVD_OK;
}

// A 3 x 3 layer over a nearest-x2 upsampled input is planned on the input grid when the folded form applies (see
// conv_wgrad_row4_kernel<.., FOLD>): whole 64-channel blocks of Cout, 128-wide input-channel tiles, a slice workspace (`may_fold`:
// the caller supplied one, or is asking for its size) and enough rows for the one-wave-per-SIMD tile.
static int wgrad_plan(const dvd_wgrad_desc* d, WgK& p, dim3& grid, int& ta, int& tb, long long& msplit, int& mode, bool may_fold) {
#ifndef DVD_WG_FOLD
#define DVD_WG_FOLD 1
#endif
    if (DVD_WG_FOLD && may_fold && d && d->up2 && d->dtype == DVD_BF16 && d->kt == 1 && d->kh == 3 && d->kw == 3 && d->T == 1 &&
        d->Cout % 64 == 0 && d->Cy >= d->Cout && !((d->H | d->W) & 1)) {
        dvd_wgrad_desc e = *d;
        e.H = d->H / 2; e.W = d->W / 2; e.up2 = 0;
        e.Cout = e.Cy = 4 * d->Cout;
        if (wgrad_plan1(&e, p, grid, ta, tb, msplit, mode, true) == DVD_OK) {
            const size_t M4 = (size_t)d->frames * d->T * d->H * d->W;
            p.dy_bytes = ((M4 - 1) * (size_t)d->ldy + d->Cy) * 2;
            p.fold = 1; p.cout_f = d->Cout;
            return DVD_OK;
        }
    }
    return wgrad_plan1(d, p, grid, ta, tb, msplit, mode, false);
}

extern "C" long long dvd_conv_wgrad_ws_floats(const dvd_wgrad_desc* d) {
    WgK p; dim3 grid; int ta, tb, mode; long long msplit;
    if (const long long thin = dvd_wgrad_thin_ws_floats(d)) return thin;       // 3 (8) channels on one side: wgrad_thin.hip
    if (wgrad_plan(d, p, grid, ta, tb, msplit, mode, true) != DVD_OK || (msplit <= 1 && !p.fold)) return 0;
    if (mode >= 1) return msplit * (long long)grid.x * d->kw * (ta * 64) * (tb * 64) + msplit * (long long)grid.x * (ta * 64);   // + bias partials
    return msplit * (long long)grid.x * (ta * 64) * (tb * 64) + msplit * (long long)p.tiles_co * (ta * 64);      // + bias partials
}

extern "C" int dvd_conv_wgrad(const dvd_wgrad_desc* d, void* stream) {
    WgK p; dim3 grid; int ta, tb, mode; long long msplit;
    const int rc = wgrad_plan(d, p, grid, ta, tb, msplit, mode, d && d->ws != nullptr);
    if (rc != DVD_OK) return rc;
    const int ntaps = d->kt * d->kh * d->kw;
    const long long Mreal = (long long)d->frames * d->T * d->H * d->W;
    if (d->ws && dvd_wgrad_thin_ws_floats(d)) {    // the thin ends of the networks (stems, RGB layer): taps folded into the matrix dimension
        if (d->overwrite && (d->s_tap != 1 || d->s_ci != ntaps || d->s_co != (long long)d->Cin_real * ntaps)) return DVD_E_ARG;
        ProfScope prof(1, 2.0 * (double)Mreal * d->Cout * d->Cin_real * ntaps, stream, (int)Mreal, d->C, d->Cout, ntaps, 0, d->relu_in << 1);
        prof.r.variant = 3;
        return dvd_wgrad_thin(d, stream);
    }
    if (d->ws && (msplit > 1 || p.fold)) p.ws = d->ws;         // two-phase reduction; a single slice adds straight into dw (not the folded form)
    const int overwrite = d->overwrite != 0;       // dw = result instead of dw += result (the caller need not zero it)
    if (overwrite) {
        if (d->s_tap != 1 || d->s_ci != ntaps || d->s_co != (long long)d->Cin_real * ntaps) return DVD_E_ARG;   // dense [co][ci][tap] only
        if (!p.ws && hipMemsetAsync(d->dw, 0, (size_t)d->Cout * d->Cin_real * ntaps * sizeof(float), (hipStream_t)stream) != hipSuccess)
            return DVD_E_LAUNCH;                   // single slice: the kernel adds with atomics
    }
    p.wsb = !p.ws ? nullptr : mode >= 1 ? p.ws + msplit * (long long)grid.x * d->kw * (ta * 64) * (tb * 64)
                                        : p.ws + msplit * (long long)grid.x * (ta * 64) * (tb * 64);
    ProfScope prof(1, 2.0 * (double)Mreal * d->Cout * d->Cin_real * ntaps, stream, (int)Mreal, d->C, d->Cout, ntaps, (int)msplit,
                   d->up2 | (d->relu_in << 1));
    hipStream_t st = (hipStream_t)stream;
    prof.r.variant = mode == 2 ? 4 : mode == 1 ? 1 : 2;        // (dvd_prof_report_variants: 1 = 8-wave filter-row tiles, 2 = one tap, 3 = thin ends, 4 = one wave per SIMD)
    if (mode >= 1) {
#define LAUNCH_ROW(WM_, KW_)                                                                        \
        do { if (d->relu_in) conv_wgrad_row_kernel<WM_, KW_, true><<<grid, WM_ * 128, 0, st>>>(p);      \
             else conv_wgrad_row_kernel<WM_, KW_, false><<<grid, WM_ * 128, 0, st>>>(p); } while (0)
#define LAUNCH_ROW_UP(WM_)                                                                          \
        do { if (d->relu_in) conv_wgrad_row_kernel<WM_, 3, true, true><<<grid, WM_ * 128, 0, st>>>(p);  \
             else conv_wgrad_row_kernel<WM_, 3, false, true><<<grid, WM_ * 128, 0, st>>>(p); } while (0)
#define LAUNCH_ROW_W(KW_)                                                                           \
        do { if (d->relu_in) conv_wgrad_row_kernel<2, KW_, true, false, 4><<<grid, 512, 0, st>>>(p);    \
             else conv_wgrad_row_kernel<2, KW_, false, false, 4><<<grid, 512, 0, st>>>(p); } while (0)
        if (mode == 2) {
#define LAUNCH_ROW4(KW_, AW_, BW_, WCO_, WCI_, DEPTH_)                                                                        \
            do { if (d->relu_in) conv_wgrad_row4_kernel<KW_, AW_, BW_, WCO_, WCI_, true, DEPTH_><<<grid, 256, 0, st>>>(p);       \
                 else conv_wgrad_row4_kernel<KW_, AW_, BW_, WCO_, WCI_, false, DEPTH_><<<grid, 256, 0, st>>>(p); } while (0)
            if (p.fold) {
                if (d->relu_in) conv_wgrad_row4_kernel<3, 4, 1, 1, 4, true, 2, true><<<grid, 256, 0, st>>>(p);
                else conv_wgrad_row4_kernel<3, 4, 1, 1, 4, false, 2, true><<<grid, 256, 0, st>>>(p);
                WgFoldRedK r{p.ws, p.dw, (int)msplit, (int)grid.x, p.tiles_ci, d->Cout, p.Cin_real, p.s_co, p.s_ci, p.s_tap, overwrite};
                wgrad_fold_reduce_kernel<<<cdiv(9ll * d->Cout * ((p.Cin_real + 3) / 4), 256), 256, 0, st>>>(r);
                if (p.dbias) {
                    WgFoldBiasK b{p.wsb, p.dbias, (int)msplit, (int)grid.x, p.tiles_ci, d->Cout};
                    wgrad_fold_bias_reduce_kernel<<<cdiv(d->Cout, 32), 1024, 0, st>>>(b);
                }
                return launch_status();
            }
            if (d->kw == 3) { if (ta == 4 && p.dbias) LAUNCH_ROW4(3, 4, 2, 2, 2, 1); else if (ta == 4) LAUNCH_ROW4(3, 4, 2, 2, 2, 2); else LAUNCH_ROW4(3, 4, 1, 1, 4, DVD_ROW4_DEPTH); }
            else            { if (ta == 4) LAUNCH_ROW4(5, 4, 1, 2, 2, 2); else LAUNCH_ROW4(5, 4, 1, 1, 4, 2); }
#undef LAUNCH_ROW4
        } else
        if (d->up2) { if (ta == 4) LAUNCH_ROW_UP(4); else if (ta == 2) LAUNCH_ROW_UP(2); else LAUNCH_ROW_UP(1); }
        else if (tb == 2) { if (d->kw == 5) LAUNCH_ROW_W(5); else LAUNCH_ROW_W(3); }
        else
        if (ta == 4)      { if (d->kw == 5) LAUNCH_ROW(4, 5); else LAUNCH_ROW(4, 3); }
        else if (ta == 2) { if (d->kw == 5) LAUNCH_ROW(2, 5); else LAUNCH_ROW(2, 3); }
        else              { if (d->kw == 5) LAUNCH_ROW(1, 5); else LAUNCH_ROW(1, 3); }
#undef LAUNCH_ROW
#undef LAUNCH_ROW_UP
#undef LAUNCH_ROW_W
        if (p.ws) {
            WgRowRedK r{p.ws, p.dw, (int)msplit, (int)grid.x, p.tiles_co, p.tiles_ci, ta * 64, tb * 64, d->kw, p.Cout, p.Cin_real,
                        p.s_co, p.s_ci, p.s_tap, overwrite};
            const long long n = (long long)grid.x * r.KW * r.BMc * r.BNc;
            wgrad_row_reduce_kernel<<<cdiv(n / 4, 256), 256, 0, st>>>(r);
            if (p.dbias) {
                WgBiasRedK b{p.wsb, p.dbias, (int)msplit, (int)grid.x, p.tiles_co, p.tiles_ci, d->kw, d->kt, ta * 64, p.Cout};
                wgrad_row_bias_reduce_kernel<<<p.tiles_co * (ta * 64 / 32), 1024, 0, st>>>(b);
            }
        }
        return launch_status();
    }
    if (d->dtype == DVD_BF16) {
        if (ta == 4) conv_wgrad_kernel<bf16_t, 4, 2><<<grid, NT, 0, st>>>(p);
        else if (tb == 4) conv_wgrad_kernel<bf16_t, 2, 4><<<grid, NT, 0, st>>>(p);
        else conv_wgrad_kernel<bf16_t, 2, 2><<<grid, NT, 0, st>>>(p);
    } else conv_wgrad_kernel<float, 2, 2><<<grid, NT, 0, st>>>(p);
    if (p.ws) {
        WgRedK r{p.ws, p.dw, (int)msplit, (int)grid.x, p.tiles_co * p.tiles_ci, p.tiles_ci, ta * 64, tb * 64, p.Cout,
                 p.Cin_real, p.s_co, p.s_ci, p.s_tap, overwrite};
        const long long n = (long long)grid.x * r.BMc * r.BNc;
        wgrad_reduce_kernel<<<cdiv(n / 4, 256), 256, 0, st>>>(r);
        if (p.dbias) wgrad_bias_reduce_kernel<<<cdiv(p.Cout, 64), 1024, 0, st>>>(p.wsb, p.dbias, (int)msplit, p.tiles_co, ta * 64, p.Cout);
    }
    return launch_status();
}
