// Weight gradients of the THIN ends of the networks (bf16 mode): 3 x 3 (x 3) convolutions with 3 (padded to 8) channels on one
// side and 64 on the other -- the discriminator stems (Discriminators.py:186, 335: 3 -> 64) and the generator's RGB layer
// (Generator.py:59: 64 -> 3).  The filter-row kernel of conv_igemm.hip pads the thin side to a 64-channel tile (30 TF/s on
// 3.1 M x 8 -> 64, 27 taps: 2.1 ms); here the thin side's KW taps are FOLDED into the matrix dimension instead:
//     dw[wide ch][tap = (dt, dy, dx)][thin ch] = sum over pixels p of  wide[p][ch] * thin[p + (dt, dy, dx) - pad][c]
// With 8 channels per pixel (16 bytes) four consecutive pixels of a line ARE a 32-wide row: the B operand (k = pixels, columns
// = (dx, c)) of tap row (dt, dy) for pixel x is the 64-byte window that starts one pixel to the left of x in a zero-haloed copy
// of line (t + dt, y + dy) -- read straight from LDS with ds_read_b64_tr_b16, the windows of neighbouring pixels overlapping.
// One 32 x 32 accumulator per (tap row, 32 wide channels): 18 for a 3 x 3 x 3 stem, spread over the four waves.
// Workgroups are persistent (they walk the image lines with a stride and keep their accumulators), leave them in a workspace
// and a two-stage reduce adds them up in a fixed order (deterministic like the other weight-gradient paths).
// The same kernel serves both roles: `thin` = x (stems: taps as they are) or `thin` = dy (RGB layer: x[p + o] dy[p] =
// x[p'] dy[p' - o], i.e. the tap index mirrored) -- only the final reduce knows which.
// Bias gradient: sums over the WIDE side ride along as one more column (a 1.0 in the unused 8th channel of the centre window),
// sums over the THIN side are taken by one wave from the centre line.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int WP = 144;                   // LDS row pitch of the wide line image (64 channels + 16 bytes)
constexpr int TENT = 68;                  // entries (pixels) per thin footprint line: x = -1 .. 66
constexpr int TPB = TENT * 16;
constexpr int PARTS = 16;                 // first reduce stage: workgroups are summed in 16 groups

struct ThinK {
    const bf16_t* wide; const bf16_t* thin;
    int ldw, ldt, F, T, H, W, kt;
    int relu_wide, relu_thin, ones, tsum;
    long long lines;
    float* ws;
};

__device__ __forceinline__ u32x4 relu8(u32x4 v) {
    auto r2 = [](uint32_t a) { const uint32_t m = ((a >> 15) & 0x00010001u) * 0xffffu; return a & ~m; };
    v.x = r2(v.x); v.y = r2(v.y); v.z = r2(v.z); v.w = r2(v.w);
    return v;
}
__device__ __forceinline__ bf16x8 tr2(const char* lo, int hi_off) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lo);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lo + hi_off));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// grid.x workgroups of 4 waves; wave w owns the accumulators (tap row (w >> 1) + 2 i, channel block w & 1), i = 0 .. NI - 1
template <int NI>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(ThinK p) {
    __shared__ __attribute__((aligned(16))) char sm[64 * WP + 9 * TPB];
    char* const Wl = sm;
    char* const Tl = sm + 64 * WP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntr = p.kt * 3, pt = p.kt >> 1;
    for (int i = tid; i < 9 * TPB / 16; i += 256) reinterpret_cast<u32x4*>(Tl)[i] = u32x4{0u, 0u, 0u, 0u};   // halo entries stay zero
    const int h = lane >> 5, i16 = lane & 15, cb16 = (lane >> 4) & 1;
    const int rb = wave & 1, tr0 = wave >> 1;
    f32x16 acc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float ts[4] = {0.f, 0.f, 0.f, 0.f};
    const char* const a_lo = Wl + (8 * h + (i16 >> 2)) * WP + (rb * 32 + cb16 * 16 + (i16 & 3) * 4) * 2;
    const int b_lo = (8 * h + (i16 >> 2)) * 16 + cb16 * 32 + (i16 & 3) * 8;
    // the next line's pieces are requested before this line is multiplied (2 + 3 pieces of 16 bytes per thread)
    u32x4 wv[2], tv[3];
    auto fetch = [&](long long line) __attribute__((always_inline)) {
        const int y = (int)(line % p.H);
        const long long ft = line / p.H;
        const int t = (int)(ft % p.T);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + 256 * j, px = i >> 3, ck = i & 7;
            wv[j] = u32x4{0u, 0u, 0u, 0u};
            if (i < p.W * 8) wv[j] = *reinterpret_cast<const u32x4*>(p.wide + ((size_t)line * p.W + px) * p.ldw + ck * 8);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int i = tid + 256 * j, tr = i / p.W, px = i - tr * p.W;
            const int dtI = tr / 3, dyI = tr - dtI * 3;
            const int tt = t + dtI - pt, yy = y + dyI - 1;
            tv[j] = u32x4{0u, 0u, 0u, 0u};
            if (i < ntr * p.W && (unsigned)tt < (unsigned)p.T && (unsigned)yy < (unsigned)p.H)
                tv[j] = *reinterpret_cast<const u32x4*>(p.thin + (((size_t)(ft - t + tt) * p.H + yy) * p.W + px) * p.ldt);
        }
    };
    fetch(blockIdx.x);
    for (long long line = blockIdx.x; line < p.lines; line += gridDim.x) {
        __syncthreads();                                   // the previous line's fragments are read (and the zero fill has landed)
#pragma unroll
        for (int j = 0; j < 2; ++j) {                      // wide line: W pixels x 8 pieces of 16 bytes
            const int i = tid + 256 * j, px = i >> 3, ck = i & 7;
            if (i < p.W * 8) *reinterpret_cast<u32x4*>(Wl + px * WP + ck * 16) = p.relu_wide ? relu8(wv[j]) : wv[j];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {                      // thin footprint: lines (t + dt, y + dy), entry px + 1 = pixel px
            const int i = tid + 256 * j, tr = i / p.W, px = i - tr * p.W;
            if (i < ntr * p.W) {
                u32x4 v = p.relu_thin ? relu8(tv[j]) : tv[j];
                if (p.ones && tr == pt * 3 + 1) v.w = (v.w & 0xffffu) | 0x3f800000u;      // channel 7 := 1.0 (bias column)
                *reinterpret_cast<u32x4*>(Tl + tr * TPB + (px + 1) * 16) = v;
            }
        }
        __syncthreads();
        if (line + gridDim.x < p.lines) fetch(line + gridDim.x);
        for (int k0 = 0; k0 < p.W; k0 += 16) {
            const bf16x8 a = tr2(a_lo + k0 * WP, 4 * WP);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int tr = tr0 + 2 * i;
                if (tr < ntr)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, tr2(Tl + tr * TPB + k0 * 16 + b_lo, 4 * 16), acc[i], 0, 0, 0);
            }
        }
        if (p.tsum && wave == 3 && lane < p.W) {           // sums over the thin side: the centre line, 4 channels
            const uint2 v = *reinterpret_cast<const uint2*>(Tl + (pt * 3 + 1) * TPB + (lane + 1) * 16);
            ts[0] += __uint_as_float(v.x << 16); ts[1] += __uint_as_float(v.x & 0xffff0000u);
            ts[2] += __uint_as_float(v.y << 16); ts[3] += __uint_as_float(v.y & 0xffff0000u);
        }
    }
    const int NU = ntr * 2;
    float* out = p.ws + (size_t)blockIdx.x * (NU * 1024 + 8);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int u = wave + 4 * i;
        if (u < NU)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[(u * 16 + r) * 64 + lane] = acc[i][r];
    }
    if (p.tsum && wave == 3) {
#pragma unroll
        for (int c = 0; c < 4; ++c) ts[c] = wave_sum(ts[c]);
        if (lane < 4) out[NU * 1024 + lane] = lane == 0 ? ts[0] : lane == 1 ? ts[1] : lane == 2 ? ts[2] : ts[3];
    } else if (wave == 3 && lane < 4) out[NU * 1024 + lane] = 0.f;
}

// stage 1: out[part][e] = sum over the workgroups g = part, part + PARTS, ... of in[g][e]      (e < per)
__global__ void thin_reduce1_kernel(const float* in, float* out, int G, int per) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x, part = blockIdx.y;
    if (e >= per) return;
    float a = 0.f;
    for (int g = part; g < G; g += PARTS) a += in[(size_t)g * per + e];
    out[(size_t)part * per + e] = a;
}
struct ThinRedK {
    const float* ws2; float* dw; float* dbias;
    int per, kt, creal, mirror, thin_is_ci, overwrite;
    long long s_co, s_ci, s_tap;
};
// stage 2: one thread per accumulator element: adds the PARTS partials, decodes (wide channel, tap, thin channel), writes dw
__global__ void thin_reduce2_kernel(ThinRedK p) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int NU = p.kt * 3 * 2;
    if (e >= NU * 1024 + 4) return;
    float v = 0.f;
    for (int part = 0; part < PARTS; ++part) v += p.ws2[(size_t)part * p.per + e];
    if (e >= NU * 1024) {                                  // sums over the thin side (its bias gradient)
        const int c = e - NU * 1024;
        if (p.dbias && !p.thin_is_ci && c < p.creal) p.dbias[c] += v;
        return;
    }
    const int lane = e & 63, r = (e >> 6) & 15, u = e >> 10;
    const int trI = u >> 1, rb = u & 1;
    const int cw = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int n = lane & 31, s = n >> 3, c = n & 7;
    const int dtI = trI / 3, dyI = trI - dtI * 3;
    if (p.dbias && p.thin_is_ci && n == 15 && dtI == (p.kt >> 1) && dyI == 1) p.dbias[cw] += v;       // the 1.0 column: sums over the wide side
    if (s >= 3 || c >= p.creal) return;
    const int dt = p.mirror ? p.kt - 1 - dtI : dtI, dy = p.mirror ? 2 - dyI : dyI, dx = p.mirror ? 2 - s : s;
    const int tap = (dt * 3 + dy) * 3 + dx;
    const long long o = p.thin_is_ci ? cw * p.s_co + c * p.s_ci + tap * p.s_tap : c * p.s_co + cw * p.s_ci + tap * p.s_tap;
    if (p.overwrite) p.dw[o] = v; else p.dw[o] += v;
}

struct ThinPlan { bool ok; bool thin_is_ci; int G, NU, per; long long lines; };
ThinPlan thin_plan(const dvd_wgrad_desc* d) {
    ThinPlan q = {};
    constexpr int use = 1;
    if (!use || !d || d->dtype != DVD_BF16 || d->up2 || d->kh != 3 || d->kw != 3 || (d->kt != 1 && d->kt != 3)) return q;
    if (d->W < 16 || d->W > 64 || (d->W & 15) || d->H < 1 || d->T < 1 || d->frames < 1) return q;
    if (d->msplit > 1) return q;                            // a caller that asks for a specific row split gets the general kernels
    const bool a = d->C == 8 && d->ldx == 8 && d->Cin_real <= 7 && d->Cout == 64 && d->Cy >= 64;          // x thin (stems)
    const bool b = d->Cy == 8 && d->ldy == 8 && d->Cout <= 4 && d->C == 64 && d->Cin_real == 64;            // dy thin (RGB layer)
    if (!a && !b) return q;
    q.lines = (long long)d->frames * d->T * d->H;
    if (q.lines < 4 * PARTS) return q;
    q.G = (int)(q.lines < 2048 ? q.lines / PARTS * PARTS : 2048);         // persistent workgroups (8 per CU fit the LDS)
    q.NU = d->kt * 3 * 2;
    q.per = q.NU * 1024 + 8;
    q.thin_is_ci = a;
    q.ok = true;
    return q;
}

}  // namespace

// (internal, called by dvd_conv_wgrad / dvd_conv_wgrad_ws_floats of conv_igemm.hip)  floats of workspace, 0 = not served here
long long dvd_wgrad_thin_ws_floats(const dvd_wgrad_desc* d) {
    const ThinPlan q = thin_plan(d);
    return q.ok ? (long long)(q.G + PARTS) * q.per : 0;
}

int dvd_wgrad_thin(const dvd_wgrad_desc* d, void* stream) {
    const ThinPlan q = thin_plan(d);
    if (!q.ok || !d->ws || !d->x || !d->dy || !d->dw) return DVD_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    ThinK p = {};
    if (q.thin_is_ci) { p.wide = (const bf16_t*)d->dy; p.ldw = d->ldy; p.thin = (const bf16_t*)d->x; p.ldt = d->ldx; p.relu_thin = d->relu_in; p.ones = d->dbias != nullptr; }
    else { p.wide = (const bf16_t*)d->x; p.ldw = d->ldx; p.thin = (const bf16_t*)d->dy; p.ldt = d->ldy; p.relu_wide = d->relu_in; p.tsum = d->dbias != nullptr; }
    p.F = d->frames; p.T = d->T; p.H = d->H; p.W = d->W; p.kt = d->kt; p.lines = q.lines; p.ws = d->ws;
    if (d->kt == 3) wgrad_thin_kernel<5><<<q.G, 256, 0, st>>>(p);
    else wgrad_thin_kernel<2><<<q.G, 256, 0, st>>>(p);
    float* ws2 = d->ws + (size_t)q.G * q.per;
    thin_reduce1_kernel<<<dim3(cdiv(q.per, 256), PARTS), 256, 0, st>>>(d->ws, ws2, q.G, q.per);
    ThinRedK r = {ws2, d->dw, d->dbias, q.per, d->kt, q.thin_is_ci ? d->Cin_real : d->Cout, q.thin_is_ci ? 0 : 1, q.thin_is_ci ? 1 : 0,
                  d->overwrite != 0, d->s_co, d->s_ci, d->s_tap};
    thin_reduce2_kernel<<<cdiv(q.NU * 1024 + 4, 256), 256, 0, st>>>(r);
    return launch_status();
}
