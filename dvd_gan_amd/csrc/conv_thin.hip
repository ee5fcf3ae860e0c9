// 3 x 3 (x 3) convolutions with 3 (padded to 8) INPUT channels and 64 output channels, bf16 mode: the discriminator stems
// (Discriminators.py:186, 335) and the backward-data pass of the generator's RGB layer (Generator.py:59, dy has 3 channels).
// The halo-staged kernel of conv_igemm.hip spends one 32-wide K step per tap on them (8 real channels of 32: 27 K steps for a
// 3 x 3 x 3 stem).  Here the KW taps are folded into the K dimension, as in wgrad_thin.hip: with 16 bytes per pixel the four
// consecutive pixels x - 1 .. x + 2 of a zero-haloed line copy ARE one 32-wide K step (the fourth meets zero weights), so a tap
// ROW (dt, dy) is ONE K step: 9 instead of 27.  out^T = W^T (rows = out channels, k = (dx, c)) x windows (k, columns = pixels):
// a lane ends up with 4 x 4 consecutive out channels of ONE pixel (the tile leaves through LDS as whole lines), the weight fragments of all tap rows stay in
// registers for the life of the (persistent) workgroup, and the only traffic in the loop is the 16-byte-per-pixel input and the
// output itself.  Epilogue: bias, ReLU, ReLU mask of a backward-data result.
#include "common.h"

namespace {

constexpr int TENT = 68;                  // entries (pixels) per footprint line: x = -1 .. 66
constexpr int PG = 64;                    // pixels per group = 64 / W consecutive lines

struct ThinFwdK {
    const bf16_t* in; const bf16_t* wt; const float* bias; const bf16_t* mask; bf16_t* out;
    int ldmask, ldo, T, H, W, relu_in, act, out_f32;
    long long groups;
};

__device__ __forceinline__ u32x4 relu8(u32x4 v) {
    auto r2 = [](uint32_t a) { const uint32_t m = ((a >> 15) & 0x00010001u) * 0xffffu; return a & ~m; };
    v.x = r2(v.x); v.y = r2(v.y); v.z = r2(v.z); v.w = r2(v.w);
    return v;
}

// wave w: pixels (w >> 1) * 32 .. + 31 of the group, out channels (w & 1) * 32 .. + 31
template <int KT>
__global__ __launch_bounds__(256) void conv_thin_in_kernel(ThinFwdK p) {
    constexpr int NTR = KT * 3, PT = KT / 2;
    constexpr int NP = (PG * NTR + 255) / 256;             // 16-byte pieces of a group's footprint per thread
    constexpr int OP = 144;                                // row pitch of the output tile (64 channels + 16 bytes)
    __shared__ __attribute__((aligned(16))) char sm[2 * NTR * TENT * 16 + PG * OP];
    char* const ol = sm + 2 * NTR * TENT * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5, rbk = wave >> 1, cb = wave & 1;
    const int lpg = PG / p.W;                              // lines per group (1 or 2)
    for (int i = tid; i < 2 * NTR * TENT; i += 256) reinterpret_cast<u32x4*>(sm)[i] = u32x4{0u, 0u, 0u, 0u};     // halo entries stay zero
    bf16x8 wf[NTR][2];
#pragma unroll
    for (int tr = 0; tr < NTR; ++tr)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            wf[tr][kk] = *reinterpret_cast<const bf16x8*>(p.wt + ((size_t)((tr * 2 + kk) * 2 + cb) * 64 + lane) * 8);
    f32x4 b4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        b4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias) b4[i] = *reinterpret_cast<const f32x4*>(p.bias + cb * 32 + 8 * i + 4 * h);
    }
    const int pix = rbk * 32 + l31, li = pix / p.W, x = pix - li * p.W;
    const char* const win0 = sm + ((li * NTR) * TENT + x + h) * 16;
    u32x4 tv[NP];
    auto fetch = [&](long long g) __attribute__((always_inline)) {
        const long long line0 = g * lpg;
        const int y0 = (int)(line0 % p.H);
        const long long ft = line0 / p.H;
        const int t = (int)(ft % p.T);
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = tid + 256 * j, l = i / (NTR * p.W), rem = i - l * (NTR * p.W), tr = rem / p.W, px = rem - tr * p.W;
            const int dtI = tr / 3, dyI = tr - dtI * 3;
            const int tt = t + dtI - PT, yy = y0 + l + dyI - 1;
            tv[j] = u32x4{0u, 0u, 0u, 0u};
            if (i < PG * NTR && (unsigned)tt < (unsigned)p.T && (unsigned)yy < (unsigned)p.H)
                tv[j] = *reinterpret_cast<const u32x4*>(p.in + (((size_t)(ft - t + tt) * p.H + yy) * p.W + px) * 8);
        }
    };
    fetch(blockIdx.x);
    for (long long g = blockIdx.x; g < p.groups; g += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = tid + 256 * j, l = i / (NTR * p.W), rem = i - l * (NTR * p.W), tr = rem / p.W, px = rem - tr * p.W;
            if (i < PG * NTR)
                *reinterpret_cast<u32x4*>(sm + ((l * NTR + tr) * TENT + px + 1) * 16) = p.relu_in ? relu8(tv[j]) : tv[j];
        }
        __syncthreads();
        if (g + gridDim.x < p.groups) fetch(g + gridDim.x);
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tr = 0; tr < NTR; ++tr)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)      // window of pixel x in tap row tr: entries x .. x + 3; this lane's 8 k values = entry x + 2 kk + h
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    wf[tr][kk], *reinterpret_cast<const bf16x8*>(win0 + (tr * TENT + 2 * kk) * 16), acc, 0, 0, 0);
        // this lane: pixel `pix` of the group, out channels cb * 32 + 8 i + 4 h + {0..3} in registers 4 i .. 4 i + 3.  The tile goes
        // through LDS so that the rows leave as whole 128-byte lines (8-byte stores per lane straight from the accumulators touch 32
        // lines per instruction: 1.25 ms instead of 0.78 on the RGB layer's backward-data pass)
        if (p.out_f32) {
            // dvd_conv_desc.out_f32 (parity tests: the sums before the bf16 rounding, as the general kernels offer them): straight
            // from the accumulators, 16 bytes per lane
            const size_t row = (size_t)g * PG + pix;
            float* o32 = reinterpret_cast<float*>(p.out) + row * p.ldo;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = cb * 32 + 8 * i + 4 * h + e;
                    float t = acc[4 * i + e] + b4[i][e];
                    if (p.act == DVD_ACT_RELU) t = fmaxf(t, 0.f);
                    if (p.mask && !(bf16_to_f32(p.mask[row * p.ldmask + c]) > 0.f)) t = 0.f;
                    v[e] = t;
                }
                *reinterpret_cast<f32x4*>(o32 + cb * 32 + 8 * i + 4 * h) = v;
            }
            __syncthreads();
            continue;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc[4 * i + e] + b4[i][e];
                if (p.act == DVD_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
            }
            uint2 o;
            o.x = pack2_bf16(v[0], v[1]); o.y = pack2_bf16(v[2], v[3]);
            *reinterpret_cast<uint2*>(ol + pix * OP + (cb * 32 + 8 * i + 4 * h) * 2) = o;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + 256 * j, prow = i >> 3, ck = i & 7;
            u32x4 v = *reinterpret_cast<const u32x4*>(ol + prow * OP + ck * 16);
            const size_t row = (size_t)g * PG + prow;
            if (p.mask) {
                const u32x4 m = *reinterpret_cast<const u32x4*>(p.mask + row * p.ldmask + ck * 8);
                auto keep = [](uint32_t val, uint32_t mk) -> uint32_t {      // per bf16 half: value where mask > 0, else 0
                    const uint32_t lo = (mk & 0x8000u) || !(mk & 0x7fffu) ? 0u : 0xffffu;
                    const uint32_t hi = (mk & 0x80000000u) || !(mk & 0x7fff0000u) ? 0u : 0xffff0000u;
                    return val & (lo | hi);
                };
                v.x = keep(v.x, m.x); v.y = keep(v.y, m.y); v.z = keep(v.z, m.z); v.w = keep(v.w, m.w);
            }
            *reinterpret_cast<u32x4*>(p.out + row * p.ldo + ck * 8) = v;
        }
    }
}

// [ntaps = kt * 9][64][8] forward pack -> [tap row][kk][channel block][lane][8]: lane (out channel cb * 32 + (l & 31), half l >> 5) holds
// the 8 input channels of tap dx = 2 kk + half (zeros for dx = 3)
__global__ void thin_image_kernel(const bf16_t* w, bf16_t* wt, int ntr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntr * 4 * 64) return;
    const int lane = i & 63, cb = (i >> 6) & 1, kk = (i >> 7) & 1, tr = i >> 8;
    const int s = 2 * kk + (lane >> 5), co = cb * 32 + (lane & 31);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (s < 3) v = *reinterpret_cast<const u32x4*>(w + ((size_t)(tr * 3 + s) * 64 + co) * 8);
    *reinterpret_cast<u32x4*>(wt + (size_t)i * 8) = v;
}

// ---------------------------------------------------------------------------- 64 -> 3 (8) channels, 3 x 3
// The generator's RGB layer forward (Generator.py:59,113-115: ReLU -> conv -> tanh) and the backward-data pass of the spatial
// discriminator's stem: the halo-staged kernel computes a 64-column tile for 3 real columns (31 TF/s, 1.4 ms on the RGB layer).
// Here out^T (rows = out channels: 32, 3 real; columns = 32 pixels) = W (rows, k = channels) x footprint (k, columns): the B
// operand is the natural [pixel][64 channels] footprint read 16 bytes per lane at the tap's offset, the A operand the weight image
// kept in LDS; 36 MFMAs per 32 pixels, one wave each.  A workgroup takes 128 consecutive pixels (2 or 4 lines) at a time.
constexpr int OPIT = 144;                 // footprint pixel pitch: 64 channels + 16 bytes
constexpr int OG = 128;                   // pixels per group

struct ThinOutK {
    const bf16_t* in; const bf16_t* wt; const float* bias; const bf16_t* mask; bf16_t* out;
    int ldi, ldmask, ldo, Cout, H, W, relu_in, act, out_f32;
    long long groups;
};

__global__ __launch_bounds__(256) void conv_thin_out_kernel(ThinOutK p) {
    __shared__ __attribute__((aligned(16))) char sm[36 * 1024 + 4 * 66 * OPIT];
    char* const wl = sm;
    char* const fl = sm + 36 * 1024;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int lpg = OG / p.W, ents = p.W + 2;              // lines per group (2 or 4); footprint: lpg + 2 lines of W + 2 entries
    for (int i = tid; i < 36 * 64; i += 256) reinterpret_cast<u32x4*>(wl)[i] = reinterpret_cast<const u32x4*>(p.wt)[i];
    for (int i = tid; i < 4 * 66 * OPIT / 16; i += 256) reinterpret_cast<u32x4*>(fl)[i] = u32x4{0u, 0u, 0u, 0u};   // halo entries stay zero
    const int pix = wave * 32 + l31, li = pix / p.W, x = pix - li * p.W;
    const char* const b0 = fl + (li * ents + x) * OPIT + h * 16;
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias)
#pragma unroll
        for (int c = 0; c < 4; ++c) if (4 * h + c < p.Cout) bias4[c] = p.bias[4 * h + c];
    constexpr int NP = 8;                                  // (lpg + 2) * W * 8 <= 2048 pieces of 16 bytes
    u32x4 tv[NP];
    auto fetch = [&](long long g) __attribute__((always_inline)) {
        const long long line0 = g * lpg;
        const int y0 = (int)(line0 % p.H);
        const long long f = line0 / p.H;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = tid + 256 * j, ck = i & 7, px = (i >> 3) % p.W, l = (i >> 3) / p.W;
            const int yy = y0 + l - 1;
            tv[j] = u32x4{0u, 0u, 0u, 0u};
            if (l < lpg + 2 && (unsigned)yy < (unsigned)p.H)
                tv[j] = *reinterpret_cast<const u32x4*>(p.in + ((size_t)(f * p.H + yy) * p.W + px) * p.ldi + ck * 8);
        }
    };
    fetch(blockIdx.x);
    for (long long g = blockIdx.x; g < p.groups; g += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = tid + 256 * j, ck = i & 7, px = (i >> 3) % p.W, l = (i >> 3) / p.W;
            if (l < lpg + 2) *reinterpret_cast<u32x4*>(fl + (l * ents + px + 1) * OPIT + ck * 16) = p.relu_in ? relu8(tv[j]) : tv[j];
        }
        __syncthreads();
        if (g + gridDim.x < p.groups) fetch(g + gridDim.x);
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int kc = 0; kc < 4; ++kc)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        *reinterpret_cast<const bf16x8*>(wl + (((dy * 3 + dx) * 4 + kc) * 64 + lane) * 16),
                        *reinterpret_cast<const bf16x8*>(b0 + (dy * ents + dx) * OPIT + kc * 32), acc, 0, 0, 0);
        // lane (pixel, half): registers 0 .. 3 = out channels 4 half + {0..3}; the lower half-lane collects all 8 and stores the pixel
        float v[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = acc[c] + bias4[c];
            if (p.act == DVD_ACT_RELU) a = fmaxf(a, 0.f);
            else if (p.act == DVD_ACT_TANH) a = gate_tanh<bf16_t>(a);
            a = (4 * h + c < p.Cout) ? a : 0.f;            // padded channels stay exactly zero
            const float o = __shfl_xor(a, 32, 64);
            v[c] = h ? o : a; v[4 + c] = h ? a : o;
        }
        if (h == 0 && p.out_f32) {                       // dvd_conv_desc.out_f32 (parity tests): the sums before the bf16 rounding
            const size_t row = (size_t)g * OG + pix;
            if (p.mask)
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (!(bf16_to_f32(p.mask[row * p.ldmask + c]) > 0.f)) v[c] = 0.f;
            float* o32 = reinterpret_cast<float*>(p.out) + row * p.ldo;
            const f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
            *reinterpret_cast<f32x4*>(o32) = lo; *reinterpret_cast<f32x4*>(o32 + 4) = hi;
        } else if (h == 0) {
            const size_t row = (size_t)g * OG + pix;
            u32x4 o;
            o.x = pack2_bf16(v[0], v[1]); o.y = pack2_bf16(v[2], v[3]); o.z = pack2_bf16(v[4], v[5]); o.w = pack2_bf16(v[6], v[7]);
            if (p.mask) {
                const u32x4 m = *reinterpret_cast<const u32x4*>(p.mask + row * p.ldmask);
                auto keep = [](uint32_t val, uint32_t mk) -> uint32_t {
                    const uint32_t lo = (mk & 0x8000u) || !(mk & 0x7fffu) ? 0u : 0xffffu;
                    const uint32_t hi = (mk & 0x80000000u) || !(mk & 0x7fff0000u) ? 0u : 0xffff0000u;
                    return val & (lo | hi);
                };
                o.x = keep(o.x, m.x); o.y = keep(o.y, m.y); o.z = keep(o.z, m.z); o.w = keep(o.w, m.w);
            }
            *reinterpret_cast<u32x4*>(p.out + row * p.ldo) = o;
        }
    }
}

// [9][R][64] pack (R <= 8 rows) -> [tap][kc][lane][8]: lane (out channel l & 31, half l >> 5) holds input channels 16 kc + 8 half .. + 7
__global__ void thin_out_image_kernel(const bf16_t* w, bf16_t* wt, int R) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 36 * 64) return;
    const int lane = i & 63, kc = (i >> 6) & 3, tap = i >> 8;
    const int co = lane & 31;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (co < R) v = *reinterpret_cast<const u32x4*>(w + ((size_t)tap * R + co) * 64 + kc * 16 + (lane >> 5) * 8);
    *reinterpret_cast<u32x4*>(wt + (size_t)i * 8) = v;
}

}  // namespace

// (internal) 1 when the request is served by conv_thin_in_kernel, given its weight image in d->wq
int dvd_conv_thin_in_ok(const dvd_conv_desc* d) {
    constexpr int use = 1;
    if (!use || !d || d->dtype != DVD_BF16 || d->C != 8 || d->ldi != 8 || d->Cout != 64 || d->ldo < 64 || (d->ldo & 7)) return 0;
    if (d->kh != 3 || d->kw != 3 || (d->kt != 1 && d->kt != 3) || d->up2 || d->res || d->ws || d->nsplit > 1) return 0;
    if (d->act != DVD_ACT_NONE && d->act != DVD_ACT_RELU) return 0;
    if (d->W != 32 && d->W != 64) return 0;
    if (d->H % (PG / d->W) || d->frames < 1 || d->T < 1) return 0;
    if (d->mask && (d->ldmask & 7)) return 0;
    return 1;
}

int dvd_conv_thin_in(const dvd_conv_desc* d, void* stream) {
    if (!dvd_conv_thin_in_ok(d) || !d->wq || !d->in || !d->out) return DVD_E_ARG;
    ThinFwdK p = {};
    p.in = (const bf16_t*)d->in; p.wt = (const bf16_t*)d->wq; p.bias = d->bias; p.mask = (const bf16_t*)d->mask; p.out = (bf16_t*)d->out;
    p.ldmask = d->ldmask; p.ldo = d->ldo; p.T = d->T; p.H = d->H; p.W = d->W; p.relu_in = d->relu_in; p.act = d->act; p.out_f32 = d->out_f32;
    p.groups = (long long)d->frames * d->T * d->H * d->W / PG;
    const unsigned grid = (unsigned)(p.groups < 2048 ? p.groups : 2048);
    if (d->kt == 3) conv_thin_in_kernel<3><<<grid, 256, 0, (hipStream_t)stream>>>(p);
    else conv_thin_in_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(p);
    return launch_status();
}

extern "C" long long dvd_conv_thin_image_bytes(int kt) { return (kt == 1 || kt == 3) ? (long long)kt * 3 * 4 * 1024 : 0; }
extern "C" int dvd_conv_thin_image(const void* w, void* wt, int kt, void* stream) {
    if (!w || !wt || (kt != 1 && kt != 3)) return DVD_E_ARG;
    const int ntr = kt * 3;
    thin_image_kernel<<<cdiv(ntr * 256, 256), 256, 0, (hipStream_t)stream>>>((const bf16_t*)w, (bf16_t*)wt, ntr);
    return launch_status();
}

// (internal) 1 when the request is served by conv_thin_out_kernel, given its weight image in d->wq
int dvd_conv_thin_out_ok(const dvd_conv_desc* d) {
    constexpr int use = 1;
    if (!use || !d || d->dtype != DVD_BF16 || d->C != 64 || d->ldi < 64 || (d->ldi & 7) || d->Cout < 1 || d->Cout > 8 || d->ldo < 8 || (d->ldo & 7)) return 0;
    if (d->kt != 1 || d->kh != 3 || d->kw != 3 || d->T != 1 || d->up2 || d->res || d->ws || d->nsplit > 1) return 0;
    if (d->act != DVD_ACT_NONE && d->act != DVD_ACT_RELU && d->act != DVD_ACT_TANH) return 0;
    if (d->W != 32 && d->W != 64) return 0;
    if (d->H % (OG / d->W) || d->frames < 1) return 0;
    if (d->mask && (d->ldmask & 7)) return 0;
    return 1;
}

int dvd_conv_thin_out(const dvd_conv_desc* d, void* stream) {
    if (!dvd_conv_thin_out_ok(d) || !d->wq || !d->in || !d->out) return DVD_E_ARG;
    ThinOutK p = {};
    p.in = (const bf16_t*)d->in; p.wt = (const bf16_t*)d->wq; p.bias = d->bias; p.mask = (const bf16_t*)d->mask; p.out = (bf16_t*)d->out;
    p.ldi = d->ldi; p.ldmask = d->ldmask; p.ldo = d->ldo; p.Cout = d->Cout; p.H = d->H; p.W = d->W; p.relu_in = d->relu_in; p.act = d->act; p.out_f32 = d->out_f32;
    p.groups = (long long)d->frames * d->H * d->W / OG;
    const unsigned grid = (unsigned)(p.groups < 512 ? p.groups : 512);
    conv_thin_out_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(p);
    return launch_status();
}

extern "C" long long dvd_conv_thin_out_image_bytes(void) { return 36 * 1024; }
extern "C" int dvd_conv_thin_out_image(const void* w, void* wt, int rows, void* stream) {
    if (!w || !wt || rows < 1 || rows > 8) return DVD_E_ARG;
    thin_out_image_kernel<<<cdiv(36 * 64, 256), 256, 0, (hipStream_t)stream>>>((const bf16_t*)w, (bf16_t*)wt, rows);
    return launch_status();
}
