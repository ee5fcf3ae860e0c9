// ConvGRU layer: the serial (time-recurrent) half of Module/ConvGRU.py:29-54 for one layer and
// all T steps, forward and BPTT.
//
// The reference evaluates three convolutions per cell-step on cat[x, h].  Here each gate conv is
// split by linearity into its x-part and its h-part:
//     conv(cat[x, h]) = conv_x(x) + conv_h(h)
// The x-parts of all three gates and all T steps do not depend on the recurrence; the caller
// computes them as ONE large batched convolution (gx[t] = Wx * x_t + b, columns u|r|o).  Only the
// h-parts remain on the serial chain and are driven from here, without returning to Python:
//     step t :  [u|r] = sigmoid(gx[t][u|r] + conv(h_{t-1}; Wh_ur))        (one N=2h conv)
//               o     = tanh   (gx[t][o]   + conv(h_{t-1}*r; Wh_o))
//               h_t   = h_{t-1}*(1-u) + o*u                                ConvGRU.py:47-52
// Small spatial sizes (4x4, 8x8) give few output tiles, so these convs run split-K and the gate
// kernels below reduce the fp32 slabs while applying the non-linearities.
// Backward walks t = T-1..0 with two backward-data convs per step; every weight gradient and the
// x-path gradient are batched over all T by the caller afterwards (dg holds d(pre-activation)).
#include "common.h"
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <queue>
#include <string>
#include <vector>

namespace {


// acc[0..7] += sum over the ns split-K slabs of 8 floats at `off` (slab s starts s * stride floats further).  The loads of four
// slabs are requested before the first is added: these kernels run a few hundred waves on mostly empty CUs and a loop of
// load -> add -> load is a chain of ns L2 / HBM round trips (16 slabs: 12-19 us per launch).  The additions keep slab order.
__device__ __forceinline__ void slab_sum8(const float* ws, int ns, size_t stride, size_t off, float (&acc)[8]) {
    int s = 0;
    for (; s + 4 <= ns; s += 4) {
        float a[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j) load8<float>(ws + (size_t)(s + j) * stride + off, a[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += a[j][k];
    }
    for (; s < ns; ++s) {
        float a[8];
        load8<float>(ws + (size_t)s * stride + off, a);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += a[k];
    }
}

// u, r, hr for one step.  ws: [ns][M][2h] fp32 partial sums of conv(h_prev; Wh_ur) (ns may be 0).
template <typename T>
__global__ void gru_gates_ur_kernel(const float* ws, int ns, const T* gx, int ldg, const T* hprev, T* u, T* r, T* hr,
                                    long long M, int h) {
    const int cg = h / 8;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * cg) return;
    const long long row = (unsigned)i / (unsigned)cg;          // M * cg < 2^31 (checked by the layer entry points)
    const int c = (int)(i - row * cg) * 8;
    float pu[8], pr[8], hp[8];
    load8<T>(gx + (size_t)row * ldg + c, pu);
    load8<T>(gx + (size_t)row * ldg + h + c, pr);
    if (hprev) load8<T>(hprev + (size_t)row * h + c, hp);
    {
        const size_t stride = (size_t)M * 2 * h, off = (size_t)row * 2 * h + c;
        int s = 0;
        for (; s + 4 <= ns; s += 4) {
            float a[4][8], b[4][8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                load8<float>(ws + (size_t)(s + j) * stride + off, a[j]);
                load8<float>(ws + (size_t)(s + j) * stride + off + h, b[j]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < 8; ++k) { pu[k] += a[j][k]; pr[k] += b[j][k]; }
        }
        for (; s < ns; ++s) {
            float a[8], b[8];
            load8<float>(ws + (size_t)s * stride + off, a);
            load8<float>(ws + (size_t)s * stride + off + h, b);
#pragma unroll
            for (int k = 0; k < 8; ++k) { pu[k] += a[k]; pr[k] += b[k]; }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        pu[k] = gate_sigmoid<T>(pu[k]);
        pr[k] = round_to<T>(gate_sigmoid<T>(pr[k]));       // the stored r is the r the cell uses
        hp[k] = hprev ? hp[k] * pr[k] : 0.f;
    }
    store8<T>(u + (size_t)row * h + c, pu);
    if (r) store8<T>(r + (size_t)row * h + c, pr);           // (not kept by an inference-mode forward)
    store8<T>(hr + (size_t)row * h + c, hp);
}

// o, h_t.  ws: [ns][M][h] partial sums of conv(h_prev*r; Wh_o).  h32p/h32n: optional fp32 carry.
template <typename T>
__global__ void gru_out_kernel(const float* ws, int ns, const T* gx, int ldg, const T* hprev, const float* h32p,
                               const T* u, T* o, T* hn, float* h32n, long long M, int h) {
    const int cg = h / 8;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * cg) return;
    const long long row = (unsigned)i / (unsigned)cg;          // M * cg < 2^31 (checked by the layer entry points)
    const int c = (int)(i - row * cg) * 8;
    float po[8], hp[8], uu[8];
    load8<T>(gx + (size_t)row * ldg + 2 * h + c, po);
    if (h32p) load8<float>(h32p + (size_t)row * h + c, hp);
    else if (hprev) load8<T>(hprev + (size_t)row * h + c, hp);
    else {
#pragma unroll
        for (int k = 0; k < 8; ++k) hp[k] = 0.f;
    }
    load8<T>(u + (size_t)row * h + c, uu);
    slab_sum8(ws, ns, (size_t)M * h, (size_t)row * h + c, po);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        po[k] = round_to<T>(gate_tanh<T>(po[k]));
        hp[k] = hp[k] * (1.f - uu[k]) + po[k] * uu[k];
    }
    if (o) store8<T>(o + (size_t)row * h + c, po);
    store8<T>(hn + (size_t)row * h + c, hp);
    if (h32n) store8<float>(h32n + (size_t)row * h + c, hp);
}

// BPTT, first half of step t: dh = dh_out[t] + carry + sum(slabs);  writes d(pre_u), d(pre_o), new carry.
template <typename T>
__global__ void gru_bwd_out_kernel(const T* dh_out, float* carry, const float* ws, int ns, const T* u, const T* o,
                                   const T* hprev, T* dg, int ldg, long long M, int h) {
    const int cg = h / 8;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * cg) return;
    const long long row = (unsigned)i / (unsigned)cg;          // M * cg < 2^31 (checked by the layer entry points)
    const int c = (int)(i - row * cg) * 8;
    const size_t off = (size_t)row * h + c;
    float dh[8], t8[8], uu[8], oo[8], hp[8], dpu[8], dpo[8];
    load8<float>(carry + off, dh);
    if (dh_out) {
        load8<T>(dh_out + off, t8);
#pragma unroll
        for (int k = 0; k < 8; ++k) dh[k] += t8[k];
    }
    load8<T>(u + off, uu);
    load8<T>(o + off, oo);
    if (hprev) load8<T>(hprev + off, hp);
    slab_sum8(ws, ns, (size_t)M * h, off, dh);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float hpk = hprev ? hp[k] : 0.f;
        dpo[k] = dh[k] * uu[k] * (1.f - oo[k] * oo[k]);
        dpu[k] = dh[k] * (oo[k] - hpk) * uu[k] * (1.f - uu[k]);
        dh[k] = dh[k] * (1.f - uu[k]);
    }
    store8<float>(carry + off, dh);
    store8<T>(dg + (size_t)row * ldg + c, dpu);
    store8<T>(dg + (size_t)row * ldg + 2 * h + c, dpo);
}

// BPTT, second half: d(h*r) = sum(slabs);  carry += d(hr)*r;  d(pre_r) = d(hr)*h_prev*r(1-r).
template <typename T>
__global__ void gru_bwd_r_kernel(float* carry, const float* ws, int ns, const T* r, const T* hprev, T* dg, int ldg,
                                 long long M, int h) {
    const int cg = h / 8;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * cg) return;
    const long long row = (unsigned)i / (unsigned)cg;          // M * cg < 2^31 (checked by the layer entry points)
    const int c = (int)(i - row * cg) * 8;
    const size_t off = (size_t)row * h + c;
    float dhr[8], t8[8], rr[8], hp[8], cy[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) dhr[k] = 0.f;
    slab_sum8(ws, ns, (size_t)M * h, off, dhr);
    if (hprev) {
        load8<T>(r + off, rr);
        load8<T>(hprev + off, hp);
        load8<float>(carry + off, cy);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            cy[k] += dhr[k] * rr[k];
            dhr[k] = dhr[k] * hp[k] * rr[k] * (1.f - rr[k]);
        }
        store8<float>(carry + off, cy);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) dhr[k] = 0.f;
    }
    store8<T>(dg + (size_t)row * ldg + h + c, dhr);
}

// out = carry + sum(slabs): gradient wrt the supplied initial hidden state
__global__ void gru_dh0_kernel(const float* carry, const float* ws, int ns, float* out, long long M, int h) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * h) return;
    float v = carry[i];
    for (int s = 0; s < ns; ++s) v += ws[(size_t)s * M * h + i];
    out[i] = v;
}

// recurrent conv whose epilogue applies the gate math directly (no split-K, no fp32 slabs)
// (nsplit > 1: in-launch split-K, the tile's last workgroup to arrive runs the gate epilogue -- g.slabs / g.tickets)
int conv_fused(int dtype, int B, int H, int W, int k, const void* in, int C, const void* w, const void* wq, int Cout, const GruEpi& g,
               int nsplit, void* stream) {
    dvd_conv_desc d = {};
    d.dtype = dtype; d.frames = B; d.T = 1; d.H = H; d.W = W; d.C = C; d.ldi = C; d.Cout = Cout; d.ldo = Cout;
    d.kt = 1; d.kh = k; d.kw = k; d.nsplit = nsplit; d.in = in; d.w = w; d.wq = wq; d.out = g.mode == 1 ? g.u : g.hn;   // (`out` itself is not written)
    return dvd_conv_forward_gru(&d, &g, stream);
}

// backward-data conv of the BPTT whose epilogue folds the result into the carry / gate gradients (modes 3, 4)
int conv_fused_bwd(int dtype, int B, int H, int W, int k, const void* in, int C, int ldi, const void* w, const void* wq, int Cout,
                   const GruEpi& g, int nsplit, void* stream) {
    dvd_conv_desc d = {};
    d.dtype = dtype; d.frames = B; d.T = 1; d.H = H; d.W = W; d.C = C; d.ldi = ldi; d.Cout = Cout; d.ldo = Cout;
    d.kt = 1; d.kh = k; d.kw = k; d.nsplit = nsplit; d.in = in; d.w = w; d.wq = wq; d.out = g.h32n;     // `out` itself is not written
    return dvd_conv_forward_gru(&d, &g, stream);
}

int conv_slabs(int dtype, int B, int H, int W, int k, const void* in, int C, int ldi, const void* w, const void* wq, int Cout,
               int nsplit, float* ws, void* stream) {
    dvd_conv_desc d = {};
    d.dtype = dtype; d.frames = B; d.T = 1; d.H = H; d.W = W; d.C = C; d.ldi = ldi; d.Cout = Cout; d.ldo = Cout;
    d.kt = 1; d.kh = k; d.kw = k; d.nsplit = nsplit; d.in = in; d.w = w; d.wq = wq; d.ws = ws;
    return dvd_conv_forward(&d, stream);
}

}  // namespace

#define S_ ((hipStream_t)stream)
#define BY_DTYPE(dtype, ...)                                                   \
    do {                                                                       \
        if ((dtype) == DVD_BF16) { using T = bf16_t; __VA_ARGS__; }            \
        else if ((dtype) == DVD_F32) { using T = float; __VA_ARGS__; }         \
        else return DVD_E_ARG;                                                 \
    } while (0)

// Split-K factor that brings a conv with few output tiles up to ~two workgroups per CU (measured on the
// full step: target 128 -> 948 ms, 256 -> 922, 384 -> 917, 512 -> 913; re-swept with the halo kernels, where a split
// shape gets the 256 x 128 tile at two workgroups per CU: 512 -> 642 ms, 768 -> 634, 1024 -> 626, 1536 -> 661).
// Per layer (tools/gru_microbench.py, round 2) the optimum depends on the filter: the 3 x 3 layers have a short K loop
// (36-72 steps), so a fused gate epilogue without split-K beats a better-filled split launch + gate kernel there
// (S = 32, h = 128: 13.3 -> 11.1 ms; S = 8 / 16, h = 256: 7.2 -> 5.9 and 11.4 -> 10.6 ms per layer); the 5 x 5 layers
// keep 1024 (S = 16, h = 512: 52.0 vs 60.2 ms at 512).
extern "C" int dvd_conv_pick_nsplit(int dtype, long long M, int Cout, int C, int ntaps) {
    const int bk = dtype == DVD_BF16 ? 32 : 16;
    const long long nk = (long long)ntaps * ((C + bk - 1) / bk);
    const long long tiles = ((M + 127) / 128) * ((Cout + 127) / 128);
    // round 3 (gate math batched in the conv epilogue, tools/gru_microbench.py sweep): a 5 x 5 conv with a short K loop
    // (C = 128: 100 steps) no longer gains from a split plus gate kernel once it has 512 tiles (S = 32, h = 128:
    // 180.5 / 208.7 -> 171.0 / 196.4 us per step forward / backward); the long loops (C = 512: 400 steps) still do
    // (round 4, weights from L2: nk = 200 -- the d[u|r] conv of gru3.l2 -- is better off unsplit too: 187.6 -> 174.2 us per step)
    const long long target = ntaps <= 9 || nk <= 200 ? 512 : 1024;
    long long ns = (target + tiles - 1) / tiles;
    // (cap: 8 since the pixel-major tile order skips the out-of-frame filter rows of the 4 x 4 convs -- their K loops are
    //  shorter, and 16 slabs of 2 MB cost more in the gate kernels than they return: 39.1 / 39.4 -> 34.3 / 34.1 us per step)
    constexpr long long cap = 8;
    if (ns > cap) ns = cap;
    // (round 4, whole-frame footprint kernel: the 3 x 3 layers on 8 x 8 frames -- 18 K steps per slice at 4 -- lose more in the gate
    //  kernel's eight slabs than the fuller launch returns: 114.6 / 111.8 -> 100.6 / 102.1 us per step for gru1.l0 / l2, forward + backward)
    if (ntaps <= 9 && M >= 4096 && ns > 4) ns = 4;
    if (ns > nk) ns = nk;
    if (ns < 1) ns = 1;
    return (int)ns;
}

// Largest split-K factor whose slices are combined inside the launch (by the tile's last workgroup to arrive) instead of by a gate
// kernel.  Measured per layer (tools/gru_microbench.py, B = 64, us per step forward / backward, same box): at 2 slices the combine is
// one 64-128 KB slab read and replaces a launch on the chain -- S = 16 layers 102.0 / 106.1 -> 84.3 / 92.1 (3 x 3), 544.5 / 549.3 ->
// 509.2 / 536.1 (5 x 5); at 4 slices it still gains a little (S = 8, 3 x 3: 50.5 / 53.3 -> 47.6 / 50.9); at 8 the serial read of eight
// slabs by ONE workgroup per tile costs more than the gate kernel, which spreads the same reads over the whole chip (S = 4:
// 33.9 / 35.9 -> 37.9 / 39.4, 67.3 / 69.6 -> 72.6 / 80.3).  Fewer, longer slices so that everything combines in-launch lose as well
// (at most 2 slices everywhere: the S = 4 / 8 layers 56.4 -> 70.4 ms over a pass pair).  dvd_gru_desc.combine_max overrides the 4.
static int inlaunch_max() {
    return 4;
}

// floats of dvd_gru_desc.ws: nsplit slabs of whole output tiles (up to 256 rows x 256 columns) for the widest of the three
// recurrent convolutions of a layer
extern "C" long long dvd_convgru_ws_floats(int dtype, int B, int H, int W, int hidden, int k) {
    const long long M = (long long)B * H * W, Mp = (M + 255) / 256 * 256;
    const int ntaps = k * k, h = hidden;
    auto need = [&](int Cout, int C) -> long long {
        return (long long)dvd_conv_pick_nsplit(dtype, M, Cout, C, ntaps) * Mp * ((Cout + 255) / 256 * 256);
    };
    long long n = need(2 * h, h);
    if (need(h, h) > n) n = need(h, h);
    if (need(h, 2 * h) > n) n = need(h, 2 * h);
    return n;
}

static int capped_nsplit(const dvd_gru_desc* d, long long M, int Cout, int C, int ntaps) {
    const int ns = dvd_conv_pick_nsplit(d->dtype, M, Cout, C, ntaps);
    return (d->ns_cap > 0 && ns > d->ns_cap) ? d->ns_cap : ns;
}

extern "C" int dvd_convgru_layer_forward(const dvd_gru_desc* d, void* stream) {
    if (!d || !d->gx || !d->w_ur || !d->w_o || !d->h_all || !d->u_all || !d->hr_all || !d->ws) return DVD_E_ARG;
    if (!d->infer && (!d->r_all || !d->o_all)) return DVD_E_ARG;
    if (d->T <= 0 || d->B <= 0 || d->hidden <= 0 || !(d->k & 1)) return DVD_E_ARG;
    if (d->hidden & 7) return DVD_E_SHAPE;
    const int h = d->hidden, ntaps = d->k * d->k;
    const long long M = (long long)d->B * d->H * d->W;
    if (M * (d->hidden / 8) >= (1ll << 31)) return DVD_E_SHAPE;
    const size_t esz = d->dtype == DVD_BF16 ? 2 : 4;
    const size_t step = (size_t)M * h * esz;
    const size_t astep = d->infer ? 0 : step;       // inference: u / h*r are one-step scratch, r and o are not stored at all
    const int ns_ur = capped_nsplit(d, M, 2 * h, h, ntaps);
    const int ns_o = capped_nsplit(d, M, h, h, ntaps);
    const int nmax = !d->tickets ? 1 : d->combine_max > 0 ? d->combine_max : inlaunch_max();    // convs with up to nmax slices apply the gates in their epilogue
    const unsigned grid = cdiv(M * (h / 8), 256);
    for (int t = 0; t < d->T; ++t) {
        const char* hprev = t > 0 ? (const char*)d->h_all + (t - 1) * step : (const char*)d->h0;
        const char* gx = (const char*)d->gx + (size_t)t * d->gx_stride * esz;
        char* u = (char*)d->u_all + t * astep; char* r = d->infer ? nullptr : (char*)d->r_all + t * step;
        char* o = d->infer ? nullptr : (char*)d->o_all + t * step; char* hr = (char*)d->hr_all + t * astep;
        char* hn = (char*)d->h_all + t * step;
        const float* h32p = (d->h32 && t > 0) ? d->h32 + (size_t)(t & 1) * M * h : nullptr;
        float* h32n = d->h32 ? d->h32 + (size_t)((t + 1) & 1) * M * h : nullptr;
        int rc, ns = 0;
        GruEpi g = {};
        g.h = h; g.ldg = 3 * h; g.gx = gx; g.hprev = hprev; g.h32p = h32p; g.u_in = u;
        g.u = u; g.r = r; g.hr = hr; g.o = o; g.hn = hn; g.h32n = h32n;
        g.slabs = d->ws; g.tickets = d->tickets;
        if (hprev && ns_ur <= nmax) {                 // gates applied in the conv epilogue (split-K: by the tile's last workgroup)
            g.mode = 1;
            rc = conv_fused(d->dtype, d->B, d->H, d->W, d->k, hprev, h, d->w_ur, d->w_ur_q, 2 * h, g, ns_ur, stream);
            if (rc) return rc;
        } else {
            if (hprev) {
                rc = conv_slabs(d->dtype, d->B, d->H, d->W, d->k, hprev, h, h, d->w_ur, d->w_ur_q, 2 * h, ns_ur, d->ws, stream);
                if (rc) return rc;
                ns = ns_ur;
            }
            BY_DTYPE(d->dtype, gru_gates_ur_kernel<T><<<grid, 256, 0, S_>>>(d->ws, ns, (const T*)gx, 3 * h,
                                                                            (const T*)hprev, (T*)u, (T*)r, (T*)hr, M, h));
        }
        ns = 0;
        if (hprev && ns_o <= nmax) {
            g.mode = 2;
            rc = conv_fused(d->dtype, d->B, d->H, d->W, d->k, hr, h, d->w_o, d->w_o_q, h, g, ns_o, stream);
            if (rc) return rc;
        } else {
            if (hprev) {
                rc = conv_slabs(d->dtype, d->B, d->H, d->W, d->k, hr, h, h, d->w_o, d->w_o_q, h, ns_o, d->ws, stream);
                if (rc) return rc;
                ns = ns_o;
            }
            BY_DTYPE(d->dtype, gru_out_kernel<T><<<grid, 256, 0, S_>>>(d->ws, ns, (const T*)gx, 3 * h, (const T*)hprev,
                                                                       h32p, (const T*)u, (T*)o, (T*)hn, h32n, M, h));
        }
    }
    return launch_status();
}

extern "C" int dvd_convgru_layer_backward(const dvd_gru_desc* d, void* stream) {
    if (!d || !d->wd_ur || !d->wd_o || !d->h_all || !d->u_all || !d->r_all || !d->o_all || !d->dg || !d->carry || !d->ws)
        return DVD_E_ARG;
    if (d->T <= 0 || d->B <= 0 || d->hidden <= 0 || !(d->k & 1)) return DVD_E_ARG;
    if (d->hidden & 7) return DVD_E_SHAPE;
    const int h = d->hidden, ntaps = d->k * d->k;
    const long long M = (long long)d->B * d->H * d->W;
    if (M * (d->hidden / 8) >= (1ll << 31)) return DVD_E_SHAPE;
    const size_t esz = d->dtype == DVD_BF16 ? 2 : 4;
    const size_t step = (size_t)M * h * esz;
    const int ns_o = capped_nsplit(d, M, h, h, ntaps);        // d(hr)  = convT(d pre_o)
    const int ns_ur = capped_nsplit(d, M, h, 2 * h, ntaps);   // dh    += convT(d pre_u | d pre_r)
    const int nmax = !d->tickets ? 1 : d->combine_max > 0 ? d->combine_max : inlaunch_max();
    const unsigned grid = cdiv(M * (h / 8), 256);
    hipError_t e = hipMemsetAsync(d->carry, 0, (size_t)M * h * sizeof(float), S_);
    if (e != hipSuccess) return DVD_E_LAUNCH;
    int ns_pending = 0;      // slabs of the ur backward-data conv of step t+1 waiting in ws
    bool out_done = false;   // first half of this step already applied by the previous step's conv epilogue (mode 5)
    for (int t = d->T - 1; t >= 0; --t) {
        const char* hprev = t > 0 ? (const char*)d->h_all + (t - 1) * step : (const char*)d->h0;
        const char* u = (const char*)d->u_all + t * step; const char* r = (const char*)d->r_all + t * step;
        const char* o = (const char*)d->o_all + t * step;
        const char* dho = d->dh_out ? (const char*)d->dh_out + t * step : nullptr;
        char* dg = (char*)d->dg + (size_t)t * M * 3 * h * esz;
        if (!out_done)
            BY_DTYPE(d->dtype, gru_bwd_out_kernel<T><<<grid, 256, 0, S_>>>((const T*)dho, d->carry, d->ws, ns_pending,
                                                                           (const T*)u, (const T*)o, (const T*)hprev,
                                                                           (T*)dg, 3 * h, M, h));
        ns_pending = 0;
        out_done = false;
        int ns = 0, rc;
        GruEpi g = {};
        g.h = h; g.ldg = 3 * h; g.r = const_cast<char*>(r); g.hprev = hprev; g.h32n = d->carry; g.o = dg;
        g.slabs = d->ws; g.tickets = d->tickets;
        if (hprev && ns_o <= nmax) {                  // d(h*r) conv applies the reset-gate step in its epilogue
            g.mode = 3;
            rc = conv_fused_bwd(d->dtype, d->B, d->H, d->W, d->k, dg + (size_t)2 * h * esz, h, 3 * h, d->wd_o, d->wd_o_q, h, g, ns_o,
                                stream);
            if (rc) return rc;
        } else {
            if (hprev) {
                rc = conv_slabs(d->dtype, d->B, d->H, d->W, d->k, dg + (size_t)2 * h * esz, h, 3 * h, d->wd_o, d->wd_o_q, h, ns_o, d->ws,
                                stream);
                if (rc) return rc;
                ns = ns_o;
            }
            BY_DTYPE(d->dtype, gru_bwd_r_kernel<T><<<grid, 256, 0, S_>>>(d->carry, d->ws, ns, (const T*)r, (const T*)hprev,
                                                                         (T*)dg, 3 * h, M, h));
        }
        if (hprev) {
            if (ns_ur <= nmax) {             // [u|r] backward-data conv adds straight into the carry and goes on with
                g.mode = 4;                  // the first half of step t-1 (mode 5) unless this is step 0 with an h0
                if (t > 0) {
                    const size_t tp = (size_t)(t - 1);
                    g.mode = 5;
                    g.gx = d->dh_out ? (const char*)d->dh_out + tp * step : nullptr;
                    g.u_in = (const char*)d->u_all + tp * step;
                    g.hr = const_cast<char*>((const char*)d->o_all + tp * step);
                    g.hprev = t - 1 > 0 ? (const char*)d->h_all + (tp - 1) * step : (const char*)d->h0;
                    g.o = (char*)d->dg + tp * M * 3 * h * esz;
                    out_done = true;
                }
                rc = conv_fused_bwd(d->dtype, d->B, d->H, d->W, d->k, dg, 2 * h, 3 * h, d->wd_ur, d->wd_ur_q, h, g, ns_ur, stream);
                if (rc) return rc;
            } else {
                rc = conv_slabs(d->dtype, d->B, d->H, d->W, d->k, dg, 2 * h, 3 * h, d->wd_ur, d->wd_ur_q, h, ns_ur, d->ws, stream);
                if (rc) return rc;
                ns_pending = ns_ur;
            }
        }
    }
    if (d->dh0) gru_dh0_kernel<<<cdiv(M * h, 256), 256, 0, S_>>>(d->carry, d->ws, ns_pending, d->dh0, M, h);
    return launch_status();
}

// ============================================================================ layer wavefront over a ConvGRU stack (round 5)
// See include/dvdgan_hip.h (dvd_gru_stack_desc).  Launch pair k of the forward pass:
//   U group:  [u|r] convolution of (layer l, step t = k - 2 l) for every layer with 0 <= t < T
//   O group:  out-gate convolution of the same (l, t), plus the x-part convolution of (layer l >= 1, step k - 2 l + 1) -- its input,
//             layer l-1's state of that step, was finished by the O group of pair k - 1; its result is read by the U group of pair k + 1
// and of the backward pass (layers in reverse, layer l works on step t = T - 1 - (k - 2 (L - 1 - l))):
//   A group:  d(h*r) convolution (epilogue: reset-gate step)
//   B group:  d[u|r] convolution (epilogue: carry + first half of step t - 1), plus for l >= 1 the x-part backward-data convolution of
//             step t (all three gate gradients of the step are complete), whose result is layer l-1's dh_out of step t -- needed by
//             layer l-1's B convolution of step t + 1 in pair k + 1.
// The steps without a previous state (t = 0, no h0) and the first BPTT step of a layer run the elementwise gate kernels.
#ifndef DVD_STACK_SEARCH_KINDS
#define DVD_STACK_SEARCH_KINDS(kind) 1
#endif
namespace {

struct Member { dvd_conv_desc d; long long tiles; int kchunks; int gate; };
// debug bookkeeping behind dvd_debug_stack_ws (tests only): the last sizing answer and the largest slab cursor a LAUNCHED group used
std::atomic<long long> g_ws_sized{0}, g_ws_high{0};
constexpr int kMaxMember = 2 * DVD_GRU_STACK_MAX;

// kernel family serving a stack: 0 = frames >= 16 pixels (256 x 128 tiles), 2 / 3 = 8 x 8 frames (256- / 128-row tiles), 4 = 4 x 4
int stack_kind(const dvd_gru_stack_desc* s) {
    const dvd_gru_desc& a = s->layer[0];
    if (a.H != a.W || ilog2_exact(a.H) < 0) return -1;
    if (a.H >= 16) return 0;                         // (128 x 128 tiles, three workgroups per CU: 4-8 % slower on the 16 x 16 / 32 x 32 stages)
    if (a.H == 4) return 4;
    if (a.H != 8) return -1;
    long long t256 = 0;                              // tiles of the U group on 256-row tiles
    for (int l = 0; l < s->n_layers; ++l) t256 += (long long)cdiv(a.B, 4) * cdiv(2 * s->layer[l].hidden, 128);
    return t256 >= 256 ? 2 : 3;
}
long long kind_mtiles(int kind, int B, int H, int W) {
    const long long M = (long long)B * H * W;
    return kind == 0 ? cdiv(M, 256) : kind == 1 ? cdiv(M, 128) : cdiv(B, kind == 2 ? 4 : kind == 3 ? 2 : 8);
}
long long kind_tile_floats(int kind) { return (kind == 0 || kind == 2) ? 32768 : 16384; }     // accumulators of one output tile

int stack_check(const dvd_gru_stack_desc* s, bool backward) {
    if (!s || s->n_layers < 1 || s->n_layers > DVD_GRU_STACK_MAX) return DVD_E_ARG;
    const dvd_gru_desc& a = s->layer[0];
    if (a.dtype != DVD_BF16 || a.T <= 0 || a.B <= 0 || !a.tickets || !s->ws) return DVD_E_ARG;
    if (stack_kind(s) < 0) return DVD_E_SHAPE;
    for (int l = 0; l < s->n_layers; ++l) {
        const dvd_gru_desc& d = s->layer[l];
        if (d.dtype != a.dtype || d.T != a.T || d.B != a.B || d.H != a.H || d.W != a.W) return DVD_E_ARG;
        if (d.hidden <= 0 || (d.hidden & 7) || (d.k != 3 && d.k != 5)) return DVD_E_SHAPE;
        if ((long long)d.B * d.H * d.W * (d.hidden / 8) >= (1ll << 31)) return DVD_E_SHAPE;
        if (!d.gx || !d.h_all || !d.u_all || !d.hr_all) return DVD_E_ARG;
        if (l > 0 && (s->cin[l] != s->layer[l - 1].hidden || d.gx_stride != (long long)d.B * d.H * d.W * 3 * d.hidden)) return DVD_E_ARG;
        if (!backward) {
            if (!d.w_ur || !d.w_o || !d.w_ur_q || !d.w_o_q) return DVD_E_ARG;
            if (!d.infer && (!d.r_all || !d.o_all)) return DVD_E_ARG;
            if (l > 0 && (!s->wx[l] || !s->wx_q[l] || !s->bx[l])) return DVD_E_ARG;
        } else {
            if (!d.wd_ur || !d.wd_o || !d.wd_ur_q || !d.wd_o_q || !d.r_all || !d.o_all || !d.dg || !d.carry) return DVD_E_ARG;
            if (l > 0 && (!s->wdx[l] || !s->wdx_q[l] || !s->dh_mid[l])) return DVD_E_ARG;
        }
    }
    return DVD_OK;
}

void member_conv(Member& m, const dvd_gru_desc& L, const void* in, int C, int ldi, const void* w, const void* wq, int Cout, int kind) {
    m = Member{};
    dvd_conv_desc& d = m.d;
    d.dtype = L.dtype; d.frames = L.B; d.T = 1; d.H = L.H; d.W = L.W; d.C = C; d.ldi = ldi; d.Cout = Cout; d.ldo = Cout;
    d.kt = 1; d.kh = L.k; d.kw = L.k; d.nsplit = 1; d.in = in; d.w = w; d.wq = wq; d.wq_kind = 1;
    m.tiles = kind_mtiles(kind, L.B, L.H, L.W) * cdiv(Cout, 128);
    m.kchunks = (C + 31) / 32;
}

// ---- split-K factors of a grouped launch: a search over per-member factors against a model of the launch.
// The members of a group differ 5x in K length (3 x 3 on 256 channels beside 5 x 5 on 768): one factor for all of them -- the
// first round-5 policy, "fill 512 workgroups" -- splits short tiles for nothing and leaves the long tiles of the 5 x 5 layer as
// the launch's makespan.  Model: an XCD runs its share of the workgroups on `slots` concurrent slots in the
// order group_dispatch issues them (members by decreasing length); a workgroup costs its K steps plus a fixed prologue / epilogue
// (in K steps), a split tile a little more (slab traffic, the combine by its last workgroup).  Plans are cached per signature.
struct SplitPlan { int ns[2 * DVD_GRU_STACK_MAX]; };
double model_makespan(int n, const long long* tiles, const int* units, const int* ns, int slots, double fixed, double split_cost) {
    int order[2 * DVD_GRU_STACK_MAX];
    double len[2 * DVD_GRU_STACK_MAX];
    for (int i = 0; i < n; ++i) { order[i] = i; len[i] = (double)units[i] / ns[i] + fixed + (ns[i] > 1 ? split_cost : 0.0); }
    std::sort(order, order + n, [&](int a, int b) { return (double)units[a] / ns[a] > (double)units[b] / ns[b]; });
    std::priority_queue<double, std::vector<double>, std::greater<double>> q;
    for (int i = 0; i < slots; ++i) q.push(0.0);
    double end = 0.0;
    for (int oi = 0; oi < n; ++oi) {
        const int i = order[oi];
        const long long w = (tiles[i] * ns[i] + 7) / 8;              // this XCD's workgroups of member i
        for (long long k = 0; k < w; ++k) {
            const double t = q.top() + len[i];
            q.pop(); q.push(t);
            if (t > end) end = t;
        }
    }
    return end;
}
SplitPlan plan_splits(int kind, bool backward, int n, const Member* m, long long cap) {
    static std::map<std::string, SplitPlan> cache;
    static std::mutex mu;
    long long tiles[2 * DVD_GRU_STACK_MAX]; int units[2 * DVD_GRU_STACK_MAX], capi[2 * DVD_GRU_STACK_MAX];
    std::string key;
    key.append((const char*)&kind, sizeof kind); key.push_back(backward ? 1 : 0);
    for (int i = 0; i < n; ++i) {
        tiles[i] = m[i].tiles; units[i] = m[i].kchunks * m[i].d.kh * m[i].d.kw;
        capi[i] = (int)std::min<long long>(cap, m[i].kchunks);
        if (m[i].tiles > 1024) capi[i] = 1;                               // one ticket per tile, 1024 tickets per member
        key.append((const char*)&tiles[i], sizeof tiles[i]); key.append((const char*)&units[i], sizeof units[i]);
        key.append((const char*)&capi[i], sizeof capi[i]);
    }
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    const int slots = (kind == 0 || kind == 2) ? 64 : 96;
    // fitted on tools/gru_microbench.py stack (B = 64; K steps of the launch's tile shape): a split tile is cheap on 4 x 4 frames
    // (64 KB slabs, few tiles) and dear on 256-row tiles (128 KB slabs; the tile's last workgroup combines, then runs the epilogue):
    // 8 x 8 forward 13.85 -> 12.87 ms with only the 5 x 5 layer split, 4 x 4 5.39 / 6.09 -> 5.03 / 5.56, 16 x 16 backward 46.1 -> 45.0
    double fixed = (kind == 0 || kind == 2) ? 12.0 : 16.0;
    double split_cost = kind == 4 ? 3.0 : kind == 0 ? 150.0 : backward ? 50.0 : 80.0;
#ifdef DVD_STACK_SEARCH_DEBUG
    if (const char* e = getenv("DVD_SS_FIXED")) fixed = atof(e);
    if (const char* e = getenv("DVD_SS_SPLIT")) split_cost = atof(e);
#endif
    static const int opts[] = {1, 2, 3, 4, 8};
    SplitPlan best{}; double best_t = 1e30;
    int idx[2 * DVD_GRU_STACK_MAX] = {0};
    if (n > 5) {
        // 5^n combinations stop being "a few ms once" beyond five members (four-layer stacks: 7-8 members = 78 k - 390 k makespan
        // simulations under the lock): coordinate descent from the unsplit plan -- one member's factor at a time, until no change helps
        int ns[2 * DVD_GRU_STACK_MAX];
        for (int i = 0; i < n; ++i) ns[i] = 1;
        auto cost = [&]() { int k = 0; for (int i = 0; i < n; ++i) k += ns[i] > 1;
                            return model_makespan(n, tiles, units, ns, slots, fixed, split_cost) * (1.0 + 0.004 * k); };
        best_t = cost();
        for (bool moved = true; moved;) {
            moved = false;
            for (int i = 0; i < n; ++i) {
                const int keep = ns[i]; int pick = keep;
                for (int o : opts) {
                    if (o > capi[i] || o == keep) continue;
                    ns[i] = o;
                    const double t = cost();
                    if (t < best_t * (1.0 - 1e-9)) { best_t = t; pick = o; moved = true; }
                }
                ns[i] = pick;
            }
        }
        for (int i = 0; i < n; ++i) best.ns[i] = ns[i];
    } else
    for (;;) {
        int ns[2 * DVD_GRU_STACK_MAX]; bool ok = true; int nsplit_members = 0;
        for (int i = 0; i < n; ++i) { ns[i] = opts[idx[i]]; if (ns[i] > capi[i]) ok = false; nsplit_members += ns[i] > 1; }
        if (ok) {
            const double t = model_makespan(n, tiles, units, ns, slots, fixed, split_cost) * (1.0 + 0.004 * nsplit_members);
            if (t < best_t) { best_t = t; for (int i = 0; i < n; ++i) best.ns[i] = ns[i]; }
        }
        int j = 0;
        while (j < n && ++idx[j] == (int)(sizeof opts / sizeof opts[0])) idx[j++] = 0;
        if (j == n) break;
    }
#ifdef DVD_STACK_SEARCH_DEBUG
    {
        int one[2 * DVD_GRU_STACK_MAX]; for (int i = 0; i < n; ++i) one[i] = 1;
        fprintf(stderr, "plan kind %d %s:", kind, backward ? "bwd" : "fwd");
        for (int i = 0; i < n; ++i) fprintf(stderr, " [tiles %lld units %d cap %d -> ns %d]", tiles[i], units[i], capi[i], best.ns[i]);
        fprintf(stderr, "  model %.0f (unsplit %.0f)\n", best_t, model_makespan(n, tiles, units, one, slots, fixed, split_cost));
    }
#endif
    cache[key] = best;
    return best;
}

// Split-K factors of one grouped launch and the slab space behind them; launches unless `dry`.
int run_group(const dvd_gru_stack_desc* s, int kind, Member* m, GruEpi* g, int n, void* stream, bool dry, long long& ws_need, bool backward) {
    if (n == 0) return DVD_OK;
    long long total = 0;
    for (int i = 0; i < n; ++i) total += m[i].tiles;
    // measured (tools/gru_microbench.py stack, B = 64): (target, cap) = (768, 4) 5.46 / 6.99 ms forward / backward on 4 x 4 frames,
    // (768, 8) 5.66 / 6.16; 8 x 8 frames: (768, 4) 14.86 / 15.09, (768, 8) 14.93 / 15.95, (384, 4) 14.46 / 16.53, (1536, *) slower
    const long long cap = (kind == 4 && backward) ? 8 : 4;
    const long long target = (kind == 0 || kind == 2) ? 512 : 768;
    long long want = (target + total - 1) / total;
    if (want > cap) want = cap;
    SplitPlan plan{};
    bool searched = !s->layer_policy && DVD_STACK_SEARCH_KINDS(kind);
#ifdef DVD_STACK_SEARCH_DEBUG
    if (const char* e = getenv("DVD_SS_MASK")) searched = searched && ((atoi(e) >> kind) & 1) && ((atoi(e) >> (backward ? 9 : 8)) & 1);
#endif
    if (searched) plan = plan_splits(kind, backward, n, m, cap);
    long long cursor = 0;
    dvd_conv_desc d[kMaxMember];
    for (int i = 0; i < n; ++i) {
        long long ns = searched ? plan.ns[i] : want;
        if (s->layer_policy)
            ns = m[i].gate ? dvd_conv_pick_nsplit(DVD_BF16, (long long)m[i].d.frames * m[i].d.H * m[i].d.W, m[i].d.Cout, m[i].d.C,
                                                  m[i].d.kh * m[i].d.kw) : 1;
        if (s->layer[0].ns_cap > 0 && ns > s->layer[0].ns_cap) ns = s->layer[0].ns_cap;
        if (ns > m[i].kchunks) ns = m[i].kchunks;
        if (ns > 1 && m[i].tiles > 1024) ns = 1;
        m[i].d.nsplit = (int)ns;
        if (ns > 1) {
            if (g[i].mode == 0) g[i].mode = 6;            // direct epilogue behind the in-launch combine
            g[i].slabs = s->ws + cursor; g[i].tickets = s->layer[0].tickets;
            cursor += ns * m[i].tiles * kind_tile_floats(kind);
        }
        d[i] = m[i].d;
    }
    if (dry) {
        if (cursor > ws_need) ws_need = cursor;
        return DVD_OK;
    }
    for (long long seen = g_ws_high.load(); cursor > seen && !g_ws_high.compare_exchange_weak(seen, cursor);) {}
    if (cursor > ws_need) return DVD_E_SHAPE;      // (cannot happen: the launch walks the schedule the sizing query walked -- but a slab overrun is a memory fault)
    return dvd_conv_forward_group(d, g, n, kind, s->run, stream);
}

int stack_forward(const dvd_gru_stack_desc* s, void* stream, bool dry, long long& ws_need) {
    const int L = s->n_layers, T = s->layer[0].T, kind = stack_kind(s);
    const long long M = (long long)s->layer[0].B * s->layer[0].H * s->layer[0].W;
    const size_t esz = 2;
    for (int k = 0; k < T + 2 * (L - 1); ++k) {
        Member mem[kMaxMember];
        GruEpi epi[kMaxMember];
        for (int phase = 0; phase < 2; ++phase) {                 // 0 = U group, 1 = O group
            int n = 0;
            for (int l = 0; l < L; ++l) {
                const dvd_gru_desc& d = s->layer[l];
                const int t = k - 2 * l, h = d.hidden;
                if (t < 0 || t >= T) continue;
                const size_t step = (size_t)M * h * esz, astep = d.infer ? 0 : step;
                const char* hprev = t > 0 ? (const char*)d.h_all + (t - 1) * step : (const char*)d.h0;
                const char* gx = (const char*)d.gx + (size_t)t * d.gx_stride * esz;
                char* u = (char*)d.u_all + t * astep; char* r = d.infer ? nullptr : (char*)d.r_all + t * step;
                char* o = d.infer ? nullptr : (char*)d.o_all + t * step; char* hr = (char*)d.hr_all + t * astep;
                char* hn = (char*)d.h_all + t * step;
                const float* h32p = (d.h32 && t > 0) ? d.h32 + (size_t)(t & 1) * M * h : nullptr;
                float* h32n = d.h32 ? d.h32 + (size_t)((t + 1) & 1) * M * h : nullptr;
                const unsigned grid = cdiv(M * (h / 8), 256);
                const bool has_prev = t > 0 || d.h0 != nullptr;             // (the dry run sizes the workspace for exactly this schedule)
                if (!has_prev) {                                  // step 0 without a supplied state: gates of the x-part alone
                    if (dry) continue;
                    using T_ = bf16_t;
                    if (phase == 0)
                        gru_gates_ur_kernel<T_><<<grid, 256, 0, S_>>>(nullptr, 0, (const T_*)gx, 3 * h, (const T_*)nullptr, (T_*)u, (T_*)r, (T_*)hr, M, h);
                    else
                        gru_out_kernel<T_><<<grid, 256, 0, S_>>>(nullptr, 0, (const T_*)gx, 3 * h, (const T_*)nullptr, h32p, (const T_*)u, (T_*)o,
                                                                 (T_*)hn, h32n, M, h);
                    continue;
                }
                GruEpi& g = epi[n];
                g = GruEpi{};
                g.h = h; g.ldg = 3 * h; g.gx = gx; g.hprev = hprev; g.h32p = h32p; g.u_in = u;
                g.u = u; g.r = r; g.hr = hr; g.o = o; g.hn = hn; g.h32n = h32n;
                if (phase == 0) {
                    g.mode = 1;
                    member_conv(mem[n], d, hprev, h, h, d.w_ur, d.w_ur_q, 2 * h, kind);
                    mem[n].d.out = u;
                } else {
                    g.mode = 2;
                    member_conv(mem[n], d, hr, h, h, d.w_o, d.w_o_q, h, kind);
                    mem[n].d.out = hn;
                }
                mem[n].gate = 1;
                ++n;
            }
            // bit l: the x-part of layer l rides in the U group instead of the O group (both are behind its producer).  Measured
            // (tools/gru_microbench.py stack): no difference beyond noise except on 8 x 8 frames, where the top layer's x-part in the
            // U group balances the two launches of a pair (14.63 -> 13.93 ms forward, 15.77 -> 14.76 backward)
            const int x_in_u = s->layer[0].H == 8 ? 4 : 0;
            for (int l = 1; l < L; ++l)
                if (phase == (((x_in_u >> l) & 1) ? 0 : 1)) {     // x-part of layer l for step k - 2 l + 1
                    const dvd_gru_desc& d = s->layer[l];
                    const dvd_gru_desc& b = s->layer[l - 1];
                    const int t = k - 2 * l + 1;
                    if (t < 0 || t >= T) continue;
                    epi[n] = GruEpi{};
                    member_conv(mem[n], d, (const char*)b.h_all + (size_t)t * M * b.hidden * esz, b.hidden, b.hidden, s->wx[l], s->wx_q[l],
                                3 * d.hidden, kind);
                    mem[n].d.bias = s->bx[l];
                    mem[n].d.out = (char*)d.gx + (size_t)t * d.gx_stride * esz;
                    ++n;
                }
            const int rc = run_group(s, kind, mem, epi, n, stream, dry, ws_need, false);
            if (rc) return rc;
        }
    }
    return dry ? DVD_OK : launch_status();
}

int stack_backward(const dvd_gru_stack_desc* s, void* stream, bool dry, long long& ws_need) {
    const int L = s->n_layers, T = s->layer[0].T, kind_ = stack_kind(s);
    const long long M = (long long)s->layer[0].B * s->layer[0].H * s->layer[0].W;
    const size_t esz = 2;
    using T_ = bf16_t;
    const int dx_in_a = s->layer[0].H == 8 ? 4 : 0;       // bit l: layer l's x-part backward-data rides in the NEXT pair's A group (see x_in_u)
    if (!dry)
        for (int l = 0; l < L; ++l)
            if (hipMemsetAsync(s->layer[l].carry, 0, (size_t)M * s->layer[l].hidden * sizeof(float), S_) != hipSuccess) return DVD_E_LAUNCH;
    // gradient wrt layer l's state of step t: from outside the stack (top layer) or from the x-path of the layer above
    auto dh_of = [&](int l, int t) -> const char* {
        const size_t step = (size_t)M * s->layer[l].hidden * esz;
        if (l == L - 1) return s->layer[l].dh_out ? (const char*)s->layer[l].dh_out + t * step : nullptr;
        return (const char*)s->dh_mid[l + 1] + t * step;
    };
    for (int k = 0; k < T + 2 * (L - 1); ++k) {
        Member mem[kMaxMember];
        GruEpi epi[kMaxMember];
        for (int phase = 0; phase < 2; ++phase) {                 // 0 = A group, 1 = B group
            int n = 0;
#ifndef DVD_A_GROUP_KIND                  // experiment: 1 = the A group of the >= 16-pixel stages on 128-row tiles (three workgroups per CU)
#define DVD_A_GROUP_KIND 0
#endif
            const int kind = (phase == 0 && kind_ == 0 && DVD_A_GROUP_KIND) ? 1 : kind_;
            for (int l = L - 1; l >= 0; --l) {
                const dvd_gru_desc& d = s->layer[l];
                const int t = T - 1 - (k - 2 * (L - 1 - l)), h = d.hidden;
                if (phase == 0 && l > 0 && ((dx_in_a >> l) & 1) && t >= -1 && t + 1 < T) {     // x-part backward-data of the step finished in the previous pair
                    const int ci = s->cin[l];
                    char* dgp = (char*)d.dg + (size_t)(t + 1) * M * 3 * h * esz;
                    epi[n] = GruEpi{};
                    member_conv(mem[n], d, dgp, 3 * h, 3 * h, s->wdx[l], s->wdx_q[l], ci, kind);
                    mem[n].d.out = (char*)s->dh_mid[l] + (size_t)(t + 1) * M * ci * esz;
                    mem[n].d.ldo = ci;
                    if (s->layer[l - 1].dh_out) {
                        mem[n].d.res = (const char*)s->layer[l - 1].dh_out + (size_t)(t + 1) * M * ci * esz;
                        mem[n].d.ldres = ci;
                    }
                    ++n;
                }
                if (t < 0 || t >= T) continue;
                const size_t step = (size_t)M * h * esz;
                const char* hprev = t > 0 ? (const char*)d.h_all + (t - 1) * step : (const char*)d.h0;
                const char* u = (const char*)d.u_all + t * step; const char* r = (const char*)d.r_all + t * step;
                const char* o = (const char*)d.o_all + t * step;
                char* dg = (char*)d.dg + (size_t)t * M * 3 * h * esz;
                const unsigned grid = cdiv(M * (h / 8), 256);
                const bool has_prev = t > 0 || d.h0 != nullptr;
                if (phase == 0) {
                    if (t == T - 1 && !dry)                       // first BPTT step of the layer: nothing upstream to ride on
                        gru_bwd_out_kernel<T_><<<grid, 256, 0, S_>>>((const T_*)dh_of(l, t), d.carry, nullptr, 0, (const T_*)u, (const T_*)o,
                                                                     (const T_*)hprev, (T_*)dg, 3 * h, M, h);
                    if (!has_prev) {
                        if (!dry) gru_bwd_r_kernel<T_><<<grid, 256, 0, S_>>>(d.carry, nullptr, 0, (const T_*)r, (const T_*)nullptr, (T_*)dg, 3 * h, M, h);
                        continue;
                    }
                    GruEpi& g = epi[n];
                    g = GruEpi{};
                    g.mode = 3; g.h = h; g.ldg = 3 * h; g.r = const_cast<char*>(r); g.hprev = hprev; g.h32n = d.carry; g.o = dg;
                    member_conv(mem[n], d, dg + (size_t)2 * h * esz, h, 3 * h, d.wd_o, d.wd_o_q, h, kind);
                    mem[n].d.out = d.carry; mem[n].gate = 1;
                    ++n;
                } else {
                    if (has_prev) {
                        GruEpi& g = epi[n];
                        g = GruEpi{};
                        g.h = h; g.ldg = 3 * h; g.h32n = d.carry;
                        g.mode = 4;
                        if (t > 0) {                              // ... and the first half of step t - 1
                            const size_t tp = (size_t)(t - 1);
                            g.mode = 5;
                            g.gx = dh_of(l, t - 1);
                            g.u_in = (const char*)d.u_all + tp * step;
                            g.hr = const_cast<char*>((const char*)d.o_all + tp * step);
                            g.hprev = t - 1 > 0 ? (const char*)d.h_all + (tp - 1) * step : (const char*)d.h0;
                            g.o = (char*)d.dg + tp * M * 3 * h * esz;
                        }
                        member_conv(mem[n], d, dg, 2 * h, 3 * h, d.wd_ur, d.wd_ur_q, h, kind);
                        mem[n].d.out = d.carry; mem[n].gate = 1;
                        ++n;
                    }
                    if (l > 0 && !((dx_in_a >> l) & 1)) {         // x-part backward-data: gradient reaching layer l-1's state of step t
                        const int ci = s->cin[l];
                        epi[n] = GruEpi{};
                        member_conv(mem[n], d, dg, 3 * h, 3 * h, s->wdx[l], s->wdx_q[l], ci, kind);
                        mem[n].d.out = (char*)s->dh_mid[l] + (size_t)t * M * ci * esz;
                        mem[n].d.ldo = ci;
                        if (s->layer[l - 1].dh_out) {
                            mem[n].d.res = (const char*)s->layer[l - 1].dh_out + (size_t)t * M * ci * esz;
                            mem[n].d.ldres = ci;
                        }
                        ++n;
                    }
                }
            }
            const int rc = run_group(s, kind, mem, epi, n, stream, dry, ws_need, true);
            if (rc) return rc;
        }
    }
    if (!dry)
        for (int l = 0; l < L; ++l) {
            const dvd_gru_desc& d = s->layer[l];
            if (d.dh0) gru_dh0_kernel<<<cdiv(M * d.hidden, 256), 256, 0, S_>>>(d.carry, nullptr, 0, d.dh0, M, d.hidden);
        }
    return dry ? DVD_OK : launch_status();
}

}  // namespace

extern "C" int dvd_convgru_stack_ok(const dvd_gru_stack_desc* d, int backward) {
    dvd_gru_stack_desc t;
    if (!d) return 0;
    t = *d;
    static float dummy;
    if (!t.ws) t.ws = &dummy;                                     // (a geometry / pointer-completeness query: the workspace may not exist yet)
    return stack_check(&t, backward != 0) == DVD_OK ? 1 : 0;
}
extern "C" long long dvd_convgru_stack_ws_floats(const dvd_gru_stack_desc* d) {
    if (!d || d->n_layers < 1 || d->n_layers > DVD_GRU_STACK_MAX || stack_kind(d) < 0) return 0;
    long long need = 0, nb = 0;
    stack_forward(d, nullptr, true, need);
    stack_backward(d, nullptr, true, nb);
    if (nb > need) need = nb;
    g_ws_sized.store(need);
    return need > 0 ? need : 1;
}
// Test hook: out[0] = floats the last dvd_convgru_stack_ws_floats call asked for (0 = no member of any group is split),
// out[1] = the largest slab cursor any grouped launch has used since the last reset.  reset != 0 clears out[1] afterwards.
extern "C" void dvd_debug_stack_ws(long long* out, int reset) {
    if (out) { out[0] = g_ws_sized.load(); out[1] = g_ws_high.load(); }
    if (reset) g_ws_high.store(0);
}
extern "C" int dvd_convgru_stack_forward(const dvd_gru_stack_desc* d, void* stream) {
    const int rc = stack_check(d, false);
    if (rc) return rc;
    long long need = dvd_convgru_stack_ws_floats(d);      // what the caller was told to allocate: the launch refuses to go past it
    return stack_forward(d, stream, false, need);
}
extern "C" int dvd_convgru_stack_backward(const dvd_gru_stack_desc* d, void* stream) {
    const int rc = stack_check(d, true);
    if (rc) return rc;
    long long need = dvd_convgru_stack_ws_floats(d);
    return stack_backward(d, stream, false, need);
}
