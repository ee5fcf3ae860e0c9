// Memory-bound kernels around the convolutions: batch-norm statistics, conditional batch norm
// apply / backward, pooling / nearest resampling, bias-gradient column sums, small elementwise
// helpers.  All of them move 16-byte (bf16) / 32-byte (f32) channel groups per thread and are
// bounded by HBM bandwidth; statistics accumulate in fp64.
#include "common.h"
#include <cstdlib>

namespace {

// ----------------------------------------------------------------------------- BN statistics
// sums[rep][0..C) += sum_rows x, sums[rep][C..2C) += sum_rows x^2, rep = block % DVD_BN_NREP   (fp64 atomics, caller zeroes).
// The atomics of all blocks on ONE copy serialise at ~0.17 us each per address (512 blocks: 87 us of a 126 us launch on
// 3072 x 8 x 8 x 512); DVD_BN_NREP copies, added up by bn_finalize_kernel, divide that tail by DVD_BN_NREP.
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, long long rows, int C, int ld, double* sums) {
    __shared__ double red[256][17];
    const int cg = (C + 7) / 8, nj = 256 / cg;
    const int g = threadIdx.x % cg, j = threadIdx.x / cg;
    double s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = q[i] = 0.0;
    if (j < nj) {
        constexpr int U = 4;                                  // independent 16-byte loads in flight per thread
        const long long stride = (long long)gridDim.x * nj;
        for (long long r = (long long)blockIdx.x * nj + j; r < rows; r += stride * U) {
            float v[U][8];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long ru = r + u * stride;
                if (ru < rows) load8<T>(x + (size_t)ru * ld + g * 8, v[u]);
                else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[u][i] = 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) { s[i] += v[u][i]; q[i] += (double)v[u][i] * v[u][i]; }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[threadIdx.x][i] = s[i]; red[threadIdx.x][8 + i] = q[i]; }
    __syncthreads();
    if (j == 0) {
        for (int jj = 1; jj < nj; ++jj)
#pragma unroll
            for (int i = 0; i < 8; ++i) { s[i] += red[jj * cg + g][i]; q[i] += red[jj * cg + g][8 + i]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = g * 8 + i;
            double* dst = sums + (size_t)(blockIdx.x % DVD_BN_NREP) * 2 * C;
            if (c < C) { atomicAdd(dst + c, s[i]); atomicAdd(dst + C + c, q[i]); }
        }
    }
}

// mean / rstd from the sums (training) or from the running buffers (eval); running-stat update.
__global__ void bn_finalize_kernel(double* sums, double n, int C, float eps, float momentum, int training,
                                   float* mean, float* rstd, float* run_mean, float* run_var, int rezero) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (training) {
        double s1 = 0, s2 = 0;
        for (int r = 0; r < DVD_BN_NREP; ++r) { s1 += sums[(size_t)r * 2 * C + c]; s2 += sums[(size_t)r * 2 * C + C + c]; }
        if (rezero)                                // persistent workspace: left zeroed for the next statistics pass
            for (int r = 0; r < DVD_BN_NREP; ++r) { sums[(size_t)r * 2 * C + c] = 0.0; sums[(size_t)r * 2 * C + C + c] = 0.0; }
        const double m = s1 / n;
        double var = s2 / n - m * m;
        if (var < 0) var = 0;
        mean[c] = (float)m;
        rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (run_mean) {
            const double unbiased = n > 1 ? var * n / (n - 1) : var;
            run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * m);
            run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unbiased);
        }
    } else {
        mean[c] = run_mean[c];
        rstd[c] = (float)(1.0 / sqrt((double)run_var[c] + (double)eps));
    }
}

// the conditional affine map of one element, spelled once: the backward kernels re-evaluate it to get the ReLU mask (t > 0) from x
// instead of reading the stored activation, and must land on the same side of zero as the forward did
__device__ __forceinline__ float cbn_affine(float v, float mu, float rs, float gam, float bet) {
    return __builtin_fmaf(gam, (v - mu) * rs, bet);
}

// ----------------------------------------------------------------------------- CBN apply
// y = act(gb[s][c] * (x - mean[c]) * rstd[c] + gb[s][C + c]),  s = samp[row / P]
// grid (frames, pixel chunks): every pixel of a frame shares one condition row, so the 4 x 8 per-channel
// constants live in registers and the loop is load -> 3 flops -> store with U vectors in flight.  (The first form
// looked all 32 constants up per 16-byte vector: 0.6-1.2 TB/s.)
template <typename T>
__global__ __launch_bounds__(256) void cbn_apply_kernel(const T* __restrict__ x, T* __restrict__ y, int P, int C, int ld,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gb, const int* __restrict__ samp,
                                                        int relu, int chunk) {
    const int cg = ld / 8, nj = 256 / cg;
    const int g = threadIdx.x % cg, j = threadIdx.x / cg;
    if (j >= nj) return;
    const int frame = blockIdx.x;
    const float* gbs = gb + (size_t)samp[frame] * 2 * C;
    float gam[8], bet[8], mu[8], rs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = g * 8 + k;
        const bool ok = c < C;
        gam[k] = ok ? gbs[c] : 0.f; bet[k] = ok ? gbs[C + c] : 0.f;
        mu[k] = ok ? mean[c] : 0.f; rs[k] = ok ? rstd[c] : 0.f;
    }
    const int p0 = blockIdx.y * chunk, p1 = min(P, p0 + chunk);
    const size_t base = (size_t)frame * P * ld + g * 8;
    constexpr int U = 4;
    for (int p = p0 + j; p < p1; p += nj * U) {
        float v[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (p + u * nj < p1) load8<T>(x + base + (size_t)(p + u * nj) * ld, v[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p + u * nj >= p1) break;
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float t = cbn_affine(v[u][k], mu[k], rs[k], gam[k], bet[k]);
                o[k] = relu ? fmaxf(t, 0.f) : t;
            }
            store8<T>(y + base + (size_t)(p + u * nj) * ld, o);
        }
    }
}

// ----------------------------------------------------------------------------- CBN backward
// dgb[s][c]   += sum gm * xhat,  dgb[s][C+c] += sum gm     with gm = g * (a > 0), over one frame
template <typename T>
__global__ __launch_bounds__(256) void cbn_bwd_reduce_kernel(const T* g, const T* x, int P, int C,
                                                             int ld, const float* mean, const float* rstd, const float* gb,
                                                             const int* samp, float* dgb, int relu, int chunk, float* part) {
    __shared__ float red[256][17];
    const int cg = (C + 7) / 8, nj = 256 / cg;
    const int gi = threadIdx.x % cg, j = threadIdx.x / cg;
    const int frame = blockIdx.x;
    const int p0 = blockIdx.y * chunk, p1 = min(P, p0 + chunk);
    float dg[8], db[8], mu[8], rs[8], gam[8], bet[8];
    const float* gbs = gb + (size_t)samp[frame] * 2 * C;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        dg[k] = db[k] = 0.f;
        const int c = gi * 8 + k;
        mu[k] = c < C ? mean[c] : 0.f;
        rs[k] = c < C ? rstd[c] : 0.f;
        gam[k] = (relu && c < C) ? gbs[c] : 0.f;
        bet[k] = (relu && c < C) ? gbs[C + c] : 0.f;
    }
    if (j < nj) {
        constexpr int U = 2;      // (4 in flight: 5-10 % slower on all four generator shapes)
        for (int p = p0 + j; p < p1; p += nj * U) {
            float gv[U][8], xv[U][8];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (p + u * nj < p1) {
                    const size_t off = ((size_t)frame * P + p + u * nj) * ld + gi * 8;
                    load8<T>(g + off, gv[u]);
                    load8<T>(x + off, xv[u]);
                }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (p + u * nj >= p1) break;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const bool on = !relu || cbn_affine(xv[u][k], mu[k], rs[k], gam[k], bet[k]) > 0.f;
                    const float gm = on ? gv[u][k] : 0.f;
                    dg[k] += gm * ((xv[u][k] - mu[k]) * rs[k]);
                    db[k] += gm;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[threadIdx.x][k] = dg[k]; red[threadIdx.x][8 + k] = db[k]; }
    __syncthreads();
    if (j == 0) {
        for (int jj = 1; jj < nj; ++jj)
#pragma unroll
            for (int k = 0; k < 8; ++k) { dg[k] += red[jj * cg + gi][k]; db[k] += red[jj * cg + gi][8 + k]; }
        // with a workspace: this block's partial sums, gathered per condition row in frame order by cbn_dgb_gather_kernel
        float* o = part ? part + ((size_t)frame * gridDim.y + blockIdx.y) * 2 * C : dgb + (size_t)samp[frame] * 2 * C;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = gi * 8 + k;
            if (c < C) {
                if (part) { o[c] = dg[k]; o[C + c] = db[k]; }
                else { atomicAdd(o + c, dg[k]); atomicAdd(o + C + c, db[k]); }
            }
        }
    }
}

// dgb[s][:] += sum over the frames conditioned on row s and their pixel chunks of the partial sums, in an order that depends on
// `samp` alone.  Block (s, 64 columns) = 4 frame lanes x 64 columns: the frames of row s are compacted IN FRAME ORDER into an LDS
// list, 2048 frames at a time (wave ballots + a prefix over the four waves); lane q sums hits q, q+4, ... of the list, the four lane
// sums are added in lane order.  (One thread per column walking the flags of all frames: 194 us per call, 3.1 ms per step.)
__global__ __launch_bounds__(256) void cbn_dgb_gather_kernel(const float* part, const int* samp, long long frames, int nchunk, int C2, float* dgb) {
    constexpr int kRound = 2048;
    __shared__ int list[kRound];
    __shared__ int wcount[4];
    __shared__ float red[4][64];
    const int s = blockIdx.x, col = threadIdx.x & 63, q = threadIdx.x >> 6, c = blockIdx.y * 64 + col;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a = 0.f;
    int carry = 0;                                   // hits of earlier rounds: keeps lane q on hits q, q+4, ... of the WHOLE list
    for (long long f0 = 0; f0 < frames; f0 += kRound) {
        int n = 0;
        __syncthreads();                             // the previous round's list has been consumed
        for (int i = 0; i < kRound; i += 256) {
            const long long f = f0 + i + threadIdx.x;
            const bool hit = f < frames && samp[f] == s;
            const unsigned long long m = __ballot(hit);
            if (lane == 0) wcount[wave] = __popcll(m);
            __syncthreads();
            int base = n;
            for (int w = 0; w < wave; ++w) base += wcount[w];
            if (hit) list[base + __popcll(m & ((1ull << lane) - 1))] = (int)(f - f0);
            n += wcount[0] + wcount[1] + wcount[2] + wcount[3];
            __syncthreads();
        }
        if (c < C2)
            for (int i = (q - carry) & 3; i < n; i += 4) {
                const float* pp = part + (size_t)(f0 + list[i]) * nchunk * C2 + c;
                for (int k = 0; k < nchunk; ++k) a += pp[(size_t)k * C2];
            }
        carry = (carry + n) & 3;
    }
    red[q][col] = a;
    __syncthreads();
    if (q == 0 && c < C2) dgb[(size_t)s * C2 + c] += ((red[0][col] + red[1][col]) + red[2][col]) + red[3][col];
}

// s12[c] = sum_s gb[s][c] * dgb[s][C+c]   (= sum dxhat),  s12[C+c] = sum_s gb[s][c] * dgb[s][c]  (= sum dxhat*xhat)
__global__ void cbn_bwd_sums_kernel(const float* gb, const float* dgb, int B, int C, float* s12) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a = 0, b = 0;
    for (int s = 0; s < B; ++s) {
        const double gam = gb[(size_t)s * 2 * C + c];
        a += gam * dgb[(size_t)s * 2 * C + C + c];
        b += gam * dgb[(size_t)s * 2 * C + c];
    }
    s12[c] = (float)a;
    s12[C + c] = (float)b;
}

// dx = rstd * (gm * gamma_s - s1/N - xhat * s2/N);  grid (pixel chunks, frames), constants in registers
template <typename T>
__global__ __launch_bounds__(256) void cbn_bwd_apply_kernel(const T* __restrict__ g,
                                                            const T* __restrict__ x, T* __restrict__ dx, int P, int C, int ld,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gb, const int* __restrict__ samp,
                                                            const float* __restrict__ s12, float inv_n, int relu, int chunk) {
    const int cg = ld / 8, nj = 256 / cg;
    const int gi = threadIdx.x % cg, j = threadIdx.x / cg;
    if (j >= nj) return;
    const int frame = blockIdx.x;
    const float* gbs = gb + (size_t)samp[frame] * 2 * C;
    float gam[8], bet[8], mu[8], rs[8], s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = gi * 8 + k;
        const bool ok = c < C;
        gam[k] = ok ? gbs[c] : 0.f; bet[k] = ok ? gbs[C + c] : 0.f; mu[k] = ok ? mean[c] : 0.f; rs[k] = ok ? rstd[c] : 0.f;
        s1[k] = ok ? s12[c] * inv_n : 0.f; s2[k] = ok ? s12[C + c] * inv_n : 0.f;
    }
    const int p0 = blockIdx.y * chunk, p1 = min(P, p0 + chunk);
    const size_t base = (size_t)frame * P * ld + gi * 8;
#ifndef DVD_CBN_U
#define DVD_CBN_U 2
#endif
    constexpr int U = DVD_CBN_U;
    for (int p = p0 + j; p < p1; p += nj * U) {
        float gv[U][8], xv[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (p + u * nj < p1) {
                const size_t off = base + (size_t)(p + u * nj) * ld;
                load8<T>(g + off, gv[u]);
                load8<T>(x + off, xv[u]);
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p + u * nj >= p1) break;
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool on = !relu || cbn_affine(xv[u][k], mu[k], rs[k], gam[k], bet[k]) > 0.f;
                const float gm = on ? gv[u][k] : 0.f;
                const float xh = (xv[u][k] - mu[k]) * rs[k];
                o[k] = rs[k] * (gm * gam[k] - s1[k] - xh * s2[k]);
            }
            store8<T>(dx + base + (size_t)(p + u * nj) * ld, o);
        }
    }
}

// ----------------------------------------------------------------------------- pooling / resampling
// y[f][t][y][x][c] = scale * sum over the (pt x 2 x 2) window of x.  Input grid Ti x Hi x Wi with Ti = To * pt, Hi / 2 = Ho, Wi / 2 = Wo:
// an odd Hi / Wi loses its last line / column, like F.avg_pool2d's floor (a 3 x 3 map pools to 1 x 1 in D_t at 96 x 96 frames)
template <typename T>
__global__ void pool_kernel(const T* x, T* y, long long nout, int To, int Ho, int Wo, int Hi, int Wi, int ld, int pt, float scale,
                            const T* mask) {
    const int cg = ld / 8;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nout * cg) return;
    long long r = i / cg;
    const int g = (int)(i - r * cg);
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho); r /= Ho;
    const int to = (int)(r % To);
    const long long f = r / To;
    const int Ti = To * pt;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int a = 0; a < pt; ++a)
        for (int b = 0; b < 2; ++b)
            for (int c = 0; c < 2; ++c) {
                float v[8];
                load8<T>(x + ((((size_t)f * Ti + to * pt + a) * Hi + yo * 2 + b) * Wi + xo * 2 + c) * ld + g * 8, v);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += v[k];
            }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] *= scale;
    if (mask) {          // ReLU mask of the pooled result's own grid (backward of relu -> nearest x2 -> conv): zero where mask <= 0
        float m[8];
        load8<T>(mask + (size_t)(i / cg) * ld + g * 8, m);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = m[k] > 0.f ? acc[k] : 0.f;
    }
    store8<T>(y + (size_t)(i / cg) * ld + g * 8, acc);
}
// 2x2x2 max pooling (stride 2) of a channels-last [f][T][H][W][c] tensor: Module/Attention.py:148 (nn.MaxPool3d(2, 2)).
template <typename T>
__global__ void maxpool3d_kernel(const T* x, T* y, long long nout, int To, int Ho, int Wo, int ld) {
    const int cg = ld / 8;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nout * cg) return;
    long long r = i / cg;
    const int g = (int)(i - r * cg);
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho); r /= Ho;
    const int to = (int)(r % To);
    const long long f = r / To;
    const int Ti = To * 2, Hi = Ho * 2, Wi = Wo * 2;
    float m[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) m[k] = -INFINITY;
    for (int w = 0; w < 8; ++w) {
        float v[8];
        load8<T>(x + ((((size_t)f * Ti + to * 2 + (w >> 2)) * Hi + yo * 2 + ((w >> 1) & 1)) * Wi + xo * 2 + (w & 1)) * ld + g * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = v[k] > m[k] ? v[k] : m[k];
    }
    store8<T>(y + (size_t)(i / cg) * ld + g * 8, m);
}
// backward: the gradient of a window goes to its FIRST maximum in (t, h, w) scan order (torch's tie rule: strict >);
// windows do not overlap, so every input element is written exactly once
template <typename T>
__global__ void maxpool3d_bwd_kernel(const T* x, const T* dy, T* dx, long long nout, int To, int Ho, int Wo, int ld) {
    const int cg = ld / 8;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nout * cg) return;
    long long r = i / cg;
    const int g = (int)(i - r * cg);
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho); r /= Ho;
    const int to = (int)(r % To);
    const long long f = r / To;
    const int Ti = To * 2, Hi = Ho * 2, Wi = Wo * 2;
    float m[8], gy[8];
    int arg[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { m[k] = -INFINITY; arg[k] = 0; }
    for (int w = 0; w < 8; ++w) {
        float v[8];
        load8<T>(x + ((((size_t)f * Ti + to * 2 + (w >> 2)) * Hi + yo * 2 + ((w >> 1) & 1)) * Wi + xo * 2 + (w & 1)) * ld + g * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) if (v[k] > m[k]) { m[k] = v[k]; arg[k] = w; }
    }
    load8<T>(dy + (size_t)(i / cg) * ld + g * 8, gy);
    for (int w = 0; w < 8; ++w) {
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = arg[k] == w ? gy[k] : 0.f;
        store8<T>(dx + ((((size_t)f * Ti + to * 2 + (w >> 2)) * Hi + yo * 2 + ((w >> 1) & 1)) * Wi + xo * 2 + (w & 1)) * ld + g * 8, o);
    }
}
// y[f][t][y][x][c] = scale * x[f][t/pt][y/2][x/2][c]; the last line / column of an odd output grid lies outside every pooling
// window (floor) and is zero
template <typename T>
__global__ void unpool_kernel(const T* x, T* y, long long nout, int To, int Ho, int Wo, int ld, int pt, float scale) {
    const int cg = ld / 8;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nout * cg) return;
    long long r = i / cg;
    const int g = (int)(i - r * cg);
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho); r /= Ho;
    const int to = (int)(r % To);
    const long long f = r / To;
    const int Ti = To / pt, Hi = Ho / 2, Wi = Wo / 2;
    float v[8];
    if (yo / 2 < Hi && xo / 2 < Wi) {
        load8<T>(x + ((((size_t)f * Ti + to / pt) * Hi + yo / 2) * Wi + xo / 2) * ld + g * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= scale;
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0.f;
    }
    store8<T>(y + (size_t)(i / cg) * ld + g * 8, v);
}

// ----------------------------------------------------------------------------- column sums (bias grads)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* x, long long rows, int C, int ld, float* out) {
    __shared__ float red[256][9];
    const int cg = (C + 7) / 8, nj = 256 / cg;
    const int g = threadIdx.x % cg, j = threadIdx.x / cg;
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = 0.f;
    if (j < nj)
        for (long long r = (long long)blockIdx.x * nj + j; r < rows; r += (long long)gridDim.x * nj) {
            float v[8];
            load8<T>(x + (size_t)r * ld + g * 8, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] += v[i];
        }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = s[i];
    __syncthreads();
    if (j == 0) {
        for (int jj = 1; jj < nj; ++jj)
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] += red[jj * cg + g][i];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (g * 8 + i < C) atomicAdd(out + g * 8 + i, s[i]);
    }
}

// ----------------------------------------------------------------------------- small elementwise
template <typename T>
__global__ void add_kernel(const T* a, const T* b, T* o, long long n8) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float x[8], y[8];
    load8<T>(a + i * 8, x);
    load8<T>(b + i * 8, y);
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] += y[k];
    store8<T>(o + i * 8, x);
}
template <typename T>
__global__ void sum_leading_kernel(const T* in, T* out, int L, long long n8) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int l = 0; l < L; ++l) {
        float v[8];
        load8<T>(in + ((size_t)l * n8 + i) * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
    store8<T>(out + i * 8, acc);
}
// dpre = dy * act'(y) expressed with the activation OUTPUT y (tanh: 1 - y^2, relu: y > 0)
template <typename T>
__global__ void act_bwd_kernel(const T* dy, const T* y, T* dx, long long n8, int act) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float g[8], v[8];
    load8<T>(dy + i * 8, g);
    load8<T>(y + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (act == DVD_ACT_TANH) g[k] *= 1.f - v[k] * v[k];
        else if (act == DVD_ACT_RELU) g[k] = v[k] > 0.f ? g[k] : 0.f;
        else if (act == DVD_ACT_SIGMOID) g[k] *= v[k] * (1.f - v[k]);
    }
    store8<T>(dx + i * 8, g);
}

// ----------------------------------------------------------------------------- reference-layout helpers
// utils.py:77-83  src f32 [B][T][C][H][W] -> dst f32 [B][C][T][H/2][W/2]   (bwd: transpose of it)
__global__ void vid_down_kernel(const float* src, float* dst, int B, int T, int C, int H, int W, int bwd) {
    const int Ho = H / 2, Wo = W / 2;
    const long long n = bwd ? (long long)B * T * C * H * W : (long long)B * C * T * Ho * Wo;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!bwd) {
        long long r = i;
        const int x = (int)(r % Wo); r /= Wo;
        const int y = (int)(r % Ho); r /= Ho;
        const int t = (int)(r % T); r /= T;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        const float* s = src + ((((size_t)b * T + t) * C + c) * H + 2 * y) * W + 2 * x;
        dst[i] = (s[0] + s[1] + s[W] + s[W + 1]) * 0.25f;
    } else {   // src = d(dst of fwd) [B][C][T][Ho][Wo], dst = d(src of fwd) [B][T][C][H][W]
        long long r = i;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H); r /= H;
        const int c = (int)(r % C); r /= C;
        const int t = (int)(r % T);
        const int b = (int)(r / T);
        dst[i] = 0.25f * src[((((size_t)b * C + c) * T + t) * Ho + y / 2) * Wo + x / 2];
    }
}
// dst row r = src row idx[r] (gather) or dst row idx[r] = src row r (scatter); rows of L floats
__global__ void row_copy_kernel(const float* src, float* dst, const int* idx, long long nrows, long long L, int scatter) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * L) return;
    const long long r = i / L, k = i - r * L;
    if (scatter) dst[(size_t)idx[r] * L + k] = src[i];
    else dst[i] = src[(size_t)idx[r] * L + k];
}

}  // namespace

#define S_ ((hipStream_t)stream)
// run `body` with the storage type bound to T
#define BY_DTYPE(dtype, ...)                                                   \
    do {                                                                       \
        if ((dtype) == DVD_BF16) { using T = bf16_t; __VA_ARGS__; }            \
        else if ((dtype) == DVD_F32) { using T = float; __VA_ARGS__; }         \
        else return DVD_E_ARG;                                                 \
    } while (0)

extern "C" int dvd_bn_stats(int dtype, const void* x, long long rows, int C, int ld, double* sums, void* stream) {
    if (!x || !sums || rows <= 0 || C <= 0) return DVD_E_ARG;
    if ((ld & 7) || C > ld || (C + 7) / 8 > 256) return DVD_E_SHAPE;
    const int nj = 256 / ((C + 7) / 8);
    unsigned grid = cdiv(rows, (long long)nj * 16);
    // every block ends with 2C fp64 atomics: 1024 blocks while that stays <= 256 k atomics per launch (3072 x 64 x 64 x 64: 459 us with
    // 512 blocks on one copy of the sums, 418 on 16 copies, 296 with 1024 blocks = 5.4 TB/s), 512 blocks for wider tensors
    // (C = 256 / 512: 112 / 63 us against 123 / 80 with 1024); 64 copies of the sums: no further gain
    constexpr unsigned forced = 0u;
    const unsigned cap = forced ? forced : (C <= 128 ? 1024u : 512u);
    if (grid > cap) grid = cap;
    BY_DTYPE(dtype, bn_stats_kernel<T><<<grid, 256, 0, S_>>>((const T*)x, rows, C, ld, sums));
    return launch_status();
}

extern "C" int dvd_bn_finalize(double* sums, long long rows, int C, float eps, float momentum, int training,
                               float* mean, float* rstd, float* run_mean, float* run_var, int rezero, void* stream) {
    if (!mean || !rstd || C <= 0 || (training ? !sums : (!run_mean || !run_var))) return DVD_E_ARG;
    bn_finalize_kernel<<<cdiv(C, 128), 128, 0, S_>>>(sums, (double)rows, C, eps, momentum, training, mean, rstd,
                                                     run_mean, run_var, rezero);
    return launch_status();
}

extern "C" int dvd_cbn_apply(int dtype, const void* x, void* y, long long frames, int P, int C, int ld,
                             const float* mean, const float* rstd, const float* gb, const int* samp, int relu,
                             void* stream) {
    if (!x || !y || !mean || !rstd || !gb || !samp || frames <= 0 || P <= 0) return DVD_E_ARG;
    if ((ld & 7) || C > ld) return DVD_E_SHAPE;
    if (ld / 8 > 256) return DVD_E_SHAPE;
    const int nj = 256 / (ld / 8), chunk = nj * 16;          // 16 pixels per thread
    dim3 grid((unsigned)frames, cdiv(P, chunk));
    BY_DTYPE(dtype, cbn_apply_kernel<T><<<grid, 256, 0, S_>>>((const T*)x, (T*)y, P, C, ld, mean, rstd, gb, samp, relu, chunk));
    return launch_status();
}

// Backward of the conditional batch norm in two stages so a data-parallel caller can all-reduce the two per-channel
// sums between them (cross-replica batch norm, Generator.py:57 TODO / SURVEY section 8e):
//   reduce: dgb[s][0..C) += sum g*xhat, dgb[s][C..2C) += sum g over the frames conditioned on row s;  s12 = the two
//           gamma-weighted totals over all condition rows (sum dxhat | sum dxhat*xhat)
//   apply : dx = rstd * (g*gamma_s - s12[c]/rows_total - xhat * s12[C+c]/rows_total)
extern "C" int dvd_cbn_backward_reduce(int dtype, const void* g, const void* a, const void* x, long long frames, int P, int C,
                                       int ld, const float* mean, const float* rstd, const float* gb, const int* samp, int B,
                                       float* dgb, float* s12, int relu, float* part, void* stream) {
    (void)a;          // the ReLU mask is re-evaluated from x (cbn_affine): the stored activation is not read
    if (!g || !x || !mean || !rstd || !gb || !samp || !dgb || !s12) return DVD_E_ARG;
    if (frames <= 0 || P <= 0 || B <= 0) return DVD_E_ARG;
    if ((ld & 7) || C > ld || ld / 8 > 256) return DVD_E_SHAPE;
    const int chunk = 2048;
    dim3 grid((unsigned)frames, cdiv(P, chunk));
    BY_DTYPE(dtype, cbn_bwd_reduce_kernel<T><<<grid, 256, 0, S_>>>((const T*)g, (const T*)x, P, C, ld,
                                                                   mean, rstd, gb, samp, dgb, relu, chunk, part));
    if (part) cbn_dgb_gather_kernel<<<dim3((unsigned)B, cdiv(2 * C, 64)), 256, 0, S_>>>(part, samp, frames, (int)grid.y, 2 * C, dgb);
    cbn_bwd_sums_kernel<<<cdiv(C, 128), 128, 0, S_>>>(gb, dgb, B, C, s12);
    return launch_status();
}

extern "C" long long dvd_cbn_backward_ws_floats(long long frames, int P, int C) {
    if (frames <= 0 || P <= 0 || C <= 0) return 0;
    return frames * cdiv(P, 2048) * 2 * C;            // (2048 = the pixel chunk of dvd_cbn_backward_reduce)
}

extern "C" int dvd_cbn_backward_apply(int dtype, const void* g, const void* a, const void* x, void* dx, long long frames,
                                      int P, int C, int ld, const float* mean, const float* rstd, const float* gb,
                                      const int* samp, const float* s12, long long rows_total, int relu, void* stream) {
    (void)a;
    if (!g || !x || !dx || !mean || !rstd || !gb || !samp || !s12) return DVD_E_ARG;
    if (frames <= 0 || P <= 0 || rows_total < frames * P) return DVD_E_ARG;
    if ((ld & 7) || C > ld || ld / 8 > 256) return DVD_E_SHAPE;
    const float inv_n = (float)(1.0 / (double)rows_total);
#ifndef DVD_CBN_CH                     // rows per workgroup = nj * 64 (16: 32 x 32 x 128 frames 930 -> 850 us, 16 x 16 x 256 543 -> 495 incl. the reduce pass; U = 4: nothing)
#define DVD_CBN_CH 64
#endif
    const int nj = 256 / (ld / 8), chunk2 = nj * DVD_CBN_CH;
    dim3 grid2((unsigned)frames, cdiv(P, chunk2));
    BY_DTYPE(dtype, cbn_bwd_apply_kernel<T><<<grid2, 256, 0, S_>>>((const T*)g, (const T*)x, (T*)dx, P, C,
                                                                   ld, mean, rstd, gb, samp, s12, inv_n, relu, chunk2));
    return launch_status();
}

extern "C" int dvd_cbn_backward(int dtype, const void* g, const void* a, const void* x, void* dx, long long frames,
                                int P, int C, int ld, const float* mean, const float* rstd, const float* gb,
                                const int* samp, int B, float* dgb, float* s12, int relu, float* part, void* stream) {
    if (!dx) return DVD_E_ARG;
    const int rc = dvd_cbn_backward_reduce(dtype, g, a, x, frames, P, C, ld, mean, rstd, gb, samp, B, dgb, s12, relu, part, stream);
    if (rc != DVD_OK) return rc;
    return dvd_cbn_backward_apply(dtype, g, a, x, dx, frames, P, C, ld, mean, rstd, gb, samp, s12, frames * P, relu, stream);
}

// y = scale * sum over (pt,2,2) windows; output grid frames x To x Ho x Wo, input grid frames x (To * pt) x Hi x Wi
extern "C" int dvd_pool(int dtype, const void* x, void* y, long long frames, int To, int Ho, int Wo, int Hi, int Wi, int ld, int pt,
                        float scale, void* stream) {
    if (!x || !y || frames <= 0 || To <= 0 || Ho <= 0 || Wo <= 0 || (pt != 1 && pt != 2)) return DVD_E_ARG;
    if ((ld & 7) || Hi / 2 != Ho || Wi / 2 != Wo) return DVD_E_SHAPE;
    const long long nout = frames * To * Ho * Wo, n = nout * (ld / 8);
    BY_DTYPE(dtype, pool_kernel<T><<<cdiv(n, 256), 256, 0, S_>>>((const T*)x, (T*)y, nout, To, Ho, Wo, Hi, Wi, ld, pt, scale,
                                                                 (const T*)nullptr));
    return launch_status();
}
// the same with a ReLU mask on the OUTPUT grid (same layout and leading dimension as y): y = mask > 0 ? pooled : 0
extern "C" int dvd_pool_masked(int dtype, const void* x, const void* mask, void* y, long long frames, int To, int Ho, int Wo, int ld,
                               int pt, float scale, void* stream) {
    if (!x || !y || !mask || frames <= 0 || To <= 0 || Ho <= 0 || Wo <= 0 || (pt != 1 && pt != 2)) return DVD_E_ARG;
    if (ld & 7) return DVD_E_SHAPE;
    const long long nout = frames * To * Ho * Wo, n = nout * (ld / 8);
    BY_DTYPE(dtype, pool_kernel<T><<<cdiv(n, 256), 256, 0, S_>>>((const T*)x, (T*)y, nout, To, Ho, Wo, 2 * Ho, 2 * Wo, ld, pt, scale,
                                                                 (const T*)mask));
    return launch_status();
}
// 2x2x2 max pooling; output grid frames x To x Ho x Wo (input 2To x 2Ho x 2Wo)
extern "C" int dvd_maxpool3d(int dtype, const void* x, void* y, long long frames, int To, int Ho, int Wo, int ld, void* stream) {
    if (!x || !y || frames <= 0 || To <= 0 || Ho <= 0 || Wo <= 0) return DVD_E_ARG;
    if (ld & 7) return DVD_E_SHAPE;
    const long long nout = frames * To * Ho * Wo, n = nout * (ld / 8);
    BY_DTYPE(dtype, maxpool3d_kernel<T><<<cdiv(n, 256), 256, 0, S_>>>((const T*)x, (T*)y, nout, To, Ho, Wo, ld));
    return launch_status();
}
extern "C" int dvd_maxpool3d_backward(int dtype, const void* x, const void* dy, void* dx, long long frames, int To, int Ho,
                                      int Wo, int ld, void* stream) {
    if (!x || !dy || !dx || frames <= 0 || To <= 0 || Ho <= 0 || Wo <= 0) return DVD_E_ARG;
    if (ld & 7) return DVD_E_SHAPE;
    const long long nout = frames * To * Ho * Wo, n = nout * (ld / 8);
    BY_DTYPE(dtype, maxpool3d_bwd_kernel<T><<<cdiv(n, 256), 256, 0, S_>>>((const T*)x, (const T*)dy, (T*)dx, nout, To, Ho,
                                                                          Wo, ld));
    return launch_status();
}
// y[t][y][x] = scale * x[t/pt][y/2][x/2]; output grid frames x To x Ho x Wo (input grid To / pt x Ho / 2 x Wo / 2, floor: the
// odd line / column of the output is zero -- the transpose of dvd_pool on an odd grid)
extern "C" int dvd_unpool(int dtype, const void* x, void* y, long long frames, int To, int Ho, int Wo, int ld, int pt,
                          float scale, void* stream) {
    if (!x || !y || frames <= 0 || To <= 0 || Ho <= 0 || Wo <= 0 || (pt != 1 && pt != 2)) return DVD_E_ARG;
    if ((ld & 7) || (To % pt) || Ho < 2 || Wo < 2) return DVD_E_SHAPE;
    const long long nout = frames * To * Ho * Wo, n = nout * (ld / 8);
    BY_DTYPE(dtype, unpool_kernel<T><<<cdiv(n, 256), 256, 0, S_>>>((const T*)x, (T*)y, nout, To, Ho, Wo, ld, pt, scale));
    return launch_status();
}

extern "C" int dvd_colsum(int dtype, const void* x, long long rows, int C, int ld, float* out, void* stream) {
    if (!x || !out || rows <= 0 || C <= 0) return DVD_E_ARG;
    if ((ld & 7) || C > ld || (C + 7) / 8 > 256) return DVD_E_SHAPE;
    const int nj = 256 / ((C + 7) / 8);
    unsigned grid = cdiv(rows, (long long)nj * 8);
    if (grid > 1024) grid = 1024;
    BY_DTYPE(dtype, colsum_kernel<T><<<grid, 256, 0, S_>>>((const T*)x, rows, C, ld, out));
    return launch_status();
}

extern "C" int dvd_add(int dtype, const void* a, const void* b, void* out, long long n, void* stream) {
    if (!a || !b || !out || n <= 0) return DVD_E_ARG;
    if (n & 7) return DVD_E_SHAPE;
    BY_DTYPE(dtype, add_kernel<T><<<cdiv(n / 8, 256), 256, 0, S_>>>((const T*)a, (const T*)b, (T*)out, n / 8));
    return launch_status();
}
extern "C" int dvd_sum_leading(int dtype, const void* in, void* out, int L, long long n, void* stream) {
    if (!in || !out || n <= 0 || L <= 0) return DVD_E_ARG;
    if (n & 7) return DVD_E_SHAPE;
    BY_DTYPE(dtype, sum_leading_kernel<T><<<cdiv(n / 8, 256), 256, 0, S_>>>((const T*)in, (T*)out, L, n / 8));
    return launch_status();
}
extern "C" int dvd_act_backward(int dtype, const void* dy, const void* y, void* dx, long long n, int act, void* stream) {
    if (!dy || !y || !dx || n <= 0) return DVD_E_ARG;
    if (n & 7) return DVD_E_SHAPE;
    BY_DTYPE(dtype, act_bwd_kernel<T><<<cdiv(n / 8, 256), 256, 0, S_>>>((const T*)dy, (const T*)y, (T*)dx, n / 8, act));
    return launch_status();
}

extern "C" int dvd_vid_downsample(const float* src, float* dst, int B, int T, int C, int H, int W, int backward,
                                  void* stream) {
    if (!src || !dst || B <= 0 || T <= 0 || C <= 0 || H <= 0 || W <= 0) return DVD_E_ARG;
    if ((H | W) & 1) return DVD_E_SHAPE;
    const long long n = backward ? (long long)B * T * C * H * W : (long long)B * C * T * (H / 2) * (W / 2);
    vid_down_kernel<<<cdiv(n, 256), 256, 0, S_>>>(src, dst, B, T, C, H, W, backward);
    return launch_status();
}
extern "C" int dvd_row_copy(const float* src, float* dst, const int* idx, long long nrows, long long L, int scatter,
                            void* stream) {
    if (!src || !dst || !idx || nrows <= 0 || L <= 0) return DVD_E_ARG;
    row_copy_kernel<<<cdiv(nrows * L, 256), 256, 0, S_>>>(src, dst, idx, nrows, L, scatter);
    return launch_status();
}
