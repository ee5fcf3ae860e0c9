// Spectral norm (power iteration, backward), small fp32 linear / embedding layers, projection
// head of the discriminators, adversarial loss and Adam.  All fp32; sizes are small (the largest
// matrix is 512 x 4608), so these kernels are latency- not bandwidth-critical.
#include "common.h"

namespace {

__device__ float block_sum(float v, float* sh) {          // blockDim.x <= 1024, result broadcast
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (l == 0) sh[w] = v;
    __syncthreads();
    float t = (threadIdx.x < nw) ? sh[threadIdx.x] : 0.f;
    if (w == 0) { t = wave_sum(t); if (l == 0) sh[0] = t; }
    __syncthreads();
    return sh[0];
}

// ----------------------------------------------------------------------------- spectral norm
// Normalization.py:19-31 -- one power iteration on W [h][w]:
//   v <- W^T u / (|W^T u| + eps);  u <- W v / (|W v| + eps);  sigma = u . (W v)
// Three launches so that the two matrix-vector products use the whole chip (the largest matrix is
// 512 x 4608 and there are 127 iterations per training step):
//   1. v_raw = W^T u          (grid over column blocks x row chunks, fp32 atomics into v, pre-zeroed)
//   2. t     = W v_raw        (one wave per row; written over u)
//   3. n_v = |v_raw|+eps, v = v_raw/n_v, (W v) = t/n_v, n = |W v|, u = (W v)/(n+eps), sigma = n^2/(n+eps)
__global__ __launch_bounds__(256) void sn_wtu_kernel(const float* W, int h, int w, const float* u, float* v) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i0 = blockIdx.y * 64, i1 = min(h, i0 + 64);
    if (j >= w) return;
    float a = 0.f;
    for (int i = i0; i < i1; ++i) a += W[(size_t)i * w + j] * u[i];
    atomicAdd(v + j, a);
}
__global__ __launch_bounds__(256) void sn_wv_kernel(const float* W, int h, int w, const float* v, float* t) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= h) return;
    float a = 0.f;
    for (int j = lane; j < w; j += 64) a += W[(size_t)i * w + j] * v[j];
    a = wave_sum(a);
    if (lane == 0) t[i] = a;
}
__global__ __launch_bounds__(1024) void sn_finish_kernel(int h, int w, float* u, float* v, float* sigma, float eps) {
    __shared__ float sh[32];
    float ss = 0.f;
    for (int j = threadIdx.x; j < w; j += blockDim.x) ss += v[j] * v[j];
    const float nv = sqrtf(block_sum(ss, sh)) + eps;
    for (int j = threadIdx.x; j < w; j += blockDim.x) v[j] = v[j] / nv;
    float s2 = 0.f;
    for (int i = threadIdx.x; i < h; i += blockDim.x) { const float t = u[i] / nv; u[i] = t; s2 += t * t; }
    const float n2 = block_sum(s2, sh);
    const float nu = sqrtf(n2) + eps;
    for (int i = threadIdx.x; i < h; i += blockDim.x) u[i] = u[i] / nu;
    if (threadIdx.x == 0) *sigma = n2 / nu;        // u . (W v) = |W v|^2 / (|W v| + eps)
}

// ---- all matrices of a network at once (dvd_sn_batched).  An item's blocks are found through the block offsets the host
// stored in the item table; W^T u needs no atomics here: one block owns 256 columns for all rows.
__device__ __forceinline__ int sn_find_item(const dvd_sn_item* it, int n, int blk, int which) {
    int lo = 0, hi = n - 1;                      // last item whose first block is <= blk
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        const int first = which == 0 ? it[mid].blk_wtu : which == 1 ? it[mid].blk_wv : it[mid].blk_pack;
        if (first <= blk) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__global__ __launch_bounds__(256) void sn_wtu_batched_kernel(const dvd_sn_item* items, int n) {
    const int k = sn_find_item(items, n, blockIdx.x, 0);
    const dvd_sn_item it = items[k];
    const int j = (blockIdx.x - it.blk_wtu) * 256 + threadIdx.x;
    if (j >= it.w) return;
    float a = 0.f;
    for (int i = 0; i < it.h; ++i) a += it.W[(size_t)i * it.w + j] * it.u[i];
    it.v[j] = a;
}
__global__ __launch_bounds__(256) void sn_wv_batched_kernel(const dvd_sn_item* items, int n, float* t_all) {
    const int k = sn_find_item(items, n, blockIdx.x, 1);
    const dvd_sn_item it = items[k];
    const int i = (blockIdx.x - it.blk_wv) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= it.h) return;
    float a = 0.f;
    for (int j = lane; j < it.w; j += 64) a += it.W[(size_t)i * it.w + j] * it.v[j];
    a = wave_sum(a);
    if (lane == 0) t_all[(size_t)it.blk_wv * 4 + i] = a;     // (u is still needed by other blocks' W^T u? no: that launch is over)
}
__global__ __launch_bounds__(1024) void sn_finish_batched_kernel(const dvd_sn_item* items, const float* t_all, float eps) {
    __shared__ float sh[32];
    const dvd_sn_item it = items[blockIdx.x];
    const float* t = t_all + (size_t)it.blk_wv * 4;
    float ss = 0.f;
    for (int j = threadIdx.x; j < it.w; j += blockDim.x) ss += it.v[j] * it.v[j];
    const float nv = sqrtf(block_sum(ss, sh)) + eps;
    for (int j = threadIdx.x; j < it.w; j += blockDim.x) it.v[j] = it.v[j] / nv;
    float s2 = 0.f;
    for (int i = threadIdx.x; i < it.h; i += blockDim.x) { const float q = t[i] / nv; s2 += q * q; }
    const float n2 = block_sum(s2, sh);
    const float nu = sqrtf(n2) + eps;
    for (int i = threadIdx.x; i < it.h; i += blockDim.x) it.u[i] = (t[i] / nv) / nu;
    if (threadIdx.x == 0) *it.sigma = n2 / nu;
}
template <typename T>
__device__ __forceinline__ void sn_pack_one(const dvd_sn_item& it, long long i) {
    const int tap = (int)(i % it.ntaps);
    const long long r = i / it.ntaps;
    const int ci = (int)(r % it.cip), co = (int)(r / it.cip);
    float v = 0.f;
    if (ci < it.cin) v = it.W[((size_t)co * it.cin + ci) * it.ntaps + tap] / *it.sigma;
    if (it.wf) stf(reinterpret_cast<T*>(it.wf) + ((size_t)tap * it.cout + co) * it.cip + ci, v);
    if (it.wd) stf(reinterpret_cast<T*>(it.wd) + ((size_t)(it.ntaps - 1 - tap) * it.cip + ci) * it.cop + co, v);
}
__global__ __launch_bounds__(256) void sn_pack_batched_kernel(const dvd_sn_item* items, int n) {
    const int k = sn_find_item(items, n, blockIdx.x, 2);
    const dvd_sn_item it = items[k];
    const long long i = (long long)(blockIdx.x - it.blk_pack) * 256 + threadIdx.x;
    if (i >= (long long)it.cout * it.cip * it.ntaps) return;
    if (it.dtype == DVD_BF16) sn_pack_one<bf16_t>(it, i); else sn_pack_one<float>(it, i);
}

// partial[block] = this block's share of sum G*W (no atomics: sn_grad_kernel adds the partials in a fixed order)
__global__ void sn_dot_kernel(const float* G, const float* W, long long n, float* partial) {
    __shared__ float sh[32];
    float a = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        a += G[i] * W[i];
    a = block_sum(a, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = a;
}
// W_sn = W/sigma, sigma = u.(W v)  =>  dL/dW = G/sigma - (sum G*W)/sigma^2 * u v^T        (npart <= 512 partials of sum G*W)
__global__ __launch_bounds__(256) void sn_grad_kernel(const float* G, const float* u, const float* v, const float* sigma, const float* partial,
                                                      int npart, int h, int w, float* dW) {
    __shared__ float sh[32];
    const int t = threadIdx.x;
    const float dot = block_sum((t < npart ? partial[t] : 0.f) + (t + 256 < npart ? partial[t + 256] : 0.f), sh);   // same order in every block
    const long long n = (long long)h * w;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = (int)(i / w), c = (int)(i - (long long)r * w);
    const float s = *sigma;
    dW[i] += G[i] / s - dot / (s * s) * u[r] * v[c];
}
// out = W / sigma
__global__ void sn_scale_kernel(const float* W, const float* sigma, float* out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = W[i] / *sigma;
}

// ----------------------------------------------------------------------------- fp32 linear
// out[b][j] = bias[j] + sum_k in[b][k] * W[j][k]      (wave per output)
__global__ void linear_fwd_kernel(const float* in, const float* W, const float* bias, float* out, int B, int K, int J) {
    const long long o = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (o >= (long long)B * J) return;
    const int b = (int)(o / J), j = (int)(o - (long long)b * J);
    float a = 0.f;
    for (int k = lane; k < K; k += 64) a += in[(size_t)b * K + k] * W[(size_t)j * K + k];
    a = wave_sum(a);
    if (lane == 0) out[o] = a + (bias ? bias[j] : 0.f);
}
// din[b][k] (+)= sum_j dout[b][j] W[j][k].  Block = 64 columns k x 4 j-lanes of one row b: the J loop (up to 4096
// for the generator's first Linear) is split four ways and folded through LDS, grid (B, K/64).
__global__ __launch_bounds__(256) void linear_bwd_in_kernel(const float* dout, const float* W, float* din, int B, int K,
                                                            int J, int accumulate) {
    __shared__ float red[4][64];
    const int b = blockIdx.x, kx = threadIdx.x & 63, jl = threadIdx.x >> 6;
    const int k = blockIdx.y * 64 + kx;
    float a = 0.f;
    if (k < K) {
#pragma unroll 8
        for (int j = jl; j < J; j += 4) a += dout[(size_t)b * J + j] * W[(size_t)j * K + k];      // 8 loads in flight: the J / 4 steps are latency
    }
    red[jl][kx] = a;
    __syncthreads();
    if (jl == 0 && k < K) {
        a = (red[0][kx] + red[1][kx]) + (red[2][kx] + red[3][kx]);
        const size_t i = (size_t)b * K + k;
        din[i] = accumulate ? din[i] + a : a;
    }
}
// dW[j][k] += sum_b dout[b][j] in[b][k];  dbias[j] += sum_b dout[b][j]
__global__ void linear_bwd_w_kernel(const float* dout, const float* in, float* dW, float* dbias, int B, int K, int J) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)J * K) return;
    const int j = (int)(i / K), k = (int)(i - (long long)j * K);
    float a = 0.f, s = 0.f;
    for (int b = 0; b < B; ++b) {
        const float d = dout[(size_t)b * J + j];
        a += d * in[(size_t)b * K + k];
        s += d;
    }
    dW[i] += a;
    if (dbias && k == 0) dbias[j] += s;
}
// dW[idx[i]][:] += dout[i][:].  The rows that share an index are added by the thread of their FIRST occurrence, in row order
// (fp32 atomics per row gave a run-to-run order whenever a class came up three times in the batch).
__global__ void embedding_bwd_kernel(const float* dout, const int* idx, float* dW, long long n, int D) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * D) return;
    const long long r = i / D;
    const int c = (int)(i - r * D), row = idx[r];
    for (long long q = 0; q < r; ++q)
        if (idx[q] == row) return;
    float a = 0.f;
    for (long long q = r; q < n; ++q)
        if (idx[q] == row) a += dout[q * D + c];
    dW[(size_t)row * D + c] += a;
}

// ----------------------------------------------------------------------------- projection head
// hsum[f][c] = sum_p relu(feat[f][p][c])                 Discriminators.py:264-271 / 421-428
template <typename T>
__global__ void relu_spatial_sum_kernel(const T* feat, float* hsum, long long F, int P, int C, int ld) {
    const int cg = ld / 8;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F * cg) return;
    const long long f = i / cg;
    const int c = (int)(i - f * cg) * 8;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int p = 0; p < P; ++p) {
        float v[8];
        load8<T>(feat + ((size_t)f * P + p) * ld + c, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += fmaxf(v[k], 0.f);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (c + k < C) hsum[(size_t)f * C + c + k] = acc[k];
}
// dfeat[f][p][c] = dh[f][c] * (feat > 0)
template <typename T>
__global__ void relu_spatial_sum_bwd_kernel(const float* dh, const T* feat, T* dfeat, long long F, int P, int C, int ld) {
    const int cg = ld / 8;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F * P * cg) return;
    const long long row = i / cg;
    const int c = (int)(i - row * cg) * 8;
    const long long f = row / P;
    float v[8], o[8];
    load8<T>(feat + (size_t)row * ld + c, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (c + k < C && v[k] > 0.f) ? dh[(size_t)f * C + c + k] : 0.f;
    store8<T>(dfeat + (size_t)row * ld + c, o);
}
// out[f] = b + sum_c hsum[f][c] * (wl[c]/sl + emb[cls[f]][c]/se)       (wave per frame)
__global__ void proj_head_fwd_kernel(const float* hsum, const float* wl, const float* sl, const float* bias,
                                     const float* emb, const float* se, const int* cls, float* out, long long F, int C) {
    const long long f = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (f >= F) return;
    const float isl = 1.f / *sl, ise = 1.f / *se;
    const float* e = emb + (size_t)cls[f] * C;
    float a = 0.f;
    for (int c = lane; c < C; c += 64) a += hsum[(size_t)f * C + c] * (wl[c] * isl + e[c] * ise);
    a = wave_sum(a);
    if (lane == 0) out[f] = a + bias[0];
}
// dh[f][c] = dout[f] * (wl[c]/sl + emb[cls][c]/se);  G_lin[c] += dout[f] h[f][c];  G_emb[cls][c] += dout[f] h[f][c].
// All three sums run in frame order inside ONE thread each (G_emb: the thread of the first frame of a class), no atomics.
__global__ void proj_head_bwd_kernel(const float* dout, const float* hsum, const float* wl, const float* sl,
                                     const float* emb, const float* se, const int* cls, float* dh, float* g_lin,
                                     float* g_emb, float* g_bias, long long F, int C) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F * C) return;
    const long long f = i / C;
    const int c = (int)(i - f * C);
    const float d = dout[f];
    const int k = cls[f];
    const size_t e = (size_t)k * C + c;
    dh[i] = d * (wl[c] / *sl + emb[e] / *se);
    if (!g_lin) return;
    if (f == 0) {                                   // column c of the linear weight; the bias rides with column 0
        float a = 0.f, b = 0.f;
        for (long long q = 0; q < F; ++q) { a += dout[q] * hsum[q * C + c]; b += dout[q]; }
        g_lin[c] += a;
        if (c == 0) *g_bias += b;
    }
    for (long long q = 0; q < f; ++q)
        if (cls[q] == k) return;
    float a = 0.f;
    for (long long q = f; q < F; ++q)
        if (cls[q] == k) a += dout[q] * hsum[q * C + c];
    g_emb[e] += a;
}

// ----------------------------------------------------------------------------- adversarial loss
// trainer.py:114-121.  x = real ? -out : out;  hinge: mean(relu(1+x));  wgan-gp: mean(x).
// *loss += value;  dout[i] = d(value)/d(out[i]) * gscale
__global__ void adv_loss_kernel(const float* out, long long n, int hinge, int real, float* loss, float* dout, float gscale) {
    __shared__ float sh[32];
    const float sgn = real ? -1.f : 1.f;
    float a = 0.f;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
        const float x = sgn * out[i];
        float l = hinge ? fmaxf(1.f + x, 0.f) : x;
        a += l;
        if (dout) dout[i] = ((hinge && !(1.f + x > 0.f)) ? 0.f : sgn) * gscale / (float)n;
    }
    a = block_sum(a, sh);
    if (threadIdx.x == 0) *loss += a / (float)n;
}

// ----------------------------------------------------------------------------- Adam
// torch.optim.Adam (no weight decay / amsgrad), trainer.py:136-141
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long long n, float lr_over_bc1, float b1,
                            float b2, float eps, float bc2_sqrt) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= lr_over_bc1 * mi / (sqrtf(vi) / bc2_sqrt + eps);
}

}  // namespace

#define S_ ((hipStream_t)stream)
#define BY_DTYPE(dtype, ...)                                                   \
    do {                                                                       \
        if ((dtype) == DVD_BF16) { using T = bf16_t; __VA_ARGS__; }            \
        else if ((dtype) == DVD_F32) { using T = float; __VA_ARGS__; }         \
        else return DVD_E_ARG;                                                 \
    } while (0)

extern "C" int dvd_sn_power_iter(const float* W, int h, int w, float* u, float* v, float* sigma, void* stream) {
    if (!W || !u || !v || !sigma || h <= 0 || w <= 0) return DVD_E_ARG;
    if (hipMemsetAsync(v, 0, (size_t)w * sizeof(float), S_) != hipSuccess) return DVD_E_LAUNCH;
    sn_wtu_kernel<<<dim3(cdiv(w, 256), cdiv(h, 64)), 256, 0, S_>>>(W, h, w, u, v);
    sn_wv_kernel<<<cdiv(h, 4), 256, 0, S_>>>(W, h, w, v, u);
    sn_finish_kernel<<<1, 1024, 0, S_>>>(h, w, u, v, sigma, 1e-12f);
    return launch_status();
}
extern "C" int dvd_sn_batched_prepare(dvd_sn_item* host, int n, long long* scratch_floats) {
    if (!host || n <= 0 || !scratch_floats) return DVD_E_ARG;
    long long b0 = 0, b1 = 0, b2 = 0;
    for (int k = 0; k < n; ++k) {
        dvd_sn_item& it = host[k];
        if (!it.W || !it.u || !it.v || !it.sigma || it.h <= 0 || it.w <= 0) return DVD_E_ARG;
        if ((it.wf || it.wd) && (it.cout != it.h || it.cin * it.ntaps != it.w || (it.cip & 7) || it.cip < it.cin ||
                                 (it.dtype != DVD_BF16 && it.dtype != DVD_F32)))
            return DVD_E_ARG;
        it.blk_wtu = (int)b0; it.blk_wv = (int)b1; it.blk_pack = (int)b2;
        b0 += cdiv(it.w, 256); b1 += cdiv(it.h, 4);
        b2 += (it.wf || it.wd) ? cdiv((long long)it.cout * it.cip * it.ntaps, 256) : 0;
        if (b2 >= (1ll << 31)) return DVD_E_SHAPE;
    }
    *scratch_floats = b1 * 4;                   // the products W v of all items: 4 floats per block of the second launch
    return DVD_OK;
}
extern "C" int dvd_sn_batched(const dvd_sn_item* host, const dvd_sn_item* dev, int n, float* scratch, void* stream) {
    if (!host || !dev || !scratch || n <= 0) return DVD_E_ARG;
    const dvd_sn_item& last = host[n - 1];
    const unsigned g0 = last.blk_wtu + cdiv(last.w, 256), g1 = last.blk_wv + cdiv(last.h, 4);
    const unsigned g2 = last.blk_pack + ((last.wf || last.wd) ? cdiv((long long)last.cout * last.cip * last.ntaps, 256) : 0);
    float* t_all = scratch;
    sn_wtu_batched_kernel<<<g0, 256, 0, S_>>>(dev, n);
    sn_wv_batched_kernel<<<g1, 256, 0, S_>>>(dev, n, t_all);
    sn_finish_batched_kernel<<<n, 1024, 0, S_>>>(dev, t_all, 1e-12f);
    if (g2) sn_pack_batched_kernel<<<g2, 256, 0, S_>>>(dev, n);
    return launch_status();
}
extern "C" int dvd_sn_backward(const float* G, const float* W, const float* u, const float* v, const float* sigma,
                               int h, int w, float* dW, float* scratch, void* stream) {
    if (!G || !W || !u || !v || !sigma || !dW || !scratch || h <= 0 || w <= 0) return DVD_E_ARG;
    const long long n = (long long)h * w;
    unsigned g = cdiv(n, 256 * 8);
    if (g > DVD_SN_SCRATCH) g = DVD_SN_SCRATCH;
    sn_dot_kernel<<<g, 256, 0, S_>>>(G, W, n, scratch);
    sn_grad_kernel<<<cdiv(n, 256), 256, 0, S_>>>(G, u, v, sigma, scratch, (int)g, h, w, dW);
    return launch_status();
}
extern "C" int dvd_sn_scale(const float* W, const float* sigma, float* out, long long n, void* stream) {
    if (!W || !sigma || !out || n <= 0) return DVD_E_ARG;
    sn_scale_kernel<<<cdiv(n, 256), 256, 0, S_>>>(W, sigma, out, n);
    return launch_status();
}

extern "C" int dvd_linear_forward(const float* in, const float* W, const float* bias, float* out, int B, int K, int J,
                                  void* stream) {
    if (!in || !W || !out || B <= 0 || K <= 0 || J <= 0) return DVD_E_ARG;
    linear_fwd_kernel<<<cdiv((long long)B * J * 64, 256), 256, 0, S_>>>(in, W, bias, out, B, K, J);
    return launch_status();
}
extern "C" int dvd_linear_backward(const float* dout, const float* in, const float* W, float* din, int din_accumulate,
                                   float* dW, float* dbias, int B, int K, int J, void* stream) {
    if (!dout || !in || !W || B <= 0 || K <= 0 || J <= 0) return DVD_E_ARG;
    if (din) linear_bwd_in_kernel<<<dim3((unsigned)B, cdiv(K, 64)), 256, 0, S_>>>(dout, W, din, B, K, J, din_accumulate);
    if (dW) linear_bwd_w_kernel<<<cdiv((long long)J * K, 256), 256, 0, S_>>>(dout, in, dW, dbias, B, K, J);
    return launch_status();
}
extern "C" int dvd_embedding_backward(const float* dout, const int* idx, float* dW, long long n, int D, void* stream) {
    if (!dout || !idx || !dW || n <= 0 || D <= 0) return DVD_E_ARG;
    embedding_bwd_kernel<<<cdiv(n * D, 256), 256, 0, S_>>>(dout, idx, dW, n, D);
    return launch_status();
}

extern "C" int dvd_relu_spatial_sum(int dtype, const void* feat, float* hsum, long long F, int P, int C, int ld,
                                    void* stream) {
    if (!feat || !hsum || F <= 0 || P <= 0 || C <= 0) return DVD_E_ARG;
    if ((ld & 7) || C > ld) return DVD_E_SHAPE;
    BY_DTYPE(dtype, relu_spatial_sum_kernel<T><<<cdiv(F * (ld / 8), 256), 256, 0, S_>>>((const T*)feat, hsum, F, P, C, ld));
    return launch_status();
}
extern "C" int dvd_relu_spatial_sum_backward(int dtype, const float* dh, const void* feat, void* dfeat, long long F,
                                             int P, int C, int ld, void* stream) {
    if (!dh || !feat || !dfeat || F <= 0 || P <= 0 || C <= 0) return DVD_E_ARG;
    if ((ld & 7) || C > ld) return DVD_E_SHAPE;
    BY_DTYPE(dtype, relu_spatial_sum_bwd_kernel<T><<<cdiv(F * P * (ld / 8), 256), 256, 0, S_>>>(dh, (const T*)feat,
                                                                                               (T*)dfeat, F, P, C, ld));
    return launch_status();
}
extern "C" int dvd_proj_head_forward(const float* hsum, const float* wl, const float* sl, const float* bias,
                                     const float* emb, const float* se, const int* cls, float* out, long long F, int C,
                                     void* stream) {
    if (!hsum || !wl || !sl || !bias || !emb || !se || !cls || !out || F <= 0 || C <= 0) return DVD_E_ARG;
    proj_head_fwd_kernel<<<cdiv(F * 64, 256), 256, 0, S_>>>(hsum, wl, sl, bias, emb, se, cls, out, F, C);
    return launch_status();
}
extern "C" int dvd_proj_head_backward(const float* dout, const float* hsum, const float* wl, const float* sl,
                                      const float* emb, const float* se, const int* cls, float* dh, float* g_lin,
                                      float* g_emb, float* g_bias, long long F, int C, void* stream) {
    if (!dout || !hsum || !wl || !sl || !emb || !se || !cls || !dh || F <= 0 || C <= 0) return DVD_E_ARG;
    if (g_lin && (!g_emb || !g_bias)) return DVD_E_ARG;
    proj_head_bwd_kernel<<<cdiv(F * C, 256), 256, 0, S_>>>(dout, hsum, wl, sl, emb, se, cls, dh, g_lin, g_emb, g_bias, F, C);
    return launch_status();
}

extern "C" int dvd_adv_loss(const float* out, long long n, int hinge, int real_flag, float* loss, float* dout,
                            float grad_scale, void* stream) {
    if (!out || !loss || n <= 0) return DVD_E_ARG;
    adv_loss_kernel<<<1, 1024, 0, S_>>>(out, n, hinge, real_flag, loss, dout, grad_scale);
    return launch_status();
}

extern "C" int dvd_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                             float beta2, float eps, int step, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step <= 0) return DVD_E_ARG;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    adam_kernel<<<cdiv(n, 256), 256, 0, S_>>>(p, g, m, v, n, (float)((double)lr / bc1), beta1, beta2, eps, (float)sqrt(bc2));
    return launch_status();
}
