// SeparableAttnCell of Module/Attention.py:24-111 (the T / W / H cells of SeparableAttn, :8-21) on the device.
//
// The reference builds its "attention along one axis" from RAW RESHAPES of contiguous NCDHW tensors, not from
// transposes: with o = x with the attended axis swapped to position 2 (sizes D0 = A, D1, D2),
//     q = conv1x1(o)                  contiguous [C/2, A, D1, D2]   is REINTERPRETED as  Qf [A][L],   L = C/2 * D1 * D2
//     k = maxpool(2,1,1)(conv1x1(o))  contiguous [C/2, A/2, D1, D2] is REINTERPRETED as  Kp [L][A/2]
//     v = maxpool(2,1,1)(conv1x1(o))  contiguous [C,   A/2, D1, D2] is REINTERPRETED as  Vp [R][A/2], R = C * D1 * D2
//     att = softmax_j(Qf Kp)  [A][A/2];   out = Vp att^T  [R][A], reinterpreted as [C, (other two axes), A] and permuted
//     y = gamma * out + x
// so a "row" of Qf is a run of L consecutive elements of the flat q tensor, whatever (channel, position) they belong to.
// This file restates exactly that on channels-last activations: `gather` materialises the three flat fp32 operands
// (q | k | v come from ONE fused 1x1 convolution, columns [0,Cq) | [koff,koff+Cq) | [voff,voff+C) of `qkv`) with the
// index arithmetic of the reshapes, the small products run on them, and `scatter` routes the gradients back (the max-pool
// gradient goes to the first maximum of each pair, like F.max_pool3d).  The block is defined by the reference but not
// invoked by its generator; it is < 0.1 % of a step's FLOPs when enabled, so everything is plain fp32 on the vector pipe.
// Round 5: the two small products run as tiled products (sep_prod_kernel); the gather / scatter / output kernels are still one
// thread per flat element with 2-byte scattered accesses (bench.py --g-attn sep shows what that costs).
#include "common.h"

namespace {

struct SepGeo {
    int T, W, H, axis;         // axis: 0 = T, 1 = W, 2 = H
    int A, D1, D2;             // permuted sizes (D0 = A)
    int C, Cq;
    long long N;               // T * W * H
};

__host__ __device__ inline SepGeo make_geo(int T, int W, int H, int axis, int C, int Cq) {
    SepGeo g;
    g.T = T; g.W = W; g.H = H; g.axis = axis; g.C = C; g.Cq = Cq; g.N = (long long)T * W * H;
    if (axis == 0) { g.A = T; g.D1 = W; g.D2 = H; }
    else if (axis == 1) { g.A = W; g.D1 = T; g.D2 = H; }
    else { g.A = H; g.D1 = W; g.D2 = T; }
    return g;
}

// permuted position (d0, d1, d2) -> token index (t * W + w) * H + h of the channels-last tensor
__device__ __forceinline__ long long token(const SepGeo& g, int d0, int d1, int d2) {
    int t, w, h;
    if (g.axis == 0) { t = d0; w = d1; h = d2; }
    else if (g.axis == 1) { t = d1; w = d0; h = d2; }          // transpose(2, 3): [C, W, T, H]
    else { t = d2; w = d1; h = d0; }                           // transpose(2, 4): [C, H, W, T]
    return ((long long)t * g.W + w) * g.H + h;
}

// flat output index o of out [R][A] viewed as [C, E1, E2, E3] (Attention.py:101-106) -> (channel, token)
__device__ __forceinline__ void out_dest(const SepGeo& g, long long o, int& c, long long& tok) {
    int e3, e2, e1;
    if (g.axis == 0) {            // view (C, W, H, T), permute(0, 1, 4, 2, 3)
        e3 = (int)(o % g.T); o /= g.T; e2 = (int)(o % g.H); o /= g.H; e1 = (int)(o % g.W); c = (int)(o / g.W);
        tok = ((long long)e3 * g.W + e1) * g.H + e2;
    } else if (g.axis == 1) {     // view (C, T, H, W), permute(0, 1, 2, 4, 3)
        e3 = (int)(o % g.W); o /= g.W; e2 = (int)(o % g.H); o /= g.H; e1 = (int)(o % g.T); c = (int)(o / g.T);
        tok = ((long long)e1 * g.W + e3) * g.H + e2;
    } else {                      // view (C, T, W, H)
        e3 = (int)(o % g.H); o /= g.H; e2 = (int)(o % g.W); o /= g.W; e1 = (int)(o % g.T); c = (int)(o / g.T);
        tok = ((long long)e1 * g.W + e2) * g.H + e3;
    }
}

// one thread per element of Qf, Kp, Vp (in that order)
template <typename T>
__global__ void sep_gather_kernel(SepGeo g, const T* qkv, int ldq, int koff, int voff, float* Qf, float* Kp, float* Vp,
                                  unsigned char* ksel, unsigned char* vsel, long long B) {
    const long long nq = (long long)g.Cq * g.N, nk = nq / 2, nv = (long long)g.C * g.N / 2, per = nq + nk + nv;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * per) return;
    const long long b = i / per;
    long long e = i - b * per;
    const T* base = qkv + (size_t)b * g.N * ldq;
    if (e < nq) {
        const int d2 = (int)(e % g.D2); long long r = e / g.D2;
        const int d1 = (int)(r % g.D1); r /= g.D1;
        const int d0 = (int)(r % g.A); const int c = (int)(r / g.A);
        Qf[b * nq + e] = ldf(base + (size_t)token(g, d0, d1, d2) * ldq + c);
        return;
    }
    e -= nq;
    const bool isv = e >= nk;
    if (isv) e -= nk;
    const int Ah = g.A / 2;
    const int d2 = (int)(e % g.D2); long long r = e / g.D2;
    const int d1 = (int)(r % g.D1); r /= g.D1;
    const int dp = (int)(r % Ah); const int c = (int)(r / Ah);
    const int col = (isv ? voff : koff) + c;
    const float a0 = ldf(base + (size_t)token(g, 2 * dp, d1, d2) * ldq + col);
    const float a1 = ldf(base + (size_t)token(g, 2 * dp + 1, d1, d2) * ldq + col);
    const bool second = a1 > a0;                                 // ties -> the first element, like F.max_pool3d
    if (isv) { Vp[b * nv + e] = second ? a1 : a0; vsel[b * nv + e] = second; }
    else { Kp[b * nk + e] = second ? a1 : a0; ksel[b * nk + e] = second; }
}

__device__ float blk_sum256(float v, float* sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sh[w] = v;
    __syncthreads();
    const float t = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return t;
}

// The two [A][A/2] products of a cell -- scores[a][j] = sum_l Qf[a*L + l] * Kp[l*(A/2) + j] (forward) and
// datt[a][j] = sum_r dO[r*A + a] * Vp[r*(A/2) + j] (backward) -- as tiled products (round 5; the first form ran one block per
// output row a with a block reduction per column j over STRIDED operand columns: 29 / 194 ms per call at 48 x 32 x 32, C = 128).
// Block (split s of the long axis, clip b), 1024 threads: 64 rows of both operands at a time through LDS (coalesced loads; the
// first operand is stored [l][a] whichever way it lies in memory), thread = (4 x 4 block of outputs, one of 8 row lanes), the 8 row
// lanes folded by a fixed butterfly.  P[b][s][a][j] = the partial sums of split s, added in split order by the finishing kernels.
constexpr int kSepSplit = 8, kSepRows = 64;
template <bool FIRST_ROWS>      // true: U[a*L + l] (Qf), false: U[l*A + a] (dO)
__global__ __launch_bounds__(1024) void sep_prod_kernel(const float* U, const float* V, float* P, int A, int Ah, long long L) {
    __shared__ float Ut[kSepRows][68];              // [l][a], a < 64 (+4: rows 16-byte aligned, banks spread)
    __shared__ float Vt[kSepRows][36];              // [l][j], j < 32
    const int tid = threadIdx.x, s = blockIdx.x;
    const long long b = blockIdx.y;
    U += b * A * L; V += b * L * Ah;
    const int SJ = (Ah + 3) / 4, NSB = ((A + 3) / 4) * SJ;
    const int ll = tid & 7, sb = tid >> 3;
    const bool live = sb < NSB;
    const int a0 = (sb / SJ) * 4, j0 = (sb % SJ) * 4;
    long long per = (L + kSepSplit - 1) / kSepSplit;
    per = (per + kSepRows - 1) / kSepRows * kSepRows;
    const long long l0 = s * per, l1 = l0 + per < L ? l0 + per : L;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (long long lc = l0; lc < l1; lc += kSepRows) {
        __syncthreads();
        for (int idx = tid; idx < 64 * kSepRows; idx += 1024) {      // first operand, zero-padded to 64 columns
            int a, l;
            if (FIRST_ROWS) { a = idx / kSepRows; l = idx - a * kSepRows; } else { l = idx / 64; a = idx - l * 64; }
            float v = 0.f;
            if (a < A && lc + l < l1) v = FIRST_ROWS ? U[(long long)a * L + lc + l] : U[(lc + l) * A + a];
            Ut[l][a] = v;
        }
        for (int idx = tid; idx < 32 * kSepRows; idx += 1024) {      // second operand, zero-padded to 32 columns
            const int l = idx >> 5, j = idx & 31;
            Vt[l][j] = (j < Ah && lc + l < l1) ? V[(lc + l) * Ah + j] : 0.f;
        }
        __syncthreads();
        if (live)
#pragma unroll
            for (int l = ll; l < kSepRows; l += 8) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(&Ut[l][a0]);
                const f32x4 k = *reinterpret_cast<const f32x4*>(&Vt[l][j0]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] += q[i] * k[j];
            }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[i][j];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
            if (live && ll == 0 && a0 + i < A && j0 + j < Ah) P[((b * kSepSplit + s) * A + a0 + i) * Ah + j0 + j] = v;
        }
}

// att[b][a][:] = softmax_j of the summed score partials;  grid (A, B), one wave
__global__ __launch_bounds__(64) void sep_softmax_kernel(const float* P, float* att, int A, int Ah) {
    const int a = blockIdx.x, j = threadIdx.x;
    const long long b = blockIdx.y;
    float sc = -INFINITY;
    if (j < Ah) {
        sc = 0.f;
        for (int s = 0; s < kSepSplit; ++s) sc += P[((b * kSepSplit + s) * A + a) * Ah + j];
    }
    const float m = wave_max(sc);
    const float e = j < Ah ? expf(sc - m) : 0.f;
    const float sum = wave_sum(e);
    if (j < Ah) att[(b * A + a) * Ah + j] = e / sum;
}

// y = gamma * out + x with out[r][a] = sum_j Vp[r*(A/2)+j] * att[a][j];  one thread per flat output index
template <typename T>
__global__ void sep_out_kernel(SepGeo g, const float* Vp, const float* att, const T* x, int ldx, const float* gamma, T* y,
                               long long B) {
    const long long per = (long long)g.C * g.N;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * per) return;
    const long long b = i / per, o = i - b * per;
    const int Ah = g.A / 2;
    const long long r = o / g.A;
    const int a = (int)(o - r * g.A);
    const float* v = Vp + b * (per / 2) + r * Ah;
    const float* at = att + (b * g.A + a) * Ah;
    float s = 0.f;
    for (int j = 0; j < Ah; ++j) s += v[j] * at[j];
    int c; long long tok;
    out_dest(g, o, c, tok);
    const size_t off = ((size_t)b * g.N + tok) * ldx + c;
    stf(y + off, *gamma * s + ldf(x + off));
}

// dO[b][o] = gamma * dy[dest(o)];  dgamma += sum dy * out
template <typename T>
__global__ __launch_bounds__(256) void sep_dout_kernel(SepGeo g, const float* Vp, const float* att, const T* dy, int ldx,
                                                       const float* gamma, float* dO, float* dgamma, long long B) {
    __shared__ float sh[4];
    const long long per = (long long)g.C * g.N;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float part = 0.f;
    if (i < B * per) {
        const long long b = i / per, o = i - b * per;
        const int Ah = g.A / 2;
        const long long r = o / g.A;
        const int a = (int)(o - r * g.A);
        const float* v = Vp + b * (per / 2) + r * Ah;
        const float* at = att + (b * g.A + a) * Ah;
        float s = 0.f;
        for (int j = 0; j < Ah; ++j) s += v[j] * at[j];
        int c; long long tok;
        out_dest(g, o, c, tok);
        const float d = ldf(dy + ((size_t)b * g.N + tok) * ldx + c);
        dO[i] = *gamma * d;
        part = d * s;
    }
    part = blk_sum256(part, sh);
    if (threadIdx.x == 0 && part != 0.f) atomicAdd(dgamma, part);
}

// dS[b][a][:] = softmax backward of the summed datt partials (sep_prod_kernel<false> over dO, Vp);  grid (A, B), one wave
__global__ __launch_bounds__(64) void sep_dsoft_kernel(const float* P, const float* att, float* dS, int A, int Ah) {
    const int a = blockIdx.x, j = threadIdx.x;
    const long long b = blockIdx.y;
    float da = 0.f, at = 0.f;
    if (j < Ah) {
        for (int s = 0; s < kSepSplit; ++s) da += P[((b * kSepSplit + s) * A + a) * Ah + j];
        at = att[(b * A + a) * Ah + j];
    }
    const float dot = wave_sum(da * at);
    if (j < Ah) dS[(b * A + a) * Ah + j] = at * (da - dot);
}

// gradients of the three flat operands; one thread per element of (dQf | dKp | dVp)
__global__ void sep_dops_kernel(SepGeo g, const float* dO, const float* att, const float* dS, const float* Qf, const float* Kp,
                                float* dQf, float* dKp, float* dVp, long long B) {
    const long long nq = (long long)g.Cq * g.N, nk = nq / 2, nv = (long long)g.C * g.N / 2, per = nq + nk + nv;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * per) return;
    const long long b = i / per;
    long long e = i - b * per;
    const int Ah = g.A / 2;
    const long long L = nq / g.A;
    if (e < nq) {                                   // dQf[a*L + l] = sum_j dS[a][j] * Kp[l*(A/2) + j]
        const long long a = e / L, l = e - a * L;
        const float* ds = dS + (b * g.A + a) * Ah;
        const float* k = Kp + b * nk + l * Ah;
        float s = 0.f;
        for (int j = 0; j < Ah; ++j) s += ds[j] * k[j];
        dQf[b * nq + e] = s;
        return;
    }
    e -= nq;
    if (e < nk) {                                   // dKp[l*(A/2) + j] = sum_a dS[a][j] * Qf[a*L + l]
        const long long l = e / Ah;
        const int j = (int)(e - l * Ah);
        float s = 0.f;
        for (int a = 0; a < g.A; ++a) s += dS[(b * g.A + a) * Ah + j] * Qf[b * nq + (long long)a * L + l];
        dKp[b * nk + e] = s;
        return;
    }
    e -= nk;                                        // dVp[r*(A/2) + j] = sum_a dO[r*A + a] * att[a][j]
    const long long r = e / Ah;
    const int j = (int)(e - r * Ah);
    float s = 0.f;
    for (int a = 0; a < g.A; ++a) s += dO[b * 2 * nv + r * g.A + a] * att[(b * g.A + a) * Ah + j];
    dVp[b * nv + e] = s;
}

// dqkv (channels-last, q | k | v columns; other columns untouched) from the flat gradients
template <typename T>
__global__ void sep_scatter_kernel(SepGeo g, const float* dQf, const float* dKp, const float* dVp, const unsigned char* ksel,
                                   const unsigned char* vsel, T* dqkv, int ldq, int koff, int voff, long long B) {
    const int ncol = 2 * g.Cq + g.C;
    const long long per = g.N * ncol;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * per) return;
    const long long b = i / per, e = i - b * per;
    const long long tok = e / ncol;
    int cc = (int)(e - tok * ncol);
    const int h = (int)(tok % g.H), w = (int)((tok / g.H) % g.W), t = (int)(tok / ((long long)g.H * g.W));
    int d0, d1, d2;
    if (g.axis == 0) { d0 = t; d1 = w; d2 = h; }
    else if (g.axis == 1) { d0 = w; d1 = t; d2 = h; }
    else { d0 = h; d1 = w; d2 = t; }
    const long long nq = (long long)g.Cq * g.N, nk = nq / 2, nv = (long long)g.C * g.N / 2;
    const int Ah = g.A / 2;
    T* dst = dqkv + ((size_t)b * g.N + tok) * ldq;
    if (cc < g.Cq) {
        stf(dst + cc, dQf[b * nq + (((long long)cc * g.A + d0) * g.D1 + d1) * g.D2 + d2]);
        return;
    }
    cc -= g.Cq;
    const bool isv = cc >= g.Cq;
    if (isv) cc -= g.Cq;
    const long long pe = (((long long)cc * Ah + (d0 >> 1)) * g.D1 + d1) * g.D2 + d2;
    const bool mine = (isv ? vsel[b * nv + pe] : ksel[b * nk + pe]) == (d0 & 1);
    const float gr = mine ? (isv ? dVp[b * nv + pe] : dKp[b * nk + pe]) : 0.f;
    stf(dst + (isv ? voff : koff) + cc, gr);
}

}  // namespace

#define S_ ((hipStream_t)stream)
#define BY_DTYPE(dtype, ...)                                                   \
    do {                                                                       \
        if ((dtype) == DVD_BF16) { using T = bf16_t; __VA_ARGS__; }            \
        else if ((dtype) == DVD_F32) { using T = float; __VA_ARGS__; }         \
        else return DVD_E_ARG;                                                 \
    } while (0)

static int sep_check(int T, int W, int H, int axis, int C, int Cq) {
    if (T <= 0 || W <= 0 || H <= 0 || C <= 0 || Cq <= 0 || axis < 0 || axis > 2) return DVD_E_ARG;
    if ((T | W | H) & 1) return DVD_E_SHAPE;                       // Attention.py:67: "T, W, H is not even"
    const int A = axis == 0 ? T : axis == 1 ? W : H;
    if (A > 64) return DVD_E_SHAPE;                                 // (tiles of sep_prod_kernel: 64 x 32 outputs)
    return DVD_OK;
}

extern "C" long long dvd_sepattn_work_floats(int T, int W, int H, int axis, int C, int Cq) {
    if (sep_check(T, W, H, axis, C, Cq) != DVD_OK) return 0;
    const SepGeo g = make_geo(T, W, H, axis, C, Cq);
    return (long long)Cq * g.N * 3 / 2 + (long long)C * g.N / 2 + (long long)g.A * (g.A / 2);     // Qf | Kp | Vp | att per clip
}

extern "C" int dvd_sepattn_forward(int dtype, const void* qkv, int ldq, int Cq, int koff, int voff, const void* x, int ldx,
                                   int C, const float* gamma, void* y, float* Qf, float* Kp, float* Vp, unsigned char* ksel,
                                   unsigned char* vsel, float* att, long long B, int T, int W, int H, int axis, void* stream) {
    if (!qkv || !x || !gamma || !y || !Qf || !Kp || !Vp || !ksel || !vsel || !att || B <= 0) return DVD_E_ARG;
    const int rc = sep_check(T, W, H, axis, C, Cq);
    if (rc != DVD_OK) return rc;
    const SepGeo g = make_geo(T, W, H, axis, C, Cq);
    const long long per = (long long)Cq * g.N * 3 / 2 + (long long)C * g.N / 2;
    BY_DTYPE(dtype, sep_gather_kernel<T><<<cdiv(B * per, 256), 256, 0, S_>>>(g, (const T*)qkv, ldq, koff, voff, Qf, Kp, Vp,
                                                                             ksel, vsel, B));
    // the partial score sums live in `y` until sep_out_kernel writes it (B * 8 * A * A/2 floats; y holds B * N * ldx elements)
    float* part = reinterpret_cast<float*>(y);
    if ((long long)kSepSplit * g.A * (g.A / 2) * 4 > g.N * ldx * (dtype == DVD_BF16 ? 2 : 4)) return DVD_E_SHAPE;       // (never: N >= 8 A)
    sep_prod_kernel<true><<<dim3(kSepSplit, (unsigned)B), 1024, 0, S_>>>(Qf, Kp, part, g.A, g.A / 2, (long long)Cq * g.D1 * g.D2);
    sep_softmax_kernel<<<dim3(g.A, (unsigned)B), 64, 0, S_>>>(part, att, g.A, g.A / 2);
    if (ldx != C &&      // padded channel columns of y must read zero afterwards (sep_out_kernel writes the real ones only)
        hipMemsetAsync(y, 0, (size_t)B * kSepSplit * g.A * (g.A / 2) * sizeof(float), S_) != hipSuccess) return DVD_E_LAUNCH;
    BY_DTYPE(dtype, sep_out_kernel<T><<<cdiv(B * C * g.N, 256), 256, 0, S_>>>(g, Vp, att, (const T*)x, ldx, gamma, (T*)y, B));
    return launch_status();
}

extern "C" int dvd_sepattn_backward(int dtype, const void* dy, int ldx, int C, int Cq, const float* gamma, const float* Qf,
                                    const float* Kp, const float* Vp, const unsigned char* ksel, const unsigned char* vsel,
                                    const float* att, float* dO, float* dS, float* dQf, float* dKp, float* dVp, void* dqkv,
                                    int ldq, int koff, int voff, float* dgamma, long long B, int T, int W, int H, int axis,
                                    void* stream) {
    if (!dy || !gamma || !Qf || !Kp || !Vp || !ksel || !vsel || !att || !dO || !dS || !dQf || !dKp || !dVp || !dqkv ||
        !dgamma || B <= 0)
        return DVD_E_ARG;
    const int rc = sep_check(T, W, H, axis, C, Cq);
    if (rc != DVD_OK) return rc;
    const SepGeo g = make_geo(T, W, H, axis, C, Cq);
    const long long per = (long long)Cq * g.N * 3 / 2 + (long long)C * g.N / 2;
    BY_DTYPE(dtype, sep_dout_kernel<T><<<cdiv(B * C * g.N, 256), 256, 0, S_>>>(g, Vp, att, (const T*)dy, ldx, gamma, dO,
                                                                               dgamma, B));
    // the partial sums of datt live in dQf until sep_dops_kernel writes it (B * 8 * A * A/2 of its B * Cq * N floats)
    if ((long long)kSepSplit * g.A * (g.A / 2) > (long long)Cq * g.N) return DVD_E_SHAPE;
    sep_prod_kernel<false><<<dim3(kSepSplit, (unsigned)B), 1024, 0, S_>>>(dO, Vp, dQf, g.A, g.A / 2, (long long)C * g.D1 * g.D2);
    sep_dsoft_kernel<<<dim3(g.A, (unsigned)B), 64, 0, S_>>>(dQf, att, dS, g.A, g.A / 2);
    sep_dops_kernel<<<cdiv(B * per, 256), 256, 0, S_>>>(g, dO, att, dS, Qf, Kp, dQf, dKp, dVp, B);
    BY_DTYPE(dtype, sep_scatter_kernel<T><<<cdiv(B * g.N * (2 * Cq + C), 256), 256, 0, S_>>>(
                        g, dQf, dKp, dVp, ksel, vsel, (T*)dqkv, ldq, koff, voff, B));
    return launch_status();
}
