// Per-frame 2-D self attention of the discriminators (Module/Discriminators.py:100-119):
//     A = softmax_j(q_i . k_j)   (no 1/sqrt(d) scale),   out_i = sum_j A_ij v_j,   y = gamma*out + x
// q|k|v come from ONE fused 1x1 convolution (columns [0,dq) | [koff,koff+dq) | [voff,voff+C) of
// `qkv`).  N = H*W tokens per frame (256 / 64 at the 64x64 configuration), dq = C/8.
// fp32 arithmetic on the vector pipe: the whole attention is < 0.2 % of the step's FLOPs, so the
// kernels are organised for exactness and simple, coalesced access rather than MFMA.
// A ([F][N][Nk] fp32) and out are kept for the backward pass.
// The kernels take the keys / values from a separate row set (`kv`, Nk rows per frame): Nk = N and kv = qkv for the
// discriminators' 2-D block; for the spatio-temporal block of Module/Attention.py:114-185 the keys and values are the
// 2x2x2 max-pooled projections (Nk = N/8) and the query axis runs over all T*H*W tokens of a clip.
#include "common.h"

namespace {

// QB = query rows per workgroup (template parameter: 16, or 8 when 16 rows of scores do not fit 64 KB of LDS --
// N = 1024 tokens, the 128 x 128 configuration)

__device__ float blk_sum(float v, float* sh) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sh[w] = v;
    __syncthreads();
    float t = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return t;
}

// Forward.  LDS: St [Nk][QB] scores -> probabilities (transposed: the QB values of one key are contiguous),
// ql [QB][dqp] the block's query rows as floats, red [PARTS][QB][C8*8] partial outputs.
//   scores : thread j owns key j: its row is read once with 16-byte loads, the QB query rows come from LDS
//   softmax: one wave per query row, wave-reduced max / sum
//   A v    : thread = (8-channel group, 4-row group, key part); 32 accumulators, one 16-byte v load and one
//            16-byte LDS read of St per key; the key parts are folded through LDS
// (the first version did one scalar load per multiply: 8 TF/s, 7 ms per step for D_s at N = 256)
template <typename T, int QB>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* qkv, int ldq, int dq, const T* kv, int ldk, int koff,
                                                       int voff, const T* x, int ldx, int C, const float* gamma, T* y,
                                                       T* att_out, float* A, int N, int Nk) {
    extern __shared__ float S[];
    const int f = blockIdx.y, i0 = blockIdx.x * QB, tid = threadIdx.x;
    const int dqp = (dq + 7) & ~7, C8 = (C + 7) / 8;
    float* St = S;                                  // [Nk][QB]
    float* ql = S + (size_t)Nk * QB;                // [QB][dqp]
    float* red = ql + QB * dqp;                     // [PARTS][QB][C8*8]
    const T* qf = qkv + (size_t)f * N * ldq;
    const T* kf = kv + (size_t)f * Nk * ldk;
    for (int idx = tid; idx < QB * dqp; idx += 256) {
        const int i = idx / dqp, d = idx - i * dqp;
        ql[idx] = (i0 + i < N && d < dq) ? ldf(qf + (size_t)(i0 + i) * ldq + d) : 0.f;
    }
    __syncthreads();
    // ---- scores ----
    for (int j = tid; j < Nk; j += 256) {
        float acc[QB];
#pragma unroll
        for (int i = 0; i < QB; ++i) acc[i] = 0.f;
        const T* k = kf + (size_t)j * ldk + koff;
        for (int d = 0; d < dqp; d += 8) {
            float kk[8];
            load8<T>(k + d, kk);                    // columns >= dq only ever meet zeros of ql
#pragma unroll
            for (int i = 0; i < QB; ++i) {
                const float* q = ql + i * dqp + d;
                acc[i] += q[0] * kk[0] + q[1] * kk[1] + q[2] * kk[2] + q[3] * kk[3] + q[4] * kk[4] + q[5] * kk[5] +
                          q[6] * kk[6] + q[7] * kk[7];
            }
        }
#pragma unroll
        for (int i = 0; i < QB; ++i) St[j * QB + i] = acc[i];
    }
    __syncthreads();
    // ---- row softmax (one wave per row) ----
    const int wave = tid >> 6, lane = tid & 63;
    for (int i = wave; i < QB; i += 4) {
        float m = -INFINITY;
        for (int j = lane; j < Nk; j += 64) m = fmaxf(m, St[j * QB + i]);
        m = wave_max(m);
        float sum = 0.f;
        for (int j = lane; j < Nk; j += 64) { const float e = expf(St[j * QB + i] - m); St[j * QB + i] = e; sum += e; }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        for (int j = lane; j < Nk; j += 64) {
            const float a = St[j * QB + i] * inv;
            St[j * QB + i] = a;
            if (A && i0 + i < N) A[((size_t)f * N + i0 + i) * Nk + j] = a;
        }
    }
    __syncthreads();
    // ---- out = A v ----
    constexpr int RG = QB / 4;                      // 4-row groups
    const int items = C8 * RG;                      // (channel group, row group) pairs
    const int parts = items >= 256 ? 1 : (256 / items > 4 ? 4 : 256 / items);
    const float g = *gamma;
    if (tid < items * parts) {
        const int part = tid / items, it = tid - part * items;
        const int cg = it % C8, rg = it / C8;
        float acc[4][8];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[r][c] = 0.f;
        for (int j = part; j < Nk; j += parts) {
            float vv[8];
            load8<T>(kf + (size_t)j * ldk + voff + cg * 8, vv);
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(St + j * QB + rg * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[r][c] += a4[r] * vv[c];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) red[((size_t)part * QB + rg * 4 + r) * (C8 * 8) + cg * 8 + c] = acc[r][c];
    }
    __syncthreads();
    for (int idx = tid; idx < QB * C8; idx += 256) {
        const int i = idx / C8, cg = idx - i * C8;
        if (i0 + i >= N) continue;
        float o[8], xv[8], yv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] = 0.f;
        for (int part = 0; part < parts; ++part)
#pragma unroll
            for (int c = 0; c < 8; ++c) o[c] += red[((size_t)part * QB + i) * (C8 * 8) + cg * 8 + c];
        const size_t off = ((size_t)f * N + i0 + i) * ldx + cg * 8;
        load8<T>(x + off, xv);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const bool ok = cg * 8 + c < C;
            o[c] = ok ? o[c] : 0.f;
            yv[c] = ok ? g * o[c] + xv[c] : 0.f;
        }
        if (att_out) store8<T>(att_out + off, o);
        store8<T>(y + off, yv);
    }
}

// Backward, row pass: dA, dS (written over `dS` [F][N][N]), dq, dgamma.
template <typename T, int QB>
__global__ __launch_bounds__(256) void attn_bwd_rows_kernel(const T* kv, int ldk, int dq, int koff, int voff, const T* dy,
                                                            int ldx, int C, const float* gamma, const T* att_out,
                                                            const float* A, float* dS, T* dqo, int ldq, float* dgamma,
                                                            int N, int Nk) {
    extern __shared__ float S[];                   // [QB][Nk] dA -> dS
    __shared__ float sh[4];
    const int f = blockIdx.y, i0 = blockIdx.x * QB, tid = threadIdx.x;
    const T* kf = kv + (size_t)f * Nk * ldk;
    const float g = *gamma;
    (void)dgamma;                                  // dgamma: attn_dgamma_kernel below (fixed summation order)
    // dA[i][j] = gamma * sum_c dy[i][c] v[j][c]: the block's QB rows of dy are staged in LDS as
    // floats; thread j streams its v row with 16-byte loads and keeps QB accumulators.
    float* dyl = S + QB * Nk;                       // [QB][C]
    for (int idx = tid; idx < QB * C; idx += 256) {
        const int i = idx / C, c = idx - i * C;
        dyl[idx] = (i0 + i < N) ? ldf(dy + ((size_t)f * N + i0 + i) * ldx + c) : 0.f;
    }
    __syncthreads();
    for (int j = tid; j < Nk; j += 256) {
        float acc[QB];
#pragma unroll
        for (int i = 0; i < QB; ++i) acc[i] = 0.f;
        const T* v = kf + (size_t)j * ldk + voff;
        for (int c = 0; c < C; c += 8) {
            float vv[8];
            load8<T>(v + c, vv);                    // C is a multiple of 8, voff too
#pragma unroll
            for (int i = 0; i < QB; ++i) {
                const float* d = dyl + i * C + c;
                acc[i] += d[0] * vv[0] + d[1] * vv[1] + d[2] * vv[2] + d[3] * vv[3] + d[4] * vv[4] + d[5] * vv[5] +
                          d[6] * vv[6] + d[7] * vv[7];
            }
        }
#pragma unroll
        for (int i = 0; i < QB; ++i) S[i * Nk + j] = acc[i] * g;
    }
    __syncthreads();
    // dS = A * (dA - sum_j A dA)
    const int wave = tid >> 6, lane = tid & 63;
    for (int i = wave; i < QB; i += 4) {
        if (i0 + i >= N) continue;
        const float* a = A + ((size_t)f * N + i0 + i) * Nk;
        float dot = 0.f;
        for (int j = lane; j < Nk; j += 64) dot += a[j] * S[i * Nk + j];
        dot = wave_sum(dot);
        for (int j = lane; j < Nk; j += 64) {
            const float v = a[j] * (S[i * Nk + j] - dot);
            S[i * Nk + j] = v;
            dS[((size_t)f * N + i0 + i) * Nk + j] = v;
        }
    }
    __syncthreads();
    // dq[i][d] = sum_j dS[i][j] k[j][d]
    for (int idx = tid; idx < QB * dq; idx += 256) {
        const int i = idx / dq, d = idx - i * dq;
        if (i0 + i >= N) continue;
        float s = 0.f;
        for (int j = 0; j < Nk; ++j) s += S[i * Nk + j] * ldf(kf + (size_t)j * ldk + koff + d);
        stf(dqo + ((size_t)f * N + i0 + i) * ldq + d, s);
    }
}

// Backward, column pass: dv[j][c] = gamma * sum_i A[i][j] dy[i][c];  dk[j][d] = sum_i dS[i][j] q[i][d]
template <typename T, int QB>
__global__ __launch_bounds__(256) void attn_bwd_cols_kernel(const T* qkv, int ldq, int dq, int koff, int voff, const T* dy,
                                                            int ldx, int C, const float* gamma, const float* A,
                                                            const float* dS, T* dkv, int ldk, int N, int Nk) {
    extern __shared__ float S[];                   // [N][QB] slice of A, then of dS (column block j0..j0+QB)
    const int f = blockIdx.y, j0 = blockIdx.x * QB, tid = threadIdx.x;
    const T* qf = qkv + (size_t)f * N * ldq;
    const float g = *gamma;
    for (int idx = tid; idx < N * QB; idx += 256) {
        const int i = idx / QB, jj = idx - i * QB;
        S[idx] = (j0 + jj < Nk) ? A[((size_t)f * N + i) * Nk + j0 + jj] : 0.f;
    }
    __syncthreads();
    for (int idx = tid; idx < (QB / 4) * C; idx += 256) {
        const int jb = (idx / C) * 4, c = idx % C;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < N; ++i) {
            const float d = ldf(dy + ((size_t)f * N + i) * ldx + c);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += S[i * QB + jb + r] * d;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (j0 + jb + r < Nk) stf(dkv + ((size_t)f * Nk + j0 + jb + r) * ldk + voff + c, g * acc[r]);
    }
    __syncthreads();
    for (int idx = tid; idx < N * QB; idx += 256) {
        const int i = idx / QB, jj = idx - i * QB;
        S[idx] = (j0 + jj < Nk) ? dS[((size_t)f * N + i) * Nk + j0 + jj] : 0.f;
    }
    __syncthreads();
    for (int idx = tid; idx < QB * dq; idx += 256) {
        const int jj = idx / dq, d = idx - jj * dq;
        if (j0 + jj >= Nk) continue;
        float s = 0.f;
        for (int i = 0; i < N; ++i) s += S[i * QB + jj] * ldf(qf + (size_t)i * ldq + d);
        stf(dkv + ((size_t)f * Nk + j0 + jj) * ldk + koff + d, s);
    }
}

}  // namespace

#define S_ ((hipStream_t)stream)
#define BY_DTYPE(dtype, ...)                                                   \
    do {                                                                       \
        if ((dtype) == DVD_BF16) { using T = bf16_t; __VA_ARGS__; }            \
        else if ((dtype) == DVD_F32) { using T = float; __VA_ARGS__; }         \
        else return DVD_E_ARG;                                                 \
    } while (0)

static int attention_fwd(int dtype, const void* q, int ldq, int dq, const void* kv, int ldk, int koff, int voff, const void* x,
                         int ldx, int C, const float* gamma, void* y, void* att_out, float* A, long long frames, int N, int Nk,
                         void* stream) {
    if (!q || !kv || !x || !gamma || !y || frames <= 0 || N <= 0 || Nk <= 0 || dq <= 0 || C <= 0) return DVD_E_ARG;
    if (frames > 65535) return DVD_E_SHAPE;
    if ((ldq & 7) || (ldk & 7) || (ldx & 7) || (koff & 7) || (voff & 7) || C > ldx) return DVD_E_SHAPE;   // 16-byte vectors
    const int dqp = (dq + 7) & ~7, c8 = (C + 7) / 8 * 8;
    auto lds_bytes = [&](int qb) { return ((size_t)qb * Nk + (size_t)qb * dqp + (size_t)4 * qb * c8) * sizeof(float); };
    const int qb = lds_bytes(16) <= 64 * 1024 ? 16 : 8;
    if (lds_bytes(qb) > 64 * 1024) return DVD_E_SHAPE;
    dim3 grid(cdiv(N, qb), (unsigned)frames);
    const size_t sh = lds_bytes(qb);
#define ATT_FWD(QB_) BY_DTYPE(dtype, attn_fwd_kernel<T, QB_><<<grid, 256, sh, S_>>>((const T*)q, ldq, dq, (const T*)kv, ldk, koff, \
                                       voff, (const T*)x, ldx, C, gamma, (T*)y, (T*)att_out, A, N, Nk))
    if (qb == 16) ATT_FWD(16); else ATT_FWD(8);
#undef ATT_FWD
    return launch_status();
}

// dgamma += sum over all rows and channels of dy * att_out in a FIXED order: 256 blocks leave partial sums (in the dS scratch, dead
// behind the column pass), one block adds them up (per-block atomics made the gradient of gamma differ from run to run)
template <typename T>
__global__ __launch_bounds__(256) void attn_dgamma_partial_kernel(const T* dy, const T* att_out, long long rows, int C, int ldx, float* partial) {
    __shared__ float shw[4];
    const int cg = C / 8;
    float a = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < rows * cg; i += (long long)gridDim.x * 256) {
        const long long r = i / cg;
        const int c = (int)(i - r * cg) * 8;
        float u[8], v[8];
        load8<T>(dy + (size_t)r * ldx + c, u);
        load8<T>(att_out + (size_t)r * ldx + c, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) a += u[k] * v[k];
    }
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) shw[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (shw[0] + shw[1]) + (shw[2] + shw[3]);
}
__global__ void attn_dgamma_final_kernel(const float* partial, int n, float* dgamma) {
    if (threadIdx.x || blockIdx.x) return;
    float t = 0.f;
    for (int i = 0; i < n; ++i) t += partial[i];
    *dgamma += t;
}

static int attention_bwd(int dtype, const void* q, int ldq, int dq, const void* kv, int ldk, int koff, int voff, const void* dy,
                         int ldx, int C, const float* gamma, const void* att_out, const float* A, float* dS, void* dqo,
                         void* dkv, float* dgamma, long long frames, int N, int Nk, void* stream) {
    if (!q || !kv || !dy || !gamma || !att_out || !A || !dS || !dqo || !dkv || frames <= 0 || N <= 0 || Nk <= 0) return DVD_E_ARG;
    if (frames > 65535 || (C & 7)) return DVD_E_SHAPE;
    const int qb = ((size_t)16 * (Nk + C) * sizeof(float) <= 64 * 1024 && (size_t)16 * N * sizeof(float) <= 64 * 1024) ? 16 : 8;
    if ((size_t)qb * (Nk + C) * sizeof(float) > 64 * 1024 || (size_t)qb * N * sizeof(float) > 64 * 1024) return DVD_E_SHAPE;
    dim3 grid_r(cdiv(N, qb), (unsigned)frames), grid_c(cdiv(Nk, qb), (unsigned)frames);
    const size_t sh_rows = (size_t)qb * (Nk + C) * sizeof(float), sh_cols = (size_t)qb * N * sizeof(float);
#define ATT_BWD(QB_)                                                                                                    \
    do {                                                                                                                \
        BY_DTYPE(dtype, attn_bwd_rows_kernel<T, QB_><<<grid_r, 256, sh_rows, S_>>>((const T*)kv, ldk, dq, koff, voff,   \
                            (const T*)dy, ldx, C, gamma, (const T*)att_out, A, dS, (T*)dqo, ldq, dgamma, N, Nk));       \
        BY_DTYPE(dtype, attn_bwd_cols_kernel<T, QB_><<<grid_c, 256, sh_cols, S_>>>((const T*)q, ldq, dq, koff, voff,    \
                            (const T*)dy, ldx, C, gamma, A, dS, (T*)dkv, ldk, N, Nk));                                  \
    } while (0)
    if (qb == 16) ATT_BWD(16); else ATT_BWD(8);
#undef ATT_BWD
    if (dgamma) {
        const int nb = (long long)frames * N * Nk >= 256 ? 256 : 1;          // partials live in dS ([frames][N][Nk] floats)
        BY_DTYPE(dtype, attn_dgamma_partial_kernel<T><<<nb, 256, 0, S_>>>((const T*)dy, (const T*)att_out, frames * N, C, ldx, dS));
        attn_dgamma_final_kernel<<<1, 64, 0, S_>>>(dS, nb, dgamma);
    }
    return launch_status();
}

extern "C" int dvd_attention_forward(int dtype, const void* qkv, int ldq, int dq, int koff, int voff, const void* x,
                                     int ldx, int C, const float* gamma, void* y, void* att_out, float* A,
                                     long long frames, int N, void* stream) {
    return attention_fwd(dtype, qkv, ldq, dq, qkv, ldq, koff, voff, x, ldx, C, gamma, y, att_out, A, frames, N, N, stream);
}

extern "C" int dvd_attention_backward(int dtype, const void* qkv, int ldq, int dq, int koff, int voff, const void* dy,
                                      int ldx, int C, const float* gamma, const void* att_out, const float* A,
                                      float* dS, void* dqkv, float* dgamma, long long frames, int N, void* stream) {
    return attention_bwd(dtype, qkv, ldq, dq, qkv, ldq, koff, voff, dy, ldx, C, gamma, att_out, A, dS, dqkv, dqkv, dgamma,
                         frames, N, N, stream);
}

// Queries from `q` (N rows per frame), keys / values from `kv` (Nk rows per frame, columns [koff,koff+dq) / [voff,voff+C)).
extern "C" int dvd_attention_kv_forward(int dtype, const void* q, int ldq, int dq, const void* kv, int ldk, int koff, int voff,
                                        const void* x, int ldx, int C, const float* gamma, void* y, void* att_out, float* A,
                                        long long frames, int N, int Nk, void* stream) {
    return attention_fwd(dtype, q, ldq, dq, kv, ldk, koff, voff, x, ldx, C, gamma, y, att_out, A, frames, N, Nk, stream);
}

extern "C" int dvd_attention_kv_backward(int dtype, const void* q, int ldq, int dq, const void* kv, int ldk, int koff, int voff,
                                         const void* dy, int ldx, int C, const float* gamma, const void* att_out,
                                         const float* A, float* dS, void* dq_out, void* dkv_out, float* dgamma,
                                         long long frames, int N, int Nk, void* stream) {
    return attention_bwd(dtype, q, ldq, dq, kv, ldk, koff, voff, dy, ldx, C, gamma, att_out, A, dS, dq_out, dkv_out, dgamma,
                         frames, N, Nk, stream);
}
