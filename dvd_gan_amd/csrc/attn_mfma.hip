// Per-frame 2-D self attention of the discriminators (Module/Discriminators.py:100-119) on the matrix cores, bf16 mode:
//     A = softmax_j(q_i . k_j)   (no 1/sqrt(d) scale),   out_i = sum_j A_ij v_j,   y = gamma*out + x
// attn.hip computes the same on the vector pipe in fp32 and keeps the N x N map A (and dS) in HBM for the backward pass
// (134 MB per pass at N = 256, 512 frames); it stays the exact-mode path and the path of every shape not covered here.
// Here nothing N x N ever leaves the registers: the forward keeps one number per query (log-sum-exp of its score row), the
// backward recomputes the probabilities from q, k and that number.
//
// One workgroup = 128 consecutive tokens of one frame (4 waves, one 32-row block each: queries in the forward and dq pass, keys
// in the dk / dv pass).  The other token axis is walked in chunks of up to 256 tokens whose k and v (forward, dq pass) or q and dy
// (dk / dv pass) are staged in LDS in their natural [token][channel] layout -- any N that is a multiple of 32 (1024 tokens at the
// 128 x 128 configuration); everything else comes from global memory / registers in fragment shape.  All products are v_mfma_f32_32x32x16_bf16:
//     A operand: lane l holds row l & 31, k = 8 (l >> 5) .. + 7;   B operand: lane l holds column l & 31, same k
//     C / D    : lane l holds column l & 31, rows (r & 3) + 8 (r >> 2) + 4 (l >> 5), r = 0 .. 15
// Scores are computed TRANSPOSED (S^T = K Q^T: rows = keys, columns = queries), so a lane holds 16 scores of ONE query: the row
// softmax is a loop over registers plus one exchange with lane ^ 32, and the probabilities are already a B operand
// (k = keys) for the next product -- registers 8j .. 8j + 7 are k-step j, in the key order
//     key(j, half, e) = 16 j + 4 half + (e & 3) + 8 (e >> 2)
// which the A operand (V^T: rows = channels, k = keys) reproduces by addressing: its fragments come from the natural LDS image
// through ds_read_b64_tr_b16 (a 16-lane group reads 4 rows x 16 channels and hands lane i column i), rows chosen per lane.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int NMAX = 256;                 // tokens per staged chunk (the LDS images are sized for it)
constexpr int RSK = 48;                   // row pitch (bytes) of a [N][16] image: 16-byte aligned, 8-byte aligned chunks
__host__ __device__ constexpr int rsv(int C) { return C * 2 + 16; }

__device__ __forceinline__ f32x16 zero16() {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return z;
}
__device__ __forceinline__ f32x16 mma(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// [N][W] columns col0 .. col0 + W of a row-major bf16 matrix -> LDS image with row pitch rs (16-byte pieces)
__device__ __forceinline__ void stage(char* dst, int rs, const bf16_t* src, int ld, int col0, int W, int N, int tid) {
    const int cpr = W / 8;
    for (int i = tid; i < N * cpr; i += 256) {
        const int r = i / cpr, ck = i - r * cpr;
        *reinterpret_cast<u32x4*>(dst + r * rs + ck * 16) = *reinterpret_cast<const u32x4*>(src + (size_t)r * ld + col0 + ck * 8);
    }
}
// A fragment (rows = 32 channels starting at col, k = 8 tokens in the key order of the header) of the TRANSPOSE of a natural
// [token][channel] LDS image: tokens row0 + {0..3} and row0 + 8 + {0..3}, row0 = block + 16 j + 4 half
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int rs, int row0, int col, int lane) {
    const int i16 = lane & 15;
    const char* lo = img + (row0 + (i16 >> 2)) * rs + (col + (i16 & 3) * 4) * 2;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lo);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lo + 8 * rs));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ bf16x8 pack8(const float* v) {
    u32x4 r;
    r.x = pack2_bf16(v[0], v[1]); r.y = pack2_bf16(v[2], v[3]); r.z = pack2_bf16(v[4], v[5]); r.w = pack2_bf16(v[6], v[7]);
    return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ void unpack4(uint32_t lo, uint32_t hi, float (&v)[4]) {
    v[0] = __uint_as_float(lo << 16); v[1] = __uint_as_float(lo & 0xffff0000u);
    v[2] = __uint_as_float(hi << 16); v[3] = __uint_as_float(hi & 0xffff0000u);
}

// ---------------------------------------------------------------------------- forward
// qkv: [F][N][ldq], q = columns 0..15, k = 16..31, v = 32..32+C;  x, y, att: [F][N][C];  lse: [F][N]
template <int CB>
__global__ __launch_bounds__(256) void attn_fwd_mfma_kernel(const bf16_t* qkv, int ldq, const bf16_t* x, const float* gamma,
                                                            bf16_t* y, bf16_t* att, float* lse, int N) {
    constexpr int C = CB * 32, RSV = rsv(C);
    __shared__ __attribute__((aligned(16))) char sm[NMAX * RSK + NMAX * RSV];
    char* const Kl = sm;
    char* const Vl = sm + NMAX * RSK;
    const int qgroups = (N + 127) / 128;
    const int f = blockIdx.x / qgroups, qb = (blockIdx.x % qgroups) * 4 + (threadIdx.x >> 6);
    const int tid = threadIdx.x, lane = tid & 63;
    const bf16_t* qf = qkv + (size_t)f * N * ldq;
    const int l31 = lane & 31, h = lane >> 5, cblk = ((lane >> 4) & 1) * 16;
    const float g = *gamma;
    const bool active = qb * 32 < N;                       // (a frame of 64 tokens leaves two waves idle: they only stage)
    const int q = active ? qb * 32 + l31 : 0;
    const bf16x8 qfrag = *reinterpret_cast<const bf16x8*>(qf + (size_t)q * ldq + h * 8);
    f32x16 o[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) o[cb] = zero16();
    float m = -INFINITY, l = 0.f;
    for (int c0 = 0; c0 < N; c0 += NMAX) {
        const int nc = min(NMAX, N - c0);
        if (c0) __syncthreads();                           // every wave is done with the previous chunk's images
        stage(Kl, RSK, qf + (size_t)c0 * ldq, ldq, 16, 16, nc, tid);
        stage(Vl, RSV, qf + (size_t)c0 * ldq, ldq, 32, C, nc, tid);
        __syncthreads();
        if (!active) continue;
        for (int kb = 0; kb < nc / 32; ++kb) {
            const bf16x8 kfrag = *reinterpret_cast<const bf16x8*>(Kl + (kb * 32 + l31) * RSK + h * 16);
            const f32x16 s = mma(kfrag, qfrag, zero16());
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mn = fmaxf(m, mx);
            const float alpha = __expf(m - mn);
            float p[16], ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[r] = __expf(s[r] - mn); ps += p[r]; }
            l = l * alpha + ps;
            m = mn;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) o[cb] *= alpha;
            const bf16x8 pf[2] = {pack8(p), pack8(p + 8)};
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    o[cb] = mma(tr_frag(Vl, RSV, kb * 32 + 16 * j + 4 * h, cb * 32 + cblk, lane), pf[j], o[cb]);
        }
    }
    if (!active) return;
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.f / l;
    if (h == 0) lse[(size_t)f * N + q] = m + __logf(l);
    // this lane holds, of query q, channels cb * 32 + 8 i + 4 h + {0..3} (registers 4 i .. 4 i + 3)
    const size_t row = ((size_t)f * N + q) * C;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = cb * 32 + 8 * i + 4 * h;
            const uint2 xv = *reinterpret_cast<const uint2*>(x + row + c);
            float xf[4];
            unpack4(xv.x, xv.y, xf);
            const float o0 = o[cb][4 * i] * inv, o1 = o[cb][4 * i + 1] * inv, o2 = o[cb][4 * i + 2] * inv, o3 = o[cb][4 * i + 3] * inv;
            uint2 ov, yv;
            ov.x = pack2_bf16(o0, o1); ov.y = pack2_bf16(o2, o3);
            yv.x = pack2_bf16(g * o0 + xf[0], g * o1 + xf[1]); yv.y = pack2_bf16(g * o2 + xf[2], g * o3 + xf[3]);
            *reinterpret_cast<uint2*>(att + row + c) = ov;
            *reinterpret_cast<uint2*>(y + row + c) = yv;
        }
}

// ---------------------------------------------------------------------------- backward, query pass
// D_i = sum_c dy_ic out_ic;  dS_ij = gamma A_ij (sum_c dy_ic v_jc - D_i);  dq_i = sum_j dS_ij k_j;  dgamma += sum_i D_i.
// LDS: k and v of the frame.  Writes dq (columns 0..15 of dqkv) and D ([F][N], read by the key pass).
template <int CB>
__global__ __launch_bounds__(256) void attn_bwd_q_mfma_kernel(const bf16_t* qkv, int ldq, const bf16_t* dy, const bf16_t* att,
                                                              const float* gamma, const float* lse, float* Dout, bf16_t* dqkv,
                                                              float* dgamma, int N) {
    constexpr int C = CB * 32, RSV = rsv(C), KC = C / 16;
    __shared__ __attribute__((aligned(16))) char sm[NMAX * RSK + NMAX * RSV];
    char* const Kl = sm;
    char* const Vl = sm + NMAX * RSK;
    const int qgroups = (N + 127) / 128;
    const int f = blockIdx.x / qgroups, qb = (blockIdx.x % qgroups) * 4 + (threadIdx.x >> 6);
    const int tid = threadIdx.x, lane = tid & 63;
    const bf16_t* qf = qkv + (size_t)f * N * ldq;
    const int l31 = lane & 31, h = lane >> 5;
    const float g = *gamma;
    const bool active = qb * 32 < N;
    const int q = active ? qb * 32 + l31 : 0;
    const size_t row = ((size_t)f * N + q) * C;
    const bf16x8 qfrag = *reinterpret_cast<const bf16x8*>(qf + (size_t)q * ldq + h * 8);
    bf16x8 dyf[KC];                                   // B operand of dP^T = V dY^T: channels 16 kc + 8 h .. + 7 of row q
    float Dp = 0.f;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(dy + row + 16 * kc + 8 * h);
        const u32x4 b = *reinterpret_cast<const u32x4*>(att + row + 16 * kc + 8 * h);
        dyf[kc] = __builtin_bit_cast(bf16x8, a);
        Dp += __uint_as_float(a.x << 16) * __uint_as_float(b.x << 16) + __uint_as_float(a.x & 0xffff0000u) * __uint_as_float(b.x & 0xffff0000u)
            + __uint_as_float(a.y << 16) * __uint_as_float(b.y << 16) + __uint_as_float(a.y & 0xffff0000u) * __uint_as_float(b.y & 0xffff0000u)
            + __uint_as_float(a.z << 16) * __uint_as_float(b.z << 16) + __uint_as_float(a.z & 0xffff0000u) * __uint_as_float(b.z & 0xffff0000u)
            + __uint_as_float(a.w << 16) * __uint_as_float(b.w << 16) + __uint_as_float(a.w & 0xffff0000u) * __uint_as_float(b.w & 0xffff0000u);
    }
    Dp += __shfl_xor(Dp, 32, 64);
    const float ls = lse[(size_t)f * N + q];
    f32x16 dq = zero16();
    for (int c0 = 0; c0 < N; c0 += NMAX) {
        const int nc = min(NMAX, N - c0);
        if (c0) __syncthreads();
        stage(Kl, RSK, qf + (size_t)c0 * ldq, ldq, 16, 16, nc, tid);
        stage(Vl, RSV, qf + (size_t)c0 * ldq, ldq, 32, C, nc, tid);
        __syncthreads();
        if (!active) continue;
        for (int kb = 0; kb < nc / 32; ++kb) {
            const char* krow = Kl + (kb * 32 + l31) * RSK;
            const char* vrow = Vl + (kb * 32 + l31) * RSV;
            const f32x16 s = mma(*reinterpret_cast<const bf16x8*>(krow + h * 16), qfrag, zero16());
            f32x16 dp = zero16();
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) dp = mma(*reinterpret_cast<const bf16x8*>(vrow + (16 * kc + 8 * h) * 2), dyf[kc], dp);
            float ds[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) ds[r] = g * __expf(s[r] - ls) * (dp[r] - Dp);
            // dq^T (16 valid rows of 32) += K^T dS^T: the lanes of the upper 16 rows read the same image, their results are dropped
#pragma unroll
            for (int j = 0; j < 2; ++j) dq = mma(tr_frag(Kl, RSK, kb * 32 + 16 * j + 4 * h, 0, lane), pack8(ds + 8 * j), dq);
        }
    }
    if (active) {
        bf16_t* dst = dqkv + ((size_t)f * N + q) * ldq;
        uint2 a, b;
        a.x = pack2_bf16(dq[0], dq[1]); a.y = pack2_bf16(dq[2], dq[3]);          // d = 4 h + {0..3}
        b.x = pack2_bf16(dq[4], dq[5]); b.y = pack2_bf16(dq[6], dq[7]);          // d = 8 + 4 h + {0..3}
        *reinterpret_cast<uint2*>(dst + 4 * h) = a;
        *reinterpret_cast<uint2*>(dst + 8 + 4 * h) = b;
        if (h == 0) Dout[(size_t)f * N + q] = Dp;
    }
}

// dgamma += sum over all (frame, query) of D, in a fixed order (one block; the query pass leaves D complete) -- the first form
// added per-wave partial sums with fp32 atomics, whose order changed from run to run
__global__ __launch_bounds__(1024) void attn_dgamma_kernel(const float* D, long long n, float* dgamma) {
    __shared__ float sh[16];
    float a = 0.f;
    for (long long i = threadIdx.x; i < n; i += 1024) a += D[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += sh[w];
        *dgamma += t;
    }
}

// ---------------------------------------------------------------------------- backward, key pass
// dv_j = gamma sum_i A_ij dy_i;   dk_j = sum_i dS_ij q_i.   LDS: q and dy of the frame, lse and D.
template <int CB>
__global__ __launch_bounds__(256) void attn_bwd_kv_mfma_kernel(const bf16_t* qkv, int ldq, const bf16_t* dy, const float* gamma,
                                                               const float* lse, const float* Din, bf16_t* dqkv, int N) {
    constexpr int C = CB * 32, RSV = rsv(C), KC = C / 16;
    __shared__ __attribute__((aligned(16))) char sm[NMAX * RSK + NMAX * RSV + NMAX * 8];
    char* const Ql = sm;
    char* const Yl = sm + NMAX * RSK;
    float* const lsl = reinterpret_cast<float*>(sm + NMAX * RSK + NMAX * RSV);
    float* const Dl = lsl + NMAX;
    const int kgroups = (N + 127) / 128;
    const int f = blockIdx.x / kgroups, kb = (blockIdx.x % kgroups) * 4 + (threadIdx.x >> 6);
    const int tid = threadIdx.x, lane = tid & 63;
    const bf16_t* qf = qkv + (size_t)f * N * ldq;
    const int l31 = lane & 31, h = lane >> 5, cblk = ((lane >> 4) & 1) * 16;
    const float g = *gamma;
    const bool active = kb * 32 < N;
    const int key = active ? kb * 32 + l31 : 0;
    const bf16_t* krow = qf + (size_t)key * ldq;
    const bf16x8 kfrag = *reinterpret_cast<const bf16x8*>(krow + 16 + h * 8);          // B operand: k = d
    bf16x8 vf[KC];                                                                      // B operand: k = channels
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) vf[kc] = *reinterpret_cast<const bf16x8*>(krow + 32 + 16 * kc + 8 * h);
    f32x16 dv[CB], dk = zero16();
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) dv[cb] = zero16();
    for (int c0 = 0; c0 < N; c0 += NMAX) {
        const int nc = min(NMAX, N - c0);
        if (c0) __syncthreads();
        stage(Ql, RSK, qf + (size_t)c0 * ldq, ldq, 0, 16, nc, tid);
        stage(Yl, RSV, dy + ((size_t)f * N + c0) * C, C, 0, C, nc, tid);
        for (int i = tid; i < nc; i += 256) { lsl[i] = lse[(size_t)f * N + c0 + i]; Dl[i] = Din[(size_t)f * N + c0 + i]; }
        __syncthreads();
        if (!active) continue;
        for (int qb = 0; qb < nc / 32; ++qb) {
            const char* qrow = Ql + (qb * 32 + l31) * RSK;
            const char* yrow = Yl + (qb * 32 + l31) * RSV;
            const f32x16 s = mma(*reinterpret_cast<const bf16x8*>(qrow + h * 16), kfrag, zero16());   // rows = queries, columns = keys
            f32x16 dp = zero16();
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) dp = mma(*reinterpret_cast<const bf16x8*>(yrow + (16 * kc + 8 * h) * 2), vf[kc], dp);
            float p[16], ds[16];
#pragma unroll
            for (int i = 0; i < 4; ++i) {                  // registers 4 i .. 4 i + 3 = queries qb * 32 + 8 i + 4 h + {0..3}
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lsl + qb * 32 + 8 * i + 4 * h);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(Dl + qb * 32 + 8 * i + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p[4 * i + e] = __expf(s[4 * i + e] - l4[e]);
                    ds[4 * i + e] = g * p[4 * i + e] * (dp[4 * i + e] - d4[e]);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 pf = pack8(p + 8 * j);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    dv[cb] = mma(tr_frag(Yl, RSV, qb * 32 + 16 * j + 4 * h, cb * 32 + cblk, lane), pf, dv[cb]);
                dk = mma(tr_frag(Ql, RSK, qb * 32 + 16 * j + 4 * h, 0, lane), pack8(ds + 8 * j), dk);
            }
        }
    }
    if (!active) return;
    bf16_t* dst = dqkv + ((size_t)f * N + key) * ldq;
    uint2 a, b;
    a.x = pack2_bf16(dk[0], dk[1]); a.y = pack2_bf16(dk[2], dk[3]);
    b.x = pack2_bf16(dk[4], dk[5]); b.y = pack2_bf16(dk[6], dk[7]);
    *reinterpret_cast<uint2*>(dst + 16 + 4 * h) = a;
    *reinterpret_cast<uint2*>(dst + 24 + 4 * h) = b;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 v;
            v.x = pack2_bf16(g * dv[cb][4 * i], g * dv[cb][4 * i + 1]);
            v.y = pack2_bf16(g * dv[cb][4 * i + 2], g * dv[cb][4 * i + 3]);
            *reinterpret_cast<uint2*>(dst + 32 + cb * 32 + 8 * i + 4 * h) = v;
        }
}

}  // namespace

#define S_ ((hipStream_t)stream)

// 1 when the MFMA kernels serve this attention call: bf16 storage, q | k | v at columns 0 | 16 | 32 with 16 query / key channels,
// 32 / 64 / 128 value channels without padding, 32 .. 4096 tokens per frame in whole 32-token blocks
extern "C" int dvd_attention_mfma_ok(int dtype, int ldq, int dq, int koff, int voff, int ldx, int C, int N) {
    return dtype == DVD_BF16 && dq == 16 && koff == 16 && voff == 32 && (C == 32 || C == 64 || C == 128) && ldx == C &&
           ldq >= 32 + C && !(ldq & 7) && N >= 32 && N <= 4096 && !(N & 31);
}

extern "C" int dvd_attention_mfma_forward(const void* qkv, int ldq, const void* x, int C, const float* gamma, void* y,
                                          void* att_out, float* lse, long long frames, int N, void* stream) {
    if (!qkv || !x || !gamma || !y || !att_out || !lse || frames <= 0) return DVD_E_ARG;
    if (!dvd_attention_mfma_ok(DVD_BF16, ldq, 16, 16, 32, C, C, N) || frames * ((N + 127) / 128) >= (1ll << 31)) return DVD_E_SHAPE;
#define FWD(CB_) attn_fwd_mfma_kernel<CB_><<<(unsigned)(frames * ((N + 127) / 128)), 256, 0, S_>>>((const bf16_t*)qkv, ldq, (const bf16_t*)x, gamma, \
                                                                            (bf16_t*)y, (bf16_t*)att_out, lse, N)
    if (C == 128) FWD(4); else if (C == 64) FWD(2); else FWD(1);
#undef FWD
    return launch_status();
}

// D: scratch [frames][N] floats.  dqkv: every column of q | k | v is written.  dgamma is ADDED to.
extern "C" int dvd_attention_mfma_backward(const void* qkv, int ldq, const void* dy, int C, const float* gamma,
                                           const void* att_out, const float* lse, float* D, void* dqkv, float* dgamma,
                                           long long frames, int N, void* stream) {
    if (!qkv || !dy || !gamma || !att_out || !lse || !D || !dqkv || frames <= 0) return DVD_E_ARG;
    if (!dvd_attention_mfma_ok(DVD_BF16, ldq, 16, 16, 32, C, C, N) || frames * ((N + 127) / 128) >= (1ll << 31)) return DVD_E_SHAPE;
#define BWD(CB_)                                                                                                              \
    do {                                                                                                                      \
        attn_bwd_q_mfma_kernel<CB_><<<(unsigned)(frames * ((N + 127) / 128)), 256, 0, S_>>>((const bf16_t*)qkv, ldq, (const bf16_t*)dy,               \
                                                                      (const bf16_t*)att_out, gamma, lse, D, (bf16_t*)dqkv, dgamma, N); \
        attn_bwd_kv_mfma_kernel<CB_><<<(unsigned)(frames * ((N + 127) / 128)), 256, 0, S_>>>((const bf16_t*)qkv, ldq, (const bf16_t*)dy, gamma, lse, D, \
                                                                       (bf16_t*)dqkv, N);                                      \
    } while (0)
    if (C == 128) BWD(4); else if (C == 64) BWD(2); else BWD(1);
#undef BWD
    if (dgamma) attn_dgamma_kernel<<<1, 1024, 0, S_>>>(D, frames * N, dgamma);
    return launch_status();
}
