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

// Qf, Kp, Vp (+ the max-pool winners) from the channels-last projection.  Round 5: a tile of TP positions of the (d1, d2) plane for
// the two attended coordinates 2 dp, 2 dp + 1 goes through LDS -- the rows of `qkv` are read along their channels (coalesced), the
// flat operands are written along the plane (TP consecutive floats per channel).  The first form (one thread per flat element)
// read 2 bytes from a different 512-byte row in every lane.
template <typename T, int TP>
__global__ __launch_bounds__(256) void sep_gather_kernel(SepGeo g, const T* qkv, int ldq, int koff, int voff, float* Qf, float* Kp, float* Vp,
                                                         unsigned char* ksel, unsigned char* vsel) {
    extern __shared__ float tile[];                 // [2][TP][NC + 1]
    const int NC = 2 * g.Cq + g.C, NCP = NC + 1;
    const int PL = g.D1 * g.D2, Ah = g.A / 2;
    const int p0 = blockIdx.x * TP, dp = blockIdx.y, tid = threadIdx.x;
    const long long b = blockIdx.z;
    const long long nq = (long long)g.Cq * g.N, nk = nq / 2, nv = (long long)g.C * g.N / 2;
    const T* base = qkv + (size_t)b * g.N * ldq;
    for (int idx = tid; idx < 2 * TP * NC; idx += 256) {
        const int col = idx % NC, rp = idx / NC, p = rp % TP, s2 = rp / TP;
        float v = 0.f;
        if (p0 + p < PL) {
            const int pos = p0 + p, d1 = pos / g.D2, d2 = pos - d1 * g.D2;
            const int src = col < g.Cq ? col : col < 2 * g.Cq ? koff + col - g.Cq : voff + col - 2 * g.Cq;
            v = ldf(base + (size_t)token(g, 2 * dp + s2, d1, d2) * ldq + src);
        }
        tile[(s2 * TP + p) * NCP + col] = v;
    }
    __syncthreads();
    for (int idx = tid; idx < 2 * g.Cq * TP; idx += 256) {          // q: both attended coordinates
        const int p = idx % TP, r = idx / TP, c = r % g.Cq, s2 = r / g.Cq;
        if (p0 + p < PL) Qf[b * nq + ((long long)c * g.A + 2 * dp + s2) * PL + p0 + p] = tile[(s2 * TP + p) * NCP + c];
    }
    for (int idx = tid; idx < (g.Cq + g.C) * TP; idx += 256) {      // k, v: max over the pair; ties -> the first element, like F.max_pool3d
        const int p = idx % TP, c = idx / TP;
        if (p0 + p >= PL) continue;
        const float a0 = tile[p * NCP + g.Cq + c], a1 = tile[(TP + p) * NCP + g.Cq + c];
        const bool second = a1 > a0;
        if (c < g.Cq) {
            const long long e = ((long long)c * Ah + dp) * PL + p0 + p;
            Kp[b * nk + e] = second ? a1 : a0; ksel[b * nk + e] = second;
        } else {
            const long long e = ((long long)(c - g.Cq) * Ah + dp) * PL + p0 + p;
            Vp[b * nv + e] = second ? a1 : a0; vsel[b * nv + e] = second;
        }
    }
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

// y = gamma * out + x with out[r][a] = sum_j Vp[r*(A/2)+j] * att[a][j], out [R][A] viewed as [C, E1, E2, E3] (Attention.py:101-106).
// Round 5: a block owns QT rows r of each of 32 channels (r = c * (N / A) + q): thread (channel, q) keeps its Vp row in registers
// and produces the A outputs of its row, the tile goes through LDS and leaves as 64-byte channel runs of the destination tokens
// (the first form wrote -- and read x -- 2 bytes per lane into a different token row each).  sep_dout_kernel is the same tiling in
// the other direction: dy rows in, dO[r][0..A) out, and the partial sums of dgamma.
constexpr int kSepCB = 32;
__device__ __forceinline__ int sep_qt(int A) { return A <= 16 ? 8 : A <= 32 ? 4 : A <= 48 ? 4 : 2; }        // QT * A <= 256 tile rows ... (A <= 64)
template <typename T, bool BACKWARD>
__global__ __launch_bounds__(256) void sep_outio_kernel(SepGeo g, const float* Vp, const float* att, const T* xy, int ldx, const float* gamma,
                                                        T* y, float* dO, float* dgamma) {
    __shared__ float att_s[64 * 32];
    __shared__ float tile[256][kSepCB + 1];
    __shared__ float sh[4];
    const int A = g.A, Ah = A / 2, QT = sep_qt(A);
    const long long PLr = g.N / A, per = (long long)g.C * g.N;
    const long long b = blockIdx.z;
    const int q0 = blockIdx.x * QT, c0 = blockIdx.y * kSepCB, tid = threadIdx.x;
    const int cl = tid & (kSepCB - 1), ql = tid >> 5;
    for (int i = tid; i < A * Ah; i += 256) att_s[i] = att[b * A * Ah + i];
    const bool live = ql < QT && q0 + ql < PLr && c0 + cl < g.C;
    const long long r = (long long)(c0 + cl) * PLr + q0 + ql;
    float v[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) v[j] = (live && j < Ah) ? Vp[b * (per / 2) + r * Ah + j] : 0.f;
    const float gm = *gamma;
    if (BACKWARD) {                                   // dy rows of the tile's tokens -> LDS (channel runs)
        for (int idx = tid; idx < QT * A * kSepCB; idx += 256) {
            const int c = idx & (kSepCB - 1), row = idx >> 5, qq = row / A, a = row - qq * A;
            float d = 0.f;
            if (q0 + qq < PLr && c0 + c < g.C) {
                int cc; long long tok;
                out_dest(g, ((long long)c0 * PLr + q0 + qq) * A + a, cc, tok);          // (the token does not depend on the channel)
                d = ldf(xy + ((size_t)b * g.N + tok) * ldx + c0 + c);
            }
            tile[row][c] = d;
        }
    }
    __syncthreads();
    float part = 0.f;
    if (live || !BACKWARD)
        for (int a = 0; a < A; ++a) {
            float o = 0.f;
#pragma unroll
            for (int j = 0; j < 32; ++j) o += v[j] * att_s[a * Ah + (j < Ah ? j : 0)] * (j < Ah ? 1.f : 0.f);
            if (BACKWARD) {
                if (live) {
                    const float d = tile[ql * A + a][cl];
                    dO[b * per + r * A + a] = gm * d;
                    part += d * o;
                }
            } else if (ql < QT) tile[ql * A + a][cl] = o;
        }
    if (BACKWARD) {
        part = blk_sum256(part, sh);
        if (tid == 0 && part != 0.f) atomicAdd(dgamma, part);
        return;
    }
    __syncthreads();
    for (int idx = tid; idx < QT * A * kSepCB; idx += 256) {
        const int c = idx & (kSepCB - 1), row = idx >> 5, qq = row / A, a = row - qq * A;
        if (q0 + qq >= PLr || c0 + c >= g.C) continue;
        int cc; long long tok;
        out_dest(g, ((long long)c0 * PLr + q0 + qq) * A + a, cc, tok);
        const size_t off = ((size_t)b * g.N + tok) * ldx + c0 + c;
        stf(y + off, gm * tile[row][c] + ldf(xy + off));
    }
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

// gradients of the flat operands.  Round 5: one thread per ROW of the long axis (l of Qf / Kp, r of dO / Vp) with dS / att in LDS, so
// every global access is a per-thread contiguous run or coalesced across the block:
//   dQf[a*L + l] = sum_j dS[a][j] * Kp[l*(A/2) + j];   dKp[l*(A/2) + j] = sum_a dS[a][j] * Qf[a*L + l]      (thread l)
//   dVp[r*(A/2) + j] = sum_a dO[r*A + a] * att[a][j]                                                        (thread r)
__global__ __launch_bounds__(256) void sep_dqk_kernel(SepGeo g, const float* dS, const float* Qf, const float* Kp, float* dQf, float* dKp) {
    __shared__ float ds_s[64 * 32];
    const int A = g.A, Ah = A / 2;
    const long long b = blockIdx.y, L = (long long)g.Cq * g.N / A;
    const long long nq = (long long)g.Cq * g.N, nk = nq / 2;
    for (int i = threadIdx.x; i < A * Ah; i += 256) ds_s[i] = dS[b * A * Ah + i];
    __syncthreads();
    const long long l = (long long)blockIdx.x * 256 + threadIdx.x;
    if (l >= L) return;
    float k[32], dk[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) { k[j] = j < Ah ? Kp[b * nk + l * Ah + j] : 0.f; dk[j] = 0.f; }
    for (int a = 0; a < A; ++a) {
        const float q = Qf[b * nq + (long long)a * L + l];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const float d = j < Ah ? ds_s[a * Ah + j] : 0.f;
            s += d * k[j];
            dk[j] += d * q;
        }
        dQf[b * nq + (long long)a * L + l] = s;
    }
#pragma unroll
    for (int j = 0; j < 32; ++j)
        if (j < Ah) dKp[b * nk + l * Ah + j] = dk[j];
}
__global__ __launch_bounds__(256) void sep_dv_kernel(SepGeo g, const float* dO, const float* att, float* dVp) {
    __shared__ float att_s[64 * 32];
    const int A = g.A, Ah = A / 2;
    const long long b = blockIdx.y, R = (long long)g.C * g.N / A, nv = (long long)g.C * g.N / 2;
    for (int i = threadIdx.x; i < A * Ah; i += 256) att_s[i] = att[b * A * Ah + i];
    __syncthreads();
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    float dv[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) dv[j] = 0.f;
    for (int a = 0; a < A; ++a) {
        const float d = dO[b * 2 * nv + r * A + a];
#pragma unroll
        for (int j = 0; j < 32; ++j) dv[j] += d * (j < Ah ? att_s[a * Ah + j] : 0.f);
    }
#pragma unroll
    for (int j = 0; j < 32; ++j)
        if (j < Ah) dVp[b * nv + r * Ah + j] = dv[j];
}

// dqkv (channels-last, q | k | v columns; other columns untouched) from the flat gradients: the tiling of sep_gather_kernel in the
// other direction (the max-pool gradient goes to the remembered winner of each pair)
template <typename T, int TP>
__global__ __launch_bounds__(256) void sep_scatter_kernel(SepGeo g, const float* dQf, const float* dKp, const float* dVp, const unsigned char* ksel,
                                                          const unsigned char* vsel, T* dqkv, int ldq, int koff, int voff) {
    extern __shared__ float tile[];                 // [2][TP][NC + 1]
    const int NC = 2 * g.Cq + g.C, NCP = NC + 1;
    const int PL = g.D1 * g.D2, Ah = g.A / 2;
    const int p0 = blockIdx.x * TP, dp = blockIdx.y, tid = threadIdx.x;
    const long long b = blockIdx.z;
    const long long nq = (long long)g.Cq * g.N, nk = nq / 2, nv = (long long)g.C * g.N / 2;
    for (int idx = tid; idx < 2 * g.Cq * TP; idx += 256) {
        const int p = idx % TP, r = idx / TP, c = r % g.Cq, s2 = r / g.Cq;
        tile[(s2 * TP + p) * NCP + c] = p0 + p < PL ? dQf[b * nq + ((long long)c * g.A + 2 * dp + s2) * PL + p0 + p] : 0.f;
    }
    for (int idx = tid; idx < (g.Cq + g.C) * TP; idx += 256) {
        const int p = idx % TP, c = idx / TP;
        float gr = 0.f; int sel = 0;
        if (p0 + p < PL) {
            if (c < g.Cq) { const long long e = ((long long)c * Ah + dp) * PL + p0 + p; gr = dKp[b * nk + e]; sel = ksel[b * nk + e]; }
            else { const long long e = ((long long)(c - g.Cq) * Ah + dp) * PL + p0 + p; gr = dVp[b * nv + e]; sel = vsel[b * nv + e]; }
        }
        tile[p * NCP + g.Cq + c] = sel == 0 ? gr : 0.f;
        tile[(TP + p) * NCP + g.Cq + c] = sel == 1 ? gr : 0.f;
    }
    __syncthreads();
    T* base = dqkv + (size_t)b * g.N * ldq;
    for (int idx = tid; idx < 2 * TP * NC; idx += 256) {
        const int col = idx % NC, rp = idx / NC, p = rp % TP, s2 = rp / TP;
        if (p0 + p >= PL) continue;
        const int pos = p0 + p, d1 = pos / g.D2, d2 = pos - d1 * g.D2;
        const int dst = col < g.Cq ? col : col < 2 * g.Cq ? koff + col - g.Cq : voff + col - 2 * g.Cq;
        stf(base + (size_t)token(g, 2 * dp + s2, d1, d2) * ldq + dst, tile[(s2 * TP + p) * NCP + col]);
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
    const int PL = g.D1 * g.D2, NC = 2 * Cq + C;
    if ((size_t)2 * 8 * (NC + 1) * sizeof(float) > 64 * 1024) return DVD_E_SHAPE;
    // positions per tile: as many as 64 KB of LDS hold (two attended coordinates x TP rows of NC + 1 floats)
#define SEP_TILED(KERNEL, ...)                                                                                                         \
    do {                                                                                                                               \
        if ((size_t)2 * 32 * (NC + 1) * sizeof(float) <= 64 * 1024)                                                                    \
            BY_DTYPE(dtype, KERNEL<T, 32><<<dim3(cdiv(PL, 32), g.A / 2, (unsigned)B), 256, (size_t)2 * 32 * (NC + 1) * sizeof(float), S_>>>(__VA_ARGS__));   \
        else if ((size_t)2 * 16 * (NC + 1) * sizeof(float) <= 64 * 1024)                                                               \
            BY_DTYPE(dtype, KERNEL<T, 16><<<dim3(cdiv(PL, 16), g.A / 2, (unsigned)B), 256, (size_t)2 * 16 * (NC + 1) * sizeof(float), S_>>>(__VA_ARGS__));   \
        else                                                                                                                           \
            BY_DTYPE(dtype, KERNEL<T, 8><<<dim3(cdiv(PL, 8), g.A / 2, (unsigned)B), 256, (size_t)2 * 8 * (NC + 1) * sizeof(float), S_>>>(__VA_ARGS__));      \
    } while (0)
    SEP_TILED(sep_gather_kernel, g, (const T*)qkv, ldq, koff, voff, Qf, Kp, Vp, ksel, vsel);
    // the partial score sums live in `y` until sep_out_kernel writes it (B * 8 * A * A/2 floats; y holds B * N * ldx elements)
    float* part = reinterpret_cast<float*>(y);
    if ((long long)kSepSplit * g.A * (g.A / 2) * 4 > g.N * ldx * (dtype == DVD_BF16 ? 2 : 4)) return DVD_E_SHAPE;       // (never: N >= 8 A)
    sep_prod_kernel<true><<<dim3(kSepSplit, (unsigned)B), 1024, 0, S_>>>(Qf, Kp, part, g.A, g.A / 2, (long long)Cq * g.D1 * g.D2);
    sep_softmax_kernel<<<dim3(g.A, (unsigned)B), 64, 0, S_>>>(part, att, g.A, g.A / 2);
    if (ldx != C &&      // padded channel columns of y must read zero afterwards (sep_out_kernel writes the real ones only)
        hipMemsetAsync(y, 0, (size_t)B * kSepSplit * g.A * (g.A / 2) * sizeof(float), S_) != hipSuccess) return DVD_E_LAUNCH;
    {
        const int QT = g.A <= 16 ? 8 : g.A <= 48 ? 4 : 2;
        const dim3 grid(cdiv(g.N / g.A, QT), cdiv(C, 32), (unsigned)B);
        BY_DTYPE(dtype, sep_outio_kernel<T, false><<<grid, 256, 0, S_>>>(g, Vp, att, (const T*)x, ldx, gamma, (T*)y, nullptr, nullptr));
    }
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
    {
        const int QT = g.A <= 16 ? 8 : g.A <= 48 ? 4 : 2;
        const dim3 grid(cdiv(g.N / g.A, QT), cdiv(C, 32), (unsigned)B);
        BY_DTYPE(dtype, sep_outio_kernel<T, true><<<grid, 256, 0, S_>>>(g, Vp, att, (const T*)dy, ldx, gamma, (T*)nullptr, dO, dgamma));
    }
    // the partial sums of datt live in dQf until sep_dops_kernel writes it (B * 8 * A * A/2 of its B * Cq * N floats)
    if ((long long)kSepSplit * g.A * (g.A / 2) > (long long)Cq * g.N) return DVD_E_SHAPE;
    sep_prod_kernel<false><<<dim3(kSepSplit, (unsigned)B), 1024, 0, S_>>>(dO, Vp, dQf, g.A, g.A / 2, (long long)C * g.D1 * g.D2);
    sep_dsoft_kernel<<<dim3(g.A, (unsigned)B), 64, 0, S_>>>(dQf, att, dS, g.A, g.A / 2);
    sep_dqk_kernel<<<dim3(cdiv((long long)Cq * g.N / g.A, 256), (unsigned)B), 256, 0, S_>>>(g, dS, Qf, Kp, dQf, dKp);
    sep_dv_kernel<<<dim3(cdiv((long long)C * g.N / g.A, 256), (unsigned)B), 256, 0, S_>>>(g, dO, att, dVp);
    const int PL = g.D1 * g.D2, NC = 2 * Cq + C;
    if ((size_t)2 * 8 * (NC + 1) * sizeof(float) > 64 * 1024) return DVD_E_SHAPE;
    SEP_TILED(sep_scatter_kernel, g, dQf, dKp, dVp, ksel, vsel, (T*)dqkv, ldq, koff, voff);
    return launch_status();
}
