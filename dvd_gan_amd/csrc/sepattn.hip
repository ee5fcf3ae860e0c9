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
// invoked by its generator; it is < 0.1 % of a step's FLOPs when enabled, so everything is plain fp32 on the vector pipe and the
// cost is memory passes: gather (qkv -> Qf | Kp | Vp), scores (tiled product) + softmax, output (Vp att^T -> y) forward;
// output-backward (dy -> dO, dVp, dgamma partials), datt (tiled product) + softmax backward, dQf / dKp, scatter backward.
// Round 5 took one cell at [64, 48, 32, 32, 128] from 20.9 to 7.4 ms (profiles/HISTORY.md): 8-channel-wide tiles without per-element
// divisions in gather / scatter, att read at wave-uniform addresses (scalar loads) instead of LDS, dVp fused into the output
// backward, register-prefetched products, dO written in runs, every sum in a fixed order (dgamma included).
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

// The same tile with 8-channel (16-byte for bf16) accesses to `qkv` and no per-element divisions: thread = (8-channel chunk, row of
// the tile) on the channels-last side, (position, channel) on the flat side; the token of each tile row is worked out once per
// block.  Needs Cq, C, koff, voff and ldq to be multiples of 8 (the generator's widths); other shapes take the scalar form above.
template <typename T, int TP>
__global__ __launch_bounds__(256) void sep_gather8_kernel(SepGeo g, const T* qkv, int ldq, int koff, int voff, float* Qf, float* Kp, float* Vp,
                                                          unsigned char* ksel, unsigned char* vsel) {
    extern __shared__ float tile[];                 // [2][TP][NC + 1]
    __shared__ long long tok_s[2 * TP];
    const int NC = 2 * g.Cq + g.C, NCP = NC + 1, NCH = NC / 8;
    const int PL = g.D1 * g.D2, Ah = g.A / 2;
    const int p0 = blockIdx.x * TP, dp = blockIdx.y, tid = threadIdx.x;
    const long long b = blockIdx.z;
    const long long nq = (long long)g.Cq * g.N, nk = nq / 2, nv = (long long)g.C * g.N / 2;
    const T* base = qkv + (size_t)b * g.N * ldq;
    if (tid < 2 * TP) {
        const int p = tid % TP, s2 = tid / TP, pos = p0 + p;
        tok_s[tid] = pos < PL ? token(g, 2 * dp + s2, pos / g.D2, pos % g.D2) : -1;
    }
    __syncthreads();
    {
        const int ch = tid % NCH, r0 = tid / NCH, rstep = 256 / NCH > 0 ? 256 / NCH : 1;
        const int col = ch * 8;
        const int src = col < g.Cq ? col : col < 2 * g.Cq ? koff + col - g.Cq : voff + col - 2 * g.Cq;
        if (tid < NCH * rstep)
            for (int row = r0; row < 2 * TP; row += rstep) {
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const long long tk = tok_s[row];
                if (tk >= 0) load8<T>(base + (size_t)tk * ldq + src, v);
#pragma unroll
                for (int k = 0; k < 8; ++k) tile[row * NCP + col + k] = v[k];
            }
    }
    __syncthreads();
    const int p = tid % TP, cc = tid / TP;
    constexpr int CSTEP = 256 / TP;
    if (p0 + p >= PL) return;
    for (int s2 = 0; s2 < 2; ++s2)                                   // q: both attended coordinates
        for (int c = cc; c < g.Cq; c += CSTEP)
            Qf[b * nq + ((long long)c * g.A + 2 * dp + s2) * PL + p0 + p] = tile[(s2 * TP + p) * NCP + c];
    for (int c = cc; c < g.Cq + g.C; c += CSTEP) {                   // k, v: max over the pair; ties -> the first element, like F.max_pool3d
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
    // the next 64 rows of both operands wait in registers while the current ones are multiplied (loads overlap the products)
    float ru[4], rv[2];
    auto fetch = [&](long long lc) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {                                // first operand, zero-padded to 64 columns
            const int idx = tid + t * 1024;
            int a, l;
            if (FIRST_ROWS) { a = idx / kSepRows; l = idx - a * kSepRows; } else { l = idx / 64; a = idx - l * 64; }
            ru[t] = (a < A && lc + l < l1) ? (FIRST_ROWS ? U[(long long)a * L + lc + l] : U[(lc + l) * A + a]) : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {                                // second operand, zero-padded to 32 columns
            const int idx = tid + t * 1024, l = idx >> 5, j = idx & 31;
            rv[t] = (j < Ah && lc + l < l1) ? V[(lc + l) * Ah + j] : 0.f;
        }
    };
    if (l0 < l1) fetch(l0);
    for (long long lc = l0; lc < l1; lc += kSepRows) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int idx = tid + t * 1024;
            int a, l;
            if (FIRST_ROWS) { a = idx / kSepRows; l = idx - a * kSepRows; } else { l = idx / 64; a = idx - l * 64; }
            Ut[l][a] = ru[t];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int idx = tid + t * 1024;
            Vt[idx >> 5][idx & 31] = rv[t];
        }
        __syncthreads();
        if (lc + kSepRows < l1) fetch(lc + kSepRows);
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

// four consecutive values of a row that EVERY lane reads at the same (wave-uniform) address: the compiler turns these into scalar
// loads (s_load_dwordx4 through the scalar cache), so the small attention matrix occupies neither LDS bandwidth nor vector registers.
// `n` = valid length of the row: a group that would cross it is read value by value (never past the row).
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ f32x4 row4(const float* __restrict__ row, int j, int n) {
    if (j + 4 <= n) { const f32x4u t = *reinterpret_cast<const f32x4u*>(row + j); return f32x4{t[0], t[1], t[2], t[3]}; }
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (j < n) t[0] = row[j];
    if (j + 1 < n) t[1] = row[j + 1];
    if (j + 2 < n) t[2] = row[j + 2];
    return t;
}

// y = gamma * out + x with out[r][a] = sum_j Vp[r*(A/2)+j] * att[a][j], out [R][A] viewed as [C, E1, E2, E3] (Attention.py:101-106).
// A block owns QT rows r of each of 64 channels (r = c * (N / A) + q): thread (channel, q) keeps its Vp row in registers and produces
// the A outputs of its row; the tile goes through LDS and leaves as 128-byte channel runs of the destination tokens.  For a fixed q
// the A outputs lie on tokens base(q) + a * stride (sep_tok: the view + permute of the reference reduces to that), so the tile
// loops carry no divisions.  att is read at wave-uniform addresses (row4: scalar loads), not staged in LDS.
// BACKWARD is the same tiling in the other direction: dy rows in; dO[r][0..A) = gamma * dy out (operand of the datt product),
// dVp[r][:] = sum_a dO[r][a] * att[a][:] from the same att reads, and the block's partial sum of dgamma.
constexpr int kSepCB = 64;
__host__ __device__ inline int sep_qt(int A) { return A <= 48 ? 4 : 3; }        // QT * A <= 192 tile rows (A <= 64)
// token of output (q, a = 0): out flat index o = q * A + a of one channel, viewed / permuted as Attention.py:101-106
__device__ __forceinline__ long long sep_tok(const SepGeo& g, long long q, long long& stride) {
    if (g.axis == 0) { stride = (long long)g.W * g.H; return q; }                               // (W, H, T) -> token (t, w, h)
    if (g.axis == 1) { stride = g.H; return (q / g.H) * g.W * g.H + q % g.H; }                    // (T, H, W)
    stride = 1; return q * g.H;                                                                   // (T, W, H)
}
// NJ: A/2 = 4 * NJ known at compile time (straight-line inner products; 0 = any A/2 <= 32, bounds checked at run time)
// dO leaves in RUNS: the QT rows of one channel are QT * A consecutive floats = one column of the dy tile, written with the lanes
// along the run.  (Staging Vp / dVp through the tile the same way cost more in barriers than the per-lane 16-byte accesses.)
template <typename T, bool BACKWARD, int NJ>
__global__ __launch_bounds__(256) void sep_outio_kernel(SepGeo g, const float* __restrict__ Vp, const float* __restrict__ att,
                                                        const T* __restrict__ xy, int ldx, const float* __restrict__ gamma,
                                                        T* __restrict__ y, float* __restrict__ dO, float* __restrict__ dVp, float* __restrict__ gpart) {
    __shared__ float tile[192][kSepCB + 1];
    __shared__ float sh[4];
    __shared__ long long tok_s[4], str_s;
    constexpr int JN = NJ ? 4 * NJ : 32;
    const int A = g.A, Ah = NJ ? JN : A / 2, QT = sep_qt(A);
    const long long PLr = g.N / A, per = (long long)g.C * g.N;
    const long long b = blockIdx.z;
    const int q0 = blockIdx.x * QT, c0 = blockIdx.y * kSepCB, tid = threadIdx.x;
    const int cl = tid & (kSepCB - 1), ql = tid >> 6;
    const float* __restrict__ attb = att + b * A * Ah;
    if (tid < QT) { long long st; tok_s[tid] = sep_tok(g, q0 + tid, st); if (tid == 0) str_s = st; }
    const int cw = min(kSepCB, g.C - c0), nq = (int)min((long long)QT, PLr - q0);
    const bool live = ql < nq && cl < cw;
    const float gm = *gamma;
    // runs of channel c0 + c: rows (c0 + c) * PLr + q0 ... + nq of a flat [R][w] operand = nq * w consecutive floats <-> tile[i][c]
    auto run_out = [&](float* __restrict__ dst, int w, float scale) __attribute__((always_inline)) {
        const int n = nq * w;
        for (int c = ql; c < cw; c += 4) {
            float* dp = dst + ((long long)(c0 + c) * PLr + q0) * w;
            for (int i = cl; i < n; i += 64) dp[i] = scale * tile[i][c];
        }
    };
    const long long r = (long long)(c0 + cl) * PLr + q0 + ql;
    float v[JN];
#pragma unroll
    for (int j = 0; j < JN; ++j) v[j] = 0.f;
    if (live) {                                       // the row's A/2 values: per-lane 16-byte loads (staging them through the tile was slower)
        const float* vp = Vp + b * (per / 2) + r * Ah;
        if (NJ || (Ah & 3) == 0) {
#pragma unroll
            for (int j = 0; j < JN; j += 4)
                if (NJ || j < Ah) { const f32x4 t = *reinterpret_cast<const f32x4*>(vp + j); v[j] = t[0]; v[j + 1] = t[1]; v[j + 2] = t[2]; v[j + 3] = t[3]; }
        } else {
#pragma unroll
            for (int j = 0; j < JN; ++j) if (j < Ah) v[j] = vp[j];
        }
    }
    __syncthreads();
    const long long tstr = str_s;
    if (BACKWARD) {                                   // dy rows of the tile's tokens -> LDS (channel runs)
        for (int qq = 0; qq < nq; ++qq) {
            const T* src = xy + ((size_t)b * g.N + tok_s[qq]) * ldx + c0 + cl;
            for (int a = ql; a < A; a += 4) tile[qq * A + a][cl] = cl < cw ? ldf(src + (size_t)a * tstr * ldx) : 0.f;
        }
        __syncthreads();
        run_out(dO + b * per, A, gm);                 // dO = gamma * dy, the operand of the datt product
        float part = 0.f;
        float dv[JN];
#pragma unroll
        for (int j = 0; j < JN; ++j) dv[j] = 0.f;
        if (live)
            for (int a = 0; a < A; ++a) {
                const float d = tile[ql * A + a][cl], gd = gm * d;
                float o = 0.f;
#pragma unroll
                for (int j = 0; j < JN; j += 4)
                    if (NJ || j < Ah) {
                        const f32x4 t = row4(attb + a * Ah, j, NJ ? j + 4 : Ah);
#pragma unroll
                        for (int k = 0; k < 4; ++k) { o += v[j + k] * t[k]; dv[j + k] += gd * t[k]; }
                    }
                part += d * o;
            }
        if (live) {
            float* dvr = dVp + b * (per / 2) + r * Ah;
            if (NJ || (Ah & 3) == 0) {
#pragma unroll
                for (int j = 0; j < JN; j += 4)
                    if (NJ || j < Ah) *reinterpret_cast<f32x4*>(dvr + j) = f32x4{dv[j], dv[j + 1], dv[j + 2], dv[j + 3]};
            } else {
#pragma unroll
                for (int j = 0; j < JN; ++j) if (j < Ah) dvr[j] = dv[j];
            }
        }
        part = blk_sum256(part, sh);
        if (tid == 0) gpart[((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = part;
        return;
    }
    if (ql < QT)
        for (int a = 0; a < A; ++a) {
            float o = 0.f;
#pragma unroll
            for (int j = 0; j < JN; j += 4)
                if (NJ || j < Ah) {
                    const f32x4 t = row4(attb + a * Ah, j, NJ ? j + 4 : Ah);
#pragma unroll
                    for (int k = 0; k < 4; ++k) o += v[j + k] * t[k];
                }
            tile[ql * A + a][cl] = o;
        }
    __syncthreads();
    if (cl < cw)
        for (int qq = 0; qq < nq; ++qq) {
            const size_t base = ((size_t)b * g.N + tok_s[qq]) * ldx + c0 + cl;
            for (int a = ql; a < A; a += 4) {
                const size_t off = base + (size_t)a * tstr * ldx;
                stf(y + off, gm * tile[qq * A + a][cl] + ldf(xy + off));
            }
        }
}

// dgamma += the block partials of sep_outio_kernel<BACKWARD>, in a fixed order (one block)
__global__ __launch_bounds__(1024) void sep_dgamma_kernel(const float* gpart, long long n, float* dgamma) {
    __shared__ float red[1024];
    float a = 0.f;
    for (long long i = threadIdx.x; i < n; i += 1024) a += gpart[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *dgamma += red[0];
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
//   (dVp[r*(A/2) + j] = sum_a dO[r*A + a] * att[a][j] comes out of sep_outio_kernel<BACKWARD>, which holds both operands.)
__global__ __launch_bounds__(256) void sep_dqk_kernel(SepGeo g, const float* __restrict__ dS, const float* __restrict__ Qf,
                                                      const float* __restrict__ Kp, float* __restrict__ dQf, float* __restrict__ dKp) {
    const int A = g.A, Ah = A / 2;
    const long long b = blockIdx.y, L = (long long)g.Cq * g.N / A;
    const long long nq = (long long)g.Cq * g.N, nk = nq / 2;
    const float* __restrict__ dsb = dS + b * A * Ah;                  // read at wave-uniform addresses (row4)
    const long long l = (long long)blockIdx.x * 256 + threadIdx.x;
    if (l >= L) return;
    float k[32], dk[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) { k[j] = 0.f; dk[j] = 0.f; }
    const float* kr = Kp + b * nk + l * Ah;
    if ((Ah & 3) == 0) {
#pragma unroll
        for (int j = 0; j < 32; j += 4)
            if (j < Ah) { const f32x4 t = *reinterpret_cast<const f32x4*>(kr + j); k[j] = t[0]; k[j + 1] = t[1]; k[j + 2] = t[2]; k[j + 3] = t[3]; }
    } else {
#pragma unroll
        for (int j = 0; j < 32; ++j) if (j < Ah) k[j] = kr[j];
    }
    for (int a = 0; a < A; ++a) {
        const float q = Qf[b * nq + (long long)a * L + l];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 32; j += 4)
            if (j < Ah) {
                const f32x4 d = row4(dsb + a * Ah, j, Ah);
#pragma unroll
                for (int i = 0; i < 4; ++i) { s += d[i] * k[j + i]; dk[j + i] += d[i] * q; }
            }
        dQf[b * nq + (long long)a * L + l] = s;
    }
    float* dkr = dKp + b * nk + l * Ah;
    if ((Ah & 3) == 0) {
#pragma unroll
        for (int j = 0; j < 32; j += 4)
            if (j < Ah) *reinterpret_cast<f32x4*>(dkr + j) = f32x4{dk[j], dk[j + 1], dk[j + 2], dk[j + 3]};
    } else {
#pragma unroll
        for (int j = 0; j < 32; ++j) if (j < Ah) dkr[j] = dk[j];
    }
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

// sep_scatter_kernel with the accesses of sep_gather8_kernel (same conditions)
template <typename T, int TP>
__global__ __launch_bounds__(256) void sep_scatter8_kernel(SepGeo g, const float* dQf, const float* dKp, const float* dVp, const unsigned char* ksel,
                                                           const unsigned char* vsel, T* dqkv, int ldq, int koff, int voff) {
    extern __shared__ float tile[];                 // [2][TP][NC + 1]
    __shared__ long long tok_s[2 * TP];
    const int NC = 2 * g.Cq + g.C, NCP = NC + 1, NCH = NC / 8;
    const int PL = g.D1 * g.D2, Ah = g.A / 2;
    const int p0 = blockIdx.x * TP, dp = blockIdx.y, tid = threadIdx.x;
    const long long b = blockIdx.z;
    const long long nq = (long long)g.Cq * g.N, nk = nq / 2, nv = (long long)g.C * g.N / 2;
    if (tid < 2 * TP) {
        const int p = tid % TP, s2 = tid / TP, pos = p0 + p;
        tok_s[tid] = pos < PL ? token(g, 2 * dp + s2, pos / g.D2, pos % g.D2) : -1;
    }
    {
        const int p = tid % TP, cc = tid / TP;
        constexpr int CSTEP = 256 / TP;
        const bool in = p0 + p < PL;
        for (int s2 = 0; s2 < 2; ++s2)
            for (int c = cc; c < g.Cq; c += CSTEP)
                tile[(s2 * TP + p) * NCP + c] = in ? dQf[b * nq + ((long long)c * g.A + 2 * dp + s2) * PL + p0 + p] : 0.f;
        for (int c = cc; c < g.Cq + g.C; c += CSTEP) {
            float gr = 0.f; int sel = 0;
            if (in) {
                if (c < g.Cq) { const long long e = ((long long)c * Ah + dp) * PL + p0 + p; gr = dKp[b * nk + e]; sel = ksel[b * nk + e]; }
                else { const long long e = ((long long)(c - g.Cq) * Ah + dp) * PL + p0 + p; gr = dVp[b * nv + e]; sel = vsel[b * nv + e]; }
            }
            tile[p * NCP + g.Cq + c] = sel == 0 ? gr : 0.f;
            tile[(TP + p) * NCP + g.Cq + c] = sel == 1 ? gr : 0.f;
        }
    }
    __syncthreads();
    T* base = dqkv + (size_t)b * g.N * ldq;
    const int ch = tid % NCH, r0 = tid / NCH, rstep = 256 / NCH > 0 ? 256 / NCH : 1;
    const int col = ch * 8;
    const int dst = col < g.Cq ? col : col < 2 * g.Cq ? koff + col - g.Cq : voff + col - 2 * g.Cq;
    if (tid < NCH * rstep)
        for (int row = r0; row < 2 * TP; row += rstep) {
            const long long tk = tok_s[row];
            if (tk < 0) continue;
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = tile[row * NCP + col + k];
            store8<T>(base + (size_t)tk * ldq + dst, v);
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
    // (every validation before the first launch) the partial score sums live in `y` until sep_out_kernel writes it
    if ((long long)kSepSplit * g.A * (g.A / 2) * 4 > g.N * ldx * (dtype == DVD_BF16 ? 2 : 4)) return DVD_E_SHAPE;       // (never: N >= 8 A)
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
    const bool wide = !((Cq | C | koff | voff | ldq) & 7) && NC / 8 <= 256;
    if (wide) SEP_TILED(sep_gather8_kernel, g, (const T*)qkv, ldq, koff, voff, Qf, Kp, Vp, ksel, vsel);
    else SEP_TILED(sep_gather_kernel, g, (const T*)qkv, ldq, koff, voff, Qf, Kp, Vp, ksel, vsel);
    // the partial score sums live in `y` until sep_out_kernel writes it (B * 8 * A * A/2 floats; y holds B * N * ldx elements)
    float* part = reinterpret_cast<float*>(y);
    sep_prod_kernel<true><<<dim3(kSepSplit, (unsigned)B), 1024, 0, S_>>>(Qf, Kp, part, g.A, g.A / 2, (long long)Cq * g.D1 * g.D2);
    sep_softmax_kernel<<<dim3(g.A, (unsigned)B), 64, 0, S_>>>(part, att, g.A, g.A / 2);
    if (ldx != C &&      // padded channel columns of y must read zero afterwards (sep_out_kernel writes the real ones only)
        hipMemsetAsync(y, 0, (size_t)B * kSepSplit * g.A * (g.A / 2) * sizeof(float), S_) != hipSuccess) return DVD_E_LAUNCH;
    {
        const dim3 grid(cdiv(g.N / g.A, sep_qt(g.A)), cdiv(C, kSepCB), (unsigned)B);
#define SEP_OUTIO(BWD, ...)                                                                                         \
    do {                                                                                                            \
        if (g.A == 48) BY_DTYPE(dtype, sep_outio_kernel<T, BWD, 6><<<grid, 256, 0, S_>>>(__VA_ARGS__));             \
        else if (g.A == 32) BY_DTYPE(dtype, sep_outio_kernel<T, BWD, 4><<<grid, 256, 0, S_>>>(__VA_ARGS__));        \
        else BY_DTYPE(dtype, sep_outio_kernel<T, BWD, 0><<<grid, 256, 0, S_>>>(__VA_ARGS__));                       \
    } while (0)
        SEP_OUTIO(false, g, Vp, att, (const T*)x, ldx, gamma, (T*)y, nullptr, nullptr, nullptr);
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
        // the block partials of dgamma wait in dKp (B * blocks-per-clip of its B * Cq * N / 2 floats) until sep_dqk_kernel writes it
        const dim3 grid(cdiv(g.N / g.A, sep_qt(g.A)), cdiv(C, kSepCB), (unsigned)B);
        const long long nblk = (long long)grid.x * grid.y * B;
        if (nblk > (long long)B * Cq * g.N / 2) return DVD_E_SHAPE;
        SEP_OUTIO(true, g, Vp, att, (const T*)dy, ldx, gamma, (T*)nullptr, dO, dVp, dKp);
        sep_dgamma_kernel<<<1, 1024, 0, S_>>>(dKp, nblk, dgamma);
    }
    // the partial sums of datt live in dQf until sep_dops_kernel writes it (B * 8 * A * A/2 of its B * Cq * N floats)
    if ((long long)kSepSplit * g.A * (g.A / 2) > (long long)Cq * g.N) return DVD_E_SHAPE;
    sep_prod_kernel<false><<<dim3(kSepSplit, (unsigned)B), 1024, 0, S_>>>(dO, Vp, dQf, g.A, g.A / 2, (long long)C * g.D1 * g.D2);
    sep_dsoft_kernel<<<dim3(g.A, (unsigned)B), 64, 0, S_>>>(dQf, att, dS, g.A, g.A / 2);
    sep_dqk_kernel<<<dim3(cdiv((long long)Cq * g.N / g.A, 256), (unsigned)B), 256, 0, S_>>>(g, dS, Qf, Kp, dQf, dKp);
    const int PL = g.D1 * g.D2, NC = 2 * Cq + C;
    if ((size_t)2 * 8 * (NC + 1) * sizeof(float) > 64 * 1024) return DVD_E_SHAPE;
    const bool wide = !((Cq | C | koff | voff | ldq) & 7) && NC / 8 <= 256;
    if (wide) SEP_TILED(sep_scatter8_kernel, g, dQf, dKp, dVp, ksel, vsel, (T*)dqkv, ldq, koff, voff);
    else SEP_TILED(sep_scatter_kernel, g, dQf, dKp, dVp, ksel, vsel, (T*)dqkv, ldq, koff, voff);
    return launch_status();
}
