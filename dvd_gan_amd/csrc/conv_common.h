// Shared device code of the convolution kernels (conv_igemm.hip, conv_gb.hip, conv_wgrad.hip): the kernel parameter block,
// LDS tile addressing, and the batched epilogue (direct / split-K slabs / fused ConvGRU gate math / in-launch split-K combine).
#pragma once
#include "common.h"

namespace dvdk {

constexpr int BM = 128, BN = 128, NT = 256;
constexpr int TILEB = BM * 80;            // f32 wgrad: 10240 B per operand tile ([16 rows][WG_LD floats] fits)
constexpr int WG_LD = 132;               // f32 wgrad LDS row length in floats (128 + 4 pad)

__device__ __forceinline__ u32x4 relu16_f32(u32x4 v) {
    v.x = (int32_t)v.x < 0 ? 0u : v.x; v.y = (int32_t)v.y < 0 ? 0u : v.y;
    v.z = (int32_t)v.z < 0 ? 0u : v.z; v.w = (int32_t)v.w < 0 ? 0u : v.w;
    return v;
}
__device__ __forceinline__ uint32_t relu2_bf16(uint32_t v) {
    uint32_t m = ((v >> 15) & 0x00010001u) * 0xffffu;      // 0xffff in each half whose sign bit is set
    return v & ~m;
}
__device__ __forceinline__ u32x4 relu16_bf16(u32x4 v) {
    v.x = relu2_bf16(v.x); v.y = relu2_bf16(v.y); v.z = relu2_bf16(v.z); v.w = relu2_bf16(v.w);
    return v;
}
template <typename T> __device__ __forceinline__ u32x4 relu16(u32x4 v);
template <> __device__ __forceinline__ u32x4 relu16<float>(u32x4 v) { return relu16_f32(v); }
template <> __device__ __forceinline__ u32x4 relu16<bf16_t>(u32x4 v) { return relu16_bf16(v); }

// row index of a [frames][H][W] grid -> (frame, y, x).  Power-of-two extents (every size the 64 x 64 / 128 x 128 models
// produce) take shifts; any other extent (latent_dim 3, 6, ...: logW < 0) takes divisions -- those sizes run through the
// tap-by-tap kernels only, where the decomposition is outside the K loop (forward) or the launch is small.
__device__ __forceinline__ void grid_pos(int m, int H, int W, int logH, int logW, int& f, int& y, int& x) {
    if (logW >= 0) { x = m & (W - 1); y = (m >> logW) & (H - 1); f = m >> (logW + logH); }
    else { f = m / (H * W); const int r = m - f * (H * W); y = r / W; x = r - y * W; }
}

// ============================================================================ forward
struct ConvK {
    const char* in; const char* w; const float* bias; const char* res; const char* mask;
    const char* wq;                      // optional: the same weights in fragment-major order (conv_halo_gb_tile), else nullptr
    char* out; float* ws;
    int M, C, ldi, Cout, ldo, ldres, ldmask, res_up2;
    int T, H, W, logH, logW, Hin, Win;
    int kt, kh, kw, kchunks, nk, nsplit, tilesN;
    int up2, relu_in, act, out_f32;
    int nb32;                            // 32-column blocks per (tap, chunk) of the fragment-major image `wq`
    int nmajor;                          // tile order: consecutive workgroups (one XCD's run) share the N tile, not the M tile
    int pm;                              // > 0 (tap-by-tap kernel, small frames): GEMM rows in PIXEL-major order, pm = frames (see conv_igemm_kernel)
    int pool2;                           // halo kernels: `out` / `mask` on the half-size grid, 2 x 2 sums of the result (dvd_conv_desc.pool2)
    size_t in_bytes; unsigned w_bytes;   // extents for the buffer descriptors (hardware zero-fill past them)
    int maxshift;                        // largest |tap shift| in rows
    GruEpi g;                            // optional fused ConvGRU gate epilogue (mode 0 = off)
};

// 16-byte-per-lane LDS-DMA: LDS[lds + lane*16 .. +16) <- buffer[off]; zeros when off is out of range.
// `lds` must be wave-uniform.  (Kept in a non-template helper: inside a kernel template hipcc's HOST
// pass rejects the target builtin as a silent substitution failure and drops the kernel stub.)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds, unsigned off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
}

// LDS tile image: 64-byte rows (one K chunk), no padding; the 16-byte slot of a row is XOR-swizzled
// with bits 2..3 of the row index.  Conflict-free for the loader's ds_write_b128 (8 consecutive lanes
// = 2 rows x 4 slots = all 32 banks) and for the fragment ds_read_b128 (a 16-lane group reads 16 rows
// whose (row&3, slot) pairs are all distinct).  Measured before this layout (80-byte padded rows):
// SQ_LDS_BANK_CONFLICT = 33 % of SQ_LDS_IDX_ACTIVE, all of it on the stores.
__device__ __forceinline__ int lds_off(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

// ---------------------------------------------------------------------------- batched epilogue
// The epilogue used to be a chain of dependent round trips: per 32-row sub-tile four rounds of (load the gate / residual
// operands of 8 rows -> wait -> math -> store), each behind lane-divergent branches at whose joins hipcc waits vmcnt(0),
// with the bias read element by element in every round: 16 serialized L2 / HBM latencies per wave, ~14 us of a launch
// that hides nothing behind them (all workgroups of a one-round launch reach their epilogue together).  Here every
// operand goes through a buffer descriptor, so validity is an out-of-range offset instead of a branch (loads return
// zeros, stores are dropped; an absent optional operand is an EMPTY descriptor), the operands of all four rounds of a
// sub-tile are requested before the first is used, and the bias is read once per wave.
constexpr unsigned kOOB = 0x80000000u;            // offsets at / above 2 GiB are out of range for every descriptor built here

__device__ __forceinline__ __amdgpu_buffer_rsrc_t epi_rsrc(const void* base, long long row0, unsigned ld_bytes, long long rows) {
    const unsigned long long bytes = base ? (unsigned long long)rows * ld_bytes : 0ull;
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + (size_t)row0 * ld_bytes), 0,
                                             bytes > 0x7ffffff0ull ? 0x7ffffff0u : (unsigned)bytes, 0x00020000);
}
template <typename T> struct Raw8;                // 8 elements as they sit in memory
template <> struct Raw8<bf16_t> { u32x4 a; };
template <> struct Raw8<float> { u32x4 a, b; };
template <typename T> __device__ __forceinline__ Raw8<T> bld8(__amdgpu_buffer_rsrc_t r, unsigned off);
template <> __device__ __forceinline__ Raw8<bf16_t> bld8<bf16_t>(__amdgpu_buffer_rsrc_t r, unsigned off) {
    Raw8<bf16_t> v; v.a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); return v;
}
template <> __device__ __forceinline__ Raw8<float> bld8<float>(__amdgpu_buffer_rsrc_t r, unsigned off) {
    Raw8<float> v;
    v.a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    v.b = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, 0);
    return v;
}
__device__ __forceinline__ void unpack8(const Raw8<bf16_t>& r, float (&v)[8]) {
    v[0] = __uint_as_float(r.a.x << 16); v[1] = __uint_as_float(r.a.x & 0xffff0000u);
    v[2] = __uint_as_float(r.a.y << 16); v[3] = __uint_as_float(r.a.y & 0xffff0000u);
    v[4] = __uint_as_float(r.a.z << 16); v[5] = __uint_as_float(r.a.z & 0xffff0000u);
    v[6] = __uint_as_float(r.a.w << 16); v[7] = __uint_as_float(r.a.w & 0xffff0000u);
}
__device__ __forceinline__ void unpack8(const Raw8<float>& r, float (&v)[8]) {
    v[0] = __uint_as_float(r.a.x); v[1] = __uint_as_float(r.a.y); v[2] = __uint_as_float(r.a.z); v[3] = __uint_as_float(r.a.w);
    v[4] = __uint_as_float(r.b.x); v[5] = __uint_as_float(r.b.y); v[6] = __uint_as_float(r.b.z); v[7] = __uint_as_float(r.b.w);
}
template <typename T> __device__ __forceinline__ void bst8(__amdgpu_buffer_rsrc_t r, unsigned off, const float (&v)[8]);
template <> __device__ __forceinline__ void bst8<bf16_t>(__amdgpu_buffer_rsrc_t r, unsigned off, const float (&v)[8]) {
    u32x4 a;
    a.x = pack2_bf16(v[0], v[1]); a.y = pack2_bf16(v[2], v[3]); a.z = pack2_bf16(v[4], v[5]); a.w = pack2_bf16(v[6], v[7]);
    __builtin_amdgcn_raw_buffer_store_b128(a, r, off, 0, 0);
}
template <> __device__ __forceinline__ void bst8<float>(__amdgpu_buffer_rsrc_t r, unsigned off, const float (&v)[8]) {
    u32x4 a = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    u32x4 b = {__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])};
    __builtin_amdgcn_raw_buffer_store_b128(a, r, off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(b, r, off + 16, 0, 0);
}

// Epilogue of one wave: TM sub-tiles of 32 rows x 64 columns = R = 4*TM rounds of 8 rows x 64 columns (8 columns per lane).
// The rounds run as a software pipeline: the operands of round i + D are requested before round i is computed (D + 1
// register slots, static after unrolling; D per epilogue kind so that the 256 x 128 tile stays within 128 VGPRs beside its
// 128 accumulator registers, i.e. two workgroups per CU).  stage(tm) moves sub-tile tm's accumulators into the wave's
// LDS block before its first round.
template <int R, int D, class Stage, class Load, class Compute>
__device__ __forceinline__ void epi_run(Stage stage, Load load, Compute compute) {
#pragma unroll
    for (int i = 0; i < D && i < R; ++i) load(i);
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if ((i & 3) == 0) stage(i >> 2);
        if (i + D < R) load(i + D);
        compute(i);
        if ((i & 3) == 3) __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------- in-launch split-K combine (round 4)
// A split-K recurrent convolution used to leave ns fp32 slabs [z][M][Cout] for a gate kernel that summed them and applied the
// ConvGRU gate math: two launches per convolution on a chain of dependent launches.  Here the LAST workgroup of a tile to finish
// sums the slices and runs the fused gate epilogue itself -- no spinning, nobody waits for anybody:
//   every slice workgroup   writes its accumulators, in register order, to its slab with write-through (sc1) 16-byte stores
//                           -> every wave drains vmcnt -> barrier -> lane 0 draws a ticket (relaxed agent-scope fetch_add)
//   ticket != ns - 1        done
//   ticket == ns - 1        all other slabs are complete and in memory (their writers drained before drawing): read them with sc1
//                           loads (they bypass this CU's L1 and this XCD's L2 copy of a previous launch's data), add them in slice
//                           order, reset the ticket for the next launch on the stream, go on into the gate epilogue.
// The sum is the same whichever workgroup arrives last: ns == 2 adds the other slab onto the registers (a + b == b + a), ns > 2
// re-reads all ns slabs, its own included, into zeroed accumulators in slice order.  Slab layout (private to this function):
// [tile][slice][wave][tm][tn][quad][lane] x 16 bytes.  cdna_hip_programming.md section 5 (split-K reduction recipe).
template <int TM>
__device__ __forceinline__ bool splitk_combine(const ConvK& p, f32x16 (&acc)[TM][2], float* lds0, int lane, int z, int ltile) {
    constexpr unsigned kWaveBytes = TM * 2 * 4 * 1024;
    constexpr int kSc1 = 16;                          // cache-policy bit 4 on gfx950: sc1
    const int ns = p.nsplit;
    const int wave = threadIdx.x >> 6;
    const unsigned tileBytes = (blockDim.x >> 6) * kWaveBytes;
    const size_t tile = (size_t)ltile;          // index of this output tile among the convolution's tiles (== blockIdx.x outside grouped launches)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((char*)p.g.slabs + tile * ns * (size_t)tileBytes), 0, (unsigned)ns * tileBytes, 0x00020000);
    const unsigned lo = wave * kWaveBytes + lane * 16;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 v = {__float_as_uint(acc[tm][tn][4 * q]), __float_as_uint(acc[tm][tn][4 * q + 1]),
                                 __float_as_uint(acc[tm][tn][4 * q + 2]), __float_as_uint(acc[tm][tn][4 * q + 3])};
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, z * tileBytes + lo + ((tm * 2 + tn) * 4 + q) * 1024, 0, kSc1);
            }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        *reinterpret_cast<volatile unsigned*>(lds0) =
            __hip_atomic_fetch_add(p.g.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned ticket = *reinterpret_cast<volatile unsigned*>(lds0);
    if (ticket != (unsigned)(ns - 1)) return false;
    __syncthreads();                                   // the word is read before the staging area is written again
    if (threadIdx.x == 0) __hip_atomic_store(p.g.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int s_begin = 0, s_end = ns;
    if (ns == 2) { s_begin = z ^ 1; s_end = s_begin + 1; }
    else {
        const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = zacc;
    }
    // groups of 4 loads (one 32 x 32 accumulator block), the next group requested before this one is added: 8 loads per lane in
    // flight on 32 registers (the scheduling fences keep hipcc from hoisting every load of the unrolled body to the top -- 348
    // registers in the 256 x 128 kernels).  The group behind the last slab gets an out-of-range offset: zeros, no branch.
    u32x4 buf[2][4];
    auto issue = [&](int s, int g, u32x4 (&b)[4]) __attribute__((always_inline)) {
        const unsigned so = s < s_end ? (unsigned)s * tileBytes : kOOB;      // (the range check sees the vector offset only)
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, so + lo + (g * 4 + i) * 1024, 0, kSc1);
    };
    issue(s_begin, 0, buf[0]);
    for (int s = s_begin; s < s_end; ++s) {
#pragma unroll
        for (int g = 0; g < 2 * TM; ++g) {
            if (g + 1 < 2 * TM) issue(s, g + 1, buf[(g + 1) & 1]);
            else issue(s + 1, 0, buf[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4 v = buf[g & 1][i];
                f32x16& a = acc[g >> 1][g & 1];
                a[4 * i] += __uint_as_float(v.x); a[4 * i + 1] += __uint_as_float(v.y);
                a[4 * i + 2] += __uint_as_float(v.z); a[4 * i + 3] += __uint_as_float(v.w);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return true;
}

// `ep`: the wave's LDS staging block (32 x 64 floats); col: first of this lane's 8 output columns; row0: first row every
// row of this workgroup's tile is relative to (the buffer descriptors start there, offsets stay 32-bit); relrow(tm, j): this
// lane's row of round j of sub-tile tm, relative to row0, or a negative number when it lies past M.
// DEEP: deeper operand pipelines for the gate epilogues (kernels whose accumulators live in the unified register file can spend the
// registers of already-staged sub-tiles on operands in flight)
template <typename T, int TM, int DEEP = 0, bool COMBINE = true, class RelRow>
__device__ __forceinline__ void conv_epilogue(const ConvK& p, f32x16 (&acc)[TM][2], float* ep, int lane, int col, int z, int ltile,
                                              long long row0, RelRow relrow) {
    constexpr unsigned esz = sizeof(T);
    constexpr int R = 4 * TM;
    constexpr bool kB = sizeof(T) == 2;           // bf16: deeper pipelines fit
    const int erow = lane >> 3, ecol = (lane & 7) * 8;
    const bool colv = col < p.Cout;
    const long long rows = (long long)p.M - row0;
    auto stage = [&](int tm) __attribute__((always_inline)) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ep[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 64 + tn * 32 + (lane & 31)] = acc[tm][tn][r];
        __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): this wave's LDS writes landed
        __builtin_amdgcn_wave_barrier();
    };
    auto staged = [&](int i, float (&v)[8]) __attribute__((always_inline)) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(ep + ((i & 3) * 8 + erow) * 64 + ecol);
        const f32x4 b = *reinterpret_cast<const f32x4*>(ep + ((i & 3) * 8 + erow) * 64 + ecol + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    };
    auto rrow = [&](int i) __attribute__((always_inline)) -> int { return relrow(i >> 2, i & 3); };
    // byte offset of (row, column c) in a tensor with `ld` elements of `eb` bytes per row; kOOB when the row is past M or
    // `ok` is false
    auto offs = [&](int rr, unsigned ld, unsigned eb, int c, bool ok) __attribute__((always_inline)) -> unsigned {
        return (rr >= 0 && ok) ? (unsigned)rr * (ld * eb) + (unsigned)c * eb : kOOB;
    };
    const int mode = p.g.mode;
#ifdef DVD_EXP_NOEPI           // compile-time measurement variant (tools/build_variant.sh): the epilogue is skipped, results are garbage
    return;
#endif
    // split-K with a gate epilogue: only the last slice workgroup of the tile to arrive goes on, holding the full sums
    if constexpr (COMBINE)
        if (mode != 0 && p.nsplit > 1 && !splitk_combine<TM>(p, acc, ep - (threadIdx.x >> 6) * (32 * 64), lane, z, ltile)) return;
    if (mode == 1) {              // [u|r] = sigmoid(acc + gx);  hr = h_prev * r        (ConvGRU.py:47-49)
        constexpr int D = kB ? 3 : 1;
        const int h = p.g.h;
        const bool isr = col >= h;
        const int c2 = isr ? col - h : col;
        const auto rgx = epi_rsrc(p.g.gx, row0, p.g.ldg * esz, rows), rhp = epi_rsrc(p.g.hprev, row0, h * esz, rows);
        const auto ru = epi_rsrc(p.g.u, row0, h * esz, rows), rr_ = epi_rsrc(p.g.r, row0, h * esz, rows);
        const auto rhr = epi_rsrc(p.g.hr, row0, h * esz, rows);
        Raw8<T> gxv[D + 1], hpv[D + 1];
        epi_run<R, D>(stage,
            [&](int i) __attribute__((always_inline)) {
                const int rr = rrow(i);
                gxv[i % (D + 1)] = bld8<T>(rgx, offs(rr, p.g.ldg, esz, col, colv));
                hpv[i % (D + 1)] = bld8<T>(rhp, offs(rr, h, esz, c2, colv && isr));
            },
            [&](int i) __attribute__((always_inline)) {
                const int rr = rrow(i);
                float v[8], g[8], hp[8];
                staged(i, v); unpack8(gxv[i % (D + 1)], g); unpack8(hpv[i % (D + 1)], hp);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v[k] = gate_sigmoid<T>(v[k] + g[k]);
                    const float rk = round_to<T>(v[k]);          // the stored r is the r the cell uses
                    hp[k] *= rk;
                    v[k] = isr ? rk : v[k];
                }
                bst8<T>(ru, offs(rr, h, esz, c2, colv && !isr), v);
                bst8<T>(rr_, offs(rr, h, esz, c2, colv && isr), v);
                bst8<T>(rhr, offs(rr, h, esz, c2, colv && isr), hp);
            });
        return;
    }
    if (mode == 2) {                // o = tanh(acc + gx_o);  h = h_prev (1 - u) + o u          (ConvGRU.py:50-52)
        constexpr int D = (kB && DEEP) ? DEEP : 1;
        const int h = p.g.h;
        const bool c32 = p.g.h32p != nullptr;
        const auto rgx = epi_rsrc(p.g.gx, row0, p.g.ldg * esz, rows), ru = epi_rsrc(p.g.u_in, row0, h * esz, rows);
        const auto rhp = epi_rsrc(c32 ? nullptr : p.g.hprev, row0, h * esz, rows), rh32 = epi_rsrc(p.g.h32p, row0, h * 4, rows);
        const auto ro = epi_rsrc(p.g.o, row0, h * esz, rows), rhn = epi_rsrc(p.g.hn, row0, h * esz, rows);
        const auto rn32 = epi_rsrc(p.g.h32n, row0, h * 4, rows);
        Raw8<T> gxv[D + 1], uv[D + 1], hpv[D + 1];
        Raw8<float> h32v[D + 1];
        epi_run<R, D>(stage,
            [&](int i) __attribute__((always_inline)) {
                const int rr = rrow(i);
                gxv[i % (D + 1)] = bld8<T>(rgx, offs(rr, p.g.ldg, esz, 2 * h + col, colv));
                uv[i % (D + 1)] = bld8<T>(ru, offs(rr, h, esz, col, colv));
                hpv[i % (D + 1)] = bld8<T>(rhp, offs(rr, h, esz, col, colv));
                h32v[i % (D + 1)] = bld8<float>(rh32, offs(rr, h, 4, col, colv));
            },
            [&](int i) __attribute__((always_inline)) {
                const int rr = rrow(i);
                float v[8], g[8], uu[8], hp[8], hq[8];
                staged(i, v); unpack8(gxv[i % (D + 1)], g); unpack8(uv[i % (D + 1)], uu);
                unpack8(hpv[i % (D + 1)], hp); unpack8(h32v[i % (D + 1)], hq);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float hk = c32 ? hq[k] : hp[k];
                    v[k] = round_to<T>(gate_tanh<T>(v[k] + g[k]));
                    hp[k] = hk * (1.f - uu[k]) + v[k] * uu[k];
                }
                bst8<T>(ro, offs(rr, h, esz, col, colv), v);
                bst8<T>(rhn, offs(rr, h, esz, col, colv), hp);
                bst8<float>(rn32, offs(rr, h, 4, col, colv), hp);
            });
        return;
    }
    if (mode == 3) {                // BPTT: acc = d(h*r);  carry += acc*r;  d(pre_r) = acc*h_prev*r(1-r)   (gru.hip gru_bwd_r)
        constexpr int D = kB ? (DEEP ? 3 : 2) : 1;
        const int h = p.g.h;
        const auto rr_ = epi_rsrc(p.g.r, row0, h * esz, rows), rhp = epi_rsrc(p.g.hprev, row0, h * esz, rows);
        const auto rcy = epi_rsrc(p.g.h32n, row0, h * 4, rows), rdg = epi_rsrc(p.g.o, row0, p.g.ldg * esz, rows);
        Raw8<T> rv[D + 1], hpv[D + 1];
        Raw8<float> cyv[D + 1];
        epi_run<R, D>(stage,
            [&](int i) __attribute__((always_inline)) {
                const int rr = rrow(i);
                rv[i % (D + 1)] = bld8<T>(rr_, offs(rr, h, esz, col, colv));
                hpv[i % (D + 1)] = bld8<T>(rhp, offs(rr, h, esz, col, colv));
                cyv[i % (D + 1)] = bld8<float>(rcy, offs(rr, h, 4, col, colv));
            },
            [&](int i) __attribute__((always_inline)) {
                const int rr = rrow(i);
                float v[8], r8[8], hp[8], cy[8];
                staged(i, v); unpack8(rv[i % (D + 1)], r8); unpack8(hpv[i % (D + 1)], hp); unpack8(cyv[i % (D + 1)], cy);
#pragma unroll
                for (int k = 0; k < 8; ++k) { cy[k] += v[k] * r8[k]; v[k] = v[k] * hp[k] * r8[k] * (1.f - r8[k]); }
                bst8<float>(rcy, offs(rr, h, 4, col, colv), cy);
                bst8<T>(rdg, offs(rr, p.g.ldg, esz, h + col, colv), v);
            });
        return;
    }
    if (mode == 5) {                // BPTT: mode 4 followed by the first half of the NEXT step to process (gru_bwd_out):
        constexpr int D = (kB && DEEP) ? DEEP : 1;        // dh = carry + acc + dh_out;  d(pre_o), d(pre_u) of that step;  carry = dh (1 - u)
        const int h = p.g.h;
        const auto rcy = epi_rsrc(p.g.h32n, row0, h * 4, rows), rdh = epi_rsrc(p.g.gx, row0, h * esz, rows);
        const auto ru = epi_rsrc(p.g.u_in, row0, h * esz, rows), rog = epi_rsrc(p.g.hr, row0, h * esz, rows);
        const auto rhp = epi_rsrc(p.g.hprev, row0, h * esz, rows), rdg = epi_rsrc(p.g.o, row0, p.g.ldg * esz, rows);
        Raw8<float> cyv[D + 1];
        Raw8<T> dhv[D + 1], uv[D + 1], ov[D + 1], hpv[D + 1];
        epi_run<R, D>(stage,
            [&](int i) __attribute__((always_inline)) {
                const int rr = rrow(i);
                const unsigned o = offs(rr, h, esz, col, colv);
                cyv[i % (D + 1)] = bld8<float>(rcy, offs(rr, h, 4, col, colv));
                dhv[i % (D + 1)] = bld8<T>(rdh, o); uv[i % (D + 1)] = bld8<T>(ru, o);
                ov[i % (D + 1)] = bld8<T>(rog, o); hpv[i % (D + 1)] = bld8<T>(rhp, o);
            },
            [&](int i) __attribute__((always_inline)) {
                const int rr = rrow(i);
                float v[8], dh[8], t8[8], uu[8], oo[8], hp[8], dpu[8];
                staged(i, v); unpack8(cyv[i % (D + 1)], dh); unpack8(dhv[i % (D + 1)], t8); unpack8(uv[i % (D + 1)], uu);
                unpack8(ov[i % (D + 1)], oo); unpack8(hpv[i % (D + 1)], hp);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float d = (dh[k] + t8[k]) + v[k];
                    v[k] = d * uu[k] * (1.f - oo[k] * oo[k]);                 // d(pre_o)
                    dpu[k] = d * (oo[k] - hp[k]) * uu[k] * (1.f - uu[k]);
                    dh[k] = d * (1.f - uu[k]);
                }
                bst8<float>(rcy, offs(rr, h, 4, col, colv), dh);
                bst8<T>(rdg, offs(rr, p.g.ldg, esz, col, colv), dpu);
                bst8<T>(rdg, offs(rr, p.g.ldg, esz, 2 * h + col, colv), v);
            });
        return;
    }
    if (mode == 4) {                // BPTT: carry += acc  (dh contribution of the [u|r] backward-data conv)
        constexpr int D = 3;
        const int h = p.g.h;
        const auto rcy = epi_rsrc(p.g.h32n, row0, h * 4, rows);
        Raw8<float> cyv[D + 1];
        epi_run<R, D>(stage,
            [&](int i) __attribute__((always_inline)) { cyv[i % (D + 1)] = bld8<float>(rcy, offs(rrow(i), h, 4, col, colv)); },
            [&](int i) __attribute__((always_inline)) {
                float v[8], cy[8];
                staged(i, v); unpack8(cyv[i % (D + 1)], cy);
#pragma unroll
                for (int k = 0; k < 8; ++k) cy[k] += v[k];
                bst8<float>(rcy, offs(rrow(i), h, 4, col, colv), cy);
            });
        return;
    }
    if (p.ws) {                     // raw split-K partial sums [z][M][Cout]
        if (!(p.Cout & 7)) {
            const auto rws = epi_rsrc(p.ws, (long long)z * p.M + row0, p.Cout * 4, rows);
            epi_run<R, 0>(stage, [&](int) __attribute__((always_inline)) {},
                [&](int i) __attribute__((always_inline)) {
                    float v[8];
                    staged(i, v);
                    bst8<float>(rws, offs(rrow(i), p.Cout, 4, col, colv), v);
                });
        } else {                    // ragged channel count: element stores (tm unrolled: a runtime index into acc parks it in scratch)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                stage(tm);
#pragma unroll 1
                for (int j = 0; j < 4; ++j) {
                    const int rr = relrow(tm, j);
                    if (rr >= 0 && colv) {
                        float* dst = p.ws + ((size_t)z * p.M + row0 + rr) * p.Cout + col;
                        const float* src = ep + (j * 8 + erow) * 64 + ecol;
                        for (int k = 0; k < min(8, p.Cout - col); ++k) dst[k] = src[k];
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        return;
    }
    // direct epilogue: bias, residual (optionally through a nearest x2 upsample), activation, ReLU mask of a backward-data result
    constexpr int D = kB ? 3 : 1;
    const int nvalid = min(8, p.Cout - col);
    float bias8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bias8[k] = (p.bias && k < nvalid) ? p.bias[col + k] : 0.f;
    const bool has_mask = p.mask != nullptr;
    if (p.pool2) {
        // 2 x 2 sums onto the half-size grid (halo kernels: a 32-row sub-tile is two 16-pixel lines of the patch, i.e. eight whole
        // 2 x 2 blocks; staged rows 2j, 2j + 1, 16 + 2j, 17 + 2j -> pooled pixel j = lane >> 3, eight columns per lane)
        const long long lrow0 = row0 >> 2, lrows = (long long)(p.M >> 2) - lrow0;
        const auto rmaskp = epi_rsrc(p.mask, lrow0, p.ldmask * esz, lrows);
        const auto routp = epi_rsrc(p.out, lrow0, p.ldo * (p.out_f32 ? 4u : esz), lrows);
#pragma unroll                                                           // (a runtime tm would index the accumulators dynamically: scratch for the whole kernel)
        for (int tm = 0; tm < TM; ++tm) {
            stage(tm);
            const int base = rrow(tm * 4) - erow;                        // first pixel of the sub-tile's first line, row within the frame
            const int y = base >> p.logW, x = base & (p.W - 1);
            const int lr = (y >> 1) * (p.W >> 1) + (x >> 1) + erow;
            const Raw8<T> mraw = bld8<T>(rmaskp, offs(lr, p.ldmask, esz, col, colv));
            float v[8], m8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = 4.f * bias8[k];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* src = ep + ((q >> 1) * 16 + 2 * erow + (q & 1)) * 64 + ecol;
                const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
                v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3]; v[4] += b[0]; v[5] += b[1]; v[6] += b[2]; v[7] += b[3];
            }
            unpack8(mraw, m8);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v[k] = (has_mask && !(m8[k] > 0.f)) ? 0.f : v[k];
                v[k] = (k < nvalid) ? v[k] : 0.f;
            }
            if (p.out_f32) bst8<float>(routp, offs(lr, p.ldo, 4, col, colv), v);
            else bst8<T>(routp, offs(lr, p.ldo, esz, col, colv), v);
            __builtin_amdgcn_s_waitcnt(0xc07f);                          // the staged rows are consumed before the next sub-tile overwrites them
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    // the residual of a res_up2 conv lives on the half-size grid; its descriptor starts at the tile's first frame there
    const long long res_row0 = p.res_up2 ? (row0 / (p.H * p.W)) * ((p.H >> 1) * (p.W >> 1)) : row0;
    const auto rres = epi_rsrc(p.res, res_row0, p.ldres * esz, (p.res_up2 ? (long long)(p.M >> 2) : (long long)p.M) - res_row0);
    const auto rmask = epi_rsrc(p.mask, row0, p.ldmask * esz, rows);
    const auto rout = epi_rsrc(p.out, row0, p.ldo * (p.out_f32 ? 4u : esz), rows);
    Raw8<T> resv[D + 1], mv[D + 1];
    epi_run<R, D>(stage,
        [&](int i) __attribute__((always_inline)) {
            int rr = rrow(i);
            mv[i % (D + 1)] = bld8<T>(rmask, offs(rr, p.ldmask, esz, col, colv));
            if (p.res_up2 && rr >= 0) {            // residual kept at H/2 x W/2: nearest x2 while reading
                const int row = (int)row0 + rr;
                int f, y, x;
                grid_pos(row, p.H, p.W, p.logH, p.logW, f, y, x);
                rr = (f * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + (x >> 1) - (int)res_row0;
            }
            resv[i % (D + 1)] = bld8<T>(rres, offs(rr, p.ldres, esz, col, colv));
        },
        [&](int i) __attribute__((always_inline)) {
            const int rr = rrow(i);
            float v[8], rv[8], m8[8];
            staged(i, v); unpack8(resv[i % (D + 1)], rv); unpack8(mv[i % (D + 1)], m8);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (v[k] + bias8[k]) + rv[k];
            if (p.act == DVD_ACT_RELU) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
            } else if (p.act == DVD_ACT_TANH) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = gate_tanh<T>(v[k]);
            } else if (p.act == DVD_ACT_SIGMOID) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = gate_sigmoid<T>(v[k]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v[k] = (has_mask && !(m8[k] > 0.f)) ? 0.f : v[k];
                v[k] = (k < nvalid) ? v[k] : 0.f;               // padded channels stay exactly zero
            }
            if (p.out_f32) bst8<float>(rout, offs(rr, p.ldo, 4, col, colv), v);
            else bst8<T>(rout, offs(rr, p.ldo, esz, col, colv), v);
        });
}

// ============================================================================ forward, halo-staged
// Same GEMM as conv_igemm_kernel, but the activation operand is staged ONCE per channel chunk instead of
// once per (chunk, tap).  The M tile is a 2-D patch of one frame (PH x 16 output pixels); its input
// footprint -- the patch plus the filter halo, e.g. 20 x 20 rows for 5 x 5 taps on a 16 x 16 patch --
// is brought into LDS by LDS-DMA (rows outside the frame use an out-of-range offset -> zeros), and the
// 25 (9) taps then read their A fragments from that footprint at a per-tap row offset.  Only the weight
// tile still moves per tap.  Why: with the tap-by-tap gather the DMA *issue* cost (not the traffic) was
// the limiter -- timing experiments with the DMAs predicated off (DVD_CONV_DBG) put the MFMA/ds_read
// loop alone at 1.3-1.7 PF/s and the same loop with only one A tile per chunk at +19..37 %.
//   LDS: 2 halo buffers (chunk cc is multiplied while cc+1 lands) + 3-stage weight ring.
//   kt = 3 (D_t's 3-D convs): the outer index runs over (chunk, dt) and the footprint comes from frame t+dt.
// Footprint image in LDS: 64-byte rows (one channel chunk of one input pixel), row = hy * PITCH + hx with a
// compile-time PITCH (20 = 16 + 2*2 columns; 12 when the nearest-x2 upsample is folded in), 16-byte slot
// XOR-swizzled with ((hx >> 2) + SWA * hy) & 3.  Brute-forced over all taps and both ds_read_b128 lane groups:
// conflict-free for 32 lanes = two 16-pixel patch lines (the row-index swizzle of the tap-by-tap kernel is
// 2-way conflicted here because consecutive patch lines are PITCH, not 16, rows apart).
template <bool UP2> struct HaloGeo {
    static constexpr int PITCH = UP2 ? 12 : 20;
    static constexpr int SWA = UP2 ? 2 : 0;
    static __device__ __forceinline__ int sw(int hy, int hx) { return ((hx >> 2) + SWA * hy) & 3; }
};

// Several independent convolutions served by ONE grid (conv_gb.hip: group_dispatch).  Workgroup b runs on XCD b % 8 as that XCD's
// (b / 8)-th workgroup, and an XCD hands its i-th workgroup to CU i % 32 (tools/place_probe.hip): every XCD takes a contiguous
// eighth of each member's workgroups (tiles x K slices) and walks them in the order
//     `head` of order[0] | `head` of order[1] | rest of order[0] | rest of order[1] | order[2] | ...
// with the members sorted by decreasing K length: the two workgroups that share a CU in the first round belong to different
// members, so they reach their epilogues at different times, and the short tiles fill the tail.  Passed by value (kernel arguments).
constexpr int kGroupMax = 6;
struct ConvGroup {
    int n, head, nslots;
    int order[kGroupMax];                   // members by decreasing K length
    int wgs[kGroupMax];                     // workgroups of member g
    ConvK c[kGroupMax];
};
static_assert(sizeof(ConvGroup) <= 3584, "ConvGroup must fit the kernel argument segment");

// conv_gb.hip (weights from L2, fragment-major): launchers for conv_igemm.hip's dispatch
void launch_gb(const ConvK& p, int variant, bool relu_in, bool up2, dim3 grid, hipStream_t st);
void launch_gbs(const ConvK& p, int S, bool big, bool relu_in, dim3 grid, hipStream_t st);
void launch_group(const ConvGroup& grp, int kind, hipStream_t st);

}  // namespace dvdk
using namespace dvdk;
