// Convolutions of the DVD-GAN step for gfx950: forward / backward-data (implicit GEMM, no im2col) and
// backward-weight, bf16 MFMA with fp32 accumulation or exact fp32 MFMA.
//
//   conv_halo_kernel       3x3 / 5x5 / 3x3x3 filters on frames >= 16 pixels wide: the input footprint of a
//                          16 x 16 pixel patch is staged in LDS once per channel chunk (LDS-DMA), every tap
//                          reads it at an immediate offset; only the weight tile moves per tap.
//   conv_igemm_kernel      1x1 filters and narrow frames: activation tile gathered per (chunk, tap).
//   conv_wgrad_row_kernel  weight gradients, one filter row (all KW taps) per workgroup, reduction over pixels
//                          through transposing LDS reads (ds_read_b64_tr_b16).
//   conv_wgrad_kernel      weight gradients, one tap per workgroup (1x1, upsampling convs, fp32 mode).
//
//   bf16 : v_mfma_f32_32x32x16_bf16   (A: lane l holds row l&31, k = 8*(l>>5)..+7)
//   f32  : v_mfma_f32_32x32x2_f32     (A: lane l holds row l&31, k = l>>5)        exact mode
//   C/D  : col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
// Environment switches read here are for A/B measurements only (DVD_CONV_HALO, DVD_WG_ROW, DVD_WG_TGT, ...);
// none of them changes results.
#include "common.h"

namespace {

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

// acc[tm][tn] += A(TM*32 rows of this wave) x B(64 cols of this wave) over one 64-byte K chunk.
template <typename T, int TM, bool RELU>
__device__ __forceinline__ void mma_swz(const char* At, const char* Bt, int arow, int brow, int lane,
                                        f32x16 (&acc)[TM][2]) {
    const int kh = lane >> 5;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int slot = kk * 2 + kh;
            bf16x8 b[2], a[TM];
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) b[tn] = *reinterpret_cast<const bf16x8*>(Bt + lds_off(brow + tn * 32, slot));
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                u32x4 raw = *reinterpret_cast<const u32x4*>(At + lds_off(arow + tm * 32, slot));
                // ReLU of the conv input (LDS-DMA cannot transform data) is applied on the fragment; it is a
                // compile-time variant: a runtime branch here makes hipcc wait lgkmcnt(0) after every ds_read
                if constexpr (RELU) raw = relu16_bf16(raw);
                a[tm] = __builtin_bit_cast(bf16x8, raw);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int k = kk * 2 + kh;                 // float index 0..15 inside the 64-byte row
            float b[2], a[TM];
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
                b[tn] = *reinterpret_cast<const float*>(Bt + lds_off(brow + tn * 32, k >> 2) + (k & 3) * 4);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                a[tm] = *reinterpret_cast<const float*>(At + lds_off(arow + tm * 32, k >> 2) + (k & 3) * 4);
                if constexpr (RELU) a[tm] = fmaxf(a[tm], 0.f);
            }
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
    }
}

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
// [tile = blockIdx.x][slice][wave][tm][tn][quad][lane] x 16 bytes.  cdna_hip_programming.md section 5 (split-K reduction recipe).
template <int TM>
__device__ __forceinline__ bool splitk_combine(const ConvK& p, f32x16 (&acc)[TM][2], float* lds0, int lane, int z) {
    constexpr unsigned kWaveBytes = TM * 2 * 4 * 1024;
    constexpr int kSc1 = 16;                          // cache-policy bit 4 on gfx950: sc1
    const int ns = p.nsplit;
    const int wave = threadIdx.x >> 6;
    const unsigned tileBytes = (blockDim.x >> 6) * kWaveBytes;
    const size_t tile = blockIdx.x;
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
__device__ __forceinline__ void conv_epilogue(const ConvK& p, f32x16 (&acc)[TM][2], float* ep, int lane, int col, int z,
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
        if (mode != 0 && p.nsplit > 1 && !splitk_combine<TM>(p, acc, ep - (threadIdx.x >> 6) * (32 * 64), lane, z)) return;
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

// Workgroup = 2 x 2 waves, wave tile = (TM*32) x 64  =>  block tile BM = TM*64 rows x 128 columns.
// TM = 4 (256 x 128) is the production shape: 16 MFMAs per wave between barriers and 6 instead of 8
// fragment reads per 8 MFMAs; TM = 2 (128 x 128) serves problems with few rows.
template <typename T, int TM, int WN, bool RELU>   // waves: 2 (M) x WN (N); block tile (TM*64) x (WN*64)
__global__ __launch_bounds__(128 * WN) void conv_igemm_kernel(ConvK p) {
    constexpr int E16 = ElemTraits<T>::kPer16B;
    constexpr int BK = 4 * E16;
    constexpr int BMt = TM * 64, BNt = WN * 64;
    constexpr int NWAVE = 2 * WN;
    constexpr int NA = BMt / 16 / NWAVE;               // 16-row DMA groups of the activation tile per wave
    constexpr int NB = BNt / 16 / NWAVE;               // ... of the weight tile
    constexpr int ABYTES = BMt * 64, BBYTES = BNt * 64;
    constexpr int NSTAGE = 3;                          // LDS ring: tile k is multiplied while k+1, k+2 are in flight
    __shared__ __attribute__((aligned(16))) char smem[NSTAGE][ABYTES + BBYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch order, used for speed only);
    // give each XCD a contiguous run of tiles so the N-tiles of one M-tile share that XCD's L2.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, qd = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (qd + 1) : rr * (qd + 1) + (xcd - rr) * qd) + (bid >> 3);
    }
    // M-major (default): one XCD's contiguous run of tiles shares the activation rows.  N-major, for the recurrent convs of
    // the 4 x 4 / 8 x 8 stages whose weights (up to 26 MB, re-fetched from the Infinity Cache every time step because a
    // step's weights exceed the 4 MiB L2) dwarf the activations: a run shares the WEIGHT tile instead, so each weight block
    // enters one or two L2s rather than all eight.
    int mt, nt;
    if (p.nmajor) { const int tilesM = gridDim.x / p.tilesN; nt = bid / tilesM; mt = bid - nt * tilesM; }
    else { mt = bid / p.tilesN; nt = bid - mt * p.tilesN; }
    const int m0 = mt * BMt, n0 = nt * BNt;
    const int z = blockIdx.z;
    // Pixel-major row order (p.pm = frames F > 0; 4 x 4 / 8 x 8 recurrent convs): GEMM row m' = pixel * F + frame, so a tile
    // holds a few pixels of ONE image line for many frames instead of whole frames -- every row of the tile then has the same
    // set of filter rows that fall outside the frame (2 of 5 at the top / bottom line of a 5 x 5 conv: 30 % of all (line, row)
    // pairs at 4 x 4, 15 % at 8 x 8), and those K steps are skipped for the whole tile instead of multiplying zeros.  The
    // tensors keep their frame-major layout: only the row <-> workgroup assignment changes (memrow), results are identical.
    const int HWp = p.H * p.W;
    auto memrow = [&](int mp) __attribute__((always_inline)) -> int {
        if (!p.pm) return mp;
        const int px = mp / p.pm;
        return (mp - px * p.pm) * HWp + px;
    };
    int iy_lo = 0, iy_hi = p.kh;                       // filter rows that touch the frame for this tile's image line
    if (p.pm) {
        const int y_t = (m0 / p.pm) >> p.logW, pad_ = p.kh >> 1;
        iy_lo = max(0, pad_ - y_t); iy_hi = min(p.kh, p.H + pad_ - y_t);
    }
    const int ntaps = p.pm ? (iy_hi - iy_lo) * p.kw : p.kt * p.kh * p.kw;      // K steps per channel chunk
    const int nk = ntaps * p.kchunks;
    const int per = (nk + p.nsplit - 1) / p.nsplit;
    const int k_begin = z * per;
    const int k_end = min(nk, k_begin + per);
    const size_t esz = sizeof(T);

    // ---- staging: LDS-DMA (buffer_load ... lds).  One wave-instruction moves 64 x 16 B = 16 tile rows
    // straight from L2 into LDS (destination = wave-uniform base + lane*16, i.e. linear), so the tile
    // never passes through VGPRs and no ds_write is issued.  The XOR swizzle of lds_off() is applied
    // to the SOURCE: lane l lands on physical slot l&3 of row l>>2 and therefore fetches logical slot
    // (l&3) ^ ((row>>2)&3).  Rows outside the frame / channels past C use offset 0xFFFFFFFF, for
    // which the buffer range check stores zeros.  Wave w moves row groups w, w+4, ...
    const int wu = __builtin_amdgcn_readfirstlane(wave);
    const int lrow = lane >> 2;                               // row inside a 16-row group
    const int q = (lane & 3) ^ ((lane >> 4) & 3);             // logical 16-byte slot this lane fetches
    int am[NA], ax[NA], ay[NA], at[NA];
    bool av[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int mp = m0 + (i * NWAVE + wu) * 16 + lrow;
        const int m = memrow(mp);
        am[i] = m; av[i] = mp < p.M;
        int f_;
        grid_pos(m, p.H, p.W, p.logH, p.logW, f_, ay[i], ax[i]);
        at[i] = p.kt > 1 ? f_ % p.T : 0;
    }
    int cob[NB];
    bool cov[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) { cob[j] = n0 + (j * NWAVE + wu) * 16 + lrow; cov[j] = cob[j] < p.Cout; }
    // Offsets are 32-bit, activations can exceed 4 GiB: the descriptor of the activation tensor starts
    // at the first input row this tile can touch (wave-uniform), offsets are relative to it.
    const unsigned ldb = (unsigned)p.ldi * (unsigned)esz;        // input row pitch in bytes
    const int base_row = p.pm ? 0 : p.up2 ? (m0 / (p.H * p.W)) * p.Hin * p.Win : max(0, m0 - p.maxshift);
    const size_t base_b = (size_t)base_row * ldb;
    const size_t left_b = p.in_bytes - base_b;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.in + base_b), 0, left_b > 0xfffffffeull ? 0xfffffffeu : (unsigned)left_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    unsigned aoff[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) aoff[i] = (unsigned)(am[i] - base_row) * ldb + q * 16;
    unsigned woff[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) woff[j] = ((unsigned)cob[j] * p.C + q * E16) * (unsigned)esz;
    // wave-uniform K-step state (tap decomposition kept incrementally)
    // K order: channel chunk OUTER, taps INNER -- the 25 (9, 27) taps of one 64-byte channel chunk re-read
    // the same few input rows back to back, so they hit L1/L2 instead of coming back after a whole sweep
    // over C (which at 2 workgroups per CU is tens of MB per XCD, far beyond its 4 MiB L2).
    int tap = 0, cc = 0, it = 0, iy = iy_lo, ix = 0;       // tap = index into the weight pack: (it * kh + iy) * kw + ix
    if (k_begin < k_end) {
        cc = k_begin / ntaps; int rem = k_begin - cc * ntaps;
        if (p.pm) { iy = iy_lo + rem / p.kw; ix = rem - (iy - iy_lo) * p.kw; }
        else {
            it = rem / (p.kh * p.kw); rem -= it * p.kh * p.kw;
            iy = rem / p.kw; ix = rem - iy * p.kw;
        }
        tap = (it * p.kh + iy) * p.kw + ix;
    }
    auto dma = [&](int buf) __attribute__((always_inline)) {
        const int dt_ = it - (p.kt >> 1), dy_ = iy - (p.kh >> 1), dx_ = ix - (p.kw >> 1);
        const bool cv_ = cc * BK + q * E16 < p.C;
        // wave-uniform part of the offset: tap shift + channel chunk
        const unsigned udelta_ = (unsigned)(((dt_ * p.H + dy_) * p.W + dx_) * (int)ldb + cc * 64);
        char* abase_ = &smem[buf][wu * 1024];
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int yy_ = ay[i] + dy_, xx_ = ax[i] + dx_, tt_ = at[i] + dt_;
            const bool ok_ = av[i] && cv_ && (unsigned)yy_ < (unsigned)p.H && (unsigned)xx_ < (unsigned)p.W &&
                             (unsigned)tt_ < (unsigned)p.T;
            unsigned off_ = aoff[i] + udelta_;
            if (p.up2) {
                const int f_ = p.logW >= 0 ? am[i] >> (p.logW + p.logH) : am[i] / (p.H * p.W);
                off_ = (unsigned)(((f_ + dt_) * p.Hin + (yy_ >> 1)) * p.Win + (xx_ >> 1) - base_row) * ldb + cc * 64 + q * 16;
            }
            dma16(rin, abase_ + i * (NWAVE * 1024), ok_ ? off_ : 0xffffffffu);
        }
        const unsigned uw_ = (unsigned)(tap * p.Cout) * (unsigned)p.C * (unsigned)esz + cc * 64;
        char* bbase_ = &smem[buf][ABYTES + wu * 1024];
#pragma unroll
        for (int j = 0; j < NB; ++j) dma16(rw, bbase_ + j * (NWAVE * 1024), (cov[j] && cv_) ? woff[j] + uw_ : 0xffffffffu);
        ++tap;
        if (++ix == p.kw) {
            ix = 0;
            if (++iy == iy_hi) {                        // (iy_lo, iy_hi) = (0, kh) unless pixel-major
                iy = iy_lo;
                if (++it == p.kt) { it = 0; ++cc; }
                tap = (it * p.kh + iy) * p.kw;
            }
        }
    };

    f32x16 acc[TM][2];
    {
        const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = zacc;   // (a per-element loop of 128 stores is not unrolled -> scratch)
    }

    const int arow = wm * (TM * 32) + (lane & 31), brow = wn * 64 + (lane & 31);
#ifdef DVD_EXP_NOMAIN          // compile-time measurement variant: the main loop is skipped
    constexpr bool kMain = false;
#else
    constexpr bool kMain = true;
#endif
    if (kMain && k_begin < k_end) {
        // 3-stage ring, two tiles in flight.  Waits are COUNTED (vmcnt(NA+2) keeps the youngest tile's
        // DMAs outstanding) and the barrier is the raw s_barrier: __syncthreads() would make hipcc drain
        // vmcnt(0) while an LDS-DMA is pending and collapse the pipeline to depth 1.
        constexpr int kDmaPerTile = NA + NB;
        constexpr int kWaitOne = (kDmaPerTile & 0xf) | (7 << 4) | (0xf << 8) | ((kDmaPerTile >> 4) << 14);
        constexpr int kWaitAll = 0 | (7 << 4) | (0xf << 8);
        const int nsteps = k_end - k_begin;
        dma(0);
        if (nsteps > 1) {
            dma(1);
            __builtin_amdgcn_s_waitcnt(kWaitOne);
        } else {
            __builtin_amdgcn_s_waitcnt(kWaitAll);
        }
        __builtin_amdgcn_s_barrier();
        int st = 0, st2 = 2;                                   // stage of tile s, stage of tile s+2
        constexpr bool late = true;
        for (int sidx = 0; sidx < nsteps; ++sidx) {
            const bool issue = sidx + 2 < nsteps;
            // The (slow to issue) LDS-DMAs of a step go out AFTER its MFMA block, while the matrix pipe drains, instead of in
            // front of it where they delay the first fragment reads: +2 % on the 8 x 8 recurrent convolutions (first form:
            // only the second half of the waves of the 8-wave tile did this), see also conv_halo_kernel.
            if (issue && !late) dma(st2);                      // stage st2 was last read in step sidx-1
            mma_swz<T, TM, RELU>(&smem[st][0], &smem[st][ABYTES], arow, brow, lane, acc);
            if (issue && late) dma(st2);
            if (issue) __builtin_amdgcn_s_waitcnt(kWaitOne);   // tile sidx+1 has landed (this wave's part)
            else __builtin_amdgcn_s_waitcnt(kWaitAll);
            __builtin_amdgcn_s_barrier();                      // ... and every other wave's part
            st = st == NSTAGE - 1 ? 0 : st + 1;
            st2 = st2 == NSTAGE - 1 ? 0 : st2 + 1;
        }
    }
    __syncthreads();

    // ---- epilogue: accumulators -> LDS (per wave, 32 rows x 64 columns at a time) -> 8-column vectors.
    // Going through LDS keeps the register->LDS part trivially unrollable (a large branchy epilogue
    // indexed by acc[][][r] is NOT fully unrolled by hipcc and then parks all accumulators in scratch,
    // 6x slower) and turns the global stores into coalesced 16/32-byte vectors.
    float* ep = reinterpret_cast<float*>(&smem[0][0]) + wave * (32 * 64);
    const int ecol = (lane & 7) * 8, erow = lane >> 3;
    if (p.pm) {          // rows of the tile are scattered over the tensor: offsets from its start (small tensors only, see conv_plan)
        conv_epilogue<T, TM, 0, WN == 2>(p, acc, ep, lane, n0 + wn * 64 + ecol, z, 0ll, [&](int tm, int j) __attribute__((always_inline)) {
            const int mp = m0 + wm * (TM * 32) + tm * 32 + j * 8 + erow;
            return mp < p.M ? memrow(mp) : -1;
        });
        return;
    }
    // (WN == 2: the 8-wave tile has no registers to spare for the in-launch split-K combine)
    conv_epilogue<T, TM, 0, WN == 2>(p, acc, ep, lane, n0 + wn * 64 + ecol, z, (long long)m0, [&](int tm, int j) __attribute__((always_inline)) {
        const int rr = wm * (TM * 32) + tm * 32 + j * 8 + erow;
        return m0 + rr < p.M ? rr : -1;
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

// waves: WMV (M) x WN (N); block tile (WMV*TM*32 pixels) x (WN*64).  WMV = 4, TM = 2, WN = 1 is the 256 x 64 tile of
// the thin convs (Cout <= 64: a 128-wide N tile would multiply zeros half of the time)
// compile-time geometry of one halo-kernel variant
template <typename T, int TM, int WN, bool UP2, int WMV> struct HaloCfg {
    static constexpr int PITCH = HaloGeo<UP2>::PITCH;
    static constexpr int BMt = WMV * TM * 32, BNt = WN * 64;
    static constexpr int NWAVE = WMV * WN;
    static constexpr int PH = BMt / 16;                              // patch: PH rows x 16 columns of one frame
    static constexpr int HG = UP2 ? ((PH / 2 + 3) * PITCH + 15) / 16 // 16-row DMA groups of the largest footprint
                                  : ((PH + 4) * PITCH + 15) / 16;    //   (5 x 5 taps)
    static constexpr int HBYTES = HG * 1024, BBYTES = BNt * 64;
    // weight ring depth: NSTAGE-1 tiles in flight.  4 where LDS allows (the 256 x 128 variant runs 2 workgroups per CU)
    // (a fourth stage for the 3 x 3 filters, whose smaller footprint leaves room for it at two workgroups per CU, measured
    //  +0.6 %: the ring depth is not what limits the loop)
    static constexpr int NSTAGE = (TM == 4 && WN == 2) ? 3 : 4;
    static constexpr int EPI = NWAVE * 32 * 64 * 4;
    static constexpr int LDSB = 2 * HBYTES + 1024 + NSTAGE * BBYTES > EPI ? 2 * HBYTES + 1024 + NSTAGE * BBYTES : EPI;
};

// One output tile (M tile mt = a patch of one frame, N tile nt, K slice z) of the halo-staged convolution; `smem`: the
// workgroup's LDS block of HaloCfg::LDSB bytes (the ONLY __shared__ object of the calling kernel: a second one makes hipcc
// drain vmcnt before the LDS reads of every K step).  conv_halo_kernel calls it once per workgroup; a caller that runs
// several tiles in one workgroup (the persistent time-loop experiment of round 3) places a workgroup
// barrier between two tiles.
template <typename T, int TM, int WN, bool RELU, bool UP2, int WMV = 2>
__device__ __forceinline__ void conv_halo_tile(const ConvK& p, char* const smem, const int mt, const int nt, const int z) {
    using G = HaloGeo<UP2>;
    using Cfg = HaloCfg<T, TM, WN, UP2, WMV>;
    constexpr int PITCH = G::PITCH;
    constexpr int E16 = ElemTraits<T>::kPer16B;
    constexpr int BK = 4 * E16;
    constexpr int BMt = Cfg::BMt, BNt = Cfg::BNt;
    constexpr int NWAVE = Cfg::NWAVE;
    constexpr int PH = Cfg::PH;
    constexpr int HG = Cfg::HG;
    constexpr int NH = (HG + NWAVE - 1) / NWAVE;              // footprint DMAs per wave
    constexpr int NB = BNt / 16 / NWAVE;                      // weight-tile DMAs per wave
    constexpr int HBYTES = Cfg::HBYTES, BBYTES = Cfg::BBYTES;
    constexpr int NSTAGE = Cfg::NSTAGE;
    char* const hbuf0 = &smem[0];
    char* const dump = &smem[2 * HBYTES];                     // landing zone of the DMAs of non-existent groups
    char* const bring = &smem[2 * HBYTES + 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = nt * BNt;
    // tile -> (frame, patch origin)
    const int pw = p.W >> 4, ppf = pw * (p.H / PH);
    const int ft = mt / ppf, pidx = mt - ft * ppf;
    const int y0 = (pidx / pw) * PH, x0 = (pidx % pw) * 16;
    const int nouter = p.kchunks * p.kt;                      // outer index oc = cc * kt + it
    const int per = (nouter + p.nsplit - 1) / p.nsplit;
    const int oc_begin = z * per, oc_end = min(nouter, oc_begin + per);
    const int ntap2 = p.kh * p.kw;
    const unsigned esz = sizeof(T);
    const int pad = p.kh >> 1, cpad = (pad + 1) >> 1;
    // footprint extent in INPUT coordinates
    const int HWa = UP2 ? ((15 + pad) >> 1) + cpad + 1 : 16 + 2 * pad;
    const int HHa = UP2 ? ((PH - 1 + pad) >> 1) + cpad + 1 : PH + 2 * pad;
    const int iy_lo = UP2 ? (y0 >> 1) - cpad : y0 - pad, ix_lo = UP2 ? (x0 >> 1) - cpad : x0 - pad;

    const int wu = __builtin_amdgcn_readfirstlane(wave);
    const int lrow = lane >> 2;
    const unsigned ldb = (unsigned)p.ldi * esz;
    const int tt = p.kt > 1 ? ft % p.T : 0;                   // time index of this tile's frame
    const int base_frame = max(0, ft - (p.kt >> 1));
    const size_t fbytes = (size_t)p.Hin * p.Win * ldb;
    const size_t base_b = (size_t)base_frame * fbytes;
    const size_t left_b = p.in_bytes - base_b;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.in + base_b), 0, left_b > 0xfffffffeull ? 0xfffffffeu : (unsigned)left_b, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    unsigned hoff[NH];
    int hq[NH];                                               // logical 16-byte slot this lane fetches, per group
    bool hval[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int g = i * NWAVE + wu;
        const int h = g * 16 + lrow;
        const int hy = h / PITCH, hx = h - hy * PITCH;
        const int yin = iy_lo + hy, xin = ix_lo + hx;
        hq[i] = (lane & 3) ^ G::sw(hy, hx);
        hval[i] = g < HG && hy < HHa && hx < HWa && (unsigned)yin < (unsigned)p.Hin && (unsigned)xin < (unsigned)p.Win;
        hoff[i] = (unsigned)(yin * p.Win + xin) * ldb + hq[i] * 16;
    }
    const int q = (lane & 3) ^ ((lane >> 4) & 3);             // weight tile: row-index swizzle as in conv_igemm_kernel
    int cob[NB];
    bool cov[NB];
    unsigned woff[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        cob[j] = n0 + (j * NWAVE + wu) * 16 + lrow; cov[j] = cob[j] < p.Cout;
        woff[j] = ((unsigned)cob[j] * p.C + q * E16) * esz;
    }
    // footprint of the outer index (cc_, it_) -> halo buffer hb.  (The (chunk, dt) pairs are kept as running counters: an
    // integer division per K step costs ~20 scalar instructions on the wave that is about to issue its MFMAs.)
    auto dmaH = [&](int hb, int cc_, int it_) __attribute__((always_inline)) {
        const int dt_ = it_ - (p.kt >> 1);
        const bool ok_ = (unsigned)(tt + dt_) < (unsigned)p.T || p.kt == 1;
        const unsigned ud_ = (unsigned)(ft + dt_ - base_frame) * (unsigned)fbytes + cc_ * 64;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int g = i * NWAVE + wu;
            char* dst_ = g < HG ? hbuf0 + hb * HBYTES + g * 1024 : dump;
            const bool cv_ = cc_ * BK + hq[i] * E16 < p.C;
            dma16(rin, dst_, (hval[i] && ok_ && cv_) ? hoff[i] + ud_ : 0xffffffffu);
        }
    };
    // weight tile of (chunk, tap): running iterator, two steps ahead of the multiply
    int b_cc = oc_begin / p.kt, b_it = oc_begin - b_cc * p.kt, b_tap = 0;
    auto dmaB = [&](int stg) __attribute__((always_inline)) {
        const bool cv_ = b_cc * BK + q * E16 < p.C;
        const unsigned uw_ = (unsigned)((b_it * ntap2 + b_tap) * p.Cout) * (unsigned)p.C * esz + b_cc * 64;
        char* bbase_ = bring + stg * BBYTES + wu * 1024;
#pragma unroll
        for (int j = 0; j < NB; ++j) dma16(rw, bbase_ + j * (NWAVE * 1024), (cov[j] && cv_) ? woff[j] + uw_ : 0xffffffffu);
        if (++b_tap == ntap2) { b_tap = 0; if (++b_it == p.kt) { b_it = 0; ++b_cc; } }
    };

    f32x16 acc[TM][2];
    {
        const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = zacc;
    }

    // this lane's A rows: pixel (py0 + 2 tm, px) of the patch for its TM 32-row sub-tiles.  Sub-tile tm sits
    // 2 patch lines below sub-tile 0 -> a compile-time LDS offset (2 * PITCH rows; PITCH rows after the x2 fold)
    const int l31 = lane & 31, kh2 = lane >> 5;
    const int px = l31 & 15, py0 = wm * (TM * 2) + (l31 >> 4);
    constexpr int TMSTRIDE = (UP2 ? 1 : 2) * PITCH * 64;
    const int brow = wn * 64 + l31;
#ifdef DVD_EXP_NOMAIN
    const int nsteps = 0;
#else
    const int nsteps = (oc_end - oc_begin) * ntap2;
#endif
    if (nsteps > 0) {
#define VMCNT(n) (((n) & 0xf) | (7 << 4) | (0xf << 8) | (((n) >> 4) << 14))
        constexpr int FL = NSTAGE - 2;                         // weight tiles allowed to stay in flight past tile s+1
        int h_cc = b_cc, h_it = b_it;                          // outer index of the NEXT footprint to fetch
        dmaH(0, h_cc, h_it);
        if (++h_it == p.kt) { h_it = 0; ++h_cc; }
#pragma unroll
        for (int i = 0; i < NSTAGE - 1; ++i)
            if (i < nsteps) dmaB(i);
        if (nsteps > FL) __builtin_amdgcn_s_waitcnt(VMCNT(FL * NB));     // footprint + tile 0 landed
        else __builtin_amdgcn_s_waitcnt(VMCNT(0));
        __builtin_amdgcn_s_barrier();
        // The DMAs of a step go out AFTER its MFMA block (all ds_reads issued, the matrix pipe still draining): +2...7 % on the
        // 5 x 5 shapes, +0...3 % on 3 x 3 against issuing them first.  Measured alternatives: in the middle of the block (after
        // unit 1 / 3 / 5 of 8) -7 %; every other wave or every other workgroup late -4...-8 %.
        constexpr bool late = true;
        int st = 0, st2 = NSTAGE - 1, hb = 0, m_oc = oc_begin, m_tap = 0, iy = 0, ix = 0;
        for (int sidx = 0; sidx < nsteps; ++sidx) {
            const bool issueB = sidx + NSTAGE - 1 < nsteps;
            const bool issueH = m_tap == 0 && m_oc + 1 < oc_end;
            if (!late) {
                if (issueB) dmaB(st2);                         // stage st2 was last read in step sidx-1
                if (issueH) { dmaH(hb ^ 1, h_cc, h_it); if (++h_it == p.kt) { h_it = 0; ++h_cc; } }
            }
            {
                // footprint row of this lane's sub-tile 0 for tap (iy, ix)
                const int hy = UP2 ? ((py0 + iy - pad) >> 1) + cpad : py0 + iy;
                const int hx = UP2 ? ((px + ix - pad) >> 1) + cpad : px + ix;
                const int swz = G::sw(hy, hx);
                const char* Ah = hbuf0 + hb * HBYTES + (hy * PITCH + hx) * 64;
                const char* Bt = bring + st * BBYTES;
                if constexpr (sizeof(T) == 2) {
                    // Explicit software pipeline over the 2*TM (k-half, sub-tile) units: the fragment of unit
                    // u+2 is requested right before the two MFMAs of unit u, pinned with sched_barrier.  (Left
                    // alone hipcc recycles ONE A register set and waits lgkmcnt(0) before every MFMA pair;
                    // all 12 reads up front instead stall on LDS issue: SQ_WAIT_INST_LDS 7x.)
                    constexpr int NU = 2 * TM;
                    bf16x8 b[2][2], a[NU];
                    auto ldB = [&](int kk) __attribute__((always_inline)) {
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn)
                            b[kk][tn] = *reinterpret_cast<const bf16x8*>(Bt + lds_off(brow + tn * 32, kk * 2 + kh2));
                    };
                    auto ldA = [&](int u) __attribute__((always_inline)) {
                        const int kk = u / TM, tm = u % TM, slot = kk * 2 + kh2;
                        // after the x2 fold hy advances by 1 per sub-tile and SWA = 2: odd sub-tiles flip slot bit 1
                        const int sl = (UP2 && (tm & 1)) ? (slot ^ 2) : slot;
                        a[u] = *reinterpret_cast<const bf16x8*>(Ah + tm * TMSTRIDE + ((sl ^ swz) << 4));
                    };
                    ldB(0); ldA(0); ldA(1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        if (u + 2 < NU) {
                            if ((u + 2) % TM == 0) ldB((u + 2) / TM);
                            ldA(u + 2);
                        }
                        const int kk = u / TM, tm = u % TM;
                        if constexpr (RELU) a[u] = __builtin_bit_cast(bf16x8, relu16_bf16(__builtin_bit_cast(u32x4, a[u])));
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[kk][tn], acc[tm][tn], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        const int k = kk * 2 + kh2;
                        float b[2], a[TM];
#pragma unroll
                        for (int tn = 0; tn < 2; ++tn)
                            b[tn] = *reinterpret_cast<const float*>(Bt + lds_off(brow + tn * 32, k >> 2) + (k & 3) * 4);
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm) {
                            const int sl = (UP2 && (tm & 1)) ? ((k >> 2) ^ 2) : (k >> 2);
                            a[tm] = *reinterpret_cast<const float*>(Ah + tm * TMSTRIDE + ((sl ^ swz) << 4) + (k & 3) * 4);
                            if constexpr (RELU) a[tm] = fmaxf(a[tm], 0.f);
                        }
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                            for (int tn = 0; tn < 2; ++tn)
                                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
                    }
                }
            }
#ifndef DVD_EXP_NODMA          // DVD_EXP_*: compile-time measurement variants (tools/build_variant.sh), results are garbage
            if (late) {
                if (issueB) dmaB(st2);
                if (issueH) { dmaH(hb ^ 1, h_cc, h_it); if (++h_it == p.kt) { h_it = 0; ++h_cc; } }
            }
#endif
            // In-order completion: weight tile sidx+1 has landed once at most the FL younger tiles (plus, for the
            // NSTAGE-1 steps after a footprint went out behind tile sidx+NSTAGE-1, that footprint) are outstanding.
#ifndef DVD_EXP_NOWAIT
            {
                const int younger = min(FL, nsteps - 2 - sidx);          // tiles issued beyond sidx+1
                const bool hpend = m_tap < NSTAGE - 1 && m_oc + 1 < oc_end;
                if (younger == FL) {
                    if (hpend) __builtin_amdgcn_s_waitcnt(VMCNT(FL * NB + NH));
                    else __builtin_amdgcn_s_waitcnt(VMCNT(FL * NB));
                } else if (FL == 2 && younger == 1) {
                    __builtin_amdgcn_s_waitcnt(VMCNT(NB));
                } else {
                    __builtin_amdgcn_s_waitcnt(VMCNT(0));
                }
            }
#endif
#ifndef DVD_EXP_NOBAR
            __builtin_amdgcn_s_barrier();
#endif
            st = st == NSTAGE - 1 ? 0 : st + 1;
            st2 = st2 == NSTAGE - 1 ? 0 : st2 + 1;
            ++m_tap;
            if (++ix == p.kw) { ix = 0; ++iy; }
            if (m_tap == ntap2) { m_tap = 0; iy = 0; ix = 0; ++m_oc; hb ^= 1; }
        }
#undef VMCNT
    }
    __syncthreads();

    float* ep = reinterpret_cast<float*>(&smem[0]) + wave * (32 * 64);
    const int ecol = (lane & 7) * 8, erow = lane >> 3;
    const long long frame_row0 = (long long)ft * (p.H * p.W);
    conv_epilogue<T, TM>(p, acc, ep, lane, n0 + wn * 64 + ecol, z, frame_row0, [&](int tm, int j) __attribute__((always_inline)) {
        const int pi = wm * (TM * 32) + tm * 32 + j * 8 + erow;          // pixel of the patch, row-major 16 wide
        return (y0 + (pi >> 4)) * p.W + x0 + (pi & 15);
    });
}

template <typename T, int TM, int WN, bool RELU, bool UP2, int WMV = 2>
// (bf16: two workgroups per CU, i.e. at most 256 registers incl. the accumulators -- without the bound hipcc moves all 128 accumulators
//  into arch VGPRs for the in-launch split-K combine: 348 registers, one workgroup per CU)
__global__ __launch_bounds__(64 * WMV * WN, sizeof(T) == 2 ? 2 : 1) void conv_halo_kernel(ConvK p) {
    __shared__ __attribute__((aligned(16))) char smem[HaloCfg<T, TM, WN, UP2, WMV>::LDSB];
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, qd = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (qd + 1) : rr * (qd + 1) + (xcd - rr) * qd) + (bid >> 3);
    }
    int mt = bid / p.tilesN, nt = bid - mt * p.tilesN;
    if (p.nmajor) { const int tilesM = gridDim.x / p.tilesN; nt = bid / tilesM; mt = bid - nt * tilesM; }
    conv_halo_tile<T, TM, WN, RELU, UP2, WMV>(p, smem, mt, nt, blockIdx.z);
}

// ============================================================================ forward, halo-staged, weights from L2
// conv_halo_tile with the weight operand read STRAIGHT INTO REGISTERS instead of through LDS.  The weights are kept a second
// time in fragment-major order (dvd_conv_fragment_major): record (tap, chunk, 32-column block nb, k-half pair kk) is the 1 KiB
// a wave needs for one B fragment -- lane l = (kh2 = l >> 5, col = l & 31) owns the 8 channels chunk*32 + (2 kk + kh2)*8 .. +7 of
// output column nb*32 + col at byte l*16 -- so a fragment is ONE fully coalesced buffer_load_dwordx4 per lane (8 whole 128-byte
// lines per wave instruction), four per wave and K step.  What that buys: the activation footprint of a channel chunk stays in
// LDS for all 9 / 25 taps, so with the weight tile gone from LDS nothing is handed between waves inside a chunk -- the per-tap
// LDS-DMA of the weight tile (the slowest instruction of the old loop to issue), its counted wait and the per-tap s_barrier all
// disappear; the waves of a workgroup meet once per chunk (footprint hand-over) instead of once per tap.  Price: the two waves
// that share an N half both fetch its fragments (16 KB per workgroup and K step from L1 / L2 instead of 8 KB by DMA).
// Prefetch distance one K step: the fragments of step s+1 are requested right after the first MFMA pair of step s, i.e. behind the
// point where the compiler waits for step s's own fragments (hipcc drains vmcnt(0) there while an LDS-DMA may be pending).
// waves: WMV (M) x WN (N), wave tile (TM*32 pixels) x 64 columns: 4 x 2 x 2 = the 256 x 128 tile, 2 x 2 x 2 = 128 x 128 (launches with
// few rows), 2 x 1 x 4 = 256 x 64 (thin outputs)
#ifndef DVD_GB_EPI_DEEP
#define DVD_GB_EPI_DEEP 0
#endif
template <int TM, int WN, int WMV, bool UP2> struct HaloGbCfg {
    static constexpr int PITCH = HaloGeo<UP2>::PITCH;
    static constexpr int NWAVE = WMV * WN;
    static constexpr int PH = WMV * TM * 2;                             // patch: PH lines x 16 columns
    static constexpr int HG = UP2 ? ((PH / 2 + 3) * PITCH + 15) / 16 : ((PH + 4) * PITCH + 15) / 16;
    static constexpr int HBYTES = HG * 1024;
    static constexpr int EPI = NWAVE * 32 * 64 * 4;
    static constexpr int LDSB = 2 * HBYTES + 1024 > EPI ? 2 * HBYTES + 1024 : EPI;
};

template <int TM, int WN, int WMV, bool RELU, bool UP2>
__device__ __forceinline__ void conv_halo_gb_tile(const ConvK& p, char* const smem, const int mt, const int nt, const int z) {
    using T = bf16_t;
    using G = HaloGeo<UP2>;
    using Cfg = HaloGbCfg<TM, WN, WMV, UP2>;
    constexpr int NWAVE = Cfg::NWAVE;
    constexpr int PITCH = G::PITCH;
    constexpr int BNt = WN * 64;
    constexpr int PH = Cfg::PH, HG = Cfg::HG, HBYTES = Cfg::HBYTES;
    constexpr int NH = (HG + NWAVE - 1) / NWAVE;
    char* const hbuf0 = &smem[0];
    char* const dump = &smem[2 * HBYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = nt * BNt;
    const int pw = p.W >> 4, ppf = pw * (p.H / PH);
    const int ft = mt / ppf, pidx = mt - ft * ppf;
    const int y0 = (pidx / pw) * PH, x0 = (pidx % pw) * 16;
    const int nouter = p.kchunks * p.kt;
    const int per = (nouter + p.nsplit - 1) / p.nsplit;
    const int oc_begin = z * per, oc_end = min(nouter, oc_begin + per);
    const int ntap2 = p.kh * p.kw;
    constexpr unsigned esz = 2;
    const int pad = p.kh >> 1, cpad = (pad + 1) >> 1;
    const int HWa = UP2 ? ((15 + pad) >> 1) + cpad + 1 : 16 + 2 * pad;
    const int HHa = UP2 ? ((PH - 1 + pad) >> 1) + cpad + 1 : PH + 2 * pad;
    const int iy_lo = UP2 ? (y0 >> 1) - cpad : y0 - pad, ix_lo = UP2 ? (x0 >> 1) - cpad : x0 - pad;

    const int wu = __builtin_amdgcn_readfirstlane(wave);
    const int lrow = lane >> 2;
    const unsigned ldb = (unsigned)p.ldi * esz;
    const int tt = p.kt > 1 ? ft % p.T : 0;
    const int base_frame = max(0, ft - (p.kt >> 1));
    const size_t fbytes = (size_t)p.Hin * p.Win * ldb;
    const size_t base_b = (size_t)base_frame * fbytes;
    const size_t left_b = p.in_bytes - base_b;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.in + base_b), 0, left_b > 0xfffffffeull ? 0xfffffffeu : (unsigned)left_b, 0x00020000);
    // fragment-major weights: [tap][chunk][nb32][kk][lane][16 B]; nb32 = 32-column blocks, padded to whole 128-column tiles
    const int nb32 = p.nb32;
    const __amdgpu_buffer_rsrc_t rwq = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.wq, 0, (unsigned)((size_t)p.kt * ntap2 * p.kchunks * nb32 * 2048), 0x00020000);
    unsigned hoff[NH];
    int hq[NH];
    bool hval[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int g = i * NWAVE + wu;
        const int h = g * 16 + lrow;
        const int hy = h / PITCH, hx = h - hy * PITCH;
        const int yin = iy_lo + hy, xin = ix_lo + hx;
        hq[i] = (lane & 3) ^ G::sw(hy, hx);
        hval[i] = g < HG && hy < HHa && hx < HWa && (unsigned)yin < (unsigned)p.Hin && (unsigned)xin < (unsigned)p.Win;
        hoff[i] = (unsigned)(yin * p.Win + xin) * ldb + hq[i] * 16;
    }
    auto dmaH = [&](int hb, int cc_, int it_) __attribute__((always_inline)) {
        const int dt_ = it_ - (p.kt >> 1);
        const bool ok_ = (unsigned)(tt + dt_) < (unsigned)p.T || p.kt == 1;
        const unsigned ud_ = (unsigned)(ft + dt_ - base_frame) * (unsigned)fbytes + cc_ * 64;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int g = i * NWAVE + wu;
            char* dst_ = g < HG ? hbuf0 + hb * HBYTES + g * 1024 : dump;
            const bool cv_ = cc_ * 32 + hq[i] * 8 < p.C;
            dma16(rin, dst_, (hval[i] && ok_ && cv_) ? hoff[i] + ud_ : 0xffffffffu);
        }
    };
    // B fragments: voffset = this wave's column half + lane slot (per lane), soffset = record of (tap, chunk, N tile) (uniform)
    const unsigned bvoff = (unsigned)(wn * 2 * 2048 + lane * 16);
    const unsigned bnt = (unsigned)(nt * (BNt / 32)) * 2048u;
    const unsigned brec = (unsigned)nb32 * 2048u;                       // bytes per (tap, chunk)
    auto ldBq = [&](bf16x8 (&b)[2][2], unsigned rec) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
                b[kk][tn] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rwq, bvoff, rec + (unsigned)(tn * 2048 + kk * 1024), 0));
    };

    f32x16 acc[TM][2];
    {
        const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = zacc;
    }
    const int l31 = lane & 31, kh2 = lane >> 5;
    const int px = l31 & 15, py0 = wm * (TM * 2) + (l31 >> 4);
    constexpr int TMSTRIDE = (UP2 ? 1 : 2) * PITCH * 64;
#ifdef DVD_EXP_NOMAIN
    const int nsteps = 0;
#else
    const int nsteps = (oc_end - oc_begin) * ntap2;
#endif
    if (nsteps > 0) {
        // running (chunk, dt) of the step being multiplied / of the next footprint, and the FETCH iterator of the weight records
        // (memory order [tap][chunk]), which runs GB_DIST K steps ahead and stops at the last record (the trailing requests re-read it:
        // they are UNCONDITIONAL so that the compiler's counted waits know them outstanding -- behind a branch hipcc assumes the
        // smaller count and drains the prefetch in the middle of a step)
#ifdef DVD_GB_DIST2
        constexpr int GB_DIST = 2;
#else
        constexpr int GB_DIST = 1;
#endif
        int m_cc = oc_begin / p.kt, m_it = oc_begin - m_cc * p.kt;
        int h_cc = m_cc, h_it = m_it;
        int f_cc = m_cc, f_it = m_it, f_tap = 0, f_left = nsteps;
        auto frec = [&]() __attribute__((always_inline)) -> unsigned {
            return ((unsigned)(f_it * ntap2 + f_tap) * p.kchunks + f_cc) * brec + bnt;
        };
        auto fadv = [&]() __attribute__((always_inline)) {
            if (f_left > 1) { --f_left; if (++f_tap == ntap2) { f_tap = 0; if (++f_it == p.kt) { f_it = 0; ++f_cc; } } }
        };
        bf16x8 bq0[2][2], bq1[2][2];
        dmaH(0, h_cc, h_it);
        if (++h_it == p.kt) { h_it = 0; ++h_cc; }
        ldBq(bq0, frec()); fadv();
#ifdef DVD_GB_DIST2
        bf16x8 bq2[2][2];
        ldBq(bq1, frec()); fadv();
#endif
        __builtin_amdgcn_s_waitcnt(0x0070 | (0xf << 8));                // vmcnt(0): footprint 0 (this wave's part) landed
        __builtin_amdgcn_s_barrier();
#ifdef DVD_GB_PRIO
        if (__builtin_amdgcn_readfirstlane((int)(blockIdx.x >> 8) & 1)) __builtin_amdgcn_s_setprio(1);   // the second workgroup of a CU
#endif
        int hb = 0, m_tap = 0, m_oc = oc_begin, iy = 0, ix = 0;
        auto step = [&](bf16x8 (&b)[2][2], bf16x8 (&bn)[2][2], int sidx) __attribute__((always_inline)) {
            const bool more = sidx + 1 < nsteps;
            const bool last_tap = m_tap + 1 == ntap2;
            const bool issueH = m_tap == 0 && m_oc + 1 < oc_end;
            const int hy = UP2 ? ((py0 + iy - pad) >> 1) + cpad : py0 + iy;
            const int hx = UP2 ? ((px + ix - pad) >> 1) + cpad : px + ix;
            const int swz = G::sw(hy, hx);
            const char* Ah = hbuf0 + hb * HBYTES + (hy * PITCH + hx) * 64;
            constexpr int NU = 2 * TM;
            bf16x8 a[NU];
            auto ldA = [&](int u) __attribute__((always_inline)) {
                const int kk = u / TM, tm = u % TM, slot = kk * 2 + kh2;
                const int sl = (UP2 && (tm & 1)) ? (slot ^ 2) : slot;
                a[u] = *reinterpret_cast<const bf16x8*>(Ah + tm * TMSTRIDE + ((sl ^ swz) << 4));
            };
            ldA(0); ldA(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (u + 2 < NU) ldA(u + 2);
                const int kk = u / TM, tm = u % TM;
                if constexpr (RELU) a[u] = __builtin_bit_cast(bf16x8, relu16_bf16(__builtin_bit_cast(u32x4, a[u])));
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[kk][tn], acc[tm][tn], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (u == 0) {
                    // behind the first MFMA pair (the compiler's wait for THIS step's fragments sits in front of it): request the
                    // fragments GB_DIST steps ahead, and at the first tap of a chunk the next chunk's footprint
#ifndef DVD_EXP_NODMA
                    ldBq(bn, frec()); fadv();
                    if (issueH) { dmaH(hb ^ 1, h_cc, h_it); if (++h_it == p.kt) { h_it = 0; ++h_cc; } }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            ++m_tap;
            if (++ix == p.kw) { ix = 0; ++iy; }
            if (last_tap) {
                // chunk boundary: every wave has finished reading footprint hb (it is overwritten one chunk from now) and has seen
                // its own part of footprint hb ^ 1 land (the in-order drain in front of the steps since): one barrier per chunk
                m_tap = 0; iy = 0; ix = 0; ++m_oc; hb ^= 1;
                if (++m_it == p.kt) { m_it = 0; ++m_cc; }
                if (more) {
#ifndef DVD_EXP_NOWAIT
                    __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): this wave's fragment reads are done
#endif
#ifndef DVD_EXP_NOBAR
                    __builtin_amdgcn_s_barrier();
#endif
                }
            }
        };
        int sidx = 0;
#ifdef DVD_GB_DIST2
        for (; sidx + 2 < nsteps; sidx += 3) { step(bq0, bq2, sidx); step(bq1, bq0, sidx + 1); step(bq2, bq1, sidx + 2); }
        if (sidx < nsteps) { step(bq0, bq2, sidx); ++sidx; }
        if (sidx < nsteps) { step(bq1, bq0, sidx); ++sidx; }
#else
        for (; sidx + 1 < nsteps; sidx += 2) { step(bq0, bq1, sidx); step(bq1, bq0, sidx + 1); }
        if (sidx < nsteps) step(bq0, bq1, sidx);
#endif
    }
    __syncthreads();

    float* ep = reinterpret_cast<float*>(&smem[0]) + wave * (32 * 64);
    const int ecol = (lane & 7) * 8, erow = lane >> 3;
    const long long frame_row0 = (long long)ft * (p.H * p.W);
    conv_epilogue<T, TM, DVD_GB_EPI_DEEP>(p, acc, ep, lane, n0 + wn * 64 + ecol, z, frame_row0, [&](int tm, int j) __attribute__((always_inline)) {
        const int pi = wm * (TM * 32) + tm * 32 + j * 8 + erow;
        return (y0 + (pi >> 4)) * p.W + x0 + (pi & 15);
    });
}

template <int TM, int WN, int WMV, bool RELU, bool UP2>
__global__ __launch_bounds__(256, 2) void conv_halo_gb_kernel(ConvK p) {
    __shared__ __attribute__((aligned(16))) char smem[HaloGbCfg<TM, WN, WMV, UP2>::LDSB];
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, qd = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (qd + 1) : rr * (qd + 1) + (xcd - rr) * qd) + (bid >> 3);
    }
    int mt = bid / p.tilesN, nt = bid - mt * p.tilesN;
    if (p.nmajor) { const int tilesM = gridDim.x / p.tilesN; nt = bid / tilesM; mt = bid - nt * tilesM; }
    conv_halo_gb_tile<TM, WN, WMV, RELU, UP2>(p, smem, mt, nt, blockIdx.z);
}

// ---------------------------------------------------------------------------- the same for frames of 8 x 8 and 4 x 4 pixels
// The recurrent convolutions of the first two generator stages (Generator.py:39,43: ConvGRUs on 4 x 4 / 8 x 8 latents) ran through
// the tap-by-tap kernel: per (chunk, tap) one LDS-DMA gather of the activation tile, one of the weight tile, a counted wait and a
// barrier -- K steps of pure issue / wait latency.  Here the M tile is G WHOLE frames (TM*64 / S^2 of them), whose zero-padded
// footprints ((S+4) x (S+4) rows of 64 bytes per frame, PITCH = S + 4 whatever the filter size) sit in LDS for all 9 / 25 taps of a
// channel chunk, and the weights come from L2 in fragment-major order as in conv_halo_gb_tile: no per-tap DMA, wait or barrier.
// 16-byte slot swizzle: (line of the footprint) & 3 -- brute-forced conflict-free for both ds_read_b128 lane groups, every tap,
// S = 8 (a 32-row sub-tile = 4 lines of one frame) and S = 4 (= 2 frames).
template <int TM, int S> struct HaloGbsCfg {
    static constexpr int PITCH = S + 4, FR = PITCH * PITCH;             // rows per frame footprint
    static constexpr int G = TM * 64 / (S * S);                         // frames per tile
    static constexpr int HG = (G * FR + 15) / 16;
    static constexpr int HBYTES = HG * 1024;
    static constexpr int EPI = 4 * 32 * 64 * 4;
    static constexpr int LDSB = 2 * HBYTES + 1024 > EPI ? 2 * HBYTES + 1024 : EPI;
};

template <int TM, int S, bool RELU>
__device__ __forceinline__ void conv_halo_gbs_tile(const ConvK& p, char* const smem, const int mt, const int nt, const int z) {
    using T = bf16_t;
    using Cfg = HaloGbsCfg<TM, S>;
    constexpr int WN = 2, NWAVE = 4, BNt = 128;
    constexpr int PITCH = Cfg::PITCH, FR = Cfg::FR, G = Cfg::G, HG = Cfg::HG, HBYTES = Cfg::HBYTES;
    constexpr int NH = (HG + NWAVE - 1) / NWAVE;
    constexpr int SS = S * S;
    char* const hbuf0 = &smem[0];
    char* const dump = &smem[2 * HBYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n0 = nt * BNt;
    const int ft0 = mt * G, nframes = p.M / SS;
    const int per = (p.kchunks + p.nsplit - 1) / p.nsplit;
    const int cc_begin = z * per, cc_end = min(p.kchunks, cc_begin + per);
    const int ntap2 = p.kh * p.kw;
    constexpr unsigned esz = 2;
    const int pad = p.kh >> 1, ext = S + 2 * pad;                      // footprint extent actually used by this filter

    const int wu = __builtin_amdgcn_readfirstlane(wave);
    const int lrow = lane >> 2;
    const unsigned ldb = (unsigned)p.ldi * esz;
    const size_t base_b = (size_t)ft0 * SS * ldb;
    const size_t left_b = p.in_bytes - base_b;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.in + base_b), 0, left_b > 0xfffffffeull ? 0xfffffffeu : (unsigned)left_b, 0x00020000);
    const int nb32 = p.nb32;
    const __amdgpu_buffer_rsrc_t rwq = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.wq, 0, (unsigned)((size_t)ntap2 * p.kchunks * nb32 * 2048), 0x00020000);
    unsigned hoff[NH];
    int hq[NH];
    bool hval[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int g = i * NWAVE + wu;
        const int h = g * 16 + lrow;
        const int fl = h / FR, r = h - fl * FR;
        const int hy = r / PITCH, hx = r - hy * PITCH;
        const int yin = hy - pad, xin = hx - pad;
        hq[i] = (lane & 3) ^ (hy & 3);
        hval[i] = g < HG && fl < G && ft0 + fl < nframes && hy < ext && hx < ext && (unsigned)yin < (unsigned)S && (unsigned)xin < (unsigned)S;
        hoff[i] = (unsigned)((fl * S + yin) * S + xin) * ldb + hq[i] * 16;
    }
    auto dmaH = [&](int hb, int cc_) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int g = i * NWAVE + wu;
            char* dst_ = g < HG ? hbuf0 + hb * HBYTES + g * 1024 : dump;
            const bool cv_ = cc_ * 32 + hq[i] * 8 < p.C;
            dma16(rin, dst_, (hval[i] && cv_) ? hoff[i] + cc_ * 64 : 0xffffffffu);
        }
    };
    const unsigned bvoff = (unsigned)(wn * 2 * 2048 + lane * 16);
    const unsigned bnt = (unsigned)(nt * (BNt / 32)) * 2048u;
    const unsigned brec = (unsigned)nb32 * 2048u;
    auto ldBq = [&](bf16x8 (&b)[2][2], unsigned rec) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
                b[kk][tn] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rwq, bvoff, rec + (unsigned)(tn * 2048 + kk * 1024), 0));
    };

    f32x16 acc[TM][2];
    {
        const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = zacc;
    }
    // this lane's A rows: pixel pi = (wm*TM + tm)*32 + l31 of the tile -> (frame, y, x); sub-tile tm sits a compile-time number of
    // footprint rows further (S = 8: 4 lines or a whole frame, S = 4: two frames); y & 3 is the same for every tm
    const int l31 = lane & 31, kh2 = lane >> 5;
    const int pi0 = wm * (TM * 32) + l31;
    const int f0 = pi0 / SS, y0 = (pi0 % SS) / S, x0 = pi0 % S;
    const int arow0 = f0 * FR + y0 * PITCH + x0;
    auto tmoff = [](int tm) constexpr -> int { return S == 8 ? ((tm >> 1) * FR + (tm & 1) * 4 * PITCH) * 64 : tm * 2 * FR * 64; };
#ifdef DVD_EXP_NOMAIN
    const int nsteps = 0;
#else
    const int nsteps = (cc_end - cc_begin) * ntap2;
#endif
    if (nsteps > 0) {
        int m_cc = cc_begin, h_cc = cc_begin;
        unsigned rec = (unsigned)m_cc * brec + bnt;                     // tap 0 of chunk m_cc; records are [tap][chunk]
        const unsigned tapstep = (unsigned)p.kchunks * brec;
        bf16x8 bq0[2][2], bq1[2][2];
        dmaH(0, h_cc); ++h_cc;
        ldBq(bq0, rec);
        __builtin_amdgcn_s_waitcnt(0x0070 | (0xf << 8));                // vmcnt(0)
        __builtin_amdgcn_s_barrier();
        int hb = 0, m_tap = 0, iy = 0, ix = 0;
        auto step = [&](bf16x8 (&b)[2][2], bf16x8 (&bn)[2][2], int sidx) __attribute__((always_inline)) {
            const bool more = sidx + 1 < nsteps;
            const bool last_tap = m_tap + 1 == ntap2;
            const bool issueH = m_tap == 0 && m_cc + 1 < cc_end;
            unsigned nrec = more ? rec + tapstep : rec;
            if (last_tap && more) nrec = (unsigned)(m_cc + 1) * brec + bnt;
            const int swz = (y0 + iy) & 3;
            const char* Ah = hbuf0 + hb * HBYTES + (arow0 + iy * PITCH + ix) * 64;
            constexpr int NU = 2 * TM;
            bf16x8 a[NU];
            auto ldA = [&](int u) __attribute__((always_inline)) {
                const int kk = u / TM, tm = u % TM, slot = kk * 2 + kh2;
                a[u] = *reinterpret_cast<const bf16x8*>(Ah + tmoff(tm) + ((slot ^ swz) << 4));
            };
            ldA(0); ldA(1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (u + 2 < NU) ldA(u + 2);
                const int kk = u / TM, tm = u % TM;
                if constexpr (RELU) a[u] = __builtin_bit_cast(bf16x8, relu16_bf16(__builtin_bit_cast(u32x4, a[u])));
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[kk][tn], acc[tm][tn], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (u == 0) {
                    ldBq(bn, nrec);
                    if (issueH) { dmaH(hb ^ 1, h_cc); ++h_cc; }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            rec = nrec;
            ++m_tap;
            if (++ix == p.kw) { ix = 0; ++iy; }
            if (last_tap) {
                m_tap = 0; iy = 0; ix = 0; ++m_cc; hb ^= 1;
                if (more) {
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_s_barrier();
                }
            }
        };
        int sidx = 0;
        for (; sidx + 1 < nsteps; sidx += 2) { step(bq0, bq1, sidx); step(bq1, bq0, sidx + 1); }
        if (sidx < nsteps) step(bq0, bq1, sidx);
    }
    __syncthreads();

    float* ep = reinterpret_cast<float*>(&smem[0]) + wave * (32 * 64);
    const int ecol = (lane & 7) * 8, erow = lane >> 3;
    const long long row0 = (long long)ft0 * SS;
    conv_epilogue<T, TM, DVD_GB_EPI_DEEP>(p, acc, ep, lane, n0 + wn * 64 + ecol, z, row0, [&](int tm, int j) __attribute__((always_inline)) {
        const int pi = wm * (TM * 32) + tm * 32 + j * 8 + erow;          // the tile's rows are G consecutive whole frames
        return row0 + pi < p.M ? pi : -1;
    });
}

template <int TM, int S, bool RELU>
__global__ __launch_bounds__(256, 2) void conv_halo_gbs_kernel(ConvK p) {
    __shared__ __attribute__((aligned(16))) char smem[HaloGbsCfg<TM, S>::LDSB];
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, qd = nwg >> 3, rr = nwg & 7;
        bid = (xcd < rr ? xcd * (qd + 1) : rr * (qd + 1) + (xcd - rr) * qd) + (bid >> 3);
    }
    int mt = bid / p.tilesN, nt = bid - mt * p.tilesN;
    if (p.nmajor) { const int tilesM = gridDim.x / p.tilesN; nt = bid / tilesM; mt = bid - nt * tilesM; }
    conv_halo_gbs_tile<TM, S, RELU>(p, smem, mt, nt, blockIdx.z);
}

// standard forward pack [tap][Cout][C] (bf16) -> fragment-major [tap][chunk][nb32][kk][lane][8] (zeros in every padded position)
struct FragK { const bf16_t* w; bf16_t* wq; int ntaps, Cout, C, kchunks, nb32; };
__global__ void fragment_major_kernel(FragK p) {
    const long long n = (long long)p.ntaps * p.kchunks * p.nb32 * 2 * 64;             // 16-byte units
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int l = (int)(i & 63);
    long long r = i >> 6;
    const int kk = (int)(r & 1); r >>= 1;
    const int nb = (int)(r % p.nb32); r /= p.nb32;
    const int cc = (int)(r % p.kchunks);
    const int tap = (int)(r / p.kchunks);
    const int co = nb * 32 + (l & 31), ci = cc * 32 + (kk * 2 + (l >> 5)) * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (co < p.Cout && ci < p.C) v = *reinterpret_cast<const u32x4*>(p.w + ((size_t)tap * p.Cout + co) * p.C + ci);   // C % 8 == 0
    *reinterpret_cast<u32x4*>(p.wq + i * 8) = v;
}

// ============================================================================ backward-weight
struct WgK {
    const char* x; const char* dy; float* dw;
    int M, C, ldx, Cin_real, Cout, Cy, ldy;
    int T, H, W, logH, logW, Hin, Win;
    int kt, kh, kw, up2, relu_in;
    int tiles_co, tiles_ci, rows_per_split;
    long long s_co, s_ci, s_tap;
    size_t x_bytes, dy_bytes;
    int maxshift, xcd_remap;
    float* dbias;
    float* wsb;             // row kernel, workspace path: bias partial sums [slice][workgroup][BMc] behind the tile partials
    float* ws;                           // partial tiles [slice][x-block][BMc*BNc] when non-null
};

// D[co][ci] = sum over rows m of dy[m][co] * x[pos(m)+tap][ci].  The reduction index (rows) is the
// MFMA K dimension, so both operand tiles must be K(row)-contiguous per channel in LDS:
//   bf16: tiles are staged in their natural [row][channel] image with 16-byte loads and the
//         fragments are built by the LDS transpose read ds_read_b64_tr_b16;
//   f32 : the natural [row][channel] image already matches the 32x32x2 fragment (1 float/lane).
template <typename T, int TA, int TB>      // wave tile (TA*32 out-channels) x (TB*32 in-channels); block = 2 x 2 waves
__global__ __launch_bounds__(NT) void conv_wgrad_kernel(WgK p) {
    constexpr bool kBf16 = sizeof(T) == 2;
    static_assert(kBf16 || (TA == 2 && TB == 2), "exact mode uses the 128 x 128 tile");
    constexpr int BKW = kBf16 ? 32 : 16;            // rows reduced per step
    constexpr int BMc = TA * 64, BNc = TB * 64;     // block tile: out-channels x in-channels
    constexpr int RSA = BMc * 2 + 64, RSB = BNc * 2 + 64;          // bf16 LDS row strides (bytes)
    constexpr int TA_BYTES = kBf16 ? 32 * RSA : TILEB, TB_BYTES = kBf16 ? 32 * RSB : TILEB;
    __shared__ __attribute__((aligned(16))) char smem[2][TA_BYTES + TB_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware order over (x, z): each XCD gets a contiguous run of row slices, whose workgroups (all
    // taps / channel tiles of a slice) then share the slice's rows in that XCD's L2 (hit rate 0.42 -> 0.73
    // on the 3x3 shapes, 0.81 -> 0.89 on the 5x5 ones; tools/pmc_l2.txt)
    int bx = blockIdx.x, bz = blockIdx.z;
    if (p.xcd_remap) {
        const int gx = gridDim.x, total = gx * gridDim.z, lin = bx + gx * bz;
        const int xcd = lin & 7, qd = total >> 3, rr = total & 7;
        const int lp = (xcd < rr ? xcd * (qd + 1) : rr * (qd + 1) + (xcd - rr) * qd) + (lin >> 3);
        bz = lp / gx; bx = lp - bz * gx;
    }
    const int tiles = p.tiles_co * p.tiles_ci;
    const int tap = bx / tiles;
    const int rem = bx - tap * tiles;
    const int tco = rem / p.tiles_ci, tci = rem - tco * p.tiles_ci;
    const int co0 = tco * BMc, ci0 = tci * BNc;
    const int it = tap / (p.kh * p.kw), r2 = tap - it * p.kh * p.kw;
    const int iy = r2 / p.kw, ix = r2 - iy * p.kw;
    const int dt = it - (p.kt >> 1), dy_ = iy - (p.kh >> 1), dx = ix - (p.kw >> 1);
    const int m_begin = bz * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);

    f32x16 acc[TA][TB];
    {
        const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int b = 0; b < TB; ++b) acc[a][b] = zacc;
    }

    // Buffer descriptors: invalid rows / channels use offset 0xFFFFFFFF -> hardware returns zeros.
    // (32-bit offsets relative to the first row this workgroup's row slice can touch: tensors > 4 GiB ok)
    constexpr unsigned ESZ = sizeof(T);
    const int xbase_row = p.up2 ? (m_begin / (p.H * p.W)) * p.Hin * p.Win : max(0, m_begin - p.maxshift);
    const size_t xbase_b = (size_t)xbase_row * p.ldx * ESZ, ybase_b = (size_t)m_begin * p.ldy * ESZ;
    const size_t xleft = p.x_bytes - xbase_b, yleft = p.dy_bytes > ybase_b ? p.dy_bytes - ybase_b : 0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + xbase_b), 0, xleft > 0xfffffffeull ? 0xfffffffeu : (unsigned)xleft, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.dy + ybase_b), 0, yleft > 0xfffffffeull ? 0xfffffffeu : (unsigned)yleft, 0x00020000);
    // Byte offset of the shifted x row for output row m.  Without upsampling the input grid equals
    // the output grid, so the shifted row is simply m + delta.
    const int delta = (dt * p.H + dy_) * p.W + dx;
    auto xoff = [&](int m, unsigned chan_bytes, bool cvalid) __attribute__((always_inline)) -> unsigned {
        int fm, ym, xm;
        grid_pos(m, p.H, p.W, p.logH, p.logW, fm, ym, xm);
        const int xx = xm + dx, yy = ym + dy_;
        int tt = dt;
        if (p.kt > 1) tt += fm % p.T;
        const bool ok = cvalid && m < m_end && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W &&
                        (unsigned)tt < (unsigned)p.T;
        int row = m + delta;
        if (p.up2) row = ((fm + dt) * p.Hin + (yy >> 1)) * p.Win + (xx >> 1);
        return ok ? (unsigned)(row - xbase_row) * ((unsigned)p.ldx * ESZ) + chan_bytes : 0xffffffffu;
    };
    auto yoff = [&](int m, unsigned chan_bytes, bool cvalid) __attribute__((always_inline)) -> unsigned {
        return (cvalid && m < m_end) ? (unsigned)(m - m_begin) * ((unsigned)p.ldy * ESZ) + chan_bytes : 0xffffffffu;
    };

    if constexpr (kBf16) {
        // Natural [row][channel] LDS image (row stride 320 B = 256 B of channels + 64 B: the four
        // rows a 16-lane group touches land on disjoint bank quarters).  MFMA fragments need 8
        // consecutive ROWS of one channel per lane: ds_read_b64_tr_b16 delivers exactly that
        // (each 16-lane group reads a 4-row x 16-channel block and hands lane i column i).
        // operand A = dy (BMc channels per row), operand B = x (BNc channels per row)
        constexpr int CPRA = BMc / 8, RPPA = NT / CPRA, NPA = 32 / RPPA;       // chunks/row, rows/pass, passes
        constexpr int CPRB = BNc / 8, RPPB = NT / CPRB, NPB = 32 / RPPB;
        const int rra = tid / CPRA, cka = tid % CPRA, rrb = tid / CPRB, ckb = tid % CPRB;
        const int cy = co0 + cka * 8, cx = ci0 + ckb * 8;
        const bool cyv = cy < p.Cy, cxv = cx < p.C;
        u32x4 ra[NPA], rb[NPB];
        auto wg_gload = [&](int mk) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NPA; ++i)
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rdy, yoff(mk + rra + i * RPPA, cy * 2, cyv), 0, 0);
#pragma unroll
            for (int i = 0; i < NPB; ++i)
                rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff(mk + rrb + i * RPPB, cx * 2, cxv), 0, 0);
        };
        // fused bias gradient: the centre-tap / first-ci-tile workgroups sum the dy chunks they stage
        const bool do_bias = p.dbias != nullptr && dt == 0 && dy_ == 0 && dx == 0 && tci == 0;
        float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        auto wg_lstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < NPA; ++i) {
                *reinterpret_cast<u32x4*>(&smem[buf][(rra + i * RPPA) * RSA + cka * 16]) = ra[i];
                if (do_bias) {
                    bs[0] += __uint_as_float(ra[i].x << 16); bs[1] += __uint_as_float(ra[i].x & 0xffff0000u);
                    bs[2] += __uint_as_float(ra[i].y << 16); bs[3] += __uint_as_float(ra[i].y & 0xffff0000u);
                    bs[4] += __uint_as_float(ra[i].z << 16); bs[5] += __uint_as_float(ra[i].z & 0xffff0000u);
                    bs[6] += __uint_as_float(ra[i].w << 16); bs[7] += __uint_as_float(ra[i].w & 0xffff0000u);
                }
            }
#pragma unroll
            for (int i = 0; i < NPB; ++i)
                *reinterpret_cast<u32x4*>(&smem[buf][TA_BYTES + (rrb + i * RPPB) * RSB + ckb * 16]) =
                    p.relu_in ? relu16_bf16(rb[i]) : rb[i];
        };
#define WG_GLOAD(mk) wg_gload(mk)
#define WG_LSTORE(buf) wg_lstore(buf)
        // per-lane offset of its 8-byte chunk inside a 4-row x 16-channel block of a fragment:
        // rows (g16>>1)*8 + (i16>>2), channels (g16&1)*16 + (i16&3)*4
        const int g16 = lane >> 4, i16 = lane & 15;
        const int frow = (g16 >> 1) * 8 + (i16 >> 2), fcol2 = ((g16 & 1) * 16 + (i16 & 3) * 4) * 2;
        auto frag = [&](const char* tile, int rs, int col0, int kb) __attribute__((always_inline)) -> bf16x8 {
            typedef __attribute__((ext_vector_type(4))) short s16x4;
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            const char* pz = tile + (frow + kb) * rs + fcol2 + col0 * 2;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pz);
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pz + 4 * rs));
            s16x8 f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            return __builtin_bit_cast(bf16x8, f);
        };
        auto mma = [&](int buf) __attribute__((always_inline)) {
            const char* At = &smem[buf][0];
            const char* Bt = &smem[buf][TA_BYTES];
            // all fragment reads of the 32-row step are issued up front: the second half's transpose
            // reads land while the first half's MFMAs run
            bf16x8 fa0[TA], fb0[TB], fa1[TA], fb1[TB];
#pragma unroll
            for (int a = 0; a < TA; ++a) fa0[a] = frag(At, RSA, wm * (TA * 32) + a * 32, 0);
#pragma unroll
            for (int b = 0; b < TB; ++b) fb0[b] = frag(Bt, RSB, wn * (TB * 32) + b * 32, 0);
#pragma unroll
            for (int a = 0; a < TA; ++a) fa1[a] = frag(At, RSA, wm * (TA * 32) + a * 32, 16);
#pragma unroll
            for (int b = 0; b < TB; ++b) fb1[b] = frag(Bt, RSB, wn * (TB * 32) + b * 32, 16);
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int b = 0; b < TB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[a], fb0[b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int b = 0; b < TB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[a], fb1[b], acc[a][b], 0, 0, 0);
        };
        if (m_begin < m_end) {
            WG_GLOAD(m_begin);
            WG_LSTORE(0);
            __syncthreads();
            int buf = 0;
            for (int mk = m_begin; mk < m_end; mk += BKW, buf ^= 1) {
                const bool more = mk + BKW < m_end;
                if (more) WG_GLOAD(mk + BKW);
                mma(buf);
                if (more) WG_LSTORE(buf ^ 1);
                __syncthreads();
            }
        }
#undef WG_GLOAD
#undef WG_LSTORE
        if (do_bias) {                                   // threads sharing a channel chunk: rra = 0..RPPA-1
            float* red = reinterpret_cast<float*>(&smem[0][0]);
#pragma unroll
            for (int k = 0; k < 8; ++k) red[tid * 8 + k] = bs[k];
            __syncthreads();
            if (tid < CPRA) {
                for (int k = 0; k < 8; ++k) {
                    float a = 0.f;
                    for (int r = 0; r < RPPA; ++r) a += red[(r * CPRA + tid) * 8 + k];
                    const int co = co0 + tid * 8 + k;
                    if (co < p.Cout && a != 0.f) atomicAdd(p.dbias + co, a);
                }
            }
            __syncthreads();
        }
    } else {
        // f32: LDS image [16 rows][WG_LD floats]; thread stages rows kr, kr+8, 16-byte chunk ch
        const int ch = tid & 31, kr = tid >> 5;
        const int cy = co0 + ch * 4, cx = ci0 + ch * 4;
        const bool cyv = cy < p.Cy, cxv = cx < p.C;
        u32x4 a0, a1, b0, b1;
        auto gload = [&](int mk) __attribute__((always_inline)) {
            a0 = __builtin_amdgcn_raw_buffer_load_b128(rdy, yoff(mk + kr, cy * 4, cyv), 0, 0);
            a1 = __builtin_amdgcn_raw_buffer_load_b128(rdy, yoff(mk + kr + 8, cy * 4, cyv), 0, 0);
            b0 = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff(mk + kr, cx * 4, cxv), 0, 0);
            b1 = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff(mk + kr + 8, cx * 4, cxv), 0, 0);
        };
        const bool do_bias = p.dbias != nullptr && dt == 0 && dy_ == 0 && dx == 0 && tci == 0;
        float bs[4] = {0.f, 0.f, 0.f, 0.f};
        auto lstore = [&](int buf) __attribute__((always_inline)) {
            if (do_bias) {
                bs[0] += __uint_as_float(a0.x) + __uint_as_float(a1.x); bs[1] += __uint_as_float(a0.y) + __uint_as_float(a1.y);
                bs[2] += __uint_as_float(a0.z) + __uint_as_float(a1.z); bs[3] += __uint_as_float(a0.w) + __uint_as_float(a1.w);
            }
            *reinterpret_cast<u32x4*>(&smem[buf][(kr * WG_LD + ch * 4) * 4]) = a0;
            *reinterpret_cast<u32x4*>(&smem[buf][((kr + 8) * WG_LD + ch * 4) * 4]) = a1;
            *reinterpret_cast<u32x4*>(&smem[buf][TA_BYTES + (kr * WG_LD + ch * 4) * 4]) = p.relu_in ? relu16_f32(b0) : b0;
            *reinterpret_cast<u32x4*>(&smem[buf][TA_BYTES + ((kr + 8) * WG_LD + ch * 4) * 4]) = p.relu_in ? relu16_f32(b1) : b1;
        };
        auto mma = [&](int buf) {
            const float* As = reinterpret_cast<const float*>(&smem[buf][0]) + wm * 64 + (lane & 31);
            const float* Bs = reinterpret_cast<const float*>(&smem[buf][TA_BYTES]) + wn * 64 + (lane & 31);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int k = kk * 2 + (lane >> 5);
                const float a0 = As[k * WG_LD], a1 = As[k * WG_LD + 32];
                const float b0 = Bs[k * WG_LD], b1 = Bs[k * WG_LD + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        };
        if (m_begin < m_end) {
            gload(m_begin);
            lstore(0);
            __syncthreads();
            int buf = 0;
            for (int mk = m_begin; mk < m_end; mk += BKW, buf ^= 1) {
                const bool more = mk + BKW < m_end;
                if (more) gload(mk + BKW);
                mma(buf);
                if (more) lstore(buf ^ 1);
                __syncthreads();
            }
        }
        if (do_bias) {                                   // threads sharing a channel chunk: kr = 0..7
            float* red = reinterpret_cast<float*>(&smem[0][0]);
#pragma unroll
            for (int k = 0; k < 4; ++k) red[tid * 4 + k] = bs[k];
            __syncthreads();
            if (tid < 32) {
                for (int k = 0; k < 4; ++k) {
                    float a = 0.f;
                    for (int r = 0; r < 8; ++r) a += red[(r * 32 + tid) * 4 + k];
                    const int co = co0 + tid * 4 + k;
                    if (co < p.Cout && a != 0.f) atomicAdd(p.dbias + co, a);
                }
            }
            __syncthreads();
        }
    }

    // epilogue: each 32x32 accumulator tile goes through a per-wave LDS block and is then added to
    // dw with a rolled loop (an unrolled 128-atomic epilogue costs ~170 VGPRs of addresses)
    float* ep = reinterpret_cast<float*>(&smem[0][0]) + wave * (32 * 32);
#pragma unroll
    for (int ta = 0; ta < TA; ++ta)
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ep[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[ta][tb][r];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            const int ci = ci0 + wn * (TB * 32) + tb * 32 + (lane & 31);
            const int cob = co0 + wm * (TA * 32) + ta * 32 + (lane >> 5);
            float* dst = p.dw + ci * p.s_ci + tap * p.s_tap;
            if (p.ws) {
                // partial tile of this row slice, tile-local [row][col] layout, coalesced plain stores
                float* wt = p.ws + ((size_t)bz * gridDim.x + bx) * (size_t)(BMc * BNc) +
                            (size_t)(wm * (TA * 32) + ta * 32) * BNc + wn * (TB * 32) + tb * 32;
#pragma unroll 1
                for (int j = 0; j < 16; ++j)
                    wt[(size_t)(2 * j + (lane >> 5)) * BNc + (lane & 31)] = ep[(2 * j + (lane >> 5)) * 32 + (lane & 31)];
            } else if (ci < p.Cin_real) {
#pragma unroll 1
                for (int j = 0; j < 16; ++j) {
                    const int co = cob + 2 * j;
                    const float v = ep[(2 * j + (lane >> 5)) * 32 + (lane & 31)];
                    if (co < p.Cout && v != 0.f) atomicAdd(dst + co * p.s_co, v);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
}

// ============================================================================ backward-weight, one filter ROW per workgroup
// Same contraction as conv_wgrad_kernel, regrouped so the staged operands are reused across taps: a workgroup owns
// (filter row iy, 64*WM out-channels, 64 in-channels) and ALL KW taps of that row.  Per 32-pixel step it stages
// the dy rows once and the input FOOTPRINT of the step -- the 32 pixels plus KW-1 halo columns (two 16-pixel lines
// when W = 16) -- once; the B fragment of tap ix is the same footprint read ix rows further on.  Bytes staged per
// MFMA drop from 384 (256 x 128 tile, one tap) to ~130, and dy / x are fetched from L2 KH instead of KH*KW times.
//   waves: WM (64 out-channels each) x 2 (32 in-channels each); wave tile 64 x (KW taps x 32) = 2 x KW accumulators
//   LDS:   dy [32 rows][64*WM ch] (+64 B pad, transposing reads as in conv_wgrad_kernel);
//          x  [2 halves of 32 ch][48 rows][64 B]: the 4 consecutive rows a ds_read_b64_tr_b16 pass touches are
//             256 contiguous bytes (conflict-free without a swizzle) and a tap is a +64-byte immediate offset.
// UP2: the convolution reads a nearest-x2 upsampled input (GResBlock.py:57-58): the footprint is kept in INPUT
// coordinates (half the columns), tap ix of step pixel pk reads row ((pk + ix - pad) >> 1) + 1 of it.
// 3 x 3 filters stay near 0.9 PF/s: with three taps per row the LDS is the limit, not the staging traffic -- per stage of the
// 256-channel tile 640 cycles of fragment reads + 624 of ds_write_b128 (13 cycles each) against 1536 MFMA cycles, 82 % busy
// (5 taps: 59 %).  A nine-tap variant (128 x 64 channel tile, dy staged once for all three filter rows, 146 staged bytes per
// MFMA instead of 212) was built and passed the tests: +3 % on 786 k x 256 -> 256, -2 % on 3.1 M x 128 -> 128 (its 10 fragment
// reads per 9 MFMAs keep the LDS as busy); removed.
// NH: 32-channel blocks of input channels per workgroup (2 = 64 channels; 4 = 128, used with WM = 2 so that the 128-channel
// output tile also runs as ONE 8-wave workgroup per CU and can stagger its two halves, see the main loop).
template <int WM, int KW, bool RELU, bool UP2 = false, int NH = 2>
__global__ __launch_bounds__(WM * NH * 64) void conv_wgrad_row_kernel(WgK p) {
    constexpr int NWAVE = WM * NH, NTt = NWAVE * 64, BMc = WM * 64, BNc = NH * 32;
    constexpr int RSA = BMc * 2 + 64;
    constexpr int TA_BYTES = 32 * RSA;
    constexpr int XROWS = KW == 5 ? 64 : 48, XHALF = XROWS * 64, TB_BYTES = NH * XHALF;   // W = 8, 5 taps: 4 lines x 12 rows; W = 4: 8 lines x 8 (6) rows
    constexpr int NS = 2;      // 32-pixel sub-steps per barrier (3x3: 0.72 -> 0.80 PF/s, 5x5: +3 %; three sub-steps cost occupancy)
    constexpr int SUB = TA_BYTES + TB_BYTES, STAGE = NS * SUB;
    constexpr int EPIB = NWAVE * 32 * 32 * 4;
    constexpr int REDB = NTt * 8 * 4;                            // bias partial sums
    constexpr int LDSB = 2 * STAGE > EPIB ? (2 * STAGE > REDB ? 2 * STAGE : REDB) : (EPIB > REDB ? EPIB : REDB);
    __shared__ __attribute__((aligned(16))) char smem[LDSB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NH, wn = wave % NH;
    int bx = blockIdx.x, bz = blockIdx.z;
    if (p.xcd_remap) {
        const int gx = gridDim.x, total = gx * gridDim.z, lin = bx + gx * bz;
        const int xcd = lin & 7, qd = total >> 3, rr = total & 7;
        const int lp = (xcd < rr ? xcd * (qd + 1) : rr * (qd + 1) + (xcd - rr) * qd) + (lin >> 3);
        bz = lp / gx; bx = lp - bz * gx;
    }
    const int tiles = p.tiles_co * p.tiles_ci;
    const int irow = bx / tiles;                                 // filter row index: it * kh + iy
    const int rem = bx - irow * tiles;
    const int tco = rem / p.tiles_ci, tci = rem - tco * p.tiles_ci;
    const int co0 = tco * BMc, ci0 = tci * BNc;
    constexpr int pad = KW >> 1;
    const int it = irow / KW, iy = irow - it * KW;               // kh == kw
    const int dyl = iy - pad, dtl = it - (p.kt >> 1);
    const int m_begin = bz * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);

    f32x16 acc[2][KW];
    {
        const f32x16 zacc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int t = 0; t < KW; ++t) acc[a][t] = zacc;
    }
    const int xbase_row = UP2 ? (m_begin >> (p.logW + p.logH)) * p.Hin * p.Win : max(0, m_begin - p.maxshift);
    const size_t xbase_b = (size_t)xbase_row * p.ldx * 2, ybase_b = (size_t)m_begin * p.ldy * 2;
    const size_t xleft = p.x_bytes - xbase_b, yleft = p.dy_bytes > ybase_b ? p.dy_bytes - ybase_b : 0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + xbase_b), 0, xleft > 0xfffffffeull ? 0xfffffffeu : (unsigned)xleft, 0x00020000);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.dy + ybase_b), 0, yleft > 0xfffffffeull ? 0xfffffffeu : (unsigned)yleft, 0x00020000);
    // step geometry: 32 pixels = one 32-pixel segment of a line (W >= 32), two 16-pixel lines or four 8-pixel lines
    constexpr int cpad = (pad + 1) >> 1;
    const int segw = min(p.W, 32), logsegw = min(p.logW, 5);
    const int fpr = UP2 ? (segw >> 1) + 2 : segw + KW - 1, frows = (32 >> logsegw) * fpr;
    // dy loader: NPA 16-byte chunks per thread
    constexpr int CPRA = BMc / 8, RPPA = NTt / CPRA, NPA = 32 / RPPA;
    const int rra = tid / CPRA, cka = tid % CPRA;
    const int cy = co0 + cka * 8;
    const bool cyv = cy < p.Cy;
    // x footprint loader: XROWS * CPX 16-byte chunks over NTt threads
    constexpr int CPX = NH * 4;
    constexpr int NXL = (XROWS * CPX + NTt - 1) / NTt;
    int xseg[NXL], xj[NXL], xdst[NXL];
    unsigned xcb[NXL];
    bool xcv[NXL];
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
        const int q = tid + i * NTt, row = q / CPX, c8 = q % CPX;
        xseg[i] = row / fpr;
        xj[i] = row - xseg[i] * fpr;
        const int cx = ci0 + c8 * 8;
        xcv[i] = row < frows && cx < p.C;
        xcb[i] = (unsigned)cx * 2;
        xdst[i] = q < XROWS * CPX ? (c8 >> 2) * XHALF + row * 64 + (c8 & 3) * 16 : -1;
    }
    u32x4 ra[NS][NPA], rb[NS][NXL];
    auto gload1 = [&](int mk, u32x4 (&ra)[NPA], u32x4 (&rb)[NXL]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int m = mk + rra + i * RPPA;
            const unsigned off = (cyv && m < m_end) ? (unsigned)(m - m_begin) * ((unsigned)p.ldy * 2) + cy * 2 : 0xffffffffu;
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rdy, off, 0, 0);
        }
        const int x0 = mk & (p.W - 1), y = (mk >> p.logW) & (p.H - 1);
        int frow0 = mk - (y << p.logW) - x0;                       // first row of the frame
        bool tok = mk < m_end;
        if (p.kt > 1) {                                            // 3-D: the footprint comes from frame t + dt
            const int tt = (mk >> (p.logW + p.logH)) % p.T + dtl;
            tok = tok && (unsigned)tt < (unsigned)p.T;
            frow0 += dtl << (p.logW + p.logH);
        }
#pragma unroll
        for (int i = 0; i < NXL; ++i) {
            // output line the tap row looks at.  A 32-pixel step of 4 x 4 frames spans TWO frames (8 lines): line y + xseg of the
            // step is line (.. & (H-1)) of frame (.. >> logH) counted from the step's first frame
            const int lf = y + xseg[i], fo = lf >> p.logH;
            const int yy = (lf & (p.H - 1)) + dyl;
            bool ok = xcv[i] && tok && (unsigned)yy < (unsigned)p.H;
            int row;
            if (UP2) {
                const int xin = (x0 >> 1) - cpad + xj[i];
                ok = ok && (unsigned)xin < (unsigned)p.Win;
                row = ((mk >> (p.logW + p.logH)) + fo) * (p.Hin * p.Win) + (yy >> 1) * p.Win + xin;
            } else {
                const int xx = x0 + xj[i] - pad;
                ok = ok && (unsigned)xx < (unsigned)p.W;
                row = frow0 + (fo << (p.logW + p.logH)) + (yy << p.logW) + xx;
            }
            const unsigned off = ok ? (unsigned)(row - xbase_row) * ((unsigned)p.ldx * 2) + xcb[i] : 0xffffffffu;
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
        }
    };
    auto gload = [&](int mk) __attribute__((always_inline)) {
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) gload1(mk + 32 * s_, ra[s_], rb[s_]);
    };
    // Bias gradient = column sums of dy.  The dy tile is tap-independent, so ANY workgroup of a (tco, slice) can sum any of its
    // 32-pixel sub-steps.  With the slice workspace the KW * tiles_ci workgroups of the centre time slab take the sub-steps in turn
    // (sub-step n belongs to share n % nshare) and leave their partial sums in the workspace for wgrad_row_bias_reduce_kernel:
    // when only the centre-row, tci == 0 workgroups summed -- every sub-step, ~16 VALU per 16-byte chunk beside their MFMAs -- they
    // were the stragglers of the launch (+6...11 % on the large shapes), and their atomics formed one chain per channel.
    const bool spread = p.ws != nullptr;
    const bool do_bias = p.dbias != nullptr && dtl == 0 && (spread || (dyl == 0 && tci == 0));
    const int nshare = spread ? KW * p.tiles_ci : 1, myshare = spread ? iy * p.tiles_ci + tci : 0;
    int bphase = 0;                                              // share of the next sub-step to be stored
    float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto lstore1 = [&](char* st, const u32x4 (&ra)[NPA], const u32x4 (&rb)[NXL]) __attribute__((always_inline)) {
        const bool mine = do_bias && bphase == myshare;          // wave-uniform
        if (++bphase == nshare) bphase = 0;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            *reinterpret_cast<u32x4*>(st + (rra + i * RPPA) * RSA + cka * 16) = ra[i];
            if (mine) {
                bs[0] += __uint_as_float(ra[i].x << 16); bs[1] += __uint_as_float(ra[i].x & 0xffff0000u);
                bs[2] += __uint_as_float(ra[i].y << 16); bs[3] += __uint_as_float(ra[i].y & 0xffff0000u);
                bs[4] += __uint_as_float(ra[i].z << 16); bs[5] += __uint_as_float(ra[i].z & 0xffff0000u);
                bs[6] += __uint_as_float(ra[i].w << 16); bs[7] += __uint_as_float(ra[i].w & 0xffff0000u);
            }
        }
#pragma unroll
        for (int i = 0; i < NXL; ++i)
            if (xdst[i] >= 0) *reinterpret_cast<u32x4*>(st + TA_BYTES + xdst[i]) = RELU ? relu16_bf16(rb[i]) : rb[i];
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) lstore1(&smem[buf * STAGE + s_ * SUB], ra[s_], rb[s_]);
    };
    // fragment addressing (ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-channel block)
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int g16 = lane >> 4, i16 = lane & 15;
    const int frow = (g16 >> 1) * 8 + (i16 >> 2), fcol2 = ((g16 & 1) * 16 + (i16 & 3) * 4) * 2;
    auto tr2 = [&](const char* lo_, const char* hi_) __attribute__((always_inline)) -> bf16x8 {
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lo_);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)hi_);
        s16x8 f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, f);
    };
    // footprint row of step pixel pk (before the tap offset): line s of a multi-line step starts at row s * fpr
    constexpr int NT_ = UP2 ? KW : 1;                            // x2 fold: the row depends on the tap's parity
    int xoffs[2][2][NT_];                                        // [k half][lo / hi 4-row block][tap]
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
#pragma unroll
            for (int t = 0; t < NT_; ++t) {
                const int pk = kb * 16 + frow + hl * 4;
                const int px_ = pk & (segw - 1);
                const int fr = (pk >> logsegw) * fpr + (UP2 ? ((px_ + t - pad) >> 1) + cpad : px_);
                xoffs[kb][hl][t] = TA_BYTES + wn * XHALF + fr * 64 + fcol2;
            }
    const int aoff = frow * RSA + fcol2 + (wm * 64) * 2;
    auto mma1 = [&](const char* st) __attribute__((always_inline)) {
        constexpr int NU = 2 * KW;                               // units: (k half, tap), 2 MFMAs each
        bf16x8 fa[2][2], fb[NU];
        auto ldA = [&](int kb) __attribute__((always_inline)) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const char* pz = st + aoff + kb * 16 * RSA + a * 64;
                fa[kb][a] = tr2(pz, pz + 4 * RSA);
            }
        };
        auto ldB = [&](int u) __attribute__((always_inline)) {
            const int kb = u / KW, t = u % KW;
            if constexpr (UP2) fb[u] = tr2(st + xoffs[kb][0][t], st + xoffs[kb][1][t]);
            else fb[u] = tr2(st + xoffs[kb][0][0] + t * 64, st + xoffs[kb][1][0] + t * 64);
        };
        ldA(0); ldB(0); ldB(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (u + 2 < NU) ldB(u + 2);
            if (u == KW - 2) ldA(1);
            const int kb = u / KW, t = u % KW;
            acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb][0], fb[u], acc[0][t], 0, 0, 0);
            acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb][1], fb[u], acc[1][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto mma = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) mma1(&smem[buf * STAGE + s_ * SUB]);
    };
    if (m_begin < m_end) {
        // One 8-wave workgroup per CU: waves w and w+4 share a SIMD and would run their load-issue / MFMA / LDS-store phases in
        // lockstep, leaving the matrix pipe idle while both issue memory operations.  The upper half of the waves therefore
        // works one stage further ahead in registers and does its LDS stores and global loads BETWEEN the two sub-steps, while
        // the lower half does them at the stage boundaries: 1.16 -> 1.29 PF/s (5 x 5), 0.86 -> 0.92 (3 x 3, 256-channel tile).
        // Measured alternatives: loads / stores compiled out 1.41 / 1.58 PF/s; both halves one stage ahead (stores at the top
        // of the stage) -1.3 %; LDS-DMA staging in a 3-stage ring (no registers, no ds_write; swizzled unpadded rows) 1.13 PF/s,
        // 1.22 staggered: a `buffer_load ... lds` costs more issue time beside MFMAs than a register load plus its ds_write.
        const bool late = NWAVE == 8 && NS == 2 && wave >= 4;
        gload(m_begin);
        lstore(0);
        if (late) gload(m_begin + 32 * NS);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        int buf = 0;
        for (int mk = m_begin; mk < m_end; mk += 32 * NS, buf ^= 1) {
            if (!late) {
                gload(mk + 32 * NS);              // past the slice: out-of-range offsets -> zeros
                mma(buf);
                lstore(buf ^ 1);
            } else {
                mma1(&smem[buf * STAGE]);
                lstore(buf ^ 1);
                gload(mk + 2 * 32 * NS);
                mma1(&smem[buf * STAGE + (NS - 1) * SUB]);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_waitcnt(0x0070);      // vmcnt(0): the upper half still has a prefetch in flight
    }
    if (do_bias) {
        float* red = reinterpret_cast<float*>(&smem[0]);
#pragma unroll
        for (int k = 0; k < 8; ++k) red[tid * 8 + k] = bs[k];
        __syncthreads();
        if (tid < CPRA) {
            for (int k = 0; k < 8; ++k) {
                float a = 0.f;
                for (int r = 0; r < RPPA; ++r) a += red[(r * CPRA + tid) * 8 + k];
                const int co = co0 + tid * 8 + k;
                if (spread) p.wsb[((size_t)bz * gridDim.x + bx) * BMc + tid * 8 + k] = a;      // partial of (slice, workgroup)
                else if (co < p.Cout && a != 0.f) atomicAdd(p.dbias + co, a);
            }
        }
        __syncthreads();
    }
    // epilogue: 32 x 32 accumulator tiles through a per-wave LDS block
    float* ep = reinterpret_cast<float*>(&smem[0]) + wave * (32 * 32);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < KW; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ep[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[a][t][r];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            const int ci = ci0 + wn * 32 + (lane & 31);
            const int cob = co0 + wm * 64 + a * 32 + (lane >> 5);
            const int tap = irow * KW + t;
            if (p.ws) {
                // partial tile of this row slice: [slice][x-block][tap of the row][BMc][BNc], plain coalesced stores
                float* wt = p.ws + (((size_t)bz * gridDim.x + bx) * KW + t) * (size_t)(BMc * BNc) +
                            (size_t)(wm * 64 + a * 32) * BNc + wn * 32;
#pragma unroll 1
                for (int j = 0; j < 16; ++j)
                    wt[(size_t)(2 * j + (lane >> 5)) * BNc + (lane & 31)] = ep[(2 * j + (lane >> 5)) * 32 + (lane & 31)];
            } else if (ci < p.Cin_real) {
                float* dst = p.dw + ci * p.s_ci + tap * p.s_tap;
#pragma unroll 1
                for (int j = 0; j < 16; ++j) {
                    const int co = cob + 2 * j;
                    const float v = ep[(2 * j + (lane >> 5)) * 32 + (lane & 31)];
                    if (co < p.Cout && v != 0.f) atomicAdd(dst + co * p.s_co, v);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
}

// reduction of the row kernel's partial tiles: dw[co][ci][iy*KW + t] += sum over slices
struct WgRowRedK { const float* ws; float* dw; int nslice, gx, tiles_co, tiles_ci, BMc, BNc, KW, Cout, Cin_real; long long s_co, s_ci, s_tap; int overwrite; };
// sum over the slices of four consecutive partial-tile elements (16-byte loads, four slices requested before the first addition;
// slice order kept): the reduce kernels stream the whole workspace once and were running at 2.5 TB/s with scalar loads
__device__ __forceinline__ f32x4 slice_sum4(const float* ws, int nslice, size_t stride, size_t off) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 4 <= nslice; z += 4) {
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4*>(ws + (size_t)(z + j) * stride + off);
#pragma unroll
        for (int j = 0; j < 4; ++j) a += v[j];
    }
    for (; z < nslice; ++z) a += *reinterpret_cast<const f32x4*>(ws + (size_t)z * stride + off);
    return a;
}

__global__ void wgrad_row_reduce_kernel(WgRowRedK p) {
    const int tile_elems = p.KW * p.BMc * p.BNc;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;      // 4 consecutive in-channels per thread
    if (i >= (long long)p.gx * tile_elems) return;
    const int bx = (int)(i / tile_elems), e = (int)(i - (long long)bx * tile_elems);
    const int t = e / (p.BMc * p.BNc), e2 = e - t * (p.BMc * p.BNc);
    const int r = e2 / p.BNc, c = e2 - r * p.BNc;
    const int tiles = p.tiles_co * p.tiles_ci;
    const int iy = bx / tiles, rem = bx - iy * tiles;
    const int tco = rem / p.tiles_ci, tci = rem - tco * p.tiles_ci;
    const int co = tco * p.BMc + r, ci = tci * p.BNc + c;
    if (co >= p.Cout || ci >= p.Cin_real) return;
    const f32x4 a = slice_sum4(p.ws, p.nslice, (size_t)p.gx * tile_elems, (size_t)bx * tile_elems + e);
    float* d = p.dw + co * p.s_co + ci * p.s_ci + (iy * p.KW + t) * p.s_tap;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (ci + j < p.Cin_real) d[j * p.s_ci] = p.overwrite ? a[j] : d[j * p.s_ci] + a[j];
}

// bias gradient of the row kernel's workspace path: dbias[co] += sum over (slice, filter row of the centre time slab, ci tile) of the
// partial column sums, in a fixed order (block = 32 channels x 8 partial lanes)
struct WgBiasRedK { const float* wsb; float* dbias; int nslice, gx, tiles_co, tiles_ci, KW, kt, BMc, Cout; };
__global__ __launch_bounds__(1024) void wgrad_row_bias_reduce_kernel(WgBiasRedK p) {
    // block = 32 channels x 32 partial lanes; a lane walks its items eight loads at a time (a few thousand partials per channel on
    // the large shapes: as a serial chain on 8 lanes this kernel took ~250 us)
    __shared__ float red[32][33];
    const int gpt = p.BMc / 32;
    const int tco = blockIdx.x / gpt, cg = blockIdx.x - tco * gpt;
    const int c = threadIdx.x & 31, l = threadIdx.x >> 5;
    const int tiles = p.tiles_co * p.tiles_ci, per = p.KW * p.tiles_ci, nitem = p.nslice * per;
    const int irow0 = (p.kt >> 1) * p.KW;
    auto item = [&](int i) __attribute__((always_inline)) -> float {
        if (i >= nitem) return 0.f;
        const int bz = i / per, r = i - bz * per;
        const int iy = r / p.tiles_ci, tci = r - iy * p.tiles_ci;
        const int bx = (irow0 + iy) * tiles + tco * p.tiles_ci + tci;
        return p.wsb[((size_t)bz * p.gx + bx) * p.BMc + cg * 32 + c];
    };
    float a = 0.f;
    for (int i = l; i < nitem; i += 32 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = item(i + 32 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    red[l][c] = a;
    __syncthreads();
    if (l == 0) {
        for (int j = 1; j < 32; ++j) a += red[j][c];
        const int co = tco * p.BMc + cg * 32 + c;
        if (co < p.Cout) p.dbias[co] += a;
    }
}

// Second phase of the workspace path: dw[co][ci][tap] += sum over row slices of the partial tiles.
struct WgRedK { const float* ws; float* dw; int nslice, gx, gx_per_tap, tiles_ci, BMc, BNc, Cout, Cin_real; long long s_co, s_ci, s_tap; int overwrite; };
__global__ void wgrad_reduce_kernel(WgRedK p) {
    const int tile_elems = p.BMc * p.BNc;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= (long long)p.gx * tile_elems) return;
    const int bx = (int)(i / tile_elems), e = (int)(i - (long long)bx * tile_elems);
    const int r = e / p.BNc, c = e - r * p.BNc;
    // decode bx exactly like the kernel: tap = bx / (tiles_co*tiles_ci), rem -> (tco, tci)
    const int per_tap = p.gx_per_tap;
    const int tap = bx / per_tap, rem = bx - tap * per_tap;
    const int tco = rem / p.tiles_ci, tci = rem - tco * p.tiles_ci;
    const int co = tco * p.BMc + r, ci = tci * p.BNc + c;
    if (co >= p.Cout || ci >= p.Cin_real) return;
    const f32x4 a = slice_sum4(p.ws, p.nslice, (size_t)p.gx * tile_elems, (size_t)bx * tile_elems + e);
    float* d = p.dw + co * p.s_co + ci * p.s_ci + tap * p.s_tap;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (ci + j < p.Cin_real) d[j * p.s_ci] = p.overwrite ? a[j] : d[j * p.s_ci] + a[j];
}

// ============================================================================ weight packing
struct PackK {
    const float* w; const float* sigma; char* wf; char* wd;
    int Cout, Cin, ntaps, Cip, co_off, co_tot_f, co_tot_d, kt, kh, kw, ci_off, ci_tot;
};
// one thread per (co, ci_pad, tap) of the forward pack; writes both packs
template <typename T>
__global__ void pack_weight_kernel(PackK p) {
    const long long n = (long long)p.Cout * p.Cip * p.ntaps;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int tap = (int)(i % p.ntaps);
    const long long r = i / p.ntaps;
    const int ci = (int)(r % p.Cip), co = (int)(r / p.Cip);
    float v = 0.f;
    if (ci < p.Cin) {
        v = p.w[((size_t)co * p.ci_tot + p.ci_off + ci) * p.ntaps + tap];
        if (p.sigma) v = v / *p.sigma;
    }
    if (p.wf) stf(reinterpret_cast<T*>(p.wf) + ((size_t)tap * p.co_tot_f + p.co_off + co) * p.Cip + ci, v);
    if (p.wd) {
        const int ftap = p.ntaps - 1 - tap;     // flipping every axis == reversing the flat tap index
        stf(reinterpret_cast<T*>(p.wd) + ((size_t)ftap * p.Cip + ci) * p.co_tot_d + p.co_off + co, v);
    }
}

}  // namespace

// ============================================================================ optional profiling
// bench.py needs the average duration of the dominant kernel measured with HIP events on the
// launch stream.  When enabled, every conv launch is bracketed by an event pair; dvd_prof_report
// synchronises and sums them.  Off by default (no events, no global state touched).
#include <vector>
#include <mutex>
#include <cstdio>
#include <cstdlib>
namespace {
struct ProfRec { hipEvent_t a, b; double flops; int kind; long long M; int C, Cout, taps, split, flags, variant; };
bool g_prof = false;
std::vector<ProfRec> g_recs;
std::mutex g_prof_mu;
struct ProfScope {
    ProfRec r; bool on; hipStream_t s;
    ProfScope(int kind, double flops, void* stream, long long M, int C, int Cout, int taps, int split, int flags)
        : on(g_prof), s((hipStream_t)stream) {
        if (!on) return;
        r.kind = kind; r.flops = flops; r.M = M; r.C = C; r.Cout = Cout; r.taps = taps; r.split = split; r.flags = flags;
        r.variant = 0;
        hipEventCreate(&r.a); hipEventCreate(&r.b);
        hipEventRecord(r.a, s);
    }
    ~ProfScope() {
        if (!on) return;
        hipEventRecord(r.b, s);
        std::lock_guard<std::mutex> l(g_prof_mu);
        g_recs.push_back(r);
    }
};
}  // namespace
extern "C" void dvd_prof_enable(int on) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    g_prof = on != 0;
}
// kind 0 = conv_igemm (forward / backward-data), 1 = conv_wgrad.  Drains the records of `kind` and returns the number of
// launches; n / ms / flops (each [nvar] or NULL) receive the per-variant totals -- kind 0: 1 = conv_halo 256 x 128,
// 2 = conv_halo 128 x 128, 3 = conv_halo 256 x 64 (thin outputs), 4 = conv_igemm 128 x 128, 5 = conv_igemm 256 x 128,
// 6 = conv_igemm 256 x 256 (8 waves), 7 / 8 = whole-frame footprint kernel (4 x 4 / 8 x 8 frames) 256 x 128 / 128 x 128, 9 = thin-input kernel (conv_thin.hip); kind 1: 1 = filter-row kernel, 2 = one-tap kernel, 3 = thin-end kernel (wgrad_thin.hip); index 0 = everything.
// If the environment variable DVD_PROF_CSV is set, every drained record is appended to that file.
extern "C" long long dvd_prof_report_variants(int kind, int nvar, long long* n, double* ms, double* flops) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    for (int v = 0; v < nvar; ++v) { if (n) n[v] = 0; if (ms) ms[v] = 0; if (flops) flops[v] = 0; }
    long long total = 0;
    std::vector<ProfRec> keep;
    const char* csv = getenv("DVD_PROF_CSV");
    FILE* f = csv ? fopen(csv, "a") : nullptr;
    for (auto& r : g_recs) {
        if (r.kind != kind) { keep.push_back(r); continue; }
        hipEventSynchronize(r.b);
        float t = 0; hipEventElapsedTime(&t, r.a, r.b);
        if (f) fprintf(f, "%d,%lld,%d,%d,%d,%d,%d,%.4f,%.0f,%d\n", r.kind, r.M, r.C, r.Cout, r.taps, r.split, r.flags, t, r.flops, r.variant);
        const int slots[2] = {0, r.variant};                    // slot 0 = all launches, plus the record's own slot
        for (int j = 0; j < (r.variant > 0 ? 2 : 1); ++j) {
            const int v = slots[j];
            if (v >= nvar) continue;
            if (n) ++n[v];
            if (ms) ms[v] += t;
            if (flops) flops[v] += r.flops;
        }
        ++total;
        hipEventDestroy(r.a); hipEventDestroy(r.b);
    }
    if (f) fclose(f);
    g_recs.swap(keep);
    return total;
}
extern "C" long long dvd_prof_report(int kind, double* total_ms, double* total_flops) {
    long long n = 0;
    return dvd_prof_report_variants(kind, 1, &n, total_ms, total_flops);
}

// ============================================================================ C ABI
extern "C" int dvd_conv_forward(const dvd_conv_desc* d, void* stream) { return dvd_conv_forward_gru(d, nullptr, stream); }

// Validates a forward / backward-data request and derives the kernel parameters and the variant that serves it.
struct ConvPlan { long long M; bool halo, thin, wide, big, smallf; };
static int conv_plan(const dvd_conv_desc* d, const GruEpi* g, ConvK& p, ConvPlan& pl) {
    if (!d || !d->in || !d->w || (!d->ws && (!d->out || (d->nsplit > 1 && !g)))) return DVD_E_ARG;
    if (g && (d->ws || (g->h & 7) || (d->nsplit > 1 && (!g->slabs || !g->tickets)))) return DVD_E_ARG;
    if (d->frames <= 0 || d->T <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->Cout <= 0) return DVD_E_ARG;
    int logH = ilog2_exact(d->H), logW = ilog2_exact(d->W);
    const bool pow2 = logH >= 0 && logW >= 0;
    if (!pow2) logH = logW = -1;           // any other extent: division-based indexing in the tap-by-tap kernels (grid_pos)
    if ((d->C & 7) || (d->ldi & 7) || !(d->kt & d->kh & d->kw & 1)) return DVD_E_SHAPE;
    if (d->up2 && ((d->H | d->W) & 1)) return DVD_E_SHAPE;
    if (d->kt > 1 && d->up2) return DVD_E_SHAPE;
    const long long M = (long long)d->frames * d->T * d->H * d->W;
    if (M >= (1ll << 31) - BM) return DVD_E_SHAPE;
    p.in = (const char*)d->in; p.w = (const char*)d->w; p.bias = d->bias; p.res = (const char*)d->res;
    p.mask = (const char*)d->mask; p.out = (char*)d->out; p.ws = d->ws; p.wq = (const char*)d->wq;
    p.nb32 = (d->Cout + 127) / 128 * 4;
    p.M = (int)M; p.C = d->C; p.ldi = d->ldi; p.Cout = d->Cout; p.ldo = d->ldo; p.ldres = d->ldres; p.ldmask = d->ldmask;
    p.res_up2 = d->res && d->res_up2;
    if (p.res_up2 && (d->H < 2 || d->W < 2 || ((d->H | d->W) & 1))) return DVD_E_SHAPE;
    p.T = d->T; p.H = d->H; p.W = d->W; p.logH = logH; p.logW = logW;
    p.Hin = d->up2 ? d->H / 2 : d->H; p.Win = d->up2 ? d->W / 2 : d->W;
    p.kt = d->kt; p.kh = d->kh; p.kw = d->kw;
    const int bk = d->dtype == DVD_BF16 ? 32 : 16;
    p.kchunks = (d->C + bk - 1) / bk;
    p.nk = d->kt * d->kh * d->kw * p.kchunks;
    p.nsplit = d->nsplit < 1 ? 1 : d->nsplit;
    if (p.nsplit > p.nk) p.nsplit = p.nk;
    if (p.nsplit != (d->nsplit < 1 ? 1 : d->nsplit)) return DVD_E_ARG;   // caller sized ws for d->nsplit slabs
    // 8-wave 256 x 256 tile when the output is wide enough (no more than 1/8 of the last N tile wasted)
    // and there is at least one tile per CU; else 256 x 128 (>= 2 tiles per CU) or 128 x 128.
    const long long t256 = cdiv(M, 256) * (long long)cdiv(d->Cout, 256) * p.nsplit;
    const int rem256 = d->Cout % 256;
    // halo-staged kernel: square 3x3 / 5x5 (x3) filters on frames at least one 16 x 16 patch large.  It runs as
    // 4-wave workgroups only: with the activation DMAs gone, two 256 x 128 workgroups per CU beat one 8-wave
    // 256 x 256 workgroup (1.13-1.34 vs 0.86-1.12 PF/s on the S = 16 / 32 shapes of config C2).
    static const int use_halo = getenv("DVD_CONV_HALO") ? atoi(getenv("DVD_CONV_HALO")) : 1;
    const bool halo = use_halo && pow2 && d->kh == d->kw && (d->kh == 3 || d->kh == 5) && d->W >= 16 && d->H >= 16 &&
                      p.nsplit <= p.kchunks * d->kt;
    // frames of 4 x 4 / 8 x 8 pixels (bf16, 2-D taps, no upsample): whole-frame footprints in LDS, weights from L2 -- needs the
    // fragment-major image
    static const int use_small = getenv("DVD_CONV_SMALLF") ? atoi(getenv("DVD_CONV_SMALLF")) : 1;
    const bool smallf = use_small && !halo && pow2 && d->dtype == DVD_BF16 && d->wq && d->kt == 1 && d->T == 1 && !d->up2 &&
                        d->kh == d->kw && (d->kh == 3 || d->kh == 5) && d->H == d->W && (d->W == 4 || d->W == 8) &&
                        p.nsplit <= p.kchunks && M < (1ll << 24);
    const bool thin = halo && d->dtype == DVD_BF16 && d->Cout <= 64 && cdiv(M, 256) * (long long)p.nsplit >= 512;
    const bool wide = !halo && !smallf && d->dtype == DVD_BF16 && d->Cout >= 256 && (rem256 == 0 || rem256 > 224) && t256 >= 256 &&
                      !(g && p.nsplit > 1);
    p.tilesN = wide ? (d->Cout + 255) / 256 : thin ? 1 : (d->Cout + BN - 1) / BN;
    p.up2 = d->up2; p.relu_in = d->relu_in; p.act = d->act; p.out_f32 = d->out_f32;
    if (g) p.g = *g; else p.g = GruEpi{};
    p.nmajor = 0;
    {   // extents of the two buffer descriptors (32-bit byte offsets): tensors must stay below 4 GiB
        const size_t esz = d->dtype == DVD_BF16 ? 2 : 4;
        const size_t rows_in = (size_t)d->frames * d->T * p.Hin * p.Win;
        const size_t inb = ((rows_in - 1) * (size_t)d->ldi + d->C) * esz;
        const size_t wb = (size_t)d->kt * d->kh * d->kw * d->Cout * d->C * esz;
        if (wb >= 0xffffffffull) return DVD_E_SHAPE;
        p.in_bytes = inb; p.w_bytes = (unsigned)wb;
        p.maxshift = ((d->kt >> 1) * d->H + (d->kh >> 1)) * d->W + (d->kw >> 1);
        static const int nmaj = getenv("DVD_CONV_NMAJOR") ? atoi(getenv("DVD_CONV_NMAJOR")) : 1;
        p.nmajor = nmaj && !halo && wb > inb && wb > (4u << 20);      // weights beyond one L2 (also for the small-frame halo kernel)
        // halo kernel: with several N tiles and weights well beyond one L2, a run of workgroups that shares the weight tile
        // (and streams the activations once per N tile) misses less than one that shares the footprint and cycles through
        // all the weights: 1272 -> 1355 TF/s on 786 k x 256 -> 1536 (19.7 MB of weights); neutral from 5 MiB, -1.5 % at 4.9 MB
        static const long long nmaj_mb = getenv("DVD_CONV_NMAJOR_MB") ? atoll(getenv("DVD_CONV_NMAJOR_MB")) : 5;
        if (nmaj && halo && p.tilesN > 1 && wb > ((size_t)nmaj_mb << 20)) p.nmajor = 1;
        if (nmaj && halo && p.tilesN >= 6 && wb > (2u << 20)) p.nmajor = 1;      // 3.1 M x 128 -> 768 (4.9 MB, 6 N tiles): +2 %
    }
    // 256-row tiles when they still give every CU work; 128-row tiles for the small recurrent convs
    static const long long big_thr = getenv("DVD_CONV_BIGT") ? atoll(getenv("DVD_CONV_BIGT")) : 512;
    // 1 x 1 filters are HBM-bound streams with 2-8 K steps: the 256-row tap-by-tap tile holds 433 registers = ONE workgroup per CU, the
    // 128-row tile 240 = two.  3.1 M x 128 -> 128: 665 -> 505 us, 128 -> 64: 691 -> 420, 64 -> 128: 574 -> 422; step 493.4 -> 490.9 ms
    // (DVD_CONV_BIG1 = 1 restores the 256-row tile)
    static const int big1 = getenv("DVD_CONV_BIG1") ? atoi(getenv("DVD_CONV_BIG1")) : 0;
    const bool one_tap = d->kt * d->kh * d->kw == 1 && !halo && !smallf;
    const bool big = wide || (cdiv(M, 256) * (long long)p.tilesN * p.nsplit >= big_thr && !(one_tap && !big1));   // >= 2 workgroups per CU
    p.pm = 0;
    {   // pixel-major row order for the tap-by-tap kernel on small frames: the rows of a tile must share their image line
        static const int use_pm = getenv("DVD_CONV_PIXMAJOR") ? atoi(getenv("DVD_CONV_PIXMAJOR")) : 1;
        const int bmt = (wide || big) ? 256 : 128;
        const int F = d->frames;
        const bool lines = (F % bmt == 0) || (bmt % F == 0 && d->W % (bmt / F) == 0);
        const size_t esz = d->dtype == DVD_BF16 ? 2 : 4;
        if (use_pm && !halo && !smallf && pow2 && d->kt == 1 && d->T == 1 && !d->up2 && d->kh >= 3 && d->H <= 8 && lines &&
            (size_t)M * (size_t)(d->ldi > 3 * d->Cout ? d->ldi : 3 * d->Cout) * 4 < (1ull << 31))     // every epilogue offset from row 0 fits
            p.pm = F;
    }
    pl.M = M; pl.halo = halo; pl.thin = thin; pl.wide = wide; pl.big = big && !(smallf && d->W == 4); pl.smallf = smallf;
    return DVD_OK;
}

extern "C" int dvd_conv_forward_gru(const dvd_conv_desc* d, const GruEpi* g, void* stream) {
    ConvK p; ConvPlan pl;
    const int rc = conv_plan(d, g, p, pl);
    if (rc != DVD_OK) return rc;
    const long long M = pl.M;
    const bool halo = pl.halo, thin = pl.thin, wide = pl.wide, big = pl.big;
    if (d->wq_kind >= 2 && !(d->wq_kind == 2 ? dvd_conv_thin_in_ok(d) : dvd_conv_thin_out_ok(d))) return DVD_E_ARG;   // a thin image the request cannot use
    if (!g && d->wq && d->wq_kind == 2 && dvd_conv_thin_in_ok(d)) {      // the stems / the RGB layer's backward-data pass: taps folded into K (conv_thin.hip)
        ProfScope prof(0, 2.0 * (double)M * d->Cout * d->C * d->kt * d->kh * d->kw, stream, M, d->C, d->Cout, d->kt * d->kh * d->kw, 1,
                       d->relu_in << 1);
        prof.r.variant = 9;
        return dvd_conv_thin_in(d, stream);
    }
    if (!g && d->wq && d->wq_kind == 3 && dvd_conv_thin_out_ok(d)) {     // the RGB layer / the stems' backward-data pass (conv_thin.hip)
        ProfScope prof(0, 2.0 * (double)M * d->Cout * d->C * d->kt * d->kh * d->kw, stream, M, d->C, d->Cout, d->kt * d->kh * d->kw, 1,
                       d->relu_in << 1);
        prof.r.variant = 9;
        return dvd_conv_thin_out(d, stream);
    }
    dim3 grid(cdiv(M, big ? 256 : 128) * p.tilesN, 1, p.nsplit);
    if (g && p.nsplit > 1 && grid.x > DVD_GRU_TICKETS) return DVD_E_SHAPE;      // one ticket per output tile (the small-frame grid below is no larger)
    ProfScope prof(0, 2.0 * (double)M * d->Cout * d->C * d->kt * d->kh * d->kw, stream, M, d->C, d->Cout,
                   d->kt * d->kh * d->kw, p.nsplit, d->up2 | (d->relu_in << 1) | ((d->ws != nullptr) << 2));
    hipStream_t st = (hipStream_t)stream;
    prof.r.variant = halo ? (thin ? 3 : big ? 1 : 2) : pl.smallf ? (big ? 7 : 8) : (wide ? 6 : big ? 5 : 4);
    if (pl.smallf) {
        const int S_ = d->W, Gf = (big ? 256 : 128) / (S_ * S_);
        grid = dim3(cdiv(d->frames, Gf) * p.tilesN, 1, p.nsplit);
#define LAUNCH_GBS(TM_, SZ_) do { if (d->relu_in) conv_halo_gbs_kernel<TM_, SZ_, true><<<grid, 256, 0, st>>>(p);   \
                                  else conv_halo_gbs_kernel<TM_, SZ_, false><<<grid, 256, 0, st>>>(p); } while (0)
        if (S_ == 8) { if (big) LAUNCH_GBS(4, 8); else LAUNCH_GBS(2, 8); }
        else LAUNCH_GBS(2, 4);
#undef LAUNCH_GBS
        return launch_status();
    }
    if (halo) {
#define LAUNCH_HALO2(TT, TM_, WN_, RL_)                                                             \
        do { if (d->up2) conv_halo_kernel<TT, TM_, WN_, RL_, true><<<grid, 128 * WN_, 0, st>>>(p);      \
             else conv_halo_kernel<TT, TM_, WN_, RL_, false><<<grid, 128 * WN_, 0, st>>>(p); } while (0)
#define LAUNCH_HALO(TT, TM_, WN_)                                                                   \
        do { if (d->relu_in) LAUNCH_HALO2(TT, TM_, WN_, true); else LAUNCH_HALO2(TT, TM_, WN_, false); } while (0)
#define LAUNCH_GB(TM_, WN_, WMV_)                                                                       \
        do { if (d->relu_in) { if (d->up2) conv_halo_gb_kernel<TM_, WN_, WMV_, true, true><<<grid, 256, 0, st>>>(p);       \
                               else conv_halo_gb_kernel<TM_, WN_, WMV_, true, false><<<grid, 256, 0, st>>>(p); }           \
             else            { if (d->up2) conv_halo_gb_kernel<TM_, WN_, WMV_, false, true><<<grid, 256, 0, st>>>(p);      \
                               else conv_halo_gb_kernel<TM_, WN_, WMV_, false, false><<<grid, 256, 0, st>>>(p); } } while (0)
        static const int gb_mask = getenv("DVD_CONV_GB") ? atoi(getenv("DVD_CONV_GB")) : 7;     // A/B aid: bit 0 = 256 x 128, 1 = 128 x 128, 2 = thin
        if (d->dtype == DVD_BF16 && p.wq && (gb_mask & (thin ? 4 : big ? 1 : 2))) {   // weights straight from L2 (fragment-major image supplied)
            if (thin) LAUNCH_GB(2, 1, 4); else if (big) LAUNCH_GB(4, 2, 2); else LAUNCH_GB(2, 2, 2);
            return launch_status();
        }
#undef LAUNCH_GB
        if (thin) {
            if (d->relu_in) { if (d->up2) conv_halo_kernel<bf16_t, 2, 1, true, true, 4><<<grid, 256, 0, st>>>(p);
                              else conv_halo_kernel<bf16_t, 2, 1, true, false, 4><<<grid, 256, 0, st>>>(p); }
            else            { if (d->up2) conv_halo_kernel<bf16_t, 2, 1, false, true, 4><<<grid, 256, 0, st>>>(p);
                              else conv_halo_kernel<bf16_t, 2, 1, false, false, 4><<<grid, 256, 0, st>>>(p); }
        } else if (d->dtype == DVD_BF16) { if (big) LAUNCH_HALO(bf16_t, 4, 2); else LAUNCH_HALO(bf16_t, 2, 2); }
        else if (d->dtype == DVD_F32) { if (big) LAUNCH_HALO(float, 4, 2); else LAUNCH_HALO(float, 2, 2); }
        else return DVD_E_ARG;
#undef LAUNCH_HALO
#undef LAUNCH_HALO2
        return launch_status();
    }
#define LAUNCH_CONV(TT)                                                                       \
    do {                                                                                      \
        if (big) { if (d->relu_in) conv_igemm_kernel<TT, 4, 2, true><<<grid, NT, 0, st>>>(p);   \
                   else conv_igemm_kernel<TT, 4, 2, false><<<grid, NT, 0, st>>>(p); }           \
        else     { if (d->relu_in) conv_igemm_kernel<TT, 2, 2, true><<<grid, NT, 0, st>>>(p);   \
                   else conv_igemm_kernel<TT, 2, 2, false><<<grid, NT, 0, st>>>(p); }           \
    } while (0)
    if (wide) {
        if (d->relu_in) conv_igemm_kernel<bf16_t, 4, 4, true><<<grid, 512, 0, st>>>(p);
        else conv_igemm_kernel<bf16_t, 4, 4, false><<<grid, 512, 0, st>>>(p);
    } else if (d->dtype == DVD_BF16) LAUNCH_CONV(bf16_t);
    else if (d->dtype == DVD_F32) LAUNCH_CONV(float);
    else return DVD_E_ARG;
#undef LAUNCH_CONV
    return launch_status();
}

// Validates a weight-gradient request and derives tile shape, grid and row split.
// mode: 0 = one tap per workgroup (conv_wgrad_kernel), 1 = one filter row per workgroup (conv_wgrad_row_kernel; ta = WM)
static int wgrad_plan(const dvd_wgrad_desc* d, WgK& p, dim3& grid, int& ta, int& tb, long long& msplit, int& mode) {
    if (!d || !d->x || !d->dy || !d->dw) return DVD_E_ARG;
    int logH = ilog2_exact(d->H), logW = ilog2_exact(d->W);
    const bool pow2 = logH >= 0 && logW >= 0;
    if (!pow2) logH = logW = -1;           // division-based indexing, one-tap kernel only
    if (d->up2 && ((d->H | d->W) & 1)) return DVD_E_SHAPE;
    if ((d->C & 7) || (d->ldx & 7) || (d->Cy & 7) || (d->ldy & 7) || !(d->kt & d->kh & d->kw & 1)) return DVD_E_SHAPE;
    if (d->Cout > d->Cy || d->Cin_real > d->C) return DVD_E_ARG;
    if (d->dtype != DVD_BF16 && d->dtype != DVD_F32) return DVD_E_ARG;
    const long long M = (long long)d->frames * d->T * d->H * d->W;
    if (M >= (1ll << 31) - 64) return DVD_E_SHAPE;
    p.x = (const char*)d->x; p.dy = (const char*)d->dy; p.dw = d->dw;
    p.M = (int)M; p.C = d->C; p.ldx = d->ldx; p.Cin_real = d->Cin_real; p.Cout = d->Cout; p.Cy = d->Cy; p.ldy = d->ldy;
    p.T = d->T; p.H = d->H; p.W = d->W; p.logH = logH; p.logW = logW;
    p.Hin = d->up2 ? d->H / 2 : d->H; p.Win = d->up2 ? d->W / 2 : d->W;
    p.kt = d->kt; p.kh = d->kh; p.kw = d->kw; p.up2 = d->up2; p.relu_in = d->relu_in;
    // bf16: 256-wide tile along whichever channel axis is long enough (2x the MFMAs per barrier)
    ta = 2; tb = 2;
    if (d->dtype == DVD_BF16) {
        if (d->Cout >= 192) ta = 4;
        else if (d->Cin_real >= 192) tb = 4;
    }
    static const int use_row = getenv("DVD_WG_ROW") ? atoi(getenv("DVD_WG_ROW")) : 1;
    mode = (use_row && pow2 && d->dtype == DVD_BF16 && d->kh == d->kw && (d->kw == 3 || d->kw == 5) && (!d->up2 || (d->kw == 3 && d->kt == 1)) &&
            ((d->W >= 8 && (d->H * d->W) % 32 == 0) ||
             // 4 x 4 frames (round 4): a 32-pixel step = two whole frames
             (d->W == 4 && d->H == 4 && d->kt == 1 && d->T == 1 && !d->up2 && (d->frames & 1) == 0))) ? 1 : 0;
    if (mode == 1) {   // 64 channels for thin outputs, else 256 or 128, whichever pads Cout less (256 on a tie)
        const int w4 = (d->Cout + 255) / 256 * 256, w2 = (d->Cout + 127) / 128 * 128;
        static const int force_wm = getenv("DVD_WGR_WM") ? atoi(getenv("DVD_WGR_WM")) : 0;
        ta = d->Cout <= 64 ? 1 : (w4 <= w2 ? 4 : 2); tb = 1;
        if (force_wm && ta > force_wm) ta = force_wm;
        // 128-channel output tile, 5 taps: take 128 input channels too (one 8-wave workgroup per CU whose halves stagger their
        // loads, like the 256-channel tile) when that pads Cin no further: 1.20 -> 1.33 PF/s on 3.1 M x 256 -> 384; with 3 taps
        // the two 4-wave workgroups per CU of the 64-channel form stay 3 % ahead
        static const int wide = getenv("DVD_WGR_NH4") ? atoi(getenv("DVD_WGR_NH4")) : 1;
        if (wide && ta == 2 && d->kw == 5 && !d->up2 && (d->Cin_real + 127) / 128 * 128 == (d->Cin_real + 63) / 64 * 64) tb = 2;
    }
    p.tiles_co = (d->Cout + ta * 64 - 1) / (ta * 64); p.tiles_ci = (d->Cin_real + tb * 64 - 1) / (tb * 64);
    p.s_co = d->s_co; p.s_ci = d->s_ci; p.s_tap = d->s_tap; p.dbias = d->dbias; p.ws = nullptr;
    { static const int xr = getenv("DVD_WG_XCD") ? atoi(getenv("DVD_WG_XCD")) : 1; p.xcd_remap = xr; }
    {
        const size_t esz = d->dtype == DVD_BF16 ? 2 : 4;
        const size_t rows_in = (size_t)d->frames * d->T * p.Hin * p.Win;
        p.x_bytes = ((rows_in - 1) * (size_t)d->ldx + d->C) * esz;
        p.dy_bytes = (((size_t)M - 1) * (size_t)d->ldy + d->Cy) * esz;
        p.maxshift = ((d->kt >> 1) * d->H + (d->kh >> 1)) * d->W + (d->kw >> 1);
    }
    const int ntaps = d->kt * d->kh * d->kw;
    msplit = d->msplit;
    if (msplit < 1) {   // auto: ~16 workgroups per CU, every workgroup keeping >= 4k rows of reduction.  Swept on the
                        // full step with the workspace reduction (ms of wgrad per step): (1024,8192) 292,
                        // (1536,8192) 267, (3072,4096) 247, (4096,4096) 243, (6144,4096) 242, (8192,2048) 252
        static const long long tgt = getenv("DVD_WG_TGT") ? atoll(getenv("DVD_WG_TGT")) : 2048;   // re-swept with whole-round grids: 4096 185.6, 2048 183.2, 1024 183.5 ms
        static const long long minrows = getenv("DVD_WG_ROWS") ? atoll(getenv("DVD_WG_ROWS")) : 4096;
        // (rounds 1-3, swept with whole-round grids: 1024 187.7, 1536 185.1, 2048 190.8, 3072 192.2 ms of weight gradients.  End of round 4 -- the
        //  chain's kernels no longer leave the side stream the CUs they used to -- fewer, longer slices win on the STEP: 256 503.4 / 503.8,
        //  384 498.1 / 498.1, 512 495.6 / 495.7, 768 497.7 / 498.5, 1024 498.4 / 497.3, 1536 500.4 / 500.3, 3072 501.3 ms, one box)
        static const long long tgt_row = getenv("DVD_WGR_TGT") ? atoll(getenv("DVD_WGR_TGT")) : 512;
        const long long base = (long long)p.tiles_co * p.tiles_ci * (mode == 1 ? d->kt * d->kh : ntaps);
        msplit = ((mode == 1 ? tgt_row : tgt) + base - 1) / base;
        const long long cap = M / minrows > 0 ? M / minrows : 1;
        if (msplit > cap) msplit = cap;
        // whole rounds: the workgroups run `conc` at a time (register / LDS limited); a grid a few workgroups over a
        // multiple of that pays a full extra round (20 x 103 = 2060 workgroups = 8.05 rounds of 256 -> 9 rounds)
        static const int quant = getenv("DVD_WG_QUANT") ? atoi(getenv("DVD_WG_QUANT")) : 1;
        const long long conc = 256ll * ((mode == 1 && (ta == 4 || tb == 2)) ? 1 : 2);
        if (quant && base * msplit > conc) {
            const long long rounds = (base * msplit + conc / 2) / conc;           // nearest
            long long ms2 = rounds * conc / base;
            if (ms2 >= 1 && ms2 <= cap) msplit = ms2;
        }
    }
    long long rows = (M + msplit - 1) / msplit;
    {   // a workgroup's row slice is addressed with 32-bit byte offsets
        const long long ldmax = (d->ldx > d->ldy ? d->ldx : d->ldy) * (d->dtype == DVD_BF16 ? 2ll : 4ll);
        const long long cap = (1ll << 31) / ldmax;
        if (rows > cap) rows = cap;
    }
    rows = (rows + 31) / 32 * 32;
    if (rows < 32) rows = 32;
    msplit = (M + rows - 1) / rows;
    p.rows_per_split = (int)rows;
    grid = dim3(p.tiles_co * p.tiles_ci * (mode == 1 ? d->kt * d->kh : ntaps), 1, (unsigned)msplit);
    return DVD_OK;
}

extern "C" long long dvd_conv_wgrad_ws_floats(const dvd_wgrad_desc* d) {
    WgK p; dim3 grid; int ta, tb, mode; long long msplit;
    if (const long long thin = dvd_wgrad_thin_ws_floats(d)) return thin;       // 3 (8) channels on one side: wgrad_thin.hip
    if (wgrad_plan(d, p, grid, ta, tb, msplit, mode) != DVD_OK || msplit <= 1) return 0;
    if (mode == 1) return msplit * (long long)grid.x * d->kw * (ta * 64) * (tb * 64) + msplit * (long long)grid.x * (ta * 64);   // + bias partials
    return msplit * (long long)grid.x * (ta * 64) * (tb * 64);
}

extern "C" int dvd_conv_wgrad(const dvd_wgrad_desc* d, void* stream) {
    WgK p; dim3 grid; int ta, tb, mode; long long msplit;
    const int rc = wgrad_plan(d, p, grid, ta, tb, msplit, mode);
    if (rc != DVD_OK) return rc;
    const int ntaps = d->kt * d->kh * d->kw;
    if (d->ws && dvd_wgrad_thin_ws_floats(d)) {    // the thin ends of the networks (stems, RGB layer): taps folded into the matrix dimension
        if (d->overwrite && (d->s_tap != 1 || d->s_ci != ntaps || d->s_co != (long long)d->Cin_real * ntaps)) return DVD_E_ARG;
        ProfScope prof(1, 2.0 * (double)p.M * d->Cout * d->Cin_real * ntaps, stream, p.M, d->C, d->Cout, ntaps, 0, d->relu_in << 1);
        prof.r.variant = 3;
        return dvd_wgrad_thin(d, stream);
    }
    if (d->ws && msplit > 1) p.ws = d->ws;         // two-phase reduction; a single slice adds straight into dw
    const int overwrite = d->overwrite != 0;       // dw = result instead of dw += result (the caller need not zero it)
    if (overwrite) {
        if (d->s_tap != 1 || d->s_ci != ntaps || d->s_co != (long long)d->Cin_real * ntaps) return DVD_E_ARG;   // dense [co][ci][tap] only
        if (!p.ws && hipMemsetAsync(d->dw, 0, (size_t)d->Cout * d->Cin_real * ntaps * sizeof(float), (hipStream_t)stream) != hipSuccess)
            return DVD_E_LAUNCH;                   // single slice: the kernel adds with atomics
    }
    p.wsb = (p.ws && mode == 1) ? p.ws + msplit * (long long)grid.x * d->kw * (ta * 64) * (tb * 64) : nullptr;
    ProfScope prof(1, 2.0 * (double)p.M * d->Cout * d->Cin_real * ntaps, stream, p.M, d->C, d->Cout, ntaps, (int)msplit,
                   d->up2 | (d->relu_in << 1));
    hipStream_t st = (hipStream_t)stream;
    prof.r.variant = mode == 1 ? 1 : 2;
    if (mode == 1) {
#define LAUNCH_ROW(WM_, KW_)                                                                        \
        do { if (d->relu_in) conv_wgrad_row_kernel<WM_, KW_, true><<<grid, WM_ * 128, 0, st>>>(p);      \
             else conv_wgrad_row_kernel<WM_, KW_, false><<<grid, WM_ * 128, 0, st>>>(p); } while (0)
#define LAUNCH_ROW_UP(WM_)                                                                          \
        do { if (d->relu_in) conv_wgrad_row_kernel<WM_, 3, true, true><<<grid, WM_ * 128, 0, st>>>(p);  \
             else conv_wgrad_row_kernel<WM_, 3, false, true><<<grid, WM_ * 128, 0, st>>>(p); } while (0)
#define LAUNCH_ROW_W(KW_)                                                                           \
        do { if (d->relu_in) conv_wgrad_row_kernel<2, KW_, true, false, 4><<<grid, 512, 0, st>>>(p);    \
             else conv_wgrad_row_kernel<2, KW_, false, false, 4><<<grid, 512, 0, st>>>(p); } while (0)
        if (d->up2) { if (ta == 4) LAUNCH_ROW_UP(4); else if (ta == 2) LAUNCH_ROW_UP(2); else LAUNCH_ROW_UP(1); }
        else if (tb == 2) { if (d->kw == 5) LAUNCH_ROW_W(5); else LAUNCH_ROW_W(3); }
        else
        if (ta == 4)      { if (d->kw == 5) LAUNCH_ROW(4, 5); else LAUNCH_ROW(4, 3); }
        else if (ta == 2) { if (d->kw == 5) LAUNCH_ROW(2, 5); else LAUNCH_ROW(2, 3); }
        else              { if (d->kw == 5) LAUNCH_ROW(1, 5); else LAUNCH_ROW(1, 3); }
#undef LAUNCH_ROW
#undef LAUNCH_ROW_UP
#undef LAUNCH_ROW_W
        if (p.ws) {
            WgRowRedK r{p.ws, p.dw, (int)msplit, (int)grid.x, p.tiles_co, p.tiles_ci, ta * 64, tb * 64, d->kw, p.Cout, p.Cin_real,
                        p.s_co, p.s_ci, p.s_tap, overwrite};
            const long long n = (long long)grid.x * r.KW * r.BMc * r.BNc;
            wgrad_row_reduce_kernel<<<cdiv(n / 4, 256), 256, 0, st>>>(r);
            if (p.dbias) {
                WgBiasRedK b{p.wsb, p.dbias, (int)msplit, (int)grid.x, p.tiles_co, p.tiles_ci, d->kw, d->kt, ta * 64, p.Cout};
                wgrad_row_bias_reduce_kernel<<<p.tiles_co * (ta * 64 / 32), 1024, 0, st>>>(b);
            }
        }
        return launch_status();
    }
    if (d->dtype == DVD_BF16) {
        if (ta == 4) conv_wgrad_kernel<bf16_t, 4, 2><<<grid, NT, 0, st>>>(p);
        else if (tb == 4) conv_wgrad_kernel<bf16_t, 2, 4><<<grid, NT, 0, st>>>(p);
        else conv_wgrad_kernel<bf16_t, 2, 2><<<grid, NT, 0, st>>>(p);
    } else conv_wgrad_kernel<float, 2, 2><<<grid, NT, 0, st>>>(p);
    if (p.ws) {
        WgRedK r{p.ws, p.dw, (int)msplit, (int)grid.x, p.tiles_co * p.tiles_ci, p.tiles_ci, ta * 64, tb * 64, p.Cout,
                 p.Cin_real, p.s_co, p.s_ci, p.s_tap, overwrite};
        const long long n = (long long)grid.x * r.BMc * r.BNc;
        wgrad_reduce_kernel<<<cdiv(n / 4, 256), 256, 0, st>>>(r);
    }
    return launch_status();
}

// Fragment-major image of a forward (or backward-data) pack for conv_halo_gb_kernel; see the comment there.
extern "C" long long dvd_conv_fragment_major_bytes(int ntaps, int Cout, int C) {
    if (ntaps <= 0 || Cout <= 0 || C <= 0) return 0;
    const long long kchunks = (C + 31) / 32, nb32 = (Cout + 127) / 128 * 4;
    return (long long)ntaps * kchunks * nb32 * 2048;
}
extern "C" int dvd_conv_fragment_major(int dtype, const void* w, void* wq, int ntaps, int Cout, int C, void* stream) {
    if (!w || !wq || ntaps <= 0 || Cout <= 0 || C <= 0) return DVD_E_ARG;
    if (dtype != DVD_BF16 || (C & 7)) return DVD_E_SHAPE;
    FragK p{(const bf16_t*)w, (bf16_t*)wq, ntaps, Cout, C, (C + 31) / 32, (Cout + 127) / 128 * 4};
    const long long n = (long long)ntaps * p.kchunks * p.nb32 * 128;
    fragment_major_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(p);
    return launch_status();
}
// 1 when dvd_conv_forward would run this request through the kernel that reads a fragment-major weight image (d->wq)
extern "C" int dvd_conv_wants_fragment_major(const dvd_conv_desc* d) {
    ConvK p; ConvPlan pl;
    dvd_conv_desc t = *d;
    if (!t.w) t.w = t.in;                 // (only the geometry matters here)
    t.wq = t.w;                           // "if the image were supplied"
    if (conv_plan(&t, nullptr, p, pl) != DVD_OK) return 0;
    if (dvd_conv_thin_in_ok(d)) return 2;      // 3 (8) input channels -> 64: wants the image of dvd_conv_thin_image in `wq` instead
    if (dvd_conv_thin_out_ok(d)) return 3;     // 64 -> 3 (8) channels: dvd_conv_thin_out_image
    return ((pl.halo || pl.smallf) && d->dtype == DVD_BF16) ? 1 : 0;
}

extern "C" int dvd_pack_conv_weight(int dtype, const float* w, const float* sigma, int Cout, int Cin, int ntaps,
                                    int Cip, int co_off, int co_tot_f, int co_tot_d, void* wf, void* wd,
                                    int kt, int kh, int kw, int ci_off, int ci_tot, void* stream) {
    if (!w || (!wf && !wd) || Cout <= 0 || Cin <= 0 || ntaps != kt * kh * kw) return DVD_E_ARG;
    if (ci_tot <= 0) { ci_off = 0; ci_tot = Cin; }
    if (ci_off < 0 || ci_off + Cin > ci_tot) return DVD_E_ARG;
    if ((Cip & 7) || Cip < Cin || (wd && (co_tot_d & 7))) return DVD_E_SHAPE;
    PackK p{w, sigma, (char*)wf, (char*)wd, Cout, Cin, ntaps, Cip, co_off, co_tot_f, co_tot_d, kt, kh, kw, ci_off, ci_tot};
    const long long n = (long long)Cout * Cip * ntaps;
    if (dtype == DVD_BF16) pack_weight_kernel<bf16_t><<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(p);
    else if (dtype == DVD_F32) pack_weight_kernel<float><<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(p);
    else return DVD_E_ARG;
    return launch_status();
}
