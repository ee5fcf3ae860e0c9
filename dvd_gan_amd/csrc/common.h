// Shared device helpers for the gfx950 kernels of libdvdgan_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dvdgan_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;   // raw storage

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;   // a real register vector (a struct here ends up in scratch)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, NaN kept quiet: the hardware conversion (v_cvt_pk_bf16_f32, one instruction per PAIR of values).
// The former bit-twiddling form cost five VALU operations and -- through its NaN test -- a lane-divergent branch per element.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
// two values -> one 32-bit word (lo in bits 0..15)
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    const f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int kPer16B = 4;
    static __device__ __forceinline__ float load(const float* p) { return *p; }
    static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct ElemTraits<bf16_t> {
    static constexpr int kPer16B = 8;
    static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

template <typename T> __device__ __forceinline__ float ldf(const T* p) { return ElemTraits<T>::load(p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v) { ElemTraits<T>::store(p, v); }

// 8 consecutive elements <-> 8 floats (16 B for bf16, 32 B for f32); p must be 16-B aligned.
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
    u32x4 a = *reinterpret_cast<const u32x4*>(p);
    v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xffff0000u);
    v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xffff0000u);
    v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xffff0000u);
    v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<f32x4*>(p) = a; *reinterpret_cast<f32x4*>(p + 4) = b;
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
    u32x4 a;
    a.x = pack2_bf16(v[0], v[1]); a.y = pack2_bf16(v[2], v[3]); a.z = pack2_bf16(v[4], v[5]); a.w = pack2_bf16(v[6], v[7]);
    *reinterpret_cast<u32x4*>(p) = a;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// value a T-typed store would keep (bf16 rounding; identity for fp32)
template <typename T> __device__ __forceinline__ float round_to(float v) { return v; }
template <> __device__ __forceinline__ float round_to<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }

// Gate non-linearities of the ConvGRU (ConvGRU.py:47-51).  Exact mode (fp32 storage): libm expf / tanhf and an IEEE divide.
// bf16 mode: the result is rounded to 8 mantissa bits anyway, so the hardware exponential and reciprocal (about 1e-6
// relative) replace ~50 VALU instructions per element, two lane-divergent branches of tanhf included, by ~6; the fused gate
// epilogues of the large recurrent convolutions were VALU-bound on them.
template <typename T> __device__ __forceinline__ float gate_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
template <> __device__ __forceinline__ float gate_sigmoid<bf16_t>(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
template <typename T> __device__ __forceinline__ float gate_tanh(float x) { return tanhf(x); }
// odd-symmetric form: t = exp(-2|x|) never overflows, (1 - t) / (1 + t) loses relative accuracy only below |x| ~ 1e-4 (the
// cancellation in 1 - t is ~6e-8 absolute), where tanh x = x to 1e-6 relative takes over -- relative error below one bf16 ulp
// everywhere, also for the conv activation epilogue that writes the generated clips (DVD_ACT_TANH, Generator.py:115)
template <> __device__ __forceinline__ float gate_tanh<bf16_t>(float x) {
    const float ax = fabsf(x), t = __expf(-2.f * ax);
    const float r = ax < 2e-3f ? ax : (1.f - t) * __builtin_amdgcn_rcpf(1.f + t);
    return copysignf(r, x);
}

// Internal (not part of the public ABI; hidden visibility: the library exports exactly what include/dvdgan_hip.h declares):
// ConvGRU gate math fused into the convolution epilogue, used
// by gru.hip when the recurrent conv needs no split-K.  mode 1: [u|r] conv, mode 2: out-gate conv (forward);
// mode 3: d(h*r) conv of the backward pass (r, hprev, h32n = fp32 carry, o = dg base), mode 4: carry += conv,
// mode 5: mode 4 + first half of the next BPTT step (gx = dh_out, u_in = u, hr = o-gate, hprev, o = dg of that step).
// nsplit > 1 with a gate epilogue (round 4): `slabs` / `tickets` = workspace of the in-launch split-K combine (splitk_combine,
// conv_igemm.hip): nsplit * tiles * tile floats, one zero-initialised counter per tile (left at zero by every launch).
struct GruEpi {
    int mode, h, ldg;
    const void* gx; const void* hprev; const float* h32p; const void* u_in;
    void* u; void* r; void* hr; void* o; void* hn; float* h32n;
    float* slabs; unsigned* tickets;
};
extern "C" __attribute__((visibility("hidden"))) int dvd_conv_forward_gru(const dvd_conv_desc* d, const GruEpi* g, void* stream);
// Internal: n independent convolutions (d[i], gate epilogue g[i], mode 0 = none / 6 = direct epilogue behind the in-launch split-K
// combine) in ONE launch of kernel `kind`; see conv_igemm.hip
extern "C" __attribute__((visibility("hidden"))) int dvd_conv_forward_group(const dvd_conv_desc* d, const GruEpi* g, int n, int kind, int run, void* stream);
// Internal: weight gradients with 3 (8) channels on one side and 64 on the other (wgrad_thin.hip); 0 floats = not served there
long long dvd_wgrad_thin_ws_floats(const dvd_wgrad_desc* d);
int dvd_wgrad_thin(const dvd_wgrad_desc* d, void* stream);
// Internal: 3 x 3 (x 3) convolutions with 3 (8) input and 64 output channels (conv_thin.hip); d->wq = dvd_conv_thin_image
int dvd_conv_thin_in_ok(const dvd_conv_desc* d);
int dvd_conv_thin_in(const dvd_conv_desc* d, void* stream);
int dvd_conv_thin_out_ok(const dvd_conv_desc* d);      // 64 -> 3 (8) channels, 3 x 3; d->wq = dvd_conv_thin_out_image
int dvd_conv_thin_out(const dvd_conv_desc* d, void* stream);

static inline int ilog2_exact(int v) {   // host: log2 of a power of two, -1 otherwise
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
static inline int launch_status() { return hipGetLastError() == hipSuccess ? DVD_OK : DVD_E_LAUNCH; }
static inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }
