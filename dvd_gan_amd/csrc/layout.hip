// Boundary layout / dtype conversion: reference NCHW-style fp32 <-> internal channels-last.
#include "common.h"

namespace {

struct LayK { const char* src; char* dst; long long F, P; int C, Cp, A, B, swap; };

__device__ __forceinline__ long long map_frame(const LayK& p, long long f) {
    if (!p.swap) return f;
    const long long a = f / p.B, b = f - a * p.B;       // src frame (a,b) of an [A][B] grid
    return b * p.A + a;                                  // -> internal frame (b,a)
}

// src fp32 [F][C][P] -> dst T [F'][P][Cp]
template <typename T>
__global__ void to_cl_kernel(LayK p) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.F * p.P) return;
    const long long f = i / p.P, pos = i - f * p.P;
    const float* s = reinterpret_cast<const float*>(p.src) + (size_t)f * p.C * p.P + pos;
    T* d = reinterpret_cast<T*>(p.dst) + ((size_t)map_frame(p, f) * p.P + pos) * p.Cp;
    for (int c0 = 0; c0 < p.Cp; c0 += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (c0 + j < p.C) ? s[(size_t)(c0 + j) * p.P] : 0.f;
        store8<T>(d + c0, v);
    }
}
// src T [F'][P][Cp] -> dst fp32 [F][C][P]
template <typename T>
__global__ void from_cl_kernel(LayK p) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.F * p.P) return;
    const long long f = i / p.P, pos = i - f * p.P;
    float* d = reinterpret_cast<float*>(p.dst) + (size_t)f * p.C * p.P + pos;
    const T* s = reinterpret_cast<const T*>(p.src) + ((size_t)map_frame(p, f) * p.P + pos) * p.Cp;
    for (int c0 = 0; c0 < p.C; c0 += 8) {
        float v[8];
        load8<T>(s + c0, v);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (c0 + j < p.C) d[(size_t)(c0 + j) * p.P] = v[j];
    }
}

template <typename S, typename D>
__global__ void convert_kernel(const S* s, D* d, long long n) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= n) {
        float v[8];
        load8<S>(s + i, v);
        store8<D>(d + i, v);
    } else {
        for (long long j = i; j < n; ++j) stf(d + j, ldf(s + j));
    }
}

}  // namespace

extern "C" int dvd_to_channels_last(int dtype, const float* src, void* dst, long long F, int C, long long P,
                                    int Cp, int A, int B, int swap_ab, void* stream) {
    if (!src || !dst || F <= 0 || C <= 0 || P <= 0) return DVD_E_ARG;
    if ((Cp & 7) || Cp < C || (swap_ab && (long long)A * B != F)) return DVD_E_SHAPE;
    LayK p{(const char*)src, (char*)dst, F, P, C, Cp, A, B, swap_ab};
    const unsigned g = cdiv(F * P, 256);
    if (dtype == DVD_BF16) to_cl_kernel<bf16_t><<<g, 256, 0, (hipStream_t)stream>>>(p);
    else if (dtype == DVD_F32) to_cl_kernel<float><<<g, 256, 0, (hipStream_t)stream>>>(p);
    else return DVD_E_ARG;
    return launch_status();
}

extern "C" int dvd_from_channels_last(int dtype, const void* src, float* dst, long long F, int C, long long P,
                                      int Cp, int A, int B, int swap_ab, void* stream) {
    if (!src || !dst || F <= 0 || C <= 0 || P <= 0) return DVD_E_ARG;
    if ((Cp & 7) || Cp < C || (swap_ab && (long long)A * B != F)) return DVD_E_SHAPE;
    LayK p{(const char*)src, (char*)dst, F, P, C, Cp, A, B, swap_ab};
    const unsigned g = cdiv(F * P, 256);
    if (dtype == DVD_BF16) from_cl_kernel<bf16_t><<<g, 256, 0, (hipStream_t)stream>>>(p);
    else if (dtype == DVD_F32) from_cl_kernel<float><<<g, 256, 0, (hipStream_t)stream>>>(p);
    else return DVD_E_ARG;
    return launch_status();
}

extern "C" int dvd_convert(int sdt, const void* src, int ddt, void* dst, long long n, void* stream) {
    if (!src || !dst || n <= 0) return DVD_E_ARG;
    const unsigned g = cdiv(n, 256 * 8);
    hipStream_t s = (hipStream_t)stream;
    if (sdt == DVD_F32 && ddt == DVD_BF16) convert_kernel<float, bf16_t><<<g, 256, 0, s>>>((const float*)src, (bf16_t*)dst, n);
    else if (sdt == DVD_BF16 && ddt == DVD_F32) convert_kernel<bf16_t, float><<<g, 256, 0, s>>>((const bf16_t*)src, (float*)dst, n);
    else if (sdt == DVD_F32 && ddt == DVD_F32) convert_kernel<float, float><<<g, 256, 0, s>>>((const float*)src, (float*)dst, n);
    else if (sdt == DVD_BF16 && ddt == DVD_BF16) convert_kernel<bf16_t, bf16_t><<<g, 256, 0, s>>>((const bf16_t*)src, (bf16_t*)dst, n);
    else return DVD_E_ARG;
    return launch_status();
}

// uint8 clips [B][T][H][W][3] -> fp32 [B][3][T][H][W]: ((x / norm) - mean[c]) / std[c], clips with flip[b] != 0 mirrored in x.
// (ToTensor + Normalize + RandomHorizontalFlip of Dataloader/transform/spatial_transforms.py:38-122,253-268 after the crop.)
namespace {
__global__ void clip_to_tensor_kernel(const unsigned char* src, const unsigned char* flip, float* dst, long long B, int T, int H,
                                      int W, float norm, const float* ms) {
    const long long per = (long long)3 * T * H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * per) return;
    const long long b = i / per;
    long long e = i - b * per;
    const int x = (int)(e % W); e /= W;
    const int y = (int)(e % H); e /= H;
    const int t = (int)(e % T);
    const int c = (int)(e / T);
    const int xs = flip[b] ? W - 1 - x : x;
    const float v = (float)src[((((size_t)b * T + t) * H + y) * W + xs) * 3 + c];
    dst[i] = (v / norm - ms[c]) / ms[3 + c];
}
}  // namespace
extern "C" int dvd_clip_to_tensor(const unsigned char* src, const unsigned char* flip, float* dst, long long B, int T, int H, int W,
                                  float norm_value, const float* mean_std, void* stream) {
    if (!src || !flip || !dst || !mean_std || B <= 0 || T <= 0 || H <= 0 || W <= 0 || norm_value == 0.f) return DVD_E_ARG;
    const long long n = B * 3 * T * H * W;
    clip_to_tensor_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(src, flip, dst, B, T, H, W, norm_value, mean_std);
    return launch_status();
}

extern "C" int dvd_abi_version(void) { return 13; }
extern "C" int dvd_struct_size(int which) {
    switch (which) {
        case DVD_STRUCT_CONV: return (int)sizeof(dvd_conv_desc);
        case DVD_STRUCT_WGRAD: return (int)sizeof(dvd_wgrad_desc);
        case DVD_STRUCT_GRU: return (int)sizeof(dvd_gru_desc);
        case DVD_STRUCT_SN_ITEM: return (int)sizeof(dvd_sn_item);
        case DVD_STRUCT_GRU_STACK: return (int)sizeof(dvd_gru_stack_desc);
        case DVD_STRUCT_PACK_ITEM: return (int)sizeof(dvd_pack_item);
        case DVD_STRUCT_FRAG_ITEM: return (int)sizeof(dvd_frag_item);
        default: return -1;
    }
}
extern "C" const char* dvd_strerror(int code) {
    switch (code) {
        case DVD_OK: return "ok";
        case DVD_E_ARG: return "bad argument (null pointer / non-positive size / inconsistent sizes)";
        case DVD_E_SHAPE: return "unsupported shape (channels % 8, non power-of-two H/W, even kernel, too many rows)";
        case DVD_E_LAUNCH: return "kernel launch failed";
        default: return "unknown error";
    }
}
