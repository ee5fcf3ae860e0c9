/*
 * dvdgan_hip.h -- C ABI of libdvdgan_hip.so, the MI355X (gfx950) hot path of the DVD-GAN
 * G + D_s + D_t training step.
 *
 * The reference (Harrypotterrrr/DVD-GAN) has no FFI layer: its hot path is torch ATen calls made
 * from nn.Module.forward.  Each entry point below therefore names the reference call site(s)
 * whose vendor kernel it replaces (file:line under /root/reference).  Conventions:
 *   - plain C: device pointers + sizes, no torch types; every call is asynchronous on `stream`
 *     (a hipStream_t passed as void*), never synchronises the device, keeps no global state;
 *   - return value: 0 = ok, <0 = DVD_E_* (bad argument / unsupported shape / launch failure);
 *   - activations are channels-last: [frames][T][H][W][C] with C padded to a multiple of 8,
 *     row stride `ld*` in ELEMENTS; spatial extents H, W must be powers of two;
 *   - `dtype` selects the STORAGE / MFMA operand type of activations and packed weights:
 *     DVD_F32 (exact mode, v_mfma_f32_32x32x2_f32) or DVD_BF16 (v_mfma_f32_32x32x16_bf16);
 *     accumulation, statistics, master weights, gradients of weights are always fp32.
 */
#ifndef DVDGAN_HIP_H
#define DVDGAN_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DVD_F32 0
#define DVD_BF16 1

#define DVD_OK 0
#define DVD_E_ARG (-1)      /* null pointer / non-positive size                                */
#define DVD_E_SHAPE (-2)    /* unsupported shape (C % 8, non power-of-two H/W, even kernel...) */
#define DVD_E_LAUNCH (-3)   /* hipGetLastError() != hipSuccess after a launch                  */

#define DVD_ACT_NONE 0
#define DVD_ACT_RELU 1
#define DVD_ACT_TANH 2
#define DVD_ACT_SIGMOID 3

int dvd_abi_version(void);              /* bumps when a signature changes                      */
const char* dvd_strerror(int code);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution, stride 1, "same" zero padding, 1-D/2-D/3-D taps (kt,kh,kw odd).
 *   out[m][co] = epilogue( sum_{tap,ci} in[pos(m)+tap][ci] * w[tap][co][ci] )
 * Replaces F.conv2d / F.conv3d issued by nn.Conv2d / nn.Conv3d at
 *   Module/ConvGRU.py:49-51 (gate convs, split into x-part and h-part),
 *   Module/GResBlock.py:57,64,73, Module/Generator.py:114 (colorize),
 *   Module/Discriminators.py:186-206 (GBlock), :335-366 (Res3dBlock), :221-226,:376-382 (stems),
 *   :94-96 (q/k/v 1x1), and -- run on the flipped/transposed pack -- their backward-data passes.
 * Also fuses: ReLU on the input (GResBlock.py:52,62 / Discriminators.py:186,192), nearest x2
 * upsample of the input (GResBlock.py:55,72), bias, residual add (GResBlock.py:80), ReLU/tanh
 * on the output (Generator.py:113-115), ReLU-mask of a backward-data result.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int dtype;                 /* DVD_F32 | DVD_BF16                                            */
    int frames, T, H, W;       /* OUTPUT grid; M = frames*T*H*W rows. T>1 only for 3-D convs.   */
    int C, ldi;                /* input channels (multiple of 8) and input row stride           */
    int Cout, ldo;             /* output channels, output row stride                            */
    int kt, kh, kw;            /* taps                                                          */
    int up2;                   /* 1: input stored at H/2 x W/2, nearest-upsampled while loading */
    int relu_in;               /* 1: max(x,0) applied to the input while loading                */
    int nsplit;                /* K is cut in nsplit slices (grid.z); >1 requires `ws`          */
    int act;                   /* DVD_ACT_* (direct epilogue only)                              */
    int out_f32;               /* direct epilogue stores fp32 instead of `dtype`                */
    int ldres, ldmask;         /* row strides of res / mask                                     */
    const void* in;            /* [rows_in][ldi]                                                */
    const void* w;             /* packed [ntaps][Cout][C] (see dvd_pack_conv_weight)            */
    const float* bias;         /* [Cout] or NULL                                                */
    const void* res;           /* [M][ldres] added before act, or NULL                          */
    const void* mask;          /* [M][ldmask]: result multiplied by (mask > 0), or NULL         */
    void* out;                 /* [M][ldo]                                                      */
    float* ws;                 /* non-NULL: raw fp32 partial sums [nsplit][M][Cout] are written
                                  here INSTEAD of the direct epilogue (bias/res/act/mask/out)   */
} dvd_conv_desc;
int dvd_conv_forward(const dvd_conv_desc* d, void* stream);

/* Backward-weight of the same convolution:
 *   dw[co*s_co + ci*s_ci + tap*s_tap] += sum_m dy[m][co] * x[pos(m)+tap][ci]       (fp32 atomics)
 * for co < Cout, ci < Cin_real.  With s_co = Cin_real*ntaps, s_ci = ntaps, s_tap = 1 this is the
 * reference nn.Conv{2,3}d .weight.grad layout [co][ci][kt][kh][kw] (autograd of the sites above). */
typedef struct {
    int dtype;
    int frames, T, H, W;       /* grid of dy (= output grid of the forward conv)                */
    int C, ldx;                /* padded channels of x (multiple of 8), row stride              */
    int Cin_real;              /* only ci < Cin_real is written                                  */
    int Cout, Cy, ldy;         /* real out channels, padded channels present in dy, row stride  */
    int kt, kh, kw;
    int up2, relu_in;          /* same meaning as in dvd_conv_desc (applied to x)               */
    int msplit;                /* row range is cut in msplit slices (grid.z); >=1               */
    long long s_co, s_ci, s_tap;
    const void* x;
    const void* dy;
    float* dw;
} dvd_wgrad_desc;
int dvd_conv_wgrad(const dvd_wgrad_desc* d, void* stream);

/* fp32 master weight [Cout][Cin][ntaps] (reference layout) -> the two operand packs:
 *   wf[tap][co][ci_pad]          forward            (Cin padded with zeros to Cip, multiple of 8)
 *   wd[flip(tap)][ci_pad][co_pad] backward-data     (Cout padded to Cop, multiple of 8)
 * every element divided by *sigma when sigma != NULL (spectral norm, Normalization.py:30-31).
 * co_off/co_tot place this weight as rows [co_off, co_off+Cout) of a wider fused pack
 * (e.g. update|reset|out gates, q|k|v): wf row length stays Cip, wd row length is co_tot. */
int dvd_pack_conv_weight(int dtype, const float* w, const float* sigma, int Cout, int Cin, int ntaps,
                         int Cip, int co_off, int co_tot_f, int co_tot_d, void* wf, void* wd,
                         int kt, int kh, int kw, void* stream);

/* Layout / dtype conversion at the module boundary (reference tensors are NCHW-style fp32):
 *   to_cl:   src fp32 [F][C][P]  -> dst [F][P][Cp]  (pad channels zero-filled)
 *   from_cl: src      [F][P][Cp] -> dst fp32 [F][C][P]
 * F = frames (arbitrary leading dims), P = spatial positions per frame.  `f_outer/f_inner` let the
 * frame index be permuted on the way: src frame (a,b) of an [A][B] grid <-> dst frame (b,a) when
 * swap_ab != 0 (b-major reference frames <-> t-major internal frames, Generator.py:103-106). */
int dvd_to_channels_last(int dtype, const float* src, void* dst, long long F, int C, long long P,
                         int Cp, int A, int B, int swap_ab, void* stream);
int dvd_from_channels_last(int dtype, const void* src, float* dst, long long F, int C, long long P,
                           int Cp, int A, int B, int swap_ab, void* stream);
/* dst[i] (+)= (float)src[i] * alpha, n elements; src/dst dtype given separately */
int dvd_convert(int src_dtype, const void* src, int dst_dtype, void* dst, long long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
