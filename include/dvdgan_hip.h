/*
 * dvdgan_hip.h -- C ABI of libdvdgan_hip.so, the MI355X (gfx950) hot path of the DVD-GAN
 * G + D_s + D_t training step.
 *
 * The reference (Harrypotterrrr/DVD-GAN) has no FFI layer: its hot path is torch ATen calls made
 * from nn.Module.forward.  Each entry point below therefore names the reference call site(s)
 * whose vendor kernel it replaces (file:line under /root/reference).  Conventions:
 *   - plain C: device pointers + sizes, no torch types; every call is asynchronous on `stream`
 *     (a hipStream_t passed as void*), never synchronises the device.  Process-wide state is limited to three
 *     caches / debug aids that never change a result: the split-K plan cache of the ConvGRU wavefront (keyed by the
 *     launch geometry, mutex-protected), the optional profiling hooks (dvd_prof_*) and the dvd_debug_* counters;
 *   - return value: 0 = ok, <0 = DVD_E_* (bad argument / unsupported shape / launch failure);
 *   - activations are channels-last: [frames][T][H][W][C] with C padded to a multiple of 8,
 *     row stride `ld*` in ELEMENTS; spatial extents H, W: powers of two take the fast kernels (LDS-staged footprints,
 *     shift indexing), any other extent (latent_dim 3, 6, ... of Generator.py:15) the tap-by-tap kernels with division indexing;
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
/* the library is built with -fvisibility=hidden: exactly the declarations of this header are exported */
#pragma GCC visibility push(default)

#define DVD_F32 0
#define DVD_BF16 1

#define DVD_OK 0
#define DVD_E_ARG (-1)      /* null pointer / non-positive size                                */
#define DVD_E_SHAPE (-2)    /* unsupported shape (C % 8, even kernel, odd extent under a x2 upsample ...) */
#define DVD_E_LAUNCH (-3)   /* hipGetLastError() != hipSuccess after a launch                  */

#define DVD_ACT_NONE 0
#define DVD_ACT_RELU 1
#define DVD_ACT_TANH 2
#define DVD_ACT_SIGMOID 3

int dvd_abi_version(void);              /* bumps when a signature changes                      */
/* Layout handshake: sizeof() of descriptor struct `which` AS THIS LIBRARY WAS COMPILED (0 = dvd_conv_desc, 1 = dvd_wgrad_desc,
 * 2 = dvd_gru_desc, 3 = dvd_sn_item, 4 = dvd_gru_stack_desc, 5 = dvd_pack_item, 6 = dvd_frag_item; -1 = unknown index).  A binding compares it with the size of its own
 * mirror of the struct at load time (dvd_gan_amd/lib.py does; tests/test_abi_cpu.py also checks the stub printed in
 * INTEGRATION.md): a descriptor that grew on one side only is caught before the library reads past the caller's struct.
 * The reference boundary these structs stand in for is the nn.Module constructor / forward argument lists
 * (Module/Generator.py:15, Module/Discriminators.py:219,371). */
#define DVD_STRUCT_CONV 0
#define DVD_STRUCT_WGRAD 1
#define DVD_STRUCT_GRU 2
#define DVD_STRUCT_SN_ITEM 3
#define DVD_STRUCT_GRU_STACK 4
#define DVD_STRUCT_PACK_ITEM 5
#define DVD_STRUCT_FRAG_ITEM 6
int dvd_struct_size(int which);
/* Optional measurement aid for bench.py: bracket every conv launch with HIP events on its stream.
 * kind 0 = conv_igemm (forward / backward-data), 1 = conv_wgrad.  Report drains the records,
 * returns the launch count, total milliseconds and total algorithmic FLOPs (2*M*Cout*C*taps). */
void dvd_prof_enable(int on);
long long dvd_prof_report(int kind, double* total_ms, double* total_flops);
/* The same, split by kernel variant: n / ms / flops are arrays of nvar entries (or NULL); entry 0 = all launches of `kind`,
 * kind 0: 1 = conv_halo 256x128 tile, 2 = conv_halo 128x128, 3 = conv_halo 256x64, 4 = conv_igemm 128x128,
 * 5 = conv_igemm 256x128, 6 = conv_igemm 256x256, 7 / 8 = whole-frame footprint kernel (4x4 / 8x8 frames) 256x128 / 128x128,
 * 9 = thin-input kernel (3 -> 64 channels); kind 1: 1 = filter-row kernel, 2 = one-tap kernel, 3 = thin-end kernel. */
long long dvd_prof_report_variants(int kind, int nvar, long long* n, double* ms, double* flops);
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
 * upsample of the input (GResBlock.py:55,72) or of the residual, bias, residual add (GResBlock.py:80), ReLU/tanh
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
    int res_up2;               /* 1: res is stored at H/2 x W/2 ([frames*T][H/2][W/2][ldres]) and read through a
                                  nearest x2 upsample -- the 1x1 shortcut of GResBlock.py:72-73 commutes with the
                                  upsample, so it runs on the small grid and is expanded here                  */
    const void* in;            /* [rows_in][ldi]                                                */
    const void* w;             /* packed [ntaps][Cout][C] (see dvd_pack_conv_weight)            */
    const float* bias;         /* [Cout] or NULL                                                */
    const void* res;           /* [M][ldres] added before act, or NULL                          */
    const void* mask;          /* [M][ldmask]: result multiplied by (mask > 0), or NULL         */
    void* out;                 /* [M][ldo]                                                      */
    float* ws;                 /* non-NULL: raw fp32 partial sums [nsplit][M][Cout] are written
                                  here INSTEAD of the direct epilogue (bias/res/act/mask/out)   */
    const void* wq;            /* optional (bf16): the same weights as `w` in fragment-major order
                                  (dvd_conv_fragment_major); requests for which
                                  dvd_conv_wants_fragment_major() is 1 then read their weight operand
                                  straight from L2 into registers instead of staging it in LDS       */
    int wq_kind;               /* what `wq` holds: 0 / 1 = the fragment-major image, 2 = dvd_conv_thin_image, 3 = dvd_conv_thin_out_image
                                  (the value dvd_conv_wants_fragment_major returned for this request) */
    int pool2;                 /* ABI 12: != 0 -> `out` (and `mask`) live on the HALF-size grid [frames][H/2][W/2][ldo] and receive the
                                  2 x 2 SUMS of the convolution result -- the transpose of a nearest x2 upsample, i.e. the input gradient
                                  of a convolution that read an upsampled input (GResBlock.py:57-58) -- formed on the fp32 accumulators
                                  in the epilogue instead of by a dvd_pool pass over a full-size intermediate.  Served by the
                                  halo-staged kernels only: ask dvd_conv_pool2_ok() first.                                   */
} dvd_conv_desc;
int dvd_conv_forward(const dvd_conv_desc* d, void* stream);
/* 1 when dvd_conv_forward serves `d` with d->pool2 set (square 3 / 5 taps, power-of-two frames >= 16 pixels, one K slice,
 * no residual / activation / ConvGRU epilogue), else 0 (the caller runs the plain request and dvd_pool). */
int dvd_conv_pool2_ok(const dvd_conv_desc* d);
/* Fragment-major weight image: [tap][32-channel chunk][32-column block, padded to whole 128-column tiles][k-half pair][lane]
 * [8 channels] -- 1 KiB per MFMA B fragment, fetched by one coalesced 16-byte load per lane.  `w`: a forward or backward-data
 * pack of dvd_pack_conv_weight ([ntaps][Cout][C], bf16, C % 8 == 0).  Re-run after every re-pack (spectral norm: every forward). */
long long dvd_conv_fragment_major_bytes(int ntaps, int Cout, int C);
int dvd_conv_fragment_major(int dtype, const void* w, void* wq, int ntaps, int Cout, int C, void* stream);
/* n images in one launch per 32 items (ABI 13; `items`: HOST array, copied into the kernel arguments): the packs of a whole ConvGRU. */
typedef struct { const void* w; void* wq; int ntaps, Cout, C; } dvd_frag_item;
int dvd_conv_fragment_major_batched(int dtype, const dvd_frag_item* items, int n, void* stream);
int dvd_conv_wants_fragment_major(const dvd_conv_desc* d);     /* 0 = no, 1 = fragment-major image, 2 / 3 = the thin-input / thin-output image below */
/* 3 x 3 (x 3) convolutions from 3 (padded to 8) input channels to 64 output channels (the discriminator stems, the backward-data
 * pass of the RGB layer) fold their KW taps into the K dimension; they take, in `wq`, this image of their [kt*9][64][8] pack:
 * [tap row][k half][channel block][lane][8] (ABI 10). */
long long dvd_conv_thin_image_bytes(int kt);
int dvd_conv_thin_image(const void* w, void* wt, int kt, void* stream);
/* ... and 3 x 3 convolutions from 64 to at most 8 output channels (the RGB layer, the spatial stem's backward-data pass; wants = 3)
 * this image of their [9][rows][64] pack (rows = output rows present in the pack, <= 8): [tap][16-channel step][lane][8] */
long long dvd_conv_thin_out_image_bytes(void);
int dvd_conv_thin_out_image(const void* w, void* wt, int rows, void* stream);

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
    float* dbias;              /* optional: dbias[co] += sum_m dy[m][co] (bias gradient, fused: the workgroups of a
                                  channel tile already stream every dy row and take the 32-row steps in turn; with
                                  `ws` their partial sums go through the workspace and are added in a fixed order,
                                  without it the centre-tap workgroups add with atomics), co < Cout; NULL = skip */
    float* ws;                 /* optional workspace of dvd_conv_wgrad_ws_floats(d) floats: row slices write
                                  their partial tiles there with plain stores and a second kernel reduces
                                  them into dw (deterministic, no atomics).  NULL = fp32 atomics on dw.
                                  With a workspace, 3 x 3 layers over a x2-upsampled input (up2, GResBlock.py:57-58) whose
                                  Cout is a multiple of 64 are worked on the INPUT grid (four dy phases as output channels,
                                  2 / 3 of the products; same bf16 products, another fp32 summation order); without one
                                  they take the direct x2 tiles                                              */
    int overwrite;             /* 1: dw = result instead of dw += result (dense [co][ci][tap] dw only): no zero fill by the caller */
} dvd_wgrad_desc;
int dvd_conv_wgrad(const dvd_wgrad_desc* d, void* stream);
long long dvd_conv_wgrad_ws_floats(const dvd_wgrad_desc* d);   /* 0 = no workspace needed (single slice) */

/* fp32 master weight [Cout][Cin][ntaps] (reference layout) -> the two operand packs:
 *   wf[tap][co][ci_pad]          forward            (Cin padded with zeros to Cip, multiple of 8)
 *   wd[flip(tap)][ci_pad][co_pad] backward-data     (Cout padded to Cop, multiple of 8)
 * every element divided by *sigma when sigma != NULL (spectral norm, Normalization.py:30-31).
 * co_off/co_tot place this weight as rows [co_off, co_off+Cout) of a wider fused pack
 * (e.g. update|reset|out gates, q|k|v): wf row length stays Cip, wd row length is co_tot. */
int dvd_pack_conv_weight(int dtype, const float* w, const float* sigma, int Cout, int Cin, int ntaps,
                         int Cip, int co_off, int co_tot_f, int co_tot_d, void* wf, void* wd,
                         int kt, int kh, int kw, int ci_off, int ci_tot, void* stream);
/* ci_off/ci_tot: the master tensor is [Cout][ci_tot][ntaps] and only input channels
 * [ci_off, ci_off+Cin) are packed (x-part / h-part of a ConvGRU gate, ConvGRU.py:16-18). */
/* The same for n requests in ONE launch per 24 items (ABI 13): the x-part / h-part packs of the three gates of every layer of a
 * ConvGRU (Module/ConvGRU.py:16-18, :104-133: 18 fills per three-layer ConvGRU) were 18 launches of ~18 us in front of its first
 * convolution.  `items` is a HOST array (copied into the kernel arguments); fields as the arguments of dvd_pack_conv_weight. */
typedef struct {
    const float* w; const float* sigma; void* wf; void* wd;
    int Cout, Cin, ntaps, Cip, co_off, co_tot_f, co_tot_d, kt, kh, kw, ci_off, ci_tot;
} dvd_pack_item;
int dvd_pack_conv_weight_batched(int dtype, const dvd_pack_item* items, int n, void* stream);

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


/* ------------------------------------------------------------------------------------------
 * ConvGRU layer, serial half (Module/ConvGRU.py:29-54 for ONE layer, all T steps; the T loop of
 * Module/Generator.py:87-97 runs inside the library).  The caller supplies the batched x-path
 * pre-activations gx[t] = Wx * x_t + b (columns u|r|o, 3*hidden) and receives h_t plus the
 * tensors BPTT needs.  Backward fills dg[t] = d loss / d pre-activation (u|r|o) for every step;
 * weight and input gradients are then batched over T with dvd_conv_wgrad / dvd_conv_forward.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int dtype;
    int T, B, H, W;            /* steps, batch, spatial grid: M = B*H*W rows per step            */
    int hidden, k;             /* hidden channels (multiple of 8), square kernel size            */
    long long gx_stride;       /* elements between gx[t] and gx[t+1]; 0 = same gx for every t    */
    const void* gx;            /* [T|1][M][3*hidden]                                            */
    const void* w_ur;          /* forward pack  [k*k][2*hidden][hidden]  (update|reset, h-part)  */
    const void* w_o;           /* forward pack  [k*k][hidden][hidden]    (out gate, h-part)      */
    const void* wd_ur;         /* backward-data pack [k*k][hidden][2*hidden]                     */
    const void* wd_o;          /* backward-data pack [k*k][hidden][hidden]                       */
    const void* h0;            /* optional initial state [M][hidden] (ConvGRU.py:104), or NULL   */
    void* h_all; void* u_all; void* r_all; void* o_all; void* hr_all;   /* each [T][M][hidden]   */
    float* h32;                /* optional fp32 carry of h, [2][M][hidden] (bf16 mode), or NULL  */
    float* ws;                 /* split-K workspace, see dvd_convgru_ws_floats                   */
    /* backward only */
    const void* dh_out;        /* [T][M][hidden] gradient wrt every h_t, or NULL                 */
    void* dg;                  /* out [T][M][3*hidden]                                           */
    float* carry;              /* scratch [M][hidden] fp32                                       */
    float* dh0;                /* optional out [M][hidden] fp32: gradient wrt h0                 */
    /* forward only */
    int infer;                 /* 1 (sampling path, trainer.py:323-334: no backward will follow): u_all / hr_all are ONE-step
                                  scratch buffers [M][hidden], r and o are not stored (r_all, o_all may be NULL)           */
    /* optional: the four packs once more in fragment-major order (dvd_conv_fragment_major), or NULL -- handed to the recurrent
       convolutions as dvd_conv_desc.wq */
    const void* w_ur_q; const void* w_o_q; const void* wd_ur_q; const void* wd_o_q;
    /* optional (ABI 10): DVD_GRU_TICKETS zero-initialised counters.  When given, a split-K recurrent convolution sums its slices
       and applies the gate math INSIDE the launch (the last workgroup of an output tile to finish does it; nobody waits) instead
       of leaving fp32 slabs to a separate gate kernel: two launches per step and pass instead of four.  Every launch leaves the
       counters at zero; one buffer serves all layers issued on one stream, launches on different streams need their own.
       NULL = slabs + gate kernels (ABI 9 behaviour). */
    unsigned* tickets;
    int combine_max;           /* largest split-K factor combined in-launch; 0 = the library's measured default (4: beyond that the
                                  gate kernel, which spreads the slab reads over the whole chip, is faster)                       */
    int ns_cap;                /* > 0: upper bound on the split-K factor of the recurrent convolutions (tests: the same factors in
                                  the layer-by-layer and the wavefront path -> bit-equal results); 0 = the measured policy        */
} dvd_gru_desc;
#define DVD_GRU_TICKETS 8192
int dvd_convgru_layer_forward(const dvd_gru_desc* d, void* stream);
int dvd_convgru_layer_backward(const dvd_gru_desc* d, void* stream);
/* floats the caller provides as dvd_gru_desc.ws for a layer of this shape (slabs of whole output tiles) */
long long dvd_convgru_ws_floats(int dtype, int B, int H, int W, int hidden, int k);
int dvd_conv_pick_nsplit(int dtype, long long M, int Cout, int C, int ntaps);

/* ------------------------------------------------------------------------------------------
 * A whole ConvGRU (Module/ConvGRU.py:57-133: n_layers chained cells, layer l's input at step t = layer l-1's new state)
 * as a LAYER WAVEFRONT: layer l works on step t while layer l-1 works on step t + 2, so the recurrent convolutions of all
 * layers -- and the x-part convolutions of the layers above the first, now one per step -- are independent of each other
 * and run as GROUPED launches (several convolutions in one grid, their tiles interleaved).  A one-round launch of one
 * layer runs its workgroups in lock step (main loops, then the HBM-bound gate epilogues, overlapping with nothing); in a
 * grouped launch one tile's epilogue runs under another's main loop and the 4 x 4 / 8 x 8 stages issue 2 instead of 6-12
 * dependent launches per step.  Results equal dvd_convgru_layer_* applied layer by layer (bit-equal when the split-K
 * factors are the same: `layer_policy`).  bf16, square power-of-two frames of 4, 8 or >= 16 pixels, 3 x 3 / 5 x 5 filters,
 * fragment-major weight images required: dvd_convgru_stack_ok() says whether a stack is served; callers fall back to the
 * per-layer entry points otherwise.
 * ---------------------------------------------------------------------------------------- */
#define DVD_GRU_STACK_MAX 4
typedef struct {
    int n_layers;                                   /* 1 .. DVD_GRU_STACK_MAX                                          */
    int layer_policy;                               /* 0: split-K factors chosen per grouped launch; 1: dvd_conv_pick_nsplit per
                                                       convolution, x-part unsplit (the per-layer path's arithmetic, for tests) */
    int run;                                        /* tiles per pattern entry of a grouped launch; 0 = default           */
    int cin[DVD_GRU_STACK_MAX];                     /* stored (padded) input channels of layer l's x-part; l >= 1: hidden of l-1 */
    dvd_gru_desc layer[DVD_GRU_STACK_MAX];          /* as for dvd_convgru_layer_*: same dtype, T, B, H, W in every layer.
                                                       layer[l >= 1].gx ([T][M][3 hidden_l], gx_stride = M * 3 hidden_l) is
                                                       WRITTEN by the forward call; layer[l].ws is not used (`ws` below);
                                                       layer[0].tickets serves the whole stack; backward: layer[n-1].dh_out =
                                                       gradient wrt the top layer's states, layer[l < n-1].dh_out = optional
                                                       gradient wrt that layer's states from outside the stack, or NULL        */
    const void* wx[DVD_GRU_STACK_MAX];              /* l >= 1: x-part forward pack [k*k][3 hidden_l][cin_l] ...               */
    const void* wx_q[DVD_GRU_STACK_MAX];            /* ... its fragment-major image ...                                      */
    const float* bx[DVD_GRU_STACK_MAX];             /* ... and bias [3 hidden_l] (update | reset | out)                      */
    const void* wdx[DVD_GRU_STACK_MAX];             /* backward, l >= 1: x-part backward-data pack [k*k][cin_l][3 hidden_l]   */
    const void* wdx_q[DVD_GRU_STACK_MAX];           /* ... its fragment-major image                                          */
    void* dh_mid[DVD_GRU_STACK_MAX];                /* backward, l >= 1: out [T][M][cin_l] = gradient reaching layer l-1's states
                                                       (x-path of layer l + layer[l-1].dh_out)                               */
    float* ws;                                      /* dvd_convgru_stack_ws_floats(d) floats (split-K slabs of the grouped launches) */
} dvd_gru_stack_desc;
int dvd_convgru_stack_ok(const dvd_gru_stack_desc* d, int backward);
long long dvd_convgru_stack_ws_floats(const dvd_gru_stack_desc* d);
int dvd_convgru_stack_forward(const dvd_gru_stack_desc* d, void* stream);
int dvd_convgru_stack_backward(const dvd_gru_stack_desc* d, void* stream);
/* Test hook (tests/test_gpu_gru_stack.py): out[0] = floats the last dvd_convgru_stack_ws_floats call asked for (0 when no
 * member of any grouped launch is split), out[1] = the largest slab cursor any LAUNCHED group has used since the last reset;
 * reset != 0 clears out[1] afterwards.  After one forward + backward of a stack the two must be equal. */
void dvd_debug_stack_ws(long long* out, int reset);

/* ------------------------------------------------------------------------------------------
 * Batch norm statistics + conditional batch norm (Module/Normalization.py:78-88; F.batch_norm +
 * F.linear + mul/add).  `samp[frame]` = row of the condition matrix used by that frame (this
 * is where the reference's condition mis-ordering, Generator.py:109-110, is expressed).
 * gb = Linear(cond) as [B][2*C] (gamma | beta), computed with dvd_linear_forward.
 * ---------------------------------------------------------------------------------------- */
#define DVD_BN_NREP 16      /* copies of the [2C] sums the statistics kernel spreads its atomics over (dvd_bn_finalize adds them up) */
int dvd_bn_stats(int dtype, const void* x, long long rows, int C, int ld, double* sums /*[DVD_BN_NREP][2C], zeroed*/, void* stream);
int dvd_bn_finalize(double* sums /* read, then reset to zero when `rezero`: a persistent workspace needs no fill per call */,
                    long long rows, int C, float eps, float momentum, int training,
                    float* mean, float* rstd, float* run_mean, float* run_var, int rezero, void* stream);
int dvd_cbn_apply(int dtype, const void* x, void* y, long long frames, int P, int C, int ld, const float* mean,
                  const float* rstd, const float* gb, const int* samp, int relu, void* stream);
/* g: gradient wrt the (ReLU'd) output, x: CBN input.  Produces dx and accumulates dgb[B][2C] (zeroed by the caller);
 * s12[2C] is scratch.  a (the stored output) is NOT read and may be NULL: with relu the mask is re-evaluated from x by the
 * same fused multiply-add the forward used (two HBM passes fewer than reading it). */
int dvd_cbn_backward(int dtype, const void* g, const void* a, const void* x, void* dx, long long frames, int P,
                     int C, int ld, const float* mean, const float* rstd, const float* gb, const int* samp, int B,
                     float* dgb, float* s12, int relu, float* part, void* stream);
/* `part`: optional workspace of dvd_cbn_backward_ws_floats() floats -- the per-(frame, pixel chunk) partial sums of dgb are written
 * there and added per condition row in frame order (run-to-run reproducible); NULL = fp32 atomics straight into dgb. */
long long dvd_cbn_backward_ws_floats(long long frames, int P, int C);
/* The same in two stages, for cross-replica batch norm (the reference leaves it as a TODO, Generator.py:57; autograd of
 * F.batch_norm over a global batch): `reduce` accumulates dgb and writes the two per-channel sums s12[2C]; the caller
 * all-reduces s12 over the replicas; `apply` produces dx with rows_total = frames * P summed over the replicas.
 * dvd_cbn_backward == reduce + apply with rows_total = frames * P. */
int dvd_cbn_backward_reduce(int dtype, const void* g, const void* a, const void* x, long long frames, int P, int C, int ld,
                            const float* mean, const float* rstd, const float* gb, const int* samp, int B, float* dgb,
                            float* s12, int relu, float* part, void* stream);
int dvd_cbn_backward_apply(int dtype, const void* g, const void* a, const void* x, void* dx, long long frames, int P, int C,
                           int ld, const float* mean, const float* rstd, const float* gb, const int* samp, const float* s12,
                           long long rows_total, int relu, void* stream);

/* avg / sum pooling over (pt,2,2) windows and its transpose (nearest replication), channels-last.
 * F.avg_pool2d / F.avg_pool3d at Discriminators.py:197,206,225,249,352,361,380,408 and the
 * gradient of F.interpolate(scale_factor=2) (GResBlock.py:55,72). Output grid given; dvd_unpool writes zeros to the last
 * line / column of an odd output grid (the transpose of a floored pooling). */
int dvd_pool(int dtype, const void* x, void* y, long long frames, int To, int Ho, int Wo, int Hi, int Wi /* input grid: Hi/2 = Ho,
             Wi/2 = Wo; an odd extent loses its last line like F.avg_pool2d */, int ld, int pt, float scale, void* stream);
/* ... followed by a ReLU mask on the output grid (mask: same layout as y): the backward of relu -> nearest x2 -> conv
 * (GResBlock.py:52-57) in one pass instead of a pooling pass plus a masking pass */
int dvd_pool_masked(int dtype, const void* x, const void* mask, void* y, long long frames, int To, int Ho, int Wo, int ld, int pt,
                    float scale, void* stream);
int dvd_unpool(int dtype, const void* x, void* y, long long frames, int To, int Ho, int Wo, int ld, int pt, float scale, void* stream);
int dvd_colsum(int dtype, const void* x, long long rows, int C, int ld, float* out /* += */, void* stream);
int dvd_add(int dtype, const void* a, const void* b, void* out, long long n, void* stream);
int dvd_sum_leading(int dtype, const void* in /*[L][n]*/, void* out /*[n]*/, int L, long long n, void* stream);
int dvd_act_backward(int dtype, const void* dy, const void* y, void* dx, long long n, int act, void* stream);
/* utils.py:77-83 vid_downsample on reference-layout fp32 tensors (+ its backward) and
 * utils.py:60-63 frame gather: rows of L floats copied by index (scatter = transpose). */
int dvd_vid_downsample(const float* src, float* dst, int B, int T, int C, int H, int W, int backward, void* stream);
int dvd_row_copy(const float* src, float* dst, const int* idx, long long nrows, long long L, int scatter, void* stream);

/* ------------------------------------------------------------------------------------------
 * Spectral norm (Module/Normalization.py:19-31: mv, mv, norm, dot, div) and its backward
 *   dW_bar += G/sigma - (sum G*W_bar)/sigma^2 * u v^T        (u, v: CURRENT buffers, quirk 7)
 * ---------------------------------------------------------------------------------------- */
int dvd_sn_power_iter(const float* W, int h, int w, float* u, float* v, float* sigma, void* stream);
#define DVD_SN_SCRATCH 512   /* floats of dvd_sn_backward's `scratch`: per-block partial sums of sum G*W_bar, added in a fixed order (no atomics) */
int dvd_sn_backward(const float* G, const float* W, const float* u, const float* v, const float* sigma, int h, int w,
                    float* dW, float* scratch /*DVD_SN_SCRATCH floats*/, void* stream);
int dvd_sn_scale(const float* W, const float* sigma, float* out, long long n, void* stream);
/* The power iterations AND the sigma-normalised MFMA weight images of ALL spectrally normalised convolutions of a network in
 * four launches (the reference runs Normalization.py:19-31 inside every layer's forward): W^T u for every matrix,
 * W v, the finish step (norms, u, v, sigma), the two packs of dvd_pack_conv_weight (whole weight, co_off 0).  `host`: the n
 * items; `dev`: the same array in device memory (the caller uploads it AFTER dvd_sn_batched_prepare has filled the block
 * offsets and reported the scratch size).  wf / wd may be NULL (power iteration only). */
typedef struct {
    const float* W; float* u; float* v; float* sigma;      /* [h][w] fp32, [h], [w], [1]                                    */
    void* wf; void* wd;                                    /* packs [ntaps][cout][cip], [ntaps][cip][cop] in `dtype`, or NULL */
    int h, w;                                              /* matrix view: h = cout, w = cin * ntaps                         */
    int dtype, cout, cin, ntaps, cip, cop;
    int blk_wtu, blk_wv, blk_pack;                         /* first block of this item in the three batched launches (prepare) */
    int pad_;
} dvd_sn_item;
int dvd_sn_batched_prepare(dvd_sn_item* host, int n, long long* scratch_floats);
int dvd_sn_batched(const dvd_sn_item* host, const dvd_sn_item* dev, int n, float* scratch /* *scratch_floats floats */,
                   void* stream);

/* fp32 nn.Linear (Generator.py:75, Normalization.py:80) and nn.Embedding backward (Generator.py:70) */
int dvd_linear_forward(const float* in, const float* W, const float* bias, float* out, int B, int K, int J, void* stream);
int dvd_linear_backward(const float* dout, const float* in, const float* W, float* din, int din_accumulate,
                        float* dW /* += */, float* dbias /* += */, int B, int K, int J, void* stream);
int dvd_embedding_backward(const float* dout, const int* idx, float* dW /* += */, long long n, int D, void* stream);

/* Projection head (Discriminators.py:264-291 / 421-447): relu + spatial sum, SN linear + SN embed */
int dvd_relu_spatial_sum(int dtype, const void* feat, float* hsum, long long F, int P, int C, int ld, void* stream);
int dvd_relu_spatial_sum_backward(int dtype, const float* dh, const void* feat, void* dfeat, long long F, int P, int C,
                                  int ld, void* stream);
int dvd_proj_head_forward(const float* hsum, const float* wl, const float* sl, const float* bias, const float* emb,
                          const float* se, const int* cls, float* out, long long F, int C, void* stream);
int dvd_proj_head_backward(const float* dout, const float* hsum, const float* wl, const float* sl, const float* emb,
                           const float* se, const int* cls, float* dh, float* g_lin, float* g_emb, float* g_bias,
                           long long F, int C, void* stream);

/* Trainer.calc_loss (trainer.py:114-121): *loss += mean(...), dout = grad_scale * d mean / d out */
int dvd_adv_loss(const float* out, long long n, int hinge, int real_flag, float* loss, float* dout, float grad_scale,
                 void* stream);
/* torch.optim.Adam.step (trainer.py:252,268,306) on one flat fp32 buffer */
int dvd_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                  float eps, int step, void* stream);

/* 2-D self attention (Discriminators.py:100-119: bmm, softmax, bmm, gamma*out + x) */
int dvd_attention_forward(int dtype, const void* qkv, int ldq, int dq, int koff, int voff, const void* x, int ldx, int C,
                          const float* gamma, void* y, void* att_out, float* A, long long frames, int N, void* stream);
int dvd_attention_backward(int dtype, const void* qkv, int ldq, int dq, int koff, int voff, const void* dy, int ldx,
                           int C, const float* gamma, const void* att_out, const float* A, float* dS, void* dqkv,
                           float* dgamma, long long frames, int N, void* stream);
/* The same block on the matrix cores (bf16 storage, ABI 10): nothing N x N is kept -- the forward leaves `lse` [frames][N]
 * (log-sum-exp of each query's score row), the backward recomputes the probabilities from q, k and lse; D is scratch
 * [frames][N].  q | k | v at columns 0 | 16 | 32 of qkv (16 query / key channels), C = 32 / 64 / 128 = ldx, N = 32 .. 4096 in
 * whole 32-token blocks: dvd_attention_mfma_ok says whether a call qualifies (everything else: the fp32 kernels above). */
int dvd_attention_mfma_ok(int dtype, int ldq, int dq, int koff, int voff, int ldx, int C, int N);
int dvd_attention_mfma_forward(const void* qkv, int ldq, const void* x, int C, const float* gamma, void* y, void* att_out,
                               float* lse, long long frames, int N, void* stream);
int dvd_attention_mfma_backward(const void* qkv, int ldq, const void* dy, int C, const float* gamma, const void* att_out,
                                const float* lse, float* D, void* dqkv /* every q | k | v column is written */,
                                float* dgamma /* += */, long long frames, int N, void* stream);

/* Spatio-temporal self attention (Module/Attention.py:114-185, SelfAttention over the T*H*W token axis; defined by the
 * reference, not invoked by its Generator): queries from `q` (N rows per clip), keys / values from `kv` -- the
 * 2x2x2 max-pooled projections, Nk = N/8 rows per clip, columns [koff,koff+dq) / [voff,voff+C); A is [frames][N][Nk].
 * softmax(q k^T) (:166-167), V A^T (:172), gamma*out + x (:179). */
int dvd_attention_kv_forward(int dtype, const void* q, int ldq, int dq, const void* kv, int ldk, int koff, int voff,
                             const void* x, int ldx, int C, const float* gamma, void* y, void* att_out, float* A,
                             long long frames, int N, int Nk, void* stream);
int dvd_attention_kv_backward(int dtype, const void* q, int ldq, int dq, const void* kv, int ldk, int koff, int voff,
                              const void* dy, int ldx, int C, const float* gamma, const void* att_out, const float* A,
                              float* dS, void* dq_out, void* dkv_out, float* dgamma, long long frames, int N, int Nk,
                              void* stream);
/* nn.MaxPool3d(kernel_size=2, stride=2) of Module/Attention.py:148,164,170 on a channels-last [frames][2To][2Ho][2Wo][ld]
 * tensor and its backward (gradient to the first maximum of each window) */
int dvd_maxpool3d(int dtype, const void* x, void* y, long long frames, int To, int Ho, int Wo, int ld, void* stream);
int dvd_maxpool3d_backward(int dtype, const void* x, const void* dy, void* dx, long long frames, int To, int Ho, int Wo,
                           int ld, void* stream);

/* ------------------------------------------------------------------------------------------
 * SeparableAttnCell (Module/Attention.py:24-111; the T / W / H cells of SeparableAttn, :8-21): attention along ONE axis of a
 * [B, T, W, H, C] channels-last clip, built -- like the reference -- from raw reshapes of the contiguous NCDHW projections
 * (see csrc/sepattn.hip).  q | k | v are columns [0,Cq) | [koff,koff+Cq) | [voff,voff+C) of `qkv` (one fused 1x1 conv);
 * axis: 0 = T, 1 = W, 2 = H; T, W, H even; attended size A <= 64 (DVD_E_SHAPE otherwise: the score / gradient product tiles are
 * 64 x 32; n_frames > 64 or a 128-pixel attended axis is not served).  Every shape / scratch check runs BEFORE the first launch.
 * Work arrays are per call, sized by the caller:
 *   Qf [B][Cq*N], Kp [B][Cq*N/2], Vp [B][C*N/2] fp32, ksel / vsel bytes of the same counts (max-pool winners),
 *   att [B][A][A/2]   (N = T*W*H; dvd_sepattn_work_floats = floats of Qf + Kp + Vp + att per clip)
 * backward additionally: dO [B][C*N], dS like att, dQf / dKp / dVp like Qf / Kp / Vp; writes the q | k | v columns of dqkv
 * and ADDS to dgamma[1].  The residual path (dx += dy) and the 1x1 convolutions belong to the caller.
 * ---------------------------------------------------------------------------------------- */
long long dvd_sepattn_work_floats(int T, int W, int H, int axis, int C, int Cq);
int dvd_sepattn_forward(int dtype, const void* qkv, int ldq, int Cq, int koff, int voff, const void* x, int ldx, int C,
                        const float* gamma, void* y, float* Qf, float* Kp, float* Vp, unsigned char* ksel,
                        unsigned char* vsel, float* att, long long B, int T, int W, int H, int axis, void* stream);
int dvd_sepattn_backward(int dtype, const void* dy, int ldx, int C, int Cq, const float* gamma, const float* Qf,
                         const float* Kp, const float* Vp, const unsigned char* ksel, const unsigned char* vsel,
                         const float* att, float* dO, float* dS, float* dQf, float* dKp, float* dVp, void* dqkv, int ldq,
                         int koff, int voff, float* dgamma, long long B, int T, int W, int H, int axis, void* stream);

/* Real-data input (Dataloader/transform/spatial_transforms.py:38-122,253-268 after the host-side crop / resize): uint8 clips
 * [B][T][H][W][3] -> fp32 [B][3][T][H][W] = ((x / norm_value) - mean[c]) / std[c], clips with flip[b] != 0 mirrored
 * horizontally.  mean_std: device float[6] = mean RGB | std RGB. */
int dvd_clip_to_tensor(const unsigned char* src, const unsigned char* flip, float* dst, long long B, int T, int H, int W,
                       float norm_value, const float* mean_std, void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
