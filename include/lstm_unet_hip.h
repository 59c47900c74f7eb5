/*
 * lstm_unet_hip.h -- C ABI of the MI355X (gfx950) ConvLSTM-UNet kernel library.
 *
 * The reference (arbellea/LSTM-UNet) has no FFI of its own: every op on the hot path is a
 * TensorFlow/Keras layer call made from Networks.py / losses.py / train2D.py.  Each entry
 * point below replaces one of those call sites (cited per function, file:line into the
 * reference).  The Python host (lstm-unet_amd/lu_native/ops.py) binds them with ctypes;
 * INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - tensors are fp32 unless an entry point says otherwise (the bf16 BPTT tape of the mixed-precision mode travels as raw
 *     bf16 bit patterns behind const void* / explicit dtype fields), channels-last: [frames, H, W, C]; "frame" = one
 *     (batch-slot, time) image
 *   - no process-global state: nothing is read from the environment; kernel-variant overrides used by tests and A/B
 *     tools travel in the descriptors' `flags` fields
 *   - raw device pointers, caller allocates everything (workspace size via *_workspace_bytes)
 *   - functions only ENQUEUE on `stream` (a hipStream_t); they never synchronise, allocate or throw
 *   - return 0 on success, non-zero on error; text via lu_last_error() (thread-local)
 *   - weights use the Keras layout [kh][kw][Cin][Cout] (Cout fastest)
 */
#ifndef LSTM_UNET_HIP_H
#define LSTM_UNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* lu_stream_t; /* hipStream_t */

const char* lu_last_error(void);
int lu_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 *   out[f, oy, ox, n] = bias[n] + sum_{s<n_src} sum_{kh,kw,c} in_s[f, vy, vx, c] * w_s[kh,kw,c,n]
 *   vy = oy*stride + kh - pad_t (vx likewise); with dil == 2 the input is the zero-dilated
 *   tensor (taps with odd vy/vx contribute 0, real row = vy/2) -- the dgrad of a stride-2 conv.
 * Replaces: k.layers.Conv2D call (Networks.py:70,147), the 8 convs of the ConvLSTM2D cell
 * (Networks.py:62-63; two sources = x_t and h_{t-1}), tf.concat + Conv2D in UpBlock2D
 * (Networks.py:145-147; two sources = upsampled and skip), and their input-gradients.
 * ------------------------------------------------------------------------------------------- */
typedef struct lu_conv_src {
    const float* x;         /* activations of this source */
    const float* w;         /* weight row (tap, c) lives at w + tap*w_tap_stride + c*w_row_stride */
    int64_t frame_stride;   /* elements between consecutive frames of x */
    int64_t w_tap_stride;   /* elements between taps of w */
    int32_t pix_stride;     /* elements between consecutive pixels of x (>= C) */
    int32_t C;              /* channels read from this source */
    int32_t w_row_stride;   /* elements between channel rows of w (>= N) */
    int32_t dtype;          /* LU_F32 (0) or LU_BF16 (1): element type of x.  LU_BF16 needs precision == 1, C % 8 == 0, 16-byte
                             * aligned x, strides in ELEMENTS (multiples of 8); the kernel then stages the source without
                             * converting it.  All sources of one launch share the element type. */
} lu_conv_src;

enum { LU_F32 = 0, LU_BF16 = 1 };
enum { LU_EPI_BIAS = 0, LU_EPI_LSTM = 1 };
/* lu_conv_desc.flags: kernel-variant overrides (tests / A-B tools; 0 = the library's own choice) and tape options */
enum {
    LU_CONV_F_PATCH8 = 1,        /* bf16 halo kernel: force 8 x 32-pixel patches */
    LU_CONV_F_PATCH16 = 2,       /* ... force 16 x 32-pixel patches (5x5 only) */
    LU_CONV_F_NO_HALO = 4,       /* fp32: take the general gather kernel where the halo kernel would apply */
    LU_CONV_F_XCD_BY_N = 8,      /* halo kernel: give every XCD its own column tiles */
    LU_CONV_F_GENERAL = 64,      /* general fp32 kernel: the fully general (dilation-capable) instantiation */
    LU_CONV_F_GATES_BF16 = 256,  /* LU_EPI_LSTM, precision 1: gates_out is a bf16 tensor (the bf16 BPTT tape) */
    LU_CONV_F_SRC1_CENTER = 512, /* precision 1 halo kernel, two sources: src[1] contributes its CENTRE tap only (k*k = 1):
                                  * the im2col image of a thin input (lu_im2col_bf16) as one 32-channel chunk, weights
                                  * packed as ONE tap by lu_pack_weights_taps_bf16 */
    LU_CONV_F_NO_BALANCE = 1024, /* few-tile launches (<= 2048 tile x split work items): keep the m-tile-per-XCD block
                                  * numbering instead of the balanced one (equal runs of work items per XCD) -- A/B */
    LU_CONV_F_SLABS_ONLY = 2048, /* LU_EPI_BIAS, splits > 1: stop after the partial slabs -- workspace[s][frames*Hout*Wout][N],
                                  * bias NOT added, `out` unused; the consumer sums them (lu_lstm_gates_fwd_slabs) */
    LU_CONV_F_NO_NARROW = 4096,  /* precision 1, stride-1 3x3 / 5x5 with N = 32 / 64: take the gather kernel instead of the narrow
                                  * blocks of the halo kernel (A/B runs and tests: the two must agree) */
    LU_CONV_F_HALF_BLOCK = 8192, /* precision 1 halo kernel, N > 64 (5x5; 3x3 on bf16 sources): 4-wave blocks on 8 x 32 patches, two
                                  * independent blocks per CU, instead of one 8-wave block on a 16 x 32 patch (bit-identical) */
    LU_CONV_F_H16_SPLIT = 65536  /* LU_EPI_LSTM, precision 1 (ABI v11): h16_out receives the lu_split6 image of h -- [pixel][6][F] bf16, channel blocks
                                  * lo, mid, hi, mid, hi, hi of the exact three-way bf16 split -- instead of its rounded bf16 copy: the recurrent
                                  * operand of the next step of precision 'bf16x3' (fp32 arithmetic on the bf16 MFMA) */
                                 /* (bits 128, 16384, 32768 -- LOOP_GEN1, SPLIT_TAPS, LOOP_GEN2 -- selected A/B forms of rounds 2-5 that ABI v12 no longer ships) */
};

typedef struct lu_conv_desc {
    lu_conv_src src[2];
    int32_t n_src;
    int32_t frames, Hin, Win;       /* real input extent (all sources share it) */
    int32_t Hout, Wout;
    int32_t k, stride, dil;         /* dil in {1,2}; stride in {1,2}; at most one of them is 2 */
    int32_t pad_t, pad_l;
    int32_t N;                      /* output channels */
    int32_t out_pix_stride;
    int32_t epilogue;               /* LU_EPI_* */
    const float* bias;              /* [N] or NULL */
    float* out;                     /* LU_EPI_BIAS: [frames,Hout,Wout,N] */
    int64_t out_frame_stride;
    /* LU_EPI_LSTM (N == 4F, F % 32 == 0): fused ConvLSTM gate block, see lu_lstm_gates_fwd */
    const float* c_prev;            /* [frames,H,W,F] */
    float* c_out;                   /* [frames,H,W,F] */
    float* h_out;                   /* frame stride h_frame_stride, pixel stride F */
    float* gates_out;               /* [frames,H,W,4F] post-activation i,f,g,o or NULL (inference) */
    int64_t c_prev_frame_stride, c_out_frame_stride, h_frame_stride, gates_frame_stride;
    /* optional split of the K axis (taps x channel chunks) for problems with too few output tiles to fill
     * 256 CUs (coarse-level recurrent dgrads): partial tiles go to `workspace`
     * (lu_conv2d_workspace_bytes) and are summed in a fixed order.  LU_EPI_BIAS only; 0/1 = off. */
    int32_t splits;
    int32_t precision;              /* 0: fp32 MFMA (v_mfma_f32_32x32x2_f32).  1: bf16 MFMA operands, fp32 accumulate
                                     * (v_mfma_f32_32x32x16_bf16) -- activations stay fp32 in HBM and are rounded to bf16 while
                                     * staged; every src[i].w must then point to weights packed by lu_pack_weights_bf16
                                     * (w_tap_stride / w_row_stride ignored).  Stride-1 3x3 / 5x5, N > 64, 16-byte aligned sources
                                     * with C % 4 == 0 only (pad a thin input with zero channels: the packed image is zero there).
                                     * (2 -- fp32 MFMA on fragment-packed weights, measured neutral -- left with ABI v12.) */
    void* workspace;
    int64_t out_row_stride;         /* elements between output rows; 0 = dense (Wout * out_pix_stride).  Lets a launch
                                     * write one parity plane of a stride-2 input gradient in place. */
    int32_t k_h;                    /* kernel HEIGHT; 0 = k (square).  With k_h != k the tap window is k_h rows x k columns and
                                     * src[i].w holds k_h*k taps (LU_EPI_BIAS, general kernels only): the parity planes of a
                                     * stride-2 input gradient have 2x2, 2x1, 1x2 and 1x1 windows. */
    int32_t flags;                  /* LU_CONV_F_* */
    void* h16_out;                  /* LU_EPI_LSTM, precision 1, optional: a second copy of h rounded to bf16 (pixel stride F) --
                                     * the recurrent operand of the next step and the x operand of the hoisted weight
                                     * gradient, so neither re-reads / re-rounds the fp32 sequence */
    int64_t h16_frame_stride;
    /* LU_EPI_BIAS, optional (both or neither): the stored value becomes
     *     lrelu(post_scale[n] * (acc + bias[n]) + post_shift[n], post_alpha)
     * i.e. the inference-mode BatchNormalization (lu_bn_finalize_infer) + LeakyReLU of a conv unit (reference
     * Networks.py:69-72,146-151 with training=False) in the same call: with splits > 1 the slab reduce applies it (no extra
     * pass: the tile-starved streaming case), otherwise a short in-place pass over `out` follows the tile kernel.  `out`
     * must be dense.  Same arithmetic as lu_bn_lrelu_apply on the plain conv output (bit-identical). */
    const float* post_scale;        /* [N] */
    const float* post_shift;        /* [N] */
    float post_alpha;
} lu_conv_desc;

/* bf16 weight image for precision == 1: [tap][ceil(C/32)][N][32] bf16 (zero-filled beyond C), built from a
 * [k*k][C][N] fp32 matrix addressed as w + tap*w_tap_stride + c*w_row_stride + n. */
size_t lu_pack_weights_bf16_bytes(int k, int C, int N);
int lu_pack_weights_bf16(const float* w, int64_t w_tap_stride, int w_row_stride, int k, int C, int N, void* out,
                         lu_stream_t stream);
/* precision 'bf16x3' (ABI v11): the same image of the kernel's three-way bf16 split laid out along its row axis -- [tap][6][cp][N], block j = the
 * piece lu_split6's `order` names (rows [C, cp) of a block zero) -- in one pass; lu_pack_weights_bf16_bytes(k, 6 * cp, N) bytes. */
int lu_pack_weights_split6_bf16(const float* w, int64_t w_tap_stride, int w_row_stride, int k, int C, int cp, int N, int order,
                                void* out, lu_stream_t stream);

/* the same image for a rectangular k_h x k tap window given as a list of `taps` = k_h*k taps (size: taps instead of k*k) */
int lu_pack_weights_taps_bf16(const float* w, int64_t w_tap_stride, int w_row_stride, int taps, int C, int N, void* out,
                              lu_stream_t stream);

size_t lu_conv2d_workspace_bytes(const lu_conv_desc* d);
int lu_conv2d_fwd(const lu_conv_desc* d, lu_stream_t stream);

/* wt[kh'][kw'][co][ci] = w[k-1-kh'][k-1-kw'][c_off+ci][co]  for ci < C_sub -- the weight of the
 * input-gradient convolution.  w is [k][k][C_tot][N]. */
int lu_weight_flip_transpose(const float* w, float* wt, int k, int C_tot, int N, int c_off, int C_sub,
                             lu_stream_t stream);

/* All derived weight images of a step in one launch per dependency level (the optimiser changes every parameter at once; the
 * reference has no counterpart: Keras layers read their variables directly, Networks.py:48-58).  One record per
 * lu_weight_flip_transpose (kind 0: src = w, dst = wt, k, C_tot, N, c_off, C = C_sub; 256-thread blocks, one per (tap, 32 x 32
 * tile): nblk = k*k * ceil(C_sub/32) * ceil(N/32)) or lu_pack_weights_bf16 (kind 1: src, tap_stride, row_stride, kk = k*k, C, N,
 * dst; any nblk >= 1, the blocks stride over one thread per 8 packed elements).  The table lives in DEVICE memory, sorted by
 * blk0, blk0 = running sum of nblk; results are those of the single calls, bit for bit. */
typedef struct lu_prep_op {
    int32_t kind;
    int32_t blk0, nblk;
    int32_t k;
    const float* src;
    void* dst;
    int64_t tap_stride;
    int32_t row_stride, kk;
    int32_t C, N;
    int32_t C_tot, c_off;
} lu_prep_op;
int lu_weight_prep_batch(const void* dev_ops, int n_ops, int total_blocks, lu_stream_t stream);

/* Input gradient of a STRIDE-2 convolution without multiplying zeros: the four output parity classes
 * (py, px) are four small stride-1 convolutions of dy.  This packs their kernels:
 *   plane cls = 2*py+px:  sub_cls[ty][tx][n][c] = w[kh][kw][c][n],  kh = py + pad_t - 2*(ty - pady[py]), kw likewise,
 * for the ny[py] x nx[px] taps that fall inside the k x k support (ny[p] = #{kh : (p + pad_t - kh) even}); the four planes
 * are stored back to back.  The caller then launches lu_conv2d_fwd four times with k = nx[px], k_h = ny[py], pads
 * (pady[py], padx[px]) and out_row_stride / out_pix_stride doubled.  (`ks` = ceil(k/2) is kept for sizing: the buffer never
 * needs more than 4*ks*ks*N*C floats.) */
int lu_stride2_dgrad_weights(const float* w, float* sub, int k, int ks, int C, int N, int pad_t, int pad_l,
                             int pady0, int pady1, int padx0, int padx1, lu_stream_t stream);

/* bf16 mode, stride-2 3x3 layers on even input extents: the whole input gradient (all four parity classes) in ONE launch
 * on bf16 MFMA operands.  packed = lu_pack_weights_taps_bf16(sub viewed as [9][1][Nf][C]) of the nine tap matrices
 * lu_stride2_dgrad_weights(w, sub, 3, 2, C, Nf, 0, 0, 1, 0, 1, 0) wrote; dy [frames, Hd, Wd, Nf] fp32, dx dense
 * [frames, 2 Hd, 2 Wd, C] fp32. */
int lu_conv2d_s2_dgrad_bf16(const float* dy, int64_t dy_frame_stride, int32_t dy_pix_stride, const void* packed, int32_t frames,
                            int32_t Hd, int32_t Wd, int32_t Nf, int32_t C, float* dx, lu_stream_t stream);

/* bf16 mode, the same stride-2 3x3 layers FORWARD: reads a bf16 tensor (the bf16 copy of the ConvLSTM output the tape keeps),
 * weights packed by lu_pack_weights_bf16; out = conv + bias, fp32 dense [frames, Hin / 2, Win / 2, N].  Even extents. */
int lu_conv2d_s2_fwd_bf16(const void* x_bf16, int64_t x_frame_stride, int32_t x_pix_stride, const void* packed, const float* bias,
                          int32_t frames, int32_t Hin, int32_t Win, int32_t C, int32_t N, float* out, lu_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Weight gradient:  dw[kh,kw,c,n] (+)= sum_{f,oy,ox} x[f, oy*stride+kh-pad_t, ox*stride+kw-pad_l, c] * dy[f,oy,ox,n]
 * (tape.gradient w.r.t. every conv kernel, train2D.py:92).  Split over pixels into `splits` slabs
 * in `workspace`, then reduced deterministically into dw (dw = beta*dw + sum).
 * ------------------------------------------------------------------------------------------- */
typedef struct lu_wgrad_desc {
    const float* x;  int64_t x_frame_stride;  int32_t x_pix_stride;  int32_t C;   /* input of the conv */
    const float* dy; int64_t dy_frame_stride; int32_t dy_pix_stride; int32_t N;   /* grad of its output */
    int32_t frames, Hin, Win, Hout, Wout;
    int32_t k, stride, pad_t, pad_l;
    float* dw;                /* row (tap,c) at dw + tap*dw_tap_stride + c*dw_row_stride */
    int64_t dw_tap_stride; int32_t dw_row_stride;
    int32_t splits;           /* >= 1 */
    float beta;               /* 0: overwrite, 1: accumulate */
    int32_t precision;        /* 0: fp32 MFMA.  1: x and dy may be rounded to bf16 MFMA operands (fp32 accumulate) where the
                               * bf16 kernel applies (stride-1 3x3 / 5x5, C >= 64, W % 32 == 0); other shapes stay fp32 */
    void* workspace;          /* lu_conv2d_wgrad_workspace_bytes(d) bytes */
    float* dbias;             /* optional: bias gradient dbias[n] = dbias_beta*dbias[n] + sum_p dy[p,n], summed on the side by
                               * the kernel-row variants (stride-1 3x3 / 5x5, C >= 64, W % 16 == 0, 16-byte aligned operands;
                               * an error otherwise -- use lu_colsum).  Replaces `tape.gradient` w.r.t. the Conv2D / ConvLSTM2D
                               * bias (train2D.py:92) without a second pass over dy. */
    float dbias_beta;
    int32_t phase;            /* 0: partial sums + reduce (default).  1: partial sums into the workspace only.  2: the
                               * deterministic reduce of a previous phase-1 call (same descriptor).  Lets a profiler time
                               * the MFMA kernel alone. */
    int32_t x_dtype, dy_dtype;/* LU_F32 / LU_BF16: x / dy are bf16 tensors (precision 1, kernel-row bf16 variant only:
                               * stride-1 3x3 / 5x5, C >= 64, C % 8 == 0, N % 8 == 0, W % 32 == 0); strides in elements */
    int32_t flags;            /* LU_WGRAD_F_* */
    int32_t terms;            /* 0 / 1: plain.  > 1 (ABI v11, precision 'bf16x3'; bf16 kernel-row variant, stride 1, bf16 operands only, no
                               * dbias unless every piece of dy occurs exactly once): the sum runs over `terms` x `frames` frames -- frame f of
                               * term t reads x at x + t * x_term_stride + f * x_frame_stride and dy at dy + t * dy_term_stride + f *
                               * dy_frame_stride (strides in elements).  With the channel blocks of two lu_split6 tensors as the terms (x in
                               * order 0, dy in order 1, term stride = one block) ONE launch sums the six bf16 products of the exact three-way
                               * split: the fp32 weight gradient on the bf16 MFMA, accumulated inside the blocks instead of across launches.
                               * terms == 6 with LU_WGRAD_F_PIECES3 (ABI v12): the PIECE-AWARE form of the same sum -- x and dy are lu_split6 tensors whose first
                               * three channel blocks (block stride x_term_stride / dy_term_stride) are the three pieces of the operand (x, order
                               * 0: lo, mid, hi; dy, order 1: hi, mid, lo); one pass over `frames` frames stages each piece once and issues the
                               * six products from registers (half the staged bytes and LDS fragment reads per MFMA of the terms-as-frames form).
                               * Stride-1 3x3 / 5x5, C % 128 == 0, W % 32 == 0; dbias = column sums of hi + mid + lo = dy is allowed. */
    int64_t x_term_stride, dy_term_stride;
} lu_wgrad_desc;

enum {
    LU_WGRAD_F_NO_ROW = 1,       /* do not take the kernel-row variants */
    LU_WGRAD_F_NO_SMALL3 = 2,    /* do not take the all-taps kernel of the narrow 3x3 layers */
    LU_WGRAD_F_CT64 = 4,         /* bf16 kernel-row variant: force 64-channel tiles */
    LU_WGRAD_F_CT128 = 8,        /* ... force 128-channel tiles */
    LU_WGRAD_F_SMALL_TILE = 16,  /* general kernel: 128 x 128 instead of 128 x 256 tiles */
    LU_WGRAD_F_PRB32 = 32,       /* bf16 kernel-row variant: 32-pixel stages where 64-pixel ones would be taken (A/B, tests) */
    LU_WGRAD_F_NO_RAGGED = 64,   /* fp32 kernel-row variant: only for W % 16 == 0 (other widths: the one-tap-per-block kernel) -- A/B */
    LU_WGRAD_F_NO_NARROW_BF16 = 128, /* precision 1: keep the narrow layers (C < 64) on the fp32 all-taps / general kernels -- A/B */
    LU_WGRAD_F_NO_TAPS9 = 2048,  /* precision 1, stride-1 3x3, C >= 64: keep the kernel-row form (one block = three taps of a kernel row) instead of
                                  * the all-taps form (one block = nine taps of a 64-channel x 128-column tile, 8 waves) -- the previous form; tests */
    LU_WGRAD_F_NO_DMA = 8192,    /* all-taps 3x3 form on bf16 operands: staging registers + ds_write instead of global_load_lds into swizzled LDS rows
                                  * (the library's own choice there, +3.5 %; bit-identical) -- the previous form; tests */
    LU_WGRAD_F_PIECES3 = 131072  /* terms == 6 (ABI v12, precision 'bf16x3'): the piece-aware kernel -- each of the three pieces of x and dy staged once per
                                  * 32-pixel run, the six products issued from registers (wgrad_row_x3_kernel); C % 128 == 0, stride-1 3x3 / 5x5.
                                  * (Bits 256, 512, 1024, 4096, 16384, 32768, 65536 -- KP32, NO_SLIDE, TAPS9, DMA, KP16, XREALIGN, HALF_BLOCK -- selected
                                  * A/B instances of rounds 3-5 that ABI v12 no longer ships; their measurements are in DESIGN 9a.) */
};

size_t lu_conv2d_wgrad_workspace_bytes(const lu_wgrad_desc* d);
int lu_conv2d_wgrad(const lu_wgrad_desc* d, lu_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * ConvLSTM2D gate block (Keras cell maths, constructed Networks.py:48-50):
 *   z = pre-activations [rows,4F], order i,f,g,o;  i,f,o = clip(0.2 z + 0.5, 0, 1), g = tanh
 *   c = f*c_prev + i*g ;  h = o*tanh(c)
 * rows = frames*H*W of one timestep; h is written with its own frame stride (h_seq[b,t]).
 * ------------------------------------------------------------------------------------------- */
int lu_lstm_gates_fwd(const float* z, const float* c_prev, float* c_out, float* h_out, float* gates_out,
                      int32_t frames, int64_t pix_per_frame, int32_t F,
                      int64_t h_frame_stride, lu_stream_t stream);
/* The same gate block fed by the partial slabs of a K-split convolution (LU_CONV_F_SLABS_ONLY):
 *   z = bias + slabs[0] + slabs[1] + ... + slabs[splits-1]   (this order: the sum lu_conv2d_fwd's own reduce forms),
 * slabs = [splits][rows][4F].  Tile-starved steps (streaming inference, B = 1) skip one pass over z this way. */
int lu_lstm_gates_fwd_slabs(const float* slabs, int32_t splits, const float* bias, const float* c_prev, float* c_out,
                            float* h_out, float* gates_out, int32_t frames, int64_t pix_per_frame, int32_t F,
                            int64_t h_frame_stride, lu_stream_t stream);

/* backward of the gate block for one timestep:
 *   dh = dh_a (+ dh_b if non-NULL); dc = dh*o*(1-tanh(c)^2) + dc_in(if non-NULL)
 *   dz[rows,4F] (pre-activation grads; may alias `gates` for an in-place update), dc_prev_out = dc*f   */
int lu_lstm_gates_bwd(const float* gates, const float* c_prev, const float* c_cur,
                      const float* dh_a, int64_t dh_a_frame_stride, const float* dh_b, const float* dc_in,
                      float* dz, float* dc_prev_out,
                      int32_t frames, int64_t pix_per_frame, int32_t F, lu_stream_t stream);

/* The same step on the bf16 BPTT tape of the mixed-precision mode: `gates` holds the saved post-activation gates as
 * bf16 and receives dz as bf16 in place (what the recurrent / input gradients and the weight gradients consume as MFMA
 * operands anyway); c, dh and dc stay fp32. */
int lu_lstm_gates_bwd_bf16(void* gates_dz, const float* c_prev, const float* c_cur,
                           const float* dh_a, int64_t dh_a_frame_stride, const float* dh_b, const float* dc_in,
                           float* dc_prev_out, int32_t frames, int64_t pix_per_frame, int32_t F, lu_stream_t stream);

/* The fp32 gate backward (lu_lstm_gates_bwd, dz in place of the saved gates) that also writes the lu_split6 image of dz --
 * dz6 [frames * pix_per_frame][6][4F] bf16, channel blocks in order 1 (hi, mid, lo, hi, mid, hi) -- in the same pass: the operand of
 * the recurrent / input gradients and the weight gradients of precision 'bf16x3' (ABI v11).  F % 4 == 0, 16-byte aligned tensors. */
int lu_lstm_gates_bwd_split(float* gates_dz, const float* c_prev, const float* c_cur,
                            const float* dh_a, int64_t dh_a_frame_stride, const float* dh_b, const float* dc_in,
                            void* dz6, float* dc_prev_out, int32_t frames, int64_t pix_per_frame, int32_t F, lu_stream_t stream);

/* element-wise conversions between fp32 and bf16 (round to nearest even) on n elements */
int lu_convert_f32_bf16(const float* x, void* y, int64_t n, lu_stream_t stream);
int lu_convert_bf16_f32(const void* x, float* y, int64_t n, lu_stream_t stream);

/* Three-way bf16 split along a reduction axis (precision 'bf16x3': fp32 convolutions on the bf16 MFMA at fp32 accuracy).
 * x [rows][L] fp32 (row stride x_row_stride) -> y [rows][6][Lp] (row stride y_row_stride >= 6 Lp, zero for l in [L, Lp)), where
 * x = hi + mid + lo exactly (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)) and block j holds piece
 * order 0 (activations): lo, mid, hi, mid, hi, hi      order 1 (weights): hi, mid, lo, hi, mid, hi
 * so that a bf16 GEMM over the 6 Lp-long axis of an order-0 operand and an order-1 operand sums the six products
 * lo hi + mid mid + hi lo + mid hi + hi mid + hi hi = x w (1 + O(2^-26)), small terms first.  out_dtype LU_BF16: y is bf16;
 * LU_F32: the same values as fp32 (a weight image on its way into lu_pack_weights_bf16, which rounds them without change).
 * New in this build; no counterpart in the reference (its arithmetic is TensorFlow's fp32 convolution, Networks.py:48-50). */
int lu_split6(const float* x, int64_t rows, int32_t L, int64_t x_row_stride, void* y, int64_t y_row_stride, int32_t Lp,
              int32_t order, int32_t out_dtype, lu_stream_t stream);

/* im2col image of a thin input for LU_CONV_F_SRC1_CENTER: y[f, oy, ox, (kh*k + kw)*C + c] = bf16(x[f, oy+kh-p, ox+kw-p, c])
 * (zero outside the frame, zero for channels >= k*k*C); y is [frames, H, W, 32] bf16, k*k*C <= 32, p = (k-1)/2. */
int lu_im2col_bf16(const float* x, void* y, int32_t frames, int32_t H, int32_t W, int32_t C, int32_t k,
                   lu_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Column reductions over rows of a [rows, C] matrix (row stride ld):  deterministic two-stage.
 *   lu_colsum: out[c] = beta*out[c] + sum_r x[r,c]                       (bias gradients)
 *   lu_bn_stats: sums[0:C] = sum x, sums[C:2C] = sum x^2   (double)      (BatchNormalization batch stats,
 *                                                                          Networks.py:71,150)
 * workspace: lu_colreduce_workspace_bytes(rows, C) bytes.
 * ------------------------------------------------------------------------------------------- */
size_t lu_colreduce_workspace_bytes(int64_t rows, int32_t C);
int lu_colsum(const float* x, int64_t rows, int32_t C, int32_t ld, float* out, float beta, void* workspace,
              lu_stream_t stream);
int lu_bn_stats(const float* x, int64_t rows, int32_t C, double* sums, void* workspace, lu_stream_t stream);

/* from (possibly all-reduced) sums over `count` rows: mean/var -> scale = gamma*rsqrt(var+eps),
 * shift = beta - mean*scale; saves mean and invstd for backward; moving stats <- momentum update
 * with the UNBIASED variance (Keras fused BN).  moving_* may be NULL. */
int lu_bn_finalize_train(const double* sums, double count, const float* gamma, const float* beta, float eps,
                         float momentum, float* moving_mean, float* moving_var, float* scale, float* shift,
                         float* save_mean, float* save_invstd, int32_t C, lu_stream_t stream);
/* inference: scale/shift from the moving statistics */
int lu_bn_finalize_infer(const float* gamma, const float* beta, const float* moving_mean,
                         const float* moving_var, float eps, float* scale, float* shift, int32_t C,
                         lu_stream_t stream);
/* y = leaky_relu(x*scale[c] + shift[c], alpha)   (BN + LeakyReLU(0.3), Networks.py:71-72,150-151) */
int lu_bn_lrelu_apply(const float* x, float* y, const float* scale, const float* shift, float alpha,
                      int64_t rows, int32_t C, lu_stream_t stream);
/* backward, stage 1: dz = dy * (z > 0 ? 1 : alpha), z recomputed from x; sums[0:C] = sum dz,
 * sums[C:2C] = sum dz*xhat (double) */
int lu_bn_lrelu_bwd_reduce(const float* x, const float* dy, const float* scale, const float* shift,
                           const float* save_mean, const float* save_invstd, float alpha, int64_t rows,
                           int32_t C, double* sums, void* workspace, lu_stream_t stream);
/* stage 2: dx = scale * (dz - sum_dz/count - xhat*sum_dz_xhat/count); dgamma = sum_dz_xhat, dbeta = sum_dz */
int lu_bn_lrelu_bwd_apply(const float* x, const float* dy, const float* scale, const float* shift,
                          const float* save_mean, const float* save_invstd, float alpha, const double* sums,
                          double count, float* dx, float* dgamma, float* dbeta, int64_t rows, int32_t C,
                          lu_stream_t stream);
/* bf16 mode: the same with dx stored as bf16 -- for layers whose input gradient and weight gradient both run on bf16 MFMA
 * operands, i.e. whose every reader of dx rounds it to bf16 anyway (same values, half the bytes).  C % 4 == 0, aligned. */
int lu_bn_lrelu_bwd_apply_bf16(const float* x, const float* dy, const float* scale, const float* shift,
                               const float* save_mean, const float* save_invstd, float alpha, const double* sums,
                               double count, void* dx_bf16, float* dgamma, float* dbeta, int64_t rows, int32_t C,
                               lu_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Bilinear x2 up-sampling, edge clamp (k.backend.resize_images(..., 'bilinear'), Networks.py:143).  The source coordinate
 * depends on the TensorFlow release behind the reference:
 *   legacy = 1: src = o / 2 -- the v1 resize_bilinear op (align_corners=False, half_pixel_centers=False) that
 *               keras.backend.resize_images calls in TF 2.0 / 2.1, the release the reference pins (README: 2.0.0a0)
 *   legacy = 0: src = (o + 0.5) / 2 - 0.5 -- half-pixel centres (tf.image.resize v2; later Keras releases)
 * ------------------------------------------------------------------------------------------- */
int lu_upsample2x_fwd(const float* x, float* y, int32_t frames, int32_t H, int32_t W, int32_t C, int32_t legacy,
                      lu_stream_t stream);
/* bf16-mode variants with a bf16 RESULT (round to nearest even after the fp32 arithmetic): for activations whose every consumer
 * rounds them to bf16 MFMA operands anyway -- the values the convolutions see are unchanged, the bytes halve.  C % 4 == 0. */
int lu_upsample2x_fwd_bf16(const float* x, void* y_bf16, int32_t frames, int32_t H, int32_t W, int32_t C, int32_t legacy,
                           lu_stream_t stream);
int lu_bn_lrelu_apply_bf16(const float* x, void* y_bf16, const float* scale, const float* shift, float alpha, int64_t rows,
                           int32_t C, lu_stream_t stream);
/* dx[frames,H,W,C] = transpose of the above applied to dy (pixel stride dy_pix_stride, first C channels) */
int lu_upsample2x_bwd(const float* dy, int32_t dy_pix_stride, float* dx, int32_t frames, int32_t H, int32_t W,
                      int32_t C, int32_t legacy, lu_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Spatial window copy with REFLECT or ZERO fill:
 *   y[f, oy, ox, :] = x[f, map(oy - off_y), map(ox - off_x), :]
 * mode 0 = zero outside, mode 1 = reflect (tf.pad "REFLECT", Networks.py:232).  Negative offsets crop
 * (Networks.py:250).  x may have a pixel stride (channel slice of a wider tensor).
 * ------------------------------------------------------------------------------------------- */
int lu_window_copy(const float* x, int32_t x_pix_stride, float* y, int32_t frames, int32_t Hx, int32_t Wx,
                   int32_t Hy, int32_t Wy, int32_t C, int32_t off_y, int32_t off_x, int32_t mode, float beta,
                   lu_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * 3-class softmax + weighted, validity-masked cross entropy (losses.py:13-27) and its gradient.
 *   stage 1: sums[0] = sum ce*w[gt]*valid, sums[1] = sum valid  (double), optional softmax output
 *   stage 2: dlogits = grad_scale * w[gt]*valid*(softmax - onehot) / (sums[1] + 1e-5)
 * logits [rows,3], gt [rows] float in {-1,0,1,2}.
 * ------------------------------------------------------------------------------------------- */
size_t lu_wce_workspace_bytes(int64_t rows);
int lu_softmax_wce_fwd(const float* logits, const float* gt, const float* class_w, float* softmax_out,
                       double* sums, int64_t rows, void* workspace, lu_stream_t stream);
int lu_softmax_wce_bwd(const float* logits, const float* gt, const float* class_w, const double* sums,
                       float grad_scale, float* dlogits, int64_t rows, lu_stream_t stream);
/* k.layers.Softmax over the 3 classes (Networks.py:206,252): out[r,:] = softmax(logits[r,:]) */
int lu_softmax3(const float* logits, float* out, int64_t rows, lu_stream_t stream);
/* ... for any class count (reference Networks.py:205-206: `last_depth` = filters of the last up-block kernel; ABI v8) */
int lu_softmax_rows(const float* logits, float* out, int64_t rows, int32_t classes, lu_stream_t stream);
/* loss[0] = sums[0] / (sums[1] + 1e-5) */
int lu_wce_finalize(const double* sums, float* loss, lu_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * tf.keras Adam (train2D.py:61,93): m,v update, p -= alpha*m/(sqrt(v)+eps), alpha precomputed by the host
 * as lr*sqrt(1-b2^t)/(1-b1^t); g is multiplied by grad_scale first (1/world for DP mean).
 * ------------------------------------------------------------------------------------------- */
int lu_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float alpha, float b1, float b2,
                 float eps, float grad_scale, lu_stream_t stream);

/* state mask (Networks.py:77-84): x[f, :] *= keep[f] */
int lu_scale_frames(float* x, const float* keep, int32_t frames, int64_t per_frame, lu_stream_t stream);

/* the same mask applied while a training window STARTS (one pass instead of mask + copy + bf16 copy): dst[f, :] = src[f, :] *
 * keep[f] (src NULL: zeros -- reset_states(None); keep NULL: plain copy), dst_bf16 (optional) = the rounded copy of dst the
 * bf16 kernels read.  dst is slot 0 of the h / c tape of a ConvLSTM layer, src the state carried from the previous window. */
int lu_state_begin(float* dst, void* dst_bf16, const float* src, const float* keep, int32_t frames, int64_t per_frame,
                   lu_stream_t stream);

/* [n, a, b] -> [n, b, a]  (NCHW <-> NHWC at the public boundary, losses.py:17-18, train2D.py:98-100) */
int lu_transpose_inner(const float* x, float* y, int64_t n, int32_t a, int32_t b, lu_stream_t stream);

/* HOST utility (no device work): CRC-32C of n bytes, continuing from `crc` (0 to start) -- the checksum of TensorFlow
 * tensor bundles, i.e. of the reference's saved-model files (`model.save_weights(..., save_format='tf')`, train2D.py:235;
 * `model.load_weights`, Inference2D.py:34) read and written by tf_bundle.py without TensorFlow.
 * lu_crc32c("123456789", 9, 0) == 0xE3069283. */
uint32_t lu_crc32c(const void* data, size_t n, uint32_t crc);

/* y = x + y on n elements (gradient fan-in of skip connections) */
int lu_add_inplace(float* y, const float* x, int64_t n, lu_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Inference post-processing on the device: softmax [3,H,W] -> instance labels (Inference2D.py:66-123, host numpy / scipy /
 * OpenCV in the reference).  Integer work, bit-exact; the host (Inference2D.postprocess) sequences these calls and reads
 * back the small per-label arrays.  One workspace (lu_post_workspace_bytes) serves all of them; per-label arrays hold
 * lu_post_max_labels(H, W) entries.
 *   lu_post_label        :66-78  edge threshold 0.2 / cell mask, binary_fill_holes, 8-connected components numbered in
 *                                cv2.connectedComponentsWithStats' order with their areas, nearest-label absorption of edge
 *                                pixels closer than edge_dist (scipy distance_transform_edt tie-break)
 *   lu_post_label_stats          per-label bounding box, 4 x Euler number (8-conn) and component count: an object has
 *                                holes iff components - Euler > 0
 *   lu_post_fill_object  :80-91  per-object binary_fill_holes with the reference's additive label quirk
 *   lu_post_fill_all     :80-91  the same for ALL objects in one launch, driven by the device-side statistics (no host
 *                                read in between); reports when the reference's label order matters (nested objects) or a crop
 *                                exceeds its LDS staging -- lu_post_frame then replays the frame in label order with one
 *                                workgroup on the device
 *   lu_post_newid        :93-123 size / FOV filter and consecutive numbering from the device-side count and areas
 *   lu_post_bbox_of_label        bounding box of one label value in the current map
 *   lu_post_present      :93-103 labels present inside the field of view
 *   lu_post_relabel      :113-123 consecutive uint16 ids of the kept labels
 * ------------------------------------------------------------------------------------------- */
size_t lu_post_workspace_bytes(int32_t H, int32_t W);
int32_t lu_post_max_labels(int32_t H, int32_t W);
int lu_post_label(const float* softmax_chw, int32_t H, int32_t W, float edge_thresh, double edge_dist, void* workspace,
                  int32_t* labels, int32_t* num_labels, int32_t* area, lu_stream_t stream);
int lu_post_label_stats(const int32_t* labels, int32_t H, int32_t W, int32_t num_labels, void* workspace, int32_t* bbox,
                        int32_t* e4, int32_t* ncomp, lu_stream_t stream);
int lu_post_fill_object(int32_t* labels, int32_t H, int32_t W, int32_t n, int32_t x0, int32_t y0, int32_t w, int32_t h,
                        void* workspace, int32_t* dirty, lu_stream_t stream);
int lu_post_fill_all(int32_t* labels, int32_t H, int32_t W, const int32_t* num_labels, const int32_t* bbox, const int32_t* e4,
                     const int32_t* ncomp, int32_t* flags, lu_stream_t stream);
int lu_post_newid(const int32_t* num_labels, const int32_t* area, const int32_t* present, int32_t min_size, int32_t max_size,
                  int32_t table_size, int32_t* newid, const int32_t* flags, int32_t* tail, lu_stream_t stream);
/* One frame in ONE call (label + statistics + snapshot + device-driven hole fill + presence + numbering + relabel + the
 * copy to pinned host memory), and its tail alone (after the host's exact replay of a nested-object frame).  tables: int32
 * [4 + 8 * lu_post_max_labels] = num | dirty | oversize | pad | area | bbox | e4 | ncomp | present; out / host_out: the uint16 map
 * padded to whole int32 words, then {num, dirty, oversize, 0}. */
int lu_post_frame(const float* softmax_chw, int32_t H, int32_t W, float edge_thresh, double edge_dist, int32_t min_size,
                  int32_t max_size, int32_t fov, int32_t single_column, void* workspace, int32_t* labels, int32_t* snapshot,
                  int32_t* tables, int32_t* newid, void* out, void* host_out, lu_stream_t stream);
int lu_post_frame_tail(int32_t H, int32_t W, int32_t min_size, int32_t max_size, int32_t fov, int32_t single_column,
                       const int32_t* labels, int32_t* tables, int32_t* newid, void* out, void* host_out, lu_stream_t stream);
int lu_post_bbox_of_label(const int32_t* labels, int32_t H, int32_t W, int32_t n, int32_t* box, lu_stream_t stream);
int lu_post_present(const int32_t* labels, int32_t H, int32_t W, int32_t fov, int32_t single_column, int32_t num_labels,
                    int32_t* present, lu_stream_t stream);
int lu_post_relabel(const int32_t* labels, int32_t H, int32_t W, const int32_t* newid, int32_t num_labels, uint16_t* out,
                    lu_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LSTM_UNET_HIP_H */
