/*
 * spectrum_hip.h - C ABI of libspectrum_hip.so, the gfx950 (MI355X) kernel library
 * behind the spectrum-vae model hot path.
 *
 * The reference (tomsasani/spectrum-vae) has no FFI: its model classes call
 * tf.keras layers, which dispatch to TensorFlow kernels.  Each entry point below
 * replaces one of those tf.keras layer / loss / optimizer call sites; the comment on
 * every declaration cites the reference call sites it stands in for.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (no allocation, no ownership
 *    transfer); workspaces are sized by the matching *_workspace_bytes query;
 *  - every call only enqueues work on `stream` (a hipStream_t) and returns at once;
 *  - return value 0 = ok, negative = SV_E_* code; sv_last_error() gives the text of the
 *    calling thread's last failure;
 *  - activations are NHWC, Conv2D kernels HWIO (kh,kw,Cin,Cout), Conv2DTranspose kernels
 *    (kh,kw,Cout,Cin), Dense kernels (in,out): the tf.keras layouts;
 *  - `dtype` selects the storage type of activations, filters and activation gradients: SV_F32 everywhere, SV_BF16 in
 *    the strided / transposed conv family (sv_conv2d_{fwd,dgrad,wgrad}[_ws], sv_conv2d_transpose_*, sv_conv2d_wgrad_static),
 *    sv_act_*, sv_bias_grad, sv_batchnorm_*, sv_concat_channels_*, sv_axpy, sv_mask_input_dt and the losses (prediction
 *    and its gradient; targets, masks and weights stay f32).  Sums are always f32; biases, BatchNorm parameters and
 *    statistics, loss values and every kernel / bias gradient (dw, db, dgamma, dbeta) are f32 in both modes: the master
 *    weights and the Adam step stay f32, a bf16 model keeps a bf16 copy of its filters (sv_cast_f32_bf16).
 */
#ifndef SPECTRUM_HIP_H
#define SPECTRUM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sv_stream_t; /* hipStream_t */

enum { SV_OK = 0, SV_E_INVALID = -1, SV_E_UNSUPPORTED = -2, SV_E_LAUNCH = -3, SV_E_WORKSPACE = -4 };
enum { SV_ACT_NONE = 0, SV_ACT_RELU = 1, SV_ACT_ELU = 2, SV_ACT_LRELU = 3, SV_ACT_TANH = 4, SV_ACT_SIGMOID = 5 };
enum { SV_F32 = 0, SV_BF16 = 1 };

/* Geometry of one cross-correlation y[n,oh,ow,co] = sum x[n, oh*sh+kh-pad_t, ow*sw+kw-pad_l, ci] w[kh,kw,ci,co].
 * H, W are the LOGICAL input extent; when up_h/up_w > 1 the stored tensor is (H/up_h, W/up_w) and is read
 * through nearest-neighbour repetition (UpSampling2D folded into the gather, autoencoder.py:364,503).
 * For the transposed entry points the same struct describes the forward conv that the layer is the
 * input-gradient of: (H,W,Cin) is the transposed layer's OUTPUT, (Ho,Wo,Cout) its INPUT. */
typedef struct sv_conv_geom {
    int32_t N, H, W, Cin;
    int32_t KH, KW, SH, SW;
    int32_t pad_t, pad_l;
    int32_t Ho, Wo, Cout;
    int32_t up_h, up_w;
} sv_conv_geom;

const char* sv_last_error(void);
int sv_version(void);

/* ---- Plan-time options.  No counterpart in the reference (TensorFlow picks its kernels by itself).  Where the library
 *      holds several bodies or launch shapes for the same arithmetic, an option selects one; none changes what an entry
 *      point computes.  They exist for the tests (one body held against another) and for measurements; -1, the initial
 *      value of every option, leaves the choice to the planner.  Process-wide; the library never reads the environment on
 *      a launch path.  Names: dgrad_nhwc, dgrad_phase, dgrad_pair, wgrad_nhwc, netfwd_pair, netfwd_phase, small_convt,
 *      s1_mfma (0 = the other body); dgrad_valu, wgrad_valu, netfwd_valu, netfwd_generic (1 = the other body);
 *      dgrad_grid, netfwd_s, netfwd_grid (> 0: workgroups / samples per workgroup).  Unknown names: SV_E_INVALID. */
int sv_set_option(const char* name, int value);
int sv_get_option(const char* name, int* value);

/* ---- Conv2D (autoencoder.py:443,510,521; unet.py:40; context_encoder.py:57,72,172,185,230,242,334,346) */
/* y = act(conv(x, w) + bias); bias may be NULL. */
int sv_conv2d_fwd(const sv_conv_geom* g, const void* x, const void* w, const void* bias, void* y,
                  int act, float alpha, int dtype, sv_stream_t stream);
/* dx = conv input-gradient of dz (dz = dL/d pre-activation). */
int sv_conv2d_dgrad(const sv_conv_geom* g, const void* dz, const void* w, void* dx, int dtype, sv_stream_t stream);
/* dw (+)= conv kernel-gradient; accumulate != 0 adds into dw instead of overwriting it. */
size_t sv_conv2d_wgrad_workspace_bytes(const sv_conv_geom* g);
int sv_conv2d_wgrad(const sv_conv_geom* g, const void* x, const void* dz, void* dw, int accumulate,
                    void* workspace, size_t workspace_bytes, int dtype, sv_stream_t stream);

/* ---- sv_conv2d_wgrad without a workspace argument and without any allocation.  No counterpart in the reference (its
 *      kernel gradients come out of TensorFlow's GradientTape, which allocates freely): this is for callers that cannot
 *      allocate - code under HIP graph capture, tight serving / fine-tuning loops, harnesses that forbid hipMalloc.
 *      The partial sums go through a 16 MiB device array inside the library (one copy per GPU, held until the process
 *      ends; sv_static_scratch_bytes() returns its size); the call uses the CURRENT device's copy.  Geometries whose
 *      sv_conv2d_wgrad_workspace_bytes exceeds that size are refused with SV_E_WORKSPACE.  Results are bit-identical to
 *      sv_conv2d_wgrad.  All calls on one GPU share the array: calls that overlap in time on one GPU (different
 *      streams, different threads) race - serialise them.  The first use on a GPU looks the array's address up, which
 *      must not happen under stream capture: sv_static_scratch_bytes() also performs that lookup for the current
 *      device, so call it once per GPU before recording (a first use inside a capture is refused with SV_E_INVALID).
 *      Inside a sv_deferred_reduce_begin / _flush bracket the sum of this call's partial sums is NOT deferred (the
 *      shared array would be overwritten before the flush): it launches at once. */
size_t sv_static_scratch_bytes(void);
int sv_conv2d_wgrad_static(const sv_conv_geom* g, const void* x, const void* dz, void* dw, int accumulate, int dtype,
                           sv_stream_t stream);

/* ---- fused forms for stride-1 SAME convs with a 3x3 or 5x5 kernel (every conv of CAEPadded,
 *      autoencoder.py:438-459,496-526).  sv_conv2d_fused_supported tells whether a geometry is in the family
 *      (pooled != 0: also that the output extent is even).
 *  sv_conv2d_pool_fwd: y = maxpool2x2(act(conv(x, w) + bias)) with the argmax (0..3, row-major in the window, first
 *      maximum wins) in idx; the full-resolution activation is never written (conv -> MaxPooling2D((2,2)) pairs,
 *      autoencoder.py:443-451).
 *  sv_conv2d_bwd: whole backward of one conv layer.  dz = dy * act'(y) is formed on the fly from the saved
 *      activation OUTPUT y; with idx != NULL, dy / y / idx are at the pooled resolution and dz is that product
 *      placed at each window's argmax, zero elsewhere.  dw += kernel gradient, db += sum(dz) (db may be NULL),
 *      dx = input gradient.  Either half is skipped when its output pointer is NULL (dw == NULL: input gradient
 *      only; dx == NULL: weight gradients only), so the two halves can be issued on different streams.
 *      workspace (needed when dw != NULL): sv_conv2d_wgrad_workspace_bytes(g). */
int sv_conv2d_fused_supported(const sv_conv_geom* g, int pooled);
/* 1 when sv_conv2d_bwd runs this geometry on the matrix-core kernels for 16..128-channel layers (the shapes of the
 * run_model.py:164-219 grid beyond the default stack): dz is then formed once in the workspace and feeds both the
 * input-gradient and the kernel-gradient kernel, so a caller gains nothing from batching this layer's kernel gradient
 * with other layers' (sv_conv2d_wgrad_batch) and should call sv_conv2d_bwd with dx and dw together. */
int sv_conv2d_s1_mfma_bwd(const sv_conv_geom* g);
int sv_conv2d_pool_fwd(const sv_conv_geom* g, const void* x, const void* w, const void* bias, void* y_pooled,
                       uint8_t* idx, int act, float alpha, int dtype, sv_stream_t stream);
int sv_conv2d_bwd(const sv_conv_geom* g, const void* x, const void* w, const void* dy, const void* y,
                  const uint8_t* idx, void* dx, void* dw, void* db, int act, float alpha, void* workspace,
                  size_t workspace_bytes, int dtype, sv_stream_t stream);

/* ---- Whole-network forward of a conv chain: the encoder + decoder stacks of CAEPadded (autoencoder.py:408-536,
 *      `decoded = self.decoder(self.encoder(x))` :530-536) when every layer is a 3x3 stride-1 SAME Conv2D, optionally
 *      preceded by UpSampling2D((2,2)) (up = 2) and/or followed by MaxPooling2D((2,2)) (pool = 1), without
 *      BatchNormalization / Dense.  One launch carries each sample through all layers inside one workgroup; every
 *      layer's output (pooled extent when pool = 1, NHWC) and pooling argmax are also written to y / idx, the tensors
 *      the per-layer backward entry points read.  params: the flat parameter buffer the w_off / b_off (in floats,
 *      b_off < 0: no bias) refer to.  H, W: extents of x.  sv_convnet_supported: 1 when the chain fits the kernel. */
typedef struct sv_net_layer {
    int32_t Cin, Cout, K, up, pool, act;
    float alpha;
    int32_t reserved;
    int64_t w_off, b_off;
    void* y;
    uint8_t* idx;
} sv_net_layer;
int sv_convnet_supported(const sv_net_layer* layers, int n_layers, int N, int H, int W);
int sv_convnet_fwd(const sv_net_layer* layers, int n_layers, int N, int H, int W, const void* x, const void* params,
                   int dtype, sv_stream_t stream);
/* The same launch with the reconstruction loss of the training step (`self.loss(x, decoded)` with MeanSquaredError,
 * run_model.py:140,281) evaluated while the output leaves the workgroup: dpred = loss_scale * dMSE/dy, and
 * partials[0..*n_partials) (room for sv_loss_workspace_bytes()) receive per-workgroup sums of squared errors; the loss is
 * their sum / (N * Hout * Wout * Cout), folded later on the same stream by sv_fold_scalar or by sv_adam_keras_step's
 * fold arguments.  The last layer must not be pooled. */
int sv_convnet_fwd_mse(const sv_net_layer* layers, int n_layers, int N, int H, int W, const void* x, const void* params,
                       const float* y_true, float* dpred, float loss_scale, float* partials, int* n_partials, int dtype,
                       sv_stream_t stream);
/* sv_convnet_fwd_mse with the shuffled batch of `model.fit(..., shuffle=True)` (run_model.py:281-289 feeds the step from
 * a shuffled, batched dataset) gathered inside the launch instead of by a sv_gather_rows launch in front of it.  Sample
 * n of the call is row row_index[first + n] of x_rows, first = (counter[0] - counter_base[0]) * batch_stride (0 when
 * counter is NULL); positions are clamped into [0, n_index) and rows into [0, n_rows).  counter is a device scalar that
 * grows by one per step - the update count in sv_adam_keras_step's step_state[0] - and counter_base a device scalar
 * holding its value at position 0 of row_index (a device-to-device copy when an epoch's order starts): a recorded step
 * then walks the epoch without any per-step argument.  The gathered batch is also written to x_batch ([N][H][W][Cin],
 * float32), which is what the backward entry points and y_true (usually x_batch itself) read.
 * sv_convnet_rows_supported: 1 when the chain takes this form (the layer table the forward kernel is built for). */
typedef struct sv_batch_rows {
    const float* x_rows;         /* [n_rows][H * W * Cin] float32, 16-byte aligned */
    int64_t n_rows;
    const int64_t* row_index;    /* [n_index] */
    int64_t n_index;
    const int64_t* counter;      /* device scalar or NULL */
    const int64_t* counter_base; /* device scalar, NULL with counter */
    int64_t batch_stride;        /* positions per step */
    float* x_batch;              /* [N][H][W][Cin] */
} sv_batch_rows;
int sv_convnet_rows_supported(const sv_net_layer* layers, int n_layers, int N, int H, int W);
int sv_convnet_fwd_mse_rows(const sv_net_layer* layers, int n_layers, int N, int H, int W, const sv_batch_rows* rows,
                            const void* params, const float* y_true, float* dpred, float loss_scale, float* partials,
                            int* n_partials, int dtype, sv_stream_t stream);

/* ---- The kernel/bias-gradient half of sv_conv2d_bwd in two steps, so that the reductions of all layers of one backward
 *      pass (the gradient tape of run_model.py:281-289) run as ONE launch: sv_conv2d_bwd_partial leaves per-workgroup
 *      partial sums in `workspace` (sv_conv2d_wgrad_workspace_bytes(g); one workspace per layer until the finish),
 *      sv_conv2d_wgrad_finish sums them in a fixed order into dw (and db when not NULL; the partial call must have
 *      been made with with_bias = 1), adding to the previous contents when accumulate != 0. */
typedef struct sv_wgrad_job {
    sv_conv_geom g;
    void* workspace;
    void* dw;
    void* db;
    int32_t accumulate;
    int32_t blocks;   /* workgroups per layer the producer was asked for (sv_wgrad_batch_job.blocks); 0: default */
} sv_wgrad_job;
int sv_conv2d_bwd_partial(const sv_conv_geom* g, const void* x, const void* dy, const void* y, const uint8_t* idx,
                          int act, float alpha, int with_bias, void* workspace, size_t workspace_bytes, int dtype,
                          sv_stream_t stream);
int sv_conv2d_wgrad_finish(const sv_wgrad_job* jobs, int n_jobs, int dtype, sv_stream_t stream);

/* ---- The kernel/bias gradients of SEVERAL layers of one backward pass in one launch: the same arithmetic as the
 *      dw/db half of sv_conv2d_bwd per job.  The layers' gradients are independent once every dy exists (the
 *      input-gradient chain has run), so their workgroups share the chip.  Each job needs its own workspace of
 *      sv_conv2d_wgrad_workspace_bytes(g).  reduce != 0: also sum the partial sums into dw / db (a second launch);
 *      reduce == 0: leave them in the workspaces for sv_conv2d_wgrad_finish (db != NULL here decides whether the
 *      bias sums are formed).  3x3 kernels. */
typedef struct sv_wgrad_batch_job {
    sv_conv_geom g;
    const void* x;        /* layer input */
    const void* dy;       /* gradient w.r.t. the layer output (pooled extent when idx != NULL) */
    const void* y;        /* layer output (after activation, pooled when idx != NULL) */
    const uint8_t* idx;   /* pooling argmax or NULL */
    void* dw;
    void* db;             /* or NULL */
    void* workspace;
    uint64_t workspace_bytes;
    int32_t act;
    float alpha;
    int32_t accumulate;
    int32_t blocks;       /* workgroups for this layer, 0: default (512).  Fewer (e.g. 256) suits a batch of several
                             layers: longer-lived workgroups, half the partial sums.  Never more than the default, which
                             sizes the workspace. */
} sv_wgrad_batch_job;
int sv_conv2d_wgrad_batch(const sv_wgrad_batch_job* jobs, int n_jobs, int reduce, int dtype, sv_stream_t stream);

/* ---- The same launch for a training step whose optimizer reads its gradients as sums of partial sums
 *      (`optimizer.apply_gradients` right behind the tape, run_model.py:281-289): the per-workgroup slabs stay in the
 *      jobs' workspaces and slabs_out[j] describes job j's (n_split rows of slab_floats floats; a row holds the kernel
 *      gradient in HWIO order, n_dw floats, then the bias gradient, n_db floats).  dw / db are not touched;
 *      sv_adam_keras_step_slabs consumes the descriptions on the same stream.  Only the few-channel 3x3 family has this
 *      form: sv_conv2d_wgrad_batch_slabs_supported tells (1) whether ALL jobs are in it; otherwise the call fails with
 *      SV_E_UNSUPPORTED and nothing is launched.  At most 8 jobs. */
typedef struct sv_grad_slabs {
    const void* slabs;
    void* dw;             /* where the sums belong: the job's dw ... */
    void* db;             /* ... and db (NULL: the layer has no bias; the bias columns of the slabs are ignored) */
    int32_t slab_floats, n_split, n_dw, n_db;
} sv_grad_slabs;
int sv_conv2d_wgrad_batch_slabs_supported(const sv_wgrad_batch_job* jobs, int n_jobs);
int sv_conv2d_wgrad_batch_slabs(const sv_wgrad_batch_job* jobs, int n_jobs, sv_grad_slabs* slabs_out, int dtype,
                                sv_stream_t stream);

/* ---- The input gradients of SEVERAL consecutive layers of one backward pass (the activation-gradient chain of the tape
 *      in run_model.py:281-289 through the conv stack of autoencoder.py:437-526) in one launch: the same arithmetic as
 *      the dx half of sv_conv2d_bwd per job, jobs in list order.  A job may read, as its dy, the dx of an EARLIER job of
 *      the list (sample n of it depends on sample n only, so a workgroup takes its samples through the whole list
 *      without waiting for any other); no other overlap between one job's dx and another job's tensors is allowed.
 *      All jobs share the batch size.  Only the few-channel 3x3 family has this form:
 *      sv_conv2d_dgrad_chain_supported tells (1) whether the whole list is in it; otherwise the call fails with
 *      SV_E_UNSUPPORTED, nothing is launched, and the caller issues sv_conv2d_bwd per job.  At most 8 jobs. */
typedef struct sv_dgrad_job {
    sv_conv_geom g;
    const void* w;        /* the forward conv's HWIO filters */
    const void* dy;       /* gradient w.r.t. the layer output (pooled extent when idx != NULL) */
    const void* y;        /* layer output (after activation, pooled when idx != NULL) */
    const uint8_t* idx;   /* pooling argmax or NULL */
    void* dx;             /* gradient w.r.t. the layer input (summed over the 2x2 windows when the conv upsampled) */
    int32_t act;
    float alpha;
} sv_dgrad_job;
int sv_conv2d_dgrad_chain_supported(const sv_dgrad_job* jobs, int n_jobs);
int sv_conv2d_dgrad_chain(const sv_dgrad_job* jobs, int n_jobs, int dtype, sv_stream_t stream);

/* ---- sv_convnet_fwd_mse (x given, rows NULL) or sv_convnet_fwd_mse_rows (rows given, x ignored) followed IN THE SAME
 *      LAUNCH by the input-gradient jobs of the backward pass (sv_conv2d_dgrad_chain's list): the first half of the
 *      training step of run_model.py:281-289 (forward, loss, activation-gradient chain of the tape).  The jobs may
 *      read dpred and the layers' y / idx buffers of this call: a workgroup carries its samples from the input to the
 *      last job.  Only the default stack's forward kernel has this form and the jobs must be in the chain's family:
 *      sv_convnet_fwd_mse_dgrad_supported tells (1); otherwise SV_E_UNSUPPORTED and nothing is launched. */
int sv_convnet_fwd_mse_dgrad_supported(const sv_net_layer* layers, int n_layers, int N, int H, int W,
                                       const sv_dgrad_job* jobs, int n_jobs);
int sv_convnet_fwd_mse_dgrad(const sv_net_layer* layers, int n_layers, int N, int H, int W, const void* x,
                             const sv_batch_rows* rows, const void* params, const float* y_true, float* dpred,
                             float loss_scale, float* partials, int* n_partials, const sv_dgrad_job* jobs, int n_jobs,
                             int dtype, sv_stream_t stream);

/* ---- Conv2DTranspose (autoencoder.py:371,382; unet.py:69,123; context_encoder.py:92,105,119,260,276,290) */
/* y[(H,W,Cin) side] = act(conv_transpose(x[(Ho,Wo,Cout) side], w) + bias), w laid out (kh,kw,Cin_of_geom,Cout_of_geom). */
int sv_conv2d_transpose_fwd(const sv_conv_geom* g, const void* x, const void* w, const void* bias, void* y,
                            int act, float alpha, int dtype, sv_stream_t stream);
int sv_conv2d_transpose_dgrad(const sv_conv_geom* g, const void* dz, const void* w, void* dx, int dtype,
                              sv_stream_t stream);
int sv_conv2d_transpose_wgrad(const sv_conv_geom* g, const void* x, const void* dz, void* dw, int accumulate,
                              void* workspace, size_t workspace_bytes, int dtype, sv_stream_t stream);

/* ---- The four operations above with a caller-provided scratch buffer.  The strided 4x4 convs and transposed convs
 *      deep in the U-Net / context-encoder / discriminator stacks (unet.py:45-52,69-78,123; context_encoder.py:63-119,
 *      189-290: 512..1000 channels on 4x4..1x1 maps at batch 32 / 64) have a result of only a few tiles and a reduction
 *      of thousands of terms; with scratch the reduction is cut over many workgroups (partial results in the scratch,
 *      summed in a fixed order together with bias + activation), so results stay bit-reproducible.
 *      sv_conv2d_scratch_bytes gives the size for an operation (0: the geometry gains nothing, pass NULL);
 *      workspace == NULL or too small is not an error: the operation then runs as the plain entry point does. */
#define SV_CONV_OP_FWD 0
#define SV_CONV_OP_DGRAD 1
#define SV_CONV_OP_TRANSPOSE_FWD 2
#define SV_CONV_OP_TRANSPOSE_DGRAD 3
size_t sv_conv2d_scratch_bytes(const sv_conv_geom* g, int op);
int sv_conv2d_fwd_ws(const sv_conv_geom* g, const void* x, const void* w, const void* bias, void* y, int act,
                     float alpha, void* workspace, size_t workspace_bytes, int dtype, sv_stream_t stream);
int sv_conv2d_dgrad_ws(const sv_conv_geom* g, const void* dz, const void* w, void* dx, void* workspace,
                       size_t workspace_bytes, int dtype, sv_stream_t stream);
int sv_conv2d_transpose_fwd_ws(const sv_conv_geom* g, const void* x, const void* w, const void* bias, void* y, int act,
                               float alpha, void* workspace, size_t workspace_bytes, int dtype, sv_stream_t stream);
int sv_conv2d_transpose_dgrad_ws(const sv_conv_geom* g, const void* dz, const void* w, void* dx, void* workspace,
                                 size_t workspace_bytes, int dtype, sv_stream_t stream);

/* ---- activation layers and the backward of fused activations
 *      (LeakyReLU unet.py:53,79; activation= arguments of every Conv2D/Dense above) */
int sv_act_fwd(const void* x, void* y, int64_t n, int act, float alpha, int dtype, sv_stream_t stream);
/* dz = dy * act'(y) expressed through the activation OUTPUT y; if db != NULL also db[c] (+)= sum_rows dz[r,c]. */
/* Workspace (bytes) of the per-channel reductions below: act_bwd (when db != NULL), bias_grad, batchnorm. */
size_t sv_colreduce_workspace_bytes(int64_t rows, int C);
int sv_act_bwd(const void* dy, const void* y, void* dz, void* db, int db_accumulate, int64_t rows, int C,
               int act, float alpha, void* workspace, size_t workspace_bytes, int dtype, sv_stream_t stream);
/* db[c] (+)= sum_rows dz[r,c] */
int sv_bias_grad(const void* dz, void* db, int accumulate, int64_t rows, int C, void* workspace,
                 size_t workspace_bytes, int dtype, sv_stream_t stream);

/* ---- MaxPooling2D, stride == pool (autoencoder.py:33,98,167,319,451); idx is uint8 [N,Ho,Wo,C] = kh*pw+kw of
 *      the first maximum in window scan order */
int sv_maxpool2d_fwd(const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, int ph, int pw, int pad_t,
                     int pad_l, int Ho, int Wo, int dtype, sv_stream_t stream);
int sv_maxpool2d_bwd(const void* dy, const uint8_t* idx, void* dx, int N, int H, int W, int C, int ph, int pw,
                     int pad_t, int pad_l, int Ho, int Wo, int dtype, sv_stream_t stream);

/* ---- UpSampling2D nearest (autoencoder.py:364,503); (H,W) is the INPUT extent */
int sv_upsample2d_fwd(const void* x, void* y, int N, int H, int W, int C, int uh, int uw, int dtype, sv_stream_t stream);
int sv_upsample2d_bwd(const void* dy, void* dx, int N, int H, int W, int C, int uh, int uw, int dtype,
                      sv_stream_t stream);

/* ---- BatchNormalization, axis -1 (autoencoder.py:318,450,468; unet.py:51;
 *      context_encoder.py:64,88,179,237,267,283,341).  rows = product of all other axes.
 *      Training: y = gamma*(x-mean)/sqrt(var_biased+eps)+beta; saves mean and 1/sqrt(var+eps);
 *      moving <- momentum*moving + (1-momentum)*batch, the variance fed to it being unbiased when
 *      unbiased_moving_var != 0 (the fused 4-D kernel of Keras 2).  moving_* may both be NULL.
 *      workspace: sv_colreduce_workspace_bytes(rows, C). */
int sv_batchnorm_fwd_train(const void* x, const float* gamma, const float* beta, void* y, float* save_mean,
                           float* save_invstd, float* moving_mean, float* moving_var, int64_t rows, int C, float eps,
                           float momentum, int unbiased_moving_var, void* workspace, size_t workspace_bytes,
                           int dtype, sv_stream_t stream);
int sv_batchnorm_fwd_infer(const void* x, const float* gamma, const float* beta, const float* moving_mean,
                           const float* moving_var, void* y, int64_t rows, int C, float eps, int dtype,
                           sv_stream_t stream);
/* dgamma/dbeta are overwritten (accumulate == 0) or added to. */
int sv_batchnorm_bwd(const void* x, const void* dy, const float* gamma, const float* save_mean,
                     const float* save_invstd, void* dx, float* dgamma, float* dbeta, int accumulate, int64_t rows,
                     int C, void* workspace, size_t workspace_bytes, int dtype, sv_stream_t stream);
/* The same three operations with what surrounds the layer in the GAN models folded into their launches, and with
 * statistics GROUPS: the rows split into `groups` equal consecutive parts that are normalised independently
 * (save_mean / save_invstd are [groups][C]; the moving statistics receive one update per group, in order) - the
 * discriminator of run_context_encoder.py:165-187 / run_unet.py:128-161 is called on the real and on the generated
 * batch, which is one pass over both here.  workspace: sv_batchnorm_workspace_bytes(rows, C, groups).
 *  - act / alpha (forward): a following activation layer (unet.py:51-53, BatchNormalization -> LeakyReLU) applied to the
 *    normalised value in the same launch; y is then the activation's output.
 *  - y_act / act_out (backward): that output; the incoming gradient is dy * act_out'(y_act).
 *  - act_in / alpha_in / dbias_in (backward): the activation of the conv that PRODUCED x (context_encoder.py:61-64,
 *    Conv2D(activation=LeakyReLU) -> BatchNormalization): dx is multiplied by act_in'(x), i.e. it is the gradient at the
 *    conv's pre-activation, and dbias_in (may be NULL) receives its column sums, the conv's bias gradient. */
size_t sv_batchnorm_workspace_bytes(int64_t rows, int C, int groups);
int sv_batchnorm_act_fwd_train(const void* x, const float* gamma, const float* beta, void* y, float* save_mean,
                               float* save_invstd, float* moving_mean, float* moving_var, int64_t rows, int C, float eps,
                               float momentum, int unbiased_moving_var, int groups, int act, float alpha, void* workspace,
                               size_t workspace_bytes, int dtype, sv_stream_t stream);
int sv_batchnorm_act_fwd_infer(const void* x, const float* gamma, const float* beta, const float* moving_mean,
                               const float* moving_var, void* y, int64_t rows, int C, float eps, int act, float alpha,
                               int dtype, sv_stream_t stream);
int sv_batchnorm_act_bwd(const void* x, const void* dy, const void* y_act, int act_out, float alpha_out, const float* gamma,
                         const float* save_mean, const float* save_invstd, void* dx, float* dgamma, float* dbeta,
                         int accumulate, int act_in, float alpha_in, float* dbias_in, int dbias_accumulate, int64_t rows,
                         int C, int groups, void* workspace, size_t workspace_bytes, int dtype, sv_stream_t stream);

/* BatchNormalization whose batch statistics run over the rows of EVERY data-parallel replica (the reference trains on one
 * device, run_context_encoder.py:225-260; N replicas at batch B then equal one device at batch N * B), as the layer's three
 * steps with the replicas' exchange - the caller's, e.g. RCCL - between them:
 *   forward:  sv_batchnorm_sync_moments -> moments[groups][3][C] (double): rows, mean, sum of squared deviations about it;
 *             all-gather, combine (count-weighted means; deviations + count * (mean - combined mean)^2), invstd =
 *             1 / sqrt(var + eps), moving statistics: the caller;  sv_batchnorm_sync_fwd_apply normalises with them.
 *   backward: sv_batchnorm_sync_bwd_sums -> sums[groups][2][C] = column sums of dy' and dy' * xhat over this replica's rows
 *             (dy' = dy * act_out'(y_act)), and dgamma / dbeta (+)= those same local sums;  all-reduce (sum) of sums;
 *             sv_batchnorm_sync_bwd_apply forms dx with the summed sums and all_rows_per_group = the rows of a statistics
 *             group over all replicas.  Other arguments as in sv_batchnorm_act_fwd_train / sv_batchnorm_act_bwd. */
int sv_batchnorm_sync_moments(const void* x, double* moments, int64_t rows, int C, int groups, void* workspace,
                              size_t workspace_bytes, int dtype, sv_stream_t stream);
int sv_batchnorm_sync_fwd_apply(const void* x, const float* gamma, const float* beta, const float* mean,
                                const float* invstd, void* y, int64_t rows, int C, int groups, int act, float alpha,
                                int dtype, sv_stream_t stream);
int sv_batchnorm_sync_bwd_sums(const void* x, const void* dy, const void* y_act, int act_out, float alpha_out,
                               const float* mean, const float* invstd, float* sums, float* dgamma, float* dbeta,
                               int accumulate, int64_t rows, int C, int groups, void* workspace, size_t workspace_bytes,
                               int dtype, sv_stream_t stream);
int sv_batchnorm_sync_bwd_apply(const void* x, const void* dy, const void* y_act, int act_out, float alpha_out,
                                const float* gamma, const float* mean, const float* invstd, const float* sums,
                                int64_t all_rows_per_group, void* dx, int act_in, float alpha_in, float* dbias_in,
                                int dbias_accumulate, int64_t rows, int C, int groups, void* workspace,
                                size_t workspace_bytes, int dtype, sv_stream_t stream);

/* ---- Concatenate on the channel axis (unet.py:144): y[r] = [a[r], b[r]] and its split backward */
int sv_concat_channels_fwd(const void* a, const void* b, void* y, int64_t rows, int Ca, int Cb, int dtype,
                           sv_stream_t stream);
int sv_concat_channels_bwd(const void* dy, void* da, void* db, int64_t rows, int Ca, int Cb, int dtype,
                           sv_stream_t stream);

/* ---- Dense (autoencoder.py:38-40,107,110,176,179,222-241,330-352,466-492): y = act(x[M,K] w[K,N] + b) */
int sv_dense_fwd(const void* x, const void* w, const void* bias, void* y, int M, int K, int N, int act, float alpha,
                 int dtype, sv_stream_t stream);
int sv_dense_dgrad(const void* dz, const void* w, void* dx, int M, int K, int N, int dtype, sv_stream_t stream);
int sv_dense_wgrad(const void* x, const void* dz, void* dw, int accumulate, int M, int K, int N, int dtype,
                   sv_stream_t stream);
/* sv_dense_wgrad with scratch: the reduction over the batch is cut over about 256 workgroups (a Dense kernel gradient has
 * few result tiles and M terms per element), partial sums go through `workspace` (sv_dense_wgrad_workspace_bytes; 0: the
 * shape gains nothing) and are added in a fixed order.  workspace == NULL or too small is not an error: the call then
 * runs as sv_dense_wgrad does. */
size_t sv_dense_wgrad_workspace_bytes(int M, int K, int N);
int sv_dense_wgrad_ws(const void* x, const void* dz, void* dw, int accumulate, int M, int K, int N, void* workspace,
                      size_t workspace_bytes, int dtype, sv_stream_t stream);

/* ---- Dropout (autoencoder.py:108,177,223,331): inverted dropout with a counter-based generator;
 *      element i is kept iff hash(seed, offset + i) >= rate. */
int sv_dropout_fwd(const void* x, void* y, int64_t n, float rate, uint64_t seed, uint64_t offset, int dtype,
                   sv_stream_t stream);
int sv_dropout_bwd(const void* dy, void* dx, int64_t n, float rate, uint64_t seed, uint64_t offset, int dtype,
                   sv_stream_t stream);

/* ---- reduce_sum over the H axis of NHWC (autoencoder.py:36) */
int sv_reduce_sum_h_fwd(const void* x, void* y, int N, int H, int W, int C, int dtype, sv_stream_t stream);
int sv_reduce_sum_h_bwd(const void* dy, void* dx, int N, int H, int W, int C, int dtype, sv_stream_t stream);

/* ---- Sampling (autoencoder.py:264-272): z = mean + exp(0.5*logvar)*eps, eps ~ N(0,1) from a counter-based
 *      generator (seed, offset) or, when eps_in != NULL, supplied by the caller.  kl_out (1 float) receives
 *      mean_b( -1/2 sum_d(1 + logvar - mean^2 - exp(logvar)) ); the KL term has no counterpart in the reference.
 *      eps_out (optional) stores the epsilon used, for the backward pass.  rng_counter (optional, one uint64 on the
 *      device): its value is added to offset and the call advances it by B*D after drawing, so a recorded (HIP graph)
 *      step draws fresh values at every replay. */
int sv_reparam_kl_fwd(const float* mean, const float* logvar, const float* eps_in, float* z, float* eps_out,
                      float* kl_out, float* partials /* sv_loss_workspace_bytes() */, int B, int D, uint64_t seed,
                      uint64_t offset, uint64_t* rng_counter, sv_stream_t stream);
/* dmean = dz + kl_weight*mean/B ; dlogvar = dz*0.5*exp(0.5*logvar)*eps + kl_weight*0.5*(exp(logvar)-1)/B */
int sv_reparam_kl_bwd(const float* mean, const float* logvar, const float* eps, const float* dz, float* dmean,
                      float* dlogvar, float kl_weight, int B, int D, sv_stream_t stream);

/* ---- losses.  Each writes the scalar loss to loss_out[0] (device; += when accumulate != 0) and, if dpred != NULL,
 *      loss_scale * dL/dpred.  partials is scratch of sv_loss_workspace_bytes().
 *  MSE (run_model.py:140,281; run_unet.py:102): mean over all elements.  With weight != NULL
 *  (run_context_encoder.py:190-198) weight is [n_pix] and the loss is sum_p w[p]*mean_c(d^2) / n_pix.
 *  weight_stride lets the caller pass channel 0 of an NHWC mask directly (stride = C). */
size_t sv_loss_workspace_bytes(void);
int sv_loss_mse(const void* y_true, const void* y_pred, const float* weight, int weight_stride, float weight_scale,
                float* loss_out, int accumulate, void* dpred, float loss_scale, int64_t n_pix, int C, float* partials,
                float* running, float running_weight, int dtype, sv_stream_t stream);
/* running (may be NULL): running[0] += running_weight * loss in the same launch - the sample-weighted epoch loss that
 * model.fit reports (run_model.py:141-148), without a launch of its own. */
/* per-pixel map, reduction="none" (run_context_encoder.py:493-501; run_model.py:361-362): out[p] = w[p]*mean_c(d^2) */
int sv_loss_mse_map(const void* y_true, const void* y_pred, const float* weight, int weight_stride,
                    float weight_scale, float* out, int64_t n_pix, int C, int dtype, sv_stream_t stream);
/* Per-sample reconstruction error of the evaluation after training (run_model.py:356-362:
 * tf.reduce_mean(losses.mse(reconstructions, data), axis=(1, 2))): out[n] = mean over the sample_elems elements of
 * sample n of (y_pred - y_true)^2.  y_true and out are f32, y_pred has the storage type `dtype`. */
int sv_sample_mse(const float* y_true, const void* y_pred, float* out, int64_t n_samples, int64_t sample_elems,
                  int dtype, sv_stream_t stream);
/* Pair counts of the ROC AUC (run_model.py:390-398: sklearn.metrics.roc_curve + auc with y_true = 1 for the first score
 * set and 0 for the second, y_pred = the score): counts[0] = #{(i, j): s1[i] > s0[j]}, counts[1] = #{(i, j): s1[i] ==
 * s0[j]}, so AUC = (counts[0] + counts[1] / 2) / (n1 n0).  counts is two uint64 on the device, cleared by the call. */
int sv_roc_auc_counts(const float* scores_label1, int64_t n1, const float* scores_label0, int64_t n0,
                      uint64_t* counts, sv_stream_t stream);
/* MeanAbsoluteError (run_unet.py:147) */
int sv_loss_mae(const void* y_true, const void* y_pred, float* loss_out, int accumulate, void* dpred,
                float loss_scale, int64_t n, float* partials, int dtype, sv_stream_t stream);
/* BinaryCrossentropy against a constant label (run_context_encoder.py:165-187: from_logits=False, clip 1e-7;
 * run_unet.py:84-90,148-154: from_logits=True) */
int sv_loss_bce(const void* pred, float label, int from_logits, float* loss_out, int accumulate, void* dpred,
                float loss_scale, int64_t n, float* partials, int dtype, sv_stream_t stream);

/* ---- optimizer: tf.keras.optimizers.legacy.Adam + ExponentialDecay(staircase)
 *      (run_model.py:77-89; run_unet.py:48-66; run_context_encoder.py:144-162) over one flat buffer.
 *      step_state is a 16-byte device buffer: int64 count of updates applied so far, then two floats of
 *      scratch.  sv_adam_keras_prepare reads the count `it`, stores
 *        lr = lr0 * decay_rate^floor(it/decay_steps) (decay_steps <= 0: constant), t = it+1,
 *        alpha = lr*sqrt(1-b2^t)/(1-b1^t)
 *      (in double) into the scratch and increments the count, so a captured graph replays the schedule with
 *      no host input.  sv_adam_keras_apply then does, per element,
 *        g *= grad_scale; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; theta -= alpha m/(sqrt(v)+eps)
 *      (grad_scale = 1/world_size after a sum all-reduce).  zero_grad != 0 also clears grad once it has been read, so
 *      the next backward pass of a training loop (which accumulates) needs no separate fill.  Buffers must be 16-byte
 *      aligned. */
/* Both steps in one launch (every thread derives the step size from the count, the workgroup that finishes last advances
 * it): what a training step uses.  step_state is 24 bytes here: the 16 of the two-launch form (same contents afterwards)
 * and a third word holding an arrival counter that must be zero before the first launch (a zero-filled state is) and is
 * left zero by every launch.
 * theta_bf16 (may be NULL): a bf16 copy of theta written in the same pass, for models whose convs read bf16 filters.
 * fold_partials (may be NULL): a pending scalar fold carried by one workgroup of this launch:
 * fold_out[0] = fold_scale * sum(fold_partials[0..fold_n)) in a fixed order, fold_running[0] += fold_running_weight *
 * fold_out[0] when fold_running != NULL (the epoch's running loss of the fit loop). */
int sv_adam_keras_step(float* theta, float* grad, float* m, float* v, int64_t n, int64_t* step_state, float lr0,
                       float beta1, float beta2, float eps, int decay_steps, float decay_rate, int staircase,
                       float grad_scale, int zero_grad, void* theta_bf16, const float* fold_partials, int fold_n,
                       float fold_scale, float* fold_out, float* fold_running, float fold_running_weight,
                       sv_stream_t stream);
/* sv_adam_keras_step whose gradient of element i is grad[i] + the fixed-order sum of the slab rows that
 * sv_conv2d_wgrad_batch_slabs left for it (dw / db of each description must lie inside grad[0..n); elements no description
 * covers take grad[i] alone): the "sum the partial sums" launch and the optimizer launch of a training step in one.  The
 * sums are formed in the order sv_conv2d_wgrad_batch(reduce = 1) forms them, so both routes give the same bits.  At most
 * 16 descriptions that do not overlap. */
int sv_adam_keras_step_slabs(float* theta, float* grad, float* m, float* v, int64_t n, int64_t* step_state, float lr0,
                             float beta1, float beta2, float eps, int decay_steps, float decay_rate, int staircase,
                             float grad_scale, int zero_grad, void* theta_bf16, const float* fold_partials, int fold_n,
                             float fold_scale, float* fold_out, float* fold_running, float fold_running_weight,
                             const sv_grad_slabs* slabs, int n_slabs, sv_stream_t stream);
/* out[0] = [out[0] +] scale * sum(partials[0..n)) in a fixed order (one workgroup); running as above. */
int sv_fold_scalar(const float* partials, int n, float scale, float* out, int accumulate, float* running,
                   float running_weight, sv_stream_t stream);
/* Loss bookkeeping of the GAN steps in one launch (run_context_encoder.py:246-251: disc_loss = 0.5 * (true + fake),
 * total = w * l2 + (1 - w) * disc_loss; run_unet.py:175-187): n_statements (1..4) scalar statements, run in order,
 *   outs[s][0] = [outs[s][0] +] sum_k weights[.] * terms[.][0],   n_terms[s] (0..3) terms each,
 * `terms` / `weights` listing all statements' terms back to back.  A later statement may read an earlier one's result.
 * Each term enters through one fused multiply-add: the same bits as sv_fill_f32 + one sv_axpy_f32 per term. */
int sv_scalar_program(int n_statements, float* const* outs, const int* accumulate, const int* n_terms,
                      const float* const* terms, const float* weights, sv_stream_t stream);
int sv_adam_keras_prepare(int64_t* step_state, float lr0, float beta1, float beta2, int decay_steps,
                          float decay_rate, int staircase, sv_stream_t stream);
int sv_adam_keras_apply(float* theta, float* grad, float* m, float* v, int64_t n, const int64_t* step_state,
                        float beta1, float beta2, float eps, float grad_scale, int zero_grad, sv_stream_t stream);

/* ---- deferred reductions (no reference counterpart; launch-count control for the backward pass of the GAN steps,
 *      run_context_encoder.py:225-260, run_unet.py:164-203).  Between _begin and _flush on one host thread, every
 *      "sum the partial slabs into dw / db" launch inside sv_conv2d_wgrad[_ws], sv_conv2d_transpose_wgrad, sv_act_bwd and
 *      sv_bias_grad is queued instead; _flush(launch = 1) runs all queued sums in ONE launch on `stream` (the stream
 *      the producers ran on), _flush(launch = 0) drops them (error paths).  While a reduction is queued its workspace
 *      must not be reused: give every call in the bracket its own workspace.  Results are the same fixed-order sums. */
int sv_deferred_reduce_begin(void);
int sv_deferred_reduce_flush(int launch, sv_stream_t stream);

/* ---- batching helpers for Model.fit / predict (run_model.py:141-146,283-289,300):
 *      dst[i, :] = src[idx[i], :] with rows of row_elems elements; src may be f64 or f32 (src_is_f64), dst is f32:
 *      Keras casts the float64 arrays np.zeros/np.concatenate produce (run_model.py:23,110,245) to float32. */
int sv_gather_rows(const void* src, int src_is_f64, const int64_t* idx, float* dst, int64_t n_rows,
                   int64_t row_elems, sv_stream_t stream);
/* Random block masks generated on the device (batch_random_mask / create_random_image_mask,
 * run_context_encoder.py:24-58): per image the block [seg, 3 seg)^2 with seg = W / 8, then seg x seg blocks at uniform
 * positions in [0, W - seg)^2 until at least pct_mask of W^2 pixels are 1; mask is [N][H][W][C] f32, equal over C.
 * The reference draws from NumPy's unseeded global generator; here draw d of image n is a counter hash of
 * (seed, offset + (n << 20) + 2d [+ 1]), so a mask is a pure function of the arguments. */
int sv_random_block_mask(float* mask, int N, int H, int W, int C, double pct_mask, uint64_t seed, uint64_t offset,
                         sv_stream_t stream);
/* The masked input of the context encoder folded into its first convolution (run_context_encoder.py:242-243:
 * `X_masked = X * (1 - mask)`, `pred = context_encoder(X_masked)`): the forward conv and its kernel gradient read x as
 * x * (1 - mask), mask laid out like x, while they gather it; the masked tensor never exists in memory.  Same results as
 * sv_mask_input followed by sv_conv2d_fwd_ws / sv_conv2d_wgrad, bit for bit.  Geometries of the implicit-GEMM family
 * only (sv_conv2d_masked_supported; strided convs with a reduction of >= 16 terms), f32. */
int sv_conv2d_masked_supported(const sv_conv_geom* g);
int sv_conv2d_fwd_masked(const sv_conv_geom* g, const void* x, const void* mask, const void* w, const void* bias, void* y,
                         int act, float alpha, void* workspace, size_t workspace_bytes, int dtype, sv_stream_t stream);
int sv_conv2d_wgrad_masked(const sv_conv_geom* g, const void* x, const void* mask, const void* dz, void* dw, int accumulate,
                           void* workspace, size_t workspace_bytes, int dtype, sv_stream_t stream);
/* y = x * (1 - mask) (run_context_encoder.py:242) */
int sv_mask_input(const float* x, const float* mask, float* y, int64_t n, sv_stream_t stream);
/* the same with y in the storage type `dtype` (the masked input of a bf16 model) */
int sv_mask_input_dt(const float* x, const float* mask, void* y, int64_t n, int dtype, sv_stream_t stream);
int sv_fill_f32(float* p, float value, int64_t n, sv_stream_t stream);
int sv_axpy_f32(float* y, const float* x, float a, int64_t n, sv_stream_t stream); /* y += a*x */
/* y += a*x on activation-typed tensors (the sum of the gradients arriving over a skip connection, unet.py:144, and of the
 * two loss terms' gradients in run_unet.py:164-203) */
int sv_axpy(void* y, const void* x, float a, int64_t n, int dtype, sv_stream_t stream);
int sv_cast_f32_bf16(const float* x, void* y, int64_t n, sv_stream_t stream);
int sv_cast_bf16_f32(const void* x, float* y, int64_t n, sv_stream_t stream);

/* ---- data preparation, the step upstream of the models (generator.py:92-98 builds every training image with it):
 *      util.process_region (util.py:152-225) with inter_snp_distances (util.py:34-45) and major_minor (util.py:228-256)
 *      for a whole batch of ragged regions.  x holds the regions' (n_sites, n_haps, n_mut) genotype arrays back to back,
 *      positions their int64 site positions, site_offsets[r] .. site_offsets[r + 1] the sites of region r (every region
 *      needs at least one site).  out[r] is the (n_haps, n_snps, n_channels) float32 image: channel 0 the centre-cropped /
 *      zero-padded, minor-allele-polarised haplotype matrix, the last channel the inter-SNP distances scaled to [-1, 1]
 *      (float64 arithmetic in the reference's operation order, rounded to float32 on the store), other channels 0.
 *      distances (optional, [total sites], float64) receives each region's full normalised distance vector.
 *      status[r] = 1 where the reference's `assert np.max(X) == 1` (util.py:198) would fail.  n_snps even,
 *      n_haps * n_snps <= 8192.  Bit-identical to the NumPy path. */
int sv_process_regions(const float* x, const int64_t* positions, const int64_t* site_offsets, int64_t n_regions,
                       int n_haps, int n_mut, int n_snps, int n_channels, int64_t norm_len, float* out, double* distances,
                       int32_t* status, sv_stream_t stream);
/* util.sort_min_diff (util.py:48-58) for a batch: order[r] = the haplotype rows of image r sorted by Manhattan distance
 * (negative entries clamped to 0, util.py:54) from the row whose summed distance to all rows is smallest (first such
 * row); equal distances keep row order (the reference inherits an unspecified tie order from scikit-learn).
 * images: [n_images][n_haps][n_snps][n_channels], the matrix is channel `channel`; n_haps <= 64. */
int sv_sort_min_diff(const float* images, int64_t n_images, int n_haps, int n_snps, int n_channels, int channel,
                     int64_t* order, sv_stream_t stream);
/* out[r][k] = images[r][order[r][k]] with rows of row_elems floats: sort_batch (run_model.py:22-29), batch_sort
 * (run_context_encoder.py:101-112).  Not in place. */
int sv_reorder_rows(const float* images, const int64_t* order, float* out, int64_t n_images, int n_rows, int row_elems,
                    sv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SPECTRUM_HIP_H */
