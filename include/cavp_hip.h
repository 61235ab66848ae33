/*
 * cavp_hip.h — C-ABI of libcavp_hip.so: the MI355X (gfx950) kernels behind the CAVP forward hot path.
 *
 * The reference (cyh-0/CAVP) has no FFI / plugin layer: its "operators" are stock torch.nn modules called from
 * models/cavp_model.py::CAVP.forward.  Each entry point below therefore names the reference call site(s) whose
 * arithmetic it replaces (file:line relative to the reference repo).  The Python host in cavp_amd/ binds these
 * with ctypes and keeps the reference's nn.Module contract; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain pointers + sizes; no torch / C++ types.  All pointers are DEVICE pointers unless stated.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Kernels are launched asynchronously;
 *     nothing here allocates, frees or synchronises, so every call is hipGraph-capturable.
 *   - activations are NHWC ("channels-last"): element (n,h,w,c) of a tensor with pixel stride `ld` lives at
 *     ((n*H + h)*W + w)*ld + c.  `ld >= C` lets a kernel read/write a channel slice of a wider tensor
 *     (that is how torch.cat along C is done without a copy: encoder_decoder.py:104,139).
 *   - conv / linear weights are "OHWI": [Cout][KH][KW][Cin], i.e. the GEMM K dimension is contiguous.
 *   - dtype: CAVP_F32 (parity path, exact-f32 MFMA) or CAVP_BF16 (bf16 storage, f32 accumulate).
 *   - return value: 0 on success, negative cavp_status_t otherwise (see cavp_error_string).
 */
#ifndef CAVP_HIP_H_
#define CAVP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CAVP_ABI_VERSION 12

typedef enum { CAVP_F32 = 0, CAVP_BF16 = 1 } cavp_dtype_t;
typedef enum { CAVP_ACT_NONE = 0, CAVP_ACT_RELU = 1, CAVP_ACT_LEAKY = 2, CAVP_ACT_GELU = 3 } cavp_act_t;
typedef enum {
  CAVP_OK = 0,
  CAVP_ERR_BAD_ARG = -1,      /* null pointer / non-positive size */
  CAVP_ERR_UNSUPPORTED = -2,  /* shape the kernels do not cover (e.g. Cin not a multiple of the 16-byte vector) */
  CAVP_ERR_ALIGN = -3,        /* pointer or leading dimension not 16-byte aligned where required */
  CAVP_ERR_WORKSPACE = -4,    /* workspace too small */
  CAVP_ERR_LAUNCH = -5        /* hipLaunch failed (hipGetLastError != hipSuccess) */
} cavp_status_t;

int cavp_abi_version(void);

/* Clear nranges [start, end) element ranges (table_dev: int64 pairs in device memory; 4-float aligned) of one f32 buffer in one
 * launch: the per-step reset of the flat gradient arena.  max_len = the longest range (sizes the grid). */
int cavp_zero_ranges_f32(float* base, const int64_t* table_dev, int32_t nranges, int64_t max_len, void* stream);
/* ABI 11: clear nbytes (a multiple of 4, 4-byte aligned) of device memory: the training step's scratch pools and zero-initialised
 * gradients without a framework fill kernel on the path. */
int cavp_zero_bytes(void* p, size_t nbytes, void* stream);
/* ABI 11: *(int64_t*)table_dev[i] += inc for the n device addresses in table_dev (device memory): nn.BatchNorm2d's
 * num_batches_tracked counters of all layers (resnet.py:75-98 in train mode) in one launch. */
int cavp_i64_add_table(const int64_t* table_dev, int32_t n, int64_t inc, void* stream);
/* Opt-in deterministic training (the reference runs with cudnn.deterministic = True, main_vpo_mono.py:39-41).  With a scratch
 * buffer registered (>= 1 MiB, 16-byte aligned device memory that stays alive; 8 MiB covers every CAVP shape) the reductions
 * that otherwise finish with f32 atomics - cavp_colsum / cavp_colstats / cavp_bn_act_bwd_reduce, the dgamma / dbeta of
 * cavp_layernorm_bwd and the dk / dv of cavp_attn_gate_bwd - write per-workgroup partials and add them in a fixed order:
 * a training step is then bit-reproducible run to run.  scratch = NULL switches the mode off.  Process-wide (one process per
 * GPU); launches that need more scratch than registered return CAVP_ERR_WORKSPACE. */
int cavp_set_deterministic(void* scratch, size_t bytes);
int cavp_get_deterministic(void);
const char* cavp_error_string(int status);

/* ---------------------------------------------------------------------------------------------------------
 * Fused conv / linear: y = act( (conv(x, w) + nbias[n, :]) * scale + shift + residual )
 *
 * Replaces every nn.Conv2d(+BatchNorm2d eval)(+ReLU/LeakyReLU)(+residual add) group and every nn.Linear(+GELU /
 * ReLU) on the path:  resnet.py:75-98,107-121,186-190 (bottlenecks, stem convs 2-3) · encoder_decoder.py:62-75
 * (decoder head), :97-105 (reduce), :137-156 (ASPP) · vgg.py:17-36 (audio convs + FCs) · cavp_model.py:123-128,146
 * (projector Mlp) · attn.py:30-39,64-71,103-105,136-143 (patch_embed, q/k/v/proj, Mlp).
 *   scale/shift : per-Cout f32 (folded BN: scale = gamma/sqrt(var+eps), shift = beta - mean*scale; or bias).
 *                 scale == NULL means 1, shift == NULL means 0.
 *   nbias       : optional f32 [N][Cout], added per image before scale (ASPP pooled branch, :150-153).
 *   residual    : optional, same dtype as y, pixel stride ldr (Bottleneck `out += residual`, resnet.py:92-96;
 *                 attention residuals attn.py:148-149).
 *   A linear layer over T tokens is the 1x1 case with N=1, H=1, W=T.
 * Implicit GEMM on MFMA; taps that can never be in-bounds (dilation > extent) are skipped.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct cavp_conv_desc {
  int32_t dtype;      /* cavp_dtype_t of x, w, residual and y */
  int32_t N, H, W;    /* input batch / height / width */
  int32_t Cin, ldx;   /* input channels, input pixel stride (elements) */
  int32_t Cout, ldy;  /* output channels, output pixel stride */
  int32_t KH, KW, stride, pad, dil;
  int32_t ldr;        /* residual pixel stride (ignored when residual == NULL) */
  int32_t act;        /* cavp_act_t */
  int32_t splitk;     /* 0 = let the library choose; >= 1 forces that many K slices */
  int32_t tile;       /* 0 = auto; otherwise id (1..9) + profiling digits (tools/bench_conv.py): testing / tuning only */
  int32_t up;         /* 0/1 = ordinary conv.  up = s > 1 (power of two): x is read as if zero-upsampled by s, i.e. the
                         data-gradient of a stride-s conv (transposed conv); Ho/Wo below then give the output size */
  int32_t Ho, Wo;     /* only read when up > 1 (the forward conv's input extent) */
  int32_t stride_w;   /* 0 = same as `stride`; otherwise the horizontal stride (PVT spatial-reduction convs are run as
                         KH = sr, KW = 1 convs over the input viewed as [N][H][W/sr][sr*C] with stride (sr, 1)) */
  int32_t dw_oihw;    /* cavp_conv2d_wgrad_nhwc only: 1 = accumulate into a torch-layout [Cout][Cin][KH][KW] gradient (default 0: OHWI) */
  int32_t dw_overwrite; /* cavp_conv2d_wgrad_nhwc only: 1 = dw = gradient (beta = 0: dw is neither read nor required to be zeroed;
                           dead taps of a dilated kernel are written as zeros); default 0: dw += gradient */
  /* ---- ABI 6: epilogue fusions of the token path (cavp_conv2d_nhwc_aux) ---- */
  int32_t res_rows;   /* 0 = the residual has one row per output pixel; > 0: it has res_rows pixel rows and output pixel p
                         adds row p mod res_rows - the un-duplicated half of forward_train's `torch.cat((x, x.clone()))`
                         (cavp_model.py:181) read twice instead of copied (must be a multiple of 256 rows) */
  int32_t aux_mode;   /* 0 = none.  1 (act must be CAVP_ACT_GELU): besides y = gelu(t) the epilogue stores gelu'(t) into
                         `aux` (y's shape, stride ld_aux) - what the backward of timm's Mlp needs (attn.py:136-150,
                         cavp_model.py:123-128), so the pre-activation is never written.  2: the result is multiplied
                         element-wise by `aux` before the residual is added: d(pre) = d(hidden) * gelu'(pre) fused into
                         the data-gradient GEMM that produces d(hidden) */
  int32_t ld_aux;     /* pixel stride of aux (elements) */
} cavp_conv_desc;

size_t cavp_conv2d_workspace_bytes(const cavp_conv_desc* d);
/* tile_stats (optional, training): f32 [tiles][Cout][2]; the epilogue stores each pixel tile's per-channel mean and
 * centred second moment of the raw conv output, so BatchNorm batch statistics cost no extra pass (combine with
 * cavp_bn_finalize_tiles).  Only for plain convs (no scale/shift/nbias/residual/act) whose launch uses the staged
 * epilogue: query cavp_conv2d_tile_stats_layout first (returns 0 -> fall back to cavp_colsum + cavp_colstats). */
int cavp_conv2d_tile_stats_layout(const cavp_conv_desc* d, int32_t* tiles, int32_t* rows_per_tile);
int cavp_conv2d_nhwc(const cavp_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift,
                     const float* nbias, const void* residual, void* y, void* workspace, size_t workspace_bytes,
                     float* tile_stats, void* stream);
/* The same with the auxiliary epilogue tensor of d->aux_mode (dtype of y; written for mode 1, read for mode 2).  Launches
 * that use res_rows / aux_mode need the 16-byte epilogue (Cout, ldy, ldr, ld_aux multiples of 8 bf16 / 4 f32 elements,
 * 16-byte aligned pointers) and are never split over K: CAVP_ERR_UNSUPPORTED otherwise. */
int cavp_conv2d_nhwc_aux(const cavp_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift,
                         const float* nbias, const void* residual, void* y, void* aux, void* workspace,
                         size_t workspace_bytes, float* tile_stats, void* stream);
/* ABI 10: BatchNorm-backward statistics fused into the data-gradient launch that PRODUCES the gradient of a BatchNorm + activation
 * output.  torch.autograd runs ReLU.backward and BatchNorm.backward as passes of their own behind every conv's backward
 * (resnet.py:75-98 under trainer_cavp_vpo_mono.py:190 `loss.backward()`); here the conv launch (a data gradient: d = the
 * transposed conv, see `up`) finishes v = conv(x, w) + residual, multiplies by the activation's derivative, stores
 *   g = v * act'(.)                      (act' from `out`, the forward's activation output, or re-derived from z*fwd_scale + fwd_shift)
 * and writes, per pixel tile, partials[tile][c] = (sum g, sum g * (z - mean) * rstd) over the tile's rows - the two reductions of
 * cavp_bn_act_bwd_reduce, which then only have to be summed over the tiles (cavp_bn_bwd_sum_tiles; fixed order, no atomics).
 * cavp_bn_act_bwd_apply runs on g with act = NONE.  cavp_conv2d_bnbwd_layout returns 0 when the launch cannot carry the
 * statistics (split over K, 256 x 256 tile, unaligned operands): the caller keeps the separate reduce. */
typedef struct cavp_bnbwd_args {
  const void* z;            /* the BatchNorm's input (dtype and shape of y), pixel stride ld_z */
  const void* out;          /* the activation's output (BN + residual + act), pixel stride ld_out; NULL: mask from z, fwd_scale, fwd_shift */
  int32_t ld_z, ld_out;
  const float* fwd_scale;   /* the forward's folded scale / shift (only read when out == NULL and act != NONE) */
  const float* fwd_shift;
  const float* mean;        /* batch mean and 1 / sqrt(var + eps) of z */
  const float* rstd;
  int32_t act;              /* CAVP_ACT_NONE / RELU / LEAKY */
  int32_t pad_;
  float* partials;          /* f32 [tiles][Cout][2] (cavp_conv2d_bnbwd_layout): per pixel tile (sum g, sum g * zhat), summed by cavp_bn_bwd_sum_tiles */
} cavp_bnbwd_args;   /* (ABI 11: the f32-atomic route of ABI 10 - sum_g / sum_gz - is gone: measured slower, profiles/r05_notes.md 3) */
int cavp_conv2d_bnbwd_layout(const cavp_conv_desc* d, int32_t* tiles, int32_t* rows_per_tile);
int cavp_conv2d_nhwc_bnbwd(const cavp_conv_desc* d, const void* x, const void* w, const void* residual, void* y,
                           const cavp_bnbwd_args* b, void* workspace, size_t workspace_bytes, void* stream);
/* sum_g[c] += sum_t partials[t][c][0], sum_gz[c] += sum_t partials[t][c][1] (ascending tiles within 16 interleaved lanes, then a
 * fixed tree: deterministic). */
int cavp_bn_bwd_sum_tiles(const float* partials, int32_t tiles, int32_t C, float* sum_g, float* sum_gz, void* stream);
/* cavp_bn_act_bwd_apply (below) that also adds the two sums to the BatchNorm's affine gradients (dbeta_acc += sum_g, dgamma_acc +=
 * sum_gz; both or neither): for sums that arrive in scratch memory (the atomic route of cavp_conv2d_nhwc_bnbwd). */
int cavp_bn_act_bwd_apply_acc(int32_t dtype, const void* dy, const void* y, const void* z, const float* mean, const float* rstd,
                              const float* gamma, const float* sum_g, const float* sum_gz, int64_t rows, int32_t C, int32_t ld_dy,
                              int32_t ld_y, int32_t ld_z, int32_t act, void* dz, int32_t ld_dz, void* g_out, int32_t ld_g,
                              const float* fwd_scale, const float* fwd_shift, float* dbeta_acc, float* dgamma_acc, void* stream);
/* An automatically planned launch of the 256x256 tile whose last round of 256 tiles is at most a quarter full (the decoder head
 * convs, encoder_decoder.py:62-75, at 2B x 56 x 56: 784 tiles = 3.06 rounds) is issued as the leading images on the 256x256 tile
 * + the remaining images on the small tiles.  Process-wide switch for A/B runs and tests (default on); results are identical
 * either way up to the summation order inside the tail images. */
int cavp_set_tail_split(int32_t on);
/* ABI 9: the 256 x 256 weight-gradient tile (csrc/conv_wgrad_big.hip; bf16: one workgroup per CU, four-stage LDS-DMA
 * ring, v_mfma_f32_32x32x16_bf16) for the weight gradients of encoder_decoder.py:62-75 (decoder head), models/attn.py:136-143 and
 * cavp_model.py:123-128 (token / projector Mlp) - the jobs with >= 16384 pixel rows and >= 192 input and output channels.
 * mode: 0 (default) = those jobs, 1 = never (the 128 x 128 tile everywhere), 2 = every bf16 job (tests).  schedule: 2 (default) =
 * sixteen waves per workgroup (64 x 64 wave tiles, four waves per SIMD, fragments double-buffered by k step), 1 = eight waves
 * (64 x 128 wave tiles) with every LDS read and DMA issue in the gaps between the MFMAs, 0 = eight waves: read, multiply, retire,
 * fetch.  Same products, same summation order in all three.  The choice of tile depends on the job alone, so grouped and single
 * launches of a job with one split count stay bit-identical.  Process-wide switch for A/B runs and tests. */
int cavp_set_wgrad_big(int32_t mode, int32_t schedule);

/* Direct 3x3 conv for Cin in {1,2,3} reading an NCHW f32 tensor and writing NHWC (dtype) with scale/shift + act:
 * the ResNet deep-stem first conv (resnet.py:108-110, stride 2) and the first VGGish conv (vgg.py:26-36). */
int cavp_conv3x3_smallcin_nchw(int32_t dtype, const float* x_nchw, const float* w_oihw, const float* scale,
                               const float* shift, void* y_nhwc, int32_t N, int32_t Cin, int32_t H, int32_t W,
                               int32_t Cout, int32_t stride, int32_t act, void* stream);

/* nn.MaxPool2d(k, s, p) on NHWC (resnet.py:139,190: 3/2/1; vgg.py:30: 2/2/0).  argmax (optional, u8 [N][Ho][Wo][C]):
 * window-relative position kh*k + kw of the first maximum in scan order (the element ATen routes the gradient to),
 * consumed by cavp_maxpool_bwd_nhwc. */
int cavp_maxpool_nhwc(int32_t dtype, const void* x, void* y, uint8_t* argmax, int32_t N, int32_t H, int32_t W,
                      int32_t C, int32_t k, int32_t stride, int32_t pad, void* stream);

/* ABI 12.  The same pool over act(x * scale + shift) (per-channel f32 scale / shift, rounded to dtype before the comparison): the training step's
 * BatchNorm apply + ReLU in front of the stem pool (resnet.py:187-190: bn1 -> relu -> maxpool) without the activation tensor in memory.
 * Values and arg-max are bit-identical to cavp_scale_shift_act followed by cavp_maxpool_nhwc. */
int cavp_maxpool_affine_nhwc(int32_t dtype, const void* x, const float* scale, const float* shift, int32_t act, void* y,
                             uint8_t* argmax, int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t pad,
                             void* stream);

/* x.view(N, C, -1).mean(-1) for NHWC x; output f32 [N][C] (ASPP._global_pooling, encoder_decoder.py:158-161). */
int cavp_global_avgpool_nhwc(int32_t dtype, const void* x, float* y, int32_t N, int32_t HW, int32_t C, int32_t ldx,
                             void* stream);

/* F.interpolate(mode="bilinear") NHWC -> NHWC channel slice (encoder_decoder.py:103, align_corners=True). */
int cavp_bilinear_nhwc(int32_t dtype, const void* x, void* y, int32_t N, int32_t Hi, int32_t Wi, int32_t C,
                       int32_t ldx, int32_t Ho, int32_t Wo, int32_t ldy, int32_t align_corners, void* stream);

/* Final F.interpolate(..., align_corners=False) of the logits: NHWC (dtype) in, NCHW f32 out (cavp_model.py:140). */
int cavp_bilinear_nhwc_to_nchw(int32_t dtype, const void* x, float* y_nchw, int32_t N, int32_t Hi, int32_t Wi,
                               int32_t C, int32_t ldx, int32_t Ho, int32_t Wo, int32_t align_corners, void* stream);

/* nn.LayerNorm(C, eps) over the last dim of [rows][C] (attn.py:130,136,229 -> :154-155,149,242). */
int cavp_layernorm(int32_t dtype, const void* x, const float* gamma, const float* beta, void* y, int32_t rows,
                   int32_t C, int32_t ldx, int32_t ldy, float eps, void* stream);
/* residual + DropPath + LayerNorm of a transformer block's stream in one pass (pvt.py:252-256 with timm's drop_path):
 * y_sum = x + row_scale[row / rows_per_group] * branch (dense [rows][C], x's dtype), y = LayerNorm(y_sum).  branch = NULL: plain
 * cavp_layernorm. */
int cavp_layernorm_residual(int32_t dtype, const void* x, const void* branch, const float* row_scale, int32_t rows_per_group,
                            const float* gamma, const float* beta, void* y_sum, void* y, int32_t rows, int32_t C, int32_t ldx,
                            int32_t ldy, float eps, void* stream);

/* Sigmoid-gated single-key attention (attn.py:73-106 with N_kv == 1):
 *   s[b,h,t] = sigmoid(scale * <q[b,t,h,:], k[b,h,:]>);  o[b,t,h,:] = s[b,h,t] * v[b,h,:];  attn[b,h,t] = s.
 * q: [q_batch][T][heads*hd], o: [B][T][heads*hd] (dtype); k,v: [B][heads*hd] (dtype); attn: f32 [B][heads][T].
 * q_batch = B, or a divisor of B: batch item b reads q[b mod q_batch] (forward_train runs the query projection once on the B
 * images and gates it with the 2B audio clips, cavp_model.py:181). */
int cavp_attn_gate(int32_t dtype, const void* q, const void* k, const void* v, void* o, float* attn, int32_t B,
                   int32_t T, int32_t heads, int32_t hd, float scale, int32_t q_batch, void* stream);

/* BatchNorm (eval) folding: scale = gamma * rsqrt(var + eps), shift = beta - mean * scale (f32, C entries). */
int cavp_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* scale,
                 float* shift, int32_t C, void* stream);

/* Weight packing: OIHW f32 (torch layout) -> OHWI (dtype).  Linear weights are the KH=KW=1 case (pure cast). */
int cavp_pack_weight_ohwi(int32_t dtype, const float* w_oihw, void* w_ohwi, int32_t Cout, int32_t Cin, int32_t KH,
                          int32_t KW, void* stream);

/* Element-wise cast between f32 and dtype (n elements): src_dtype -> dst_dtype. */
int cavp_cast(int32_t src_dtype, const void* src, int32_t dst_dtype, void* dst, int64_t n, void* stream);

/* =========================================================================================================
 * Training side: batch-statistics BatchNorm and the backward kernels.  The reference gets all
 * of this from torch.autograd over the modules cited above (trainer_cavp_vpo_mono.py:168-193 `loss.backward()`).
 * Data gradients of conv / linear reuse cavp_conv2d_nhwc with cavp_pack_weight_dgrad weights (+ `up` for strides).
 * ======================================================================================================= */

/* Weight gradient of a conv / linear: dw[co][kh][kw][ci] += sum_pixels dy[pix][co] * x[pix @ tap][ci] (f32, OHWI; the
 * result is ADDED to dw).  Fields of `d` describe the FORWARD conv (x is its input, dy its output gradient with pixel
 * stride d->ldy).  The pixel reduction is split over workgroups; partial slabs go to `workspace`
 * (cavp_conv2d_wgrad_workspace_bytes) and are reduced in a fixed order: deterministic, no atomics.
 * dbias (optional, f32 [Cout]): dbias[co] += sum_pixels dy[pix][co], the bias gradient, summed from the dY tiles the kernel
 * streams anyway (saves the separate column-sum pass over dY). */
size_t cavp_conv2d_wgrad_workspace_bytes(const cavp_conv_desc* d);
int cavp_conv2d_wgrad_nhwc(const cavp_conv_desc* d, const void* x, const void* dy, float* dw_ohwi, float* dbias,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ABI 7: up to CAVP_WGRAD_GROUP_MAX independent weight gradients (the jobs of cavp_conv2d_wgrad_nhwc, one dtype) in ONE launch
 * (+ one launch for the slab reduces of the jobs that split their pixel range).  torch.autograd computes every weight
 * gradient where its layer's backward runs (trainer_cavp_vpo_mono.py:168-193 `loss.backward()`); nothing in the backward
 * chain consumes them, so the host defers the weight gradients of a stage of small layers (ResNet layer3: 19 convs on
 * 32 x 14 x 14 pixels, resnet.py:75-98; the 52 PVTv2 blocks, pvt.py:137-170) and issues them together: the launches fill the
 * chip with far fewer pixel splits (less slab traffic, one reduce instead of 19).  `jobs` is a HOST array; the job table
 * travels as kernel arguments.  Two jobs must not share a dw or dbias (CAVP_ERR_BAD_ARG).  The split reduction stays a
 * fixed-order slab sum: deterministic. */
#define CAVP_WGRAD_GROUP_MAX 16
typedef struct cavp_wgrad_job {
  cavp_conv_desc desc;   /* the FORWARD conv, as for cavp_conv2d_wgrad_nhwc (dw_oihw / dw_overwrite / splitk honoured per job) */
  const void* x;
  const void* dy;
  float* dw;
  float* dbias;          /* optional */
} cavp_wgrad_job;
size_t cavp_conv2d_wgrad_group_workspace_bytes(const cavp_wgrad_job* jobs, int32_t njobs);
int cavp_conv2d_wgrad_group(const cavp_wgrad_job* jobs, int32_t njobs, void* workspace, size_t workspace_bytes, void* stream);

/* OIHW f32 -> [Cin][KH][KW][Cout] (dtype), taps rotated by 180 degrees: the OHWI weight of the transposed conv. */
/* All weight re-packs of one training step in one launch per <= 48 tensors (the per-tensor entry points cost ~170
 * launches of ~8 us each per step, profiles/r01_notes.md): for every job, w_oihw f32 [Cout][Cin][KH][KW] ->
 * ohwi (cavp_pack_weight_ohwi layout, may be NULL) and dgrad (cavp_pack_weight_dgrad layout, may be NULL), both `dtype`.
 * `jobs` is a HOST array; device pointers inside. */
typedef struct cavp_pack_job {
  const float* w_oihw;
  void* ohwi;
  void* dgrad;
  int32_t Cout, Cin, KH, KW;
} cavp_pack_job;
int cavp_pack_weights_multi(int32_t dtype, const cavp_pack_job* jobs, int32_t njobs, void* stream);
int cavp_pack_weight_dgrad(int32_t dtype, const float* w_oihw, void* w_t, int32_t Cout, int32_t Cin, int32_t KH,
                           int32_t KW, void* stream);
/* OHWI f32 gradient -> OIHW f32 (.grad layout); accumulate != 0 adds into the destination. */
int cavp_unpack_weight_grad(const float* g_ohwi, float* g_oihw, int32_t Cout, int32_t Cin, int32_t KH, int32_t KW,
                            int32_t accumulate, void* stream);
/* Weight gradient of cavp_conv3x3_smallcin_nchw: dw_oihw (f32) += im2col(x)^T dy on the MFMA weight-gradient kernel
 * (deterministic).  workspace: cavp_conv3x3_smallcin_wgrad_workspace_bytes(...) bytes, 16-byte aligned. */
size_t cavp_conv3x3_smallcin_wgrad_workspace_bytes(int32_t dtype, int32_t N, int32_t Cin, int32_t H, int32_t W,
                                                   int32_t Cout, int32_t stride);
int cavp_conv3x3_smallcin_wgrad(int32_t dtype, const float* x_nchw, const void* dy_nhwc, float* dw_oihw, int32_t N,
                                int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t stride, void* workspace,
                                size_t workspace_bytes, void* stream);

/* nn.BatchNorm2d in training mode (resnet.py / encoder_decoder.py BN layers under model.train()):
 *   cavp_colstats     sum[c] += sum_rows (x - s_c), sumsq[c] += sum_rows (x - s_c)^2 with an optional per-channel
 *                     shift s (f32 atomics; caller zeroes).  Two passes (s = 0, then s = mean) give a cancellation-free
 *                     variance, which matters for the tiny-M BatchNorms (ASPP pooled branch: M = batch size).
 *   cavp_bn_finalize  mean = s + sum/M, var = sumsq/M - (sum/M)^2 -> scale = gamma*rstd, shift = beta - mean*scale;
 *                     saves mean, rstd; updates the running
 *                     statistics in place (momentum, unbiased variance) when running_mean/var are non-NULL
 *   cavp_scale_shift_act  y = act(x*scale + shift + residual)
 *   cavp_bn_act_bwd_reduce / _apply  g = dy*act'(y); dbeta = sum g; dgamma = sum g*zhat;
 *                     dz = gamma*rstd*(g - dbeta/M - zhat*dgamma/M); optional g_out = g (skip-path gradient) */
int cavp_colstats(int32_t dtype, const void* x, const float* shift, int64_t rows, int32_t C, int32_t ldx, float* sum,
                  float* sumsq, void* stream);
int cavp_scale_f32(const float* in, float alpha, float* out, int32_t n, void* stream); /* out = alpha * in */
int cavp_bn_finalize(const float* sum, const float* sumsq, const float* stat_shift, int64_t count, const float* gamma,
                     const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                     float* mean, float* rstd, int32_t C, void* stream);
int cavp_bn_finalize_tiles(const float* tile_stats, int32_t tiles, int32_t rows_per_tile, int64_t count,
                           const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                           float* running_var, float* scale, float* shift, float* mean, float* rstd, int32_t C,
                           void* stream);
/* SyncBatchNorm (main_vpo_mono.py:130): this rank's per-channel (mean, M2) from its tile statistics, f32 [C][2].  The ranks'
 * moments are all-gathered into [world][C][2] and combined by cavp_bn_finalize_tiles(tiles = world, rows_per_tile = the
 * per-rank row count): ONE collective per BatchNorm layer in the forward. */
int cavp_bn_tiles_to_moments(const float* tile_stats, int32_t tiles, int32_t rows_per_tile, int64_t count, float* moments,
                             int32_t C, void* stream);
int cavp_scale_shift_act(int32_t dtype, const void* x, const float* scale, const float* shift, const void* residual,
                         void* y, int64_t rows, int32_t C, int32_t ldx, int32_t ldr, int32_t ldy, int32_t act,
                         void* stream);
/* y may be NULL for a BN + activation WITHOUT residual: the activation mask is then re-derived from z with the forward's
 * folded fwd_scale / fwd_shift (y = act(z*scale + shift) > 0 <=> z*scale + shift > 0), which saves reading y in both passes */
int cavp_bn_act_bwd_reduce(int32_t dtype, const void* dy, const void* y, const void* z, const float* mean,
                           const float* rstd, int64_t rows, int32_t C, int32_t ld_dy, int32_t ld_y, int32_t ld_z,
                           int32_t act, float* sum_g, float* sum_gz, const float* fwd_scale, const float* fwd_shift,
                           void* stream);
int cavp_bn_act_bwd_apply(int32_t dtype, const void* dy, const void* y, const void* z, const float* mean,
                          const float* rstd, const float* gamma, const float* sum_g, const float* sum_gz, int64_t rows,
                          int32_t C, int32_t ld_dy, int32_t ld_y, int32_t ld_z, int32_t act, void* dz, int32_t ld_dz,
                          void* g_out, int32_t ld_g, const float* fwd_scale, const float* fwd_shift, void* stream);
/* dx = dy * act'(.): ReLU / LeakyReLU from the activation OUTPUT, GELU from the pre-activation (ref). */
int cavp_act_bwd(int32_t dtype, const void* dy, const void* ref, void* dx, int64_t rows, int32_t C, int32_t ld_dy,
                 int32_t ld_ref, int32_t ld_dx, int32_t act, void* stream);
int cavp_add(int32_t dtype, const void* a, const void* b, void* out, int64_t n, void* stream);
/* out[c] += sum_rows x[r][c]  (bias gradients; f32 atomics, caller zeroes) */
int cavp_colsum(int32_t dtype, const void* x, int64_t rows, int32_t C, int32_t ldx, float* out, void* stream);
/* BatchNorm batch statistics of a tensor whose producer could not fuse them: per-tile (mean, M2) of 128-row tiles in one pass,
 * tile_stats f32 [ceil(rows / 128)][C][2] - the layout cavp_bn_finalize_tiles(tiles, rows_per_tile = 128) combines. */
int cavp_col_tile_stats(int32_t dtype, const void* x, int64_t rows, int32_t C, int32_t ldx, float* tile_stats, void* stream);
/* out[g][c] += sum of rows g*rows_per_group .. (g+1)*rows_per_group - 1: the per-image bias gradient of the ASPP pooled branch
 * (encoder_decoder.py:150-154) for all images in one launch.  Deterministic (one owner per output). */
int cavp_colsum_groups(int32_t dtype, const void* x, int32_t groups, int32_t rows_per_group, int32_t C, int32_t ldx, float* out,
                       void* stream);
/* nn.LayerNorm backward; dgamma / dbeta are accumulated with f32 atomics (caller zeroes). */
int cavp_layernorm_bwd(int32_t dtype, const void* dy, const void* x, const float* gamma, void* dx, float* dgamma,
                       float* dbeta, int32_t rows, int32_t C, int32_t ld_dy, int32_t ld_x, int32_t ld_dx, float eps,
                       void* stream);
/* same, with the gradient the input already holds added in: dx = layernorm_bwd(dy) + dx_add (dx_add may be NULL, and may be dx
 * itself).  The residual stream of a transformer block (pvt.py:252-256, attn.py:194-197) reaches the norm's input twice.
 * dx_scaled (optional, dense [rows][C]): a second copy of dx with row r multiplied by row_scale[r / rows_per_group] - the
 * gradient of a DropPath branch (timm drop_path: mask / keep per image) that hangs off the same residual stream. */
int cavp_layernorm_bwd_add(int32_t dtype, const void* dy, const void* x, const float* gamma, const void* dx_add, int32_t ld_add,
                           void* dx, void* dx_scaled, const float* row_scale, int32_t rows_per_group, float* dgamma,
                           float* dbeta, int32_t rows, int32_t C, int32_t ld_dy, int32_t ld_x, int32_t ld_dx, float eps,
                           void* stream);
/* ---- cross-modal attention with ONE key / value token per batch item, collapsed to rank-H operations (ABI 8) ------------------
 * attn.py:73-106 as CAVP calls it (x_k = x_v = the audio token, cavp_model.py:145-149; 4 heads, q / k / v without bias, sigmoid):
 *   q = x Wq^T, s[t,h] = scale q[t,h,:] . k[h,:], g = sigmoid(s), o[t,h,:] = g[t,h] v[h,:], out = x + o Wp^T + bp   (attn.py:153-156)
 * equals, with u[b,h,:] = scale Wq[h-slice,:]^T k[b,h-slice] and p[b,h,:] = Wp[:,h-slice] v[b,h-slice] (H x C per batch item),
 *   g[b,t,h] = sigmoid(x[t,:] . u[b,h,:]),   out[b,t,:] = x[t,:] + bp + sum_h g[b,t,h] p[b,h,:]
 * so the two C x C token GEMMs (and their data / weight gradient GEMMs) become one pass over the tokens per direction.
 * wq, wp: the f32 [C][C] weights of attn.q / attn.proj (nn.Linear layout); k, v: [B][C] in `dtype`; U, P: f32 [B][heads][C].
 * x: [xb][T][C] with xb dividing B (forward_train duplicates the images, cavp_model.py:181: batch item b reads x[b % xb]).
 * Supported: heads == 4, C <= 512, C % 8 == 0 (cavp_attn1_supported); callers keep cavp_attn_gate + two linears otherwise. */
int cavp_attn1_supported(int32_t C, int32_t heads);
int cavp_attn1_prepare(int32_t dtype, const float* wq, const float* wp, const void* k, const void* v, float* U, float* P, int32_t B,
                       int32_t C, int32_t heads, float scale, void* stream);
/* out: [B][T][C] in `dtype`; attn: f32 [B][heads][T] (the reference's attn_v, cavp_model.py:150); bp may be NULL. */
int cavp_attn1_fwd(int32_t dtype, const void* x, const float* U, const float* P, const float* bp, void* out, float* attn, int32_t B,
                   int32_t xb, int32_t T, int32_t C, int32_t heads, void* stream);
/* Backward over the tokens: dx [xb][T][C] = sum over the batch items that share a row of (dout + sum_h ds[t,h] u[b,h,:]) is
 * WRITTEN (not accumulated); dU / dP: f32 [B][heads][C] written; dbp (optional, f32 [C]) accumulated.  All sums in a fixed
 * order (per-workgroup slabs in `workspace`, cavp_attn1_bwd_workspace_bytes). */
size_t cavp_attn1_bwd_workspace_bytes(int32_t B, int32_t xb, int32_t T, int32_t C, int32_t heads);
int cavp_attn1_bwd(int32_t dtype, const void* dout, const void* x, const float* U, const float* P, void* dx, float* dU, float* dP,
                   float* dbp, void* workspace, size_t workspace_bytes, int32_t B, int32_t xb, int32_t T, int32_t C, int32_t heads,
                   void* stream);
/* dwq, dwp (f32 [C][C], nn.Linear layout) accumulated; dk, dv: f32 [B][C] written. */
int cavp_attn1_finish(int32_t dtype, const float* wq, const float* wp, const void* k, const void* v, const float* dU, const float* dP,
                      float* dwq, float* dwp, float* dk, float* dv, int32_t B, int32_t C, int32_t heads, float scale, void* stream);
/* backward of cavp_attn_gate; dk, dv: f32 [B][heads*hd] accumulated with atomics (caller zeroes); dattn optional.
 * q: [q_batch][T][heads*hd] as in the forward; dq: [B][T][heads*hd] (the caller sums the q_batch-periodic parts). */
int cavp_attn_gate_bwd(int32_t dtype, const void* dout, const void* q, const void* k, const void* v, const float* attn,
                       const float* dattn, void* dq, float* dk, float* dv, int32_t B, int32_t T, int32_t heads,
                       int32_t hd, float scale, int32_t q_batch, void* stream);
/* dx [N][H][W][C] = gradient routed to the recorded arg-max of every window (argmax from cavp_maxpool_nhwc) */
int cavp_maxpool_bwd_nhwc(int32_t dtype, const uint8_t* argmax, const void* dy, void* dx, int32_t N, int32_t H, int32_t W,
                          int32_t C, int32_t k, int32_t stride, int32_t pad, void* stream);
int cavp_bilinear_bwd_nhwc(int32_t dtype, const void* dy, void* dx, int32_t N, int32_t Hi, int32_t Wi, int32_t C,
                           int32_t ld_dx, int32_t Ho, int32_t Wo, int32_t ld_dy, int32_t align_corners, void* stream);
/* backward of cavp_bilinear_nhwc_to_nchw: dy NCHW f32 (images >= n_valid carry zero gradient), dx NHWC (dtype) */
int cavp_bilinear_bwd_nchw_to_nhwc(int32_t dtype, const float* dy_nchw, void* dx, int32_t N, int32_t n_valid,
                                   int32_t Hi, int32_t Wi, int32_t C, int32_t ld_dx, int32_t Ho, int32_t Wo,
                                   int32_t align_corners, void* stream);
/* x[n, p, :] += alpha * v[n, :]  (global-average-pool backward) */
int cavp_bcast_add_nhwc(int32_t dtype, void* x, const float* v, float alpha, int32_t N, int32_t HW, int32_t C,
                        int32_t ld, void* stream);
/* nn.CrossEntropyLoss(ignore_index) on NCHW f32 logits of the first n_img images (loss/losser.py:60-62 applied to
 * `out[:B] + out[B:]*0`, trainer_cavp_vpo_mono.py:171,187): loss[0] = mean over valid pixels; dlogits (optional,
 * [n_total][C][HW]) = grad_scale * dloss/dlogits, zero for images >= n_img.  scratch: CAVP_CE_SCRATCH_FLOATS floats
 * (per-workgroup partial sums, combined in a fixed order: the loss is bit-reproducible run to run). */
#define CAVP_CE_SCRATCH_FLOATS (2 + 2 * 1024)
int cavp_ce_loss_nchw(const float* logits, const int64_t* labels, int32_t n_img, int32_t n_total, int32_t C, int64_t HW,
                      int32_t ignore_index, float grad_scale, float* loss, float* dlogits, float* scratch,
                      void* stream);
/* SURVEY.md §8f row f1 - the head of the training step in one op: F.interpolate(bilinear) of the low-resolution logits
 * (models/cavp_model.py:143-146) + CrossEntropyLoss(ignore_index) on `out[:n_img] + out[n_img:]*0`
 * (trainer_cavp_vpo_mono.py:171,187; loss/losser.py:60-62) + the gradient w.r.t. the low-resolution logits, without
 * the [n_total][C][Ho][Wo] f32 prediction or its gradient ever reaching HBM.
 *   lo, dlo: [n_total][Hi][Wi][ld] (dtype), classes in columns [0, C); dlo (optional) = grad_scale * dloss/dlo with every
 *            column written (padding columns and images >= n_img: zero)
 *   labels:  int64 [n_img][Ho][Wo];  lse: f32 [n_img][Ho][Wo] workspace (per-pixel log-sum-exp);
 *   scratch: CAVP_CE_SCRATCH_FLOATS floats;  loss[0] = mean over the valid pixels (fixed summation order). */
int cavp_upsample_ce_head(int32_t dtype, const void* lo, const int64_t* labels, int32_t n_img, int32_t n_total, int32_t C,
                          int32_t Hi, int32_t Wi, int32_t ld, int32_t Ho, int32_t Wo, int32_t align_corners,
                          int32_t ignore_index, float grad_scale, float* loss, void* dlo, float* lse, float* scratch,
                          void* stream);

/* ---- fused optimiser step (harness row of SURVEY.md §8c): torch.optim.SGD(momentum, weight_decay) on the visual groups +
 * torch.optim.Adam on the audio encoder (main_vpo_mono.py:45-65,118-125), all tensors in one launch.
 * jobs_device: DEVICE array (built once; blk0 = running sum of cavp_optimizer_blocks(n) over the preceding jobs, ascending).
 * kind 0 = SGD (m = momentum buffer, v unused), 1 = Adam (m, v = first / second moments, zero-initialised).
 * lr of a job = lr_sgd (or lr_adam) * lr_mult; step = 1 for the first call (SGD: buf = d; Adam bias corrections). */
typedef struct cavp_opt_job {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  float lr_mult;
  float weight_decay;
  int32_t blk0;
  int32_t kind;
  int32_t vec;   /* p, g, m 16-byte aligned: float4 path */
  int32_t pad_;
} cavp_opt_job;
int32_t cavp_optimizer_blocks(int64_t n);
int cavp_optimizer_step(const cavp_opt_job* jobs_device, int32_t njobs, int32_t total_blocks, float lr_sgd, float lr_adam,
                        float momentum, float beta1, float beta2, float eps, int64_t step, void* stream);

/* ---- log-mel front-end = trainers' preprocess_audio (trainer_cavp_vpo_mono.py:43-52,59-69;
 * utils/sourcesep.py:23-47): STFT(n_fft 512, hop, centred window, reflect padding) -> |.|^2 -> mel filterbank ->
 * 20 log10(max(amin, x)) -> 2 (x - spec_min) / (spec_max - spec_min) - 1.
 * wave: f32 [N][A]; window: f32 [n_fft] (Hann of win_length, zero-padded to the frame centre); fb: f32 [n_fft/2+1][n_mels];
 * out: f32 [N][n_frames][n_mels] (n_frames <= 1 + A / hop). */
int cavp_mel_frontend(const float* wave, int32_t N, int32_t A, const float* window, const float* fb, float* out,
                      int32_t n_fft, int32_t hop, int32_t n_frames, int32_t n_mels, float amin, float spec_min,
                      float spec_max, void* stream);

/* ---- pixel-level audio-visual InfoNCE (loss/contrastive_aud.py::ContrastLoss, config #5 / SURVEY.md §8a row a13) ----
 * The class-balanced sampling (torch.randperm on the CPU generator, contrastive_aud.py:76-141) stays on the host; these
 * are the device stages for the N sampled anchors.  S = A A^T / T and dA = G A run on cavp_conv2d_nhwc / _wgrad. */
/* Nearest-neighbour down-sampling of the int64 label maps [B][H][W] to the feature resolution, int32 [B][h][w]
 * (contrastive_aud.py:18-22 `F.interpolate(gt.unsqueeze(1).float(), size, mode='nearest')`): only B*h*w int32 labels travel to the
 * host for the class-balanced sampling instead of the full-resolution int64 maps. */
int cavp_label_nearest(const int64_t* gt, int32_t* out, int32_t B, int32_t H, int32_t W, int32_t h, int32_t w, void* stream);
/* A[i] = x[b_i, :, p_i] / max(||.||, eps); x addressed by element strides (works for NCHW memory and for the NHWC
 * memory behind out_fusion); also saves the norms (F.normalize, contrastive_aud.py:25-26 + gathers :97-139). */
int cavp_gather_l2norm(const float* x, int64_t stride_b, int64_t stride_c, int64_t stride_p, const int32_t* idx_b,
                       const int32_t* idx_p, int32_t N, int32_t C, float eps, float* A, float* norms, void* stream);
/* info_nce (contrastive_aud.py:41-74) on S[ld][ld] (already / temperature): row_mlpp[i], loss[0] = -mean(row_mlpp);
 * optional dS = grad_scale * dloss/dS (zero in the padding rows / columns >= N). */
int cavp_infonce_rows(const float* S, const int32_t* labels, int32_t N, int32_t ld, float eps, float* row_mlpp,
                      float* loss, float* dS, float grad_scale, void* stream);
int cavp_symm_add(const float* d, float* g, int32_t n, float scale, void* stream); /* g = (d + d^T) * scale */
/* g = (d + d^T) * scale * scale_dev[0]: the loss's upstream gradient (a device scalar in torch.autograd) stays on the device -
 * reading it on the host stalled the launch queue behind the whole forward pass (config #5, trainer_cavp_vpo_mono.py:178-190) */
int cavp_symm_add_scaled(const float* d, float* g, int32_t n, float scale, const float* scale_dev, void* stream);
int cavp_l2norm_bwd_scatter(const float* dA, const float* A, const float* norms, const int32_t* idx_b,
                            const int32_t* idx_p, int32_t N, int32_t C, float* dx, int64_t stride_b, int64_t stride_c,
                            int64_t stride_p, void* stream);

/* ---- PVTv2-B5 visual backbone (models/visual/backbones/pvt/pvt.py, config #4 / SURVEY.md §8a row a12) ---- */
/* Attention.forward (pvt.py:102-130): softmax(q k^T * scale) v per head with the spatially-reduced K/V (Nk <= 256,
 * head_dim 64).  q, o: [B][Nq][heads*64]; kv: [B][Nk][2*heads*64] (k then v, as written by the `kv` Linear). */
int cavp_sra_attention(int32_t dtype, const void* q, const void* kv, void* o, int32_t B, int32_t Nq, int32_t Nk,
                       int32_t heads, int32_t head_dim, float scale, void* stream);
/* DWConv (pvt.py:315-326): depth-wise 3x3 pad 1 + bias on NHWC, optional fused act (GELU of Mlp.forward :46-55). */
int cavp_dwconv3x3_nhwc(int32_t dtype, const void* x, const float* w9c, const float* bias, void* y, int32_t N, int32_t H,
                        int32_t W, int32_t C, int32_t act, void* stream);
/* ABI 7: the same with act = CAVP_ACT_GELU and gelu'(t) stored into aux (y's shape; NULL = plain call): what the backward of
 * Mlp.forward (pvt.py:46-55) needs - it is multiplied into the data-gradient GEMM of fc2 (cavp_conv_desc.aux_mode 2), so the
 * training pass has no separate GELU and GELU-backward launches and the pre-activation is never stored. */
int cavp_dwconv3x3_nhwc_aux(int32_t dtype, const void* x, const float* w9c, const float* bias, void* y, void* aux, int32_t N,
                            int32_t H, int32_t W, int32_t C, int32_t act, void* stream);
/* data gradient of the depth-wise conv (pvt.py:46-55 backward): dx = dy correlated with the reversed taps of the same packed
 * [9][C] weights; dense NHWC of C channels. */
int cavp_dwconv3x3_bwd_data_nhwc(int32_t dtype, const void* dy, const float* w9c, void* dx, int32_t N, int32_t H, int32_t W,
                                 int32_t C, void* stream);
int cavp_pack_dwconv_weight(const float* w_c133, float* w9c, int32_t C, void* stream); /* [C][1][3][3] -> [9][C] */
/* OverlapPatchEmbed.proj of stage 1 (pvt.py:187-188): KSxKS conv, Cin <= 3, NCHW f32 in, NHWC out, + bias. */
int cavp_conv_smallcin_kxk_nchw(int32_t dtype, const float* x_nchw, const float* w_oihw, const float* bias, void* y_nhwc,
                                int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t Cout, int32_t KS, int32_t stride,
                                int32_t pad, void* stream);

/* ---- PVTv2-B5 training pass (backward of the kernels above; pvt.py:102-130,46-55,187-188,167-168) ---- */
/* Attention backward: dq [B][Nq][heads*64] (dtype), dkv f32 [B][Nk][2*heads*64] (dk | dv; overwritten) from q, kv, dout.
 * The softmax is recomputed from q and kv (nothing is saved by the forward).  workspace: row statistics,
 * cavp_sra_attention_bwd_workspace_bytes(B, Nq, heads) bytes, 16-byte aligned. */
size_t cavp_sra_attention_bwd_workspace_bytes(int32_t B, int32_t Nq, int32_t heads);
int cavp_sra_attention_bwd(int32_t dtype, const void* q, const void* kv, const void* dout, void* dq, float* dkv, int32_t B,
                           int32_t Nq, int32_t Nk, int32_t heads, int32_t head_dim, float scale, void* workspace,
                           size_t workspace_bytes, void* stream);
/* same, with dkv written in `dkv_dtype` (CAVP_F32 / CAVP_BF16): the fixed-order sum of the query splits' partials stores the
 * consumer's dtype directly. */
int cavp_sra_attention_bwd_to(int32_t dtype, const void* q, const void* kv, const void* dout, void* dq, void* dkv,
                              int32_t dkv_dtype, int32_t B, int32_t Nq, int32_t Nk, int32_t heads, int32_t head_dim, float scale,
                              void* workspace, size_t workspace_bytes, void* stream);
/* DWConv weight / bias gradient: dw_c133 f32 [C][1][3][3] += , dbias f32 [C] += (may be NULL).  The data gradient is
 * cavp_dwconv3x3_nhwc with the taps reversed. */
int cavp_dwconv3x3_wgrad(int32_t dtype, const void* x, const void* dy, float* dw_c133, float* dbias, int32_t N, int32_t H,
                         int32_t W, int32_t C, void* stream);
/* weight / bias gradient AND data gradient of the depth-wise conv in one walk over dy (w9c: the forward's packed [9][C] taps;
 * dx: dense NHWC of x's dtype, overwritten).  w9c = dx = NULL: the weight gradient alone. */
int cavp_dwconv3x3_bwd(int32_t dtype, const void* x, const void* dy, const float* w9c, void* dx, float* dw_c133, float* dbias,
                       int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
/* Weight gradient of cavp_conv_smallcin_kxk_nchw, step 1: im2col of the NCHW f32 input into cols [N*Ho*Wo][Kpad] (dtype;
 * column order (ci, kh, kw) = the OIHW weight's, zero beyond Cin*KS*KS; Kpad a multiple of 8).  Step 2 is
 * cavp_conv2d_wgrad_nhwc as a 1x1 layer with x = cols, giving dw [Cout][Kpad]. */
int cavp_smallcin_kxk_im2col(int32_t dtype, const float* x_nchw, void* cols, int32_t N, int32_t Cin, int32_t H, int32_t W,
                             int32_t KS, int32_t stride, int32_t pad, int32_t Kpad, void* stream);
/* [B][H][W][C] -> [B][H/s][W/s][s*s*C] (inverse != 0: back): the spatial-reduction conv (kernel = stride = s,
 * pvt.py:76-79) becomes a token GEMM over the rearranged rows, forward and backward. */
int cavp_space_to_depth(int32_t dtype, const void* src, void* dst, int32_t B, int32_t H, int32_t W, int32_t C, int32_t s,
                        int32_t inverse, void* stream);
/* timm DropPath around a residual branch (pvt.py:167-168): out = x + sample_scale[b] * branch; x == NULL: the branch's
 * gradient sample_scale[b] * g.  per_sample = elements per batch item. */
int cavp_row_scale_add(int32_t dtype, const void* x, const void* branch, const float* sample_scale, void* out, int32_t B,
                       int64_t per_sample, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CAVP_HIP_H_ */
