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

#define CAVP_ABI_VERSION 1

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
  int32_t tile;       /* 0 = auto; otherwise a tile-config id (testing / tuning) */
} cavp_conv_desc;

size_t cavp_conv2d_workspace_bytes(const cavp_conv_desc* d);
int cavp_conv2d_nhwc(const cavp_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift,
                     const float* nbias, const void* residual, void* y, void* workspace, size_t workspace_bytes,
                     void* stream);

/* Direct 3x3 conv for Cin in {1,2,3} reading an NCHW f32 tensor and writing NHWC (dtype) with scale/shift + act:
 * the ResNet deep-stem first conv (resnet.py:108-110, stride 2) and the first VGGish conv (vgg.py:26-36). */
int cavp_conv3x3_smallcin_nchw(int32_t dtype, const float* x_nchw, const float* w_oihw, const float* scale,
                               const float* shift, void* y_nhwc, int32_t N, int32_t Cin, int32_t H, int32_t W,
                               int32_t Cout, int32_t stride, int32_t act, void* stream);

/* nn.MaxPool2d(k, s, p) on NHWC (resnet.py:139,190: 3/2/1; vgg.py:30: 2/2/0). */
int cavp_maxpool_nhwc(int32_t dtype, const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C,
                      int32_t k, int32_t stride, int32_t pad, void* stream);

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

/* Sigmoid-gated single-key attention (attn.py:73-106 with N_kv == 1):
 *   s[b,h,t] = sigmoid(scale * <q[b,t,h,:], k[b,h,:]>);  o[b,t,h,:] = s[b,h,t] * v[b,h,:];  attn[b,h,t] = s.
 * q,o: [B][T][heads*hd] (dtype); k,v: [B][heads*hd] (dtype); attn: f32 [B][heads][T]. */
int cavp_attn_gate(int32_t dtype, const void* q, const void* k, const void* v, void* o, float* attn, int32_t B,
                   int32_t T, int32_t heads, int32_t hd, float scale, void* stream);

/* BatchNorm (eval) folding: scale = gamma * rsqrt(var + eps), shift = beta - mean * scale (f32, C entries). */
int cavp_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* scale,
                 float* shift, int32_t C, void* stream);

/* Weight packing: OIHW f32 (torch layout) -> OHWI (dtype).  Linear weights are the KH=KW=1 case (pure cast). */
int cavp_pack_weight_ohwi(int32_t dtype, const float* w_oihw, void* w_ohwi, int32_t Cout, int32_t Cin, int32_t KH,
                          int32_t KW, void* stream);

/* Element-wise cast between f32 and dtype (n elements): src_dtype -> dst_dtype. */
int cavp_cast(int32_t src_dtype, const void* src, int32_t dst_dtype, void* dst, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CAVP_HIP_H_ */
