/*
 * stlpose_hip.h -- C ABI of libstlpose_hip.so (gfx950 / MI355X).
 *
 * The reference (angelvillar96/STLPose) has no FFI: its hot path is ATen/cuDNN ops invoked
 * from Python nn.Modules.  This header is the boundary the build creates for that path
 * (SURVEY.md 8(b) "What the native side must export").  Every entry point names the reference
 * computation it replaces.  Conventions:
 *   - plain pointers + sizes only; all tensor pointers are DEVICE pointers owned by the caller
 *     (the Python host allocates them with torch); kernels never allocate;
 *   - activations are NHWC ("channels last"), element type `dtype` = STL_F32 | STL_BF16 | STL_F16 (mixed 16-bit mode:
 *     forward tensors STL_F16, gradients STL_BF16, see STL_DT2 / ydtype), accumulation always fp32, BatchNorm statistics fp64;
 *   - every call enqueues on `stream` (a hipStream_t passed as void*) and returns without
 *     synchronising, so a whole step can be captured in a hipGraph;
 *   - return value 0 = ok, negative = error (message via stl_last_error()); the Python mirror
 *     raises RuntimeError like the reference's ATen calls would.
 */
#ifndef STLPOSE_HIP_H
#define STLPOSE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define STL_F32 0
#define STL_BF16 1
#define STL_F16 2 /* IEEE half: the FORWARD tensors (raw conv outputs, materialised sums, MFMA operands, kernel-layout weights of
                     the forward convs) of the mixed 16-bit mode -- BatchNorm bounds their range, and 10 mantissa bits instead of
                     bf16's 7 cut the distance from the fp32 reference 6-8 x at the same bytes and MFMA rate.
                     Gradients have no such bound and stay bf16: the backward kernels take the type of the forward tensors they
                     read (BatchNorm-backward source y, mask_y, mask_z, the weight gradient's h) as a second dtype, `ydtype`. */
#define STL_NSHARD 2 /* BatchNorm sum buffers are [STL_NSHARD][2*C] doubles (atomic de-contention).  Every consumer
                        reads all shards of its channels at the start of the launch (a dependent round trip in
                        front of ~800 launches per step): 8 / 4 / 2 / 1 shards = 16.25 / 15.89 / 15.65 / 15.87 ms */

/* How a tensor is read ("normalise on load"): a conv / sum kernel applies the producing
 * BatchNorm (+ReLU) while staging its input, so BN never costs its own pass over HBM. */
/* Two element types in one int argument: low byte = type of the gradient-side tensors, bits 8-15 = type of the forward-side
 * tensors when it differs (0 = the same).  stl_head_backward: (dx, x); stl_weight_prep*: (data-gradient layouts, forward layouts). */
#define STL_DT2(grad_dtype, fwd_dtype) ((grad_dtype) | ((fwd_dtype) << 8))

#define STL_SRC_PLAIN 0 /* v = x                                              */
#define STL_SRC_BN 1    /* v = [relu](a*x + b), a,b from batch or running stats  */
#define STL_SRC_BNBWD 2 /* v = a*(dt - r1/n - yhat*r2/n): BatchNorm backward on load */
#define STL_SRC_BNADD 3 /* v = [relu](a*x + b + y): a residual block end z = ReLU(BN(x) + y) (HRnet.py:58-59) formed while the
                           NEXT unit's first convolution stages its input; x = raw conv output, y = the skip tensor (same
                           shape, PLAIN).  stl_conv_forward only (3x3 stride 1, and 1x1 with Ci, Co >= 64: layer1's
                           bottleneck units, HRnet.py:88-100); with stl_conv.src_out the sum is also written out once, for the
                           skip connection and the backward pass */

typedef struct stl_src {
    const void* x;        /* PLAIN/BN: tensor; BNBWD: dt (grad wrt BN output, post ReLU mask) */
    const void* y;        /* BNBWD: raw conv output the BN normalised; BNADD: the tensor added after the BN */
    int32_t mode;         /* STL_SRC_*                                                    */
    int32_t relu;         /* BN: apply ReLU after the affine                              */
    const double* stats;  /* [NSHARD][2C] batch sum / sum-of-squares (train) or NULL (eval) */
    const double* rstats; /* BNBWD: [NSHARD][2C] r1 = sum dt, r2 = sum dt*yhat             */
    const float* gamma;   /* [C]                                                          */
    const float* beta;    /* [C]                                                          */
    const float* rmean;   /* [C] running mean (eval)                                      */
    const float* rvar;    /* [C] running var  (eval)                                      */
    float inv_count;      /* 1 / (B*H*W) of the normalised tensor                         */
    float eps;
} stl_src;

/* Implicit-GEMM convolution (no im2col), 3x3 pad 1 or 1x1 pad 0, MFMA 16x16 tiles.
 * Replaces nn.Conv2d forward (reference src/models/HRnet.py:26-29,69-75,146-150,201-205,
 * 219-223,290-294,352-356,371-373) and, with `transposed` weights, aten::convolution_backward's
 * data gradient.  out[b,oy,ox,co] = sum_{ky,kx,ci} src(b, oy*stride+ky-pad, ox*stride+kx-pad, ci)
 * * w[co][ky*ks+kx][ci].   stuff=1: the source is a stride-2 zero-stuffed view of a half-size
 * tensor (data gradient of a stride-2 conv).  Epilogue (all optional): +bias, +addend, ReLU
 * mask from the BatchNorm that produced this conv's input in the forward pass (mask_*),
 * ReLU, per-channel sum/sumsq of the stored output (out_stats) or r1/r2 against mask_y (red). */
typedef struct stl_conv {
    int32_t dtype;
    int32_t B, Hi, Wi, Ci; /* source tensor dims (for stuff=1: the half-size tensor) */
    int32_t Ho, Wo, Co;    /* output dims                                          */
    int32_t ks, stride, stuff;
    int32_t TH, TW;        /* output pixel tile; 0,0 = let stl_conv_plan choose      */
    int32_t shape;         /* block shape id chosen by stl_conv_plan (-1 = choose at launch) */
    stl_src src;
    const void* w;         /* [Co][ks*ks][Ci] dtype                               */
    void* out;             /* [B,Ho,Wo,Co] dtype                                  */
    const float* bias;     /* [Co] or NULL                                        */
    int32_t out_relu;
    double* out_stats;     /* [NSHARD][2*Co] += or NULL                           */
    const void* addend;    /* [B,Ho,Wo,Co] dtype or NULL                          */
    const void* mask_y;    /* [B,Ho,Wo,Co] raw tensor whose BN(+ReLU) gates `out`  */
    stl_src mask_bn;       /* BN parameters for mask_y (mode must be STL_SRC_BN)   */
    double* red;           /* [NSHARD][2*Co] += (r1,r2) against mask_y or NULL     */
    const void* mask_z;    /* [B,Ho,Wo,Co] dtype or NULL: ReLU output z whose sign gates `out` (out = 0 where
                              z <= 0).  With addend / mask_y (mask_bn.relu = 0) / red this is the backward of a
                              residual block end  z = ReLU(BN(y) + x)  fused into the data gradient that
                              produces the last contribution to dz (HRnet.py:58-59,99-100). */
    void* src_out;         /* BNADD source: [B,Hi,Wi,Ci] dtype -- the transformed source (the block-end sum) is stored here, every
                              pixel by the one block whose tile owns it; NULL = not stored */
    int32_t ydtype, pad_;       /* data gradient (BNBWD source) of the mixed mode: element type of the FORWARD tensors it reads -- src.y,
                              mask_y, mask_z -- when it differs from dtype (STL_F16 with dtype STL_BF16); 0 = same as dtype */
} stl_conv;
int stl_conv_forward(const stl_conv* p, void* stream);
/* Fill p->shape / p->TH / p->TW once (host-side tile search) so that launches are cheap. */
int stl_conv_plan(stl_conv* p);
/* 1 when stl_conv_forward has a kernel variant that takes p (already planned) with a STL_SRC_BNADD source. */
int stl_conv_bnadd_ok(const stl_conv* p);

/* Weight gradient of the same convolution (aten::convolution_backward, weight part).
 * partial[s][co][tap][ci] (fp32) for s < nsplit; summed later by stl_reduce_slabs.
 * h = the conv's forward input (PLAIN or BN source), g = gradient of its output (PLAIN or BNBWD). */
typedef struct stl_wgrad {
    int32_t dtype;
    int32_t B, Hi, Wi, Ci, Ho, Wo, Co, ks, stride;
    int32_t TH, TW, nsplit;
    stl_src h;
    stl_src g;
    float* partial; /* [nsplit][Co][ks*ks][Ci] */
    int32_t ydtype; /* mixed mode: element type of the forward tensors h.x and g.y (STL_F16) when dtype (g.x = dt, the MFMA operands)
                       is STL_BF16; 0 = same as dtype */
} stl_wgrad;
int stl_conv_wgrad(const stl_wgrad* p, void* stream);
/* Up to STL_WGRAD_GROUP_MAX weight gradients of IDENTICAL geometry, tile, nsplit and gradient-source mode in one
 * launch (grid.x = n * nsplit).  Same results as n stl_conv_wgrad calls; replaces n of the per-layer
 * aten::convolution_backward weight parts of one branch (HRnet.py:140-186: the 3x3 convolutions of
 * _make_one_branch share one shape). */
#define STL_WGRAD_GROUP_MAX 8
typedef struct stl_wgrad_io { stl_src h; stl_src g; float* partial; } stl_wgrad_io;
typedef struct stl_wgrad_group { int32_t n, pad_; const stl_wgrad* p[STL_WGRAD_GROUP_MAX]; } stl_wgrad_group;
int stl_conv_wgrad_group(const stl_wgrad_group* g, void* stream);
/* Channel tile (32 or 64) of the kernel variant stl_conv_wgrad uses for this problem: the grid is
 * nsplit x ceil(Co/tile) x ceil(Ci/tile) blocks, which is what a caller sizes nsplit against. */
int stl_wgrad_chunk(const stl_wgrad* p);

/* Sum of up to 4 terms + optional ReLU, each term read through an stl_src and optionally
 * nearest-upsampled by 2^shift.  Replaces the residual add+ReLU of the reference's blocks
 * (HRnet.py:58-59,99-100), the materialisation of transition outputs (:352-358,371-376) and the
 * multi-resolution exchange sum  out_i = ReLU(sum_j f_ij(x_j))  incl. nn.Upsample (:207,:255-264). */
typedef struct stl_term {
    stl_src src;
    int32_t shift; /* source is (H>>shift, W>>shift) */
} stl_term;
typedef struct stl_fuse {
    int32_t dtype, B, H, W, C, nterms, relu;
    stl_term t[4];
    void* out;
} stl_fuse;
int stl_fuse_forward(const stl_fuse* p, void* stream);

/* Backward of stl_fuse_forward: du = (sum_k dz_k) * (z > 0); for every same-resolution BN term
 * accumulates r1 = sum du, r2 = sum du*yhat into that term's rstats.  (aten::threshold_backward
 * + native_batch_norm_backward reductions.) */
typedef struct stl_fuse_bwd {
    int32_t dtype, B, H, W, C, ngrads, relu, nbn; /* dtype: the gradients dz / du */
    const void* dz[4];
    const void* z;
    void* du;
    stl_src bn[4];    /* same-res BN terms (mode STL_SRC_BN): x = raw y, stats, gamma */
    double* rstats[4];
    int32_t ydtype, pad_; /* mixed mode: element type of the forward tensors z and bn[].x (STL_F16 with dtype STL_BF16); 0 = dtype */
} stl_fuse_bwd;
int stl_fuse_backward(const stl_fuse_bwd* p, void* stream);

/* Backward of one nearest-upsampled BN term (upsample_nearest2d_backward + BN reductions):
 * dt[b,y,x,c] = sum over the 2^shift x 2^shift patch of du; r1/r2 against the low-res raw y. */
typedef struct stl_upbwd {
    int32_t dtype, B, H, W, C, shift; /* H,W = LOW-res dims */
    const void* du;                   /* [B, H<<shift, W<<shift, C] */
    void* dt;                         /* [B,H,W,C] */
    stl_src bn;
    double* rstats;
    int32_t ydtype, pad_; /* mixed mode: element type of bn.x (STL_F16 with dtype STL_BF16); 0 = dtype */
} stl_upbwd;
int stl_upsample_backward(const stl_upbwd* p, void* stream);

/* 3x3 patches of an NCHW fp32 image batch (3 channels) as a 32-wide NHWC tensor
 * (k = (ky*3+kx)*3 + c, zero for k >= 27 and outside the image), optionally ImageNet-normalised
 * first ((x-mean)/std, reference lib/loss.py:46-47).  Turns the 3-channel stem conv
 * (HRnet.py:290) / VGG conv1_1 into a 1x1 convolution with Ci=32. */
int stl_patch3x3(int dtype, const float* img, void* out, int B, int H, int W, int stride,
                 const float* mean3, const float* std3, void* stream);

/* 1x1 head with bias: NHWC dtype [B,H,W,Ci] -> NCHW fp32 [B,J,H,W]  (HRnet.py:331-337,466). */
int stl_head_forward(int dtype, const void* x, const float* w, const float* bias, float* out,
                     int B, int H, int W, int Ci, int J, void* stream);
/* its backward: dx (dtype NHWC), per-block partials of dw [J][Ci] and db [J] (fp32). */
int stl_head_backward(int dtype, const void* x, const float* w, const float* dout, void* dx,
                      float* partial, int nblk, int B, int H, int W, int Ci, int J, void* stream);

/* Gaussian heatmap targets on device (reference data/JointsDataset.py:230-286 generate_target):
 * joints_xy [B][J][2] in image pixels, vis [B][J] in {0,1} -> target [B][J][Hh][Wh] (unnormalised
 * gaussian, centre value 1, support 3*sigma, clipped at the border), tweight [B][J] (vis, or 0 when
 * the gaussian lies completely outside the map).  stride = image_size / heatmap_size per axis. */
int stl_gaussian_targets(const float* joints_xy, const float* vis, float* target, float* tweight, int B, int J,
                         int Hh, int Wh, float stride_x, float stride_y, float sigma, void* stream);

/* PersonMSELoss forward+backward (reference lib/loss.py:71-94):
 * loss = 0.5*mean(((o-t)*w)^2) over all B*J*H*W;  dout = (o-t)*w^2 / (B*J*H*W) * gscale.
 * loss == NULL: only dout and the per-block partial sums are written; the caller finishes the scalar with
 * stl_sum_partials(partial, nblk, 0.5 / (B*J*H*W), loss, 0, stream) whenever it likes (the train step: behind the
 * backward program, so that the one-block sum does not sit in front of the head's gradient). */
int stl_mse_loss(const float* out, const float* target, const float* tweight, float* dout,
                 double* partial, int nblk, float* loss, int B, int J, int HW, float gscale, void* stream);

/* get_max_preds_hrnet (reference lib/pose_parsing.py:16-55): first-max flat argmax + max per
 * (b,joint); idx int32 [B*J], maxval f32 [B*J], preds f32 [B*J*2] (x,y, zeroed when max<=0). */
int stl_heatmap_argmax(const float* hm, int32_t* idx, float* maxval, float* preds, int BJ, int H,
                       int W, void* stream);
/* flip_back + 1px shift + average of forward_pass(flip=True) (lib/inference.py:20-26,
 * lib/transforms.py:147-164) on device: out = 0.5*(a + shift(flip(b))). */
int stl_flip_merge(const float* a, const float* bflip, float* out, const int32_t* perm, int B, int J,
                   int H, int W, void* stream);
/* +-0.25 px refinement + inverse affine of get_final_preds_hrnet (lib/pose_parsing.py:58-92). */
int stl_final_preds(const float* hm, const float* center, const float* scale, float* preds,
                    float* maxval, int B, int J, int H, int W, void* stream);

/* Table-driven batched helpers (one launch for the whole network). */
typedef struct stl_wprep { /* one conv weight: OIHW fp32 master -> kernel layouts */
    int64_t src_off;  /* element offset in master */
    int64_t fwd_off;  /* element offset in `wk`: [Co][tap][Cip] (Cip = padded Ci)          */
    int64_t bwd_off;  /* element offset in `wk`: [Ci][8-tap][Co] for the data gradient, or -1 */
    int32_t Co, Ci, ks, Cip; /* Cip: Ci padded (stem patches: 27 -> 32)                     */
    int32_t patch;    /* 1: Ci*ks*ks flattened as k=(tap*Ci + ci) into Cip (stem / VGG conv1_1) */
    int32_t blk0;     /* first block of this entry */
} stl_wprep;
int stl_weight_prep(int dtype, const float* master, void* wk, const stl_wprep* tab, int n,
                    int nblocks, void* stream);
/* The same over a sub-range of the table: `tab` points at the first entry of the range, blk_base is
 * that entry's blk0.  Used per gradient bucket, right after the bucket's optimiser slice. */
int stl_weight_prep_range(int dtype, const float* master, void* wk, const stl_wprep* tab, int n,
                          int blk_base, int nblocks, void* stream);

typedef struct stl_slab { /* one wgrad result: sum partial[s] -> grad (OIHW fp32) */
    int64_t part_off; /* element offset into `partials`; a multiple of 4 (16-byte aligned) when Ci % 4 == 0 */
    int64_t grad_off; /* element offset into `grads`    */
    int32_t nsplit, Co, Ci, ks, Cip, patch, blk0;
    int32_t pad; /* if non-zero: element stride between consecutive splits (default Co*taps*Ci) */
} stl_slab;
int stl_reduce_slabs(const float* partials, float* grads, const stl_slab* tab, int n, int nblocks,
                     void* stream);

typedef struct stl_bnrec { /* one BatchNorm layer */
    int64_t stats_off;  /* offset (doubles) of its [NSHARD][2C] block in the stats arena  */
    int64_t param_off;  /* offset of gamma in the fp32 master (beta follows at +C)        */
    int64_t buf_off;    /* offset of running_mean in the buffer arena (running_var at +C) */
    int32_t C;
    float inv_count;
} stl_bnrec;
/* running_mean/var momentum update (unbiased var), reference nn.BatchNorm2d(momentum=0.1).
 * overflow (device int32, NULL = off; the caller initialises it to INT32_MAX): range guard of the 16-bit forward tensors.  A raw
 * conv output beyond the storage type's range (STL_F16: |y| > 65504, e.g. a badly scaled checkpoint, lib/model_setup.py:38-42
 * loads arbitrary ones) is stored as infinity and shows in the layer's sums; such a layer keeps its running statistics and the
 * smallest index i of tab[] with non-finite sums is left in *overflow (atomic min).  The optimisers below skip elements whose
 * gradient is not finite, so the step that overflowed leaves the weights as they were; the host reads the word when it reads
 * the loss (TrainStep.check_forward_range) and raises with the layer's name. */
int stl_bn_running_update(const double* stats, float* buffers, int64_t* num_batches_tracked /* [n] or NULL */,
                          const stl_bnrec* tab, int n, float momentum, int32_t* overflow, void* stream);
/* dgamma = r2, dbeta = r1 from the backward reduction arena into the flat grad buffer. */
int stl_bn_param_grads(const double* rstats, float* grads, const stl_bnrec* tab, int n, void* stream);

/* Optimisers over the flat fp32 master (torch.optim.Adam / SGD semantics, reference
 * lib/model_setup.py:135-141).  hyper = device float[8]: lr, beta1, beta2, eps, weight_decay,
 * momentum, nesterov, gscale;  step = device int32 (incremented by the kernel).  overflow: the word stl_bn_running_update
 * maintains -- when it reports a non-finite forward tensor the WHOLE step is skipped (step is left negative, not counted:
 * ReLU(NaN) = 0 can let the loss and the gradients of such a step come out finite and wrong); independently of it an element
 * whose gradient is NaN or infinite is left untouched (weight and moments). */
int stl_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                  int32_t* step, const int32_t* overflow /* or NULL */, void* stream);
int stl_sgd_step(float* p, const float* g, float* mom, int64_t n, const float* hyper, int32_t* step,
                 const int32_t* overflow /* or NULL */, void* stream);
/* Per-bucket form: stl_optim_begin_step increments `step` once, the *_slice calls then update any
 * contiguous slices of the flat buffers (pointers already offset) with that step count -- the
 * optimiser of a gradient bucket runs as soon as the bucket is final, overlapped with backward. */
int stl_optim_begin_step(int32_t* step, const int32_t* overflow /* or NULL */, void* stream);
int stl_adam_slice(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                   const int32_t* step, void* stream);
int stl_sgd_slice(float* p, const float* g, float* mom, int64_t n, const float* hyper, const int32_t* step,
                  void* stream);

/* Device affine crop + normalisation of a batch (reference data/JointsDataset.py:189-200: cv2.warpAffine(img,
 * get_affine_transform(c, s, r, image_size), INTER_LINEAR) followed by ToTensor + Normalize, data_loaders.py:59-61).
 * src: uint8 HWC RGB images packed in one buffer (src_off[b] bytes, src_hw[b] = (H, W)); minv[b] = the 2x3 matrix
 * that maps OUTPUT pixel (x, y) to source coordinates (the inverse of the reference's `trans`); flip[b] != 0 reads the
 * source mirrored left-right (the flip augmentation, :183-186); out = fp32 NCHW [B,3,Ho,Wo]; mean3/std3 may be NULL.
 * Bilinear taps outside the source contribute 0 (BORDER_CONSTANT).  cv2's 1/32-pixel fixed-point coordinate rounding
 * is NOT reproduced (exact float bilinear): parity for that rounding is unpinned (cv2 absent). */
int stl_affine_crop(const uint8_t* src, const int64_t* src_off, const int32_t* src_hw, const float* minv, const int32_t* flip,
                    float* out, int B, int Ho, int Wo, const float* mean3, const float* std3, void* stream);
/* VGG perceptual path (reference lib/loss.py:17-58). */
int stl_maxpool2x2(int dtype, const void* x, void* out, int B, int H, int W, int C, void* stream);
int stl_l1_partial(int dtype, const void* a, const void* b, int64_t n, double* partial, int nblk,
                   void* stream); /* partial[i] = sum |a-b| over block i's share */
/* Same with squared differences (content / Gram MSE of the VGG19 style loss, V2 -- no reference counterpart). */
int stl_l2_partial(int dtype, const void* a, const void* b, int64_t n, double* partial, int nblk, void* stream);
int stl_bilinear_nchw(const float* in, float* out, int B, int C, int H, int W, int Ho, int Wo,
                      void* stream); /* F.interpolate(mode='bilinear', align_corners=False) */
int stl_sum_partials(const double* partial, int n, double scale, float* out, int accumulate, void* stream);

/* Layout / dtype utilities. */
int stl_nchw_to_nhwc(int dtype, const float* in, void* out, int B, int C, int H, int W, void* stream);
int stl_nhwc_to_nchw(int dtype, const void* in, float* out, int B, int C, int H, int W, void* stream);

/* ---- native program replay (csrc/program.hip) --------------------------------------------------
 * The Python planner turns the network into a flat list of the calls above; stl_program_run()
 * enqueues all of them in one C call, on up to 16 HIP streams, with cross-stream RAW dependencies
 * expressed as events (wait[] = indices of earlier ops that record). */
#define STL_OP_CONV 0
#define STL_OP_WGRAD 1
#define STL_OP_FUSE 2
#define STL_OP_FUSE_BWD 3
#define STL_OP_UP_BWD 4
#define STL_OP_PATCH 5
#define STL_OP_HEAD 6
#define STL_OP_HEAD_BWD 7
#define STL_OP_REDUCE_RANGE 8 /* stl_reduce_slabs over a sub-range of the table (a gradient bucket) */
#define STL_OP_BN_GRADS_RANGE 9 /* stl_bn_param_grads over a sub-range of the table */
#define STL_OP_WGRAD_GROUP 10   /* stl_conv_wgrad_group */
#define STL_OP_OPTIM_SLICE 11   /* stl_adam_slice / stl_sgd_slice of a gradient bucket, behind its reductions */
#define STL_OP_WPREP_RANGE 12   /* stl_weight_prep_range of the bucket's convolutions (the NEXT step's kernel-layout weights) */
/* A gradient bucket = a contiguous slice of the flat gradient buffer whose weight-gradient slabs and
 * BatchNorm reductions are complete at some point of the backward program.  Reducing it there (and
 * recording an event) lets the data-parallel all-reduce of that slice start while the rest of
 * backward still runs (reference: the per-step gradient gather of nn.DataParallel, 02_train.py:109). */
typedef struct stl_reduce_range { const float* partials; float* grads; const stl_slab* tab; int32_t n, blk_base, nblocks, pad_; } stl_reduce_range;
typedef struct stl_bn_range { const double* rstats; float* grads; const stl_bnrec* tab; int32_t n, pad_; } stl_bn_range;
int stl_reduce_slabs_range(const stl_reduce_range* r, void* stream);
/* Optimiser of one gradient bucket as a program op (single-process training: with data parallelism the collective sits
 * between a bucket's reduction and its optimiser, and the host issues the slices, see train_step.py).  kind 0 = Adam,
 * 1 = SGD (v unused).  Reference: optimizer.step() after loss.backward(), 02_train.py:113-114. */
typedef struct stl_optim_slice { int32_t kind, pad_; float* p; const float* g; float* m; float* v; int64_t n; const float* hyper; const int32_t* step; } stl_optim_slice;
typedef struct stl_wprep_range { int32_t dtype /* STL_DT2(data-gradient layouts, forward layouts) */, n, blk_base, nblocks; const float* master; void* wk; const stl_wprep* tab; } stl_wprep_range;
typedef struct stl_patch { int32_t dtype, B, H, W, stride, pad_; const float* img; void* out; const float* mean3; const float* std3; } stl_patch;
typedef struct stl_head { int32_t dtype, B, H, W, Ci, J; const void* x; const float* w; const float* bias; float* out; } stl_head;
typedef struct stl_head_bwd { int32_t dtype /* STL_DT2(dx, x) */, B, H, W, Ci, J, nblk, pad_; const void* x; const float* w; const float* dout; void* dx; float* partial; } stl_head_bwd;
typedef struct stl_op {
    int32_t kind;    /* STL_OP_*                                        */
    int32_t stream;  /* index into the streams array given to run()      */
    const void* desc; /* the op's descriptor struct (kept alive by the caller) */
    int32_t nwait;
    int32_t wait[8];
    int32_t record;  /* 1: record an event after this op (someone waits on it) */
} stl_op;
int stl_program_create(const stl_op* ops, int n, int nstreams, void** out_handle);
int stl_program_run(void* program, void* const* streams /* hipStream_t[nstreams]; [0] = main */);
/* Ops [first, last) only; first == 0 forks the side streams, last == n joins them.  A data-parallel host runs the backward program
 * bucket by bucket and enqueues each bucket's all-reduce between two ranges, i.e. right behind the bucket in every in-order queue. */
int stl_program_run_range(void* program, void* const* streams, int first, int last);
int stl_program_destroy(void* program);
/* The same program as ONE explicit HIP graph: stl_program_graph_build records every op's kernel launches (nothing runs) and
 * adds them as kernel nodes with the program's dependencies (in-order streams + waits) -- built, not captured, because stream
 * capture of a plan that forks onto three or more streams crashes in hipStreamEndCapture on ROCm 7.2; stl_program_graph_launch
 * replays it on `stream`.  No events exist in this mode: stl_program_wait_op (the data-parallel bucket pick-up) needs
 * stl_program_run. */
int stl_program_graph_build(void* program);
int stl_program_graph_launch(void* program, void* stream);
/* Make `stream` wait for op `op` (which must record) of the LAST run of the program: how a
 * communication stream picks up a finished gradient bucket. */
int stl_program_wait_op(void* program, int op, void* stream);

/* Self-checks that need no reference: MFMA / LDS-transpose lane maps (used by tests). */
int stl_selftest_mfma(float* out /* [4] max abs err: bf16 mfma, f32 mfma, tr-read, f64 atomic */, void* stream);

const char* stl_last_error(void);
int stl_version(void);
/* Hash (16 hex digits) of the kernel and header sources this library was compiled from (stlpose_amd/build.py). */
const char* stl_build_id(void);
/* Name of the kernel instantiation the calling thread launched last, e.g. "conv_core_kernel<bf16,3,4,2,4,2,3,1,0,1,-1,0>"
 * (template arguments in declaration order) -- measurement only: bench.py groups per-launch timings by it. */
const char* stl_last_kernel(void);

#ifdef __cplusplus
}
#endif
#endif
