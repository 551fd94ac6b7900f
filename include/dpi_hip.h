/*
 * dpi_hip.h — C ABI of libdpi_hip.so: the MI355X (gfx950) kernels behind the deep-prior
 * interpolation hot path (MulResUnet3D / MulResUnet / Skip3D forward + backward, masked loss,
 * metrics, Adam, overlap-add reassembly).
 *
 * The reference (polimi-ispl/deep_prior_interpolation) has NO native/FFI interface: its hot path
 * sits behind `architectures.get_net(args, outchannel) -> torch.nn.Module`
 * (architectures/__init__.py:10) and delegates all arithmetic to torch's aten kernels.  Each entry
 * point below therefore cites the torch call site in the reference whose work it takes over; the
 * Python binding a maintainer adds is shown in INTEGRATION.md.
 *
 * Conventions
 *  - All pointers are DEVICE pointers owned by the caller (PyTorch caching allocator), fp32 unless
 *    stated, batch N = 1 (the reference optimises one patch at a time, main.py:131-135), layout
 *    channel-planar [C][D][H][W] with W contiguous.  2-D tensors use D = 1 and kd = 1.
 *    A channel slice [c0:c1] of such a tensor is itself a valid tensor (zero-copy concat).
 *  - `stream` is a hipStream_t passed as void*; every call only enqueues work on it (graph-capturable:
 *    no allocation, no synchronisation, no host read-back).
 *  - Return value: 0 on success, negative DPI_E_* otherwise; dpi_last_error() gives the message of
 *    the calling thread's last failure.  No exceptions cross the ABI.
 *  - "chain": optional per-channel input transform applied while a tensor is LOADED, 5 floats per
 *    channel {ps, pb, slope, qs, qb}:  T(x) = qs * act(ps*x + pb) + qb,  act(v) = v>0 ? v : slope*v.
 *    It lets BatchNorm-apply + LeakyReLU be fused into the consumer instead of a separate pass.
 *    NULL = identity.
 *  - "stat partials": double[nblk][C][2] = per-block {sum, sum of squares}; reduced in fixed order by
 *    dpi_bn_finalize (deterministic, double precision).
 */
#ifndef DPI_HIP_H
#define DPI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPI_OK 0
#define DPI_E_ARG (-1)      /* invalid argument / unsupported shape */
#define DPI_E_LAUNCH (-2)   /* HIP launch or runtime error */
#define DPI_E_WORKSPACE (-3)/* workspace too small */

#define DPI_CHAIN_STRIDE 5

const char* dpi_last_error(void);
/* ABI version (404 = round 6: the ten dpi_set_* tuning functions left the ABI for dpi_set_option): 300 = round 3 (dpi_conv_desc carries its own size as first field), 301 adds dpi_conv_fwd_ws / dpi_conv_bwd_data_ws /
 * dpi_conv_bwd_data_dual, 401 = dpi_pack_scratch_bytes / dpi_pack_release, 402 = dpi_pack_forget (round 5), 403 = dpi_join_bwd (round 5), 400 = round 4: dpi_conv_desc grows the `io` field (bf16 storage of activations) and the elementwise entry
 * points get `_io` twins that take the storage types of their tensors.  A binding checks `>=` the version it was written against and
 * dpi_conv_desc_size() == its own struct size. */
int dpi_version(void);
/* Number of devices / properties as HIP sees them (no torch involved). */
int dpi_device_info(int device, int* cus, int* lds_bytes, size_t* hbm_bytes, char* name, int name_len);
/* Profiling aid (no reference counterpart): an empty kernel `dpi_marker_kernel` launched with `id` workgroups of one wave.
 * rocprofv3's kernel trace records launch grids but not kernel arguments, so tools/rocpd_stats.py uses these markers to
 * attribute the dispatches that follow (in host launch order) to the layer the caller tagged; id = 1 ends a scope. */
int dpi_profile_marker(int id, void* stream);
/* ABI 404.  The ONE process-global switchboard of the library, for tests, tools and A/B experiments only — the product path (ops.py, main.py,
 * parallel.py, bench.py's timed region) never calls it: what a launch computes is a function of its descriptor and arguments alone (SURVEY §8(b):
 * "no hidden global state").  Every key selects between launch PLANS or kernel VARIANTS that compute the same result (or, for the *_debug keys,
 * skips phases for timing experiments and says so); none changes the arithmetic contract of an entry point.  Returns DPI_OK, or DPI_E_ARG for
 * an unknown key / a value out of range.  Keys (defaults in brackets; the environment variable that sets the initial value, if any):
 *   "splitk"                   [1]  0: no input-channel split of coarse-level launches (DPI_SPLITK)
 *   "dual_bwd_data"            [1]  0: dpi_conv_bwd_data_dual always runs two launches (DPI_NO_DUAL)
 *   "bw_pair"                  [2]  backward-weight, two 4-channel groups per workgroup: 0 off, 1 every layer with >= 2 full groups, 2 the long
 *                                   group loops of the finest level only (DPI_BW_PAIR)
 *   "bw_workgroups"            [-]  workgroups the backward-weight MFMA launch plan aims at (> 0)
 *   "bw_xcd_order"             [1]  XCD-aware workgroup order of that plan (0 / 1)
 *   "mfma_min_cout"            [8]  3x3(x3) stride-1 forward / backward-data with at least this many output channels run the fp32-MFMA stencil
 *   "bwd_weight_mfma_min_cout" [8]  ... the same threshold of the backward-weight dispatch
 *   "fewco_mfma"               [1]  0: forward 3x3x3 with Cout <= 4 back on the VALU kernel
 *   "q4"                       [1]  4x4x1-MFMA kernel for <= 8 output channels: 0 off, 1 where it pays, 2 wherever it can run (tests)
 *   "q4_ck"                    [0]  its chunk variant: 2 planar, 4 channel-interleaved, 0 by shape
 *   "q4_debug"                 [0]  its phase-skipping bits (timing experiments, WRONG results): 0 no input loads after the first chunk, 1 no LDS
 *                                   stores after it, 2 no MFMAs, 3 no output stores, 5 no barriers; bits 8-15 = KiB of extra dynamic LDS
 *   "bf16_debug"               [0]  bf16 stencil kernel: bits 0-2 skip phases (WRONG results), bit 3 routes EVERY 3x3(x3) stride-1 convolution of
 *                                   a precision = 1 descriptor through it (default: only the shapes where it beats the fp32 kernels)
 * (the per-knob functions behind the keys live in csrc/dpi_hip_internal.h and are not exported). */
int dpi_set_option(const char* key, int value);

/* ---------------------------------------------------------------- convolution ------------------
 * Replaces nn.Conv3d / nn.Conv2d built at architectures/base.py:123,176 (k in {1,3}, stride in {1,2},
 * zero padding (k-1)/2) and their autograd backward (aten convolution_backward).
 * x: [Cin][D][H][W]   w: [Cout][Cin][kd][k][k]   y: [Cout][Do][Ho][Wo],  Xo = (X + 2*pad - k)/stride + 1.
 * kd = k for 3-D, kd = 1 for 2-D (D must be 1).
 */
typedef struct {
  /* = sizeof(dpi_conv_desc) of the header the CALLER was built against (dpi_conv_desc_size() gives the library's).  Every entry
   * point that takes a descriptor rejects a mismatch with DPI_E_ARG instead of reading fields past a shorter struct: the
   * descriptor grew a `precision` field in round 2 and a binding with the old 8-int layout read it from adjacent memory. */
  int size;
  int Cin, Cout;
  int D, H, W;      /* input spatial size */
  int k, kd;        /* kernel extent in H/W and in D */
  int stride;       /* 1 or 2 (applies to D only when kd > 1) */
  /* 0 (default): fp32 operands, the reference's arithmetic.  1: mixed precision of BASELINE configs[4] — the operands of 3x3(x3)
   * stride-1 convolutions (forward, backward-data, backward-weight) are rounded to bf16 on their way into the matrix cores
   * (v_mfma_f32_16x16x32_bf16, fp32 accumulate); tensors in HBM, master weights, BatchNorm statistics and Adam stay fp32.
   * 2: "split" mode — forward / backward-data operands are split exactly into three bf16 terms and six partial products are
   * accumulated in fp32: fp32-class accuracy (same tolerance as precision 0 against the fp64 oracle) on the bf16 matrix cores. */
  int precision;
  /* ABI 400: storage type of the ACTIVATION tensors of this layer in HBM, a mask of DPI_IO_*: bit set = that tensor is bf16
   * (2 bytes per element, same [C][D][H][W] layout), clear = fp32.  The pointer arguments keep their `float*` spelling; with a bit
   * set the library reads / writes that tensor as bf16: widened exactly on load, rounded to nearest-even on store, all arithmetic
   * and accumulation in fp32 (BASELINE configs[4]: "bf16 activations + fp32 Adam master weights").  Weights, biases, weight
   * gradients, chains and statistic partials are always fp32 / double.  BatchNorm partials emitted by dpi_conv_fwd describe the
   * stored (rounded) output.  0 = every tensor fp32 (the reference's storage, main.py:112). */
  int io;
} dpi_conv_desc;
#define DPI_IO_X_BF16 1   /* x: input of dpi_conv_fwd / dpi_conv_bwd_weight */
#define DPI_IO_Y_BF16 2   /* y: output of dpi_conv_fwd */
#define DPI_IO_DY_BF16 4  /* dy: input of dpi_conv_bwd_data / dpi_conv_bwd_weight */
#define DPI_IO_DX_BF16 8  /* dx: output (and, with accumulate, input) of dpi_conv_bwd_data */
/* sizeof(dpi_conv_desc) as the library was compiled; a binding asserts it equals its own struct size at load time. */
int dpi_conv_desc_size(void);

/* number of stat-partial blocks dpi_conv_fwd writes for this problem (0 if desc invalid) */
int dpi_conv_fwd_stat_blocks(const dpi_conv_desc* d);
/* y = conv(T(x), w) + bias.  bias may be NULL.  stat_partials (may be NULL): double[nblk][Cout][2]. */
int dpi_conv_fwd(const dpi_conv_desc* d, const float* x, const float* x_chain, const float* w,
                 const float* bias, float* y, double* stat_partials, void* stream);
/* dx (+)= conv_transpose(dy, w):  dx [Cin][D][H][W], dy [Cout][Do][Ho][Wo].  accumulate != 0 adds. */
int dpi_conv_bwd_data(const dpi_conv_desc* d, const float* dy, const float* w, float* dx,
                      int accumulate, void* stream);
/* The same two operations with a caller-owned workspace (ABI 301).  At the coarse levels of the U-Net a convolution is a few
 * dozen output tiles with hundreds of input channels: the library then splits the input-channel loop over more workgroups, writes
 * partial outputs into `ws` and sums them in a fixed order (deterministic).  dpi_conv_*_ws_floats(d) = floats wanted (0: the
 * problem is not split and ws may be NULL); with ws == NULL these are exactly dpi_conv_fwd / dpi_conv_bwd_data.  Results of the
 * split and the unsplit launch differ by fp32 rounding of the partial sums only. */
size_t dpi_conv_fwd_ws_floats(const dpi_conv_desc* d);
size_t dpi_conv_bwd_data_ws_floats(const dpi_conv_desc* d);
int dpi_conv_fwd_ws(const dpi_conv_desc* d, const float* x, const float* x_chain, const float* w,
                    const float* bias, float* y, double* stat_partials, float* ws, size_t ws_floats, void* stream);
int dpi_conv_bwd_data_ws(const dpi_conv_desc* d, const float* dy, const float* w, float* dx,
                         int accumulate, float* ws, size_t ws_floats, void* stream);
/* dx (+)= conv_transpose(dy3, w3) + conv_transpose(dy1, w1) for a 3x3(x3) stride-1 layer d3 and a 1x1(x1) layer d1 that read the SAME
 * tensor (equal Cin, D, H, W): Block3d.conv1 + shortcut (mulresunet.py:72-96) and ResPath3d.conv3x3 + conv1x1 (mulresunet.py:99-113),
 * whose input gradient is the sum of the two transposed convolutions.  Where the fp32-MFMA stencil kernel serves d3, the 1x1x1 term is
 * accumulated in the same pass (dx written once instead of written by one launch and read-modified-written by the next); every other
 * case runs the two launches.  ws / ws_floats: dpi_conv_bwd_data_ws_floats(d3) (may be NULL / 0). */
int dpi_conv_bwd_data_dual(const dpi_conv_desc* d3, const float* dy3, const float* w3,
                           const dpi_conv_desc* d1, const float* dy1, const float* w1,
                           float* dx, int accumulate, float* ws, size_t ws_floats, void* stream);
/* dw[Cout][Cin][kd][k][k] = sum_p dy[co][p] * T(x)[ci][p*stride + tap - pad].
 * workspace: float[dpi_conv_bwd_weight_ws_floats(d)].  */
size_t dpi_conv_bwd_weight_ws_floats(const dpi_conv_desc* d);
int dpi_conv_bwd_weight(const dpi_conv_desc* d, const float* x, const float* x_chain, const float* dy,
                        float* dw, float* ws, size_t ws_floats, void* stream);


/* ---------------------------------------------------------------- BatchNorm / activations -------
 * Replaces nn.BatchNorm3d/2d in training mode (base.py:164,214; mulresunet.py:80-81,104,225) and
 * LeakyReLU(0.2) (base.py:102), plus their backward.
 */
int dpi_stat_blocks(int C, size_t V);
/* partials[nblk][C][2] of T(x) over the V voxels of each channel */
int dpi_channel_stats(const float* x, const float* chain, int C, size_t V, double* partials, void* stream);
/* Reduce partials -> mean, invstd (biased var, eps); update running stats (momentum, unbiased var);
 * write chain_out[c] = {gamma*invstd, beta - mean*gamma*invstd, slope, 1, 0}  (BN-apply followed by act with
 * `slope`; slope = 1 means no activation) or, with act_first != 0, {1, 0, slope, gamma*invstd, beta - mean*...}
 * (BN of act(x), the partials then being statistics of act(x)).  running_* / nbt / chain_out may be NULL.
 * in_chain != NULL (requires slope == 1, act_first == 0): the partials are statistics of T_in(x) and chain_out is the
 * composition BN(T_in(x)) = {in.ps, in.pb, in.slope, a*in.qs, a*in.qb + shift} — bn1 over the never-materialised
 * concat of Block3d (mulresunet.py:90-91).
 * mean_invstd: float[2][C]. */
int dpi_bn_finalize(const double* partials, int nblk, int C, size_t count, const float* gamma,
                    const float* beta, float eps, float momentum, float slope, int act_first,
                    const float* in_chain, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float* mean_invstd, float* chain_out, void* stream);
/* y = T(x) elementwise */
int dpi_chain_apply(const float* x, const float* chain, int C, size_t V, float* y, void* stream);
/* BatchNorm backward, two-phase, with the surrounding LeakyReLU folded in (slope 1 = none):
 *   pre_slope  (act -> BN, mulresunet.py:93-94,110-111): u = act(x) was normalised; dx = du * act'(x)
 *   post_slope (BN -> act, base.py:214-215):            dy is w.r.t. act(BN(x)); g = dy * act'(gamma*xhat + beta)
 * phase 1: partials[nblk][C][2] = {sum g, sum g*xhat}, xhat = (act_pre(x) - mean) * invstd
 * phase 2: dx = gamma*invstd*(g - sum_g/V - xhat*sum_gxhat/V) * act_pre'(x); dgamma = sum_gxhat; dbeta = sum_g
 * in_chain != NULL (requires pre_slope == 1): the normalised tensor was u = T_in(x) (value only) and dx is the
 * gradient w.r.t. u, not x. */
int dpi_bn_bwd_reduce(const float* dy, const float* x, const float* mean_invstd, const float* gamma,
                      const float* beta, const float* in_chain, float pre_slope, float post_slope, int C,
                      size_t V, double* partials, void* stream);
int dpi_bn_bwd_apply(const float* dy, const float* x, const float* mean_invstd, const float* gamma,
                     const float* beta, const float* in_chain, float pre_slope, float post_slope,
                     const double* partials, int nblk, int C, size_t V, float* dx, float* dgamma,
                     float* dbeta, void* stream);
/* dpi_bn_bwd_apply fused with phase 1 (dpi_bn_bwd_reduce) of up to two BatchNorms whose incoming gradient is this dx
 * (the residual joins of Block3d / ResPath3d, mulresunet.py:92-94,109-111): partials_a/b[dpi_stat_blocks][C][2] receive
 * {sum g, sum g*xhat} of dx against (xa, mi_a, gamma_a, beta_a, chain_a, post_a) resp. (xb, ...); xa / xb may be NULL. */
int dpi_bn_bwd_apply_fork(const float* dy, const float* x, const float* mean_invstd, const float* gamma,
                          const float* beta, const float* in_chain, float pre_slope, float post_slope,
                          const double* partials, int nblk, int C, size_t V, float* dx, float* dgamma,
                          float* dbeta, const float* xa, const float* mi_a, const float* gamma_a,
                          const float* beta_a, const float* chain_a, float post_a, double* partials_a,
                          const float* xb, const float* mi_b, const float* gamma_b, const float* beta_b,
                          const float* chain_b, float post_b, double* partials_b, void* stream);
/* Phase 2 of TWO BatchNorm backward passes that share the incoming gradient dy (the two branches of a residual join), from
 * their phase-1 partials, in one pass; optionally phase 1 of one more BatchNorm whose incoming gradient is dxb on the
 * channels [f_lo, f_hi) and whose input is xb itself (raw): f_partials[dpi_stat_blocks][f_hi - f_lo][2]. */
int dpi_bn_bwd_apply_dual(const float* dy, int nblk, int C, size_t V, const float* xa, const float* mi_a,
                          const float* gamma_a, const float* beta_a, const float* chain_a, float post_a,
                          const double* partials_a, float* dxa, float* dgamma_a, float* dbeta_a, const float* xb,
                          const float* mi_b, const float* gamma_b, const float* beta_b, const float* chain_b,
                          float post_b, const double* partials_b, float* dxb, float* dgamma_b, float* dbeta_b,
                          int f_lo, int f_hi, const float* f_mi, const float* f_gamma, const float* f_beta,
                          float f_post, double* f_partials, void* stream);
/* t = T_a(a) + T_b(b);  partials[nblk][C][2] = {sum, sum^2} of act(t) with LeakyReLU(slope): the residual join
 * followed by act -> BatchNorm of Block3d / ResPath3d (mulresunet.py:92-94,109-111) in one pass. */
int dpi_chain_add_stats(const float* a, const float* chain_a, const float* b, const float* chain_b, int C,
                        size_t V, float slope, float* t, double* partials, void* stream);
/* dx = dy * act'(x) with act(v) = v>0 ? v : slope*v  (x = the activation INPUT or OUTPUT: same sign) */
int dpi_lrelu_bwd(const float* dy, const float* x, float slope, size_t n, float* dx, void* stream);
/* The other activations of the reference's get_activation (architectures/base.py:97-114): ELU (alpha = 1), Tanh, Sigmoid.
 * Backward takes the activation OUTPUT y. */
#define DPI_ACT_ELU 1
#define DPI_ACT_TANH 2
#define DPI_ACT_SIGMOID 3
int dpi_act_fwd(const float* x, size_t n, int kind, float* y, void* stream);
int dpi_act_bwd(const float* dy, const float* y, size_t n, int kind, float* dx, void* stream);
/* y = a + b */
int dpi_add(const float* a, const float* b, size_t n, float* y, void* stream);
/* per-channel sum: out[c] = sum_v x[c][v]   (bias gradients) ; ws: double[dpi_stat_blocks*C*2] */
int dpi_channel_sum(const float* x, int C, size_t V, double* ws, float* out, void* stream);

/* ---------------------------------------------------------------- bf16 storage of activations (ABI 400) ---------
 * `_io` twins of the entry points above that read or write activation tensors.  `io` says how those tensors live in HBM:
 *   DPI_STORE_FWD_BF16   the FORWARD tensors of the call (activations: conv outputs, join results, block outputs) are bf16
 *   DPI_STORE_GRAD_BF16  the GRADIENT tensors of the call (dy, dx: gradients of such activations) are bf16
 * Per call, [F] / [G] below mark which arguments each bit covers.  Everything else — chains, mean / invstd, gamma / beta and their
 * gradients, statistic partials (double), workspaces — keeps its type.  Arithmetic is fp32 (double for the reductions); a bf16 element
 * is widened exactly on load and a result is rounded to nearest-even on store; statistics emitted together with a stored tensor
 * describe the ROUNDED values.  io = 0 is the fp32 entry point.  Unknown bits are rejected (DPI_E_ARG). */
#define DPI_STORE_FWD_BF16 1u
#define DPI_STORE_GRAD_BF16 2u
int dpi_channel_stats_io(const float* x /*F*/, const float* chain, int C, size_t V, double* partials, unsigned io, void* stream);
int dpi_chain_apply_io(const float* x /*F*/, const float* chain, int C, size_t V, float* y /*F*/, unsigned io, void* stream);
int dpi_chain_add_stats_io(const float* a /*F*/, const float* chain_a, const float* b /*F*/, const float* chain_b, int C, size_t V,
                           float slope, float* t /*F*/, double* partials, unsigned io, void* stream);
int dpi_bn_bwd_reduce_io(const float* dy /*G*/, const float* x /*F*/, const float* mean_invstd, const float* gamma, const float* beta,
                         const float* in_chain, float pre_slope, float post_slope, int C, size_t V, double* partials, unsigned io,
                         void* stream);
int dpi_bn_bwd_apply_io(const float* dy /*G*/, const float* x /*F*/, const float* mean_invstd, const float* gamma, const float* beta,
                        const float* in_chain, float pre_slope, float post_slope, const double* partials, int nblk, int C, size_t V,
                        float* dx /*G*/, float* dgamma, float* dbeta, unsigned io, void* stream);
int dpi_bn_bwd_apply_fork_io(const float* dy /*G*/, const float* x /*F*/, const float* mean_invstd, const float* gamma, const float* beta,
                             const float* in_chain, float pre_slope, float post_slope, const double* partials, int nblk, int C,
                             size_t V, float* dx /*G*/, float* dgamma, float* dbeta, const float* xa /*F*/, const float* mi_a,
                             const float* gamma_a, const float* beta_a, const float* chain_a, float post_a, double* partials_a,
                             const float* xb /*F*/, const float* mi_b, const float* gamma_b, const float* beta_b, const float* chain_b,
                             float post_b, double* partials_b, unsigned io, void* stream);
int dpi_bn_bwd_apply_dual_io(const float* dy /*G*/, int nblk, int C, size_t V, const float* xa /*F*/, const float* mi_a,
                             const float* gamma_a, const float* beta_a, const float* chain_a, float post_a,
                             const double* partials_a, float* dxa /*G*/, float* dgamma_a, float* dbeta_a, const float* xb /*F*/,
                             const float* mi_b, const float* gamma_b, const float* beta_b, const float* chain_b,
                             float post_b, const double* partials_b, float* dxb /*G*/, float* dgamma_b, float* dbeta_b,
                             int f_lo, int f_hi, const float* f_mi, const float* f_gamma, const float* f_beta,
                             float f_post, double* f_partials, unsigned io, void* stream);
int dpi_upsample2x_fwd_io(const float* x /*F*/, const float* chain, int C, int D, int H, int W, int Do, int Ho, int Wo, int linear,
                          float* y /*F*/, unsigned io, void* stream);
/* ws (fp32, dpi_upsample2x_bwd_ws_floats) keeps its type */
int dpi_upsample2x_bwd_io(const float* dy /*G*/, int C, int D, int H, int W, int Do, int Ho, int Wo, int linear, float* dx /*G*/,
                          float* ws, unsigned io, void* stream);
/* z stays fp32 (it is the fixed network input of main.py:62-64); out [F] is the perturbed input the first layers read */
int dpi_noise_add_io(const float* z, size_t n, float std, uint64_t seed, const uint64_t* step_ptr, float* out /*F*/, unsigned io,
                     void* stream);

/* ---------------------------------------------------------------- up-sampling / concat ----------
 * Replaces nn.Upsample(scale_factor=2, mode=nearest|bilinear|trilinear, align_corners=False)
 * (mulresunet.py:168,242; skip.py:128,231) fused with the centre-crop of Concat/Concat3D
 * (base.py:297-319,333-359): the output may be cropped to (Do,Ho,Wo) <= 2*(D,H,W), offset 0.
 * 2-D: D = Do = 1 (no scaling along D when D == Do == 1).
 */
int dpi_upsample2x_fwd(const float* x, const float* chain, int C, int D, int H, int W, int Do, int Ho,
                       int Wo, int linear, float* y, void* stream);
/* ws (optional, linear only): dpi_upsample2x_bwd_ws_floats floats; with it the adjoint runs as three separable,
 * coalesced passes (W, H, D) instead of a 64-tap gather per voxel.  NULL = gather. */
size_t dpi_upsample2x_bwd_ws_floats(int C, int D, int H, int W, int Do, int Ho, int Wo, int linear);
int dpi_upsample2x_bwd(const float* dy, int C, int D, int H, int W, int Do, int Ho, int Wo, int linear,
                       float* dx, float* ws, void* stream);
/* centre-crop copy [C][D][H][W] -> [C][Do][Ho][Wo] starting at (od,oh,ow); and its adjoint (zero-fill) */
int dpi_crop_copy(const float* x, int C, int D, int H, int W, int od, int oh, int ow, int Do, int Ho,
                  int Wo, float* y, void* stream);
int dpi_crop_copy_bwd(const float* dy, int C, int D, int H, int W, int od, int oh, int ow, int Do,
                      int Ho, int Wo, float* dx, void* stream);

/* ---------------------------------------------------------------- plain 2-D UNet extras ----------
 * Replaces nn.MaxPool2d(2, 2) (unet.py:42) and nn.ConvTranspose2d(Cin, Cout, 4, stride=2, padding=1) (unet.py:59) with
 * their backward passes.  x: [C][H][W]; pooled / transposed outputs are [C][H/2][W/2] and [Cout][2H][2W];
 * deconv weight layout is torch's [Cin][Cout][4][4]. */
int dpi_maxpool2x2_fwd(const float* x, int C, int H, int W, float* y, void* stream);
int dpi_maxpool2x2_bwd(const float* dy, const float* x, int C, int H, int W, float* dx, void* stream);
int dpi_deconv4x4s2_fwd(const float* x, const float* w, const float* bias, int Cin, int Cout, int H, int W,
                        float* y, void* stream);
int dpi_deconv4x4s2_bwd_data(const float* dy, const float* w, int Cin, int Cout, int H, int W, float* dx,
                             void* stream);
int dpi_deconv4x4s2_bwd_weight(const float* x, const float* dy, int Cin, int Cout, int H, int W, float* dw,
                               void* stream);

/* ---------------------------------------------------------------- loss + metrics ----------------
 * Replaces main.py:161-167: loss = mean(|out*m - img*m|) (kind 0, L1) or mean((.)^2) (kind 1, MSE)
 * over all n elements; dout = dloss/dout * grad_scale; plus the sums for snr/pcorr
 * (utils/metrics.py:15,32-36).  ws: double[dpi_loss_ws_doubles(n)].
 * result (device, double[8]): {loss, snr_dB, pcorr, sum_t2, sum_(t-o)^2, sum_o, sum_t, reserved}.
 */
size_t dpi_loss_ws_doubles(size_t n);
int dpi_masked_loss(const float* out, const float* img, const float* mask, size_t n, int kind,
                    float grad_scale, float* dout, double* ws, double* result, void* stream);

/* ---------------------------------------------------------------- optimiser ---------------------
 * Replaces torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8).step() (main.py:200,213) for a list of
 * tensors in ONE launch.  ptrs: device array of {p, g, m, v} pointers per tensor; sizes: element counts;
 * step_lr (device, float[2]): {step count as float (already incremented), lr}.  `active` (device int,
 * may be NULL): when *active == 0 the update is skipped (device-side early stop).  Betas/eps are doubles so that
 * 1-beta and the bias corrections are derived in double and then rounded, exactly as torch does on the host.
 */
typedef struct { float* p; const float* g; float* m; float* v; } dpi_adam_tensor;
int dpi_adam_multi(const dpi_adam_tensor* tensors, const int64_t* sizes, int ntensors,
                   const float* step_lr, double beta1, double beta2, double eps, const int* active,
                   void* stream);

/* ---------------------------------------------------------------- device-resident loop control --
 * Replaces the host-side bookkeeping of main.py:165-182,214-217 so that a captured hipGraph of one iteration can be
 * replayed without host synchronisation: appends {loss, snr, pcorr, lr} to hist[iter], raises *improved when
 * loss <= loss_min (best-output tracking), runs ReduceLROnPlateau(mode=min, rel threshold) on step_lr[1] and
 * EarlyStopping(percentage) / NaN stop on *active.  state: double[8], zero-initialised except state[2] = +inf.
 * dpi_copy_if copies src -> dst only when *flag != 0 (keeps out_best on the device). */
int dpi_loop_control(const double* metrics, double* state, double* hist, int max_iters, float* step_lr,
                     int* active, int* improved, int use_plateau, double factor, double threshold,
                     int patience, double min_lr, double lr_eps, int es_patience, double es_min_delta,
                     void* stream);
int dpi_copy_if(const int* flag, const float* src, float* dst, size_t n, void* stream);

/* ---------------------------------------------------------------- input perturbation ------------
 * Replaces main.py:148-150: out = z + std * N(0,1), Philox4x32-10 + Box-Muller, counter = element index,
 * key = (seed, *step_ptr) so that every replay of a captured graph draws fresh noise.
 * step_ptr (device, uint64) may be NULL (then step = 0).
 */
int dpi_noise_add(const float* z, size_t n, float std, uint64_t seed, const uint64_t* step_ptr,
                  float* out, void* stream);
int dpi_fill_normal(float* out, size_t n, float mean, float std, uint64_t seed, uint64_t stream_id,
                    void* stream);
/* dpi_noise_add for a z that IS such a fill (z = dpi_fill_normal(mean 0, z_std, z_seed, z_stream_id), untouched since): the kernel re-draws z
 * from its Philox stream instead of reading it — out = z + std * N(0,1) with the same bits as dpi_noise_add_io(z, ...), minus the HBM read of
 * z (64 channels at full resolution, every iteration).  io: DPI_STORE_FWD_BF16 = out is bf16. */
int dpi_noise_add_regen_io(size_t n, float z_std, uint64_t z_seed, uint64_t z_stream_id, float std, uint64_t seed,
                           const uint64_t* step_ptr, float* out /*F*/, unsigned io, void* stream);
/* Replaces utils/processing.py:34-67 (ConvolveKernel_1d: grouped conv_transposeNd with a 1-D kernel along the time
 * axis, used by --filter_noise_with_wavelet / --lowpass_*): y[c][t][s] = sum_k taps[k] * x[c][t + K/2 - k][s],
 * x: [C][T][S] with S = product of the remaining spatial axes; K odd; taps on the device. */
int dpi_fir_axis0(const float* x, const float* taps, int K, int C, int T, size_t S, float* y, void* stream);
/* y += a * x  (data-forgetting term of main.py:153-154) */
int dpi_axpy(float a, const float* x, size_t n, float* y, void* stream);

/* ---------------------------------------------------------------- patch reassembly --------------
 * Replaces PatchExtractor.reconstruct (utils/patch_extractor.py:395-428): overlap-add of one patch
 * (pd,ph,pw) at origin (od,oh,ow) into acc[D][H][W]; dpi_overlap_normalize divides by the analytic hit
 * count of the regular window grid and by `gain` (data.py:116).
 */
int dpi_overlap_add(const float* patch, int pd, int ph, int pw, int od, int oh, int ow, float* acc,
                    int D, int H, int W, void* stream);
int dpi_overlap_normalize(float* acc, int D, int H, int W, int pd, int ph, int pw, int sd, int sh,
                          int sw, float gain, void* stream);

/* ---------------------------------------------------------------- anti-aliasing operators -------
 * Linear operators of the anti-aliasing add-on (BASELINE configs[3]); `adjoint` != 0 applies the exact transpose, which is
 * what the backward of a loss term built on the operator needs (reference operators/base.py:53-67 `dottest` is the unit test).
 *
 * dpi_diff_axis replaces utils/processing.py:139-181 first_derivative (stencil 0 forward, 1 backward, 2 centered) and
 * second_derivative (stencil 3) along the middle axis of a contiguous [outer][n][inner] view, and with stencil 0, spacing 1 on the
 * H axis of BCHW it is operators/derivative.py:8-21 VerticalGrad.forward / .adjoint.   x != y. */
int dpi_diff_axis(const float* x, size_t outer, int n, size_t inner, int stencil, float spacing, int adjoint, float* y,
                  void* stream);
/* Replaces utils/slopes.py:51-105 directional_laplacian / Hale2D.forward on [N][H][W] planes with coefficient planes
 * a = cos^2, b = -cos sin, c = sin^2 of the dips (same shape as x):  y = -(Dh(a Dv x + b Dh x) + Dv(b Dv x + c Dh x)). */
int dpi_hale2d(const float* x, const float* a, const float* b, const float* c, size_t N, int H, int W, int adjoint, float* y,
               void* stream);
/* Replaces utils/slopes.py:19-22 (forward-difference gradients and their products) and :35-46 (eigen-analysis: dip angle phi
 * with NaN -> 0, anisotropy 1 - l2/l1); the Gaussian smoothing in between (:25-32) is dpi_fir_axis0 along H and W. */
int dpi_structure_tensor(const float* x, size_t N, int H, int W, float dv, float dh, float* gvv, float* gvh, float* ghh,
                         void* stream);
int dpi_dips(const float* gvv, const float* gvh, const float* ghh, size_t n, float* phi, float* anisotropy, void* stream);

/* ---------------------------------------------------------------- POCS regulariser --------------
 * Replaces utils/pocs.py:5-19 (threshold / compute_threshold: out = max(x) * scale with scale = perc/100, kept on the device;
 * y = x * ((x > th) + (x < -th)) on the real view of the spectrum) and :80-84 (y = weighted_data + weighted_mask * x).
 * The transform between them (reference: torch.rfft / irfft, main_pocs.py:156-157) is the rocFFT library call. */
size_t dpi_max_ws_floats(size_t n);
int dpi_scaled_max(const float* x, size_t n, float scale, float* ws, float* out, void* stream);
int dpi_threshold(const float* x, size_t n, const float* thresh, float* y, void* stream);
int dpi_pocs_project(const float* x, const float* wdata, const float* wmask, size_t n, float* y, void* stream);

/* ---------------------------------------------------------------- BatchNorm backward of a residual join in two passes (ABI 403) -----
 * Replaces autograd's chain through the tail of Block3d / ResPath3d (reference architectures/mulresunet.py:85-96, 109-112):
 *     y = BN(act_pre(t)),   t = act_a(BN_a(T_a(xa))) + act_b(BN_b(T_b(xb)))      [per channel; T_* optional value-only input chains]
 * and, on the channel range [f_lo, f_hi) of side B, one more BatchNorm BN_f whose OUTPUT (after act_f) is xb's chain input there and whose
 * input is the raw xb (Block3d: the third 3x3x3 layer's BatchNorm under bn1).  Given dy = dL/dy it writes dxa = dL/d(T_a(xa)),
 * dxb = dL/d(T_b(xb)) on the channels outside the fork range, dxf [f_hi - f_lo][V] = dL/d(xb) through BN_f on the fork range, and the
 * (dgamma, dbeta) pairs of BN, BN_a, BN_b as dgb [6][C] = rows {dgamma, dbeta, dgamma_a, dbeta_a, dgamma_b, dbeta_b} and of BN_f as dgb_f
 * [2][f_hi - f_lo] (every row a contiguous gradient vector).  The same per-element expressions as dpi_bn_bwd_reduce / _apply_fork / _apply_dual / _apply in sequence (13.6 tensor
 * passes); the nested per-channel sums are expanded so that ONE reduction pass (24 sums per channel, double precision) and ONE apply
 * pass suffice (10.5 passes, the intermediate gradient dL/dt is never stored).  io (DPI_STORE_FWD_BF16: t, xa, xb; DPI_STORE_GRAD_BF16: dy, dxa,
 * dxb, dxf): a bf16 element is widened on load, results are rounded on store; the nested sums describe the UNROUNDED intermediate gradients
 * (the rounds 1-4 sequence took them of the bf16-rounded stored ones), so with bf16 tensors the outputs equal round_bf16 of what the fp32
 * call computes on the widened operands.
 * ws: dpi_join_bwd_ws_doubles(C, V) doubles; coef: C x 8 floats of scratch (the per-channel constants the apply pass reads).
 * mi* = {mean[C], invstd[C]} as dpi_bn_finalize wrote them; post_* = slope of the activation BEHIND that BatchNorm (1 = none);
 * pre = slope of the activation in FRONT of the top BatchNorm.  t may be NULL (see dpi_chain_add_apply below). */
size_t dpi_join_bwd_ws_doubles(int C, size_t V);
int dpi_join_bwd(const float* dy, const float* t, const float* mi, const float* gamma, const float* beta, float pre, int C, size_t V,
                 const float* xa, const float* mi_a, const float* gamma_a, const float* beta_a, const float* chain_a, float post_a,
                 const float* xb, const float* mi_b, const float* gamma_b, const float* beta_b, const float* chain_b, float post_b,
                 const float* fwd_chain_a, const float* fwd_chain_b,
                 int f_lo, int f_hi, const float* f_mi, const float* f_gamma, const float* f_beta, float f_post,
                 double* ws, float* coef, float* dxa, float* dxb, float* dxf, float* dgb, float* dgb_f, unsigned io, void* stream);
/* The join itself without a stored t (ABI 403): dpi_chain_add_stats(.., t = NULL, ..) takes the statistics of act(T_a(a) + T_b(b)) only,
 * dpi_chain_add_apply writes y = T_out(T_a(a) + T_b(b)) (T_out = the BatchNorm dpi_bn_finalize made of those statistics), and dpi_join_bwd
 * with t = NULL recomputes t from xa, xb through fwd_chain_a / fwd_chain_b (= chain_a / chain_b of the forward calls): the three form the
 * sum with the same expression, so all of them see the same fp32 values.  Saves one tensor write forward and two reads backward per join. */
int dpi_chain_add_apply(const float* a, const float* chain_a, const float* b, const float* chain_b, const float* chain_out, int C, size_t V,
                        float* y, unsigned io, void* stream);

/* ---------------------------------------------------------------- packed-weight scratch (ABI 401) --------------------------------
 * No reference counterpart (torch's convolutions own their workspaces the same way).  In the bf16 arithmetic modes the 3x3(x3) stride-1
 * stencil kernel reads its weights from a bf16 copy in MFMA-fragment order that a small kernel writes in front of EVERY launch, on the
 * launch's stream, into the layer's slot of a library-owned scratch: one slot per (weight pointer, shape, direction), carved from 64 MB
 * hipMalloc chunks at the layer's first launch and kept.  Consequences for a caller:
 *   - the weight tensor itself is never cached: whatever wrote it before the launch is what the launch uses;
 *   - a layer's FIRST launch must not happen inside a stream capture unless a chunk with room already exists (hipMalloc is not
 *     capturable): run one eager iteration before capturing, as for any captured workload;
 *   - the scratch grows with the number of distinct (device, weight pointer, shape) triples seen, up to 4 GB per device; beyond that
 *     launches fail with DPI_E_LAUNCH until slots are handed back.  A caller that builds a new network per patch (reference
 *     main.py:286: 343 patches per configs[2] volume) calls dpi_pack_forget() for the weight tensors of the network it drops.
 * dpi_pack_scratch_bytes(): bytes currently held, all devices.
 * dpi_pack_forget(w) (ABI 402): the slots of weight tensor `w` go to a per-device free list, from which the next layer that needs a slot of
 *     the same size takes it; returns the number of slots handed back.  The caller guarantees that no launch that reads them is in flight or
 *     sits in a captured graph that will still be replayed (deep_prior_interpolation_amd.main.Interpolator calls it when a synchronised
 *     patch's network is replaced).
 * dpi_pack_release(): synchronises every device that owns a chunk, then frees every chunk and forgets every slot; the caller guarantees
 *     that no graph captured before the call is launched after it (its kernels hold slot addresses). */
size_t dpi_pack_scratch_bytes(void);
size_t dpi_pack_slot_count(void);      /* ABI 402: live (device, weight tensor, shape) slots — what dpi_pack_forget() brings down */
int dpi_pack_forget(const void* w);
int dpi_pack_release(void);

#ifdef __cplusplus
}
#endif
#endif /* DPI_HIP_H */
