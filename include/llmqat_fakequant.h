/*
 * llmqat_fakequant.h -- C ABI of the MI355X (gfx950) fake-quantization kernel library.
 *
 * This is the drop-in boundary for LLM-QAT's fake-quant hot path.  The reference has no
 * native layer: its "kernels" are chains of eager ATen ops inside two
 * torch.autograd.Functions.  Each entry point below replaces one of those chains; the
 * reference lines are cited per function (paths relative to the LLM-QAT repo root).
 *
 * Conventions
 *   - plain pointers + sizes only; no torch / HIP types in signatures (`stream` is a
 *     hipStream_t passed as void*; NULL = the legacy default stream).
 *   - every pointer is DEVICE memory owned by the caller (PyTorch's caching allocator);
 *     the library never allocates, frees or retains a pointer, and never synchronises.
 *     All work is enqueued on `stream`, so every call is hipGraph-capturable.
 *   - tensors are contiguous; a "row" is the unit that shares one scale:
 *         ndim <= 3 : rows = numel / shape[-1], cols = shape[-1]    (utils_quant.py:53-59)
 *         ndim == 4 : rows = d0*d1, cols = d2*d3                    (utils_quant.py:60-68)
 *         layerwise : rows = 1, cols = numel                        (utils_quant.py:50-51)
 *   - return value: 0 on success, negative FQ_ERR_* otherwise; fq_last_error() gives a
 *     thread-local message.  Nothing throws across the boundary.  FQ_ERR_LAUNCH is the status hipLaunchKernel() RETURNED for this
 *     call's own launches: the library neither reads nor clears the thread's hipGetLastError() slot, so an error another library
 *     left pending there is not hidden, and not mistaken for this library's.
 *   - stateless and re-entrant: safe from the autograd thread, under
 *     torch.utils.checkpoint recompute and inside DDP/FSDP hooks.
 *   - arithmetic: bit-exact replay of the reference's op-by-op rounding in the tensor
 *     dtype (see DESIGN.md "Numerics"); `sem` picks how Python scalars met the tensor.
 */
#ifndef LLMQAT_FAKEQUANT_H
#define LLMQAT_FAKEQUANT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FQ_ABI_VERSION 5 /* 5: + the *_v entry points: rows that do not follow one another in memory (fq_rows_view);  2: + multi-tensor launches, export, row scales, fq_w12_fwd_rows; 3: the STE mask is a plain row bitmap;
                            4: fq_sym_fwd_autocast takes `sem` (and the autocast modes of pair / multi / export / row_scales honour it),
                               launch status from hipLaunchKernel's return value (the hipGetLastError slot is left alone), fq_qlinear_fwd (an experiment with test hooks in its signature) left the library, fq_w12_fwd_rows sums
                               in ATen's own order (no `sem`) */

/* element types */
#define FQ_DTYPE_F32 0
#define FQ_DTYPE_BF16 1
#define FQ_DTYPE_F16 2
/* float64 (the reference has no dtype restriction): a correctness path in double arithmetic, served by fq_sym_fwd / fq_asym_fwd
 * (+ _debug; no row bounds), fq_ste_bwd and fq_w12_fwd; every other entry point answers FQ_ERR_DTYPE for it. */
#define FQ_DTYPE_F64 3

/* scalar semantics: how a Python scalar ADDED to a 16-bit tensor is treated (`max + 1e-6`, `alpha + 1e-8`), and Asym's `.div(S)`.
 * The two differ only where 1e-6 is within a rounding step of the row's |max|: bf16 rows with |max| below ~3e-4, fp16 rows with
 * |max| in [2^-13, 2^-12) (1e-6 lands on a tie there); and for fp32 Asym.  Under the autocast arithmetic (fq_sym_fwd_autocast and
 * the autocast modes of pair / multi / export / row_scales) `sem` governs `max + 1e-6` alone -- everything behind it is fp32 -- and
 * a real torch.autocast("cuda") run is FQ_SEM_DEVICE_EAGER by definition. */
#define FQ_SEM_CPU_EAGER 0    /* canonical: what ATen's CPU kernels do; pinned by tests/golden */
#define FQ_SEM_DEVICE_EAGER 1 /* what ATen's GPU kernels do (fp32 "opmath" scalars, div-by-scalar = mul by 1/S) */

/* error codes */
#define FQ_OK 0
#define FQ_ERR_DTYPE (-1)
#define FQ_ERR_BITS (-2)
#define FQ_ERR_SHAPE (-3)
#define FQ_ERR_NULL (-4)
#define FQ_ERR_WORKSPACE (-5)
#define FQ_ERR_LAUNCH (-6)
#define FQ_ERR_ARG (-7)
#define FQ_ERR_UNSUPPORTED (-8) /* shape/alignment not served by this entry point: use the general one */

int fq_version(void);               /* == FQ_ABI_VERSION */
const char* fq_build_info(void);    /* e.g. "llmqat_fakequant abi 1, gfx950, hip 7.2" */
const char* fq_last_error(void);    /* thread-local; "" if the last call on this thread succeeded */

/*
 * Bytes of scratch the row-wise forwards need for this shape (0 for rows that fit the
 * single-pass register-resident kernels; 8 bytes per row for the two-pass path that very
 * long rows -- e.g. layerwise -- take).  The caller allocates it; contents are don't-care.
 */
size_t fq_rowwise_workspace_bytes(int64_t rows, int64_t cols, int dtype);

/*
 * SymQuantizer.forward -- models/utils_quant.py:37-74
 *   m = max|x| per row; s = reciprocal(m + 1e-6) * (2^(bits-1)-1); y = round(x*s) / (s + 1e-6)
 * Replaces 9 ATen kernels (abs, max, add, reciprocal, mul, mul, round, add, div) by one pass:
 * x is read once, y written once.  No clamp (the reference has none): 8-bit bf16 bins reach +-128.
 *   x, y      [rows, cols] dtype, contiguous; y may not alias x
 *   bits      1..31 (the callers use 3..16; with 1 bit the only level is 0 -- qmax = 2^0 - 1 -- and everything quantizes to +-0,
 *             as in the reference, which the KV hooks' `kv_bits < 32` gate can reach)
 *   row_bounds_out  optional float[rows][2] = {+m, -m}: bounds of the row's values, which
 *             fq_ste_bwd_rows can use to skip re-reading x (pass NULL if unused)
 */
int fq_sym_fwd(const void* x, void* y, int64_t rows, int64_t cols, int bits, int dtype, int sem,
               float* row_bounds_out, void* workspace, size_t workspace_bytes, void* stream);

/*
 * AsymQuantizer.forward -- models/utils_quant.py:96-149
 *   beta = min x, alpha = max x - min x per row; n = (x-beta)/(alpha+1e-8);
 *   y = round(n*(2^bits-1)) / (2^bits-1) * (alpha+1e-8) + beta
 * Replaces ~13 ATen kernels (incl. the duplicated min-reduce :119/:124).
 *   row_bounds_out  optional float[rows][2] = {max, min}
 */
int fq_asym_fwd(const void* x, void* y, int64_t rows, int64_t cols, int bits, int dtype, int sem,
                float* row_bounds_out, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Test/diagnostic variants: same arithmetic, additionally emit the integer bin index of
 * every element (int32; NaN -> INT32_MIN, +-Inf -> +-INT32_MAX) and the per-row scale
 * terms as fp32:  Sym: scale_out[rows] = s ;  Asym: scale_out[rows][2] = {alpha, beta}.
 * Any of idx_out / scale_out may be NULL.
 */
int fq_sym_fwd_debug(const void* x, void* y, int32_t* idx_out, float* scale_out, int64_t rows, int64_t cols,
                     int bits, int dtype, int sem, void* workspace, size_t workspace_bytes, void* stream);
int fq_asym_fwd_debug(const void* x, void* y, int32_t* idx_out, float* scale_out, int64_t rows, int64_t cols,
                      int bits, int dtype, int sem, void* workspace, size_t workspace_bytes, void* stream);

/*
 * SymQuantizer.backward / AsymQuantizer.backward (identical) -- models/utils_quant.py:77-87, :152-162
 *   gx = g.clone(); gx[x >= hi] = 0; gx[x <= lo] = 0        (NaN x passes g)
 * Replaces 5 ATen kernels (clone, ge, index_put_, le, index_put_).  lo/hi are first rounded to
 * the tensor dtype, as the reference's comparison does.  g, x, gx: n elements of `dtype`.
 */
int fq_ste_bwd(const void* g, const void* x, void* gx, int64_t n, float lo, float hi, int dtype, void* stream);

/*
 * Same result as fq_ste_bwd on a [rows, cols] tensor, but consults the per-row bounds a
 * forward call recorded: rows whose bounds lie strictly inside (lo, hi) cannot be masked,
 * so their x is not re-read (gx = g: 4 instead of 6 bytes/element for bf16).  Rows with
 * NaN/out-of-range bounds take the normal path.  row_bounds must describe the same x.
 */
int fq_ste_bwd_rows(const void* g, const void* x, void* gx, int64_t rows, int64_t cols, float lo, float hi,
                    const float* row_bounds, int dtype, void* stream);

/*
 * Training-mode pair: the forward also records, for the backward, (a) the per-row bounds and (b) a
 * 1-bit-per-element STE mask (only written for rows whose bounds do not already prove that nothing
 * is clipped), so the backward never re-reads x:
 *     forward  bf16: read x 2 + write y 2 (+ 1/8 mask)   backward: read g 2 (+ 1/8 mask) + write gx 2   B/element
 * instead of 4 + 6.  Results are bit-identical to fq_*_fwd + fq_ste_bwd.  The autograd Functions need not
 * keep `input` alive for the backward in this mode (the reference saves it, utils_quant.py:45).
 *
 * Mask layout (ABI 3; one layout for every producer and consumer): a plain bitmap per row.  Row r starts at byte
 *                    r * 8 * ceil(cols / 64); bit (j % 8) of its byte (j / 8) is the flag of element j of the row,
 *                    1 = the backward zeroes that gradient (x >= hi || x <= lo; a NaN x has flag 0).  Rows whose recorded
 *                    bounds lie strictly inside (lo, hi) are NOT written (the backward decides from the bounds first).
 * fq_ste_mask_bytes  size of the mask buffer for this shape = rows * 8 * ceil(cols / 64); 0 if the shape is not served
 *                    (rows that do not fit the register-resident kernels, or cols not a multiple of a 16-byte vector):
 *                    use fq_*_fwd + fq_ste_bwd[_rows] then.  Contents need no initialisation.
 * fq_*_fwd_train     lo/hi = the STE clip (clip_val[0], clip_val[1]); row_bounds_out and mask_out required.
 *                    Returns FQ_ERR_UNSUPPORTED if x/y are not 16-byte aligned.
 * fq_ste_bwd_mask    SymQuantizer.backward / AsymQuantizer.backward (utils_quant.py:77-87, :152-162) from
 *                    (row_bounds, mask) of the matching forward; lo/hi must be the same values.
 *                    IN PLACE: gx may be the same pointer as g (also per tensor in fq_ste_bwd_mask_pair / _multi).  The
 *                    gradient is then masked where it stands, and a row whose bounds prove that nothing is clipped is not
 *                    touched at all: for a weight (whose rows practically never reach the clip) the STE backward moves no
 *                    bytes -- the reference's `grad_output.clone()` (:84) exists only to be mutated by :85-86.  An in-place
 *                    tensor gets one workgroup per 256 rows (it reads their bounds and walks the clippable ones), so it is
 *                    meant for tensors whose rows rarely clip; an activation's gradient belongs in a copying call.
 */
size_t fq_ste_mask_bytes(int64_t rows, int64_t cols, int dtype);
int fq_sym_fwd_train(const void* x, void* y, int64_t rows, int64_t cols, int bits, int dtype, int sem, float lo, float hi,
                     float* row_bounds_out, void* mask_out, size_t mask_bytes, void* stream);
int fq_asym_fwd_train(const void* x, void* y, int64_t rows, int64_t cols, int bits, int dtype, int sem, float lo, float hi,
                      float* row_bounds_out, void* mask_out, size_t mask_bytes, void* stream);
int fq_ste_bwd_mask(const void* g, void* gx, int64_t rows, int64_t cols, float lo, float hi, const float* row_bounds,
                    const void* mask, size_t mask_bytes, int dtype, void* stream);

/*
 * SymQuantizer.forward as it executes under CUDA autocast -- torch.autocast("cuda", dtype=bf16|fp16), which is how
 * LLM-QAT trains (`--bf16 True`, utils/kd_trainer.py:106).  `reciprocal` is on autocast's fp32 list, so for a 16-bit
 * input  s = (2^(b-1)-1) / (max + 1e-6)  (models/utils_quant.py:71) comes back in fp32 and every later op of :72 is
 * promoted:  t1 = round_to_dtype(max + 1e-6);  s = (1/t1)*qmax;  y = round(x*s) / (s + 1e-6)  in fp32; the reference
 * returns an fp32 tensor.  (fp32 inputs are unaffected by autocast: use fq_sym_fwd.  AsymQuantizer and the 1-/2-bit
 * branches contain no autocast-listed op: use fq_asym_fwd / fq_w12_fwd.)
 *   dtype     FQ_DTYPE_BF16 or FQ_DTYPE_F16 (the input)
 *   sem       FQ_SEM_DEVICE_EAGER is what the device computes under torch.autocast("cuda"); FQ_SEM_CPU_EAGER rounds the 1e-6 of
 *             `max + 1e-6` to the tensor dtype first (the reference's CPU behaviour: tests/golden/autocast.npz holds both)
 *   wide_out  1: y is fp32 [rows, cols], exactly what the reference returns (KV hooks, direct callers)
 *             0: y has the input dtype = that fp32 result rounded once to it -- bit-identical to what F.linear's own
 *                autocast cast makes of it next, so QuantizeLinear can skip the fp32 round trip (2 instead of 4+4+2 B/elem)
 *   row_bounds_out / mask_out  optional training-mode outputs, as in fq_sym_fwd_train (mask_out needs row_bounds_out).
 *             wide_out = 0: the backward is fq_ste_bwd_mask / fq_ste_bwd on the input dtype.
 *             wide_out = 1: the gradient of the fp32 result is fp32 and the autograd engine casts it to the input dtype
 *             (grad of a bf16 leaf is bf16); fq_ste_bwd_mask_wide does cast + masking in one pass (the mask is the same
 *             row bitmap whatever wide_out is, so fq_ste_bwd_mask on an already cast gradient serves it too).  Without a
 *             mask: cast grad_output, then fq_ste_bwd_rows / fq_ste_bwd.
 *   workspace  fq_rowwise_workspace_bytes(rows, cols, dtype) bytes (only rows longer than 32768 elements use it)
 * Returns FQ_ERR_UNSUPPORTED when a mask is requested for a shape fq_ste_mask_bytes rejects or for misaligned rows.
 */
int fq_sym_fwd_autocast(const void* x, void* y, int64_t rows, int64_t cols, int bits, int dtype, int sem, int wide_out, float lo, float hi,
                        float* row_bounds_out, void* mask_out, size_t mask_bytes, void* workspace, size_t workspace_bytes,
                        void* stream);

/*
 * QuantizeLinear.forward (models/utils_quant.py:195-248) fake-quantizes its weight [out, in] (w_bits, per output
 * channel) and its input [tokens, in] (a_bits, per token) at the same moment, and both reduce over `in`: the same row
 * length, hence the same launch shape.  These entry points do the two tensors in ONE launch (each launch carries
 * ~2.8 us of fixed cost on MI355X), with results bit-identical to two separate calls.
 *   tensor 0 / tensor 1  same dtype and `cols`; own rows, bit width, outputs.  row_bounds / mask are optional per
 *                        tensor (training mode; a mask needs its row_bounds).
 *   autocast             0: the arithmetic of fq_sym_fwd_train;  1: of fq_sym_fwd_autocast with wide_out = 0;
 *                        2: with wide_out = 1 (y0 / y1 are fp32) -- the K and V hooks of the attention block
 *                        (models/modeling_llama_quant.py:320-327: two SymQuantizer calls on [bsz, q_len, hidden] tensors)
 * Only the register-resident kernels serve pairs: FQ_ERR_UNSUPPORTED for misaligned rows or rows longer than 8192
 * 16-byte vectors (fall back to two calls).
 */
int fq_sym_fwd_pair(const void* x0, void* y0, int64_t rows0, int bits0, float* row_bounds0, void* mask0, size_t mask_bytes0,
                    const void* x1, void* y1, int64_t rows1, int bits1, float* row_bounds1, void* mask1, size_t mask_bytes1,
                    int64_t cols, int dtype, int sem, int autocast, float lo, float hi, void* stream);
/* STE backward of both tensors of a pair in one launch (F.linear's backward produces both gradients together). */
int fq_ste_bwd_mask_pair(const void* g0, void* gx0, int64_t rows0, const float* row_bounds0, const void* mask0,
                         const void* g1, void* gx1, int64_t rows1, const float* row_bounds1, const void* mask1,
                         int64_t cols, float lo, float hi, int dtype, void* stream);

/*
 * The N-tensor form of the two entry points above (1 <= n <= 4 tensors of one dtype and one row length per launch).  The
 * attention block quantizes the q/k/v weights [4096, 4096] x 3 and their shared input [tokens, 4096] at the same moment
 * (models/modeling_llama_quant.py:313,317,318), the MLP its gate/up weights and their input (:235): every tensor reduces over
 * the same `in`, so one launch (forward) and one (backward) serve the whole sibling group.  Tensor i owns rows
 * [sum(rows[<i]), +rows[i]) of the launch; results are bit-identical to separate calls.
 *   fq_fwd_tensor   x, y, rows, bits, optional row_bounds + mask (training mode; a mask needs its row_bounds)
 *   fq_bwd_tensor   g, gx, rows, row_bounds, mask (all required)
 *   autocast        as fq_sym_fwd_pair (0 / 1 / 2);  wide_grad = 1: the g are fp32 gradients of fp32 results (autocast = 2
 *                   forward) and the gx have `dtype`, as fq_ste_bwd_mask_wide
 * fq_sym_fwd_pair / fq_ste_bwd_mask_pair / fq_ste_bwd_mask_wide are the n = 2 (or 1) forms of these.
 */
typedef struct { const void* x; void* y; int64_t rows; int bits; float* row_bounds; void* mask; size_t mask_bytes; } fq_fwd_tensor;
typedef struct { const void* g; void* gx; int64_t rows; const float* row_bounds; const void* mask; } fq_bwd_tensor;
int fq_sym_fwd_multi(int n, const fq_fwd_tensor* tensors, int64_t cols, int dtype, int sem, int autocast, float lo, float hi, void* stream);
int fq_ste_bwd_mask_multi(int n, const fq_bwd_tensor* tensors, int64_t cols, float lo, float hi, int dtype, int wide_grad, void* stream);

/*
 * ---- rows that do not follow one another in memory (ABI 5) ------------------------------------------------------------------------
 * The reference accepts tensors of any strides (models/utils_quant.py:37: ATen handles them) and its elementwise results keep the
 * input's layout.  The entry points above take contiguous [rows, cols]; the `_v` forms below take, per tensor, an optional
 * fq_rows_view describing "last dimension contiguous, rows strided" layouts -- a slice / chunk() of the last dimension, the
 * transpose(0, 1) of a 3-D tensor -- so that the host needs no .contiguous() copy in front of the kernel and no copy back behind it:
 *     row r of the [rows, cols] view = index (r / n_inner, r % n_inner), starting (r / n_inner) * stride_outer +
 *     (r % n_inner) * stride_inner ELEMENTS behind the base pointer;  n_inner = 0: contiguous (the view is ignored).
 * 2-D [R, C] with row stride S: {n_inner = R, stride_outer = 0, stride_inner = S};  3-D [A, B, C] with strides (sa, sb, 1):
 * {n_inner = B, stride_outer = sa, stride_inner = sb}.  Row bounds and STE masks are indexed by the row number r and are always dense.
 * Results are bit-identical to the contiguous entry points on the gathered rows.  Rows must fit the single-pass kernels (two-pass /
 * layerwise shapes: FQ_ERR_UNSUPPORTED); the vector kernels additionally need every row start 16-byte aligned (the forward falls back
 * to its element-wise kernel otherwise; the mask / x-based backwards answer FQ_ERR_UNSUPPORTED).  rows < 2^31.
 *   fq_rowwise_fwd_v         SymQuantizer / AsymQuantizer forward (asym = 0 / 1), optionally in training mode (row_bounds_out, + mask_out),
 *                            no autocast arithmetic
 *   fq_sym_fwd_multi_v       fq_sym_fwd_multi (n = 1..4, autocast 0 / 1 / 2) with a view per x and per y
 *   fq_ste_bwd_mask_multi_v  fq_ste_bwd_mask_multi with a view per g and per gx (an in-place tensor: gx == g and equal views)
 *   fq_ste_bwd_v             the x-re-reading STE backward (fq_ste_bwd / fq_ste_bwd_rows) with views for g, x and gx; row_bounds may be NULL
 */
typedef struct { int64_t n_inner; int64_t stride_outer; int64_t stride_inner; } fq_rows_view;
typedef struct { const void* x; void* y; int64_t rows; int bits; float* row_bounds; void* mask; size_t mask_bytes; fq_rows_view xv, yv; } fq_fwd_tensor_v;
typedef struct { const void* g; void* gx; int64_t rows; const float* row_bounds; const void* mask; fq_rows_view gv, gxv; } fq_bwd_tensor_v;
int fq_rowwise_fwd_v(int asym, const void* x, const fq_rows_view* xv, void* y, const fq_rows_view* yv, int64_t rows, int64_t cols, int bits, int dtype,
                     int sem, float lo, float hi, float* row_bounds_out, void* mask_out, size_t mask_bytes, void* stream);
int fq_sym_fwd_multi_v(int n, const fq_fwd_tensor_v* tensors, int64_t cols, int dtype, int sem, int autocast, float lo, float hi, void* stream);
int fq_ste_bwd_mask_multi_v(int n, const fq_bwd_tensor_v* tensors, int64_t cols, float lo, float hi, int dtype, int wide_grad, void* stream);
int fq_ste_bwd_v(const void* g, const fq_rows_view* gv, const void* x, const fq_rows_view* xv, void* gx, const fq_rows_view* gxv, int64_t rows,
                 int64_t cols, float lo, float hi, const float* row_bounds, int dtype, void* stream);

/*
 * STE backward behind a fp32-result forward (fq_sym_fwd_autocast wide_out = 1 / fq_sym_fwd_pair autocast = 2), the
 * reference's `grad_input = grad_output.clone(); grad_input[mask] = 0` (models/utils_quant.py:83-87) followed by the
 * autograd engine's cast of that fp32 gradient to the input's dtype, in one pass:  gx = mask ? 0 : round_to_dtype(g).
 *   g0 / g1    fp32 [rows, cols];  gx0 / gx1  `dtype` (bf16 / fp16) [rows, cols];  rows1 = 0: one tensor only
 *   row_bounds / mask   as written by that forward (required)
 * FQ_ERR_UNSUPPORTED for shapes fq_ste_mask_bytes rejects, cols > 32768, or g not 16-byte / gx not 8-byte aligned.
 */
int fq_ste_bwd_mask_wide(const void* g0, void* gx0, int64_t rows0, const float* row_bounds0, const void* mask0,
                         const void* g1, void* gx1, int64_t rows1, const float* row_bounds1, const void* mask1,
                         int64_t cols, float lo, float hi, int dtype, void* stream);

/*
 * QuantizeLinear's 1- and 2-bit weight branches -- models/utils_quant.py:202-242, elementwise part:
 *   w_bits==1 : q = sc * sign(w / sc)
 *   w_bits==2 : q = sc * (round(clamp(w / sc, -0.99, 0.99) * 2 - 0.5) + 0.5) / 2
 *   out = (q - w) + w          (the forward value of `q.detach() - w.detach() + w`, :240-242)
 * `scale` = the mean-|w| scaling factor the caller reduced (mean|w| for 1 bit, 2*mean|w| for 2 bits; :205-209,
 * :219-224), in the tensor dtype: [rows] values if scale_per_row, else one value (weight_layerwise).  It is an input
 * because a floating-point SUM depends on the reduction order: keeping ATen's reduction keeps the result
 * bit-identical to the reference's; this call fuses the ~10 elementwise ATen kernels that follow it into one.
 * The gradient of these branches is the identity (no kernel).
 */
int fq_w12_fwd(const void* w, const void* scale, void* out, int64_t rows, int64_t cols, int w_bits, int scale_per_row,
               int dtype, void* stream);

/*
 * The same branches in ONE launch (read w + write out instead of ATen's abs + mean + fq_w12_fwd: ~10 B/elem in three launches), with the
 * per-row scale `abs().mean(dim=1)` (:205-209 / :219-224; weight_layerwise is not served) reduced INSIDE the kernel IN ATen'S OWN
 * SUMMATION ORDER -- the fixed tree torch's GPU reduce kernel uses for a contiguous [rows, cols] tensor (groups of 4 elements,
 * four sequential fp32 accumulators per thread, the ROCm shuffle order, the 8-wave combine, the multiply by float(1/cols)) -- so the
 * scale, and with it every output element, is bit-identical to the reference's eager ops on this device.  (ABI <= 3 had an opt-in
 * kernel with its own summation order here, one bf16 ulp off ATen's on <= 0.5 % of the rows; the `sem` parameter left with it.)
 *   scale_out  optional [rows] values of `dtype`: the scaling factor used (mean|w|, or 2*mean|w| for 2 bits)
 * FQ_ERR_UNSUPPORTED where that order is not the restated one or the row does not fit the kernel's registers: rows < 8, cols < 256,
 * cols % 4 != 0, tensors not 16-byte aligned, cols > 32768, fp32 with 4096 < cols <= 8128 -- callers then use ATen + fq_w12_fwd.
 */
int fq_w12_fwd_rows(const void* w, void* out, void* scale_out, int64_t rows, int64_t cols, int w_bits, int dtype, void* stream);

/*
 * ---- the integer side of the fake-quant forward ---------------------------------------------------------
 *
 * The reference never materialises integers: `output = torch.round(input * s).div(s + 1e-6)` (models/utils_quant.py:72)
 * keeps the bins as floats, and AsymQuantizer likewise (`torch.round(input_normalized * s)`, :146).  These entry points
 * emit those bins as PACKED INTEGERS plus the per-row scale terms -- what an int8 / int4 inference export of a QAT
 * checkpoint needs (the published configs are W4 / W8, README.md:45-54), and the pre-pass of a GEMM that applies the
 * fake-quant while it loads its operands (models/utils_quant.py:250 fed by :195-201 and :244-248).
 *
 * bins container (two's complement for SymQuantizer, unsigned for AsymQuantizer whose bins are 0 .. 2^bits-1):
 *   FQ_BINS_INT4   two bins per byte, element 2k in the low nibble of byte k; rows start on a byte ((cols+1)/2 bytes/row)
 *   FQ_BINS_INT8   one byte per bin
 *   FQ_BINS_INT16  two bytes per bin (little endian)
 * The reference has NO clamp (:46-48 are commented out): in bf16 the top bin of an 8-bit row can be +128 (127.5 is a
 * bf16 value and rounds half-to-even), and 16-bit rows reach bins beyond 32767.  A container therefore SATURATES, and
 * overflow_out[row] (optional) counts the elements of that row that did not fit (NaN bins are stored as 0 and counted).
 * overflow_out[row] == 0  <=>  dequantising the export reproduces the fake-quant forward of that row bit for bit (up to the
 * sign of zero: the reference's round(-0.3) is -0.0 and dequantises to -0.0, an integer bin 0 to +0.0):
 *   Sym :  y = round_to_dtype(bin / t2)                                     scales_out[row] = {s, t2}   (t2 = s + 1e-6)
 *   Asym:  y = round_to_dtype(round_to_dtype(round_to_dtype(bin / S) * a) + beta),  S = 2^bits - 1
 *                                                                         scales_out[row] = {a, beta}  (a = alpha + 1e-8)
 * A caller that must be lossless for 8-bit bf16 tensors asks for FQ_BINS_INT16 (or checks the count: it is 0 unless the
 * row's maximum lands on the +128 bin).
 *   x          [rows, cols] `dtype`, contiguous;  bins_out  fq_export_bins_bytes(rows, cols, container) bytes
 *   scales_out float[rows][2] (optional);  overflow_out int32[rows] (optional)
 *   autocast   (Sym, bf16 / fp16 only) 1: the bins of the reference's arithmetic under torch.autocast("cuda") -- fp32 behind
 *              the reciprocal, see fq_sym_fwd_autocast -- which is how LLM-QAT trains; 0: the tensor-dtype arithmetic
 * Traffic (bf16): read 2 B/elem + write 1 (int8) or 0.5 (int4) B/elem, non-temporal.  Rows that fit the register kernels
 * (16-byte aligned, cols a multiple of a 16-byte vector, <= 8192 vectors) take one pass; anything else a two-sweep
 * element kernel (one workgroup per row: a correctness path).
 */
#define FQ_BINS_NONE 0
#define FQ_BINS_INT4 1
#define FQ_BINS_INT8 2
#define FQ_BINS_INT16 3
size_t fq_export_bins_bytes(int64_t rows, int64_t cols, int container);
int fq_sym_export(const void* x, void* bins_out, float* scales_out, int32_t* overflow_out, int64_t rows, int64_t cols, int bits,
                  int container, int dtype, int sem, int autocast, void* stream);
int fq_asym_export(const void* x, void* bins_out, float* scales_out, int32_t* overflow_out, int64_t rows, int64_t cols, int bits,
                   int container, int dtype, int sem, void* stream);

/*
 * The reduction half of SymQuantizer.forward alone (models/utils_quant.py:53-59,:71): per-row  s = reciprocal(max|x| +
 * 1e-6) * qmax  and  t2 = s + 1e-6  -> scales_out[rows][2] = {s, t2}; x is read once (2 B/elem bf16), nothing
 * elementwise is written.  Optionally also records what a training-mode forward records for the STE backward
 * (row_bounds_out + mask_out, exactly as fq_sym_fwd_train; lo / hi = the clip).  With {s, t2} known per row, `round(x * s) / t2`
 * is a pure elementwise map: what a consumer that applies the fake-quant itself needs (an export's dequantisation check; the
 * quantize-on-load GEMM experiment under tools/qlinear/).
 */
int fq_sym_row_scales(const void* x, float* scales_out, int64_t rows, int64_t cols, int bits, int dtype, int sem, int autocast, float lo,
                      float hi, float* row_bounds_out, void* mask_out, size_t mask_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LLMQAT_FAKEQUANT_H */
