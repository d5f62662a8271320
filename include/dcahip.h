/*
 * dcahip.h -- C ABI of libdcahip.so: the MI355X (gfx950) kernels behind the DCA
 * ZINB-autoencoder training path.
 *
 * The reference (theislab/dca v0.3.3) has NO native / FFI interface on this path: every
 * FLOP is a stock TensorFlow CPU kernel reached through Keras.  Each entry
 * point below therefore cites the reference Python call site(s) whose TensorFlow ops it
 * replaces; a maintainer binds them with ctypes exactly as dca_amd/hip.py does (see
 * INTEGRATION.md for the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to row-major fp32 unless stated; the caller
 *     (host code / PyTorch as allocator) owns every buffer, the library allocates nothing
 *     and keeps no global state (no environment variables, no setters: every choice of kernel is a pure function of the
 *     arguments of the call; kernels that were measured and lost -- the pipelined one-wave-per-SIMD K-HEADS, the
 *     non-zero-only first-layer forward, the small-batch byte-store weight gradient, the four-wave matrix-pipe forward --
 *     exist only as -D builds of the sources (DCA_EXP_*), never in the product library);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), re-entrant
 *     across streams and capturable into a hipGraph (no host synchronisation inside);
 *   - return value: 0 on success, otherwise the hipError_t of the failed launch or
 *     DCAHIP_EINVAL for an argument the kernels cannot take (never throws, never aborts);
 *   - "gather": kernels that consume a minibatch read row r of the batch from storage row
 *     perm[*cursor + r] of the resident matrix, so the shuffled batch is never materialised
 *     (reference: Keras slices index_array[batch_start:batch_end] on the host and feeds
 *     copies, dca/train.py:91-98).  perm == NULL means identity (row r); cursor == NULL
 *     means offset 0.  `cursor` lives in device memory so a captured step graph can be
 *     replayed for every batch of an epoch.
 */
#ifndef DCAHIP_H
#define DCAHIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define DCAHIP_VERSION 1
#define DCAHIP_EINVAL (-22)

/* flags for dcahip_zinb_nll */
#define DCAHIP_NLL_HAS_PI      1   /* ZINB (pi head present); otherwise plain NB          */
#define DCAHIP_NLL_CONST_DISP  2   /* theta = clip(exp(theta_w[g]),1e-3,1e4) per gene      */
#define DCAHIP_NLL_POISSON     4   /* mean head only: poisson_loss, dca/loss.py:36-55     */
#define DCAHIP_NLL_MSE         8   /* linear mean head only: mse_loss, dca/loss.py:24-27  */

int dcahip_version(void);

/* Upper bound of `double` slots dcahip_zinb_nll writes to loss_partials. */
int dcahip_zinb_max_partials(void);

/*
 * NB / ZINB negative log-likelihood of one minibatch and its gradient with respect to the
 * head PRE-activations, one pass over the B x G tile (28 B/element: reads 3 pre-activation
 * planes + y, writes 3 gradient planes).
 * Replaces: MeanAct/DispAct/sigmoid (dca/network.py:38-39,369-380), ColwiseMultLayer
 * (dca/layers.py:85), NB.loss (dca/loss.py:72-114), ZINB.loss (dca/loss.py:122-156),
 * ConstantDispersionLayer.theta_exp (dca/layers.py:21) and TensorFlow's autodiff of them.
 * With DCAHIP_NLL_POISSON / DCAHIP_NLL_MSE (ae_type 'poisson' / 'normal', dca/network.py:143-156,
 * 233-246) only a_mean / d_mean are used: poisson_loss on clip(exp(a)) * sf, mse_loss on a * sf.
 *
 *   a_mean, a_disp, a_pi : [B, lda] pre-activations (a_disp NULL with CONST_DISP, a_pi NULL
 *                          without HAS_PI)
 *   theta_w              : [G] log-dispersion (CONST_DISP only)
 *   y, ldy               : resident raw-count matrix (gathered by perm/cursor), sf: [n] size
 *                          factors indexed by the same storage row
 *   inv_n                : 1 / (B_global * G): the reduce_mean divisor folded into the grads
 *   d_mean, d_disp, d_pi : [B, ldd] gradients out (all NULL => loss only, e.g. validation).
 *                          With CONST_DISP d_disp receives d nll/d theta * inv_n per element
 *                          (column-reduce it with dcahip_colsum_chain).
 *                          Columns g in [G, ldd-plane width) are not touched; columns in the
 *                          last partial quad (g >= G, < roundup4(G)) are written as 0.
 *   loss_partials        : [>= dcahip_zinb_max_partials()] doubles; *n_partials_out (host
 *                          int, may be NULL) = slots written = grid size (deterministic).
 */
int dcahip_zinb_nll(const float* a_mean, const float* a_disp, const float* a_pi, long lda,
                    const float* theta_w,
                    const float* y, long ldy, const float* sf,
                    const int* perm, const long long* cursor,
                    int B, int G, float ridge, float inv_n, int flags,
                    float* d_mean, float* d_disp, float* d_pi, long ldd,
                    double* loss_partials, int* n_partials_out, void* stream);

/*
 * dcahip_zinb_nll with the gradient planes leaving as PRE-SPLIT bf16 pieces (see dcahip_gemm_p3): the operand of the
 * heads' weight- and input-gradient products is written once in the form both read, instead of fp32 planes that each
 * product splits again.  d_planes [3][>= B][ldp] bf16 (plane_stride elements between pieces); the head planes start at
 * columns col_mean / col_disp / col_pi of a row (multiples of 4; columns nobody writes must be zero: the products read
 * whole rows).  Constant dispersion: d nll / d theta still goes to the fp32 plane d_theta (ld ldd_theta) that
 * dcahip_colsum_chain reduces, col_disp is ignored.  NB / ZINB only, gradient always, 16-byte aligned vector operands
 * (lda, ldy multiples of 4).  Same loss partials as dcahip_zinb_nll.
 */
int dcahip_zinb_nll_planes(const float* a_mean, const float* a_disp, const float* a_pi, long lda,
                           const float* theta_w, const float* y, long ldy, const float* sf,
                           const int* perm, const long long* cursor, int B, int G, float ridge,
                           float inv_n, int flags, void* d_planes, long ldp, long plane_stride,
                           long col_mean, long col_disp, long col_pi, float* d_theta, long ldd_theta,
                           double* loss_partials, int* n_partials_out, void* stream);

/* The same with the head planes as TWO fp16 pieces of the UNSCALED gradient g 2^d_exp (g = d nll / d pre-activation without the
 * 1 / n factor; dcahip_gemm_h2 takes the factor and the exponent out): d_planes [2][>= B][ldp] fp16.  The caller picks d_exp
 * from the bound |g| <= max(1e4, 2 y_max + 50) (+ ridge / 2) of the likelihood's formulas, so that nothing leaves the fp16
 * range: d_exp = 2 for counts up to 8 000.  A per-gene dispersion's fp32 plane d_theta keeps inv_n. */
int dcahip_zinb_nll_planes_h2(const float* a_mean, const float* a_disp, const float* a_pi, long lda,
                              const float* theta_w, const float* y, long ldy, const float* sf,
                              const int* perm, const long long* cursor, int B, int G, float ridge,
                              float inv_n, int flags, int d_exp, void* d_planes, long ldp, long plane_stride,
                              long col_mean, long col_disp, long col_pi, float* d_theta, long ldd_theta,
                              double* loss_partials, int* n_partials_out, void* stream);

/*
 * Deterministic second stage of the loss reduction: loss = scale * sum(partials) with
 * nan -> inf (dca/loss.py:148), written to *loss_out (fp32, device).
 */
int dcahip_loss_finalize(const double* partials, int n_partials, double scale,
                         float* loss_out, void* stream);

/*
 * End-of-step bookkeeping in one tiny launch: hist[*cursor / rows_per_slot] = *loss (if
 * hist), *acc += *loss * weight (Keras' sample-weighted epoch mean, if acc), *cursor += advance.
 */
int dcahip_step_end(const float* loss, double weight, float* hist, int rows_per_slot,
                    double* acc, long long* cursor, int advance, void* stream);

/*
 * Inference heads: mean*sf, theta, pi from pre-activations (in-place allowed: out == in).
 * Replaces model.predict / extra_models['dispersion'|'pi'].predict (dca/network.py:188-211,
 * 395-405).  sf is indexed by batch row (no gather).  Outputs may be NULL.  flags &
 * DCAHIP_NLL_MSE: the mean head is linear (ae_type 'normal': mean_sf = a_mean * sf).
 */
int dcahip_zinb_heads_infer(const float* a_mean, const float* a_disp, const float* a_pi,
                            long lda, const float* sf, int B, int G,
                            float* mean_sf, float* theta, float* pi, long ldo, int flags,
                            void* stream);

/*
 * K-HEADS: the output heads of one training step in a single pass -- forward GEMM of the heads
 * (pre-activations = H Wh + bh), NB / ZINB NLL + gradient, weight + bias gradient, input
 * gradient -- without materialising the [B, G] pre-activation / gradient planes in HBM.
 * Equivalent to dcahip_sgemm(NN, bias) + dcahip_zinb_nll + dcahip_sgemm(TN, colsum_row) +
 * dcahip_colsum_chain + dcahip_sgemm(NT) on the same operands.
 * Replaces: dca/network.py:369-385 (the three Dense heads, ColwiseMultLayer, SliceLayer, loss
 * closure), dca/loss.py:72-156 and TensorFlow's autodiff of them.
 *
 *   H [B, ldh]      : decoder output (last hidden activation), 16-byte aligned, ldh % 4 == 0
 *   Wh [hL, ldw]    : head weights, head planes in the order [mean | dispersion | pi] (absent
 *                     heads skipped), `plane` columns apart (plane % 4 == 0, G <= plane <=
 *                     roundup32(G)); bh [nheads*plane] the biases in the same layout
 *   theta_w [G]     : log-dispersion (CONST_DISP only)
 *   y, ldy, sf, perm, cursor, ridge, inv_n, flags : as dcahip_zinb_nll
 *   gW [hL+1, ldg]  : OUT weight gradient in the layout of Wh; row hL = bias gradient
 *   g_theta [G]     : OUT d loss / d theta_w (CONST_DISP only)
 *   dH [B, lddh]    : OUT gradient w.r.t. H (columns < hL)
 *   loss_partials   : as dcahip_zinb_nll (finish with dcahip_loss_finalize)
 *   workspace       : >= dcahip_heads_fused_workspace_bytes(...) bytes, 16-byte aligned
 * Supported: 1 <= hL <= 64 (workspace_bytes query returns 0 otherwise -> use the separate
 * kernels).  Deterministic (fixed summation order, no atomics).
 */
long dcahip_heads_fused_workspace_bytes(int B, int hL, int G, long plane, int flags);
int dcahip_heads_fused(const float* H, long ldh, const float* Wh, long ldw, const float* bh,
                       long plane, const float* theta_w,
                       const float* y, long ldy, const float* sf,
                       const int* perm, const long long* cursor,
                       int B, int hL, int G, float ridge, float inv_n, int flags,
                       float* gW, long ldg, float* g_theta, float* dH, long lddh,
                       double* loss_partials, int* n_partials_out,
                       void* workspace, long workspace_bytes, void* stream);
/* The arithmetic of K-HEADS' three matrix products, exposed for testing: C [32, 32] = A [32, K] B [K, 32]
 * (row-major fp32, K % 16 == 0) computed as the kernel does -- each operand scaled by the power of two that brings its
 * largest magnitude into [2^13, 2^14) and split into TWO fp16 pieces (round-to-nearest residual: 2^-22 relative, 2^-25
 * absolute where the second piece is an fp16 denormal, which the matrix pipe preserves), the THREE products a1b1, a1b2, a2b1
 * accumulated in fp32 by v_mfma_f32_32x32x16_f16, the scales taken out at the end (exact).  Contract
 * (tests/test_heads_fused_gpu.py): |C - A B| <= 5e-7 sum|a b| per element, i.e. the accuracy of an fp32 dot product
 * (the fp32 MFMA measures 1.8e-7 .. 2.1e-7 on the same inputs; three bf16 pieces with six products, rounds 2-5: <= 2.5e-7). */
int dcahip_x3_product_32x32(const float* A, const float* B, float* C, int K, void* stream);
/* Same, with the order in which workgroups take the 32-gene tiles: tile_order = device array of
 * dcahip_heads_tile_order_len(G) ints, a permutation of 0 .. ceil(G/32)-1 (padded with values >= ceil(G/32));
 * every 2 consecutive entries share a workgroup.  Results do not depend on it beyond fp32 re-association (the weight gradients
 * are per gene tile: bitwise the same for every order, except that a launch whose plan ends in a poorly filled round of
 * workgroups hands the LAST tiles of the order to a second launch with more batch splits -- e.g. 25 000 genes x 4 096 rows --
 * and a tile's gradient is then the sum of 8 or 16 partial sums instead of 2); for a FIXED order every result is bit for
 * bit reproducible.  A workgroup
 * lasts as long as its slower tile, so pairing tiles of similar non-zero load (sort by the non-zero count of the
 * tile's 32 count columns) removes the imbalance: measured 15 % between the two tiles of a workgroup in file order,
 * 8 % of the kernel.  NULL = identity (what dcahip_heads_fused passes). */
int dcahip_heads_tile_order_len(int G);

int dcahip_heads_fused_ordered(const float* H, long ldh, const float* Wh, long ldw, const float* bh,
                               long plane, const float* theta_w,
                               const float* y, long ldy, const float* sf,
                               const int* perm, const long long* cursor,
                               int B, int hL, int G, float ridge, float inv_n, int flags,
                               float* gW, long ldg, float* g_theta, float* dH, long lddh,
                               double* loss_partials, int* n_partials_out,
                               void* workspace, long workspace_bytes, const int* tile_order, void* stream);
/* Same; when loss_out != NULL the batch loss (inv_n * sum of the partials, nan -> inf: what dcahip_loss_finalize
 * computes) is written there by the last launch of the call -- one launch less per step. */
int dcahip_heads_fused_loss(const float* H, long ldh, const float* Wh, long ldw, const float* bh,
                            long plane, const float* theta_w,
                            const float* y, long ldy, const float* sf,
                            const int* perm, const long long* cursor,
                            int B, int hL, int G, float ridge, float inv_n, int flags,
                            float* gW, long ldg, float* g_theta, float* dH, long lddh,
                            double* loss_partials, int* n_partials_out,
                            void* workspace, long workspace_bytes, const int* tile_order,
                            float* loss_out, void* stream);

/*
 * C[M,N] = op(A) * op(B) (+ bias), fp32 in / fp32 out, LDS-tiled, deterministic split-K through `workspace`.
 * Arithmetic: every operand element is split into three bf16 pieces and the six products a1b1, a1b2, a2b1, a1b3,
 * a2b2, a3b1 are accumulated in fp32 on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16) -- the accuracy of an fp32
 * dot product (dcahip_x3_product_32x32 states the contract) at 6/16 of the cycles of the fp32 MFMA, which runs at the
 * vector rate on gfx950.  split_k < 0 selects the exact-fp32 MFMA kernel (v_mfma_f32_32x32x2_f32, a k-ordered fmaf
 * chain) with the library's split heuristic: the yardstick of the accuracy tests.
 * Replaces the MatMul/BiasAdd kernels of every keras Dense layer forward and backward
 * (dca/network.py:124-126,369-380 and autodiff).
 *   ta == 0: A stored [M,K] (lda >= K)      ta == 1: A stored [K,M]   (C = A^T B, weight grads)
 *   tb == 0: B stored [K,N]                 tb == 1: B stored [N,K]   (C = A B^T, input grads)
 *   bias     : [N] added to every row of C, or NULL
 *   perm/cursor : gather on A's STORAGE rows (batch rows: m index if ta==0, k index if ta==1)
 *   colsum_row  : if non-zero (ta==1 only) additionally writes colsum_k B[k,:] into row M of
 *                 C (C + M*ldc): the Dense bias gradient, free while B streams through LDS
 *   split_k  : 0 = library heuristic (a pure function of the shape), > 0 forced, < 0 exact-fp32 kernel + heuristic
 *   workspace: >= dcahip_sgemm_workspace_bytes(...) bytes, may be NULL when that is 0
 */
int dcahip_sgemm(int ta, int tb, int M, int N, int K,
                 const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                 const float* bias, const int* perm, const long long* cursor,
                 int colsum_row, int split_k, void* workspace, long workspace_bytes,
                 void* stream);
long dcahip_sgemm_workspace_bytes(int ta, int tb, int M, int N, int K, int colsum_row, int split_k);
/*
 * The same products from PRE-SPLIT operands.  "Planes" = the three bf16 pieces of a matrix as three images
 * [3][rows][ld] of bf16 (ld % 8 == 0; plane_stride elements from one piece to the next): x = p0 + p1 + p2 to 2^-24 |x|,
 * the split dcahip_sgemm performs on the fly, done ONCE for an operand that enters several products (the gradient
 * planes of the heads, the head weights, the normalised counts) or written directly by the kernel that produces it.
 * dcahip_gemm_p3 has no vector arithmetic in its K loop, and either operand may be contiguous along k or along m / n
 * (the m / n-contiguous one goes through the transposing LDS read), so ONE stored layout serves the products that
 * contract over a matrix's rows and over its columns: no transposed copies.  Same arithmetic and accuracy contract as
 * dcahip_sgemm (dcahip_x3_product_32x32), same argument meaning: ta / tb as there, lda / ldb in elements,
 * perm / cursor gather A's storage rows, colsum_row (tb == 0) writes the column sums of B into row M of C, split_k 0 =
 * library heuristic.  An operand contiguous along k is read in units of 8 k: K % 8 == 0 (pad BOTH operands with zeros);
 * rows of an m / n-contiguous operand must be allocated up to the next multiple of 8 columns.
 * dcahip_split_planes: fp32 [R, C] (rows gathered through perm / cursor when given) -> planes, columns C .. ldp - 1 zero.
 * Replaces the same MatMul kernels as dcahip_sgemm (dca/network.py:124-126, 369-380 and autodiff) where an operand is
 * reused. */
int dcahip_split_planes(const float* src, long ld, const int* perm, const long long* cursor, long R, int C,
                        void* planes, long ldp, long plane_stride, void* stream);
int dcahip_gemm_p3(int ta, int tb, int M, int N, int K, const void* A, long lda, long plane_a,
                   const void* B, long ldb, long plane_b, float* C, long ldc, const float* bias,
                   const int* perm, const long long* cursor, int colsum_row, int split_k,
                   void* workspace, long workspace_bytes, void* stream);
long dcahip_gemm_p3_workspace_bytes(int M, int N, int K, int colsum_row, int split_k);
/*
 * The plane products on TWO fp16 pieces per operand and THREE products per fp32 product (round 6: the arithmetic of K-HEADS,
 * dcahip_x3_product_32x32; half the matrix instructions of dcahip_gemm_p3).  An operand matrix is scaled by the power of two
 * that brings its largest magnitude into [2^13, 2^14) before it is split: the exponent lives in a DEVICE word, so nothing
 * leaves the stream (capturable).
 *   dcahip_absmax_exp       *exp_out = that exponent for src [R, C] (scratch_word: 4 bytes of device memory the call may use)
 *   dcahip_split_planes_h2  fp32 [R, C] (rows gathered through perm / cursor) x 2^*exp -> planes [2][R][ldp] fp16 (exp NULL: 0)
 *   dcahip_gemm_h2          C = alpha 2^-(ea + eb) op(A) op(B) (+ bias), ea = (exp_a ? *exp_a : 0) + exp_a_add, eb likewise;
 *                           colsum_row: row M of C = alpha 2^-eb x the column sums of B.  Shapes: those of the 256 x 256 kernel
 *                           (dcahip_gemm_h2_supported; M, N >= 256, M N >= 2^18, K % 16 == 0, no row gather); layouts, ta / tb,
 *                           split_k and the workspace as dcahip_gemm_p3.
 * Contract (tests/test_gemm_h2_gpu.py): |C - alpha op(A) op(B)| <= 1e-6 alpha sum|a b| per element (a priori 3 x 2^-22 = 7.2e-7
 * plus the fp32 accumulation over K; the worst of 3 M elements at K = 512 measured 5.7e-7), deterministic.
 * A producer that writes planes directly (dcahip_zinb_nll_planes_h2: the gradient planes as g 2^d_exp) passes its static
 * exponent through exp_*_add.  Replaces the same MatMul kernels as dcahip_sgemm (dca/network.py:124-126, 369-380 and autodiff).
 */
int dcahip_absmax_exp(const float* src, long ld, long R, int C, int* exp_out, void* scratch_word, void* stream);
int dcahip_split_planes_h2(const float* src, long ld, const int* perm, const long long* cursor, long R, int C,
                           void* planes, long ldp, long plane_stride, const int* exp, void* stream);
int dcahip_gemm_h2_supported(int M, int N, int K);
long dcahip_gemm_h2_workspace_bytes(int M, int N, int K, int colsum_row, int split_k);
int dcahip_gemm_h2(int ta, int tb, int M, int N, int K, const void* A, long lda, long plane_a, const int* exp_a, int exp_a_add,
                   const void* B, long ldb, long plane_b, const int* exp_b, int exp_b_add, float alpha,
                   float* C, long ldc, const float* bias, int colsum_row, int split_k,
                   void* workspace, long workspace_bytes, void* stream);
/* dst [C, ld_dst] = src [R, ld_src]^T.  The first Dense layer's kernel W0 [genes, h1] is transposed once per step at
 * throughput batches so that its forward product X W0 runs in the NT form (both operands contiguous along the
 * contraction: the fast operand path of dcahip_sgemm). */
int dcahip_transpose(const float* src, long ld_src, int R, int C, float* dst, long ld_dst, void* stream);
/* The same with gathered source rows: dst[c][r] = src[perm[*cursor + r]][c] (perm NULL: r).  Wide networks at throughput
 * batches (first layer >= 128 units): the minibatch of the resident matrix is transposed once per step and the first
 * layer's weight gradient X^T dZ runs as a plain product of a k-contiguous A (see dcahip_sgemm). */
int dcahip_transpose_rows(const float* src, long ld_src, const int* perm, const long long* cursor, int R, int C,
                          float* dst, long ld_dst, void* stream);

/*
 * Batch normalisation (center=True, scale=False, eps inside rsqrt) + ReLU, training mode.
 * Replaces keras BatchNormalization + Activation('relu') (dca/network.py:127-135).
 * Three stages so that data-parallel runs can exchange the statistics in between:
 *  1. dcahip_col_moments: per row-chunk (mean, M2) of Z[B,H]  -> part[R][2][H], R returned
 *  2. (DP only) dcahip_moments_combine: merge E entries (counts[e] rows each) into one
 *  3. dcahip_bn_relu_apply: merge the E entries it is given (Chan et al.), then
 *        xhat = (z - mean) * rsqrt(var + eps); h = max(xhat + beta, 0)
 *     writes h, xhat (may be NULL), inv_std[H], and (training) updates
 *        moving = moving - (moving - batch) * (1 - momentum)   (biased variance).
 *     With entries == NULL it runs in INFERENCE mode on moving_mean / moving_var.
 *     `relu` is the activation code: 0 linear, 1 relu, 2 tanh, 3 sigmoid, 4 elu, 5 selu, 6 softplus,
 *     7 softsign, 8 LeakyReLU(0.3) (Activation(self.activation) / advanced activations,
 *     dca/network.py:132-135); the backward entry points take the same code as `act` and derive
 *     the slope from the forward OUTPUT h.  B == 0 is legal (only the moving statistics are
 *     updated): a data-parallel rank with an exhausted shard still joins the exchange.
 */
int dcahip_col_moments_chunks(int B);
int dcahip_col_moments(const float* Z, long ldz, int B, int H, float* part, void* stream);
int dcahip_moments_combine(const float* entries, const float* counts, int E, int H,
                           float* out /*[2][H]*/, void* stream);
int dcahip_bn_relu_apply(const float* Z, long ldz, int B, int H,
                         const float* entries, const float* counts, int E,
                         const float* beta, float* moving_mean, float* moving_var,
                         float momentum, float eps, int relu,
                         float* Hout, long ldh, float* xhat, long ldx, float* inv_std,
                         void* stream);

/* Batch normalisation for small batches on one GPU (B <= dcahip_bn_fused_max_rows()): batch statistics, moving
 * averages, normalisation + activation (dcahip_col_moments + dcahip_bn_relu_apply) in ONE launch, and the backward
 * pass (dcahip_bn_bwd_sums + dcahip_bn_bwd_apply) in one: the reference-default batch of 32 cells (dca/train.py:37)
 * is bound by launch gaps.  Same arguments and formulas as the two-call forms (one chunk of up to 64 rows). */
int dcahip_bn_fused_max_rows(void);
/* A whole hidden layer per launch for the same small batches (B <= dcahip_bn_fused_max_rows(), K <= dcahip_dense_small_max_k()
 * inputs): Dense -> BatchNormalization (batchnorm != 0) -> activation forward (dca/network.py:124-135; Z receives the
 * pre-activation when non-NULL: the latent code of the centre layer), and its backward: d beta, the weight gradient gW
 * [K, h] with the bias gradient in row K, the gradient w.r.t. the layer input dHp (NULL for none); the backward needs
 * h <= dcahip_dense_small_max_k() as well (one workgroup owns the layer).  Equivalent to dcahip_sgemm + the batch-norm
 * kernels (+ 2 x dcahip_sgemm backward) on the same operands. */
int dcahip_dense_small_max_k(void);
int dcahip_dense_bn_small(const float* Hp, long ldp, const float* W, long ldw, const float* bias,
                          int B, int K, int H, int batchnorm, const float* beta,
                          float* moving_mean, float* moving_var, float momentum, float eps, int act,
                          float* Z, long ldz, float* xhat, long ldx, float* Hout, long ldh,
                          float* inv_std, void* stream);
int dcahip_dense_bn_bwd_small(const float* dH, long ldd, const float* Hact, long ldh,
                              const float* xhat, long ldx, const float* inv_std,
                              const float* Hp, long ldp, const float* W, long ldw,
                              int B, int K, int H, int batchnorm, float n_total, int act,
                              float* gW, long ldg, float* dbeta, float* dHp, long lddp, void* stream);
/* The hidden stack behind the first layer's product in ONE launch (every layer <= 64 units, B <= dcahip_bn_fused_max_rows()):
 * entry 0 with W == NULL normalises + activates its own Z (written by dcahip_sgemm), every entry with a kernel is
 * Dense -> BatchNormalization -> activation on the previous entry's Hout (entry 0 with a kernel reads Hin).  Same
 * formulas and outputs as dcahip_bn_relu_train_small / dcahip_dense_bn_small called one after the other
 * (dca/network.py:124-135). */
typedef struct dcahip_small_layer {
    const float* W; long ldw;           /* kernel [K, H] or NULL */
    const float* bias;                  /* [H] */
    int K, H;
    const float* beta; float* moving_mean; float* moving_var;
    float* Z; long ldz;                 /* pre-activation: input when W == NULL, optional output otherwise */
    float* xhat; long ldx; float* Hout; long ldh; float* inv_std;
} dcahip_small_layer;
int dcahip_hidden_small_chain(const dcahip_small_layer* layers, int n, const float* Hin, long ldin, int B,
                              int batchnorm, float momentum, float eps, int act, void* stream);
/* K-STACK: the same hidden stack at THROUGHPUT batches (B up to dcahip_hidden_stack_max_rows(), every layer <= 64
 * units, batch normalisation on, one GPU) in one launch per direction: workgroups own blocks of batch rows and exchange
 * only the batch-norm statistics (and, at the end of the backward pass, the weight-gradient partials) through grid-wide
 * barriers -- at most 256 workgroups, all resident: launch it on an otherwise idle device (the stream order of a training
 * step guarantees that).  Forward: entry 0 (W == NULL) normalises + activates its own Z (written by dcahip_sgemm), every
 * further entry is Dense -> BatchNormalization -> activation (dca/network.py:124-135); outputs as dcahip_bn_relu_apply /
 * dcahip_sgemm (Z optional when the whole pass is one launch).  Backward: per layer dbeta, for every layer
 * behind the first gW [K + 1, H] (row K = bias gradient), and dZ0 = gradient w.r.t. the first layer's pre-activation
 * (dcahip_bn_bwd_* + dcahip_sgemm x 2 per layer on the same operands).  workspace >=
 * dcahip_hidden_stack_workspace_bytes(n, B) bytes, 16-byte aligned, ZERO before the first call (arrival counters; every
 * call leaves them at zero), shared by both directions.  Deterministic. */
typedef struct dcahip_stack_bwd_layer {
    const float* W; long ldw; int K, H;             /* kernel [K, H] (unused for the first layer) */
    const float* Hact; long ldh;                    /* the layer's output (activation derivative through the output) */
    const float* xhat; long ldx; const float* inv_std;
    const float* Hprev; long ldp;                   /* the layer's input activations [B, K] (unused for the first layer) */
    float* gW; long ldg; float* dbeta;              /* OUT (gW unused for the first layer) */
    float* dH; long lddh;                           /* gradient w.r.t. the layer's output: IN for the last layer, scratch
                                                       (written, then read by the next launch) for the others */
} dcahip_stack_bwd_layer;
/* Both passes are a sequence of STEPS separated by a batch-wide dependency (the batch-norm statistics):
 *   forward  (n + 1 steps): 0 = partial statistics of the first layer's pre-activation; 1 + i = layer i (merge the
 *             statistics, normalise + activate, next layer's pre-activation and ITS partial statistics);
 *   backward (n + 2 steps): 0 = dy and the batch sums of the last layer; 1 + j = layer n - 1 - j (dZ, d beta, weight-gradient
 *             partial, input gradient, dy and sums of the layer below); n + 1 = sum of the weight-gradient partials.
 * A call runs steps [first_step, last_step] in ONE launch: a range of several steps synchronises with grid barriers
 * (every workgroup resident: at most 256 of them -- rows_per_wg * 256 >= B), a single step needs none (the kernel
 * boundary is the barrier; up to 1024 workgroups).  rows_per_wg (16 .. 64) fixes the row partition and must be the same
 * for every call of a pass.  A whole backward pass (steps 0 .. n + 1) over B <= 64 rows -- the reference's default batch of
 * 32 -- is ONE workgroup's work: the batch sums are its own, the input gradients stay in registers between layers, the
 * operands of the layer below are requested while a layer computes; that call takes no workspace (NULL, 0). */
int dcahip_hidden_stack_max_rows(void);
long dcahip_hidden_stack_workspace_bytes(int n_layers, int B);
int dcahip_hidden_stack_fwd(const dcahip_small_layer* layers, int n, int B, float momentum, float eps, int act,
                            int rows_per_wg, int first_step, int last_step,
                            void* workspace, long workspace_bytes, void* stream);
int dcahip_hidden_stack_bwd(const dcahip_stack_bwd_layer* layers, int n, int B, float n_total, int act,
                            float* dZ0, long ldz0, int rows_per_wg, int first_step, int last_step,
                            void* workspace, long workspace_bytes, void* stream);

/* K-STACK inside a data-parallel step (SyncBN; dca/network.py:127-128 BatchNormalization with the statistics of the GLOBAL
 * batch): ONE step per call (the steps of dcahip_hidden_stack_fwd / _bwd at 32 rows per workgroup) with the batch-wide
 * quantity of the step's input layer handed in from outside and the one the step produces merged over this rank's row
 * blocks for the exchange that follows:
 *   forward, step s = 0..n: s = 0 only measures layer 0; s >= 1 normalises layer s - 1 with ext_entries [ext_E][2][H]
 *     (mean, M2 per rank) and ext_counts [ext_E] (rows per rank), applies the activation, multiplies by the next layer's
 *     kernel; stat_out [2][H'] <- (mean, M2) of this rank's rows of the layer made (all-gather it; NULL after the last).
 *   backward, step s = 0..n: s = 0 only sums the top layer; s >= 1 handles layer n - s with ext_sums [2][H] (sum dy,
 *     sum dy xhat over ALL ranks, all-reduced); sums_out [2][K] <- this rank's sums of the layer below (all-reduce them),
 *     whose first half is also stored as that layer's LOCAL d beta.  The weight-gradient reduction is step n + 1 of
 *     dcahip_hidden_stack_bwd (rows_per_wg = 32), unchanged.
 * dcahip_hidden_stack_step_blocks(B): workgroups (= row blocks) of these launches. */
int dcahip_hidden_stack_step_blocks(int B);
int dcahip_hidden_stack_fwd_sync(const dcahip_small_layer* layers, int n, int B, float momentum, float eps, int act,
                                 int step, const float* ext_entries, const float* ext_counts, int ext_E,
                                 float* stat_out, void* workspace, long workspace_bytes, void* stream);
int dcahip_hidden_stack_bwd_sync(const dcahip_stack_bwd_layer* layers, int n, int B, float n_total, int act,
                                 float* dZ0, long ldz0, int step, const float* ext_sums, float* sums_out,
                                 void* workspace, long workspace_bytes, void* stream);
int dcahip_bn_relu_train_small(const float* Z, long ldz, int B, int H, const float* beta,
                               float* moving_mean, float* moving_var, float momentum, float eps,
                               int act, float* Hout, long ldh, float* xhat, long ldx,
                               float* inv_std, void* stream);
int dcahip_bn_bwd_small(const float* dH, long ldd, const float* Hact, long ldh,
                        const float* xhat, long ldx, const float* inv_std, float n_total,
                        int B, int H, float* dZ, long ldz, float* dbeta, int act, void* stream);

/*
 * Backward of ReLU + batch norm.  mask = the forward output h (h > 0 <=> pre-ReLU > 0).
 *  1. dcahip_bn_bwd_sums: per row-chunk sums of dy and dy*xhat, dy = dh*[h>0] -> part[R][2][H]
 *  2. (DP only) all-reduce the combined sums
 *  3. dcahip_bn_bwd_apply: dz = inv_std * (dy - S1/n - xhat * S2/n), dbeta = S1
 *     (sums: E entries [E][2][H] are added in order; n_total = global batch rows).
 * Without batch norm use dcahip_relu_bwd (dz = dh*[h>0]).
 */
int dcahip_bn_bwd_sums(const float* dH, long ldd, const float* Hact, long ldh,
                       const float* xhat, long ldx, int B, int H, float* part, int act, void* stream);
int dcahip_bn_bwd_apply(const float* dH, long ldd, const float* Hact, long ldh,
                        const float* xhat, long ldx, const float* inv_std,
                        const float* sums, int E, float n_total, int B, int H,
                        float* dZ, long ldz, float* dbeta, int act, void* stream);
int dcahip_relu_bwd(const float* dH, long ldd, const float* Hact, long ldh, int B, int H,
                    float* dZ, long ldz, int act, void* stream);
/* h = max(z, 0): Activation('relu') of a stack built with batchnorm=False (network.py:132-135). */
int dcahip_relu_fwd(const float* Z, long ldz, int B, int H, float* Hout, long ldh, int act, void* stream);
/* Shared heads (NBSharedAutoencoder / ZINBSharedAutoencoder, network.py:343-362, 464-491): dispersion and
 * dropout come from Dense(1) layers and broadcast over the genes inside the loss (loss.py:85-88,130-137).
 * dcahip_bcast_cols: out[r, c] = s[r * lds] for c < G (the scalar pre-activation spread into the plane
 * the loss kernel reads).  dcahip_row_sums_strided: out[r * ldo] = sum_c x[r, c] (the plane of
 * pre-activation gradients folded back into the gradient of the scalar); fp64 accumulation, fixed order. */
/* keras.layers.PReLU (activation='PReLU', network.py:132-133): out = max(x, 0) + alpha[c] min(x, 0), one trainable
 * slope per unit.  x = the batch-norm (or bias) output with the LINEAR activation code.  dcahip_prelu_bwd: d holds
 * dL/dout on entry and dL/dx on return; galpha[c] = sum_r dL/dout min(x, 0) (fp64 partials, fixed order);
 * workspace: dcahip_prelu_workspace_doubles(h) doubles. */
int dcahip_prelu_workspace_doubles(int h);
int dcahip_prelu_fwd(const float* x, long ldx, const float* alpha, int B, int h, float* out, long ldo, void* stream);
int dcahip_prelu_bwd(float* d, long ldd, const float* x, long ldx, const float* alpha, int B, int h,
                     float* galpha, double* workspace, void* stream);
/* zinb-elempi (ZINBAutoencoderElemPi network.py:424-461, ElementwiseDense layers.py:50-82): the mean head's Dense
 * output is negated (m = -a feeds MeanAct) and the dropout logit is a_pi[r, g] = k[g] m[r, g] + c[g].
 * dcahip_elempi_fwd: a_mean <- m in place, a_pi plane written.  dcahip_elempi_bwd: d_mean holds dL/dm on entry and
 * dL/d(Dense output) = -(dL/dm + k dL/da_pi) on return; gk[g] = sum_r dL/da_pi m, gc[g] = sum_r dL/da_pi (fp64
 * partials over a fixed row assignment: deterministic); workspace: dcahip_elempi_workspace_doubles(G) doubles. */
int dcahip_elempi_workspace_doubles(int G);
int dcahip_elempi_fwd(float* a_mean, long lda, const float* k, const float* c, int B, int G,
                      float* a_pi, long ldp, void* stream);
int dcahip_elempi_bwd(const float* m, long lda, float* d_mean, const float* d_pi, long ldd, const float* k,
                      int B, int G, float* gk, float* gc, double* workspace, void* stream);
int dcahip_bcast_cols(const float* s, long lds, int B, int G, float* out, long ldo, void* stream);
int dcahip_row_sums_strided(const float* x, long ldx, int B, int G, float* out, long ldo, void* stream);

/* out[c] (+)= chain(c) * sum_r x[r, c]; chain = d clip(exp(w),1e-3,1e4)/dw if theta_w given
 * (ConstantDispersionLayer gradient, dca/layers.py:17-21), else 1. */
int dcahip_colsum_chain(const float* x, long ldx, int B, int N, const float* theta_w,
                        float* out, void* stream);

/*
 * K-PREP: dca/io.py:88-111 (normalize: scanpy filter counts, normalize_per_cell, log1p, scale)
 * on the resident count matrix Y [n, ldy].
 *   dcahip_prep_row_sums   out[r] = sum_g Y[r, g]                 (n_counts; exact for counts)
 *   dcahip_prep_col_pass   x = Y[r, g]; if fac: x /= fac[r]; if do_log: x = log1p(x);
 *                          X[r, g] = x (X may be NULL, may alias Y); per-gene sums of x and of
 *                          fl32(x*x) in fp64 per row chunk -> col_part[R][2][roundup4(G)],
 *                          R = dcahip_prep_chunks(n)
 *   dcahip_prep_col_finish ordered sum of the R chunks (x E ranks' all-reduced partials:
 *                          pass the summed [1][2][Gp] block with R = 1): sums[g] (gene counts
 *                          of filter_genes, may be NULL) and/or mean[g], std[g] exactly as
 *                          sc.pp.scale: var = (E[x^2] - E[x]^2) n/(n-1), std 0 -> 1
 *   dcahip_prep_scale      X = (X - mean) / std in place (fp32 division)
 */
int dcahip_prep_chunks(int n);
int dcahip_prep_row_sums(const float* Y, long ldy, int n, int G, float* out, void* stream);
int dcahip_prep_col_pass(const float* Y, long ldy, int n, int G, const float* fac, int do_log,
                         float* X, long ldx, double* col_part, void* stream);
int dcahip_prep_col_finish(const double* col_part, int R, int G, double n_total,
                           float* sums, float* mean, float* stdv, void* stream);
int dcahip_prep_scale(float* X, long ldx, int n, int G, const float* mean, const float* stdv,
                      void* stream);

/*
 * Keras clipvalue + Keras RMSprop (momentum 0) on one flat parameter buffer:
 *   g = clip(g, -clip, clip); ms = rho*ms + (1-rho)*g*g; w -= lr * g / (sqrt(ms) + eps)
 * (epsilon OUTSIDE the root: standalone keras 2.2 / 2.3 `p - lr * g / (K.sqrt(new_a) + self.epsilon)` and tf.keras
 * OptimizerV2's dense update without momentum `var - lr_t * grad / (sqrt(rms_t) + epsilon)` alike; only TF's fused
 * ApplyRMSProp kernel, taken with momentum > 0, puts it inside.)
 * Replaces opt.RMSprop(lr, clipvalue) (dca/train.py:54-57).  *lr is read from device memory
 * so ReduceLROnPlateau does not invalidate a captured graph.  clip <= 0 disables clipping.
 */
int dcahip_rmsprop_clip(float* w, const float* g, float* ms, long n, const float* lr,
                        float rho, float eps, float clip, void* stream);
/* The same launch followed by what dcahip_step_end does (loss slot -> history / epoch accumulator, batch cursor
 * += advance): one launch less per step, which is what the reference-default batch of 32 is made of. */
int dcahip_rmsprop_clip_end(float* w, const float* g, float* ms, long n, const float* lr,
                            float rho, float eps, float clip, const float* loss, double weight,
                            float* hist, int rows_per_slot, double* acc, long long* cursor, int advance,
                            void* stream);

/*
 * K-OPT: the other Keras optimizers selectable through dca/train.py:54-57 and the l1 / l2 kernel
 * regularisers of dca/network.py:114-126,144-146,369-380, on the flat parameter buffer.
 * dcahip_optimizer_step: g clipped to [-clip, clip] (clip <= 0: off), then the tf.keras update
 * with default hyper-parameters (formulas in dcahip_opt.hip); slot1 / slot2 are the optimizer's
 * state buffers ([n], may be NULL where the optimizer has none: SGD none, Adagrad slot1 (init 0.1),
 * Adadelta / Adam / Adamax both).  *iter = completed steps (device memory; Adam / Adamax bias
 * correction), advanced by dcahip_counter_add.  RMSprop is dcahip_rmsprop_clip.
 * dcahip_l1l2_apply: for each segment [start, end) of the flat buffer: g += l1 sign(w) + 2 l2 w
 * (g may be NULL: penalty only) and *loss_inout += sum l1 |w| + l2 w^2 (Keras adds the
 * regularisation losses to the reported loss, training and validation).
 */
#define DCAHIP_OPT_SGD       0
#define DCAHIP_OPT_RMSPROP   1
#define DCAHIP_OPT_ADAGRAD   2
#define DCAHIP_OPT_ADADELTA  3
#define DCAHIP_OPT_ADAM      4
#define DCAHIP_OPT_ADAMAX    5
int dcahip_optimizer_step(int kind, float* w, const float* g, float* slot1, float* slot2, long n,
                          const float* lr, const long long* iter, float clip, void* stream);
int dcahip_counter_add(long long* counter, int v, void* stream);
/* tf.keras Nadam (beta_1 .9, beta_2 .999, epsilon 1e-7, schedule decay .004): m, v = the two slots ([n], zero
 * initialised), *m_schedule = running product of the momentum schedule (device float, initialised to 1 by the
 * host; the call multiplies it by mu_t after the update has read it), *iter = completed steps as above. */
int dcahip_nadam_step(float* w, const float* g, float* m, float* v, long n, const float* lr,
                      const long long* iter, float* m_schedule, float clip, void* stream);
#define DCAHIP_REG_MAX_SEGS 16
typedef struct {
    int nseg;
    long start[DCAHIP_REG_MAX_SEGS];
    long end[DCAHIP_REG_MAX_SEGS];
    float l1[DCAHIP_REG_MAX_SEGS];
    float l2[DCAHIP_REG_MAX_SEGS];
} dcahip_reg_desc;
int dcahip_l1l2_workspace_doubles(void);
int dcahip_l1l2_apply(const dcahip_reg_desc* d, const float* w, float* g, float* loss_inout,
                      double* workspace, void* stream);

/*
 * K-DROP: Keras Dropout of dca/network.py:98-99 (input_dropout) and :137-138 (hidden_dropout),
 * training mode only: out[r, c] = x[src(r), c] * keep(r, c) / (1 - rate).
 * src(r) = perm[*cursor + r] when perm != NULL (the minibatch gather of the input layer), else r;
 * in place (out == x) is allowed when perm == NULL.  The same call applied to the incoming
 * gradient is the backward pass (the mask is recomputed, never stored).
 * keep(r, c): Philox4x32-10 (Salmon et al., SC'11) with key = seed, counter =
 * (group lo, group hi, (uint32)*step, layer), group = (row0 + r) * ceil(h / 4) + c / 4, word c % 4
 * of the output block; u = (word >> 8) * 2^-24 and the unit is kept when u >= rate (TF:
 * random_uniform >= rate).  row0 = index of this rank's first row inside the global batch, so a
 * data-parallel run draws the masks of the single-process run.  *step is device memory (captured
 * step graphs stay valid); the host advances it with dcahip_counter_add.  The reference's masks
 * come from TF's stateful RNG and are not reproducible across TF builds: parity here is statistical
 * (keep frequency, scaling) and exact against oracle/net_np.py's restatement of this generator,
 * which is pinned on the published Random123 known-answer vectors.
 */
int dcahip_dropout_apply(const float* x, long ldx, const int* perm, const long long* cursor, int B, int h,
                         float rate, unsigned long long seed, const long long* step, int layer, long row0,
                         float* out, long ldo, void* stream);

/*
 * K-SPARSE: compact count storage and the first Dense layer on the non-zero counts only.
 *
 * The reference turns the ~93 %-zero count matrix into a DENSE fp32 input X = scale(log1p(counts / size factor))
 * (dca/io.py:88-111; scanpy normalize_per_cell, log1p, scale) and feeds it to the first Dense layer
 * (dca/network.py:124-126); TensorFlow's autodiff forms the weight gradient X^T dZ from the same dense matrix.
 * With x[c, g] = (L[c, g] - mean[g]) / std[g], L = log1p(y / fac[c]) (0 where y = 0; each step optional) both
 * products need the non-zero counts only (formulas in dca_amd/csrc/dcahip_sparse.hip).
 *
 * Compact counts: Yc [n, ldc] bytes, ldc = dcahip_counts_compact_ld(G) (G rounded up to 16, pad columns 0); a count
 * 0 .. 254 is stored as is, 255 is an escape whose value is looked up in a per-row overflow list: entries
 * ovf_ptr[r] .. ovf_ptr[r + 1] - 1 of (ovf_col, ovf_val), sorted by column (all three may be NULL when no count
 * reaches 255).  dcahip_counts_compact writes Yc from the fp32 counts and ADDS to status[0] the number of values that
 * are not counts (negative, fractional, not finite: stored as 0 -- the caller must not use the compact store then)
 * and to status[1] the number of escapes (the caller builds the overflow list from Y >= 255).
 * K-HEADS takes the same store through dcahip_heads_fused_compact.
 */
long dcahip_counts_compact_ld(int G);
int dcahip_counts_compact(const float* Y, long ldy, int n, int G, unsigned char* Yc, long ldc, int* status,
                          void* stream);
/* K-HEADS (dcahip_heads_fused_loss) reading the counts from the compact store: yc != NULL selects it (y / ldy are then
 * unused and may be NULL / 0), yc == NULL is dcahip_heads_fused_loss.  4 x fewer count bytes per launch.
 * d_exp (<= 0; 0 for count matrices whose largest count stays below ~8 000, what the other entry points pass): the kernel
 * carries the gradient planes as g 2^(1 + d_exp) in two fp16 pieces, g = the UNSCALED d nll / d pre-activation, |g| <=
 * max(1e4, ~2 y_max).  Values beyond the fp16 range are still exact (their tile is rescaled as a whole: a slow path); a caller
 * that knows its counts reach y_max passes d_exp = -ceil(log2(y_max / 8192)) and keeps every tile on the fast path.
 * ridge must lie in [0, 1e3]. */
int dcahip_heads_fused_compact(const float* H, long ldh, const float* Wh, long ldw, const float* bh,
                               long plane, const float* theta_w,
                               const float* y, long ldy,
                               const unsigned char* yc, long ldc, const int* ovf_ptr, const int* ovf_col,
                               const float* ovf_val, const float* sf,
                               const int* perm, const long long* cursor,
                               int B, int hL, int G, float ridge, float inv_n, int flags,
                               float* gW, long ldg, float* g_theta, float* dH, long lddh,
                               double* loss_partials, int* n_partials_out,
                               void* workspace, long workspace_bytes, const int* tile_order,
                               float* loss_out, int d_exp, void* stream);
/* First-layer widths the kernels below take (32, 64, 128). */
int dcahip_enc0_sparse_supported(int H1);
/*
 * Weight (+ bias) gradient of the first Dense layer from the compact counts:
 *   gW [G + 1, ldg]: rows g < G = sum_c x[c, g] dZ[c, :], row G = column sums of dZ (what dcahip_sgemm(ta = 1,
 *   colsum_row = 1) writes from the dense X).  Batch row c = storage row perm[*cursor + row_base + c] (perm NULL:
 *   *cursor + row_base + c) of Yc / fac.  fac NULL: no size-factor division; do_log 0: no log1p; mean NULL: 0;
 *   stdv NULL: 1.  lutp = dcahip_enc0_lut(fac, do_log) of the same cells.  n_cells * ldc must stay below 2^32.
 *   Arithmetic: X^T dZ on the matrix pipe, six bf16 products per fp32 product as dcahip_sgemm; deterministic.
 *   workspace >= dcahip_enc0_dw_sparse_workspace_bytes(B, G, H1), 16-byte aligned.
 * Replaces the autodiff of dca/network.py:124-126 w.r.t. the first kernel on the input of dca/io.py:88-111.
 */
/* lutp [n, dcahip_enc0_lut_entries() = 128] entries of 8 bytes: entry k of cell r = f(k / fac[r]) (f = log1p if do_log,
 * fac NULL: 1) split into the three bf16 pieces the matrix products use ({p0 | p1 << 16, p2}) -- made once per dataset, so
 * that the first-layer kernels LOOK UP their operand instead of dividing, taking logarithms and splitting.  The kernels
 * hold the first 32 (forward), 64 (first weight-gradient kernel) or all 128 (ring weight-gradient kernel) entries of their
 * cells in LDS; counts beyond take the formula (hardware log2 with an exact-ratio correction, ~2e-7 relative). */
int dcahip_enc0_lut(const float* fac, int do_log, int n, void* lutp, void* stream);
int dcahip_enc0_lut_entries(void);
long dcahip_enc0_dw_sparse_workspace_bytes(int B, int G, int H1);
/* The 64-unit weight gradient has two kernels (csrc/dcahip_sparse.hip): enc0_dw_kernel (counts through an LDS tile, lookups
 * and products of a 64-row block between two barriers) and enc0_dw2_kernel (512 genes per workgroup, operands staged
 * global -> LDS into a five-stage ring four K steps ahead, a wave's LDS / vector work behind its own matrix instructions).
 * `form` (an argument of the call: the library keeps no switch): 0 = by the shape -- the ring kernel from 1024 batch rows
 * up, the first kernel below --, 1 = always the first, 2 = always the ring kernel (A/B runs and the parity tests of both).
 * Same products, same row order inside a split; the number of splits differs.  The workspace size covers either. */
int dcahip_enc0_dw_sparse(const unsigned char* Yc, long ldc, const int* ovf_ptr, const int* ovf_col,
                          const float* ovf_val, const float* fac, int do_log, const void* lutp, const float* mean,
                          const float* stdv, const int* perm, const long long* cursor, long row_base,
                          int B, int G, int H1, const float* dZ, long ldz, float* gW, long ldg,
                          void* workspace, long workspace_bytes, int form, void* stream);
/*
 * Forward of the first Dense layer from the compact counts, Z [B, ldz] = X W + bias with X as above, W [G, ldw], on the
 * matrix pipe (H1 = 32 or 64; 0 workspace bytes = width not taken): the A
 * operand is looked up from the byte store through lutp (dcahip_enc0_lut of the same cells: counts 0 .. 31 from the
 * table, larger ones and escapes by the formula), W / std is split into bf16 pieces once per call, six bf16 products per
 * fp32 product as dcahip_sgemm, gene chunks added in a fixed order: deterministic.  workspace >=
 * dcahip_enc0_fwd_lut_workspace_bytes(B, G, H1), 16-byte aligned, any content; n_cells * ldc must stay below 2^32 (lutp).
 * Replaces Dense(hidden_size[0]) of dca/network.py:124-126 on the input of dca/io.py:88-111.
 */
long dcahip_enc0_fwd_lut_workspace_bytes(int B, int G, int H1);
int dcahip_enc0_fwd_lut(const unsigned char* Yc, long ldc, const int* ovf_ptr, const int* ovf_col,
                        const float* ovf_val, const float* fac, int do_log, const void* lutp, const float* mean,
                        const float* stdv, const int* perm, const long long* cursor, long row_base,
                        int B, int G, int H1, const float* W, long ldw, const float* bias,
                        float* Z, long ldz, void* workspace, long workspace_bytes, void* stream);

/*
 * K-PEER (dca_amd/csrc/dcahip_peer.hip): the small exchanges of the data-parallel step without a library call.
 *   slots[q], flags[q]: device arrays of `world` pointers to every rank's exchange buffers (rank's own: its local
 *   allocation; the others: mapped with hipIpcOpenMemHandle), each dcahip_peer_slot_bytes(world, nmax) /
 *   dcahip_peer_flag_bytes(world) bytes, zero-initialised once, in FINE-GRAINED device memory
 *   (hipExtMallocWithFlags(hipDeviceMallocFinegrained)): a peer's stores over xGMI are not guaranteed visible to the owner's
 *   spinning loads in a coarse-grained (plain hipMalloc) allocation.  epoch: one device counter per communicator, 0 at the start,
 *   advanced by every call (all ranks call in the same order).  reduce = 0: out [world * n] = concatenation of the ranks'
 *   vectors (all-gather); reduce = 1: out [n] = their sum in rank order (all-reduce; out may be local).  A peer that does
 *   not arrive within timeout_us microseconds (100 MHz wall clock) sets *status |= 1 and `out` is filled with NaN: a missed
 *   exchange poisons the step visibly instead of feeding it stale statistics (the caller checks status at its next
 *   synchronisation).  Asynchronous on `stream`; one workgroup.  Replaces torch.distributed.all_gather_into_tensor /
 *   all_reduce on <= nmax floats (SyncBN statistics); no reference call site (the reference is single-process).
 */
long dcahip_peer_slot_bytes(int world, int nmax);
long dcahip_peer_flag_bytes(int world);
int dcahip_peer_exchange(const float* local, int n, float* const* slots, unsigned* const* flags, int rank, int world, int nmax,
                         unsigned long long* epoch, float* out, int reduce, int* status, long timeout_us, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCAHIP_H */
