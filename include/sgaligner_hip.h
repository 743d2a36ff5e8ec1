/* sgaligner_hip.h -- C ABI of libsga_hip.so (sgaligner_amd/csrc), the MI355X (gfx950) implementation of
 * SGAligner's node-embedding + matching hot path.
 *
 * The reference (sayands/sgaligner) is pure Python and has no FFI; each entry point below states the
 * reference code it replaces (paths relative to the reference repository root).  A reference-side caller
 * binds these with ctypes (INTEGRATION.md shows the stub); sgaligner_amd/_lib.py is that binding.
 *
 * Conventions
 *   - plain device pointers + sizes; the library never allocates, frees or retains memory;
 *   - all matrices row-major fp32 unless stated, base pointers 16-byte aligned;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return 0 on success, non-zero on failure with the message available from sga_last_error()
 *     (thread-local); argument errors are detected before any launch.
 */
#ifndef SGALIGNER_HIP_H
#define SGALIGNER_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int sga_version(void);                 /* 100 * major + minor */
const char* sga_last_error(void);
int sga_device_cus(void);
/* The library is STATELESS (SURVEY 8(b) "Threading"): no entry point reads or writes a process-global setting.  Which arithmetic a
 * call uses is either in its name (sga_loss_multi_sums = exact fp32 MFMA, _centred = the same over centred tables, _bf16x6 = three exact
 * bf16 planes, *_f16 = fp16 inputs for wide tables) or an explicit argument (sga_pointnet_fwd_ws: mode).  The policy --
 * which of them a training step calls -- lives in the caller (sgaligner_amd.ops.set_mfma_mode, SGA_MFMA_MODE). */

/* ---- PointNet object encoder ------------------------------------------------------------------------
 * replaces PointNetfeat.forward, src/aligner/networks/pointnet.py:120-175 (called sg_aligner.py:115):
 *   y[t,c] = max_p relu(W3 relu(W2 relu(W1 x[t,p] + b1) + b2) + b3)[c]      (BN outputs are discarded there)
 * x [T,P,3] (data_dict['tot_obj_pts'] layout), w1 [64,3], w2 [128,64], w3 [C3,128], y [T,C3],
 * argmax [T,C3] int32 (first arg-max point per channel; NULL to skip).  C3 in {64,128,256}. */
int sga_pointnet_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                     const float* w3, const float* b3, float* y, int32_t* argmax, int T, int P, int C3,
                     void* stream);
/* Same, with a caller-owned workspace of sga_pointnet_fwd_ws_bytes(T, C3) bytes: when the batch has few objects (the reference's
 * own batch sizes, single-pair inference: T < 4 x CUs) every object is split over the 8 waves of a workgroup and the partial
 * (max, arg-max) pairs are folded by a second small kernel -- identical results, per-object latency / 8.  workspace may be NULL
 * (then this is sga_pointnet_fwd). */
size_t sga_pointnet_fwd_ws_bytes(int T, int C3);
/* mode: the forward's arithmetic.  0 = exact fp32 on v_mfma_f32_32x32x2_f32 (sga_pointnet_fwd is this); 4 = every fp32 operand as THREE exact
 * bf16 terms (8 + 8 + 8 significand bits, fp32's exponent range: the value itself), six bf16 MFMAs per product into fp32 accumulators --
 * fp32 arithmetic on the bf16 matrix pipe, as the default loss sweeps (sga_loss_multi_*_bf16x6), in both launch forms (identical bits);
 * with C3 = 256, T >= 4 x CUs (or no partials workspace) and a workspace of >= 81 920 bytes the kernel keeps the l planes of W2 / W3 there
 * and one workgroup serves whole objects (1.15 x faster).  (pointnet.py:140-161) */
int sga_pointnet_fwd_ws(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                        const float* w3, const float* b3, float* y, int32_t* argmax, int T, int P, int C3,
                        void* workspace, size_t ws_bytes, int mode, void* stream);
/* The forward in mode 0 (exact fp32) or 4 (three exact bf16 planes) that ALSO delivers the side effect of the reference's training forward: its three BatchNorm calls discard
 * their output but fold the batch statistics of the pre-ReLU conv outputs over all T*P points into running_mean / running_var
 * (pointnet.py:141-142,154-155,158-159).  The sums are taken inside the forward kernel (no second pass over the activations) and folded in
 * a fixed order.  bn_sums, 265 + 2 C3 doubles:
 *   [0, 9)            sum over the points of x0, x1, x2, x0x0, x0x1, x0x2, x1x1, x1x2, x2x2   (z1 = W1 x + b1 is affine in x: the caller
 *                     forms mean and variance of the 64 channels from these moments)
 *   [9, 137)          sum z2[c]         [137, 265)          sum z2[c]^2        z2 = W2 relu(z1) + b2
 *   [265, 265 + C3)   sum u[c]          [265 + C3, + 2 C3)  sum u[c]^2         u = z3 - b3 = W3 relu(z2)
 * bn_workspace: sga_pointnet_fwd_bn_ws_bytes(T, C3) bytes, caller-owned.  workspace / ws_bytes as in sga_pointnet_fwd_ws (may be NULL). */
size_t sga_pointnet_fwd_bn_ws_bytes(int T, int C3);
int sga_pointnet_fwd_bn(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                        const float* w3, const float* b3, float* y, int32_t* argmax, int T, int P, int C3,
                        void* workspace, size_t ws_bytes, void* bn_workspace, size_t bn_ws_bytes, double* bn_sums, int mode, void* stream);
/* autograd of the above wrt the six parameters (sparse through the max-pool: pointnet.py:140-161 through the arg-max points); gy [T,C3]; C3 == 256.
 * mode: the arithmetic of the three winner-row GEMMs (Z2 recomputation, dH1 = dZ2 W2, gW2 += dZ2^T H1) -- 0 = fp32 MFMA, 4 = three exact bf16
 * planes, six bf16 MFMAs per product into fp32 accumulators (as sga_pointnet_fwd_ws mode 4).  Everything else is fp32 VALU in both. */
int sga_pointnet_bwd(const float* x, const int32_t* argmax, const float* y, const float* gy, const float* w1,
                     const float* b1, const float* w2, const float* b2, const float* w3, float* gw1, float* gb1,
                     float* gw2, float* gb2, float* gw3, float* gb3, int T, int P, int C3, int mode, void* stream);

/* ---- dense layers -----------------------------------------------------------------------------------
 * C[M,N] (+)= op(A)[M,K] op(B)[K,N] (+ bias[N]);  transX = 0: X stored [rows][K]..., see gemm.hip.
 * replaces nn.Linear fwd/bwd of object_embedding / structure_embedding / meta_embedding_rel / _attr
 * (src/aligner/sg_aligner.py:112,116,119,122) and the dS*E products of the loss backward.
 * a_is_f64 != 0: A is float64 (the collated bag-of-words features, scan3r.py:196-197) converted on load. */
int sga_gemm(int transA, int transB, int M, int N, int K, const void* A, long lda, int a_is_f64, const float* B,
             long ldb, float* C, long ldc, const float* bias, int accumulate, void* stream);
int sga_colsum(const float* X, long ld, int M, int N, float* out, int accumulate, void* stream);   /* bias grads */
/* C = A B^T (+ bias) for A [M,K], B [N,K] (Conv1d k=1 / Linear over point-major rows) with the BatchNorm batch statistics of C from the same
 * launch: sums[0..N) = column sums, sums[N..2N) = column sums of squares (fp64) -- sga_bn_stats without its pass over C.  Shapes the NT
 * kernel takes only (K % 4 == 0, 16-byte aligned rows); otherwise SGA_ERR_ARG and the caller uses sga_gemm + sga_bn_stats. */
int sga_gemm_bnstats(int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc, const float* bias,
                     double* sums, void* stream);
int sga_cast_f64_f32(const double* in, float* out, size_t n, void* stream);                        /* .float(), sg_aligner.py:73-75 */

/* ---- modality fusion --------------------------------------------------------------------------------
 * replaces MultiModalFusion.forward, src/aligner/sg_aligner.py:30-35:
 *   joint[t, m*D:(m+1)*D] = softmax(weight)[m] * embs[m][t,:] / max(||embs[m][t,:]||, 1e-12)
 * embs: host array of M device pointers [T,D]; weight [M] (the [M,1] parameter); joint [T,M*D]. */
int sga_fusion_fwd(const float* const* embs, int M, const float* weight, float* joint, int T, int D, void* stream);
size_t sga_fusion_bwd_workspace_bytes(int M);
int sga_fusion_bwd(const float* const* embs, int M, const float* weight, const float* gjoint, float* const* gembs,
                   float* gweight, int T, int D, void* workspace, size_t workspace_bytes, void* stream);

/* ---- GAT structure encoder --------------------------------------------------------------------------
 * replaces torch_geometric.nn.GATConv (2.2.0, un-vendored) as used by src/aligner/networks/gat.py:36-37,44,
 * for ALL graphs of a batch in one launch (sg_aligner.py:86-110 issues 2B sequential calls):
 * H [T,256] = x W^T (2 heads x 128), att_src/att_dst/bias [256]; edges [sumE,2] int64 graph-local (col 0 source,
 * col 1 target); node_off/edge_off [G+1] int32 prefix sums; nmax = max nodes per graph (<= 256; up to 128 the features stay in LDS).
 * out[i] = sum_j softmax_j(leaky_relu(a_s[j]+a_d[i], 0.2)) H[j] + bias, self loops normalised as PyG does.
 * Limits the CALLER must respect (the Python wrapper checks the first two on the host once per batch): node ids are
 * graph-local in [0, nodes of that graph) -- an endpoint outside that range is DROPPED by the kernels, where PyG would raise;
 * duplicate edges count with their multiplicity (as PyG's scatter does) up to 255 copies of one (source, target) pair -- the
 * multiplicity matrix is 8-bit and saturates there; when that happens bit 0 of *status (device int32, caller-zeroed, may be NULL)
 * is set, and the Python wrapper turns it into an exception (deferred by at most one step, like the edge range check).
 * A graph with more than 256 nodes is rejected with SGA_ERR_ARG.
 * complete [G] uint8 (may be NULL): 1 = graph g lists every ordered pair i != j exactly once and nothing else -- what
 * preprocessing/scan3r/preprocess.py:176-182 writes for every scene (annotated relations + the supplemented 'none' pairs); its
 * multiplicities are all 1 and the kernels do not read its edges at all (16 B per edge: 2.13 GB at configs[2], eight times per step).
 * sga_gat_complete_flags fills the array from the edge lists themselves (order-independent: an N x N bitmap per graph), once per batch. */
int sga_gat_complete_flags(const int64_t* edges, const int32_t* node_off, const int32_t* edge_off, int G, uint8_t* flags, void* stream);
int sga_gat_attn_fwd(const float* H, const float* att_src, const float* att_dst, const float* bias,
                     const int64_t* edges, const int32_t* node_off, const int32_t* edge_off, int G, int nmax,
                     float* out, int32_t* status, const uint8_t* complete, void* stream);
int sga_gat_attn_bwd(const float* H, const float* dO, const float* att_src, const float* att_dst, const int64_t* edges,
                     const int32_t* node_off, const int32_t* edge_off, int G, int nmax, float* dH, float* d_att_src,
                     float* d_att_dst, const uint8_t* complete, void* stream);
int sga_elu_fwd(const float* x, float* y, size_t n, void* stream);                    /* F.elu, gat.py:45-46 */
int sga_elu_bwd(const float* x, const float* gy, float* gx, size_t n, void* stream);

/* ---- contrastive (ICL) / alignment (IAL) loss ---------------------------------------------------------
 * replace src/aligner/losses.py: calculate_prob_dist :5-15, ICLLoss.forward :43-58, IALLoss.forward :68-97.
 * Per table: Z [R,Dp] = packed, L2-normalised rows [e1i (A) | e2i (A) | e1j (J1) | e2j (J2)], Dp = D padded to 8. */
int sga_loss_gather(const float* E, int T, int D, const int32_t* idx, int R, float* Z, int Dp, float* nrm, void* stream);
int sga_loss_scatter(const float* dZ, const float* Z, const float* nrm, const int32_t* idx, int R, int D, int Dp,
                     float* dE, void* stream);                                  /* dE += J_normalize^T dZ (atomic) */
/* Scalar-accumulator buffers (sums8, sums, out, gs, gamma below) hold (1 + sga_loss_slots()) * n doubles: the first n
 * are the result, the rest per-wave partial slots (one shared set of addresses would serialise the fp64 atomics). */
int sga_loss_slots(void);
/* the four global sums of losses.py:10-11 at two temperatures: sums8[fam*2+temp], fam = s11,s12,s22,s21 */
int sga_loss_neg_sums(const float* Z, int Dp, int A, int J1, int J2, float tau0, float tau1, double* sums8, void* stream);
/* dZ += d(sums)/dZ weighted by gs8 = dL/d(sums8) (owner-stationary sweeps, atomic accumulate) */
int sga_loss_neg_grad(const float* Z, int Dp, int A, int J1, int J2, float tau0, float tau1, const double* gs8,
                      float* dZ, void* stream);
/* the same two for the anchor shard [a_lo, a_hi) this process owns: anchor-owner sweeps cover the shard only, negative-owner sweeps see
 * only the shard's anchors -- the outputs of a partition of [0, A) sum to the unsharded ones (one process per GPU all-reduces them) */
int sga_loss_neg_sums_shard(const float* Z, int Dp, int A, int J1, int J2, float tau0, float tau1, double* sums8, int a_lo, int a_hi,
                            void* stream);
int sga_loss_neg_grad_shard(const float* Z, int Dp, int A, int J1, int J2, float tau0, float tau1, const double* gs8,
                            float* dZ, int a_lo, int a_hi, void* stream);
/* anchors x anchors terms for NT tables (modalities..., joint): out = [NT icl sums | M iala | M ialb], M = NT-1 */
/* [a_lo, a_hi) (here and below): the anchor shard this process owns (0, A on one GPU).  Outputs are that shard's partial
 * contribution; the sum over a partition of [0, A) equals the unsharded result (one process per GPU all-reduces it). */
int sga_loss_anchor_fwd(const float* const* Z, const int* Dp, int NT, int A, const double* sums, float alpha,
                        float tau_icl, float tau_ial, double* out, int a_lo, int a_hi, void* stream);
/* given coef = dL/d(out): M1[k][j*A+i] = dL/dS_k[i,j] and gs[k][8] = dL/d(sums) */
int sga_loss_anchor_bwd(const float* const* Z, const int* Dp, int NT, int A, const double* sums, float alpha,
                        float tau_icl, float tau_ial, const float* coef, float* const* M1, double* gs, int a_lo, int a_hi,
                        void* stream);

/* MFMA mode 'f16' (configs[4]: tables wider than 128 columns): the same two launches with fp16 INPUTS for the similarities of every table k
 * whose Zh[k] != NULL -- Zh[k] = the fp16 copy of Z[k]'s rows that sga_wide16_prepare writes (row pitch Dp[k] halfs); fp32 accumulate, the
 * epilogue unchanged.  Zh == NULL or Zh[k] == NULL: exact fp32 for that table.  1e-2 tolerance, like the mode's sweeps.
 * ws / ws_bytes (optional; sga_loss_anchor_f16_ws_bytes(NT, A, a_hi - a_lo) bytes; the caller passes it for WIDE tables): the 2 NT
 * similarity blocks of the shard (X1 X2^T and X2 X1^T, [A][a_hi - a_lo] fp32 each) are formed first -- tables with Zh[k] on the fp16 tile
 * core of the mode's sweeps (csrc/wide16.hip: 256 x 256 tiles, LDS-DMA), the others by the exact-fp32 NT GEMM (sga_gemm) -- and the kernels
 * run their epilogue only: same arithmetic, same results up to the fp32 summation order of the products.  ws == NULL: the one-kernel form. */
size_t sga_loss_anchor_f16_ws_bytes(int NT, int A, int ns);
int sga_loss_anchor_fwd_f16(const float* const* Z, const void* const* Zh, const int* Dp, int NT, int A, const double* sums,
                            float alpha, float tau_icl, float tau_ial, double* out, int a_lo, int a_hi, void* ws, size_t ws_bytes,
                            void* stream);
int sga_loss_anchor_bwd_f16(const float* const* Z, const void* const* Zh, const int* Dp, int NT, int A, const double* sums,
                            float alpha, float tau_icl, float tau_ial, const float* coef, float* const* M1,
                            double* gs, int a_lo, int a_hi, void* ws, size_t ws_bytes, void* stream);

/* dZ[a_lo:a_hi,:] += M1^T Z[A:2A,:] ; dZ[A:2A,:] += M1 Z[a_lo:a_hi,:]  (M1 [A, a_hi-a_lo] from sga_loss_anchor_bwd; dZ zero-initialised) */
int sga_loss_stash_grad(const float* M1, const float* Z, int A, int Dp, float* dZ, int a_lo, int a_hi, void* stream);
/* fused variants for the normal pipeline, where the last table is the fusion of the M others: every joint
 * similarity is S_J = sum_m beta_m S_m (beta_m = w_m^2 / sum w^2, w = softmax(fusion.weight), sg_aligner.py:32-34
 * + losses.py:44,73), so the 300-d table is never multiplied.  Z[m] [R+32, 104] (Dp must be 104, 32 readable rows of
 * slack), D = the real embedding width (columns D..103 are zero padding; the K step that only covers padding is skipped),
 * beta [M] device.  sums/gs [(M+1)][8] with the joint in row M; gamma[m] += dL/dbeta_m through the negatives. */
int sga_loss_multi_sums(const float* const* Z, int M, int D, const float* beta, int A, int J1, int J2, float tau0, float tau1,
                        double* sums, int a_lo, int a_hi, void* stream);
int sga_loss_multi_grad(const float* const* Z, int M, int D, const float* beta, int A, int J1, int J2, float tau0, float tau1,
                        const double* gs, float* const* dZ, double* gamma, int a_lo, int a_hi, void* stream);
/* The same two fp32-MFMA sweeps with the gradient delivered in TWO parts (src/aligner/losses.py:43-58 through F.normalize's backward: a
 * table of nearly parallel rows -- meta_embedding_rel -- has an almost radial gradient, and the tangential part that survives the
 * normalisation's Jacobian must not be the rounding residue of a large radial sum).  sga_loss_centre_tables: packed table Z [R(+32), 104]
 * (R = 2A + J1 + J2) -> Zc [R(+32), 104] = (z - zbar | b = zbar.(z - zbar) + |zbar|^2/2 | 1 | 0 | 0), zbar = 0 unless |mean row|^2 >= 1/4
 * (same rule and same fixed-order column means as sga_loss_split3_tables); stat_ws: sga_loss_centre_bytes() bytes, receives the table's
 * statistics block.  The _centred sweeps take those tables (emb_dim <= 100): the owner side reads columns 100 / 101 swapped, so the MFMAs
 * still deliver S_ij = z_i . z_j; dZ[r, 0..100) += sum_j c_rj (z_j - zbar), dZ[r, 101] += sum_j c_rj.  The A x A stash products take Zc as
 * their B operand (sga_loss_stash_grad*: same two parts); sga_loss_scatter_tangent_stat projects G - rho (z_r - zbar). */
size_t sga_loss_centre_bytes(void);
int sga_loss_centre_tables(const float* Z, int A, int J1, int J2, float* Zc, void* stat_ws, void* stream);
int sga_loss_multi_sums_centred(const float* const* Zc, int M, const float* beta, int A, int J1, int J2, float tau0, float tau1,
                                double* sums, int a_lo, int a_hi, void* stream);
int sga_loss_multi_grad_centred(const float* const* Zc, int M, const float* beta, int A, int J1, int J2, float tau0, float tau1,
                                const double* gs, float* const* dZ, double* gamma, int a_lo, int a_hi, void* stream);
int sga_loss_scatter_tangent_stat(const float* dZ, const float* Z, const float* nrm, const int32_t* idx, int R, int D,
                                  const void* stat_ws, float* dE, void* stream);
/* fused anchors x anchors terms (M in {2,3,4}): same outputs as sga_loss_anchor_fwd/bwd for tables (Z_1..Z_M, joint), with the
 * joint similarities derived in registers; bwd writes M1[m] = dL/dS_m + beta_m dL/dS_J (no joint stash) and gamma[m] += dL/dbeta_m.
 * bwd with out_terms != NULL ([(M+1) + 2M] doubles + slots, like `out` of the fwd call) ALSO returns the forward term values of the
 * anchor rows [a_lo, a_hi) from the same launch: a training step whose dL/d(terms) (`coef`) is known before the terms are -- the
 * standard loss head, losses.py:114-152 -- computes the A x A similarities once (ops.FusedContrastiveFn, one-pass mode). */
int sga_loss_anchor_multi_fwd(const float* const* Z, int M, const float* beta, int A, const double* sums, float alpha,
                              float tau_icl, float tau_ial, double* out, int a_lo, int a_hi, void* stream);
int sga_loss_anchor_multi_bwd(const float* const* Z, int M, const float* beta, int A, const double* sums, float alpha,
                              float tau_icl, float tau_ial, const float* coef, float* const* M1, double* gs, double* gamma,
                              int a_lo, int a_hi, double* out_terms, void* stream);
/* Symmetric walk of the anchors x anchors terms for an UNSHARDED anchor set (M in {2,3,4}; reference arithmetic losses.py:43-97, where
 * term (i,j) and term (j,i) are made of the same two similarities S[i,j], S[j,i]): block [a_lo, a_hi) meets the columns j >= a_lo only
 * and also evaluates the mirrored elements (j, i), j >= a_hi -- every unordered anchor pair once over the whole walk instead of twice.
 * a_lo % 32 == 0; a_hi % 32 == 0 or a_hi == A.  M1[m]: [A - a_lo, a_hi - a_lo] floats, M1[m][(j-a_lo)*ns + (i-a_lo)] = dL/dS_m[i,j];
 * M2[m]: [A - a_hi, ns], M2[m][(j-a_hi)*ns + (i-a_lo)] = dL/dS_m[j,i] (may be NULL when a_hi == A).  out_terms / gs / gamma: the block's
 * share, both elements of every pair it visits; summed over the blocks of a walk they equal sga_loss_anchor_multi_bwd's over the same
 * rows.  sga_loss_stash_grad_sym applies one table's two stashes: dX1[R] += M1^T X2[a_lo:A], dX2[a_lo:A] += M1 X1[R],
 * dX1[a_hi:A] += M2 X2[R], dX2[R] += M2^T X1[a_hi:A]. */
int sga_loss_anchor_multi_bwd_sym(const float* const* Z, int M, const float* beta, int A, const double* sums, float alpha,
                                  float tau_icl, float tau_ial, const float* coef, float* const* M1, float* const* M2, double* gs,
                                  double* gamma, int a_lo, int a_hi, double* out_terms, void* stream);
int sga_loss_stash_grad_sym(const float* M1, const float* M2, const float* Z, int A, int Dp, float* dZ, int a_lo, int a_hi,
                            void* stream);
/* The same two entry points for ONE RANK of an anchor-sharded job (new design; the reference has no multi-GPU path:
 * src/engine/base_trainer.py:70,146-158 is dead): rows [a_lo, a_hi) meet the columns [j_lo, j_hi) only; tiles at j >= mir also produce the
 * mirrored element (j, i) (stash M2, rows j - mir), tiles left of mir are visited in the ordered way and must lie in the block's own
 * square.  sga_loss_anchor_multi_bwd_sym == (j_lo, j_hi, mir) = (a_lo, A, a_hi).  M1[m]: [j_hi - j_lo, a_hi - a_lo], M2[m]: [j_hi - mir, a_hi - a_lo]. */
int sga_loss_anchor_multi_bwd_symx(const float* const* Z, int M, const float* beta, int A, const double* sums, float alpha,
                                   float tau_icl, float tau_ial, const float* coef, float* const* M1, float* const* M2,
                                   double* gs, double* gamma, int a_lo, int a_hi, int j_lo, int j_hi, int mir, double* out_terms, void* stream);
int sga_loss_stash_grad_symx(const float* M1, const float* M2, const float* Z, int A, int Dp, float* dZ, int a_lo, int a_hi,
                             int j_lo, int j_hi, int mir, void* stream);

/* ZJ[r, m*104+d] = sqrt(beta_m) Z_m[r,d] for the anchor rows (operand of the anchors x anchors kernels), and its adjoint */
int sga_loss_build_joint(const float* const* Z, int M, const float* beta, int rows, float* ZJ, void* stream);
int sga_loss_fold_joint(const float* const* Z, int M, const float* beta, const float* dZJ, int rows, float* const* dZ,
                        double* gamma2, void* stream);
/* *poison = NaN if any row norm is below F.normalize's eps (the identity above would not hold): fail loudly */
int sga_loss_check_norms(const float* nrm, int n, float* poison, void* stream);

/* ---- the two fused sweeps with every fp32 operand split EXACTLY into three bf16 terms (ops.set_mfma_mode('bf16x6'); sweep3.hip) ------
 * replaces src/aligner/losses.py:5-15 (calculate_prob_dist's anchors x negatives products) and its autograd, like sga_loss_multi_sums /
 * sga_loss_multi_grad, in fp32 arithmetic on the bf16 matrix pipe: x = h + m + l (8 + 8 + 8 significand bits, fp32's exponent range: every
 * fp32 value exactly), a product = the six partial products down to 2^-16 relative on v_mfma_f32_16x16x32_bf16 into one fp32 accumulator
 * (the dropped three are <= 2^-23 of the product), i.e. 6/16 of the fp32 MFMA's matrix time at fp32's own accuracy ("fp32 MFMA
 * or split-bf16 x3").  M = 2, 3 or 4 tables, emb_dim <= 100 (columns 100, 101 of the planes carry the row centring's bookkeeping); M = 4's
 * gradient runs as two launches, each forming all four similarities and the owner gradients of two tables (the accumulators of four
 * tables do not fit a wave's registers).
 * sga_loss_split3_tables: packed fp32 table Z [R(+32), 104] -> Zb, 32-row blocks of bf16 h / m / l planes in MFMA operand order + one
 * packed K-tail image (sga_loss_split3_bytes bytes; segments X1 | X2 | N1 | N2 each padded to whole blocks; column means summed in a
 * fixed order: the planes are bitwise reproducible).  The other arguments and every output: as sga_loss_multi_sums / sga_loss_multi_grad,
 * except that the gradient arrives in TWO parts: dZ[r, 0..100) += sum_j c_rj (z_j - zbar) and dZ[r, 101] += sum_j c_rj (zbar = 0 unless the
 * table's rows are nearly parallel, |mean row|^2 >= 1/4).  Zc (optional) receives the anchor rows as fp32 [2A, 104] = z - zbar with column
 * 101 = 1: the B operand that makes sga_loss_stash_grad* deliver the A x A part in the same two-part form.
 * sga_loss_scatter_tangent = sga_loss_scatter (the autograd of F.normalize + the row gather, losses.py:44-48) for that form: it projects
 * G - rho (z_r - zbar) instead of the summed gradient -- identical in exact arithmetic (P_r z_r = 0), and free of the cancellation that
 * costs a table of nearly parallel rows 3e-4 of its gradient when the radial part is formed in fp32 first. */
size_t sga_loss_split3_bytes(int A, int J1, int J2);
int sga_loss_split3_tables(const float* Z, int A, int J1, int J2, void* Zb, float* Zc, void* stream);
/* lite != 0 (forward sums only): similarities from the h and m planes alone (products h h + h m + m h, exact K tail; 11 instead of 20 MFMAs
 * and no l-plane traffic) -- every similarity with an unbiased ~2^-16 rounding; for global sums of >= 2^24 terms each (the caller's call) they move by
 * < 1e-7 relative.  lite == 0: all six products. */
int sga_loss_multi_sums_bf16x6(const void* const* Zb, int M, const float* beta, int A, int J1, int J2, float tau0, float tau1,
                               double* sums, int a_lo, int a_hi, int lite, void* stream);
int sga_loss_multi_grad_bf16x6(const void* const* Zb, int M, const float* beta, int A, int J1, int J2, float tau0, float tau1,
                               const double* gs, float* const* dZ, double* gamma, int a_lo, int a_hi, void* stream);
int sga_loss_scatter_tangent(const float* dZ, const float* Z, const float* nrm, const int32_t* idx, int A, int J1, int J2, int D,
                             const void* Zb, float* dE, void* stream);
/* The four stash products of sga_loss_stash_grad_symx (same M1 / M2 / block arguments; a_lo = 0, a_hi .. with j_lo = 0, j_hi = mir = A it is
 * sga_loss_stash_grad) on the three exact bf16 planes Zb of sga_loss_split3_tables: fp32 coefficients split in registers, six bf16 MFMAs per
 * product, fp32 accumulation; dZ [2A.., 104] receives the two-part form (columns 0..99 and 101) that sga_loss_scatter_tangent projects. */
int sga_loss_stash_grad_symx_bf16x6(const float* M1, const float* M2, const void* Zb, int A, int J1, int J2, float* dZ,
                                    int a_lo, int a_hi, int j_lo, int j_hi, int mir, void* stream);

/* ---- loss_group = b: the same loss on G independent groups of b consecutive pairs ------------------------
 * replaces the reference trainer feeding b pairs per iteration (configs/scan3r/scan3r_ground_truth.yaml:27,
 * src/engine/epoch_based_trainer.py:91-93 -> src/aligner/losses.py:114-152) for all B/b groups of a device batch at once:
 * out[g] / gradients equal OverallLoss's raw terms evaluated on group g alone (its anchors against ITS negatives only).
 * Z[m] [R, 104] packed normalised tables (sga_loss_gather, Dp = 104), rows X1 | X2 | N1 | N2; beta [M] as in the fused
 * global path (NULL for M == 1: ICL only).  groups [G][8] int32 = {first anchor, #anchors, first N1 row, #N1, first N2
 * row, #N2, 0, 0} (contiguous ranges inside [0,A) / [0,J1) / [0,J2)); s_off [G+1] int64 float offsets of the groups'
 * similarity blocks [2 na, na+nj1+nj2] inside one table's slab of S [M][s_total].  sums [G][NT][8], out [G][NT+2M]
 * (NT = M+1, or 1 and no IAL columns when M == 1).  bwd: coef [G][NT+2M] = dL/d(out); S (as left by fwd) is overwritten
 * with dL/dS; dZ[m] += this batch's gradient (every packed row has one writer); gamma [G][M] = dL/dbeta per group.
 * use_valu: 1 = the VALU forms of the similarity / gradient kernels (kept for cross-checks), 0 = MFMA (what the product passes). */
int sga_group_loss_fwd(const float* const* Z, int M, const float* beta, int A, int J1, const int32_t* groups, int G,
                       const int64_t* s_off, int64_t s_total, float alpha, float tau_icl, float tau_ial, float* S,
                       double* sums, double* out, int use_valu, void* stream);
int sga_group_loss_bwd(const float* const* Z, int M, const float* beta, int A, int J1, const int32_t* groups, int G,
                       const int64_t* s_off, int64_t s_total, float alpha, float tau_icl, float tau_ial, float* S,
                       const double* sums, const float* coef, float* const* dZ, double* gamma, int use_valu, void* stream);

/* ---- per-pair similarity + ranking --------------------------------------------------------------------
 * replaces eval_step's emb/||emb||, sim = 1 - emb emb^T, argsort (src/inference/sgaligner/inference_align_reg.py:
 * 125-128) fused with the rank look-ups of utils/alignment.py:3-25,27-41,59-70.  The per-pair E E^T blocks run on the
 * matrix cores (exact fp32 MFMA; f16 != 0: fp16 inputs / fp32 accumulate on the normalised rows -- BASELINE.json
 * configs[4]).  pair_off [B+1] object offsets of the pairs; blk_pair / blk_row [n_blocks]: the (pair, 64-row block) of every
 * workgroup -- list exactly the blocks that contain a query object (any order; the caller knows q_idx), so that no workgroup is
 * launched for nothing and the work spreads over all XCDs.  For query object q_idx[q] (each object at most once): rank[q] = 1-based
 * rank of q_tgt[q] among its pair's other objects (q_tgt NULL or out of the pair: -1); topk_*[q,:K] its K nearest
 * others (pair-local index, distance), ascending, ties by index.  K <= 8, <= 512 objects per pair. */
size_t sga_simrank_workspace_bytes(int T);
size_t sga_simrank_workspace_bytes_f16(int T, int D);   /* workspace for f16 != 0: + the normalised fp16 table when D > 416 */
int sga_simrank(const float* E, int T, int D, const int32_t* pair_off, const int32_t* blk_pair, const int32_t* blk_row, int n_blocks, int B,
                int max_pair_objects, const int32_t* q_idx, const int32_t* q_tgt, int Q, int K, int32_t* rank,
                int32_t* topk_idx, float* topk_sim, int f16, void* workspace, size_t workspace_bytes, void* stream);
/* per pair b (queries pair_q_off[b]..pair_q_off[b+1], in sga_simrank's order): out[b][0..4] = Hits@1..5 counts,
 * [5] = #queries, [6] = sum 1/rank, [7..9] = SGAR for modes '2', '50', '100' (utils/alignment.py:13-25,3-11,27-57);
 * topk_* with leading dimension K >= 1 (column 0 = the top-1 prediction).  out [B][12] floats. */
int sga_pair_metrics(const int32_t* rank, const int32_t* topk_idx, const float* topk_sim, int K, const int32_t* q_tgt,
                     const int32_t* pair_off, const int32_t* pair_q_off, int B, float* out, void* stream);

/* ---- Point-Cloud-Transformer object encoder, inference path ('pct' module) ---------
 * sga_gemm_ex: C = act(op(A) op(B) + bias) (+ resid), act 0 none / 1 ReLU / 2 LeakyReLU(0.2): the conv1d(k=1) +
 *   eval-mode BatchNorm (folded into weight and bias by the caller) + activation (+ SA residual) layers of
 *   src/aligner/networks/pct.py:115-124 (Embedding), :223-229 (SA tail), :289-293 (linear), :311-315 (head).
 * sga_pct_attention: SA.forward's attention (pct.py:211-222) for T objects of N points each, flash style:
 *   Q [T*N,32] (= q_conv(x) = k_conv(x): shared weight, pct.py:199), V [T*N,128] (= v_conv(x) + bias) ->
 *   Xs[j,:] = sum_i softmax_row_i(Q Q^T / sqrt(32))[i,j] V[i,:].  stats: 2*T*N floats of workspace.
 * sga_segment_max: G[t,c] = max over the N points of object t (pct.py:308), argmax (nullable) = the winning point;
 * sga_segment_max_bwd: its backward, dY[t*N + argmax[t,c], c] = dG[t,c], zero elsewhere. */
int sga_gemm_ex(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                float* C, long ldc, const float* bias, int act, const float* resid, long ldr, void* stream);
int sga_pct_attention(const float* Q, long ldq, const float* V, long ldv, int T, int N, float* stats, float* Xs,
                      long ldx, void* stream);
/* backward of sga_pct_attention (autograd of pct.py:211-222): stats as left by the forward, work = T*N floats */
int sga_pct_attention_bwd(const float* Q, long ldq, const float* V, long ldv, const float* dXs, long ldd, int T, int N,
                          const float* stats, float* work, float* dQ, long ldo, float* dV, long ldw, void* stream);
int sga_segment_max(const float* Y, long ldy, int T, int N, int C, float* G, int32_t* argmax, void* stream);
int sga_segment_max_bwd(const float* dG, const int32_t* argmax, int T, int N, int C, float* dY, long ldd, void* stream);
/* Backward of conv(512->1024, no bias) + BatchNorm + LeakyReLU(slope) + max over an object's points (pct.py:282-286,306-308) without the
 * [T*N, 1024] gradient tensors: only arg-max rows carry dL/dz and the batch-statistic terms are affine in y = cat W^T, so
 *   dW = a (x) colsum(cat) + diag(b) W (cat^T cat) + sparse,   dcat = 1 (x) a^T W + cat (W^T diag(b) W) + sparse    (pct_ops.LinearBNActMaxFn).
 * head_prep: dG/G [T,C] (gradient / value of the pooled output), fin = [scale|shift|mean|rstd] of sga_bn_finalize, R = T*N rows ->
 *   coef [T,C] = scale * dL/dz at the arg-max rows, ab [4C] = a | b | dgamma | dbeta.   head_dw: WG = W (cat^T cat) [C,K], cs = colsum(cat) [K]
 *   -> dW [C,K], Wb = diag(b) W [C,K], a0 += a^T W [K] (caller-zeroed).   head_scatter: dcat[t N + amax[t,c], :] += coef[t,c] W[c,:]. */
int sga_pct_head_prep(const float* dG, const float* G, const float* gamma, const float* beta, const float* fin, int T, int C, long R,
                      int training, float slope, float* coef, float* ab, void* stream);
int sga_pct_head_dw(const float* WG, const float* W, const float* ab, const float* cs, const float* coef, const int32_t* amax,
                    const float* cat, long ldc, int T, int N, int C, int K, float* dW, float* Wb, float* a0, void* stream);
int sga_pct_head_scatter(const float* coef, const int32_t* amax, const float* W, int T, int N, int C, int K, float* dcat, long ldd,
                         void* stream);
/* G[t,c] = max_n LeakyReLU_slope(scale[c] Y[t N + n, c] + shift[c]) with its first arg-max row: BatchNorm-apply + activation folded into
 * the point max of that stage (Y read once, never rewritten); same numbers as sga_bn_apply + sga_segment_max. */
int sga_segment_max_affine(const float* Y, long ldy, int T, int N, int C, const float* scale, const float* shift, float slope, float* G,
                           int32_t* argmax, void* stream);

/* BatchNorm1d over point-major activations [R, C] fused with the following activation (0 none, 1 ReLU, 2 LeakyReLU 0.2)
 * and residual: the train-mode layers of pct.py:122-123, :226-229, :289-293, :311-315.  sums: 2*C doubles.
 *   sga_bn_stats:     sums = [sum_r x | sum_r x^2]
 *   sga_bn_apply:     Y = act(X * scale + shift) (+ resid)       (scale = gamma * rstd, shift = beta - mean * scale)
 *   sga_bn_bwd_stats: sums = [sum_r g | sum_r g * xhat],  g = dY * act'(X * scale + shift)    (= dbeta | dgamma)
 *   sga_bn_bwd_apply: dX = scale * (g - mean_g - xhat * mean_gx)   (mean_g / mean_gx NULL: eval-mode BN, dX = scale * g) */
int sga_bn_stats(const float* X, long ldx, int R, int C, double* sums, void* stream);
/* per-channel arithmetic between the passes as one launch each: out = [scale | shift | mean | rstd] (4C floats), running statistics and
 * num_batches_tracked (int64, may be NULL) updated like nn.BatchNorm1d when training != 0 (eval: the running statistics, sums unused);
 * backward: out = [dbeta | dgamma | mean_g | mean_gx] from the sums of sga_bn_bwd_stats. */
int sga_bn_finalize(const double* sums, int R, int C, const float* gamma, const float* beta, float* running_mean, float* running_var,
                    long long* num_batches_tracked, float momentum, float eps, int training, float* out, void* stream);
int sga_bn_bwd_finalize(const double* sums, int R, int C, float* out, void* stream);
int sga_bn_apply(const float* X, long ldx, int R, int C, const float* scale, const float* shift, int act,
                 const float* resid, long ldr, float* Y, long ldy, void* stream);
int sga_bn_bwd_stats(const float* X, long ldx, const float* dY, long ldd, int R, int C, const float* scale,
                     const float* shift, const float* mean, const float* rstd, int act, double* sums, void* stream);
int sga_bn_bwd_apply(const float* X, long ldx, const float* dY, long ldd, int R, int C, const float* scale,
                     const float* shift, const float* mean, const float* rstd, const float* mean_g,
                     const float* mean_gx, int act, float* dX, long ldo, void* stream);

/* ---- per-object farthest-point sampling (the step in front of the path) -----------------
 * replaces utils/point_cloud.py:61-89 pcl_farthest_sample as called by preprocessing/scan3r/preprocess.py:96-98.
 * pts [sum N,3] f32 packed per object, offsets [n_obj+1]; start[obj] = the first sample (the reference draws it with
 * np.random.randint, :77 -- the caller owns the RNG); out_idx [n_obj, npoint] object-local indices, bit-identical to
 * the reference's sequence for the same start (fp32, same summation order, first arg-max).  Every object needs
 * N >= npoint (the reference's N < npoint branch is a random draw with replacement and stays on the host).
 * work_small / work_mid / work_large: device lists of object ids with N <= 2048, <= 8192, > 8192 points (register-
 * resident vs global-walk kernels); scratch: sga_fps_scratch_floats(total points) floats, only read when n_large > 0. */
size_t sga_fps_scratch_floats(int total_points);
int sga_fps(const float* pts, const int32_t* offsets, int n_obj, const int32_t* start, int npoint,
            const int32_t* work_small, int n_small, const int32_t* work_mid, int n_mid,
            const int32_t* work_large, int n_large, int32_t* out_idx, float* scratch, void* stream);

/* ---- convex-hull candidate filter (the other half of the step in front of the path) ----------
 * serves preprocessing/scan3r/preprocess.py:93-96 (hull = ConvexHull(obj_pcl); barycentre = mean of the hull vertices):
 * keep[i] = 0 for points that are provably interior to their object's hull (strictly inside the convex hull of the object's 26
 * directional support points, by 1e-5 of its extent), 1 otherwise -- the hull of the kept points IS the object's hull, so the
 * host's Qhull call sees a few per cent of the points and returns the same vertices.  pts [sum N,3] f32 packed per object,
 * offsets [n_obj+1]; n_planes [n_obj] (nullable): facets of the filter polytope, 0 = object kept whole (degenerate). */
int sga_hull_candidates(const float* pts, const int32_t* offsets, int n_obj, unsigned char* keep, int32_t* n_planes, void* stream);
/* The hull VERTICES of every object's candidates on the device: fp64 gift wrapping with a certificate (closed surface, Euler's
 * relation, every other point behind every facet by > 1e-9 of the object's scale).  pts [sum n, 3] f64 packed per object (the survivors
 * of sga_hull_candidates, in the object's own values), offsets [n_obj+1].  status[o] == 0: is_vertex[offsets[o] ..] flags exactly the
 * vertices scipy.spatial.ConvexHull (Qhull) reports (one copy of bitwise-duplicate points); != 0 (fewer than 4 or more than
 * sga_hull_max_candidates() points, coplanar / near-degenerate facets, any failed check): is_vertex is untouched and the caller
 * must run Qhull on that object (utils/point_cloud.py does).  preprocessing/scan3r/preprocess.py:93-96. */
int sga_hull_max_candidates(void);
int sga_hull_vertices(const double* pts, const int32_t* offsets, int n_obj, unsigned char* is_vertex, int32_t* status, void* stream);

/* Wide tables (Dp > 128) of sga_loss_neg_grad: one anchor-owner sweep writes c_ij = dL/dS_ij to a caller-owned stash (anchor-row blocks
 * sized to stash_floats; sga_loss_neg_grad_wide_floats() = everything in one block), both gradients are GEMMs on it: the K = Dp
 * similarity tile is computed once instead of 2 x ceil(Dp / 320) times.  Same results as sga_loss_neg_grad up to fp32 summation order. */
size_t sga_loss_neg_grad_wide_floats(int A, int J1, int J2);
int sga_loss_neg_grad_wide(const float* Z, int Dp, int A, int J1, int J2, float tau0, float tau1, const double* gs8, float* dZ,
                           float* stash, size_t stash_floats, void* stream);

/* fp16-input / fp32-accumulate variant of the two functions above for WIDE tables (BASELINE.json configs[4]: 1024-d embeddings,
 * "similarity GEMM at fp16"); opt-in from Python (ops.set_mfma_mode('f16')), tolerance 1e-2 on gradients (tests/test_c5_gpu.py).
 * sga_wide16_prepare: packed normalised table Z fp32 [2A+J1+J2][Dp] -> Zh fp16 (same layout) and ZhT fp16 [Dp][sga_wide16_ldt()]
 * (transposed; every segment X1|X2|N1|N2 starts at a multiple of 8 columns).  sums8 / gs8 as in sga_loss_neg_sums / _grad.
 * stash: caller-owned workspace of stash_bytes (sga_loss_neg_grad_f16_bytes() = the whole batch in one pass; less -> anchor-row blocks). */
long sga_wide16_ldt(int A, int J1, int J2);
int sga_wide16_prepare(const float* Z, int Dp, int A, int J1, int J2, void* Zh, void* ZhT, void* stream);
/* [a_lo, a_hi): the anchor shard this process owns (0, A on one GPU; a_lo a multiple of 8 for the gradient) -- the shard's anchors against ALL
 * negatives: partial sums8 (all-reduced by the caller), dZ complete for the shard's anchor rows and partial for the negatives' rows. */
int sga_loss_neg_sums_f16(const void* Zh, int Dp, int A, int J1, int J2, float tau0, float tau1, double* sums8, int a_lo, int a_hi, void* stream);
size_t sga_loss_neg_grad_f16_bytes(int A, int J1, int J2);
int sga_loss_neg_grad_f16(const void* Zh, const void* ZhT, int Dp, int A, int J1, int J2, float tau0, float tau1, const double* gs8,
                          float* dZ, void* stash, size_t stash_bytes, int a_lo, int a_hi, void* stream);
/* sga_loss_stash_grad for a wide table in the same mode (replaces the two fp32 GEMMs; the autograd of src/aligner/losses.py:6,50-57,81-94
 * through S = X1 X2^T): dZ[A + j] += sum_i M1[j, i] Z[a_lo + i], dZ[a_lo + i] += sum_j M1[j, i] Z[A + j] with the coefficients as fp16 of
 * 2^e M1 (2^e: the power of two that puts the block's largest magnitude into [2^13, 2^14), found by a max pass, undone exactly at the end),
 * the rows from ZhT (sga_wide16_prepare), fp32 accumulate.  a_lo must be a multiple of 8.  ws: sga_loss_stash_grad_f16_bytes(A, a_hi - a_lo). */
size_t sga_loss_stash_grad_f16_bytes(int A, int ns);
int sga_loss_stash_grad_f16(const float* M1, const void* ZhT, int Dp, int A, int J1, int J2, float* dZ, int a_lo, int a_hi,
                            void* ws, size_t ws_bytes, void* stream);

/* ---- scalar head of OverallLoss -----------------------------------------------------------------------
 * replaces the one-element arithmetic of src/aligner/losses.py:114-152 + CustomMultiLossLayer.forward :28-34 on the raw terms
 * sums = [S_icl[M+1] | S_ial_a[M] | S_ial_b[M]] (doubles) returned by the loss kernels:
 *   out = [loss, icl_loss_unimodal, icl_loss_multimodal, ial_loss]   (doubles);  inv_a2 = 1 / A^2, z_ial / alpha = IALLoss zoom / alpha,
 *   zoom = cfg.loss.zoom.  bwd: gout[4] = dL/d(out) -> dsums [3M+1] (type of sums), dlv_ial / dlv_icl [M] floats (the two log_vars). */
int sga_loss_head_fwd(const void* sums, int sums_f64, const float* lv_ial, const float* lv_icl, int M, double inv_a2,
                      double z_ial, double alpha, double zoom, double* out, void* stream);
int sga_loss_head_bwd(const double* gout, const void* sums, int sums_f64, const float* lv_ial, const float* lv_icl, int M,
                      double inv_a2, double z_ial, double alpha, double zoom, void* dsums, float* dlv_ial, float* dlv_icl,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SGALIGNER_HIP_H */
