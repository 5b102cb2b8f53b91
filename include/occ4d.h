/*
 * occ4d.h -- C ABI of libocc4d.so: the occlusions-4d hot path (point-transformer
 * encode + cross-attention implicit decode of 4D query points) as hand-written
 * HIP kernels for MI355X (gfx950).
 *
 * The reference (basilevh/occlusions-4d) has NO FFI / plugin layer of its own:
 * the hot path sits behind Python nn.Module.forward signatures and all native
 * work is third-party (ATen, torch_cluster).  Each entry point below therefore
 * cites the reference call site(s) whose native work it replaces (paths relative
 * to the reference root).  INTEGRATION.md shows the ctypes binding a maintainer
 * of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (fp32 row-major unless noted); nothing is
 *     allocated, freed or synchronised inside the library;
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream);
 *   - row strides (`ld*`, `*_stride`) are in ELEMENTS;
 *   - return value: OCC4D_OK, or a negative OCC4D_E* code; occ4d_last_error()
 *     describes the most recent failure on the calling thread;
 *   - re-entrant: no global mutable state besides that thread-local message.
 */
#ifndef OCC4D_H_
#define OCC4D_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCC4D_OK 0
#define OCC4D_EINVAL (-1)   /* bad argument (maps to AssertionError / ValueError) */
#define OCC4D_ELAUNCH (-2)  /* HIP launch failure */

#define OCC4D_ABI_VERSION 4

int occ4d_abi_version(void);
const char* occ4d_last_error(void);
/* 0 in libocc4d.so.  1 in libocc4d_cpu.so, the g++ twin of the inference-path entry points (csrc_cpu/occ4d_twin.cpp:
 * host pointers, `stream` ignored, plain as-written loops; SURVEY.md 8(b) "each with a CPU twin compiled by g++ for
 * config 1").  The twin is loaded only on an explicit request (occlusions4d_amd.cpu_twin.enable()); it is not a fallback. */
int occ4d_is_cpu_twin(void);

/* ------------------------------------------------------------------------
 * K1 / K6 / K8  brute-force exact kNN, streaming top-k (k <= 16), never
 * materialises the N0 x N1 distance matrix.
 *   metric 0: d = ((dx*dx + dy*dy) + dz*dz), fp32, no FMA  -- square_distance +
 *             kNN_torch, model/point_transformer_layer.py:16-30,76-99; also the
 *             restated torch_cluster.knn of model/modules.py:142-146.
 *   metric 1: d = sqrt(fma(dz,dz, fma(dy,dy, dx*dx)))  -- torch.linalg.norm +
 *             topk(largest=False), utils/geometry.py:479-484 (my_knn_torch).
 * Ties: lowest data index first.  Neighbours are emitted nearest first.
 * out_idx: (n_query, k) int32, or int64 when idx_is_i64 != 0.
 * out_dist: (n_query, k) fp32 distances in the chosen metric, or NULL.
 * Requires 1 <= k <= 16 and n_data >= k.
 */
int occ4d_knn_f32(const float* query, int64_t q_stride, int n_query,
                  const float* data, int64_t d_stride, int n_data,
                  int k, int metric, void* out_idx, int idx_is_i64,
                  float* out_dist, void* stream);
/* Distances of caller-supplied neighbour lists idx (n_query, k) int32 in the same metric expressions (entries are
 * clamped to [0, n_data)): what occ4d_knn_f32 would have written to out_dist beside these indices.  Used when the
 * caller brings the reference's own lists (unstable-sort tie order, utils/geometry.py:484). */
int occ4d_knn_dists_f32(const float* query, int64_t q_stride, int n_query, const float* data, int64_t d_stride,
                        int n_data, const int32_t* idx, int k, int metric, float* out_dist, void* stream);

/* ------------------------------------------------------------------------
 * K5  farthest point sampling (restated torch_cluster.fps with
 * random_start=False, model/modules.py:133-135): start at index 0, then
 * repeatedly the first argmax of the running min of ((dx*dx+dy*dy)+dz*dz).
 * One workgroup, register-resident points.  n <= 32768, 1 <= m <= n.
 * 1536 <= n <= 16384 runs the spatially pruned kernel (csrc/fps_bucket.hip, several samples per round): same picks, ties included.
 * out_sorted: (m) int32 selected indices in ASCENDING order (the reference sorts
 * them, :135);  out_order: (m) int32 in selection order, or NULL.
 * When the cloud holds fewer than m distinct points the greedy rule re-picks (distance 0, lowest index):
 * out_order records every pick, out_sorted holds the distinct ones in its first entries and the rest is
 * unspecified.
 */
int occ4d_fps_f32(const float* xyz, int64_t stride, int n, int m,
                  int32_t* out_sorted, int32_t* out_order, void* stream);
/* The same with the first sample at index `start` (torch_cluster's random_start=True, the reference's training default
 * model/modules.py:133, draws it; the caller passes the draw). */
int occ4d_fps_start_f32(const float* xyz, int64_t stride, int n, int m, int start,
                        int32_t* out_sorted, int32_t* out_order, void* stream);

/* The same sampling over up to 16 cooperating workgroups (n <= 262144): the dataloader's reduction of
 * a whole clip to n_points (utils/geometry.py:353-364, torch_cluster.fps on ~172 K points) and the
 * 28 672-point training clouds.  `start` = index of the first sample (torch_cluster's random_start draws
 * it; the caller passes the draw).  Indices are bit-identical to occ4d_fps_f32 for start = 0.
 * workspace: occ4d_fps_coop_workspace_bytes() bytes of device memory, 8-byte aligned, reset by the call
 * itself (stream-ordered); after completion its LAST 8-byte word (the status) is 0 = ok; 2 = a bounded
 * inter-workgroup spin timed out and the selection was RECOMPUTED, in stream order, by the single-workgroup
 * kernel (occ4d_fps_repair_f32, enqueued by this call for every n <= 32768: the results are valid, a
 * consumer queued behind this call never sees a timed-out selection); 1 = timed out and not repaired
 * (n > 32768 only: results undefined, relaunch).  n_workgroups 0 = automatic.
 * occ4d_fps_repair_f32: the conditional launch itself -- does nothing when *status == 0, else overwrites
 * out_sorted / out_order with occ4d_fps_start_f32's selection and stores 2.
 * occ4d_fps_coop_debug (tests): spin_limit (0 = default 2^22 polls) and the round in which a time-out is
 * declared regardless of the polls (-1 = never); process-wide. */
int64_t occ4d_fps_coop_workspace_bytes(void);
int occ4d_fps_coop_f32(const float* xyz, int64_t stride, int n, int m, int start, int n_workgroups,
                       int32_t* out_sorted, int32_t* out_order, void* workspace, void* stream);
int occ4d_fps_repair_f32(const float* xyz, int64_t stride, int n, int m, int start, int32_t* out_sorted,
                         int32_t* out_order, void* status, void* stream);
int occ4d_fps_coop_debug(unsigned spin_limit, int fail_round);

/* ------------------------------------------------------------------------
 * K3 / K11  torch.nn.Linear on row tiles with fused prologue/epilogue, exact
 * fp32 on v_mfma_f32_32x32x2_f32:
 *     t = act_in(x) @ w^T + bias          act_in by relu_in: 0 identity, 1 relu, 2 swish = x * sigmoid(x)
 *                                         (the reference's activation options, model/implicit.py:46-64)
 *         + add_rows[row / add_div] - sub_rows[sub_idx[row]]      (each optional)
 *     t = relu_out ? relu(t) : t
 *     y = t + residual                                            (optional)
 * x (M,K) ldx; w (N,K) torch layout, ldw; y (M,N) ldy.  K % 4 == 0, ldx % 4 == 0,
 * ldw % 4 == 0, x and w 16-byte aligned.  Replaces the ATen addmm calls of
 * model/implicit.py:92-101,408,416-418,443, model/modules.py:61,64,152,
 * model/point_transformer_layer.py:170-176, model/model.py:167,189-190,204.
 */
typedef struct occ4d_linear_args {
  const float* x;        int64_t ldx;
  const float* w;        int64_t ldw;
  const float* bias;                      /* (N) or NULL */
  const float* residual; int64_t ldr;     /* (M,N) or NULL */
  float* y;              int64_t ldy;
  int32_t M, K, N;
  int32_t relu_in, relu_out;
  const float* add_rows; int64_t ld_add; int32_t add_div;        /* or NULL */
  const float* sub_rows; int64_t ld_sub; const int32_t* sub_idx; /* or NULL */
} occ4d_linear_args;

int occ4d_linear_f32(const occ4d_linear_args* args, void* stream);

/* ------------------------------------------------------------------------
 * E3 pieces (model/point_transformer_layer.py:148-183), K2/K4.
 *
 * pos_hidden: r[p,:] = relu(P1 @ (pos[i] - pos2[idx[p]]) + c1), p = i*k + j;
 *   pos (n,3) stride ps; pos2 (m,3) stride p2s; idx (n,k) int32; P1 (h,3), c1 (h);
 *   out (n*k, h) contiguous.  (:168,174 first Linear + ReLU of pos_mlp)
 */
int occ4d_pt_pos_hidden_f32(const float* pos, int64_t ps, const float* pos2, int64_t p2s,
                            const int32_t* idx, int n, int k, const float* P1, const float* c1,
                            int h, float* out, void* stream);

/* attn_in[p,:] = q[i,:] - kfeat[idx[p],:] + pe[p,:]   (:176 argument of attn_mlp) */
int occ4d_pt_attn_in_f32(const float* q, int64_t ldq, const float* kfeat, int64_t ldk,
                         const float* pe, const int32_t* idx, int n, int k, int d,
                         float* out, void* stream);

/* agg[i,c] = sum_j softmax_j(logits[i*k+j,c] / divisor) * (v[idx[i*k+j],c] + pe[i*k+j,c])
 * (:177,179; per-channel softmax over the k <= 16 neighbours, divisor = fp32(sqrt(d))).
 * pe may be NULL (treated as 0). */
int occ4d_pt_softmax_agg_f32(const float* logits, const float* v, int64_t ldv, const float* pe,
                             const int32_t* idx, int n, int k, int d, float divisor,
                             float* agg, int64_t ld_agg, void* stream);

/* Fused vector attention over k = 16 neighbours for the encoder widths (d a multiple of 4, d <= 288;
 * model/point_transformer_layer.py:168-179 as ONE kernel: the (n 16, 2d) hidden, (n 16, d) logits and (n 16, d)
 * positional encodings never reach HBM).  Replaces pos_hidden -> linear -> linear -> linear -> softmax_agg.
 * Merged form: aq (n, 2d) = (W1 Wq) x + bias, kt (m, 2d) = (W1 Wk) x2, vt (m, d) = to_v(x2),
 * wp (2d, 32) = W1 P2 row-major, w2 (d, 2d) = attn_mlp[2].weight, p2 (d, 32) / c2 (d) = pos_mlp[2], P1 (32, 3) / c1 (32)
 * = pos_mlp[0]; all in the reference's row-major layout (the kernel pads to its tile grid itself).
 * attn_mlp[2].bias is not an argument: constant over the neighbour axis, it cancels in the softmax. */
int occ4d_pt_self_attn16_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t qs, const float* apos,
                             int64_t as, const int32_t* idx, const float* kt, int64_t ld_kt, const float* vt,
                             int64_t ld_vt, const float* P1, const float* c1, const float* wp, const float* w2,
                             const float* p2, const float* c2, float* agg, int64_t ld_agg, int n, int m, int k, int d,
                             float divisor, void* stream);

/* Fused vector attention for d in {288, 416} (model/point_transformer_layer.py:168-179 in one
 * kernel; the (n*k, 2d) hidden, (n*k, d) logits and (n*k, d) positional encodings never
 * reach HBM).  With r_p = relu(P1 (qpos_i - apos_j) + c1), j = idx[i,s], p = (i,s):
 *   h_p      = relu(aq[i,:] - kt[j,:] + wp @ r_p)                (2d)   [aq, kt: see DESIGN.md
 *   logit_p  = w2 @ h_p + b2                                     (d)     refactoring (i)]
 *   agg[i,c] = sum_s softmax_s(logit_p[c] / divisor) * (vt[j,c] + (p2 @ r_p + c2)[c])
 * aq (n,2d), kt (m,2d), vt (m,d), wp (2d,32), w2 (d,2d), p2 (d,32), P1 (32,3); k <= 14
 * (9 queries x 14 neighbours are packed into the 128 MFMA rows of a workgroup).
 * aq, kt, w2, wp, p2 16-byte aligned, ld_aq % 4 == 0, ld_kt % 4 == 0. */
int occ4d_pt_cross_attn_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t qs,
                            const float* apos, int64_t as, const int32_t* idx,
                            const float* kt, int64_t ld_kt, const float* vt, int64_t ld_vt,
                            const float* P1, const float* c1, const float* wp,
                            const float* w2, const float* b2, const float* p2, const float* c2,
                            float* agg, int64_t ld_agg, int n, int m, int k, int d,
                            float divisor, void* stream);

/* (Rounds 1-4 also exported occ4d_pt_cross_attn_bf16x3_f32 / occ4d_pack_bf16x3_f32: the logit branch on TWO-piece split
 * bf16 MFMAs.  Not fp32-class; superseded by the three-piece kernels (occ4d_pt_cross_attn_bf16x6_f32,
 * occ4d_rowlin_bf16x6_f32) and removed in round 5 together with flag value 4 of the path entry points.) */

/* ------------------------------------------------------------------------
 * E6 pieces (model/modules.py:152-158), K7 / K12.
 * layernorm: y = LayerNorm(x) * gamma + beta (biased variance, eps as given; torch
 *   default 1e-5), followed by ReLU when relu_out != 0; in place allowed; gamma and
 *   beta both NULL => no affine.
 * maxpool_gather: z[i,c] = max_j y[idx[i*k+j], c].
 */
int occ4d_layernorm_f32(const float* x, int64_t ldx, const float* gamma, const float* beta,
                        float eps, int relu_out, float* y, int64_t ldy, int n, int d, void* stream);
int occ4d_maxpool_gather_f32(const float* y, int64_t ldy, const int32_t* idx, int n_out, int k,
                             int d, float* z, int64_t ldz, void* stream);

/* Strided row copy / fill as kernels (never hipMemcpy / hipMemset: kernel nodes in every capture): dst[i, 0:d] =
 * src[i, 0:d] / value -- the stride-8 xyz view of the input cloud made contiguous (model/model.py:168), the
 * pos | features concatenation of the encoder's outputs and their level-id channel (model/model.py:202-228). */
int occ4d_copy_rows_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int n, int d, void* stream);
int occ4d_fill_rows_f32(float* dst, int64_t ld, int n, int d, float value, void* stream);
/* Next level of a nested farthest-point chain (replaces the FPS launch of model/modules.py:133-135 for
 * the levels below the first when every level starts at its point 0): order (>= m) = level 0's selection order (original
 * indices), orig (n) = ascending original indices of the current cloud's points.  out_pos (m): ascending positions in the
 * current cloud of the first m picks; out_orig (m) = orig[out_pos] (the next cloud's `orig`).  n <= 32768. */
int occ4d_nested_fps_level_i32(const int32_t* order, const int32_t* orig, int n, int m, int32_t* out_pos,
                               int32_t* out_orig, void* stream);

/* gather_rows: out[i,:] = src[idx[i],:]  (index_points, :102-113; p_flat[inds], modules.py:137) */
int occ4d_gather_rows_f32(const float* src, int64_t lds, const int32_t* idx, int n_out, int d,
                          float* out, int64_t ldo, void* stream);

/* mean over rows: out[c] = (1/n) sum_i x[i,c]   (model/model.py:188) */
int occ4d_mean_rows_f32(const float* x, int64_t ldx, int n, int d, float* out, void* stream);

/* ------------------------------------------------------------------------
 * Decoder pieces (model/implicit.py).
 *
 * posenc (D5, :20-43): out (n, c*(2f+1)) = [p, sin(p w0), cos(p w0), ...],
 *   w_i = fp32(2*pi*base*2^i) computed in double then rounded, full-range sinf/cosf.
 */
int occ4d_posenc_f32(const float* pts, int64_t stride, int n, int c, int n_freq, double base_freq,
                     float* out, int64_t ldo, void* stream);

/* interp weights (D3, :336-337): w = 1/(d + 1e-4); w /= max(sum |w|, 1e-12)  (n,k) in place ok */
int occ4d_interp_weights_f32(const float* dist, int n, int k, float* w, void* stream);

/* x[i,:] += cvec[:] + sum_j w[i,j] * table[idx[i*k+j], :]
 * (D3 + lin_z of D4 with the exact-in-R refactoring of DESIGN.md; also plain D3
 * feature interpolation when x is zero-initialised and cvec is NULL) */
int occ4d_interp_add_f32(float* x, int64_t ldx, const float* cvec, const float* table, int64_t ldt,
                         const int32_t* idx, const float* w, int n, int k, int d, void* stream);

/* c (m, n) = a b in fp64 with element strides for both operands (a[i sam + l sak], b[l sbk + j sbn]; c contiguous):
 * the merged weights of the exact-in-R refactoring (W1 Wq etc., formed in fp64 and rounded once; their gradients in
 * training) -- model/point_transformer_layer.py:170-176 as rearranged in DESIGN.md 4 (i). */
int occ4d_matmul_f64(const double* a, int64_t sam, int64_t sak, const double* b, int64_t sbk, int64_t sbn, double* c,
                     int m, int n, int k, void* stream);

/* Fused vector attention for d = 416, K <= 14 on v_mfma_f32_16x16x4_f32 (csrc/crossattn16p.hip; the default): the same
 * contract as occ4d_pt_cross_attn_f32 (model/point_transformer_layer.py:168-179 with the merged first attn_mlp layer).  A wave
 * owns 16 pair rows and all 416 channels (no duplicated GEMM1, no spills); weights are DMA-streamed from a stage-packed
 * copy; TWO independent 4-wave workgroups share a CU out of phase (one's softmax epilogue / prologue / barrier waits run
 * under the other's MFMA stream): a workgroup handles its 9 queries in two passes of 64 pair rows over stages of 16
 * hidden units.  `wstream` holds
 * occ4d_pt_cross_attn16p_stream_floats() floats: 54 stages of 28 fragments x 256 floats;
 *   stage s < 52 = hidden units 16 s .. 16 s + 15:
 *     fragment t (t < 26 channel tiles):   [(g*16 + c)*4 + e] = W2[16 t + c][16 s + 4 g + e]
 *     fragment 26 + kh (kh < 2):           [(g*16 + r)*4 + e] = Wp[16 s + r][16 kh + 4 e + g]
 *   stage 52: fragment 2 t + kh (t < 14):        [(g*16 + c)*4 + e] = P2[16 t + c][16 kh + 4 e + g]
 *   stage 53: fragment 2 (t - 14) + kh (14 <= t < 26): the same for the remaining tiles; the last 4 fragments zero
 * (g < 4, c, r < 16, e < 4; W2 = attn_mlp[2].weight (416, 832), Wp = W1 P2 (832, 32), P2 = pos_mlp[2].weight (416, 32);
 * occ4d_pack_attn16p_stream_f32 builds it).  Two biases are not in the stream:
 * attn_mlp[2].bias is constant over the neighbour axis the softmax normalises over and cancels exactly; pos_mlp[2].bias
 * c2 must come folded into the value table: vt[j] = Wv f_j + c2 (the kernel adds P2 r_ij to it).
 * skew: phase offset given once to the later-placed workgroup of every CU in the first dispatch round, in units of
 * s_sleep(127) (about 8 K shader cycles); 0 = none.  Performance only -- results do not depend on it. */
int64_t occ4d_pt_cross_attn16p_stream_floats(void);
int occ4d_pt_cross_attn16p_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t q_stride, const float* apos,
                               int64_t a_stride, const int32_t* idx, const float* kt, int64_t ld_kt, const float* vt,
                               int64_t ld_vt, const float* P1, const float* c1, const float* wstream,
                               float* agg, int64_t ld_agg, int n, int m, int k, int d, float divisor, int skew,
                               void* stream);

/* occ4d_pt_cross_attn16p_f32 that also leaves the pre-softmax logits W2 relu(a) (attn_mlp[2].bias not added: it cancels
 * in the softmax) of pair p = i k + j in logits[p] (n k, 416), contiguous, 16-byte aligned: the training forward.  Backward
 * then needs no second GEMM2 (occ4d_pt_pair_mlp_f32 with logits = NULL).  With a_out (n k, 832), pe_out (n k, 416) and
 * c2 = pos_mlp[2].bias (all three or none) the hidden pre-activations a and pe = P2 r + c2 are left as well -- the three
 * pair tensors of occ4d_pt_pair_mlp_f32 on the same operands -- and backward recomputes nothing.
 * Same results in agg, bit for bit. */
int occ4d_pt_cross_attn16p_logits_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t q_stride, const float* apos,
                                      int64_t a_stride, const int32_t* idx, const float* kt, int64_t ld_kt, const float* vt,
                                      int64_t ld_vt, const float* P1, const float* c1, const float* wstream, float* agg,
                                      int64_t ld_agg, float* logits, float* a_out, float* pe_out, const float* c2, int n,
                                      int m, int k, int d, float divisor, int skew, void* stream);

/* Training companion of occ4d_pt_cross_attn16p_f32 (same file, same weight stream and MFMA chain, no softmax): the
 * pair tensors of the layer in its merged form, which backward needs (the ops are
 * model/point_transformer_layer.py:168-176 up to the softmax), for p = i k + j < n k:
 *   a_out[p]  = aq[i] - kt[idx[p]] + Wp r[p]            (n k, 832)   BEFORE the ReLU of attn_mlp
 *   logits[p] = W2 relu(a_out[p])                       (n k, 416)   attn_mlp[2].bias NOT added (it is constant over
 *                                                                    the neighbour axis of the softmax that follows)
 *   pe[p]     = P2 r[p] + c2                            (n k, 416)
 * r (n k, 32) = relu(P1 (pos_i - pos2_j) + c1) from occ4d_pt_pos_hidden_f32; idx (n, k); all three outputs contiguous.
 * One launch replaces three generic GEMM launches (K = 32 -> 832 with gathered rows, 832 -> 416, 32 -> 416).
 * 32-bit row offsets: n k 3328 B, n ld_aq 4 B and m ld_kt 4 B must stay below 4 GiB (chunk the queries).
 * logits == NULL (round 6): the caller already holds them (occ4d_pt_cross_attn16p_logits_f32 wrote them in the training
 * forward) -- GEMM2 is skipped, only a_out and pe are produced and the launch is bound by their stores. */
int occ4d_pt_pair_mlp_f32(const float* aq, int64_t ld_aq, const float* kt, int64_t ld_kt, const float* r,
                          const int32_t* idx, const float* c2, const float* wstream, float* a_out, float* logits,
                          float* pe, int n, int m, int k, int d, int skew, void* stream);

/* ------------------------------------------------------------------------
 * Row-resident fused trunk layers (csrc/trunk.hip), width 416 = d_hidden of every published configuration
 * (train.py:255-256).  A row tile's activations stay in registers across the layers of a block; weights are
 * streamed from a STAGE-PACKED copy (the exact LDS image of each 32-channel stage, built once per weight
 * update by the caller; layout below), fp32 exact on v_mfma_f32_16x16x4_f32.
 *
 * occ4d_resblock_f32:  y = x + W1 relu(W0 relu(x) + b0) + b1   -- ResnetBlockFC.forward, model/implicit.py:92-101
 *   (fc_0, fc_1, identity shortcut), optionally followed by  y += zconst + sum_j zw[:, j] ztab[zidx[:, j], :]
 *   -- the `x = x + lin_z[i+1](features_query)` of the NEXT block (model/implicit.py:416-417) in the exact-in-R
 *   form of occ4d_interp_add_f32.  y may alias x.
 * occ4d_rowlin_f32:  y[:, 0..n_out) = [res +] W [relu](x) + b [+ the same interpolation term], K = 416,
 *   n_out % 32 == 0 -- the Linear layers around the cross-attention (model/modules.py:61-65: layer1 merged into
 *   the query projection, layer3 + residual).  res may alias y.
 *
 * Packed weights (floats; occ4d_trunk_packed_floats(n_out) = (n_out / 32 + 1) * 13312 of them, the stage after
 * the last repeats stage 0 so that the kernels prefetch branch-free):
 *   "rows" packing of an (n_out, 416) weight (w0_packed, occ4d_rowlin's w_packed):
 *       P[s][(nt * 26 + t) * 256 + (g * 16 + r) * 4 + e] = W[32 s + 16 nt + r][16 t + 4 g + e]
 *   "cols" packing of the (416, 416) second layer of a residual block (w1_packed):
 *       P[j][(nt * 2 + tt) * 256 + (g * 16 + r) * 4 + e] = W[16 nt + r][32 j + 16 tt + 4 g + e]
 *   with r < 16, g < 4, e < 4, t < 26, nt < 2 (rows) / 26 (cols), tt < 2.
 * x, y, res: 16-byte aligned, row strides % 4 == 0; ztab / zconst / biases 16-byte aligned, ldz % 4 == 0. */
int occ4d_trunk_width(void);
int64_t occ4d_trunk_packed_floats(int n_out);
int occ4d_resblock_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w0_packed, const float* b0,
                       const float* w1_packed, const float* b1, const float* zconst, const float* ztab, int64_t ldz,
                       const int32_t* zidx, const float* zw, int kz, int n, void* stream);
int occ4d_rowlin_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed, const float* b,
                     int n_out, int relu_in, const float* res, int64_t ldr, const float* zconst, const float* ztab,
                     int64_t ldz, const int32_t* zidx, const float* zw, int kz, int n, void* stream);

/* occ4d_rowlin_f32 with an output mask: y = mask > 0 ? ([res +] W [relu](x) + b) : 0 -- the data gradient of a
 * relu_in Linear, dx = (x > 0) . (g W), with the ReLU mask applied in the epilogue (training, train.py:101-118). */
int occ4d_rowlin_masked_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed, const float* b,
                            int n_out, int relu_in, const float* res, int64_t ldr, const float* mask, int64_t ldm, int n,
                            void* stream);

/* Half-CU re-cut of the two kernels above (csrc/trunk4.hip): the same contracts, arithmetic and register layout, with
 * 4-wave workgroups of 64 rows and 26 KB stages of 16 channels, so that two workgroups -- of these kernels, of
 * occ4d_pt_cross_attn16p_f32, or of whatever the other decode stream runs -- share a CU and one's memory phases sit
 * under the other's MFMA stream.  occ4d_rowlin4_f32 takes n_out % 16 == 0.
 * Packed weights (floats; occ4d_trunk4_packed_floats(n_out) = (n_out / 16 + 1) * 6656 of them, the stage after the
 * last repeats stage 0):
 *   "rows" packing of an (n_out, 416) weight:   P[s][t * 256 + (g * 16 + r) * 4 + e] = W[16 s + r][16 t + 4 g + e]
 *   "cols" packing of the (416, 416) second layer of a residual block:
 *                                               P[j][nt * 256 + (g * 16 + r) * 4 + e] = W[16 nt + r][16 j + 4 g + e]
 *   with r < 16, g < 4, e < 4, t, nt, j < 26. */
int64_t occ4d_trunk4_packed_floats(int n_out);
int occ4d_resblock4_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w0_packed, const float* b0,
                        const float* w1_packed, const float* b1, const float* zconst, const float* ztab, int64_t ldz,
                        const int32_t* zidx, const float* zw, int kz, int n, void* stream);
int occ4d_rowlin4_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed, const float* b,
                      int n_out, int relu_in, const float* res, int64_t ldr, const float* zconst, const float* ztab,
                      int64_t ldz, const int32_t* zidx, const float* zw, int kz, int n, void* stream);
/* occ4d_rowlin4_f32 with the output mask of occ4d_rowlin_masked_f32 (training data gradients: at row counts that are
 * not a whole number of 256-workgroup rounds the half-CU kernel is the faster one, profiles/time_rowlin_tail.py). */
int occ4d_rowlin4_masked_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed, const float* b,
                             int n_out, int relu_in, const float* res, int64_t ldr, const float* mask, int64_t ldm,
                             int n, void* stream);
/* y = [mask > 0] ([relu](x) W^T + b) + skip: the data gradient of a residual block's first layer with the gradient of the
 * skip connection added AFTER the mask (model/implicit.py:66-85 backward: dx = dy + relu'(x) (dh W0)) -- the masked
 * kernel and the element-wise add of the autograd engine in one launch. */
int occ4d_rowlin4_masked_skip_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed,
                                  const float* b, int n_out, int relu_in, const float* skip, int64_t lds,
                                  const float* mask, int64_t ldm, int n, void* stream);

/* K13 post-ops (eval/inference.py:218-243): per channel op code in `ops` (G ints):
 * 0 = identity, 1 = sigmoid, 2 = clamp to [0,1].  In place. */
int occ4d_squash_f32(float* out, int64_t ld, int n, int g, const int32_t* ops_host, void* stream);

/* ------------------------------------------------------------------------
 * Device-side pre/post steps of perform_inference.
 * grid_points: cell-centred query grid (utils/geometry.py:1257-1283), x slowest / z fastest, rows
 *   (x,y,z,t); fp32 arithmetic in numpy's order, bit-identical to the reference's host grid.
 * split_count + split_write: density-threshold split (eval/inference.py:279-287) as an
 *   order-preserving stream compaction; solid rows (x,y,z,t, out[0..g)); air rows the same, or
 *   (x,y,z, density, argmax over the last n_classes columns of the (x,y,z,t,out) row -- numpy negative-slice
 *   semantics: a window wider than G reaches into the coordinates) when compress_air (:299-305).
 *   block_counts: ceil(n/256) ints (becomes the exclusive prefix); total_solid: 1 int (device). */
int occ4d_grid_points_f32(int nx, int ny, int nz, float x0, float sx, float y0, float sy, float z0, float sz,
                          float t, float* out, void* stream);
int occ4d_split_count_f32(const float* implicit_output, int64_t ld, int n, float threshold, int* block_counts,
                          int* total_solid, void* stream);
/* Radius test on a uniform grid (csrc/gridrad.hip): far[i] = 1.0f when NO target point lies within `radius` of query i,
 * else 0.0f -- the decision of the sampler's air / solid gap filter (utils/geometry.py:1164-1196: 1-NN distance of every
 * candidate to the whole target cloud > radius) without the 1-NN search: only the targets in the 27 cells around a
 * query are visited, every visited distance is the streaming kNN kernel's metric-1 expression, so the decisions are
 * those of occ4d_knn_f32(k = 1, metric = 1) followed by `dist > radius`, bit for bit.
 * occ4d_radius_grid_build_f32 sorts the targets into cells of edge >= radius_max / 0.95 (bounding box and cell size on
 * the device; at most 64^3 cells) inside `workspace` (occ4d_radius_grid_workspace_bytes(n) bytes, 16-byte aligned);
 * occ4d_radius_far_f32 answers queries for any radius <= radius_max.  Points outside the box need no special case. */
int64_t occ4d_radius_grid_workspace_bytes(int n);
int occ4d_radius_grid_build_f32(const float* xyz, int64_t stride, int n, float radius_max, void* workspace, void* stream);
int occ4d_radius_far_f32(const float* query, int64_t q_stride, int n_query, const void* workspace, float radius,
                         float* far, void* stream);
/* Exact k nearest neighbours through a uniform grid of `data` (built inside the call, in `workspace` of
 * occ4d_radius_grid_workspace_bytes(n_data) bytes, 16-byte aligned): the contract and the results of occ4d_knn_f32 with
 * int32 indices, bit for bit (same distance expressions, (distance, index) lexicographic order), for large searches --
 * a query visits the cells around its own in growing rings until its k-th distance is proven (csrc/gridrad.hip) instead of
 * all n_data points.  The caller chooses: the brute-force kernel wins below ~64 M pairs. */
int occ4d_knn_grid_f32(const float* query, int64_t q_stride, int n_query, const float* data, int64_t d_stride, int n_data,
                       int k, int metric, int32_t* out_idx, float* out_dist, void* workspace, void* stream);

/* Generic order-preserving row compaction (training-time sampler, filter_air_solid_gap utils/geometry.py:1190-1194):
 * keep row i when key[i] >= threshold (key[i] > threshold when strict).  compact_count fills block_counts
 * (ceil(n/256) ints -> exclusive prefix) and *total_kept (device int); compact_rows writes the kept rows
 * (row stride d) and, optionally, their keys. */
int occ4d_compact_count_f32(const float* key, int64_t ld, int n, float threshold, int strict, int* block_counts,
                            int* total_kept, void* stream);
int occ4d_compact_rows_f32(const float* src, int64_t ld, int n, int d, const float* key, int64_t ld_key,
                           float threshold, int strict, const int* block_offsets, float* out_rows, float* out_key,
                           void* stream);
int occ4d_split_write_f32(const float* points_query, const float* implicit_output, int64_t ld, int n, int g,
                          float threshold, const int* block_offsets, int compress_air, int n_classes,
                          float* solid, float* air, void* stream);

/* ========================================================================
 * Backward pass (the reference trains through torch autograd over the
 * ATen ops above, train.py:101-118).  Scatter reductions use fp32 atomics.
 * ======================================================================== */

/* dW (N,K) (+)= g^T x for y = x W^T: g (M,N) = dL/dy, x (M,K).  Split over M into partial products that a second
 * kernel adds up.  `workspace`: its size in floats comes ONLY from occ4d_linear_wgrad_workspace(M, N, K, &splits,
 * &floats) (host arithmetic, does not touch the GPU) -- do not derive it from `splits`: the wide-layer kernel
 * (csrc/wgrad16.hip) lays out (splits + 1) partial planes of N*K floats followed by (splits + 1) bias partials of N
 * floats, the generic kernel splits*N*K.  Pass the `splits` that call returned.
 * N, K, ldg, ldx multiples of 4; g, x 16-byte aligned. */
int occ4d_linear_wgrad_workspace(int M, int N, int K, int* splits_out, int64_t* floats_out);
int occ4d_linear_wgrad_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, int M, int N, int K,
                           float* dw, int accumulate, float* workspace, int splits, void* stream);
/* The same with the bias gradient db[n] (+)= sum_m g[m][n] fused (column sums of the staged g tiles; db may be
 * NULL) and, when relu_x, relu applied to x on load (weight gradient of a relu_in Linear). */
int occ4d_linear_wgrad_bias_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, int M, int N, int K,
                                int relu_x, float* dw, float* db, int accumulate, float* workspace, int splits,
                                void* stream);
/* out (d) (+)= column sums of x (n,d)  (bias gradients); workspace: chunks*d floats */
int occ4d_colsum_f32(const float* x, int64_t ldx, int n, int d, float* out, int accumulate,
                     float* workspace, int chunks, void* stream);
/* BatchNorm1d in TRAINING mode + ReLU of DownTransition(norm_type='batch') (model/modules.py:98-102,152; csrc/batchnorm.hip):
 * fwd: mean / var (biased) = the batch statistics of the n rows of y (written out: the caller updates the running
 * statistics from them, var * n / (n - 1) as torch does), out = relu(gamma (y - mean) / sqrt(var + eps) + beta);
 * bwd: g = dL/d out -> dx = dL/dy, dgamma, dbeta.  workspace: occ4d_bn_workspace_doubles(n, d) doubles. */
int64_t occ4d_bn_workspace_doubles(int n, int d);
int occ4d_bn_train_fwd_f32(const float* y, int64_t ldy, int n, int d, const float* gamma, const float* beta, float eps,
                           float* mean, float* var, float* out, int64_t ldo, double* workspace, void* stream);
int occ4d_bn_train_bwd_f32(const float* y, int64_t ldy, const float* g, int64_t ldg, const float* out, int64_t ldo, int n,
                           int d, const float* mean, const float* var, const float* gamma, float eps, float* dx, int64_t ldx,
                           float* dgamma, float* dbeta, double* workspace, void* stream);
/* swish (model/implicit.py:46-64) for the training path: y = x sigmoid(x); out = g d/dx [x sigmoid(x)] */
int occ4d_swish_f32(const float* x, int64_t ldx, int n, int d, float* y, int64_t ldy, void* stream);
int occ4d_swish_bwd_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, int n, int d, float* out, int64_t ldo,
                        void* stream);
/* out = ref > 0 ? g : 0   (ReLU backward) */
int occ4d_relu_mask_f32(const float* g, int64_t ldg, const float* ref, int64_t ldr, int n, int d,
                        float* out, int64_t ldo, void* stream);
/* out[idx[i],:] += scale * src[i,:]   (gather backward) */
int occ4d_scatter_add_rows_f32(const float* src, int64_t lds, const int32_t* idx, int n, int d, float scale,
                               float* out, int64_t ldo, void* stream);
/* out[i,:] = sum_{j<k} src[i*k+j,:]   (broadcast-over-neighbours backward) */
int occ4d_segment_sum_f32(const float* src, int n, int k, int d, float* out, int64_t ldo, void* stream);
/* dy[idx[i,j*],c] += dz[i,c], j* = first argmax_j y[idx[i,j],c]   (occ4d_maxpool_gather_f32 backward) */
int occ4d_maxpool_gather_bwd_f32(const float* y, int64_t ldy, const int32_t* idx, int n_out, int k, int d,
                                 const float* dz, int64_t ldz, float* dy, int64_t ldd, void* stream);
/* LayerNorm backward: g = dL/d(output before ReLU); dgamma/dbeta accumulate (both or neither) */
int occ4d_layernorm_bwd_f32(const float* x, int64_t ldx, const float* gamma, const float* g, int64_t ldg,
                            float eps, int n, int d, float* dx, int64_t lddx, float* dgamma, float* dbeta,
                            void* stream);
/* occ4d_pt_softmax_agg_f32 backward: dlogits (n*k,d), dpe (n*k,d) or NULL, dv (m,d) accumulates (atomics) or NULL.
 * With dv = NULL, d % 4 == 0, k in {8, 12, 14, 16} and 16-byte aligned rows a 16-byte-lane kernel runs (5.4 TB/s against
 * 2.1 TB/s); the caller then reduces dpe -- the per-pair value gradients, also written when pe = NULL -- per abstract
 * point itself (occ4d_segment_sum_sorted_f32 / occ4d_scatter_add_rows_f32).  exp through v_exp_f32 there. */
int occ4d_pt_softmax_agg_bwd_f32(const float* logits, const float* v, int64_t ldv, const float* pe,
                                 const int32_t* idx, int n, int k, int d, float divisor, const float* dagg,
                                 int64_t ldda, float* dlogits, float* dpe, float* dv, int64_t lddv, void* stream);
/* occ4d_pt_pos_hidden_f32 backward wrt P1 (h,3) and c1 (h) (accumulate); r = forward output, gr = dL/dr */
int occ4d_pt_pos_hidden_bwd_f32(const float* pos, int64_t ps, const float* pos2, int64_t p2s, const int32_t* idx,
                                int n, int k, int h, const float* r, const float* gr, float* dP1, float* dc1,
                                void* stream);
/* interpolation backward wrt the table: dtable[idx[i,j],:] += w[i,j] * dy[i,:] */
int occ4d_interp_bwd_f32(const float* dy, int64_t ldy, const int32_t* idx, const float* w, int n, int k, int d,
                         float* dtable, int64_t ldt, void* stream);
/* out = alpha*a + beta*b (b may be NULL);  out[i,:] = scale*vec  (mean backward) */
/* Deterministic (fixed summation order) counterparts of the atomic reductions above, selected by the host in its
 * deterministic mode (occlusions-4d_amd/ops.py: DETERMINISTIC): gradients are then bit-identical from run to run and
 * between an eager step and its hipGraph replay.
 * segment_gather_sum: out[r][c] = scale * sum_{t in [offsets[r], offsets[r+1])} (weights ? weights[order[t]] : 1)
 *   * src[(order[t] / div) * lds + c]: `order` = the pair indices sorted (stably) by target row, offsets (n_out + 1)
 *   = the segment bounds.  Serves scatter_add_rows (div 1), the value-table gradient of softmax_agg_bwd (src = its
 *   per-pair dpe output; occ4d_pt_softmax_agg_bwd_f32 accepts dv = NULL) and interp_bwd (div = k, weights = w).
 * pos_hidden_bwd_det: occ4d_pt_pos_hidden_bwd_f32 with the block partials written to `workspace`
 *   (occ4d_pt_pos_hidden_bwd_det_workspace floats) and added up in a fixed order. */
int occ4d_segment_gather_sum_f32(const float* src, int64_t lds, const int32_t* order, const int32_t* offsets,
                                 const float* weights, int div, int n_out, int d, float scale, float* out, int64_t ldo,
                                 void* stream);
/* segment_sum_sorted: the same sorted-segment reduction tuned for speed instead of reproducibility (the default of large
 * scatters onto few rows, e.g. the key-table gradient of the attention backward: 458752 pair rows of 832 floats onto 4248
 * abstract points): out[r][:] = scale * sum of src[order[t]][:] over the row's segment, every segment cut into `parts`
 * slices that are summed independently (float4 loads, four in flight) and combined with one atomic per element and slice
 * into the output, which the call clears first.  d, lds, ldo multiples of 4; src, out 16-byte aligned. */
/* The segments themselves (kernels only, capturable): order (n) int32 = the pair indices grouped by target row idx[p] in
 * [0, n_out), offsets (n_out + 1) = the group bounds; inside a group the order is unspecified (a multi-block counting
 * sort with LDS arrival ranks) -- for sums whose order does not matter.  n_out <= 16384; workspace:
 * occ4d_segments_workspace_ints(n_out) int32. */
int64_t occ4d_segments_workspace_ints(int n_out);
int occ4d_segments_build_i32(const int32_t* idx, int64_t n, int n_out, int32_t* order, int32_t* offsets, int32_t* workspace,
                             void* stream);
int occ4d_segment_sum_sorted_f32(const float* src, int64_t lds, const int32_t* order, const int32_t* offsets, int n_out,
                                 int d, int parts, float scale, float* out, int64_t ldo, void* stream);
int occ4d_pt_pos_hidden_bwd_det_workspace(int n, int k, int h, int64_t* floats);
int occ4d_pt_pos_hidden_bwd_det_f32(const float* pos, int64_t ps, const float* pos2, int64_t p2s, const int32_t* idx,
                                    int n, int k, int h, const float* r, const float* gr, float* dP1, float* dc1,
                                    float* workspace, void* stream);
/* Gradient clipping + AdamW for every parameter in three launches, no host read (train.py:107-109 clip_grad_norm_(0.2) + the
 * optimiser step): params_flat / exp_avg / exp_avg_sq hold the parameters and both moments of ALL tensors back to back
 * (tensor t at offsets[t], numels[t] elements); grad_ptrs[t] = device address of tensor t's gradient wherever the backward
 * pass left it, 0 = no gradient this step (skipped entirely, as torch skips .grad is None); grad_ptrs holds 2 n_tensors
 * words: behind the addresses, per tensor, the two floats (1 - beta1^k, sqrt(1 - beta2^k)) of its own update count k
 * (torch counts steps per parameter).  chunk_tensor / chunk_start: one entry per chunk of occ4d_adamw_chunk() elements
 * of a tensor.  All five tables are DEVICE arrays.
 * total_norm = sqrt(sum g^2) over all gradients, coef = max_norm > 0 ? min(1, max_norm / (total_norm + 1e-6)) : 1;
 * then torch.optim.AdamW's update with g * coef (amsgrad off).
 * workspace: n_chunks + 2 floats; afterwards workspace[n_chunks] = total_norm, [n_chunks + 1] = coef. */
int occ4d_adamw_chunk(void);
int occ4d_adamw_clip_f32(float* params_flat, float* exp_avg, float* exp_avg_sq, const int64_t* grad_ptrs,
                         const int64_t* offsets, const int64_t* numels, int n_tensors, const int32_t* chunk_tensor,
                         const int32_t* chunk_start, int n_chunks, float lr, float beta1, float beta2, float eps,
                         float weight_decay, float max_norm, float* workspace, void* stream);
int occ4d_axpby_f32(const float* a, int64_t lda, float alpha, const float* b, int64_t ldb, float beta, int n, int d,
                    float* out, int64_t ldo, void* stream);
int occ4d_broadcast_rows_f32(const float* vec, float scale, int n, int d, float* out, int64_t ldo, void* stream);

/* ========================================================================
 * Weight packers and PATH-LEVEL entry points (SURVEY.md 8(b) "minimum set": pt_layer_fwd, down_pool_fwd,
 * decoder_prepare_scene, decoder_query_fwd).  They take the reference's parameters IN THE REFERENCE'S LAYOUT (torch
 * Linear (out, in) row-major fp32, exactly the tensors of the checkpoint's state_dict) and run the whole launch sequence
 * of one reference forward on `stream`: a binder that keeps the reference's Python and only loads this library needs
 * nothing from the Python modules of occlusions-4d_amd -- the merged-weight algebra of DESIGN.md 4 and every stage packing above happen
 * inside the library (device kernels, no host round trip).  tests/test_gpu_cabi_only.py drives them with ctypes + torch
 * alone; the package's own nn.Modules call the same functions.
 *
 * Memory: three kinds of caller-provided device buffers, all sized by host-only query functions, 16-byte aligned:
 *   prepared   per WEIGHT UPDATE: merged matrices + stage-packed weight streams (+ fp64 scratch for forming them)
 *   scene      per ABSTRACT CLOUD: key / value / lin_z tables (the reference recomputes them per forward call, D7)
 *   workspace  per CALL: activations; contents undefined afterwards, may be shared by calls on the SAME stream
 * `flags` select kernel variants (A/B partners of the default, measured in DESIGN.md 6); prepare / scene / forward of
 * one object must be given the same flags.
 * ======================================================================== */
#define OCC4D_PATH_DEFAULT 0
#define OCC4D_PATH_UNFUSED 1        /* vector attention as the unfused kernel chain (pos_hidden .. softmax_agg) */
#define OCC4D_PATH_FIRST_GEN 2      /* d = 416: csrc/crossattn.hip instead of crossattn16p.hip */
/*      4 was OCC4D_PATH_BF16X3 (rounds 1-4): retired, the value is not reused */
#define OCC4D_PATH_GENERIC_LINEAR 8 /* generic Linear kernel instead of the row-resident trunk kernels */
#define OCC4D_PATH_TRUNK4 16        /* half-CU trunk kernels (csrc/trunk4.hip) */
#define OCC4D_PATH_BF16X6 64        /* opt-in, fp32-class: d = 416 attention GEMMs on 3-way split bf16 MFMAs, 6 partial products */
#define OCC4D_PATH_BF16X6_TRUNK 128 /* opt-in, fp32-class: the decoder's 416-input Linear layers on the same split (csrc/trunk_bf16x6.hip) */
#define OCC4D_PATH_FUSED_INTERP 32 /* A/B only (slower): lin_z table term of block i + 1 in block i's epilogue */
#define OCC4D_PATH_SPLIT_F16 256    /* with OCC4D_PATH_BF16X6 / _TRUNK: the split is fp16 x 2 pieces, 3 partial products (round 6;
                                     * half the matrix instructions; inference forwards only; |w| < 255, |activation| < 65504) */

/* Stage packers as device kernels (layouts: occ4d_resblock_f32 / occ4d_resblock4_f32 / occ4d_pt_cross_attn16p_f32).
 * w: (n_out, 416) row-major with row stride ldw. */
int occ4d_pack_trunk_rows_f32(const float* w, int64_t ldw, int n_out, float* packed, void* stream);
int occ4d_pack_trunk_cols_f32(const float* w, int64_t ldw, float* packed, void* stream);
int occ4d_pack_trunk4_rows_f32(const float* w, int64_t ldw, int n_out, float* packed, void* stream);
int occ4d_pack_trunk4_cols_f32(const float* w, int64_t ldw, float* packed, void* stream);
/* w2 (416, 832) = attn_mlp[2].weight, wp (832, 32) = W1 P2 (merged), p2 (416, 32) = pos_mlp[2].weight */
int occ4d_pack_attn16p_stream_f32(const float* w2, const float* wp, const float* p2, float* wstream, void* stream);

/* Optional profiling hook of the path-level forwards: the library records events[2 i] / events[2 i + 1] (hipEvent_t,
 * created by the caller with timing enabled) on `stream` right before / after the i-th launch of the chosen kernel
 * family inside the call, i < capacity; `used` (host, out) = launches bracketed.  No synchronisation. */
#define OCC4D_PROFILE_CROSS_ATTN 1
#define OCC4D_PROFILE_RESBLOCK 2
#define OCC4D_PROFILE_ROWLIN 3
typedef struct occ4d_launch_events {
  void** events;
  int32_t capacity, used, kernel, reserved;
} occ4d_launch_events;

/* E3 / E2: PointTransformerLayer.forward (model/point_transformer_layer.py:148-183), optionally wrapped as a
 * PointTransformerBlock (model/modules.py:45-67: z = x + layer3(layer2(layer1(x), p, x2, p2))).
 * Parameters in the reference layout: to_q (dim, dim), to_k / to_v (dim, dim2) (no bias); pos_mlp.0 (pos_hidden, 3) +
 * bias, pos_mlp.2 (dim, pos_hidden) + bias; attn_mlp.0 (2 dim, dim) + bias, attn_mlp.2 (dim, 2 dim) + bias.
 * pre_w (dim, d_in) / pre_b = layer1 or NULL; post_w (d_out, dim) / post_b = layer3 or NULL (then out = x + layer3(agg)
 * and d_out must equal d_in).  cross != 0: keys / values come from a second cloud (x2, pos2) with dim2 features;
 * cross == 0: self-attention (dim2 == dim, x2 / pos2 ignored). */
typedef struct occ4d_pt_layer_weights {
  int32_t dim, dim2, pos_hidden, cross, d_in, d_out, reserved0, reserved1;
  const float *to_q, *to_k, *to_v;
  const float *pos0_w, *pos0_b, *pos2_w, *pos2_b;
  const float *attn0_w, *attn0_b, *attn2_w, *attn2_b;
  const float *pre_w, *pre_b, *post_w, *post_b;
} occ4d_pt_layer_weights;

/* d = 416, K <= 14 vector attention with every GEMM on v_mfma_f32_16x16x32_bf16, both operands split into three bf16
 * truncation pieces (x = x1 + x2 + x3 exactly), six partial products accumulated in fp32 (csrc/crossattn_bf16x6.hip):
 * the contract of occ4d_pt_cross_attn16p_f32 (vtc = Wv f + c2; attn_mlp[2].bias cancels in the softmax), fp32-class
 * results.  wstream: occ4d_pack_attn_bf16x6_stream_f32(attn_mlp[2].weight (416, 832), merged W1 P2 (832, 32),
 * pos_mlp[2].weight (416, 32)) -> occ4d_pt_cross_attn_bf16x6_stream_floats() floats. */
int64_t occ4d_pt_cross_attn_bf16x6_stream_floats(void);
int occ4d_pack_attn_bf16x6_stream_f32(const float* w2, const float* wp, const float* p2, float* wstream, void* stream);
/* occ4d_pt_cross_attn_bf16x6_f32 that also leaves the pre-softmax logits of pair p = i k + j in logits[p] (n k, 416;
 * n k 416 < 2^31) and, with a_out (n k, 832; n k 832 < 2^31), pe_out (n k, 416) and c2 = pos_mlp[2].bias (all three or
 * none), the hidden pre-activations and pe: the training forward of the split-precision step (cf.
 * occ4d_pt_cross_attn16p_logits_f32). */
int occ4d_pt_cross_attn_bf16x6_logits_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t q_stride, const float* apos,
                                          int64_t a_stride, const int32_t* idx, const float* kt, int64_t ld_kt,
                                          const float* vtc, int64_t ld_vt, const float* pos0_w, const float* pos0_b,
                                          const float* wstream, float* agg, int64_t ld_agg, float* logits, float* a_out,
                                          float* pe_out, const float* c2, int n, int m, int k, int d, float divisor,
                                          void* stream);
/* y[:, 0 .. n_out) = [res +] W [relu](x) + b, K = 416, on v_mfma_f32_16x16x32_bf16 with both operands split into three
 * bf16 pieces and six partial products (csrc/trunk_bf16x6.hip): occ4d_rowlin_f32's contract (no interpolation term),
 * fp32-class results.  n_out in {208, 416, 832, 1664}; y may alias res, never x.  w_packed: occ4d_pack_rowlin_bf16x6_f32
 * of the (n_out, 416) weight (row stride ldw) -> occ4d_rowlin_bf16x6_packed_floats(n_out) floats. */
int64_t occ4d_rowlin_bf16x6_packed_floats(int n_out);
int occ4d_pack_rowlin_bf16x6_f32(const float* w, int64_t ldw, int n_out, float* packed, void* stream);
int occ4d_rowlin_bf16x6_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed, const float* b,
                            int n_out, int relu_in, const float* res, int64_t ldr, int n, void* stream);
/* Training, opt-in, fp32-class: occ4d_pt_pair_mlp_f32 (the recompute's pair tensors a / logits / pe) on the three-way split
 * bf16 MFMAs.  wstream: occ4d_pack_attn_bf16x6_stream_f32(W2, Wp, P2).  a (n k, 832), logits (n k, 416), pe (n k, 416)
 * contiguous; aq (n, >= 832) / kt (m, >= 832) rows 16-byte aligned with ld % 4 == 0; n k < 2^31, n ld_aq and m ld_kt
 * below 2^29 floats (32-bit row offsets). */
int occ4d_pt_pair_mlp_bf16x6_f32(const float* aq, int64_t ld_aq, const float* kt, int64_t ld_kt, const float* r,
                                 const int32_t* idx, const float* c2, const float* wstream, float* a_out, float* logits,
                                 float* pe, int n, int m, int k, int d, void* stream);
/* The same kernel with the epilogue of a training data gradient: y = [mask > 0] ([relu](x) W^T + b [+ res]) [+ res];
 * res_after_mask = 0: res before the mask (occ4d_rowlin4_masked_f32's contract), != 0: after it (the skip gradient of a
 * residual block, occ4d_rowlin4_masked_skip_f32's contract).  b and res may be NULL. */
int occ4d_rowlin_bf16x6_masked_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed,
                                   const float* b, int n_out, int relu_in, const float* res, int64_t ldr,
                                   int res_after_mask, const float* mask, int64_t ldm, int n, void* stream);
int occ4d_debug_x6_stamps(unsigned long long* out, int n_words);   /* debug: phase time stamps (OCC4D_X6_STAMPS=1) */
/* The same two kernels on v_mfma_f32_16x16x32_f16 with both operands split into TWO fp16 pieces, x = rn16(x) + rn16(x -
 * rn16(x)) (+ <= 2^-23 |x|), three partial products a1 b1 + a1 b2 + a2 b1 accumulated in fp32 (csrc/bf16x6.hpp, round 6):
 * half the matrix instructions of the bf16 scheme.  fp16's range is the contract: the packers store the pieces of w * 2^8
 * (|w| < 255; weights below 2^-10 keep an absolute 2^-33), activations are split as they are (|x| < 65504; below 0.25 the
 * second piece is an fp16 subnormal: absolute error <= 2^-25).  Inference forwards only (no masked / gradient variant).
 * Same argument contracts as the _bf16x6 entry points; the packed streams are NOT interchangeable. */
int64_t occ4d_pt_cross_attn_f16x3_stream_floats(void);
int occ4d_pack_attn_f16x3_stream_f32(const float* w2, const float* wp, const float* p2, float* wstream, void* stream);
int occ4d_pt_cross_attn_f16x3_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t q_stride, const float* apos,
                                  int64_t a_stride, const int32_t* idx, const float* kt, int64_t ld_kt, const float* vtc,
                                  int64_t ld_vt, const float* pos0_w, const float* pos0_b, const float* wstream, float* agg,
                                  int64_t ld_agg, int n, int m, int k, int d, float divisor, void* stream);
/* ... with aq and kt already multiplied by occ4d_pt_cross_attn_f16x3_hidden_scale() (2^4: the scale the kernel keeps its
 * hidden activations at; the path-level entry points scale the merged matrices behind Aq / Kt once per weight update, so no
 * scaling instruction is left in the kernel's loop).  |hidden activation| < 65504 / 2^4. */
float occ4d_pt_cross_attn_f16x3_hidden_scale(void);
int occ4d_pt_cross_attn_f16x3_prescaled_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t q_stride,
                                            const float* apos, int64_t a_stride, const int32_t* idx, const float* kt,
                                            int64_t ld_kt, const float* vtc, int64_t ld_vt, const float* pos0_w,
                                            const float* pos0_b, const float* wstream, float* agg, int64_t ld_agg, int n,
                                            int m, int k, int d, float divisor, void* stream);
/* The fp16 scheme's attention layer on v_mfma_f32_32x32x16_f16 (csrc/crossattn_f16w.hip, round 6): one 32-row tile x all 416
 * channels per wave, GEMM1 / split / init gathers once per pair row, Aq read once.  Contract of
 * occ4d_pt_cross_attn_f16x3_prescaled_f32 (aq and kt * 2^4); its own packed stream (occ4d_pack_attn_f16w_stream_f32).  The
 * path-level entry points take it for OCC4D_PATH_SPLIT_F16 only with OCC4D_F16W=1 in the environment (an A/B switch: measured
 * slower than the 16 x 16 x 32 kernel). */
int64_t occ4d_pt_cross_attn_f16w_stream_floats(void);
int occ4d_pack_attn_f16w_stream_f32(const float* w2, const float* wp, const float* p2, float* wstream, void* stream);
int occ4d_pt_cross_attn_f16w_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t q_stride, const float* apos,
                                 int64_t a_stride, const int32_t* idx, const float* kt, int64_t ld_kt, const float* vtc,
                                 int64_t ld_vt, const float* pos0_w, const float* pos0_b, const float* wstream, float* agg,
                                 int64_t ld_agg, int n, int m, int k, int d, float divisor, void* stream);
/* Training loss of the published configurations and its gradient in two launches (csrc/loss.hip, round 6; loss.py:50-64,
 * 156-173, 243-250, 276-277): out (cells, n, g) raw decoder outputs (row stride ldo), target (cells, n, > label_col) with the
 * density target in column 0 and the semantic label (float, < 0 = unlabelled) in column label_col;
 *   loss[0] = sum_cells [ density_lw mean_i BCEwithLogits(out[i, 0], target[i, 0])
 *                         + segmentation_lw mean_{label_i >= 0} CE(out[i, g - semantic_classes :], label_i) ] / cells
 * grad (cells, n, g) (row stride ldg) or NULL: d loss / d out.  workspace: occ4d_implicit_loss_workspace_floats(cells) floats.
 * Deterministic (fixed-order partial sums).  Colour / tracking terms are not covered (weights 0 in the published configs). */
int64_t occ4d_implicit_loss_workspace_floats(int cells);
int occ4d_implicit_loss_f32(const float* out, int64_t ldo, const float* target, int64_t ldt, int cells, int n, int g, int label_col,
                            int semantic_classes, float density_lw, float segmentation_lw, float* workspace, float* loss,
                            float* grad, int64_t ldg, void* stream);

/* ResnetBlockFC (model/implicit.py:92-101), width 416, relu, as ONE launch in the fp16 two-piece scheme (csrc/resblock_f16x3.hip,
 * round 6): y = x + W1 relu(W0 relu(x) + b0) + b1 with the hidden activation in registers (the two-launch form moves it through
 * HBM twice).  y may alias x (a workgroup owns whole rows).  w_packed: occ4d_pack_resblock_f16x3_f32 of fc_0.weight, fc_1.weight
 * (416, 416; row strides ld0, ld1) -> occ4d_resblock_f16x3_packed_floats() floats.  Range contract of the scheme:
 * |weight| < 255, |x|, |hidden| < 65504.  The path-level decoder entry points use it for OCC4D_PATH_BF16X6_TRUNK |
 * OCC4D_PATH_SPLIT_F16 unless OCC4D_F16_RESBLOCK=0 is in the environment (A/B: two occ4d_rowlin_f16x3_f32 launches). */
int64_t occ4d_resblock_f16x3_packed_floats(void);
int occ4d_pack_resblock_f16x3_f32(const float* w0, int64_t ld0, const float* w1, int64_t ld1, float* packed, void* stream);
int occ4d_resblock_f16x3_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed, const float* b0,
                             const float* b1, int n, void* stream);
int64_t occ4d_rowlin_f16x3_packed_floats(int n_out);
int occ4d_pack_rowlin_f16x3_f32(const float* w, int64_t ldw, int n_out, float* packed, void* stream);
int occ4d_rowlin_f16x3_f32(const float* x, int64_t ldx, float* y, int64_t ldy, const float* w_packed, const float* b,
                           int n_out, int relu_in, const float* res, int64_t ldr, int n, void* stream);
int occ4d_pt_cross_attn_bf16x6_f32(const float* aq, int64_t ld_aq, const float* qpos, int64_t q_stride, const float* apos,
                                   int64_t a_stride, const int32_t* idx, const float* kt, int64_t ld_kt, const float* vtc,
                                   int64_t ld_vt, const float* pos0_w, const float* pos0_b, const float* wstream, float* agg,
                                   int64_t ld_agg, int n, int m, int k, int d, float divisor, void* stream);

int64_t occ4d_pt_layer_prepared_floats(const occ4d_pt_layer_weights* w, int flags);
int occ4d_pt_layer_prepare_f32(const occ4d_pt_layer_weights* w, float* prepared, int flags, void* stream);
/* per-scene key / value tables of a cross layer: (W1 Wk) x2 (m, 2 dim), Wv x2 (m, dim), Wv x2 + c2 (m, dim) */
int64_t occ4d_pt_layer_scene_floats(const occ4d_pt_layer_weights* w, int m);
int occ4d_pt_layer_scene_f32(const occ4d_pt_layer_weights* w, const float* prepared, const float* x2, int64_t ldx2,
                             int m, float* scene, int flags, void* stream);
int64_t occ4d_pt_layer_workspace_floats(const occ4d_pt_layer_weights* w, int n, int m, int k, int flags);
/* x (n, d_in), pos (n, >= 3); cross: x2 (m, dim2), pos2 (m, >= 3).  k = num_neighbors <= 16.  knn_idx: (n, k) int32 =
 * kNN_torch(pos, pos2, k) when the caller already has it, or NULL (computed: occ4d_knn_f32 metric 0).  scene: the
 * tables of occ4d_pt_layer_scene_f32 for (x2, these weights), or NULL (computed per call, as the reference does).
 * out (n, d_out if post else dim); may alias x only when post != NULL. */
int occ4d_pt_layer_fwd_f32(const occ4d_pt_layer_weights* w, const float* prepared, const float* x, int64_t ldx,
                           const float* pos, int64_t pos_stride, int n, const float* x2, int64_t ldx2,
                           const float* pos2, int64_t pos2_stride, int m, int k, const int32_t* knn_idx,
                           const float* scene, float* out, int64_t ldo, float* workspace, int flags,
                           occ4d_launch_events* ev, void* stream);

/* occ4d_pt_layer_fwd_f32 of a cross-attention layer that the fp32 paired-workgroup kernel or the bf16 x 3 split kernel serves
 * (dim = 416, k <= 14, fused; not the fp16 scheme; OCC4D_ERR_INVALID otherwise), which also writes the pre-softmax logits of
 * every (query, neighbour) pair to logits_out (n k, 416) -- the training forward of the recompute-in-backward attention
 * (round 6): backward reads them instead of running GEMM2 a second time.  a_out (n k, 832) and pe_out (n k, 416), both or
 * neither: the other two pair tensors as well (nothing left to recompute).  `out` is bit-identical to
 * occ4d_pt_layer_fwd_f32's. */
int occ4d_pt_layer_fwd_logits_f32(const occ4d_pt_layer_weights* w, const float* prepared, const float* x, int64_t ldx,
                                  const float* pos, int64_t pos_stride, int n, const float* x2, int64_t ldx2,
                                  const float* pos2, int64_t pos2_stride, int m, int k, const int32_t* knn_idx,
                                  const float* scene, float* out, int64_t ldo, float* logits_out, float* a_out, float* pe_out,
                                  float* workspace, int flags, occ4d_launch_events* ev, void* stream);

/* E6 feature half: DownTransition.forward after its FPS / kNN (model/modules.py:152-158): y = ReLU(norm(Linear(x))) on
 * ALL n points, z[i] = max_j y[nn_idx[i, j]].  norm 0 none, 1 LayerNorm(gamma, beta, eps), 2 BatchNorm1d in eval mode
 * (running mean / var, gamma, beta, eps) -- model/modules.py:95-109.  w (d_out, d_in), b (d_out).
 * workspace: n * d_out floats.  nn_idx (n_new, k) int32 from occ4d_knn_f32(p_sub, p, k, metric 0). */
int occ4d_down_pool_fwd_f32(const float* x, int64_t ldx, int n, int d_in, const float* w, const float* b, int d_out,
                            int norm, const float* gamma, const float* beta, const float* mean, const float* var,
                            float eps, const int32_t* nn_idx, int n_new, int k, float* z, int64_t ldz,
                            float* workspace, void* stream);

/* D1-D7: LocalPclResnetFC (model/implicit.py:110-150,217-269 parameters; :271-445 forward), local_mode 'attention'
 * (n_cross >= 1) or 'feature' (n_cross == 0), B == 1.  lin_in (d_hidden, lin_in_ld >= d_in (2 n_freq + 1)), lin_z[i]
 * (d_hidden, d_latent) with the GLOBAL part in the first d_latent - d_latent_local columns (model/implicit.py:342),
 * blocks[i].fc_0 / fc_1 (d_hidden, d_hidden), lin_out (d_out, d_hidden); cross[j] = pt_blocks[j] as an
 * occ4d_pt_layer_weights with cross = 1, pre = layer1, post = layer3, applied after block cross_after[j]
 * (model/implicit.py:265: int((j + 1) n_blocks / (n_cross + 1))).  activation 0 = relu, 1 = swish (:46-64). */
#define OCC4D_MAX_BLOCKS 16
#define OCC4D_MAX_CROSS 4
typedef struct occ4d_decoder_weights {
  int32_t d_in, n_freq, d_hidden, d_out, d_latent, d_latent_local, n_blocks, n_cross, k_local, k_cross, activation,
      lin_in_ld;
  float base_frequency, reserved;
  const float *lin_in_w, *lin_in_b, *lin_out_w, *lin_out_b;
  const float* lin_z_w[OCC4D_MAX_BLOCKS];
  const float* lin_z_b[OCC4D_MAX_BLOCKS];
  const float* fc0_w[OCC4D_MAX_BLOCKS];
  const float* fc0_b[OCC4D_MAX_BLOCKS];
  const float* fc1_w[OCC4D_MAX_BLOCKS];
  const float* fc1_b[OCC4D_MAX_BLOCKS];
  int32_t cross_after[OCC4D_MAX_CROSS];
  occ4d_pt_layer_weights cross[OCC4D_MAX_CROSS];
} occ4d_decoder_weights;

int64_t occ4d_decoder_prepared_floats(const occ4d_decoder_weights* w, int flags);
int occ4d_decoder_prepare_f32(const occ4d_decoder_weights* w, float* prepared, int flags, void* stream);
/* Per abstract cloud (model/implicit.py:286-290 split of pcl_abstract; D7 hoisting; refactoring (ii) of DESIGN.md 4):
 * xyz (m, >= 3) stride xyz_stride, feats (m, d_latent_local) ld_feats, fglobal (d_latent - d_latent_local). */
int64_t occ4d_decoder_scene_floats(const occ4d_decoder_weights* w, int m);
int occ4d_decoder_prepare_scene_f32(const occ4d_decoder_weights* w, const float* prepared, const float* xyz,
                                    int64_t xyz_stride, const float* feats, int64_t ld_feats, const float* fglobal,
                                    int m, float* scene, int flags, void* stream);
/* One mini-batch of queries (any n; processed in chunks of 32768 rows): queries (n, d_in) rows (x, y, z, t) ->
 * out (n, d_out) raw network outputs, penult (n, d_hidden) or NULL.
 * knn_local (n, k_local) / knn_cross (n, k_cross) int32, each optional (NULL = searched here, lowest index first on
 * equal distances): the caller's own neighbour lists of the abstract cloud -- my_knn_torch's `inds`
 * (utils/geometry.py:484, model/implicit.py:328) and kNN_torch's `knn_idx` (model/point_transformer_layer.py:96-97,167;
 * one list serves every cross layer: same coordinates, same K).  The reference orders equidistant points with an
 * unstable sort, so on clouds with coincident points (CARLA's two-level abstract cloud, model/model.py:202-228) its
 * choice at the k-th rank is implementation-defined; a caller that must reproduce one particular run passes that run's
 * lists (tests/golden g10 / g8 `knn_local`, `knn_cross`).  The inverse-distance weights are recomputed from the given
 * indices with the search kernel's distance expression (occ4d_knn_dists_f32). */
int64_t occ4d_decoder_query_workspace_floats(const occ4d_decoder_weights* w, int n, int m, int flags);
int occ4d_decoder_query_fwd_f32(const occ4d_decoder_weights* w, const float* prepared, const float* scene, int m,
                                const float* queries, int64_t q_stride, int n, const int32_t* knn_local,
                                const int32_t* knn_cross, float* out, int64_t ld_out, float* penult, int64_t ld_pen,
                                float* workspace, int flags, occ4d_launch_events* ev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OCC4D_H_ */
