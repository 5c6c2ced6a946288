/*
 * dun_hip.h -- C ABI of libdun_hip.so: the MI355X (gfx950) kernels behind DUN's depth-marginalised
 * training / prediction path.
 *
 * The reference (cambridge-mlg/DUN) is pure Python/PyTorch and has no FFI; this header is the boundary a
 * maintainer would bind (ctypes stub in INTEGRATION.md) to replace the ATen op sequences cited at each entry.
 * All pointers are DEVICE pointers unless a comment says host.  All tensors are dense, row-major fp32 unless
 * stated.  Every entry returns DUN_OK (0) or an error code; dun_last_error() gives the message.  No entry
 * allocates or frees device memory or synchronises the device: scratch comes in through workspace arguments
 * (dun_elbo_gauss_static: a fixed array inside the library) and kernels are enqueued on the hipStream_t passed
 * as `stream`, which must belong to the current device.
 *
 * Citations are file:line in the reference repository.
 */
#ifndef DUN_HIP_H
#define DUN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DUN_OK 0
#define DUN_ERR_INVALID 1     /* bad shape / null pointer / misuse, nothing was launched */
#define DUN_ERR_LAUNCH 2      /* HIP reported a launch error */
#define DUN_ERR_UNSUPPORTED 3 /* valid request outside the implemented range (e.g. > 1024 classes) */

#define DUN_OBJ_VI 0  /* ELBO with learnt categorical q(d): DUN_VI.fit, src/DUN/training_wrappers.py:220-229 */
#define DUN_OBJ_MLL 1 /* exact marginal over depth (logsumexp): DUN.fit, src/DUN/training_wrappers.py:62-74 */

typedef void* dun_stream_t; /* hipStream_t */

int dun_abi_version(void);
const char* dun_last_error(void); /* host string, valid until the next failing call on this thread */

/* ------------------------------------------------------------------------------------------------------
 * Probability module: fused per-depth likelihood pass + depth reduction + ELBO / marginal + backward.
 * Replaces depth_categorical.get_w_joint_loglike (src/probability.py:162-174), homo_Gauss_mloglike.forward
 * (src/probability.py:54-60), nn.CrossEntropyLoss(reduction='none') (src/DUN/training_wrappers.py:29),
 * get_marg_loglike / get_depth_log_posterior (src/probability.py:176-186), depth_categorical_VI.get_KL /
 * get_E_loglike / estimate_ELBO (src/probability.py:232-253) and autograd's backward through them.
 * ---------------------------------------------------------------------------------------------------- */

/* Number of double partials per depth the likelihood passes write (one per workgroup along the batch). */
int dun_ll_num_partials(int B, int C);

/* Gaussian pass, one output dimension (the reference's view(depth,batch) only works for O == 1).
 *   act [D,B], y [B], log_std [1].
 *   partials (double [D, dun_ll_num_partials(B,1)]): per-workgroup sums of r^2 = (y-act)^2.
 *   w: NULL (forward only) or [D] depth weights; w_is_logits != 0 -> softmax(w) is applied on device
 *      (q = softmax(q_logits), src/probability.py:228-230).
 *   dact [D,B] (nullable): d(-objective/N)/d act = -(w_d * inv_batch) * (y - act) / sigma^2,
 *      sigma = max(exp(log_std), 1e-4) (src/probability.py:55); inv_batch = 1 / (global batch size).
 *   ll [D,B] (nullable): per-activation log-likelihood, as loglike_per_act at src/probability.py:171. */
int dun_ll_gauss(const float* act, const float* y, const float* log_std, const float* w, int w_is_logits,
                 int D, int B, float inv_batch, float* dact, float* ll, double* partials, dun_stream_t stream);

/* Softmax cross-entropy pass.  act [D,B,C], y int64 [B].  partials: per-workgroup sums of the row
 * log-likelihood a[y] - logsumexp(a).  dact = -(w_d * inv_batch) * (onehot(y) - softmax(a)).  C <= 1024.
 * Labels as nn.CrossEntropyLoss(reduction='none') treats them (src/DUN/training_wrappers.py:29): y == -100
 * (ignore_index) contributes 0 log-likelihood and a zero gradient row; any other label outside [0,C) -- an error in
 * the reference -- contributes NaN to its depth's sum and a zero gradient row (never an out-of-bounds access), and
 * dun_predict_softmax counts such labels in metrics[2] so that a caller can raise without an extra sync. */
int dun_ll_softmax(const float* act, const int64_t* y, const float* w, int w_is_logits, int D, int B, int C,
                   float inv_batch, float* dact, float* ll, double* partials, dun_stream_t stream);

/* Sum the per-workgroup partials: sums[d] = sum_j partials[d, j]  (double [D]); this is what a batch-sharded
 * job all-reduces across ranks before dun_depth_finalize. */
int dun_ll_reduce(const double* partials, int D, int nparts, double* sums, dun_stream_t stream);

/* Layout of the result block written by dun_depth_finalize (floats). */
#define DUN_RES_OBJECTIVE 0 /* ELBO (VI) or log marginal over depth (MLL) */
#define DUN_RES_E_LOGLIKE 1 /* sum_d q_d joint_d (VI) */
#define DUN_RES_KL 2        /* KL(q || prior) with the eps clamp of src/probability.py:235 */
#define DUN_RES_LOG_MARG 3  /* logsumexp_d(joint_d + logprior_d), always */
#define DUN_RES_DLOG_STD 4  /* d(-objective/N)/d log_std (Gaussian), 0 under the 1e-4 clamp */
#define DUN_RES_HEADER 8
/* followed by 5 arrays of D floats: joint[D], weights[D] (q for VI, exact posterior for MLL),
 * dq_logits[D] (VI; zeros for MLL), posterior[D] (exact, always), loglike_sum[D] (sum_b ll[d,b]) */
#define DUN_RES_FLOATS(D) (DUN_RES_HEADER + 5 * (D))

/* sums: double [D] from dun_ll_reduce (after any all-reduce): sum_b r^2 (gaussian != 0) or sum_b ll.
 * q_logits [D] (VI) or NULL (MLL); logprior [D]; prior_ll [D] or NULL (zeros, w_prior None:
 * src/DUN/stochastic_fc_models.py:52); log_std [1] or NULL (classification).
 * joint_d = (N_train / batch) * sum_b ll[d,b] + prior_ll[d]  (src/probability.py:173). */
int dun_depth_finalize(const double* sums, int gaussian, const float* q_logits, const float* logprior,
                       const float* prior_ll, const float* log_std, int D, double N_train, double batch,
                       double beta, double eps, int objective, float* result, dun_stream_t stream);

/* dun_ll_reduce + dun_depth_finalize in ONE launch, for jobs with nothing to exchange between the two (one device):
 * partials double [D, nparts] as written by dun_ll_gauss / dun_ll_softmax; sums (nullable) receives the per-depth
 * sums, bit-identical to dun_ll_reduce's.  Same reference lines as dun_depth_finalize. */
int dun_depth_finalize_partials(const double* partials, int nparts, double* sums, int gaussian,
                                const float* q_logits, const float* logprior, const float* prior_ll,
                                const float* log_std, int D, double N_train, double batch, double beta, double eps,
                                int objective, float* result, dun_stream_t stream);

/* The whole variational objective of a Gaussian-likelihood DUN on one device, two launches: the streaming pass of
 * dun_ll_gauss (w = q_logits as logits, inv_batch = 1/B) and dun_depth_finalize_partials (batch = B, DUN_OBJ_VI).
 * Replaces depth_categorical_VI.estimate_ELBO (src/probability.py:246-253 over :162-174 and :54-60) and autograd's
 * backward through it: dact [D,B] = d(-ELBO/N_train)/d act, result as laid out above.
 * ws: scratch of at least dun_elbo_gauss_workspace_doubles(D, B) doubles (the per-workgroup partial sums). */
size_t dun_elbo_gauss_workspace_doubles(int D, int B);
int dun_elbo_gauss(const float* act, const float* y, const float* log_std, const float* q_logits,
                   const float* logprior, const float* prior_ll, int D, int B, double N_train, double beta, double eps,
                   float* dact, float* result, double* ws, size_t ws_doubles, dun_stream_t stream);

/* dun_elbo_gauss for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for code under HIP graph
 * capture, tight serving loops and harnesses that forbid hipMalloc, which cannot bring the `ws` argument.  The partial
 * sums live in a device array inside the library instead: 512 KiB per GPU, held until the process ends; inputs with
 * D * dun_ll_num_partials(B, 1) > dun_elbo_gauss_static_max_doubles() (= 65536; e.g. D = 51 up to B = 5.2 million rows)
 * are rejected with DUN_ERR_INVALID.  The call allocates and frees nothing; results are bit-identical to
 * dun_elbo_gauss.  All calls on one GPU share the array: calls that overlap on one GPU (several streams or threads)
 * race -- serialise them.  dun_static_init() looks up the current device's copy of the array (optional: the first
 * dun_elbo_gauss_static call on a device does it otherwise; call it before a graph capture begins). */
int dun_static_init(void);
size_t dun_elbo_gauss_static_max_doubles(void);
int dun_elbo_gauss_static(const float* act, const float* y, const float* log_std, const float* q_logits,
                          const float* logprior, const float* prior_ll, int D, int B, double N_train, double beta,
                          double eps, float* dact, float* result, dun_stream_t stream);

/* Depth-marginal prediction, replaces depth_categorical.marginalise_d_predict (src/probability.py:188-208)
 * and, when y != NULL, the train/eval metrics of src/DUN/training_wrappers.py:76-87, 101-113, 231-242.
 * Gaussian: act [D,n] (n = B*O), q [D] probabilities -> mean [n], std [n] (nullable) = sqrt(sum q a^2 - mean^2)
 *   (NaN for a negative variance, as model_var.pow(0.5) in the reference).
 *   metrics (nullable, double [2]; needs y [n] and log_std [1]): sum_b NLL(mean, y, sqrt(sigma^2 + std^2)),
 *   sum_b (mean - y)^2.  Overwritten (no need to zero it); partials: double [2][dun_predict_num_partials(n, 1)].
 *   log_std WITHOUT y (y = metrics = partials = NULL): `std` receives the total predictive std
 *   sqrt(std^2 + exp(log_std)^2), i.e. DUN.predict(get_std=True) of src/DUN/training_wrappers.py:153 in the same launch
 *   (the step entries below always write the model std: their log_std belongs to the metrics). */
int dun_predict_gauss(const float* act, const float* q, int D, int n, float* mean, float* std, const float* y,
                      const float* log_std, double* partials, double* metrics, dun_stream_t stream);
/* dun_predict_gauss for the end of a training step (src/DUN/training_wrappers.py:229-245): q may be given as logits
 * (q_is_logits: q = softmax(q) is taken on device, F.softmax's fp32 arithmetic -- the post-step q(d) of DUN_VI.fit) and is
 * then also written to q_out [D] (nullable); objective_src (nullable, float [1]: DUN_RES_OBJECTIVE of the step's result
 * block) is copied to objective_dst (double [1]) so that everything fit() returns comes home in one tensor. */
int dun_predict_gauss_step(const float* act, const float* q, int q_is_logits, int D, int n, float* mean, float* std,
                           const float* y, const float* log_std, double* partials, double* metrics, float* q_out,
                           const float* objective_src, double* objective_dst, dun_stream_t stream);
/* The same with a completion word: metrics / objective_dst / done may be pinned HOST memory (device-mapped), and the
 * thread that wrote the scalars stores 1.0 to done[0] behind a system-scope fence -- a host that zeroed the word before
 * enqueueing can read the step's scalars as soon as it sees the 1.0, without a stream synchronisation.  Only where the
 * predictor is one workgroup (dun_predict_num_partials(n, 1) == 1; DUN_ERR_UNSUPPORTED otherwise). */
int dun_predict_gauss_step_signal(const float* act, const float* q, int q_is_logits, int D, int n, float* mean, float* std,
                                  const float* y, const float* log_std, double* metrics, float* q_out,
                                  const float* objective_src, double* objective_dst, double* done, dun_stream_t stream);
/* Softmax: act [D,B,C] -> probs [B,C] = sum_d q_d softmax(act[d,b,:]).
 *   metrics (nullable, double [3]; needs y int64 [B]): sum_b -log probs[b,y_b], count(argmax != y), count of labels
 *   outside [0,C) other than -100.  y == -100 adds 0 to the first (nn.NLLLoss's ignore_index,
 *   src/DUN/training_wrappers.py:30) and 1 to the second (pred.ne(y), :87); an out-of-range label adds NaN to the
 *   first.  partials: double [3][dun_predict_num_partials(B, C)]. */
int dun_predict_softmax(const float* act, const float* q, int D, int B, int C, float* probs, const int64_t* y,
                        double* partials, double* metrics, dun_stream_t stream);
/* Columns of `partials` for the two predictors (an upper bound of the workgroups a call launches; C = 1 for the
 * Gaussian predictor, which has two rows; the softmax predictor has three). */
int dun_predict_num_partials(int n, int C);

/* ------------------------------------------------------------------------------------------------------
 * fp32 MFMA GEMM (v_mfma_f32_32x32x2_f32, LDS-staged 128x128x32 tiles) with the fused epilogues the
 * trunk needs.  C[M,N] = epi( sum_k A(m,k) * B(k,n) ).
 * Replaces the nn.Linear calls of src/DUN/stochastic_fc_models.py:65-69 / src/DUN/layers.py:38 (forward)
 * and the addmm/mm nodes autograd runs for them (backward).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct dun_gemm_desc {
    const float* A; /* a_kcontig: A[m*lda + k], else A[k*lda + m] */
    const float* B; /* b_kcontig: B[n*ldb + k], else B[k*ldb + n] */
    float* C;       /* C[m*ldc + n] */
    int M, N, K;
    int lda, ldb, ldc;
    int a_kcontig, b_kcontig;
    const float* bias; /* [N] added before the activation, or NULL */
    int relu;          /* max(.,0) after bias */
    const float* R;    /* residual R[m*ldr + n] added after the activation, or NULL (may alias C) */
    int ldr;
    const float* U;    /* rank-O update: C += U[m*O + o] * V[o*ldv + n]; NULL for none */
    const float* V;
    int O, ldv;
    double* colstats;  /* NULL, or fp64 [2][T][N], T = dun_gemm_colstat_tiles(M, N): per row-tile column sums of C, C^2 */
    float* splitk_ws;  /* NULL, or scratch of splitk_ws_floats floats: lets an epilogue-free problem with few output */
    size_t splitk_ws_floats; /* tiles and a long K (dW = dU^T H) split K over two workgroups per tile; >= 2*M*N to be used */
    /* NULL, or the BatchNorm backward column statistics of the NEXT layer down, taken while the output tile is in
     * registers (the trunk's dX GEMM writes the running gradient g; the BatchNorm backward of the block below starts
     * with sum_b g and sum_b g*xhat): bnz = that block's ReLU output Z [M, ldz], bn_mean / bn_rstd [N] its batch
     * statistics; bnstats float [2][T][N], T = dun_gemm_colstat_tiles(M, N): per row-tile column sums of C and of
     * C * (Z - mean) * rstd.  Not together with bias / ReLU / colstats, nor with a rank-O update of more than one
     * column (DUN_ERR_UNSUPPORTED); a residual R is taken as the accumulators' starting value. */
    const float* bnz;
    const float* bn_mean;
    const float* bn_rstd;
    float* bnstats;
    int ldz;
    /* NULL, or the inference form of a trunk block (src/DUN/layers.py:38,42 under model.eval()): BatchNorm with running
     * statistics is a per-column affine map, so C = [R +] (max(acc + bias, 0) - ev_mean) * ev_scale + ev_shift leaves
     * the GEMM as the next hidden state (ev_scale = gamma / sqrt(running_var + eps), ev_shift = beta; all [N]).  With
     * ev_head_w [N] (the shared head's weights, one output) the row-dot of the finished tile is taken as well:
     * ev_head_part float [N/128][M] receives sum over the tile's 128 columns of C[m][n] ev_head_w[n].  Needs bias and relu,
     * whole 128 x 128 tiles (>= 128 of them), no colstats / bnz / rank update (DUN_ERR_UNSUPPORTED otherwise). */
    const float* ev_mean;
    const float* ev_scale;
    const float* ev_shift;
    const float* ev_head_w;
    float* ev_head_part;
} dun_gemm_desc;
int dun_gemm_f32(const dun_gemm_desc* desc, dun_stream_t stream);
/* Measurement hook (no counterpart in the reference; bench.py's in-step roofline): between _begin and _end every
 * dun_gemm_f32 launch -- those inside dun_fc_forward / dun_fc_backward included -- is bracketed by two HIP events on its
 * stream.  _end waits for them and fills up to `cap` records in launch order, returning how many.  `epilogue` bits:
 * 1 bias, 2 ReLU, 4 residual, 8 rank-O update, 16 colstats, 32 BatchNorm-backward statistics; `kernel`: 0 register-staged,
 * 1 LDS-DMA 128x128 tiles, 2 LDS-DMA 128x256 tiles; `ms` includes the split-K sum where ksplit > 1.  HOST calls; one
 * window at a time. */
typedef struct dun_gemm_profile_rec {
    int M, N, K, a_kcontig, b_kcontig, epilogue, kernel, ksplit;
    float ms;
} dun_gemm_profile_rec;
int dun_gemm_profile_begin(int max_launches);
int dun_gemm_profile_end(dun_gemm_profile_rec* out /* host */, int cap);
/* Row tiles the kernel uses for an M x N problem (128-row tiles; 64-row tiles when the problem is too small to fill
 * the chip with 128x128 tiles).  No reference counterpart: sizing helper for `colstats`. */
int dun_gemm_colstat_tiles(int M, int N);

/* EXPLORATORY, off by default, no counterpart in the reference and never the measured path: mode 1 runs the problems
 * that qualify for the 128 x 128 LDS-DMA kernel with every fp32 operand element split exactly into three bf16 digits
 * and the product taken as six bf16 MFMAs with fp32 accumulation (terms below 2^-23 of |a b| dropped); mode 0 (the
 * default, and what every number in DESIGN.md is measured with) is exact fp32 MFMA.  Returns the previous mode;
 * any other argument only queries.  Process-wide; DUN_GEMM_BF16X3=1 in the environment selects mode 1 at first use. */
int dun_gemm_precision(int mode);

/* ------------------------------------------------------------------------------------------------------
 * Residual-MLP trunk ("forward_get_acts"): arq_uncert_fc_resnet.forward / arq_uncert_fc_MLP
 * (src/DUN/stochastic_fc_models.py:62-80) with bern_MLPBlock[_nores] (src/DUN/layers.py:33-53),
 * the shared output head applied at every depth, and the hand-derived backward of all of it.
 * ---------------------------------------------------------------------------------------------------- */
typedef struct dun_fc_tensors {
    float* w_in;  /* [W,din] */
    float* b_in;  /* [W] */
    float* w_out; /* [O,W] */
    float* b_out; /* [O] */
    float* w;     /* [L,W,W]  (out,in) per block */
    float* b;     /* [L,W] */
    float* gamma; /* [L,W] BatchNorm1d weight */
    float* beta;  /* [L,W] BatchNorm1d bias */
} dun_fc_tensors;

typedef struct dun_fc_ctx {
    int B, din, W, O, L;
    int residual; /* 1: h + BN(ReLU(Lin h)) (layers.py:42); 0: BN(ReLU(Lin h)) (layers.py:53) */
    int train;    /* 1: batch statistics + running-stat update; 0: running statistics */
    int depth;    /* blocks to run, 0..L (predict(depth=k), training_wrappers.py:142) */
    float bn_eps, bn_momentum;
    dun_fc_tensors p;    /* parameters */
    float* running_mean; /* [L,W] */
    float* running_var;  /* [L,W] */
    const float* x;      /* [B,din] */
    float* H;            /* [depth+1,B,W] hidden state per depth (kept for the head and for backward) */
    float* Z;            /* train: [depth,B,W] = ReLU(Linear) per block; eval: [B,W] scratch */
    float* bn_mean;      /* [L,W] batch mean (train) */
    float* bn_rstd;      /* [L,W] 1/sqrt(var+eps) (train) */
    double* bn_sums;     /* [2,W] staging for the statistics a sharded job all-reduces */
    float* act;          /* [depth+1,B,O] */
    float* ws;           /* scratch, >= dun_fc_workspace_floats() floats */
    size_t ws_floats;
    /* backward only */
    const float* dact;   /* [depth+1,B,O] */
    dun_fc_tensors g;    /* gradients, same shapes as p (overwritten, not accumulated) */
    float* G;            /* [B,W] running dL/dh */
    float* dU;           /* [B,W] dL/d(Linear output) of the current block */
    /* Optional (all NULL / 0 = plain forward): the Gaussian likelihood pass of dun_ll_gauss folded into the head, O == 1.
     * The head value of a row is in a register when it is written, so y - act, d(-objective/N)/d act and the
     * per-depth sums of (y - act)^2 cost no further pass over act_vec: SURVEY.md section 7 step 5, replacing the
     * act_vec round trip between src/DUN/stochastic_fc_models.py:66,69 and src/probability.py:162-174. */
    const float* obj_y;       /* [B] targets */
    const float* obj_w;       /* [depth+1] depth weights (see dun_ll_gauss `w`), or NULL: sums only, no dact */
    int obj_w_is_logits;
    const float* obj_log_std; /* [1] */
    float obj_inv_batch;      /* 1 / global batch */
    float* obj_dact;          /* [depth+1,B] d(-objective/N)/d act, or NULL */
    double* obj_partials;     /* [depth+1, dun_fc_obj_num_partials(B, O)]: per-workgroup sums of (y - act)^2, the layout
                                 dun_ll_reduce / dun_depth_finalize_partials take */
    /* Optional, with obj_w and where dun_fc_head_grads_in_forward: the head's weight / bias gradient from the forward pass.
     * d loss / d act of a row is known as soon as its head value is, and the row of h is in registers then, so
     * sum_rows dact[row] * h[row,:] is accumulated there: per-workgroup partials [depth+1, P, W] and [depth+1, P],
     * P = dun_fc_obj_num_partials.  A backward whose `dact` IS the forward's obj_dact sets head_grads_from_forward and
     * dun_fc_bwd_begin / dun_fc_backward add the partials up instead of streaming every hidden state once more
     * (dun_head_backward's pass; autograd's mm nodes behind src/DUN/stochastic_fc_models.py:66,69). */
    float* obj_dwpart;
    float* obj_dbpart;
    int head_grads_from_forward;
    /* Optional: nn.BatchNorm1d's num_batches_tracked [L] (int64).  A train-mode forward adds 1 to the entries of the
     * blocks it runs, inside the kernels that update the running statistics (torch: functional.batch_norm's caller,
     * behind src/DUN/layers.py:38), instead of the host issuing one more launch per step. */
    int64_t* num_batches_tracked;
    /* Optional: hand-off words of the one-launch forward / backward of small nets (dun_fc_sync_bytes(B, W) bytes,
     * 256-byte aligned, ZEROED ONCE by the caller when allocated and never touched by it afterwards; NULL: the
     * one-launch kernels are not used).  One buffer per context; calls that share it must not overlap. */
    uint32_t* sync;
    size_t sync_bytes;
    /* Optional, with obj_y: finish the objective inside the forward launch where dun_fc_objective_in_forward(c) says
     * so -- the workgroup that delivers the last partial sums runs dun_depth_finalize_partials' arithmetic itself
     * (same argument meaning, same result block layout; src/probability.py:173-186, 232-253), so a training step needs
     * no separate finalisation launch.  obj_fin_result NULL: not requested. */
    float* obj_fin_result;          /* [DUN_RES_FLOATS(depth+1)] */
    double* obj_fin_sums;           /* [depth+1] per-depth sums (nullable) */
    const float* obj_fin_q_logits;  /* [depth+1] (DUN_OBJ_VI) or NULL (DUN_OBJ_MLL) */
    const float* obj_fin_logprior;  /* [depth+1] */
    const float* obj_fin_prior_ll;  /* [depth+1] or NULL */
    double obj_fin_N_train, obj_fin_batch, obj_fin_beta, obj_fin_eps;
    int obj_fin_objective;          /* DUN_OBJ_VI / DUN_OBJ_MLL */
    /* 1: dun_fc_forward only leaves the partial sums and dun_fc_backward(c) -- same context, called next -- finishes the
     * objective in a spare workgroup of its weight-gradient launch, off every critical path (the variational
     * objective: its d loss / d act needs no reduced quantity, so nothing before the optimiser reads the result block;
     * the exact marginal needs the posterior before its backward and must not defer). */
    int obj_fin_defer;
} dun_fc_ctx;

size_t dun_fc_workspace_floats(int B, int din, int W, int O, int L);
/* Partial sums per depth the head writes when obj_y is set (0: not supported for this shape, i.e. O != 1). */
int dun_fc_obj_num_partials(int B, int O);
/* 1 when block_norm(i) also produces act[i+1] (width a multiple of 256 up to 2048, O == 1, 16-byte aligned planes):
 * BatchNorm apply, residual add, head row-dot and the likelihood terms in ONE pass over the [B,W] plane, so a caller of
 * the staged entries needs dun_fc_head_depth for depth 0 only. */
int dun_fc_head_is_fused(const dun_fc_ctx* c);
/* 1 when a train-mode dun_fc_forward(c) with obj_w accumulates the head's weight / bias gradient partials into
 * obj_dwpart / obj_dbpart (where dun_fc_head_is_fused, and on the one-launch small-net forward, which leaves ONE partial
 * per row-owning workgroup, [dun_fc_obj_num_partials][W], in the first rows of obj_dwpart): a caller allocates the two
 * buffers only then. */
int dun_fc_head_grads_in_forward(const dun_fc_ctx* c);

/* Whole net in one call.  W <= 128 (W % 4 == 0) with B <= 2048 -- the toy / UCI nets of experiments/toy/train_toy.py
 * and experiments/regression -- do not run the tiled GEMM sequence:
 *   mode 2 (default; needs c->sync, one output, L <= 255): ONE launch for the whole forward and ONE for the whole
 *          backward (persist.hip): a workgroup owns 16 rows of the batch through every block, hidden state in LDS,
 *          Linear layers on f32 MFMA tiles; the only data that crosses workgroups are BatchNorm's per-column partial
 *          statistics, exchanged once per block as tagged 16-byte records -- no grid barrier, no kernel boundary;
 *   mode 1: one column-owned launch per block (smallnet.hip);
 *   mode 0: the tiled path.
 * dun_fc_small_net_mode(m) selects the mode for later calls and returns the previous one (m < 0: query only); the
 * environment sets the initial mode: DUN_SMALLNET=0 -> 0, DUN_SMALLNET_PERSIST=0 -> 1.  Results of the three agree to
 * fp32 rounding.  No counterpart in the reference (launch structure only). */
int dun_fc_forward(const dun_fc_ctx* c, dun_stream_t stream);  /* whole net, BatchNorm over the local batch */
int dun_fc_backward(const dun_fc_ctx* c, dun_stream_t stream); /* whole net, needs a train-mode forward */
int dun_fc_small_net_mode(int mode);
size_t dun_fc_sync_bytes(int B, int W); /* size of dun_fc_ctx.sync for this shape (0: shape never takes mode 2) */
/* 1 when dun_fc_forward(c) -- or, with obj_fin_defer, the dun_fc_backward(c) that follows it -- will also finish the
 * objective (obj_fin_* set and the forward is one launch). */
int dun_fc_objective_in_forward(const dun_fc_ctx* c);
/* Error word of the one-launch kernels, read back from c->sync (HOST call, synchronises the stream): 0, or the code of
 * the exchange that gave up waiting for another workgroup (bounded spins; the outputs of such a call are NaN). */
int dun_fc_sync_status(const dun_fc_ctx* c, dun_stream_t stream);

/* The same work cut at the points where a batch-sharded job exchanges BatchNorm statistics:
 * forward:  input_layer; for i: block_linear(i) -> [all-reduce bn_sums] -> block_norm(i, global_batch); head
 *           (head = head_depth(d) for d = 0..depth, row for row the same numbers: a sharded job runs head_depth(i),
 *           which only needs the hidden state block i starts from, under the exchange of block i)
 * backward: bwd_begin; for i desc: bwd_block_reduce(i) -> [all-reduce bn_sums] -> bwd_block_apply(i, global_batch); bwd_end
 * bwd_block_apply(i) = bwd_block_dx(i) (dU_i, db_i, running gradient g <- [g +] dU_i W_i + dact_i w_out) followed by
 * bwd_block_dw(i) (dW_i = dU_i^T H_i).  A sharded job calls the halves itself and runs bwd_block_dw(i) AFTER
 * bwd_block_reduce(i-1), under the BatchNorm exchange of block i-1 (dU_i stays in place until bwd_block_dx(i-1)):
 *   ... bwd_block_dx(i); bwd_block_reduce(i-1); [all-reduce in flight] bwd_block_dw(i); [wait] bwd_block_dx(i-1) ... */
int dun_fc_input_layer(const dun_fc_ctx* c, dun_stream_t stream);
int dun_fc_block_linear(const dun_fc_ctx* c, int i, dun_stream_t stream);
int dun_fc_block_norm(const dun_fc_ctx* c, int i, double global_batch, dun_stream_t stream);
/* The second half of dun_fc_block_norm alone: normalise + affine + residual (+ head and likelihood terms where
 * dun_fc_head_is_fused) with the statistics block_norm left in bn_mean / bn_rstd (eval mode: running statistics).
 * For callers that time or re-run the streaming pass by itself. */
int dun_fc_block_apply(const dun_fc_ctx* c, int i, dun_stream_t stream);
int dun_fc_head(const dun_fc_ctx* c, dun_stream_t stream);
int dun_fc_head_depth(const dun_fc_ctx* c, int d, dun_stream_t stream);
int dun_fc_bwd_begin(const dun_fc_ctx* c, dun_stream_t stream);
int dun_fc_bwd_block_reduce(const dun_fc_ctx* c, int i, dun_stream_t stream);
int dun_fc_bwd_block_apply(const dun_fc_ctx* c, int i, double global_batch, dun_stream_t stream);
int dun_fc_bwd_block_dx(const dun_fc_ctx* c, int i, double global_batch, dun_stream_t stream);
int dun_fc_bwd_block_dw(const dun_fc_ctx* c, int i, dun_stream_t stream);
int dun_fc_bwd_end(const dun_fc_ctx* c, dun_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * Shared output head as one batched contraction over all depths: act[r,o] = sum_k H[r,k] w[o,k] + b[o],
 * r over D*B rows.  FC: output_layer at every depth (src/DUN/stochastic_fc_models.py:66,69); image DUNs:
 * output_block on the pooled features of every exit (src/DUN/stochastic_img_resnets.py:283-285).
 * Backward: dw[o,k] = sum_r dact[r,o] H[r,k], db[o] = sum_r dact[r,o], dH[r,k] = sum_o dact[r,o] w[o,k]
 * (dH nullable: the FC trunk folds that term into its own backward instead).
 * ---------------------------------------------------------------------------------------------------- */
int dun_head_forward(const float* H, const float* w, const float* b, int rows, int K, int O, float* act,
                     dun_stream_t stream);
size_t dun_head_backward_workspace_floats(int rows, int K, int O);
int dun_head_backward(const float* H, const float* w, const float* dact, int rows, int K, int O, float* dw,
                      float* db, float* dH, float* ws, size_t ws_floats, dun_stream_t stream);

/* ------------------------------------------------------------------------------------------------------
 * torch.optim.SGD(momentum, weight_decay) over one flat parameter buffer, as configured at
 * src/DUN/training_wrappers.py:48-52: g += wd*p; buf = momentum*buf + g (buf = g on the first step, which a
 * zero-initialised buf reproduces); p -= lr*buf.  Any alignment (16-byte aligned buffers take the vector kernel).
 * ---------------------------------------------------------------------------------------------------- */
int dun_sgd_step(float* p, const float* g, float* buf, size_t n, float lr, float momentum, float weight_decay,
                 dun_stream_t stream);
/* The same update for up to DUN_SGD_MAX_SEGMENTS tensors in ONE launch (flat trunk parameters + q_logits + log_std of a
 * small net; the conv parameters of an image DUN in a few launches instead of one per tensor). */
#define DUN_SGD_MAX_SEGMENTS 24
typedef struct dun_sgd_segment {
    float* p;
    const float* g;
    float* buf;
    size_t n;
} dun_sgd_segment;
int dun_sgd_step_multi(const dun_sgd_segment* segments /* host array */, int nseg, float lr, float momentum,
                       float weight_decay, dun_stream_t stream);

/* The guard of nn.CrossEntropyLoss / nn.NLLLoss (src/DUN/training_wrappers.py:29-30, used at :205 and :112), which
 * raise on a label outside [0, C) before anything is updated.  A device step cannot raise in the middle, so:
 * dun_label_count_bad writes (not adds) into count[0] how many of y[0..n) are outside [0, C) and not the ignored -100
 * (one workgroup), and dun_sgd_step_multi_gated is dun_sgd_step_multi that changes NOTHING -- parameters and momentum
 * buffers -- when *skip_if_positive > 0 (a device double, e.g. that count, all-reduced when the batch is sharded).
 * The wrappers raise the reference's IndexError from the same count once the step's one host copy is home. */
int dun_label_count_bad(const int64_t* y, size_t n, int C, double* count, dun_stream_t stream);
int dun_sgd_step_multi_gated(const dun_sgd_segment* segments, int nseg, float lr, float momentum, float weight_decay,
                             const double* skip_if_positive, dun_stream_t stream);

/* Two device-to-device copies in ONE launch (the inputs of a captured training step go into its static buffers:
 * x and y of src/DUN/training_wrappers.py:215; host glue, no arithmetic).  Byte counts; any alignment. */
int dun_copy_pair(void* dst0, const void* src0, size_t bytes0, void* dst1, const void* src1, size_t bytes1,
                  dun_stream_t stream);
/* The same with the two SOURCE addresses read at run time from src_table[0..1] (device or pinned host memory): a captured
 * step keeps this launch inside its graph and the host only rewrites two pointers before each replay. */
int dun_copy_pair_indirect(void* dst0, size_t bytes0, void* dst1, size_t bytes1, const void* const* src_table,
                           dun_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DUN_HIP_H */
