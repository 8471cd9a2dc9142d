/* easyhybrid_hip.h -- C ABI of libeasyhybrid_hip.so: the MI355X (gfx950) engine for the
 * EasyHybrid.jl training-step hot path.
 *
 * The reference (EarthyScience/EasyHybrid.jl, pure Julia) has no FFI; the seam this ABI replaces is
 * the per-minibatch call
 *     Lux.Training.single_train_step!(cfg.autodiff_backend, loss_fn, (x, y), train_state)
 *                                                           src/training/epoch.jl:20-26 (run_epoch!, :13-33)
 * together with the forward/eval calls around it (evaluate_acc, src/training/train.jl:347-355;
 * evaluate_epoch, src/training/epoch.jl:53-66).  A Julia host binds these symbols with @ccall
 * (INTEGRATION.md shows the stub); the test/bench harness binds the same symbols with ctypes.
 *
 * Conventions
 *   - every function returns an eh_status (0 = ok, < 0 = error); nothing throws across the ABI;
 *     eh_last_error(h) (or eh_last_error(NULL) for eh_create failures) gives the message.
 *   - the caller owns every pointer it passes; the library copies during the call and never
 *     retains it.  The library owns all device memory behind the opaque handle.
 *   - a handle is not thread-safe (one driver thread per handle); distinct handles are independent.
 *   - flat parameter vector theta (length eh_n_theta) has the reference's ComponentArray order
 *     (src/models/GenericHybridModel.jl:236-256, src/training/initialization.jl:42-44):
 *       for each Dense layer in chain order: weight, column-major (out,in), then bias (out);
 *       then one raw (pre-sigmoid) value per global parameter, in global_param_names order.
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails with
 *     EH_EHIP.
 */
#ifndef EASYHYBRID_HIP_H
#define EASYHYBRID_HIP_H

#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define EH_ABI_VERSION 4      /* 2: eh_train_step takes the minibatch indices; eh_comm_*.  3: eh_comm_init_local, eh_dp_train_step_group, eh_set_target_loss_program.
                               * 4: eh_p2p_init_local, eh_p2p_check_local, eh_dp_moments, eh_set_target_roles */
#define EH_MAX_HIDDEN 8       /* hidden layers: up to 3 run as ONE fused kernel per step, more (or widths above 128) layer by layer (csrc/eh_lform.hpp) */
#define EH_MAX_PARAMS 8
#define EH_MAX_FORC 4
#define EH_MAX_TARG 4
#define EH_MAX_NETS 8
#define EH_MAX_PROG 64         /* instructions of a mechanistic program (EH_MECH_PROGRAM) */
#define EH_MAX_PROG_CONST 16   /* its literal constants */
#define EH_MAX_PROG_OUT 3      /* its outputs */

typedef enum eh_status {
    EH_OK = 0,
    EH_EINVAL = -1,        /* bad argument / inconsistent descriptor (reference: ArgumentError / AssertionError) */
    EH_EHIP = -2,          /* HIP runtime error or no device */
    EH_ENOMEM = -3,
    EH_EUNSUPPORTED = -4,  /* unknown mechanistic model / activation / shape outside the compiled kernels */
    EH_ESTATE = -5,        /* call order (e.g. train step before eh_opt_init / eh_set_data) */
    EH_ERCCL = -6          /* an RCCL call of the eh_comm_* / eh_dp_allreduce family failed */
} eh_status;

/* activation of the hidden Dense layers (src/models/NNModels.jl:225-230; last layer is linear) */
typedef enum eh_activation { EH_ACT_TANH = 0, EH_ACT_SIGMOID = 1, EH_ACT_RELU = 2, EH_ACT_SWISH = 3, EH_ACT_IDENTITY = 4,
                             EH_ACT_PER_NET = 5 /* n_nets >= 1: net k uses net_activation[k] (activation::NamedTuple,
                                                   GenericHybridModel.jl:168-176); n_nets == 0: hidden layer l of the single network uses
                                                   net_activation[l] (hidden_layers::Chain of Dense layers with activations of their own,
                                                   src/models/NNModels.jl:145-219); such kernels are compiled at run time */ } eh_activation;

/* registry of mechanistic models (the reference takes an arbitrary Julia closure,
 * src/models/GenericHybridModel.jl:425; a closure cannot run in a kernel, so the engine ships
 * hand-derived forward+VJP pairs for the reference's own models).  Canonical parameter /
 * forcing / output order per model:
 *   RBQ10          params (rb, Q10)               forcing (ta)  output (reco)      reco = rb*Q10^(0.1(ta-15))
 *                                                  test/test_split_data_train.jl:36-39, src/models/Respiration_Rb_Q10.jl:39-41
 *   EXPO           params (Resp0, k)              forcing (T)   output (Resp_obs)  Resp0*exp(k*T)
 *                                                  projects/ExpoHybrid/ExpoHybridEstim.jl:69-85
 *   LINEAR         params (alpha, beta)           forcing (x)   output (obs)       alpha*x+beta        src/models/LinearHM.jl:61-68
 *   EXPO2POOL      params (R0a, ka, R0b, kb)      forcing (T)   output (Resp_obs)  R0a*exp(ka*T)+R0b*exp(kb*T)
 *                                                  (build-defined 4-parameter model of BASELINE.json config 3)
 *   RS_COMPONENTS  params (Rb_het,Rb_root,Rb_myc,Q10_het,Q10_root,Q10_myc) forcing (ta) output (R_soil)
 *                                                  src/models/Rs_components.jl:40-57
 *   RS_COMPONENTS3F  the same six parameters, forcings (ta, sw_in, vpd), output (R_soil): R_het + sw_in*R_root + vpd*R_myc,
 *                  R_c = Rb_c*Q10_c^(0.1(ta-15))   (build-defined three-forcing model of BASELINE.json config 5; not in the reference)
 *   FLUXPART       params (RUE, Rb, Q10)          forcings (SW_IN, TA)  outputs (NEE, GPP, RECO)
 *                  GPP = SW_IN*RUE/12.011, RECO = Rb*Q10^(0.1(TA-15)), NEE = RECO - GPP   src/models/FluxPartModel_Q10_Lux.jl:50-79
 */
typedef enum eh_mech { EH_MECH_RBQ10 = 0, EH_MECH_EXPO = 1, EH_MECH_LINEAR = 2, EH_MECH_EXPO2POOL = 3, EH_MECH_RS_COMPONENTS = 4, EH_MECH_FLUXPART = 5,
                       EH_MECH_PROGRAM = 6, EH_MECH_RS_COMPONENTS3F = 7 } eh_mech;

/* EH_MECH_PROGRAM: any other closure `f(; forcing..., params...) -> NamedTuple` of elementwise arithmetic
 * (src/models/GenericHybridModel.jl:420-425), handed over as a straight-line program that the host binding records by
 * calling the closure once with tracer numbers.  The step kernel evaluates it per sample and runs the reverse sweep over the
 * same tape for d loss / d parameter (what Zygote derives from the closure), fp32 like the rest of the path.
 * Value slots: 0..7 the mechanistic parameters in descriptor order (physical values, after sigma-scaling), 8..11 the
 * forcings (canonical order = forcing_index[]), 12..27 prog_const[], 28+i the result of instruction i.  An instruction is
 * op | a << 8 | b << 16 | c << 24 with a, b, c slots defined before it (unused operands 0). */
typedef enum eh_prog_op {
    EH_OP_ADD = 0, EH_OP_SUB = 1, EH_OP_MUL = 2, EH_OP_DIV = 3, EH_OP_NEG = 4,
    EH_OP_EXP = 5, EH_OP_LOG = 6, EH_OP_POW = 7,        /* POW: a^b for a > 0 (exp2(b log2 a)); integer powers are recorded as products */
    EH_OP_SQRT = 8, EH_OP_TANH = 9, EH_OP_SIGMOID = 10,
    EH_OP_MAX = 11, EH_OP_MIN = 12, EH_OP_ABS = 13, EH_OP_SIN = 14, EH_OP_COS = 15,
    EH_OP_SELECT = 16,                                  /* a > 0 ? b : c   (ifelse) */
    EH_OP_GT = 17,                                      /* a > b ? 1 : 0   (no derivative) */
    EH_OP_COUNT = 18
} eh_prog_op;
#define EH_PROG_SLOT_PAR 0
#define EH_PROG_SLOT_FORC 8
#define EH_PROG_SLOT_CONST 12
#define EH_PROG_SLOT_INSTR 28
#define EH_PROG_SLOTS (EH_PROG_SLOT_INSTR + EH_MAX_PROG)

/* where a mechanistic parameter comes from (neural_param_names / global_param_names / the rest,
 * src/models/GenericHybridModel.jl:96-97,127) */
typedef enum eh_param_kind { EH_PAR_NEURAL = 0, EH_PAR_GLOBAL = 1, EH_PAR_FIXED = 2 } eh_param_kind;

typedef enum eh_split { EH_SPLIT_TRAIN = 0, EH_SPLIT_VAL = 1 } eh_split;

/* optimiser rules (Optimisers.jl; reference default Adam(0.01), src/config/TrainingConfig.jl:43) */
typedef enum eh_opt_rule { EH_OPT_ADAM = 0, EH_OPT_ADAMW = 1, EH_OPT_RMSPROP = 2, EH_OPT_DESCENT = 3 } eh_opt_rule;

/* training losses of src/losses/loss_fn.jl:58-174 that the engine can minimise (per target on the valid samples):
 *   MSE mean(r^2) | RMSE sqrt(mean(r^2)) | MAE mean(|r|) | NSELOSS sum(r^2) / sum((y - mean(y))^2)      -- one pass;
 *   PEARSONLOSS 1 - cor | KGELOSS sqrt((r-1)^2 + (alpha-1)^2 + (beta-1)^2) | PBKGELOSS sqrt((r-1)^2 + (beta-1)^2)
 *   -- two passes: d loss / d yhat_i is affine in (yhat_i, y_i) with coefficients made of the batch moments, so a
 *   forward-only pass collects the moments first (no fused_update mode; under data parallelism the moments go round first: eh_dp_moments).
 * Selected with eh_set_option(h, "training_loss", k) (TrainConfig.training_loss, src/config/TrainingConfig.jl:64; default MSE) */
typedef enum eh_loss { EH_LOSS_MSE = 0, EH_LOSS_RMSE = 1, EH_LOSS_MAE = 2, EH_LOSS_NSELOSS = 3,
                       EH_LOSS_PEARSONLOSS = 4, EH_LOSS_KGELOSS = 5, EH_LOSS_PBKGELOSS = 6,
                       EH_LOSS_PROGRAM = 7 /* a recorded custom loss, see eh_set_loss_program */ } eh_loss;

/* buffers a host may address directly on the device (data-parallel all-reduce over RCCL) */
typedef enum eh_buffer { EH_BUF_GRAD = 0, EH_BUF_THETA = 1, EH_BUF_OPT_M = 2, EH_BUF_OPT_V = 3, EH_BUF_GACC = 4, EH_BUF_BNSTAT = 5,
                         EH_BUF_TCOUNT = 6 /* [EH_MAX_TARG][3] per-target sums of the step's shard (eh_dp_counts) */,
                         EH_BUF_MOMENT = 7 /* [EH_MAX_TARG][8] moment sums of (yhat, y) of the step's shard (eh_dp_moments) */ } eh_buffer;

typedef struct eh_model_desc {
    int32_t struct_size;                     /* = sizeof(eh_model_desc) */
    int32_t device;                          /* HIP device ordinal */
    int32_t n_predictors;                    /* P  = length(predictors) */
    int32_t n_hidden;                        /* length(hidden_layers), 1..EH_MAX_HIDDEN */
    int32_t hidden[EH_MAX_HIDDEN];           /* hidden_layers */
    int32_t activation;                      /* eh_activation */
    int32_t scale_nn_outputs;                /* constructHybridModel kwarg */
    int32_t input_batchnorm;                 /* constructHybridModel kwarg: InputBatchNorm(P, affine = false) in front of the chain (src/models/NNModels.jl:89-105,226) */
    int32_t mech;                            /* eh_mech */
    int32_t n_params;                        /* must equal the registry's parameter count */
    int32_t param_kind[EH_MAX_PARAMS];       /* per canonical parameter: eh_param_kind */
    int32_t param_index[EH_MAX_PARAMS];      /* NEURAL: row of the NN output; GLOBAL: position in global_param_names */
    float param_default[EH_MAX_PARAMS];      /* the (default, lower, upper) table, helpers_for_HybridModel.jl:95-102 */
    float param_lower[EH_MAX_PARAMS];
    float param_upper[EH_MAX_PARAMS];
    int32_t n_forcings;                      /* F = number of forcing arrays handed to eh_set_data */
    int32_t forcing_index[EH_MAX_FORC];      /* per canonical forcing of the mech model: which of the F arrays feeds it */
    int32_t n_targets;                       /* T */
    int32_t target_output[EH_MAX_TARG];      /* per target: which output of the mech model it is compared with */
    /* MultiNNHybridModel (src/models/GenericHybridModel.jl:142-206,458-530): n_nets >= 1 single-output MLPs, net k predicting
     * the neural parameter with param_index k from its own predictors.  x passed to eh_set_data is then the per-net predictor
     * matrices stacked row-wise (n_predictors = sum of net_n_predictors); every net has n_hidden hidden layers of
     * net_hidden[k][.] units and the common activation; theta holds the nets one after the other.  The engine runs them as one
     * block-diagonal MLP (the off-diagonal weights are structural zeros).  n_nets = 0: SingleNNHybridModel (hidden[] above). */
    int32_t n_nets;
    int32_t net_n_predictors[EH_MAX_NETS];
    int32_t net_hidden[EH_MAX_NETS][EH_MAX_HIDDEN];
    int32_t net_activation[EH_MAX_NETS];     /* read when activation == EH_ACT_PER_NET: eh_activation (TANH..IDENTITY) of net k, or -- n_nets == 0 --
                                              * of hidden layer l of the single network (EH_MAX_NETS == EH_MAX_HIDDEN) */
    int32_t net_depth[EH_MAX_NETS];          /* hidden layers of net k (hidden_layers::NamedTuple with vectors of different length,
                                              * test/test_generic_hybrid_model.jl:346): 1..n_hidden, 0 = n_hidden; n_hidden is the deepest
                                              * net's.  A shallower net is carried through the remaining layers of the block-diagonal MLP
                                              * by identity blocks (constant weights 1, identity activation, outside theta); such models
                                              * run on kernels compiled at run time, like EH_ACT_PER_NET */
    /* EH_MECH_PROGRAM only (ignored otherwise): n_params parameters, prog_n_forc forcings, prog_n_out outputs */
    int32_t prog_len;                        /* 1..EH_MAX_PROG */
    int32_t prog_n_const;                    /* 0..EH_MAX_PROG_CONST */
    int32_t prog_n_forc;                     /* 0..EH_MAX_FORC canonical forcings the program reads */
    int32_t prog_n_out;                      /* 1..EH_MAX_PROG_OUT */
    int32_t prog_out[EH_MAX_PROG_OUT];       /* slot holding output o */
    uint32_t prog_code[EH_MAX_PROG];
    float prog_const[EH_MAX_PROG_CONST];
} eh_model_desc;

typedef struct eh_target_metrics {           /* src/losses/loss_fn.jl:58-179 on the valid samples of one target */
    double n;                                /* number of valid (non-NaN) targets */
    double mse, rmse, mae, r2, nse, pearson, kge, pbkge, beta, alpha;
    double sse;                              /* sum of squared residuals */
} eh_target_metrics;

typedef struct eh_handle_s eh_handle;

int32_t eh_version(void);
const char* eh_last_error(const eh_handle* h);

/* constructHybridModel(...) -> device-side model + workspace (src/models/GenericHybridModel.jl:89-140) */
int32_t eh_create(const eh_model_desc* desc, eh_handle** out);
int32_t eh_destroy(eh_handle* h);
int32_t eh_n_theta(const eh_handle* h, int64_t* n);

/* run every later call on this hipStream_t (NULL = the handle's own stream) */
int32_t eh_set_stream(eh_handle* h, void* hip_stream);
int32_t eh_synchronize(eh_handle* h);

/* dataset of one split, made resident in HBM once (replaces the per-batch host->device copies of
 * collect_dim_data, src/training/epoch.jl:1-11).  x: (P x N) column-major exactly as the reference
 * holds it (src/data/prepare_data.jl:6); forcings: F arrays of N; targets: T arrays of N with NaN =
 * missing (valid_mask, src/training/train.jl:221-232).  on_device: flags -- EH_DATA_ON_DEVICE (1): the pointers are device
 * pointers on the handle's device; EH_DATA_X_PLANES (2): x is given as P arrays of N (a ROW-major P x N matrix, what a NumPy host
 * holds: no transposed copy on the caller's side). */
#define EH_DATA_ON_DEVICE 1
#define EH_DATA_X_PLANES 2
#define EH_DATA_X_ROWS 4      /* host arrays only: x is really `const float* const*`, P pointers to arrays of N -- the caller's own predictor columns
                               * (a DataFrame's), interleaved into the records without ever being stacked into a matrix on the host */
int32_t eh_set_data(eh_handle* h, int32_t split, int64_t n, const float* x, const float* const* forcings,
                    const float* const* targets, int32_t on_device);

int32_t eh_set_params(eh_handle* h, const float* theta, int64_t n);
int32_t eh_get_params(eh_handle* h, float* theta, int64_t n);

/* model forward on samples [first, first+count) of a split (GenericHybridModel.jl:370-431, test mode).
 * yhat: T host arrays of count floats (or NULL); params: n_params host arrays of count floats (or NULL). */
int32_t eh_forward(eh_handle* h, int32_t split, int64_t first, int64_t count, float* const* yhat, float* const* params);

/* The mechanistic stage on its own, for a neural network that lives OUTSIDE this library (a Lux / Flux chain on the caller's GPU):
 * o = its outputs for `count` samples -> physical parameters (sigmoid scaling when scale_nn_outputs; GenericHybridModel.jl:404-411)
 * -> mechanistic model -> masked MSE summed over the targets (compute_loss.jl:50-53, loss_fn.jl:61-63) -> d loss / d o, the
 * gradient of the raw global parameters and, optionally, the predictions.  What `Zygote.pullback` hands back at the NN boundary
 * of GenericHybridModel.jl:389-425.  ALL array arguments are DEVICE pointers: o_dev / d_o_dev are [K][ld] (row k = neural
 * parameter k, ld >= count floats apart), forcings_dev / targets_dev the F / T arrays in eh_set_data order (NaN target =
 * missing), yhat_dev [T][ld] or NULL.  Global / fixed parameters are those of the handle (eh_set_params).  n_valid_in: valid
 * samples per target if the caller knows them (masks belong to the data set, train.jl:221-232), NULL = counted here by a pass
 * over the targets.  loss / grad_global (G floats, global_param_names order) / n_valid are HOST outputs; pass all three NULL to
 * leave the call asynchronous on the handle's stream.  A pure streaming kernel: 4 (K + F + T) bytes read and 4 K written per
 * sample (16-byte accesses when count, ld and the pointers allow; a recorded closure, EH_MECH_PROGRAM, is interpreted one sample
 * per lane).  The training loss is the handle's (eh_set_option "training_loss"): mse or mae; the others need batch statistics
 * before the pass and return EH_EUNSUPPORTED. */
int32_t eh_mech_loss_vjp(eh_handle* h, int64_t count, int64_t ld, const float* o_dev, const float* const* forcings_dev,
                         const float* const* targets_dev, const int64_t* n_valid_in, float* d_o_dev, float* yhat_dev,
                         float* loss, float* grad_global, int64_t* n_valid);

/* compute_loss(train_mode) value and its gradient wrt flat theta on one minibatch, no update
 * (the objective Zygote differentiates, src/training/epoch.jl:40-51; also the seam
 * train_optimization.jl:121-133 would use).  idx: optional count sample indices (host, int32) into the
 * split, NULL = the contiguous window.  An all-masked batch gives loss = NaN, grad = 0, n_valid = 0. */
int32_t eh_loss_and_grad(eh_handle* h, int32_t split, const int32_t* idx, int64_t first, int64_t count,
                         float* loss, float* grad, int64_t* n_valid);

/* running statistics of the input BatchNorm layer (the model state `st.st_nn`; Lux starts them at mean 0, var 1).
 * Training steps use the statistics of their own minibatch and update these with momentum 0.1; forward / eval use them. */
int32_t eh_get_bn_state(eh_handle* h, float* running_mean, float* running_var, int64_t n_predictors);
int32_t eh_set_bn_state(eh_handle* h, const float* running_mean, const float* running_var, int64_t n_predictors);

/* Optimisers.setup(rule, ps): zero moments, t = 0 (src/training/initialization.jl:42-44) */
int32_t eh_opt_init(eh_handle* h, int32_t rule, float lr, float beta1, float beta2, float eps, float weight_decay);
int32_t eh_get_opt_state(eh_handle* h, float* m, float* v, int64_t n, float* beta_t /* [2] */);
int32_t eh_set_opt_state(eh_handle* h, const float* m, const float* v, int64_t n, const float* beta_t);

/* one single_train_step! (src/training/epoch.jl:20-26): fused forward + mechanistic model + masked loss + VJP, then reduce +
 * optimiser update, on one minibatch of the resident train split.
 *   idx == NULL : the contiguous window [first, first+count)
 *   idx != NULL : the minibatch the CALLER's loader drew (MLUtils.DataLoader(shuffle = true), src/data/loaders.jl:1-12): sample
 *                 i of the batch is train sample idx[first + i], int32, 0-based.  idx_on_device == 0: a host array (range-checked,
 *                 copied on the handle's stream); != 0: a device array the caller keeps alive until the step has run (e.g. a
 *                 whole epoch's permutation uploaded once; not range-checked).
 * loss_out == NULL: fully asynchronous on the stream.  An all-masked batch is skipped without touching theta or the optimiser
 * state (src/training/epoch.jl:17-19). */
int32_t eh_train_step(eh_handle* h, const int32_t* idx, int32_t idx_on_device, int64_t first, int64_t count, float* loss_out);

/* one run_epoch! over the train split (src/training/epoch.jl:13-33): ceil(N/batchsize) steps, the
 * last one partial (MLUtils.DataLoader default), shuffled on the device when shuffle != 0
 * (src/data/loaders.jl:6; the permutation is the engine's own keyed bijection, not Julia's RNG
 * stream).  mean_loss: mean of the per-step losses (nullable); n_steps: steps run (nullable). */
int32_t eh_train_epoch(eh_handle* h, int64_t batchsize, uint64_t seed, int32_t shuffle, float* mean_loss, int64_t* n_steps);

/* extra_loss = (yhat, ps) -> (; l2 = lambda * weight_l2(ps; normalize)) (src/utils/extract_weights.jl:69-91,
 * src/losses/compute_loss.jl:31-34 with agg = sum): lambda times the sum -- or, normalize != 0, the mean -- of the squared
 * Dense WEIGHTS (biases and global parameters excluded) is added to the training loss and its gradient 2 lambda w to the
 * weight gradients.  lambda = 0 switches it off.  Two-kernel path only (no fused_update, no data-parallel seam).
 * Other extra_loss closures cannot run on the device. */
int32_t eh_set_weight_l2(eh_handle* h, float lambda, int32_t normalize);

/* The same extra loss with one coefficient per canonical parameter entry: extra loss = sum_i coef[i] * theta_i^2, gradient
 * 2 coef[i] theta_i.  This is what several weight_l2 terms come to -- one per network of a MultiNNHybridModel with its own
 * lambda and normalisation (the reference's own example, src/utils/extract_weights.jl:64:
 * `extra_loss = (yhat, ps) -> (; l2_Rb = lambda * weight_l2(ps.Rb; normalize = true),)`), or `key = :bias` -- the host side
 * adds lambda (or lambda / #entries of the term) at the entries each term covers.  coef: host array of n = n_theta floats,
 * >= 0; NULL or all zero switches the extra loss off; replaces whatever eh_set_weight_l2 had set (and vice versa).
 * (ABI 3, added in the same round as the other version-3 entry points.) */
int32_t eh_set_weight_l2_coef(eh_handle* h, const float* coef, int64_t n);

/* hipGraph capture of a sequence of steps: between eh_graph_begin and eh_graph_end the calls
 * eh_train_step(..., loss_out = NULL) / eh_dp_fused_step are recorded, not run; eh_graph_launch replays the
 * recorded sequence (same windows, same buffers).  The calls advance the engine's rotation state
 * (accumulator slot, parameter-set ping-pong) as if they had run, and a graph is only valid in the
 * state it was recorded in: record a multiple of 6 steps in fused_update mode (of 2 otherwise) so that
 * it can be replayed back to back, and in fused_update mode begin (and launch) with an update pending,
 * i.e. right after a step and without an eh_synchronize in between.  Violations are refused.
 * (Measured: no gain on the RbQ10 step -- the gap between two dependent kernels is on the GPU side.) */
int32_t eh_graph_begin(eh_handle* h);
int32_t eh_graph_end(eh_handle* h, int32_t* graph_id);
int32_t eh_graph_launch(eh_handle* h, int32_t graph_id);

/* evaluate_acc on samples [first, first+count) of a split: metrics per target, optionally the
 * predictions and physical parameters (host arrays as in eh_forward). */
int32_t eh_eval(eh_handle* h, int32_t split, int64_t first, int64_t count, eh_target_metrics* out,
                float* const* yhat, float* const* params);

/* ---- data-parallel seam: one process per GPU, the host all-reduces EH_BUF_GRAD over RCCL ----------
 * eh_dp_grad   : local partial sums of the UN-normalised gradient, loss and valid counts into
 *                EH_BUF_GRAD  = [ grad (n_theta) | sum m (yhat-y)^2 | n_valid per target (T) | sum (y-c) | sum (y-c)^2 ]
 * (host: all_reduce(SUM) over that buffer)
 * eh_dp_apply  : normalise by the global counts and apply the optimiser update (replicated).
 * The mean over the GLOBAL valid count is what the reference computes (src/losses/loss_fn.jl:61-63),
 * so shards exchange sums and counts, never per-shard means.
 * Multi-target models (T > 1): every target has its own normaliser, which must be known inside the pass, so a step
 * starts with  eh_dp_counts(h, first, count)  -> EH_BUF_TCOUNT = [ n_t | sum (y-c) | sum (y-c)^2 ] per target (12 floats),
 * (host: all_reduce(SUM) over that buffer), then eh_dp_grad (weights 1/n_t, or 1/sum (y-ybar)^2 for nseLoss, of the GLOBAL
 * batch; EH_BUF_GRAD then holds final sums), all-reduce, eh_dp_apply.  c is the split's target shift: the ranks must share
 * it (eh_set_target_shift, e.g. the global mean of each target); eh_set_data resets it to the shard's own means. */
int32_t eh_dp_grad(eh_handle* h, int64_t first, int64_t count);
int32_t eh_dp_counts(eh_handle* h, int64_t first, int64_t count);
/* Two-pass training losses (pearsonLoss / kgeLoss / pbkgeLoss, src/losses/loss_fn.jl:105-174; rmse on a multi-target model, :58-60):
 * d loss / d yhat_i is affine in (yhat_i, y_i) with coefficients made of the moments of the GLOBAL batch, so a step starts with
 *   eh_dp_moments(h, first, count, 0) -> this shard's sums about the common target shift in EH_BUF_MOMENT (host: all_reduce(SUM)),
 *   eh_dp_moments(h, first, count, 1) -> the same about the global mean of yhat (host: all_reduce(SUM) again),
 * then [eh_dp_counts + all-reduce for a multi-target model,] eh_dp_grad (which turns the all-reduced moments into the coefficients; the
 * sums in EH_BUF_GRAD are then final), all-reduce, eh_dp_apply.  The shift must be common to the ranks (eh_set_target_shift).  Not
 * for the layer-wise form (EH_EUNSUPPORTED).  (ABI 4) */
int32_t eh_dp_moments(eh_handle* h, int64_t first, int64_t count, int32_t stage);
int32_t eh_set_target_shift(eh_handle* h, int32_t split, const float* shift, int64_t n);
int32_t eh_dp_apply(eh_handle* h, float* loss_out);
/* per-shard epoch shuffle for the data-parallel calls (the reference shuffles the whole training set,
 * src/data/loaders.jl:6; here every rank permutes its own shard): on != 0 draws a new keyed
 * permutation of the train split -- the same generator eh_train_epoch uses -- and the windows
 * [first, first+count) of eh_dp_grad / eh_dp_fused_step / eh_dp_bn_stats then index THROUGH it;
 * on == 0 goes back to contiguous windows.  eh_set_data and a shuffled eh_train_epoch drop it. */
int32_t eh_dp_shuffle(eh_handle* h, uint64_t seed, int32_t on);

/* Input BatchNorm under data parallelism (Lux BatchNorm normalises with the statistics of the WHOLE
 * minibatch, src/models/NNModels.jl:97-105): before eh_dp_grad / eh_dp_fused_step of a step,
 *   eh_dp_bn_stats : this shard's sums into EH_BUF_BNSTAT = [ sum (x-c) (32) | sum (x-c)^2 (32) | n ]  (65 floats)
 *   (host: all_reduce(SUM) over that buffer)
 * and the step kernel normalises with the global mean / variance and updates the (replicated)
 * running statistics.  c is a per-predictor shift every rank must share (eh_set_bn_shift, e.g. the
 * mean of the whole training set); it only guards the variance against cancellation. */
int32_t eh_set_bn_shift(eh_handle* h, const float* shift, int64_t n);
int32_t eh_dp_bn_stats(eh_handle* h, int64_t first, int64_t count);
int32_t eh_device_buffer(eh_handle* h, int32_t which, void** dev_ptr, int64_t* n_floats);

/* The collective inside the library (RCCL over xGMI): a host that has no collective library of its own --
 * the Julia shim, or ONE process driving all GPUs of a node with one handle per device -- gets the all-reduce of the seam
 * above from the library itself, in stream order on the handle's stream:
 *   eh_comm_unique_id : rank 0 draws the 128-byte id of a new communicator; the host hands it to every rank (any channel)
 *   eh_comm_init      : ncclCommInitRank on the handle's device; world = number of handles in the job, rank = this one's
 *   eh_dp_allreduce   : SUM all-reduce, in place, of EH_BUF_GRAD (after eh_dp_grad), of the third `index` of EH_BUF_GACC
 *                       (*buffer_index of eh_dp_fused_step) or of EH_BUF_BNSTAT (after eh_dp_bn_stats)
 *   eh_dp_train_step  : the whole data-parallel step of one rank: [eh_dp_bn_stats + all-reduce] + eh_dp_grad + all-reduce +
 *                       eh_dp_apply, or eh_dp_fused_step + all-reduce in fused_update mode (loss_out then must be NULL)
 * A single thread that drives several handles must bracket the calls of one step over all its handles with
 * eh_comm_group_begin / eh_comm_group_end (ncclGroupStart / ncclGroupEnd), as RCCL requires; eh_comm_init likewise.
 *
 * ONE process, one host thread, one handle per device (threading row) needs no RCCL at all:
 *   eh_comm_init_local     : the n handles (all of this process; rank = position in the list) become a LOCAL group.  Their
 *                            eh_dp_allreduce calls -- every member's, inside one eh_comm_group_begin / _end bracket -- meet at
 *                            eh_comm_group_end: each member's stream waits (events) for all members' producers, one small kernel
 *                            per member reads ALL members' buffers (peer-mapped device memory over xGMI when they sit on
 *                            different GPUs; EH_EUNSUPPORTED if a pair of devices has no peer access) and adds them in rank
 *                            order -- bit-identical sums on every replica -- and a second event round guards the write-back.
 *                            Nothing blocks the host.  eh_comm_destroy on any member dissolves the group.
 *   eh_dp_train_step_group : a whole data-parallel step of n handles driven by one thread (local group or RCCL communicators):
 *                            every phase (BatchNorm sums, per-target counts, gradient pass, update) is issued to all handles
 *                            before the exchange that follows it; first[i] = handle i's window start in its own shard.
 *                            loss_out (two-kernel mode only) receives the global batch loss. */
#define EH_COMM_ID_BYTES 128
int32_t eh_comm_unique_id(void* id_out, int64_t id_bytes);
int32_t eh_comm_init(eh_handle* h, const void* unique_id, int64_t id_bytes, int32_t world, int32_t rank);
int32_t eh_comm_init_local(eh_handle* const* handles, int32_t n);
int32_t eh_comm_destroy(eh_handle* h);
int32_t eh_comm_group_begin(void);
int32_t eh_comm_group_end(void);
int32_t eh_dp_allreduce(eh_handle* h, int32_t which, int32_t index);
int32_t eh_dp_train_step(eh_handle* h, int64_t first, int64_t count, float* loss_out);
int32_t eh_dp_train_step_group(eh_handle* const* handles, int32_t n, const int64_t* first, int64_t count, float* loss_out);

/* One-kernel-per-step variant of the data-parallel seam (needs eh_set_option("fused_update", 1)):
 * eh_dp_fused_step launches the fused step kernel, whose prologue applies the (already all-reduced)
 * accumulator of the previous step and whose epilogue adds this step's raw partial sums into one
 * of three rotating accumulators inside EH_BUF_GACC ([3][8 shards][n_theta+1+T] floats);
 * *buffer_index says which third the host must all_reduce(SUM) next.  eh_synchronize applies the
 * last pending update. */
int32_t eh_dp_fused_step(eh_handle* h, int64_t first, int64_t count, int32_t* buffer_index);

/* Cross-GPU exchange without a collective call (2..8 ranks of one node, fused_update mode): every rank
 * owns an uncached receive buffer of {value, sequence number} words that all peers map over HIP IPC;
 * the last workgroup of a step stores the rank's sums into every peer's buffer (one 8-byte xGMI
 * peer-to-peer store per element, carrying its own arrival stamp), the next step's prologue reads the
 * shards of all ranks and re-reads what has not arrived yet (2 s deadline -> error at eh_synchronize,
 * never a hang).  eh_dp_fused_step then needs NO host all-reduce (*buffer_index = -1).
 *   eh_p2p_init     : allocate + export; writes the 64-byte IPC handle of this rank (world = 1: loopback)
 *   (host: all-gather the handles of all ranks, e.g. over torch.distributed)
 *   eh_p2p_attach   : map every peer; handles = world consecutive handles, `handle_stride` bytes apart
 *   eh_p2p_selftest : `rounds` exchanges of known vectors, all ranks together; *ok = 0 -> call
 *                     eh_p2p_disable on every rank and keep all-reducing EH_BUF_GACC instead */
int32_t eh_p2p_init(eh_handle* h, int32_t world, int32_t rank, void* handle_out, int64_t handle_bytes);
int32_t eh_p2p_attach(eh_handle* h, const void* handles, int64_t handle_stride);
int32_t eh_p2p_selftest(eh_handle* h, int32_t rounds, int32_t* ok);
int32_t eh_p2p_disable(eh_handle* h);
/* The same exchange between the handles of ONE process (one host thread, one handle per device -- the Julia host; SURVEY section
 * 8(b), threading row): no IPC, every member's kernels hold plain device pointers to the others' receive buffers (peer access is
 * enabled between distinct devices; EH_EUNSUPPORTED where a pair has none).  All members need fused_update on and a communicator
 * for the fall-back (eh_comm_init_local, or eh_comm_init); steps then go through eh_dp_train_step_group, which issues no
 * exchange at all while the group is healthy.
 *   eh_p2p_init_local  : allocate + wire all n members (rank = position) and run the start-up self-test on all of them
 *                        together; *ok = 0 -> the test failed, everything is released again and the members keep all-reducing
 *   eh_p2p_check_local : drain every member and read the deadline flags; *healthy = 0 -> an exchange was missed (a member
 *                        stepped alone, a device stalled > 2 s): every member has left the peer-to-peer exchange and taken
 *                        member 0's parameters and optimiser state, so the replicas are identical again
 * eh_p2p_disable on any member dissolves the whole local peer-to-peer group. */
int32_t eh_p2p_init_local(eh_handle* const* handles, int32_t n, int32_t selftest_rounds, int32_t* ok);
int32_t eh_p2p_check_local(eh_handle* const* handles, int32_t n, int32_t* healthy);

/* timing aid for bench.py: when enabled, eh_train_step brackets the fused step kernel with HIP
 * events on its stream; eh_profile_read returns the number of launches and their mean duration. */
int32_t eh_profile_enable(eh_handle* h, int32_t on);
int32_t eh_profile_read(eh_handle* h, int64_t* n_launches, double* mean_ms_step_kernel, double* mean_ms_reduce_kernel);
/* on = S > 1 brackets bursts of S consecutive steps with ONE event pair (an event between two
 * back-to-back kernels costs about as much as a small kernel; a burst measures the steady-state rate).
 * eh_profile_samples copies out the duration in ms of every recorded step (S = 1) or burst (S > 1)
 * without consuming them; call it before eh_profile_read. */
int32_t eh_profile_samples(eh_handle* h, double* ms, int64_t cap, int64_t* n_out);

/* diagnostic builds (make STAMPS=1) only: in-kernel phase stamps of workgroup 0 as (shader clock, 100 MHz clock)
 * pairs; the first call arms the buffer.  A normal build leaves the buffer zero. */
int32_t eh_debug_stamps(eh_handle* h, uint64_t* out, int32_t n);

/* tuning knobs (name/value): "max_blocks" (1..256), "variant" (tile shape), "fast_paths" (0 = generic MFMA kernels), "training_loss" (eh_loss),
 * "fused_update" (1 = one kernel per step, float-atomic accumulation: not bitwise reproducible; 2 = one kernel per step only where that IS
 * reproducible -- minibatches one workgroup covers -- and the deterministic step + reduce pair for larger ones: what a seeded run asks for), "row_split" (kernel family),
 * "jit" (EH_MECH_PROGRAM: 1 = step kernels compiled at run time around the recorded closure (default; also env EH_JIT),
 * 0 = the interpreting kernels built ahead of time),
 * "specialize" (1 = every model's step kernels compiled at run time (hiprtc, ~1 s, cached on disk) with the descriptor as a compile-time
 * constant; 2 = the same in a background thread -- steps run the kernels built ahead of time until the compiled one is ready),
 * "check_idx" (debug: 1 = the minibatch indices eh_train_step is given ON THE DEVICE are range-checked before every step -- a small kernel and
 * one synchronisation per step; an index outside the split is EH_EINVAL instead of a wild read.  Host indices are always checked),
 * "aot_spec" (default 1: a handle whose descriptor is a canonical one -- the BASELINE.json configurations -- runs the kernel specialised for it
 * at BUILD time, csrc/eh_spec.hip, whatever "specialize" says; 0 = never: tests and A/B runs of the other kernels),
 * "precision" (0 = fp32 end to end, the reference's arithmetic; 1 = bf16 forward products with fp32 accumulation and an fp32-exact
 * backward pass, BASELINE.json config 5 -- row-split shapes with tanh / sigmoid / relu / identity only; 2 = bf16 operands in BOTH passes: every
 * backward delta is rounded to bfloat16 once, in the scale the step carries it, fp32 accumulation),
 * "multi_step" (1 = default: eh_train_epoch in fused_update mode runs minibatches that one workgroup covers -- e.g. the reference's default
 * batch of 64 -- several steps per kernel launch, the state between two steps in LDS; 0 = one launch per step),
 * "bn_in_kernel" (1 = default: with input BatchNorm, minibatches of up to 512 samples get their batch statistics inside the per-wave step
 * kernel instead of from a launch in front of it; 0 = always the separate launch),
 * "eval_blocks" (workgroups of eh_eval / eh_forward; 0 = default: the per-wave kernels, which need far fewer registers without gradient
 * accumulators, run up to four workgroups per CU there),
 * "agg" (`agg::Function` of the training configuration, src/config/TrainingConfig.jl:76-77: the training loss is agg(per-target losses),
 * src/losses/compute_loss.jl:50-53, and with an extra loss agg([that, extra entries...]), :31-34 -- 0 = sum (default), 1 = mean) with
 * "extra_terms" (the number of entries the extra loss returns -- what eh_set_weight_l2 / eh_set_weight_l2_coef stand for; only mean needs it) */
int32_t eh_set_option(eh_handle* h, const char* name, int64_t value);

/* A custom training loss `training_loss::Function` (src/losses/loss_fn.jl: called as f(yhat[mask], y[mask])) of the form
 * mean_i l(yhat_i, y_i): the per-sample term l recorded as a program like a mechanistic closure (eh_prog_op; value slot 0 =
 * yhat, slot 1 = y, 12.. constants, 28+i instruction i; out_slot = the slot holding l).  The step kernel evaluates l and
 * d l / d yhat per valid sample; the mean over the valid samples and everything downstream is the MSE path.  Select it with
 * eh_set_option(h, "training_loss", EH_LOSS_PROGRAM).  It exists only in kernels compiled at run time (hiprtc): without them
 * the training calls fail with EH_EUNSUPPORTED -- there is no interpreted or CPU form. */
int32_t eh_set_loss_program(eh_handle* h, const uint32_t* code, int32_t n_instr, const float* consts, int32_t n_const, int32_t out_slot);
/* the same for ONE target of a multi-target model (PerTarget((f, g)) with different functions, src/losses/compute_loss.jl:128-145;
 * the reference's own test: test/test_compute_loss.jl:56-62): target t then uses this program where its kind is EH_LOSS_PROGRAM,
 * the others the common one of eh_set_loss_program (which also clears the per-target ones). */
int32_t eh_set_target_loss_program(eh_handle* h, int32_t target, const uint32_t* code, int32_t n_instr, const float* consts, int32_t n_const, int32_t out_slot);

/* loss_spec = PerTarget((l_1, ..., l_T)) (src/losses/compute_loss.jl:128-145): target t is trained on its own loss, the total is
 * their sum (agg = sum).  Built per target: EH_LOSS_MSE, EH_LOSS_MAE, EH_LOSS_NSELOSS -- the losses whose per-sample weight is a
 * function of the targets alone (1 / n_t, 1 / n_t, 1 / sum (y - mean y)^2), found by a pre-pass; n must equal n_targets.  The same
 * three are what eh_set_option("training_loss") accepts on a multi-target model. */
int32_t eh_set_target_losses(eh_handle* h, const int32_t* kinds, int32_t n);

/* extra_loss as a function of the PREDICTIONS (src/losses/compute_loss.jl:31-34; the reference's own test:
 * `extra_loss = (yhat, ps) -> [sum(abs, yhat.var1), sum(abs, yhat.var2)]`, test/test_compute_loss.jl:257-285).  An entry that is the sum or
 * the mean over ALL samples of the batch of a per-sample function f(yhat_o) of one model output is carried as one more TARGET of the
 * descriptor: it observes output o (target_output), its data column holds no NaN (the host binding passes zeros), its per-sample loss is
 * the recorded f (eh_set_target_loss_program; kind EH_LOSS_PROGRAM in eh_set_target_losses) -- and this call says which targets are such
 * entries: roles[t] = 0 a data target | 1 an extra-loss entry, mean over all samples | 2 an extra-loss entry, sum over all samples.
 * An entry takes the extra loss's factor under agg = mean (count it in "extra_terms"), no 1 / n when it is a sum, and is no data target.
 * n must equal n_targets; at least one data target.  (ABI 4) */
int32_t eh_set_target_roles(eh_handle* h, const int32_t* roles, int32_t n);

/* Which step kernels the handle runs.  *n_compiled = kernel pairs (train + eval) specialised for this handle's descriptor and in use:
 * compiled with hiprtc at run time (recorded closures, the "specialize" option), or -- log starts with "ahead-of-time:" -- built into the
 * library for a canonical descriptor (the BASELINE.json configurations, csrc/eh_spec.hip).  0 with a non-empty log = a build or a launch
 * was refused, or a compiled kernel disagreed with the one built ahead of time on the first batch (every kernel that merely replaces
 * a generic one is checked before it takes over), and the handle runs the generic / interpreting kernels instead (same results,
 * slower).  log (optional) receives the message, NUL-terminated. */
int32_t eh_jit_status(eh_handle* h, int32_t* n_compiled, char* log, int64_t log_bytes);

#ifdef __cplusplus
}
#endif
#endif
