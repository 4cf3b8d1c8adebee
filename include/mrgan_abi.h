/* libmrgan_hip -- C ABI of the MI355X-native mr_gan training path.
 *
 * Drop-in boundary: the three Theano-compiled callables of the reference,
 *     train_batch_disc([phase, x_lab, labels, x_unl, noise]) -> [loss_lab, loss_unl, train_err]   (mr_gan.py:169)
 *     train_batch_gen ([phase, x_unl, noise])               -> loss_gen                           (mr_gan.py:170)
 *     test_batch      ([phase, x_lab, labels])              -> test_err                           (mr_gan.py:171)
 * plus the state they close over (Keras shared variables: weights of mr_gan.py:110-128, the two Adam update
 * lists and their single iteration counter, mr_gan.py:165-167).
 *
 * Conventions: every function returns 0 on success and a negative code on failure; the message is available
 * from mrgan_last_error() (thread-local).  Nothing throws across the ABI.  All pointers named *_dev are device
 * pointers (HBM); the caller owns inputs and outputs, the handle owns weights, optimiser state and workspace.
 * Calls are asynchronous on `stream` unless a host output pointer is passed (then the call synchronises the
 * stream before returning).  One handle per device; a handle is not thread-safe.  Training state lives in the handle only
 * (tuning and the diagnostic switches of mrgan_debug.h are per handle; nothing reads the environment); the library keeps two
 * process-wide caches, both keyed by device id and safe to share between threads: which kernels have had their LDS limit
 * raised on a device, and the mel filter banks of mrgan_logmel per (device, sr, n_mels).
 * No torch types appear here: the Python host passes tensor.data_ptr() and the raw hipStream_t.
 */
#ifndef MRGAN_ABI_H
#define MRGAN_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mrgan_handle mrgan_handle;
typedef void* mrgan_stream;               /* hipStream_t */

/* arithmetic of the dense stacks: fp32 MFMA (parity) / bf16 MFMA (speed) / fp8: the discriminator's dense products on
 * the fp8 matrix cores (e4m3 activations and weights, e5m2 gradients, fp32 accumulate, delayed per-tensor power-of-two
 * scales; the generator, the loss head and evaluation stay bf16; every padded width becomes a multiple of 128) */
enum { MRGAN_F32 = 0, MRGAN_BF16 = 1, MRGAN_FP8 = 2 };
enum { MRGAN_NET_G = 0, MRGAN_NET_D = 1 };
enum {
    MRGAN_FLAG_SYNC_STATS = 1,   /* batch statistics (BN, feature-matching moments) are exchanged between phases   */
    MRGAN_FLAG_FLAT_GRADS = 2,   /* gradients are reduced into the flat buffers; Adam runs as its own phase        */
    MRGAN_FLAG_GRAPH      = 4,   /* stream-mode arguments only: mrgan_train_pair replays a captured hipGraph of the whole
                                  * pair; with MRGAN_FLAG_FLAT_GRADS every phase range [first, last] passed to
                                  * mrgan_disc_step / mrgan_gen_step is captured once and replayed (the kernels between two
                                  * collectives of a data-parallel host)                                                */
    MRGAN_FLAG_GRAD_BF16  = 8    /* with MRGAN_FLAG_FLAT_GRADS: the gradients travel as bfloat16.  The reduce phase writes
                                  * MRGAN_REGION_GRAD_*_BF16 (rounded once from the fp32 sums) instead of the fp32 flat
                                  * buffers and the Adam phase reads them back: the host all-reduces those regions in place,
                                  * nothing is cast or allocated per step.  The four fp32 scalars behind the fp32 buffers
                                  * (MRGAN_REGION_TAIL_*) still travel in fp32.  A labelled, different numerical path.      */
};

/* Hyper-parameters are literals inside mr_gan() in the reference (mr_gan.py:77-79, :111-128, :165);
 * mrgan_default_config() fills exactly those values. */
typedef struct mrgan_config {
    int32_t d_in;             /* D = X_train.shape[1]                                  (mr_gan.py:114, :117)  */
    int32_t batch;            /* rows per sub-batch on THIS rank; reference batchSize = 50      (mr_gan.py:78) */
    int32_t noise_size;       /* 100                                                            (mr_gan.py:77) */
    int32_t g_hidden[2];      /* 500, 500                                                  (mr_gan.py:111-113) */
    int32_t d_hidden[5];      /* 1000, 500, 250, 250, 250                                  (mr_gan.py:119-127) */
    int32_t num_classes;      /* 6 materials; the fake class is the implicit zero logit        (mr_gan.py:128) */
    int32_t dtype;            /* MRGAN_F32 | MRGAN_BF16 | MRGAN_FP8 */
    float sigma[5];           /* GaussianNoise std before discriminator dense 1..5: .3 .5 .5 .5 .5 (:118-126)  */
    float lr, beta1, beta2, adam_eps;      /* Adam(lr=0.0006, beta_1=0.5), Keras defaults         (mr_gan.py:165) */
    float bn_eps;             /* 2e-5                                                          (mr_gan.py:112) */
    float unlabeled_weight;   /* 1                                                              (mr_gan.py:79) */
    uint64_t seed;            /* key of the device Philox streams (layer noise, z) */
    int32_t rank, world;      /* data-parallel position: global batch = batch*world, noise rows offset by rank*batch */
    int32_t flags;
    int32_t reserved;
} mrgan_config;

int mrgan_default_config(mrgan_config* cfg, int32_t d_in, int32_t batch);

/* Workspace: all device memory of a handle is one block.  Pass workspace_dev = NULL to let the library
 * hipMalloc it, or allocate `bytes` yourself (e.g. a torch uint8 tensor) to be able to alias sub-regions. */
int mrgan_workspace_bytes(const mrgan_config* cfg, size_t* bytes);
int mrgan_create(const mrgan_config* cfg, void* workspace_dev, size_t bytes, mrgan_stream stream, mrgan_handle** out);
int mrgan_destroy(mrgan_handle* h);
const char* mrgan_last_error(void);

/* Weights in Keras order.  G: W1 b1 gamma beta W2 b2 W3 b3 (generator.trainable_weights, mr_gan.py:131);
 * D: (W,b) x 6 (discriminator.trainable_weights, mr_gan.py:132).  Dense fp32 [rows, cols] row-major. */
int mrgan_num_tensors(const mrgan_handle* h, int net, int* n);
int mrgan_tensor_shape(const mrgan_handle* h, int net, int idx, int* rows, int* cols);
int mrgan_set_weights(mrgan_handle* h, int net, int idx, const float* src_dev, mrgan_stream stream);
int mrgan_get_weights(mrgan_handle* h, int net, int idx, float* dst_dev, mrgan_stream stream);
/* which: 0 = Adam m, 1 = Adam v, 2 = last flat gradient (only meaningful with MRGAN_FLAG_FLAT_GRADS) */
int mrgan_get_slot(mrgan_handle* h, int net, int idx, int which, float* dst_dev, mrgan_stream stream);
int mrgan_set_slot(mrgan_handle* h, int net, int idx, int which, const float* src_dev, mrgan_stream stream);
int mrgan_get_iterations(mrgan_handle* h, mrgan_stream stream, uint32_t* iterations_host);
int mrgan_set_iterations(mrgan_handle* h, uint32_t iterations, uint32_t batch_counter, mrgan_stream stream);

/* train_batch_disc (mr_gan.py:169, :207).  x_* are fp32 row-major with pitch ld_x; if idx_* is non-NULL row r
 * of the batch is x[idx[r]] (device-side gather from a resident pool, replacing the host slicing of
 * mr_gan.py:190-195, :207).  z_dev = NULL draws z on the device.  stream_mode = 1: idx/labels/z address whole-epoch
 * streams and the batch offset comes from the device batch counter (advanced by the generator step). */
typedef struct mrgan_disc_args {
    const float* x_lab_dev; const int32_t* idx_lab_dev; const int32_t* labels_dev;
    const float* x_unl_dev; const int32_t* idx_unl_dev;
    const float* z_dev;
    int64_t ld_x_lab, ld_x_unl;
    int32_t stream_mode, reserved;
} mrgan_disc_args;

/* train_batch_gen (mr_gan.py:170, :213) */
typedef struct mrgan_gen_args {
    const float* x_unl_dev; const int32_t* idx_unl_dev;
    const float* z_dev;
    int64_t ld_x_unl;
    int32_t stream_mode, reserved;
} mrgan_gen_args;

/* Phases exist for data-parallel drivers: a statistic or gradient exchange (RCCL all-reduce issued by the
 * host on the regions below) sits between consecutive phases.  Single-GPU callers pass (0, -1) = everything. */
enum { MRGAN_D_GEN = 0, MRGAN_D_MAIN = 1, MRGAN_D_ADAM = 2, MRGAN_D_NPHASES = 3 };
enum { MRGAN_G_GEN = 0, MRGAN_G_FEAT = 1, MRGAN_G_BWD = 2, MRGAN_G_TAIL = 3, MRGAN_G_ADAM = 4, MRGAN_G_NPHASES = 5 };
/* out3_host (optional): loss_lab, loss_unl, train_err -- the outputs of train_batch_disc */
int mrgan_disc_step(mrgan_handle* h, const mrgan_disc_args* a, int phase_first, int phase_last,
                    float* out3_host, mrgan_stream stream);
/* out1_host (optional): loss_gen */
int mrgan_gen_step(mrgan_handle* h, const mrgan_gen_args* a, int phase_first, int phase_last,
                   float* out1_host, mrgan_stream stream);
/* fp8 mode, phase-wise callers only (data-parallel hosts; whole-sub-step callers need none of this: mrgan_disc_step /
 * mrgan_gen_step / mrgan_train_pair settle the scales themselves).  The delayed scales of a sub-step kind (0 = D, 1 = G) are
 * settled by running its forward + backward phases (D_GEN .. D_MAIN, or G_GEN .. G_BWD, with the host's exchanges in between)
 * MRGAN_FP8_DRY_PASSES times WITHOUT the update phases, between BEGIN and DONE and each followed by END_PASS.  A handle
 * with synchronised statistics refuses a phase-wise first sub-step that was not calibrated this way. */
enum { MRGAN_FP8_DRY_PASSES = 5 };
enum { MRGAN_FP8_CAL_QUERY = 0, MRGAN_FP8_CAL_BEGIN = 1, MRGAN_FP8_CAL_END_PASS = 2, MRGAN_FP8_CAL_DONE = 3 };
/* QUERY returns 1 (calibrated, or not an fp8 handle) or 0; the other actions return 0 or a negative status */
int mrgan_fp8_calibration(mrgan_handle* h, int kind, int action, mrgan_stream stream);

/* Supervised baseline on the same discriminator stack (mr_nn.py:101-118): one Keras train_on_batch of the 6-layer MLP with
 * GaussianNoise, loss = mse against the one-hot label, Adam with the handle's lr / beta_1 (Keras defaults 0.001 / 0.9: set
 * them in mrgan_config).  Uses the D network and its Adam slots only; one iteration counter step per call.  A handle should
 * run either the GAN loop or this loop.  out2_host (optional): loss, training error of the batch.
 * rows_valid (0 = batch): Keras' short last batch of an epoch -- the first rows_valid rows count (means over rows_valid),
 * the remaining rows must be readable and carry label -1. */
typedef struct mrgan_sup_args {
    const float* x_dev; const int32_t* idx_dev; const int32_t* labels_dev;
    int64_t ld_x;
    int32_t stream_mode, rows_valid;
} mrgan_sup_args;
int mrgan_sup_step(mrgan_handle* h, const mrgan_sup_args* a, float* out2_host, mrgan_stream stream);

/* Log-mel front end of the contact-microphone modality (mr_gan.py:42-47: librosa.feature.melspectrogram(y, sr, n_mels=128)
 * followed by librosa.logamplitude(S, ref_power=np.max); n_fft 2048, hop 512, Slaney mel basis, -80 dB floor).  No handle:
 * the only state is the cached filter bank per (device, sr, n_mels), built on first use under a mutex.
 * y_dev: n_trials rows of n_samples float32 (pitch ld_y); out_dev: n_trials rows of n_mels * mrgan_logmel_frames(n_samples)
 * float32, mel-major like log_S.flatten() (pitch ld_out). */
int32_t mrgan_logmel_frames(int64_t n_samples);
int mrgan_logmel(const float* y_dev, int64_t n_trials, int64_t n_samples, int64_t ld_y, int32_t sr, int32_t n_mels,
                 float* out_dev, int64_t ld_out, mrgan_stream stream);

/* one iteration of the hot loop (mr_gan.py:204-213): D step then G step */
int mrgan_train_pair(mrgan_handle* h, const mrgan_disc_args* d, const mrgan_gen_args* g, mrgan_stream stream);
/* For hosts that drive the phases themselves (data parallel) and know that the next mrgan_disc_step is followed by a
 * mrgan_gen_step whose z is drawn on the device: on != 0 lets that D sub-step also run the G sub-step's generator
 * forward as a second segment of the same launches (what mrgan_train_pair does by itself).  With
 * MRGAN_FLAG_SYNC_STATS the BN_STATS region then holds both segments and ONE all-reduce after D_GEN serves both
 * sub-steps: the host skips the exchange after that G sub-step's G_GEN.  One D sub-step per hint. */
int mrgan_pair_hint(mrgan_handle* h, int on);

/* Launch-structure knobs of one handle (results stay within rounding; defaults are the measured best).  Call between
 * steps, never inside a captured pair. */
enum {
    MRGAN_TUNE_CHAIN = 0,        /* 1 (default where the layer widths allow): the 256-wide tail D3..D5 + loss head of the
                                  * discriminator runs as row-block chain launches; 0: one launch per layer            */
    MRGAN_TUNE_KC_CFG = 1,       /* forward / input-gradient tile: -1 (default) measured table; 0 64x128/3 stages,
                                  * 1 128x128, 2 256x128, 3 256x256, 4 64x128 pipelined fragments, 5 64x128/2 stages  */
    MRGAN_TUNE_KC_PIPE = 2,      /* 1: pipelined-fragment variant for launches with <= 1 tile per CU (default 0)       */
    MRGAN_TUNE_KS_W8 = 3,        /* grouped weight-gradient launch: 0 (default) 8 waves, two blocks per CU; 1: 8 waves with a
                                  * 3-stage ring, one block per CU; 2: 4 waves, two blocks per CU                           */
    MRGAN_TUNE_KS_GROUP = 4,     /* 0: one launch per weight gradient instead of one grouped launch (default 1)        */
    MRGAN_TUNE_PAIR_GEN = 5,     /* 0: mrgan_train_pair runs the two generator forwards separately (default 1: as one) */
    MRGAN_TUNE_HEAD_MFMA = 6     /* feature layers wider than 256 columns (bf16 / fp8): 1 (default) the loss head of the D
                                  * sub-step runs on the matrix cores over 64-row blocks; 0: the scalar head kernel       */
};
int mrgan_set_tuning(mrgan_handle* h, int knob, int value);

/* regions a data-parallel host all-reduces (sum) between phases; fp32 */
enum {
    MRGAN_REGION_BN_STATS = 0,   /* after *_GEN : [2 segments][2][N1p]  sum h, sum h^2 of the generator BatchNorm
                                  * input; segment 1 carries the G sub-step's batch after mrgan_pair_hint            */
    MRGAN_REGION_FM_MOMENTS = 1, /* after G_FEAT: [2][Fp]   sum_b f(fake), sum_b f(real)                         */
    MRGAN_REGION_BN_BWD = 2,     /* after G_BWD : [2][N1p]  sum dy, sum dy*xhat                                  */
    MRGAN_REGION_GRAD_D = 3,     /* after D_MAIN: flat padded gradients of the 12 D tensors + 4 scalars          */
    MRGAN_REGION_GRAD_G = 4,     /* after G_TAIL: flat padded gradients of the 8 G tensors + 4 scalars           */
    MRGAN_REGION_WORKSPACE = 5,
    MRGAN_REGION_GRAD_D_BF16 = 6, /* MRGAN_FLAG_GRAD_BF16: bfloat16 [n] flat padded gradients of the D tensors (same element order) */
    MRGAN_REGION_GRAD_G_BF16 = 7, /* ... of the G tensors                                                                    */
    MRGAN_REGION_TAIL_D = 8,      /* the 4 fp32 scalars at the end of MRGAN_REGION_GRAD_D (loss sums of the D sub-step)        */
    MRGAN_REGION_TAIL_G = 9
};
int mrgan_region(mrgan_handle* h, int region, void** ptr_dev, size_t* bytes);

/* test_batch (mr_gan.py:171, :221-222, :230): learning phase 0, discriminator only.
 * err_host = mean(argmax(logits) != labels) over the n rows. */
int mrgan_eval_error(mrgan_handle* h, const float* x_dev, const int32_t* idx_dev, int64_t ld_x,
                     const int32_t* labels_dev, int64_t n, float* err_host, mrgan_stream stream);
/* logits_dev: fp32 [n, num_classes] */
int mrgan_predict_logits(mrgan_handle* h, const float* x_dev, const int32_t* idx_dev, int64_t ld_x, int64_t n,
                         float* logits_dev, mrgan_stream stream);

/* epoch metrics accumulated on the device (mr_gan.py:208-210 accumulate on the host and force a sync per step):
 * out8_host = sum loss_lab, sum loss_unl, sum train_err, sum loss_gen, last loss_lab, last loss_unl, last err,
 * last loss_gen.  reset != 0 zeroes the sums afterwards. */
int mrgan_read_metrics(mrgan_handle* h, float* out8_host, int reset, mrgan_stream stream);

/* Per-kernel timing with hipEvents on the launch stream (bench.py's live roofline figure): while profiling is on,
 * every kernel of the step is launched with a (start, stop) event pair stamped at its own begin and end on the device
 * (hipExtLaunchKernelGGL), i.e. the interval rocprofv3's kernel trace reports; mrgan_train_pair launches eagerly
 * (no graph replay) behind a short delay kernel per sub-step so that the kernels still run back to back.
 * mrgan_profile_end returns, per distinct kernel instantiation (named as rocprofv3 prints it, MRGAN_PROF_NAME_LEN
 * bytes each): summed time, launch count, summed ALGORITHMIC flops (2 x logical M*N*K of the dense layer) and summed
 * ALGORITHMIC bytes (operands read once + outputs written once; 0 where the library does not account them). */
enum { MRGAN_PROF_NAME_LEN = 96 };
int mrgan_profile_begin(mrgan_handle* h);
int mrgan_profile_end(mrgan_handle* h, mrgan_stream stream, int max_kernels, char* names, float* ms, int32_t* launches,
                      double* flops, double* bytes, int* n_kernels);

#ifdef __cplusplus
}
#endif
#endif
