// Argument blocks + launchers for the non-GEMM kernels of the mr_gan training path.
#pragma once
#include "common.h"

namespace mrgan {

constexpr int KMAX = 8;            // classes are padded to 8 logits (reference: 6 materials, mr_gan.py:80)
constexpr int HEAD_ROWS = 32;      // rows per loss-head block
constexpr int HEAD_CHUNK = 256;    // feature columns of the loss head held in LDS at a time (wider layers are walked in chunks)

// ---- staging: rows of the (scaled) data matrix -> noisy discriminator input; z -> generator input ----
struct StageSeg {
    const float* src; const int32_t* idx; long ld;   // row r comes from src[(idx ? idx[o+r] : o+r) * ld + c]
    int rows, cols, cols_pad;
    void* out; int ldo;                             // T [rows][ldo]; columns >= cols are zero-filled
    float sigma; uint32_t site, seg;
    int gen;                                        // 1: out = N(0,1) only (z drawn on device), src ignored
    int stream;                                     // 1: o = batch * rows from the device batch counter
    uint32_t iter_off;                              // noise key uses DevState::iter + iter_off (z of a later sub-step)
};
struct StageArgs {
    StageSeg s[5]; int nseg;
    uint64_t seed; uint32_t row0;
    const DevState* cur;
};
int launch_stage(int bf16, const StageArgs& a, hipStream_t s);

// ---- BatchNorm (batch statistics, biased variance; mr_gan.py:112) ----
struct BnApplyArgs {
    const void* h; void* out; int ld; int rows; int cols;      // cols = logical width
    const float* cs1; const float* cs2; int npart; int ldcs;   // column partial sums of h and h^2
    float count, eps;                                          // global batch size
    const float* gamma; const float* beta;
    float* mu; float* rstd;                                    // saved for the backward pass
    long cs_seg_stride;                                        // floats between the partial sums of consecutive segments
    int nseg; long seg_rows;                                   // nseg independent batches, seg_rows rows apart in h / out; their
                                                               // partial sums follow each other (npart rows each), mu / rstd ld apart
};
int launch_bn_apply(int bf16, const BnApplyArgs& a, hipStream_t s);

struct BnBwdArgs {
    const void* dy; const void* h; void* dpre; int ld; int rows; int cols;
    const float* cs1; const float* cs2; int npart; int ldcs;   // sum(dy), sum(dy*xhat) partials
    float count;
    const float* gamma; const float* mu; const float* rstd;
    float* db_part;                                            // [stat_row_blocks(rows)][ld] column sums of dpre
};
int launch_bn_bwd(int bf16, const BnBwdArgs& a, hipStream_t s);
int stat_row_blocks(int rows);                                 // row blocks of the column-statistic kernels

// ---- loss head (mr_gan.py:128, :146-149, :161-162): last dense + losses + their gradients ----
enum { HEAD_LAB = 0, HEAD_UNL = 1, HEAD_FAKE = 2, HEAD_EVAL = 3, HEAD_LOGITS = 4,
       HEAD_MSE = 5 };   // supervised baseline (mr_nn.py:112): mean squared error against the one-hot label
struct HeadArgs {
    const void* f; long f_bs; int ldf;         // features [seg][rows][ldf]
    int rows, nseg, seg_kind[3];
    int feat, feat_valid, classes;             // padded / logical feature width, number of classes
    const float* w; int ldw; const float* b;   // last dense: w[feat][ldw], b[classes]
    const int32_t* labels;                     // [rows] (stream mode: offset by batch*rows)
    const DevState* st; int labels_stream;
    float inv_count, unl_weight;               // 1/(global batch), mr_gan.py:79
    float* logits; long logits_bs;             // optional [seg][rows][KMAX]
    void* dpre; long dpre_bs; int ldd;         // dL/d(pre-activation of the feature layer), T
    float* part; long part_stride;             // per-block partial gradients: part[blk][0 .. feat*KMAX) = dW6,
    int off_db, off_dbf;                       //   [off_db .. +KMAX) = db6, [off_dbf .. +feat) = bias grad of the feature layer
    float* loss_part;                          // [blk][4] : sum loss_lab, sum loss_unl terms, sum err, 0
    int* err_count;                            // HEAD_EVAL: integer count of argmax != label
    // fp8 mode: dpre leaves as e5m2 copies scaled by q8_slot->scale (row-major [seg][rows][ldq8] and transposed [feat][ldq8t] with
    // segment s at row offset s * q8t_bs), max |dpre| -> q8_slot; dpre itself may then be null
    unsigned char* q8; long q8_bs; int ldq8;
    unsigned char* q8t; long q8t_bs; int ldq8t;
    Fp8Slot* q8_slot;
};
int launch_head(int bf16, const HeadArgs& a, hipStream_t s);
// dst[g][i] = sum over partial rows p = g, g+ngroups, ... of src[p][i]  (i < n, rows `stride` apart)
int launch_reduce_partials(const float* src, int nsrc, long stride, int n, int ngroups, float* dst, hipStream_t s);
int init_kernel_attributes();

// ---- feature matching (mr_gan.py:152-154) ----
struct FmArgs {
    const float* cs; int npart_fake, npart_real, ldcs;   // column partial sums of f: fake rows first, then real
    float count, grad_scale; int feat, feat_valid;        // rows behind each mean; 1/world for per-shard statistics
    const uint16_t* mask; int ldm;                        // lane-native relu mask (gemm.h) of the fake rows' feature layer
    void* dpre; int ldd; int rows;
    float* loss_out; float* accum;                        // step scalar + epoch accumulator (block 0 only)
    unsigned char* q8; int ldq8; Fp8Slot* q8_slot;        // fp8 mode: e5m2 copy of dpre (row-major), scaled by the slot; dpre may be null
    int rb;                                               // rows per block (set by launch_fm)
    // wide feature layers: the loss is assembled from per-column-block partials (lscratch[gridDim.x]) by the last block to
    // arrive (*lcount: ticket, left at zero again) in a fixed order, instead of by one extra block that walks every column
    float* lscratch; unsigned int* lcount;
};
int launch_fm(int bf16, const FmArgs& a, hipStream_t s);

// ---- collapse per-row-tile partial sums to one row (data-parallel statistic exchange) ----
int launch_colsum_finalize(const float* part1, const float* part2, int npart, int ld, int n, float* out /* [nseg][2][n] */, hipStream_t s,
                           int nseg = 1 /* segments: partial rows npart * ld floats apart */);

// ---- multi-tensor Adam (Keras 2.0.9 formula, mr_gan.py:165-167) ----
struct AdamTile {
    float* p; float* m; float* v;
    const float* g; int nslab; long slab_stride;       // gradient = sum of nslab slabs
    float* flat;                                       // flat gradient buffer (tile origin)
    __bf16* flat16;                                    // MRGAN_FLAG_GRAD_BF16: bfloat16 flat gradient buffer instead (null otherwise)
    __bf16* w16; __bf16* wt16;                         // bf16 copies [K][N] and [N][K] (null for fp32 / 1-D)
    unsigned char* w8; unsigned char* w8t; Fp8Slot* w8_slot;      // fp8 mode: e4m3 copies of the bf16 values (same tile origins)
    int ld, ldt;                                       // row pitch of p/m/v/g/w16 ; of wt16
    int rows, cols;                                    // tile extent (<= 64 x 64)
};
enum { ADAM_FUSED = 0, ADAM_REDUCE_ONLY = 1, ADAM_FROM_FLAT = 2 };
struct AdamArgs {
    const AdamTile* tiles; int ntiles; int mode;
    float b1, b2, eps;
    const DevState* st;
    // metrics finish (block 0): loss partials of this sub-step -> step_out[0..2] and epoch accumulators
    const float* loss_part; int nloss_part; float inv_rows; float* step_out; float* accum;
    float* flat_tail;                                  // 4 floats after the flat gradients (travel with the all-reduce)
    // The updating launch is the last kernel of a sub-step: it publishes the next sub-step's DevState slot
    // (iterations + 1, batch counter + advance_batch, lr_t of the new iteration).  Null for ADAM_REDUCE_ONLY.
    DevState* next; int advance_batch; float lr;
};
int launch_adam(const AdamArgs& a, hipStream_t s);

int launch_noise_debug(uint64_t seed, uint32_t site, uint32_t seg, uint32_t step, uint32_t row0, int rows, int cols,
                       float* out, hipStream_t s);

}  // namespace mrgan
