// Row-block chain kernel: several consecutive dense products of one 64-row block run inside ONE launch, the block's
// activations staying in LDS between them (gemm_chain.hip).
//
// Why: the narrow tail of the discriminator (mr_gan.py:123-128: 500 -> 250 -> 250 -> 250 -> 6) is 5 % of the step's
// FLOPs but, launched layer by layer, 12 of its 34 launches -- each a 6-13 us kernel whose main loop is shorter than
// its prologue + epilogue.  The weights of these layers total 0.5 MB, so every block can stream all of them from its
// XCD's L2 while the block's rows never leave the CU:
//     D sub-step :  D3 D4 D5 forward -> loss head (mr_gan.py:128, :146-149) -> dX through D5 D4 D3      (1 launch, was 7)
//     G sub-step :  D3 D4 D5 forward (+ feature-matching column sums)                                  (1 launch, was 3)
//                   feature-matching gradient (mr_gan.py:152-154) -> dX through D5 D4 D3               (1 launch, was 4)
// Row blocks are independent (the losses are per-row sums scaled by 1/B), so there is no inter-workgroup traffic at all.
#pragma once
#include "aux_kernels.h"
#include "common.h"

namespace mrgan {

constexpr int CH_ROWS = 64;            // rows per block
constexpr int CH_THREADS = 512;        // 8 waves: wave w owns columns [32 w, 32 w + 32) of a 256-column pass, all 64 rows
constexpr int CH_PW = 256;             // output columns per pass (wider layers take several passes over the resident A)
constexpr int CH_KMAX = 512;           // widest resident activation (reduction length of any product in a chain)
constexpr int CH_MAX_OPS = 8;
// LDS map (bytes): two activation images and a 2-stage ring of weight tiles [256 columns][64 k]
constexpr int CH_BUF0 = 0, CH_BUF0_BYTES = CH_ROWS * CH_KMAX * 2;                  // 64 KiB: up to 512 columns
constexpr int CH_BUF1 = CH_BUF0 + CH_BUF0_BYTES, CH_BUF1_BYTES = CH_ROWS * CH_PW * 2;   // 32 KiB: up to 256 columns
constexpr int CH_RING = CH_BUF1 + CH_BUF1_BYTES, CH_STAGE_BYTES = CH_PW * 128;     // 32 KiB per stage
constexpr int CH_LDS_BYTES = CH_RING + 2 * CH_STAGE_BYTES;                         // 160 KiB: the whole CU
// the same map for blocks of `rows` rows (64, or 32 for launches with few row blocks): image 0 | image 1 | the 2-stage ring
constexpr int chain_buf0(int rows) { return 0; }
constexpr int chain_buf1(int rows) { return rows * CH_KMAX * 2; }
constexpr int chain_ring(int rows) { return rows * (CH_KMAX + CH_PW) * 2; }
constexpr int chain_stages(int rows) { return rows <= 32 ? 3 : 2; }      // ring depth: what fits beside the images in 160 KiB
constexpr int chain_lds_bytes(int rows) { return chain_ring(rows) + chain_stages(rows) * CH_STAGE_BYTES; }

enum { CH_OP_GEMM = 0, CH_OP_HEAD = 1 };
enum { CH_FWD_RELU = 0, CH_DX_RELU = 1 };
enum { CH_A_GLOBAL = 0, CH_A_LDS = 1, CH_A_FMGRAD = 2 };
// the three chains of a training step: op lists [F F F H X X X], [F F F], [X X X] (F forward, H loss head, X dX)
enum { CH_V_DTAIL = 0, CH_V_GFWD = 1, CH_V_GBWD = 2 };
// timing experiments (results are wrong): skip the loss head / the copies to HBM / the MFMAs / the weight stream / the epilogue math
enum { CH_ABL_HEAD = 256, CH_ABL_COPY = 512, CH_ABL_MFMA = 1024, CH_ABL_STREAM = 2048, CH_ABL_EPI = 4096 };

struct ChainOp {
    int kind;                          // CH_OP_*
    // ---- CH_OP_GEMM: out[64][N] = epilogue(A[64][K] Bt[N][K]^T) ----
    int K, N, n_valid;                 // padded reduction / output widths (multiples of 64), logical output width
    const __bf16* W;                   // Bt: [N][K], reduction index contiguous (forward: W^T copy; dX: W copy)
    int a_off, o_off;                  // LDS byte offsets of the A image and of the output image
    int mode;                          // CH_FWD_RELU | CH_DX_RELU
    const float* bias;                 // forward
    float sigma; uint32_t site;        // forward: out += sigma * N(0,1) (the next layer's GaussianNoise); site of the draw
    __bf16* out; long out_bs; int ldo; // global copy of the output [seg][S][ldo]
    uint16_t* mask; long mask_bs; int ldm;      // lane-native relu mask (gemm.h): written by CH_FWD_RELU, read by CH_DX_RELU
    float* cs; int ldcs;               // optional column sums of the (unrounded) output: one partial row per (segment, row block)
};

struct ChainArgs {
    int variant;                       // CH_V_*
    int nops; ChainOp op[CH_MAX_OPS];
    int rows, nseg, S;                 // valid rows per segment, segments, segment stride (rows)
    int block_rows;                    // rows per block: 64, or 32 (CH_V_GFWD / CH_V_GBWD only)
    int a_kind;                        // how the first A image is produced
    const __bf16* a; long a_bs; int lda; int a_cols;      // CH_A_GLOBAL: rows of a[seg][S][lda], a_cols (padded) columns
    // CH_A_FMGRAD: A = relu-mask ? gj : 0 with gj from the feature-matching moments (mr_gan.py:152-154)
    FmArgs fm;
    const __bf16* fm_feat; int fm_ldf;                     // features of the generated rows [rows][fm_ldf] (their sign is the relu mask)
    // CH_OP_HEAD
    HeadArgs head;
    int head_f_off, head_scratch_off, head_o_off;         // LDS offsets: features in, scratch, dpre out
    int seg0;                          // noise segment id of segment 0
    uint64_t seed; uint32_t row0; const DevState* st;
    unsigned long long* stamps;        // diagnostic build only (make STAMPS=1): [block][8] cycles per phase
    int ablate;                        // timing experiments only (mrgan_debug_ablate): CH_ABL_* bits

};

int launch_chain(const ChainArgs& a, hipStream_t s);

// Stand-alone loss head on the matrix cores for feature layers the chain cannot hold (wider than 256 columns; BASELINE
// configs[4]: 4096): the three products of chain_head over 64-row blocks, the feature dimension walked in 256-column chunks.
// feat % 256 == 0; bf16 features; segment kinds LAB / UNL / FAKE (training); mask = the feature layer's lane-native relu mask.
struct HeadWideArgs {
    HeadArgs h;
    const uint16_t* mask; long mask_bs; int ldm;
    __bf16* w6c; __bf16* w6r;          // scratch: the bf16 addends of W6, class-major [3][KMAX][feat] and row-major [3][feat][KMAX]
};
constexpr int HEAD_WIDE_ROWS = CH_ROWS;
int launch_w6_split(const HeadWideArgs& a, hipStream_t s);       // first: the addends of the current W6
int launch_head_wide(const HeadWideArgs& a, hipStream_t s);
int chain_init_attributes();

}  // namespace mrgan
