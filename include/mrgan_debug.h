/* libmrgan_hip -- diagnostic entry points (parity tests, kernel timing experiments).  Not part of the drop-in boundary:
 * a host that replaces mr_gan.py:169-171 needs include/mrgan_abi.h only. */
#ifndef MRGAN_DEBUG_H
#define MRGAN_DEBUG_H

#include "mrgan_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

int mrgan_debug_noise(mrgan_handle* h, uint32_t site, uint32_t seg, uint32_t step, uint32_t row0, int rows, int cols,
                      float* out_dev, mrgan_stream stream);
int mrgan_debug_tr_probe(uint16_t* out1024_dev, mrgan_stream stream);
/* timing experiments only (results become wrong): 2 = skip the GEMM epilogues, 4 = skip the GEMM main loops */
int mrgan_debug_ablate(mrgan_handle* h, int bits);
int mrgan_debug_buffer(mrgan_handle* h, int kind, int l, void** ptr_dev, int* rows_per_seg, int* ld, int* elem_size);
/* average device time (us) of `reps` back-to-back launches of one bf16 product on scratch buffers:
 * op 0 forward (relu+noise+mask), 1 input-gradient (relu mask), 2 weight-gradient with `splits` slabs */
int mrgan_debug_gemm_time(int op, int m, int n, int k, int nbatch, int splits, int reps, int ablate, int kc_cfg, float* avg_us);
/* raw GEMM entry for kernel-level parity tests: op 0 = Y = act(X W + b), 1 = dX = dY W^T, 2 = dW = X^T dY.
 * fp32 device buffers in and out (converted internally when dtype = bf16). */
int mrgan_debug_gemm(int dtype, int op, int m, int n, int k, const float* a_dev, const float* b_dev, const float* bias_dev,
                     int act, int splits, float* out_dev, mrgan_stream stream);

/* fp8 (OCP e4m3) forward product on the matrix cores, operands quantised from the fp32 inputs with per-tensor scales:
 * out[m,n] = act((q(a * scale_a) q(b * scale_b)) / (scale_a scale_b) + bias).  reps > 0 also times `reps` launches.
 * kc_cfg: -1 = the launcher's choice, 1 = 128x128 blocks, 3 = 256x256 blocks. */
int mrgan_debug_gemm_fp8(int m, int n, int k, const float* a_dev, const float* b_dev, const float* bias_dev, int act, float scale_a,
                         float scale_b, float* out_dev, int reps, float* avg_us, int kc_cfg, mrgan_stream stream);

#ifdef __cplusplus
}
#endif
#endif
