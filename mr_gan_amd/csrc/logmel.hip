// Log-mel front end of the contact-microphone modality.
//
// Reference: mr_gan.py:42-47 (same lines in mr_nn.py:38-43, mr_svm.py:46-51):
//     S = librosa.feature.melspectrogram(np.array(objData['contact'][i]), sr=48000, n_mels=128)
//     log_S = librosa.logamplitude(S, ref_power=np.max)
// librosa 0.5.1 is a third-party dependency that is not in the reference tree and not installed here; what it computes
// (published algorithm, restated on the CPU in oracle/melspec_oracle.py):
//     stft    n_fft 2048, hop 512, periodic Hann window, centred frames with reflect padding -> |X|^2
//     mel     Slaney scale (linear below 1 kHz, log above), fmin 0, fmax sr / 2, area normalisation, dense [n_mels][1025]
//     log     10 log10(max(S, 1e-10)) - 10 log10(max(max S, 1e-10)), floored at (max - 80 dB)
//
// One workgroup per trial.  A trial is 0.2 s of audio (9600 samples, 19 frames): the block walks its frames, each a
// 2048-point radix-2 FFT in LDS (16 KiB, twiddles 8 KiB), turns the 1025 power bins into mel bands with the sparse rows of
// the filterbank (every triangle covers a few bins), keeps all [frames][mels] powers in LDS, and finishes with the trial-wide
// maximum and the dB conversion.  HBM traffic = the samples once (38 KB) + the features once (9.7 KB); 6000 trials of the
// MREO set make 6000 blocks, 23 per CU.
#include "logmel.h"

#include <cmath>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace mrgan {
namespace {

constexpr int NBIN = LM_NFFT / 2 + 1, LOGN = 11, THREADS = 256;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__global__ __launch_bounds__(THREADS) void logmel_kernel(const float* __restrict__ y, long ld_y, int n, int n_frames, int n_mels,
                                                         const int* __restrict__ fstart, const int* __restrict__ flen,
                                                         const int* __restrict__ foff, const float* __restrict__ fw,
                                                         float* __restrict__ out, long ld_out) {
    __shared__ float2 buf[LM_NFFT];
    __shared__ float2 tw[LM_NFFT / 2];         // exp(-2 pi i k / 2048)
    __shared__ float pw[NBIN + 3];
    __shared__ float red[THREADS / 64];
    extern __shared__ float mel[];             // [n_frames][n_mels]
    const int t = threadIdx.x;
    const float* yt = y + (long)blockIdx.x * ld_y;

    for (int k = t; k < LM_NFFT / 2; k += THREADS) {
        float sn, cs;
        sincospif(-(float)k * (1.0f / 1024.0f), &sn, &cs);
        tw[k] = make_float2(cs, sn);
    }
    __syncthreads();

    for (int f = 0; f < n_frames; ++f) {
        // frame f of the reflect-padded signal, times the periodic Hann window, stored bit-reversed
        for (int i = t; i < LM_NFFT; i += THREADS) {
            int pos = f * LM_HOP + i - LM_NFFT / 2;
            pos = pos < 0 ? -pos : pos;
            pos = pos >= n ? 2 * (n - 1) - pos : pos;
            const float c = (i < LM_NFFT / 2) ? tw[i].x : -tw[i - LM_NFFT / 2].x;       // cos(2 pi i / N)
            const float w = 0.5f - 0.5f * c;
            buf[__brev((unsigned)i) >> (32 - LOGN)] = make_float2(yt[pos] * w, 0.f);
        }
        __syncthreads();
#pragma unroll 1
        for (int s = 1; s <= LOGN; ++s) {
            const int half = 1 << (s - 1);
            for (int b = t; b < LM_NFFT / 2; b += THREADS) {
                const int j = b & (half - 1);
                const int i0 = ((b >> (s - 1)) << s) + j, i1 = i0 + half;
                const float2 a = buf[i0], v = cmul(buf[i1], tw[j << (LOGN - s)]);
                buf[i0] = make_float2(a.x + v.x, a.y + v.y);
                buf[i1] = make_float2(a.x - v.x, a.y - v.y);
            }
            __syncthreads();
        }
        for (int k = t; k < NBIN; k += THREADS) pw[k] = buf[k].x * buf[k].x + buf[k].y * buf[k].y;
        __syncthreads();
        for (int m = t; m < n_mels; m += THREADS) {
            const float* w = fw + foff[m];
            const float* p = pw + fstart[m];
            float acc = 0.f;
            for (int q = 0; q < flen[m]; ++q) acc = fmaf(w[q], p[q], acc);
            mel[f * n_mels + m] = acc;
        }
        __syncthreads();                       // pw / buf are rewritten by the next frame
    }

    // trial-wide maximum (ref_power = np.max), then dB with the 80 dB floor
    const int total = n_frames * n_mels;
    float mx = 0.f;
    for (int i = t; i < total; i += THREADS) mx = fmaxf(mx, mel[i]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float amin = 1e-10f, top_db = 80.0f;
    const float ref = log10f(fmaxf(amin, mx));
    float* o = out + (long)blockIdx.x * ld_out;
    for (int i = t; i < total; i += THREADS) {
        const int m = i / n_frames, f = i - m * n_frames;                 // log_S.flatten(): mel-major
        const float db = 10.0f * (log10f(fmaxf(amin, mel[f * n_mels + m])) - ref);      // the maximum maps to exactly 0 dB
        o[i] = fmaxf(db, -top_db);
    }
}

// ---- Slaney mel filterbank (librosa.filters.mel, htk=False, norm=1) in double precision, kept as sparse rows ----
double hz_to_mel(double f) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
    const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

struct Bank { int* start; int* len; int* off; float* w; };
std::mutex g_bank_mutex;
std::map<std::tuple<int, int, int>, Bank> g_banks;       // (device, sr, n_mels)

hipError_t build_bank(int sr, int n_mels, Bank* out) {
    std::vector<double> mel_f(n_mels + 2);
    const double m0 = hz_to_mel(0.0), m1 = hz_to_mel(sr / 2.0);
    for (int i = 0; i < n_mels + 2; ++i) mel_f[i] = mel_to_hz(m0 + (m1 - m0) * i / (n_mels + 1));
    std::vector<int> start(n_mels), len(n_mels), off(n_mels);
    std::vector<float> w;
    for (int m = 0; m < n_mels; ++m) {
        const double lo = mel_f[m], ce = mel_f[m + 1], hi = mel_f[m + 2], enorm = 2.0 / (hi - lo);
        int first = -1, last = -2;
        std::vector<double> row(NBIN);
        for (int k = 0; k < NBIN; ++k) {
            const double fr = (sr / 2.0) * k / (NBIN - 1);
            const double lower = (fr - lo) / (ce - lo), upper = (hi - fr) / (hi - ce);
            const double v = std::fmax(0.0, std::fmin(lower, upper)) * enorm;
            row[k] = v;
            if (v > 0.0) { if (first < 0) first = k; last = k; }
        }
        if (first < 0) { first = 0; last = -1; }
        start[m] = first; len[m] = last - first + 1; off[m] = (int)w.size();
        for (int k = first; k <= last; ++k) w.push_back((float)row[k]);
    }
    if (w.empty()) w.push_back(0.f);
    hipError_t e;
    if ((e = hipMalloc(&out->start, sizeof(int) * n_mels)) != hipSuccess) return e;
    if ((e = hipMalloc(&out->len, sizeof(int) * n_mels)) != hipSuccess) return e;
    if ((e = hipMalloc(&out->off, sizeof(int) * n_mels)) != hipSuccess) return e;
    if ((e = hipMalloc(&out->w, sizeof(float) * w.size())) != hipSuccess) return e;
    if ((e = hipMemcpy(out->start, start.data(), sizeof(int) * n_mels, hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMemcpy(out->len, len.data(), sizeof(int) * n_mels, hipMemcpyHostToDevice)) != hipSuccess) return e;
    if ((e = hipMemcpy(out->off, off.data(), sizeof(int) * n_mels, hipMemcpyHostToDevice)) != hipSuccess) return e;
    return hipMemcpy(out->w, w.data(), sizeof(float) * w.size(), hipMemcpyHostToDevice);
}

}  // namespace

int launch_logmel(const float* y, long n_trials, long n_samples, long ld_y, int sr, int n_mels, float* out, long ld_out,
                  hipStream_t s, const char** err) {
    *err = "";
    if (!y || !out || n_trials <= 0) { *err = "logmel: null buffer or no trials"; return -1; }
    // reflect padding of n_fft / 2 samples needs a signal longer than that (librosa raises the same way)
    if (n_samples <= LM_NFFT / 2 || n_samples > (1 << 24)) { *err = "logmel: each trial needs more than 1024 samples"; return -1; }
    if (n_mels < 1 || n_mels > LM_MAX_MELS || sr < 2) { *err = "logmel: n_mels outside [1, 256] or bad sample rate"; return -1; }
    const int n_frames = logmel_frames(n_samples);
    if (ld_y < n_samples || ld_out < (long)n_frames * n_mels) { *err = "logmel: row pitch smaller than the row"; return -1; }
    const size_t dyn = sizeof(float) * (size_t)n_frames * n_mels;
    if (dyn > 96 * 1024) { *err = "logmel: frames x mels of one trial exceed the LDS budget (96 KiB)"; return -1; }
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { *err = hipGetErrorString(e); return -10; }
    Bank bank;
    {
        std::lock_guard<std::mutex> lock(g_bank_mutex);
        const auto key = std::make_tuple(dev, sr, n_mels);
        auto it = g_banks.find(key);
        if (it == g_banks.end()) {
            Bank b = {nullptr, nullptr, nullptr, nullptr};
            e = build_bank(sr, n_mels, &b);
            if (e != hipSuccess) { *err = hipGetErrorString(e); return -10; }
            it = g_banks.emplace(key, b).first;
        }
        bank = it->second;
    }
    if (dyn > 32 * 1024) {
        e = hipFuncSetAttribute((const void*)logmel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        if (e != hipSuccess) { *err = hipGetErrorString(e); return -10; }
    }
    hipLaunchKernelGGL(logmel_kernel, dim3((unsigned)n_trials), dim3(THREADS), dyn, s, y, ld_y, (int)n_samples, n_frames, n_mels,
                       bank.start, bank.len, bank.off, bank.w, out, ld_out);
    e = hipGetLastError();
    if (e != hipSuccess) { *err = hipGetErrorString(e); return -10; }
    return 0;
}

}  // namespace mrgan
