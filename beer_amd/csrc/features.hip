// Speech feature front-end on the GPU: framing, pre-emphasis, windowing, FFT
// magnitude, triangular filter bank, log, cosine transform + liftering, energy,
// deltas, per-utterance mean normalisation -- over a RAGGED BATCH of utterances
// (one launch per stage for a whole data set shard).
//
// Reference restated: beer/features.py:95-105 (add_deltas), :107-146
// (short_term_mspec), :148-204 (fbank), and the pipeline of
// beer/cli/subcommands/features/extract.py:107-161.  The reference computes in
// float64 (numpy); so does this file -- the work is HBM / latency bound (320 B
// of new samples and <= 1 KB of output per frame), not FLOP bound, so the FFT
// is a plain radix-2 in LDS, one frame per wave.
//
// Host-side tables (window, filter matrix, cosine bases, lifter) are built once
// by the caller, as the reference builds them with numpy.

#include "common.h"

using namespace beer;

namespace {

struct FeaArgs {
    int32_t nutt, flen, fstep, fft_len, log_fft, mode;
    int32_t nfilters, apply_log, n_dct, add_energy, out_ld;
    int64_t total_frames;
    double preemph, log_offset, norm;
    const int64_t* sample_off;
    const int64_t* frame_off;
    const double* utt_mean;
    const double* window;
    const double* filters;
    const int32_t* filt_lo;
    const int32_t* filt_hi;
    const double* dct;
    const double* lifter;
    double* out;
};

__device__ __forceinline__ int find_utt(const int64_t* off, int n, int64_t x) {
    int lo = 0, hi = n;                      // off[lo] <= x < off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= x) lo = mid; else hi = mid;
    }
    return lo;
}

template <typename TIN>
__device__ __forceinline__ double sample_f32_preemph(const TIN* x, int64_t n, int64_t first,
                                                     float pre) {
    // fbank(): the whole signal is cast to float32 and pre-emphasised in float32
    // (features.py:182-184), each operation rounded separately (no FMA).
#pragma clang fp contract(off)
    const float cur = (float)x[n];
    const float prev = (float)x[n > first ? n - 1 : first];
    const float prod = pre * prev;
    return (double)(cur - prod);
}

template <typename TIN>
__device__ __forceinline__ double sample_f64_preemph(const TIN* x, int64_t s0, int i, double mean,
                                                     double pre) {
#pragma clang fp contract(off)
    const double cur = (double)x[s0 + i] - mean;
    const double prev = (double)x[s0 + (i > 0 ? i - 1 : 0)] - mean;
    const double prod = pre * prev;
    return cur - prod;
}

template <typename TIN>
__global__ __launch_bounds__(256) void features_kernel(FeaArgs a, const TIN* __restrict__ signal) {
    extern __shared__ double smem[];
    const int N = a.fft_len, H = N / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    double* tw_re = smem;
    double* tw_im = smem + H;
    double* re = smem + N + (size_t)wave * 2 * N;
    double* im = re + N;

    for (int k = threadIdx.x; k < H; k += blockDim.x) {
        double s, c;
        sincospi(-2.0 * (double)k / (double)N, &s, &c);
        tw_re[k] = c;
        tw_im[k] = s;
    }
    __syncthreads();

    const int64_t total_waves = (int64_t)gridDim.x * nwave;
    const int64_t iters = (a.total_frames + total_waves - 1) / total_waves;
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t frame = it * total_waves + (int64_t)blockIdx.x * nwave + wave;
        const bool valid = frame < a.total_frames;
        int u = 0;
        int64_t first = 0, s0 = 0;
        double mean = 0.0;
        if (valid) {
            u = find_utt(a.frame_off, a.nutt, frame);
            first = a.sample_off[u];
            s0 = first + (frame - a.frame_off[u]) * a.fstep;
            if (a.mode == 1 && a.utt_mean) mean = a.utt_mean[u];
        }
        // 1. frame -> pre-emphasis -> window, stored bit-reversed; zero padding
        for (int i = lane; i < N; i += 64) {
            double v = 0.0;
            if (valid && i < a.flen) {
                if (a.mode == 0) {
                    v = sample_f32_preemph(signal, s0 + i, first, (float)a.preemph);
                } else {
                    // short_term_mspec(): DC removed, pre-emphasis inside the frame
                    // (features.py:124-139), float64, operations rounded separately
                    v = sample_f64_preemph(signal, s0, i, mean, a.preemph);
                }
                v *= a.window[i];
            }
            const int r = (int)(__brev((unsigned)i) >> (32 - a.log_fft));
            re[r] = v;
            im[r] = 0.0;
        }
        __syncthreads();
        // 2. radix-2 decimation-in-time FFT
        for (int s = 1; s <= a.log_fft; ++s) {
            const int half = 1 << (s - 1);
            const int tstep = N >> s;
            for (int b = lane; b < H; b += 64) {
                const int pos = b & (half - 1);
                const int i0 = ((b >> (s - 1)) << s) + pos, i1 = i0 + half;
                const double wr = tw_re[pos * tstep], wi = tw_im[pos * tstep];
                const double xr = re[i1], xi = im[i1];
                const double tr = wr * xr - wi * xi, ti = wr * xi + wi * xr;
                const double ur = re[i0], ui = im[i0];
                re[i0] = ur + tr;
                im[i0] = ui + ti;
                re[i1] = ur - tr;
                im[i1] = ui - ti;
            }
            __syncthreads();
        }
        // 3. magnitude of bins 0 .. N/2-1 (the reference drops the Nyquist bin)
        for (int b = lane; b < H; b += 64) re[b] = hypot(re[b], im[b]);
        __syncthreads();
        // 4. filter bank (+ log) -> im[0 .. nf)
        const int nf = a.nfilters > 0 ? a.nfilters : H;
        for (int f = lane; f < nf; f += 64) {
            double m;
            if (a.nfilters > 0) {
                const double* F = a.filters + (size_t)f * H;
                const int lo = a.filt_lo ? a.filt_lo[f] : 0, hi = a.filt_hi ? a.filt_hi[f] : H - 1;
                m = 0.0;
                for (int b = lo; b <= hi; ++b) m += re[b] * F[b];
            } else {
                m = re[f];
            }
            im[f] = a.apply_log ? log(m + a.log_offset) : m;
        }
        __syncthreads();
        // 5. output row: [energy] + (cepstra | log filter-bank outputs)
        double* row = valid ? a.out + frame * (int64_t)a.out_ld : nullptr;
        int col0 = 0;
        if (a.add_energy) {
            double e = 0.0;
            for (int f = lane; f < nf; f += 64) e += im[f];
            e = wave_sum(e);
            if (row && lane == 0) row[0] = e * a.norm;
            col0 = 1;
        }
        if (a.n_dct > 0) {
            for (int m = lane; m < a.n_dct; m += 64) {
                double c = 0.0;
                for (int f = 0; f < nf; ++f) c += im[f] * a.dct[(size_t)f * a.n_dct + m];
                c *= a.norm;
                if (a.lifter) c *= a.lifter[m];
                if (row) row[col0 + m] = c;
            }
        } else if (row) {
            for (int f = lane; f < nf; f += 64) row[col0 + f] = im[f];
        }
        __syncthreads();
    }
}

// out[t, d] = sum_{j=-w..w} (j / den) * in[clamp(t + j), d], den = 2 sum j^2 over
// -w..w, t clamped inside the utterance (features.py:95-105: edge replication +
// lfilter).
__global__ void deltas_kernel(int nutt, const int64_t* __restrict__ frame_off, int64_t total,
                              int D, int ld, int wlen, const double* __restrict__ in,
                              double* __restrict__ out) {
    double den = 0.0;
    for (int j = -wlen; j <= wlen; ++j) den += (double)(j * j);
    den *= 2.0;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total * D;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = idx / D;
        const int d = (int)(idx % D);
        const int u = find_utt(frame_off, nutt, t);
        const int64_t lo = frame_off[u], hi = frame_off[u + 1] - 1;
        // same accumulation order as the direct-form filter: taps i = 0 .. 2w
        // multiply the samples t + w - i
        double acc = 0.0;
        for (int i = 0; i <= 2 * wlen; ++i) {
            const int j = wlen - i;
            int64_t tt = t + j;
            tt = tt < lo ? lo : (tt > hi ? hi : tt);
            acc += ((double)j / den) * in[tt * ld + d];
        }
        out[t * ld + d] = acc;
    }
}

// per-utterance mean normalisation, in place: x[t, d] -= mean_t x[t, d]
__global__ void cmn_kernel(const int64_t* __restrict__ frame_off, int D, int ld,
                           double* __restrict__ x) {
    const int u = blockIdx.x;
    const int64_t lo = frame_off[u], hi = frame_off[u + 1];
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        double s = 0.0;
        for (int64_t t = lo; t < hi; ++t) s += x[t * ld + d];
        const double m = s / (double)(hi - lo);
        for (int64_t t = lo; t < hi; ++t) x[t * ld + d] -= m;
    }
}

template <typename TIN>
__global__ void signal_mean_kernel(const int64_t* __restrict__ sample_off,
                                   const TIN* __restrict__ signal, double* __restrict__ mean) {
    __shared__ double red[8];
    const int u = blockIdx.x;
    const int64_t lo = sample_off[u], hi = sample_off[u + 1];
    double s = 0.0;
    for (int64_t n = lo + threadIdx.x; n < hi; n += blockDim.x) s += (double)signal[n];
    s = block_sum(s, red);
    if (threadIdx.x == 0) mean[u] = hi > lo ? s / (double)(hi - lo) : 0.0;
}

template <typename TIN>
int extract_launch(const FeaArgs& a, const void* signal, hipStream_t s) {
    const int nwave = 4;
    int64_t blocks = (a.total_frames + nwave - 1) / nwave;
    if (blocks > 4096) blocks = 4096;
    const size_t shmem = sizeof(double) * ((size_t)a.fft_len + (size_t)nwave * 2 * a.fft_len);
    hipLaunchKernelGGL(features_kernel<TIN>, dim3((unsigned)blocks), dim3(64 * nwave), shmem, s, a,
                       (const TIN*)signal);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

}  // namespace

extern "C" {

int beer_features_signal_mean(int in_dtype, int32_t nutt, const int64_t* sample_off,
                              const void* signal, double* mean, void* stream) {
    BEER_REQUIRE(nutt >= 0 && sample_off && mean);
    if (nutt == 0) return BEER_OK;
    BEER_REQUIRE(signal);
    hipStream_t s = as_stream(stream);
    if (in_dtype == BEER_I16)
        hipLaunchKernelGGL(signal_mean_kernel<int16_t>, dim3(nutt), dim3(256), 0, s, sample_off,
                           (const int16_t*)signal, mean);
    else if (in_dtype == BEER_F32)
        hipLaunchKernelGGL(signal_mean_kernel<float>, dim3(nutt), dim3(256), 0, s, sample_off,
                           (const float*)signal, mean);
    else if (in_dtype == BEER_F64)
        hipLaunchKernelGGL(signal_mean_kernel<double>, dim3(nutt), dim3(256), 0, s, sample_off,
                           (const double*)signal, mean);
    else
        return BEER_EINVAL;
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

int beer_features_extract(int in_dtype, int32_t nutt, const int64_t* sample_off,
                          const int64_t* frame_off, int64_t total_frames, const void* signal,
                          const double* utt_mean, const beer_feaconf* c, double* out,
                          int32_t out_ld, void* stream) {
    BEER_REQUIRE(c && nutt >= 0 && total_frames >= 0);
    if (nutt == 0 || total_frames == 0) return BEER_OK;
    BEER_REQUIRE(sample_off && frame_off && signal && out && c->window);
    BEER_REQUIRE(c->flen >= 1 && c->fstep >= 1 && c->fft_len >= 64 && c->fft_len <= 2048 &&
                 (c->fft_len & (c->fft_len - 1)) == 0 && c->flen <= c->fft_len);
    BEER_REQUIRE(c->mode == 0 || c->mode == 1);
    BEER_REQUIRE(c->nfilters >= 0 && c->nfilters <= c->fft_len / 2 && c->n_dct >= 0);
    BEER_REQUIRE(c->nfilters == 0 || c->filters);
    BEER_REQUIRE(c->n_dct == 0 || c->dct);
    const int nf = c->nfilters > 0 ? c->nfilters : c->fft_len / 2;
    const int width = (c->add_energy ? 1 : 0) + (c->n_dct > 0 ? c->n_dct : nf);
    BEER_REQUIRE(out_ld >= width);
    FeaArgs a;
    a.nutt = nutt;
    a.flen = c->flen;
    a.fstep = c->fstep;
    a.fft_len = c->fft_len;
    a.log_fft = 0;
    while ((1 << a.log_fft) < c->fft_len) ++a.log_fft;
    a.mode = c->mode;
    a.nfilters = c->nfilters;
    a.apply_log = c->apply_log;
    a.n_dct = c->n_dct;
    a.add_energy = c->add_energy;
    a.out_ld = out_ld;
    a.total_frames = total_frames;
    a.preemph = c->preemph;
    a.log_offset = c->log_offset;
    a.norm = c->norm;
    a.sample_off = sample_off;
    a.frame_off = frame_off;
    a.utt_mean = utt_mean;
    a.window = c->window;
    a.filters = c->filters;
    a.filt_lo = c->filt_lo;
    a.filt_hi = c->filt_hi;
    a.dct = c->dct;
    a.lifter = c->lifter;
    a.out = out;
    hipStream_t s = as_stream(stream);
    if (in_dtype == BEER_I16) return extract_launch<int16_t>(a, signal, s);
    if (in_dtype == BEER_F32) return extract_launch<float>(a, signal, s);
    if (in_dtype == BEER_F64) return extract_launch<double>(a, signal, s);
    return BEER_EINVAL;
}

int beer_features_deltas(int32_t nutt, const int64_t* frame_off, int64_t total_frames, int32_t D,
                         int32_t ld, int32_t wlen, const double* in, double* out, void* stream) {
    BEER_REQUIRE(nutt >= 0 && total_frames >= 0 && D >= 1 && ld >= D && wlen >= 1);
    if (nutt == 0 || total_frames == 0) return BEER_OK;
    BEER_REQUIRE(frame_off && in && out);
    const int64_t total = total_frames * D;
    const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(deltas_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), nutt, frame_off,
                       total_frames, D, ld, wlen, in, out);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

int beer_features_cmn(int32_t nutt, const int64_t* frame_off, int32_t D, int32_t ld, double* x,
                      void* stream) {
    BEER_REQUIRE(nutt >= 0 && D >= 1 && ld >= D);
    if (nutt == 0) return BEER_OK;
    BEER_REQUIRE(frame_off && x);
    hipLaunchKernelGGL(cmn_kernel, dim3(nutt), dim3(D < 64 ? 64 : (D > 256 ? 256 : D)), 0,
                       as_stream(stream), frame_off, D, ld, x);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

}  // extern "C"
