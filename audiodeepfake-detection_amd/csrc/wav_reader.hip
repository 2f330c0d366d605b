// Input side of the hot path: the windowed WAV read of the reference's
// CustomDataset.__getitem__ (src/audiofakedetect/data_loader.py:323-353: torchaudio.load(path, frame_offset,
// num_frames) + torchaudio.functional.resample) for a whole batch at once.
//
//   afd_wav_read_windows   host: `n` windows of `win` frames of 16-bit PCM (first channel) from `n` files, read by a
//                          few threads with pread straight into the caller's (pinned) int16 buffer; RIFF chunks are
//                          walked ("fmt ", then "data"; others skipped), short reads are zero-filled.  One Python
//                          DataLoader worker delivers ~1 500-5 900 frames/s (tools/loader_rate.py); the level-8
//                          step consumes 15 000 per GPU.
//   afd_pcm16_resample     device: int16 -> float32 / 32768 and, when the file rate differs from the target rate,
//                          the Hann-windowed sinc interpolator of torchaudio.functional.resample as a polyphase
//                          filter: out[j * new + ph] = sum_t K[ph][t] x[j * orig + t - width] with zero padding
//                          (the kernel bank is built by the caller: audiofakedetect.data_loader._sinc_resample_kernel).
#include "afd_common.h"
#include "../../include/afd_hip.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct WavInfo {
    int channels = 0, rate = 0, bits = 0, format = 0, block_align = 0;
    bool pcm_subformat = true;  // WAVE_FORMAT_EXTENSIBLE: the sub-format GUID starts with the PCM tag
    long long data_off = -1, data_bytes = 0;
};

unsigned rd32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((unsigned)p[3] << 24); }
unsigned rd16(const unsigned char* p) { return p[0] | (p[1] << 8); }

// 0 = ok; walks the RIFF chunks up to "data"
int parse_wav(int fd, WavInfo& w) {
    unsigned char h[12];
    if (pread(fd, h, 12, 0) != 12 || memcmp(h, "RIFF", 4) || memcmp(h + 8, "WAVE", 4)) return -1;
    long long pos = 12;
    bool have_fmt = false;
    for (int guard = 0; guard < 64; ++guard) {
        unsigned char c[8];
        if (pread(fd, c, 8, pos) != 8) return -1;
        const unsigned size = rd32(c + 4);
        if (!memcmp(c, "fmt ", 4)) {
            unsigned char f[16];
            if (size < 16 || pread(fd, f, 16, pos + 8) != 16) return -1;
            w.format = (int)rd16(f);
            w.channels = (int)rd16(f + 2);
            w.rate = (int)rd32(f + 4);
            w.block_align = (int)rd16(f + 12);
            w.bits = (int)rd16(f + 14);
            if (w.format == 0xFFFE) {  // extensible: cbSize(2) validBits(2) channelMask(4) SubFormat GUID(16)
                unsigned char g[2];
                w.pcm_subformat = size >= 40 && pread(fd, g, 2, pos + 8 + 24) == 2 && rd16(g) == 1;
            }
            have_fmt = true;
        } else if (!memcmp(c, "data", 4)) {
            if (!have_fmt) return -1;
            w.data_off = pos + 8;
            // the chunk size of a truncated or streamed file (0, 0xFFFFFFFF) is not to be trusted: the data end
            // where the file ends
            struct stat st;
            long long have = fstat(fd, &st) == 0 ? (long long)st.st_size - w.data_off : (long long)size;
            if (have < 0) have = 0;
            w.data_bytes = (size == 0 || size == 0xFFFFFFFFu || (long long)size > have) ? have : (long long)size;
            return 0;
        }
        pos += 8 + (long long)size + (size & 1);
    }
    return -1;
}

// pread until `bytes` are in or the file ends; returns the bytes read (-1 on an I/O error)
long long pread_all(int fd, void* dst, size_t bytes, long long off) {
    size_t got = 0;
    while (got < bytes) {
        const ssize_t r = pread(fd, (char*)dst + got, bytes - got, off + (long long)got);
        if (r < 0) return -1;
        if (r == 0) break;
        got += (size_t)r;
    }
    return (long long)got;
}

// window `i`: frames [offset, offset + win) of channel 0 -> out[i * win ..]; 0 = ok, 1 = io, 2 = format,
// 3 = negative frame offset.  Frames past the end of the data (also of a file shorter than its header says)
// are zero-filled.
int read_one(const char* path, long long offset, int win, int16_t* out, int* rate) {
    if (offset < 0) return 3;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return 1;
    WavInfo w;
    int rc = 0;
    if (parse_wav(fd, w)) {
        rc = 2;
    } else if ((w.format != 1 && w.format != 0xFFFE) || !w.pcm_subformat || w.bits != 16 || w.channels < 1 ||
               w.channels > 8 || w.block_align != 2 * w.channels) {
        rc = 2;
    } else {
        *rate = w.rate;
        const long long frames = w.data_bytes / (2LL * w.channels);
        long long avail = frames - offset;
        if (avail < 0) avail = 0;
        if (avail > win) avail = win;
        const size_t bytes = (size_t)avail * 2 * w.channels;
        if (w.channels == 1) {
            const long long got = bytes ? pread_all(fd, out, bytes, w.data_off + offset * 2) : 0;
            if (got < 0) rc = 1; else avail = got / 2;
        } else {
            std::vector<int16_t> tmp((size_t)avail * w.channels);
            const long long got = bytes ? pread_all(fd, tmp.data(), bytes, w.data_off + offset * 2 * w.channels) : 0;
            if (got < 0) rc = 1; else avail = got / (2LL * w.channels);
            for (long long j = 0; j < avail; ++j) out[j] = tmp[(size_t)j * w.channels];
        }
        for (long long j = avail; j < win; ++j) out[j] = 0;
    }
    close(fd);
    return rc;
}

// one output sample per thread; kernel bank row of the sample's phase from global memory (L1 / L2 resident)
__global__ void __launch_bounds__(256)
pcm16_resample_kernel(const int16_t* __restrict__ pcm, int n_in, int orig, int nnew, int width,
                      const float* __restrict__ bank, float* __restrict__ out, int n_out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    const int16_t* x = pcm + (size_t)b * n_in;
    const float s = 1.0f / 32768.0f;
    if (!bank) {
        out[(size_t)b * n_out + i] = i < n_in ? (float)x[i] * s : 0.f;
        return;
    }
    const int j = i / nnew, ph = i - j * nnew;
    const int T = 2 * width + orig;
    const float* k = bank + (size_t)ph * T;
    const int base = j * orig - width;
    const int t0 = base < 0 ? -base : 0;
    const int t1 = base + T > n_in ? n_in - base : T;
    float acc = 0.f;
    for (int t = t0; t < t1; ++t) acc = fmaf(k[t], (float)x[base + t], acc);
    out[(size_t)b * n_out + i] = acc * s;
}

}  // namespace

extern "C" int afd_wav_read_windows(const char* const* paths, const long long* frame_offsets, int n, int win,
                                    int16_t* out, int* rates, int threads) {
    if (!paths || !frame_offsets || !out || !rates || n < 0 || win < 1) return afd::fail(AFD_ERR_ARG, "wav reader: bad argument");
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    if (threads > n) threads = n > 0 ? n : 1;
    std::atomic<int> next(0), first_bad(-1), bad_code(0);
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n) return;
            const int rc = read_one(paths[i], frame_offsets[i], win, out + (size_t)i * win, rates + i);
            if (rc) {
                int expect = -1;
                if (first_bad.compare_exchange_strong(expect, i)) bad_code.store(rc);
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    const int fb = first_bad.load();
    if (fb >= 0) {
        const int bc = bad_code.load();
        return afd::fail(bc == 2 ? AFD_ERR_UNSUPPORTED : AFD_ERR_ARG, "wav reader: %s: %s", paths[fb],
                         bc == 2 ? "not a 16-bit PCM WAV file" : bc == 3 ? "negative frame offset" : "cannot read");
    }
    return AFD_OK;
}

extern "C" int afd_pcm16_resample(const int16_t* pcm, int B, int n_in, int orig, int nnew, int width,
                                  const float* bank, float* out, int n_out, afd_stream_t stream) {
    if (!pcm || !out || B < 1 || n_in < 1 || n_out < 1 || orig < 1 || nnew < 1 || width < 0)
        return afd::fail(AFD_ERR_ARG, "pcm16 resample: bad argument");
    if (B > 65535) return afd::fail(AFD_ERR_UNSUPPORTED, "pcm16 resample: batch > 65535");
    if ((orig != nnew) != (bank != nullptr)) return afd::fail(AFD_ERR_ARG, "pcm16 resample: kernel bank and rates disagree");
    hipLaunchKernelGGL(pcm16_resample_kernel, dim3((n_out + 255) / 256, B), dim3(256), 0, static_cast<hipStream_t>(stream),
                       pcm, n_in, orig, nnew, width, bank, out, n_out);
    return afd::check_launch("pcm16_resample_kernel");
}
