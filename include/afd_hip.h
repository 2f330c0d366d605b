/* libafd_hip -- C ABI of the MI355X (gfx950) hot path of audiofakedetect.
 *
 * The reference (gan-police/audiodeepfake-detection) is pure Python; the device work of
 * its hot path is launched implicitly by ptwt / torchaudio / torch.nn.  This header is the
 * boundary a maintainer binds (ctypes) to replace those launches.
 * Each entry point cites the reference call site it replaces.
 *
 * Conventions
 *   - every function returns 0 on success, a negative AFD_ERR_* code otherwise; the
 *     message is available from afd_last_error() (thread local);
 *   - pointers marked [dev] are device pointers owned by the caller (PyTorch); the library
 *     never allocates, frees or retains them; pointers marked [host] are host memory read
 *     during the call only;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and the call
 *     returns without synchronising;
 *   - tensors are dense fp32 unless stated; layouts are given per function.
 */
#ifndef AFD_HIP_H
#define AFD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFD_ERR_ARG (-1)
#define AFD_ERR_HIP (-2)
#define AFD_ERR_UNSUPPORTED (-3)
#define AFD_ERR_WORKSPACE (-4)

typedef void* afd_stream_t;

const char* afd_last_error(void);
int afd_version(void);

/* ------------------------------------------------------------------------------------
 * Wavelet-packet front end.
 * Replaces: compute_pytorch_packet_representation + Packets.forward
 *           (reference src/audiofakedetect/wavelet_math.py:167-263), i.e. the
 *           ptwt.WaveletPacket tree (:182), get_level (:185), stack (:206),
 *           log(|x|^power + eps) (:208-209), sign channel (:211-213), the final permute
 *           (:263), and torchvision Normalize with scalar statistics (:380-382).
 * ---------------------------------------------------------------------------------- */
#define AFD_WPT_LOG 1u  /* log(|x|^power + eps)                                       */
#define AFD_WPT_SIGN 2u /* second channel = +1/-1 sign pattern (loss_less), needs LOG */
#define AFD_WPT_NORM 4u /* (v - mean) / std on every channel                          */

/* Node length at `level` for frames of N samples and L-tap filters (reflect mode). */
int afd_wpt_out_len(int N, int L, int level);

/* Bytes of device workspace afd_wpt_forward needs (may be 0). */
size_t afd_wpt_workspace_bytes(int B, int N, int L, int level);

/* x   [dev]  [B][N] frames
 * dec_lo, dec_hi [host] L decomposition taps (pywt convention)
 * out [dev]  [B][C][T][P], P = 2^level packets in frequency (Gray-code) order fastest,
 *            T = afd_wpt_out_len, C = 2 with AFD_WPT_SIGN else 1.  This is the memory
 *            order of the reference's returned view (logical [B][C][P][T]).
 */
int afd_wpt_forward(const float* x, int B, int N, const float* dec_lo, const float* dec_hi,
                    int L, int level, unsigned flags, float power, float eps, float mean,
                    float std, float* out, void* ws, size_t ws_bytes, afd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AFD_HIP_H */
