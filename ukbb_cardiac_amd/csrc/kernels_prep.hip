// Device-side pre/post-processing of the short-axis / long-axis deploy loop:
//
//   ukbb_fcn_select_kth     exact order statistics of a float32 volume (4-pass radix select) -- replaces the
//                           full sort inside np.percentile(image, (1, 99)), common/image_utils.py:72
//   ukbb_fcn_rescale_pack   clip + (v - lo)/(hi - lo) + centred zero padding + (X,Y,Z,T) -> [T*Z][X2][Y2]
//                           -- common/image_utils.py:73-76 and common/deploy_network.py:97-107
//   ukbb_fcn_unpack_labels  label batch -> cropped (X,Y,Z,T) volume + per-frame class counts
//                           -- common/deploy_network.py:114-116 and the counting behind :125-130
//
// All three are HBM-bound byte/float shuffles: coalesced 16-byte loads, 32x32 LDS tile transposes between
// the volume's x-fastest order and the network's y-fastest order, no arithmetic to speak of.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/ukbb_fcn.h"
#include "kernels.h"

namespace ukbb {

namespace {

constexpr int MAXR = 8;                 // ranks per call
struct SelState {
    unsigned prefix[MAXR];              // key bits fixed so far (high bytes)
    unsigned long long rank[MAXR];      // rank remaining inside the current prefix class
    unsigned hist[MAXR][256];
};

// order-preserving map float32 -> uint32 (negative: all bits flipped, non-negative: sign bit set)
__device__ __forceinline__ unsigned fkey(float x) {
    const unsigned b = __float_as_uint(x);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __host__ inline float fkey_inv(unsigned k) {
    const unsigned b = k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    float f;
    memcpy(&f, &b, 4);
    return f;
}

// One radix pass: histogram of byte `shift/8` over the elements whose higher bytes equal prefix[r].
// LDS-privatised; runs of equal bins (MR intensities share their top byte) are counted in a register and
// flushed once, which removes the same-address atomic contention of the first pass.
template <int NR>
__global__ __launch_bounds__(256) void sel_hist_kernel(const float *__restrict__ data, size_t n, SelState *st, int shift) {
    __shared__ unsigned h[NR][256];
    for (int i = threadIdx.x; i < NR * 256; i += 256) (&h[0][0])[i] = 0;
    unsigned prefix[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) prefix[r] = st->prefix[r];
    const unsigned himask = shift == 24 ? 0u : 0xFFFFFFFFu << (shift + 8);
    __syncthreads();
    int last_bin[NR];
    unsigned run[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) { last_bin[r] = 0; run[r] = 0; }
    auto feed = [&](float x) {
        const unsigned k = fkey(x);
        const int bin = (k >> shift) & 255;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (((k ^ prefix[r]) & himask) == 0) {
                if (bin == last_bin[r]) ++run[r];
                else { if (run[r]) atomicAdd(&h[r][last_bin[r]], run[r]); last_bin[r] = bin; run[r] = 1; }
            }
        }
    };
    const size_t n4 = n / 4;
    const float4 *d4 = reinterpret_cast<const float4 *>(data);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = d4[i];
        feed(v.x); feed(v.y); feed(v.z); feed(v.w);
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) feed(data[i]);
#pragma unroll
    for (int r = 0; r < NR; ++r)
        if (run[r]) atomicAdd(&h[r][last_bin[r]], run[r]);
    __syncthreads();
    for (int i = threadIdx.x; i < NR * 256; i += 256) {
        const unsigned c = (&h[0][0])[i];
        if (c) atomicAdd(&st->hist[0][0] + i, c);
    }
}

// Per rank: find the bin holding the rank, extend the prefix, reduce the rank, clear the histogram.
__global__ void sel_pick_kernel(SelState *st, int nr, int shift) {
    const int r = threadIdx.x;
    if (r >= nr) return;
    unsigned long long k = st->rank[r], cum = 0;
    int bin = 255;
    for (int b = 0; b < 256; ++b) {
        const unsigned c = st->hist[r][b];
        if (k < cum + c) { bin = b; break; }
        cum += c;
    }
    st->prefix[r] |= (unsigned)bin << shift;
    st->rank[r] = k - cum;
    for (int b = 0; b < 256; ++b) st->hist[r][b] = 0;
}

// ---- rescale + pad + transpose ----------------------------------------------------------------------------
// One workgroup: a 32x32 (x, y) tile of one (z, t) slice.  Reads run along the volume's fastest axis when
// sx == 1 (NIfTI order), writes run along y2 (the network's fastest axis).
__global__ __launch_bounds__(256) void rescale_pack_kernel(const float *__restrict__ vol, int X, int Y, int Z, int T,
                                                           long long sx, long long sy, long long sz, long long st,
                                                           double lo, double hi, int X2, int Y2, int x_pre, int y_pre,
                                                           float *__restrict__ out) {
    __shared__ float tile[32][33];
    const int tiles_x = (X2 + 31) / 32;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int b = blockIdx.y;                           // b = t * Z + z
    const int t = b / Z, z = b - t * Z;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const float flo = (float)lo, fhi = (float)hi;       // what the in-place clip stores into the float32 array
    const double inv_den = hi - lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x2 = tx * 32 + lx, y2 = ty * 32 + ly + 8 * j;
        const int x = x2 - x_pre, y = y2 - y_pre;
        float r = 0.f;                                  // np.pad(..., 'constant') after the rescale
        if (x >= 0 && x < X && y >= 0 && y < Y) {
            float v = vol[x * sx + y * sy + z * sz + t * st];
            if ((double)v < lo) v = flo;                // image[image < val_l] = val_l   (image_utils.py:73)
            if ((double)v > hi) v = fhi;                // image[image > val_h] = val_h   (:74)
            r = (float)(((double)v - lo) / inv_den);    // (:75-76), float64 arithmetic, float32 at deploy_network.py:105
        }
        tile[ly + 8 * j][lx] = r;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x2 = tx * 32 + ly + 8 * j, y2 = ty * 32 + lx;
        if (x2 < X2 && y2 < Y2) out[((size_t)b * X2 + x2) * Y2 + y2] = tile[lx][ly + 8 * j];
    }
}

// ---- labels back to the volume + class counts -----------------------------------------------------------
__global__ __launch_bounds__(256) void unpack_labels_kernel(const int *__restrict__ pred, int X, int Y, int Z, int T,
                                                            int X2, int Y2, int x_pre, int y_pre, int n_class,
                                                            unsigned char *__restrict__ vol, unsigned long long *counts) {
    __shared__ unsigned char tile[32][33];
    __shared__ unsigned cnt[16];
    if (threadIdx.x < 16) cnt[threadIdx.x] = 0;
    const int tiles_x = (X + 31) / 32;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int b = blockIdx.y;
    const int t = b / Z, z = b - t * Z;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {                       // read along y2 (fastest in the batch)
        const int x = tx * 32 + ly + 8 * j, y = ty * 32 + lx;
        int v = 0;
        if (x < X && y < Y) {
            v = pred[((size_t)b * X2 + x + x_pre) * Y2 + y + y_pre];
            if (v >= 0 && v < n_class && v < 16) atomicAdd(&cnt[v], 1u);
        }
        tile[ly + 8 * j][lx] = (unsigned char)v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {                       // write along x (fastest in the NIfTI volume)
        const int x = tx * 32 + lx, y = ty * 32 + ly + 8 * j;
        if (x < X && y < Y) vol[x + (size_t)X * (y + (size_t)Y * (z + (size_t)Z * t))] = tile[lx][ly + 8 * j];
    }
    if (threadIdx.x < n_class && threadIdx.x < 16 && cnt[threadIdx.x])
        atomicAdd(&counts[(size_t)t * n_class + threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
}

SelState *sel_scratch() {
    static thread_local SelState *p[16] = {nullptr};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 16) return nullptr;
    if (!p[d] && hipMalloc(&p[d], sizeof(SelState)) != hipSuccess) p[d] = nullptr;
    return p[d];
}

}  // namespace
}  // namespace ukbb

using namespace ukbb;

extern "C" {

int ukbb_fcn_select_kth(const float *d_data, size_t n, const uint64_t *ranks, int nranks, float *out_host, void *stream) {
    if (!d_data || !ranks || !out_host || n == 0 || nranks < 1 || nranks > MAXR) {
        set_error("select_kth: bad argument (1..8 ranks, n > 0)");
        return UKBB_EINVAL;
    }
    for (int r = 0; r < nranks; ++r)
        if (ranks[r] >= n) { set_error("select_kth: rank outside [0, n)"); return UKBB_EINVAL; }
    SelState *st = sel_scratch();
    if (!st) { set_error("select_kth: no HIP device / scratch allocation failed (there is no CPU fallback)"); return UKBB_EDEVICE; }
    hipStream_t s = (hipStream_t)stream;
    SelState init;
    memset(&init, 0, sizeof init);
    for (int r = 0; r < MAXR; ++r) init.rank[r] = ranks[r < nranks ? r : nranks - 1];
    if (hipMemcpyAsync(st, &init, sizeof init, hipMemcpyHostToDevice, s) != hipSuccess) { set_error("select_kth: H2D failed"); return UKBB_EDEVICE; }
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (nranks <= 2) hipLaunchKernelGGL(sel_hist_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, d_data, n, st, shift);
        else if (nranks <= 4) hipLaunchKernelGGL(sel_hist_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, d_data, n, st, shift);
        else hipLaunchKernelGGL(sel_hist_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, s, d_data, n, st, shift);
        hipLaunchKernelGGL(sel_pick_kernel, dim3(1), dim3(64), 0, s, st, MAXR, shift);
    }
    unsigned keys[MAXR];
    if (hipMemcpyAsync(keys, st->prefix, sizeof keys, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        set_error("select_kth: device error: %s", hipGetErrorString(hipGetLastError()));
        return UKBB_EDEVICE;
    }
    for (int r = 0; r < nranks; ++r) out_host[r] = fkey_inv(keys[r]);
    return UKBB_OK;
}

int ukbb_fcn_rescale_pack(const float *d_vol, int X, int Y, int Z, int T, int64_t sx, int64_t sy, int64_t sz, int64_t st,
                          double lo, double hi, int X2, int Y2, int x_pre, int y_pre, float *d_batch, void *stream) {
    if (!d_vol || !d_batch || X < 1 || Y < 1 || Z < 1 || T < 1 || x_pre < 0 || y_pre < 0 || X2 < X + x_pre || Y2 < Y + y_pre ||
        (long long)Z * T > 65535) {
        set_error("rescale_pack: bad shape X=%d Y=%d Z=%d T=%d X2=%d Y2=%d pre=(%d,%d) (Z*T <= 65535)", X, Y, Z, T, X2, Y2, x_pre, y_pre);
        return UKBB_EINVAL;
    }
    dim3 grid((unsigned)(((X2 + 31) / 32) * ((Y2 + 31) / 32)), (unsigned)(Z * T));
    hipLaunchKernelGGL(rescale_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_vol, X, Y, Z, T, (long long)sx, (long long)sy,
                       (long long)sz, (long long)st, lo, hi, X2, Y2, x_pre, y_pre, d_batch);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("rescale_pack: launch failed: %s", hipGetErrorString(e)); return UKBB_EDEVICE; }
    return UKBB_OK;
}

int ukbb_fcn_unpack_labels(const int32_t *d_pred, int X, int Y, int Z, int T, int X2, int Y2, int x_pre, int y_pre, int n_class,
                           uint8_t *d_vol, uint64_t *d_counts, void *stream) {
    if (!d_pred || !d_vol || !d_counts || X < 1 || Y < 1 || Z < 1 || T < 1 || x_pre < 0 || y_pre < 0 || X2 < X + x_pre ||
        Y2 < Y + y_pre || n_class < 1 || n_class > 16 || (long long)Z * T > 65535) {
        set_error("unpack_labels: bad argument (n_class 1..16, Z*T <= 65535)");
        return UKBB_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(d_counts, 0, sizeof(uint64_t) * (size_t)T * n_class, s) != hipSuccess) { set_error("unpack_labels: memset failed"); return UKBB_EDEVICE; }
    dim3 grid((unsigned)(((X + 31) / 32) * ((Y + 31) / 32)), (unsigned)(Z * T));
    hipLaunchKernelGGL(unpack_labels_kernel, grid, dim3(256), 0, s, d_pred, X, Y, Z, T, X2, Y2, x_pre, y_pre, n_class, d_vol,
                       reinterpret_cast<unsigned long long *>(d_counts));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("unpack_labels: launch failed: %s", hipGetErrorString(e)); return UKBB_EDEVICE; }
    return UKBB_OK;
}

}  // extern "C"
