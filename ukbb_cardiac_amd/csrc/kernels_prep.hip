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
#include <vector>

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

// ---- aortic z-score (common/image_utils.py:60-67) on the device ------------------------------------------
// normalise_intensity = percentile threshold -> ROI mask -> np.mean / np.std over image[roi] -> (image - mu) / (sigma + eps).
// The two reductions are float32 PAIRWISE sums in numpy (add.reduce over the contiguous compressed array): to be bit-identical
// the device reproduces numpy's summation tree, not just its value: (1) the ROI elements are compacted in numpy's order
// (row-major index order of the (X,Y,Z,T) array, whatever its strides), (2) every leaf of the tree (<= 128 consecutive
// elements, 8 interleaved accumulators, numpy's pairwise_sum) is summed by one thread in numpy's order, (3) the host adds the
// leaf sums up the tree in the same order, one tree per 8192-element buffer of numpy's reduction iterator, buffers in sequence.  No FMA contraction (-ffp-contract=off), IEEE division.
constexpr int CCH = 1024;                            // row-major indices per workgroup of the compaction kernels

__device__ __forceinline__ float roi_elem(const float *vol, long long i, int Y, int Z, int T, long long sx, long long sy, long long sz, long long st) {
    const int t = (int)(i % T); long long r = i / T;
    const int z = (int)(r % Z); r /= Z;
    const int y = (int)(r % Y); const long long x = r / Y;
    return vol[x * sx + y * sy + z * sz + t * st];
}

__global__ __launch_bounds__(256) void roi_count_kernel(const float *__restrict__ vol, long long n, int Y, int Z, int T, long long sx, long long sy,
                                                        long long sz, long long st, float thr, unsigned *__restrict__ counts) {
    __shared__ unsigned wsum[4];
    const long long i0 = (long long)blockIdx.x * CCH + threadIdx.x * 4;
    unsigned c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i0 + k < n && roi_elem(vol, i0 + k, Y, Z, T, sx, sy, sz, st) >= thr) ++c;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of nb block counts by one workgroup; offs[nb] = total
__global__ __launch_bounds__(1024) void roi_scan_kernel(const unsigned *__restrict__ counts, int nb, unsigned long long *__restrict__ offs) {
    __shared__ unsigned long long part[1024];
    const int per = (nb + 1023) / 1024, b0 = threadIdx.x * per, b1 = b0 + per < nb ? b0 + per : nb;
    unsigned long long s = 0;
    for (int b = b0; b < b1; ++b) s += counts[b];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int k = 0; k < 1024; ++k) { const unsigned long long v = part[k]; part[k] = run; run += v; }
        offs[nb] = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (int b = b0; b < b1; ++b) { offs[b] = run; run += counts[b]; }
}

__global__ __launch_bounds__(256) void roi_write_kernel(const float *__restrict__ vol, long long n, int Y, int Z, int T, long long sx, long long sy,
                                                        long long sz, long long st, float thr, const unsigned long long *__restrict__ offs,
                                                        float *__restrict__ out) {
    __shared__ unsigned wpre[4];
    const long long i0 = (long long)blockIdx.x * CCH + threadIdx.x * 4;
    float v[4];
    bool keep[4];
    unsigned c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        keep[k] = false;
        if (i0 + k < n) { v[k] = roi_elem(vol, i0 + k, Y, Z, T, sx, sy, sz, st); keep[k] = v[k] >= thr; }
        c += keep[k];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned incl = c;                                   // inclusive scan inside the wave
    for (int o = 1; o < 64; o <<= 1) { const unsigned u = __shfl_up(incl, o); if (lane >= o) incl += u; }
    if (lane == 63) wpre[wave] = incl;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < wave; ++w) base += wpre[w];
    unsigned long long pos = offs[blockIdx.x] + base + (incl - c);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (keep[k]) out[pos++] = v[k];
}

// one thread = one leaf of numpy's pairwise summation tree (loops_utils.h.src, pairwise_sum with PW_BLOCKSIZE = 128)
__global__ __launch_bounds__(128) void pairwise_leaf_kernel(const float *__restrict__ a, const unsigned long long *__restrict__ leaf_off,
                                                            const unsigned *__restrict__ leaf_len, int nleaf, int sq, float mean,
                                                            float *__restrict__ leaf_sum) {
    const int l = blockIdx.x * 128 + threadIdx.x;
    if (l >= nleaf) return;
    const float *p = a + leaf_off[l];
    const int n = (int)leaf_len[l];
    auto val = [&](int i) { float v = p[i]; if (sq) { const float d = v - mean; v = d * d; } return v; };
    float res;
    if (n < 8) {
        res = 0.f;
        for (int i = 0; i < n; ++i) res += val(i);
    } else {
        float r0 = val(0), r1 = val(1), r2 = val(2), r3 = val(3), r4 = val(4), r5 = val(5), r6 = val(6), r7 = val(7);
        int i = 8;
        for (; i < n - (n % 8); i += 8) {
            r0 += val(i); r1 += val(i + 1); r2 += val(i + 2); r3 += val(i + 3);
            r4 += val(i + 4); r5 += val(i + 5); r6 += val(i + 6); r7 += val(i + 7);
        }
        res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
        for (; i < n; ++i) res += val(i);
    }
    leaf_sum[l] = res;
}

// (v - mu) / den in float32, centred zero padding, (X,Y,Z,T) -> [T*Z][X2][Y2]: rescale_pack_kernel with the z-score arithmetic
__global__ __launch_bounds__(256) void zscore_pack_kernel(const float *__restrict__ vol, int X, int Y, int Z, int T,
                                                          long long sx, long long sy, long long sz, long long st,
                                                          float mu, float den, int X2, int Y2, int x_pre, int y_pre, float *__restrict__ out) {
    __shared__ float tile[32][33];
    const int tiles_x = (X2 + 31) / 32;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int b = blockIdx.y;
    const int t = b / Z, z = b - t * Z;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x2 = tx * 32 + lx, y2 = ty * 32 + ly + 8 * j;
        const int x = x2 - x_pre, y = y2 - y_pre;
        float r = 0.f;                                  // np.pad(..., 'constant') after the normalisation (deploy_network_ao.py:105-108)
        if (x >= 0 && x < X && y >= 0 && y < Y) r = __fdiv_rn(vol[x * sx + y * sy + z * sz + t * st] - mu, den);
        tile[ly + 8 * j][lx] = r;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x2 = tx * 32 + ly + 8 * j, y2 = ty * 32 + lx;
        if (x2 < X2 && y2 < Y2) out[((size_t)b * X2 + x2) * Y2 + y2] = tile[lx][ly + 8 * j];
    }
}

struct Scratch {                                        // grow-only device scratch of this thread, per device
    void *p = nullptr; size_t n = 0;
    void *get(size_t bytes) {
        if (bytes <= n) return p;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        n = bytes;
        return p;
    }
};
Scratch *prep_scratch(int which) {
    static thread_local Scratch s[MAX_DEVICES][2];
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEVICES) return nullptr;
    return &s[d][which];
}

// numpy's reduction iterator hands the inner loop at most one buffer of elements at a time (np.getbufsize(), 8192 by default)
// even for a contiguous array: add.reduce = ((0 + pw(a[0:8192])) + pw(a[8192:16384])) + ...
constexpr uint64_t NPY_BUFSIZE = 8192;

void pairwise_leaves(unsigned long long off, unsigned long long n, std::vector<unsigned long long> &offs, std::vector<unsigned> &lens) {
    if (n <= 128) { offs.push_back(off); lens.push_back((unsigned)n); return; }
    unsigned long long n2 = n / 2;
    n2 -= n2 % 8;
    pairwise_leaves(off, n2, offs, lens);
    pairwise_leaves(off + n2, n - n2, offs, lens);
}
float pairwise_combine(const float *leaf, size_t &idx, unsigned long long n) {
    if (n <= 128) return leaf[idx++];
    unsigned long long n2 = n / 2;
    n2 -= n2 % 8;
    const volatile float l = pairwise_combine(leaf, idx, n2);
    const volatile float r = pairwise_combine(leaf, idx, n - n2);
    return l + r;
}

// ---- synthetic subjects generated on the device ("generated on device from seed = subject id") ----
// Voxel i of subject `seed` = a * b / 2048 with a, b the low two 12-bit fields of splitmix64's finaliser applied to
// seed * 0x9E3779B97F4A7C15 + i: the product of two uniform integers is exact in float32 (< 2^24), so numpy reproduces the
// volume bit for bit from integer arithmetic alone (ukbb_cardiac_amd/synthetic_cohort.py), and its density -ln(x) has the
// heavy right tail that makes the 1 / 99 percentile clip of common/image_utils.py:72-74 matter.  One 16-byte store per thread.
__device__ __forceinline__ float synth_voxel(unsigned long long seed_mul, unsigned long long i) {
    unsigned long long z = seed_mul + i;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const unsigned a = (unsigned)z & 0xFFFu, b = (unsigned)(z >> 12) & 0xFFFu;
    return (float)(a * b) * (1.0f / 2048.0f);
}
__global__ __launch_bounds__(256) void synth_volume_kernel(unsigned long long seed_mul, unsigned long long n, float *__restrict__ out) {
    const unsigned long long stride = (unsigned long long)gridDim.x * 256ull * 4ull;
    for (unsigned long long i = ((unsigned long long)blockIdx.x * 256ull + threadIdx.x) * 4ull; i < n; i += stride) {
        if (i + 4 <= n) {
            float4 v;
            v.x = synth_voxel(seed_mul, i); v.y = synth_voxel(seed_mul, i + 1); v.z = synth_voxel(seed_mul, i + 2); v.w = synth_voxel(seed_mul, i + 3);
            *reinterpret_cast<float4 *>(out + i) = v;
        } else {
            for (unsigned long long j = i; j < n; ++j) out[j] = synth_voxel(seed_mul, j);
        }
    }
}

SelState *sel_scratch() {
    static thread_local SelState *p[MAX_DEVICES] = {nullptr};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEVICES) return nullptr;
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

int ukbb_fcn_roi_compact(const float *d_vol, int X, int Y, int Z, int T, int64_t sx, int64_t sy, int64_t sz, int64_t st, float thr,
                         float *d_out, uint64_t *n_host, void *stream) {
    if (!d_vol || !d_out || !n_host || X < 1 || Y < 1 || Z < 1 || T < 1) { set_error("roi_compact: bad argument"); return UKBB_EINVAL; }
    const long long n = (long long)X * Y * Z * T;
    const int nb = (int)((n + CCH - 1) / CCH);
    Scratch *sc = prep_scratch(0);
    void *buf = sc ? sc->get((size_t)nb * 4 + 8 + ((size_t)nb + 1) * 8) : nullptr;
    if (!buf) { set_error("roi_compact: scratch allocation failed"); return UKBB_ENOMEM; }
    unsigned *counts = static_cast<unsigned *>(buf);
    unsigned long long *offs = reinterpret_cast<unsigned long long *>(static_cast<char *>(buf) + (((size_t)nb * 4 + 7) / 8) * 8);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(roi_count_kernel, dim3(nb), dim3(256), 0, s, d_vol, n, Y, Z, T, (long long)sx, (long long)sy, (long long)sz, (long long)st, thr, counts);
    hipLaunchKernelGGL(roi_scan_kernel, dim3(1), dim3(1024), 0, s, counts, nb, offs);
    hipLaunchKernelGGL(roi_write_kernel, dim3(nb), dim3(256), 0, s, d_vol, n, Y, Z, T, (long long)sx, (long long)sy, (long long)sz, (long long)st, thr, offs, d_out);
    unsigned long long total = 0;
    if (hipMemcpyAsync(&total, offs + nb, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        set_error("roi_compact: device error: %s", hipGetErrorString(hipGetLastError()));
        return UKBB_EDEVICE;
    }
    *n_host = total;
    return UKBB_OK;
}

int ukbb_fcn_pairwise_sum(const float *d_a, uint64_t n, int squared_dev, float mean, float *sum_host, void *stream) {
    if (!d_a || !sum_host) { set_error("pairwise_sum: NULL argument"); return UKBB_EINVAL; }
    if (n == 0) { *sum_host = 0.f; return UKBB_OK; }
    std::vector<unsigned long long> offs;
    std::vector<unsigned> lens;
    offs.reserve((size_t)(n / 64) + 2); lens.reserve((size_t)(n / 64) + 2);
    for (uint64_t c = 0; c < n; c += NPY_BUFSIZE) pairwise_leaves(c, n - c < NPY_BUFSIZE ? n - c : NPY_BUFSIZE, offs, lens);
    const size_t nl = offs.size();
    Scratch *sc = prep_scratch(1);
    char *buf = sc ? static_cast<char *>(sc->get(nl * 16)) : nullptr;
    if (!buf) { set_error("pairwise_sum: scratch allocation failed"); return UKBB_ENOMEM; }
    unsigned long long *d_off = reinterpret_cast<unsigned long long *>(buf);
    unsigned *d_len = reinterpret_cast<unsigned *>(buf + nl * 8);
    float *d_sum = reinterpret_cast<float *>(buf + nl * 12);
    hipStream_t s = (hipStream_t)stream;
    std::vector<float> leaf(nl);
    if (hipMemcpyAsync(d_off, offs.data(), nl * 8, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(d_len, lens.data(), nl * 4, hipMemcpyHostToDevice, s) != hipSuccess) { set_error("pairwise_sum: H2D failed"); return UKBB_EDEVICE; }
    hipLaunchKernelGGL(pairwise_leaf_kernel, dim3((unsigned)((nl + 127) / 128)), dim3(128), 0, s, d_a, d_off, d_len, (int)nl, squared_dev, mean, d_sum);
    if (hipMemcpyAsync(leaf.data(), d_sum, nl * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        set_error("pairwise_sum: device error: %s", hipGetErrorString(hipGetLastError()));
        return UKBB_EDEVICE;
    }
    size_t idx = 0;
    volatile float acc = 0.f;                           // add.reduce starts from the identity and adds one pairwise sum per buffer
    for (uint64_t c = 0; c < n; c += NPY_BUFSIZE) {
        const volatile float tree = pairwise_combine(leaf.data(), idx, n - c < NPY_BUFSIZE ? n - c : NPY_BUFSIZE);
        acc = acc + tree;
    }
    *sum_host = acc;
    return UKBB_OK;
}

// ---- shader clock under load (measurement only) ----
static __global__ void clock_probe_kernel(unsigned long long *out, unsigned long long spin_ticks) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < spin_ticks) { __builtin_amdgcn_s_sleep(32); r1 = __builtin_amdgcn_s_memrealtime(); }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

int ukbb_fcn_clock_probe(int device, void *stream, int spin_us, double *mhz) {
    if (!mhz || spin_us < 1 || spin_us > 1000000) { set_error("clock_probe: bad arguments"); return UKBB_EINVAL; }
    unsigned long long *d = nullptr, h[2] = {0, 0};
    hipStream_t s = (hipStream_t)stream;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&d), 16) != hipSuccess) {
        set_error("clock_probe: no HIP device %d (there is no CPU fallback)", device);
        return UKBB_EDEVICE;
    }
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, s, d, (unsigned long long)spin_us * 100ull);   // s_memrealtime: 100 MHz
    const bool ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, s) == hipSuccess &&
                    hipStreamSynchronize(s) == hipSuccess;
    (void)hipFree(d);
    if (!ok || h[1] == 0) { set_error("clock_probe: device error"); return UKBB_EDEVICE; }
    *mhz = (double)h[0] / (double)h[1] * 100.0;
    return UKBB_OK;
}

int ukbb_fcn_zscore_pack(const float *d_vol, int X, int Y, int Z, int T, int64_t sx, int64_t sy, int64_t sz, int64_t st,
                         float mu, float den, int X2, int Y2, int x_pre, int y_pre, float *d_batch, void *stream) {
    if (!d_vol || !d_batch || X < 1 || Y < 1 || Z < 1 || T < 1 || x_pre < 0 || y_pre < 0 || X2 < X + x_pre || Y2 < Y + y_pre ||
        (long long)Z * T > 65535) {
        set_error("zscore_pack: bad shape X=%d Y=%d Z=%d T=%d X2=%d Y2=%d pre=(%d,%d) (Z*T <= 65535)", X, Y, Z, T, X2, Y2, x_pre, y_pre);
        return UKBB_EINVAL;
    }
    dim3 grid((unsigned)(((X2 + 31) / 32) * ((Y2 + 31) / 32)), (unsigned)(Z * T));
    hipLaunchKernelGGL(zscore_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_vol, X, Y, Z, T, (long long)sx, (long long)sy,
                       (long long)sz, (long long)st, mu, den, X2, Y2, x_pre, y_pre, d_batch);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("zscore_pack: launch failed: %s", hipGetErrorString(e)); return UKBB_EDEVICE; }
    return UKBB_OK;
}

// ---- debugging aid (r06): fill every CU's LDS with a bit pattern.  LDS is not cleared between workgroups; a kernel that reads LDS it never wrote
//      normally sees the finite leftovers of the previous kernel of its own plan -- and NaN patterns when another stream's kernel ran there ----
static __global__ __launch_bounds__(256) void poison_lds_kernel(unsigned pattern, int dwords, unsigned *sink) {
    extern __shared__ unsigned pl_lds[];
    for (int i = threadIdx.x; i < dwords; i += 256) pl_lds[i] = pattern;
    __syncthreads();
    if (sink && pl_lds[(threadIdx.x * 97) % dwords] != pattern) sink[0] = 1;     // keeps the stores alive
}
int ukbb_fcn_debug_poison_lds(uint32_t pattern, void *stream) {
    constexpr int bytes = 160 * 1024;
    static OncePerDevice ok;
    if (allow_dynamic_lds(ok, reinterpret_cast<const void *>(poison_lds_kernel), bytes) != hipSuccess) { set_error("poison_lds: cannot get 160 KB of LDS"); return UKBB_EDEVICE; }
    hipLaunchKernelGGL(poison_lds_kernel, dim3((unsigned)(device_cu_count() * 4)), dim3(256), bytes, (hipStream_t)stream, pattern, bytes / 4, (unsigned *)nullptr);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("poison_lds: launch failed: %s", hipGetErrorString(e)); return UKBB_EDEVICE; }
    return UKBB_OK;
}

// ---- debugging aid (r06): a launch whose every workgroup does nothing but a system-scope release + acquire fence (L2 write-back and invalidate of
//      the XCD it runs on): placed between two dependent launches it takes the place of whatever the runtime does, or does not do, at that boundary ----
static __global__ __launch_bounds__(64) void fence_kernel(unsigned *sink) {
    __atomic_thread_fence(__ATOMIC_SEQ_CST);            // hipcc: system scope by default -> buffer_wbl2 sc0 sc1 + s_waitcnt + buffer_inv sc0 sc1
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");       // system scope
    if (sink && threadIdx.x == 12345) sink[0] = 1;
}
int ukbb_fcn_debug_fence_kernel(void *stream) {
    hipLaunchKernelGGL(fence_kernel, dim3((unsigned)(device_cu_count() * 2)), dim3(64), 0, (hipStream_t)stream, (unsigned *)nullptr);
    return hipGetLastError() == hipSuccess ? UKBB_OK : UKBB_EDEVICE;
}

// the same with a chosen LDS footprint per block (co-resident with other kernels' workgroups: a kernel that uses LDS beyond what it asked for reads this)
int ukbb_fcn_debug_poison_lds_sized(uint32_t pattern, int bytes, int blocks, void *stream) {
    if (bytes < 1024 || bytes > 160 * 1024 || blocks < 1) return UKBB_EINVAL;
    static OncePerDevice ok;
    if (allow_dynamic_lds(ok, reinterpret_cast<const void *>(poison_lds_kernel), 160 * 1024) != hipSuccess) return UKBB_EDEVICE;
    hipLaunchKernelGGL(poison_lds_kernel, dim3((unsigned)blocks), dim3(256), bytes, (hipStream_t)stream, pattern, bytes / 4, (unsigned *)nullptr);
    return hipGetLastError() == hipSuccess ? UKBB_OK : UKBB_EDEVICE;
}

int ukbb_fcn_synth_volume(uint64_t seed, size_t n, float *d_out, void *stream) {
    if (!d_out || n == 0 || (reinterpret_cast<uintptr_t>(d_out) & 15)) { set_error("synth_volume: bad argument (n > 0, 16-byte aligned output)"); return UKBB_EINVAL; }
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(synth_volume_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (unsigned long long)seed * 0x9E3779B97F4A7C15ull,
                       (unsigned long long)n, d_out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("synth_volume: launch failed: %s (there is no CPU fallback)", hipGetErrorString(e)); return UKBB_EDEVICE; }
    return UKBB_OK;
}

}  // extern "C"
