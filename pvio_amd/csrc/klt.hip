// klt.hip -- device-resident image front end: CLAHE, 4-level pyramid, Scharr derivatives, pyramidal Lucas-Kanade.
//
// Replaces what the reference gets from OpenCV in
//   OpenCvImage::preprocess       pvio-extra/src/pvio/extra/opencv_image.cpp:138-145  (CLAHE(6.0, 8x8) + buildOpticalFlowPyramid(21x21, 3, true))
//   OpenCvImage::track_keypoints  opencv_image.cpp:88-109                             (calcOpticalFlowPyrLK + 20-px border kill)
// The F-matrix RANSAC of :121-129 stays on the host adapter.
//
// HBM layout: every pyramid level is stored PADDED by kPad pixels on each side -- pixels with the BORDER_REFLECT_101
// values OpenCV's pyramid carries, derivatives (int16 x 2, interleaved) with a zero border -- so that the LK kernel
// never clamps or branches on image boundaries.  Row pitch is a multiple of 64 bytes.
// LK kernel: one 64-wide wavefront per track; the 21x21 template patch and its derivative patch are built once per
// level with 14-bit fixed-point bilinear weights (exactly OpenCV's integer arithmetic) and staged in LDS as int16;
// every iteration each lane resamples ~7 window pixels of the next image and the 2x1 mismatch vector is a wave
// reduction.  The float accumulators are reduced in a fixed butterfly order (deterministic, but not the scalar
// left-to-right order of the CPU restatement: positions agree to ~1e-5 px, see tests/test_gpu_klt.py).
#include "klt.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "../../include/pvio_hip.h"
#include "pv_fundamental.h"

namespace pvklt {

constexpr int kLevels = 4;   // maxLevel 3
constexpr int kWin = 21;
constexpr int kPad = 32;     // >= kWin + 1
constexpr int kWinPix = kWin * kWin;

struct LevelDesc {
    int w, h, pitch;         // pitch in pixels (u8) / in int16 pairs (derivative)
    uint8_t *img;            // (h + 2 kPad) x pitch
    int16_t *drv;            // (h + 2 kPad) x pitch x 2
};
struct Image {
    int w, h, n_levels;
    LevelDesc lv[kLevels];
    uint8_t *raw;            // unpadded upload buffer (w x h)
    uint8_t *lut;            // CLAHE LUTs [64][256]
    void *slab;              // one allocation for everything above
    size_t slab_bytes;
};

struct TrackArgs {
    int n, n_levels;
    LevelDesc prev[kLevels], next[kLevels];
    const float *prev_xy;
    float *next_xy;
    uint8_t *status;
};

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

// ---- CLAHE ----------------------------------------------------------------------------------------------------------
// one workgroup per tile: histogram (LDS atomics), clip + redistribute, cumulative LUT
__global__ void __launch_bounds__(256) k_clahe_lut(const uint8_t *src, int w, int h, int tiles_x, int tiles_y, int tw, int th, int clip, uint8_t *lut) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    __shared__ int hist[256];
    __shared__ int scan[256];
    __shared__ int s_clipped;
    const int tid = threadIdx.x, tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    hist[tid] = 0;
    if (tid == 0) s_clipped = 0;
    __syncthreads();
    for (int i = tid; i < tw * th; i += 256) {
        const int y = ty * th + i / tw, x = tx * tw + i % tw;
        atomicAdd(&hist[src[(size_t)reflect101(y, h) * w + reflect101(x, w)]], 1); // padded region = BORDER_REFLECT_101
    }
    __syncthreads();
    int v = hist[tid];
    if (clip > 0) {
        if (v > clip) {
            atomicAdd(&s_clipped, v - clip);
            v = clip;
        }
        __syncthreads();
        const int clipped = s_clipped, batch = clipped / 256, residual = clipped - batch * 256;
        v += batch;
        if (residual != 0) {
            int step = 256 / residual;
            if (step < 1) step = 1;
            if (tid % step == 0 && tid / step < residual) v += 1; // bins 0, step, 2 step, ... (residual of them)
        }
    }
    scan[tid] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) { // inclusive scan
        const int add = tid >= off ? scan[tid - off] : 0;
        __syncthreads();
        scan[tid] += add;
        __syncthreads();
    }
    const float lut_scale = 255.0f / (float)(tw * th);
    int r = (int)rintf((float)scan[tid] * lut_scale);
    lut[(size_t)blockIdx.x * 256 + tid] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
}

__global__ void k_clahe_apply(const uint8_t *src, int w, int h, int tiles_x, int tiles_y, int tw, int th, const uint8_t *lut, LevelDesc out) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
    const float tyf = (float)y * inv_th - 0.5f, txf = (float)x * inv_tw - 0.5f;
    int ty1 = (int)floorf(tyf), tx1 = (int)floorf(txf);
    int ty2 = ty1 + 1, tx2 = tx1 + 1;
    const float ya = tyf - (float)ty1, ya1 = 1.0f - ya, xa = txf - (float)tx1, xa1 = 1.0f - xa;
    ty1 = max(ty1, 0), tx1 = max(tx1, 0), ty2 = min(ty2, tiles_y - 1), tx2 = min(tx2, tiles_x - 1);
    const int v = src[(size_t)y * w + x];
    const uint8_t *p1 = lut + (size_t)(ty1 * tiles_x) * 256, *p2 = lut + (size_t)(ty2 * tiles_x) * 256;
    const int i1 = tx1 * 256 + v, i2 = tx2 * 256 + v;
    const float res = ((float)p1[i1] * xa1 + (float)p1[i2] * xa) * ya1 + ((float)p2[i1] * xa1 + (float)p2[i2] * xa) * ya;
    const int r = (int)rintf(res);
    out.img[(size_t)(y + kPad) * out.pitch + x + kPad] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
}

__global__ void k_copy_level0(const uint8_t *src, int w, int h, LevelDesc out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x < w && y < h) out.img[(size_t)(y + kPad) * out.pitch + x + kPad] = src[(size_t)y * w + x];
}

// ---- undistortion: cv::remap(INTER_LINEAR, BORDER_CONSTANT) with fixed-point maps --------------------------------------
// Replaces cv::undistort (pvio-pc/src/euroc_dataset_reader.cpp:72-75) and ImageUndistorter::undistort_image
// (pvio-extra/include/pvio/extra/image_undistorter.h:44-46).  Maps are OpenCV's CV_16SC2 + CV_16UC1 pair: integer source
// position and a 5+5-bit fraction index (INTER_BITS = 5); weights are the INTER_LINEAR table entries at 15 bits
// ((32-fx)(32-fy)*32, ...; the fraction (0,0) entry is {32767, 0, 0, 1}: saturate_cast<short>(32768) and the +1 that makes the
// table row sum to 1 << 15 again -- the result is the same pixel), result = (sum + (1 << 14)) >> 15; taps outside the
// source image are the constant border 0.  Streaming: 5 bytes of map + 1 byte out per pixel, the 2x2 source taps hit in L2.
struct Undistort {
    int w, h;          // destination (= map) size
    int16_t *xy;       // [h][w][2]
    uint16_t *frac;    // [h][w]
};
__global__ void __launch_bounds__(256) k_remap(const uint8_t *src, int sw, int sh, const int16_t *__restrict__ mxy, const uint16_t *__restrict__ mfr,
                                               int w, int h, uint8_t *__restrict__ dst) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const size_t i = (size_t)y * w + x;
    const int sx = mxy[2 * i], sy = mxy[2 * i + 1];
    const int f = mfr[i] & 1023, fx = f & 31, fy = f >> 5;
    int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    if (f == 0) w00 = 32767, w11 = 1;
    const bool x0 = sx >= 0 && sx < sw, x1 = sx + 1 >= 0 && sx + 1 < sw, y0 = sy >= 0 && sy < sh, y1 = sy + 1 >= 0 && sy + 1 < sh;
    const int p00 = (x0 && y0) ? src[(size_t)sy * sw + sx] : 0, p01 = (x1 && y0) ? src[(size_t)sy * sw + sx + 1] : 0;
    const int p10 = (x0 && y1) ? src[(size_t)(sy + 1) * sw + sx] : 0, p11 = (x1 && y1) ? src[(size_t)(sy + 1) * sw + sx + 1] : 0;
    dst[i] = (uint8_t)((p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15);
}

// BORDER_REFLECT_101 ring of kPad pixels around the interior
__global__ void k_border(LevelDesc lv) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x - kPad, y = (int)blockIdx.y - kPad;
    if (x >= lv.w + kPad || y >= lv.h + kPad) return;
    if (x >= 0 && x < lv.w && y >= 0 && y < lv.h) return;
    lv.img[(size_t)(y + kPad) * lv.pitch + x + kPad] = lv.img[(size_t)(reflect101(y, lv.h) + kPad) * lv.pitch + reflect101(x, lv.w) + kPad];
}

// calcSharrDeriv: dx = [3 10 3]^T (x) [-1 0 1], dy = [-1 0 1]^T (x) [3 10 3]; the reflected border of the image supplies
// exactly the BORDER_REFLECT_101 neighbours the reference uses
__global__ void k_scharr(LevelDesc lv) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= lv.w || y >= lv.h) return;
    const uint8_t *c = lv.img + (size_t)(y + kPad) * lv.pitch + x + kPad;
    const int p = lv.pitch;
    const int a00 = c[-p - 1], a01 = c[-p], a02 = c[-p + 1], a10 = c[-1], a12 = c[1], a20 = c[p - 1], a21 = c[p], a22 = c[p + 1];
    const int dx = ((a02 + a22) * 3 + a12 * 10) - ((a00 + a20) * 3 + a10 * 10);
    const int dy = ((a20 + a22) * 3 + a21 * 10) - ((a00 + a02) * 3 + a01 * 10);
    int16_t *d = lv.drv + 2 * ((size_t)(y + kPad) * lv.pitch + x + kPad);
    d[0] = (int16_t)dx, d[1] = (int16_t)dy;
}

// cv::pyrDown: [1 4 6 4 1] x [1 4 6 4 1], (sum + 128) >> 8
__global__ void k_pyr_down(LevelDesc src, LevelDesc dst) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dst.w || y >= dst.h) return;
    const uint8_t *c = src.img + (size_t)(2 * y + kPad) * src.pitch + 2 * x + kPad;
    int sum = 0;
#pragma unroll
    for (int j = -2; j <= 2; ++j) {
        const uint8_t *r = c + (ptrdiff_t)j * src.pitch;
        const int rs = r[-2] + 4 * r[-1] + 6 * r[0] + 4 * r[1] + r[2];
        sum += (j == 0 ? 6 : (j == -1 || j == 1) ? 4 : 1) * rs;
    }
    dst.img[(size_t)(y + kPad) * dst.pitch + x + kPad] = (uint8_t)((sum + 128) >> 8);
}

// ---- pyramidal LK: one wavefront per track --------------------------------------------------------------------------
// sum over the wave, result in every lane.  DPP moves inside the rows of 16 lanes, then the four row sums by lane reads
// in a fixed order: a `__shfl_xor` butterfly is six ds_bpermute round trips (~250 cycles), this is ~90.
__device__ __forceinline__ float dpp_f32(float x, int pattern /* 0: ^1, 1: ^2, 2: mirror in 8, 3: mirror in 16 */) {
    const int i = __float_as_int(x);
    switch (pattern) {
    case 0: return __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
    case 1: return __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
    case 2: return __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0x141, 0xF, 0xF, true)); // row_half_mirror
    default: return __int_as_float(__builtin_amdgcn_update_dpp(0, i, 0x140, 0xF, 0xF, true)); // row_mirror
    }
}
__device__ __forceinline__ float readlane_f32(float x, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src));
}
__device__ __forceinline__ float wave_sum_f(float v) {
    v += dpp_f32(v, 0);
    v += dpp_f32(v, 1);
    v += dpp_f32(v, 2);
    v += dpp_f32(v, 3);
    return ((readlane_f32(v, 0) + readlane_f32(v, 16)) + readlane_f32(v, 32)) + readlane_f32(v, 48);
}

// Lane l < 63 owns 7 consecutive pixels of one row of the 21 x 21 window (row l / 3, columns 7 (l % 3) ..): the bilinear
// taps of neighbouring pixels overlap, so a lane fetches 2 x 8 bytes per iteration instead of 7 x 4, and its share of the
// template (intensity + both derivatives, 14-bit fixed point like OpenCV) stays in registers -- no LDS in the loop.
// A bilinear sample is two v_dot2c_i32_i16 on (tap, right neighbour) x (left weight, right weight) pairs (measured: 1024
// tracks 22.4 -> 20.8 us, 6000 tracks 63.6 -> 58.9 us against four 24-bit multiplies and three adds).
// The lane's 8 taps of a row are 8 consecutive bytes at an arbitrary address, its 8 derivative pairs 32 consecutive bytes
// at a 4-byte aligned one: one 8-byte and two 16-byte loads instead of 8 + 16 scalar ones (gfx950 global loads need no
// natural alignment).
typedef uint64_t lk_u64_any __attribute__((aligned(1)));
struct __attribute__((aligned(4))) lk_i16x8 {
    int16_t v[8];
};
struct lk_drv_raw {
    lk_i16x8 lo, hi;
};
// a.lo * b.lo + a.hi * b.hi + c on signed 16-bit halves: v_dot2c_i32_i16.  The bilinear sample of OpenCV's fixed-point LK,
// (p00 w00 + p01 w01 + p10 w10 + p11 w11 + round) >> shift, is two of these on (pixel, right neighbour) pairs and
// (left weight, right weight) pairs -- exact integer arithmetic (pixels <= 255 or int16 derivatives, weights <= 2^14).
__device__ __forceinline__ int lk_dot2(int a, int b, int c) {
    typedef short lk_s2 __attribute__((vector_size(4)));
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(lk_s2, a), __builtin_bit_cast(lk_s2, b), c, false); // v_dot2_i32_i16
}
// (tap k, tap k + 1) pairs of one row of 8 taps, k = 0..6
__device__ __forceinline__ void lk_tap_pairs(uint64_t w, int *pr) {
#pragma unroll
    for (int k = 0; k < 7; ++k) pr[k] = (int)(((w >> (8 * k)) & 255u) | (((w >> (8 * k + 8)) & 255u) << 16));
}
__device__ __forceinline__ void lk_load_tap_pairs(const uint8_t *s, int *pr) { lk_tap_pairs(*reinterpret_cast<const lk_u64_any *>(s), pr); }
// derivative rows come as (x, y) int16 pairs per pixel: (x_k, x_k+1) and (y_k, y_k+1) pairs, k = 0..6 -- one v_perm_b32 each
// (bytes 0-3 = the second operand, 4-7 = the first): the low halves of (d_k, d_k+1), and their high halves
__device__ __forceinline__ void lk_deriv_pairs(const lk_drv_raw &r, int *px, int *py) {
    uint32_t d[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        d[k] = (uint32_t)(uint16_t)r.lo.v[2 * k] | ((uint32_t)(uint16_t)r.lo.v[2 * k + 1] << 16);
        d[4 + k] = (uint32_t)(uint16_t)r.hi.v[2 * k] | ((uint32_t)(uint16_t)r.hi.v[2 * k + 1] << 16);
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        px[k] = (int)__builtin_amdgcn_perm(d[k + 1], d[k], 0x05040100u);
        py[k] = (int)__builtin_amdgcn_perm(d[k + 1], d[k], 0x07060302u);
    }
}
__device__ __forceinline__ lk_drv_raw lk_load_derivs_raw(const int16_t *d) {
    lk_drv_raw r;
    r.lo = *reinterpret_cast<const lk_i16x8 *>(d), r.hi = *reinterpret_cast<const lk_i16x8 *>(d + 8);
    return r;
}

constexpr int kRun = 7; // pixels per lane; kWin = 3 * kRun
static_assert(kWin == 3 * kRun, "window / lane mapping");
constexpr int LK_W_BITS = 14;
// One level's template of one lane: position test, bilinear weights and the raw taps (kept packed until formed).
struct LkTplRaw {
    int w00, w01, w10, w11;
    uint64_t i0, i1;
    lk_drv_raw d0, d1;
};
__device__ __forceinline__ int lk_template_load(const LevelDesc &I, int level, float pxf, float pyf, int wy, int wx, LkTplRaw &t) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const float half = (kWin - 1) * 0.5f;
    const float sc = (float)(1. / (1 << level));
    const float px = pxf * sc - half, py = pyf * sc - half;
    const int ipx = (int)floorf(px), ipy = (int)floorf(py);
    if (ipx < -kWin || ipx >= I.w || ipy < -kWin || ipy >= I.h) return 1; // skipped
    const float fa = px - (float)ipx, fb = py - (float)ipy;
    t.w00 = (int)rintf((1.f - fa) * (1.f - fb) * (float)(1 << LK_W_BITS));
    t.w01 = (int)rintf(fa * (1.f - fb) * (float)(1 << LK_W_BITS));
    t.w10 = (int)rintf((1.f - fa) * fb * (float)(1 << LK_W_BITS));
    t.w11 = (1 << LK_W_BITS) - t.w00 - t.w01 - t.w10;
    const size_t o = (size_t)(ipy + wy + kPad) * I.pitch + (ipx + wx + kPad);
    const uint8_t *s0 = I.img + o, *s1 = s0 + I.pitch;
    const int16_t *d0 = I.drv + 2 * o, *d1 = d0 + 2 * I.pitch;
    static_assert(kRun + 1 == 8, "eight taps per lane and row");
    t.i0 = *reinterpret_cast<const lk_u64_any *>(s0), t.i1 = *reinterpret_cast<const lk_u64_any *>(s1);
    t.d0 = lk_load_derivs_raw(d0), t.d1 = lk_load_derivs_raw(d1);
    return 0;
}
// Template of the level (this lane's 7 pixels: intensity << 5, both derivatives) + the gradient matrix.
// Returns 1 when the level is skipped (degenerate gradient matrix).
struct LkTpl {
    int ti[kRun], tx[kRun], ty[kRun];
    float A11, A12, A22, Dinv;
};
__device__ __forceinline__ int lk_template_form(const LkTplRaw &t, bool live, LkTpl &T) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const float FLT_SCALE = 1.f / (1 << 20);
    constexpr int W_BITS = LK_W_BITS;
    const int wA = t.w00 | (t.w01 << 16), wB = t.w10 | (t.w11 << 16); // (left, right) weight pairs of the two rows
    int q0[kRun], q1[kRun], x0[kRun], x1[kRun], y0[kRun], y1[kRun];
    lk_tap_pairs(t.i0, q0), lk_tap_pairs(t.i1, q1);
    lk_deriv_pairs(t.d0, x0, y0), lk_deriv_pairs(t.d1, x1, y1);
    float sA11 = 0, sA12 = 0, sA22 = 0;
#pragma unroll
    for (int k = 0; k < kRun; ++k) {
        T.ti[k] = lk_dot2(q1[k], wB, lk_dot2(q0[k], wA, 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
        T.tx[k] = lk_dot2(x1[k], wB, lk_dot2(x0[k], wA, 1 << (W_BITS - 1))) >> W_BITS;
        T.ty[k] = lk_dot2(y1[k], wB, lk_dot2(y0[k], wA, 1 << (W_BITS - 1))) >> W_BITS;
        sA11 += (float)(T.tx[k] * T.tx[k]), sA12 += (float)(T.tx[k] * T.ty[k]), sA22 += (float)(T.ty[k] * T.ty[k]);
    }
    if (!live) sA11 = 0.f, sA12 = 0.f, sA22 = 0.f; // (selected after the sums, not a branch around each term)
    const float a11 = wave_sum_f(sA11) * FLT_SCALE, a12 = wave_sum_f(sA12) * FLT_SCALE, a22 = wave_sum_f(sA22) * FLT_SCALE;
    const float D = a11 * a22 - a12 * a12;
    const float minEig = (a22 + a11 - sqrtf((a11 - a22) * (a11 - a22) + 4.f * a12 * a12)) / (float)(2 * kWin * kWin);
    T.A11 = a11, T.A12 = a12, T.A22 = a22, T.Dinv = 0.f;
    if (minEig < 1e-4f || D < 1.1920929e-07f) return 1;
    T.Dinv = 1.f / D;
    return 0;
}

// One pyramid level of one track (one wave) in two parts: the level's template (depends on the previous image and the point only, not on the
// position handed down from the coarser level) and the search (the iterations from the handed-down position).  k_lk_track runs them back to back,
// level after level; k_lk_track_levels forms the templates of all levels at once, a wave each.
// Within ONE wave, forming the templates of all levels up front -- their loads and the first search window of the coarsest level in flight
// together -- was built and measured: 33.2 us against 31.7 us for 1500 tracks (205 VGPRs, two waves per SIMD); requesting each level's first
// search window before its template is formed: no change.
__device__ __forceinline__ int lk_level_template(const LevelDesc &I, int level, float pxf, float pyf, int lane, LkTpl &T) {
    const bool live = lane < kWin * 3;
    const int wy = live ? lane / 3 : 0, wx = live ? kRun * (lane - 3 * wy) : 0;
    LkTplRaw raw;
    int skip = lk_template_load(I, level, pxf, pyf, wy, wx, raw); // wave-uniform
    if (!skip) skip = lk_template_form(raw, live, T);
    return skip; // template outside the image / degenerate gradient matrix
}
__device__ __forceinline__ void lk_level_search(const LevelDesc &J, int level, bool coarsest, const LkTpl &T, int skip, int lane, float &outx, float &outy, int &st) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const bool live = lane < kWin * 3;
    const int wy = live ? lane / 3 : 0, wx = live ? kRun * (lane - 3 * wy) : 0;
    const float half = (kWin - 1) * 0.5f, FLT_SCALE = 1.f / (1 << 20);
    constexpr int W_BITS = LK_W_BITS;
    const float sc = (float)(1. / (1 << level));
    float nx, ny;
    if (coarsest) nx = outx * sc, ny = outy * sc; // OPTFLOW_USE_INITIAL_FLOW
    else nx = outx * 2.f, ny = outy * 2.f;
    outx = nx, outy = ny;
    if (skip) {
        if (level == 0) st = 0;
        return;
    }
    const float a11 = T.A11, a12 = T.A12, a22 = T.A22, D = T.Dinv;
    nx -= half, ny -= half;
    float pdx = 0, pdy = 0;
    int r0[kRun], r1[kRun]; // the lane's (tap, right neighbour) pairs of the search window, kept while its integer origin stays
    int cinx = -(1 << 30), ciny = -(1 << 30);
    for (int j = 0; j < 30; ++j) {
        const int inx = (int)floorf(nx), iny = (int)floorf(ny);
        if (inx < -kWin || inx >= J.w || iny < -kWin || iny >= J.h) {
            if (level == 0) st = 0;
            break;
        }
        const float fa = nx - (float)inx, fb = ny - (float)iny;
        const int iw00 = (int)rintf((1.f - fa) * (1.f - fb) * (float)(1 << W_BITS));
        const int iw01 = (int)rintf(fa * (1.f - fb) * (float)(1 << W_BITS));
        const int iw10 = (int)rintf((1.f - fa) * fb * (float)(1 << W_BITS));
        const int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        if (inx != cinx || iny != ciny) { // uniform: the window only moves to other pixels every few iterations
            const uint8_t *s0 = J.img + (size_t)(iny + wy + kPad) * J.pitch + (inx + wx + kPad), *s1 = s0 + J.pitch;
            lk_load_tap_pairs(s0, r0), lk_load_tap_pairs(s1, r1);
            cinx = inx, ciny = iny;
        }
        const int wA = iw00 | (iw01 << 16), wB = iw10 | (iw11 << 16);
        float sb1 = 0, sb2 = 0;
#pragma unroll
        for (int k = 0; k < kRun; ++k) {
            const int diff = (lk_dot2(r1[k], wB, lk_dot2(r0[k], wA, 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5)) - T.ti[k];
            sb1 += (float)__mul24(diff, T.tx[k]), sb2 += (float)__mul24(diff, T.ty[k]);
        }
        if (!live) sb1 = 0.f, sb2 = 0.f;
        const float b1 = wave_sum_f(sb1) * FLT_SCALE, b2 = wave_sum_f(sb2) * FLT_SCALE;
        const float dx = (a12 * b2 - a22 * b1) * D, dy = (a12 * b1 - a11 * b2) * D;
        nx += dx, ny += dy;
        outx = nx + half, outy = ny + half;
        // OpenCV's two termination tests are DOUBLE comparisons (lkpyramid.cpp: `delta.ddot(delta) <= criteria.epsilon` with the double
        // epsilon 0.01 squared by calcOpticalFlowPyrLK = 1.0000000000000002e-4; `std::abs(delta.x + prevDelta.x) < 0.01`: a float sum
        // against the double literal): a float comparison decides differently within an ulp of either threshold.  Wave-uniform, twice per
        // iteration (deciding in float wherever that is provably the same, FP64 only near the thresholds: measured, no gain -- profiles/r5_ab_klt_float_tests.txt).
        if ((double)dx * (double)dx + (double)dy * (double)dy <= 0.01 * 0.01) break;
        if (j > 0 && (double)fabsf(dx + pdx) < 0.01 && (double)fabsf(dy + pdy) < 0.01) {
            outx -= dx * 0.5f, outy -= dy * 0.5f;
            break;
        }
        pdx = dx, pdy = dy;
    }
    if (st && level == 0) {
        const int ix = (int)floorf(outx - half), iy = (int)floorf(outy - half);
        if (ix < -kWin || ix >= J.w || iy < -kWin || iy >= J.h) st = 0;
    }
}
__device__ __forceinline__ void lk_level(const LevelDesc &I, const LevelDesc &J, int level, bool coarsest, float pxf, float pyf, int lane, float &outx, float &outy, int &st) {
    LkTpl T;
    const int skip = lk_level_template(I, level, pxf, pyf, lane, T);
    lk_level_search(J, level, coarsest, T, skip, lane, outx, outy, st);
}
// the end of a track: opencv_image.cpp:104-109, the 20-px border of the full-resolution image
__device__ __forceinline__ void lk_finish(const TrackArgs &a, int p, int lane, float outx, float outy, int st) {
    if (outx < 20 || outx >= (float)(a.prev[0].w - 20) || outy < 20 || outy >= (float)(a.prev[0].h - 20)) st = 0;
    if (lane == 0) {
        a.next_xy[2 * p] = outx, a.next_xy[2 * p + 1] = outy;
        a.status[p] = (uint8_t)st;
    }
}

// One wave per track, its levels in turn: the form for track counts that leave every SIMD at most one wave (n <= 4 x CUs).
__global__ void __launch_bounds__(256) k_lk_track(TrackArgs a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int p = blockIdx.x * 4 + wv;
    if (p >= a.n) return; // whole wave exits together
    const float pxf = a.prev_xy[2 * p], pyf = a.prev_xy[2 * p + 1];
    float outx = a.next_xy[2 * p], outy = a.next_xy[2 * p + 1];
    int st = 1;
#pragma unroll
    for (int li = 0; li < kLevels; ++li) {
        const int level = kLevels - 1 - li;
        if (level >= a.n_levels) continue;
        lk_level(a.prev[level], a.next[level], level, level == a.n_levels - 1, pxf, pyf, lane, outx, outy, st);
    }
    lk_finish(a, p, lane, outx, outy, st);
}

// One WORKGROUP per track, one wave per pyramid level (round 6).  A track's four templates depend on the previous image and the point alone, so the
// four waves load and form them at the same time (38 % of a track's VALU work and four dependent miss latencies with a wave per track:
// tests/micro/klt_stamps.py); the searches are a chain -- level l starts from the position level l + 1 ended at -- and pass through the block in level
// order: wave s searches between workgroup barriers s and s + 1 (hardware barriers: a waiting wave issues nothing and nothing is polled, so the
// code-generation hazard of the unit queue below cannot occur), the position goes down through two LDS words.  A track's chain is template +
// searches (~18 k cycles) instead of 4 templates + searches (~28.5 k); 1500 tracks are 6000 waves, all resident at once (62 VGPRs: eight waves per
// SIMD), so every SIMD carries a quarter of ~6 tracks instead of one or two whole ones.  The arithmetic is lk_level's two halves, untouched: status and
// positions bit-identical to k_lk_track and the oracle (tests/test_gpu_klt.py).
__global__ void __launch_bounds__(64 * kLevels) k_lk_track_levels(TrackArgs a) {
    __shared__ float sh_xy[2][2]; // double-buffered by the parity of the step: a wave's lanes read one pair and write the other
    const int lane = threadIdx.x & 63, p = blockIdx.x;
    // step of this wave in the chain (0: the coarsest level).  (The hardware already starts every block on another SIMD -- tools/ubench/wave_placement.hip:
    // the four waves of a block sit on four SIMDs, wave 0's SIMD differs from block to block, blocks b, b + 256, ... share a CU; rotating the mapping by
    // hand was 10 % slower, profiles/r6_klt_levels_ab.txt)
    const int wv = threadIdx.x >> 6;
    const int level = a.n_levels - 1 - wv; // waves without a level only keep the barrier count
    const float pxf = a.prev_xy[2 * p], pyf = a.prev_xy[2 * p + 1];
    static_assert(kLevels == 4, "level selection below");
    LkTpl T;
    int skip = 1;
    if (level >= 0) {
        const LevelDesc I = level == 0 ? a.prev[0] : level == 1 ? a.prev[1] : level == 2 ? a.prev[2] : a.prev[3];
        skip = lk_level_template(I, level, pxf, pyf, lane, T);
    }
    const LevelDesc J = level <= 0 ? a.next[0] : level == 1 ? a.next[1] : level == 2 ? a.next[2] : a.next[3];
    for (int s = 0; s < a.n_levels; ++s) {
        if (s == wv) { // wave-uniform
            float outx, outy;
            if (s == 0) outx = a.next_xy[2 * p], outy = a.next_xy[2 * p + 1];
            else outx = sh_xy[(s - 1) & 1][0], outy = sh_xy[(s - 1) & 1][1];
            int st = 1; // only level 0 (the last search of a track) can clear it
            lk_level_search(J, level, s == 0, T, skip, lane, outx, outy, st);
            if (level == 0) lk_finish(a, p, lane, outx, outy, st);
            else sh_xy[s & 1][lane & 1] = (lane & 1) ? outy : outx; // every lane stores the wave-uniform value (no lane-0 branch in this loop)
        }
        __syncthreads();
    }
}

// More tracks than SIMDs: (track, level) UNITS from a queue in LDS, one workgroup of eight waves per CU.  With a wave per track, 1500
// tracks leave 476 of the 1024 SIMDs with two waves and 548 with one for the whole launch, and the launch lasts as long as the slowest
// pair (tests/micro/klt_stamps.py, profiles/r5_klt_stamps.txt: a track alone 28.5k cycles, the slowest wave of 1500 60.5k, their mean
// 34k).  Here block b owns the tracks b, b + G, b + 2 G, ... (G blocks = CUs) and its waves take units in the order (coarsest level
// of each of its tracks, then the next level of each, ...): the CU's five or six level chains move from SIMD to SIMD with the wave
// that happens to be free, so the SIMDs of a CU share its work instead of two of them carrying two tracks from start to end.  A unit
// needs the position its predecessor hands down: sh_xy[track] behind sh_done[track] (workgroup-scope release / acquire: LDS, no global
// atomics -- a queue in global memory was built first: one contended agent-scope counter serves ~80 M units/s, 125 us for 1500 tracks,
// profiles/r5_ab_klt_units_global_queue.txt).  The unit a wave waits for was taken earlier by a wave of the same block, which is
// resident: it can never wait on a unit that nobody holds.  The per-track arithmetic is lk_level's, untouched: bit-identical results.
// Measured (profiles/r5_klt_units_check.txt, same box, min of 8): 1500 tracks 31.9 -> 30.2 us, 3000 42.2 -> 39.6, 2048 31.6 -> 32.2 (two
// waves on every SIMD either way), 6000 61.1 -> 61.0: the rotation buys 5 %, not the 30 % the SIMD arithmetic promised -- what is left
// is the CU's own share (six tracks of 16 +- 4 iterations each against five) and the serial chain of a track's four levels.
constexpr int kLkUnitWaves = 8, kLkMaxOwned = 64; // waves per block; tracks a block can own (more tracks than 64 per CU: a wave per track)
__global__ void __launch_bounds__(64 * kLkUnitWaves) k_lk_track_units(TrackArgs a) {
    __shared__ int sh_next, sh_done[kLkMaxOwned];
    __shared__ float sh_xy[2 * kLkMaxOwned];
    const int lane = threadIdx.x & 63, G = gridDim.x, b = blockIdx.x;
    const int owned = (a.n - b + G - 1) / G; // tracks b + k G < n
    if (threadIdx.x == 0) sh_next = 0;
    if (threadIdx.x < kLkMaxOwned) sh_done[threadIdx.x] = 0;
    __syncthreads();
    const int total = a.n_levels * owned;
    for (;;) {
        // The wave barrier (a convergent no-op) keeps this `if (lane == 0)` apart from the `if (lane == 0)` of the hand-down at the end of the
        // previous pass: without it the compiler threads the two branches across the back edge, lanes 1-63 go round an inner loop of their
        // own and meet the readfirstlane without lane 0 -- unit 0 for ever (the first build of this kernel hung on the GPU that way;
        // tests/test_isa_guards.py checks the loop nest of the compiled kernel).
        __builtin_amdgcn_wave_barrier();
        int u = 0;
        if (lane == 0) u = __hip_atomic_fetch_add(&sh_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= total) break;
        const int li = u / owned, k = u - li * owned, p = b + k * G, level = a.n_levels - 1 - li;
        const float pxf = a.prev_xy[2 * p], pyf = a.prev_xy[2 * p + 1];
        float outx, outy;
        if (li == 0) outx = a.next_xy[2 * p], outy = a.next_xy[2 * p + 1];
        else {
            while (__hip_atomic_load(&sh_done[k], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < li) __builtin_amdgcn_s_sleep(1);
            outx = sh_xy[2 * k], outy = sh_xy[2 * k + 1];
        }
        int st = 1; // only level 0 (the last unit of a track) can clear it
        const LevelDesc I = level == 0 ? a.prev[0] : level == 1 ? a.prev[1] : level == 2 ? a.prev[2] : a.prev[3];
        const LevelDesc J = level == 0 ? a.next[0] : level == 1 ? a.next[1] : level == 2 ? a.next[2] : a.next[3];
        static_assert(kLevels == 4, "level selection above");
        lk_level(I, J, level, li == 0, pxf, pyf, lane, outx, outy, st);
        if (level == 0) lk_finish(a, p, lane, outx, outy, st);
        else if (lane == 0) {
            sh_xy[2 * k] = outx, sh_xy[2 * k + 1] = outy;
            __hip_atomic_store(&sh_done[k], li + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// ---- host -----------------------------------------------------------------------------------------------------------
static int level_sizes(int w, int h, int *ws, int *hs) {
    int n = 0;
    for (int l = 0; l < kLevels; ++l) {
        if (l > 0) {
            w = (w + 1) / 2, h = (h + 1) / 2;
            if (w <= kWin || h <= kWin) break; // buildOpticalFlowPyramid stops when a level is not larger than the window
        }
        ws[n] = w, hs[n] = h, ++n;
    }
    return n;
}

Klt::Klt(int device) : device_(device) {
    (void)hipSetDevice(device_);
    hipDeviceProp_t prop;
    n_simds_ = hipGetDeviceProperties(&prop, device_) == hipSuccess ? 4 * std::max(1, prop.multiProcessorCount) : 1024;
    if (const char *e = std::getenv("PVIO_HIP_LK_FORM")) lk_form_ = std::min(3, std::max(0, std::atoi(e)));
    if (const char *e = std::getenv("PVIO_HIP_LK_BLOCKS")) lk_blocks_ = std::max(0, std::atoi(e));
    (void)hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking);
    (void)hipEventCreate(&ev0_);
    (void)hipEventCreate(&ev1_);
}
Klt::~Klt() {
    if (d_pts_) (void)hipFree(d_pts_);
    if (d_det_) (void)hipFree(d_det_);
    if (d_src_) (void)hipFree(d_src_);
    if (h_pts_) (void)hipHostFree(h_pts_);
    if (d_fm_) (void)hipFree(d_fm_);
    if (h_fm_) (void)hipHostFree(h_fm_);
    if (det_host_) (void)hipHostFree(det_host_);
    for (auto &s : slab_pool_) (void)hipFree(s.second);
    if (staging_) (void)hipHostFree(staging_);
    if (ev0_) (void)hipEventDestroy(ev0_);
    if (ev1_) (void)hipEventDestroy(ev1_);
    if (stream_) (void)hipStreamDestroy(stream_);
}

int Klt::create_undistort(const int16_t *map_xy, const uint16_t *map_frac, int w, int h, Undistort **out) {
    if (!map_xy || !map_frac || w < 1 || h < 1) {
        err_ = "bad undistortion map";
        return PVIO_ERR_INVALID_ARGUMENT;
    }
    (void)hipSetDevice(device_);
    Undistort *u = new Undistort();
    u->w = w, u->h = h, u->xy = nullptr, u->frac = nullptr;
    const size_t n = (size_t)w * h;
    void *slab = nullptr;
    if (hipMalloc(&slab, n * 6) != hipSuccess) {
        delete u;
        err_ = "hipMalloc failed";
        return PVIO_ERR_OUT_OF_MEMORY;
    }
    u->xy = static_cast<int16_t *>(slab);
    u->frac = reinterpret_cast<uint16_t *>(static_cast<char *>(slab) + n * 4);
    if (hipMemcpy(u->xy, map_xy, n * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(u->frac, map_frac, n * 2, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(slab);
        delete u;
        err_ = "H2D failed";
        return PVIO_ERR_HIP;
    }
    *out = u;
    return PVIO_OK;
}
void Klt::release_undistort(Undistort *u) {
    if (!u) return;
    if (u->xy) (void)hipFree(u->xy);
    delete u;
}

int Klt::create_image(const uint8_t *pixels, int sw, int sh, int stride, bool clahe, Image **out, const Undistort *ud) {
    // with an undistortion map the pyramid has the MAP's size and the uploaded pixels are only the source of the remap
    const int w = ud ? ud->w : sw, h = ud ? ud->h : sh;
    if (w < 2 * kWin || h < 2 * kWin || sw < 1 || sh < 1 || stride < sw) {
        err_ = "image too small / bad stride";
        return PVIO_ERR_INVALID_ARGUMENT;
    }
    (void)hipSetDevice(device_);
    Image *im = new Image();
    std::memset(im, 0, sizeof *im);
    im->w = w, im->h = h;
    int ws[kLevels], hs[kLevels];
    im->n_levels = level_sizes(w, h, ws, hs);
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    };
    size_t o_raw = carve((size_t)w * h), o_lut = carve(64 * 256), o_img[kLevels], o_drv[kLevels];
    for (int l = 0; l < im->n_levels; ++l) {
        im->lv[l].w = ws[l], im->lv[l].h = hs[l];
        im->lv[l].pitch = (ws[l] + 2 * kPad + 63) & ~63;
        o_img[l] = carve((size_t)(hs[l] + 2 * kPad) * im->lv[l].pitch);
        o_drv[l] = carve((size_t)(hs[l] + 2 * kPad) * im->lv[l].pitch * 4);
    }
    im->slab_bytes = off;
    for (size_t i = 0; i < slab_pool_.size(); ++i)
        if (slab_pool_[i].first == off) {
            im->slab = slab_pool_[i].second;
            slab_pool_.erase(slab_pool_.begin() + i);
            break;
        }
    if (!im->slab && hipMalloc(&im->slab, off) != hipSuccess) {
        delete im;
        err_ = "hipMalloc failed";
        return PVIO_ERR_OUT_OF_MEMORY;
    }
    char *base = static_cast<char *>(im->slab);
    im->raw = reinterpret_cast<uint8_t *>(base + o_raw);
    im->lut = reinterpret_cast<uint8_t *>(base + o_lut);
    for (int l = 0; l < im->n_levels; ++l) {
        im->lv[l].img = reinterpret_cast<uint8_t *>(base + o_img[l]);
        im->lv[l].drv = reinterpret_cast<int16_t *>(base + o_drv[l]);
    }
    // derivative borders are BORDER_CONSTANT zeros; image borders are written by k_border
    (void)hipMemsetAsync(im->slab, 0, off, stream_);
    // pixels go through a pinned staging buffer (packed rows): the copy is a real asynchronous DMA then
    if ((size_t)sw * sh > staging_cap_) {
        if (staging_) (void)hipHostFree(staging_);
        staging_ = nullptr, staging_cap_ = 0;
        if (hipHostMalloc(&staging_, (size_t)sw * sh) != hipSuccess) {
            release_image(im);
            err_ = "hipHostMalloc failed";
            return PVIO_ERR_OUT_OF_MEMORY;
        }
        staging_cap_ = (size_t)sw * sh;
    }
    for (int y = 0; y < sh; ++y) std::memcpy(static_cast<uint8_t *>(staging_) + (size_t)y * sw, pixels + (size_t)y * stride, sw);
    const uint8_t *src = static_cast<const uint8_t *>(staging_);
    uint8_t *d_src = im->raw; // distorted pixels land in a scratch buffer, the remap writes im->raw
    if (ud) {
        if ((size_t)sw * sh > src_cap_) {
            if (d_src_) (void)hipFree(d_src_);
            d_src_ = nullptr, src_cap_ = 0;
            if (hipMalloc(&d_src_, (size_t)sw * sh) != hipSuccess) {
                release_image(im);
                err_ = "hipMalloc failed";
                return PVIO_ERR_OUT_OF_MEMORY;
            }
            src_cap_ = (size_t)sw * sh;
        }
        d_src = static_cast<uint8_t *>(d_src_);
    }
    if (hipMemcpyAsync(d_src, src, (size_t)sw * sh, hipMemcpyHostToDevice, stream_) != hipSuccess) {
        release_image(im);
        err_ = "H2D failed";
        return PVIO_ERR_HIP;
    }
    const dim3 blk(256);
    if (ud) hipLaunchKernelGGL(k_remap, dim3((w + 255) / 256, h), blk, 0, stream_, (const uint8_t *)d_src, sw, sh, (const int16_t *)ud->xy, (const uint16_t *)ud->frac, w, h, im->raw);
    if (clahe) {
        const int tiles = 8;
        int ew = w, eh = h;
        if (w % tiles != 0 || h % tiles != 0) ew = w + (tiles - w % tiles), eh = h + (tiles - h % tiles);
        const int tw = ew / tiles, th = eh / tiles;
        int clip = (int)(6.0 * (tw * th) / 256);
        if (clip < 1) clip = 1;
        hipLaunchKernelGGL(k_clahe_lut, dim3(tiles * tiles), blk, 0, stream_, (const uint8_t *)im->raw, w, h, tiles, tiles, tw, th, clip, im->lut);
        hipLaunchKernelGGL(k_clahe_apply, dim3((w + 255) / 256, h), blk, 0, stream_, (const uint8_t *)im->raw, w, h, tiles, tiles, tw, th,
                           (const uint8_t *)im->lut, im->lv[0]);
    } else {
        hipLaunchKernelGGL(k_copy_level0, dim3((w + 255) / 256, h), blk, 0, stream_, (const uint8_t *)im->raw, w, h, im->lv[0]);
    }
    for (int l = 0; l < im->n_levels; ++l) {
        const LevelDesc &L = im->lv[l];
        hipLaunchKernelGGL(k_border, dim3((L.w + 2 * kPad + 255) / 256, L.h + 2 * kPad), blk, 0, stream_, L);
        hipLaunchKernelGGL(k_scharr, dim3((L.w + 255) / 256, L.h), blk, 0, stream_, L);
        if (l + 1 < im->n_levels) hipLaunchKernelGGL(k_pyr_down, dim3((im->lv[l + 1].w + 255) / 256, im->lv[l + 1].h), blk, 0, stream_, L, im->lv[l + 1]);
    }
    if (hipStreamSynchronize(stream_) != hipSuccess || hipGetLastError() != hipSuccess) {
        release_image(im);
        err_ = "pyramid kernels failed";
        return PVIO_ERR_HIP;
    }
    *out = im;
    return PVIO_OK;
}

void Klt::release_image(Image *img) {
    if (!img) return;
    if (img->slab) {
        if (slab_pool_.size() < 8) slab_pool_.emplace_back(img->slab_bytes, img->slab);
        else (void)hipFree(img->slab);
    }
    delete img;
}

int Klt::download_level(const Image *img, int level, uint8_t *pixels, int16_t *deriv, int32_t *w, int32_t *h) {
    if (level < 0 || level >= img->n_levels) return PVIO_ERR_INVALID_ARGUMENT;
    const LevelDesc &L = img->lv[level];
    if (w) *w = L.w;
    if (h) *h = L.h;
    std::vector<uint8_t> hp;
    std::vector<int16_t> hd;
    const size_t rows = L.h + 2 * kPad;
    if (pixels) {
        hp.resize(rows * L.pitch);
        if (hipMemcpy(hp.data(), L.img, hp.size(), hipMemcpyDeviceToHost) != hipSuccess) return PVIO_ERR_HIP;
        for (int y = 0; y < L.h; ++y) std::memcpy(pixels + (size_t)y * L.w, &hp[(size_t)(y + kPad) * L.pitch + kPad], L.w);
    }
    if (deriv) {
        hd.resize(rows * L.pitch * 2);
        if (hipMemcpy(hd.data(), L.drv, hd.size() * 2, hipMemcpyDeviceToHost) != hipSuccess) return PVIO_ERR_HIP;
        for (int y = 0; y < L.h; ++y) std::memcpy(deriv + (size_t)y * L.w * 2, &hd[2 * ((size_t)(y + kPad) * L.pitch + kPad)], (size_t)L.w * 4);
    }
    return PVIO_OK;
}

int Klt::track(const Image *prev, const Image *next, int n, const float *prev_xy, float *next_xy, uint8_t *status) {
    if (prev->w != next->w || prev->h != next->h) {
        err_ = "image sizes differ";
        return PVIO_ERR_INVALID_ARGUMENT;
    }
    last_ms_ = 0;
    if (n == 0) return PVIO_OK;
    (void)hipSetDevice(device_);
    const size_t need = (size_t)n * (2 * sizeof(float) * 2 + 1) + 64;
    if (need > pts_cap_) {
        if (d_pts_) (void)hipFree(d_pts_);
        d_pts_ = nullptr;
        if (hipMalloc(&d_pts_, need * 2) != hipSuccess) {
            pts_cap_ = 0;
            err_ = "hipMalloc failed";
            return PVIO_ERR_OUT_OF_MEMORY;
        }
        pts_cap_ = 0; // the capacity counts only once BOTH buffers exist
        if (h_pts_) (void)hipHostFree(h_pts_);
        h_pts_ = nullptr;
        if (hipHostMalloc(&h_pts_, need * 2) != hipSuccess) {
            h_pts_ = nullptr;
            err_ = "hipHostMalloc failed";
            return PVIO_ERR_OUT_OF_MEMORY;
        }
        pts_cap_ = need * 2;
    }
    float *d_prev = static_cast<float *>(d_pts_), *d_next = d_prev + 2 * (size_t)n;
    uint8_t *d_st = reinterpret_cast<uint8_t *>(d_next + 2 * (size_t)n);
    TrackArgs a;
    a.n = n, a.n_levels = std::min(prev->n_levels, next->n_levels);
    for (int l = 0; l < a.n_levels; ++l) a.prev[l] = prev->lv[l], a.next[l] = next->lv[l];
    a.prev_xy = d_prev, a.next_xy = d_next, a.status = d_st;
    // one DMA in ([previous | initial next] points), one out ([next points | status bytes]) through a pinned buffer laid out
    // like the device one: four pageable copies cost more than the kernel
    char *hp = static_cast<char *>(h_pts_);
    std::memcpy(hp, prev_xy, (size_t)n * 8), std::memcpy(hp + (size_t)n * 8, next_xy, (size_t)n * 8);
    bool ok = hipMemcpyAsync(d_prev, hp, (size_t)n * 16, hipMemcpyHostToDevice, stream_) == hipSuccess;
    // more tracks than SIMDs: (track, level) units from a queue in LDS, a block of eight waves per CU (see k_lk_track_units); otherwise a wave per track
    const int blocks = lk_blocks_ > 0 ? lk_blocks_ : n_simds_ / 4;
    const bool units = lk_form_ == 2 && (n + blocks - 1) / blocks <= kLkMaxOwned;
    // default: a workgroup per track and a wave per level up to three tracks per CU, a wave per track beyond -- same-box A/Bs of the mean of 50 launches
    // (profiles/r6_klt_bench_ab.txt, bench.py's own point set): 150 tracks 22.2 -> 19.2 us, 300 tracks 25.5 -> 22.4 us; 600 tracks 23.3 -> 24.0,
    // 1500 tracks 35.3 -> 36.2 (with five or six chains per CU the level waves of different tracks meet on the same SIMDs: nothing is gained)
    const bool levels = lk_form_ == 0 ? (long)n * 4 <= 3L * n_simds_ : lk_form_ == 3;
    (void)hipEventRecord(ev0_, stream_);
    if (units) hipLaunchKernelGGL(k_lk_track_units, dim3(std::min(blocks, n)), dim3(64 * kLkUnitWaves), 0, stream_, a);
    else if (levels) hipLaunchKernelGGL(k_lk_track_levels, dim3(n), dim3(64 * kLevels), 0, stream_, a);
    else hipLaunchKernelGGL(k_lk_track, dim3((n + 3) / 4), dim3(256), 0, stream_, a);
    (void)hipEventRecord(ev1_, stream_);
    ok = ok && hipMemcpyAsync(hp + (size_t)n * 8, d_next, (size_t)n * 9, hipMemcpyDeviceToHost, stream_) == hipSuccess;
    ok = ok && hipStreamSynchronize(stream_) == hipSuccess && hipGetLastError() == hipSuccess;
    if (!ok) {
        err_ = "klt track failed";
        return PVIO_ERR_HIP;
    }
    std::memcpy(next_xy, hp + (size_t)n * 8, (size_t)n * 8), std::memcpy(status, hp + (size_t)n * 16, (size_t)n);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev0_, ev1_);
    last_ms_ = ms;
    return PVIO_OK;
}

// ---- corner detection: cv::goodFeaturesToTrack with the Harris measure (opencv_image.cpp:61, detector :183) ---------
// Streaming image kernels over level 0 (the CLAHE output, padded with its REFLECT_101 border); the float operations
// follow ONE fixed order (stated in oracle/oracle_gftt.cpp) so that the response map is reproducible bit for bit.
__global__ void __launch_bounds__(256) k_harris_cov(LevelDesc L, float *cxx, float *cxy, float *cyy) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= L.w) return;
    const uint8_t *p = L.img + (size_t)(y + kPad) * L.pitch + (x + kPad);
    const int a0 = p[-L.pitch - 1], a1 = p[-L.pitch], a2 = p[-L.pitch + 1], b0 = p[-1], b2 = p[1], c0 = p[L.pitch - 1], c1 = p[L.pitch], c2 = p[L.pitch + 1];
    const float s = (float)(1.0 / (4.0 * 3.0 * 255.0)), s2 = 2.0f * s;
    float dx = s * (float)((a2 - a0) + (c2 - c0));
    dx = dx + s2 * (float)(b2 - b0);
    const float dy = s * (float)((c0 + 2 * c1 + c2) - (a0 + 2 * a1 + a2));
    const size_t o = (size_t)y * L.w + x;
    cxx[o] = dx * dx, cxy[o] = dx * dy, cyy[o] = dy * dy;
}

__global__ void __launch_bounds__(256) k_harris_response(int w, int h, const float *cxx, const float *cxy, const float *cyy, float *resp, int *max_bits) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    __shared__ float smax[4];
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    float r = -INFINITY;
    if (x < w) {
        float a = 0, b = 0, c = 0;
#pragma unroll
        for (int j = -1; j <= 1; ++j) {
            const size_t row = (size_t)reflect101(y + j, h) * w;
#pragma unroll
            for (int i = -1; i <= 1; ++i) {
                const size_t o = row + reflect101(x + i, w);
                a += cxx[o], b += cxy[o], c += cyy[o];
            }
        }
        const float t = a + c;
        r = a * c;
        r = r - b * b;
        r = r - (0.04f * t) * t;
        resp[(size_t)y * w + x] = r;
    }
    // block maximum -> one atomic per block (float order == int order of the bit patterns for non-negative values; a negative
    // maximum never beats the 0 the cell starts from: the threshold is then 0 and TOZERO keeps nothing negative, as in OpenCV)
    float m = r;
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
        if (m > 0.0f) atomicMax(max_bits, __float_as_int(m));
    }
}

// keymap != nullptr: every pixel also gets its candidate response (0 = not a candidate) for the dominance filter below
__global__ void __launch_bounds__(256) k_harris_collect(int w, int h, const float *resp, const int *max_bits, float quality, int cap, int *count, float *cand_val,
                                                        int *cand_pos, float *keymap) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (keymap && x < w) keymap[(size_t)y * w + x] = 0.0f;
    if (x < 1 || x >= w - 1 || y < 1 || y >= h - 1) return;
    const float mx = __int_as_float(*max_bits);
    const float thr = (float)((double)mx * (double)quality);
    const float v0 = resp[(size_t)y * w + x];
    const float v = v0 > thr ? v0 : 0.0f;
    if (v == 0.0f) return;
    float m = v;
#pragma unroll
    for (int j = -1; j <= 1; ++j)
#pragma unroll
        for (int i = -1; i <= 1; ++i) {
            const float q = resp[(size_t)(y + j) * w + x + i];
            m = fmaxf(m, q > thr ? q : 0.0f);
        }
    if (v == m) {
        if (keymap) keymap[(size_t)y * w + x] = v;
        const int slot = atomicAdd(count, 1);
        if (slot < cap) cand_val[slot] = v, cand_pos[slot] = y * w + x;
    }
}

// ---- exact pre-filter of the greedy minimum-distance selection ---------------------------------------------------------
// goodFeaturesToTrack walks the candidates in descending (response, address) order and accepts one iff no ACCEPTED corner
// lies closer than min_distance.  A candidate that is the strongest within min_distance of itself ("dominant") is always
// accepted, hence every other candidate within min_distance of a dominant one is always rejected -- and a rejected
// candidate influences nothing.  Dropping those on the device leaves the host with the contested candidates only (a
// quarter of them on textured images) and changes no result.  Keys are unique (the address breaks ties), the distance test
// is the host's: dx^2 + dy^2 < min_distance^2 (integer offsets, compared in double).
__device__ __forceinline__ bool lk_key_greater(float va, int pa, float vb, int pb) { return va > vb || (va == vb && pa > pb); }
// One WAVE per candidate of the compacted list: the lanes share the (2R + 1)^2 positions of its disc (independent loads; a
// thread per pixel walking the disc alone paid one dependent L2 round trip per position: 1.2 ms).
__global__ void __launch_bounds__(256) k_corner_dominant(int w, int h, const float *keymap, const int *count, const float *cand_val, const int *cand_pos, int cap,
                                                         int R, double md2, uint8_t *dominant_map, uint8_t *dominant_flag) {
    const int lane = threadIdx.x & 63;
    const int n = *count < cap ? *count : cap;
    for (int idx = blockIdx.x * 4 + (threadIdx.x >> 6); idx < n; idx += gridDim.x * 4) { // whole waves; the count is only known here
        const float v = cand_val[idx];
        const int o = cand_pos[idx], y = o / w, x = o - y * w, side = 2 * R + 1;
        bool beaten = false;
        for (int e = lane; e < side * side; e += 64) {
            const int j = e / side - R, i = e - (j + R) * side - R, xx = x + i, yy = y + j;
            if (xx < 0 || xx >= w || yy < 0 || yy >= h || (double)(i * i + j * j) >= md2) continue;
            const float q = keymap[(size_t)yy * w + xx];
            beaten |= q > 0.0f && lk_key_greater(q, yy * w + xx, v, o);
        }
        const bool dom = __ballot(beaten ? 1 : 0) == 0;
        if (lane == 0) {
            dominant_flag[idx] = dom ? 1 : 0;
            if (dom) dominant_map[o] = 1; // the map was cleared before the launch
        }
    }
}
__global__ void __launch_bounds__(256) k_corner_survivors(int w, int h, const uint8_t *dominant_map, const uint8_t *dominant_flag, const int *count,
                                                          const float *cand_val, const int *cand_pos, int cap, int R, double md2, int *count_out, float *out_val,
                                                          int *out_pos) {
    const int lane = threadIdx.x & 63;
    const int n = *count < cap ? *count : cap;
    for (int idx = blockIdx.x * 4 + (threadIdx.x >> 6); idx < n; idx += gridDim.x * 4) {
        const int o = cand_pos[idx], y = o / w, x = o - y * w, side = 2 * R + 1;
        bool killed = false;
        if (!dominant_flag[idx]) { // wave-uniform
            for (int e = lane; e < side * side; e += 64) {
                const int j = e / side - R, i = e - (j + R) * side - R, xx = x + i, yy = y + j;
                if (xx < 0 || xx >= w || yy < 0 || yy >= h || (double)(i * i + j * j) >= md2) continue;
                killed |= dominant_map[(size_t)yy * w + xx] != 0; // a dominant candidate in reach is necessarily stronger than this one
            }
        }
        const bool keep = __ballot(killed ? 1 : 0) == 0;
        if (lane == 0 && keep) {
            const int slot = atomicAdd(count_out, 1);
            if (slot < cap) out_val[slot] = cand_val[idx], out_pos[slot] = o;
        }
    }
}

int Klt::detect(const Image *img, int max_corners, double quality, double min_distance, float *xy, float *response, int *n_out) {
    *n_out = 0;
    (void)hipSetDevice(device_);
    const int w = img->w, h = img->h;
    const size_t px = (size_t)w * h;
    const int cap = (int)(px / 4 + 64);
    const size_t need = px * 4 * sizeof(float) + (size_t)cap * 8 + 256;
    if (need > det_cap_) {
        if (d_det_) (void)hipFree(d_det_);
        d_det_ = nullptr;
        if (hipMalloc(&d_det_, need) != hipSuccess) {
            det_cap_ = 0;
            err_ = "hipMalloc failed";
            return PVIO_ERR_OUT_OF_MEMORY;
        }
        det_cap_ = need;
    }
    float *cxx = static_cast<float *>(d_det_), *cxy = cxx + px, *cyy = cxy + px, *resp = cyy + px, *cand_val = resp + px;
    int *cand_pos = reinterpret_cast<int *>(cand_val + cap), *scal = cand_pos + cap; // scal[0] = max bits, scal[1] = count
    (void)hipMemsetAsync(scal, 0, 12, stream_); // [0] max bits, [1] candidates, [2] contested candidates
    const dim3 grid((w + 255) / 256, h), blk(256);
    hipLaunchKernelGGL(k_harris_cov, grid, blk, 0, stream_, img->lv[0], cxx, cxy, cyy);
    hipLaunchKernelGGL(k_harris_response, grid, blk, 0, stream_, w, h, (const float *)cxx, (const float *)cxy, (const float *)cyy, resp, scal);
    const float *list_val = cand_val;
    const int *list_pos = cand_pos;
    int count_slot = 1;
    if (min_distance >= 1) {
        // all candidates -> list + key map (the xx plane is dead by now) -> dominant flags / map (the xy plane) -> contested
        // candidates, compacted into the yy plane
        float *keymap = cxx;
        uint8_t *dominant_map = reinterpret_cast<uint8_t *>(cxy), *dominant_flag = dominant_map + px;
        float *out_val = cyy;
        int *out_pos = reinterpret_cast<int *>(cyy + cap);
        static_assert(sizeof(float) == 4, "plane arithmetic");
        const int R = (int)std::ceil(min_distance);
        const double md2 = min_distance * min_distance; // compared in double like the host selection
        hipLaunchKernelGGL(k_harris_collect, grid, blk, 0, stream_, w, h, (const float *)resp, (const int *)scal, (float)quality, cap, scal + 1, cand_val, cand_pos,
                           keymap);
        (void)hipMemsetAsync(dominant_map, 0, px, stream_);
        const dim3 lgrid(std::min((cap + 3) / 4, 2048)); // the kernels loop: the candidate count is only known on the device
        hipLaunchKernelGGL(k_corner_dominant, lgrid, blk, 0, stream_, w, h, (const float *)keymap, (const int *)(scal + 1), (const float *)cand_val,
                           (const int *)cand_pos, cap, R, md2, dominant_map, dominant_flag);
        hipLaunchKernelGGL(k_corner_survivors, lgrid, blk, 0, stream_, w, h, (const uint8_t *)dominant_map, (const uint8_t *)dominant_flag, (const int *)(scal + 1),
                           (const float *)cand_val, (const int *)cand_pos, cap, R, md2, scal + 2, out_val, out_pos);
        list_val = out_val, list_pos = out_pos, count_slot = 2;
    } else {
        hipLaunchKernelGGL(k_harris_collect, grid, blk, 0, stream_, w, h, (const float *)resp, (const int *)scal, (float)quality, cap, scal + 1, cand_val, cand_pos,
                           (float *)nullptr);
    }
    static const bool timing = getenv("PVIO_KLT_TIMING") != nullptr; // diagnostics: where a detect() call spends its time
    const auto tt0 = std::chrono::steady_clock::now();
    // counts and a first slice of the list come back together (one synchronization); a longer list costs a second round
    constexpr int kFirst = 4096;
    if (!det_host_) {
        if (hipHostMalloc(&det_host_, 16 + (size_t)kFirst * 8) != hipSuccess) {
            err_ = "hipHostMalloc failed";
            return PVIO_ERR_OUT_OF_MEMORY;
        }
    }
    int *hs = static_cast<int *>(det_host_);
    float *h_val = reinterpret_cast<float *>(hs + 4);
    int *h_pos = reinterpret_cast<int *>(h_val + kFirst);
    const int first = std::min(kFirst, cap);
    bool ok = hipMemcpyAsync(hs, scal, 12, hipMemcpyDeviceToHost, stream_) == hipSuccess;
    ok = ok && hipMemcpyAsync(h_val, list_val, (size_t)first * 4, hipMemcpyDeviceToHost, stream_) == hipSuccess;
    ok = ok && hipMemcpyAsync(h_pos, list_pos, (size_t)first * 4, hipMemcpyDeviceToHost, stream_) == hipSuccess;
    ok = ok && hipStreamSynchronize(stream_) == hipSuccess && hipGetLastError() == hipSuccess;
    if (ok && hs[1] > cap) {
        err_ = "too many corner candidates"; // a quarter of the pixels: the 3 x 3 non-maximum suppression cannot produce more
        return PVIO_ERR_UNSUPPORTED;
    }
    const int nc = ok ? std::min(hs[count_slot], cap) : 0;
    std::vector<float> val((size_t)nc);
    std::vector<int> pos((size_t)nc);
    if (ok && nc > 0) {
        const int got = std::min(nc, first);
        std::memcpy(val.data(), h_val, (size_t)got * 4), std::memcpy(pos.data(), h_pos, (size_t)got * 4);
        if (nc > got) {
            ok = hipMemcpyAsync(val.data() + got, list_val + got, (size_t)(nc - got) * 4, hipMemcpyDeviceToHost, stream_) == hipSuccess;
            ok = ok && hipMemcpyAsync(pos.data() + got, list_pos + got, (size_t)(nc - got) * 4, hipMemcpyDeviceToHost, stream_) == hipSuccess &&
                 hipStreamSynchronize(stream_) == hipSuccess;
        }
    }
    if (!ok) {
        err_ = "corner detection failed";
        return PVIO_ERR_HIP;
    }
    const auto tt1 = std::chrono::steady_clock::now();
    // the order the atomics handed the slots out in is arbitrary: (response, address) descending makes it canonical again.
    // One 64-bit key per candidate -- response bits (non-negative floats order like their bit patterns) above the address --
    // sorts without indirection (an index sort with a two-array comparator took 430 us for 7 300 candidates, this 60).
    std::vector<uint64_t> &keys = det_keys_;
    keys.resize((size_t)nc);
    for (int i = 0; i < nc; ++i) {
        uint32_t bits;
        std::memcpy(&bits, &val[(size_t)i], 4);
        keys[(size_t)i] = ((uint64_t)bits << 32) | (uint32_t)pos[(size_t)i];
    }
    std::sort(keys.begin(), keys.end(), std::greater<uint64_t>());
    const auto tt2 = std::chrono::steady_clock::now();
    int n = 0;
    auto key_val = [](uint64_t k) {
        const uint32_t bits = (uint32_t)(k >> 32);
        float f;
        std::memcpy(&f, &bits, 4);
        return f;
    };
    if (min_distance >= 1) { // greedy minimum-distance selection on a grid (goodFeaturesToTrack)
        // cell = min_distance: accepted corners are >= min_distance apart, so a cell holds at most four of them (kept five
        // slots: rounding of the cell size) -- a flat table reused from call to call instead of a vector per cell
        const int cell = (int)std::lround(min_distance), gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        constexpr int kPerCell = 8;
        det_grid_cnt_.assign((size_t)gw * gh, 0);
        det_grid_xy_.resize((size_t)gw * gh * kPerCell * 2);
        const double md2 = min_distance * min_distance;
        for (int ii = 0; ii < nc; ++ii) {
            const int p1 = (int)(uint32_t)keys[(size_t)ii];
            const int y = p1 / w, x = p1 - y * w, xc = x / cell, yc = y / cell;
            bool good = true;
            for (int yy = std::max(0, yc - 1); yy <= std::min(gh - 1, yc + 1) && good; ++yy)
                for (int xx = std::max(0, xc - 1); xx <= std::min(gw - 1, xc + 1) && good; ++xx) {
                    const size_t c = (size_t)yy * gw + xx;
                    const float *q = det_grid_xy_.data() + c * kPerCell * 2;
                    for (int k = 0; k < det_grid_cnt_[c]; ++k) {
                        const float dx = x - q[2 * k], dy = y - q[2 * k + 1];
                        if (dx * dx + dy * dy < md2) {
                            good = false;
                            break;
                        }
                    }
                }
            if (!good) continue;
            const size_t c = (size_t)yc * gw + xc;
            if (det_grid_cnt_[c] >= kPerCell) { // cannot happen for min_distance >= 1 (see above); refuse rather than overflow
                err_ = "corner grid cell overflow";
                return PVIO_ERR_UNSUPPORTED;
            }
            det_grid_xy_[(c * kPerCell + det_grid_cnt_[c]) * 2] = (float)x, det_grid_xy_[(c * kPerCell + det_grid_cnt_[c]) * 2 + 1] = (float)y;
            ++det_grid_cnt_[c];
            xy[2 * n] = (float)x, xy[2 * n + 1] = (float)y, response[n] = key_val(keys[(size_t)ii]);
            if (++n >= max_corners && max_corners > 0) break;
        }
    } else {
        for (int ii = 0; ii < nc; ++ii) {
            const int p1 = (int)(uint32_t)keys[(size_t)ii];
            xy[2 * n] = (float)(p1 % w), xy[2 * n + 1] = (float)(p1 / w), response[n] = key_val(keys[(size_t)ii]);
            if (++n >= max_corners && max_corners > 0) break;
        }
    }
    *n_out = n;
    if (timing) {
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "[pvio-hip] detect: %d candidates (%d contested) -> %d corners: kernels + copies %.1f us, ordering %.1f us, selection %.1f us\n", hs[1], nc, n, us(tt0, tt1),
                us(tt1, tt2), us(tt2, std::chrono::steady_clock::now()));
    }
    return PVIO_OK;
}

int Klt::download_response(const Image *img, float *resp) { // tests: the last detect()'s response map
    const size_t px = (size_t)img->w * img->h;
    if (!d_det_ || det_cap_ < px * 16) {
        err_ = "no response map (call detect first)";
        return PVIO_ERR_INVALID_ARGUMENT;
    }
    const float *r = static_cast<const float *>(d_det_) + 3 * px;
    if (hipMemcpy(resp, r, px * 4, hipMemcpyDeviceToHost) != hipSuccess) return PVIO_ERR_HIP;
    return PVIO_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Fundamental-matrix RANSAC, hypotheses in batches (opencv_image.cpp:113-129).
//
// cv::findFundamentalMat(FM_RANSAC) is a sequential hypothesise-and-verify loop, but the only thing a hypothesis needs from its
// predecessors is WHETHER it still has to be looked at (the adaptive iteration count): the samples come from a generator and the points
// alone (pv_fundamental.h: fm_draw_sample).  So the host draws the samples of a batch, the device turns every sample into its (up to
// three) models and scores each against all the matches -- one wave per sample: the 7-point solver runs redundantly in every lane, the
// lanes then stride over the points, one ballot per 64 points gives the inlier mask words and the count -- and the host replays the
// loop's bookkeeping over the counts in the original order (a model replaces the best when it has MORE inliers, the iteration count
// shrinks as in RANSACUpdateNumIters).  Hypotheses drawn beyond the point where the sequential loop stops are simply not looked at:
// the result is the sequential algorithm's.  Batches of 48, 192, then the rest: at the inlier ratios of a tracker (>= 80 %) the first
// batch ends the run.
// ------------------------------------------------------------------------------------------------------------------------------
struct FundArgs {
    int n, nw;                  // matches; 64-bit mask words per model
    const float *p, *q;         // [n][2]
    const float *samples;       // [H][28]: the seven sample points of either image
    double thr2;
    int *counts;                // [H][4]: number of models, inliers of each
    double *models;             // [H][27]
    unsigned long long *masks;  // [H][3][nw]
};

__global__ void __launch_bounds__(64) k_fund_hypotheses(FundArgs a) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const int h = blockIdx.x, lane = threadIdx.x;
    float sp[14], sq[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) sp[k] = a.samples[(size_t)h * 28 + k], sq[k] = a.samples[(size_t)h * 28 + 14 + k];
    double F[27];
    for (int k = 0; k < 27; ++k) F[k] = 0.0;
    const int nm = pvfm::fm_seven_point(sp, sq, F); // every lane the same arithmetic on the same numbers
    for (int m = 0; m < 3; ++m) {
        int count = 0;
        for (int i0 = 0; i0 < a.n; i0 += 64) {
            const int i = i0 + lane;
            int in = 0;
            if (i < a.n && m < nm) in = pvfm::fm_error(F + 9 * m, a.p[2 * i], a.p[2 * i + 1], a.q[2 * i], a.q[2 * i + 1]) <= a.thr2 ? 1 : 0;
            const unsigned long long word = __ballot(in);
            if (lane == 0) a.masks[((size_t)h * 3 + m) * a.nw + (i0 >> 6)] = word;
            count += __popcll(word);
        }
        if (lane == 0) a.counts[4 * h + 1 + m] = count;
    }
    if (lane == 0) a.counts[4 * h] = nm;
    if (lane < 27) a.models[(size_t)h * 27 + lane] = F[lane];
}

int Klt::fundamental_ransac(int n, const float *p, const float *q, double threshold, double confidence, int max_iterations, uint8_t *mask, double *F_out, int *n_inliers) {
    constexpr int kModel = 7;
    *n_inliers = 0;
    for (int i = 0; i < n; ++i) mask[i] = 0;
    last_fm_hypotheses_ = 0;
    if (n < kModel) return PVIO_OK;
    (void)hipSetDevice(device_);
    const int nw = (n + 63) / 64;
    const int cap_h = std::max(1, std::min(max_iterations, 1000));
    // one device block + its pinned mirror: [p | q | samples | counts | models | masks]
    const size_t off_q = (size_t)n * 8, off_s = off_q + (size_t)n * 8, off_c = (off_s + (size_t)cap_h * 28 * 4 + 15) & ~(size_t)15,
                 off_m = off_c + (size_t)cap_h * 16, off_k = off_m + (size_t)cap_h * 27 * 8, total = off_k + (size_t)cap_h * 3 * nw * 8;
    if (total > fm_cap_) {
        if (d_fm_) (void)hipFree(d_fm_);
        if (h_fm_) (void)hipHostFree(h_fm_);
        d_fm_ = h_fm_ = nullptr, fm_cap_ = 0;
        if (hipMalloc(&d_fm_, total) != hipSuccess || hipHostMalloc(&h_fm_, total) != hipSuccess) {
            err_ = "fundamental_ransac: allocation failed";
            return PVIO_ERR_OUT_OF_MEMORY;
        }
        fm_cap_ = total;
    }
    char *hb = static_cast<char *>(h_fm_), *db = static_cast<char *>(d_fm_);
    std::memcpy(hb, p, (size_t)n * 8), std::memcpy(hb + off_q, q, (size_t)n * 8);
    if (hipMemcpyAsync(db, hb, off_s, hipMemcpyHostToDevice, stream_) != hipSuccess) {
        err_ = "fundamental_ransac: upload failed";
        return PVIO_ERR_HIP;
    }
    pvfm::FmRng rng((uint64_t)-1);
    int niters = n == kModel ? 1 : max_iterations, iter = 0, max_good = 0;
    double bestF[9] = {0};
    std::vector<unsigned long long> best_words((size_t)nw, 0);
    float *hs = reinterpret_cast<float *>(hb + off_s);
    int batch = 48;
    bool exhausted = false;
    while (iter < niters && !exhausted) {
        int H = std::min(std::min(batch, niters - iter), cap_h);
        batch = batch == 48 ? 192 : cap_h;
        int drawn = 0;
        for (; drawn < H; ++drawn) {
            float *sp = hs + (size_t)drawn * 28, *sq = sp + 14;
            if (n > kModel) {
                if (!pvfm::fm_draw_sample(rng, n, p, q, sp, sq)) {
                    exhausted = true; // no admissible sample any more: the sequential loop stops where it gets here
                    break;
                }
            } else {
                std::memcpy(sp, p, 56), std::memcpy(sq, q, 56);
            }
        }
        H = drawn;
        if (H == 0) break;
        FundArgs a;
        a.n = n, a.nw = nw, a.p = reinterpret_cast<const float *>(db), a.q = reinterpret_cast<const float *>(db + off_q);
        a.samples = reinterpret_cast<const float *>(db + off_s), a.thr2 = threshold * threshold;
        a.counts = reinterpret_cast<int *>(db + off_c), a.models = reinterpret_cast<double *>(db + off_m), a.masks = reinterpret_cast<unsigned long long *>(db + off_k);
        bool ok = hipMemcpyAsync(db + off_s, hb + off_s, (size_t)H * 28 * 4, hipMemcpyHostToDevice, stream_) == hipSuccess;
        hipLaunchKernelGGL(k_fund_hypotheses, dim3(H), dim3(64), 0, stream_, a);
        // counts and models come back now; the mask words of ONE model are fetched once the replay knows which (they are 3 nw words per
        // hypothesis: copying them all would be most of the traffic)
        ok = ok && hipMemcpyAsync(hb + off_c, db + off_c, (off_m - off_c) + (size_t)H * 27 * 8, hipMemcpyDeviceToHost, stream_) == hipSuccess;
        ok = ok && hipStreamSynchronize(stream_) == hipSuccess && hipGetLastError() == hipSuccess;
        if (!ok) {
            err_ = "fundamental_ransac: batch failed";
            return PVIO_ERR_HIP;
        }
        last_fm_hypotheses_ += H;
        const int *cnt = reinterpret_cast<const int *>(hb + off_c);
        const double *models = reinterpret_cast<const double *>(hb + off_m);
        int win_h = -1, win_m = -1;
        for (int hh = 0; hh < H && iter < niters; ++hh, ++iter) {
            const int nm = cnt[4 * hh];
            for (int m = 0; m < nm; ++m) {
                const int good = cnt[4 * hh + 1 + m];
                if (good > std::max(max_good, kModel - 1)) {
                    win_h = hh, win_m = m, max_good = good;
                    std::memcpy(bestF, models + (size_t)hh * 27 + 9 * m, sizeof bestF);
                    niters = pvfm::fm_update_iterations(confidence, (double)(n - good) / n, kModel, niters);
                }
            }
        }
        if (win_h >= 0) { // the best model of the run so far lives in this batch: its mask words
            if (hipMemcpy(best_words.data(), a.masks + ((size_t)win_h * 3 + win_m) * nw, (size_t)nw * 8, hipMemcpyDeviceToHost) != hipSuccess) {
                err_ = "fundamental_ransac: mask download failed";
                return PVIO_ERR_HIP;
            }
        }
    }
    if (max_good > 0) {
        for (int i = 0; i < n; ++i) mask[i] = (uint8_t)((best_words[(size_t)i >> 6] >> (i & 63)) & 1ull);
        if (F_out) std::memcpy(F_out, bestF, sizeof bestF);
    }
    *n_inliers = max_good;
    return PVIO_OK;
}

} // namespace pvklt
