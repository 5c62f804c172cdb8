// rectify_kernels.hip -- SURVEY section 8(f-1): consumers of the solve's output on MI355X (gfx950).
//
//   back_project_claim_kernel / back_project_write_kernel
//       RsFrame::backProject / backProjectGs (reference rsframe.cc:803-878 with planeToSpace :644-664,
//       cameraToWorldFrame :712-736, worldToCameraFrame :687-709, spaceToPlane :628-641): forward splat of the BGR
//       rolling-shutter image into the global-shutter image through the depth map and the per-scanline pose table.
//       The reference's sequential double loop lets the LAST writer (largest row-major scan index) win; here every
//       source pixel claims its target with an integer atomicMax on the scan index (deterministic), then one pass
//       gathers the winners.  The claim map is a PERSISTENT per-context array whose words carry an 8-bit epoch above the
//       24-bit scan index: words of earlier frames lose against the current epoch, so no clearing pass runs per frame
//       (round 1 filled 4 B/pixel with -1 before every splat: 3.7 MB and one launch at 1280x720).  Byte / index work, HBM-bound: 3 B image + 8 B depth read, 3 B + 12 B (world point as
//       float3) written per pixel = 26 B/pixel algorithmic.
//       The depth map arrives column-major (Eigen MatrixXd, what depth_write_kernel produces) while the image is
//       row-major: each workgroup stages a 64 x 16 (x, y) tile of the depth map through LDS (coalesced along y),
//       then walks the tile along x (coalesced image reads / float3 writes).
//   interpolate_cracky_kernel
//       Camera::interpolateCrackyImage (camera.cc:694-774): 4-neighbour fill of black pixels.
//   preview_claim_minmax_kernel / preview_write_kernel
//       the 8-bit depth image of evaluateSingleRun (main.cc:480-509).
//
// Arithmetic mirrors oracle/rsdsfm_oracle.c (rso_back_project, rso_interpolate_cracky, rso_depth_preview) operation
// for operation; all outputs are integers / bytes (bit-exact) except the float3 world points (bit-exact as well: the
// per-pixel chain is identical and compiled with -ffp-contract=off).
#include <math.h>

#include <algorithm>

#include "device_math.hpp"
#include "rsdsfm_internal.hpp"

namespace rsdsfm {

namespace {

constexpr int kTX = 64;  // tile width  (image columns) = one wave
constexpr int kTY = 16;  // tile height (image rows = scanlines): 16 x 8 B = one 128-byte line of the column-major depth map per column
constexpr int kCB = 512; // threads of the claim kernel: 8 waves x 2 scanlines
constexpr int kBP = 256;

// double -> int like the reference's int(x) on x86-64 (cvttsd2si): non-finite / out of range -> INT_MIN
__device__ __forceinline__ int trunc_int(double x) {
    if (!(x > -2147483649.0 && x < 2147483648.0)) return INT32_MIN;
    return (int)x;
}

// cv::norm(Vec3b) <= 15 (camera.cc:694): sqrt(b^2 + g^2 + r^2) <= 15 in double.  The sum of squares is an integer
// below 2^18 and sqrt is correctly rounded and monotone (sqrt(225) = 15 exactly, sqrt(226) > 15), so the comparison is
// decided exactly by the integers.
__device__ __forceinline__ bool is_black(unsigned b, unsigned g, unsigned r) { return b * b + g * g + r * r <= 225u; }

__device__ __forceinline__ unsigned char saturate_u8(double v) {  // cvRound (nearest even) + clamp
    const long long r = __double2ll_rn(v);
    return (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

}  // namespace

// grid: (ceil(cols / kTX), ceil(rows / kTY)); owner: rows*cols claim words `tag | scan index`, row-major (see claim_map_acquire).
// A wave walks one scanline segment of 64 pixels at a time, so the scanline index is wave-uniform and its pose (12
// doubles) comes through the scalar data path; every pixel of the image is visited exactly once, so the world point of
// skipped (marker) pixels is zeroed here instead of by a separate memset.
__device__ __forceinline__ void back_project_claim_body(int bx, int by, const unsigned char* __restrict__ img,
                                                        const double* __restrict__ depth_cm, const double* __restrict__ R,
                                                        const double* __restrict__ t, double fx, double fy, double cx, double cy, double fyp,
                                                        int rows, int cols, int mode, unsigned* __restrict__ owner, unsigned tag,
                                                        float* __restrict__ c3d) {
    constexpr int RPW = kTY / (kCB / kTX);  // scanlines per wave
    __shared__ double s_z[kTX][kTY + 1];
    const int x0 = bx * kTX, y0 = by * kTY;
    const int tid = threadIdx.x;
    const int lx = tid & (kTX - 1);
    const int x = x0 + lx;
    const int wv = __builtin_amdgcn_readfirstlane(tid / kTX);
    // everything this thread needs from global memory is requested up front, so the latencies overlap:
    // (1) its share of the depth tile (lanes along y: contiguous in the column-major map)
    double zst[kTX * kTY / kCB];
    {
        const int ly = tid & (kTY - 1);
#pragma unroll
        for (int j = 0; j < kTX * kTY / kCB; ++j) {
            const int xx = x0 + tid / kTY + j * (kCB / kTY), yy = y0 + ly;
            zst[j] = (xx < cols && yy < rows) ? depth_cm[(int64_t)xx * rows + yy] : 0.0;
        }
    }
    // (2) the pixels of its scanlines, (3) their poses (wave-uniform -> scalar loads) and the pose of scanline 0
    unsigned pb[RPW], pg_[RPW], pr[RPW];
    double Rr[RPW][9], tr[RPW][3];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int y = y0 + wv + j * (kCB / kTX);
        const bool live = y < rows && x < cols;
        const int64_t s = live ? (int64_t)y * cols + x : 0;
        pb[j] = img[3 * s], pg_[j] = img[3 * s + 1], pr[j] = img[3 * s + 2];
        const int ys = (mode == 0 && y < rows) ? y : 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) Rr[j][i] = R[(int64_t)ys * 9 + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) tr[j][i] = t[(int64_t)ys * 3 + i];
    }
    double R0[9], t0[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R0[i] = R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) t0[i] = t[i];
    {
        const int ly = tid & (kTY - 1);
#pragma unroll
        for (int j = 0; j < kTX * kTY / kCB; ++j) s_z[tid / kTY + j * (kCB / kTY)][ly] = zst[j];
    }
    __syncthreads();
    if (x >= cols) return;
    const double nx = ((double)x - cx) * 1.0 / fx;
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int ly = wv + j * (kCB / kTX);
        const int y = y0 + ly;  // wave-uniform
        if (y >= rows) break;
        const int64_t s = (int64_t)y * cols + x;
        if (pb[j] == 1 && pg_[j] == 1 && pr[j] == 1) {  // marker colour: pixel not processed (rsframe.cc:816)
            if (c3d) c3d[3 * s] = 0.0f, c3d[3 * s + 1] = 0.0f, c3d[3 * s + 2] = 0.0f;
            continue;
        }
        const double ny = ((double)y - cy) * 1.0 / fy;
        const double z = s_z[lx][ly];
        const double pc0 = z * nx, pc1 = z * ny, pc2 = z * 1.0;
        double pw[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double rt0 = Rr[j][i], rt1 = Rr[j][3 + i], rt2 = Rr[j][6 + i];  // row i of R^T
            const double ti = ((-rt0) * tr[j][0] + (-rt1) * tr[j][1]) + (-rt2) * tr[j][2];
            pw[i] = ((rt0 * pc0 + rt1 * pc1) + rt2 * pc2) + ti * 1.0;
        }
        double pg[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) pg[i] = ((R0[i * 3] * pw[0] + R0[i * 3 + 1] * pw[1]) + R0[i * 3 + 2] * pw[2]) + t0[i] * 1.0;
        const double gx = pg[0] / pg[2] * fx + cx;
        const double gy = pg[1] / pg[2] * fyp + cy;
        if (c3d) {
            c3d[3 * s] = (float)pw[0];
            c3d[3 * s + 1] = (float)pw[1];
            c3d[3 * s + 2] = (float)pw[2];
        }
        const int ix = trunc_int(gx + 0.5), iy = trunc_int(gy + 0.5);
        if (ix >= 0 && ix < cols && iy >= 0 && iy < rows) atomicMax(&owner[(int64_t)iy * cols + ix], tag | (unsigned)s);
    }
}

__global__ __launch_bounds__(kCB) void back_project_claim_kernel(const unsigned char* __restrict__ img,
                                                                const double* __restrict__ depth_cm,
                                                                const double* __restrict__ R, const double* __restrict__ t,
                                                                double fx, double fy, double cx, double cy, double fyp, int rows,
                                                                int cols, int mode, unsigned* __restrict__ owner, unsigned tag,
                                                                float* __restrict__ c3d) {
    back_project_claim_body(blockIdx.x, blockIdx.y, img, depth_cm, R, t, fx, fy, cx, cy, fyp, rows, cols, mode, owner, tag, c3d);
}

// 4 target pixels (12 bytes) per thread
// a claim word is valid for this frame iff its bits above `mask` equal `tag`; its low bits are the winner's scan index
__device__ __forceinline__ void back_project_write_body(int block, int nblocks, const unsigned char* __restrict__ img,
                                                        const unsigned* __restrict__ owner, unsigned tag, unsigned mask, int64_t npix,
                                                        unsigned char* __restrict__ gs) {
    const int64_t stride = (int64_t)nblocks * kBP * 4;
    for (int64_t p0 = ((int64_t)block * kBP + threadIdx.x) * 4; p0 < npix; p0 += stride) {
        if (p0 + 4 <= npix) {
            const uint4 o = *reinterpret_cast<const uint4*>(owner + p0);  // p0 multiple of 4: 16-byte aligned
            const unsigned oo[4] = {o.x, o.y, o.z, o.w};
            unsigned v[12];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool hit = (oo[j] & ~mask) == tag;
                const int64_t src = 3 * (int64_t)(hit ? (oo[j] & mask) : 0u);
                const unsigned b = img[src], g = img[src + 1], r = img[src + 2];
                v[3 * j] = hit ? b : 0u;
                v[3 * j + 1] = hit ? g : 0u;
                v[3 * j + 2] = hit ? r : 0u;
            }
            unsigned* dst = reinterpret_cast<unsigned*>(gs + 3 * p0);  // 12 p0 bytes: 4-byte aligned
            dst[0] = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
            dst[1] = v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24);
            dst[2] = v[8] | (v[9] << 8) | (v[10] << 16) | (v[11] << 24);
        } else {
            for (int64_t p = p0; p < npix; ++p) {
                const unsigned w = owner[p];
                const bool hit = (w & ~mask) == tag;
                const int64_t o = (int64_t)(w & mask);
                gs[3 * p] = hit ? img[3 * o] : 0;
                gs[3 * p + 1] = hit ? img[3 * o + 1] : 0;
                gs[3 * p + 2] = hit ? img[3 * o + 2] : 0;
            }
        }
    }
}

__global__ __launch_bounds__(kBP) void back_project_write_kernel(const unsigned char* __restrict__ img,
                                                                const unsigned* __restrict__ owner, unsigned tag, unsigned mask,
                                                                int64_t npix, unsigned char* __restrict__ gs) {
    back_project_write_body(blockIdx.x, gridDim.x, img, owner, tag, mask, npix, gs);
}

// one pixel of the stencil; (b, g, r) hold the pixel's own colour on entry and the result on exit
__device__ __forceinline__ void interpolate_pixel(const unsigned char* __restrict__ in, int rows, int cols, int offset, int64_t p,
                                                  unsigned& b, unsigned& g, unsigned& r) {
    const int row = (int)(p / cols), col = (int)(p - (int64_t)row * cols);
    if (!(row >= offset && row < rows - offset && col >= offset && col < cols - offset && is_black(b, g, r))) return;
    const int64_t nb[4] = {p - (int64_t)offset * cols, p + (int64_t)offset * cols, p - offset, p + offset};
    double s0 = 0, s1 = 0, s2 = 0;
    unsigned count = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned nb0 = in[3 * nb[j]], nb1 = in[3 * nb[j] + 1], nb2 = in[3 * nb[j] + 2];
        if (!is_black(nb0, nb1, nb2)) {
            s0 += (double)nb0;
            s1 += (double)nb1;
            s2 += (double)nb2;
            count++;
        }
    }
    if (count > 0) {
        const double inv = 1 / (double)count;
        b = saturate_u8(inv * s0);
        g = saturate_u8(inv * s1);
        r = saturate_u8(inv * s2);
    }
}

// 4 pixels (12 bytes = 3 dwords) per thread: most pixels are not black and are simply copied
__global__ __launch_bounds__(kBP) void interpolate_cracky_kernel(const unsigned char* __restrict__ in, int rows, int cols,
                                                                int offset, unsigned char* __restrict__ out) {
    const int64_t npix = (int64_t)rows * cols;
    const int64_t stride = (int64_t)gridDim.x * kBP * 4;
    for (int64_t p0 = ((int64_t)blockIdx.x * kBP + threadIdx.x) * 4; p0 < npix; p0 += stride) {
        if (p0 + 4 <= npix) {
            const unsigned* src = reinterpret_cast<const unsigned*>(in + 3 * p0);  // 12 p0 bytes: 4-byte aligned
            unsigned w[3] = {src[0], src[1], src[2]};
            unsigned v[12];
#pragma unroll
            for (int j = 0; j < 12; ++j) v[j] = (w[j >> 2] >> (8 * (j & 3))) & 0xffu;
#pragma unroll
            for (int j = 0; j < 4; ++j) interpolate_pixel(in, rows, cols, offset, p0 + j, v[3 * j], v[3 * j + 1], v[3 * j + 2]);
            unsigned* dst = reinterpret_cast<unsigned*>(out + 3 * p0);
            dst[0] = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
            dst[1] = v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24);
            dst[2] = v[8] | (v[9] << 8) | (v[10] << 16) | (v[11] << 24);
        } else {
            for (int64_t p = p0; p < npix; ++p) {
                unsigned b = in[3 * p], g = in[3 * p + 1], r = in[3 * p + 2];
                interpolate_pixel(in, rows, cols, offset, p, b, g, r);
                out[3 * p] = (unsigned char)b;
                out[3 * p + 1] = (unsigned char)g;
                out[3 * p + 2] = (unsigned char)r;
            }
        }
    }
}

// ---- 8-bit depth preview (main.cc:480-509) ----
// One pass over the inliers does both jobs of the reference's two loops (main.cc:484-495 min / max of z, :499-507 the splat):
//  * partials[2 * block] = min z, [2 * block + 1] = max z (start values +inf / 0 as in the reference; min / max are exact in any order);
//  * owner: cols*rows claim words COLUMN-major (x * rows + y, the order the inliers arrive in: coalesced atomics); the highest
//    inlier index wins (the reference's last writer).
// (round 2 first ran these as two kernels, each streaming the inlier array: 7.3 + 4.2 us at 1280x720)
template <int BS>
__device__ __forceinline__ void preview_claim_minmax_body(int block, int nblocks, const double* __restrict__ inl, int64_t m, double fx, double fy,
                                                          double cx, double cy, int rows, int cols, unsigned* __restrict__ owner, unsigned tag,
                                                          double* __restrict__ partials) {
    __shared__ double s_min[BS / 64], s_max[BS / 64];
    double lo = INFINITY, hi = 0.0;
    const int64_t stride = (int64_t)nblocks * BS;
    for (int64_t i = (int64_t)block * BS + threadIdx.x; i < m; i += stride) {
        const double qx = inl[3 * i], qy = inl[3 * i + 1], z = inl[3 * i + 2];
        if (z < lo) lo = z;
        if (z > hi) hi = z;
        const int x = (int)(fx * qx + cx + 0.5);
        const int y = (int)(fy * qy + cy + 0.5);
        if (x >= 0 && x < cols && y >= 0 && y < rows) atomicMax(&owner[(int64_t)x * rows + y], tag | (unsigned)i);
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const double ol = __shfl_xor(lo, off, 64), oh = __shfl_xor(hi, off, 64);
        if (ol < lo) lo = ol;
        if (oh > hi) hi = oh;
    }
    if ((threadIdx.x & 63) == 0) {
        s_min[threadIdx.x >> 6] = lo;
        s_max[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w2 = 1; w2 < BS / 64; ++w2) {
            if (s_min[w2] < lo) lo = s_min[w2];
            if (s_max[w2] > hi) hi = s_max[w2];
        }
        partials[2 * block] = lo;
        partials[2 * block + 1] = hi;
    }
}

__global__ __launch_bounds__(kBP) void preview_claim_minmax_kernel(const double* __restrict__ inl, int64_t m, double fx, double fy, double cx,
                                                                  double cy, int rows, int cols, unsigned* __restrict__ owner, unsigned tag,
                                                                  double* __restrict__ partials) {
    preview_claim_minmax_body<kBP>(blockIdx.x, gridDim.x, inl, m, fx, fy, cx, cy, rows, cols, owner, tag, partials);
}

// grid: (ceil(cols / 32), ceil(rows / 32)).  Every workgroup first reduces the (<= 1024) min / max partials of
// preview_claim_minmax_kernel itself -- min / max are exact in any order, so the redundant reduction replaces the single-workgroup
// header kernel of round 1 -- then reads the column-major owner tile along y (coalesced; the winners of neighbouring pixels are
// neighbouring inliers, so the z gather is local too), transposes the bytes through LDS and writes the row-major 8-bit image
// along x.
__device__ __forceinline__ void preview_write_body(int bx, int by, const double* __restrict__ inl, const unsigned* __restrict__ owner, unsigned tag,
                                                   unsigned mask, const double* __restrict__ partials, int nblocks, int rows, int cols,
                                                   unsigned char* __restrict__ out) {
    constexpr int T = 32;
    __shared__ unsigned char s_v[T][T + 4];  // [y][x]
    __shared__ double s_min[kBP / 64], s_max[kBP / 64];
    const int tid = threadIdx.x;
    double z_min, mult;
    {
        double lo = INFINITY, hi = 0.0;
        for (int b = tid; b < nblocks; b += kBP) {
            if (partials[2 * b] < lo) lo = partials[2 * b];
            if (partials[2 * b + 1] > hi) hi = partials[2 * b + 1];
        }
        for (int off = 32; off >= 1; off >>= 1) {
            const double ol = __shfl_xor(lo, off, 64), oh = __shfl_xor(hi, off, 64);
            if (ol < lo) lo = ol;
            if (oh > hi) hi = oh;
        }
        if ((tid & 63) == 0) {
            s_min[tid >> 6] = lo;
            s_max[tid >> 6] = hi;
        }
        __syncthreads();
        lo = s_min[0], hi = s_max[0];
        for (int w2 = 1; w2 < kBP / 64; ++w2) {
            if (s_min[w2] < lo) lo = s_min[w2];
            if (s_max[w2] > hi) hi = s_max[w2];
        }
        z_min = lo;
        mult = 244.0 / (hi - lo);  // main.cc:497
    }
    const int x0 = bx * T, y0 = by * T;
    {
        const int ly = tid & (T - 1);
        const int y = y0 + ly;
        int o[T * T / kBP];
#pragma unroll
        for (int j = 0; j < T * T / kBP; ++j) {
            const int x = x0 + (tid / T) + j * (kBP / T);
            const unsigned w = (x < cols && y < rows) ? owner[(int64_t)x * rows + y] : 0u;
            o[j] = (x < cols && y < rows && (w & ~mask) == tag) ? (int)(w & mask) : -1;
        }
        double z[T * T / kBP];
#pragma unroll
        for (int j = 0; j < T * T / kBP; ++j) z[j] = o[j] >= 0 ? inl[3 * (int64_t)o[j] + 2] : 0.0;
#pragma unroll
        for (int j = 0; j < T * T / kBP; ++j) {
            unsigned char v = 0;
            if (o[j] >= 0) {
                int zi = trunc_int((z[j] - z_min) * mult);
                if (zi == INT32_MIN) zi = 0;
                v = (unsigned char)(10 + zi);  // int -> uchar: modulo 256, like the reference's assignment
            }
            s_v[ly][(tid / T) + j * (kBP / T)] = v;
        }
    }
    __syncthreads();
    const int lx = tid & (T - 1);
    const int x = x0 + lx;
    if (x >= cols) return;
#pragma unroll
    for (int j = 0; j < T * T / kBP; ++j) {
        const int ly = (tid / T) + j * (kBP / T);
        const int y = y0 + ly;
        if (y < rows) out[(int64_t)y * cols + x] = s_v[ly][lx];
    }
}

__global__ __launch_bounds__(kBP) void preview_write_kernel(const double* __restrict__ inl, const unsigned* __restrict__ owner, unsigned tag,
                                                           unsigned mask, const double* __restrict__ partials, int nblocks, int rows,
                                                           int cols, unsigned char* __restrict__ out) {
    preview_write_body(blockIdx.x, blockIdx.y, inl, owner, tag, mask, partials, nblocks, rows, cols, out);
}

// ---- back-projection write pass + crack interpolation on one tile (rsdsfm_rectify_frame_dev with offset <= kMaxHalo) ------------------
// The interpolation (camera.cc:694-774) only reads the global-shutter image at the pixel and its four neighbours `offset` away, so a
// workgroup that forms a kWT x kHT tile of that image plus a halo of `offset` pixels from the claim map (LDS: one packed BGR word per pixel)
// can write the tile of BOTH images: no launch of its own for the interpolation, and the global-shutter image is not read back.
constexpr int kWT = 64, kHT = 16, kMaxHalo = 2;
__device__ __forceinline__ void write_interpolate_body(int bx, int by, const unsigned char* __restrict__ img, const unsigned* __restrict__ owner,
                                                       unsigned tag, unsigned mask, int rows, int cols, int offset, unsigned char* __restrict__ gs,
                                                       unsigned char* __restrict__ fixed) {
    __shared__ unsigned s_px[(kHT + 2 * kMaxHalo) * (kWT + 2 * kMaxHalo)];
    const int tid = threadIdx.x;
    const int h = offset;
    const int x0 = bx * kWT - h, y0 = by * kHT - h, W = kWT + 2 * h, H = kHT + 2 * h;
    for (int i = tid; i < W * H; i += kBP) {  // the tile and its halo: the winner's colour, 0 where nobody landed or outside the image
        const int ly = i / W, lx = i - ly * W;
        const int x = x0 + lx, y = y0 + ly;
        unsigned v = 0u;
        if (x >= 0 && x < cols && y >= 0 && y < rows) {
            const unsigned w = owner[(int64_t)y * cols + x];
            if ((w & ~mask) == tag) {
                const int64_t src = 3 * (int64_t)(w & mask);
                v = (unsigned)img[src] | ((unsigned)img[src + 1] << 8) | ((unsigned)img[src + 2] << 16);
            }
        }
        s_px[i] = v;
    }
    __syncthreads();
    // 4 consecutive pixels of a tile row per thread (kWT / 4 threads per row, kHT rows): 12 bytes = 3 dwords of each image
    const int ly = tid / (kWT / 4), lx4 = (tid - ly * (kWT / 4)) * 4;
    const int y = by * kHT + ly, xb = bx * kWT + lx4;
    if (y >= rows || xb >= cols) return;
    unsigned g[12], f[12];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = xb + j;
        const unsigned c0 = s_px[(ly + h) * W + (lx4 + j + h)];
        unsigned b = c0 & 0xffu, gg = (c0 >> 8) & 0xffu, r = c0 >> 16;
        g[3 * j] = b, g[3 * j + 1] = gg, g[3 * j + 2] = r;
        if (x < cols && y >= offset && y < rows - offset && x >= offset && x < cols - offset && is_black(b, gg, r)) {  // interpolate_pixel on the tile
            const unsigned nb[4] = {s_px[(ly + h - offset) * W + (lx4 + j + h)], s_px[(ly + h + offset) * W + (lx4 + j + h)],
                                    s_px[(ly + h) * W + (lx4 + j + h - offset)], s_px[(ly + h) * W + (lx4 + j + h + offset)]};
            double s0 = 0, s1 = 0, s2 = 0;
            unsigned count = 0;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                const unsigned n0 = nb[k2] & 0xffu, n1 = (nb[k2] >> 8) & 0xffu, n2 = nb[k2] >> 16;
                if (!is_black(n0, n1, n2)) {
                    s0 += (double)n0;
                    s1 += (double)n1;
                    s2 += (double)n2;
                    count++;
                }
            }
            if (count > 0) {
                const double inv = 1 / (double)count;
                b = saturate_u8(inv * s0);
                gg = saturate_u8(inv * s1);
                r = saturate_u8(inv * s2);
            }
        }
        f[3 * j] = b, f[3 * j + 1] = gg, f[3 * j + 2] = r;
    }
    const int64_t p0 = (int64_t)y * cols + xb;
    if (xb + 4 <= cols && ((3 * p0) & 3) == 0) {
        unsigned* dg = reinterpret_cast<unsigned*>(gs + 3 * p0);
        unsigned* df = reinterpret_cast<unsigned*>(fixed + 3 * p0);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            dg[d] = g[4 * d] | (g[4 * d + 1] << 8) | (g[4 * d + 2] << 16) | (g[4 * d + 3] << 24);
            df[d] = f[4 * d] | (f[4 * d + 1] << 8) | (f[4 * d + 2] << 16) | (f[4 * d + 3] << 24);
        }
    } else {
        for (int j = 0; j < 4 && xb + j < cols; ++j)
            for (int c2 = 0; c2 < 3; ++c2) {
                gs[3 * (p0 + j) + c2] = (unsigned char)g[3 * j + c2];
                fixed[3 * (p0 + j) + c2] = (unsigned char)f[3 * j + c2];
            }
    }
}

__global__ __launch_bounds__(kBP) void rectify_write_interpolate_kernel(const unsigned char* __restrict__ img, const unsigned* __restrict__ owner_bp,
                                                                       unsigned tag_bp, unsigned mask_bp, int rows, int cols, int offset,
                                                                       unsigned char* __restrict__ gs, unsigned char* __restrict__ fixed, int tiles_x,
                                                                       int nb_t, const double* __restrict__ inl, const unsigned* __restrict__ owner_pv,
                                                                       unsigned tag_pv, unsigned mask_pv, const double* __restrict__ partials,
                                                                       int nrows_pv, int tiles_x_pv, unsigned char* __restrict__ preview) {
    const int b = blockIdx.x;
    if (b < nb_t)
        write_interpolate_body(b % tiles_x, b / tiles_x, img, owner_bp, tag_bp, mask_bp, rows, cols, offset, gs, fixed);
    else
        preview_write_body((b - nb_t) % tiles_x_pv, (b - nb_t) / tiles_x_pv, inl, owner_pv, tag_pv, mask_pv, partials, nrows_pv, rows, cols, preview);
}

// ---- main.cc:480-523 in three launches instead of five (rsdsfm_rectify_frame_dev) ----------------------------------------------------
// The depth image and the back projection are independent chains (inliers -> claim map 1 -> 8-bit image; image + depth map -> claim map 0
// -> global-shutter image) of two launches each, every one of them short enough for the launch floor to show: the two claim passes share
// ONE launch (its first workgroups walk the back projection's tiles, the rest the inlier list) and so do the two write passes.  Same bodies,
// same results; which workgroup does what depends on the block index alone.
__global__ __launch_bounds__(kCB) void rectify_claim_kernel(const unsigned char* __restrict__ img, const double* __restrict__ depth_cm,
                                                           const double* __restrict__ R, const double* __restrict__ t, double fx, double fy,
                                                           double cx, double cy, double fyp, int rows, int cols, int mode,
                                                           unsigned* __restrict__ owner_bp, unsigned tag_bp, float* __restrict__ c3d, int tiles_x,
                                                           int nb_bp, const double* __restrict__ inl, int64_t m, unsigned* __restrict__ owner_pv,
                                                           unsigned tag_pv, double* __restrict__ partials) {
    const int b = blockIdx.x;
    if (b < nb_bp)
        back_project_claim_body(b % tiles_x, b / tiles_x, img, depth_cm, R, t, fx, fy, cx, cy, fyp, rows, cols, mode, owner_bp, tag_bp, c3d);
    else
        preview_claim_minmax_body<kCB>(b - nb_bp, (int)gridDim.x - nb_bp, inl, m, fx, fy, cx, cy, rows, cols, owner_pv, tag_pv, partials);
}

__global__ __launch_bounds__(kBP) void rectify_write_kernel(const unsigned char* __restrict__ img, const unsigned* __restrict__ owner_bp, unsigned tag_bp,
                                                           unsigned mask_bp, int64_t npix, unsigned char* __restrict__ gs, int nb_w,
                                                           const double* __restrict__ inl, const unsigned* __restrict__ owner_pv, unsigned tag_pv,
                                                           unsigned mask_pv, const double* __restrict__ partials, int nrows_pv, int rows, int cols,
                                                           int tiles_x_pv, unsigned char* __restrict__ preview) {
    const int b = blockIdx.x;
    if (b < nb_w)
        back_project_write_body(b, nb_w, img, owner_bp, tag_bp, mask_bp, npix, gs);
    else
        preview_write_body((b - nb_w) % tiles_x_pv, (b - nb_w) / tiles_x_pv, inl, owner_pv, tag_pv, mask_pv, partials, nrows_pv, rows, cols, preview);
}

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
static inline int stream_grid(int64_t n, int per_thread = 1) {
    int64_t b = (n + (int64_t)kBP * per_thread - 1) / ((int64_t)kBP * per_thread);
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// Persistent claim map `which` (0: back projection, 1: depth image, 2: the solve's depth map) of the context with room for npix words.  Words are
// `tag | index`: tag = epoch << 24 with a per-call epoch 1..255 and index < 2^24 -- a word written by an earlier call carries an
// older epoch, loses every atomicMax against the current one and is ignored by the read-back (valid iff word & ~mask == tag), so
// the map is cleared only when it is (re)allocated and when the epoch wraps (every 255 calls).  Images beyond 2^24 pixels fall
// back to a clear per call with a 1-bit tag.
// room for `words` claim words in map `which`, without starting a new epoch (the tiled solve allocates in its setup part)
int claim_map_reserve(Ctx* c, int which, size_t words) {
    if (words <= c->claim_words[which]) return RSDSFM_OK;
    RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (c->d_claim[which]) RSDSFM_HIP_CHECK(c, hipFree(c->d_claim[which]));
    c->d_claim[which] = nullptr;
    c->claim_words[which] = 0;
    RSDSFM_HIP_CHECK(c, hipMalloc(reinterpret_cast<void**>(&c->d_claim[which]), sizeof(unsigned) * words));
    c->claim_words[which] = words;
    c->claim_epoch[which] = 255u;  // the next acquire starts from a cleared map
    return RSDSFM_OK;
}

int claim_map_acquire(Ctx* c, int which, size_t npix, unsigned** map, unsigned* tag, unsigned* mask) {
    bool clear = false;
    if (npix > c->claim_words[which]) {
        RSDSFM_HIP_CHECK(c, hipStreamSynchronize(c->stream));
        if (c->d_claim[which]) RSDSFM_HIP_CHECK(c, hipFree(c->d_claim[which]));
        c->d_claim[which] = nullptr;
        c->claim_words[which] = 0;
        RSDSFM_HIP_CHECK(c, hipMalloc(reinterpret_cast<void**>(&c->d_claim[which]), sizeof(unsigned) * npix));
        c->claim_words[which] = npix;
        c->claim_epoch[which] = 0;
        clear = true;
    }
    const bool wide = npix > ((size_t)1 << 24);
    if (wide) {
        clear = true;
        *tag = 0x80000000u;
        *mask = 0x7FFFFFFFu;
        // a wide call leaves words tagged 0x80000000 | index behind, which would beat every epoch tag 0x01.. - 0x7F.. of a later
        // narrow call on the same map: the next narrow call must start from a cleared map (the epoch wrap below does that)
        c->claim_epoch[which] = 255u;
    } else {
        if (c->claim_epoch[which] >= 255u) {
            c->claim_epoch[which] = 0;
            clear = true;
        }
        c->claim_epoch[which] += 1;
        *tag = c->claim_epoch[which] << 24;
        *mask = 0x00FFFFFFu;
    }
    if (clear) RSDSFM_HIP_CHECK(c, hipMemsetAsync(c->d_claim[which], 0, sizeof(unsigned) * c->claim_words[which], c->stream));
    *map = c->d_claim[which];
    return RSDSFM_OK;
}

int back_project_launch(Ctx* c, const unsigned char* d_img, const double* d_depth_cm, const double* d_R, const double* d_t, double fx,
                        double fy, double cx, double cy, int rows, int cols, int mode, int q5_mode, unsigned char* d_gs, float* d_c3d) {
    const int64_t npix = (int64_t)rows * cols;
    unsigned *d_owner = nullptr, tag = 0, mask = 0;
    int rc = claim_map_acquire(c, 0, (size_t)npix, &d_owner, &tag, &mask);
    if (rc != RSDSFM_OK) return rc;
    dim3 grid((cols + kTX - 1) / kTX, (rows + kTY - 1) / kTY);
    hipLaunchKernelGGL(back_project_claim_kernel, grid, dim3(kCB), 0, c->stream, d_img, d_depth_cm, d_R, d_t, fx, fy, cx, cy,
                       q5_mode == 0 ? fx : fy, rows, cols, mode, d_owner, tag, d_c3d);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(back_project_write_kernel, dim3(stream_grid(npix, 4)), dim3(kBP), 0, c->stream, d_img, d_owner, tag, mask, npix, d_gs);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

int interpolate_cracky_launch(Ctx* c, const unsigned char* d_in, int rows, int cols, int offset, unsigned char* d_out) {
    hipLaunchKernelGGL(interpolate_cracky_kernel, dim3(stream_grid((int64_t)rows * cols, 4)), dim3(kBP), 0, c->stream, d_in, rows, cols,
                       offset, d_out);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

// d_partials: >= 2 * 1024 doubles
int depth_preview_launch(Ctx* c, const double* d_inl, int64_t m, double fx, double fy, double cx, double cy, int rows, int cols,
                         unsigned char* d_out, double* d_partials) {
    const int64_t npix = (int64_t)rows * cols;
    const int zb = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (m + kBP - 1) / kBP));
    unsigned *d_owner = nullptr, tag = 0, mask = 0;
    if (m >= ((int64_t)1 << 31)) return fail(c, RSDSFM_ERR_INVALID, "depth image: more than 2^31 inliers");
    // the claim word holds the INLIER index: the map's index field is sized by the larger of the two counts (as depth_map_slab_launch)
    int rc = claim_map_acquire(c, 1, (size_t)std::max<int64_t>(std::max<int64_t>(npix, m), 1), &d_owner, &tag, &mask);
    if (rc != RSDSFM_OK) return rc;
    hipLaunchKernelGGL(preview_claim_minmax_kernel, dim3(zb), dim3(kBP), 0, c->stream, d_inl, m, fx, fy, cx, cy, rows, cols, d_owner, tag,
                       d_partials);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(preview_write_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(kBP), 0, c->stream, d_inl, d_owner, tag, mask,
                       d_partials, zb, rows, cols, d_out);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

// main.cc:480-523: 8-bit depth image + back projection + crack interpolation; d_partials: >= 2 * 1024 doubles
int rectify_frame_launch(Ctx* c, const double* d_inl, int64_t m, const unsigned char* d_img, const double* d_depth_cm, const double* d_R,
                         const double* d_t, double fx, double fy, double cx, double cy, int rows, int cols, int mode, int q5_mode, int offset,
                         unsigned char* d_preview, unsigned char* d_gs, float* d_c3d, unsigned char* d_fixed, double* d_partials) {
    const int64_t npix = (int64_t)rows * cols;
    if (m >= ((int64_t)1 << 31)) return fail(c, RSDSFM_ERR_INVALID, "depth image: more than 2^31 inliers");
    unsigned *d_owner_bp = nullptr, tag_bp = 0, mask_bp = 0, *d_owner_pv = nullptr, tag_pv = 0, mask_pv = 0;
    int rc = claim_map_acquire(c, 0, (size_t)npix, &d_owner_bp, &tag_bp, &mask_bp);
    if (rc != RSDSFM_OK) return rc;
    rc = claim_map_acquire(c, 1, (size_t)std::max<int64_t>(std::max<int64_t>(npix, m), 1), &d_owner_pv, &tag_pv, &mask_pv);
    if (rc != RSDSFM_OK) return rc;
    const int tiles_x = (cols + kTX - 1) / kTX, tiles_y = (rows + kTY - 1) / kTY;
    const int nb_bp = tiles_x * tiles_y;
    const int zb = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (m + kCB - 1) / kCB));
    hipLaunchKernelGGL(rectify_claim_kernel, dim3(nb_bp + zb), dim3(kCB), 0, c->stream, d_img, d_depth_cm, d_R, d_t, fx, fy, cx, cy,
                       q5_mode == 0 ? fx : fy, rows, cols, mode, d_owner_bp, tag_bp, d_c3d, tiles_x, nb_bp, d_inl, m, d_owner_pv, tag_pv, d_partials);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    const int ptx = (cols + 31) / 32, pty = (rows + 31) / 32;
    if (offset <= kMaxHalo && (cols % 4) == 0) {  // two launches: the write pass forms the interpolated image from its own tile (+ halo)
        const int ttx = (cols + kWT - 1) / kWT, tty = (rows + kHT - 1) / kHT;
        hipLaunchKernelGGL(rectify_write_interpolate_kernel, dim3(ttx * tty + ptx * pty), dim3(kBP), 0, c->stream, d_img, d_owner_bp, tag_bp, mask_bp, rows,
                           cols, offset, d_gs, d_fixed, ttx, ttx * tty, d_inl, d_owner_pv, tag_pv, mask_pv, d_partials, zb, ptx, d_preview);
        RSDSFM_HIP_CHECK(c, hipGetLastError());
        return RSDSFM_OK;
    }
    const int nb_w = stream_grid(npix, 4);
    hipLaunchKernelGGL(rectify_write_kernel, dim3(nb_w + ptx * pty), dim3(kBP), 0, c->stream, d_img, d_owner_bp, tag_bp, mask_bp, npix, d_gs, nb_w,
                       d_inl, d_owner_pv, tag_pv, mask_pv, d_partials, zb, rows, cols, ptx, d_preview);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return interpolate_cracky_launch(c, d_gs, rows, cols, offset, d_fixed);
}

}  // namespace rsdsfm
