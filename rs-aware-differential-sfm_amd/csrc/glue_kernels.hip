// glue_kernels.hip -- RS scale factors, per-scanline pose table.
//   minimal::getAlpha / getAlphaK   reference minimal.cc:179-197
//   RsFrame::setRelativePose        reference rsframe.cc:771-800
#include <string.h>

#include <algorithm>

#include "device_math.hpp"
#include "rsdsfm_internal.hpp"

namespace rsdsfm {

__global__ __launch_bounds__(256) void alpha_kernel(const double2* __restrict__ flow_px, int64_t n, double h,
                                                    double gamma, double* __restrict__ alpha) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        alpha[i] = 1 + gamma * flow_px[i].y / h;
}

__global__ __launch_bounds__(256) void alpha_k_kernel(const double2* __restrict__ q_px, const double2* __restrict__ flow_px,
                                                      int64_t n, double h, double gamma, double* __restrict__ alpha_k) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double y = q_px[i].y, fy = flow_px[i].y;
        double part1 = gamma * y / h;
        double part2 = 1.0 + gamma * (y + fy) / h;
        alpha_k[i] = 0.5 * (part2 * part2 - part1 * part1);
    }
}

// one lane per scanline: R_i = I + beta_1(i) skew(w), t_i = beta_1(i) v  (scanline 0 = identity)
// v_dev (optional): the translation is taken from device memory (the {flipped, v'} header the depth-map stage leaves behind)
// instead of `pose.v`, so that the table can be enqueued behind that stage without a host round trip
// wk_dev (optional): (w, k) likewise from device memory (RefineState::p + 3: the table is then enqueued before the host has read the
// refinement's result)
// one scanline of the table (rsframe.cc:771-800)
__device__ __forceinline__ void pose_table_row(const Pose& pose, double gamma, int rows, int i, double* __restrict__ R, double* __restrict__ t) {
    double beta_1 = 0.0;
    if (i > 0)
        beta_1 = (gamma * i / rows + 0.5 * pose.k * (gamma * gamma * i * i) / ((double)rows * rows)) * (2.0 / (2.0 + pose.k));
    double* Ri = R + (int64_t)i * 9;
    Ri[0] = 1.0 + beta_1 * 0.0;
    Ri[1] = 0.0 + beta_1 * -pose.w[2];
    Ri[2] = 0.0 + beta_1 * pose.w[1];
    Ri[3] = 0.0 + beta_1 * pose.w[2];
    Ri[4] = 1.0 + beta_1 * 0.0;
    Ri[5] = 0.0 + beta_1 * -pose.w[0];
    Ri[6] = 0.0 + beta_1 * -pose.w[1];
    Ri[7] = 0.0 + beta_1 * pose.w[0];
    Ri[8] = 1.0 + beta_1 * 0.0;
    t[(int64_t)i * 3 + 0] = 0.0 + beta_1 * pose.v[0];
    t[(int64_t)i * 3 + 1] = 0.0 + beta_1 * pose.v[1];
    t[(int64_t)i * 3 + 2] = 0.0 + beta_1 * pose.v[2];
}

__global__ __launch_bounds__(256) void pose_table_kernel(Pose pose, double gamma, int rows, double* __restrict__ R,
                                                         double* __restrict__ t, const double* __restrict__ v_dev,
                                                         const double* __restrict__ wk_dev) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    if (v_dev) pose.v[0] = v_dev[0], pose.v[1] = v_dev[1], pose.v[2] = v_dev[2];
    if (wk_dev) pose.w[0] = wk_dev[0], pose.w[1] = wk_dev[1], pose.w[2] = wk_dev[2], pose.k = wk_dev[3];
    pose_table_row(pose, gamma, rows, i, R, t);
}

static inline int stream_grid(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

// global-shutter override of evaluateSingleRun (main.cc:441-444): alpha = alpha * 0 + 1 for every point, evaluated literally
// (a NaN / inf alpha stays NaN exactly like `alpha *= 0; alpha += 1` on the Eigen array)
__global__ __launch_bounds__(256) void alpha_ones_kernel(double* __restrict__ alpha, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) alpha[i] = alpha[i] * 0.0 + 1.0;
}

int alpha_ones_launch(Ctx* c, double* d_alpha, int64_t n) {
    if (n <= 0) return RSDSFM_OK;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, (int64_t)c->num_cus * 8);
    hipLaunchKernelGGL(alpha_ones_kernel, dim3(grid), dim3(256), 0, c->stream, d_alpha, n);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

int alpha_launch(Ctx* c, const double* flow_px, int64_t n, double h, double gamma, double* alpha) {
    if (n == 0) return RSDSFM_OK;
    hipLaunchKernelGGL(alpha_kernel, dim3(stream_grid(n)), dim3(256), 0, c->stream,
                       reinterpret_cast<const double2*>(flow_px), n, h, gamma, alpha);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

int alpha_k_launch(Ctx* c, const double* q_px, const double* flow_px, int64_t n, double h, double gamma, double* alpha_k) {
    if (n == 0) return RSDSFM_OK;
    hipLaunchKernelGGL(alpha_k_kernel, dim3(stream_grid(n)), dim3(256), 0, c->stream,
                       reinterpret_cast<const double2*>(q_px), reinterpret_cast<const double2*>(flow_px), n, h, gamma, alpha_k);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

int pose_table_launch(Ctx* c, const Pose& pose, double gamma, int rows, double* R, double* t, const double* v_dev, const double* wk_dev) {
    if (rows <= 0) return RSDSFM_OK;
    hipLaunchKernelGGL(pose_table_kernel, dim3((rows + 255) / 256), dim3(256), 0, c->stream, pose, gamma, rows, R, t, v_dev, wk_dev);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

}  // namespace rsdsfm

// =====================================================================================================
// caller-side glue of the hot path
//   flatten   : reference main.cc:398-444 / errorMeasure.cpp:66-111 (shrinking variant) fused with
//               minimal::getAlpha / getAlphaK (minimal.cc:179-197)
//   depth map : reference main.cc:466-509 (mean-z sign flip, scatter of inlier depths)
// =====================================================================================================
namespace rsdsfm {

namespace {
constexpr int kGB = 256;

}  // namespace

// ---- flatten, tiled: coalesced reads of the row-major image ----
// The scan order of the reference is column-major (main.cc:408-432) while the flow image is row-major, so walking the scan
// positions directly reads 16 bytes per lane at a stride of one image row.  Here the image is cut into tiles of
// kFT_W columns x kFT_H rows; a workgroup loads its tile with full 128-byte row segments into LDS and one WAVE compacts
// one CELL = 64 consecutive rows of one column (ballot / popcount).  Cells are numbered column-major (cell = column *
// chunks + chunk), so an exclusive scan of the cell counts yields every cell's base in the reference's order.
constexpr int kScanSegCells = 2048;  // cells per segment of the two-level scan
constexpr int kFT_W = 8;   // tile columns: 8 x 16 B = one 128-byte line per image row
constexpr int kFT_H = 64;  // tile rows = cell height = one wave

template <int SCATTER>
__global__ __launch_bounds__(kGB) void flatten_tile_kernel(const double2* __restrict__ img, int rows, int cols, double fx, double fy,
                                                          double cx, double cy, double gamma, double thr, int col0, int nchunks,
                                                          int64_t* __restrict__ cell_counts,
                                                          const int64_t* __restrict__ cell_offsets,
                                                          const int64_t* __restrict__ seg_bases, double2* __restrict__ q,
                                                          double2* __restrict__ u, double* __restrict__ alpha,
                                                          double* __restrict__ alpha_k, int nseg, int64_t* __restrict__ total_out) {
    // SCATTER: 0 = count the cells, 1 = scatter with scanned segment bases, 2 = scatter with the raw segment TOTALS of
    // cell_scan_local_kernel in `seg_bases` (nseg of them: a handful -- every wave adds up the ones in front of its cell itself, and
    // the first workgroup stores the grand total; the frame solve's variant: no launch for the second scan level)
    __shared__ double2 s_tile[kFT_H][kFT_W + 1];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c0 = blockIdx.x * kFT_W, r0 = blockIdx.y * kFT_H;
    if (SCATTER == 2 && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        int64_t run = 0;
        for (int sgi = 0; sgi < nseg; ++sgi) run += seg_bases[sgi];
        *total_out = run;
    }
#pragma unroll
    for (int k = 0; k < kFT_H * kFT_W / kGB; ++k) {
        const int lr = tid / kFT_W + k * (kGB / kFT_W), lc = tid % kFT_W;
        const int r = r0 + lr, c = c0 + lc;
        s_tile[lr][lc] = (r < rows && c < cols) ? img[(int64_t)r * cols + c] : make_double2(0.0, 0.0);
    }
    __syncthreads();
    const double h = (double)rows;
    for (int lc = wv; lc < kFT_W; lc += kGB / 64) {
        const int i = c0 + lc, j = r0 + lane;  // column i, row j
        if (i >= cols) break;
        const double2 f = s_tile[lane][lc];
        const double norm = f.x * f.x + f.y * f.y;
        const bool keep = j < rows && norm > thr;
        const unsigned long long bal = __ballot(keep);
        const int64_t cell = (int64_t)i * nchunks + blockIdx.y;
        if (!SCATTER) {
            if (lane == 0) cell_counts[cell] = __popcll(bal);
        } else if (keep) {
            int64_t seg_base;
            if (SCATTER == 2) {
                seg_base = 0;
                const int sg = (int)(cell / kScanSegCells);  // wave-uniform
                for (int sgi = 0; sgi < sg; ++sgi) seg_base += seg_bases[sgi];
            } else {
                seg_base = seg_bases[cell / kScanSegCells];
            }
            const int64_t o = seg_base + cell_offsets[cell] + __popcll(bal & ((1ull << lane) - 1ull));
            const FlatPoint fp = flatten_point(f, i + col0, j, fx, fy, cx, cy, gamma, h);
            q[o] = make_double2(fp.qx, fp.qy);
            u[o] = make_double2(fp.ux, fp.uy);
            alpha[o] = fp.alpha;
            alpha_k[o] = fp.alpha_k;
        }
    }
}

// ---- depth map ----
// fixed-order sum of z = inliers(2, i): per-block partials then one workgroup
// m_dev (optional, here and in the kernels below): the inlier count is read from device memory (RefineState::m) -- the frame solve
// enqueues this stage before the host knows it; `m` is then only the upper bound the grid was sized for
__global__ __launch_bounds__(kGB) void zsum_partial_kernel(const double* __restrict__ inl, int64_t m, double* __restrict__ partials,
                                                          const int64_t* __restrict__ m_dev) {
    __shared__ double s_red[kGB / 64];
    if (m_dev) m = *m_dev;
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * kGB;
    for (int64_t i = (int64_t)blockIdx.x * kGB + threadIdx.x; i < m; i += stride) acc += inl[3 * i + 2];
    const double r = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = s_red[0];
        for (int w2 = 1; w2 < kGB / 64; ++w2) t += s_red[w2];
        partials[blockIdx.x] = t;
    }
}

// header: [0] = flipped flag (as double), [1..3] = v after the flip
// v_dev (optional): v is read from device memory (RefineState::p) instead of `pose_v`
__global__ __launch_bounds__(256) void zsum_decide_kernel(const double* __restrict__ partials, int nblocks, int64_t m, Pose pose_v,
                                                         double* __restrict__ header, double* __restrict__ header_host,
                                                         const double* __restrict__ v_dev, const int64_t* __restrict__ m_dev,
                                                         PoseTableOut pt) {
    if (v_dev) pose_v.v[0] = v_dev[0], pose_v.v[1] = v_dev[1], pose_v.v[2] = v_dev[2];
    if (m_dev) m = *m_dev;
    __shared__ double s_red[4];
    __shared__ double s_v[3];
    double acc = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) acc += partials[b];
    const double r = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double count_z = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
        const double z_mean = count_z * 1.0 / (double)m;  // main.cc:472 (NaN for m == 0: no flip)
        const bool flip = z_mean < 0;
        header[0] = flip ? 1.0 : 0.0;
        header[1] = flip ? pose_v.v[0] * -1.0 : pose_v.v[0];
        header[2] = flip ? pose_v.v[1] * -1.0 : pose_v.v[1];
        header[3] = flip ? pose_v.v[2] * -1.0 : pose_v.v[2];
        if (header_host) {  // host-mapped pinned copy: no copy kernel behind the stage, the host only synchronises
#pragma unroll
            for (int i = 0; i < 4; ++i) header_host[i] = header[i];
        }
        s_v[0] = header[1], s_v[1] = header[2], s_v[2] = header[3];
    }
    if (pt.R) {  // frame solve: RsFrame::setRelativePose's table for (v', w, k) right here instead of in a launch of its own
        __syncthreads();
        Pose pose;
        pose.v[0] = s_v[0], pose.v[1] = s_v[1], pose.v[2] = s_v[2];
        pose.w[0] = pt.wk_dev[0], pose.w[1] = pt.wk_dev[1], pose.w[2] = pt.wk_dev[2], pose.k = pt.wk_dev[3];
        for (int i = threadIdx.x; i < pt.rows; i += 256) pose_table_row(pose, pt.gamma, pt.rows, i, pt.R, pt.t);
    }
}

// flips z in place if requested, computes pixel indices, claims pixels for the HIGHEST inlier index (the
// reference's sequential loop lets the last writer win, main.cc:499-508)
// owner: persistent claim words `tag | inlier index` (claim_map_acquire: words of earlier calls carry an older epoch and lose every
// atomicMax, so the map needs no clearing pass per call -- round 1 filled 8 B/pixel with -1 before every depth map)
__global__ __launch_bounds__(kGB) void depth_claim_kernel(double* __restrict__ inl, int64_t m, const double* __restrict__ header,
                                                         double fx, double fy, double cx, double cy, int rows, int col0,
                                                         int ncols, unsigned* __restrict__ owner, unsigned tag,
                                                         int32_t* __restrict__ xs, int32_t* __restrict__ ys,
                                                         const int64_t* __restrict__ m_dev) {
    if (m_dev) m = *m_dev;
    const bool flip = header[0] != 0.0;
    const int64_t stride = (int64_t)gridDim.x * kGB;
    for (int64_t i = (int64_t)blockIdx.x * kGB + threadIdx.x; i < m; i += stride) {
        if (flip) inl[3 * i + 2] = inl[3 * i + 2] * -1.0;
        const int x = (int)(fx * inl[3 * i] + cx + 0.5);
        const int y = (int)(fy * inl[3 * i + 1] + cy + 0.5);
        if (xs) xs[i] = x;
        if (ys) ys[i] = y;
        if (x >= col0 && x < col0 + ncols && y >= 0 && y < rows) atomicMax(&owner[(int64_t)(x - col0) * rows + y], tag | (unsigned)i);
    }
}

__global__ __launch_bounds__(kGB) void depth_write_kernel(const double* __restrict__ inl, const unsigned* __restrict__ owner, unsigned tag,
                                                         unsigned mask, int64_t npix, double* __restrict__ depth_map) {
    const int64_t stride = (int64_t)gridDim.x * kGB;
    for (int64_t p = (int64_t)blockIdx.x * kGB + threadIdx.x; p < npix; p += stride) {
        const unsigned w = owner[p];
        depth_map[p] = ((w & ~mask) == tag) ? inl[3 * (int64_t)(w & mask) + 2] : 0.0;
    }
}

// exclusive scan of the cell counts in two levels (a single workgroup walking 15 000 ... 130 000 counts serialises on load
// latency: 31 us at 1280x720).  Level 1: every workgroup scans kScanSeg consecutive counts (coalesced staging through LDS)
// and writes local exclusive offsets + its segment total; level 2: one workgroup scans the segment totals in place and
// stores the grand total.  The scatter kernel adds the segment base to the local offset.
constexpr int kScanSeg = kScanSegCells;

__global__ __launch_bounds__(256) void cell_scan_local_kernel(const int64_t* __restrict__ counts, int64_t ncells,
                                                             int64_t* __restrict__ offsets, int64_t* __restrict__ seg_totals) {
    __shared__ int s_cnt[kScanSeg];
    __shared__ int s_wave[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * kScanSeg;
#pragma unroll
    for (int k = 0; k < kScanSeg / 256; ++k) {
        const int64_t i = base + k * 256 + tid;
        s_cnt[k * 256 + tid] = i < ncells ? (int)counts[i] : 0;
    }
    __syncthreads();
    int v[kScanSeg / 256];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < kScanSeg / 256; ++k) {
        v[k] = s_cnt[tid * (kScanSeg / 256) + k];
        sum += v[k];
    }
    int incl = sum;  // inclusive scan of the thread sums inside the wave
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w2 = 0; w2 < wv; ++w2) wbase += s_wave[w2];
    int run = wbase + incl - sum;
#pragma unroll
    for (int k = 0; k < kScanSeg / 256; ++k) {
        const int64_t i = base + (int64_t)tid * (kScanSeg / 256) + k;
        if (i < ncells) offsets[i] = run;
        run += v[k];
    }
    if (tid == 255) seg_totals[blockIdx.x] = run;
}

// in-place exclusive scan of the segment totals (nseg <= a few hundred) + grand total
__global__ __launch_bounds__(256) void cell_scan_segments_kernel(int64_t* __restrict__ seg_totals, int nseg, int64_t* __restrict__ total) {
    __shared__ int64_t s_part[256];
    const int tid = threadIdx.x;
    const int per = (nseg + 255) / 256;
    int64_t sum = 0;
    for (int j = 0; j < per; ++j) {
        const int b = tid * per + j;
        if (b < nseg) sum += seg_totals[b];
    }
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int64_t run = 0;
        for (int i = 0; i < 256; ++i) {
            const int64_t v = s_part[i];
            s_part[i] = run;
            run += v;
        }
        *total = run;
    }
    __syncthreads();
    int64_t run = s_part[tid];
    for (int j = 0; j < per; ++j) {
        const int b = tid * per + j;
        if (b < nseg) {
            const int64_t v = seg_totals[b];
            seg_totals[b] = run;
            run += v;
        }
    }
}

// cells of the tiled flatten (one per column and 64-row chunk): size of the two int64 work arrays
// (d_offsets additionally holds the segment totals / bases behind the per-cell offsets)
int64_t flatten_cells(int rows, int cols) {
    const int64_t ncells = (int64_t)cols * ((rows + kFT_H - 1) / kFT_H);
    return ncells + (ncells + kScanSeg - 1) / kScanSeg + 1;
}

// d_total: int64 receiving the number of kept points (device memory, or host-mapped pinned memory when the host waits on
// `total_ready` instead of the whole stream); d_counts / d_offsets: flatten_cells(rows, cols) int64 each
int flatten_launch(Ctx* c, const double* d_img, int rows, int cols, int col0, double fx, double fy, double cx, double cy,
                   double gamma, double thr, double* d_q, double* d_u, double* d_alpha, double* d_alpha_k, int64_t* d_counts,
                   int64_t* d_offsets, int64_t* d_total, hipEvent_t total_ready) {
    const int nchunks = (rows + kFT_H - 1) / kFT_H;
    const int64_t ncells = (int64_t)cols * nchunks;
    if (ncells > (int64_t)INT32_MAX) return fail(c, RSDSFM_ERR_INVALID, "image too large for the flatten scan");
    const dim3 grid((cols + kFT_W - 1) / kFT_W, nchunks);
    const double2* img2 = reinterpret_cast<const double2*>(d_img);
    const int nseg = (int)((ncells + kScanSeg - 1) / kScanSeg);
    int64_t* d_seg = d_offsets + ncells;  // segment totals -> bases
    hipLaunchKernelGGL(flatten_tile_kernel<0>, grid, dim3(kGB), 0, c->stream, img2, rows, cols, fx, fy, cx, cy, gamma, thr, col0, nchunks,
                       d_counts, d_offsets, d_seg, reinterpret_cast<double2*>(d_q), reinterpret_cast<double2*>(d_u), d_alpha, d_alpha_k, 0,
                       (int64_t*)nullptr);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(cell_scan_local_kernel, dim3(nseg), dim3(256), 0, c->stream, d_counts, ncells, d_offsets, d_seg);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    if (!total_ready && nseg <= 256) {
        // nobody waits for the count in front of the scatter pass (the frame solve): that pass adds up the few segment totals itself
        hipLaunchKernelGGL(flatten_tile_kernel<2>, grid, dim3(kGB), 0, c->stream, img2, rows, cols, fx, fy, cx, cy, gamma, thr, col0, nchunks,
                           d_counts, d_offsets, d_seg, reinterpret_cast<double2*>(d_q), reinterpret_cast<double2*>(d_u), d_alpha, d_alpha_k, nseg,
                           d_total);
        RSDSFM_HIP_CHECK(c, hipGetLastError());
        return RSDSFM_OK;
    }
    hipLaunchKernelGGL(cell_scan_segments_kernel, dim3(1), dim3(256), 0, c->stream, d_seg, nseg, d_total);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    if (total_ready) RSDSFM_HIP_CHECK(c, hipEventRecord(total_ready, c->stream));  // the count is final here; the scatter pass follows
    hipLaunchKernelGGL(flatten_tile_kernel<1>, grid, dim3(kGB), 0, c->stream, img2, rows, cols, fx, fy, cx, cy, gamma, thr, col0, nchunks,
                       d_counts, d_offsets, d_seg, reinterpret_cast<double2*>(d_q), reinterpret_cast<double2*>(d_u), d_alpha, d_alpha_k, 0,
                       (int64_t*)nullptr);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

// the shard's z sum as one double (the all-gather payload of the row-tiled solve); d_partials: >= 1024 doubles
__global__ __launch_bounds__(256) void zsum_row_kernel(const double* __restrict__ partials, int nblocks, double* __restrict__ out) {
    __shared__ double s_red[4];
    double acc = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) acc += partials[b];
    const double r = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
}

static inline int zsum_blocks(int64_t m) { return (int)std::min<int64_t>(1024, std::max<int64_t>(1, (m + kGB - 1) / kGB)); }

int zsum_row_launch(Ctx* c, const double* d_inl, int64_t m, double* d_partials, double* d_out) {
    const int zb = zsum_blocks(m);
    hipLaunchKernelGGL(zsum_partial_kernel, dim3(zb), dim3(kGB), 0, c->stream, d_inl, m, d_partials, (const int64_t*)nullptr);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    hipLaunchKernelGGL(zsum_row_kernel, dim3(1), dim3(256), 0, c->stream, d_partials, zb, d_out);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

// sign decision from nz partial z sums over m_total points, then claim + write of the column slab [col0, col0 + ncols)
// d_header: 4 doubles (flipped, v')
int depth_map_slab_launch(Ctx* c, double* d_inl, int64_t m, const double* d_zsums, int nz, int64_t m_total, const double v[3],
                          double fx, double fy, double cx, double cy, int rows, int col0, int ncols, double* d_depth_map,
                          int32_t* d_xs, int32_t* d_ys, double* d_header, double* h_header, const double* v_dev, const int64_t* m_dev,
                          const PoseTableOut* pt) {
    const int64_t npix = (int64_t)rows * ncols;
    if (m >= ((int64_t)1 << 31)) return fail(c, RSDSFM_ERR_INVALID, "depth map: more than 2^31 inliers");
    Pose pv;
    memset(&pv, 0, sizeof(pv));
    if (v) pv.v[0] = v[0], pv.v[1] = v[1], pv.v[2] = v[2];
    hipLaunchKernelGGL(zsum_decide_kernel, dim3(1), dim3(256), 0, c->stream, d_zsums, nz, m_total, pv, d_header, h_header, v_dev, m_dev,
                       pt ? *pt : PoseTableOut());
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    // the claim word holds the inlier index: the map's index field is sized by the larger of the two counts
    unsigned *d_owner = nullptr, tag = 0, mask = 0;
    int rc = claim_map_acquire(c, 2, (size_t)std::max<int64_t>(std::max<int64_t>(npix, m), 1), &d_owner, &tag, &mask);
    if (rc != RSDSFM_OK) return rc;
    if (m > 0) {
        hipLaunchKernelGGL(depth_claim_kernel, dim3(stream_grid(m)), dim3(kGB), 0, c->stream, d_inl, m, d_header, fx, fy, cx, cy, rows,
                           col0, ncols, d_owner, tag, d_xs, d_ys, m_dev);
        RSDSFM_HIP_CHECK(c, hipGetLastError());
    }
    hipLaunchKernelGGL(depth_write_kernel, dim3(stream_grid(npix)), dim3(kGB), 0, c->stream, d_inl, d_owner, tag, mask, npix, d_depth_map);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return RSDSFM_OK;
}

// d_header: 4 doubles (flipped, v'); d_partials: >= 1024 doubles
int depth_map_launch(Ctx* c, double* d_inl, int64_t m, const double v[3], double fx, double fy, double cx, double cy, int rows,
                     int cols, double* d_depth_map, int32_t* d_xs, int32_t* d_ys, double* d_header, double* d_partials,
                     double* h_header, const double* v_dev, const int64_t* m_dev, const PoseTableOut* pt) {
    const int zb = zsum_blocks(m);
    hipLaunchKernelGGL(zsum_partial_kernel, dim3(zb), dim3(kGB), 0, c->stream, d_inl, m, d_partials, m_dev);
    RSDSFM_HIP_CHECK(c, hipGetLastError());
    return depth_map_slab_launch(c, d_inl, m, d_partials, zb, m, v, fx, fy, cx, cy, rows, 0, cols, d_depth_map, d_xs, d_ys, d_header,
                                 h_header, v_dev, m_dev, pt);
}

}  // namespace rsdsfm
