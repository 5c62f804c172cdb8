// tiled_run.cpp -- a C++ host of the column-tiled whole solve (BASELINE configs[3]) written against the C ABI
// only (include/rsdsfm.h + the HIP runtime for device memory): what a maintainer's multi-GPU driver looks like.  One process per
// rank; rank / world size / the 128-byte RCCL id come from the command line here (an MPI host would use MPI_Comm_rank and
// MPI_Bcast).  With world size 1 it runs on one GPU over a 1-rank RCCL communicator, which is how tests/test_cpp_mirror.py drives
// it.  Reads a raw flow image (rows x cols x 2 doubles, row-major), uploads THIS rank's column slab, solves, prints JSON.
//
//   tiled_run flow.bin rows cols fx fy cx cy gamma trials tol seed [rank world id_hex]
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rsdsfm.h"

#define CHECK_HIP(x)                                                            \
    do {                                                                        \
        hipError_t e_ = (x);                                                    \
        if (e_ != hipSuccess) {                                                 \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));        \
            return 4;                                                           \
        }                                                                       \
    } while (0)
#define CHECK_RS(ctx, x)                                                        \
    do {                                                                        \
        int rc_ = (x);                                                          \
        if (rc_ != RSDSFM_OK) {                                                 \
            std::fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, rsdsfm_last_error(ctx)); \
            return 5;                                                           \
        }                                                                       \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 12) {
        std::fprintf(stderr, "usage: tiled_run flow.bin rows cols fx fy cx cy gamma trials tol seed [rank world id_hex]\n");
        return 2;
    }
    const int rows = std::atoi(argv[2]), cols = std::atoi(argv[3]);
    const double fx = std::atof(argv[4]), fy = std::atof(argv[5]), cx = std::atof(argv[6]), cy = std::atof(argv[7]), gamma = std::atof(argv[8]);
    const int rank = argc > 12 ? std::atoi(argv[12]) : 0, world = argc > 13 ? std::atoi(argv[13]) : 1;

    std::vector<double> img((size_t)rows * cols * 2);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(img.data(), sizeof(double), img.size(), f) != img.size()) return 3;
    std::fclose(f);

    int ndev = 0;
    CHECK_HIP(hipGetDeviceCount(&ndev));
    rsdsfm_ctx* ctx = nullptr;
    if (rsdsfm_create(&ctx, rank % ndev, nullptr) != RSDSFM_OK) return 6;  // one GPU per rank (round-robin if fewer GPUs than ranks)

    // communicator: rank 0 creates the id; the other ranks receive it (here: as a hex string on the command line)
    unsigned char id[RSDSFM_DIST_ID_BYTES];
    if (argc > 14) {
        const std::string hex = argv[14];
        if (hex.size() != 2 * RSDSFM_DIST_ID_BYTES) return 7;
        for (int i = 0; i < RSDSFM_DIST_ID_BYTES; ++i) id[i] = (unsigned char)std::strtoul(hex.substr(2 * i, 2).c_str(), nullptr, 16);
    } else {
        CHECK_RS(ctx, rsdsfm_dist_unique_id(id));
    }
    CHECK_RS(ctx, rsdsfm_dist_init(ctx, world, rank, id));

    // this rank's column slab, packed row-major [rows][slab_cols][2], and the device buffers
    int32_t col0 = 0, slab_cols = 0, stride = 0;
    CHECK_RS(ctx, rsdsfm_tiled_slab_bounds(cols, world, rank, &col0, &slab_cols, &stride));
    std::vector<double> slab((size_t)rows * slab_cols * 2);
    for (int y = 0; y < rows; ++y)
        std::memcpy(&slab[(size_t)y * slab_cols * 2], &img[((size_t)y * cols + col0) * 2], sizeof(double) * 2 * (size_t)slab_cols);
    double *d_slab = nullptr, *d_map = nullptr, *d_R = nullptr, *d_t = nullptr;
    CHECK_HIP(hipSetDevice(rank % ndev));
    CHECK_HIP(hipMalloc(&d_slab, sizeof(double) * std::max<size_t>(slab.size(), 2)));
    CHECK_HIP(hipMalloc(&d_map, sizeof(double) * (size_t)rows * cols));
    CHECK_HIP(hipMalloc(&d_R, sizeof(double) * 9 * (size_t)rows));
    CHECK_HIP(hipMalloc(&d_t, sizeof(double) * 3 * (size_t)rows));
    if (!slab.empty()) CHECK_HIP(hipMemcpy(d_slab, slab.data(), sizeof(double) * slab.size(), hipMemcpyHostToDevice));

    rsdsfm_frame_params prm;
    std::memset(&prm, 0, sizeof(prm));
    prm.ransac_trials = std::atoi(argv[9]);
    prm.ransac_tol = std::atof(argv[10]);
    prm.seed = std::strtoull(argv[11], nullptr, 10);
    prm.flow_threshold = 1e-10;                     // main.cc:311
    prm.use_refinement = 1;                         // main.cc:307
    prm.depth_mode = RSDSFM_DEPTH_CERES_LM;
    // flow_index_mode stays 0 = RSDSFM_FLOW_COMPAT_RANK: the zero-initialised struct reproduces evaluateSingleRun (main.cc:457)
    rsdsfm_frame_result res;
    rsdsfm_tiled_info info;
    CHECK_RS(ctx, rsdsfm_solve_frame_tiled_dev(ctx, d_slab, rows, cols, fx, fy, cx, cy, gamma, &prm, d_map, d_R, d_t, &res, &info));

    std::vector<double> map((size_t)rows * cols), t((size_t)rows * 3);
    CHECK_HIP(hipMemcpy(map.data(), d_map, sizeof(double) * map.size(), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(t.data(), d_t, sizeof(double) * t.size(), hipMemcpyDeviceToHost));
    double dsum = 0;
    long long nz = 0;
    for (double z : map) {
        dsum += z;
        nz += z != 0.0;
    }
    std::printf("{\"rank\": %d, \"world\": %d, \"n\": %lld, \"num_inliers\": %lld, \"best_trial\": %d, \"flipped\": %d, "
                "\"v\": [%.17g, %.17g, %.17g], \"w\": [%.17g, %.17g, %.17g], \"k\": %.17g, \"iterations\": %d, \"depth_nonzero\": %lld, "
                "\"depth_sum\": %.17g, \"last_t\": [%.17g, %.17g, %.17g], \"shard_points\": %lld, \"host_syncs\": %d, \"collectives\": %d}\n",
                rank, world, (long long)res.n_points, (long long)res.num_inliers, res.best_trial, res.flipped, res.v[0], res.v[1], res.v[2],
                res.w[0], res.w[1], res.w[2], res.k, res.refine_summary.num_iterations, nz, dsum, t[(size_t)(rows - 1) * 3],
                t[(size_t)(rows - 1) * 3 + 1], t[(size_t)(rows - 1) * 3 + 2], (long long)info.shard_points, info.host_syncs, info.collectives);
    (void)hipFree(d_slab), (void)hipFree(d_map), (void)hipFree(d_R), (void)hipFree(d_t);
    CHECK_RS(ctx, rsdsfm_dist_finalize(ctx));
    rsdsfm_destroy(ctx);
    return 0;
}
