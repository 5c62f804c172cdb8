// sequence_run.cpp -- a C++ host of the SEQUENCE solve (BASELINE configs[4]: "batched frame pairs, sequence throughput mode") written
// against the C ABI only (include/rsdsfm.h + the HIP runtime for device memory).  The reference's evaluateSingleRun (main.cc:302-559)
// solves one pair per program run; a host with a whole sequence resident hands the pairs to ONE call and the library pipelines them
// (rsdsfm_solve_frames_dev).  Reads B raw flow images (each rows x cols x 2 doubles, row-major) from one file, solves them as a
// sequence and -- for comparison -- pair by pair with rsdsfm_solve_frame_dev on a second context; prints one JSON object per pair.
//
//   sequence_run flows.bin B rows cols fx fy cx cy gamma trials tol lanes
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/rsdsfm.h"

#define CHECK_HIP(x)                                                     \
    do {                                                                 \
        hipError_t e_ = (x);                                             \
        if (e_ != hipSuccess) {                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            return 4;                                                    \
        }                                                                \
    } while (0)
#define CHECK_RS(ctx, x)                                                                     \
    do {                                                                                     \
        int rc_ = (x);                                                                       \
        if (rc_ != RSDSFM_OK) {                                                              \
            std::fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, rsdsfm_last_error(ctx));   \
            return 5;                                                                        \
        }                                                                                    \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 13) {
        std::fprintf(stderr, "usage: sequence_run flows.bin B rows cols fx fy cx cy gamma trials tol lanes\n");
        return 2;
    }
    const int B = std::atoi(argv[2]), rows = std::atoi(argv[3]), cols = std::atoi(argv[4]);
    const double fx = std::atof(argv[5]), fy = std::atof(argv[6]), cx = std::atof(argv[7]), cy = std::atof(argv[8]), gamma = std::atof(argv[9]);
    const size_t per = (size_t)rows * cols * 2;
    std::vector<double> img(per * (size_t)B);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(img.data(), sizeof(double), img.size(), f) != img.size()) return 3;
    std::fclose(f);

    rsdsfm_ctx *seq = nullptr, *one = nullptr;
    if (rsdsfm_create(&seq, 0, nullptr) != RSDSFM_OK || rsdsfm_create(&one, 0, nullptr) != RSDSFM_OK) return 6;
    CHECK_RS(seq, rsdsfm_set_sequence_lanes(seq, std::atoi(argv[12])));

    rsdsfm_frame_params prm;
    rsdsfm_frame_params_init(&prm);  // main.cc:304-311's constants, rank-indexed flow (main.cc:457), struct_bytes stamped
    prm.ransac_trials = std::atoi(argv[10]);
    prm.ransac_tol = std::atof(argv[11]);

    double* d_img = nullptr;
    CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&d_img), sizeof(double) * img.size()));
    CHECK_HIP(hipMemcpy(d_img, img.data(), sizeof(double) * img.size(), hipMemcpyHostToDevice));
    std::vector<rsdsfm_frame_job> jobs((size_t)B);
    std::vector<double*> maps((size_t)B), maps1((size_t)B);
    for (int i = 0; i < B; ++i) {
        CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&maps[i]), sizeof(double) * (size_t)rows * cols));
        CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&maps1[i]), sizeof(double) * (size_t)rows * cols));
        rsdsfm_frame_job& j = jobs[(size_t)i];
        std::memset(&j, 0, sizeof(j));
        j.d_flow_img = d_img + per * (size_t)i;
        j.rows = rows, j.cols = cols;
        j.fx = fx, j.fy = fy, j.cx = cx, j.cy = cy, j.gamma = gamma;
        j.d_depth_map_colmajor = maps[(size_t)i];
        j.seed = 100 + 7 * (uint64_t)i;
    }
    std::vector<rsdsfm_frame_result> res((size_t)B);
    CHECK_RS(seq, rsdsfm_solve_frames_dev(seq, jobs.data(), B, &prm, res.data()));

    std::vector<double> a((size_t)rows * cols), b((size_t)rows * cols);
    for (int i = 0; i < B; ++i) {
        rsdsfm_frame_result r1;
        prm.seed = jobs[(size_t)i].seed;
        CHECK_RS(one, rsdsfm_solve_frame_dev(one, jobs[(size_t)i].d_flow_img, rows, cols, fx, fy, cx, cy, gamma, &prm, maps1[(size_t)i], nullptr, nullptr, &r1));
        CHECK_HIP(hipMemcpy(a.data(), maps[(size_t)i], sizeof(double) * a.size(), hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(b.data(), maps1[(size_t)i], sizeof(double) * b.size(), hipMemcpyDeviceToHost));
        const rsdsfm_frame_result& r = res[(size_t)i];
        const bool same = r.n_points == r1.n_points && r.num_inliers == r1.num_inliers && r.best_trial == r1.best_trial && r.flipped == r1.flipped &&
                          std::memcmp(r.v, r1.v, sizeof(r.v)) == 0 && std::memcmp(r.w, r1.w, sizeof(r.w)) == 0 && r.k == r1.k &&
                          r.refine_summary.num_iterations == r1.refine_summary.num_iterations &&
                          std::memcmp(a.data(), b.data(), sizeof(double) * a.size()) == 0;
        std::printf("{\"pair\": %d, \"n\": %lld, \"num_inliers\": %lld, \"best_trial\": %d, \"v\": [%.17g, %.17g, %.17g], \"w\": [%.17g, %.17g, %.17g], "
                    "\"iterations\": %d, \"equals_single_solve\": %s}\n",
                    i, (long long)r.n_points, (long long)r.num_inliers, r.best_trial, r.v[0], r.v[1], r.v[2], r.w[0], r.w[1], r.w[2],
                    r.refine_summary.num_iterations, same ? "true" : "false");
    }
    for (int i = 0; i < B; ++i) (void)hipFree(maps[(size_t)i]), (void)hipFree(maps1[(size_t)i]);
    (void)hipFree(d_img);
    rsdsfm_destroy(seq);
    rsdsfm_destroy(one);
    return 0;
}
