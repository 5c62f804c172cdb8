// mirror_check.cpp -- does the drop-in C++ mirror (rs-aware-differential-sfm_amd/host/*.h) meet the reference's call sites with REAL Eigen?
// INTEGRATION.md promises: with Eigen on the include path `rsdsfm::lite` is an alias of `Eigen`, so the mirror's signatures are the
// reference's, type for type.  This translation unit is compiled (and linked against librsdsfm_hip.so) by the `mirror_check` target of this
// directory's CMakeLists.txt on a machine that HAS Eigen -- the development image has not -- and is never run: it needs no GPU to prove
// that the six call sites of the reference compile unchanged.  The calls below are the reference's own argument expressions:
//   main.cc:437-438, :447, :457 (evaluateSingleRun)  and  errorMeasure.cpp:104-105, :140, :152 (evaluateVelocities).
#include <iostream>
#include <type_traits>
#include <utility>

#include <Eigen/Dense>

#include "../../rs-aware-differential-sfm_amd/host/minimal.h"
#include "../../rs-aware-differential-sfm_amd/host/nonlinearRefinement.h"

static_assert(std::is_same<rsdsfm::lite::ArrayXd, Eigen::ArrayXd>::value, "with Eigen present rsdsfm::lite IS Eigen");
static_assert(std::is_same<rsdsfm::lite::Matrix2Xd, Eigen::Matrix2Xd>::value, "with Eigen present rsdsfm::lite IS Eigen");
static_assert(std::is_same<std::decay<decltype(std::declval<RansacValues&>().w)>::type, Eigen::Vector3d>::value, "RansacValues::w is the reference's Vector3d");
static_assert(std::is_same<std::decay<decltype(std::declval<RansacValues&>().inliers)>::type, Eigen::Array3Xd>::value, "RansacValues::inliers is the reference's Array3Xd");

using namespace std;
using namespace Eigen;
using nonlinear_refinement::nonLinearRefinement;  // (main.cc calls it unqualified)

// main.cc:398-462, the solver part of evaluateSingleRun, with the reference's variable names and expressions
static RansacValues like_main_cc(int rows, int cols, double gamma, bool use_global_shutter_mode, bool use_acceleration_mode, bool use_refinement,
                                 int ransac_trials, double ransac_tol) {
    Eigen::Matrix2Xd coord = Eigen::Matrix2Xd::Ones(2, rows * cols);
    Eigen::Matrix2Xd coord_pixel = Eigen::Matrix2Xd::Ones(2, rows * cols);
    Eigen::Matrix2Xd flow = Eigen::Matrix2Xd::Zero(2, rows * cols);
    Eigen::Matrix2Xd flow_pixel = Eigen::Matrix2Xd::Zero(2, rows * cols);
    // calculate beta values
    ArrayXd alpha = minimal::getAlpha(flow_pixel, rows, gamma);
    ArrayXd alphaK = minimal::getAlphaK(coord_pixel, flow_pixel, rows, gamma);
    // set beta = 1 if GS assumption is used
    if (use_global_shutter_mode) {
        alpha *= 0;
        alpha += 1;
    }
    // run ransac
    RansacValues ransac_results = minimal::ransac(coord, flow, alpha, alphaK, use_acceleration_mode, ransac_trials, ransac_tol, true);
    cout << endl << "ransac numInliers: " << ransac_results.num_inliers << endl;
    cout << "ransac w: " << ransac_results.w.transpose() << endl;
    cout << "ransac v: " << ransac_results.v.transpose() << endl;
    cout << "ransac k: " << ransac_results.k << endl << endl;
    // optimize solution with nonlinear refinement:
    RansacValues results = ransac_results;
    if (use_refinement) {
        results = nonLinearRefinement(flow, ransac_results, use_acceleration_mode, false);
    }
    cout << "ransac w: " << results.w.transpose() << endl;
    return results;
}

// errorMeasure.cpp:66-160, the solver part of evaluateVelocities
static RansacValues like_error_measure_cpp(int rows, int cols, double gamma, bool global_shutter, bool constant_acceleration, bool optimize_results,
                                           int ransac_trials, bool show_messages) {
    const double TOL_RANSAC = 0.05;
    Eigen::Matrix2Xd coord = Eigen::Matrix2Xd::Ones(2, rows * cols);
    Eigen::Matrix2Xd coord_full = Eigen::Matrix2Xd::Ones(2, rows * cols);
    Eigen::Matrix2Xd flow = Eigen::Matrix2Xd::Zero(2, rows * cols);
    Eigen::Matrix2Xd flow_full = Eigen::Matrix2Xd::Zero(2, rows * cols);
    int position = rows * cols / 2;
    // remove 0 columns from the coord and flow matrices
    coord.conservativeResize(2, position);
    flow.conservativeResize(2, position);
    // calculate both parts of the beta factor
    Eigen::ArrayXd alpha = minimal::getAlpha(flow_full, rows, gamma);
    Eigen::ArrayXd alphaK = minimal::getAlphaK(coord_full, flow_full, rows, gamma);
    if (global_shutter) {
        alpha *= 0;
        alpha += 1;
    }
    // execute ransac
    RansacValues ransac_results = minimal::ransac(coord, flow, alpha, alphaK, constant_acceleration, ransac_trials, TOL_RANSAC, show_messages);
    if (show_messages) {
        std::cout << "ransac w: " << ransac_results.w.transpose() << std::endl;
        std::cout << "ransac k: " << ransac_results.k << std::endl << std::endl;
    }
    RansacValues results = ransac_results;  // in case no optimization is used
    if (optimize_results) {
        results = nonlinear_refinement::nonLinearRefinement(flow, ransac_results, constant_acceleration, show_messages);
        if (show_messages) std::cout << "final w: " << results.w.transpose() << std::endl;
    }
    return results;
}

int main(int argc, char**) {
    if (argc < 1000) {  // compile + link only: the calls above need an MI355X to run
        std::cout << "mirror_check: the reference's six call sites compile and link against the drop-in mirror with Eigen " << EIGEN_WORLD_VERSION << "."
                  << EIGEN_MAJOR_VERSION << "." << EIGEN_MINOR_VERSION << std::endl;
        return 0;
    }
    return (int)(like_main_cc(4, 4, 0.8, false, false, true, 5, 0.05).num_inliers + like_error_measure_cpp(4, 4, 0.8, false, false, true, 5, false).num_inliers);
}
