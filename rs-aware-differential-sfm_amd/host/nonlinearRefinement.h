// nonlinearRefinement.h -- drop-in for the reference's src/nonlinearRefinement.h (namespace nonlinear_refinement):
// same names and argument order, implemented over the MI355X C ABI.  The Ceres functor RsResidual
// (reference nonlinearRefinement.h:42-61) is kept as a plain double evaluator for callers that used it directly.
#ifndef RSDSFM_HOST_NONLINEARREFINEMENT_H
#define RSDSFM_HOST_NONLINEARREFINEMENT_H

#include "minimal.h"

namespace nonlinear_refinement {

using rsdsfm::lite::Array2Xd;
using rsdsfm::lite::Array3Xd;
using rsdsfm::lite::ArrayXd;
using rsdsfm::lite::Vector2d;
using rsdsfm::lite::Vector3d;

/** residual of one correspondence (reference nonlinearRefinement.cc:32-52, T = double) */
struct RsResidual {
    RsResidual(double coordinates[], double observed_velocity[], double alpha, double alphaK) : alpha_(alpha), alphaK_(alphaK) {
        for (int i = 0; i < 2; ++i) {
            coordinates_[i] = coordinates[i];
            observed_velocity_[i] = observed_velocity[i];
        }
    }
    bool operator()(const double* const lin_velocity, const double* const angl_velocity, const double* const k,
                    const double* const inv_depth, double* residuals) const {
        const double x = coordinates_[0], y = coordinates_[1];
        const double beta = (2.0 / (2.0 + (*k))) * (alpha_ + (*k) * alphaK_);
        const double p0 = beta * -1.0 * (*inv_depth * (x * lin_velocity[2] - lin_velocity[0]) + (x * y * angl_velocity[0]) -
                                         (1.0 + x * x) * angl_velocity[1] + y * angl_velocity[2]);
        const double p1 = beta * -1.0 * (*inv_depth * (y * lin_velocity[2] - lin_velocity[1]) + (1.0 + y * y) * angl_velocity[0] -
                                         x * y * angl_velocity[1] - x * angl_velocity[2]);
        residuals[0] = observed_velocity_[0] - p0;
        residuals[1] = observed_velocity_[1] - p1;
        return true;
    }

private:
    double coordinates_[2];
    double observed_velocity_[2];
    double alpha_;
    double alphaK_;
};

/** reference nonlinearRefinement.cc:109-180 */
inline ArrayXd estimateInverseDepths(const Array2Xd& normalized_coordinates, const Array2Xd& flow, const Vector3d& linear_velocity,
                                     const Vector3d& angular_velocity, const double& k, const ArrayXd& alpha, const ArrayXd& alphaK,
                                     bool show_messages) {
    const long n = normalized_coordinates.cols();
    ArrayXd rho(n);
    rsdsfm_lm_summary summary = {};
    rsdsfm::StopWatch watch;
    rsdsfm::check(rsdsfm_estimate_inverse_depths(rsdsfm::default_context(), normalized_coordinates.data(), flow.data(), n,
                                                 linear_velocity.data(), angular_velocity.data(), k, alpha.data(), alphaK.data(),
                                                 rsdsfm::depth_mode(), rho.data(), &summary),
                  "rsdsfm_estimate_inverse_depths");
    if (show_messages) {  // nonlinearRefinement.cc:165-169
        std::cout << std::endl;
        std::cout << rsdsfm::brief_report(summary) << std::endl;
        std::cout << "Total time for solving optimization: " << watch.seconds() << " s" << std::endl;
    }
    return rho;
}

/** reference nonlinearRefinement.cc:55-106, declared nonlinearRefinement.h:77-79 (single pixel; same signature) */
inline double estimateInverseDepth(const Vector2d& normalized_coordinates, const Vector3d& linear_velocity,
                                   const Vector3d& angular_velocity, const Vector2d& flow, const double& k, const double& alpha,
                                   const double& alphaK, bool show_messages) {
    double rho = 1.0;
    rsdsfm_lm_summary summary = {};
    rsdsfm::check(rsdsfm_estimate_inverse_depths(rsdsfm::default_context(), normalized_coordinates.data(), flow.data(), 1, linear_velocity.data(),
                                                 angular_velocity.data(), k, &alpha, &alphaK, rsdsfm::depth_mode(), &rho, &summary),
                  "rsdsfm_estimate_inverse_depths");
    if (show_messages) {  // nonlinearRefinement.cc:100-103
        std::cout << rsdsfm::brief_report(summary) << std::endl;
        std::cout << rho << std::endl;
    }
    return rho;
}

/** reference nonlinearRefinement.cc:183-252.  flow is indexed by inlier RANK exactly like the reference (quirk Q2)
 *  unless rsdsfm::flow_index_mode() is set to RSDSFM_FLOW_GATHERED, which uses inliers.inlier_idx. */
inline int& flow_index_mode() {
    static int mode = RSDSFM_FLOW_COMPAT_RANK;
    return mode;
}
inline RansacValues nonLinearRefinement(const Array2Xd& flow, const RansacValues& inliers, bool const_acceleration, bool show_messages) {
    const long m = inliers.num_inliers;
    Array3Xd out(3, m);
    Vector3d v, w;
    double k = 0;
    rsdsfm_lm_summary summary = {};
    rsdsfm::StopWatch watch;
    // (rsdsfm_tag: the RansacValues minimal::ransac has just returned are still on the device -- main.cc:447-457 passes them on unmodified)
    rsdsfm::check(rsdsfm_refine_from_ransac(rsdsfm::default_context(), inliers.rsdsfm_tag, flow.data(), flow.cols(), m, inliers.inliers.data(),
                                            inliers.alpha.data(), inliers.alpha_k.data(), inliers.inlier_idx.empty() ? nullptr : inliers.inlier_idx.data(),
                                            inliers.v.data(), inliers.w.data(), inliers.k, const_acceleration ? 1 : 0,
                                            inliers.inlier_idx.empty() ? RSDSFM_FLOW_COMPAT_RANK : flow_index_mode(), out.data(), v.data(), w.data(),
                                            &k, &summary),
                  "rsdsfm_refine_from_ransac");
    if (show_messages) {  // nonlinearRefinement.cc:230-234
        std::cout << rsdsfm::brief_report(summary) << std::endl;
        std::cout << "Total time for solving optimization: " << watch.seconds() << " s" << std::endl;
        std::cout << std::endl;
    }
    RansacValues r(inliers.num_inliers, out, inliers.alpha, inliers.alpha_k, w, v, k);
    r.inlier_idx = inliers.inlier_idx;
    return r;
}

}  // namespace nonlinear_refinement

#endif
