// camera.h -- the facade of the reference's Camera (src/camera.h) over the frames: intrinsics (incl. the five
// hard-coded phone calibrations, camera.cc:179-206), gamma, pose and depth map forwarding (camera.cc:335-371).
// plus the rectifier entry points backProject / backProjectGs (camera.cc:353-361) and interpolateCrackyImage
// (camera.cc:753-774).  Flow, CSV and visualisation members are out of scope.
#ifndef RSDSFM_HOST_CAMERA_H
#define RSDSFM_HOST_CAMERA_H

#include <cmath>
#include <iostream>
#include <string>
#include <vector>

#include "rsframe.h"

class Camera {
public:
    Camera() {}
    rsdsfm::lite::Matrix3d getIntrinsics() { return K_; }
    void setIntrinsics(const rsdsfm::lite::Matrix3d& K) {
        K_ = K;
        for (auto& f : frames_) f.setIntrinsics(K_);
    }
    /** reference camera.cc:179-206 */
    void setIntrinsics(const std::string source_camera) {
        double fx, fy, cx, cy;
        if (source_camera == "iphone") fx = 1505.1283359786307, fy = 1513.7789208311444, cx = 657.81734686405991, cy = 349.91807538147589;
        else if (source_camera == "galaxy_stabil") fx = 1803.29785922382, fy = 1799.35406531529, cx = 945.304708272490, cy = 544.684292978344;
        else if (source_camera == "galaxy") fx = 1492.41306997746, fy = 1491.09286590722, cx = 949.571146410704, cy = 554.675409391795;
        else if (source_camera == "galaxy_old") fx = 3154.53208221173, fy = 3152.28696217577, cx = 1969.87107268891, cy = 1521.27056048818;
        else if (source_camera == "galaxy_vga") fx = 484.450845764569, fy = 485.345469134313, cx = 313.442094604855, cy = 241.383116350144;
        else {
            std::cerr << "No valid source camera specified";
            return;
        }
        rsdsfm::lite::Matrix3d K = rsdsfm::lite::Matrix3d::Zero();
        K(0, 0) = fx, K(1, 1) = fy, K(0, 2) = cx, K(1, 2) = cy, K(2, 2) = 1.0;
        setIntrinsics(K);
    }
    /** frames are 1-based like the reference; only the frame geometry is kept (no pixel data) */
    void addFrame(int rows, int cols) {
        frames_.emplace_back(rows, cols);
        frames_.back().setIntrinsics(K_);
    }
    RsFrame getFrame(const int frameNr) { return frames_[(size_t)frameNr - 1]; }
    RsFrame& frame(const int frameNr) { return frames_[(size_t)frameNr - 1]; }
    rsdsfm::lite::MatrixXd getDepthMap(const int frameNr) { return frames_[(size_t)frameNr - 1].getDepthMap(); }
    void setDepthMap(const int frameNr, rsdsfm::lite::MatrixXd depth_map) { frames_[(size_t)frameNr - 1].setDepthMap(depth_map); }
    /** reference camera.cc:369-371, :416-420 (the latter overwrites the stored depth map, like the reference) */
    void setSyntheticDepthMap(const int frameNr) { frames_[(size_t)frameNr - 1].setSyntheticDepthMapRs(); }
    rsdsfm::lite::MatrixXd getGroundTruthDepthMap(const int frameNr) {
        frames_[(size_t)frameNr - 1].setSyntheticDepthMapRs();
        return frames_[(size_t)frameNr - 1].getDepthMap();
    }
    /** reference camera.cc:374-408: the eyeball check of the projection chain on frame 1 -- for every pixel with ground truth, the
     *  world point from the unprojection maps, its camera-frame coordinates, the point re-derived from the depth map, both
     *  re-projections and the world point recovered from the depth map (vectors printed on one line each) */
    void testProjection() {
        RsFrame frame = frames_[0];
        frame.setSyntheticDepthMapRs();
        auto show = [](const char* what, const double* d, int n) {
            std::cout << what;
            for (int i = 0; i < n; ++i) std::cout << (i ? " " : "") << d[i];
            std::cout << std::endl;
        };
        for (int x = 0; x < frame.getCols(); ++x)
            for (int y = 0; y < frame.getRows(); ++y) {
                const rsdsfm::lite::Vector3d true_coordinates = frame.getUnprojectedWorldCoordinates(rsdsfm::lite::Vector2d(x, y));
                const rsdsfm::lite::Vector3d camera_coordinates1 = frame.worldToCameraFrame(true_coordinates, y);
                const rsdsfm::lite::Vector3d camera_coordinates2 = frame.planeToSpace(rsdsfm::lite::Vector2d(x, y));
                const rsdsfm::lite::Vector2d image_coordinates1 = frame.spaceToPlane(camera_coordinates1);
                const rsdsfm::lite::Vector2d image_coordinates2 = frame.spaceToPlane(camera_coordinates2);
                const rsdsfm::lite::Vector3d world_coordinates = frame.cameraToWorldFrame(camera_coordinates2, y);
                const double* t = true_coordinates.data();
                if (std::sqrt((t[0] * t[0] + t[1] * t[1]) + t[2] * t[2]) > 0) {
                    std::cout << "x, y: " << x << ", " << y << std::endl;
                    show("true coordinates: ", true_coordinates.data(), 3);
                    show("camera coordinates1: ", camera_coordinates1.data(), 3);
                    show("camera coordinates2: ", camera_coordinates2.data(), 3);
                    show("calculated coordinates: ", world_coordinates.data(), 3);
                    show("image coordinates1: ", image_coordinates1.data(), 2);
                    show("image coordinates2: ", image_coordinates2.data(), 2);
                    std::cout << "---------------------------------------------" << std::endl;
                }
            }
    }
    /** reference camera.cc:340-342 */
    void setPose(const int frameNr, const double k, const rsdsfm::lite::Vector3d& linear_velocity, const rsdsfm::lite::Vector3d& angular_velocity) {
        frames_[(size_t)frameNr - 1].setRelativePose(linear_velocity, angular_velocity, k);
    }
    void setImage(const int frameNr, const rsdsfm::ImageBGR& image) { frames_[(size_t)frameNr - 1].setImage(image); }
    /** reference camera.cc:353-361 */
    void backProject(const int frameNr) { frames_[(size_t)frameNr - 1].backProject(); }
    void backProjectGs(const int frameNr) { frames_[(size_t)frameNr - 1].backProjectGs(); }
    /** reference camera.cc:753-774 */
    rsdsfm::ImageBGR interpolateCrackyImage(const rsdsfm::ImageBGR& image_in, const unsigned offset) {
        rsdsfm::ImageBGR image_out(image_in.rows(), image_in.cols());
        rsdsfm::check(rsdsfm_interpolate_cracky(rsdsfm::default_context(), image_in.data(), image_in.rows(), image_in.cols(), (int32_t)offset,
                                                image_out.data()),
                      "rsdsfm_interpolate_cracky");
        return image_out;
    }
    /** reference camera.cc:209-249: ground-truth flow from frame frameNr1 to frame frameNr2 (unprojection maps of frame 1,
     *  relative scanline poses of frame 2) */
    rsdsfm::FlowImage calculateTrueFlow(const int frameNr1, const int frameNr2) {
        return frames_[(size_t)frameNr2 - 1].trueFlowFrom(frames_[(size_t)frameNr1 - 1]);
    }
    /** reference camera.cc:594-691 */
    double meanReprojectionError(const int frameNr) { return frames_[(size_t)frameNr - 1].reprojectionError(10.0, nullptr).mean_error; }
    /** reference camera.cc:503-591: rows x cols bytes, row-major (cv::Mat CV_8U) */
    std::vector<unsigned char> createErrorImage(const int frameNr, const double max_norm) {
        std::vector<unsigned char> img;
        frames_[(size_t)frameNr - 1].reprojectionError(max_norm, &img);
        return img;
    }
    void setGamma(const double gamma) {
        for (auto& f : frames_) f.setGamma(gamma);
    }

private:
    rsdsfm::lite::Matrix3d K_;
    std::vector<RsFrame> frames_;
};

#endif
