// camera.h -- the facade of the reference's Camera (src/camera.h) over the frames: intrinsics (incl. the five
// hard-coded phone calibrations, camera.cc:179-206), gamma, pose and depth map forwarding (camera.cc:335-371).
// plus the rectifier entry points backProject / backProjectGs (camera.cc:353-361) and interpolateCrackyImage
// (camera.cc:753-774).  Flow, CSV and visualisation members are out of scope.
#ifndef RSDSFM_HOST_CAMERA_H
#define RSDSFM_HOST_CAMERA_H

#include <cmath>
#include <cstdlib>
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
    /** reference camera.cc:39-46: a frame of real data (RS image + intrinsics only) */
    void addFrameReal(const rsdsfm::ImageBGR& rs_image) {
        addFrame(rs_image.rows(), rs_image.cols());
        frames_.back().setImage(rs_image);
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
    /** reference camera.cc:494-497 */
    void smallMotionWrapping(const int frameNr, const rsdsfm::lite::Vector3d& linear_velocity, const rsdsfm::lite::Vector3d& angular_velocity, const double k) {
        frames_[(size_t)frameNr - 1].smallMotionWrapping(linear_velocity, angular_velocity, k);
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
    /** reference camera.cc:694-750: the per-pixel predicates of interpolateCrackyImage on the host (the GPU kernel behind
     *  interpolateCrackyImage evaluates the same rules).  A pixel is "black" when the Euclidean norm of its BGR triple is <= threshold;
     *  the four neighbours at distance `offset` must lie inside the image (the reference does not check either). */
    static bool isBlackPixel(const unsigned char point[3], const unsigned threshold) {
        const double n2 = (double)point[0] * point[0] + (double)point[1] * point[1] + (double)point[2] * point[2];
        return std::sqrt(n2) <= (double)threshold;
    }
    static bool isColorfulArea(const rsdsfm::ImageBGR& image_in, const unsigned row, const unsigned col, const unsigned offset) {
        const unsigned black_threshold = 15;
        const int r = (int)row, c = (int)col, o = (int)offset;
        return !isBlackPixel(px(image_in, r - o, c), black_threshold) || !isBlackPixel(px(image_in, r + o, c), black_threshold) ||
               !isBlackPixel(px(image_in, r, c - o), black_threshold) || !isBlackPixel(px(image_in, r, c + o), black_threshold);
    }
    /** average colour of the non-black neighbours above / below / left / right (camera.cc:712-750); out = (0, 0, 0) if all are black */
    static void interpolateAreaColor(const rsdsfm::ImageBGR& image_in, const unsigned row, const unsigned col, const unsigned offset, unsigned char out[3]) {
        const unsigned black_threshold = 15;
        const int r = (int)row, c = (int)col, o = (int)offset;
        const unsigned char* nb[4] = {px(image_in, r - o, c), px(image_in, r + o, c), px(image_in, r, c - o), px(image_in, r, c + o)};
        double sum[3] = {0, 0, 0};
        unsigned count = 0;
        for (int k = 0; k < 4; ++k)
            if (!isBlackPixel(nb[k], black_threshold)) {
                for (int ch = 0; ch < 3; ++ch) sum[ch] += (double)nb[k][ch];
                count++;
            }
        for (int ch = 0; ch < 3; ++ch) out[ch] = count > 0 ? saturate_u8((1 / static_cast<double>(count)) * sum[ch]) : 0;
    }
    /** reference camera.cc:777-815: per-channel gain, clamped to [0, 255], truncated (static_cast<uint8_t>) */
    static rsdsfm::ImageBGR shiftChannelBGR(const rsdsfm::ImageBGR& image_in, double shift_blue, double shift_green, double shift_red) {
        rsdsfm::ImageBGR shifted(image_in.rows(), image_in.cols());
        const double gain[3] = {shift_blue, shift_green, shift_red};
        for (int row = 0; row < image_in.rows(); row++)
            for (int col = 0; col < image_in.cols(); col++)
                for (int ch = 0; ch < 3; ++ch) {
                    double v = image_in.at(row, col, ch) * gain[ch];
                    if (v > 255) v = 255; else if (v < 0) v = 0;
                    shifted.at(row, col, ch) = static_cast<unsigned char>(v);
                }
        return shifted;
    }
    /** reference camera.cc:818-840: where shift_image is not black, blend the two pixels by their norms (cv::Vec3b arithmetic:
     *  every product and the sum are rounded / saturated to 8 bits); elsewhere the original pixel */
    static rsdsfm::ImageBGR createOverlayImage(const rsdsfm::ImageBGR& original_image, const rsdsfm::ImageBGR& shift_image) {
        const unsigned black_threshold = 15;
        rsdsfm::ImageBGR image_out(original_image.rows(), original_image.cols());
        for (int row = 0; row < original_image.rows(); row++)
            for (int col = 0; col < original_image.cols(); col++) {
                const unsigned char* ps = px(shift_image, row, col);
                const unsigned char* po = px(original_image, row, col);
                const double ns = norm3(ps), no = norm3(po);
                for (int ch = 0; ch < 3; ++ch) {
                    if (ns > black_threshold) {
                        const double multiplier = no / (no + ns);
                        const int a = saturate_u8(multiplier * po[ch]), b = saturate_u8((1 - multiplier) * ps[ch]);
                        image_out.at(row, col, ch) = (unsigned char)(a + b > 255 ? 255 : a + b);
                    } else {
                        image_out.at(row, col, ch) = po[ch];
                    }
                }
            }
        return image_out;
    }
    /** reference camera.cc:842-865: frame 1's RS image pushed forward along a flow field rounded to whole pixels; scan order
     *  columns outer / rows inner, the last writer wins, targets with x <= 0 or y <= 0 are dropped (the reference's strict test) */
    rsdsfm::ImageBGR reconstructImageFromFlow(const rsdsfm::FlowImage& flow_image) {
        const rsdsfm::ImageBGR original_image = frames_[0].getRsImage();
        rsdsfm::ImageBGR reconstructed_image(original_image.rows(), original_image.cols());
        for (int u = 0; u < flow_image.cols(); ++u)
            for (int v = 0; v < flow_image.rows(); ++v) {
                const int new_x = u + (int)std::floor(flow_image.x(v, u) + 0.5), new_y = v + (int)std::floor(flow_image.y(v, u) + 0.5);
                if (new_x > 0 && new_x < flow_image.cols() && new_y > 0 && new_y < flow_image.rows())
                    for (int ch = 0; ch < 3; ++ch) reconstructed_image.at(new_y, new_x, ch) = original_image.at(v, u, ch);
            }
        return reconstructed_image;
    }
    /** |a - b| per byte: what the reference's `abs(a - b)` on 8-bit cv::Mat evaluates to (OpenCV folds it into absdiff; main.cc:542-547) */
    static rsdsfm::ImageBGR absDiff(const rsdsfm::ImageBGR& a, const rsdsfm::ImageBGR& b) {
        rsdsfm::ImageBGR out(a.rows(), a.cols());
        const size_t n = (size_t)a.rows() * (size_t)a.cols() * 3;
        for (size_t i = 0; i < n; ++i) out.data()[i] = (unsigned char)std::abs((int)a.data()[i] - (int)b.data()[i]);
        return out;
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
    static const unsigned char* px(const rsdsfm::ImageBGR& img, int row, int col) { return img.data() + ((size_t)row * (size_t)img.cols() + (size_t)col) * 3; }
    static double norm3(const unsigned char p[3]) { return std::sqrt((double)p[0] * p[0] + (double)p[1] * p[1] + (double)p[2] * p[2]); }
    /** cv::saturate_cast<uchar>(double): round half to even (cvRound), then clamp */
    static unsigned char saturate_u8(double v) {
        const long iv = std::lrint(v);
        return (unsigned char)(iv < 0 ? 0 : iv > 255 ? 255 : iv);
    }
    rsdsfm::lite::Matrix3d K_;
    std::vector<RsFrame> frames_;
};

#endif
