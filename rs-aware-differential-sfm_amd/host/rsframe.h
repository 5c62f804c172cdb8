// rsframe.h -- the part of the reference's RsFrame (src/rsframe.h) that consumes the solver's output: frame size,
// gamma, intrinsics, depth map and the per-scanline relative pose table (setRelativePose, rsframe.cc:771-800, computed
// by the pose_table HIP kernel), plus the rectifier that consumes them: backProject / backProjectGs
// (rsframe.cc:803-878) on a plain BGR byte image (rsdsfm::ImageBGR in place of cv::Mat).  File / OpenCV / CSV
// members of the reference class are out of scope.
#ifndef RSDSFM_HOST_RSFRAME_H
#define RSDSFM_HOST_RSFRAME_H

#include <cmath>
#include <vector>

#include "minimal.h"
#include "scanline.h"

class RsFrame {
public:
    RsFrame() : rows_(0), cols_(0), gamma_(1.0) {}
    RsFrame(int rows, int cols) : rows_(rows), cols_(cols), gamma_(1.0), scanlines_((size_t)rows) {}
    void setIntrinsics(const rsdsfm::lite::Matrix3d& intrinsics) { K_ = intrinsics; }
    void setGamma(const double gamma) { gamma_ = gamma; }
    int getRows() const { return rows_; }
    int getCols() const { return cols_; }
    void setDepthMap(const rsdsfm::lite::MatrixXd& depth_map) { depth_map_ = depth_map; }
    rsdsfm::lite::MatrixXd getDepthMap() { return depth_map_; }
    const Scanline& getScanline(int i) const { return scanlines_[(size_t)i]; }
    void setImage(const rsdsfm::ImageBGR& image) { image_ = image; }
    rsdsfm::ImageBGR getRsImage() { return image_; }
    rsdsfm::ImageBGR getGsImage() { return gs_image_; }
    rsdsfm::ImageXYZf get3dCoordinates() { return coordinates_3d_; }

    Scanline& scanline(int i) { return scanlines_[(size_t)i]; }  // absolute poses are set per scanline (reference: setPoses from CSV)

    /** reference rsframe.cc:416-436: ground-truth depth of every RS pixel from the unprojection maps and the ABSOLUTE poses */
    rsdsfm::lite::MatrixXd getGroundtruthDepthMap() const {
        rsdsfm::lite::MatrixXd z = rsdsfm::lite::MatrixXd::Zero(rows_, cols_);
        for (int y = 0; y < rows_; ++y) {
            const rsdsfm::lite::Matrix3d& R = scanlines_[(size_t)y].getRotation();
            const rsdsfm::lite::Vector3d& t = scanlines_[(size_t)y].getTranslation();
            for (int x = 0; x < cols_; ++x) {
                const double X = unprojection_map_x_(y, x), Y = unprojection_map_y_(y, x), Z = unprojection_map_z_(y, x);
                if (std::sqrt((X * X + Y * Y) + Z * Z) > 0) z(y, x) = ((R(2, 0) * X + R(2, 1) * Y) + R(2, 2) * Z) + t(2) * 1.0;
            }
        }
        return z;
    }
    /** reference rsframe.cc:951-967: first scanline becomes the origin (translation subtracted, rotation left-multiplied by
     *  the inverse of the first orientation; 3x3 inverse by cofactors like Eigen's fixed-size inverse) */
    void relocatePose() {
        const rsdsfm::lite::Vector3d p0 = scanlines_[0].getTranslation();
        const rsdsfm::lite::Matrix3d A = scanlines_[0].getRotation();
        rsdsfm::lite::Matrix3d inv;
        const double c00 = A(1, 1) * A(2, 2) - A(1, 2) * A(2, 1), c10 = A(1, 2) * A(2, 0) - A(1, 0) * A(2, 2), c20 = A(1, 0) * A(2, 1) - A(1, 1) * A(2, 0);
        const double invdet = 1.0 / ((A(0, 0) * c00 + A(0, 1) * c10) + A(0, 2) * c20);
        inv(0, 0) = c00 * invdet, inv(1, 0) = c10 * invdet, inv(2, 0) = c20 * invdet;
        inv(0, 1) = (A(0, 2) * A(2, 1) - A(0, 1) * A(2, 2)) * invdet, inv(1, 1) = (A(0, 0) * A(2, 2) - A(0, 2) * A(2, 0)) * invdet;
        inv(2, 1) = (A(2, 0) * A(0, 1) - A(0, 0) * A(2, 1)) * invdet;
        inv(0, 2) = (A(0, 1) * A(1, 2) - A(0, 2) * A(1, 1)) * invdet, inv(1, 2) = (A(1, 0) * A(0, 2) - A(0, 0) * A(1, 2)) * invdet;
        inv(2, 2) = (A(0, 0) * A(1, 1) - A(1, 0) * A(0, 1)) * invdet;
        for (int i = 1; i < rows_; ++i) {
            Scanline& sl = scanlines_[(size_t)i];
            const rsdsfm::lite::Vector3d ti = sl.getTranslation();
            sl.setTranslation(rsdsfm::lite::Vector3d(ti(0) - p0(0), ti(1) - p0(1), ti(2) - p0(2)));
            const rsdsfm::lite::Matrix3d Ri = sl.getRotation();
            rsdsfm::lite::Matrix3d out;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) out(r, c) = (inv(r, 0) * Ri(0, c) + inv(r, 1) * Ri(1, c)) + inv(r, 2) * Ri(2, c);
            sl.setRotation(out);
        }
    }
    /** Camera::meanReprojectionError / createErrorImage (camera.cc:503-691) for this frame; error_image may be null */
    rsdsfm_reprojection_stats reprojectionError(double max_norm, std::vector<unsigned char>* error_image) const {
        RsFrame frame = *this;  // the reference works on a copy and relocates its poses (camera.cc:595, :612)
        const rsdsfm::lite::MatrixXd real_depth_map = frame.getGroundtruthDepthMap();
        frame.relocatePose();
        std::vector<double> R((size_t)rows_ * 9), t((size_t)rows_ * 3);
        for (int i = 0; i < rows_; ++i)
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) R[(size_t)i * 9 + (size_t)(r * 3 + c)] = frame.scanlines_[(size_t)i].getRotation()(r, c);
                t[(size_t)i * 3 + (size_t)r] = frame.scanlines_[(size_t)i].getTranslation()(r);
            }
        rsdsfm_reprojection_stats st;
        if (error_image) error_image->assign((size_t)rows_ * (size_t)cols_, 0);
        rsdsfm::check(rsdsfm_reprojection_error(rsdsfm::default_context(), coordinates_3d_.data(), real_depth_map.data(), depth_map_.data(), R.data(),
                                                t.data(), K_(0, 0), K_(1, 1), K_(0, 2), K_(1, 2), rows_, cols_, max_norm, &st,
                                                error_image ? error_image->data() : nullptr),
                      "rsdsfm_reprojection_error");
        return st;
    }

    /** reference rsframe.h:74 takes three CSV paths (file formats are out of scope); this overload takes the loaded maps */
    bool setUnprojectionMapRs(const rsdsfm::lite::MatrixXd& x, const rsdsfm::lite::MatrixXd& y, const rsdsfm::lite::MatrixXd& z) {
        unprojection_map_x_ = x, unprojection_map_y_ = y, unprojection_map_z_ = z;
        return true;
    }
    /** reference rsframe.cc:617-625 */
    rsdsfm::lite::Vector3d getUnprojectedWorldCoordinates(int x, int y) const {
        return rsdsfm::lite::Vector3d(unprojection_map_x_(y, x), unprojection_map_y_(y, x), unprojection_map_z_(y, x));
    }
    rsdsfm::lite::Vector3d getUnprojectedWorldCoordinates(const rsdsfm::lite::Vector2d& point) const {
        return getUnprojectedWorldCoordinates((int)point.x(), (int)point.y());  // Eigen's coeff(double, double) truncates
    }
    unsigned long getNrScannlines() const { return (unsigned long)scanlines_.size(); }
    void setGsImage(const rsdsfm::ImageBGR& image) { gs_image_ = image; }
    void setDepthMapGs(const rsdsfm::lite::MatrixXd& depth_map_gs) { gs_depth_map_ = depth_map_gs; }
    bool setUnprojectionMapGs(const rsdsfm::lite::MatrixXd& x, const rsdsfm::lite::MatrixXd& y, const rsdsfm::lite::MatrixXd& z) {
        gs_unprojection_map_x_ = x, gs_unprojection_map_y_ = y, gs_unprojection_map_z_ = z;
        return true;
    }

    // ---- per-point geometry of the reference class (host arithmetic, the reference's operation order) ----
    /** reference rsframe.cc:629-642; quirk Q5: the y coordinate is scaled by f_x unless q5_mode() == RSDSFM_Q5_FIXED */
    rsdsfm::lite::Vector2d spaceToPlane(const rsdsfm::lite::Vector3d& Point) const {
        const double fyp = q5_mode() == RSDSFM_Q5_FIXED ? K_(1, 1) : K_(0, 0);
        return rsdsfm::lite::Vector2d(Point.x() / Point.z() * K_(0, 0) + K_(0, 2), Point.y() / Point.z() * fyp + K_(1, 2));
    }
    /** reference rsframe.cc:646-665: z_value == 0 (the default) takes the depth from the RS depth map */
    rsdsfm::lite::Vector3d planeToSpace(const rsdsfm::lite::Vector2d& point, double z_value = 0) const {
        const double px = (point.x() - K_(0, 2)) * 1.0 / K_(0, 0), py = (point.y() - K_(1, 2)) * 1.0 / K_(1, 1);
        if (z_value == 0) z_value = depth_map_((long)int(point.y()), (long)int(point.x()));
        return rsdsfm::lite::Vector3d(z_value * px, z_value * py, z_value * 1.0);
    }
    /** reference rsframe.cc:668-684 */
    rsdsfm::lite::Vector2i coordinateToPixel(const rsdsfm::lite::Vector2d point) const {
        return rsdsfm::lite::Vector2i((int)std::floor(point.x() + 0.5), (int)std::floor(point.y() + 0.5));
    }
    rsdsfm::lite::Vector2d pixelToCoordinate(const rsdsfm::lite::Vector2i pixel) const {
        return rsdsfm::lite::Vector2d(1.0 * pixel.x(), 1.0 * pixel.y());
    }
    /** reference rsframe.cc:688-709: [R t; 0 1] * (Point, 1), Eigen's 4x4 product evaluated left to right */
    rsdsfm::lite::Vector3d worldToCameraFrame(const rsdsfm::lite::Vector3d& Point, const int scanlineNr, bool useRelative = true) const {
        const Scanline& sl = scanlines_[(size_t)scanlineNr];
        const rsdsfm::lite::Matrix3d& R = useRelative ? sl.getRelativeRotation() : sl.getRotation();
        const rsdsfm::lite::Vector3d& t = useRelative ? sl.getRelativeTranslation() : sl.getTranslation();
        rsdsfm::lite::Vector3d out;
        for (int r = 0; r < 3; ++r) out(r) = ((R(r, 0) * Point.x() + R(r, 1) * Point.y()) + R(r, 2) * Point.z()) + t(r) * 1.0;
        return out;
    }
    /** reference rsframe.cc:713-736: [R^T  -R^T t; 0 1] * (Point, 1) */
    rsdsfm::lite::Vector3d cameraToWorldFrame(const rsdsfm::lite::Vector3d& Point, const int scanlineNr, bool useRelative = true) const {
        const Scanline& sl = scanlines_[(size_t)scanlineNr];
        const rsdsfm::lite::Matrix3d& R = useRelative ? sl.getRelativeRotation() : sl.getRotation();
        const rsdsfm::lite::Vector3d& t = useRelative ? sl.getRelativeTranslation() : sl.getTranslation();
        rsdsfm::lite::Vector3d out;
        for (int r = 0; r < 3; ++r) {
            const double p3 = -((R(0, r) * t(0) + R(1, r) * t(1)) + R(2, r) * t(2));  // -R_transpose.row(r) * t
            out(r) = ((R(0, r) * Point.x() + R(1, r) * Point.y()) + R(2, r) * Point.z()) + p3 * 1.0;
        }
        return out;
    }
    /** reference rsframe.cc:587-613: RS depth map from the unprojection maps and the RELATIVE scanline poses (synthetic data) */
    void setSyntheticDepthMapRs() {
        depth_map_ = rsdsfm::lite::MatrixXd::Zero(rows_, cols_);
        for (int y = 0; y < (int)scanlines_.size(); ++y)
            for (int x = 0; x < cols_; ++x) {
                const rsdsfm::lite::Vector3d W = getUnprojectedWorldCoordinates(x, y);
                if (std::sqrt((W.x() * W.x() + W.y() * W.y()) + W.z() * W.z()) > 0) depth_map_(y, x) = worldToCameraFrame(W, y).z();
            }
    }
    /** reference rsframe.cc:565-584: GS depth map from the GS unprojection maps and the pose of scanline 0 */
    void setSyntheticDepthMapGs() {
        gs_depth_map_ = rsdsfm::lite::MatrixXd::Zero(rows_, cols_);
        for (int y = 0; y < (int)scanlines_.size(); ++y)
            for (int x = 0; x < cols_; ++x) {
                const rsdsfm::lite::Vector3d W(gs_unprojection_map_x_(y, x), gs_unprojection_map_y_(y, x), gs_unprojection_map_z_(y, x));
                if (std::sqrt((W.x() * W.x() + W.y() * W.y()) + W.z() * W.z()) > 0) gs_depth_map_(y, x) = worldToCameraFrame(W, 0).z();
            }
    }
    rsdsfm::lite::MatrixXd getDepthMapGs() const { return gs_depth_map_; }
    /** reference rsframe.cc:740-768 with the reference's signature */
    rsdsfm::lite::Vector2d calculateImageCoordinatesRsFrame(const rsdsfm::lite::Vector3d& Point) const {
        double x = 0, y = 0;
        calculateImageCoordinatesRsFrame(Point, x, y);
        return rsdsfm::lite::Vector2d(x, y);
    }
    /** reference rsframe.cc:740-768: image coordinates of a world point under the best-matching scanline pose */
    void calculateImageCoordinatesRsFrame(const rsdsfm::lite::Vector3d& Point, double& x_out, double& y_out) const {
        std::vector<double> R, t;
        poseTable(R, t);
        double flow[2] = {0, 0};
        const double wx = Point(0), wy = Point(1), wz = Point(2);
        // a 1 x 1 "frame 1" holding the point: flow = projection - (0, 0)
        rsdsfm::check(rsdsfm_true_flow(rsdsfm::default_context(), &wx, &wy, &wz, 1, 1, R.data(), t.data(), rows_, K_(0, 0), K_(1, 1), K_(0, 2),
                                       K_(1, 2), q5_mode(), flow, nullptr),
                      "rsdsfm_true_flow");
        x_out = flow[0], y_out = flow[1];
    }
    /** the search of Camera::calculateTrueFlow (camera.cc:209-249) with this frame as frame 2 */
    rsdsfm::FlowImage trueFlowFrom(const RsFrame& frame1) const {
        std::vector<double> R, t;
        poseTable(R, t);
        rsdsfm::FlowImage flow(frame1.rows_, frame1.cols_);
        rsdsfm::check(rsdsfm_true_flow(rsdsfm::default_context(), frame1.unprojection_map_x_.data(), frame1.unprojection_map_y_.data(),
                                       frame1.unprojection_map_z_.data(), frame1.rows_, frame1.cols_, R.data(), t.data(), rows_, K_(0, 0),
                                       K_(1, 1), K_(0, 2), K_(1, 2), q5_mode(), flow.data(), nullptr),
                      "rsdsfm_true_flow");
        return flow;
    }

    /** reference rsframe.cc:803-839: RS image -> 3-D -> GS image with the per-scanline relative poses */
    void backProject() { backProjectImpl(RSDSFM_BACKPROJECT_RS); }
    /** reference rsframe.cc:842-878: the same with the pose of the first scanline for every pixel */
    void backProjectGs() { backProjectImpl(RSDSFM_BACKPROJECT_GS); }
    /** reference quirk Q5 (spaceToPlane scales y by f_x, rsframe.cc:639): RSDSFM_Q5_COMPAT (default) or RSDSFM_Q5_FIXED */
    static int& q5_mode() {
        static int mode = RSDSFM_Q5_COMPAT;
        return mode;
    }

    /** reference rsframe.cc:881-949 (not called by the reference's drivers): the small-motion alternative to backProject -- every
     *  pixel (x, y >= 1) is moved against its model flow beta_1(y) * (A v / depth + B w), rounded to whole pixels, into gs_image_
     *  (pixels whose rounded flow is zero are NOT copied, as in the reference), and the 3-D coordinates are those of planeToSpace +
     *  cameraToWorldFrame.  Host loop (a diagnostic, not on the hot path).  Two guards the reference lacks: pixels without depth are
     *  skipped (the reference divides by zero), and a write is dropped when its TARGET (y - dy, x - dx) lies outside the image (the
     *  reference tests (y + dy, x + dx) and may write out of bounds). */
    void smallMotionWrapping(const rsdsfm::lite::Vector3d& linear_velocity, const rsdsfm::lite::Vector3d& angular_velocity, const double k) {
        const double f_x = K_(0, 0), f_y = K_(1, 1), c_x = K_(0, 2), c_y = K_(1, 2);
        rsdsfm::ImageBGR gs_image(rows_, cols_);
        rsdsfm::ImageXYZf coordinates_3d(rows_, cols_);
        for (int y = 1; y < rows_; ++y) {
            const double beta_1 = (gamma_ * y / rows_ + 0.5 * k * (gamma_ * gamma_ * y * y) / (rows_ * rows_)) * (2.0 / (2.0 + k));
            for (int x = 1; x < cols_; ++x) {
                const double depth = depth_map_(y, x);
                if (depth == 0) continue;
                const double u = (x - c_x) * 1.0 / f_x, v = (y - c_y) * 1.0 / f_y;
                const double inv_depth = 1.0 / depth;
                // A = -[-1 0 u; 0 -1 v], B = -[uv -(1+u^2) v; (1+v^2) -uv -u]
                const double av0 = (linear_velocity(0) - u * linear_velocity(2)) * inv_depth, av1 = (linear_velocity(1) - v * linear_velocity(2)) * inv_depth;
                const double bw0 = -(u * v * angular_velocity(0) - (1 + u * u) * angular_velocity(1) + v * angular_velocity(2));
                const double bw1 = -((1 + v * v) * angular_velocity(0) - u * v * angular_velocity(1) - u * angular_velocity(2));
                const double flow_x = beta_1 * (av0 + bw0) * (f_x * 1.0 / gamma_), flow_y = beta_1 * (av1 + bw1) * (f_y * 1.0 / gamma_);
                const int dx = (int)std::floor(flow_x + 0.5), dy = (int)std::floor(flow_y + 0.5);
                if ((dx != 0 || dy != 0) && y + dy < rows_ && y + dy >= 0 && x + dx < cols_ && x + dx >= 0 && y - dy < rows_ && y - dy >= 0 &&
                    x - dx < cols_ && x - dx >= 0)
                    for (int ch = 0; ch < 3; ++ch) gs_image.at(y - dy, x - dx, ch) = image_.at(y, x, ch);
                const rsdsfm::lite::Vector3d Point_world = cameraToWorldFrame(planeToSpace(rsdsfm::lite::Vector2d(x, y)), y);
                for (int ch = 0; ch < 3; ++ch) coordinates_3d.at(y, x, ch) = (float)Point_world(ch);
            }
        }
        coordinates_3d_ = coordinates_3d;
        gs_image_ = gs_image;
    }

    /** reference rsframe.cc:771-800 */
    void setRelativePose(const rsdsfm::lite::Vector3d& linear_velocity, const rsdsfm::lite::Vector3d& angular_velocity, const double k) {
        std::vector<double> R((size_t)rows_ * 9), t((size_t)rows_ * 3);
        rsdsfm::check(rsdsfm_pose_table(rsdsfm::default_context(), linear_velocity.data(), angular_velocity.data(), k, gamma_, rows_,
                                        R.data(), t.data()),
                      "rsdsfm_pose_table");
        for (int i = 0; i < rows_; ++i) {
            rsdsfm::lite::Matrix3d Ri;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) Ri(r, c) = R[(size_t)i * 9 + r * 3 + c];
            scanlines_[(size_t)i].setRelativeRotation(Ri);
            scanlines_[(size_t)i].setRelativeTranslation(rsdsfm::lite::Vector3d(t[(size_t)i * 3], t[(size_t)i * 3 + 1], t[(size_t)i * 3 + 2]));
        }
    }

private:
    void poseTable(std::vector<double>& R, std::vector<double>& t) const {
        R.resize((size_t)rows_ * 9), t.resize((size_t)rows_ * 3);
        for (int i = 0; i < rows_; ++i) {
            const Scanline& sl = scanlines_[(size_t)i];
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) R[(size_t)i * 9 + (size_t)(r * 3 + c)] = sl.getRelativeRotation()(r, c);
                t[(size_t)i * 3 + (size_t)r] = sl.getRelativeTranslation()(r);
            }
        }
    }
    void backProjectImpl(int mode) {
        std::vector<double> R, t;
        poseTable(R, t);
        gs_image_ = rsdsfm::ImageBGR(rows_, cols_);
        coordinates_3d_ = rsdsfm::ImageXYZf(rows_, cols_);
        rsdsfm::check(rsdsfm_back_project(rsdsfm::default_context(), image_.data(), depth_map_.data(), R.data(), t.data(), K_(0, 0), K_(1, 1),
                                          K_(0, 2), K_(1, 2), rows_, cols_, mode, q5_mode(), gs_image_.data(), coordinates_3d_.data()),
                      "rsdsfm_back_project");
    }

    int rows_, cols_;
    double gamma_;
    rsdsfm::lite::Matrix3d K_;
    rsdsfm::lite::MatrixXd depth_map_;
    std::vector<Scanline> scanlines_;
    rsdsfm::ImageBGR image_, gs_image_;
    rsdsfm::ImageXYZf coordinates_3d_;
    rsdsfm::lite::MatrixXd unprojection_map_x_, unprojection_map_y_, unprojection_map_z_;
    rsdsfm::lite::MatrixXd gs_depth_map_, gs_unprojection_map_x_, gs_unprojection_map_y_, gs_unprojection_map_z_;
};

#endif
