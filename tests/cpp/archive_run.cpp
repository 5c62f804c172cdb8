// archive_run.cpp -- the reading side of the reference's setupCameraSynthetic() (main.cc:613-671) written against the C++
// mirror's file-format layer (host/formats.h): loads A.csv, the scanline poses, the unprojection maps and the PNG frames of
// an example archive, then writes the products the reference writes (PLY point cloud, PNG) and prints checksums.
// No GPU work: usage  archive_run <images_dir/> <out_dir/>
#define RSDSFM_WITH_PNG
#include <cstdio>

#include "../../rs-aware-differential-sfm_amd/host/formats.h"

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <images_dir/> <out_dir/>\n", argv[0]);
        return 2;
    }
    const std::string prefix = argv[1], out = argv[2];
    Camera camera;
    if (!rsdsfm::loadIntrinsicsFromFile(camera, prefix + "A.csv")) return 3;
    double sums[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    unsigned long long img_sums[2] = {0, 0};
    int rows = 0, cols = 0;
    for (int n = 1; n <= 2; ++n) {
        const std::string p = prefix + std::to_string(n);
        rsdsfm::ImageBGR rs;
        if (!rsdsfm::imread_png(p + "_rs.png", rs)) return 4;
        rows = rs.rows(), cols = rs.cols();
        camera.addFrame(rows, cols);
        camera.setImage(n, rs);
        RsFrame& f = camera.frame(n);
        if (!rsdsfm::setPoses(f, p + "_rs_t.csv", p + "_rs_r.csv")) return 5;
        if (!rsdsfm::setUnprojectionMapRs(f, p + "_rs_unproject_x.csv", p + "_rs_unproject_y.csv", p + "_rs_unproject_z.csv")) return 6;
        for (int i = 0; i < rows; ++i) {
            for (int r = 0; r < 3; ++r) {
                sums[n - 1][0] += f.getScanline(i).getTranslation()(r) * (i + 1);
                for (int c = 0; c < 3; ++c) sums[n - 1][1] += f.getScanline(i).getRotation()(r, c) * (r * 3 + c + 1);
            }
            for (int x = 0; x < cols; ++x) {
                const rsdsfm::lite::Vector3d W = f.getUnprojectedWorldCoordinates(x, i);
                sums[n - 1][2] += W(0) + 2 * W(1) + 3 * W(2);
            }
        }
        for (size_t i = 0; i < (size_t)rows * (size_t)cols * 3; ++i) img_sums[n - 1] += (unsigned long long)rs.data()[i] * (i % 253 + 1);
    }
    // a line-count mismatch must be rejected (rows + 1 scanlines expected)
    RsFrame wrong(rows + 1, cols);
    const bool rejected = !rsdsfm::setPoses(wrong, prefix + "1_rs_t.csv", prefix + "1_rs_r.csv");
    // products: the frame's own image written back as PNG; a PLY whose coordinates are the unprojection map (as floats)
    if (!rsdsfm::imwrite_png(out + "copy_rs.png", camera.frame(1).getRsImage(), 6)) return 7;
    rsdsfm::lite::Matrix3d K = camera.getIntrinsics();
    std::printf("{\"K\": [%.17g, %.17g, %.17g, %.17g], \"rows\": %d, \"cols\": %d, \"t_sum\": [%.17g, %.17g], \"R_sum\": [%.17g, %.17g], "
                "\"w_sum\": [%.17g, %.17g], \"img_sum\": [%llu, %llu], \"rejected\": %d}\n",
                K(0, 0), K(1, 1), K(0, 2), K(1, 2), rows, cols, sums[0][0], sums[1][0], sums[0][1], sums[1][1], sums[0][2], sums[1][2], img_sums[0],
                img_sums[1], rejected ? 1 : 0);
    return 0;
}
