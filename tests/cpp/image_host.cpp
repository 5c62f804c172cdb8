// image_host.cpp -- the host-side image helpers of the reference's Camera in the C++ mirror (camera.cc:694-750, :777-840 and the
// `abs(a - b)` of main.cc:542-547): no GPU work.  Reads two raw BGR images (rows x cols x 3 bytes), writes
//   <out>.shift   shiftChannelBGR(a, 2, 0.5, 0.5)
//   <out>.overlay createOverlayImage(shiftChannelBGR(a, 1, 1, 1), shiftChannelBGR(absDiff(a, b), 2, 0.5, 0.5))   (main.cc:548-549)
//   <out>.cracky  the crack interpolation of `a` written with isBlackPixel / isColorfulArea / interpolateAreaColor on the host
//                 (camera.cc:753-774), offset 1
//   <out>.warp    reconstructImageFromFlow of `a` along the flow ((7u + 3v) % 11 - 5) / 2, 3 ((5u + 2v) % 7 - 3) / 4
// for the Python test to compare with the package's numpy versions and the oracle.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../rs-aware-differential-sfm_amd/host/camera.h"

static bool read_raw(const char* path, rsdsfm::ImageBGR& img) {
    FILE* f = std::fopen(path, "rb");
    const size_t n = (size_t)img.rows() * img.cols() * 3;
    const bool ok = f && std::fread(img.data(), 1, n, f) == n;
    if (f) std::fclose(f);
    return ok;
}
static bool write_raw(const std::string& path, const rsdsfm::ImageBGR& img) {
    FILE* f = std::fopen(path.c_str(), "wb");
    const size_t n = (size_t)img.rows() * img.cols() * 3;
    const bool ok = f && std::fwrite(img.data(), 1, n, f) == n;
    if (f) std::fclose(f);
    return ok;
}

int main(int argc, char** argv) {
    if (argc != 6) {
        std::fprintf(stderr, "usage: image_host a.raw b.raw rows cols out_prefix\n");
        return 2;
    }
    const int rows = std::atoi(argv[3]), cols = std::atoi(argv[4]);
    rsdsfm::ImageBGR a(rows, cols), b(rows, cols);
    if (!read_raw(argv[1], a) || !read_raw(argv[2], b)) return 3;
    const std::string out = argv[5];
    if (!write_raw(out + ".shift", Camera::shiftChannelBGR(a, 2, 0.5, 0.5))) return 4;
    const rsdsfm::ImageBGR overlay =
        Camera::createOverlayImage(Camera::shiftChannelBGR(a, 1, 1, 1), Camera::shiftChannelBGR(Camera::absDiff(a, b), 2, 0.5, 0.5));
    if (!write_raw(out + ".overlay", overlay)) return 4;
    rsdsfm::ImageBGR cracky = a.clone();
    const unsigned offset = 1, black_threshold = 15;
    for (unsigned row = offset; row + offset < (unsigned)rows; ++row)
        for (unsigned col = offset; col + offset < (unsigned)cols; ++col) {
            const unsigned char p[3] = {a.at((int)row, (int)col, 0), a.at((int)row, (int)col, 1), a.at((int)row, (int)col, 2)};
            if (Camera::isBlackPixel(p, black_threshold) && Camera::isColorfulArea(a, row, col, offset)) {
                unsigned char avg[3];
                Camera::interpolateAreaColor(a, row, col, offset, avg);
                for (int ch = 0; ch < 3; ++ch) cracky.at((int)row, (int)col, ch) = avg[ch];
            }
        }
    if (!write_raw(out + ".cracky", cracky)) return 4;
    Camera camera;
    camera.setIntrinsics("galaxy_vga");
    camera.addFrameReal(a);  // camera.cc:39-46
    if (camera.getFrame(1).getRsImage().at(rows / 2, cols / 2, 1) != a.at(rows / 2, cols / 2, 1)) return 5;
    // reconstructImageFromFlow (camera.cc:842-865) along a flow of exact quarter-pixel values (many collisions, half-pixel ties)
    rsdsfm::FlowImage flow(rows, cols);
    for (int v = 0; v < rows; ++v)
        for (int u = 0; u < cols; ++u) {
            flow.data()[((size_t)v * cols + u) * 2] = ((u * 7 + v * 3) % 11 - 5) * 0.5;
            flow.data()[((size_t)v * cols + u) * 2 + 1] = ((u * 5 + v * 2) % 7 - 3) * 0.75;
        }
    if (!write_raw(out + ".warp", camera.reconstructImageFromFlow(flow))) return 4;
    std::printf("ok\n");
    return 0;
}
